"""The hand-offs between roles under perturbed timing (tools/chaos_handoffs.py): sequences on the one-launch route and on the
three-stream per-step route, run with the diagnostic build of the library in which workgroups are held at random in front of their
waits and signals, must give what the one-stream per-step kernels give.  Needs a real MI355X:  pytest -m gpu."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("acquire", ["0", "1"], ids=["sc1_loads_alone", "with_acquires"])
def test_sequences_with_workgroups_held_at_their_handoffs(acquire):
    """Both forms of the hand-offs: the default (sc1 loads of every handed-off byte behind the poll and the barrier) and
    VJF_HANDOFF_ACQUIRE=1 (an agent-scope acquire behind every wait as well)."""
    from vjf_amd import _build
    _build.build(chaos=True)                                   # (built by __graft_entry__.build(); compiled here if it is missing or stale)
    env = dict(os.environ, VJF_LIB="chaos", VJF_HANDOFF_ACQUIRE=acquire)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "chaos_handoffs.py"), "8", "2"], capture_output=True, text=True,
                       env=env, timeout=900)
    tail = "\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l)[-3000:]
    assert r.returncode == 0, tail
    assert "deviating sequences in all: 0" in r.stdout, tail


@pytest.mark.gpu
def test_random_configurations_of_every_plan_family():
    """tools/fuzz_parity.py with a fixed seed: 48 random configurations (one launch, per-step matrix-core trial kernel, multi-launch
    RLS, GEMM-per-layer trial path; Gaussian / Poisson; control input; 1-3 layers; ragged batches), three steps each through `filter`
    or `filter_sequence`, against the fp64 oracle at the tolerances of tests/test_gpu_parity.py (an RLS tensor of an ill-conditioned
    case may instead be as close to fp64 as the oracle run in the reference's fp32 is, within a factor of 3)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_parity.py"), "48", "0"], capture_output=True, text=True, timeout=900)
    tail = "\n".join(l for l in (r.stdout + r.stderr).splitlines() if "amdgpu.ids" not in l and " ok " not in l)[-3000:]
    assert "failures: 0 of 48" in r.stdout, tail


def _hard_cases():
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from tools.fuzz_parity import KNOWN_HARD
    return KNOWN_HARD


@pytest.mark.gpu
@pytest.mark.parametrize("case,desc", _hard_cases(), ids=lambda v: (f"case{v}" if isinstance(v, int) else f"B{v['B']}_n{v['n']}_{v['lik']}"))
def test_few_trials_against_many_features(case, desc):
    """The seven configurations that failed round 2's random sweeps (seeds 21, 22, 33 and an earlier draw order), pinned by their
    full description: B = 1 ... 9 trials against 214 ... 1289 features, the precision matrix grows 10^4-fold in two steps
    (cond 1e5).  Three steps each against the fp64 oracle; where the fixed tolerance is not met the device must be no further
    from fp64 than three times the oracle run in the reference's OWN arithmetic -- an fp32 LAPACK factorisation, as
    torch.linalg.cholesky does it (vjf/module.py:99); round 2 measured against numpy's, which works in double."""
    from tools.fuzz_parity import run_case
    route, notes = run_case(case, desc)
    assert route in ("per-step", "one-launch", "two-stream")
