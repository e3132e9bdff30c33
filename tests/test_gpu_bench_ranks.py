"""`bench.py --gpus N` for N > 1, rehearsed on the one-GPU box: the bench starts its own ranks (a fresh torch.distributed.run child,
before this process's child has touched the GPU), both ranks drive cuda:0, torch.distributed runs on gloo, the sums over ranks are
the caller-side route (vjf_filter_local / all-reduce / vjf_filter_global -- RCCL cannot span two ranks of one device).  Checked: ONE
JSON line, the contract's fields, the whole-job value = trials of all ranks, and the ELBO of the first timed steps against the
fp64 oracle on the trials of BOTH ranks (every sum over trials is a sum over ranks)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env=None, timeout=600):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=timeout, cwd=ROOT)
    out = r.stdout.decode("utf-8", "replace")
    lines = [l for l in out.splitlines() if l.strip()]
    assert r.returncode == 0, f"bench.py exited {r.returncode}\n{out}\n{r.stderr.decode('utf-8', 'replace')[-4000:]}"
    assert len(lines) == 1, f"expected ONE line on stdout, got {len(lines)}:\n{out}"
    return json.loads(lines[0])


def test_bench_starts_its_own_two_ranks_on_one_gpu():
    j = _run(["--gpus", "2", "--steps", "6", "--warmup", "3", "--repeats", "2", "--no-cpu-baseline"],
             {"VJF_BENCH_BACKEND": "gloo"})
    assert j["metric"] == "trial-timesteps/sec" and j["unit"] == "trial-timesteps/s"
    assert j["n_gpus"] == 2 and j["steps"] == 6 and j["warmup"] == 3 and j["scaling"] == "weak"
    assert j["config"]["global_batch"] == 2 * 4096 and j["config"]["trials_per_gpu"] == 4096
    assert abs(j["value"] - 2 * 4096 * 6 / (j["ms_per_step"] * 6e-3)) < 1e-6 * j["value"]
    d = j["dist"]
    assert d["ranks_share_a_gpu"] is True and d["control_plane"] == "gloo" and d["sums_over_ranks"] == "caller"
    assert d["rccl_comm_ranks"] == [0, 0]
    assert "caller-side" in j["route"]
    e = j["elbo_check"]
    assert e is not None and e["ok"] and e["trials"] == 2 * 4096 and e["max_rel_err"] < e["rtol"]
    assert j["status_bits"] == 0
    assert "cpu_baseline" not in j                      # (rank 0 at N = 1 only)


def test_bench_single_gpu_line_has_the_contract_fields():
    j = _run(["--steps", "10", "--warmup", "3", "--repeats", "1", "--no-call-cost"])
    assert j["n_gpus"] == 1 and j["route"] == "one-launch" and j["dist"] is None
    assert j["roofline"]["bound"] == "mfma" and 0 < j["roofline"]["frac"] < 1
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0
    assert j["elbo_check"]["ok"]


def test_bench_falls_back_when_the_in_library_route_fails_in_its_warm_up():
    """The in-library RCCL route has never run with two real ranks on the builder's boxes; if a wait of its warm-up steps gives up
    (or the library returns an error) on ANY rank, every rank starts over with a fresh model on the caller-side sums and the line
    says so.  Injected here (rank 0 reports a failed warm-up once): the run still ends with a valid line whose ELBO matches."""
    j = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline"],
             {"VJF_BENCH_BACKEND": "gloo", "VJF_BENCH_FAKE_NATIVE_FAILURE": "1"})
    d = j["dist"]
    assert d["sums_over_ranks"].startswith("caller (") and "injected" in d["native_route_error"]
    assert j["n_gpus"] == 2 and j["elbo_check"]["ok"] and j["elbo_check"]["trials"] == 2 * 4096 and j["status_bits"] == 0


def test_bench_measures_the_caller_side_route_first_and_keeps_its_line():
    """N > 1 under somebody else's launcher: the caller-side route is measured first and its line kept; the in-library RCCL route
    follows under a watchdog.  Here (two ranks on ONE GPU, where RCCL cannot make its communicators) the in-library attempt fails
    one way or another -- an error of the communicator's creation, or no answer until the watchdog -- and the run still ends with
    exit code 0 and a valid line of the caller-side route that says what happened."""
    j = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline"],
             {"VJF_BENCH_BACKEND": "gloo", "VJF_BENCH_TEST_SAFE": "1", "VJF_BENCH_WARMUP_TIMEOUT": "40"})
    d = j["dist"]
    assert d["native_route_error"], d
    assert d["sums_over_ranks"].startswith("caller"), d
    assert j["n_gpus"] == 2 and j["elbo_check"]["ok"] and j["elbo_check"]["trials"] == 2 * 4096 and j["status_bits"] == 0
    assert abs(j["value"] - 2 * 4096 * 4 / (j["ms_per_step"] * 4e-3)) < 1e-6 * j["value"]


def test_bench_ends_with_the_kept_line_when_the_in_library_route_never_comes_back():
    j = _run(["--gpus", "2", "--steps", "4", "--warmup", "2", "--repeats", "1", "--no-cpu-baseline"],
             {"VJF_BENCH_BACKEND": "gloo", "VJF_BENCH_TEST_SAFE": "1", "VJF_BENCH_FAKE_NATIVE_HANG": "1", "VJF_BENCH_WARMUP_TIMEOUT": "25"})
    d = j["dist"]
    assert "did not finish in time" in d["native_route_error"], d
    assert d["sums_over_ranks"] == "caller" and j["elbo_check"]["ok"] and j["status_bits"] == 0
