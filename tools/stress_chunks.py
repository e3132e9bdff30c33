"""Stress: chunked sequences (several persistent-kernel launches back to back), many repetitions; prints any non-zero status."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vjf_amd
from tests import goldenio as gio
from tests.helpers import load_fixture_state
from tests.test_gpu_parity import _model_for
z, info, _ = gio.traj_case("g5_medium_gaussian_f32")
y, eps = torch.tensor(z["y"]), torch.tensor(z["eps"])
bad = 0
ref = None
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    for chunk in ("0", "3", "2"):
        os.environ["VJF_SEQ_CHUNK"] = chunk
        m = _model_for(vjf_amd, info)
        load_fixture_state(m, z, "s0")
        o = m.filter_sequence(y, None, None, eps=eps)
        st = m.status()
        blob = m._blob.clone()
        if ref is None: ref = (o, blob)
        same = all(torch.equal(a, b) for a, b in zip(ref[0], o)) and torch.equal(ref[1], blob)
        if st or not same:
            bad += 1
            first = [t for t in range(o[0].shape[0]) if not torch.equal(o[0][t], ref[0][0][t])]
            firstl = [t for t in range(o[2].shape[0]) if not torch.equal(o[2][t], ref[0][2][t])]
            print(f"rep {rep} chunk {chunk}: status 0x{st:x} same {same} first mu diff at t={first[:1]} first loss diff at t={firstl[:1]}", flush=True)
print("bad", bad)
