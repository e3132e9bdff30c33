"""Diagnostic: vjf_filter_seq schedule variants against the one-stream order on a golden fixture (bitwise)."""
import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vjf_amd
from tests import goldenio as gio
from tests.helpers import load_fixture_state
from tests.test_gpu_parity import _model_for
name = sys.argv[1] if len(sys.argv) > 1 else "g5_medium_gaussian_f32"
dist_mode = len(sys.argv) > 2 and sys.argv[2] == "dist"
z, info, _ = gio.traj_case(name)
def run(env, overlap=True):
    for k in ("VJF_NO_PERSISTENT", "VJF_K1_GATE", "VJF_SELF_PREP", "VJF_FORCE_DIST"): os.environ.pop(k, None)
    os.environ.update(env)
    m = _model_for(vjf_amd, info)
    load_fixture_state(m, z, "s0")
    if not overlap: m.set_overlap(False)
    mu, lv, loss = m.filter_sequence(torch.tensor(z["y"]), None, None, eps=torch.tensor(z["eps"]))
    torch.cuda.synchronize()
    return mu.cpu(), lv.cpu(), loss.cpu(), m._blob.cpu().clone(), m.get_state(), m.status()
b = run({}, overlap=False)
if dist_mode:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    a = run({"VJF_FORCE_DIST": "1"})
    a2 = run({"VJF_FORCE_DIST": "1", "VJF_NO_PERSISTENT": "1"})
    dist.destroy_process_group()
else:
    a = run({})
    a2 = run({"VJF_NO_PERSISTENT": "1"})
for nm, a in (("persistent", a), ("per-step", a2)):
    print(nm, "status", a[5], b[5])
    for t in range(info["T"]):
        print(t, "mu", (a[0][t]-b[0][t]).abs().max().item(), "loss", (a[2][t]-b[2][t]).abs().max().item())
    for k in a[4]:
        x, y = np.asarray(a[4][k], np.float64), np.asarray(b[4][k], np.float64)
        if not np.array_equal(x, y): print(" ", k, np.abs(x-y).max(), np.abs(y).max())
