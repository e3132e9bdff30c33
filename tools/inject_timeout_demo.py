"""Diagnostic: a long sequence whose persistent Cholesky loop reports a wait time-out at step k (VJF_DEBUG_INJECT=k): the guard must
re-run it with per-step launches -- same results as an undisturbed run -- and the lost attempt must drain quickly."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vjf_amd
torch.manual_seed(0)
B, dz, dy, n, T = 4096, 10, 50, 200, 400
def run(env):
    for k in ("VJF_DEBUG_INJECT",): os.environ.pop(k, None)
    os.environ.update(env)
    torch.manual_seed(0)
    m = vjf_amd.VJF.make_model(dy, dz, 0, n, [128], likelihood="gaussian", noise="device")
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.randn(T, B, dy, device="cuda", generator=g); eps = torch.randn(T, 2, B, dz, device="cuda", generator=g)
    m.filter_sequence(y[:20], eps=eps[:20]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    mu, lv, loss = m.filter_sequence(y[20:], eps=eps[20:]); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return mu.cpu(), loss.cpu(), m._blob.cpu().clone(), m.status(), dt
a = run({})
b = run({"VJF_DEBUG_INJECT": "150", "VJF_VERBOSE": "1"})
print(f"clean: {a[4]*1e3:.1f} ms status {a[3]};  injected at step 150: {b[4]*1e3:.1f} ms status {b[3]}")
print("outputs equal:", torch.equal(a[0], b[0]), torch.equal(a[1], b[1]), " state equal:", torch.equal(a[2], b[2]))
