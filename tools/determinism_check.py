"""Diagnostic: two 3000-step sequences at config B from the same seed must agree bit for bit (state, posterior, losses)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vjf_amd
B, dz, dy, n, T = 4096, 10, 50, 200, 3000
def run():
    torch.manual_seed(0)
    m = vjf_amd.VJF.make_model(dy, dz, 0, n, [128], likelihood="gaussian", noise="device")
    g = torch.Generator(device="cuda").manual_seed(1)
    y = torch.randn(T, B, dy, device="cuda", generator=g); eps = torch.randn(T, 2, B, dz, device="cuda", generator=g)
    mu, lv, loss = m.filter_sequence(y, eps=eps)
    st = m.check_status()
    return m._blob.clone(), mu[-1].clone(), loss.clone(), st
a = run(); b = run()
print("status", a[3], b[3], "blob equal", torch.equal(a[0], b[0]), "mu equal", torch.equal(a[1], b[1]), "loss equal", torch.equal(a[2], b[2]), "finite", bool(torch.isfinite(a[2]).all()))
