"""Full-size sanity runs of the other BASELINE configs (parity-test cases, not bench lines):
2 filter steps vs the fp64 oracle, then a short timing.  Usage: python tools/config_check.py C|E|D1"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vjf_amd
from oracle import vjf_oracle as orc
from tests.helpers import load_oracle_state

CFGS = {"C": dict(B=4096, dz=10, dy=200, n=200, hidden=[128], lik="poisson"),
        "E": dict(B=4096, dz=64, dy=512, n=1000, hidden=[512, 512], lik="gaussian"),
        "D1": dict(B=32768, dz=10, dy=50, n=200, hidden=[128], lik="gaussian")}
c = CFGS[sys.argv[1]]
torch.manual_seed(0)
m = vjf_amd.VJF.make_model(c["dy"], c["dz"], 0, c["n"], c["hidden"], likelihood=c["lik"], noise="device")
if c["dz"] >= 32:
    r = float(np.sqrt(c["dz"]))
    m.transition.velocity.feature.centroid.uniform_(-r, r)
    m.transition.velocity.feature.logwidth.fill_(float(np.log(r)))
g = torch.Generator().manual_seed(1)
T = 12 if c['n'] > 224 else 62
if c["lik"] == "poisson":
    y = torch.poisson(torch.exp(0.5 * torch.randn(T, c["B"], c["dy"], generator=g) - 0.5), generator=g)
else:
    y = torch.randn(T, c["B"], c["dy"], generator=g)
eps = torch.randn(T, 2, c["B"], c["dz"], generator=g)
s = load_oracle_state(m, np.float64)
q, mu, lv = None, None, None
for t in range(2):
    q, loss, *comp = m.filter(y[t], None, q, verbose=True, eps=(eps[t, 0], eps[t, 1]))
    o = orc.filter_step(s, y[t].numpy(), None, mu, lv, eps[t, 0].numpy(), eps[t, 1].numpy())
    mu, lv = o.mu_t, o.lv_t
    err = np.abs(q.mean.cpu().numpy() - o.mu_t).max()
    print(f"step {t}: max|d mu| {err:.2e}  loss {float(loss):.6f} vs oracle {o.loss:.6f}  rel {abs(float(loss)-o.loss)/abs(o.loss):.1e}")
    assert err < 1e-4 and abs(float(loss) - o.loss) / abs(o.loss) < 1e-4
yd, ed = y.cuda(), eps.cuda()
W = 2 + (T - 2) // 6                                  # untimed: the first sequence call also creates the side streams
mu_w, lv_w, _ = m.filter_sequence(yd[2:W], qs=q, eps=ed[2:W])
q = vjf_amd.Gaussian(mu_w[-1], lv_w[-1])
torch.cuda.synchronize(); t0 = time.perf_counter()
m.filter_sequence(yd[W:], qs=q, eps=ed[W:])
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / (T - W)
print(f"config {sys.argv[1]}: {dt*1e6:.0f} us/step, {c['B']/dt/1e6:.2f} M trial-timesteps/s, status {m.status()}")
