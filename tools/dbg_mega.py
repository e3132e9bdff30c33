"""Diagnostic: one-launch route, sequence vs stepwise vs oracle at a given batch size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vjf_amd
from oracle import vjf_oracle as orc
from tests.helpers import load_oracle_state
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dz, dy, n = 10, 50, 200
def mk():
    torch.manual_seed(13)
    return vjf_amd.VJF.make_model(dy, dz, 0, n, [128], likelihood="gaussian", lr=1e-4)
g = torch.Generator().manual_seed(23)
y = torch.randn(T, B, dy, generator=g); eps = torch.randn(T, 2, B, dz, generator=g)
yd, ed = y.cuda(), eps.cuda()
m1, m2 = mk(), mk()
s = load_oracle_state(m1, np.float64)
mu, lv, ls = m1.filter_sequence(yd, eps=ed)
q = None
om = ol = None
for t in range(T):
    q, l, *c = m2.filter(yd[t], None, q, verbose=True, eps=(ed[t, 0], ed[t, 1]))
    o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
    om, ol = o.mu_t, o.lv_t
    print(f"t={t} seq loss {ls[t].cpu().numpy()}  step {[float(l)] + [float(x) for x in c]}  oracle {[o.loss, o.recon, o.dyn, o.entropy]}")
    print("   mu eq", torch.equal(mu[t], q.mean), " max|mu-oracle|", float(np.abs(mu[t].cpu().numpy() - o.mu_t).max()))
print("blob equal:", torch.equal(m1._blob, m2._blob), " status", m1.status(), m2.status())
d = (m1._blob - m2._blob).abs()
print("max blob diff", float(d.max()), "at", int(d.argmax()))
