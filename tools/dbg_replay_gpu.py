import os, sys, warnings; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vjf_oracle as orc
from tests.helpers import load_oracle_state
import vjf_amd as vjf
warnings.simplefilter("ignore")
B, dz, dy, n, T = 40, 3, 10, 16, 4
g = torch.Generator().manual_seed(31)
y = torch.poisson(torch.exp(0.3 * torch.randn(T, B, dy, generator=g)), generator=g)
eps = torch.randn(T, 2, B, dz, generator=g)
y[:, :, 0] = 2.0
def run(overlap):
    torch.manual_seed(30)
    m = vjf.VJF.make_model(dy, dz, 0, n, [8], likelihood="poisson", lr=1e-2)
    if not overlap: m.set_overlap(False)
    mu0, lv0, _ = m.filter_sequence(y[:1], eps=eps[:1])
    with torch.no_grad(): m.decoder.decode.bias[0] = -3e38
    s = load_oracle_state(m, np.float32)
    mu, lv, loss = m.filter_sequence(y[1:], qs=vjf.Gaussian(mu0[-1], lv0[-1]), eps=eps[1:])
    print("route", m.route(), "status", hex(m.status()))
    om, ol = mu0[-1].cpu().numpy(), lv0[-1].cpu().numpy()
    for t in range(1, T):
        o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
        om, ol = o.mu_t, o.lv_t
        print(" t", t, "device", loss[t-1].cpu().numpy(), "oracle", np.array([o.loss, o.recon, o.dyn, o.entropy], np.float32), "sigma dev/orc", float(m.transition.logvar), float(s.tr_logvar))
print("== one launch"); run(True)
print("== per step"); run(False)
os.environ["VJF_SEQ_CHUNK"] = "1"
print("== one launch, one step per launch"); run(True)
