#!/usr/bin/env python3
"""Generate golden vectors from the imported reference (catniplab/vjf).

Run ONLY in the build container, where /root/reference exists:

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

It imports the reference read-only via sys.path, feeds it seeded inputs and writes
inputs + expected outputs as .npz data files next to this script.  No reference
source, bytecode or pickled reference object is written: only numpy arrays.
The fixtures pin the CPU oracle (oracle/vjf_oracle.py)
and, through it and directly, the HIP path.
"""
import math
import os
import sys

import numpy as np
import torch

REF = os.environ.get("VJF_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import vjf.model as ref_model            # noqa: E402
from vjf import kalman as ref_kalman     # noqa: E402
from vjf.distribution import Gaussian    # noqa: E402
from vjf.functional import gaussian_entropy, gaussian_loss, rbf   # noqa: E402
from vjf.likelihood import PoissonLikelihood                       # noqa: E402
from vjf.model import VJF                # noqa: E402
from vjf.module import RBF, LinearRegression                       # noqa: E402
from vjf.recognition import Recognition  # noqa: E402
from vjf.util import running_var         # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def npy(t):
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy().copy()
    return np.asarray(t)


def export_state(m, prefix):
    """Flatten everything VJF.filter reads/mutates into {prefix.key: ndarray}."""
    d = {}
    d["prior_mean"] = npy(m.mean)
    d["prior_logvar"] = npy(m.logvar)
    lik = m.likelihood
    if hasattr(lik, "logvar"):
        d["lik_logvar"] = npy(lik.logvar)
        d["n_lik"] = np.asarray(lik.n_sample)
    tr = m.transition
    d["tr_logvar"] = npy(tr.logvar)
    d["n_tr"] = np.asarray(tr.n_sample)
    vel = tr.velocity
    d["centroid"] = npy(vel.feature.centroid)
    d["logwidth"] = npy(vel.feature.logwidth)
    d["w_mean"] = npy(vel.w_mean)
    d["w_chol"] = npy(vel.w_chol)
    d["w_precision"] = npy(vel.w_precision)
    d["w_pchol"] = npy(vel.w_pchol)
    lins = [l for l in m.recognition.mlp if isinstance(l, torch.nn.Linear)]
    for k, l in enumerate(lins):
        d[f"rec_W{k}"] = npy(l.weight)
        d[f"rec_b{k}"] = npy(l.bias)
    d["mean_W"] = npy(m.recognition.mean.weight)
    d["lv_W"] = npy(m.recognition.logvar.weight)
    d["lv_b"] = npy(m.recognition.logvar.bias)
    d["dec_W"] = npy(m.decoder.decode.weight)
    d["dec_b"] = npy(m.decoder.decode.bias)
    d["lr"] = np.asarray([g["lr"] for g in m.optimizer.param_groups], np.float64)
    return {f"{prefix}.{k}": v for k, v in d.items()}


class EpsFeeder:
    """Stands in for vjf.model.reparametrize (vjf/util.py:11-13) with recorded noise."""

    def __init__(self, eps):
        self.eps = list(eps)
        self.i = 0

    def __call__(self, q):
        mean, logvar = q
        e = self.eps[self.i]
        self.i += 1
        return mean + e * torch.exp(.5 * logvar)


def traj(name, *, dtype, lik, B, dz, dy, du, n, hidden, T, warm_up, lr=1e-4, seed=0,
         keep_states=(1,), y_scale=1.0):
    torch.set_default_dtype(dtype)
    torch.manual_seed(seed)
    m = VJF.make_model(dy, dz, du, n, hidden, likelihood=lik, lr=lr)
    g = torch.Generator().manual_seed(1000 + seed)
    if lik == "poisson":
        y = torch.poisson(torch.exp(0.5 * torch.randn(T, B, dy, generator=g) - 0.5), generator=g)
    else:
        y = torch.randn(T, B, dy, generator=g) * y_scale
    u = torch.randn(T, B, du, generator=g) if du > 0 else None
    eps = torch.randn(T, 2, B, dz, generator=g)
    rec = {"y": npy(y), "eps": npy(eps)}
    if u is not None:
        rec["u"] = npy(u)
    rec["meta"] = np.asarray([B, dz, dy, du, n, T, int(warm_up)] + list(hidden))
    rec["lik"] = np.asarray(lik)
    rec.update(export_state(m, "s0"))
    feeder = EpsFeeder([eps[t, k] for t in range(T) for k in range(2)])
    orig = ref_model.reparametrize
    ref_model.reparametrize = feeder
    try:
        q = None
        per = {k: [] for k in ("mu", "lv", "loss", "rho", "sigma", "n_lik", "n_tr")}
        for t in range(T):
            ut = None if u is None else u[t]
            q, loss, *el = m.filter(y[t], ut, q, sgd=True, update=True, verbose=True, warm_up=warm_up)
            per["mu"].append(npy(q.mean))
            per["lv"].append(npy(q.logvar))
            per["loss"].append([float(loss)] + [float(e) for e in el])
            per["rho"].append(float(m.likelihood.logvar) if hasattr(m.likelihood, "logvar") else 0.0)
            per["sigma"].append(float(m.transition.logvar))
            per["n_lik"].append(getattr(m.likelihood, "n_sample", 0))
            per["n_tr"].append(m.transition.n_sample)
            if (t + 1) in keep_states:
                rec.update(export_state(m, f"s{t + 1}"))
    finally:
        ref_model.reparametrize = orig
    for k, v in per.items():
        rec[f"out.{k}"] = np.asarray(v)
    rec.update(export_state(m, "sT"))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **rec)
    print(name, "loss", per["loss"][0][0], "->", per["loss"][-1][0])


def g1_rbf():
    torch.set_default_dtype(torch.float64)
    g = torch.Generator().manual_seed(11)
    rec = {}
    for i, (B, n, d) in enumerate([(64, 200, 10), (8, 16, 3), (16, 16, 5), (1, 7, 2)]):
        x = torch.randn(B, d, generator=g)
        c = torch.rand(n, d, generator=g) * 4 - 2
        w = torch.rand(n, generator=g) + 0.5
        rec[f"x{i}"], rec[f"c{i}"], rec[f"w{i}"] = npy(x), npy(c), npy(w)
        rec[f"phi{i}"] = npy(rbf(x, c, w))
    rec["count"] = np.asarray(4)
    np.savez_compressed(os.path.join(OUT, "g1_rbf.npz"), **rec)


def g2_losses():
    torch.set_default_dtype(torch.float64)
    g = torch.Generator().manual_seed(12)
    B, d = 24, 5
    a, b = torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
    la, lb = torch.randn(B, d, generator=g) * 0.5, torch.randn(B, d, generator=g) * 0.5
    lv = torch.tensor(-0.7)
    rec = dict(a=npy(a), b=npy(b), la=npy(la), lb=npy(lb), lv=npy(lv))
    rec["tt"] = npy(gaussian_loss(a, b, lv))
    rec["gg"] = npy(gaussian_loss(Gaussian(a, la), Gaussian(b, lb), lv))
    rec["gt"] = npy(gaussian_loss(Gaussian(a, la), b, lv))
    rec["tg"] = npy(gaussian_loss(a, Gaussian(b, lb), lv))
    rec["ent"] = npy(gaussian_entropy(Gaussian(a, la)))
    eta = torch.randn(B, d, generator=g) * 6 + 4          # several entries above the clamp at 10
    tgt = torch.poisson(torch.rand(B, d, generator=g) * 5, generator=g)
    rec["eta"], rec["tgt"] = npy(eta), npy(tgt)
    rec["n_clamped"] = np.asarray(int((eta > 10).sum()))
    rec["poisson"] = npy(PoissonLikelihood.loss(eta, tgt))
    np.savez_compressed(os.path.join(OUT, "g2_losses.npz"), **rec)


def g3_recognition():
    torch.set_default_dtype(torch.float64)
    rec = {}
    for i, (dy, dz, du, hid, B) in enumerate([(10, 3, 0, [8], 12), (10, 3, 2, [5, 5], 12), (50, 10, 0, [128], 16)]):
        torch.manual_seed(20 + i)
        r = Recognition(dy, dz, du, hid)
        g = torch.Generator().manual_seed(30 + i)
        y = torch.randn(B, dy, generator=g)
        u = torch.randn(B, du, generator=g) if du else None
        mu, lv = torch.randn(B, dz, generator=g), torch.randn(B, dz, generator=g)
        out = r(y, Gaussian(mu, lv), u)
        lins = [l for l in r.mlp if isinstance(l, torch.nn.Linear)]
        rec[f"{i}.meta"] = np.asarray([dy, dz, du, B] + hid)
        for k, l in enumerate(lins):
            rec[f"{i}.rec_W{k}"], rec[f"{i}.rec_b{k}"] = npy(l.weight), npy(l.bias)
        rec[f"{i}.mean_W"], rec[f"{i}.lv_W"], rec[f"{i}.lv_b"] = npy(r.mean.weight), npy(r.logvar.weight), npy(r.logvar.bias)
        rec[f"{i}.y"], rec[f"{i}.mu"], rec[f"{i}.lv"] = npy(y), npy(mu), npy(lv)
        if u is not None:
            rec[f"{i}.u"] = npy(u)
        rec[f"{i}.out_mu"], rec[f"{i}.out_lv"] = npy(out.mean), npy(out.logvar)
    rec["count"] = np.asarray(3)
    np.savez_compressed(os.path.join(OUT, "g3_recognition.npz"), **rec)


def g4_blr():
    torch.set_default_dtype(torch.float64)
    rec = {}
    for i, (n, d, dout, B) in enumerate([(16, 3, 3, 40), (64, 10, 10, 150)]):
        torch.manual_seed(40 + i)
        blr = LinearRegression(RBF(d, n), dout)
        g = torch.Generator().manual_seed(50 + i)
        x1, t1 = torch.randn(B, d, generator=g), torch.randn(B, dout, generator=g) * 0.3
        x2, t2 = torch.randn(B, d, generator=g), torch.randn(B, dout, generator=g) * 0.3
        rec[f"{i}.centroid"], rec[f"{i}.logwidth"] = npy(blr.feature.centroid), npy(blr.feature.logwidth)
        rec[f"{i}.x1"], rec[f"{i}.t1"], rec[f"{i}.x2"], rec[f"{i}.t2"] = npy(x1), npy(t1), npy(x2), npy(t2)
        p0 = blr(x1, sampling=False)
        rec[f"{i}.p0_mean"], rec[f"{i}.p0_logvar"] = npy(p0.mean), npy(p0.logvar)
        for j, (x, t, v) in enumerate([(x1, t1, torch.tensor(1.0)), (x2, t2, torch.tensor(0.37))]):
            blr.rls(x, t, v)
            rec[f"{i}.r{j}.v"] = npy(v)
            rec[f"{i}.r{j}.P"], rec[f"{i}.r{j}.W"] = npy(blr.w_precision), npy(blr.w_mean)
            rec[f"{i}.r{j}.w_chol"], rec[f"{i}.r{j}.w_pchol"] = npy(blr.w_chol), npy(blr.w_pchol)
            p = blr(x2, sampling=False)
            rec[f"{i}.r{j}.mean"], rec[f"{i}.r{j}.logvar"] = npy(p.mean), npy(p.logvar)
    rec["count"] = np.asarray(2)
    np.savez_compressed(os.path.join(OUT, "g4_blr.npz"), **rec)


def g6_running_var():
    rows = []
    for acc_var, acc_size, new_var, new_size, cap in [(0.1, 0, 0.5, 32, 1000), (0.3, 32, 0.2, 32, 1000),
                                                      (0.3, 5000, 0.2, 4096, 1000), (1.0, 499, 0.05, 1, 500),
                                                      (1.0, 501, 0.05, 256, 500)]:
        v, n = running_var(torch.tensor(acc_var, dtype=torch.float64), acc_size, torch.tensor(new_var, dtype=torch.float64),
                           new_size, size_cap=cap)
        rows.append([acc_var, acc_size, new_var, new_size, cap, float(v), n])
    np.savez_compressed(os.path.join(OUT, "g6_running_var.npz"), table=np.asarray(rows))


def g7_kalman():
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(70)
    n, d, S = 10, 3, 20
    blr = LinearRegression(RBF(d, n), d)
    g = torch.Generator().manual_seed(71)
    x, t = torch.randn(S, d, generator=g), torch.randn(S, d, generator=g)
    rec = dict(centroid=npy(blr.feature.centroid), logwidth=npy(blr.feature.logwidth), x=npy(x), t=npy(t))
    blr.kalman(x, t, 0.5, diffusion=0.01)
    rec["W1"], rec["L1"] = npy(blr.w_mean), npy(blr.w_chol)
    blr.kalman(t, x, 0.25)
    rec["W2"], rec["L2"] = npy(blr.w_mean), npy(blr.w_chol)
    np.savez_compressed(os.path.join(OUT, "g7_kalman.npz"), **rec)


def g8_fit():
    """fit() for 3 epochs with a loose rtol so that epoch 1 leaves warm-up (decoder frozen,
    centroids re-drawn, one RLS) and epoch 2 converges; then forecast with recorded noise."""
    torch.set_default_dtype(torch.float64)
    torch.manual_seed(80)
    T, B, dy, dz, du, n, hid = 50, 4, 10, 3, 0, 12, [6]
    m = VJF.make_model(dy, dz, du, n, hid, likelihood="gaussian")
    g = torch.Generator().manual_seed(81)
    y = torch.randn(T, B, dy, generator=g)
    rec = {"y": npy(y), "meta": np.asarray([T, B, dy, dz, du, n] + hid)}
    rec.update(export_state(m, "s0"))
    drawn = []
    orig = ref_model.reparametrize

    def recording(q):
        mean, logvar = q
        e = torch.randn_like(mean)
        drawn.append(npy(e))
        return mean + e * torch.exp(.5 * logvar)

    ref_model.reparametrize = recording
    try:
        torch.manual_seed(82)
        mu, lv, epoch_loss = m.fit(y, max_iter=3, rtol=10.0)
    finally:
        ref_model.reparametrize = orig
    rec["eps"] = np.asarray(drawn)               # (epochs*T*2, B, dz), order xs, xt per step
    rec["fit_seed"] = np.asarray(82)
    rec["mu"], rec["lv"], rec["epoch_loss"] = npy(mu), npy(lv), npy(epoch_loss)
    rec.update(export_state(m, "sT"))
    # forecast with recorded weight noise (vjf/module.py:70-73 draws randn_like(w) every step)
    n_step = 6
    # randn_like(w) follows w's strides (w_mean is column-major after cholesky_solve), so record the
    # noise the reference actually used instead of re-drawing it.
    used = []
    orig_randn_like = torch.randn_like

    def rec_randn_like(t, *a, **k):
        e = orig_randn_like(t, *a, **k)
        used.append(npy(e))
        return e

    torch.randn_like = rec_randn_like
    try:
        torch.manual_seed(83)
        x0 = mu[-1]
        xf, yf = m.forecast(x0, n_step=n_step, noise=False)
    finally:
        torch.randn_like = orig_randn_like
    wn = np.asarray(used)
    rec["fc_seed"] = np.asarray(83)
    rec["fc_w_colmajor"] = np.asarray(int(not m.transition.velocity.w_mean.is_contiguous()))
    rec["fc_wnoise"], rec["fc_x0"], rec["fc_x"], rec["fc_y"] = npy(wn), npy(x0), npy(xf), npy(yf)
    np.savez_compressed(os.path.join(OUT, "g8_fit.npz"), **rec)
    print("g8 epochs drawn:", len(drawn) // (2 * T), "epoch_loss", float(epoch_loss))


def g5_seeded():
    """Un-patched filter under torch.manual_seed: pins 'init under seed' and 'draw noise in the
    reference's order from torch's CPU generator' for the host shim."""
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(0)
    m = VJF.make_model(10, 3, 2, 16, [8], likelihood="gaussian")
    rec = export_state(m, "s0")
    g = torch.Generator().manual_seed(90)
    T, B = 4, 32
    y, u = torch.randn(T, B, 10, generator=g), torch.randn(T, B, 2, generator=g)
    rec["y"], rec["u"] = npy(y), npy(u)
    torch.manual_seed(7)
    q, mus, lvs, losses = None, [], [], []
    for t in range(T):
        q, loss, *el = m.filter(y[t], u[t], q, verbose=True)
        mus.append(npy(q.mean)); lvs.append(npy(q.logvar)); losses.append([float(loss)] + [float(e) for e in el])
    rec["out.mu"], rec["out.lv"], rec["out.loss"] = np.asarray(mus), np.asarray(lvs), np.asarray(losses)
    rec["seeds"] = np.asarray([0, 7])
    rec.update(export_state(m, "sT"))
    np.savez_compressed(os.path.join(OUT, "g5_seeded_f32.npz"), **rec)


def main():
    g1_rbf(); g2_losses(); g3_recognition(); g4_blr(); g6_running_var(); g7_kalman()
    small = dict(B=32, dz=3, dy=10, n=16, hidden=[8], T=8)
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        for lik in ("gaussian", "poisson"):
            for du in (0, 2):
                for wu in (False, True):
                    traj(f"g5_{lik}_du{du}_wu{int(wu)}_{tag}", dtype=dt, lik=lik, du=du, warm_up=wu, **small)
        traj(f"g5_gaussian_lr1e-2_{tag}", dtype=dt, lik="gaussian", du=0, warm_up=False, lr=1e-2, **small)
        traj(f"g5_gaussian_h5x5_{tag}", dtype=dt, lik="gaussian", du=1, warm_up=False, lr=1e-3,
             B=20, dz=3, dy=10, n=16, hidden=[5, 5], T=6)
    # medium: BASELINE config-B dimensions at a reduced batch; final state only (size)
    traj("g5_medium_gaussian_f32", dtype=torch.float32, lik="gaussian", B=256, dz=10, dy=50, du=0, n=200,
         hidden=[128], T=6, warm_up=False, keep_states=())
    traj("g5_medium_poisson_f32", dtype=torch.float32, lik="poisson", B=128, dz=10, dy=200, du=0, n=200,
         hidden=[128], T=4, warm_up=False, keep_states=())
    g5_seeded()
    g8_fit()
    torch.set_default_dtype(torch.float32)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT) if f.endswith(".npz"))
    print("total fixture bytes:", tot)


if __name__ == "__main__":
    main()
