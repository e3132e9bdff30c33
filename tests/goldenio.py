"""Helpers shared by the tests: load golden .npz fixtures into oracle states."""
import glob
import os

import numpy as np

from oracle import vjf_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def traj_names(tag=None):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "g5_*.npz")))
    names = [n for n in names if "seeded" not in n]
    if tag:
        names = [n for n in names if n.endswith(tag)]
    return names


def state_from(z, prefix, *, ydim, xdim, udim, n_rbf, hidden, likelihood):
    """Build an OracleState from the `prefix.*` arrays of a fixture."""
    g = lambda k: z[f"{prefix}.{k}"]                                    # noqa: E731
    s = orc.OracleState(ydim, xdim, udim, n_rbf, tuple(hidden), likelihood)
    s.prior_mean, s.prior_logvar = g("prior_mean"), g("prior_logvar")
    if likelihood == orc.GAUSSIAN:
        s.lik_logvar = g("lik_logvar")
        s.n_lik = int(g("n_lik"))
    s.tr_logvar, s.n_tr = g("tr_logvar"), int(g("n_tr"))
    s.centroid, s.logwidth = g("centroid"), g("logwidth")
    s.w_mean, s.w_chol = g("w_mean"), g("w_chol")
    s.w_precision, s.w_pchol = g("w_precision"), g("w_pchol")
    s.rec_W = [g(f"rec_W{k}") for k in range(len(hidden))]
    s.rec_b = [g(f"rec_b{k}") for k in range(len(hidden))]
    s.mean_W, s.lv_W, s.lv_b = g("mean_W"), g("lv_W"), g("lv_b")
    s.dec_W, s.dec_b = g("dec_W"), g("dec_b")
    s.lr = [float(x) for x in g("lr")]
    return s


def traj_case(name):
    """Return (z, meta dict, initial OracleState) for a g5 trajectory fixture."""
    z = load(name)
    meta = [int(v) for v in z["meta"]]
    B, dz, dy, du, n, T, wu = meta[:7]
    hidden = meta[7:]
    lik = str(z["lik"])
    s0 = state_from(z, "s0", ydim=dy, xdim=dz, udim=du, n_rbf=n, hidden=hidden, likelihood=lik)
    info = dict(B=B, dz=dz, dy=dy, du=du, n=n, T=T, warm_up=bool(wu), hidden=hidden, lik=lik)
    return z, info, s0


STATE_KEYS = ["lik_logvar", "tr_logvar", "centroid", "logwidth", "w_mean", "w_chol", "w_precision",
              "w_pchol", "mean_W", "lv_W", "lv_b", "dec_W", "dec_b"]


def state_arrays(s):
    """{key: ndarray} view of an OracleState in fixture naming."""
    d = {k: getattr(s, k) for k in STATE_KEYS if getattr(s, k) is not None}
    for k, (W, b) in enumerate(zip(s.rec_W, s.rec_b)):
        d[f"rec_W{k}"], d[f"rec_b{k}"] = W, b
    return d
