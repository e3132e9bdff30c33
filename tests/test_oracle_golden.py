"""Pin the CPU oracle (oracle/vjf_oracle.py) to golden vectors captured from the reference.

CPU-only.  fp64 fixtures must match to ~1e-10; fp32 fixtures to fp32 rounding (the oracle's
operation order differs from torch's, e.g. direct squared distances vs cdist's mm expansion).
"""
import numpy as np
import pytest

from oracle import vjf_oracle as orc
from tests import goldenio as gio

F64 = dict(rtol=1e-9, atol=1e-10)


def close(a, b, **kw):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), **kw)


def test_g1_rbf():
    z = gio.load("g1_rbf")
    for i in range(int(z["count"])):
        close(orc.rbf(z[f"x{i}"], z[f"c{i}"], z[f"w{i}"]), z[f"phi{i}"], rtol=1e-9, atol=1e-13)


def test_g2_losses():
    z = gio.load("g2_losses")
    a, b, la, lb, lv = z["a"], z["b"], z["la"], z["lb"], z["lv"]
    close(orc.gaussian_loss(a, None, b, None, lv), z["tt"], **F64)
    close(orc.gaussian_loss(a, la, b, lb, lv), z["gg"], **F64)
    close(orc.gaussian_loss(a, la, b, None, lv), z["gt"], **F64)
    close(orc.gaussian_loss(a, None, b, lb, lv), z["tg"], **F64)
    close(orc.gaussian_entropy(la), z["ent"], **F64)
    assert int(z["n_clamped"]) > 0
    close(orc.poisson_loss(z["eta"], z["tgt"]), z["poisson"], **F64)


def test_g3_recognition():
    z = gio.load("g3_recognition")
    for i in range(int(z["count"])):
        meta = [int(v) for v in z[f"{i}.meta"]]
        dy, dz, du, B = meta[:4]
        hid = meta[4:]
        s = orc.OracleState(dy, dz, du, 1, tuple(hid), orc.GAUSSIAN)
        s.rec_W = [z[f"{i}.rec_W{k}"] for k in range(len(hid))]
        s.rec_b = [z[f"{i}.rec_b{k}"] for k in range(len(hid))]
        s.mean_W, s.lv_W, s.lv_b = z[f"{i}.mean_W"], z[f"{i}.lv_W"], z[f"{i}.lv_b"]
        u = z[f"{i}.u"] if du else None
        mu, lv = orc.recognition_forward(s, z[f"{i}.y"], z[f"{i}.mu"], z[f"{i}.lv"], u)
        close(mu, z[f"{i}.out_mu"], **F64)
        close(lv, z[f"{i}.out_lv"], **F64)


def test_g4_blr_predict_and_rls():
    z = gio.load("g4_blr")
    for i in range(int(z["count"])):
        c, lw = z[f"{i}.centroid"], z[f"{i}.logwidth"]
        n = c.shape[0]
        dout = z[f"{i}.t1"].shape[1]
        s = orc.OracleState(1, dout, 0, n, (1,), orc.GAUSSIAN)
        s.centroid, s.logwidth = c, lw
        s.w_mean = np.zeros((n, dout))
        s.w_chol = np.eye(n)
        s.w_precision = np.eye(n)
        s.w_pchol = np.eye(n)
        m, lv, _ = orc.blr_predict(s, z[f"{i}.x1"])
        close(m, z[f"{i}.p0_mean"], **F64)
        close(lv, z[f"{i}.p0_logvar"], rtol=1e-9, atol=1e-9)
        for j, (x, t) in enumerate([(z[f"{i}.x1"], z[f"{i}.t1"]), (z[f"{i}.x2"], z[f"{i}.t2"])]):
            feat = orc.rbf(x, c, np.exp(lw))
            assert orc.rls(s, feat, t, z[f"{i}.r{j}.v"]) == 0
            close(s.w_precision, z[f"{i}.r{j}.P"], rtol=1e-9, atol=1e-9)
            close(s.w_pchol, z[f"{i}.r{j}.w_pchol"], rtol=1e-8, atol=1e-9)
            close(s.w_mean, z[f"{i}.r{j}.W"], rtol=1e-7, atol=1e-9)
            close(s.w_chol, z[f"{i}.r{j}.w_chol"], rtol=1e-7, atol=1e-9)
            m, lv, _ = orc.blr_predict(s, z[f"{i}.x2"])
            close(m, z[f"{i}.r{j}.mean"], rtol=1e-7, atol=1e-9)
            close(lv, z[f"{i}.r{j}.logvar"], rtol=1e-7, atol=1e-8)
            # the (B,B)-diagonal form is the same number
            m2, lv2, _ = orc.blr_predict(s, z[f"{i}.x2"], faithful_cost=True)
            close(lv2, lv, rtol=1e-10, atol=1e-10)


def test_g6_running_var():
    for row in gio.load("g6_running_var")["table"]:
        av, asz, nv, nsz, cap, want_v, want_n = row
        v, n = orc.running_var(av, int(asz), nv, int(nsz), size_cap=int(cap))
        assert n == int(want_n)
        close(v, want_v, rtol=1e-14, atol=0)


def test_g7_kalman():
    z = gio.load("g7_kalman")
    n, d = z["centroid"].shape
    s = orc.OracleState(1, d, 0, n, (1,), orc.GAUSSIAN)
    s.centroid, s.logwidth = z["centroid"], z["logwidth"]
    s.w_mean, s.w_chol = np.zeros((n, d)), np.eye(n)
    orc.blr_kalman(s, z["x"], z["t"], 0.5, diffusion=0.01)
    close(s.w_mean, z["W1"], rtol=1e-8, atol=1e-10)
    close(s.w_chol, z["L1"], rtol=1e-8, atol=1e-10)
    orc.blr_kalman(s, z["t"], z["x"], 0.25)
    close(s.w_mean, z["W2"], rtol=1e-7, atol=1e-10)
    close(s.w_chol, z["L2"], rtol=1e-6, atol=1e-9)


def _run_traj(name):
    z, info, s = gio.traj_case(name)
    u = z["u"] if info["du"] else None
    outs = []
    for t in range(info["T"]):
        mu = outs[-1].mu_t if outs else None
        lv = outs[-1].lv_t if outs else None
        o = orc.filter_step(s, z["y"][t], None if u is None else u[t], mu, lv, z["eps"][t, 0], z["eps"][t, 1],
                            sgd=True, update=True, warm_up=info["warm_up"])
        outs.append(o)
        o.rho = float(s.lik_logvar) if s.lik_logvar is not None else 0.0
        o.sigma = float(s.tr_logvar)
        o.n_lik, o.n_tr = s.n_lik, s.n_tr
        if f"s{t + 1}.w_mean" in z.files:
            o.state = s.clone()
    return z, info, s, outs


@pytest.mark.parametrize("name", gio.traj_names("f64"))
def test_g5_trajectory_f64(name):
    z, info, s, outs = _run_traj(name)
    for t, o in enumerate(outs):
        close(o.mu_t, z["out.mu"][t], rtol=1e-8, atol=1e-10)
        close(o.lv_t, z["out.lv"][t], rtol=1e-8, atol=1e-10)
        close([o.loss, o.recon, o.dyn, o.entropy], z["out.loss"][t], rtol=1e-9, atol=1e-10)
        close(o.rho, z["out.rho"][t], rtol=1e-9, atol=1e-10)
        close(o.sigma, z["out.sigma"][t], rtol=1e-8, atol=1e-10)
        assert o.n_lik == int(z["out.n_lik"][t]) and o.n_tr == int(z["out.n_tr"][t])
        if hasattr(o, "state"):
            for k, v in gio.state_arrays(o.state).items():
                close(v, z[f"s{t + 1}.{k}"], rtol=1e-7, atol=1e-10)
    for k, v in gio.state_arrays(s).items():
        close(v, z[f"sT.{k}"], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("name", gio.traj_names("f32"))
def test_g5_trajectory_f32(name):
    """fp32 oracle vs fp32 reference: rounding-level agreement (restatement 2e-6 on
    O(1) posteriors; looser on the RLS factors, whose conditioning amplifies rounding)."""
    z, info, s, outs = _run_traj(name)
    assert s.dtype == np.float32
    for t, o in enumerate(outs):
        close(o.mu_t, z["out.mu"][t], rtol=2e-5, atol=2e-5)
        close(o.lv_t, z["out.lv"][t], rtol=2e-5, atol=2e-5)
        close([o.loss, o.recon, o.dyn, o.entropy], z["out.loss"][t], rtol=2e-5, atol=2e-5)
        close(o.sigma, z["out.sigma"][t], rtol=0, atol=2e-5)
        close(o.rho, z["out.rho"][t], rtol=0, atol=2e-5)
    close(s.w_mean, z["sT.w_mean"], rtol=1e-3, atol=2e-5)
    close(s.w_precision, z["sT.w_precision"], rtol=1e-4, atol=1e-4)
    close(s.w_chol, z["sT.w_chol"], rtol=1e-3, atol=2e-5)
    for k in ("mean_W", "lv_W", "lv_b", "dec_W", "dec_b", "rec_W0", "rec_b0"):
        close(gio.state_arrays(s)[k], z[f"sT.{k}"], rtol=1e-4, atol=1e-5)


def test_f64_oracle_tracks_f32_reference_medium():
    """The fp64 oracle started from the fp32 medium fixture's state stays within the stated GPU
    acceptance band of the fp32 reference -- i.e. the band is honest."""
    z, info, s0 = gio.traj_case("g5_medium_gaussian_f32")
    s = s0.cast(np.float64)
    mu = lv = None
    for t in range(info["T"]):
        o = orc.filter_step(s, z["y"][t].astype(np.float64), None, mu, lv, z["eps"][t, 0].astype(np.float64),
                            z["eps"][t, 1].astype(np.float64))
        mu, lv = o.mu_t, o.lv_t
        close(o.mu_t, z["out.mu"][t], rtol=1e-5, atol=1e-5)
        close(o.lv_t, z["out.lv"][t], rtol=1e-5, atol=1e-5)
        close(o.loss, z["out.loss"][t][0], rtol=2e-5)
    close(s.w_mean, z["sT.w_mean"], rtol=1e-3, atol=1e-5)


def test_g8_fit_harness_and_forecast():
    """Replays fit()'s epochs (warm-up -> freeze decoder -> initialize -> converge) with the oracle
    step and the recorded noise; pins RBFDS.initialize and forecast."""
    z = gio.load("g8_fit")
    T, B, dy, dz, du, n = [int(v) for v in z["meta"][:6]]
    hid = [int(v) for v in z["meta"][6:]]
    s = gio.state_from(z, "s0", ydim=dy, xdim=dz, udim=du, n_rbf=n, hidden=hid, likelihood=orc.GAUSSIAN)
    y = z["y"]
    eps = z["eps"].reshape(-1, T, 2, B, dz)
    assert eps.shape[0] == 3
    gamma = 0.9
    warm_up, running = True, float("nan")
    for ep in range(3):
        mus, lvs, losses = orc.filter_sequence(s, y, None, eps[ep], warm_up=warm_up)
        epoch_loss = losses[:, 0].mean()
        if warm_up:
            if np.isclose(epoch_loss, running, rtol=10.0):
                warm_up = False
                running = epoch_loss
                s.freeze_decoder = True
                r = np.abs(z["sT.centroid"]).max()      # recover the U(0,1) draw from the recorded centroids
                m = mus.reshape(T, B, dz)
                xs_all, xt_all = m[:-1].reshape(-1, dz), m[1:].reshape(-1, dz)
                r = float(np.sqrt((xs_all ** 2).sum(1)).max())
                draw = (z["sT.centroid"] / r + 1) / 2
                orc.rbfds_initialize(s, xt_all, xs_all, None, draw)
        else:
            if np.isclose(epoch_loss, running, rtol=10.0):
                break
        running = 0.1 * running + 0.9 * epoch_loss if ep > 0 else epoch_loss
        s.lr = [l * gamma for l in s.lr]
    assert ep == 2 and not warm_up
    close(mus, z["mu"], rtol=1e-7, atol=1e-9)
    close(lvs, z["lv"], rtol=1e-7, atol=1e-9)
    close(epoch_loss, z["epoch_loss"], rtol=1e-9)
    for k, v in gio.state_arrays(s).items():
        close(v, z[f"sT.{k}"], rtol=1e-6, atol=1e-9)
    x, yf = orc.forecast(s, z["fc_x0"], None, z["fc_wnoise"].shape[0], z["fc_wnoise"])
    close(x, z["fc_x"], rtol=1e-8, atol=1e-10)
    close(yf, z["fc_y"], rtol=1e-8, atol=1e-10)
