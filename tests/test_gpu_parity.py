"""Parity of the HIP path (through the C ABI / vjf_amd host mirror) with the golden vectors captured
from the reference and with the CPU oracle.  All tests need a real MI355X:  pytest -m gpu.

Stated fp32 tolerances: posterior mean / logvar
atol 2e-5 + rtol 2e-5, loss components rtol 2e-5, RLS weights/factors rtol 1e-3 (their
conditioning amplifies fp32 rounding; the fp32 reference itself differs from fp64 by ~4e-6..1e-4).
"""
import numpy as np
import pytest
import torch

from oracle import vjf_oracle as orc
from tests.margins import check_close
from tests import goldenio as gio
from tests.helpers import load_fixture_state, load_oracle_state, state_close

pytestmark = pytest.mark.gpu

POST = dict(rtol=1e-6, atol=1e-6)      # (set from the achieved margins: profiles/r04_parity_margins.json, tests/margins.py)


def close(a, b, **kw):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    check_close(np.asarray(a, np.float64), np.asarray(b, np.float64), **kw)      # (asserts, and records the achieved margin)


@pytest.fixture(scope="module")
def vjf():
    import vjf_amd
    assert torch.cuda.is_available()
    return vjf_amd


# ------------------------------------------------------------------ stand-alone operators vs golden
def test_rbf_golden(vjf):
    z = gio.load("g1_rbf")
    for i in range(int(z["count"])):
        out = vjf.functional.rbf(torch.tensor(z[f"x{i}"]), torch.tensor(z[f"c{i}"]), torch.tensor(z[f"w{i}"]))
        close(out, z[f"phi{i}"], rtol=5e-6, atol=1e-7)


def test_losses_golden(vjf):
    z = gio.load("g2_losses")
    G, F = vjf.Gaussian, vjf.functional
    a, b, la, lb, lv = (torch.tensor(z[k]) for k in ("a", "b", "la", "lb", "lv"))
    close(F.gaussian_loss(a, b, lv), z["tt"], rtol=1e-6)
    close(F.gaussian_loss(G(a, la), G(b, lb), lv), z["gg"], rtol=1e-6)
    close(F.gaussian_loss(G(a, la), b, lv), z["gt"], rtol=1e-6)
    close(F.gaussian_loss(a, G(b, lb), lv), z["tg"], rtol=1e-6)
    close(F.gaussian_entropy(G(a, la)), z["ent"], rtol=1e-6, atol=1e-6)
    close(vjf.likelihood.PoissonLikelihood.loss(torch.tensor(z["eta"]), torch.tensor(z["tgt"])), z["poisson"], rtol=1e-6)


def test_recognition_golden(vjf):
    z = gio.load("g3_recognition")
    for i in range(int(z["count"])):
        meta = [int(v) for v in z[f"{i}.meta"]]
        dy, dz, du, B = meta[:4]
        hid = meta[4:]
        r = vjf.recognition.Recognition(dy, dz, du, hid)
        for k, lin in enumerate(r.linears()):
            lin.weight.copy_(torch.tensor(z[f"{i}.rec_W{k}"]))
            lin.bias.copy_(torch.tensor(z[f"{i}.rec_b{k}"]))
        r.mean.weight.copy_(torch.tensor(z[f"{i}.mean_W"]))
        r.logvar.weight.copy_(torch.tensor(z[f"{i}.lv_W"]))
        r.logvar.bias.copy_(torch.tensor(z[f"{i}.lv_b"]))
        u = torch.tensor(z[f"{i}.u"]) if du else None
        q = r(torch.tensor(z[f"{i}.y"]), vjf.Gaussian(torch.tensor(z[f"{i}.mu"]), torch.tensor(z[f"{i}.lv"])), u)
        close(q.mean, z[f"{i}.out_mu"], rtol=2e-6, atol=1e-6)
        close(q.logvar, z[f"{i}.out_lv"], rtol=2e-6, atol=1e-6)


def test_blr_predict_and_rls_golden(vjf):
    z = gio.load("g4_blr")
    for i in range(int(z["count"])):
        n, d = z[f"{i}.centroid"].shape
        dout = z[f"{i}.t1"].shape[1]
        blr = vjf.module.LinearRegression(vjf.module.RBF(d, n), dout)
        blr.feature.centroid.copy_(torch.tensor(z[f"{i}.centroid"]))
        blr.feature.logwidth.copy_(torch.tensor(z[f"{i}.logwidth"]))
        p = blr(torch.tensor(z[f"{i}.x1"]), sampling=False)
        close(p.mean, z[f"{i}.p0_mean"], rtol=1e-6, atol=1e-6)
        close(p.logvar, z[f"{i}.p0_logvar"], rtol=1e-6, atol=1e-6)
        for j, (x, t) in enumerate([(z[f"{i}.x1"], z[f"{i}.t1"]), (z[f"{i}.x2"], z[f"{i}.t2"])]):
            blr.rls(torch.tensor(x), torch.tensor(t), torch.tensor(z[f"{i}.r{j}.v"]))
            close(blr.w_precision, z[f"{i}.r{j}.P"], rtol=1e-6, atol=1e-6)
            close(blr.w_pchol, z[f"{i}.r{j}.w_pchol"], rtol=5e-6, atol=1e-6)
            close(blr.w_mean, z[f"{i}.r{j}.W"], rtol=2e-5, atol=1e-6)
            close(blr.w_chol, z[f"{i}.r{j}.w_chol"], rtol=2e-5, atol=1e-6)
            p = blr(torch.tensor(z[f"{i}.x2"]), sampling=False)
            close(p.mean, z[f"{i}.r{j}.mean"], rtol=2e-5, atol=1e-6)
            close(p.logvar, z[f"{i}.r{j}.logvar"], rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ the hot path vs golden trajectories
def _model_for(vjf, info, lr=1e-4):
    return vjf.VJF.make_model(info["dy"], info["dz"], info["du"], info["n"], info["hidden"], likelihood=info["lik"], lr=lr)


@pytest.mark.parametrize("name", gio.traj_names())
def test_filter_trajectory_golden(vjf, name):
    z, info, _ = gio.traj_case(name)
    model = _model_for(vjf, info)
    load_fixture_state(model, z, "s0")
    u = z["u"] if info["du"] else None
    q = None
    for t in range(info["T"]):
        ut = None if u is None else torch.tensor(u[t])
        q, loss, *comp = model.filter(torch.tensor(z["y"][t]), ut, q, sgd=True, update=True, verbose=True,
                                      warm_up=info["warm_up"], eps=(torch.tensor(z["eps"][t, 0]), torch.tensor(z["eps"][t, 1])))
        close(q.mean, z["out.mu"][t], **POST)
        close(q.logvar, z["out.lv"][t], **POST)
        close(torch.stack([loss, *comp]), z["out.loss"][t], rtol=1e-6, atol=1e-6)
        close(model.transition.logvar, z["out.sigma"][t], rtol=0, atol=1e-6)
        if info["lik"] == "gaussian":
            close(model.likelihood.logvar, z["out.rho"][t], rtol=0, atol=1e-6)
            assert model.likelihood.n_sample == int(z["out.n_lik"][t])
        assert model.transition.n_sample == int(z["out.n_tr"][t])
        if f"s{t + 1}.w_mean" in z.files:
            state_close(model, z, prefix=f"s{t + 1}", rtol=5e-6, atol=1e-6, rls_rtol=5e-5, rls_atol=1e-6)
    state_close(model, z, prefix="sT", rtol=5e-6, atol=1e-6, rls_rtol=5e-4, rls_atol=5e-6)
    assert model.status() == 0


@pytest.mark.parametrize("name", ["g5_gaussian_du2_wu0_f32", "g5_medium_poisson_f32", "g5_gaussian_h5x5_f32"])
def test_filter_sequence_equals_steps(vjf, name):
    z, info, _ = gio.traj_case(name)
    m1, m2 = _model_for(vjf, info), _model_for(vjf, info)
    load_fixture_state(m1, z, "s0")
    load_fixture_state(m2, z, "s0")
    u = torch.tensor(z["u"]) if info["du"] else None
    mu, lv, loss = m1.filter_sequence(torch.tensor(z["y"]), u, None, eps=torch.tensor(z["eps"]))
    q = None
    for t in range(info["T"]):
        q, l, *c = m2.filter(torch.tensor(z["y"][t]), None if u is None else u[t], q, verbose=True,
                             eps=(torch.tensor(z["eps"][t, 0]), torch.tensor(z["eps"][t, 1])))
        assert torch.equal(q.mean, mu[t]) and torch.equal(q.logvar, lv[t])      # same kernels, same order: bitwise
        assert torch.equal(torch.stack([l, *c]), loss[t])
    assert torch.equal(m1._blob, m2._blob)
    close(mu, z["out.mu"], **POST)


def test_filter_sequence_in_chunks(vjf, monkeypatch):
    """Long sequences are enqueued in chunks (one launch per chunk; a chunk of ONE step is what `filter` runs): same bits as one piece."""
    z, info, _ = gio.traj_case("g5_medium_gaussian_f32")
    y, eps = torch.tensor(z["y"]), torch.tensor(z["eps"])
    outs = []
    for chunk in (None, "3", "1"):
        if chunk:
            monkeypatch.setenv("VJF_SEQ_CHUNK", chunk)
        m = _model_for(vjf, info)
        load_fixture_state(m, z, "s0")
        o = m.filter_sequence(y, None, None, eps=eps)
        assert m.status() == 0
        outs.append((o, m._blob.clone()))
    monkeypatch.delenv("VJF_SEQ_CHUNK", raising=False)
    for k in (1, 2):
        for a, b in zip(outs[0][0], outs[k][0]):
            assert torch.equal(a, b)
        assert torch.equal(outs[0][1], outs[k][1])


def test_filter_sequence_one_launch_vs_per_step_kernels(vjf):
    """vjf_filter_seq's one-launch route (every role a workgroup of one resident grid) against the per-step kernels in
    the one-stream order (`set_overlap(False)`): different kernels, different summation trees -- every output and the whole
    state agree to fp32 summation-order tolerance, over two calls (the second starts from a posterior, triangles clean)."""
    z, info, _ = gio.traj_case("g5_medium_gaussian_f32")
    m1, m2 = _model_for(vjf, info), _model_for(vjf, info)
    load_fixture_state(m1, z, "s0")
    load_fixture_state(m2, z, "s0")
    m2.set_overlap(False)
    u = torch.tensor(z["u"]) if info["du"] else None
    for rep in range(2):
        q1 = q2 = None
        if rep:
            q1, q2 = (vjf.Gaussian(o[0][-1], o[1][-1]) for o in (o1, o2))
        o1 = m1.filter_sequence(torch.tensor(z["y"]), u, q1, eps=torch.tensor(z["eps"]))
        o2 = m2.filter_sequence(torch.tensor(z["y"]), u, q2, eps=torch.tensor(z["eps"]))
        for a, b in zip(o1, o2):
            close(a, b, rtol=1e-6, atol=1e-6)
        close(m1._blob, m2._blob, rtol=1e-4, atol=1e-6)
    assert m1.status() == 0 and m2.status() == 0


def test_seeded_drop_in(vjf):
    """make_model under torch.manual_seed + noise drawn from the CPU generator in the reference's
    order reproduce the reference's un-patched trajectory."""
    z = gio.load("g5_seeded_f32")
    torch.manual_seed(int(z["seeds"][0]))
    model = vjf.VJF.make_model(10, 3, 2, 16, [8], likelihood="gaussian")
    state_close(model, z, prefix="s0", rtol=0, atol=0)
    torch.manual_seed(int(z["seeds"][1]))
    q = None
    for t in range(z["y"].shape[0]):
        q, loss, *comp = model.filter(torch.tensor(z["y"][t]), torch.tensor(z["u"][t]), q, verbose=True)
        close(q.mean, z["out.mu"][t], **POST)
        close(q.logvar, z["out.lv"][t], **POST)
        close(torch.stack([loss, *comp]), z["out.loss"][t], rtol=1e-6, atol=1e-6)
    state_close(model, z, prefix="sT", rtol=2e-6, atol=1e-6, rls_rtol=5e-5, rls_atol=1e-6)


# ------------------------------------------------------------------ vs the oracle on fresh seeded inputs
CASES = [
    dict(B=1, dz=3, dy=10, du=0, n=100, hidden=[20], lik="gaussian", T=3),          # BASELINE configs[0]-like, one trial
    dict(B=300, dz=10, dy=50, du=0, n=200, hidden=[128], lik="gaussian", T=3),      # config B dims, ragged batch
    dict(B=77, dz=10, dy=200, du=0, n=200, hidden=[128], lik="poisson", T=2),       # config C dims
    dict(B=50, dz=5, dy=7, du=3, n=33, hidden=[40, 24, 9], lik="gaussian", T=3),     # odd sizes, 3 layers, control input
    dict(B=37, dz=64, dy=512, du=0, n=300, hidden=[512, 512], lik="gaussian", T=2),  # config E layer widths (n reduced)
    dict(B=64, dz=6, dy=12, du=0, n=228, hidden=[16], lik="gaussian", T=3),          # just beyond one CU's LDS: multi-launch RLS
    dict(B=40, dz=12, dy=20, du=1, n=500, hidden=[24], lik="poisson", T=2),          # 16 blocks, last one partial
    dict(B=48, dz=4, dy=9, du=0, n=222, hidden=[12], lik="gaussian", T=3),           # n % 4 != 0: single-workgroup serial kernel
    dict(B=32, dz=20, dy=30, du=0, n=96, hidden=[32], lik="gaussian", T=3),          # 16 < dz <= 32: LDS Cholesky kernel with its own inverse / solve
    dict(B=24, dz=40, dy=16, du=0, n=64, hidden=[16], lik="gaussian", T=2),          # dz > 32, small n: single-workgroup serial kernel
    dict(B=20, dz=40, dy=16, du=2, n=260, hidden=[16], lik="poisson", T=2),          # dz > 32, n > 224: multi-launch RLS, wide latent
    dict(B=52, dz=7, dy=11, du=1, n=250, hidden=[20], lik="gaussian", T=3),          # multi-launch RLS with n % 4 != 0: float-by-float operand paths, K tails
    dict(B=300, dz=9, dy=301, du=0, n=1210, hidden=[401], lik="gaussian", T=2),      # GEMM-per-layer trial path, every extent off the tile sizes, unaligned rows
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"B{c['B']}_dz{c['dz']}_dy{c['dy']}_{c['lik']}")
def test_filter_vs_oracle(vjf, case):
    torch.manual_seed(5)
    c = dict(case)
    T = c.pop("T")
    model = vjf.VJF.make_model(c["dy"], c["dz"], c["du"], c["n"], c["hidden"], likelihood=c["lik"], lr=1e-3)
    if c["dz"] >= 32:      # default RBF init underflows at large dz: use initialize-style init
        r = float(np.sqrt(c["dz"]))
        model.transition.velocity.feature.centroid.uniform_(-r, r)
        model.transition.velocity.feature.logwidth.fill_(float(np.log(r)))
    s = load_oracle_state(model, np.float64)
    g = torch.Generator().manual_seed(9)
    B, dz = c["B"], c["dz"]
    if c["lik"] == "poisson":
        y = torch.poisson(torch.exp(0.5 * torch.randn(T, B, c["dy"], generator=g) - 0.5), generator=g)
    else:
        y = torch.randn(T, B, c["dy"], generator=g)
    u = torch.randn(T, B, c["du"], generator=g) if c["du"] else None
    eps = torch.randn(T, 2, B, dz, generator=g)
    q, mu, lv = None, None, None
    for t in range(T):
        ut = None if u is None else u[t]
        q, loss, *comp = model.filter(y[t], ut, q, verbose=True, eps=(eps[t, 0], eps[t, 1]))
        o = orc.filter_step(s, y[t].numpy(), None if u is None else ut.numpy(), mu, lv, eps[t, 0].numpy(), eps[t, 1].numpy())
        mu, lv = o.mu_t, o.lv_t
        close(q.mean, o.mu_t, rtol=2e-6, atol=2e-6)
        close(q.logvar, o.lv_t, rtol=2e-6, atol=2e-6)
        close(torch.stack([loss, *comp]), [o.loss, o.recon, o.dyn, o.entropy], rtol=2e-5, atol=2e-5)
        close(model.transition.logvar, s.tr_logvar, rtol=0, atol=2e-5)
    state_close(model, s, rtol=5e-6, atol=1e-6, rls_rtol=3e-3, rls_atol=3e-5)
    assert model.status() == 0


@pytest.mark.parametrize("case", CASES[:4], ids=lambda c: f"B{c['B']}_dz{c['dz']}_dy{c['dy']}_{c['lik']}")
def test_filter_sequence_vs_oracle(vjf, case):
    """The sequence entry point (one launch) on ragged batches, a control input, three layers, one
    trial: every step's posterior and loss, and the final state, against the oracle stepped on the same inputs."""
    torch.manual_seed(6)
    c = dict(case)
    c.pop("T")
    T = 5
    model = vjf.VJF.make_model(c["dy"], c["dz"], c["du"], c["n"], c["hidden"], likelihood=c["lik"], lr=1e-3)
    s = load_oracle_state(model, np.float64)
    g = torch.Generator().manual_seed(10)
    B, dz = c["B"], c["dz"]
    if c["lik"] == "poisson":
        y = torch.poisson(torch.exp(0.5 * torch.randn(T, B, c["dy"], generator=g) - 0.5), generator=g)
    else:
        y = torch.randn(T, B, c["dy"], generator=g)
    u = torch.randn(T, B, c["du"], generator=g) if c["du"] else None
    eps = torch.randn(T, 2, B, dz, generator=g)
    mu_s, lv_s, loss = model.filter_sequence(y, u, None, eps=eps)
    mu, lv = None, None
    for t in range(T):
        o = orc.filter_step(s, y[t].numpy(), None if u is None else u[t].numpy(), mu, lv, eps[t, 0].numpy(), eps[t, 1].numpy())
        mu, lv = o.mu_t, o.lv_t
        close(mu_s[t], o.mu_t, rtol=1e-6, atol=1e-6)
        close(lv_s[t], o.lv_t, rtol=1e-6, atol=1e-6)
        close(loss[t], [o.loss, o.recon, o.dyn, o.entropy], rtol=5e-5, atol=5e-5)
    close(model.transition.logvar, s.tr_logvar, rtol=0, atol=5e-5)
    state_close(model, s, rtol=1e-5, atol=1e-6, rls_rtol=5e-3, rls_atol=5e-5)
    assert model.status() == 0


def test_flags_sgd_update_off(vjf):
    """sgd=False leaves the optimised tensors untouched; update=False leaves RLS state / variances untouched."""
    torch.manual_seed(2)
    model = vjf.VJF.make_model(10, 3, 0, 16, [8], likelihood="gaussian", lr=1e-2)
    g = torch.Generator().manual_seed(3)
    y, eps = torch.randn(32, 10, generator=g), torch.randn(2, 32, 3, generator=g)
    before = model._blob.clone()
    q, loss = model.filter(y, eps=(eps[0], eps[1]), sgd=False, update=False)
    assert torch.equal(before, model._blob)
    s = load_oracle_state(model)
    o = orc.filter_step(s, y.numpy(), None, None, None, eps[0].numpy(), eps[1].numpy(), sgd=False, update=False)
    close(q.mean, o.mu_t, **POST)
    close(loss, o.loss, rtol=1e-6)
    model.filter(y, eps=(eps[0], eps[1]), sgd=True, update=False)
    o = orc.filter_step(s, y.numpy(), None, None, None, eps[0].numpy(), eps[1].numpy(), sgd=True, update=False)
    state_close(model, s, rtol=2e-6, atol=1e-6)
    assert model.transition.n_sample == 0 and model.likelihood.n_sample == 0


def test_nonfinite_loss_is_flagged(vjf):
    torch.manual_seed(2)
    model = vjf.VJF.make_model(10, 3, 0, 16, [8], likelihood="poisson")
    y = torch.ones(8, 10)
    y[0, 0] = float("nan")
    g = torch.Generator().manual_seed(3)
    eps = torch.randn(2, 8, 3, generator=g)
    q, loss, recon, dyn, ent = model.filter(y, eps=(eps[0], eps[1]), verbose=True)
    st = model.status()
    assert st & 1, st                      # VJF_STATUS_NONFINITE_RECON
    assert float(recon) == 0.0             # replaced by the constant 0 (model.py:138-139)


@pytest.mark.parametrize("route", ["one_launch", "one_stream", "rls_launches", "wide", "rls_launches_two_streams", "wide_two_streams"])
@pytest.mark.parametrize("which", ["recon", "dynamics"])
def test_nonfinite_component_is_dropped_like_the_reference(vjf, which, route):
    """vjf/model.py:138-149: a loss component whose batch mean is not finite becomes the constant 0 -- the step's gradient is that
    of the OTHER components, clipped and applied as usual.  Overflow in fp32 (the reference's dtype): the oracle runs in fp32 too.
      recon:    Poisson, one decoder bias at -3e38 and counts of 2 there: -y * eta overflows, everything else stays finite;
      dynamics: w_chol scaled by 1e25: the predictive variance and with it the 'trace' term exp(logvar sums) overflow.
    The flagged step is replayed inside the launch (the middle of a sequence) and behind the last step of a call (filter()).
    Routes: the one-launch route; the per-step kernels on one stream (the backward half, the gradient sums and the SGD pass are
    launched again behind the first SGD pass and return at once on ordinary steps); the same with the multi-launch RLS (RBF(260)) and on
    the GEMM-per-layer trial path (d_y = 300, RBF(1200), hidden [400]); the last two again with the RLS update on a stream of its own
    (the default for those plans: it must not overwrite W, w_chol, sigma under the replayed backward half)."""
    import warnings
    two = route.endswith("_two_streams")
    route = route.replace("_two_streams", "")
    lik = "poisson" if which == "recon" else "gaussian"
    B, dz, dy, n, T = {"wide": (20, 6, 300, 1200, 3), "rls_launches": (40, 3, 10, 260, 4)}.get(route, (40, 3, 10, 16, 4))
    hidden = [400] if route == "wide" else [8]
    g = torch.Generator().manual_seed(31)
    y = torch.poisson(torch.exp(0.3 * torch.randn(T, B, dy, generator=g)), generator=g) if lik == "poisson" else torch.randn(T, B, dy, generator=g)
    eps = torch.randn(T, 2, B, dz, generator=g)

    def fresh():
        torch.manual_seed(30)
        m = vjf.VJF.make_model(dy, dz, 0, n, hidden, likelihood=lik, lr=1e-2)
        if route != "one_launch" and not two:
            m.set_overlap(False)
        return m

    def poison(m):
        with torch.no_grad():
            if which == "recon":
                m.decoder.decode.bias[0] = -3e38
            else:
                m.transition.velocity.w_chol.mul_(1e25)
    if which == "recon":
        y[:, :, 0] = 2.0
    bit = 1 if which == "recon" else 2
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")                     # (numpy's overflow warnings in the fp32 oracle)
        # (1) the second step of a sequence
        m = fresh()
        mu0, lv0, _ = m.filter_sequence(y[:1], eps=eps[:1])
        poison(m)
        s = load_oracle_state(m, np.float32)
        mu, lv, loss = m.filter_sequence(y[1:], qs=vjf.Gaussian(mu0[-1], lv0[-1]), eps=eps[1:])
        assert m.route() == ("one-launch" if route == "one_launch" else "two-stream" if two else "per-step")
        st = m.status()
        assert st & bit and not (st & ~0x7 & ~8), hex(st)
        om, ol = mu0[-1].cpu().numpy(), lv0[-1].cpu().numpy()
        for t in range(1, T):
            o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
            om, ol = o.mu_t, o.lv_t
            close(mu[t - 1], o.mu_t, rtol=2e-6, atol=2e-6)
            close(loss[t - 1], [o.loss, o.recon, o.dyn, o.entropy], rtol=2e-6, atol=2e-6)
            if t == 1:
                assert (o.recon if which == "recon" else o.dyn) == 0.0
        state_close(m, s, rtol=2e-5, atol=2e-6, rls_rtol=2e-3, rls_atol=2e-3)
        # (2) the only step of a call: the replay runs behind the last step
        m = fresh()
        poison(m)
        s = load_oracle_state(m, np.float32)
        w_before = m.recognition.mean.weight.clone()
        q, l1, *comp = m.filter(y[0], eps=(eps[0, 0], eps[0, 1]), verbose=True)
        assert m.status() & bit
        o = orc.filter_step(s, y[0].numpy(), None, None, None, eps[0, 0].numpy(), eps[0, 1].numpy())
        close(q.mean, o.mu_t, rtol=2e-6, atol=2e-6)
        close(torch.stack([l1, *comp]), [o.loss, o.recon, o.dyn, o.entropy], rtol=2e-6, atol=2e-6)
        state_close(m, s, rtol=2e-5, atol=2e-6, rls_rtol=2e-4, rls_atol=2e-3)
    # the other components' gradient was applied (a skipped step would leave the recognition weights where they were)
    assert (m.recognition.mean.weight - w_before).abs().max() > 1e-4


def test_refused_cooperative_launch_falls_back_to_per_step_kernels(vjf, monkeypatch):
    """When the library's residency check refuses the one-launch grid (it cannot be resident as a whole: the occupancy query times
    the compute units is less than the grid; injected here -- the name is from round 2, when the launch was a cooperative one and
    the runtime refused) the context leaves the one-launch route and the same call goes on with the per-step kernels: same results
    to summation order, status clean."""
    g = torch.Generator().manual_seed(15)
    y, eps = torch.randn(6, 96, 10, generator=g), torch.randn(6, 2, 96, 3, generator=g)
    torch.manual_seed(14)
    ref = vjf.VJF.make_model(10, 3, 0, 40, [8], likelihood="gaussian")
    mu_r, lv_r, loss_r = ref.filter_sequence(y, eps=eps)
    assert ref.route() == "one-launch"
    torch.manual_seed(14)
    m = vjf.VJF.make_model(10, 3, 0, 40, [8], likelihood="gaussian")
    monkeypatch.setenv("VJF_DEBUG_REFUSE_COOP", "1")
    mu, lv, loss = m.filter_sequence(y, eps=eps)
    monkeypatch.delenv("VJF_DEBUG_REFUSE_COOP")
    assert m.route() == "per-step" and m.check_status() == 0
    close(mu, mu_r, rtol=1e-6, atol=1e-6)
    close(loss, loss_r, rtol=1e-6, atol=1e-6)
    close(m.recognition.mean.weight, ref.recognition.mean.weight, rtol=5e-6, atol=1e-6)
    close(m.transition.velocity.w_mean, ref.transition.velocity.w_mean, rtol=1e-3, atol=1e-5)
    close(m.transition.logvar, ref.transition.logvar, rtol=0, atol=1e-6)
    q, l1 = m.filter(y[0], eps=(eps[0, 0], eps[0, 1]))            # (and single steps stay there)
    assert m.check_status() == 0


def test_timed_out_handoff_ends_the_launch_and_raises(vjf, monkeypatch):
    """A hand-off inside the one-launch route that runs out of its bound (injected: the Cholesky loop reports its statistics wait
    of step 3 as timed out): every role leaves its loop at its next wait (nothing hangs, the grid drains), the sticky status
    carries a WAIT bit and `check_status` -- what fit() calls after every sequence -- raises."""
    torch.manual_seed(8)
    model = vjf.VJF.make_model(10, 3, 0, 40, [8], likelihood="gaussian")
    g = torch.Generator().manual_seed(9)
    y, eps = torch.randn(12, 64, 10, generator=g), torch.randn(12, 2, 64, 3, generator=g)
    model.filter_sequence(y[:2], eps=eps[:2])
    assert model.route() == "one-launch" and model.check_status() == 0
    monkeypatch.setenv("VJF_DEBUG_INJECT", "4")
    model.filter_sequence(y, eps=eps)
    monkeypatch.delenv("VJF_DEBUG_INJECT")
    # the NEXT call of the context does not chain on the invalid results: it fails at once (the device told the host through the
    # pinned page; no synchronisation on the usual path) until the status has been read
    torch.cuda.synchronize()
    with pytest.raises(vjf._native.VjfError, match="timed out"):
        model.filter_sequence(y[:2], eps=eps[:2])
    with pytest.raises(vjf._native.VjfError, match="timed out"):
        model.filter(y[0], eps=(eps[0, 0], eps[0, 1]))
    with pytest.raises(RuntimeError, match="timed out"):
        model.check_status()
    assert model.status() == 0                                   # (cleared by the read)
    # ... and the context has left the route on which the wait ran out: it goes on with the per-step kernels
    assert model.route() == "per-step"
    model.filter_sequence(y[:2], eps=eps[:2])
    assert model.check_status() == 0


def test_grid_that_is_not_resident_as_a_whole_leaves_the_state_untouched(vjf, monkeypatch):
    """Another process's kernels holding compute units (injected: the grid's residency count waits for one workgroup more than the
    launch has): the launch ends within its short bound before any role has written to the state, VJF_STATUS_NOT_RESIDENT is
    raised, the next call fails instead of chaining, and after the status has been read the context runs on the per-step kernels
    from the SAME state -- the result equals a model that never tried."""
    import time
    g = torch.Generator().manual_seed(21)
    y, eps = torch.randn(4, 64, 10, generator=g), torch.randn(4, 2, 64, 3, generator=g)
    torch.manual_seed(20)
    ref = vjf.VJF.make_model(10, 3, 0, 40, [8], likelihood="gaussian")
    ref.set_overlap(False)
    torch.manual_seed(20)
    m = vjf.VJF.make_model(10, 3, 0, 40, [8], likelihood="gaussian")
    m.filter_sequence(y[:1], eps=eps[:1]); ref.filter_sequence(y[:1], eps=eps[:1])
    assert m.route() == "one-launch" and m.check_status() == 0
    before = m._blob.clone()
    monkeypatch.setenv("VJF_DEBUG_ABSENT", "1")
    t0 = time.perf_counter()
    m.filter_sequence(y[1:], eps=eps[1:])
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 2.0                         # (a quarter of a second's bound, not the 4 s of a hand-off)
    monkeypatch.delenv("VJF_DEBUG_ABSENT")
    st = m._blob.clone()
    st[m._scalars.storage_offset() + 7] = 0.0                     # (everything but the status word)
    assert torch.equal(st, before)
    with pytest.raises(vjf._native.VjfError, match="not resident"):
        m.filter_sequence(y[1:], eps=eps[1:])
    with pytest.raises(RuntimeError, match="timed out"):
        m.check_status()
    assert m.route() == "per-step"
    a = m.filter_sequence(y[1:], eps=eps[1:])
    b = ref.filter_sequence(y[1:], eps=eps[1:])
    for u_, v_ in zip(a, b):
        close(u_, v_, rtol=1e-6, atol=1e-6)
    assert m.check_status() == 0


def test_two_models_on_two_streams_interleaved(vjf):
    """Several contexts coexist (include/vjf_hip.h): two models, each on a torch stream of its own, their `filter_sequence` calls
    interleaved without a synchronisation in between.  Every call is a grid that wants every compute unit's LDS; the library
    chains them (each launch waits for the completion of the previous one of ANY context): results bitwise equal to the two
    models run one after the other, no wait gives up."""
    g = torch.Generator().manual_seed(31)
    T, B = 6, 256
    ya, ea = torch.randn(T, B, 10, generator=g).cuda(), torch.randn(T, 2, B, 3, generator=g).cuda()
    yb, eb = torch.randn(T, B, 10, generator=g).cuda(), torch.randn(T, 2, B, 3, generator=g).cuda()

    def pair():
        torch.manual_seed(30)
        ma = vjf.VJF.make_model(10, 3, 0, 40, [8], likelihood="gaussian", lr=1e-3)
        mb = vjf.VJF.make_model(10, 3, 0, 40, [8], likelihood="gaussian", lr=1e-3)
        return ma, mb
    ra, rb = pair()
    outs_ref = []
    for k in range(3):
        outs_ref.append((ra.filter_sequence(ya, eps=ea, qs=None if k == 0 else vjf.Gaussian(outs_ref[-1][0][0][-1], outs_ref[-1][0][1][-1])),
                         rb.filter_sequence(yb, eps=eb, qs=None if k == 0 else vjf.Gaussian(outs_ref[-1][1][0][-1], outs_ref[-1][1][1][-1]))))
        torch.cuda.synchronize()
    ma, mb = pair()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    for k in range(3):
        with torch.cuda.stream(sa):
            oa = ma.filter_sequence(ya, eps=ea, qs=None if k == 0 else vjf.Gaussian(outs[-1][0][0][-1], outs[-1][0][1][-1]))
        with torch.cuda.stream(sb):
            ob = mb.filter_sequence(yb, eps=eb, qs=None if k == 0 else vjf.Gaussian(outs[-1][1][0][-1], outs[-1][1][1][-1]))
        outs.append((oa, ob))
    torch.cuda.synchronize()
    assert ma.route() == "one-launch" and mb.route() == "one-launch"
    for (xa, xb), (wa, wb) in zip(outs, outs_ref):
        for u_, v_ in zip(xa + xb, wa + wb):
            assert torch.equal(u_, v_)
    assert torch.equal(ma._blob, ra._blob) and torch.equal(mb._blob, rb._blob)
    with torch.cuda.stream(sa):
        assert ma.check_status() == 0
    with torch.cuda.stream(sb):
        assert mb.check_status() == 0


def test_rls_failure_is_flagged_and_leaves_rls_state(vjf):
    """A precision matrix that is not positive definite: the reference's fallback raises (module.py:104-112); here the RLS
    tensors stay as they were, the status bit is raised and nothing hangs -- through the single-step path and through the
    sequence path, where the post kernel waits on the Cholesky kernel's column flags."""
    for n_rbf, bad_from in ((16, 0), (72, 0), (72, 40), (260, 100)):   # 1 block; 3 blocks failing in the first / second column;
                                                                        # beyond one CU's LDS (multi-launch path), failing in the 4th
        torch.manual_seed(4)
        model = vjf.VJF.make_model(10, 3, 0, n_rbf, [8], likelihood="gaussian")
        g = torch.Generator().manual_seed(5)
        T, B = 3, 64
        y, eps = torch.randn(T, B, 10, generator=g), torch.randn(T, 2, B, 3, generator=g)
        lr = model.transition.velocity
        with torch.no_grad():
            P = lr.w_precision.clone()
            P[bad_from:, bad_from:] -= 1e6 * torch.eye(n_rbf - bad_from, device=P.device)
            lr.w_precision.copy_(P)
        keep = {k: getattr(lr, k).clone() for k in ("w_mean", "w_chol", "w_pchol", "w_precision")}
        sig0 = model.transition.logvar.clone()
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model.filter(y[0], eps=(eps[0, 0], eps[0, 1]))
            assert model.status() & 8                                # VJF_STATUS_RLS_FAILED
            for k in ("w_mean", "w_chol", "w_pchol"):
                assert torch.equal(getattr(lr, k), keep[k]), k
            close(lr.w_precision, keep["w_precision"], rtol=1e-6, atol=1e-4)
            assert not torch.equal(model.transition.logvar, sig0)    # the state-noise estimate still moves (model.py:373-377)
            model.filter_sequence(y, eps=eps)
            assert model.status() & 8
            for k in ("w_mean", "w_chol", "w_pchol"):
                assert torch.equal(getattr(lr, k), keep[k]), k
            # (sequence path: the operand kernel adds G / v beside the Cholesky loop, the y / W loop takes it back on failure)
            close(lr.w_precision, keep["w_precision"], rtol=1e-6, atol=5e-4)
            assert torch.isfinite(model.transition.logvar).all()


def test_sharded_path_one_rank_nccl(vjf, monkeypatch):
    """The multi-GPU protocol (local half -> all-reduce of the reduce buffer over RCCL -> global half) with ONE rank on
    this GPU: a one-rank sum is the identity, so the trajectory must match the plain path (same kernels: bitwise)."""
    import os
    import torch.distributed as dist
    z, info, _ = gio.traj_case("g5_medium_gaussian_f32")
    m1, m2 = _model_for(vjf, info), _model_for(vjf, info)
    load_fixture_state(m1, z, "s0")
    load_fixture_state(m2, z, "s0")
    y, eps = torch.tensor(z["y"][:4]), torch.tensor(z["eps"][:4])
    m1.set_overlap(False)
    o1 = m1.filter_sequence(y, None, None, eps=eps)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    os.environ["VJF_FORCE_DIST"] = "1"
    try:
        o2 = m2.filter_sequence(y, None, None, eps=eps)
        q, loss = m2.filter(torch.tensor(z["y"][4]), eps=(torch.tensor(z["eps"][4, 0]), torch.tensor(z["eps"][4, 1])))
        torch.cuda.synchronize()
    finally:
        os.environ.pop("VJF_FORCE_DIST", None)
        dist.destroy_process_group()
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    q1, loss1 = m1.filter(torch.tensor(z["y"][4]), eps=(torch.tensor(z["eps"][4, 0]), torch.tensor(z["eps"][4, 1])))
    assert torch.equal(q.mean, q1.mean) and torch.equal(loss, loss1)
    assert torch.equal(m1._blob, m2._blob)


def test_sharded_route_fake_world_of_two(vjf, monkeypatch):
    """B_local != B_total on the in-library RCCL route with ONE GPU: VJF_DEBUG_FAKE_WORLD=2 makes the one-rank communicators
    stand for two ranks that hold the same trials (every all-reduced buffer x 2, B_total = 2 B).  That must equal an
    unsharded run on the batch written twice, [y; y]: same posterior per trial, same loss, same state (summation order)."""
    import os
    import torch.distributed as dist
    z, info, _ = gio.traj_case("g5_medium_gaussian_f32")
    m1, m2 = _model_for(vjf, info), _model_for(vjf, info)
    load_fixture_state(m1, z, "s0")
    load_fixture_state(m2, z, "s0")
    y, eps = torch.tensor(z["y"][:4]), torch.tensor(z["eps"][:4])
    o1 = m1.filter_sequence(torch.cat([y, y], 1), None, None, eps=torch.cat([eps, eps], 2))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29537")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    monkeypatch.setenv("VJF_DEBUG_FAKE_WORLD", "2")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    os.environ["VJF_FORCE_DIST"] = "1"
    try:
        o2 = m2.filter_sequence(y, None, None, eps=eps)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("VJF_FORCE_DIST", None)
        dist.destroy_process_group()
    B = y.shape[1]
    close(o1[0][:, :B], o2[0], rtol=1e-6, atol=1e-6)
    close(o1[1][:, :B], o2[1], rtol=1e-6, atol=1e-6)
    close(o1[2], o2[2], rtol=1e-6, atol=1e-6)
    close(m1._blob, m2._blob, rtol=2e-4, atol=2e-6)
    assert m1.status() == 0 and m2.status() == 0


def test_one_collective_per_step_route(vjf, monkeypatch):
    """`set_collectives(1)` (SURVEY.md 8e's layout: ONE all-reduce of the packed reduce buffer per step, inside the library): with a
    one-rank communicator it is the one-stream order of the per-step kernels bit for bit (a one-rank sum is the identity), for
    every flag set; with VJF_DEBUG_FAKE_WORLD=2 it equals the unsharded run on the batch written twice."""
    import os
    import torch.distributed as dist
    z, info, _ = gio.traj_case("g5_medium_gaussian_f32")
    y, eps = torch.tensor(z["y"][:4]), torch.tensor(z["eps"][:4])
    refs = {}
    for name, kw in (("train", {}), ("warmup", dict(warm_up=True))):
        m = _model_for(vjf, info)
        load_fixture_state(m, z, "s0")
        m.set_overlap(False)
        refs[name] = (m.filter_sequence(y, None, None, eps=eps, **kw), m._blob.clone())
    m_twice = _model_for(vjf, info)
    load_fixture_state(m_twice, z, "s0")
    o_twice = m_twice.filter_sequence(torch.cat([y, y], 1), None, None, eps=torch.cat([eps, eps], 2))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    os.environ["VJF_FORCE_DIST"] = "1"
    try:
        for name, kw in (("train", {}), ("warmup", dict(warm_up=True))):
            m = _model_for(vjf, info)
            load_fixture_state(m, z, "s0")
            m.set_collectives(1)
            o = m.filter_sequence(y, None, None, eps=eps, **kw)
            torch.cuda.synchronize()
            assert m.route(**{"warm_up": kw.get("warm_up", False)}) == "packed" and m.comm_ranks() == [1, 1]
            for a_, b_ in zip(o, refs[name][0]):
                assert torch.equal(a_, b_)
            assert torch.equal(m._blob, refs[name][1]) and m.status() == 0
            m.close()
        monkeypatch.setenv("VJF_DEBUG_FAKE_WORLD", "2")
        m2 = _model_for(vjf, info)
        load_fixture_state(m2, z, "s0")
        m2.set_collectives(1)
        o2 = m2.filter_sequence(y, None, None, eps=eps)
        torch.cuda.synchronize()
    finally:
        os.environ.pop("VJF_FORCE_DIST", None)
        dist.destroy_process_group()
    B = y.shape[1]
    close(o_twice[0][:, :B], o2[0], rtol=2e-6, atol=2e-6)
    close(o_twice[2], o2[2], rtol=2e-6, atol=2e-6)
    close(m_twice._blob, m2._blob, rtol=2e-4, atol=2e-6)
    assert m2.status() == 0


def test_nonfinite_component_on_the_sharded_route(vjf, monkeypatch):
    """vjf/model.py:138-149 where trials are sharded over ranks (the in-library RCCL route, VJF_DEBUG_FAKE_WORLD=2: two ranks holding
    the same trials): with `exact_nonfinite` the step whose dynamics term overflows is replayed -- the verdict is taken on the summed
    loss terms, the backward half runs again without that component's seeds, its gradient goes through a second sum over ranks --
    and posterior, loss terms and state follow the fp32 oracle on the batch written twice.  Without the flag the step's SGD update
    is skipped (the documented deviation), which the oracle comparison would not survive."""
    import os
    import warnings
    import torch.distributed as dist
    B, dz, dy, n, T = 48, 3, 10, 16, 4
    g = torch.Generator().manual_seed(51)
    y = torch.randn(T, B, dy, generator=g)
    eps = torch.randn(T, 2, B, dz, generator=g)
    torch.manual_seed(50)
    m = vjf.VJF.make_model(dy, dz, 0, n, [8], likelihood="gaussian", lr=1e-2)
    m.exact_nonfinite = True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29539")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    monkeypatch.setenv("VJF_DEBUG_FAKE_WORLD", "2")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    os.environ["VJF_FORCE_DIST"] = "1"
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mu0, lv0, _ = m.filter_sequence(y[:2], eps=eps[:2])
            assert m.route() == "streams"
            with torch.no_grad():
                m.transition.velocity.w_chol.mul_(1e25)           # the predictive variance, and with it the dynamics term, overflows
            s = load_oracle_state(m, np.float32)
            w_before = m.recognition.mean.weight.clone()
            mu, lv, loss = m.filter_sequence(y[2:], qs=vjf.Gaussian(mu0[-1], lv0[-1]), eps=eps[2:])
            torch.cuda.synchronize()
            st = m.status()
    finally:
        os.environ.pop("VJF_FORCE_DIST", None)
        dist.destroy_process_group()
    assert st & 2 and not (st & 0x3ff00), hex(st)                  # VJF_STATUS_NONFINITE_DYN, no time-out
    y2, e2 = torch.cat([y, y], 1), torch.cat([eps, eps], 2)       # the two "ranks" hold the same trials
    om, ol = torch.cat([mu0[-1], mu0[-1]]).cpu().numpy(), torch.cat([lv0[-1], lv0[-1]]).cpu().numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(2, T):
            o = orc.filter_step(s, y2[t].numpy(), None, om, ol, e2[t, 0].numpy(), e2[t, 1].numpy())
            om, ol = o.mu_t, o.lv_t
            close(mu[t - 2], o.mu_t[:B], rtol=2e-6, atol=2e-6)
            close(loss[t - 2], [o.loss, o.recon, o.dyn, o.entropy], rtol=2e-6, atol=2e-6)
            if t == 2:
                assert o.dyn == 0.0
    state_close(m, s, rtol=2e-5, atol=2e-6, rls_rtol=2e-4, rls_atol=2e-3)
    assert (m.recognition.mean.weight - w_before).abs().max() > 1e-4   # the other components' gradient was applied


def test_state_io_resume_is_bit_exact(vjf, tmp_path):
    """save_state after some steps, load into a fresh model, continue: identical to the uninterrupted run."""
    z, info, _ = gio.traj_case("g5_medium_gaussian_f32")
    m1, m2 = _model_for(vjf, info), _model_for(vjf, info)
    load_fixture_state(m1, z, "s0")
    y, eps = torch.tensor(z["y"]), torch.tensor(z["eps"])
    k = y.shape[0] // 2
    mu, lv, _ = m1.filter_sequence(y[:k], eps=eps[:k])
    m1.save_state(tmp_path / "s.npz")
    m2.load_state(tmp_path / "s.npz")
    q = vjf.Gaussian(mu[-1], lv[-1])
    o1 = m1.filter_sequence(y[k:], qs=q, eps=eps[k:])
    o2 = m2.filter_sequence(y[k:], qs=vjf.Gaussian(mu[-1].clone(), lv[-1].clone()), eps=eps[k:])
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    st1, st2 = m1.get_state(), m2.get_state()
    for kk in st1:
        assert np.array_equal(st1[kk], st2[kk]), kk


def test_kalman_golden(vjf):
    """LinearRegression.kalman (information form on the GPU) against the reference's Joseph-form update, fixture G7."""
    z = gio.load("g7_kalman")
    n, d = z["centroid"].shape
    blr = vjf.module.LinearRegression(vjf.module.RBF(d, n), d)
    with torch.no_grad():
        blr.feature.centroid.copy_(torch.tensor(z["centroid"], dtype=torch.float32))
        blr.feature.logwidth.copy_(torch.tensor(z["logwidth"], dtype=torch.float32))
    blr.kalman(torch.tensor(z["x"]), torch.tensor(z["t"]), 0.5, diffusion=0.01)
    close(blr.w_mean, z["W1"], rtol=1e-5, atol=1e-6)
    close(blr.w_chol, z["L1"], rtol=1e-5, atol=1e-6)
    blr.kalman(torch.tensor(z["t"]), torch.tensor(z["x"]), 0.25)
    close(blr.w_mean, z["W2"], rtol=5e-5, atol=5e-6)
    close(blr.w_chol, z["L2"], rtol=5e-5, atol=5e-6)
    # B >> n, where the reference's (samples x samples) form is O(B^3): the same update through the oracle at a size it can do
    g = torch.Generator().manual_seed(9)
    B = 600
    x, t = torch.randn(B, d, generator=g), torch.randn(B, d, generator=g)
    s = orc.OracleState(1, d, 0, n, (1,), orc.GAUSSIAN)
    s.centroid, s.logwidth = z["centroid"], z["logwidth"]
    s.w_mean, s.w_chol = blr.w_mean.cpu().numpy().astype(np.float64), blr.w_chol.cpu().numpy().astype(np.float64)
    orc.blr_kalman(s, x.numpy().astype(np.float64), t.numpy().astype(np.float64), 0.3, diffusion=0.02)
    blr.kalman(x, t, 0.3, diffusion=0.02)
    close(blr.w_mean, s.w_mean, rtol=2e-5, atol=2e-6)
    close(blr.w_chol, s.w_chol, rtol=1e-4, atol=1e-6)
    x, t = torch.randn(8192, d, generator=g), torch.randn(8192, d, generator=g)     # runs where the reference cannot
    blr.kalman(x, t, 0.3, diffusion=0.02)
    assert torch.isfinite(blr.w_mean).all() and torch.isfinite(blr.w_chol).all()
    with pytest.raises(AssertionError):
        blr.kalman(x, t, 0.3, diffusion=-1.)


def test_bad_arguments(vjf):
    model = vjf.VJF.make_model(10, 3, 2, 16, [8], likelihood="gaussian")
    with pytest.raises(TypeError):
        model.filter(torch.randn(4, 10))                 # u missing
    with pytest.raises(AssertionError):
        model.filter(torch.randn(4, 9), torch.randn(4, 2))


# ------------------------------------------------------------------ full BASELINE size: size-independent properties
def test_full_size_shard_sum_and_permutation(vjf):
    """Config B (B=4096, dz=10, dy=50, n=200, h=[128]).  (1) the two-call protocol used for multi-GPU
    -- local halves, summed reduce buffers, one global update -- gives the single-call result;
    (2) permuting trials permutes the posterior and leaves the loss and the shared state unchanged
    up to fp32 summation order; (3) a few steps match the fp64 oracle."""
    import ctypes
    from vjf_amd import _native as N
    torch.manual_seed(0)
    B, dz, dy, n, hid = 4096, 10, 50, 200, [128]
    mk = lambda: vjf.VJF.make_model(dy, dz, 0, n, hid, likelihood="gaussian")     # noqa: E731
    torch.manual_seed(0); m_full = mk()
    torch.manual_seed(0); m_half = mk()
    torch.manual_seed(0); m_perm = mk()
    g = torch.Generator().manual_seed(1)
    y = torch.randn(B, dy, generator=g).cuda()
    eps = torch.randn(2, B, dz, generator=g).cuda()
    s = load_oracle_state(m_full)
    q, loss, *comp = m_full.filter(y, eps=(eps[0], eps[1]), verbose=True)
    o = orc.filter_step(s, y.cpu().numpy(), None, None, None, eps[0].cpu().numpy(), eps[1].cpu().numpy())
    close(q.mean, o.mu_t, rtol=1e-6, atol=1e-6)
    close(torch.stack([loss, *comp]), [o.loss, o.recon, o.dyn, o.entropy], rtol=1e-6)
    state_close(m_full, s, rtol=5e-6, atol=1e-6, rls_rtol=2e-4, rls_atol=2e-6)
    # (1) shards
    m_half._ensure_ctx(B)
    L, ctx = m_half._backend(), m_half._ctx
    flags = N.FLAG_SGD | N.FLAG_UPDATE
    h = B // 2
    mu = torch.empty(B, dz, device="cuda"); lv = torch.empty(B, dz, device="cuda"); loss4 = torch.empty(4, device="cuda")
    acc = None
    for a, b in ((0, h), (h, B)):
        N.check(L.vjf_filter_local(ctx, b - a, N.ptr(y[a:b]), None, None, None, N.ptr(eps[0][a:b]), N.ptr(eps[1][a:b]),
                                   N.ptr(mu[a:b]), N.ptr(lv[a:b]), flags))
        acc = m_half._reduce.clone() if acc is None else acc + m_half._reduce
    m_half._reduce.copy_(acc)
    N.check(L.vjf_filter_global(ctx, B, N.ptr(loss4), flags))
    close(mu, q.mean, rtol=1e-6, atol=1e-6)                 # per-trial work does not depend on the shard (nor on the route)
    close(loss4[0], loss, rtol=1e-6)
    close(m_half._blob, m_full._blob, rtol=2e-4, atol=2e-6)
    # (2) permutation
    perm = torch.randperm(B, generator=g).cuda()
    qp, lossp = m_perm.filter(y[perm], eps=(eps[0][perm], eps[1][perm]))
    close(qp.mean, q.mean[perm], rtol=0, atol=0)
    close(lossp, loss, rtol=1e-6)
    close(m_perm._blob, m_full._blob, rtol=1e-4, atol=1e-6)


def test_fit_harness_golden(vjf):
    """fit(): warm-up -> decoder freeze -> RBF re-initialisation -> convergence, against G8."""
    z = gio.load("g8_fit")
    T, B, dy, dz, du, n = [int(v) for v in z["meta"][:6]]
    hid = [int(v) for v in z["meta"][6:]]
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)      # the fixture drew its noise in float64
    try:
        model = vjf.VJF.make_model(dy, dz, du, n, hid, likelihood="gaussian")
        load_fixture_state(model, z, "s0")
        torch.manual_seed(int(z["fit_seed"]))
        mu, lv, epoch_loss = model.fit(torch.tensor(z["y"]), max_iter=3, rtol=10.0)
        close(mu, z["mu"], rtol=1e-6, atol=1e-6)
        close(lv, z["lv"], rtol=1e-6, atol=1e-6)
        close(epoch_loss, z["epoch_loss"], rtol=2e-6)
        # 150 fp32 RLS steps against the fp64 fixture: the weights agree to ~1e-3 absolute
        state_close(model, z, prefix="sT", rtol=1e-5, atol=1e-6, rls_rtol=5e-3, rls_atol=1e-3)
        torch.manual_seed(int(z["fc_seed"]))
        x, yf = model.forecast(torch.tensor(z["fc_x0"]), n_step=z["fc_wnoise"].shape[0], noise=False)
        close(x, z["fc_x"], rtol=2e-4, atol=2e-4)
        close(yf, z["fc_y"], rtol=1e-4, atol=1e-4)
    finally:
        torch.set_default_dtype(old)


@pytest.mark.gpu
@pytest.mark.parametrize("n,hidden,dy", [(300, [40], 20), (1150, [400], 300)], ids=["rls_launches", "wide"])
def test_state_noise_when_the_weights_nearly_interpolate(vjf, n, hidden, dy):
    """B << n: the RLS weights reproduce dx almost exactly, so the residual of vjf/model.py:373-374 is a small difference of large
    terms.  A rank that holds every trial forms it as the reference does (dx - Phi W, then the mean of squares): the state-noise
    log-variance stays at fp32 rounding of the fp64 oracle (the quadratic form of the reduced statistics, which ranks holding
    shards use, is at 1.5e-5 here)."""
    import warnings
    B, dz, T = 4, 3, 3
    torch.manual_seed(5)
    m = vjf.VJF.make_model(dy, dz, 0, n, hidden, likelihood="gaussian", lr=1e-3)
    s = load_oracle_state(m, np.float64)
    g = torch.Generator().manual_seed(6)
    y = torch.randn(T, B, dy, generator=g)
    eps = torch.randn(T, 2, B, dz, generator=g)
    q = mu = lv = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(T):
            o = orc.filter_step(s, y[t].numpy(), None, mu, lv, eps[t, 0].numpy(), eps[t, 1].numpy())
            mu, lv = o.mu_t, o.lv_t
            q, _ = m.filter(y[t], None, q, eps=(eps[t, 0], eps[t, 1]))
            assert abs(float(m.transition.logvar) - float(s.tr_logvar)) < 4e-6, t
    assert m.status() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("n,hidden,dy", [(300, [40], 20), (1150, [400], 300)], ids=["rls_launches", "wide"])
def test_split_entry_points_on_the_multi_launch_rls_routes(vjf, monkeypatch, n, hidden, dy):
    """vjf_filter_local / vjf_filter_global as a rank holding a shard calls them (here: the only rank, the sum over ranks is the
    identity), on the plans whose RLS update is a sequence of launches: the state-noise update then comes from the reduced
    statistics (the quadratic form), everything else from the same kernels as the step entry point."""
    import warnings
    B, dz, T = 48, 4, 3
    torch.manual_seed(7)
    m = vjf.VJF.make_model(dy, dz, 0, n, hidden, likelihood="gaussian", lr=1e-3)
    m2 = vjf.VJF.make_model(dy, dz, 0, n, hidden, likelihood="gaussian", lr=1e-3)
    m2.set_state(m.get_state())
    s = load_oracle_state(m, np.float64)
    g = torch.Generator().manual_seed(8)
    y = torch.randn(T, B, dy, generator=g)
    eps = torch.randn(T, 2, B, dz, generator=g)
    monkeypatch.setattr(type(m), "_world", staticmethod(lambda: (1, True)))
    monkeypatch.setattr(type(m), "_all_reduce_sum", staticmethod(lambda t: None))
    q = mu = lv = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(T):
            o = orc.filter_step(s, y[t].numpy(), None, mu, lv, eps[t, 0].numpy(), eps[t, 1].numpy())
            mu, lv = o.mu_t, o.lv_t
            q, l1, *comp = m.filter(y[t], None, q, verbose=True, eps=(eps[t, 0], eps[t, 1]))
            close(q.mean, o.mu_t, rtol=1e-6, atol=1e-6)
            close(q.logvar, o.lv_t, rtol=1e-6, atol=1e-6)
            close(torch.stack([l1, *comp]), [o.loss, o.recon, o.dyn, o.entropy], rtol=5e-6, atol=5e-6)
    state_close(m, s, rtol=5e-6, atol=1e-6, rls_rtol=5e-3, rls_atol=5e-5)
    assert m.status() == 0
    monkeypatch.undo()
    q = None
    for t in range(T):                                   # the step entry point: same kernels but the state-noise update's
        q, _ = m2.filter(y[t], None, q, eps=(eps[t, 0], eps[t, 1]))
    close(m2.transition.logvar, m.transition.logvar, rtol=0, atol=1e-6)
    close(m2.transition.velocity.w_mean, m.transition.velocity.w_mean, rtol=1e-3, atol=1e-4)     # (B < n: conditioning)


@pytest.mark.gpu
@pytest.mark.parametrize("n,hidden,dy,B", [(300, [40], 20, 48), (1150, [400], 300, 48), (260, [8], 10, 700)], ids=["rls_launches", "wide", "rls_launches_B700"])
def test_rls_update_on_its_own_stream_gives_the_one_stream_bits(vjf, monkeypatch, n, hidden, dy, B):
    """Plans whose RLS update is a sequence of launches (n_rbf > 224): `filter_sequence` runs that update on a second stream beside
    the backward half of its step and the forward half of the next (route 'two-stream'); `set_overlap(False)` keeps the one-stream
    order.  Same kernels, same operands: every output and the whole state blob must agree bit for bit -- also from a workspace of
    NaNs, also across two calls (the second starts from the first one's state), also with an RLS failure in the middle (a
    precision matrix made indefinite between the calls: the update is dropped, the status bit raised, the sequence goes on)."""
    g = torch.Generator().manual_seed(91)
    T, dz = 5, 4
    y, eps = torch.randn(2 * T, B, dy, generator=g), torch.randn(2 * T, 2, B, dz, generator=g)
    outs = []
    for overlap, pattern in ((1, "0xFF"), (0, "0x00")):
        monkeypatch.setenv("VJF_DEBUG_POISON_WS", pattern)
        torch.manual_seed(9)
        m = vjf.VJF.make_model(dy, dz, 0, n, hidden, likelihood="gaussian", lr=1e-2)
        m.set_overlap(overlap)
        mu, lv, ls = m.filter_sequence(y[:T].cuda(), None, None, eps=eps[:T].cuda())
        assert m.route() == ("two-stream" if overlap else "per-step")
        assert m.check_status() == 0
        blob1 = m._blob.cpu().numpy().copy()
        lr = m.transition.velocity
        with torch.no_grad():
            P = lr.w_precision.clone()
            P[n - 40:, n - 40:] -= 1e7 * torch.eye(40, device=P.device)
            lr.w_precision.copy_(P)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mu2, lv2, ls2 = m.filter_sequence(y[T:].cuda(), None, vjf.Gaussian(mu[-1], lv[-1]), eps=eps[T:].cuda())
        assert m.status() & 8
        outs.append((mu.cpu().numpy(), lv.cpu().numpy(), ls.cpu().numpy(), blob1, mu2.cpu().numpy(), ls2.cpu().numpy(), m._blob.cpu().numpy().copy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [1, 0, 3], ids=["one_launch", "one_stream", "three_streams"])
def test_results_do_not_depend_on_workspace_contents(vjf, monkeypatch, overlap):
    """The caller's workspace arrives uninitialised (torch.empty).  A workspace of NaNs (every byte 0xFF, also in every counter
    word) must give the bits a workspace of zeros gives: nothing is read before the library has written it."""
    g = torch.Generator().manual_seed(77)
    T, B = 4, 100
    y, u, eps = torch.randn(T, B, 10, generator=g), torch.randn(T, B, 2, generator=g), torch.randn(T, 2, B, 3, generator=g)
    outs = []
    for pattern in ("0x00", "0xFF"):
        monkeypatch.setenv("VJF_DEBUG_POISON_WS", pattern)
        torch.manual_seed(8)
        m = vjf.VJF.make_model(10, 3, 2, 20, [16], likelihood="gaussian", lr=1e-2)
        m.set_overlap(overlap)
        mu, lv, ls = m.filter_sequence(y.cuda(), u.cuda(), None, eps=eps.cuda())
        assert m.check_status() == 0
        outs.append((mu.cpu().numpy(), ls.cpu().numpy(), m._blob.cpu().numpy().copy()))
    for a, b in zip(outs[0], outs[1]):
        assert np.isfinite(a).all()
        np.testing.assert_array_equal(a, b)


@pytest.mark.gpu
def test_gemm_dispatcher_shapes_against_fp64(vjf):
    """The GEMM kernels of the wide route behind `vjf_linear_forward` (out = x W^T + b through the C ABI), over the shapes that
    pick each of them and their edges: narrow outputs (K-split 32 x 32 tiles), 128 x 64 and 128 x 128 tiles with two LDS images,
    the 64 x 64 kernel for small / unaligned operands; M, N, K off the tile sizes; K not a multiple of 4 (float-by-float path)."""
    from vjf_amd import _native as N
    from vjf_amd.util import stream_ptr
    g = torch.Generator().manual_seed(123)
    shapes = [(300, 640, 512), (257, 96, 65), (4096, 64, 512), (1000, 1000, 64), (1024, 1000, 1000), (513, 200, 300),
              (33, 7, 5), (100, 37, 64), (256, 130, 256), (2000, 33, 129), (16, 512, 512), (384, 1000, 40), (4096, 10, 50),
              # >= 192 tiles of 128 x 128: vjf_wide_gemm3_kernel<128, 16, 4, false, true> (configs[4]'s layers at B = 4096); rows,
              # columns and depth off the tile and chunk sizes
              (4096, 1000, 1000), (4096, 640, 1024), (4096, 512, 1000), (3100, 516, 900)]
    for (B, din, dout) in shapes:
        x = torch.randn(B, din, generator=g)
        W = torch.randn(dout, din, generator=g) / din ** 0.5
        b = torch.randn(dout, generator=g)
        xd, Wd, bd = x.cuda(), W.cuda(), b.cuda()
        out = torch.empty(B, dout, device="cuda")
        N.check(N.lib().vjf_linear_forward(N.ptr(xd), N.ptr(Wd), N.ptr(bd), N.ptr(out), B, din, dout, stream_ptr()), "vjf_linear_forward")
        ref = (x.double() @ W.double().T + b.double())
        err = (out.cpu().double() - ref).abs().max().item()
        scale = (x.double().abs() @ W.double().abs().T).max().item()
        assert err <= 4e-7 * scale + 1e-6, (B, din, dout, err, scale)
        out2 = torch.empty(B, dout, device="cuda")                         # no bias
        N.check(N.lib().vjf_linear_forward(N.ptr(xd), N.ptr(Wd), None, N.ptr(out2), B, din, dout, stream_ptr()), "vjf_linear_forward")
        assert ((out2.cpu().double() - (ref - b.double())).abs().max().item()) <= 4e-7 * scale + 1e-6, (B, din, dout)


@pytest.mark.gpu
def test_large_k_major_product_against_fp64(vjf):
    """The 128 x 128-tile GEMM with a k-major B operand (`vjf_wide_gemm3_kernel<128, 16, 4, false, false>`: Z = Phi w_chol of
    configs[4] at B = 4096) on its own, through `vjf_blr_sample`: w = w_mean + w_chol @ noise (vjf/module.py:71) is an
    (n, n) x (n, dout) product with n = 2048 rows and dout = 1536 columns = 192 tiles; the sample Phi w against fp64."""
    g = torch.Generator().manual_seed(77)
    n, d, dout, B = 2048, 6, 1536, 40
    torch.manual_seed(n)
    blr = vjf.module.LinearRegression(vjf.module.RBF(d, n), dout)
    blr.w_mean.copy_(torch.randn(n, dout, generator=g) * 0.3)
    blr.w_chol.copy_(torch.triu(torch.randn(n, n, generator=g)) * 0.05)
    x = torch.randn(B, d, generator=g)
    noise = torch.randn(n, dout, generator=g)
    smp = blr(x, sampling=True, noise=noise)
    c, lw = blr.feature.centroid.cpu().double(), blr.feature.logwidth.cpu().double()
    phi = torch.exp(-0.5 * ((x.double()[:, None, :] - c[None]) ** 2).sum(-1) / torch.exp(lw).reshape(1, -1) ** 2)
    w = blr.w_mean.cpu().double() + blr.w_chol.cpu().double() @ noise.double()
    ref = phi @ w
    scale = (phi.abs() @ (blr.w_mean.cpu().double().abs() + blr.w_chol.cpu().double().abs() @ noise.double().abs())).max().item()
    assert (smp.cpu().double() - ref).abs().max().item() <= 1e-6 * scale + 1e-6


@pytest.mark.gpu
def test_last_step_without_a_gradient_keeps_the_earlier_steps(vjf):
    """A sequence whose LAST step has no finite loss component (a NaN observation: the reconstruction term, and through the
    posterior of that trial the dynamics term and the entropy): the reference skips that step's optimizer.step() alone
    (vjf/model.py:206-214) -- the SGD steps before it stay.  On the one-launch route the SGD role keeps its first round of
    parameters in registers and writes the state blob at the last step of the launch: also when that step has no gradient.
    The trained tensors after T steps against the fp32 oracle after the T - 1 steps that had a gradient."""
    import warnings
    from tests.helpers import model_arrays
    B, dz, dy, n, T = 40, 3, 10, 16, 4
    g = torch.Generator().manual_seed(41)
    y = torch.randn(T, B, dy, generator=g)
    eps = torch.randn(T, 2, B, dz, generator=g)
    y[T - 1, 0, 0] = float("nan")
    torch.manual_seed(40)
    m = vjf.VJF.make_model(dy, dz, 0, n, [8], likelihood="gaussian", lr=1e-2)
    s = load_oracle_state(m, np.float32)
    w0 = m.recognition.mean.weight.clone()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mu, lv, loss = m.filter_sequence(y, eps=eps)
        assert m.route() == "one-launch"
        st = m.status()
        assert (st & 7) == 7 and not (st & 0x3ff00), hex(st)
        om = ol = None
        for t in range(T - 1):
            o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
            om, ol = o.mu_t, o.lv_t
            close(mu[t], o.mu_t, rtol=2e-6, atol=2e-6)
            close(loss[t], [o.loss, o.recon, o.dyn, o.entropy], rtol=2e-6, atol=2e-6)
    assert float(loss[T - 1, 0]) == 0.0                               # every component replaced by the constant 0 (model.py:138-145)
    assert (m.recognition.mean.weight - w0).abs().max() > 1e-4       # the steps before it are in the blob
    got = model_arrays(m)
    want = {"mean_W": s.mean_W, "lv_W": s.lv_W, "lv_b": s.lv_b, "dec_W": s.dec_W, "dec_b": s.dec_b, "rec_W0": s.rec_W[0], "rec_b0": s.rec_b[0]}
    for k, v in want.items():
        close(got[k], np.asarray(v).reshape(got[k].shape), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
def test_operator_kernels_odd_shapes_against_fp64(vjf):
    """The matrix-core operator kernels behind `Recognition.forward` and `LinearRegression.forward` (prediction and sampling) and
    the tiled `functional.rbf`, on shapes off every tile size (batch not a multiple of 16, widths not multiples of 16 / 4,
    three layers, latent width beyond one tile), against fp64 torch."""
    g = torch.Generator().manual_seed(321)
    for (B, dy, du, dz, hidden) in [(37, 11, 2, 5, [19, 7, 33]), (16, 64, 0, 20, [48]), (1, 3, 1, 2, [5]), (130, 200, 0, 10, [128])]:
        torch.manual_seed(B)
        rec = vjf.recognition.Recognition(dy, dz, du, hidden)
        y, u = torch.randn(B, dy, generator=g), torch.randn(B, du, generator=g) if du else None
        mu, lv = torch.randn(B, dz, generator=g), torch.randn(B, dz, generator=g)
        q = rec(y, vjf.Gaussian(mu, lv), u)
        h = torch.cat([t for t in (y, u, mu, lv) if t is not None], -1).double()
        for lin in rec.linears():
            h = torch.tanh(h @ lin.weight.cpu().double().T + lin.bias.cpu().double())
        close(q.mean, (h @ rec.mean.weight.cpu().double().T).float(), rtol=1e-5, atol=1e-6)
        close(q.logvar, (h @ rec.logvar.weight.cpu().double().T + rec.logvar.bias.cpu().double()).float(), rtol=1e-5, atol=1e-6)
    for (B, d, n, dout) in [(37, 5, 50, 3), (16, 12, 200, 10), (3, 2, 7, 1), (100, 7, 333, 64)]:    # (larger d: the fp32 features underflow)
        torch.manual_seed(n)
        blr = vjf.module.LinearRegression(vjf.module.RBF(d, n), dout)
        blr.w_mean.copy_(torch.randn(n, dout, generator=g) * 0.3)
        blr.w_chol.copy_(torch.triu(torch.randn(n, n, generator=g)) * 0.1)
        x = torch.randn(B, d, generator=g)
        c, lw = blr.feature.centroid.cpu().double(), blr.feature.logwidth.cpu().double()
        phi = torch.exp(-0.5 * ((x.double()[:, None, :] - c[None]) ** 2).sum(-1) / torch.exp(lw).reshape(1, -1) ** 2)
        close(vjf.functional.rbf(x, blr.feature.centroid, torch.exp(blr.feature.logwidth)), phi.float(), rtol=5e-6, atol=1e-7)   # (takes the WIDTH)
        p = blr(x, sampling=False)
        close(p.mean, (phi @ blr.w_mean.cpu().double()).float(), rtol=1e-5, atol=1e-6)
        z = phi @ blr.w_chol.cpu().double()
        close(p.logvar, torch.log((z * z).sum(-1, keepdim=True)).expand(B, dout).float(), rtol=1e-6, atol=1e-6)
        noise = torch.randn(n, dout, generator=g)
        smp = blr(x, sampling=True, noise=noise)
        w = blr.w_mean.cpu().double() + blr.w_chol.cpu().double() @ noise.double()
        close(smp, (phi @ w).float(), rtol=5e-5, atol=5e-6)


# ------------------------------------------------------------------ the one-launch route for the other flag sets of VJF.filter
FLAG_SETS = {"warmup": dict(sgd=True, update=True, warm_up=True), "infer": dict(sgd=False, update=False, warm_up=False),
             "sgd-only": dict(sgd=True, update=False, warm_up=False), "infer-warm": dict(sgd=False, update=True, warm_up=True)}


@pytest.mark.parametrize("flags", sorted(FLAG_SETS))
@pytest.mark.parametrize("name", ["g5_gaussian_du2_wu1_f32", "g5_poisson_du0_wu1_f32", "g5_medium_gaussian_f32"])
def test_flag_sets_on_the_one_launch_route(vjf, name, flags):
    """warm_up=True (the first epochs of fit: vjf/model.py:243-259), sgd=False / update=False (a deployed filter: model.py:180, 206,
    215) run as ONE launch of trial + SGD roles (vjf_mega_lite_kernel): (1) the route says so; (2) `filter_sequence` == stepwise
    `filter`, bit for bit, outputs and blob; (3) against the per-step kernels of the one-stream order; (4) every step against the
    fp64 oracle; (5) a second call continues from the first one's posterior."""
    kw = FLAG_SETS[flags]
    z, info, _ = gio.traj_case(name)
    m1, m2, m3 = (_model_for(vjf, info, lr=1e-3) for _ in range(3))
    for m in (m1, m2, m3):
        load_fixture_state(m, z, "s0")
    m3.set_overlap(False)
    s = load_oracle_state(m1, np.float64)
    u = torch.tensor(z["u"]) if info["du"] else None
    y, eps = torch.tensor(z["y"]), torch.tensor(z["eps"])
    T, k = info["T"], info["T"] // 2
    mu_a, lv_a, loss_a = m1.filter_sequence(y[:k], None if u is None else u[:k], None, eps=eps[:k], **kw)
    assert m1.route(**kw) == "one-launch"
    mu_b, lv_b, loss_b = m1.filter_sequence(y[k:], None if u is None else u[k:], vjf.Gaussian(mu_a[-1], lv_a[-1]), eps=eps[k:], **kw)
    mu, lv, loss = torch.cat([mu_a, mu_b]), torch.cat([lv_a, lv_b]), torch.cat([loss_a, loss_b])
    o3 = m3.filter_sequence(y, u, None, eps=eps, **kw)
    assert m3.route(**kw) == "per-step"
    q = None
    om = ol = None
    for t in range(T):
        q, l, *c = m2.filter(y[t], None if u is None else u[t], q, verbose=True, eps=(eps[t, 0], eps[t, 1]), **kw)
        assert torch.equal(q.mean, mu[t]) and torch.equal(q.logvar, lv[t])
        assert torch.equal(torch.stack([l, *c]), loss[t])
        o = orc.filter_step(s, z["y"][t].astype(np.float64), None if u is None else z["u"][t].astype(np.float64), om, ol,
                            z["eps"][t, 0].astype(np.float64), z["eps"][t, 1].astype(np.float64), **kw)
        om, ol = o.mu_t, o.lv_t
        close(mu[t], o.mu_t, **POST)
        close(lv[t], o.lv_t, **POST)
        close(loss[t], [o.loss, o.recon, o.dyn, o.entropy], rtol=5e-6, atol=5e-6)
    assert torch.equal(m1._blob, m2._blob)
    for a, b in zip((mu, lv, loss), o3):
        close(a, b, rtol=2e-6, atol=2e-6)
    close(m1._blob, m3._blob, rtol=2e-5, atol=1e-6)
    state_close(m1, s, rtol=2e-6, atol=1e-6, rls_rtol=2e-5)
    if not kw["update"]:
        assert m1.transition.n_sample == int(z["s0.n_tr"])
    assert m1.status() == 0 and m2.status() == 0 and m3.status() == 0


def test_update_without_sgd_stays_on_the_per_step_kernels(vjf):
    """sgd=False, update=True, no warm-up (nobody's usual call) is the one flag set the one-launch route leaves to the per-step
    kernels; it still matches the oracle."""
    z, info, _ = gio.traj_case("g5_gaussian_du0_wu0_f32")
    m = _model_for(vjf, info)
    load_fixture_state(m, z, "s0")
    s = load_oracle_state(m, np.float64)
    kw = dict(sgd=False, update=True, warm_up=False)
    mu, lv, loss = m.filter_sequence(torch.tensor(z["y"]), None, None, eps=torch.tensor(z["eps"]), **kw)
    assert m.route(**kw) == "per-step"
    ro = orc.filter_sequence(s, z["y"].astype(np.float64), None, z["eps"].astype(np.float64), **kw)
    close(mu, ro[0], **POST)
    state_close(m, s, rtol=2e-6, atol=1e-6, rls_rtol=5e-4, rls_atol=5e-6)


@pytest.mark.parametrize("flags", ["warmup", "infer"])
def test_flag_sets_at_bench_size(vjf, flags):
    """`bench.py --flags warmup|infer` at its size (B = 4096, config B): more steps than the ring of loss sums holds (the launch
    without parameter updates has no gate between its steps), against the fp64 oracle on the first steps, bitwise against two chunks."""
    kw = FLAG_SETS[flags]
    torch.manual_seed(5)
    m1 = vjf.VJF.make_model(50, 10, 0, 200, [128], likelihood="gaussian", lr=1e-3)
    torch.manual_seed(5)
    m2 = vjf.VJF.make_model(50, 10, 0, 200, [128], likelihood="gaussian", lr=1e-3)
    s = load_oracle_state(m1, np.float64)
    g = torch.Generator().manual_seed(9)
    T, B = 80, 4096
    y, eps = torch.randn(T, B, 50, generator=g).cuda(), torch.randn(T, 2, B, 10, generator=g).cuda()
    mu, lv, loss = m1.filter_sequence(y, eps=eps, **kw)
    assert m1.route(**kw) == "one-launch" and m1.status() == 0
    a = m2.filter_sequence(y[:47], eps=eps[:47], **kw)
    b = m2.filter_sequence(y[47:], qs=vjf.Gaussian(a[0][-1], a[1][-1]), eps=eps[47:], **kw)
    assert torch.equal(torch.cat([a[0], b[0]]), mu) and torch.equal(torch.cat([a[2], b[2]]), loss) and torch.equal(m1._blob, m2._blob)
    om = ol = None
    yc, ec = y[:6].cpu().numpy().astype(np.float64), eps[:6].cpu().numpy().astype(np.float64)
    for t in range(6):
        o = orc.filter_step(s, yc[t], None, om, ol, ec[t, 0], ec[t, 1], **kw)
        om, ol = o.mu_t, o.lv_t
        close(mu[t], o.mu_t, rtol=1e-6, atol=1e-6)
        close(loss[t], [o.loss, o.recon, o.dyn, o.entropy], rtol=1e-6, atol=1e-6)
    assert torch.isfinite(loss).all()


@pytest.mark.parametrize("shape", ["full", "upper"])
@pytest.mark.parametrize("flags", ["infer", "sgd-only"])
def test_launch_without_rls_update_looks_at_w_chol_itself(vjf, flags, shape):
    """A launch without an RLS update has `w_chol` as a constant and no triangle flag to go by when the state was just loaded (the flag
    is an RLS update's to set): the trial workgroups look at the matrix while they transpose it.  A FULL `w_chol` (what `kalman`
    leaves, module.py:140-142) must take the square product, an upper triangular one the triangle -- both against the fp64 oracle,
    and the one-launch route against the per-step kernels, which go by the flag alone."""
    kw = FLAG_SETS[flags]
    z, info, _ = gio.traj_case("g5_medium_gaussian_f32")
    m1, m3 = (_model_for(vjf, info, lr=1e-3) for _ in range(2))
    for m in (m1, m3):
        load_fixture_state(m, z, "s0")
    st = m1.get_state()
    n = st["transition.velocity.w_chol"].shape[0]
    g = np.random.default_rng(3)
    w = 0.05 * g.standard_normal((n, n)).astype(np.float32) + 0.3 * np.eye(n, dtype=np.float32)
    st["transition.velocity.w_chol"] = np.triu(w) if shape == "upper" else w
    for m in (m1, m3):
        m.set_state(st)
    m3.set_overlap(False)
    s = load_oracle_state(m1, np.float64)
    y, eps = torch.tensor(z["y"]), torch.tensor(z["eps"])
    mu, lv, loss = m1.filter_sequence(y, None, None, eps=eps, **kw)
    assert m1.route(**kw) == "one-launch"
    o3 = m3.filter_sequence(y, None, None, eps=eps, **kw)
    ro = orc.filter_sequence(s, z["y"].astype(np.float64), None, z["eps"].astype(np.float64), **kw)
    close(mu, ro[0], **POST)
    close(lv, ro[1], **POST)
    close(loss, ro[2], rtol=5e-6, atol=5e-6)
    for a, b in zip((mu, lv, loss), o3):
        close(a, b, rtol=2e-6, atol=2e-6)
    assert m1.status() == 0 and m3.status() == 0
