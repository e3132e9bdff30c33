"""BASELINE.json configs at (or near) their full sizes on the GPU, through the entry points the bench times.

  * configs[1] (B): `filter_sequence` at B = 4096 for T = 50 steps -- the schedule and the size `bench.py` runs -- bitwise
    against stepwise `filter`, and every step against the fp64 oracle (vjf/model.py:179-221);
  * configs[4] (E): n_rbf = 1000 (32 block columns of the multi-launch RLS), d_z = 64, d_y = 512, hidden [512, 512];
  * configs[3] (D): B = 32768 on one GPU, and the 8-shard sum property through vjf_filter_local / vjf_filter_global;
  * configs[2] (C): Poisson, d_y = 200 at B = 4096 through `filter_sequence`.
Tolerances as in tests/test_gpu_parity.py (fp32 device path against the fp64 oracle)."""
import numpy as np
import pytest
import torch

from oracle import vjf_oracle as orc
from tests.margins import check_close
from tests.helpers import load_oracle_state, state_close

pytestmark = pytest.mark.gpu


def close(a, b, **kw):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    b = b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else b
    check_close(np.asarray(a, np.float64), np.asarray(b, np.float64), **kw)      # (asserts, and records the achieved margin)


@pytest.fixture(scope="module")
def vjf():
    import vjf_amd
    assert torch.cuda.is_available()
    return vjf_amd


def _data(c, T, seed):
    g = torch.Generator().manual_seed(seed)
    if c["lik"] == "poisson":
        y = torch.poisson(torch.exp(0.5 * torch.randn(T, c["B"], c["dy"], generator=g) - 0.5), generator=g)
    else:
        y = torch.randn(T, c["B"], c["dy"], generator=g)
    eps = torch.randn(T, 2, c["B"], c["dz"], generator=g)
    return y, eps


def _model(vjf, c, lr=1e-3):
    m = vjf.VJF.make_model(c["dy"], c["dz"], 0, c["n"], c["hidden"], likelihood=c["lik"], lr=lr)
    if c["dz"] >= 32:      # the default RBF init underflows every feature at d_z = 64: SURVEY 8d's init
        r = float(np.sqrt(c["dz"]))
        m.transition.velocity.feature.centroid.uniform_(-r, r)
        m.transition.velocity.feature.logwidth.fill_(float(np.log(r)))
    return m


CFG_B = dict(B=4096, dz=10, dy=50, n=200, hidden=[128], lik="gaussian")
CFG_C = dict(B=4096, dz=10, dy=200, n=200, hidden=[128], lik="poisson")
CFG_D1 = dict(B=32768, dz=10, dy=50, n=200, hidden=[128], lik="gaussian")
CFG_E = dict(B=64, dz=64, dy=512, n=1000, hidden=[512, 512], lik="gaussian")


@pytest.mark.parametrize("cfg", [CFG_B, CFG_C], ids=["configB", "configC"])
def test_sequence_at_bench_size_bitwise_and_oracle(vjf, cfg):
    """The benchmarked entry point at the benchmarked size (256 trial tiles, every hand-off of the schedule under its real
    workgroup count): (1) `filter_sequence` == stepwise `filter`, bit for bit, outputs and state blob; (2) two chunks == one
    piece; (3) every step's posterior and loss terms, and the final state, against the fp64 oracle."""
    T = 50 if cfg is CFG_B else 12
    torch.manual_seed(11)
    m_seq = _model(vjf, cfg)
    torch.manual_seed(11)
    m_stp = _model(vjf, cfg)
    torch.manual_seed(11)
    m_chk = _model(vjf, cfg)
    s = load_oracle_state(m_seq, np.float64)
    y, eps = _data(cfg, T, 21)
    yd, ed = y.cuda(), eps.cuda()
    mu, lv, loss = m_seq.filter_sequence(yd, eps=ed)
    assert m_seq.status() == 0
    k = T // 2 + 1
    mu1, lv1, l1 = m_chk.filter_sequence(yd[:k], eps=ed[:k])
    mu2, lv2, l2 = m_chk.filter_sequence(yd[k:], qs=vjf.Gaussian(mu1[-1], lv1[-1]), eps=ed[k:])
    assert torch.equal(torch.cat([mu1, mu2]), mu) and torch.equal(torch.cat([lv1, lv2]), lv) and torch.equal(torch.cat([l1, l2]), loss)
    assert torch.equal(m_chk._blob, m_seq._blob)
    q = None
    nstep = min(T, 12)                                          # (stepwise calls: a prefix is enough for the bitwise claim)
    for t in range(nstep):
        q, l, *c = m_stp.filter(yd[t], None, q, verbose=True, eps=(ed[t, 0], ed[t, 1]))
        assert torch.equal(q.mean, mu[t]) and torch.equal(q.logvar, lv[t]), t
        assert torch.equal(torch.stack([l, *c]), loss[t]), t
    om, ol = None, None
    for t in range(T):
        o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
        om, ol = o.mu_t, o.lv_t
        close(mu[t], o.mu_t, rtol=1e-6, atol=1e-6)
        close(lv[t], o.lv_t, rtol=1e-6, atol=1e-6)
        close(loss[t], [o.loss, o.recon, o.dyn, o.entropy], rtol=1e-6, atol=1e-6)
    close(m_seq.transition.logvar, s.tr_logvar, rtol=0, atol=1e-6)
    state_close(m_seq, s, rtol=5e-6, atol=1e-6, rls_rtol=5e-3, rls_atol=5e-5)


def test_config_E_full_width(vjf):
    """configs[4]: RBF(1000) = 32 block columns of the chip-wide RLS, d_z = 64, d_y = 512, hidden [512, 512]; 2 steps and a
    2-step sequence against the oracle, then a precision matrix that fails in block column 20 (state left as it was)."""
    c = CFG_E
    torch.manual_seed(12)
    m = _model(vjf, c)
    s = load_oracle_state(m, np.float64)
    T = 4
    y, eps = _data(c, T, 22)
    q, om, ol = None, None, None
    for t in range(2):
        q, loss, *comp = m.filter(y[t], None, q, verbose=True, eps=(eps[t, 0], eps[t, 1]))
        o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
        om, ol = o.mu_t, o.lv_t
        close(q.mean, o.mu_t, rtol=1e-6, atol=1e-6)
        close(q.logvar, o.lv_t, rtol=1e-6, atol=1e-6)
        close(torch.stack([loss, *comp]), [o.loss, o.recon, o.dyn, o.entropy], rtol=1e-6, atol=1e-6)
    mu, lv, ls = m.filter_sequence(y[2:], qs=q, eps=eps[2:])
    for t in range(2, T):
        o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
        om, ol = o.mu_t, o.lv_t
        close(mu[t - 2], o.mu_t, rtol=1e-6, atol=1e-6)
        close(ls[t - 2], [o.loss, o.recon, o.dyn, o.entropy], rtol=1e-6, atol=1e-6)
    state_close(m, s, rtol=5e-6, atol=1e-6, rls_rtol=5e-5, rls_atol=5e-5)
    assert m.status() == 0
    # RLS failure at 32 block columns
    lr = m.transition.velocity
    n = c["n"]
    with torch.no_grad():
        P = lr.w_precision.clone()
        P[640:, 640:] -= 1e7 * torch.eye(n - 640, device=P.device)
        lr.w_precision.copy_(P)
    keep = {k: getattr(lr, k).clone() for k in ("w_mean", "w_chol", "w_pchol")}
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.filter(y[0], eps=(eps[0, 0], eps[0, 1]))
    assert m.status() & 8
    for k2 in keep:
        assert torch.equal(getattr(lr, k2), keep[k2]), k2


def test_config_E_at_bench_size(vjf):
    """configs[4] at the size `bench.py --config E` times: B = 4096 trials, so that the dispatcher (vjf_abi.hip, launch_wide_gemm) takes
    the 128 x 128-tile kernel `vjf_wide_gemm3_kernel<128, 16, 4, ...>` -- both operand layouts: W^T as torch stores it for the 512-wide
    layers (`nt`), k-major for Z = Phi w_chol (4096 x 1000 x 1000) -- the kernels that carry the benchmarked step.  One `filter` step
    and a two-step sequence against the fp64 oracle (posterior, loss terms, state)."""
    c = dict(CFG_E, B=4096)
    torch.manual_seed(14)
    m = _model(vjf, c)
    s = load_oracle_state(m, np.float64)
    T = 3
    y, eps = _data(c, T, 24)
    yd, ed = y.cuda(), eps.cuda()
    q, loss, *comp = m.filter(yd[0], None, None, verbose=True, eps=(ed[0, 0], ed[0, 1]))
    o = orc.filter_step(s, y[0].numpy(), None, None, None, eps[0, 0].numpy(), eps[0, 1].numpy())
    close(q.mean, o.mu_t, rtol=1e-6, atol=1e-6)
    close(q.logvar, o.lv_t, rtol=1e-6, atol=1e-6)
    close(torch.stack([loss, *comp]), [o.loss, o.recon, o.dyn, o.entropy], rtol=1e-6, atol=1e-6)
    mu, lv, ls = m.filter_sequence(yd[1:], qs=q, eps=ed[1:])
    om, ol = o.mu_t, o.lv_t
    for t in range(1, T):
        o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
        om, ol = o.mu_t, o.lv_t
        close(mu[t - 1], o.mu_t, rtol=2e-6, atol=2e-6)
        close(lv[t - 1], o.lv_t, rtol=2e-6, atol=2e-6)
        close(ls[t - 1], [o.loss, o.recon, o.dyn, o.entropy], rtol=1e-6, atol=1e-6)
    close(m.transition.logvar, s.tr_logvar, rtol=0, atol=1e-6)
    state_close(m, s, rtol=5e-6, atol=1e-6, rls_rtol=5e-5, rls_atol=5e-5)
    assert m.status() == 0 and m.route() == "two-stream"


def test_config_D_one_gpu_and_shard_sum(vjf):
    """configs[3]: 32768 trials.  (1) all of them on one GPU, two steps against the oracle (single step and sequence entry
    points); (2) the 8-shard protocol of the multi-GPU path on one device: eight `vjf_filter_local` calls of 4096 trials, their
    reduce buffers summed (the all-reduce), one `vjf_filter_global` with B_total = 32768 -- same posterior bits, same loss and
    state up to summation order."""
    from vjf_amd import _native as N
    c = CFG_D1
    B, dz = c["B"], c["dz"]
    torch.manual_seed(13)
    m = _model(vjf, c, lr=1e-4)
    torch.manual_seed(13)
    m_sh = _model(vjf, c, lr=1e-4)
    torch.manual_seed(13)
    m_seq = _model(vjf, c, lr=1e-4)
    s = load_oracle_state(m, np.float64)
    y, eps = _data(c, 2, 23)
    yd, ed = y.cuda(), eps.cuda()
    q, om, ol = None, None, None
    outs = []
    for t in range(2):
        q, loss, *comp = m.filter(yd[t], None, q, verbose=True, eps=(ed[t, 0], ed[t, 1]))
        o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
        om, ol = o.mu_t, o.lv_t
        close(q.mean, o.mu_t, rtol=1e-6, atol=1e-6)
        close(torch.stack([loss, *comp]), [o.loss, o.recon, o.dyn, o.entropy], rtol=1e-6, atol=1e-6)
        outs.append((q.mean.clone(), loss.clone()))
    state_close(m, s, rtol=5e-6, atol=1e-6, rls_rtol=5e-4, rls_atol=5e-6)
    mu, lv, ls = m_seq.filter_sequence(yd, eps=ed)
    assert torch.equal(mu[1], outs[1][0]) and torch.equal(ls[1, 0], outs[1][1])
    assert torch.equal(m_seq._blob, m._blob) and m_seq.status() == 0
    # (2) eight shards
    m_sh._ensure_ctx(B // 8)
    L, ctx = m_sh._backend(), m_sh._ctx
    flags = N.FLAG_SGD | N.FLAG_UPDATE
    h = B // 8
    mu_s = torch.empty(B, dz, device="cuda"); lv_s = torch.empty(B, dz, device="cuda"); loss4 = torch.empty(4, device="cuda")
    acc = None
    for k in range(8):
        a, b = k * h, (k + 1) * h
        N.check(L.vjf_filter_local(ctx, h, N.ptr(yd[0, a:b]), None, None, None, N.ptr(ed[0, 0, a:b]), N.ptr(ed[0, 1, a:b]),
                                   N.ptr(mu_s[a:b]), N.ptr(lv_s[a:b]), flags))
        acc = m_sh._reduce.clone() if acc is None else acc + m_sh._reduce
    m_sh._reduce.copy_(acc)
    N.check(L.vjf_filter_global(ctx, B, N.ptr(loss4), flags))
    s_sh = load_oracle_state(_fresh(vjf, c), np.float64)
    o = orc.filter_step(s_sh, y[0].numpy(), None, None, None, eps[0, 0].numpy(), eps[0, 1].numpy())
    close(mu_s, o.mu_t, rtol=1e-6, atol=1e-6)
    close(loss4, [o.loss, o.recon, o.dyn, o.entropy], rtol=1e-6, atol=1e-6)
    state_close(m_sh, s_sh, rtol=5e-6, atol=1e-6, rls_rtol=5e-4, rls_atol=5e-6)
    close(mu_s, outs[0][0], rtol=2e-6, atol=2e-6)            # per-trial work does not depend on the shard
    close(loss4[0], outs[0][1], rtol=1e-6)


def _fresh(vjf, c):
    torch.manual_seed(13)
    return _model(vjf, c, lr=1e-4)


def test_replay_with_several_tiles_per_workgroup(vjf):
    """The replay of a step with a non-finite loss component (vjf/model.py:138-149) when a trial workgroup owns more than one tile
    (B = 8192: two tiles each at 256 CUs; the predictive moments come back from their per-trial save, the late slab is stored by the
    first tile and added to by the second), in the middle of a sequence; against the fp32 oracle."""
    import warnings
    c = dict(B=8192, dz=10, dy=50, n=200, hidden=[128], lik="gaussian")
    T = 3
    torch.manual_seed(17)
    m = _model(vjf, c, lr=1e-2)
    y, eps = _data(c, T, 27)
    yd, ed = y.cuda(), eps.cuda()
    mu0, lv0, _ = m.filter_sequence(yd[:1], eps=ed[:1])
    with torch.no_grad():
        m.transition.velocity.w_chol.mul_(1e25)                  # the predictive variance, and with it the dynamics term, overflows
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        s = load_oracle_state(m, np.float32)
        mu, lv, loss = m.filter_sequence(yd[1:], qs=vjf.Gaussian(mu0[-1], lv0[-1]), eps=ed[1:])
        assert m.status() & 2                                    # VJF_STATUS_NONFINITE_DYN
        om, ol = mu0[-1].cpu().numpy(), lv0[-1].cpu().numpy()
        for t in range(1, T):
            o = orc.filter_step(s, y[t].numpy(), None, om, ol, eps[t, 0].numpy(), eps[t, 1].numpy())
            om, ol = o.mu_t, o.lv_t
            close(mu[t - 1], o.mu_t, rtol=2e-6, atol=2e-6)
            close(loss[t - 1], [o.loss, o.recon, o.dyn, o.entropy], rtol=2e-6, atol=2e-6)
        state_close(m, s, rtol=2e-5, atol=2e-6, rls_rtol=2e-4, rls_atol=2e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("config", ["B", "E"])
def test_bench_line_contract(config):
    """`bench.py` as the round driver runs it (fewer steps): ONE JSON line with the contract's fields, the roofline and cpu_baseline
    objects, the ELBO of the timed steps checked against the oracle inside the run, status clean.  `--config E`: the extra line of
    configs[4] (its ELBO check is the oracle at B = 4096 on the GEMM-per-layer trial path)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    extra = [] if config == "B" else ["--config", "E", "--elbo-steps", "2"]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "12", "--warmup", "3", "--repeats", "2"] + extra,
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"] == "trial-timesteps/sec" and d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 3
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(d["value"] - 4096 * 12 / (d["ms_per_step"] * 1e-3 * 12)) / d["value"] < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] == os.cpu_count() and 1 <= c["blas_threads"] <= c["cores"] and c["sample"]
    assert d["elbo_check"]["ok"] and d["status_bits"] == 0
    if config == "B":                                          # the fixed cost of a call, measured behind the timed regions
        cc = d["call_cost"]
        assert 0.0 < cc["fixed_us_per_call"] < 400.0 and 0.0 < cc["single_filter_call_us"] < 600.0 and 20.0 < cc["steady_us_per_step"] < 200.0


def test_resident_column_loop_barrier_timeout_raises(vjf, monkeypatch):
    """The resident column loop of the multi-launch RLS update (vjf_rlsc_loop_kernel, the two-stream route) with one workgroup that
    never arrives at the step barrier (injected): the update is dropped like one with a failed pivot AND the status word carries a
    wait bit -- check_status() raises instead of the sequence going on with a stale W."""
    c = dict(B=64, dz=3, dy=10, n=256, hidden=[8], lik="gaussian")
    torch.manual_seed(3)
    m = _model(vjf, c)
    y, eps = _data(c, 3, 5)
    m.filter_sequence(y[:2].cuda(), eps=eps[:2].cuda())
    assert m.route() == "two-stream" and m.check_status() == 0
    w_before = m.transition.velocity.w_mean.clone()
    monkeypatch.setenv("VJF_DEBUG_RLSC_ABSENT", "2")
    m.filter_sequence(y[1:].cuda(), eps=eps[1:].cuda())
    monkeypatch.delenv("VJF_DEBUG_RLSC_ABSENT")
    torch.cuda.synchronize()
    assert torch.equal(m.transition.velocity.w_mean, w_before)        # (no update was applied)
    with pytest.raises(RuntimeError, match="timed out"):
        m.check_status()
