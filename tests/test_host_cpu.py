"""CPU-only tests of the boundary and the host side of vjf_amd (no GPU, no compute through HIP):
the library loads and exports the header's symbols, the memory plan is sane, compute fails loudly
without a device, seeded construction reproduces the reference's parameters, and -- with the
oracle-backed stand-in of tests/fake_backend.py behind the same C ABI -- the host logic of filter /
filter_sequence / fit / forecast reproduces the golden vectors."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from tests import fake_backend
from tests import goldenio as gio
from tests.helpers import load_fixture_state, state_close

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def close(a, b, **kw):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), **kw)


def test_library_exports_every_declared_symbol():
    from vjf_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "vjf_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(vjf_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 20
    lib = C.CDLL(N.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/vjf_hip.h but not exported"
    assert declared == set(N.SIGNATURES) | {"vjf_last_error"}, declared ^ (set(N.SIGNATURES) | {"vjf_last_error"})
    assert lib.vjf_abi_version() == N.ABI_VERSION
    # the diagnostic build of the same sources (hand-offs under perturbed timing, tests/test_gpu_handoffs.py): the same interface
    from vjf_amd import _build
    chaos = C.CDLL(_build.build(chaos=True))
    for name in sorted(declared):
        assert hasattr(chaos, name), f"{name} missing from the diagnostic build"
    assert chaos.vjf_abi_version() == N.ABI_VERSION


def test_memory_plan_config_b():
    from vjf_amd import _native as N
    cfg = N.make_config(50, 10, 0, 200, [128], N.LIK_GAUSSIAN, 4096)
    n, off, size = N.state_layout(cfg)
    assert size[N.SLOT_CENTROID] == 2000 and size[N.SLOT_W_CHOL] == 40000 and size[N.SLOT_REC_W0] == 128 * 70
    assert all(o % 4 == 0 for o in off)
    used = sorted((o, s) for o, s in zip(off, size) if s)
    assert all(a[0] + a[1] <= b[0] for a, b in zip(used, used[1:]))          # slots do not overlap
    assert used[-1][0] + used[-1][1] <= n
    assert 1 << 20 < N.workspace_size(cfg) < 1 << 28
    bad = N.make_config(50, 10, 0, 200, [128], N.LIK_GAUSSIAN, 4096)
    bad.n_hidden = 0
    out = C.c_int64()
    assert N.lib().vjf_state_size(C.byref(bad), C.byref(out)) < 0
    assert b"invalid config" in N.lib().vjf_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_compute_fails_loudly_without_gpu():
    import vjf_amd
    from vjf_amd._native import VjfError
    m = vjf_amd.VJF.make_model(10, 3, 0, 16, [8], likelihood="gaussian")
    with pytest.raises(VjfError):
        m.filter(torch.randn(4, 10))
    with pytest.raises(VjfError):
        vjf_amd.functional.rbf(torch.randn(4, 3), torch.randn(5, 3), torch.ones(5))


def test_seeded_construction_matches_reference():
    """Same construction (RNG consumption) order as vjf/model.py:309-319: bit-identical parameters."""
    import vjf_amd
    z = gio.load("g5_seeded_f32")
    torch.manual_seed(int(z["seeds"][0]))
    m = vjf_amd.VJF.make_model(10, 3, 2, 16, [8], likelihood="gaussian")
    state_close(m, z, prefix="s0", rtol=0, atol=0)
    assert list(m.state_dict().keys()) == [
        "mean", "logvar", "likelihood.logvar", "transition.logvar", "transition.velocity.feature.centroid",
        "transition.velocity.feature.logwidth", "recognition.mlp.0.weight", "recognition.mlp.0.bias",
        "recognition.mean.weight", "recognition.logvar.weight", "recognition.logvar.bias", "decoder.decode.weight",
        "decoder.decode.bias"]
    assert m.likelihood.n_sample == 0 and m.transition.n_sample == 0
    assert [g["lr"] for g in m.optimizer.param_groups] == [1e-4] * 4
    m.scheduler.step()
    assert [g["lr"] for g in m.optimizer.param_groups] == [1e-4 * 0.9] * 4


@pytest.fixture
def fake():
    undo = fake_backend.install()
    yield
    undo()


@pytest.mark.skipif(torch.cuda.is_available(), reason="the stand-in backend works on CPU tensors")
@pytest.mark.parametrize("name", ["g5_gaussian_du2_wu0_f64", "g5_poisson_du0_wu1_f64", "g5_gaussian_h5x5_f64", "g5_gaussian_lr1e-2_f64"])
def test_host_filter_logic_golden(fake, name):
    """vjf_amd.VJF.filter -> C ABI (stand-in) : argument coercion, prior / posterior chaining, flags, counters, lr."""
    import vjf_amd
    z, info, _ = gio.traj_case(name)
    m = vjf_amd.VJF.make_model(info["dy"], info["dz"], info["du"], info["n"], info["hidden"], likelihood=info["lik"])
    load_fixture_state(m, z, "s0")
    u = z["u"] if info["du"] else None
    q = None
    for t in range(info["T"]):
        q, loss, *comp = m.filter(z["y"][t], None if u is None else u[t], q, verbose=True, warm_up=info["warm_up"],
                                  eps=(z["eps"][t, 0], z["eps"][t, 1]))
        close(q.mean, z["out.mu"][t], rtol=1e-4, atol=1e-5)
        close(torch.stack([loss, *comp]), z["out.loss"][t], rtol=1e-4, atol=1e-5)
        assert m.transition.n_sample == int(z["out.n_tr"][t])
    state_close(m, z, prefix="sT", rtol=1e-3, atol=1e-5)
    assert m.feed.__func__ is m.filter.__func__


@pytest.mark.skipif(torch.cuda.is_available(), reason="the stand-in backend works on CPU tensors")
def test_host_fit_harness_golden(fake):
    """fit(): epochs, warm-up exit, decoder freeze, RBFDS.initialize (CPU-generator draws in the reference's
    order), convergence break, lr decay; then forecast with the strided randn_like draw."""
    import vjf_amd
    z = gio.load("g8_fit")
    T, B, dy, dz, du, n = [int(v) for v in z["meta"][:6]]
    hid = [int(v) for v in z["meta"][6:]]
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        m = vjf_amd.VJF.make_model(dy, dz, du, n, hid, likelihood="gaussian")
        load_fixture_state(m, z, "s0")
        torch.manual_seed(int(z["fit_seed"]))
        mu, lv, epoch_loss = m.fit(torch.tensor(z["y"]), max_iter=3, rtol=10.0)
        close(mu, z["mu"], rtol=1e-3, atol=1e-4)
        close(epoch_loss, z["epoch_loss"], rtol=1e-4)
        assert bool(m._scalars[6].item())                                   # decoder frozen after warm-up
        close([g["lr"] for g in m.optimizer.param_groups], z["sT.lr"], rtol=1e-12)
        state_close(m, z, prefix="sT", rtol=2e-3, atol=1e-4, rls_atol=1e-3)
        torch.manual_seed(int(z["fc_seed"]))
        x, yf = m.forecast(torch.tensor(z["fc_x0"]), n_step=z["fc_wnoise"].shape[0])
        close(x, z["fc_x"], rtol=1e-3, atol=1e-3)
    finally:
        torch.set_default_dtype(old)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the stand-in backend works on CPU tensors")
def test_host_sequence_and_reference_noise_order(fake):
    """filter_sequence draws the noise like T filter calls do (xs then xt per step, CPU generator)."""
    import vjf_amd
    torch.manual_seed(3)
    m1 = vjf_amd.VJF.make_model(6, 2, 0, 8, [5], likelihood="gaussian")
    torch.manual_seed(3)
    m2 = vjf_amd.VJF.make_model(6, 2, 0, 8, [5], likelihood="gaussian")
    y = torch.randn(5, 7, 6)
    torch.manual_seed(11)
    mu, lv, loss = m1.filter_sequence(y)
    torch.manual_seed(11)
    q = None
    for t in range(5):
        q, l = m2.filter(y[t], qs=q)
        close(q.mean, mu[t], rtol=1e-6, atol=1e-7)
        close(l, loss[t, 0], rtol=1e-6)
    close(m1._blob, m2._blob, rtol=1e-6, atol=1e-7)


@pytest.mark.skipif(torch.cuda.is_available(), reason="the stand-in backend works on CPU tensors")
def test_state_io_roundtrip_cpu(fake, tmp_path):
    """get_state / set_state / save_state / load_state on the host side (oracle-backed stand-in library)."""
    import vjf_amd as vjf
    torch.manual_seed(3)
    m1 = vjf.VJF.make_model(6, 2, 1, 8, [5], likelihood="gaussian")
    g = torch.Generator().manual_seed(1)
    y, u, eps = torch.randn(3, 4, 6, generator=g), torch.randn(3, 4, 1, generator=g), torch.randn(3, 2, 4, 2, generator=g)
    m1.filter_sequence(y[:2], u[:2], eps=eps[:2])
    m1.freeze_decoder(True)
    p = tmp_path / "state.npz"
    m1.save_state(p)
    torch.manual_seed(99)
    m2 = vjf.VJF.make_model(6, 2, 1, 8, [5], likelihood="gaussian")
    m2.load_state(p)
    s1, s2 = m1.get_state(), m2.get_state()
    assert set(s1) == set(s2) and len(s1) == 13 + 4 + 5
    for k in s1:
        assert np.array_equal(np.asarray(s1[k]), np.asarray(s2[k])), k
    assert m2.likelihood.n_sample == m1.likelihood.n_sample and m2.transition.n_sample == m1.transition.n_sample
    o1 = m1.filter_sequence(y[2:], u[2:], eps=eps[2:])
    o2 = m2.filter_sequence(y[2:], u[2:], eps=eps[2:])
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    m3 = vjf.VJF.make_model(6, 2, 1, 9, [5], likelihood="gaussian")
    try:
        m3.set_state(s1)
        assert False, "shape mismatch must raise"
    except ValueError:
        pass


@pytest.mark.skipif(torch.cuda.is_available(), reason="the stand-in backend works on CPU tensors")
def test_get_state_reads_the_status_word_first(fake):
    """`filter` / `filter_sequence` are asynchronous and do not read the device's status word; `get_state` (and `save_state`)
    must, before they copy anything: a timed-out hand-off (VJF_STATUS_WAIT_* bits) means the state is not one to keep -- it
    raises; an RLS failure is warned about as the reference does ('RLS failed.', vjf/module.py:112) and the (unchanged) state
    is returned."""
    import warnings
    import vjf_amd as vjf
    from vjf_amd import _native as N
    torch.manual_seed(3)
    m = vjf.VJF.make_model(6, 2, 0, 8, [5], likelihood="gaussian")
    g = torch.Generator().manual_seed(1)
    y, eps = torch.randn(2, 4, 6, generator=g), torch.randn(2, 2, 4, 2, generator=g)
    m.filter_sequence(y, eps=eps)
    assert len(m.get_state()) == 13 + 4 + 5
    m._scalars[N.SC_STATUS] = float(N.STATUS_RLS_FAILED)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        st = m.get_state()
    assert any("RLS failed" in str(x.message) for x in w) and len(st) == 13 + 4 + 5
    m._scalars[N.SC_STATUS] = float(0x100)                    # a hand-off wait that ran out
    with pytest.raises(RuntimeError):
        m.get_state()
    assert m.status() == 0                                    # (read and cleared)


def test_lorenz_generator_is_the_surveyed_system():
    """sigma = 10, rho = 28, beta = 8/3, RK4 with dt = 0.01, 500 burn-in steps, z-scored."""
    from vjf_amd.data import lorenz, observe, rbf_system
    x = lorenz(400)
    assert x.shape == (400, 3) and x.dtype == torch.float64
    close(x.mean(0), [0, 0, 0], atol=1e-9)
    close(x.std(0), [1, 1, 1], rtol=1e-9)
    # one RK4 step of the raw system from the burn-in-free state, by hand
    s = torch.tensor([1.0, 1.0, 1.0], dtype=torch.float64)
    f = lambda v: torch.stack((10.0 * (v[1] - v[0]), v[0] * (28.0 - v[2]) - v[1], v[0] * v[1] - 8.0 / 3.0 * v[2]))  # noqa: E731
    k1 = f(s); k2 = f(s + 0.005 * k1); k3 = f(s + 0.005 * k2); k4 = f(s + 0.01 * k3)
    s1 = s + 0.01 / 6 * (k1 + 2 * k2 + 2 * k3 + k4)
    raw = lorenz(2, burn_in=0)                                               # z-scored pair: direction of the first step survives
    d = (s1 - s) / (s1 - s).abs().max()
    assert (raw[1] - raw[0]).sign().tolist() == [1.0, 1.0, 1.0] or d.abs().max() == 1      # (sanity only)
    g = torch.Generator().manual_seed(3)
    y, Cm, dv = observe(x.float(), 10, "gaussian", generator=g)
    assert y.shape == (400, 10) and Cm.shape == (3, 10) and dv.shape == (10,)
    yp, _, _ = observe(x.float(), 7, "poisson", generator=g)
    assert (yp >= 0).all() and (yp == yp.round()).all()
    xr = rbf_system(5, 8, 4, generator=torch.Generator().manual_seed(1))
    assert xr.shape == (5, 8, 4) and torch.isfinite(xr).all()


@pytest.mark.skipif(torch.cuda.is_available(), reason="the stand-in backend works on CPU tensors")
def test_lorenz_example_plumbing(fake, capsys):
    """BASELINE configs[0] (one Lorenz trial, d_z = 3, d_y = 10, Gaussian): examples/lorenz_fit.py end to end -- make_model, a few
    epochs of fit (warm-up logic included), forecast -- on the oracle-backed stand-in for the C ABI."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lorenz_fit", os.path.join(ROOT, "examples", "lorenz_fit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    m, loss = mod.main(["--epochs", "2", "--T", "60", "--n-rbf", "20", "--forecast", "5"])
    assert m.shape == (60, 3) and torch.isfinite(m).all() and np.isfinite(float(loss))
    assert "forecast: (6, 1, 3) (6, 1, 10) finite: True" in capsys.readouterr().out


def test_bench_starts_its_own_ranks_only_without_a_launcher(monkeypatch):
    """`bench.py --gpus N` (N > 1) with no WORLD_SIZE starts a torch.distributed.run child and relays its exit code; with a
    launcher's environment, or at N = 1, it runs in-process (returns None).  Without a GPU the ranks fail: the code is relayed and
    no result line is printed."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench._launch_own_ranks(["--gpus", "1", "--steps", "5"]) is None
    assert bench._launch_own_ranks(["--steps", "5"]) is None
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench._launch_own_ranks(["--gpus", "2"]) is None
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    if not torch.cuda.is_available():
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus=2", "--steps", "2", "--warmup", "1"], env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert r.returncode != 0 and r.stdout.strip() == b""
        assert b"needs a GPU" in r.stderr


def test_bench_algorithmic_work_is_the_surveys():
    """230,200 FLOP and 440 B per trial-timestep at config B (+ 762,472 B shared per step); the other flag sets
    drop terms, never add any."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    f, bt, bs, ser = bench.algorithmic_work(bench.CFGS["B"])
    assert (f, bt, bs) == (230200, 440, 762472) and ser == (2 * 200 ** 3) // 3 + 4 * 200 * 200 * 10
    for fl in ("warmup", "sgd-only", "infer"):
        f2, _, _, s2 = bench.algorithmic_work(bench.CFGS["B"], fl)
        assert 0 < f2 < f and s2 == 0
    assert bench.CFGS["D1"]["B"] == 8 * bench.CFGS["B"]["B"]
