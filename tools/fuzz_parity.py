"""Diagnostic: random small configurations (every plan family: one-launch, per-step matrix-core trial kernel, multi-launch RLS, GEMM-per-layer
trial path; Gaussian / Poisson; control input; 1-3 layers; ragged batches) through `filter` and `filter_sequence` against the fp64 oracle
with the tolerances of tests/test_gpu_parity.py.

    python tools/fuzz_parity.py [n_cases] [seed] [only_case]
    FUZZ_FLAGS=warmup|infer|sgd-only|infer-warm|mix python tools/fuzz_parity.py ...   the other flag sets of VJF.filter (mix: by case index)

`draw_cases` / `run_case` are also what tests/test_gpu_handoffs.py calls: the random sweep with a fixed seed, and the configurations
that once failed (B << n: ill-conditioned precision matrices), pinned by their full description."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def draw_cases(N, seed):
    """[(case index, description)] -- the draw order is part of the tool: a (seed, index) pair names a configuration."""
    rng = np.random.default_rng(seed)
    out = []
    for case in range(N):
        fam = case % 4
        dz = int(rng.integers(1, 17)) if fam != 3 else int(rng.integers(2, 12))
        dy = int(rng.integers(2, 60)) if fam != 3 else int(rng.integers(280, 420))
        du = int(rng.integers(0, 4))
        n = int(rng.integers(5, 224)) if fam in (0, 1) else (int(rng.integers(226, 420)) if fam == 2 else int(rng.integers(1100, 1300)))
        if fam == 0:
            n = max(8, n & ~3)                                  # (the one-launch route wants n % 4 == 0)
        L = int(rng.integers(1, 4))
        hidden = [int(rng.integers(3, 70)) for _ in range(L)] if fam != 3 else [int(rng.integers(380, 440))]
        lik = "poisson" if rng.integers(0, 2) else "gaussian"
        B = int(rng.integers(1, 130))
        seq = bool(rng.integers(0, 2))
        overlap = fam != 1
        out.append((case, dict(fam=fam, B=B, dz=dz, dy=dy, du=du, n=n, hidden=hidden, lik=lik, seq=seq, overlap=overlap)))
    return out


FLAG_KW = {"train": dict(sgd=True, update=True, warm_up=False), "warmup": dict(sgd=True, update=True, warm_up=True),
           "infer": dict(sgd=False, update=False, warm_up=False), "sgd-only": dict(sgd=True, update=False, warm_up=False),
           "infer-warm": dict(sgd=False, update=True, warm_up=True)}


def flag_kw(case):
    """the flags of VJF.filter a case runs with (FUZZ_FLAGS; the draw of the configurations does not depend on it)"""
    f = os.environ.get("FUZZ_FLAGS", "train")
    if f == "mix":
        f = ("warmup", "infer", "sgd-only", "infer-warm", "train")[(case // 4) % 5]
    return FLAG_KW[f]


class Judge:
    """Fixed tolerance against the fp64 oracle first; beyond it, the device may be at most `slack` times as far from fp64 as the oracle run
    in the reference's fp32 arithmetic is (ill-conditioned cases: B << n, Poisson rates of 1e4) -- recorded as a note, not hidden."""

    def __init__(self, verbose=False, slack=3.0):
        self.notes, self.verbose, self.slack, self.lines = [], verbose, slack, []

    def __call__(self, name, a, r64, r32, rtol, atol):
        a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
        a = np.asarray(a, np.float64); r64 = np.asarray(r64, np.float64); r32 = np.asarray(r32, np.float64)
        dg = np.abs(a - r64).max(); d32 = np.abs(r32 - r64).max()
        if self.verbose:                                        # (one case, everything printed, nothing raised)
            self.lines.append(f"   {name}: device {dg:.3e}, fp32 oracle {d32:.3e}, scale {np.abs(r64).max():.3e}")
            print(self.lines[-1], flush=True)
            return
        if np.allclose(a, r64, rtol=rtol, atol=atol):
            return
        if dg > self.slack * d32 + atol + 1e-5 * np.abs(r64).max():
            raise AssertionError(f"{name}: device {dg:.3e} from the fp64 oracle, the fp32 oracle {d32:.3e} (scale {np.abs(r64).max():.3e})")
        self.notes.append(f"{name} {dg:.2e} vs fp32 oracle {d32:.2e}")


def run_case(case, desc, verbose=False, T=3):
    """Three steps of configuration `desc` (model seed 100 + case, data seed 200 + case) on the device against the oracle in fp64 and in
    fp32.  Raises AssertionError on a deviation; returns (route, notes)."""
    import vjf_amd as vjf
    from oracle import vjf_oracle as orc
    from tests.helpers import load_oracle_state, state_close
    dz, dy, du, n, hidden, lik, B, seq, overlap = (desc[k] for k in ("dz", "dy", "du", "n", "hidden", "lik", "B", "seq", "overlap"))
    kw = flag_kw(case)
    judge = Judge(verbose)
    torch.manual_seed(100 + case)
    m = vjf.VJF.make_model(dy, dz, du, n, hidden, likelihood=lik, lr=1e-3)
    if not overlap:
        m.set_overlap(False)
    s = load_oracle_state(m, np.float64)
    s32 = load_oracle_state(m, np.float32)                # the reference's own arithmetic
    g = torch.Generator().manual_seed(200 + case)
    y = torch.poisson(torch.exp(0.5 * torch.randn(T, B, dy, generator=g) - 0.5), generator=g) if lik == "poisson" else torch.randn(T, B, dy, generator=g)
    u = torch.randn(T, B, du, generator=g) if du else None
    eps = torch.randn(T, 2, B, dz, generator=g)
    if seq:
        mus, lvs, losses = m.filter_sequence(y, u, None, eps=eps, **kw)
    mu = lv = mu32 = lv32 = None; q = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(T):
            ut = None if u is None else u[t]
            un = None if u is None else ut.numpy()
            o = orc.filter_step(s, y[t].numpy(), un, mu, lv, eps[t, 0].numpy(), eps[t, 1].numpy(), **kw)
            o32 = orc.filter_step(s32, y[t].numpy(), un, mu32, lv32, eps[t, 0].numpy(), eps[t, 1].numpy(), **kw)
            mu, lv, mu32, lv32 = o.mu_t, o.lv_t, o32.mu_t, o32.lv_t
            if seq:
                gm, gl, gloss = mus[t], lvs[t], losses[t]
            else:
                q, l1, *comp = m.filter(y[t], ut, q, verbose=True, eps=(eps[t, 0], eps[t, 1]), **kw)
                gm, gl, gloss = q.mean, q.logvar, torch.stack([l1, *comp])
            judge(f"t={t} mean", gm, o.mu_t, o32.mu_t, 5e-5, 5e-5)
            judge(f"t={t} logvar", gl, o.lv_t, o32.lv_t, 5e-5, 5e-5)
            judge(f"t={t} losses", gloss, [o.loss, o.recon, o.dyn, o.entropy], [o32.loss, o32.recon, o32.dyn, o32.entropy], 5e-5, 5e-5)
            if verbose and not seq:
                for nm in ("w_mean", "w_chol", "w_precision", "w_pchol"):
                    judge(f"t={t} {nm}", getattr(m.transition.velocity, nm), getattr(s, nm), getattr(s32, nm), 0.0, 5e-5)
    step_notes = list(judge.notes)
    try:
        state_close(m, s, rtol=5e-4, atol=5e-5, rls_rtol=5e-3)
        assert not verbose
    except AssertionError:
        # state_close's tolerances do not hold in the ill-conditioned cases: EVERY state tensor -- trained parameters, the two
        # log-variances, the RLS tensors -- is then judged against the fp32 oracle's own distance from fp64 (the judge raises on a
        # tensor beyond the slack), with a list of notes of its own: the per-step notes above excuse nothing here
        from tests import goldenio as gio
        from tests.helpers import model_arrays
        judge.notes = []
        got, w64, w32 = model_arrays(m), gio.state_arrays(s), gio.state_arrays(s32)
        for nm in sorted(got):
            if nm in w64 and w64[nm] is not None:
                rls = nm in ("w_mean", "w_chol", "w_precision", "w_pchol")
                judge("state " + nm, got[nm], np.asarray(w64[nm]).reshape(got[nm].shape), np.asarray(w32[nm]).reshape(got[nm].shape),
                      0.0 if rls else 5e-4, 5e-5)
        state_notes, judge.notes = judge.notes, step_notes + judge.notes
        if not state_notes and not verbose:
            raise                                          # (state_close failed, yet no tensor is out of its tolerance: not explained)
    st = m.status()
    assert st == 0, hex(st)
    return m.route(**kw), list(judge.notes)


# Configurations that failed in round 2's sweeps (seeds 21, 22, 33 and an earlier draw order): B = 1 ... 9 trials against 214 ... 1289
# features -- the precision grows 10^4-fold in two steps -- where `w_chol` / `w_pchol` were 4-7 times as far from fp64 as the oracle
# run in fp32.  (case index, description): the index seeds the model and the data.
KNOWN_HARD = [
    (11, dict(fam=3, B=3, dz=2, dy=349, du=0, n=1138, hidden=[384], lik="poisson", seq=False, overlap=True)),
    (39, dict(fam=3, B=9, dz=3, dy=389, du=0, n=1189, hidden=[413], lik="gaussian", seq=False, overlap=True)),
    (151, dict(fam=3, B=4, dz=3, dy=379, du=1, n=1109, hidden=[411], lik="gaussian", seq=False, overlap=True)),
    (211, dict(fam=3, B=5, dz=2, dy=306, du=1, n=1289, hidden=[390], lik="poisson", seq=True, overlap=True)),
    (267, dict(fam=3, B=5, dz=4, dy=291, du=0, n=1254, hidden=[392], lik="gaussian", seq=False, overlap=True)),
    (67, dict(fam=3, B=2, dz=2, dy=328, du=0, n=1225, hidden=[418], lik="gaussian", seq=False, overlap=True)),
    (185, dict(fam=1, B=1, dz=1, dy=55, du=2, n=214, hidden=[40], lik="poisson", seq=True, overlap=False)),
]


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    cases = KNOWN_HARD if seed < 0 else draw_cases(N, seed)    # seed -1: the pinned hard configurations
    bad = 0
    for case, desc in cases:
        if only >= 0 and case != only:
            continue
        try:
            route, notes = run_case(case, desc, verbose=only >= 0)
            if notes:
                print(case, "note:", "; ".join(notes), flush=True)
            print(case, "ok", route, desc, flush=True)
        except Exception as e:            # noqa
            bad += 1
            print(case, "FAIL", desc, str(e).replace("\n", " ")[:300], flush=True)
    print("failures:", bad, "of", len(cases))


if __name__ == "__main__":
    main()
