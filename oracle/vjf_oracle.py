"""CPU oracle for the VJF online filtering step  --  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch numpy restatement of the reference algorithm
(catniplab/vjf, `vjf.model.VJF.filter` and the operators it reaches).  It is
the *checker* for the HIP path: only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it.  Nothing under `vjf_amd/`
imports it, and the product path raises when the HIP library is missing.

Parity status: PINNED.  Every function here is checked against golden vectors
captured from the imported reference (tests/golden/make_golden.py, run in the
build container where /root/reference is importable) by
tests/test_oracle_golden.py: fp64 to <=1e-10, fp32 to <=2e-5.

All citations `file:line` are into /root/reference/.

Differences from the reference that do not change results beyond rounding:
  * predictive variance is the row sum of squares  sum_j (Phi w_chol)[b,j]^2,
    i.e. the diagonal of the reference's (B,B) product, never the product itself
    (vjf/module.py:75-76);  `faithful_cost=True` materialises the (B,B) matrix
    so the oracle can also represent the reference's real CPU cost;
  * squared distances are formed directly (sum_d (x-c)^2) instead of through
    torch.cdist's matmul expansion (vjf/functional.py:20);
  * noise is an explicit input (eps_s, eps_t) instead of a draw from the global
    generator inside reparametrize (vjf/util.py:11-13); draw order is xs then xt;
  * the backward pass is hand-derived (the reference uses autograd).
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np
import scipy.linalg as sla

GAUSSIAN = "gaussian"
POISSON = "poisson"

LIK_SIZE_CAP = 1000   # vjf/util.py:20 default size_cap, used by likelihood.py:38
TR_SIZE_CAP = 500     # vjf/model.py:375


# --------------------------------------------------------------------------- state
@dataclass
class OracleState:
    """Everything `VJF.filter` reads or mutates (checkpoint row)."""
    ydim: int
    xdim: int
    udim: int
    n_rbf: int
    hidden: Tuple[int, ...]
    likelihood: str
    # prior (vjf/model.py:66-67) -- registered, never stepped
    prior_mean: np.ndarray = None
    prior_logvar: np.ndarray = None
    # likelihood (vjf/likelihood.py:16-17); None for Poisson
    lik_logvar: Optional[np.ndarray] = None
    n_lik: int = 0
    # transition (vjf/model.py:330-332, vjf/module.py:20-21, 46-54)
    tr_logvar: np.ndarray = None
    n_tr: int = 0
    centroid: np.ndarray = None
    logwidth: np.ndarray = None
    w_mean: np.ndarray = None
    w_chol: np.ndarray = None
    w_precision: np.ndarray = None
    w_pchol: np.ndarray = None
    # recognition (vjf/recognition.py:20-28)
    rec_W: List[np.ndarray] = field(default_factory=list)
    rec_b: List[np.ndarray] = field(default_factory=list)
    mean_W: np.ndarray = None
    lv_W: np.ndarray = None
    lv_b: np.ndarray = None
    # decoder (vjf/model.py:24)
    dec_W: np.ndarray = None
    dec_b: np.ndarray = None
    # optimiser (vjf/model.py:69-78): one lr per group [likelihood, decoder, transition, recognition]
    lr: List[float] = field(default_factory=lambda: [1e-4] * 4)
    freeze_decoder: bool = False      # vjf/model.py:283

    @property
    def dtype(self):
        return self.centroid.dtype

    def clone(self) -> "OracleState":
        return copy.deepcopy(self)

    def cast(self, dtype) -> "OracleState":
        s = self.clone()
        for k, v in vars(s).items():
            if isinstance(v, np.ndarray):
                setattr(s, k, v.astype(dtype))
            elif isinstance(v, list) and v and isinstance(v[0], np.ndarray):
                setattr(s, k, [a.astype(dtype) for a in v])
        return s


def init_state(ydim, xdim, udim, n_rbf, hidden, likelihood, rng: np.random.Generator,
               dtype=np.float64, lr=1e-4) -> OracleState:
    """Random initial state with the reference's shapes and constant inits
    (vjf/model.py:309-319).  Random tensors use numpy, NOT torch's generator;
    seed-exact initialisation parity lives in the host shim (vjf_amd.model)."""
    hidden = tuple(int(h) for h in hidden)
    din = ydim + udim + 2 * xdim
    s = OracleState(ydim, xdim, udim, n_rbf, hidden, likelihood.lower())
    s.prior_mean = np.zeros(xdim, dtype)
    s.prior_logvar = np.zeros(xdim, dtype)
    s.lik_logvar = np.asarray(math.log(0.1), dtype) if s.likelihood == GAUSSIAN else None
    s.tr_logvar = np.asarray(0.0, dtype)
    s.centroid = (rng.random((n_rbf, xdim + udim)) * 4 - 2).astype(dtype)
    s.logwidth = np.zeros(n_rbf, dtype)
    s.w_mean = np.zeros((n_rbf, xdim), dtype)
    s.w_chol = np.eye(n_rbf, dtype=dtype)
    s.w_precision = np.eye(n_rbf, dtype=dtype)
    s.w_pchol = np.eye(n_rbf, dtype=dtype)

    def lin(fan_out, fan_in, bias=True):
        k = 1.0 / math.sqrt(fan_in)
        W = rng.uniform(-k, k, (fan_out, fan_in)).astype(dtype)
        b = rng.uniform(-k, k, fan_out).astype(dtype) if bias else None
        return W, b

    sizes = (din,) + hidden
    for a, b_ in zip(sizes[:-1], sizes[1:]):
        W, b = lin(b_, a)
        s.rec_W.append(W)
        s.rec_b.append(b)
    s.mean_W, _ = lin(xdim, hidden[-1], bias=False)
    s.lv_W, s.lv_b = lin(xdim, hidden[-1])
    s.dec_W, s.dec_b = lin(ydim, xdim)
    s.lr = [lr] * 4
    return s


# --------------------------------------------------------------------------- operators
RBF_GEMM = False      # bench.py's cpu_baseline leg sets this: squared distances by the matmul expansion, as torch.cdist computes them
                      # for more than 25 rows (vjf/functional.py:20) -- one BLAS call instead of a pass per input dimension


def rbf(x: np.ndarray, c: np.ndarray, w: np.ndarray) -> np.ndarray:
    """Gaussian radial basis features  exp(-1/2 (|x-c|/w)^2)   (vjf/functional.py:11-22)."""
    if RBF_GEMM:
        d2 = (x * x).sum(1)[:, None] + (c * c).sum(1)[None, :] - 2.0 * (x @ c.T)
        np.maximum(d2, 0.0, out=d2)
        return np.exp(-0.5 * d2 / (w * w)[None, :]).astype(x.dtype)
    d2 = np.zeros((x.shape[0], c.shape[0]), x.dtype)
    for j in range(x.shape[1]):                  # one (batch, basis) pass per input dimension: no 3-D temporary
        diff = x[:, j:j + 1] - c[None, :, j]
        d2 += diff * diff
    return np.exp(-0.5 * d2 / (w * w)[None, :]).astype(x.dtype)


def nonecat(a: np.ndarray, u: Optional[np.ndarray]) -> np.ndarray:
    """[a, u] allowing u=None or zero-width u   (vjf/util.py:38-49)."""
    a = np.atleast_2d(a)
    if u is not None and u.shape[-1] > 0:
        return np.concatenate([a, np.atleast_2d(u)], axis=-1)
    return a


def blr_predict(s: OracleState, xu: np.ndarray, faithful_cost: bool = False):
    """Bayesian-linear-regression predictive (mean, logvar), sampling=False branch
    (vjf/module.py:64-77).  logvar has one value per row, tiled over outputs."""
    feat = rbf(xu, s.centroid, np.exp(s.logwidth))
    FL = feat @ s.w_chol
    if faithful_cost:
        var = np.diagonal(FL @ FL.T)          # the reference's (B,B) product
    else:
        var = np.sum(FL * FL, axis=1)
    with np.errstate(divide="ignore"):
        logvar = np.log(var)
    logvar = np.tile(logvar[:, None], (1, s.w_mean.shape[1]))
    return feat @ s.w_mean, logvar.astype(xu.dtype), feat


def recognition_forward(s: OracleState, y, mu_s, lv_s, u=None, keep=False):
    """tanh MLP on [y, u, mu_s, logvar_s] with a bias-free mean head and a biased
    logvar head   (vjf/recognition.py:31-42; layer structure :20-28)."""
    yu = nonecat(y, u)
    h = np.concatenate([yu, mu_s, lv_s], axis=-1)
    acts = [h]
    for W, b in zip(s.rec_W, s.rec_b):
        h = np.tanh(h @ W.T + b)
        acts.append(h)
    mu_t = h @ s.mean_W.T
    lv_t = h @ s.lv_W.T + s.lv_b
    if keep:
        return mu_t, lv_t, acts
    return mu_t, lv_t


def gaussian_loss(m1, lv1, m2, lv2, logvar):
    """Expected Gaussian negative log-likelihood with the reference's 'trace' term
    (vjf/functional.py:32-75).  lv1/lv2 = None for point arguments."""
    p = np.exp(-0.5 * logvar)
    mse = (m1 * p - m2 * p) ** 2
    assert np.all(np.isfinite(mse)), "non-finite squared error"     # functional.py:60
    nll = 0.5 * (mse + logvar)
    if lv1 is None and lv2 is None:
        trace = 0.0
    elif lv2 is None:
        trace = np.exp(lv1 - logvar)
    elif lv1 is None:
        trace = np.exp(lv2 - logvar)
    else:
        trace = np.exp(lv1 + lv2 - logvar)
    nll = nll + 0.5 * trace
    return nll.sum(-1).mean()


def gaussian_entropy(lv):
    """1/2 mean_b sum_j logvar   (vjf/functional.py:25-29)."""
    assert lv.ndim >= 2
    return 0.5 * lv.sum(-1).mean()


def poisson_loss(eta, target):
    """Poisson NLL with log link, eta clamped to <= 10, no Stirling term
    (vjf/likelihood.py:51-62 -> torch poisson_nll_loss(log_input=True, full=False))."""
    e = np.minimum(eta, 10.0)
    nll = np.exp(e) - target * e
    assert nll.ndim == 2
    return nll.sum(-1).mean()


def running_var(acc_var, acc_size, new_var, new_size, size_cap=1000):
    """Capped running variance   (vjf/util.py:20-35)."""
    acc_size = min(acc_size, size_cap)
    tot = acc_size + new_size
    f1 = acc_size / tot
    f2 = new_size / tot
    return f1 * acc_var + f2 * new_var, tot


def _cholesky(P: np.ndarray) -> np.ndarray:
    """Lower Cholesky factor IN THE DTYPE OF P (LAPACK spotrf / dpotrf through scipy), as torch.linalg.cholesky computes it for
    the reference (vjf/module.py:99).  `np.linalg.cholesky` would not do: numpy evaluates every linalg routine in double and rounds
    the result (numpy/linalg/_linalg.py, `_commonType`: "use higher precision (always double or cdouble)"), so on a float32 matrix
    it returns the correctly rounded fp64 factor -- for an ill-conditioned precision matrix (few trials against many features)
    two to three orders of magnitude closer to the exact factor than any fp32 factorisation, the reference's included
    (tools/rls_accuracy_emu.py)."""
    return sla.cholesky(P, lower=True, check_finite=False)


def rls(s: OracleState, feat: np.ndarray, target: np.ndarray, v, shrink: float = 1.0):
    """Information-form recursive least squares on the RBF weights
    (vjf/module.py:79-112).  Mutates s.w_precision / w_pchol / w_mean / w_chol.
    Returns 0, or 1 when the Cholesky failed and the jitter fallback ran (the
    reference's fallback calls the removed torch.eig and is dead on modern torch)."""
    dt = feat.dtype
    sd = np.sqrt(np.asarray(v, dt))
    sf = feat / sd
    st = target / sd
    P = s.w_precision
    g = (P @ s.w_mean) * shrink + sf.T @ st
    P = P * shrink + sf.T @ sf
    status = 0
    try:
        L = _cholesky(P)
    except np.linalg.LinAlgError:
        lam = np.linalg.eigvalsh(P.astype(np.float64)).min()
        L = _cholesky(P + np.eye(P.shape[0], dtype=dt) * dt.type(abs(lam) * 2))
        status = 1
    s.w_pchol = L.astype(dt)
    s.w_precision = P.astype(dt)
    s.w_mean = sla.cho_solve((L, True), g).astype(dt)
    s.w_chol = sla.solve_triangular(L.T, np.eye(P.shape[0], dtype=dt), lower=False).astype(dt)
    return status


# --------------------------------------------------------------------------- the step
@dataclass
class StepOut:
    mu_t: np.ndarray
    lv_t: np.ndarray
    loss: float
    recon: float        # = -l_recon  (vjf/model.py:152)
    dyn: float          # = -l_dynamics
    entropy: float
    grads: dict
    xs: np.ndarray
    xt: np.ndarray
    py: np.ndarray
    pt_mean: np.ndarray
    pt_logvar: np.ndarray
    rls_status: int = 0


def _clip(g):
    return np.clip(g, -1.0, 1.0)       # vjf/model.py:210  clip_grad_value_(…, 1.)


def filter_step(s: OracleState, y, u, mu_s, lv_s, eps_s, eps_t, *, sgd=True, update=True,
                warm_up=False, faithful_cost=False) -> StepOut:
    """One call of VJF.filter (vjf/model.py:179-221) on a batch of B trials.

    mu_s / lv_s = None selects the prior (vjf/model.py:80-95, 107-108).  Mutates `s`
    exactly where the reference mutates the model: SGD on the optimised tensors,
    the likelihood running variance, the RLS state and the state-noise variance.
    """
    dt = s.dtype
    y = np.atleast_2d(np.asarray(y, dt))
    B = y.shape[0]
    if u is not None:
        u = np.atleast_2d(np.asarray(u, dt))
    if mu_s is None:
        mu_s = np.ones((B, s.xdim), dt) * s.prior_mean
        lv_s = np.ones((B, s.xdim), dt) * s.prior_logvar
    eps_s = np.asarray(eps_s, dt)
    eps_t = np.asarray(eps_t, dt)

    # ---- forward (vjf/model.py:97-122)
    xs = mu_s + eps_s * np.exp(0.5 * lv_s)                       # util.py:11-13, draw #1
    xu = nonecat(xs, u)
    dmean, pt_lv, feat = blr_predict(s, xu, faithful_cost)       # model.py:334-340
    pt_mean = xs + dmean                                         # leak = 0
    mu_t, lv_t, acts = recognition_forward(s, y, mu_s, lv_s, u, keep=True)
    xt = mu_t + eps_t * np.exp(0.5 * lv_t)                       # draw #2
    py = xt @ s.dec_W.T + s.dec_b                                # model.py:28-30

    # ---- loss (vjf/model.py:124-154)
    if s.likelihood == GAUSSIAN:
        l_recon = gaussian_loss(y, None, py, None, s.lik_logvar)     # likelihood.py:26
    else:
        l_recon = poisson_loss(py, y)
    l_dyn = gaussian_loss(pt_mean, pt_lv, mu_t, lv_t, s.tr_logvar)   # model.py:390-391
    h = gaussian_entropy(lv_t)
    ok_recon = bool(np.isfinite(l_recon))
    ok_dyn = bool(np.isfinite(l_dyn))
    ok_h = bool(np.isfinite(h))
    l_recon = l_recon if ok_recon else dt.type(0)                # model.py:138-145
    l_dyn = l_dyn if ok_dyn else dt.type(0)
    h = h if ok_h else dt.type(0)
    loss = l_recon - h
    if not warm_up:
        loss = loss + l_dyn

    # ---- backward + clipped SGD (vjf/model.py:206-214); SURVEY.md 8a-bwd
    grads = {}
    if sgd:
        inv_b = dt.type(1.0 / B)
        r = py - y
        g_rho = None
        if not ok_recon:
            d_py = np.zeros_like(py)
        elif s.likelihood == GAUSSIAN:
            e = np.exp(-s.lik_logvar)
            d_py = e * r * inv_b
            g_rho = np.sum(0.5 * (1.0 - e * r * r)) * inv_b
        else:
            d_py = np.where(py <= 10.0, np.exp(np.minimum(py, 10.0)) - y, 0.0).astype(dt) * inv_b
        d_mu = np.zeros_like(mu_t)
        d_lv = np.zeros_like(lv_t)
        if (not warm_up) and ok_dyn:
            es = np.exp(-s.tr_logvar)
            d_mu += -es * (pt_mean - mu_t) * inv_b
            d_lv += 0.5 * np.exp(pt_lv + lv_t - s.tr_logvar) * inv_b
        if ok_h:
            d_lv += -0.5 * inv_b
        g_decW = d_py.T @ xt
        g_decb = d_py.sum(0)
        d_xt = d_py @ s.dec_W
        d_mu = d_mu + d_xt
        d_lv = d_lv + d_xt * eps_t * 0.5 * np.exp(0.5 * lv_t)
        hL = acts[-1]
        g_meanW = d_mu.T @ hL
        g_lvW = d_lv.T @ hL
        g_lvb = d_lv.sum(0)
        d_h = d_mu @ s.mean_W + d_lv @ s.lv_W
        g_recW = [None] * len(s.rec_W)
        g_recb = [None] * len(s.rec_W)
        for k in reversed(range(len(s.rec_W))):
            d_a = d_h * (1.0 - acts[k + 1] ** 2)
            g_recW[k] = d_a.T @ acts[k]
            g_recb[k] = d_a.sum(0)
            d_h = d_a @ s.rec_W[k]
        grads = dict(lik_logvar=g_rho, dec_W=g_decW, dec_b=g_decb, mean_W=g_meanW,
                     lv_W=g_lvW, lv_b=g_lvb, rec_W=g_recW, rec_b=g_recb)
        lr_lik, lr_dec, _lr_tr, lr_rec = (dt.type(x) for x in s.lr)
        if g_rho is not None:
            s.lik_logvar = (s.lik_logvar - lr_lik * _clip(g_rho)).astype(dt)
        if not s.freeze_decoder:
            s.dec_W = (s.dec_W - lr_dec * _clip(g_decW)).astype(dt)
            s.dec_b = (s.dec_b - lr_dec * _clip(g_decb)).astype(dt)
        s.mean_W = (s.mean_W - lr_rec * _clip(g_meanW)).astype(dt)
        s.lv_W = (s.lv_W - lr_rec * _clip(g_lvW)).astype(dt)
        s.lv_b = (s.lv_b - lr_rec * _clip(g_lvb)).astype(dt)
        for k in range(len(s.rec_W)):
            s.rec_W[k] = (s.rec_W[k] - lr_rec * _clip(g_recW[k])).astype(dt)
            s.rec_b[k] = (s.rec_b[k] - lr_rec * _clip(g_recb[k])).astype(dt)

    # ---- closed-form updates (vjf/model.py:156-177) with the forward pass's py, xt, xs
    rls_status = 0
    if update:
        if s.likelihood == GAUSSIAN:                              # likelihood.py:28-40
            mse = np.mean((y - py) ** 2)
            var, n = running_var(np.exp(s.lik_logvar), s.n_lik, mse, B, LIK_SIZE_CAP)
            s.lik_logvar = np.asarray(np.log(var), dt)
            s.n_lik = n
        dx = xt - xs                                              # model.py:363-377
        if not warm_up:
            rls_status = rls(s, feat, dx, np.exp(s.tr_logvar), 1.0)
        residual = dx - feat @ s.w_mean
        mse = np.mean(residual ** 2)
        var, n = running_var(np.exp(s.tr_logvar), s.n_tr, mse, B, TR_SIZE_CAP)
        s.tr_logvar = np.asarray(np.log(var), dt)
        s.n_tr = n

    return StepOut(mu_t, lv_t, float(loss), float(-l_recon), float(-l_dyn), float(h), grads,
                   xs, xt, py, pt_mean, pt_lv, rls_status)


def filter_sequence(s: OracleState, y, u, eps, *, sgd=True, update=True, warm_up=False,
                    mu0=None, lv0=None, faithful_cost=False):
    """T successive filter calls, each fed the previous posterior (the inner loop of
    VJF.fit, vjf/model.py:252-261).  y (T,B,dy); u (T,B,du) or None; eps (T,2,B,dz)."""
    T = y.shape[0]
    mu, lv = mu0, lv0
    mus, lvs, losses = [], [], []
    for t in range(T):
        out = filter_step(s, y[t], None if u is None else u[t], mu, lv, eps[t, 0], eps[t, 1],
                          sgd=sgd, update=update, warm_up=warm_up, faithful_cost=faithful_cost)
        mu, lv = out.mu_t, out.lv_t
        mus.append(mu)
        lvs.append(lv)
        losses.append([out.loss, out.recon, out.dyn, out.entropy])
    return np.stack(mus), np.stack(lvs), np.asarray(losses)


# --------------------------------------------------------------------------- harness ("next" rows)
def blr_initialize(s: OracleState, xu, target, v, centroid_draw):
    """LinearRegression.initialize (vjf/module.py:144-150).  `centroid_draw` is the
    U(0,1) array the caller drew in place of nn.init.uniform_; it is mapped to U(-r, r)."""
    dt = s.dtype
    r = float(np.sqrt((xu * xu).sum(1)).max())
    s.centroid = ((centroid_draw * 2 - 1) * r).astype(dt)
    s.logwidth = np.full(s.n_rbf, math.log(r), dt)
    feat = rbf(xu, s.centroid, np.exp(s.logwidth))
    return rls(s, feat, target, v)


def rbfds_initialize(s: OracleState, xt, xs, ut, centroid_draw):
    """RBFDS.initialize (vjf/model.py:379-388)."""
    xu = nonecat(xs, ut)
    mse = np.mean((xt - xs) ** 2)
    blr_initialize(s, xu, xt - xs, mse, centroid_draw)
    d, _, _ = blr_predict(s, xu)
    mse = np.mean((xt - xs - d) ** 2)
    s.tr_logvar = np.asarray(np.log(mse), s.dtype)


def forecast(s: OracleState, x0, u, n_step, w_noise, state_noise=None):
    """RBFDS.forecast + decoder (vjf/model.py:321-324, 342-361; sampling branch of
    vjf/module.py:70-73).  w_noise (n_step, n_rbf, xdim) replaces randn_like(w);
    state_noise (n_step, B, xdim) or None replaces the optional process noise."""
    dt = s.dtype
    x0 = np.atleast_2d(np.asarray(x0, dt))
    x = np.empty((n_step + 1,) + x0.shape, dt)
    x[0] = x0
    sd = np.exp(0.5 * s.tr_logvar)
    for t in range(n_step):
        xu = nonecat(x[t], None if u is None else u[t])
        feat = rbf(xu, s.centroid, np.exp(s.logwidth))
        w = s.w_mean + s.w_chol @ w_noise[t]
        x[t + 1] = x[t] + feat @ w
        if state_noise is not None:
            x[t + 1] = x[t + 1] + state_noise[t] * sd
    return x, x @ s.dec_W.T + s.dec_b


# --------------------------------------------------------------------------- off-path Kalman form
def kalman_predict(x, L, A, Q, H):
    """vjf/kalman.py:15-50 with cholesky=True."""
    xhat = A @ x
    AL = A @ L
    Vhat = _cholesky(AL @ AL.T + Q)
    return H @ xhat, xhat, Vhat


def kalman_joseph_update(y, yhat, xhat, Lhat, H, R):
    """vjf/kalman.py:102-145 with cholesky=True."""
    e = y - yhat
    Vhat = Lhat @ Lhat.T
    HL = H @ Lhat
    S = HL @ HL.T + R
    L = _cholesky(S)
    G = sla.cho_solve((L, True), H @ Vhat).T
    x = xhat + G @ sla.cho_solve((L, True), e)
    ImKH = np.eye(Vhat.shape[0], dtype=Vhat.dtype) - G @ sla.cho_solve((L, True), H)
    ImKHL = ImKH @ Lhat
    KR = G @ sla.cho_solve((L, True), np.sqrt(R))
    V = ImKHL @ ImKHL.T + KR @ KR.T
    return x, _cholesky(V)


def blr_kalman(s: OracleState, xu, target, v, diffusion=0.0):
    """LinearRegression.kalman (vjf/module.py:114-142).  O(samples^3): off the live path."""
    assert diffusion >= 0.0
    dt = s.dtype
    n = s.w_mean.shape[0]
    eye = np.eye(n, dtype=dt)
    H = rbf(xu, s.centroid, np.exp(s.logwidth))
    R = np.eye(H.shape[0], dtype=dt) * dt.type(v)
    yhat, mhat, Vhat = kalman_predict(s.w_mean, s.w_chol, eye, diffusion * eye, H)
    s.w_mean, s.w_chol = kalman_joseph_update(target, yhat, mhat, Vhat, H, R)
