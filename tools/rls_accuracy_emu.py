"""CPU emulation (numpy, fp32) of the blocked RLS update of vjf_rlsb_kernels.h, to compare the accuracy of its variants on the
ill-conditioned configurations of tools/fuzz_parity.py (B << n) without a GPU:

    python tools/rls_accuracy_emu.py [case ...]

For each pinned hard case it runs the fp64 oracle for three steps, takes the precision matrix A = P + G / v each RLS update sees
(rounded to fp32), and factors / inverts it in fp32 with
  lapack   : numpy / scipy in fp32 (what the fp32 oracle -- the reference's arithmetic -- does)
  explicit : 32 x 32 diagonal blocks factored together with their inverse, panels L_ik = T Dinv^T, X = L^-1 block row by block
             row through Dinv (the device algorithm of round 2)
  variants : see VARIANTS
and prints the largest distance of L and of w_chol = L^-T from the fp64 factor of the same fp32 matrix."""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.linalg as sla

f32 = np.float32
NB = 32


def potrf_inv_block(D):
    """chol of a 32x32 tile with the inverse of its factor, column by column in fp32 (potrf_inv_chain2's arithmetic up to the
    pairing of columns: pivot -> rsqrt -> scaled column -> rank-1 update of the tile and of the inverse's accumulator)."""
    n = D.shape[0]
    A = D.astype(f32).copy()
    R = np.eye(n, dtype=f32)
    L = np.zeros((n, n), f32)
    X = np.zeros((n, n), f32)
    for j in range(n):
        d = A[j, j]
        if not d > 0:
            raise np.linalg.LinAlgError("pivot")
        s = f32(1.0) / np.sqrt(d, dtype=f32)
        l = (A[:, j] * s).astype(f32)
        l[:j] = 0
        x = (R[j, :] * s).astype(f32)
        L[:, j] = l
        X[j, :] = x
        A = (A - np.outer(l, l).astype(f32)).astype(f32)
        R = (R - np.outer(l, x).astype(f32)).astype(f32)
    return L, np.tril(X)


def mm(a, b):
    return (a.astype(f32) @ b.astype(f32)).astype(f32)


def trsm_right_lt(T, Lkk):
    """X Lkk^T = T by substitution in fp32 (column by column)."""
    n = Lkk.shape[0]
    X = np.zeros_like(T, dtype=f32)
    Tm = T.astype(f32).copy()
    for c in range(n):
        X[:, c] = Tm[:, c] / Lkk[c, c]
        if c + 1 < n:
            Tm[:, c + 1:] = (Tm[:, c + 1:] - np.outer(X[:, c], Lkk[c + 1:, c]).astype(f32)).astype(f32)
    return X


def colsum(Aik, Li, Lk, k, order):
    """A_ik - sum_{j<k} L_ij L_kj^T in fp32 the way the kernels order it.
      'gemm'    : one fp32 GEMM over all finished columns, subtracted from A at the end (numpy's blocking)
      'strided' : round 2's kernels -- eight partial sums from zero over block columns w, w + 8, .., folded, the term of column k - 1
                  added, the whole subtracted from A at the end
      'ordered' : eight CONTIGUOUS ranges of block columns; the first range's accumulator starts from A and subtracts its terms in
                  ascending order (the large, cancelling terms of the leading columns meet A at once); the other ranges' partial sums
                  are subtracted in range order"""
    if k == 0:
        return Aik.copy()
    if order == "gemm":
        return (Aik - mm(Li[:, :k * NB], Lk[:, :k * NB].T)).astype(f32)
    blk = lambda M, j: M[:, j * NB:(j + 1) * NB]
    if order == "strided":
        parts = []
        for w in range(8):
            acc = np.zeros((NB, NB), f32)
            for j in range(w, k - 1, 8):
                acc = (acc + mm(blk(Li, j), blk(Lk, j).T)).astype(f32)
            parts.append(acc)
        for w in range(4):
            parts[w] = (parts[w] + parts[w + 4]).astype(f32)
        pacc = (((parts[0] + parts[1]).astype(f32) + parts[2]).astype(f32) + parts[3]).astype(f32)
        last = mm(blk(Li, k - 1), blk(Lk, k - 1).T)
        return (Aik - (pacc + last).astype(f32)).astype(f32)
    per = (k + 7) // 8
    acc = Aik.astype(f32).copy()
    parts = []
    for w in range(8):
        a = acc if w == 0 else np.zeros((NB, NB), f32)
        for j in range(w * per, min(k, (w + 1) * per)):
            a = (a - mm(blk(Li, j), blk(Lk, j).T)).astype(f32)
        parts.append(a)
    out = parts[0]
    for w in range(1, 8):
        out = (out + parts[w]).astype(f32)
    return out


def blocked(A32, *, panel="inverse", newton_diag=False, refine_panel=False, inv="blocks", newton_x=False, order="strided"):
    n = A32.shape[0]
    nbl = (n + NB - 1) // NB
    npad = nbl * NB
    A = np.eye(npad, dtype=f32)
    A[:n, :n] = A32
    L = np.zeros((npad, npad), f32)
    Dinv = []
    sl = lambda i: slice(i * NB, (i + 1) * NB)
    for k in range(nbl):
        D = colsum(A[sl(k), sl(k)], L[sl(k)], L[sl(k)], k, order)
        Lkk, Xkk = potrf_inv_block(D)
        if newton_diag:                                   # X <- X + X (I - L X)
            Rm = (np.eye(NB, dtype=f32) - mm(Lkk, Xkk)).astype(f32)
            Xkk = np.tril((Xkk + mm(Xkk, Rm)).astype(f32))
        L[sl(k), sl(k)] = Lkk
        Dinv.append(Xkk)
        for i in range(k + 1, nbl):
            T = colsum(A[sl(i), sl(k)], L[sl(i)], L[sl(k)], k, order)
            if panel == "inverse":
                Lik = mm(T, Xkk.T)
                if refine_panel:                          # L_ik += (T - L_ik L_kk^T) L_kk^-T
                    Rm = (T - mm(Lik, Lkk.T)).astype(f32)
                    Lik = (Lik + mm(Rm, Xkk.T)).astype(f32)
            else:
                Lik = trsm_right_lt(T, Lkk)
            L[sl(i), sl(k)] = Lik
    X = np.zeros((npad, npad), f32)
    if inv == "blocks":
        for r in range(nbl):
            X[sl(r), sl(r)] = Dinv[r]
            for j in range(r):
                S = mm(L[sl(r), j * NB:r * NB], X[j * NB:r * NB, sl(j)])
                X[sl(r), sl(j)] = (-mm(Dinv[r], S)).astype(f32)
    if newton_x:                                          # X <- X + X (I - L X), triangular
        Rm = np.tril((np.eye(npad, dtype=f32) - mm(L, X)).astype(f32))
        X = np.tril((X + mm(X, Rm)).astype(f32))
    return L[:n, :n], X[:n, :n]


VARIANTS = {
    "round 2 (strided sums, explicit)": dict(),
    "strided + trsm panels": dict(panel="trsm"),
    "gemm sums, explicit": dict(order="gemm"),
    "ordered sums, explicit": dict(order="ordered"),
    "ordered + newton_diag": dict(order="ordered", newton_diag=True),
    "ordered + refine_panel": dict(order="ordered", refine_panel=True),
    "ordered + trsm panels": dict(order="ordered", panel="trsm"),
    "ordered + newton_x": dict(order="ordered", newton_x=True),
}


def matrices_of_case(case, desc, T=3):
    """A = P + G / v of every RLS update of the case (fp64 oracle run), as the oracle forms it."""
    import torch
    from oracle import vjf_oracle as orc
    dz, dy, du, n, hidden, lik, B = (desc[k] for k in ("dz", "dy", "du", "n", "hidden", "lik", "B"))
    torch.manual_seed(100 + case)
    # the model's initial parameters as vjf_amd.VJF.make_model draws them (CPU generator, reference order) -- without a GPU:
    # build the oracle state from the same draws through the host mirror's constructors on the CPU
    import vjf_amd
    m = vjf_amd.VJF.make_model(dy, dz, du, n, hidden, likelihood=lik, lr=1e-3)
    from tests.helpers import load_oracle_state
    s = load_oracle_state(m, np.float64)
    g = torch.Generator().manual_seed(200 + case)
    y = torch.poisson(torch.exp(0.5 * torch.randn(T, B, dy, generator=g) - 0.5), generator=g) if lik == "poisson" else torch.randn(T, B, dy, generator=g)
    u = torch.randn(T, B, du, generator=g) if du else None
    eps = torch.randn(T, 2, B, dz, generator=g)
    mats = []
    mu = lv = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for t in range(T):
            un = None if u is None else u[t].numpy()
            o = orc.filter_step(s, y[t].numpy(), un, mu, lv, eps[t, 0].numpy(), eps[t, 1].numpy())
            mu, lv = o.mu_t, o.lv_t
            mats.append((s.w_precision.copy(), s.w_mean.copy()))
    return mats


def main():
    from tools.fuzz_parity import KNOWN_HARD
    want = [int(a) for a in sys.argv[1:]] or [39]
    for case, desc in KNOWN_HARD:
        if case not in want:
            continue
        mats = matrices_of_case(case, desc)
        for t, (A64, W64) in enumerate(mats):
            A32 = A64.astype(f32)
            g32 = (A64 @ W64).astype(f32)                  # the right-hand side the update solved (g = A W)
            Wr = sla.cho_solve((np.linalg.cholesky(A32.astype(np.float64)), True), g32.astype(np.float64))
            Lr = np.linalg.cholesky(A32.astype(np.float64))
            Xr = sla.solve_triangular(Lr, np.eye(len(Lr)), lower=True)
            ev = np.linalg.eigvalsh(A32.astype(np.float64))
            print(f"case {case} step {t}: n={len(Lr)} cond(A)={ev[-1] / ev[0]:.2e} |L|max={np.abs(Lr).max():.2e} |X|max={np.abs(Xr).max():.2e}")
            Ll = np.linalg.cholesky(A32)
            Xl = sla.solve_triangular(Ll, np.eye(len(Ll), dtype=f32), lower=True)
            def wline(W):
                dW = W.astype(np.float64) - Wr
                return f"W {np.abs(dW).max():.3e}  |A dW|/|g| {np.abs(A32.astype(np.float64) @ dW).max() / np.abs(g32).max():.3e}"
            Wl = sla.cho_solve((Ll, True), g32)
            print(f"   {'lapack fp32':38s} L {np.abs(Ll - Lr).max():.3e}   X {np.abs(Xl - Xr).max():.3e}   {wline(Wl)}")
            for name, kw in VARIANTS.items():
                L, X = blocked(A32, **kw)
                W = mm(X.T, mm(X, g32))                    # W = X^T (X g)
                r = (g32 - mm(A32, W)).astype(f32)         # one step of refinement: W += X^T X (g - A W)
                W2 = (W + mm(X.T, mm(X, r))).astype(f32)
                print(f"   {name:38s} L {np.abs(L - Lr).max():.3e}   X {np.abs(X - Xr).max():.3e}   {wline(W)}   refined: {wline(W2)}")


if __name__ == "__main__":
    main()
