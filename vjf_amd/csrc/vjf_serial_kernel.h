// vjf_serial_kernel.h -- K2: the once-per-step serial half of VJF.filter, one workgroup.
//
// Consumes the (all-reduced) reduce buffer and applies, in the reference's order:
//   finite guards + loss                               model.py:138-154
//   clip to +-1 and SGD on the optimised tensors       model.py:206-211
//   likelihood running variance (uses rho AFTER SGD)   likelihood.py:28-40, util.py:20-35
//   RLS: g = P W + Phi^T dx / v ; P += Phi^T Phi / v ; L = chol(P) ; W = P^-1 g ;
//        w_chol = inv(L^T)                             module.py:79-102
//   state-noise running variance from the residual     model.py:373-377
// The residual mean-square uses  sum|dx - Phi W|^2 = sum|dx|^2 - 2 tr(W^T Phi^T dx) + tr(W^T G W)
// (evaluated in fp64 from the fp32 statistics) instead of a second pass over the trials.
//
// Cholesky is right-looking, blocked by 32: the diagonal block is factored inside one wavefront
// with the row on the lane and v_readlane broadcasts (no LDS, no barriers), the panel is a
// per-row forward substitution against the LDS copy of the diagonal block, the trailing update
// reads the panel from LDS.  L^-1 is built block row by block row from the inverted diagonal
// blocks; W = L^-T (L^-1 g).
#pragma once
#include <hip/hip_runtime.h>
#include "vjf_plan.h"

#define VJF_K2_THREADS 512
#define VJF_NB 32

struct VjfSerialArgs {
    float* state;
    const float* red;
    float* work;          // see vjf_serial_work_floats
    float* loss4;         // device, may be null
    int B_total;
    unsigned flags;
    const float* E;       // non-null: this rank's rows [Phi | dx | 0] of ALL B_total trials (ldE floats each), and B_total < n -- the
                          //   state-noise residual is then formed directly, dx - Phi W (model.py:373-374), not from the statistics
};

static inline size_t vjf_serial_work_floats(const VjfPlan& P) {
    const size_t nblk = (P.n + VJF_NB - 1) / VJF_NB;
    return (size_t)P.n * P.n            // X = L^-1
           + 2 * (size_t)P.n * P.dz     // g, y
           + nblk * VJF_NB * VJF_NB;    // inverted diagonal blocks
}
static inline size_t vjf_serial_lds_floats(const VjfPlan& P) {
    size_t panel = (size_t)(P.n > VJF_NB ? P.n - VJF_NB : 0) * (VJF_NB + 1);
    size_t tbuf = (size_t)VJF_NB * P.n;
    size_t big = panel > tbuf ? panel : tbuf;
    return big + VJF_NB * (VJF_NB + 1) + 64;
}

__device__ __forceinline__ float rl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// In-wave Cholesky of a 32x32 block: lane r (< 32) holds row r in D[0..31] (lower part used).
// Returns false (wave-uniform) when a pivot is not > 0.
__device__ __forceinline__ bool chol32_inwave(float (&D)[VJF_NB], int r) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < VJF_NB; ++j) {
        const float djj = rl(D[j], j);
        if (!(djj > 0.f) || !(djj < 3.0e38f)) ok = false;
        const float s = sqrtf(djj);
        const float inv = 1.0f / s;
        D[j] = (r == j) ? s : (r > j ? D[j] * inv : 0.f);
#pragma unroll
        for (int c = j + 1; c < VJF_NB; ++c) {
            const float lcj = rl(D[j], c);
            D[c] = fmaf(-D[j], lcj, D[c]);
        }
    }
    return ok;
}

// In-wave inverse of a 32x32 lower-triangular block (row r on lane r): X = L^-1 (lower).
__device__ __forceinline__ void trinv32_inwave(const float (&Lr)[VJF_NB], float (&X)[VJF_NB], int r) {
#pragma unroll
    for (int c = 0; c < VJF_NB; ++c) X[c] = (r == c) ? 1.f : 0.f;
#pragma unroll
    for (int m = 0; m < VJF_NB; ++m) {
        const float inv = 1.0f / rl(Lr[m], m);
#pragma unroll
        for (int c = 0; c <= m; ++c) {
            if (r == m) X[c] *= inv;
            const float xm = rl(X[c], m);
            if (r > m) X[c] = fmaf(-Lr[m], xm, X[c]);
        }
    }
}

// Blocked Cholesky of the n x n matrix A (row-major, lower triangle referenced and overwritten
// with L, strict upper triangle zeroed).  Returns false when a pivot failed.  All threads call.
__device__ __forceinline__ bool vjf_chol_blocked(float* A, int n, float* lds) {
    const int tid = threadIdx.x;
    float* s_diag = lds;                                  // 32 x 33
    int* s_flag = (int*)(lds + VJF_NB * (VJF_NB + 1));    // [0]: ok
    float* s_panel = lds + VJF_NB * (VJF_NB + 1) + 64;    // rows x 33
    if (tid == 0) s_flag[0] = 1;
    __syncthreads();
    for (int k0 = 0; k0 < n; k0 += VJF_NB) {
        const int kb = min(VJF_NB, n - k0);
        if (tid < 64) {                                   // (a) diagonal block in wave 0
            const int r = tid & 31;
            float D[VJF_NB];
#pragma unroll
            for (int c = 0; c < VJF_NB; ++c)
                D[c] = (r < kb && c < kb) ? (c <= r ? A[(size_t)(k0 + r) * n + k0 + c] : 0.f) : (r == c ? 1.f : 0.f);
            const bool ok = chol32_inwave(D, r);
            if (tid < 32) {
#pragma unroll
                for (int c = 0; c < VJF_NB; ++c) {
                    s_diag[r * (VJF_NB + 1) + c] = D[c];
                    if (r < kb && c < kb) A[(size_t)(k0 + r) * n + k0 + c] = D[c];
                }
            }
            if (tid == 0 && !ok) s_flag[0] = 0;
        }
        __syncthreads();
        if (!s_flag[0]) return false;
        const int rows = n - k0 - kb;                     // rows below the diagonal block
        // (b) panel: row i of A[:, k0:k0+kb] <- row * L_kk^-T  (forward substitution per row)
        // (the row lives in LDS, not in 32 registers of a fully unrolled substitution: that form -- 528 multiply-adds with as many
        //  LDS operands in flight -- cost the three kernels that use this routine 1.3-1.5 KB per lane of scratch; same
        //  operations in the same order)
        for (int i = tid; i < rows; i += VJF_K2_THREADS) {
            float* arow = A + (size_t)(k0 + kb + i) * n + k0;
            float* prow = s_panel + (size_t)i * (VJF_NB + 1);
            for (int c = 0; c < VJF_NB; ++c) prow[c] = c < kb ? arow[c] : 0.f;
#pragma unroll 1
            for (int c = 0; c < VJF_NB; ++c) {
                const float* drow = s_diag + c * (VJF_NB + 1);
                float acc = prow[c];
#pragma unroll 4
                for (int m = 0; m < c; ++m) acc = fmaf(-prow[m], drow[m], acc);
                prow[c] = acc / drow[c];
            }
            for (int c = 0; c < kb; ++c) arow[c] = prow[c];
        }
        __syncthreads();
        // (c) trailing update of the lower triangle: A[i][j] -= sum_m Lp[i][m] Lp[j][m]
        const int tot = rows * rows;
        for (int e = tid; e < tot; e += VJF_K2_THREADS) {
            const int i = e / rows, j = e - i * rows;
            if (j > i) continue;
            float acc = 0.f;
#pragma unroll
            for (int m = 0; m < VJF_NB; ++m) acc = fmaf(s_panel[i * (VJF_NB + 1) + m], s_panel[j * (VJF_NB + 1) + m], acc);
            A[(size_t)(k0 + kb + i) * n + k0 + kb + j] -= acc;
        }
        __syncthreads();
    }
    // zero the strict upper triangle (w_pchol is a lower-triangular matrix, module.py:99)
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) {
        const int i = e / n, j = e - i * n;
        if (j > i) A[e] = 0.f;
    }
    __syncthreads();
    return true;
}

// X = L^-1 (n x n row-major, lower; strict upper zeroed).  Dinv: per-block inverted diagonals.
__device__ __forceinline__ void vjf_trinv_blocked(const float* L, float* X, float* Dinv, int n, float* lds) {
    const int tid = threadIdx.x;
    const int nblk = (n + VJF_NB - 1) / VJF_NB;
    const int wave = tid >> 6, lane = tid & 63;
    // inverted diagonal blocks: one wavefront per block, round-robin
    for (int blk = wave; blk < nblk; blk += VJF_K2_THREADS / 64) {
        const int k0 = blk * VJF_NB, kb = min(VJF_NB, n - k0), r = lane & 31;
        float Lr[VJF_NB], Xr[VJF_NB];
#pragma unroll
        for (int c = 0; c < VJF_NB; ++c)
            Lr[c] = (r < kb && c < kb) ? (c <= r ? L[(size_t)(k0 + r) * n + k0 + c] : 0.f) : (r == c ? 1.f : 0.f);
        trinv32_inwave(Lr, Xr, r);
        if (lane < 32) {
#pragma unroll
            for (int c = 0; c < VJF_NB; ++c) Dinv[(size_t)blk * VJF_NB * VJF_NB + r * VJF_NB + c] = Xr[c];
        }
    }
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) X[e] = 0.f;
    __syncthreads();
    for (int e = tid; e < nblk * VJF_NB * VJF_NB; e += VJF_K2_THREADS) {      // diagonal blocks of X
        const int blk = e / (VJF_NB * VJF_NB), rc = e - blk * VJF_NB * VJF_NB, r = rc / VJF_NB, c = rc - r * VJF_NB;
        const int gr = blk * VJF_NB + r, gc = blk * VJF_NB + c;
        if (gr < n && gc < n) X[(size_t)gr * n + gc] = Dinv[e];
    }
    __syncthreads();
    float* s_T = lds;                                      // 32 x n
    for (int ib = 1; ib < nblk; ++ib) {
        const int i0 = ib * VJF_NB, kb = min(VJF_NB, n - i0);
        // T[r][c] = sum_{k = blockstart(c)}^{i0-1} L[i0+r][k] X[k][c],   c < i0
        for (int e = tid; e < kb * i0; e += VJF_K2_THREADS) {
            const int r = e / i0, c = e - r * i0;
            const int kstart = (c / VJF_NB) * VJF_NB;
            const float* lrow = L + (size_t)(i0 + r) * n;
            float acc = 0.f;
            for (int k = kstart; k < i0; ++k) acc = fmaf(lrow[k], X[(size_t)k * n + c], acc);
            s_T[r * n + c] = acc;
        }
        __syncthreads();
        // X[i0+r][c] = - sum_{m<=r} Dinv_ib[r][m] T[m][c]
        const float* Di = Dinv + (size_t)ib * VJF_NB * VJF_NB;
        for (int e = tid; e < kb * i0; e += VJF_K2_THREADS) {
            const int r = e / i0, c = e - r * i0;
            float acc = 0.f;
            for (int m = 0; m <= r; ++m) acc = fmaf(Di[r * VJF_NB + m], s_T[m * n + c], acc);
            X[(size_t)(i0 + r) * n + c] = -acc;
        }
        __syncthreads();
    }
}

// LinearRegression.rls (module.py:79-102) from the sufficient statistics G = Phi^T Phi and
// FDX = Phi^T target:  g = shrink P W + FDX / v ;  P = shrink P + G / v ;  L = chol(P) ;
// W = P^-1 g ;  w_chol = L^-T.  work: X (n*n) | g (n*dz) | y (n*dz) | Dinv.  All threads call.
// Returns 0, or VJF_STATUS_RLS_FAILED with P, W, w_chol, L left as they were.
__device__ __forceinline__ unsigned vjf_rls_device(int n, int dz, float inv_v, float shrink, float* Pm, float* Wm, float* Wc, float* Lm,
                                   const float* G, const float* FDX, float* work, float* lds) {
    const int tid = threadIdx.x;
    float* X = work;
    float* gbuf = X + (size_t)n * n;
    float* ybuf = gbuf + (size_t)n * dz;
    float* Dinv = ybuf + (size_t)n * dz;
    for (int e = tid; e < n * dz; e += VJF_K2_THREADS) {                      // g
        const int i = e / dz, j = e - i * dz;
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc = fmaf(Pm[(size_t)i * n + k], Wm[(size_t)k * dz + j], acc);
        gbuf[e] = acc * shrink + FDX[e] * inv_v;
    }
    // Factor P_new in place in the w_pchol buffer.  If a pivot fails, the reference's fallback
    // (module.py:104-112) calls the removed torch.eig and raises; here the second pass of the loop
    // restores L = chol(P_old), the RLS state stays as it was and the failure is reported.
    bool failed = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        for (int e = tid; e < n * n; e += VJF_K2_THREADS) Lm[e] = attempt == 0 ? Pm[e] * shrink + G[e] * inv_v : Pm[e];
        __syncthreads();
        if (vjf_chol_blocked(Lm, n, lds)) break;
        failed = true;
        __syncthreads();
    }
    if (failed) return VJF_STATUS_RLS_FAILED;
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) Pm[e] = Pm[e] * shrink + G[e] * inv_v;
    vjf_trinv_blocked(Lm, X, Dinv, n, lds);
    for (int e = tid; e < n * dz; e += VJF_K2_THREADS) {                      // y = X g
        const int r = e / dz, j = e - r * dz;
        float acc = 0.f;
        for (int k = 0; k <= r; ++k) acc = fmaf(X[(size_t)r * n + k], gbuf[(size_t)k * dz + j], acc);
        ybuf[e] = acc;
    }
    __syncthreads();
    for (int e = tid; e < n * dz; e += VJF_K2_THREADS) {                      // W = X^T y  (cholesky_solve, module.py:101)
        const int k = e / dz, j = e - k * dz;
        float acc = 0.f;
        for (int r = k; r < n; ++r) acc = fmaf(X[(size_t)r * n + k], ybuf[(size_t)r * dz + j], acc);
        Wm[e] = acc;
    }
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) {                       // w_chol = inv(L^T) = X^T  (module.py:102)
        const int i = e / n, j = e - i * n;
        Wc[e] = X[(size_t)j * n + i];
    }
    __syncthreads();
    return 0u;
}

// Stand-alone LinearRegression.rls: one workgroup, statistics already reduced into `red`.
struct VjfRlsArgs {
    float* Pm; float* Wm; float* Wc; float* Lm;
    const float* G; const float* FDX; const float* v;
    float* work; unsigned* status;
    int n, dout; float shrink;
};
__global__ __launch_bounds__(VJF_K2_THREADS) void vjf_rls_kernel(VjfRlsArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned st = vjf_rls_device(A.n, A.dout, 1.0f / A.v[0], A.shrink, A.Pm, A.Wm, A.Wc, A.Lm, A.G, A.FDX, A.work, lds);
    if (threadIdx.x == 0 && A.status) A.status[0] = st;
}

// Stand-alone LinearRegression.kalman (vjf/module.py:114-142; vjf/kalman.py:15-50, 102-145) without the reference's
// (samples x samples) innovation covariance S = H Vhat H^T + v I, which is O(B^3) and unusable at B >> n.
// What the reference computes (A = I, Q = diffusion I, R = v I; Lhat = chol(w_chol w_chol^T + Q), Vhat = Lhat Lhat^T):
//     Gk = Vhat H^T S^-1                                  (kalman.py:134)
//     m  = m_prev + Gk S^-1 (y - H m_prev)                (kalman.py:135)   -- S^-1 enters twice: the effective gain is
//     V  = (I - Gk S^-1 H) Vhat (.)^T + (Gk S^-1) R (.)^T (kalman.py:137-142)  K = Vhat H^T S^-2, not the textbook one
// and w_chol <- chol(V).  This is the behaviour fixture G7 pins, so it is what is reproduced.  With the push-through identity
//     H^T S^-1 = (H^T H Vhat + v I)^-1 H^T = Lhat^-T N^-1 Lhat^T H^T,     N = Lhat^T (H^T H) Lhat + v I   (n x n, SPD)
// the gain is K = J H^T with J = Lhat N^-2 Lhat^T (symmetric), and everything reduces to n x n algebra on the Gram
// statistics G = H^T H, Fy = H^T y:
//     m = m_prev + J (Fy - G m_prev),    V = (I - J G) Vhat (I - J G)^T + v J G J.
// One workgroup; six n x n scratch matrices, Dinv as vjf_trinv_blocked wants it.
struct VjfKalmanArgs {
    float* Wm; float* Wc;
    const float* G; const float* Fy; const float* v;
    float* T[6]; float* Dinv;
    unsigned* status;
    int n, dout; float diffusion;
};
__global__ __launch_bounds__(VJF_K2_THREADS) void vjf_kalman_kernel(VjfKalmanArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, n = A.n, dz = A.dout;
    const float v = A.v[0];
    auto fail_out = [&]() { if (tid == 0 && A.status) A.status[0] = VJF_STATUS_RLS_FAILED; };
    // C (n x m) = op(X) op(Y): X, Y n x n row-major (tx / ty: use the transpose), or Y n x m when m != n
    auto mm = [&](float* C, const float* X, bool tx, const float* Y, bool ty, int m) {
        for (int e = tid; e < n * m; e += VJF_K2_THREADS) {
            const int i = e / m, j = e - i * m;
            float acc = 0.f;
            for (int k = 0; k < n; ++k)
                acc = fmaf(tx ? X[(size_t)k * n + i] : X[(size_t)i * n + k], ty ? Y[(size_t)j * n + k] : Y[(size_t)k * m + j], acc);
            C[e] = acc;
        }
        __syncthreads();
    };
    float* Vh = A.T[0]; float* Lh = A.T[1]; float* X1 = A.T[2]; float* X2 = A.T[3]; float* X3 = A.T[4]; float* X4 = A.T[5];
    mm(Vh, A.Wc, false, A.Wc, true, n);                            // Vhat = L L^T + Q   (kalman.py:41-44)
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) {
        if (e / n == e % n) Vh[e] += A.diffusion;
        Lh[e] = Vh[e];
    }
    __syncthreads();
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) if (e / n == e % n) Lh[e] = Vh[e];
    __syncthreads();
    if (!vjf_chol_blocked(Lh, n, lds)) { fail_out(); return; }     // Lhat (kalman.py:47)
    __syncthreads();
    mm(X1, A.G, false, Lh, false, n);                              // G Lhat
    mm(X2, Lh, true, X1, false, n);                                // N = Lhat^T G Lhat + v I
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) if (e / n == e % n) X2[e] += v;
    __syncthreads();
    if (!vjf_chol_blocked(X2, n, lds)) { fail_out(); return; }
    __syncthreads();
    vjf_trinv_blocked(X2, X1, A.Dinv, n, lds);                     // X1 = C^-1, N = C C^T
    __syncthreads();
    mm(X2, X1, true, X1, false, n);                                // N^-1 = C^-T C^-1
    mm(X1, X2, false, X2, false, n);                               // N^-2
    mm(X3, Lh, false, X1, false, n);                               // Lhat N^-2
    mm(X4, X3, false, Lh, true, n);                                // J = Lhat N^-2 Lhat^T
    // m = m_prev + J (Fy - G m_prev)
    for (int e = tid; e < n * dz; e += VJF_K2_THREADS) {
        const int i = e / dz, j = e - i * dz;
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc = fmaf(A.G[(size_t)i * n + k], A.Wm[(size_t)k * dz + j], acc);
        X1[e] = A.Fy[e] - acc;
    }
    __syncthreads();
    for (int e = tid; e < n * dz; e += VJF_K2_THREADS) {
        const int i = e / dz, j = e - i * dz;
        float acc = 0.f;
        for (int k = 0; k < n; ++k) acc = fmaf(X4[(size_t)i * n + k], X1[(size_t)k * dz + j], acc);
        X2[e] = A.Wm[e] + acc;                                     // new mean, parked until the covariance is known to be good
    }
    __syncthreads();
    mm(X3, X4, false, A.G, false, n);                              // J G
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) X1[e] = ((e / n == e % n) ? 1.f : 0.f) - X3[e];   // I - J G
    __syncthreads();
    mm(Lh, X1, false, Vh, false, n);                               // (I - J G) Vhat          (Lhat is no longer needed)
    mm(Vh, Lh, false, X1, true, n);                                // (I - J G) Vhat (I - J G)^T
    mm(Lh, X3, false, X4, false, n);                               // J G J   (J symmetric)
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) Vh[e] = fmaf(v, Lh[e], Vh[e]);
    __syncthreads();
    // symmetrise against rounding before the factorisation reads the lower triangle
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) { const int i = e / n, j = e - i * n; Lh[e] = 0.5f * (Vh[e] + Vh[(size_t)j * n + i]); }
    __syncthreads();
    if (!vjf_chol_blocked(Lh, n, lds)) { fail_out(); return; }     // w_chol = chol(V)   (kalman.py:144-145)
    __syncthreads();
    for (int e = tid; e < n * dz; e += VJF_K2_THREADS) A.Wm[e] = X2[e];
    for (int e = tid; e < n * n; e += VJF_K2_THREADS) A.Wc[e] = Lh[e];
    if (tid == 0 && A.status) A.status[0] = 0u;
}

__global__ __launch_bounds__(VJF_K2_THREADS) void vjf_serial_kernel(VjfPlan P, VjfSerialArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ double s_dred[VJF_K2_THREADS / 64];
    __shared__ float s_bcast[4];
    const int tid = threadIdx.x;
    float* S = A.state;
    float* SC = S + P.off[VJF_SLOT_SCALARS];
    const float* RSC = A.red + P.red_SCA;                      // the loss sums; sum |dx|^2 sits with the RLS statistics
    const int n = P.n, dz = P.dz;
    const bool do_sgd = A.flags & VJF_FLAG_SGD, do_upd = A.flags & VJF_FLAG_UPDATE, warm = A.flags & VJF_FLAG_WARM_UP;
    const float Bf = (float)A.B_total, invB = 1.0f / Bf;

    // ---- losses with the finite guards (model.py:138-149); every thread evaluates the same scalars
    float l_recon = RSC[RS_LRECON] * invB, l_dyn = RSC[RS_LDYN] * invB, ent = RSC[RS_ENT] * invB;
    const float sse_y = RSC[RS_SSEY], sdx2 = A.red[P.red_SC + RS_SDX2];
    const bool ok_r = isfinite(l_recon), ok_d = isfinite(l_dyn), ok_h = isfinite(ent);
    if (!ok_r) l_recon = 0.f;
    if (!ok_d) l_dyn = 0.f;
    if (!ok_h) ent = 0.f;
    float loss = l_recon - ent;
    if (!warm) loss += l_dyn;
    unsigned st = (ok_r ? 0u : VJF_STATUS_NONFINITE_RECON) | (ok_d ? 0u : VJF_STATUS_NONFINITE_DYN) | (ok_h ? 0u : VJF_STATUS_NONFINITE_ENT);
    if (tid == 0 && A.loss4) { A.loss4[0] = loss; A.loss4[1] = -l_recon; A.loss4[2] = -l_dyn; A.loss4[3] = ent; }
    // A non-finite component drops out of the reference's graph; its gradient cannot be separated
    // after the fact, so the SGD step is skipped and the status bit reports it.
    const bool grad_ok = ok_r && ok_h && (warm || ok_d);

    // ---- clip + SGD (model.py:210-211); groups: likelihood, decoder, transition (none), recognition
    const float lr_lik = SC[VJF_SC_LR_LIK], lr_dec = SC[VJF_SC_LR_DEC], lr_rec = SC[VJF_SC_LR_REC];
    const bool freeze = SC[VJF_SC_FREEZE_DEC] != 0.f;
    float rho = S[P.off[VJF_SLOT_LIK_LOGVAR]];
    float sig = S[P.off[VJF_SLOT_TR_LOGVAR]];
    float n_lik = SC[VJF_SC_N_LIK], n_tr = SC[VJF_SC_N_TR];
    __syncthreads();      // everyone has read the scalars before thread 0 rewrites them
    if (do_sgd && grad_ok) {
        const int dec_rel = P.dec_off - P.train_off;
        for (int i = tid; i < P.train_len; i += VJF_K2_THREADS) {
            const bool dec = i >= dec_rel;
            if (dec && freeze) continue;
            float g = A.red[i] * invB;
            g = fminf(fmaxf(g, -1.f), 1.f);
            S[P.train_off + i] -= (dec ? lr_dec : lr_rec) * g;
        }
        if (P.lik == VJF_LIK_GAUSSIAN) {
            float g = 0.5f * ((float)P.dy - expf(-rho) * sse_y * invB);
            g = fminf(fmaxf(g, -1.f), 1.f);
            rho -= lr_lik * g;
        }
    }
    // ---- likelihood running variance (likelihood.py:28-40)
    if (do_upd && P.lik == VJF_LIK_GAUSSIAN) {
        const float mse = sse_y / (Bf * (float)P.dy);
        const float acc = fminf(n_lik, 1000.f), tot = acc + Bf;
        const float var = (acc / tot) * expf(rho) + (Bf / tot) * mse;
        rho = logf(var);
        n_lik = tot;
    }

    // ---- transition update (model.py:363-377)
    float* Wm = S + P.off[VJF_SLOT_W_MEAN];
    float* Wc = S + P.off[VJF_SLOT_W_CHOL];
    float* Pm = S + P.off[VJF_SLOT_W_PREC];
    float* Lm = S + P.off[VJF_SLOT_W_PCHOL];
    const float* G = A.red + P.red_G;
    const float* FDX = A.red + P.red_FDX;
    if (do_upd) {
        if (!warm)                                              // LinearRegression.rls, module.py:79-102
            st |= vjf_rls_device(n, dz, expf(-sig), 1.0f, Pm, Wm, Wc, Lm, G, FDX, A.work, lds);
        // residual mean square with the (possibly new) W, fp64 accumulation
        double part = 0.0;
        if (A.E) {
            // few trials against many features: the weights reproduce dx almost exactly and the quadratic form below loses the
            // residual under the rounding of its terms; the difference in fp32 as the reference forms it, the sum in fp64
            for (int e = tid; e < A.B_total * dz; e += VJF_K2_THREADS) {
                const int b = e / dz, c = e - b * dz;
                const float* row = A.E + (size_t)b * P.ldE;
                float r = 0.f;
                for (int k = 0; k < n; ++k) r = fmaf(row[k], Wm[(size_t)k * dz + c], r);
                r = row[n + c] - r;
                part += (double)r * (double)r;
            }
        } else {
        for (int e = tid; e < n * n; e += VJF_K2_THREADS) {
            const int i = e / n, j = e - i * n;
            float d = 0.f;
            for (int c = 0; c < dz; ++c) d = fmaf(Wm[(size_t)i * dz + c], Wm[(size_t)j * dz + c], d);
            part += (double)G[e] * (double)d;
        }
        for (int e = tid; e < n * dz; e += VJF_K2_THREADS) part -= 2.0 * (double)Wm[e] * (double)FDX[e];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if ((tid & 63) == 0) s_dred[tid >> 6] = part;
        __syncthreads();
        if (tid == 0) {
            double t = A.E ? 0.0 : (double)sdx2;
            for (int w = 0; w < VJF_K2_THREADS / 64; ++w) t += s_dred[w];
            if (t < 0.0) t = 0.0;
            const float mse = (float)(t / ((double)Bf * (double)dz));
            const float acc = fminf(n_tr, 500.f), tot = acc + Bf;
            const float var = (acc / tot) * expf(sig) + (Bf / tot) * mse;
            s_bcast[0] = logf(var);
            s_bcast[1] = tot;
        }
        __syncthreads();
        sig = s_bcast[0];
        n_tr = s_bcast[1];
    }
    if (tid == 0) {
        S[P.off[VJF_SLOT_LIK_LOGVAR]] = rho;
        S[P.off[VJF_SLOT_TR_LOGVAR]] = sig;
        SC[VJF_SC_N_LIK] = n_lik;
        SC[VJF_SC_N_TR] = n_tr;
        if (st) SC[VJF_SC_STATUS] = (float)((unsigned)SC[VJF_SC_STATUS] | st);
        if (do_upd && !warm && !(st & VJF_STATUS_RLS_FAILED)) SC[VJF_SC_TRI_CLEAN] = 1.f;   // w_chol / w_pchol fully rewritten
    }
}
