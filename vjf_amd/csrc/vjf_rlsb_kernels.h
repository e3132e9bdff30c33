// vjf_rlsb_kernels.h -- the RLS update (module.py:79-112) for feature counts that do not fit one compute unit's LDS
// (n_rbf > 224; BASELINE config E has 1000): the same right-looking blocked Cholesky, inverse and solves as the
// LDS-resident kernels, but with the matrix in global memory (L2-resident: 4 MB at n = 1000) and one launch per phase,
// so that the trailing updates and the block rows of the inverse spread over the whole chip.
//
//   GEMM + vjf_rlsb_prep_kernel   g = P W + Phi^T dx / v (module.py:94);  A := P + Phi^T Phi / v into a work buffer (module.py:96)
//   per block column k:     vjf_rlsb_diag_kernel   L_kk, L_kk^-1      (one wavefront: the rank-2 column chain)
//                           vjf_rlsb_panel_kernel  L_ik = A_ik L_kk^-T (one wavefront per block)
//                           vjf_rlsb_trail_kernel  A_ij -= L_ik L_jk^T (one wavefront per lower block)
//   inverse X = L^-1 (module.py:102): vjf_rlsb_inv_diag_kernel (diagonal blocks), then per level of a binary recursion over the
//                           block rows vjf_rlsb_inv_t_kernel / vjf_rlsb_inv_x_kernel (T = L_21 X_11, X_21 = -X_22 T)
//   two GEMMs (vjf_wide_gemm_kernel)        y = X g,  W = X^T y                      (module.py:101)
//   vjf_rlsb_final_kernel   w_chol = X^T, w_pchol = L, P += Phi^T Phi / v -- or, after a failed pivot, nothing but the status
// All block products on v_mfma_f32_32x32x2_f32 through the helpers of vjf_chol_kernel.h.  Matrices are padded to a
// multiple of 32 with the identity on the fly (loads) and never stored outside n x n.
#pragma once
#include <hip/hip_runtime.h>
#include "vjf_chol_kernel.h"
#include "vjf_plan.h"

struct VjfRlsbArgs {
    float* state;
    const float* red;
    float* Lw;           // (n, n)  work copy: P + Phi^T Phi / v -> L (w_pchol itself is written only when the factor is good)
    float* X;            // (n, n)  L^-1
    float* gbuf;         // (n, dz)
    float* ybuf;         // (n, dz)
    float* Dinv;         // nbl blocks (32x32 row-major): inverted diagonal blocks of L
    int* ok;             // [0]: 1 while every pivot so far was positive
    int k;               // block column (factorisation) or block row (inverse) of this launch
};

// 32x32 tile (bi, bj) of the n x n row-major matrix M -> XOR-swizzled LDS tile; outside the matrix: the identity
__device__ __forceinline__ void rlsb_tile_in(float* dst, const float* M, int n, int bi, int bj, int lane) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = lane + 64 * q, r = e >> 5, c = e & 31;
        const int gi = bi * 32 + r, gj = bj * 32 + c;
        dst[vsw(r, c)] = (gi < n && gj < n) ? M[(size_t)gi * n + gj] : (gi == gj ? 1.f : 0.f);
    }
}
// accumulator (row = vrow(reg, half), column = lane & 31) -> tile (bi, bj) of M, inside the matrix only
__device__ __forceinline__ void rlsb_acc_out(const vjf_f32x16& acc, float* M, int n, int bi, int bj, int lane, bool lower_only) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = vrow(r, h), gi = bi * 32 + row, gj = bj * 32 + c;
        if (gi < n && gj < n) M[(size_t)gi * n + gj] = (!lower_only || c <= row) ? acc[r] : 0.f;
    }
}

__global__ __launch_bounds__(256) void vjf_rlsb_prep_kernel(VjfPlan P, VjfRlsbArgs A) {
    const int n = P.n, dz = P.dz;
    float* S = A.state;
    const float inv_v = expf(-S[P.off[VJF_SLOT_TR_LOGVAR]]);
    const float* Pm = S + P.off[VJF_SLOT_W_PREC];
    float* Lm = A.Lw;
    const float* G = A.red + P.red_G;
    const float* FDX = A.red + P.red_FDX;
    const int gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    if (gid == 0) A.ok[0] = 1;
    for (int e = gid; e < n * dz; e += gsz) A.gbuf[e] = A.gbuf[e] + FDX[e] * inv_v;   // gbuf holds P W (the GEMM before this kernel)
    for (int e = gid; e < n * n; e += gsz) Lm[e] = Pm[e] + G[e] * inv_v;
}

__global__ __launch_bounds__(64) void vjf_rlsb_diag_kernel(VjfPlan P, VjfRlsbArgs A) {
    __shared__ __attribute__((aligned(16))) float s_d[1024], s_i[1024];
    if (A.ok[0] == 0) return;
    const int n = P.n, k = A.k, lane = threadIdx.x;
    float* Lm = A.Lw;
    rlsb_tile_in(s_d, Lm, n, k, k, lane);
    const bool good = potrf_inv_chain2(s_d, s_i, lane);
    if (!good) { if (lane == 0) A.ok[0] = 0; return; }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = lane + 64 * q, r = e >> 5, c = e & 31;
        const int gi = k * 32 + r, gj = k * 32 + c;
        if (gi < n && gj < n) Lm[(size_t)gi * n + gj] = c <= r ? s_d[vsw(r, c)] : 0.f;
        A.Dinv[(size_t)k * 1024 + e] = s_i[vsw(r, c)];
    }
}

__global__ __launch_bounds__(64) void vjf_rlsb_panel_kernel(VjfPlan P, VjfRlsbArgs A) {
    __shared__ __attribute__((aligned(16))) float s_a[1024], s_d[1024];
    if (A.ok[0] == 0) return;
    const int n = P.n, k = A.k, bi = k + 1 + (int)blockIdx.x, lane = threadIdx.x;
    float* Lm = A.Lw;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = lane + 64 * q, r = e >> 5, c = e & 31;
        const int gi = bi * 32 + r, gj = k * 32 + c;
        s_a[vsw(r, c)] = (gi < n && gj < n) ? Lm[(size_t)gi * n + gj] : 0.f;
        s_d[vsw(r, c)] = A.Dinv[(size_t)k * 1024 + e];
    }
    vjf_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    blk_mma<true>(acc, s_a, s_d, 1.f, lane);                   // A_ik L_kk^-T
    rlsb_acc_out(acc, Lm, n, bi, k, lane, false);
}

__global__ __launch_bounds__(64) void vjf_rlsb_trail_kernel(VjfPlan P, VjfRlsbArgs A) {
    __shared__ __attribute__((aligned(16))) float s_a[1024], s_b[1024];
    if (A.ok[0] == 0) return;
    const int n = P.n, k = A.k, lane = threadIdx.x;
    int ii = 0;
    const int t = blockIdx.x;
    while ((ii + 1) * (ii + 2) / 2 <= t) ++ii;
    const int jj = t - ii * (ii + 1) / 2, bi = k + 1 + ii, bj = k + 1 + jj;
    float* Lm = A.Lw;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int e = lane + 64 * q, r = e >> 5, c = e & 31;
        const int gj = k * 32 + c;
        s_a[vsw(r, c)] = (bi * 32 + r < n && gj < n) ? Lm[(size_t)(bi * 32 + r) * n + gj] : 0.f;
        s_b[vsw(r, c)] = (bj * 32 + r < n && gj < n) ? Lm[(size_t)(bj * 32 + r) * n + gj] : 0.f;
    }
    vjf_f32x16 acc;
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int gi = bi * 32 + vrow(r, h), gj = bj * 32 + c;
        acc[r] = (gi < n && gj < n) ? Lm[(size_t)gi * n + gj] : (gi == gj ? 1.f : 0.f);
    }
    blk_mma<true>(acc, s_a, s_b, -1.f, lane);                  // A_ij -= L_ik L_jk^T
    rlsb_acc_out(acc, Lm, n, bi, bj, lane, false);
}


// X = L^-1 by recursive doubling instead of block row after block row (nbl dependent launches with up to nbl - 1 chained
// block products each): with the diagonal blocks inverted, level l joins pairs of 2^l-tile diagonal blocks,
//     inv([[A, 0], [C, B]]) = [[A^-1, 0], [-B^-1 C A^-1, B^-1]],
// as two launches of independent tiles:  T = C A^-1  (vjf_rlsb_inv_t_kernel),  X21 = -B^-1 T  (vjf_rlsb_inv_x_kernel).
// 2 log2(nbl) launches, every tile of a level in parallel.  Tiles of `half` x `half` 32-blocks; workgroup = one output tile.
struct VjfRlsbLevel { int sb; };      // tiles per half at this level (1, 2, 4, ..)
__device__ __forceinline__ bool rlsb_level_tile(int nbl, int sb, int t, int& bi, int& bj, int& s0) {
    // output tiles of a level: for every pair starting at s0 = 2 sb p: rows [s0 + sb, min(s0 + 2 sb, nbl)), columns [s0, s0 + sb)
    const int per = sb * sb, p = t / per, r = t - p * per;
    s0 = 2 * sb * p;
    bi = s0 + sb + r / sb;
    bj = s0 + r % sb;
    return bi < nbl;
}
__global__ __launch_bounds__(64) void vjf_rlsb_inv_t_kernel(VjfPlan P, VjfRlsbArgs A, VjfRlsbLevel Lv, float* T) {
    __shared__ __attribute__((aligned(16))) float s_l[1024], s_x[1024];
    if (A.ok[0] == 0) return;
    const int n = P.n, nbl = (n + 31) / 32, lane = threadIdx.x;
    int bi, bj, s0;
    if (!rlsb_level_tile(nbl, Lv.sb, blockIdx.x, bi, bj, s0)) return;
    vjf_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = bj; k < s0 + Lv.sb; ++k) {                     // T_ij = sum_k L_ik X_kj, X11 block-lower: k >= bj
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = lane + 64 * q, r = e >> 5, c = e & 31;
            const int li = bi * 32 + r, lj = k * 32 + c, xi = k * 32 + r, xj = bj * 32 + c;
            s_l[vsw(r, c)] = (li < n && lj < n) ? A.Lw[(size_t)li * n + lj] : 0.f;
            s_x[vsw(r, c)] = (xi < n && xj < n) ? A.X[(size_t)xi * n + xj] : 0.f;
        }
        blk_mma<false>(acc, s_l, s_x, 1.f, lane);
    }
    rlsb_acc_out(acc, T, n, bi, bj, lane, false);
}
__global__ __launch_bounds__(64) void vjf_rlsb_inv_x_kernel(VjfPlan P, VjfRlsbArgs A, VjfRlsbLevel Lv, const float* T) {
    __shared__ __attribute__((aligned(16))) float s_l[1024], s_x[1024];
    if (A.ok[0] == 0) return;
    const int n = P.n, nbl = (n + 31) / 32, lane = threadIdx.x;
    int bi, bj, s0;
    if (!rlsb_level_tile(nbl, Lv.sb, blockIdx.x, bi, bj, s0)) return;
    vjf_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k = s0 + Lv.sb; k <= bi; ++k) {                    // X_ij = - sum_k X22_ik T_kj, X22 block-lower: k <= bi
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int e = lane + 64 * q, r = e >> 5, c = e & 31;
            const int xi = bi * 32 + r, xj = k * 32 + c, ti = k * 32 + r, tj = bj * 32 + c;
            s_l[vsw(r, c)] = (xi < n && xj < n) ? A.X[(size_t)xi * n + xj] : 0.f;
            s_x[vsw(r, c)] = (ti < n && tj < n) ? T[(size_t)ti * n + tj] : 0.f;
        }
        blk_mma<false>(acc, s_l, s_x, -1.f, lane);
    }
    rlsb_acc_out(acc, A.X, n, bi, bj, lane, false);
}
// X_ii = L_ii^-1 for every diagonal block (the base of the recursion)
__global__ __launch_bounds__(256) void vjf_rlsb_inv_diag_kernel(VjfPlan P, VjfRlsbArgs A) {
    if (A.ok[0] == 0) return;
    const int n = P.n, nbl = (n + 31) / 32;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nbl * 1024; e += gridDim.x * 256) {
        const int b = e >> 10, r = (e >> 5) & 31, c = e & 31;
        const int gi = b * 32 + r, gj = b * 32 + c;
        if (gi < n && gj < n) A.X[(size_t)gi * n + gj] = A.Dinv[e];
    }
}

// w_chol = X^T (module.py:102), zero halves of w_chol / w_pchol, P += Phi^T Phi / v; after a failed pivot only the status bit
// (the reference's fallback calls the removed torch.eig and raises, module.py:104-112: the RLS state stays as it was)
__global__ __launch_bounds__(256) void vjf_rlsb_final_kernel(VjfPlan P, VjfRlsbArgs A) {
    const int n = P.n;
    float* S = A.state;
    float* SC = S + P.off[VJF_SLOT_SCALARS];
    const int gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    if (A.ok[0] == 0) {
        if (gid == 0) vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED);
        return;
    }
    const float inv_v = expf(-S[P.off[VJF_SLOT_TR_LOGVAR]]);
    float* Pm = S + P.off[VJF_SLOT_W_PREC];
    float* Wc = S + P.off[VJF_SLOT_W_CHOL];
    float* Lm = S + P.off[VJF_SLOT_W_PCHOL];
    const float* G = A.red + P.red_G;
    for (int e = gid; e < n * n; e += gsz) {
        const int i = e / n, j = e - i * n;
        Wc[e] = (j >> 5) >= (i >> 5) ? A.X[(size_t)j * n + i] : 0.f;
        Lm[e] = (j >> 5) > (i >> 5) ? 0.f : A.Lw[e];             // w_pchol = L (module.py:99-100)
        Pm[e] += G[e] * inv_v;
    }
}
