// vjf_post_kernel.h -- everything of the RLS update that follows the Cholesky factor and does NOT
// have to run on one compute unit (module.py:101-102, model.py:373-377):
//
//   vjf_rls_post_kernel   2*nbl + 1 workgroups.  Workgroup (j, half) solves  L X = I  for 16 columns of
//       block column j by blocked forward substitution against L and the inverted diagonal blocks
//       and writes them, transposed, into w_chol = L^-T (module.py:102).  The last workgroup solves
//       L y = g the same way, then L^T W = y backwards, and writes w_mean (module.py:101).
//       Products run on v_mfma_f32_16x16x4_f32 with the right-hand side columns on the MFMA column.
//   vjf_resid_kernel      partial sums of  sum|dx|^2 - 2 tr(W^T FDX) + tr(W^T G W)  (model.py:373-374),
//       G symmetric: lower triangle only, fp64 partials, one per workgroup.
//   vjf_sigma_kernel      one thread: fixed-order sum of the partials, state-noise running variance
//       (model.py:375-377).
#pragma once
#include <hip/hip_runtime.h>
#include "vjf_plan.h"
#include "vjf_trial_mfma_kernel.h"   // vjf_f32x4

#define VJF_POST_THREADS 512
#define VJF_POST_KPAR 4                // wavefronts = 2 row tiles x 4 interleaved block sums
#define VJF_POST_LDB 33               // padded leading dimension of a 32x32 block in LDS
#define VJF_POST_LDX 17               // right-hand sides: [row][16 columns + 1 pad]
#define VJF_RESID_BLOCKS 64

struct VjfPostArgs {
    float* state;
    const float* dinv;      // nbl blocks (32x32 row-major) of inverted diagonal blocks, from vjf_chol_lds_kernel
    const float* gbuf;      // (n, dz) g
    const int* ok;          // device flag written by the Cholesky kernel: 1 = factor valid
};

static inline size_t vjf_post_lds_bytes(const VjfPlan& P) {
    const int nbl = (P.n + 31) / 32;
    return ((size_t)(nbl * (nbl - 1) / 2 + nbl) * 32 * VJF_POST_LDB + (size_t)nbl * 32 * VJF_POST_LDX + VJF_POST_KPAR * 32 * VJF_POST_LDX + 16) * 4;
}

// acc(row = 4*(lane>>4)+r of the 16-row tile, col = lane&15) += sum_m A(tile row, m) * B[m][col], m < 32
template <class FA>
__device__ __forceinline__ void post_mma32(vjf_f32x4& acc, const float* Bs, int lane, FA fa) {
    const int i = lane & 15, kk = lane >> 4;
    float a[8], b[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { a[s] = fa(i, 4 * s + kk); b[s] = Bs[(4 * s + kk) * VJF_POST_LDX + i]; }
    __builtin_amdgcn_sched_barrier(0);
    vjf_f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};                     // two independent chains: the MFMAs issue back to back
#pragma unroll
    for (int s = 0; s < 8; s += 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s + 1], b[s + 1], acc1, 0, 0, 0);
    }
    acc += acc1;
}

__global__ __launch_bounds__(VJF_POST_THREADS) void vjf_rls_post_kernel(VjfPlan P, VjfPostArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (A.ok[0] == 0) return;                                  // factorisation failed: RLS state stays as it was
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.n, dz = P.dz, nbl = (n + 31) / 32, ntri = nbl * (nbl + 1) / 2;
    constexpr int LB = VJF_POST_LDB, LX = VJF_POST_LDX;
    float* s_L = lds;                                          // strictly-lower blocks [32][33] of L (bi > bj)
    float* s_D = s_L + (size_t)(ntri - nbl) * 32 * LB;         // nbl blocks [32][33]: inverted diagonal blocks
    float* s_x = s_D + (size_t)nbl * 32 * LB;                  // [npad][17] right-hand sides -> solution
    float* s_t = s_x + (size_t)nbl * 32 * LX;                  // VJF_POST_KPAR x [32][17] partial sums
    const float* S = A.state;
    const float* Lm = S + P.off[VJF_SLOT_W_PCHOL];
    const bool solve = (int)blockIdx.x == 2 * nbl;             // the y / W workgroup
    const int j0 = solve ? 0 : (int)blockIdx.x >> 1;           // first block row of the substitution
    const int c0 = solve ? 0 : 16 * ((int)blockIdx.x & 1);
    auto tri = [](int bi, int bj) { return bi * (bi - 1) / 2 + bj; };       // strictly lower: bi > bj

    // ---- stage L (blocks with row > col >= j0) and the inverted diagonal blocks: 8 float4 loads in flight per thread
    {
        const int nlow = ntri - nbl;                                           // strictly-lower blocks, index tri(bi, bj)
        const int nchunk = (nlow + nbl) * 256;                                 // then the nbl Dinv blocks
        for (int e0 = tid; e0 < nchunk; e0 += 8 * VJF_POST_THREADS) {
            float4 v[8];
            float* dst[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + q * VJF_POST_THREADS;
                dst[q] = nullptr;
                v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < nchunk) {
                    const int b = e >> 8, r = (e >> 3) & 31, c4 = (e & 7) * 4;
                    if (b < nlow) {
                        int bi = 1;
                        while ((bi + 1) * bi / 2 <= b) ++bi;                   // tri(bi, 0) <= b < tri(bi + 1, 0)
                        const int bj = b - bi * (bi - 1) / 2;
                        if (bj >= j0) {
                            const int gi = bi * 32 + r, gj = bj * 32 + c4;
                            if (gi < n && gj < n) v[q] = *reinterpret_cast<const float4*>(Lm + (size_t)gi * n + gj);
                            dst[q] = s_L + ((size_t)b * 32 + r) * LB + c4;
                        }
                    } else if (b - nlow >= j0) {
                        v[q] = *reinterpret_cast<const float4*>(A.dinv + (size_t)(b - nlow) * 1024 + r * 32 + c4);
                        dst[q] = s_D + ((size_t)(b - nlow) * 32 + r) * LB + c4;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (dst[q]) { dst[q][0] = v[q].x; dst[q][1] = v[q].y; dst[q][2] = v[q].z; dst[q][3] = v[q].w; }
        }
    }
    // right-hand side: 16 columns of the identity (inverse) or g padded to 16 columns (solve)
    for (int e = tid; e < nbl * 32 * 16; e += VJF_POST_THREADS) {
        const int r = e >> 4, c = e & 15;
        float v;
        if (solve) v = (r < n && c < dz) ? A.gbuf[(size_t)r * dz + c] : 0.f;
        else v = (r == j0 * 32 + c0 + c) ? 1.f : 0.f;
        s_x[r * LX + c] = v;
    }
    __syncthreads();

    const int tile = wave & 1, kpar = wave >> 1;               // 16-row tile of the block row; blocks k = first + kpar, + KPAR, ..
    constexpr int KP = VJF_POST_KPAR;
    // ---- forward substitution  Y_i = Dinv_i (R_i - sum_{k<i} L_ik Y_k),  i = j0 .. nbl-1, in place in s_x
    for (int bi = j0; bi < nbl; ++bi) {
        vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = j0 + kpar; k < bi; k += KP) {
            const float* Lb = s_L + (size_t)tri(bi, k) * 32 * LB;
            post_mma32(acc, s_x + (size_t)k * 32 * LX, lane, [&](int i, int m) { return Lb[(16 * tile + i) * LB + m]; });
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_t[(kpar * 32 + 16 * tile + 4 * (lane >> 4) + r) * LX + (lane & 15)] = acc[r];
        __syncthreads();
        for (int e = tid; e < 32 * 16; e += VJF_POST_THREADS) {                // T = R_i - T0 - T1  (into s_t[0])
            const int r = e >> 4, c = e & 15;
            s_t[r * LX + c] = s_x[(bi * 32 + r) * LX + c] - ((s_t[r * LX + c] + s_t[(32 + r) * LX + c]) + (s_t[(64 + r) * LX + c] + s_t[(96 + r) * LX + c]));
        }
        __syncthreads();
        if (wave < 2) {
            const float* Db = s_D + (size_t)bi * 32 * LB;
            vjf_f32x4 y = {0.f, 0.f, 0.f, 0.f};
            post_mma32(y, s_t, lane, [&](int i, int m) { return Db[(16 * wave + i) * LB + m]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) s_x[(bi * 32 + 16 * wave + 4 * (lane >> 4) + r) * LX + (lane & 15)] = y[r];
        }
        __syncthreads();
    }

    if (!solve) {
        // ---- w_chol[(j0*32 + c0 + c)][i] = X[i][c]: rows of w_chol, contiguous over i  (module.py:102)
        float* Wc = A.state + P.off[VJF_SLOT_W_CHOL];
        const int first = j0 * 32;
        for (int c = wave; c < 16; c += VJF_POST_THREADS / 64) {
            const int gc = j0 * 32 + c0 + c;
            if (gc >= n) continue;
            for (int i = first + lane; i < n; i += 64) Wc[(size_t)gc * n + i] = s_x[i * LX + c];
        }
        return;
    }
    // ---- backward substitution  W_i = Dinv_i^T (Y_i - sum_{k>i} L_ki^T W_k),  i = nbl-1 .. 0  (module.py:101)
    for (int bi = nbl - 1; bi >= 0; --bi) {
        vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int k = bi + 1 + kpar; k < nbl; k += KP) {
            const float* Lb = s_L + (size_t)tri(k, bi) * 32 * LB;
            post_mma32(acc, s_x + (size_t)k * 32 * LX, lane, [&](int i, int m) { return Lb[m * LB + 16 * tile + i]; });
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_t[(kpar * 32 + 16 * tile + 4 * (lane >> 4) + r) * LX + (lane & 15)] = acc[r];
        __syncthreads();
        for (int e = tid; e < 32 * 16; e += VJF_POST_THREADS) {
            const int r = e >> 4, c = e & 15;
            s_t[r * LX + c] = s_x[(bi * 32 + r) * LX + c] - ((s_t[r * LX + c] + s_t[(32 + r) * LX + c]) + (s_t[(64 + r) * LX + c] + s_t[(96 + r) * LX + c]));
        }
        __syncthreads();
        if (wave < 2) {
            const float* Db = s_D + (size_t)bi * 32 * LB;
            vjf_f32x4 w = {0.f, 0.f, 0.f, 0.f};
            post_mma32(w, s_t, lane, [&](int i, int m) { return Db[m * LB + 16 * wave + i]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) s_x[(bi * 32 + 16 * wave + 4 * (lane >> 4) + r) * LX + (lane & 15)] = w[r];
        }
        __syncthreads();
    }
    float* Wm = A.state + P.off[VJF_SLOT_W_MEAN];
    for (int e = tid; e < n * dz; e += VJF_POST_THREADS) {
        const int r = e / dz, c = e - r * dz;
        Wm[e] = s_x[r * LX + c];
    }
}

struct VjfResidArgs {
    float* state;
    const float* red;
    double* partial;        // VJF_RESID_BLOCKS partial sums
    int B_total;
    unsigned flags;
};

// q = -2 tr(W^T FDX) + tr(W^T G W), G symmetric: rows i and n-1-i of the lower triangle per work item.
__global__ __launch_bounds__(256) void vjf_resid_kernel(VjfPlan P, VjfResidArgs A) {
    __shared__ double s_d[4];
    const int tid = threadIdx.x, n = P.n, dz = P.dz;
    const float* S = A.state;
    const float* W = S + P.off[VJF_SLOT_W_MEAN];
    const float* G = A.red + P.red_G;
    const float* FDX = A.red + P.red_FDX;
    double part = 0.0;
    const int nitems = n * n;                                   // element (i, j), only j <= i contributes
    for (int e = blockIdx.x * 256 + tid; e < nitems; e += VJF_RESID_BLOCKS * 256) {
        const int i = e / n, j = e - i * n;
        if (j > i) continue;
        float d = 0.f;
        for (int c = 0; c < dz; ++c) d = fmaf(W[(size_t)i * dz + c], W[(size_t)j * dz + c], d);
        part += (double)G[e] * (double)d * (i == j ? 1.0 : 2.0);
    }
    for (int e = blockIdx.x * 256 + tid; e < n * dz; e += VJF_RESID_BLOCKS * 256) part -= 2.0 * (double)W[e] * (double)FDX[e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((tid & 63) == 0) s_d[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) A.partial[blockIdx.x] = ((s_d[0] + s_d[1]) + s_d[2]) + s_d[3];
}

__global__ __launch_bounds__(64) void vjf_sigma_kernel(VjfPlan P, VjfResidArgs A, const int* ok) {
    // one wavefront: lane b holds partial b (VJF_RESID_BLOCKS == 64), fixed-order xor tree
    double t = A.partial[threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (threadIdx.x != 0) return;
    float* S = A.state;
    float* SC = S + P.off[VJF_SLOT_SCALARS];
    t += (double)A.red[P.red_SC + RS_SDX2];
    if (t < 0.0) t = 0.0;
    const float Bf = (float)A.B_total;
    const float sig = S[P.off[VJF_SLOT_TR_LOGVAR]];
    const float mse = (float)(t / ((double)Bf * (double)P.dz));
    const float acc = fminf(SC[VJF_SC_N_TR], 500.f), tot = acc + Bf;           // running_var, size_cap=500 (model.py:375)
    S[P.off[VJF_SLOT_TR_LOGVAR]] = logf((acc / tot) * expf(sig) + (Bf / tot) * mse);
    SC[VJF_SC_N_TR] = tot;
    if (ok && ok[0] == 0) vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED);
}
