// vjf_trial_wide.h -- the trial-parallel half of a step for layer / feature widths whose 16-trial working set does not fit
// one compute unit's LDS (BASELINE config E: d_z = 64, d_y = 512, RBF(1000), hidden [512, 512]).  Instead of one fused
// workgroup per 16 trials, every product over the batch is a GEMM over all B trials on v_mfma_f32_32x32x2_f32
// (`vjf_wide_gemm_kernel`, 64x64 tiles, LDS-staged K chunks, fused epilogues), with small element-wise kernels between:
//     inputs / xs (util.py:11-13) -> RBF features (functional.py:11-22) -> recognition layers tanh(A W^T + b) and heads
//     (recognition.py:31-42) -> xt, dx -> decoder (model.py:28-30) -> pt.mean = xs + Phi W, pt.logvar = log rowsum (Phi w_chol)^2
//     (module.py:64-77) -> per-trial losses and seeds (model.py:124-154) -> dxt = dpy C -> dh_L -> da_l.
// It leaves exactly what the fused kernels leave: rows of E / ACT / DEL, the posterior, per-workgroup loss partials.
#pragma once
#include <hip/hip_runtime.h>
#include "vjf_chol_kernel.h"   // vjf_f32x16, vrow
#include "vjf_plan.h"
#include "vjf_trial_kernel.h"  // VjfTrialArgs

enum { WEPI_NONE = 0, WEPI_BIAS = 1, WEPI_TANH_BIAS = 2, WEPI_ADD_SRC = 3, WEPI_DTANH = 4, WEPI_ADDC_DTANH = 5, WEPI_SEED = 6 };

struct VjfWideGemm {
    const float* A; int lda;       // (M, K) row-major
    const float* Bm; int ldb;      // nt: (N, K) row-major [a torch Linear weight]; else (K, N) row-major
    float* C; int ldc;             // (M, N)
    int M, N, K, nt, epi;
    int ta;                        // 1: A is given transposed, (K, M) row-major
    float src_scale;               // WEPI_ADD_SRC: C = acc + src_scale * src
    const float* bias;             // WEPI_BIAS / WEPI_TANH_BIAS: [N]
    const float* src; int lds;     // WEPI_ADD_SRC: added;  WEPI_DTANH / WEPI_ADDC_DTANH: h of (1 - h^2)
    // WEPI_SEED (dxt = dpy C):  dmu += dxt;  dlv += dxt * eps_t * exp(lv_t / 2) / 2   (C -> dmu, C + N -> dlv; ldc = ldD)
    const float* eps_t; const float* lv_t;
    const int* ok;                 // non-null: nothing is done when ok[0] == 0 (the RLS path after a failed factorisation)
    int va, vb;                    // vjf_wide_gemm2_kernel: A / B can be read 16 bytes at a time (set by the launcher)
};

#define VJF_WG_KC 16
__global__ __launch_bounds__(256) void vjf_wide_gemm_kernel(VjfWideGemm g) {
    __shared__ float s_a[64][VJF_WG_KC + 1];
    __shared__ float s_b[VJF_WG_KC][64 + 1];
    if (g.ok && g.ok[0] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wr = wave >> 1, wc = wave & 1;
    vjf_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < g.K; k0 += VJF_WG_KC) {
        {   // A chunk: 64 rows x 16 k, 4 consecutive k per thread
            const int row = tid >> 2, kq = (tid & 3) * 4, m = m0 + row;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + kq + q;
                s_a[row][kq + q] = (m < g.M && k < g.K) ? (g.ta ? g.A[(size_t)k * g.lda + m] : g.A[(size_t)m * g.lda + k]) : 0.f;
            }
        }
        if (g.nt) {   // B chunk from (N, K): 64 n x 16 k
            const int nn = tid >> 2, kq = (tid & 3) * 4, n = n0 + nn;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + kq + q;
                s_b[kq + q][nn] = (n < g.N && k < g.K) ? g.Bm[(size_t)n * g.ldb + k] : 0.f;
            }
        } else {      // from (K, N): 16 k x 64 n, 4 consecutive n per thread
            const int kk = tid >> 4, nq = (tid & 15) * 4, k = k0 + kk;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + nq + q;
                s_b[kk][nq + q] = (n < g.N && k < g.K) ? g.Bm[(size_t)k * g.ldb + n] : 0.f;
            }
        }
        __syncthreads();
        float a[VJF_WG_KC / 2], b[VJF_WG_KC / 2];
#pragma unroll
        for (int t = 0; t < VJF_WG_KC / 2; ++t) {
            a[t] = s_a[wr * 32 + (lane & 31)][2 * t + (lane >> 5)];
            b[t] = s_b[2 * t + (lane >> 5)][wc * 32 + (lane & 31)];
        }
#pragma unroll
        for (int t = 0; t < VJF_WG_KC / 2; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wc * 32 + (lane & 31), h = lane >> 5;
    if (n >= g.N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wr * 32 + vrow(r, h);
        if (m >= g.M) continue;
        float v = acc[r];
        float* c = g.C + (size_t)m * g.ldc + n;
        switch (g.epi) {
            case WEPI_BIAS: v += g.bias[n]; break;
            case WEPI_TANH_BIAS: v = tanhf(v + g.bias[n]); break;
            case WEPI_ADD_SRC: v = fmaf(g.src_scale, g.src[(size_t)m * g.lds + n], v); break;
            case WEPI_DTANH: { const float hv = g.src[(size_t)m * g.lds + n]; v *= (1.f - hv * hv); break; }
            case WEPI_ADDC_DTANH: { const float hv = g.src[(size_t)m * g.lds + n]; v = (*c + v) * (1.f - hv * hv); break; }
            case WEPI_SEED: {
                const float e = g.eps_t[(size_t)m * g.N + n], lv = g.lv_t[(size_t)m * g.N + n];
                c[g.N] = fmaf(v * e, 0.5f * expf(0.5f * lv), c[g.N]);            // dlv
                v += *c;                                                         // dmu
                break;
            }
            default: break;
        }
        *c = v;
    }
}

// The same product for large shapes: 128 x TN tiles (TN = 128 or 64), 2 x 2 wavefronts of 64 x TN/2 each (four or two
// 32x32 accumulators), K chunks of 16 through LDS; the NEXT chunk's 16-byte global loads are issued before the current
// chunk's MFMAs (register double buffering), so a wavefront has 32 (16) MFMAs per 32 (24) LDS reads and several workgroups
// share a CU.  An operand whose rows start on 16-byte boundaries (leading dimension a multiple of 4, aligned base: g.va / g.vb, set
// by the launcher) is read 16 bytes at a time, another one float by float.
template <int TN>
__global__ __launch_bounds__(256) void vjf_wide_gemm2_kernel(VjfWideGemm g) {
    constexpr int TM = 128, KC = 16, NB = TN / 64;       // NB: 32-column blocks per wavefront
    __shared__ float s_a[TM][KC + 1];
    __shared__ float s_b[KC][TN + 1];
    if (g.ok && g.ok[0] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int wr = wave >> 1, wc = wave & 1;
    vjf_f32x16 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // loader roles: "k-contiguous" operands (A row-major, B as (N, K)): thread = (row, 8 consecutive k); "row-contiguous" ones
    // (A transposed, B as (K, N)): thread = (k, 8 consecutive rows)
    float4 ra[2], rb[TN / 64];
    // four consecutive floats at p (the first `valid` of them inside the matrix): one 16-byte load when the operand allows it
    auto ld4 = [](const float* p, int valid, bool vec) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
        if (valid > 0) v.x = p[0];
        if (valid > 1) v.y = p[1];
        if (valid > 2) v.z = p[2];
        if (valid > 3) v.w = p[3];
        return v;
    };
    auto load_a = [&](int k0) {
        if (!g.ta) {
            const int row = tid >> 1, kq = (tid & 1) * 8, m = m0 + row;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int k = k0 + kq + 4 * q;
                ra[q] = ld4(g.A + (size_t)(m < g.M ? m : 0) * g.lda + (k < g.K ? k : 0), m < g.M ? g.K - k : 0, g.va != 0);
            }
        } else {
            const int kk = tid >> 4, mq = (tid & 15) * 8, k = k0 + kk;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = m0 + mq + 4 * q;
                ra[q] = ld4(g.A + (size_t)(k < g.K ? k : 0) * g.lda + (m < g.M ? m : 0), k < g.K ? g.M - m : 0, g.va != 0);
            }
        }
    };
    auto store_a = [&]() {
        if (!g.ta) {
            const int row = tid >> 1, kq = (tid & 1) * 8;
#pragma unroll
            for (int q = 0; q < 2; ++q) { float* d = &s_a[row][kq + 4 * q]; d[0] = ra[q].x; d[1] = ra[q].y; d[2] = ra[q].z; d[3] = ra[q].w; }
        } else {
            const int kk = tid >> 4, mq = (tid & 15) * 8;
#pragma unroll
            for (int q = 0; q < 2; ++q) { s_a[mq + 4 * q][kk] = ra[q].x; s_a[mq + 4 * q + 1][kk] = ra[q].y; s_a[mq + 4 * q + 2][kk] = ra[q].z; s_a[mq + 4 * q + 3][kk] = ra[q].w; }
        }
    };
    auto load_b = [&](int k0) {
        if (g.nt) {          // (N, K): TN rows x 16 k; TN = 128: (row, 8 k) as A; TN = 64: (row, 4 k)
            constexpr int KPT = TN / 16;                                  // k per thread
            const int nn = tid / (16 / KPT), kq = (tid % (16 / KPT)) * KPT, n = n0 + nn;
#pragma unroll
            for (int q = 0; q < TN / 64; ++q) {
                const int k = k0 + kq + 4 * q;
                rb[q] = ld4(g.Bm + (size_t)(n < g.N ? n : 0) * g.ldb + (k < g.K ? k : 0), n < g.N ? g.K - k : 0, g.vb != 0);
            }
        } else {             // (K, N): 16 k x TN columns: thread = (k, TN / 16 consecutive columns)
            constexpr int NPT = TN / 16;
            const int kk = tid >> 4, nq = (tid & 15) * NPT, k = k0 + kk;
#pragma unroll
            for (int q = 0; q < TN / 64; ++q) {
                const int n = n0 + nq + 4 * q;
                rb[q] = ld4(g.Bm + (size_t)(k < g.K ? k : 0) * g.ldb + (n < g.N ? n : 0), k < g.K ? g.N - n : 0, g.vb != 0);
            }
        }
    };
    auto store_b = [&]() {
        if (g.nt) {
            constexpr int KPT = TN / 16;
            const int nn = tid / (16 / KPT), kq = (tid % (16 / KPT)) * KPT;
#pragma unroll
            for (int q = 0; q < TN / 64; ++q) { s_b[kq + 4 * q][nn] = rb[q].x; s_b[kq + 4 * q + 1][nn] = rb[q].y; s_b[kq + 4 * q + 2][nn] = rb[q].z; s_b[kq + 4 * q + 3][nn] = rb[q].w; }
        } else {
            constexpr int NPT = TN / 16;
            const int kk = tid >> 4, nq = (tid & 15) * NPT;
#pragma unroll
            for (int q = 0; q < TN / 64; ++q) { float* d = &s_b[kk][nq + 4 * q]; d[0] = rb[q].x; d[1] = rb[q].y; d[2] = rb[q].z; d[3] = rb[q].w; }
        }
    };
    load_a(0); load_b(0);
    for (int k0 = 0; k0 < g.K; k0 += KC) {
        __syncthreads();                                   // (the previous chunk's readers are done)
        store_a(); store_b();
        __syncthreads();
        if (k0 + KC < g.K) { load_a(k0 + KC); load_b(k0 + KC); }
        const int rl = lane & 31, h = lane >> 5;
#pragma unroll
        for (int t = 0; t < KC / 2; ++t) {
            float a[2], b[NB];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = s_a[wr * 64 + i * 32 + rl][2 * t + h];
#pragma unroll
            for (int j = 0; j < NB; ++j) b[j] = s_b[2 * t + h][wc * (TN / 2) + j * 32 + rl];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    const int h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + wc * (TN / 2) + j * 32 + (lane & 31);
            if (n >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + i * 32 + vrow(r, h);
                if (m >= g.M) continue;
                float v = acc[i][j][r];
                float* c = g.C + (size_t)m * g.ldc + n;
                switch (g.epi) {
                    case WEPI_BIAS: v += g.bias[n]; break;
                    case WEPI_TANH_BIAS: v = tanhf(v + g.bias[n]); break;
                    case WEPI_ADD_SRC: v = fmaf(g.src_scale, g.src[(size_t)m * g.lds + n], v); break;
                    case WEPI_DTANH: { const float hv = g.src[(size_t)m * g.lds + n]; v *= (1.f - hv * hv); break; }
                    case WEPI_ADDC_DTANH: { const float hv = g.src[(size_t)m * g.lds + n]; v = (*c + v) * (1.f - hv * hv); break; }
                    case WEPI_SEED: {
                        const float e = g.eps_t[(size_t)m * g.N + n], lv = g.lv_t[(size_t)m * g.N + n];
                        c[g.N] = fmaf(v * e, 0.5f * expf(0.5f * lv), c[g.N]);            // dlv
                        v += *c;                                                         // dmu
                        break;
                    }
                    default: break;
                }
                *c = v;
            }
        }
}

struct VjfWideArgs {
    VjfTrialArgs t;
    float* XU;      // (B, dxu)  [xs | u]
    float* PM;      // (B, dz)   pt.mean
    float* PLV;     // (B)       pt.logvar
    float* PY;      // (B, dy)   decoder output
    float* Z;       // (B, n)    Phi w_chol
};

// ACT = [in | 1 | . | 1 | .. | xt | 1 | 0], in = [y | u | mu_s | lv_s]; XU = [xs | u]
__global__ __launch_bounds__(256) void vjf_wide_in_kernel(VjfPlan P, VjfWideArgs W) {
    const VjfTrialArgs& A = W.t;
    const int dz = P.dz, dy = P.dy, du = P.du, din = P.din, dxu = P.dxu;
    const float* S = A.state;
    const bool prior = A.mu_s == nullptr;
    const size_t tot = (size_t)A.B * P.ldA;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (size_t)gridDim.x * 256) {
        const size_t b = e / P.ldA;
        const int c = (int)(e - b * P.ldA);
        float v = 0.f;
        if (c < dy) v = A.y[b * dy + c];
        else if (c < dy + du) v = A.u[b * du + (c - dy)];
        else if (c < dy + du + dz) { const int j = c - dy - du; v = prior ? S[P.off[VJF_SLOT_PRIOR_MEAN] + j] : A.mu_s[b * dz + j]; }
        else if (c < din) { const int j = c - dy - du - dz; v = prior ? S[P.off[VJF_SLOT_PRIOR_LOGVAR] + j] : A.lv_s[b * dz + j]; }
        else {
            // the ones that follow every segment (the bias column of the gradient Gram); everything else is written later
            bool one = c == din || c == P.colA_xt + dz;
            for (int l = 0; l < P.L; ++l) one = one || c == P.colA_act[l + 1] + P.h[l];
            if (!one && c < P.colA_xt + dz) continue;            // hidden activations / xt: by the GEMMs / vjf_wide_mid_kernel
            v = one ? 1.f : 0.f;
        }
        A.ACT[e] = v;
    }
    const size_t totx = (size_t)A.B * dxu;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < totx; e += (size_t)gridDim.x * 256) {
        const size_t b = e / dxu;
        const int c = (int)(e - b * dxu);
        float v;
        if (c < dz) {
            const float mu = prior ? S[P.off[VJF_SLOT_PRIOR_MEAN] + c] : A.mu_s[b * dz + c];
            const float lv = prior ? S[P.off[VJF_SLOT_PRIOR_LOGVAR] + c] : A.lv_s[b * dz + c];
            v = fmaf(A.eps_s[b * dz + c], expf(0.5f * lv), mu);
        } else v = A.u[b * du + (c - dz)];
        W.XU[e] = v;
    }
}

// RBF features: workgroup = 256 centroids x 16 trials.  Each thread keeps ONE centroid row in registers (read once, 16-byte
// loads of its own contiguous row) and runs it against the 16 trials' [xs | u] rows staged in LDS (same address on every
// lane: broadcast).  grid = (ceil(n / 256), ceil(B / 16)).
#define VJF_WIDE_RBF_MAXD 96
__global__ __launch_bounds__(256) void vjf_wide_rbf_kernel(VjfPlan P, VjfWideArgs W) {
    extern __shared__ __attribute__((aligned(16))) float s_x[];   // 16 x dxu
    const VjfTrialArgs& A = W.t;
    const int n = P.n, dxu = P.dxu, tid = threadIdx.x;
    const int k = blockIdx.x * 256 + tid, b0 = blockIdx.y * 16, nb = min(16, A.B - b0);
    for (int e = tid; e < 16 * dxu; e += 256) s_x[e] = (e / dxu) < nb ? W.XU[(size_t)b0 * dxu + e] : 0.f;
    __syncthreads();
    if (k >= n) return;
    const float* cen = A.state + P.off[VJF_SLOT_CENTROID] + (size_t)k * dxu;
    const float w = expf(A.state[P.off[VJF_SLOT_LOGWIDTH] + k]);
    const float sc = -0.5f / (w * w);
    float d2[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) d2[b] = 0.f;
    for (int c0 = 0; c0 < dxu; c0 += 8) {                       // 8 centroid coordinates at a time
        float cv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cv[q] = c0 + q < dxu ? cen[c0 + q] : 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (c0 + q < dxu) {
#pragma unroll
                for (int b = 0; b < 16; ++b) { const float d = s_x[b * dxu + c0 + q] - cv[q]; d2[b] = fmaf(d, d, d2[b]); }
            }
        }
    }
#pragma unroll
    for (int b = 0; b < 16; ++b)
        if (b < nb) A.E[(size_t)(b0 + b) * P.ldE + k] = expf(d2[b] * sc);
}

// after the heads: xt (model.py:119), its place in ACT, dx and the zero padding of the E rows
__global__ __launch_bounds__(256) void vjf_wide_mid_kernel(VjfPlan P, VjfWideArgs W) {
    const VjfTrialArgs& A = W.t;
    const int dz = P.dz, n = P.n, wE = P.ldE - n;
    const size_t tot = (size_t)A.B * wE;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < tot; e += (size_t)gridDim.x * 256) {
        const size_t b = e / wE;
        const int j = (int)(e - b * wE);
        float v = 0.f;
        if (j < dz) {
            const float xt = fmaf(A.eps_t[b * dz + j], expf(0.5f * A.lv_t[b * dz + j]), A.mu_t[b * dz + j]);
            A.ACT[b * P.ldA + P.colA_xt + j] = xt;
            v = xt - W.XU[b * P.dxu + j];
        }
        A.E[b * P.ldE + n + j] = v;
    }
}

// pt.logvar = log sum_j Z[b][j]^2 : one wavefront per trial, fixed-order lane sums
__global__ __launch_bounds__(256) void vjf_wide_rownorm_kernel(VjfPlan P, VjfWideArgs W) {
    const int n = P.n, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= W.t.B) return;
    float v = 0.f;
    for (int j = lane; j < n; j += 64) { const float z = W.Z[(size_t)b * n + j]; v = fmaf(z, z, v); }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if (lane == 0) W.PLV[b] = logf(v);
}

// per-trial loss terms and backward seeds (no 1/B): one wavefront per trial, 4 trials per workgroup; partial sums per workgroup
__global__ __launch_bounds__(256) void vjf_wide_loss_kernel(VjfPlan P, VjfWideArgs W) {
    __shared__ float s_sc[4][RS_N];
    const VjfTrialArgs& A = W.t;
    const int dz = P.dz, dy = P.dy, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    const bool ok = b < A.B;
    const bool warm = (A.flags & VJF_FLAG_WARM_UP) != 0;
    const float* S = A.state;
    const float rho = S[P.off[VJF_SLOT_LIK_LOGVAR]], sig = S[P.off[VJF_SLOT_TR_LOGVAR]];
    float lrec = 0.f, ssey = 0.f, ldyn = 0.f, ent = 0.f, sdx2 = 0.f;
    if (ok) {
        const float* yrow = A.ACT + (size_t)b * P.ldA;            // in = [y | ..]
        const float* py = W.PY + (size_t)b * dy;
        float* drow = A.DEL + (size_t)b * P.ldD;
        if (P.lik == VJF_LIK_GAUSSIAN) {                       // likelihood.py:19-26, functional.py:54-73
            const float p = expf(-0.5f * rho), e = expf(-rho);
            for (int i = lane; i < dy; i += 64) {
                const float yv = yrow[i], pv = py[i];
                const float r = pv - yv, dsc = yv * p - pv * p;
                lrec += 0.5f * (dsc * dsc + rho);
                ssey = fmaf(r, r, ssey);
                drow[P.colD_dpy + i] = e * r;
            }
        } else {                                               // likelihood.py:51-62
            for (int i = lane; i < dy; i += 64) {
                const float yv = yrow[i], pv = py[i];
                const float eta = fminf(pv, 10.f), ex = expf(eta);
                lrec += ex - yv * eta;
                const float r = pv - yv;
                ssey = fmaf(r, r, ssey);
                drow[P.colD_dpy + i] = (pv <= 10.f) ? (ex - yv) : 0.f;
            }
        }
        const float p = expf(-0.5f * sig), e = expf(-sig), plv = W.PLV[b];
        for (int j = lane; j < dz; j += 64) {                  // model.py:390-391, functional.py:62-75
            const float mp = W.PM[(size_t)b * dz + j], mu = A.mu_t[(size_t)b * dz + j], lv = A.lv_t[(size_t)b * dz + j];
            const float dsc = mp * p - mu * p;
            const float tr = expf(plv + lv - sig);
            ldyn += 0.5f * (dsc * dsc + sig) + 0.5f * tr;
            ent += 0.5f * lv;                                  // functional.py:25-29
            const float dx = A.E[(size_t)b * P.ldE + P.n + j];
            sdx2 = fmaf(dx, dx, sdx2);
            float dmu = 0.f, dlv = -0.5f;
            if (!warm) { dmu = -e * (mp - mu); dlv += 0.5f * tr; }
            drow[P.colD_dmu + j] = dmu;                        // the decoder path is added by the dxt GEMM's epilogue
            drow[P.colD_dlv + j] = dlv;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lrec += __shfl_xor(lrec, o, 64); ssey += __shfl_xor(ssey, o, 64); ldyn += __shfl_xor(ldyn, o, 64);
        ent += __shfl_xor(ent, o, 64); sdx2 += __shfl_xor(sdx2, o, 64);
    }
    if (lane == 0) {
        s_sc[w][RS_LRECON] = ok ? lrec : 0.f; s_sc[w][RS_LDYN] = ok ? ldyn : 0.f; s_sc[w][RS_ENT] = ok ? ent : 0.f;
        s_sc[w][RS_SSEY] = ok ? ssey : 0.f; s_sc[w][RS_SDX2] = ok ? sdx2 : 0.f;
    }
    __syncthreads();
    if (threadIdx.x < RS_N) {
        float v = 0.f;
        if (threadIdx.x <= RS_SDX2) v = ((s_sc[0][threadIdx.x] + s_sc[1][threadIdx.x]) + s_sc[2][threadIdx.x]) + s_sc[3][threadIdx.x];
        A.partial[(size_t)blockIdx.x * RS_N + threadIdx.x] = v;
    }
}
