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

enum { WEPI_NONE = 0, WEPI_BIAS = 1, WEPI_TANH_BIAS = 2, WEPI_ADD_SRC = 3, WEPI_DTANH = 4, WEPI_ADDC_DTANH = 5, WEPI_SEED = 6, WEPI_HEADS = 7 };

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
    float* C2;                     // WEPI_HEADS (both recognition heads in one product, N = 2 dz): columns < dz -> C (mean, no bias), the
                                   // others + bias[n - dz] -> C2 (log-variance); both (M, dz) with ldc
    const int* ok;                 // non-null: nothing is done when ok[0] == 0 (the RLS path after a failed factorisation)
    int va, vb;                    // vjf_wide_gemm2_kernel: A / B can be read 16 bytes at a time (set by the launcher)
};

// the fused epilogues of the wide GEMM kernels: C[m][n] <- f(v)
__device__ __forceinline__ void vjf_wide_epilogue(const VjfWideGemm& g, int m, int n, float v) {
    if (g.epi == WEPI_HEADS) {
        const int dz = g.N >> 1;
        if (n < dz) g.C[(size_t)m * g.ldc + n] = v;
        else g.C2[(size_t)m * g.ldc + n - dz] = v + g.bias[n - dz];
        return;
    }
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
        vjf_wide_epilogue(g, m, n, acc[r]);
    }
}

// The same product for large shapes: 128 x TN tiles (TN = 128 or 64), NWR x 2 wavefronts of (128 / NWR) x TN/2 each, K chunks
// of KC through two LDS images (one barrier per chunk); global loads run two chunks ahead of their LDS store, in registers.  Both operands
// sit in LDS k-major, [k][row]: every MFMA operand read is 32 consecutive floats per lane half (conflict-free), issued eight k
// steps at a time (the compiler's own order waits for every step's reads before its MFMAs).
// Global loads are 16 bytes per lane (a k-contiguous operand: 16 consecutive lanes take 16 rows at one k offset, the next 16
// the same rows 16 bytes on, so that the transposing LDS stores -- row stride = 4 mod 32 banks -- are conflict-free) -- the
// launcher picks this kernel only for operands that allow that (16-byte aligned rows, extent along the contiguous direction a
// multiple of 4: g.va and g.vb).
// (TA / BT: the layouts g.ta / g.nt as template parameters -- with run-time branches inside the loop the compiler can no longer
//  count which loads a store has to wait for, and waits for all of them.)
template <int TN, int KC, int NWR, bool TA, bool BT>
__global__ __launch_bounds__(NWR * 128) __attribute__((amdgpu_waves_per_eu(4))) void vjf_wide_gemm3_kernel(VjfWideGemm g) {
    constexpr int TM = 128, NB = TN / 64;                                   // NB: 32-column blocks per wavefront
    constexpr int NT = NWR * 128, RB = TM / NWR / 32;                       // NWR x 2 wavefronts of (RB x 32) x TN/2 each
    constexpr int FA = KC * TM / (4 * NT), FB = KC * TN / (4 * NT);         // 16-byte pieces per thread and chunk
    constexpr int L = KC / 4;                                               // lanes per row of a k-contiguous operand
    __shared__ __attribute__((aligned(16))) float s_a[2][KC * (TM + 4)];
    __shared__ __attribute__((aligned(16))) float s_b[2][KC * (TN + 4)];
    if (g.ok && g.ok[0] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // workgroup -> tile: consecutive workgroups go to the eight XCDs in turn, so workgroup id = 8 t + x gets row tile x + 8 (t / gn),
    // column tile t % gn -- an XCD's L2 then holds a few row tiles of A and the matrix B instead of streaming all of A
    int bm = blockIdx.y, bn = blockIdx.x;
    {
        const int gn = gridDim.x, gm = gridDim.y;
        if ((gm & 7) == 0) {
            const int id = blockIdx.y * gn + blockIdx.x, x = id & 7, t = id >> 3;
            bm = x + 8 * (t / gn); bn = t % gn;
        }
    }
    const int m0 = bm * TM, n0 = bn * TN;
    const int wr = wave >> 1, wc = wave & 1;
    constexpr int sa = TM + 4, sb = TN + 4;                                 // LDS row strides
    vjf_f32x16 acc[RB][NB];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[2][FA], rb[2][FB];                            // two chunks ahead of the MFMAs
    // k-contiguous operand, R rows of the tile starting at r0 (A row-major, B as (N, K)): piece q = (row q RP + kc_row, k 4 kc_kq)
    const int kc_kq = (tid >> 4) % L, kc_row = (tid & 15) + 16 * (tid / (16 * L));
    // (pieces outside the matrix are read from a clamped address and zeroed when they are STORED to LDS: a select on a load's
    //  destination right behind the load would wait for it and undo the prefetch)
    auto load_kc = [&](float4* rv, int F, const float* base, int ld, int r0, int rows, int k0) {
        constexpr int RP = NT / L;
        const int k = k0 + 4 * kc_kq;
#pragma unroll
        for (int q = 0; q < F; ++q) {
            const int gr = min(r0 + q * RP + kc_row, rows - 1);
            rv[q] = *reinterpret_cast<const float4*>(base + (size_t)gr * ld + (k < g.K ? k : 0));
        }
    };
    auto store_kc = [&](float* sx, int S, const float4* rv, int F, int k0) {
        constexpr int RP = NT / L;
        const bool in = k0 + 4 * kc_kq < g.K;
#pragma unroll
        for (int q = 0; q < F; ++q) {
            float* d = sx + (size_t)(4 * kc_kq) * S + q * RP + kc_row;
            d[0] = in ? rv[q].x : 0.f; d[S] = in ? rv[q].y : 0.f; d[2 * S] = in ? rv[q].z : 0.f; d[3 * S] = in ? rv[q].w : 0.f;
        }
    };
    // row-contiguous operand (A transposed, B as (K, N)), R columns of the tile starting at c0: piece f = tid + NT q of the KC x R/4 grid
    auto load_rc = [&](float4* rv, int F, const float* base, int ld, int c0, int cols, int R, int k0) {
#pragma unroll
        for (int q = 0; q < F; ++q) {
            const int f = tid + NT * q, k = k0 + f / (R / 4), c = c0 + (f % (R / 4)) * 4;
            const bool in = k < g.K && c < cols;
            rv[q] = *reinterpret_cast<const float4*>(base + (in ? (size_t)k * ld + c : 0));
        }
    };
    auto store_rc = [&](float* sx, int S, const float4* rv, int F, int c0, int cols, int R, int k0) {
#pragma unroll
        for (int q = 0; q < F; ++q) {
            const int f = tid + NT * q, kk = f / (R / 4), c = (f % (R / 4)) * 4;
            const bool in = k0 + kk < g.K && c0 + c < cols;
            float4 v = rv[q];
            if (!in) v = make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(sx + (size_t)kk * S + c) = v;
        }
    };
    auto load = [&](int buf, int k0) {
        if constexpr (!TA) load_kc(ra[buf], FA, g.A, g.lda, m0, g.M, k0); else load_rc(ra[buf], FA, g.A, g.lda, m0, g.M, TM, k0);
        if constexpr (BT) load_kc(rb[buf], FB, g.Bm, g.ldb, n0, g.N, k0); else load_rc(rb[buf], FB, g.Bm, g.ldb, n0, g.N, TN, k0);
    };
    auto store = [&](int buf, int k0) {                    // register set `buf` (chunk at k0) -> LDS image `buf`
        if constexpr (!TA) store_kc(s_a[buf], sa, ra[buf], FA, k0); else store_rc(s_a[buf], sa, ra[buf], FA, m0, g.M, TM, k0);
        if constexpr (BT) store_kc(s_b[buf], sb, rb[buf], FB, k0); else store_rc(s_b[buf], sb, rb[buf], FB, n0, g.N, TN, k0);
    };
    const int rl = lane & 31, h = lane >> 5;
    const int oa = h * sa + wr * RB * 32 + rl, ob = h * sb + wc * (TN / 2) + rl;
    constexpr int NBAT = KC / 16;                          // batches of eight k steps
    // Chunk c (at k0) is computed from LDS image c & 1 while chunk c + 1 goes from registers into the other image and the loads of
    // chunk c + 3 start (register set (c + 1) & 1; the other set holds chunk c + 2): ONE barrier per chunk, global loads two
    // chunks ahead of their LDS store.
    auto chunk = [&](int cur, int k0) {
        const int nxt = cur ^ 1;
        if (k0 + KC < g.K) store(nxt, k0 + KC);
        if (k0 + 3 * KC < g.K) load(nxt, k0 + 3 * KC);
        const float* pa = s_a[cur] + oa;
        const float* pb = s_b[cur] + ob;
#pragma unroll
        for (int bt = 0; bt < NBAT; ++bt) {                // eight k steps: their operand reads, then their MFMAs (the other
            float av[8][RB], bv[8][NB];                    // wavefronts of the SIMD fill the matrix pipe meanwhile)
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                const int t = bt * 8 + st;
#pragma unroll
                for (int i = 0; i < RB; ++i) av[st][i] = pa[2 * t * sa + i * 32];
#pragma unroll
                for (int j = 0; j < NB; ++j) bv[st][j] = pb[2 * t * sb + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 8; ++st)
#pragma unroll
                for (int i = 0; i < RB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st][i], bv[st][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    };
    load(0, 0);
    store(0, 0);
    if (KC < g.K) load(1, KC);
    if (2 * KC < g.K) load(0, 2 * KC);
    __syncthreads();
    for (int k0 = 0; k0 < g.K; k0 += 2 * KC) {
        chunk(0, k0);
        if (k0 + KC < g.K) chunk(1, k0 + KC);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int n = n0 + wc * (TN / 2) + j * 32 + (lane & 31);
            if (n >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wr * RB + i) * 32 + vrow(r, h);
                if (m >= g.M) continue;
                vjf_wide_epilogue(g, m, n, acc[i][j][r]);
            }
        }
}

// C = A B for the products with a narrow output (N <= 128: the two heads in one product, pt.mean, dxt, the RLS solves) -- a 128 x 64 tile grid would
// leave most of the chip idle behind a serial K loop.  Workgroup = one 32 x 32 tile of C; its NW wavefronts split K (contiguous
// ranges), every wavefront reads its operands from global memory straight into the MFMA operand layout -- the order of the
// k indices inside a group of 8 is free as long as A and B agree: lane (row, half) takes k = k0 + 4 half + {0..3}, four
// consecutive floats of a k-contiguous operand in one 16-byte load -- no LDS staging and no barrier in the K loop; the NW partial
// tiles are summed through LDS in a fixed order.  Rows / columns beyond M / N are computed on clamped addresses and dropped.
template <int NW>
__global__ __launch_bounds__(NW * 64) void vjf_skinny_gemm_kernel(VjfWideGemm g) {
    __shared__ float s_p[NW][1024];
    if (g.ok && g.ok[0] == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int m = min(m0 + r, g.M - 1), n = min(n0 + r, g.N - 1);
    const int kw = ((g.K + NW - 1) / NW + 7) / 8 * 8;           // k range of a wavefront: a multiple of 8
    const int kb = wave * kw, ke = min(g.K, kb + kw);
    vjf_f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    // four k values of the group at k0 for this lane; `full`: all of them inside [0, K)
    auto ld_a = [&](float (&a)[4], int k0, bool full) {
        const int kk = k0 + 4 * h;
        if (!g.ta) {
            const float* p = g.A + (size_t)m * g.lda + kk;
            if (full && g.va) { const float4 v = *reinterpret_cast<const float4*>(p); a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) a[e] = (full || kk + e < ke) ? p[e] : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = (full || kk + e < ke) ? g.A[(size_t)(kk + e) * g.lda + m] : 0.f;
        }
    };
    auto ld_b = [&](float (&b)[4], int k0, bool full) {
        const int kk = k0 + 4 * h;
        if (g.nt) {
            const float* p = g.Bm + (size_t)n * g.ldb + kk;
            if (full && g.vb) { const float4 v = *reinterpret_cast<const float4*>(p); b[0] = v.x; b[1] = v.y; b[2] = v.z; b[3] = v.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = (full || kk + e < ke) ? p[e] : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) b[e] = (full || kk + e < ke) ? g.Bm[(size_t)(kk + e) * g.ldb + n] : 0.f;
        }
    };
    // rounds of 32 k (four groups of 8): the next round's loads are in flight while this round's MFMAs issue
    float a0[4][4], p0[4][4], a1[4][4], p1[4][4];
    auto load = [&](float (&a)[4][4], float (&p)[4][4], int kr) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { ld_a(a[u], kr + 8 * u, true); ld_b(p[u], kr + 8 * u, true); }
    };
    auto mma = [&](const float (&a)[4][4], const float (&p)[4][4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][e], p[u][e], acc, 0, 0, 0);
    };
    int k0 = kb;
    const int kfull = kb + (ke > kb ? (ke - kb) / 32 * 32 : 0);
    if (k0 < kfull) load(a0, p0, k0);
    while (k0 < kfull) {
        if (k0 + 32 < kfull) load(a1, p1, k0 + 32);
        mma(a0, p0);
        k0 += 32;
        if (k0 >= kfull) break;
        if (k0 + 32 < kfull) load(a0, p0, k0 + 32);
        mma(a1, p1);
        k0 += 32;
    }
    for (; k0 < ke; k0 += 8) {                                  // the tail of the range, element-wise guards
        float a[4], b[4];
        ld_a(a, k0, false);
        ld_b(b, k0, false);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) s_p[wave][q * 64 + lane] = acc[q];
    __syncthreads();
    for (int e = tid; e < 1024; e += NW * 64) {
        float v = s_p[0][e];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += s_p[w][e];
        const int mm = m0 + vrow(e >> 6, (e >> 5) & 1), nn = n0 + (e & 31);
        if (mm < g.M && nn < g.N) vjf_wide_epilogue(g, mm, nn, v);
    }
}

struct VjfWideArgs {
    VjfTrialArgs t;
    float* XU;      // (B, dxu)  [xs | u]
    float* PM;      // (B, dz)   pt.mean
    float* PY;      // (B, dy)   decoder output
    float* Z;       // (B, n)    Phi w_chol
};

// ACT = [in | 1 | . | 1 | .. | xt | 1 | 0], in = [y | u | mu_s | lv_s]; XU = [xs | u]
__global__ __launch_bounds__(256) void vjf_wide_in_kernel(VjfPlan P, VjfWideArgs W) {
    const VjfTrialArgs& A = W.t;
    const int dz = P.dz, dy = P.dy, du = P.du, din = P.din, dxu = P.dxu;
    const float* S = A.state;
    const bool prior = A.mu_s == nullptr;
    for (size_t b = blockIdx.x; b < (size_t)A.B; b += gridDim.x) {     // one trial row per workgroup and round
        float* arow = A.ACT + b * P.ldA;
        // (only the columns this kernel owns are visited: the inputs, the ones behind every segment -- the bias column of the gradient
        //  Gram -- and the zeros behind the last one; the hidden activations / xt in between belong to the GEMMs and
        //  vjf_wide_mid_kernel.  Scanning the whole row with a test per column made this 23 us of config E's step.)
        for (int c = threadIdx.x; c < din; c += 256) {
            float v;
            if (c < dy) v = A.y[b * dy + c];
            else if (c < dy + du) v = A.u[b * du + (c - dy)];
            else if (c < dy + du + dz) { const int j = c - dy - du; v = prior ? S[P.off[VJF_SLOT_PRIOR_MEAN] + j] : A.mu_s[b * dz + j]; }
            else { const int j = c - dy - du - dz; v = prior ? S[P.off[VJF_SLOT_PRIOR_LOGVAR] + j] : A.lv_s[b * dz + j]; }
            arow[c] = v;
        }
        if ((int)threadIdx.x <= P.L + 1) {
            const int i = threadIdx.x;
            int col = din;
            if (i == P.L + 1) col = P.colA_xt + dz;
            else for (int l = 0; l < P.L; ++l) if (i == l + 1) col = P.colA_act[l + 1] + P.h[l];
            arow[col] = 1.f;
        }
        for (int c = P.colA_xt + dz + 1 + threadIdx.x; c < P.ldA; c += 256) arow[c] = 0.f;
        for (int c = threadIdx.x; c < dxu; c += 256) {
            float v;
            if (c < dz) {
                const float mu = prior ? S[P.off[VJF_SLOT_PRIOR_MEAN] + c] : A.mu_s[b * dz + c];
                const float lv = prior ? S[P.off[VJF_SLOT_PRIOR_LOGVAR] + c] : A.lv_s[b * dz + c];
                v = fmaf(A.eps_s[b * dz + c], expf(0.5f * lv), mu);
            } else v = A.u[b * du + (c - dz)];
            W.XU[b * dxu + c] = v;
        }
    }
}

// RBF features: workgroup = 256 centroids x 16 trials.  Each thread keeps ONE centroid row in registers (read once, 16-byte
// loads of its own contiguous row) and runs it against the 16 trials' [xs | u] rows.  Those are the same for every lane: they
// are read through the constant address space at wave-uniform addresses -- scalar loads into SGPRs, which the subtract takes as
// its scalar operand -- so the kernel issues no LDS instruction at all.  (Round 2 staged the rows in LDS and read them back as
// broadcasts: a broadcast still returns 64 copies, 1 KB per 16-byte read, and the LDS pipe, not the vector unit, bounded the
// kernel: 27 us of config E's step.)  The rows were written by the kernel before this one and are not written here.
// grid = (ceil(n / 256), ceil(B / 16)).
#define VJF_WIDE_RBF_MAXD 96
typedef const float __attribute__((address_space(4))) vjf_cfloat;
__global__ __launch_bounds__(256) void vjf_wide_rbf_kernel(VjfPlan P, VjfWideArgs W) {
    const VjfTrialArgs& A = W.t;
    const int n = P.n, dxu = P.dxu, tid = threadIdx.x;
    const int k = blockIdx.x * 256 + tid, b0 = blockIdx.y * 16, nb = min(16, A.B - b0);
    if (k >= n) return;
    vjf_cfloat* xu = (vjf_cfloat*)(W.XU + (size_t)b0 * dxu);
    const float* cen = A.state + P.off[VJF_SLOT_CENTROID] + (size_t)k * dxu;
    const float w = expf(A.state[P.off[VJF_SLOT_LOGWIDTH] + k]);
    const float sc = -0.5f / (w * w);
    float d2[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) d2[b] = 0.f;
    for (int c0 = 0; c0 < dxu; c0 += 8) {                       // 8 centroid coordinates at a time
        float cv[8];
        if ((dxu & 3) == 0) {
            const float4 ca = *reinterpret_cast<const float4*>(cen + c0);
            const float4 cb = c0 + 4 < dxu ? *reinterpret_cast<const float4*>(cen + c0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            cv[0] = ca.x; cv[1] = ca.y; cv[2] = ca.z; cv[3] = ca.w; cv[4] = cb.x; cv[5] = cb.y; cv[6] = cb.z; cv[7] = cb.w;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) cv[q] = c0 + q < dxu ? cen[c0 + q] : 0.f;
        }
        const int nq = min(8, dxu - c0);
        if (nq == 8) {                                          // (eight consecutive coordinates of a row: one wide scalar load)
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                vjf_cfloat* xr = xu + (size_t)min(b, nb - 1) * dxu + c0;   // (rows beyond the batch: the last one again, never stored)
                float x[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) x[q] = xr[q];
#pragma unroll
                for (int q = 0; q < 8; ++q) { const float d = x[q] - cv[q]; d2[b] = fmaf(d, d, d2[b]); }
            }
        } else {
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                vjf_cfloat* xr = xu + (size_t)min(b, nb - 1) * dxu + c0;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < nq) { const float d = xr[q] - cv[q]; d2[b] = fmaf(d, d, d2[b]); }
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

// per-trial loss terms and backward seeds (no 1/B): one wavefront per trial, 4 trials per workgroup; partial sums per workgroup
__global__ __launch_bounds__(256) void vjf_wide_loss_kernel(VjfPlan P, VjfWideArgs W) {
    __shared__ float s_sc[4][RS_N];
    const VjfTrialArgs& A = W.t;
    const int dz = P.dz, dy = P.dy, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int b = blockIdx.x * 4 + w;
    const bool ok = b < A.B;
    const bool warm = (A.flags & VJF_FLAG_WARM_UP) != 0;
    const float* S = A.state;
    const unsigned rbits = A.replay ? __hip_atomic_load(A.replay_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (A.replay && rbits == 0u) return;                        // (uniform: the usual step)
    const bool m_r = !(rbits & 1u), m_d = !(rbits & 2u), m_h = !(rbits & 4u);   // loss components kept (vjf_trial_kernel.h)
    const float rho = A.replay ? __hip_atomic_load(A.replay_rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : S[P.off[VJF_SLOT_LIK_LOGVAR]];
    const float sig = S[P.off[VJF_SLOT_TR_LOGVAR]];
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
                drow[P.colD_dpy + i] = m_r ? e * r : 0.f;
            }
        } else {                                               // likelihood.py:51-62
            for (int i = lane; i < dy; i += 64) {
                const float yv = yrow[i], pv = py[i];
                const float eta = fminf(pv, 10.f), ex = expf(eta);
                lrec += ex - yv * eta;
                const float r = pv - yv;
                ssey = fmaf(r, r, ssey);
                drow[P.colD_dpy + i] = (m_r && pv <= 10.f) ? (ex - yv) : 0.f;
            }
        }
        float zz = 0.f;                                        // pt.logvar = log sum_j Z[b][j]^2 (module.py:76), fixed-order lane sums
        for (int j = lane; j < P.n; j += 64) { const float z = W.Z[(size_t)b * P.n + j]; zz = fmaf(z, z, zz); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) zz += __shfl_xor(zz, o, 64);
        const float p = expf(-0.5f * sig), e = expf(-sig), plv = logf(zz);
        for (int j = lane; j < dz; j += 64) {                  // model.py:390-391, functional.py:62-75
            const float mp = W.PM[(size_t)b * dz + j], mu = A.mu_t[(size_t)b * dz + j], lv = A.lv_t[(size_t)b * dz + j];
            const float dsc = mp * p - mu * p;
            const float tr = expf(plv + lv - sig);
            ldyn += 0.5f * (dsc * dsc + sig) + 0.5f * tr;
            ent += 0.5f * lv;                                  // functional.py:25-29
            const float dx = A.E[(size_t)b * P.ldE + P.n + j];
            sdx2 = fmaf(dx, dx, sdx2);
            float dmu = 0.f, dlv = m_h ? -0.5f : 0.f;
            if (!warm && m_d) { dmu = -e * (mp - mu); dlv += 0.5f * tr; }
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
    if (threadIdx.x < RS_N && !A.replay) {                         // (a replay leaves the loss sums alone)
        float v = 0.f;
        if (threadIdx.x <= RS_SDX2) v = ((s_sc[0][threadIdx.x] + s_sc[1][threadIdx.x]) + s_sc[2][threadIdx.x]) + s_sc[3][threadIdx.x];
        A.partial[(size_t)blockIdx.x * RS_N + threadIdx.x] = v;
    }
}
