// vjf_trial_mfma_kernel.h -- K1 on the f32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
//
// Same contract as vjf_trial_kernel (see vjf_trial_kernel.h for the reference citations); this
// version is used when 16 trials' working set fits LDS.  One workgroup = 4 wavefronts = 16 trials.
// Every dense product is evaluated transposed, out^T (features x 16 trials) = A (features x K) *
// X^T (K x 16 trials), so that
//   * the trial index sits on the MFMA column (lane & 15): per-trial reductions stay inside a lane,
//   * the B operand is the LDS-resident activation matrix, stored feature-major [feature][17]
//     (17 = 16 trials + 1 pad word: operand reads and transposed copies are bank-conflict free),
//   * the A operand comes straight from L2 in "k-major" matrices (row k contiguous over features),
//     64-byte segments per k: w_chol, w_mean and the backward weights already have that layout,
//     the forward weights are read from their transposed copies in the aux buffer.
// w_chol is upper triangular (module.py:102), so variance tile j0 only runs k <= j0 + 15.
#pragma once
#include <hip/hip_runtime.h>
#include "vjf_plan.h"
#include "vjf_trial_kernel.h"      // VjfTrialArgs, group_sum

#define VJF_LDT 17
#define VJF_K1M_WAVES 16           // wavefronts per workgroup (two per SIMD: one hides the other's L2 operand latency)
#define VJF_K1M_THREADS (64 * VJF_K1M_WAVES)
typedef float vjf_f32x4 __attribute__((ext_vector_type(4)));

// acc(row = 4*(lane>>4)+r, col = lane&15) += sum_{k<K} Ag[k*lda + m0 + row] * Xs[k*17 + col]
// rows m0+i >= M contribute 0.
// (NT: the matrix as torch stores a Linear weight, A[m][k] = Ag[m * lda + k], instead of k-major -- the stand-alone operators)
template <bool NT = false>
__device__ __forceinline__ void mma_tile(vjf_f32x4& acc, const float* __restrict__ Ag, int lda_, int M, int m0,
                                         const float* Xs, int K, int lane) {
    const int i = lane & 15, kk = lane >> 4;
    const bool rv = (m0 + i) < M;
    const size_t lda = NT ? 1 : (size_t)lda_;            // distance of two k
    // (rows beyond M: read from a valid address and masked where the value is USED -- a select right behind a load makes the
    //  compiler wait for the load there, and the next batch would not be in flight beside this batch's MFMAs)
    const float* ap = Ag + (size_t)(rv ? m0 + i : 0) * (NT ? (size_t)lda_ : 1) + (size_t)kk * lda;
    const float* xp = Xs + kk * VJF_LDT + i;
    const int K32 = K & ~31;
    int k0 = 0;
    if (K32 > 0) {                                     // batches of 8 steps, the next batch's 16 operand loads in flight
        float a0[8], x0[8], a1[8], x1[8];              // while the current batch's MFMAs issue
#pragma unroll
        for (int q = 0; q < 8; ++q) { a0[q] = ap[(size_t)(4 * q) * lda]; x0[q] = xp[(4 * q) * VJF_LDT]; }
        for (; k0 < K32; k0 += 64) {
            const bool more1 = k0 + 32 < K32;
            if (more1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { a1[q] = ap[(size_t)(k0 + 32 + 4 * q) * lda]; x1[q] = xp[(k0 + 32 + 4 * q) * VJF_LDT]; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rv ? a0[q] : 0.f, x0[q], acc, 0, 0, 0);
            if (!more1) { k0 += 32; break; }
            const bool more0 = k0 + 64 < K32;
            if (more0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { a0[q] = ap[(size_t)(k0 + 64 + 4 * q) * lda]; x0[q] = xp[(k0 + 64 + 4 * q) * VJF_LDT]; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rv ? a1[q] : 0.f, x1[q], acc, 0, 0, 0);
            if (!more0) { k0 += 64; break; }
        }
    }
    // remainder (< 32 rows): its up to 8 steps' operands in flight together (clamped addresses, masked at use), then the same
    // MFMA steps in the same order as a step-by-step loop would issue them
    if (k0 < K) {
        const float* apc = ap;                                          // (rows beyond M: a valid address, masked below)
        float ar[8], xr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + 4 * q + kk;
            const bool kv = k < K;
            const float av = apc[(size_t)(kv ? k0 + 4 * q : 0) * lda], xv = xp[(kv ? k0 + 4 * q : 0) * VJF_LDT];
            ar[q] = (rv && kv) ? av : 0.f;
            xr[q] = kv ? xv : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (k0 + 4 * q < K) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[q], xr[q], acc, 0, 0, 0);   // (uniform)
    }
}

struct VjfTrialMfmaArgs {
    VjfTrialArgs t;
    const float* aux;      // transposed weights (VjfPlan::aux_*)
    unsigned long long* stamps;   // diagnostic only (null in normal runs)
    unsigned* done;        // += 1 per workgroup once it has read W, w_chol, sigma for the last time (null: not counted):
                           //   the post kernel, on another stream, waits for the count before it overwrites them
    unsigned* fwd_done;    // forward half: += 1 per workgroup once its E / ACT rows and posterior are written back to memory
                           //   (release at agent scope): the statistics Gram on another stream starts behind vjf_gate_kernel on it
    const unsigned* rls_done;  // backward half: workgroups of the previous step's post kernel that have their W, w_chol, sigma in
    unsigned rls_target;       //   memory; non-null -> the workgroup waits (bounded) for the count before stage 2, its reloads done
    int part;              // 0: whole step; 1: forward half (features, recognition, E / ACT rows, posterior);
                           // 2: backward half (predictive mean / variance, losses, backward, DEL rows) -- reloads the
                           //    forward half's rows, so that it can run after the RLS update of the previous step while
                           //    the forward half of this step ran beside it (vjf_filter_seq, two streams)
};

#define VJF_K1_STAMP(i)                                                                     \
    do {                                                                                    \
        if (AA.stamps && blockIdx.x == 0 && threadIdx.x == 0) {                             \
            unsigned long long t_;                                                          \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
            AA.stamps[i] = t_;                                                              \
        }                                                                                   \
    } while (0)

static inline size_t vjf_trial_mfma_lds_floats(const VjfPlan& P) {
    // Wide observations (dy >= hmax): the first delta buffer lives in the (by then dead) decoder-mean rows and the
    // second one exists only for networks with more than one hidden layer -- two workgroups per CU at dy = 200.
    const bool compact = P.dy >= P.hmax;
    const size_t nd = compact ? (P.L > 1 ? 1 : 0) : 2;
    const size_t feat = (size_t)P.din + P.dxu + P.n + P.hsum + nd * (size_t)P.hmax + 8 * (size_t)P.dz + 2 * (size_t)P.dy;
    return feat * VJF_LDT + 16 * RS_N + VJF_K1M_WAVES * 16 + 16 + 64;
}

__global__ __launch_bounds__(VJF_K1M_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void vjf_trial_mfma_kernel(VjfPlan P, VjfTrialMfmaArgs AA) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const VjfTrialArgs& A = AA.t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b0 = blockIdx.x * 16;
    const int nb = min(16, A.B - b0);
    const int dz = P.dz, dy = P.dy, du = P.du, n = P.n, din = P.din, dxu = P.dxu;
    const float* S = A.state;
    const bool prior = (A.mu_s == nullptr);
    const bool warm = (A.flags & VJF_FLAG_WARM_UP) != 0;
    const bool tri = S[P.off[VJF_SLOT_SCALARS] + VJF_SC_TRI_CLEAN] != 0.f;   // w_chol known upper triangular
    const bool fwd = AA.part != 2, bwd = AA.part != 1;
    const unsigned rbits = A.replay ? __hip_atomic_load(A.replay_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (A.replay && rbits == 0u) {                              // (uniform: the usual step)
        if (AA.done && tid == 0) __hip_atomic_fetch_add(AA.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a counted replay: the RLS update waits for it)
        return;
    }
    const bool m_r = !(rbits & 1u), m_d = !(rbits & 2u), m_h = !(rbits & 4u);   // loss components kept
    const bool handoff = AA.fwd_done != nullptr && !bwd;        // part 1 inside vjf_filter_seq
    constexpr int LD = VJF_LDT;
    constexpr int NW = VJF_K1M_WAVES;

    // ---- LDS carve: feature-major [feature][17] matrices
    float* s_in = smem;                          // din   [y | u | mu_s | lv_s]
    float* s_xu = s_in + din * LD;               // dxu   [xs | u]
    float* s_phi = s_xu + dxu * LD;              // n
    float* s_act = s_phi + n * LD;               // hsum  hidden activations, layer after layer
    const bool compact = dy >= P.hmax;           // see vjf_trial_mfma_lds_floats
    float* s_dd = s_act + P.hsum * LD;           // hmax  deltas (ping), hmax deltas (pong) unless compact
    float* s_mu = s_dd + (compact ? (P.L > 1 ? 1 : 0) : 2) * P.hmax * LD;   // dz    mu_t
    float* s_lv = s_mu + dz * LD;                // dz    lv_t   (directly after s_mu: the heads write 2dz rows)
    float* s_xt = s_lv + dz * LD;                // dz
    float* s_e2 = s_xt + dz * LD;                // dz    eps_t
    float* s_pm = s_e2 + dz * LD;                // dz    pt.mean
    float* s_dmu = s_pm + dz * LD;               // dz
    float* s_dlv = s_dmu + dz * LD;              // dz    (directly after s_dmu)
    float* s_dxt = s_dlv + dz * LD;              // dz
    float* s_py = s_dxt + dz * LD;               // dy
    float* s_dpy = s_py + dy * LD;               // dy
    float* s_d0 = compact ? s_py : s_dd;         // compact: written only after the losses have consumed s_py
    float* s_d1 = compact ? s_dd : s_dd + P.hmax * LD;   // used only when n_hidden > 1
    float* s_sc = s_dpy + dy * LD;               // 16 x RS_N per-trial scalars
    float* s_red = s_sc + 16 * RS_N;             // 4 x 16 variance partials
    float* s_plv = s_red + 16 * NW;              // 16 pt.logvar

    VJF_K1_STAMP(22);
    unsigned long long t_begin_ = 0;
    if (AA.stamps && tid == 0) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin_)::"memory");
    // ---- stage 0: inputs (coalesced global reads, transposed LDS writes), eps_t, xs
    for (int b = wave; b < 16; b += NW) {            // wavefront w stages trials w, w+4, ..; the lane walks the columns
        const bool ok = b < nb;
        const size_t g = (size_t)(b0 + b);
        for (int c = lane; c < din; c += 64) {
            float v = 0.f;
            if (ok) {
                if (c < dy) v = A.y[g * dy + c];
                else if (c < dy + du) v = A.u[g * du + (c - dy)];
                else if (c < dy + du + dz) { const int j = c - dy - du; v = prior ? S[P.off[VJF_SLOT_PRIOR_MEAN] + j] : A.mu_s[g * dz + j]; }
                else { const int j = c - dy - du - dz; v = prior ? S[P.off[VJF_SLOT_PRIOR_LOGVAR] + j] : A.lv_s[g * dz + j]; }
            }
            s_in[c * LD + b] = v;
        }
    }
    for (int e = tid; e < 16 * dz; e += VJF_K1M_THREADS) {
        const int b = e / dz, j = e - b * dz;
        s_e2[j * LD + b] = (b < nb) ? A.eps_t[(size_t)(b0 + b) * dz + j] : 0.f;
        s_xt[j * LD + b] = (b < nb) ? A.eps_s[(size_t)(b0 + b) * dz + j] : 0.f;      // eps_s parked in s_xt
    }
    __syncthreads();
    for (int e = tid; e < 16 * dxu; e += VJF_K1M_THREADS) {
        const int c = e >> 4, b = e & 15;
        float v;
        if (c < dz) v = fmaf(s_xt[c * LD + b], expf(0.5f * s_in[(dy + du + dz + c) * LD + b]), s_in[(dy + du + c) * LD + b]);
        else v = s_in[(dy + c - dz) * LD + b];
        s_xu[c * LD + b] = v;
    }
    __syncthreads();

    float* s_cen = s_d0;                           // n * dxu + n floats needed; available: 2 * hmax * 17, or s_py + s_dpy
    float* s_iw = s_cen + n * dxu;
    const bool stage_c = (n * dxu + n) <= 2 * (compact ? dy : P.hmax) * LD;
    VJF_K1_STAMP(23);
    // ---- stage 1: RBF features (functional.py:11-22); lanes walk the trial index
    if (!fwd) {                                   // backward half: the forward half left Phi in the E rows
        for (int b = wave; b < 16; b += NW)
            for (int c = lane; c < n; c += 64) s_phi[c * LD + b] = b < nb ? A.E[(size_t)(b0 + b) * P.ldE + c] : 0.f;
    } else {
        // centroids and -1/(2 w^2) staged in LDS (the delta buffers are free until the backward pass)
        const float* cen = S + P.off[VJF_SLOT_CENTROID];
        const float* lw = S + P.off[VJF_SLOT_LOGWIDTH];
        if (stage_c) {
            for (int e = tid; e < n * dxu; e += VJF_K1M_THREADS) s_cen[e] = cen[e];
            for (int e = tid; e < n; e += VJF_K1M_THREADS) { const float w = expf(lw[e]); s_iw[e] = -0.5f / (w * w); }
            __syncthreads();
        }
        for (int e = tid; e < 16 * n; e += VJF_K1M_THREADS) {
            const int k = e >> 4, b = e & 15;
            float d2 = 0.f;
            if (stage_c) {
                for (int c = 0; c < dxu; ++c) { const float d = s_xu[c * LD + b] - s_cen[k * dxu + c]; d2 = fmaf(d, d, d2); }
                s_phi[k * LD + b] = expf(d2 * s_iw[k]);
            } else {
                for (int c = 0; c < dxu; ++c) { const float d = s_xu[c * LD + b] - cen[k * dxu + c]; d2 = fmaf(d, d, d2); }
                const float w = expf(lw[k]);
                s_phi[k * LD + b] = expf(-0.5f * d2 / (w * w));
            }
        }
    }
    __syncthreads();

    VJF_K1_STAMP(25);
    // ---- stage 3: recognition forward (recognition.py:31-42)
    if (!fwd) {                                   // backward half: hidden activations from the ACT rows, posterior from the outputs
        for (int b = wave; b < 16; b += NW) {
            int aoff = 0;
            for (int l = 0; l < P.L; ++l) {
                const int hl = P.h[l], c0 = P.colA_act[l + 1];
                for (int k = lane; k < hl; k += 64) s_act[(aoff + k) * LD + b] = b < nb ? A.ACT[(size_t)(b0 + b) * P.ldA + c0 + k] : 0.f;
                aoff += hl;
            }
            for (int j = lane; j < dz; j += 64) {
                s_mu[j * LD + b] = b < nb ? A.mu_t[(size_t)(b0 + b) * dz + j] : 0.f;
                s_lv[j * LD + b] = b < nb ? A.lv_t[(size_t)(b0 + b) * dz + j] : 0.f;
            }
        }
    } else {
        const float* xin = s_in;
        int kin = din, aoff = 0;
        for (int l = 0; l < P.L; ++l) {
            const float* WT = AA.aux + P.aux_recT[l];                  // (kin, hl)
            const float* bias = S + P.off[VJF_SLOT_REC_B0 + 2 * l];
            float* out = s_act + aoff * LD;
            const int hl = P.h[l], mt = (hl + 15) >> 4;
            for (int t = wave; t < mt; t += NW) {
                vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                mma_tile(acc, WT, hl, hl, t * 16, xin, kin, lane);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int f = t * 16 + 4 * (lane >> 4) + r;
                    if (f < hl) out[f * LD + (lane & 15)] = tanhf(acc[r] + bias[f]);
                }
            }
            __syncthreads();
            xin = out; kin = hl; aoff += hl;
        }
        const float* HT = AA.aux + P.aux_headT;                        // (hL, 2dz): mean rows then logvar rows
        const float* bl = S + P.off[VJF_SLOT_LV_B];
        const int mt = (2 * dz + 15) >> 4;
        for (int t = wave; t < mt; t += NW) {
            vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_tile(acc, HT, 2 * dz, 2 * dz, t * 16, xin, kin, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = t * 16 + 4 * (lane >> 4) + r;
                if (f < 2 * dz) s_mu[f * LD + (lane & 15)] = acc[r] + (f >= dz ? bl[f - dz] : 0.f);   // rows dz.. land in s_lv
            }
        }
    }
    __syncthreads();

    VJF_K1_STAMP(26);
    // ---- stage 4: xt, posterior outputs, py = xt C^T + d (model.py:28-30)
    for (int e = tid; e < 16 * dz; e += VJF_K1M_THREADS) {
        const int j = e >> 4, b = e & 15;
        s_xt[j * LD + b] = fmaf(s_e2[j * LD + b], expf(0.5f * s_lv[j * LD + b]), s_mu[j * LD + b]);
    }
    if (fwd) for (int e = tid; e < nb * dz; e += VJF_K1M_THREADS) {                 // coalesced posterior stores
        const int b = e / dz, j = e - b * dz;
        A.mu_t[(size_t)(b0 + b) * dz + j] = s_mu[j * LD + b];
        A.lv_t[(size_t)(b0 + b) * dz + j] = s_lv[j * LD + b];
    }
    __syncthreads();
    if (bwd) {
        const float* CT = AA.aux + P.aux_decT;                         // (dz, dy)
        const float* d = S + P.off[VJF_SLOT_DEC_B];
        const int mt = (dy + 15) >> 4;
        for (int t = wave; t < mt; t += NW) {
            vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_tile(acc, CT, dy, dy, t * 16, s_xt, dz, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = t * 16 + 4 * (lane >> 4) + r;
                if (f < dy) s_py[f * LD + (lane & 15)] = acc[r] + d[f];
            }
        }
    }
    __syncthreads();

    VJF_K1_STAMP(27);
    if (!bwd) {
        // forward half: of the loss scalars only sum |dx|^2 (the RLS chain's residual identity needs it), summed
        // exactly as stage 5 does; then the E and ACT rows
        constexpr int LPT = VJF_K1M_THREADS / 16;
        const int b = tid / LPT, s = tid % LPT;
        float sdx2 = 0.f;
        for (int j = s; j < dz; j += LPT) {
            const float dx = s_xt[j * LD + b] - s_xu[j * LD + b];
            sdx2 = fmaf(dx, dx, sdx2);
        }
        sdx2 = group_sum<LPT>(sdx2);
        if (s == 0) s_sc[b * RS_N + RS_SDX2] = b < nb ? sdx2 : 0.f;
        __syncthreads();
        if (tid == RS_SDX2) {
            float v = 0.f;
            for (int bb = 0; bb < 16; ++bb) v += s_sc[bb * RS_N + tid];
            if (handoff) __hip_atomic_store(A.partial + (size_t)blockIdx.x * RS_N + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else A.partial[(size_t)blockIdx.x * RS_N + tid] = v;
        }
    }
    // (every part: as late as it can be -- recognition or its reload, xt and the decoder above do not need the RLS update)
    if (AA.rls_done && bwd) {
        // W, w_chol, sigma come from the post kernel of the previous step on another stream: the host only lets this kernel
        // start once that kernel's workgroups are resident (vjf_prep_kernel's last workgroup checks), so the wait cannot starve it
        if (tid == 0) {
            bool there = false;
            for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
                if ((int)(__hip_atomic_load(AA.rls_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - AA.rls_target) >= 0) { there = true; break; }
                if ((spins & 255u) == 255u && vjf_abort_seen(A.state + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS)) break;
                __builtin_amdgcn_s_sleep(VJF_POLL_SLEEP);
            }
            if (!there) vjf_status_or(const_cast<float*>(A.state) + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_K1);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        }
    VJF_K1_STAMP(24);
    // ---- stage 2 (runs in front of stage 5): predictive variance sum_j (Phi w_chol)_j^2 (module.py:75-76) and
    //      pt.mean = xs + Phi W (module.py:77)
    if (bwd) {
        const float* Wc = S + P.off[VJF_SLOT_W_CHOL];
        const int ntile = (n + 15) >> 4;
        float v2 = 0.f;
        // tiles in descending cost, dealt to the 4 wavefronts in a snake so that the triangular work balances
        for (int r = 0;; ++r) {
            const int idx = (r & 1) ? r * NW + NW - 1 - wave : r * NW + wave;
            if (idx >= ntile) { if (r * NW >= ntile) break; else continue; }
            const int t = ntile - 1 - idx, j0 = t * 16;
            const int K = tri ? min(n, j0 + 16) : n;
            vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_tile(acc, Wc, n, n, j0, s_phi, K, lane);
            v2 = fmaf(acc[0], acc[0], fmaf(acc[1], acc[1], fmaf(acc[2], acc[2], fmaf(acc[3], acc[3], v2))));
        }
        v2 += __shfl_xor(v2, 16, 64);
        v2 += __shfl_xor(v2, 32, 64);
        if (lane < 16) s_red[wave * 16 + lane] = v2;
        // mean tiles, dealt from the last wavefront backwards (it has the lightest variance share)
        const float* Wm = S + P.off[VJF_SLOT_W_MEAN];
        const int mt = (dz + 15) >> 4;
        for (int t = NW - 1 - wave; t < mt; t += NW) {
            vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_tile(acc, Wm, dz, dz, t * 16, s_phi, n, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = t * 16 + 4 * (lane >> 4) + r, b = lane & 15;
                if (j < dz) s_pm[j * LD + b] = s_xu[j * LD + b] + acc[r];
            }
        }
    }
    __syncthreads();
    if (bwd && tid < 16) {
        float v = 0.f;
        for (int w = 0; w < NW; ++w) v += s_red[w * 16 + tid];
        s_plv[tid] = logf(v);
    }
    __syncthreads();
    // ---- stage 5: per-trial loss terms and backward seeds (no 1/B); 16 lanes per trial
    if (bwd) {
        constexpr int LPT = VJF_K1M_THREADS / 16;          // lanes per trial
        const int b = tid / LPT, s = tid % LPT;
        const float rho = A.replay ? __hip_atomic_load(A.replay_rho, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : S[P.off[VJF_SLOT_LIK_LOGVAR]];
        // (sigma may have been written while this kernel was already running: a load that bypasses L1 / the scalar cache)
        const float sig = __hip_atomic_load(S + P.off[VJF_SLOT_TR_LOGVAR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float lrec = 0.f, ssey = 0.f;
        if (P.lik == VJF_LIK_GAUSSIAN) {                               // likelihood.py:19-26, functional.py:54-73
            const float p = expf(-0.5f * rho), e = expf(-rho);
            for (int i = s; i < dy; i += LPT) {
                const float yv = s_in[i * LD + b], pv = s_py[i * LD + b];
                const float r = pv - yv, dsc = yv * p - pv * p;
                lrec += 0.5f * (dsc * dsc + rho);
                ssey = fmaf(r, r, ssey);
                s_dpy[i * LD + b] = m_r ? e * r : 0.f;
            }
        } else {                                                       // likelihood.py:51-62
            for (int i = s; i < dy; i += LPT) {
                const float yv = s_in[i * LD + b], pv = s_py[i * LD + b];
                const float eta = fminf(pv, 10.f), ex = expf(eta);
                lrec += ex - yv * eta;
                const float r = pv - yv;
                ssey = fmaf(r, r, ssey);
                s_dpy[i * LD + b] = (m_r && pv <= 10.f) ? (ex - yv) : 0.f;
            }
        }
        lrec = group_sum<LPT>(lrec);
        ssey = group_sum<LPT>(ssey);
        float ldyn = 0.f, ent = 0.f, sdx2 = 0.f;
        {
            const float p = expf(-0.5f * sig), e = expf(-sig), plv = s_plv[b];
            for (int j = s; j < dz; j += LPT) {                         // model.py:390-391, functional.py:62-75
                const float mp = s_pm[j * LD + b], mu = s_mu[j * LD + b], lv = s_lv[j * LD + b];
                const float dsc = mp * p - mu * p;
                const float tr = expf(plv + lv - sig);
                ldyn += 0.5f * (dsc * dsc + sig) + 0.5f * tr;
                ent += 0.5f * lv;                                      // functional.py:25-29
                const float dx = s_xt[j * LD + b] - s_xu[j * LD + b];
                sdx2 = fmaf(dx, dx, sdx2);
                float dmu = 0.f, dlv = m_h ? -0.5f : 0.f;
                if (!warm && m_d) { dmu = -e * (mp - mu); dlv += 0.5f * tr; }
                s_dmu[j * LD + b] = dmu;
                s_dlv[j * LD + b] = dlv;
            }
        }
        ldyn = group_sum<LPT>(ldyn);
        ent = group_sum<LPT>(ent);
        sdx2 = group_sum<LPT>(sdx2);
        if (s == 0) {
            const bool ok = b < nb;
            s_sc[b * RS_N + RS_LRECON] = ok ? lrec : 0.f;
            s_sc[b * RS_N + RS_LDYN] = ok ? ldyn : 0.f;
            s_sc[b * RS_N + RS_ENT] = ok ? ent : 0.f;
            s_sc[b * RS_N + RS_SSEY] = ok ? ssey : 0.f;
            s_sc[b * RS_N + RS_SDX2] = ok ? sdx2 : 0.f;
        }
    }
    __syncthreads();
    if (bwd && !A.replay && tid < RS_N && (fwd || tid != RS_SDX2)) {   // (the forward half / part owns sum |dx|^2; a replay leaves the sums alone)
        float v = 0.f;
        if (tid <= RS_SDX2) for (int b = 0; b < 16; ++b) v += s_sc[b * RS_N + tid];
        A.partial[(size_t)blockIdx.x * RS_N + tid] = v;
    }

    VJF_K1_STAMP(28);
    // ---- stage 6: backward.  dxt = dpy C ; dmu += dxt ; dlv += dxt eps_t exp(lv/2)/2
    if (bwd) {
        const float* C = S + P.off[VJF_SLOT_DEC_W];                    // (dy, dz): k-major for this product
        const int mt = (dz + 15) >> 4;
        for (int t = wave; t < mt; t += NW) {
            vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_tile(acc, C, dz, dz, t * 16, s_dpy, dy, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = t * 16 + 4 * (lane >> 4) + r, b = lane & 15;
                if (j < dz) {
                    s_dmu[j * LD + b] += acc[r];
                    s_dlv[j * LD + b] = fmaf(acc[r] * s_e2[j * LD + b], 0.5f * expf(0.5f * s_lv[j * LD + b]), s_dlv[j * LD + b]);
                }
            }
        }
    }
    __syncthreads();
    if (bwd) {
        const int hL = P.h[P.L - 1];
        const float* Wm = S + P.off[VJF_SLOT_MEAN_W];                  // (dz, hL): k-major for dh = dmu Wm + dlv Wl
        const float* Wl = S + P.off[VJF_SLOT_LV_W];
        const float* hact = s_act + (P.hsum - hL) * LD;
        int mt = (hL + 15) >> 4;
        for (int t = wave; t < mt; t += NW) {
            vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            mma_tile(acc, Wm, hL, hL, t * 16, s_dmu, dz, lane);
            mma_tile(acc, Wl, hL, hL, t * 16, s_dlv, dz, lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = t * 16 + 4 * (lane >> 4) + r, b = lane & 15;
                if (k < hL) { const float hv = hact[k * LD + b]; s_d0[k * LD + b] = acc[r] * (1.f - hv * hv); }
            }
        }
        __syncthreads();
        int aoff = P.hsum - hL;
        float* cur = s_d0; float* nxt = s_d1;
        for (int l = P.L - 1; l >= 0; --l) {
            const int hl = P.h[l];
            for (int e = tid; e < nb * hl; e += VJF_K1M_THREADS) {                  // da_l -> DEL, coalesced over k
                const int b = e / hl, k = e - b * hl;
                A.DEL[(size_t)(b0 + b) * P.ldD + P.colD_da[l] + k] = cur[k * LD + b];
            }
            if (l > 0) {
                const int hp = P.h[l - 1];
                const float* W = S + P.off[VJF_SLOT_REC_W0 + 2 * l];   // (hl, hp): k-major for dh_{l-1} = da_l W
                const float* hprev = s_act + (aoff - hp) * LD;
                mt = (hp + 15) >> 4;
                for (int t = wave; t < mt; t += NW) {
                    vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    mma_tile(acc, W, hp, hp, t * 16, cur, hl, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = t * 16 + 4 * (lane >> 4) + r, b = lane & 15;
                        if (k < hp) { const float hv = hprev[k * LD + b]; nxt[k * LD + b] = acc[r] * (1.f - hv * hv); }
                    }
                }
                __syncthreads();
                float* tmp = cur; cur = nxt; nxt = tmp;
                aoff -= hp;
            }
        }
    }

    // ---- stage 7: rows of E = [Phi | dx | 0], ACT = [in|1|h_1|1|..|h_L|1|xt|1|0], DEL = [.. | dmu | dlv | dpy].
    //      wavefront w writes the rows of trials w, w+4, ...; the lane walks the columns (coalesced, no divisions)
    for (int b = wave; b < nb; b += NW) {
        if (fwd) {
            float* erow = A.E + (size_t)(b0 + b) * P.ldE;
            for (int c = lane; c < P.ldE; c += 64) {
                float v = 0.f;
                if (c < n) v = s_phi[c * LD + b];
                else if (c < n + dz) v = s_xt[(c - n) * LD + b] - s_xu[(c - n) * LD + b];
                // (forward half of the sequence: the statistics Gram on another stream takes these rows -- write-through)
                if (handoff) __hip_atomic_store(erow + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else erow[c] = v;
            }
            float* arow = A.ACT + (size_t)(b0 + b) * P.ldA;
            for (int c = lane; c <= din; c += 64) arow[c] = c < din ? s_in[c * LD + b] : 1.f;
            int aoff = 0;
            for (int l = 0; l < P.L; ++l) {
                const int hl = P.h[l], c0 = P.colA_act[l + 1];
                for (int k = lane; k <= hl; k += 64) arow[c0 + k] = k < hl ? s_act[(aoff + k) * LD + b] : 1.f;
                aoff += hl;
            }
            for (int j = lane; P.colA_xt + j < P.ldA; j += 64) arow[P.colA_xt + j] = j < dz ? s_xt[j * LD + b] : (j == dz ? 1.f : 0.f);
        }
        if (!bwd) continue;
        float* drow = A.DEL + (size_t)(b0 + b) * P.ldD;
        for (int c = lane; c < 2 * dz + dy; c += 64) {                      // (dlv follows dmu in DEL as s_dlv follows s_dmu)
            if (c < 2 * dz) drow[P.colD_dmu + c] = s_dmu[c * LD + b];
            else drow[P.colD_dpy + c - 2 * dz] = s_dpy[(c - 2 * dz) * LD + b];
        }
    }
    VJF_K1_STAMP(30);
    if (AA.stamps && tid == 0) {
        // diagnostic: when the LAST workgroup ends (slot 29), and per workgroup its start / end (10 ns ticks after block 0's start)
        // and placement, in the unused columns of its loss partials (tools/k1_tail.py)
        unsigned long long t_;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        atomicMax(AA.stamps + 29, t_);
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        A.partial[(size_t)blockIdx.x * RS_N + 7] = (float)(long long)(t_ - AA.stamps[22]);
        A.partial[(size_t)blockIdx.x * RS_N + 5] = (float)(long long)(t_begin_ - AA.stamps[22]);
        A.partial[(size_t)blockIdx.x * RS_N + 6] = (float)(((xcc & 15u) << 8) | ((hw >> 8) & 15u) | (((hw >> 13) & 7u) << 4));   // xcc | se | cu
    }
    if (AA.done && bwd && tid == 0) __hip_atomic_fetch_add(AA.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (AA.fwd_done && !bwd) {
        // producer side of a hand-off between launches on different streams: what the other stream takes (E rows, sum |dx|^2)
        // went out as write-through stores, in memory once every storing wavefront's vmcnt has drained; the workgroup barrier,
        // then the relaxed agent-scope count -- no L2 write-back by 256 workgroups
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(AA.fwd_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Refresh the transposed weight copies from the canonical tensors (run at the start of an API call:
// the caller may have written the state blob; within a sequence vjf_prep_kernel keeps them in step).
__global__ void vjf_aux_kernel(VjfPlan P, const float* state, float* aux) {
    for (int t = 0; t < P.n_train; ++t) {
        if (P.tr_aux[t] < 0) continue;
        const int rows = P.tr_rows[t], cols = P.tr_cols[t];
        for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < rows * cols; e += gridDim.x * blockDim.x) {
            const int r = e / cols, c = e - r * cols;
            aux[P.tr_aux[t] + (size_t)c * P.tr_auxld[t] + P.tr_auxcol[t] + r] = state[P.tr_off[t] + e];
        }
    }
}
