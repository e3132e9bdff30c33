// vjf_mega_kernel.h -- vjf_filter_seq / vjf_filter_step on a single rank as ONE cooperative launch per chunk of steps.
//
// Every piece of a filtering step (vjf/model.py:179-221) is a ROLE played by workgroups of the same grid, one workgroup per
// compute unit, all resident for the whole chunk (the launch is cooperative: the runtime refuses a grid that does not fit):
//
//   RLS roles        workgroup 0: the Cholesky loop (vjf_chol_loop), 1: the y / W loop, 2 .. 1 + 2 nbl: the inverse loops
//                    (vjf_rls_post_loop) -- module.py:94-102, model.py:373-377, exactly as before
//   trial role       n_trial workgroups; workgroup w owns the 32-trial tiles w, w + n_trial, ..  Per tile and step: reparametrise,
//                    RBF features, recognition network, posterior, decoder (model.py:97-122), Phi^T dx of the tile; then -- behind the
//                    RLS update of the previous step -- predictive mean / variance, loss terms (model.py:124-154), hand-derived backward
//                     and the tile's weight / bias gradients on the matrix cores (K = 32 trials); last the features of
//                    the NEXT step (they depend only on this step's posterior) into the rows the Gram role reads.  A workgroup's sums
//                    over its tiles leave as two slabs: early [Phi^T dx | sum dx^2], late [gradients | loss sums].
//   Gram role        n_gram workgroups: Phi^T Phi (module.py:96) one step AHEAD -- rows staged in LDS, all 28 lower 32x32 tiles per
//                    workgroup on v_mfma_f32_32x32x2_f32, partial tiles to a slab, then every workgroup sums its share of the slabs
//   operand role     ceil(n / 16) workgroups: sum of the early slabs -> Phi^T dx, g = P W + Phi^T dx / v, P += Phi^T Phi / v (module.py:94-96)
//   SGD role         n_sgd workgroups: sum of the late slabs, finite guards and loss (model.py:138-154), clip + SGD (model.py:210-211),
//                    likelihood running variance (likelihood.py:28-40)
//
// Hand-offs are monotone workgroup counters in memory: producer = write-through stores, every storing wavefront drains vmcnt, the
// workgroup barrier, one relaxed agent-scope add; consumer = one lane polls (bounded), one agent-scope acquire, vmcnt drained,
// barrier, plain loads (MI355X guide, Guideline 16).  Nothing ever waits for work of a launch that has not been submitted: every
// producer is a workgroup of this grid, and the grid is resident as a whole.  All sums are taken in a fixed order: results do not
// depend on timing, and a sequence cut into chunks gives the same bits as one piece.
#pragma once
#include <hip/hip_runtime.h>
#include "vjf_chol_kernel.h"
#include "vjf_plan.h"
#include "vjf_post_kernel.h"
#include "vjf_trial_mfma_kernel.h"   // vjf_f32x4

#define VJF_MG_THREADS 512
#define VJF_MG_WAVES 8
#define VJF_MG_TR 32                 // trials per tile: two column groups of v_mfma_f32_16x16x4_f32 share every A operand
#define VJF_MG_LD 33                 // LDS matrices are feature-major [feature][32 trials + 1 pad]
#define VJF_MG_GROWS 64              // rows of Phi staged per pass of the Gram role
#define VJF_MG_MAXQ 4                // 32x32 tiles of Phi^T Phi per wavefront of a Gram workgroup (28 lower tiles / 8)

// counters: one per 64-byte line of the block that the host zeroes before every launch
enum {
    MG_C_PHI = 0,        // trial workgroups whose Phi rows of an EVEN event e are in memory    target (e / 2 + 1) n_trial
    MG_C_PHI1 = 176,     // ... of an odd event.  (Two counters: a workgroup signals event e + 1 at the end of step e without waiting for
                         // anybody's event e -- at step 0 both are its own -- so one count could reach an event's target with a
                         // workgroup missing.  It cannot run two events ahead: step e + 1 starts behind everybody's late slab of step e.)
    MG_C_FWD = 16,       // trial workgroups whose early slab of step t is in memory           target (t + 1) n_trial
    MG_C_K1 = 32,        // trial workgroups that have read W, w_chol, sigma of step t - 1      target (t + 1) n_trial
    MG_C_BWD = 48,       // trial workgroups whose late slab of step t is in memory            target (t + 1) n_trial
    MG_C_GRAM = 64,      // Gram workgroups whose partial tiles of event e are in memory       target (e + 1) n_gram
    MG_C_STAT = 80,      // Gram workgroups whose share of Phi^T Phi of event e is reduced      target (e + 1) n_gram
    MG_C_PREP = 96,      // operand workgroups done with step t                                target (t + 1) n_prep
    MG_C_SGD = 112,      // SGD workgroups done with step t                                    target (t + 1) n_sgd
    MG_C_PDONE = 128,    // RLS workgroups (y / W loop + inverse loops) done with step t       target (t + 1) (2 nbl + 1)
    MG_C_STARTED = 144,
    MG_C_COLFLAGS = 160, // [0 .. VJF_CHOL_MAXBLK]: column flags of the Cholesky loop; [VJF_CHOL_MAXBLK + 2]: its "operands loaded" word
    MG_C_WORDS = 256
};

struct VjfMegaArgs {
    int T, B, ntiles;
    int n_rls, n_trial, n_gram, n_prep, n_sgd;        // grid = their sum
    const float* y; const float* u; const float* eps; const float* mu0; const float* lv0;
    float* mu; float* lv; float* loss;
    float* state; float* aux;
    float* E0; float* E1;                             // Phi rows of even / odd events, (B, ldE)
    float* slab_early; float* slab_late; float* gslab;
    float* red0; float* red1;                         // reduce buffers of even / odd steps ([G | FDX | sums], as the RLS loops read them)
    float* gbuf;                                      // g (n, dz)
    unsigned* cnt;
    unsigned flags;
    int early_len, late_len;                          // floats per trial workgroup
    int gram_rows;                                    // rows of Phi per Gram workgroup (a multiple of 2)
    unsigned long long* stamps;                       // diagnostic (null in normal runs): s_memrealtime of trial workgroup 0, 16 per step
};

// ---- LDS of the trial role (floats); the host uses the same function to size the launch
struct VjfMegaTrialLds {
    int cen, iw, in, xu, phi, act, dd, mu, lv, xt, e2, pm, dmu, dlv, dx, py, dpy, one, zero, sc, red, plv, wg, total;
    int nd;
};
__host__ __device__ inline VjfMegaTrialLds vjf_mega_trial_lds(const VjfPlan& P) {
    VjfMegaTrialLds l;
    const int LD = VJF_MG_LD;
    int o = 0;
    auto take = [&](int nfl) { const int at = o; o += (nfl + 3) & ~3; return at; };
    l.cen = take(P.n * P.dxu); l.iw = take(P.n);
    l.in = take(P.din * LD); l.xu = take(P.dxu * LD); l.phi = take(P.n * LD); l.act = take(P.hsum * LD);
    const bool compact = P.dy >= P.hmax;              // the first delta buffer lives in the (by then dead) decoder-mean rows
    l.nd = compact ? (P.L > 1 ? 1 : 0) : (P.L > 1 ? 2 : 1);
    l.dd = take(l.nd * P.hmax * LD);
    l.mu = take(P.dz * LD); l.lv = take(P.dz * LD); l.xt = take(P.dz * LD); l.e2 = take(P.dz * LD); l.pm = take(P.dz * LD);
    l.dmu = take(P.dz * LD); l.dlv = take(P.dz * LD); l.dx = take(P.dz * LD);
    l.py = take(P.dy * LD); l.dpy = take(P.dy * LD);
    l.one = take(LD); l.zero = take(LD);
    l.sc = take(VJF_MG_TR * RS_N); l.red = take(VJF_MG_WAVES * VJF_MG_TR); l.plv = take(VJF_MG_TR); l.wg = take(16);
    l.total = o;
    return l;
}
// the mu / lv / xt / e2 / pm / dmu / dlv / dx rows must be adjacent in this order (the heads write 2 dz rows at mu, the ahead
// features park xs' in the 3 dz rows at dmu): take() pads to 4 floats, so dz * LD must be a multiple of 4 or the code below
// addresses through the struct's offsets only -- it does (no pointer arithmetic across fields except mu -> lv and dmu -> dlv,
// which are handled explicitly).

static inline size_t vjf_mega_gram_lds_floats(const VjfPlan& P) { return (size_t)VJF_MG_GROWS * P.ldE + 64; }
static inline size_t vjf_mega_prep_lds_floats(const VjfPlan& P) {
    return (size_t)16 * VJF_PREPG_LDP(P.n) + (size_t)P.n * 17 + (size_t)VJF_MG_WAVES * 16 * 17 + 16 * 17 + 64;
}

// acc_g(row = 4*(lane>>4)+r, col = lane&15) += sum_{k<K} Ag[k*lda + m0 + row] * Xs[k*LD + 16 g + col]   (g = 0, 1)
// rows m0 + i >= M contribute 0.  k runs in steps of 4, the operands of 8 steps in flight while the previous 8 steps' MFMAs issue.
__device__ __forceinline__ void mg_mma2(vjf_f32x4& acc0, vjf_f32x4& acc1, const float* __restrict__ Ag, int lda, int M, int m0,
                                        const float* Xs, int K, int lane) {
    constexpr int LD = VJF_MG_LD;
    const int i = lane & 15, kk = lane >> 4;
    const bool rv = (m0 + i) < M;
    const float* ap = Ag + (rv ? m0 + i : 0) + (size_t)kk * lda;
    const float* xp = Xs + kk * LD + i;
    const int K32 = K & ~31;
    int k0 = 0;
    if (K32 > 0) {
        float a0[8], x0[8], y0[8], a1[8], x1[8], y1[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { a0[q] = ap[(size_t)(4 * q) * lda]; x0[q] = xp[(4 * q) * LD]; y0[q] = xp[(4 * q) * LD + 16]; }
        for (; k0 < K32; k0 += 64) {
            const bool more1 = k0 + 32 < K32;
            if (more1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { a1[q] = ap[(size_t)(k0 + 32 + 4 * q) * lda]; x1[q] = xp[(k0 + 32 + 4 * q) * LD]; y1[q] = xp[(k0 + 32 + 4 * q) * LD + 16]; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float a = rv ? a0[q] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x0[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, y0[q], acc1, 0, 0, 0);
            }
            if (!more1) { k0 += 32; break; }
            const bool more0 = k0 + 64 < K32;
            if (more0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { a0[q] = ap[(size_t)(k0 + 64 + 4 * q) * lda]; x0[q] = xp[(k0 + 64 + 4 * q) * LD]; y0[q] = xp[(k0 + 64 + 4 * q) * LD + 16]; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float a = rv ? a1[q] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, x1[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, y1[q], acc1, 0, 0, 0);
            }
            if (!more0) { k0 += 64; break; }
        }
    }
    if (k0 < K) {                                      // remainder (< 32 rows): clamped addresses, masked at use
        float ar[8], xr[8], yr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + 4 * q + kk;
            const bool kv = k < K;
            const int kc = kv ? k0 + 4 * q : 0;
            const float av = ap[(size_t)kc * lda], xv = xp[kc * LD], yv = xp[kc * LD + 16];
            ar[q] = (rv && kv) ? av : 0.f;
            xr[q] = kv ? xv : 0.f;
            yr[q] = kv ? yv : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (k0 + 4 * q < K) {                      // (uniform)
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[q], xr[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(ar[q], yr[q], acc1, 0, 0, 0);
            }
    }
}

__device__ __forceinline__ float mg_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mg_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// slab entry: the first tile of a workgroup in a step stores, later tiles add (the workgroup's own bytes, all through sc1)
__device__ __forceinline__ void mg_slab(float* p, float v, bool first) { mg_st(p, first ? v : mg_ld(p) + v); }

#define VJF_MG_STAMP(i)                                                                     \
    do {                                                                                    \
        if (A.stamps && wg == 0 && tid == 0) {                                              \
            unsigned long long t_;                                                          \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
            A.stamps[(size_t)(t & 31) * 16 + (i)] = t_;                                     \
        }                                                                                   \
    } while (0)

// one 16x16 tile of  G[m][j] = sum_{b<32} D[m0+m][b] * Bop[j0+j][b],  Bop = [Bact (Kin rows) | ones | 0..]  -> slab
__device__ __forceinline__ void mg_grad_tile(const float* D, int M, int m0, const float* Bact, int Kin, int j0, const float* s_one,
                                             const float* s_zero, float* slab, int dstW, int ld, int dstB, bool first, int lane) {
    constexpr int LD = VJF_MG_LD;
    const int i = lane & 15, kk = lane >> 4;
    const float* arow = ((m0 + i) < M ? D + (size_t)(m0 + i) * LD : s_zero) + kk;
    const int jj = j0 + i;
    const float* brow = (jj < Kin ? Bact + (size_t)jj * LD : (jj == Kin ? s_one : s_zero)) + kk;
    float a[8], b[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { a[s] = arow[4 * s]; b[s] = brow[4 * s]; }
    vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 8; s += 2) {                   // two chains: the MFMAs issue back to back
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s + 1], b[s + 1], acc1, 0, 0, 0);
    }
    acc += acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * (lane >> 4) + r, j = j0 + (lane & 15);
        if (m >= M) continue;
        if (j < Kin) mg_slab(slab + dstW + (size_t)m * ld + j, acc[r], first);
        else if (j == Kin && dstB >= 0) mg_slab(slab + dstB + m, acc[r], first);
    }
}

// ------------------------------------------------------------------------------------------------ trial role
__device__ __forceinline__ void vjf_mega_trial(const VjfPlan& P, const VjfMegaArgs& A, float* smem, const int wg) {
    constexpr int LD = VJF_MG_LD, NW = VJF_MG_WAVES, NT = VJF_MG_THREADS, TR = VJF_MG_TR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int dz = P.dz, dy = P.dy, du = P.du, n = P.n, din = P.din, dxu = P.dxu;
    const float* S = A.state;
    float* SCW = A.state + P.off[VJF_SLOT_SCALARS];
    const bool warm = false;                           // (this launch only runs sgd + update without warm-up)
    const VjfMegaTrialLds Lo = vjf_mega_trial_lds(P);
    float* s_cen = smem + Lo.cen; float* s_iw = smem + Lo.iw;
    float* s_in = smem + Lo.in; float* s_xu = smem + Lo.xu; float* s_phi = smem + Lo.phi; float* s_act = smem + Lo.act;
    float* s_dd = smem + Lo.dd;
    float* s_mu = smem + Lo.mu; float* s_lv = smem + Lo.lv; float* s_xt = smem + Lo.xt; float* s_e2 = smem + Lo.e2; float* s_pm = smem + Lo.pm;
    float* s_dmu = smem + Lo.dmu; float* s_dlv = smem + Lo.dlv; float* s_dx = smem + Lo.dx;
    float* s_py = smem + Lo.py; float* s_dpy = smem + Lo.dpy;
    float* s_one = smem + Lo.one; float* s_zero = smem + Lo.zero;
    float* s_sc = smem + Lo.sc; float* s_red = smem + Lo.red; float* s_plv = smem + Lo.plv; float* s_wg = smem + Lo.wg;
    const bool compact = dy >= P.hmax;
    float* s_d0 = compact ? s_py : s_dd;               // compact: written only after the losses have consumed s_py
    float* s_d1 = compact ? s_dd : s_dd + P.hmax * LD; // used only when n_hidden > 1
    unsigned* cnt = A.cnt;
    const unsigned npost = (unsigned)(A.n_rls - 1);
    float* late = A.slab_late + (size_t)wg * A.late_len;
    const size_t sy = (size_t)A.B * dy, su = (size_t)A.B * du, sz = (size_t)A.B * dz;
    int ntl = 0;
    for (int tile = wg; tile < A.ntiles; tile += A.n_trial) ++ntl;

    // centroids and -1/(2 w^2): constant for the launch (functional.py:11-22)
    {
        const float* cen = S + P.off[VJF_SLOT_CENTROID];
        const float* lw = S + P.off[VJF_SLOT_LOGWIDTH];
        for (int e = tid; e < n * dxu; e += NT) s_cen[e] = cen[e];
        for (int e = tid; e < n; e += NT) { const float w = expf(lw[e]); s_iw[e] = -0.5f / (w * w); }
        if (tid < LD) s_zero[tid] = 0.f;
    }
    __syncthreads();

    // Phi(x) of the trials of one tile from xs' rows parked at `s_xn` -> rows of Eout (write-through: the Gram role takes them)
    auto phi_rows = [&](const float* s_xn, float* Eout, int b0, int nb) {
        for (int b = wave; b < nb; b += NW) {
            float* erow = Eout + (size_t)(b0 + b) * P.ldE;
            for (int k = lane; k < P.ldE; k += 64) {
                float v = 0.f;
                if (k < n) {
                    float d2 = 0.f;
                    for (int c = 0; c < dxu; ++c) { const float d = s_xn[c * LD + b] - s_cen[k * dxu + c]; d2 = fmaf(d, d, d2); }
                    v = expf(d2 * s_iw[k]);
                }
                mg_st(erow + k, v);
            }
        }
    };

    for (int t = 0; t < A.T; ++t) {
        const float* y_t = A.y + (size_t)t * sy;
        const float* u_t = A.u ? A.u + (size_t)t * su : nullptr;
        const float* mu_s = t ? A.mu + (size_t)(t - 1) * sz : A.mu0;
        const float* lv_s = t ? A.lv + (size_t)(t - 1) * sz : A.lv0;
        const float* eps_s = A.eps + (size_t)t * 2 * sz;
        const float* eps_t = eps_s + sz;
        float* mu_t = A.mu + (size_t)t * sz;
        float* lv_t = A.lv + (size_t)t * sz;
        const bool prior = (mu_s == nullptr);
        float* E_now = (t & 1) ? A.E1 : A.E0;
        float* E_next = (t & 1) ? A.E0 : A.E1;
        // early slabs alternate between two sets: the operand role may read step t's long after this workgroup has started
        // step t + 1 (it also waits for the Gram of step t); step t + 2 starts behind the RLS update of step t, which consumed them
        float* early = A.slab_early + ((size_t)(t & 1) * A.n_trial + wg) * A.early_len;
        VJF_MG_STAMP(0);
        // theta of the previous step (the SGD role's write-through stores)
        if (t > 0 && !vjf_wg_wait(cnt + MG_C_SGD, (unsigned)t * (unsigned)A.n_sgd, tid, SCW + VJF_SC_STATUS)) {
            vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE);
        }
        if (vjf_abort_seen(SCW + VJF_SC_STATUS)) return;
        VJF_MG_STAMP(1);
        if (tid < 16) s_wg[tid] = 0.f;
        float sig = 0.f, rho = 0.f;
        bool tri = false;
        int it = 0;
        for (int tile = wg; tile < A.ntiles; tile += A.n_trial, ++it) {
            const bool first = it == 0, last = it == ntl - 1;
            const int b0 = tile * TR;
            const int nb = min(TR, A.B - b0);
            __syncthreads();                           // (the previous tile's readers of the LDS matrices are done)
            // ---- stage 0: inputs (coalesced global reads, transposed LDS writes), eps_t, xs
            for (int b = wave; b < TR; b += NW) {
                const bool ok = b < nb;
                const size_t g = (size_t)(b0 + b);
                for (int c = lane; c < din; c += 64) {
                    float v = 0.f;
                    if (ok) {
                        if (c < dy) v = y_t[g * dy + c];
                        else if (c < dy + du) v = u_t[g * du + (c - dy)];
                        else if (c < dy + du + dz) { const int j = c - dy - du; v = prior ? S[P.off[VJF_SLOT_PRIOR_MEAN] + j] : mu_s[g * dz + j]; }
                        else { const int j = c - dy - du - dz; v = prior ? S[P.off[VJF_SLOT_PRIOR_LOGVAR] + j] : lv_s[g * dz + j]; }
                    }
                    s_in[c * LD + b] = v;
                }
            }
            for (int e = tid; e < TR * dz; e += NT) {
                const int b = e / dz, j = e - b * dz;
                s_e2[j * LD + b] = (b < nb) ? eps_t[(size_t)(b0 + b) * dz + j] : 0.f;
                s_xt[j * LD + b] = (b < nb) ? eps_s[(size_t)(b0 + b) * dz + j] : 0.f;      // eps_s parked in s_xt
            }
            if (tid < LD) s_one[tid] = tid < nb ? 1.f : 0.f;
            __syncthreads();
            for (int e = tid; e < TR * dxu; e += NT) {
                const int c = e >> 5, b = e & 31;
                float v;
                if (c < dz) v = fmaf(s_xt[c * LD + b], expf(0.5f * s_in[(dy + du + dz + c) * LD + b]), s_in[(dy + du + c) * LD + b]);
                else v = s_in[(dy + c - dz) * LD + b];
                s_xu[c * LD + b] = v;
            }
            __syncthreads();
            // ---- stage 1: RBF features (functional.py:11-22)
            for (int e = tid; e < TR * n; e += NT) {
                const int k = e >> 5, b = e & 31;
                float d2 = 0.f;
                for (int c = 0; c < dxu; ++c) { const float d = s_xu[c * LD + b] - s_cen[k * dxu + c]; d2 = fmaf(d, d, d2); }
                s_phi[k * LD + b] = expf(d2 * s_iw[k]);
            }
            __syncthreads();
            if (t == 0) {                              // first step of the launch: nobody wrote this step's rows a step ahead
                phi_rows(s_xu, E_now, b0, nb);
                if (last) vjf_wg_signal_wt(cnt + MG_C_PHI, tid);                   // event 0
            }
            if (first) VJF_MG_STAMP(2);
            // ---- stage 3: recognition forward (recognition.py:31-42)
            {
                const float* xin = s_in;
                int kin = din, aoff = 0;
                for (int l = 0; l < P.L; ++l) {
                    const float* WT = A.aux + P.aux_recT[l];                   // (kin, hl)
                    const float* bias = S + P.off[VJF_SLOT_REC_B0 + 2 * l];
                    float* out = s_act + aoff * LD;
                    const int hl = P.h[l], mt = (hl + 15) >> 4;
                    for (int tt = wave; tt < mt; tt += NW) {
                        vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                        mg_mma2(acc0, acc1, WT, hl, hl, tt * 16, xin, kin, lane);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = tt * 16 + 4 * (lane >> 4) + r;
                            if (f < hl) {
                                const float bf = bias[f];
                                out[f * LD + (lane & 15)] = tanhf(acc0[r] + bf);
                                out[f * LD + 16 + (lane & 15)] = tanhf(acc1[r] + bf);
                            }
                        }
                    }
                    __syncthreads();
                    xin = out; kin = hl; aoff += hl;
                }
                const float* HT = A.aux + P.aux_headT;                         // (hL, 2dz): mean rows then logvar rows
                const float* bl = S + P.off[VJF_SLOT_LV_B];
                const int mt = (2 * dz + 15) >> 4;
                for (int tt = wave; tt < mt; tt += NW) {
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    mg_mma2(acc0, acc1, HT, 2 * dz, 2 * dz, tt * 16, xin, kin, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = tt * 16 + 4 * (lane >> 4) + r;
                        if (f < 2 * dz) {
                            float* row = f < dz ? s_mu + f * LD : s_lv + (f - dz) * LD;
                            const float bf = f >= dz ? bl[f - dz] : 0.f;
                            row[lane & 15] = acc0[r] + bf;
                            row[16 + (lane & 15)] = acc1[r] + bf;
                        }
                    }
                }
            }
            __syncthreads();
            if (first) VJF_MG_STAMP(3);
            // ---- stage 4: xt, dx, posterior out, py = xt C^T + d (model.py:28-30)
            for (int e = tid; e < TR * dz; e += NT) {
                const int j = e >> 5, b = e & 31;
                const float xt = fmaf(s_e2[j * LD + b], expf(0.5f * s_lv[j * LD + b]), s_mu[j * LD + b]);
                s_xt[j * LD + b] = xt;
                s_dx[j * LD + b] = b < nb ? xt - s_xu[j * LD + b] : 0.f;
            }
            for (int e = tid; e < nb * dz; e += NT) {                          // coalesced posterior stores
                const int b = e / dz, j = e - b * dz;
                mu_t[(size_t)(b0 + b) * dz + j] = s_mu[j * LD + b];
                lv_t[(size_t)(b0 + b) * dz + j] = s_lv[j * LD + b];
            }
            __syncthreads();
            {
                // sum |dx|^2 per trial (16 lanes each), then the tile's sum in trial order
                constexpr int LPT = NT / TR;
                const int b = tid / LPT, sl = tid % LPT;
                float sdx2 = 0.f;
                for (int j = sl; j < dz; j += LPT) { const float dx = s_dx[j * LD + b]; sdx2 = fmaf(dx, dx, sdx2); }
                sdx2 = group_sum<LPT>(sdx2);
                if (sl == 0) s_sc[b * RS_N + RS_SDX2] = sdx2;
            }
            {
                const float* CT = A.aux + P.aux_decT;                          // (dz, dy)
                const float* d = S + P.off[VJF_SLOT_DEC_B];
                const int mt = (dy + 15) >> 4;
                for (int tt = wave; tt < mt; tt += NW) {
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    mg_mma2(acc0, acc1, CT, dy, dy, tt * 16, s_xt, dz, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = tt * 16 + 4 * (lane >> 4) + r;
                        if (f < dy) { const float df = d[f]; s_py[f * LD + (lane & 15)] = acc0[r] + df; s_py[f * LD + 16 + (lane & 15)] = acc1[r] + df; }
                    }
                }
            }
            // early slab: Phi^T dx of this tile (module.py:94), 16 features x 16 columns per MFMA tile, K = 32 trials
            {
                const int mt = (n + 15) >> 4;
                for (int tt = NW - 1 - wave; tt < mt; tt += NW) {
                    const int m0 = tt * 16, i = lane & 15, kk = lane >> 4;
                    const float* arow = ((m0 + i) < n ? s_phi + (size_t)(m0 + i) * LD : s_zero) + kk;
                    const float* brow = (i < dz ? s_dx + (size_t)i * LD : s_zero) + kk;
                    float a[8], b[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) { a[s] = arow[4 * s]; b[s] = brow[4 * s]; }
                    vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < 8; s += 2) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acc, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s + 1], b[s + 1], acc1, 0, 0, 0);
                    }
                    acc += acc1;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = m0 + 4 * (lane >> 4) + r;
                        if (f < n) mg_slab(early + (size_t)f * 16 + (lane & 15), acc[r], first);
                    }
                }
            }
            __syncthreads();
            if (tid == 0) {
                float v = 0.f;
                for (int bb = 0; bb < TR; ++bb) v += s_sc[bb * RS_N + RS_SDX2];
                s_wg[RS_SDX2] += v;
                if (last) mg_st(early + (size_t)n * 16 + RS_SDX2, s_wg[RS_SDX2]);
            }
            if (last) vjf_wg_signal_wt(cnt + MG_C_FWD, tid);
            if (first) VJF_MG_STAMP(4);
            // ---- the RLS update of the previous step (W, w_chol, sigma: write-through stores of the RLS roles)
            if (first) {
                if (t > 0 && !vjf_wg_wait(cnt + MG_C_PDONE, (unsigned)t * npost, tid, SCW + VJF_SC_STATUS))
                    vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_K1);
                if (vjf_abort_seen(SCW + VJF_SC_STATUS)) return;
                sig = mg_ld(S + P.off[VJF_SLOT_TR_LOGVAR]);
                rho = mg_ld(S + P.off[VJF_SLOT_LIK_LOGVAR]);
                tri = mg_ld(SCW + VJF_SC_TRI_CLEAN) != 0.f;                   // w_chol known upper triangular
                VJF_MG_STAMP(5);
            }
            // ---- stage 2: predictive variance sum_j (Phi w_chol)_j^2 (module.py:75-76) and pt.mean = xs + Phi W (module.py:77)
            {
                const float* Wc = S + P.off[VJF_SLOT_W_CHOL];
                const int ntile = (n + 15) >> 4;
                float v2a = 0.f, v2b = 0.f;
                // tiles in descending cost, dealt to the wavefronts in a snake so that the triangular work balances
                for (int r = 0;; ++r) {
                    const int idx = (r & 1) ? r * NW + NW - 1 - wave : r * NW + wave;
                    if (idx >= ntile) { if (r * NW >= ntile) break; else continue; }
                    const int tt = ntile - 1 - idx, j0 = tt * 16;
                    const int K = tri ? min(n, j0 + 16) : n;
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    mg_mma2(acc0, acc1, Wc, n, n, j0, s_phi, K, lane);
                    v2a = fmaf(acc0[0], acc0[0], fmaf(acc0[1], acc0[1], fmaf(acc0[2], acc0[2], fmaf(acc0[3], acc0[3], v2a))));
                    v2b = fmaf(acc1[0], acc1[0], fmaf(acc1[1], acc1[1], fmaf(acc1[2], acc1[2], fmaf(acc1[3], acc1[3], v2b))));
                }
                v2a += __shfl_xor(v2a, 16, 64); v2a += __shfl_xor(v2a, 32, 64);
                v2b += __shfl_xor(v2b, 16, 64); v2b += __shfl_xor(v2b, 32, 64);
                if (lane < 16) { s_red[wave * TR + lane] = v2a; s_red[wave * TR + 16 + lane] = v2b; }
                const float* Wm = S + P.off[VJF_SLOT_W_MEAN];
                const int mt = (dz + 15) >> 4;
                for (int tt = NW - 1 - wave; tt < mt; tt += NW) {              // (the last wavefront has the lightest variance share)
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    mg_mma2(acc0, acc1, Wm, dz, dz, tt * 16, s_phi, n, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = tt * 16 + 4 * (lane >> 4) + r, b = lane & 15;
                        if (j < dz) { s_pm[j * LD + b] = s_xu[j * LD + b] + acc0[r]; s_pm[j * LD + 16 + b] = s_xu[j * LD + 16 + b] + acc1[r]; }
                    }
                }
            }
            __syncthreads();
            if (last && tid == 0) __hip_atomic_fetch_add(cnt + MG_C_K1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // W, w_chol, sigma read
            if (tid < TR) {
                float v = 0.f;
                for (int w = 0; w < NW; ++w) v += s_red[w * TR + tid];
                s_plv[tid] = logf(v);
            }
            __syncthreads();
            if (first) VJF_MG_STAMP(6);
            // ---- stage 5: per-trial loss terms and backward seeds (no 1/B); 16 lanes per trial
            {
                constexpr int LPT = NT / TR;
                const int b = tid / LPT, s = tid % LPT;
                const bool ok = b < nb;
                float lrec = 0.f, ssey = 0.f;
                if (P.lik == VJF_LIK_GAUSSIAN) {                               // likelihood.py:19-26, functional.py:54-73
                    const float p = expf(-0.5f * rho), e = expf(-rho);
                    for (int i = s; i < dy; i += LPT) {
                        const float yv = s_in[i * LD + b], pv = s_py[i * LD + b];
                        const float r = pv - yv, dsc = yv * p - pv * p;
                        lrec += 0.5f * (dsc * dsc + rho);
                        ssey = fmaf(r, r, ssey);
                        s_dpy[i * LD + b] = ok ? e * r : 0.f;
                    }
                } else {                                                       // likelihood.py:51-62
                    for (int i = s; i < dy; i += LPT) {
                        const float yv = s_in[i * LD + b], pv = s_py[i * LD + b];
                        const float eta = fminf(pv, 10.f), ex = expf(eta);
                        lrec += ex - yv * eta;
                        const float r = pv - yv;
                        ssey = fmaf(r, r, ssey);
                        s_dpy[i * LD + b] = (ok && pv <= 10.f) ? (ex - yv) : 0.f;
                    }
                }
                lrec = group_sum<LPT>(lrec);
                ssey = group_sum<LPT>(ssey);
                float ldyn = 0.f, ent = 0.f;
                {
                    const float p = expf(-0.5f * sig), e = expf(-sig), plv = s_plv[b];
                    for (int j = s; j < dz; j += LPT) {                         // model.py:390-391, functional.py:62-75
                        const float mp = s_pm[j * LD + b], mu = s_mu[j * LD + b], lv = s_lv[j * LD + b];
                        const float dsc = mp * p - mu * p;
                        const float tr = expf(plv + lv - sig);
                        ldyn += 0.5f * (dsc * dsc + sig) + 0.5f * tr;
                        ent += 0.5f * lv;                                      // functional.py:25-29
                        float dmu = 0.f, dlv = -0.5f;
                        if (!warm) { dmu = -e * (mp - mu); dlv += 0.5f * tr; }
                        s_dmu[j * LD + b] = ok ? dmu : 0.f;
                        s_dlv[j * LD + b] = ok ? dlv : 0.f;
                    }
                }
                ldyn = group_sum<LPT>(ldyn);
                ent = group_sum<LPT>(ent);
                if (s == 0) {
                    s_sc[b * RS_N + RS_LRECON] = ok ? lrec : 0.f;
                    s_sc[b * RS_N + RS_LDYN] = ok ? ldyn : 0.f;
                    s_sc[b * RS_N + RS_ENT] = ok ? ent : 0.f;
                    s_sc[b * RS_N + RS_SSEY] = ok ? ssey : 0.f;
                }
            }
            __syncthreads();
            if (tid < RS_SDX2) {                                               // (RS_LRECON, RS_LDYN, RS_ENT, RS_SSEY)
                float v = 0.f;
                for (int bb = 0; bb < TR; ++bb) v += s_sc[bb * RS_N + tid];
                s_wg[tid] += v;
            }
            // ---- stage 6: backward.  dxt = dpy C ; dmu += dxt ; dlv += dxt eps_t exp(lv/2)/2
            int gbase = 0;                                                     // running tile count: gradient tiles go round the wavefronts
            auto grad_tensor = [&](const float* D, int M, const float* Bact, int Kin, int slotW, int slotB) {
                const int dstW = P.off[slotW] - P.train_off, dstB = slotB >= 0 ? P.off[slotB] - P.train_off : -1;
                const int ntm = (M + 15) >> 4, ntj = (Kin + 1 + 15) >> 4;
                for (int q = 0; q < ntm * ntj; ++q)
                    if (((gbase + q) & (NW - 1)) == wave) {
                        const int tm = q / ntj, tj = q - tm * ntj;
                        mg_grad_tile(D, M, tm * 16, Bact, Kin, tj * 16, s_one, s_zero, late, dstW, Kin, dstB, first, lane);
                    }
                gbase += ntm * ntj;
            };
            // decoder gradients need only dpy and xt (model.py:28-30): before dpy's consumers move on
            grad_tensor(s_dpy, dy, s_xt, dz, VJF_SLOT_DEC_W, VJF_SLOT_DEC_B);
            {
                const float* C = S + P.off[VJF_SLOT_DEC_W];                    // (dy, dz): k-major for this product
                const int mt = (dz + 15) >> 4;
                for (int tt = wave; tt < mt; tt += NW) {
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    mg_mma2(acc0, acc1, C, dz, dz, tt * 16, s_dpy, dy, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = tt * 16 + 4 * (lane >> 4) + r;
                        if (j < dz) {
#pragma unroll
                            for (int g = 0; g < 2; ++g) {
                                const int b = 16 * g + (lane & 15);
                                const float a = g ? acc1[r] : acc0[r];
                                const float m = (b < nb) ? 1.f : 0.f;          // (padding trials carry no gradient)
                                s_dmu[j * LD + b] += m * a;
                                s_dlv[j * LD + b] = fmaf(m * a * s_e2[j * LD + b], 0.5f * expf(0.5f * s_lv[j * LD + b]), s_dlv[j * LD + b]);
                            }
                        }
                    }
                }
            }
            __syncthreads();
            if (first) VJF_MG_STAMP(7);
            {
                const int hL = P.h[P.L - 1];
                const float* Wm = S + P.off[VJF_SLOT_MEAN_W];                  // (dz, hL): k-major for dh = dmu Wm + dlv Wl
                const float* Wl = S + P.off[VJF_SLOT_LV_W];
                const float* hact = s_act + (P.hsum - hL) * LD;
                // head gradients: [dmu ; dlv]^T [h_L | 1]
                grad_tensor(s_dmu, dz, hact, hL, VJF_SLOT_MEAN_W, -1);
                grad_tensor(s_dlv, dz, hact, hL, VJF_SLOT_LV_W, VJF_SLOT_LV_B);
                int mt = (hL + 15) >> 4;
                for (int tt = wave; tt < mt; tt += NW) {
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    mg_mma2(acc0, acc1, Wm, hL, hL, tt * 16, s_dmu, dz, lane);
                    mg_mma2(acc0, acc1, Wl, hL, hL, tt * 16, s_dlv, dz, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = tt * 16 + 4 * (lane >> 4) + r, b = lane & 15;
                        if (k < hL) {
                            const float h0 = hact[k * LD + b], h1 = hact[k * LD + 16 + b];
                            s_d0[k * LD + b] = acc0[r] * (1.f - h0 * h0);
                            s_d0[k * LD + 16 + b] = acc1[r] * (1.f - h1 * h1);
                        }
                    }
                }
                __syncthreads();
                int aoff = P.hsum - hL;
                float* cur = s_d0; float* nxt = s_d1;
                for (int l = P.L - 1; l >= 0; --l) {
                    const int hl = P.h[l];
                    const int hp = l > 0 ? P.h[l - 1] : din;
                    const float* hprev = l > 0 ? s_act + (aoff - hp) * LD : s_in;
                    grad_tensor(cur, hl, hprev, hp, VJF_SLOT_REC_W0 + 2 * l, VJF_SLOT_REC_B0 + 2 * l);
                    if (l > 0) {
                        const float* W = S + P.off[VJF_SLOT_REC_W0 + 2 * l];   // (hl, hp): k-major for dh_{l-1} = da_l W
                        mt = (hp + 15) >> 4;
                        for (int tt = wave; tt < mt; tt += NW) {
                            vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                            mg_mma2(acc0, acc1, W, hp, hp, tt * 16, cur, hl, lane);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int k = tt * 16 + 4 * (lane >> 4) + r, b = lane & 15;
                                if (k < hp) {
                                    const float h0 = hprev[k * LD + b], h1 = hprev[k * LD + 16 + b];
                                    nxt[k * LD + b] = acc0[r] * (1.f - h0 * h0);
                                    nxt[k * LD + 16 + b] = acc1[r] * (1.f - h1 * h1);
                                }
                            }
                        }
                        __syncthreads();
                        float* tmp = cur; cur = nxt; nxt = tmp;
                        aoff -= hp;
                    }
                }
            }
            if (first) VJF_MG_STAMP(8);
            if (last) {
                // the workgroup's late slab is complete: loss sums, then the signal the SGD role waits for
                __syncthreads();
                if (tid < RS_SDX2) mg_st(late + P.train_len + tid, s_wg[tid]);
                vjf_wg_signal_wt(cnt + MG_C_BWD, tid);
                VJF_MG_STAMP(9);
            }
            // ---- features of the NEXT step (they depend only on this posterior): xs' = mu_t + eps_s' e^{lv_t/2}, the same operations
            //      in the same order as stages 0 / 1 of the next step.  Behind the RLS wait: the rows they overwrite were read by the
            //      Gram of two events ago, which the Cholesky loop of the previous step waited for.
            if (t + 1 < A.T) {
                const float* eps_n = A.eps + (size_t)(t + 1) * 2 * sz;
                const float* u_n = A.u ? A.u + (size_t)(t + 1) * su : nullptr;
                float* s_xn = s_dmu;                                           // dxu <= 3 dz rows: dmu, dlv, dx (host checks)
                __syncthreads();
                for (int e = tid; e < TR * dxu; e += NT) {
                    const int c = e >> 5, b = e & 31;
                    const size_t g = (size_t)(b0 + (b < nb ? b : 0));
                    float v;
                    if (c < dz) v = fmaf(eps_n[g * dz + c], expf(0.5f * s_lv[c * LD + b]), s_mu[c * LD + b]);
                    else v = u_n[g * du + (c - dz)];
                    s_xn[c * LD + b] = v;
                }
                __syncthreads();
                phi_rows(s_xn, E_next, b0, nb);
                if (last) vjf_wg_signal_wt(cnt + (((t + 1) & 1) ? MG_C_PHI1 : MG_C_PHI), tid);   // event t + 1
            }
        }
        VJF_MG_STAMP(10);
    }
}

// ------------------------------------------------------------------------------------------------ Gram role
// Phi^T Phi of event e (the features of step e), lower 32x32 tiles, from the rows the trial role wrote
__device__ __forceinline__ void vjf_mega_gram(const VjfPlan& P, const VjfMegaArgs& A, float* lds, const int hg) {
    constexpr int NT = VJF_MG_THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.n, nbl = (n + 31) / 32, ntri = nbl * (nbl + 1) / 2, ldE = P.ldE;
    float* SCW = A.state + P.off[VJF_SLOT_SCALARS];
    float* s_rows = lds;                               // [VJF_MG_GROWS][ldE]
    int* s_tab = reinterpret_cast<int*>(s_rows + (size_t)VJF_MG_GROWS * ldE);   // tile -> (bi << 8) | bj
    if (tid < ntri) {
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= tid) ++bi;
        s_tab[tid] = (bi << 8) | (tid - bi * (bi + 1) / 2);
    }
    __syncthreads();
    const int r0 = hg * A.gram_rows, r1 = min(A.B, r0 + A.gram_rows);
    float* myslab = A.gslab + (size_t)hg * ntri * 1024;
    const int c = lane & 31, kh = lane >> 5;
    for (int e = 0; e < A.T; ++e) {
        const float* E = (e & 1) ? A.E1 : A.E0;
        float* red = (e & 1) ? A.red1 : A.red0;
        if (!vjf_wg_wait(A.cnt + ((e & 1) ? MG_C_PHI1 : MG_C_PHI), (unsigned)(e / 2 + 1) * (unsigned)A.n_trial, tid, SCW + VJF_SC_STATUS))
            vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE2);
        // (the slab of the previous event: every Gram workgroup has summed its share)
        if (e > 0 && !vjf_wg_wait(A.cnt + MG_C_STAT, (unsigned)e * (unsigned)A.n_gram, tid, SCW + VJF_SC_STATUS))
            vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE2);
        if (vjf_abort_seen(SCW + VJF_SC_STATUS)) return;
        vjf_f32x16 acc[VJF_MG_MAXQ];
#pragma unroll
        for (int q = 0; q < VJF_MG_MAXQ; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
        for (int c0 = r0; c0 < r1; c0 += VJF_MG_GROWS) {
            __syncthreads();
            const int l4 = ldE >> 2;
            for (int idx = tid; idx < VJF_MG_GROWS * l4; idx += NT) {          // rows beyond the range: zero
                const int r = idx / l4, q4 = idx - r * l4;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 + r < r1) v = *reinterpret_cast<const float4*>(E + (size_t)(c0 + r) * ldE + 4 * q4);
                *reinterpret_cast<float4*>(s_rows + (size_t)r * ldE + 4 * q4) = v;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < VJF_MG_MAXQ; ++q) {
                const int tt = wave + VJF_MG_WAVES * q;
                if (tt < ntri) {
                    const int code = s_tab[tt], bi = code >> 8, bj = code & 255;
                    const float* pa = s_rows + (size_t)kh * ldE + bi * 32 + c;
                    const float* pb = s_rows + (size_t)kh * ldE + bj * 32 + c;
#pragma unroll 4
                    for (int s = 0; s < VJF_MG_GROWS / 2; s += 8) {
                        float a[8], b[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { a[u] = pa[(size_t)(2 * (s + u)) * ldE]; b[u] = pb[(size_t)(2 * (s + u)) * ldE]; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[q], 0, 0, 0);
                    }
                }
            }
        }
        // partial tiles out (write-through), accumulator layout: column = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
        for (int q = 0; q < VJF_MG_MAXQ; ++q) {
            const int tt = wave + VJF_MG_WAVES * q;
            if (tt < ntri) {
                float* sl = myslab + (size_t)tt * 1024;
#pragma unroll
                for (int i = 0; i < 16; ++i) mg_st(sl + ((i & 3) + 8 * (i >> 2) + 4 * kh) * 32 + c, acc[q][i]);
            }
        }
        vjf_wg_signal_wt(A.cnt + MG_C_GRAM, tid);
        if (!vjf_wg_wait(A.cnt + MG_C_GRAM, (unsigned)(e + 1) * (unsigned)A.n_gram, tid, SCW + VJF_SC_STATUS))
            vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE2);
        if (vjf_abort_seen(SCW + VJF_SC_STATUS)) return;
        // this workgroup's share of the sum over the slabs, in workgroup order
        for (int idx = hg * NT + tid; idx < ntri * 1024; idx += A.n_gram * NT) {
            float v = 0.f;
            int h = 0;
            for (; h + 16 <= A.n_gram; h += 16) {
                float tq[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) tq[q] = A.gslab[(size_t)(h + q) * ntri * 1024 + idx];
#pragma unroll
                for (int q = 0; q < 16; ++q) v += tq[q];
            }
            for (; h < A.n_gram; ++h) v += A.gslab[(size_t)h * ntri * 1024 + idx];
            const int tt = idx >> 10, el = idx & 1023, code = s_tab[tt];
            const int gr = (code >> 8) * 32 + (el >> 5), gc = (code & 255) * 32 + (el & 31);
            if (gr < n && gc <= gr) {
                mg_st(red + P.red_G + (size_t)gr * n + gc, v);
                mg_st(red + P.red_G + (size_t)gc * n + gr, v);
            }
        }
        vjf_wg_signal_wt(A.cnt + MG_C_STAT, tid);
    }
}

// ------------------------------------------------------------------------------------------------ operand role
// g = P W + Phi^T dx / v and P += Phi^T Phi / v for 16 rows (module.py:94-96); Phi^T dx = sum of the trial workgroups' early slabs
__device__ __forceinline__ void vjf_mega_prep(const VjfPlan& P, const VjfMegaArgs& A, float* lds, const int pw) {
    constexpr int NT = VJF_MG_THREADS, NW = VJF_MG_WAVES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.n, dz = P.dz, i0 = pw * 16, ldp = VJF_PREPG_LDP(n);
    float* s_p = lds;                                  // [16][n + 4]  rows of P before the update
    float* s_w = s_p + 16 * ldp;                       // [n][17]      W, columns dz..15 zero
    float* s_r = s_w + (size_t)n * 17;                 // [NW][16][17] per-wavefront partial products
    float* s_f = s_r + NW * 16 * 17;                   // [16][17]     Phi^T dx rows
    float* S = A.state;
    float* SCW = S + P.off[VJF_SLOT_SCALARS];
    const unsigned npost = (unsigned)(A.n_rls - 1);
    const unsigned* runw = A.cnt + MG_C_COLFLAGS + VJF_CHOL_MAXBLK + 2;
    for (int t = 0; t < A.T; ++t) {
        float* red = (t & 1) ? A.red1 : A.red0;
        bool ok = vjf_wg_wait(A.cnt + MG_C_FWD, (unsigned)(t + 1) * (unsigned)A.n_trial, tid, SCW + VJF_SC_STATUS);
        ok = vjf_wg_wait(A.cnt + MG_C_STAT, (unsigned)(t + 1) * (unsigned)A.n_gram, tid, SCW + VJF_SC_STATUS) && ok;
        if (t > 0) ok = vjf_wg_wait(A.cnt + MG_C_PDONE, (unsigned)t * npost, tid, SCW + VJF_SC_STATUS) && ok;
        ok = vjf_wg_wait(runw, (unsigned)(t + 1), tid, SCW + VJF_SC_STATUS) && ok;      // the Cholesky loop holds its operands (it reads the state's P at step 0)
        if (!ok) vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_OPERAND);
        if (vjf_abort_seen(SCW + VJF_SC_STATUS)) return;
        // Phi^T dx rows i0 .. i0 + 15: sum over the trial workgroups in workgroup order
        if (tid < 256) {
            const int r = tid >> 4, cc = tid & 15;
            float v = 0.f;
            if (i0 + r < n) {
                const float* src = A.slab_early + (size_t)(t & 1) * A.n_trial * A.early_len + (size_t)(i0 + r) * 16 + cc;
                int w = 0;
                for (; w + 16 <= A.n_trial; w += 16) {
                    float tq[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) tq[q] = src[(size_t)(w + q) * A.early_len];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v += tq[q];
                }
                for (; w < A.n_trial; ++w) v += src[(size_t)w * A.early_len];
                if (cc < dz) mg_st(red + P.red_FDX + (size_t)(i0 + r) * dz + cc, v);
            }
            s_f[r * 17 + cc] = v;
        } else if (pw == 0 && tid == 256) {
            float v = 0.f;
            for (int w = 0; w < A.n_trial; ++w) v += A.slab_early[((size_t)(t & 1) * A.n_trial + w) * A.early_len + (size_t)n * 16 + RS_SDX2];
            mg_st(red + P.red_SC + RS_SDX2, v);
        }
        const float inv_v = expf(-mg_ld(S + P.off[VJF_SLOT_TR_LOGVAR]));
        float* Pm = S + P.off[VJF_SLOT_W_PREC];
        const float* Wm = S + P.off[VJF_SLOT_W_MEAN];
        const float* G = red + P.red_G;
        const int n4 = n >> 2;
        for (int e0 = tid; e0 < 16 * n4; e0 += 4 * NT) {
            float4 p[4], g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * NT, row = e / n4, c4 = (e - row * n4) * 4;
                const bool in = e < 16 * n4 && i0 + row < n;
                const size_t off = in ? (size_t)(i0 + row) * n + c4 : 0;
                p[q] = *reinterpret_cast<const float4*>(Pm + off);
                g[q] = *reinterpret_cast<const float4*>(G + off);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * NT, row = e / n4, c4 = (e - row * n4) * 4;
                if (e >= 16 * n4) continue;
                const bool in = i0 + row < n;
                float* d = s_p + row * ldp + c4;
                d[0] = in ? p[q].x : 0.f; d[1] = in ? p[q].y : 0.f; d[2] = in ? p[q].z : 0.f; d[3] = in ? p[q].w : 0.f;
                if (in) {
                    float* dstp = Pm + (size_t)(i0 + row) * n + c4;
                    mg_st(dstp, fmaf(g[q].x, inv_v, p[q].x)); mg_st(dstp + 1, fmaf(g[q].y, inv_v, p[q].y));
                    mg_st(dstp + 2, fmaf(g[q].z, inv_v, p[q].z)); mg_st(dstp + 3, fmaf(g[q].w, inv_v, p[q].w));
                }
            }
        }
        for (int e = tid; e < n * 16; e += NT) {
            const int k = e >> 4, cc = e & 15;
            s_w[k * 17 + cc] = cc < dz ? Wm[(size_t)k * dz + cc] : 0.f;
        }
        __syncthreads();
        {
            const int i = lane & 15, kk = lane >> 4;
            vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int s4 = wave; s4 < n4; s4 += NW) {       // k-step s4 covers k = 4 s4 .. 4 s4 + 3
                const float a = s_p[i * ldp + 4 * s4 + kk];
                const float b = s_w[(4 * s4 + kk) * 17 + i];
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) s_r[(wave * 16 + 4 * (lane >> 4) + r) * 17 + (lane & 15)] = acc[r];
        }
        __syncthreads();
        if (tid < 256) {
            const int r = tid >> 4, cc = tid & 15;
            if (cc < dz && i0 + r < n) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) v += s_r[(w * 16 + r) * 17 + cc];
                mg_st(A.gbuf + (size_t)(i0 + r) * dz + cc, v + s_f[r * 17 + cc] * inv_v);
            }
        }
        vjf_wg_signal_wt(A.cnt + MG_C_PREP, tid);
    }
}

// ------------------------------------------------------------------------------------------------ SGD role
__device__ __forceinline__ void vjf_mega_sgd(const VjfPlan& P, const VjfMegaArgs& A, float* lds, const int sw) {
    constexpr int NT = VJF_MG_THREADS;
    const int tid = threadIdx.x;
    float* s_sc = lds;                                 // RS_N loss sums
    float* S = A.state;
    float* SC = S + P.off[VJF_SLOT_SCALARS];
    const float Bf = (float)A.B, invB = 1.0f / Bf;
    for (int t = 0; t < A.T; ++t) {
        if (!vjf_wg_wait(A.cnt + MG_C_BWD, (unsigned)(t + 1) * (unsigned)A.n_trial, tid, SC + VJF_SC_STATUS))
            vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_RESIDENT);
        if (vjf_abort_seen(SC + VJF_SC_STATUS)) return;
        // loss sums of the step: fp64, 32 strided partial sums per scalar, then a fixed xor tree (every SGD workgroup, for the guards)
        if (tid < 32 * RS_SDX2) {
            const int sc = tid >> 5, l = tid & 31;
            double v = 0.0;
            for (int w = l; w < A.n_trial; w += 32) v += (double)A.slab_late[(size_t)w * A.late_len + P.train_len + sc];
            v = vjf_sum32(v);
            if (l == 0) s_sc[sc] = (float)v;
        }
        __syncthreads();
        float l_recon = s_sc[RS_LRECON] * invB, l_dyn = s_sc[RS_LDYN] * invB, ent = s_sc[RS_ENT] * invB;
        const bool ok_r = isfinite(l_recon), ok_d = isfinite(l_dyn), ok_h = isfinite(ent);
        const bool grad_ok = ok_r && ok_h && ok_d;
        if (grad_ok) {
            const float lr_dec = SC[VJF_SC_LR_DEC], lr_rec = SC[VJF_SC_LR_REC];
            const bool freeze = SC[VJF_SC_FREEZE_DEC] != 0.f;
            for (int e = sw * NT + tid; e < P.train_len; e += A.n_sgd * NT) {
                int tens = -1;
                for (int q = 0; q < P.n_train; ++q) {
                    const int o = P.tr_off[q] - P.train_off;
                    if (e >= o && e < o + P.tr_rows[q] * P.tr_cols[q]) { tens = q; break; }
                }
                if (tens < 0) continue;                                        // (alignment padding between tensors)
                const bool dec = P.tr_dec[tens] != 0;
                if (dec && freeze) continue;
                float v = 0.f;
                int w = 0;
                const float* src = A.slab_late + e;
                for (; w + 16 <= A.n_trial; w += 16) {
                    float tq[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) tq[q] = src[(size_t)(w + q) * A.late_len];
#pragma unroll
                    for (int q = 0; q < 16; ++q) v += tq[q];
                }
                for (; w < A.n_trial; ++w) v += src[(size_t)w * A.late_len];
                float g = v * invB;
                g = fminf(fmaxf(g, -1.f), 1.f);                                // clip_grad_value_ (model.py:210)
                const float wn = S[P.train_off + e] - (dec ? lr_dec : lr_rec) * g;
                mg_st(S + P.train_off + e, wn);
                if (P.tr_aux[tens] >= 0) {
                    const int cols = P.tr_cols[tens], rel = e - (P.tr_off[tens] - P.train_off);
                    const int r = rel / cols, cc = rel - r * cols;
                    mg_st(A.aux + P.tr_aux[tens] + (size_t)cc * P.tr_auxld[tens] + P.tr_auxcol[tens] + r, wn);
                }
            }
        }
        if (t == 0 && mg_ld(SC + VJF_SC_TRI_CLEAN) == 0.f) {
            // one-time clearing of the halves the inverse loops never write (block-lower part of w_chol, block-upper part of
            // w_pchol): every reader of the dense w_chol of step 0 has signalled its late slab
            float* Wc = S + P.off[VJF_SLOT_W_CHOL];
            float* Lm = S + P.off[VJF_SLOT_W_PCHOL];
            const int n = P.n;
            for (int e = sw * NT + tid; e < n * n; e += A.n_sgd * NT) {
                const int i = e / n, j = e - i * n;
                if ((i >> 5) < (j >> 5)) mg_st(Lm + e, 0.f);
                if ((i >> 5) > (j >> 5)) mg_st(Wc + e, 0.f);
            }
        }
        if (sw == 0 && tid == 0) {                                             // ---- scalars: loss, likelihood log-variance
            if (!ok_r) l_recon = 0.f;
            if (!ok_d) l_dyn = 0.f;
            if (!ok_h) ent = 0.f;
            const float loss = l_recon - ent + l_dyn;
            if (A.loss) { float* l4 = A.loss + 4 * (size_t)t; l4[0] = loss; l4[1] = -l_recon; l4[2] = -l_dyn; l4[3] = ent; }
            const unsigned st = (ok_r ? 0u : VJF_STATUS_NONFINITE_RECON) | (ok_d ? 0u : VJF_STATUS_NONFINITE_DYN) |
                                (ok_h ? 0u : VJF_STATUS_NONFINITE_ENT);
            if (st) vjf_status_or(SC + VJF_SC_STATUS, st);
            if (P.lik == VJF_LIK_GAUSSIAN) {
                const float sse_y = s_sc[RS_SSEY];
                float rho = S[P.off[VJF_SLOT_LIK_LOGVAR]];
                if (grad_ok) {
                    float g = 0.5f * ((float)P.dy - expf(-rho) * sse_y * invB);
                    g = fminf(fmaxf(g, -1.f), 1.f);
                    rho -= SC[VJF_SC_LR_LIK] * g;
                }
                const float mse = sse_y / (Bf * (float)P.dy);
                const float acc = fminf(SC[VJF_SC_N_LIK], 1000.f), tot = acc + Bf;
                rho = logf((acc / tot) * expf(rho) + (Bf / tot) * mse);
                mg_st(SC + VJF_SC_N_LIK, tot);
                mg_st(S + P.off[VJF_SLOT_LIK_LOGVAR], rho);
            }
        }
        __syncthreads();
        vjf_wg_signal_wt(A.cnt + MG_C_SGD, tid);
    }
    // (the launch's last act on the triangle flag: set once every SGD workgroup has cleared its share -- they all have signalled
    //  step 0 by then; the kernel boundary makes it visible to the next launch)
    if (sw == 0 && tid == 0 && SC[VJF_SC_TRI_CLEAN] == 0.f) {
        bool there = false;
        for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
            if ((int)(__hip_atomic_load(A.cnt + MG_C_SGD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)A.n_sgd) >= 0) { there = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (there) mg_st(SC + VJF_SC_TRI_CLEAN, 1.f);
    }
}

// ------------------------------------------------------------------------------------------------ the kernel
__global__ __launch_bounds__(VJF_MG_THREADS) void vjf_mega_kernel(VjfPlan P, VjfMegaArgs A, VjfCholArgs C, VjfPostArgs Q) {
    static_assert(VJF_CHOL_THREADS == VJF_MG_THREADS && VJF_POST_THREADS == VJF_MG_THREADS, "one workgroup size for every role");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_dead;
    if (threadIdx.x == 0) s_dead = 0;
    __syncthreads();
    int b = (int)blockIdx.x;
    if (b == 0) { vjf_chol_loop<16>(P, C, lds, &s_dead); return; }
    if (b == 1) { vjf_rls_post_loop(P, Q, lds, &s_dead, 2, 0); return; }
    if (b < A.n_rls) { vjf_rls_post_loop(P, Q, lds, &s_dead, 1, b - 2); return; }
    b -= A.n_rls;
    if (b < A.n_trial) { vjf_mega_trial(P, A, lds, b); return; }
    b -= A.n_trial;
    if (b < A.n_gram) { vjf_mega_gram(P, A, lds, b); return; }
    b -= A.n_gram;
    if (b < A.n_prep) { vjf_mega_prep(P, A, lds, b); return; }
    b -= A.n_prep;
    vjf_mega_sgd(P, A, lds, b);
}
