// vjf_mega_kernel.h -- vjf_filter_seq / vjf_filter_step on a single rank as ONE launch (a grid resident as a whole) per chunk of steps.
//
// Every piece of a filtering step (vjf/model.py:179-221) is a ROLE played by workgroups of the same grid, one workgroup per
// compute unit, all resident for the whole chunk (the host launches the grid only if the occupancy query says all of it fits):
//
//   RLS roles        workgroup 0: the Cholesky loop (vjf_chol_loop), 1: the y / W loop, 2 .. 1 + 2 nbl: the inverse loops
//                    (vjf_rls_post_loop) -- module.py:94-102, model.py:373-377, exactly as before
//   trial role       n_trial workgroups; workgroup w owns the 32-trial tiles w, w + n_trial, ..  Per tile and step: reparametrise,
//                    RBF features, recognition network, posterior, decoder (model.py:97-122), Phi^T dx of the tile; then -- behind the
//                    RLS update of the previous step -- predictive mean / variance, loss terms (model.py:124-154), hand-derived backward
//                     and the tile's weight / bias gradients on the matrix cores (K = 32 trials).  A workgroup's sums
//                    over its tiles leave as two slabs: early [Phi^T dx | sum dx^2], late [gradients | loss sums].
//   Gram role        n_gram workgroups: Phi^T Phi (module.py:96) one step AHEAD -- its rows of Phi formed here from the posterior, in LDS, all 28 lower 32x32 tiles per
//                    workgroup on v_mfma_f32_32x32x2_f32, partial tiles to a slab, then every workgroup sums its share of the slabs
//   operand role     ceil(n / 16) workgroups: sum of the early slabs -> Phi^T dx, g = P W + Phi^T dx / v, P += Phi^T Phi / v (module.py:94-96)
//   SGD role         n_sgd workgroups: sum of the late slabs, finite guards and loss (model.py:138-154), clip + SGD (model.py:210-211),
//                    likelihood running variance (likelihood.py:28-40)
//
// Hand-offs are monotone workgroup counters in memory: producer = write-through stores, every storing wavefront drains vmcnt, the
// workgroup barrier, one relaxed agent-scope add; consumer = one lane polls about once a microsecond (bounded; VJF_POLL_SLEEP,
// vjf_plan.h), the workgroup barrier, and every handed-off byte is read with an sc1 load (MI355X guide, "sc1 loads in place of the
// acquire"; VJF_HANDOFF_ACQUIRE=1 adds an agent-scope acquire behind every wait).  Nothing ever waits for work of a launch that has not been submitted: every
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
#define VJF_MG_GROWS 96              // rows of Phi formed per pass of the Gram role
#define VJF_MG_MAXQ 4                // 32x32 tiles of Phi^T Phi per wavefront of a Gram workgroup (28 lower tiles / 8)
#define VJF_MG_RING 32               // loss sums of a late slab: a ring over the steps (a launch without parameter updates has no gate
                                     // between its steps: the trial role may run this many steps ahead of the role that sums them)
#define VJF_MG_TAG_TILES 512         // most tiles a launch with a moments role has (B <= 16384)
#define RS_RESID 5                   // late slab only: sum |dx - Phi W|^2 of a workgroup's trials (warm-up: the state-noise update
                                     // without an RLS update, model.py:373-377 with the old W)

// counters: one per 64-byte line of the block -- times MG_C_SPREAD (experiment: 64 puts every counter into a 4-KB page of its own)
#ifndef MG_C_SPREAD
#define MG_C_SPREAD 1
#endif
enum {
    MG_C_FWD = 16 * MG_C_SPREAD,       // trial workgroups whose early slab of step t is in memory           target (t + 1) n_trial
    MG_C_K1 = 32 * MG_C_SPREAD,        // trial workgroups that have read W, w_chol, sigma of step t - 1      target (t + 1) n_trial
    MG_C_BWD = 48 * MG_C_SPREAD,       // trial workgroups whose late slab of step t is in memory            target (t + 1) n_trial
    MG_C_GRAM = 64 * MG_C_SPREAD,      // Gram workgroups whose partial tiles of event e are in memory       target (e + 1) n_gram
    MG_C_STAT = 80 * MG_C_SPREAD,      // Gram workgroups whose share of Phi^T Phi of event e is reduced      target (e + 1) n_gram
    MG_C_PREP = 96 * MG_C_SPREAD,      // operand workgroups done with step t                                target (t + 1) n_prep
    MG_C_SGD = 112 * MG_C_SPREAD,      // SGD workgroups done with step t                                    target (t + 1) n_sgd
    MG_C_PDONE = 128 * MG_C_SPREAD,    // RLS workgroups (y / W loop + inverse loops) done with step t       target (t + 1) (2 nbl + 1)
    MG_C_STARTED = 144 * MG_C_SPREAD,
    MG_C_REDO_B = 0 * MG_C_SPREAD,     // trial workgroups whose REPLAYED late slab is in memory             target (replays so far) n_trial
    MG_C_REDO_S = 176 * MG_C_SPREAD,   // SGD workgroups done with a replayed step                         target (replays so far) n_sgd
    MG_C_IMG = 208 * MG_C_SPREAD,      // SGD workgroups whose share of the parameter image is in memory (start of the launch)  target n_sgd
    MG_C_SIGW = 224 * MG_C_SPREAD,     // 8 bytes: {epoch, sigma} from the y / W loop to the Cholesky loop of the next step
    MG_C_XT = 240 * MG_C_SPREAD,       // inverse workgroups whose share of xt = w_chol^T is in memory (start of the launch)              target 2 nbl
                                       // [+ 1]: launches without an RLS update: trial workgroups that met a nonzero BELOW the diagonal of w_chol
    MG_C_MASK = 192 * MG_C_SPREAD,     // (step + 1) << 8 | non-finite loss components (1 recon, 2 dynamics, 4 entropy) of the last step that had one
    MG_C_COLFLAGS = 160 * MG_C_SPREAD, // [0 .. VJF_CHOL_MAXBLK]: column flags of the Cholesky loop; [VJF_CHOL_MAXBLK + 2]: its "operands loaded" word
    MG_C_ALIVE = 256 * MG_C_SPREAD,    // workgroups of the grid that have started (all of them: the launch goes on; else it ends untouched)  target gridDim.x
    // per-TILE step tags of the launches without an RLS update that have a moments role (vjf_mega_moments): one producer, one consumer each
    MG_C_ARR = 272 * MG_C_SPREAD,      // [step % VJF_MG_RING]: trial workgroups whose loss sums of that step are in memory (a launch without
                                       // parameter updates: the LAST arriver sums them; it puts the word back to 0)
    MG_C_TAG_POST = (272 + 32) * MG_C_SPREAD, // [tile]: t + 1 once the posterior of step t of the tile is in memory (trial role -> moments role)
    MG_C_TAG_MOM = MG_C_TAG_POST + VJF_MG_TAG_TILES,   // [tile]: t + 1 once the predictive moments of step t of the tile are (moments role -> trial role)
    MG_C_WORDS = MG_C_TAG_MOM + VJF_MG_TAG_TILES
};

// The last act of every workgroup of a one-launch grid: if a wait of the launch has been given up (by this workgroup or another),
// say so where the host sees it without a synchronisation (vjf_plan.h, VJF_MIRROR_SLOT).
__device__ __forceinline__ void mg_tell_host(const float* status, unsigned* host_word) {
    if (threadIdx.x == 0 && host_word && vjf_abort_seen(status)) __hip_atomic_store(host_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The first act of every workgroup of a one-launch grid: count itself in and wait until the WHOLE grid has -- every wait of the
// launch is for a workgroup of the same grid.  Within a process the launches of this route are chained (vjf_abi.hip), so a grid
// never shares the device with another one of its kind; a grid of ANOTHER process can hold compute units (each of these
// workgroups wants a whole unit's LDS), and then neither would ever be placed as a whole.  The bound is short (2^17 polls, about a
// quarter of a second: a grid starts within a microsecond on a free device): the launch ends before any role has written to the
// state, VJF_STATUS_NOT_RESIDENT says so, and the context takes the per-step kernels from its next call on.
__device__ __forceinline__ bool mg_grid_resident(unsigned* cnt, float* status, int extra) {
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt + MG_C_ALIVE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool there = false;
        for (unsigned spins = 0; spins < (1u << 17); ++spins) {
            if (__hip_atomic_load(cnt + MG_C_ALIVE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x + (unsigned)extra) { there = true; break; }
            if ((spins & 255u) == 255u && vjf_abort_seen(status)) break;
            __builtin_amdgcn_s_sleep(VJF_POLL_SLEEP);
        }
        if (!there && !vjf_abort_seen(status)) vjf_status_or(status, VJF_STATUS_RLS_FAILED | VJF_STATUS_NOT_RESIDENT);
        vjf_s_abort_word = there ? 0 : 1;
    }
    __syncthreads();
    return vjf_s_abort_word == 0;
}

struct VjfMegaArgs {
    int T, B, ntiles;
    int n_rls, n_trial, n_gram, n_prep, n_sgd;        // grid = their sum
    unsigned* host_word;                              // this context's word of the pinned host page (null: none), see mg_tell_host
    int alive_extra;                                  // test hook (VJF_DEBUG_ABSENT=1): workgroups the residency count waits for beyond the grid's own
    int n_mom;                                        // launches without an RLS update: workgroups of the moments role (0: the trial role forms the moments itself)
    float* mom;                                       // [tile][step parity][(2 dz + 1) x 32]: pt.mean | Phi W | pt.logvar of the tile's trials, moments role -> trial role
    int n_sgd_live;                                   // SGD workgroups that stay for the steps (all of them; ONE when flags has no VJF_FLAG_SGD:
                                                      // the others only help to build the parameter image at the start of the launch)
    const float* y; const float* u; const float* eps; const float* mu0; const float* lv0;
    float* mu; float* lv; float* loss;
    float* state; float* aux;
    const float* img;                                 // the optimised parameters as the trial role's LDS holds them (vjf_mega_trial_lds: theta region)
    float* pmsave;                                    // (B, dz + 1): pt.mean | pt.logvar of every trial at its last step (for a replayed backward pass)
    float* slab_early; float* slab_late; float* gslab;
    float* red0; float* red1;                         // reduce buffers of even / odd steps ([G | FDX | sums], as the RLS loops read them)
    float* gbuf;                                      // g (n, dz)
    const float* xt;                                  // (n, n) row-major L^-1 = w_chol^T: the inverse loops keep it beside w_chol (vjf_post_kernel.h)
    unsigned* cnt;
    unsigned* cnt_next;                               // the other counter block: zeroed by this launch for the next one
    unsigned flags;
    int early_len, late_len;                          // floats per trial workgroup
    int lds_floats;                                   // dynamic LDS of the launch (floats): decides whether the parameters are staged in it
    int gram_rows;                                    // rows of Phi per Gram workgroup (a multiple of 2)
    int slab_len;                                     // floats of gradient per late slab (vjf_mega_slab_layout)
    const int* sl_pidx;                               // per slab float: the parameter it is the gradient of (index in the train region; -1: padding)
    const int* sl_cidx;                               // per slab float: that parameter's copy the trial role reads (LDS image, or the transposed aux copy; -1: none)
    const int* sl_grp;                                // per slab QUAD: 0 recognition, 1 decoder group (learning rate, freeze flag)
    unsigned long long* stamps;                       // diagnostic (null in normal runs): s_memrealtime of workgroup 0 of each role, 32 per step
};

// ---- LDS of the trial role (floats); the host uses the same function to size the launch
struct VjfMegaTrialLds {
    int cen, iw, in, xu, phi, act, dd, mu, lv, xt, e2, pm, dmu, dlv, dx, xn, py, dpy, one, zero, sc, red, plv, wg, part, total;
    int nd;
    // the optimised parameters, staged once per step when they fit (theta = 1): matrices in their torch layout [rows][ld], ld = the
    // row length rounded up to 2 (mod 4) -- the rows an MFMA operand read walks then fall on distinct banks
    int th0, th_len;                                  // first float / length (a multiple of 4) of the region
    int theta, th_w[VJF_MAX_HIDDEN], th_ldw[VJF_MAX_HIDDEN], th_head, th_ldh, th_dec, th_ldd, th_b[VJF_MAX_HIDDEN], th_bl, th_bd;
};
__host__ __device__ inline int vjf_mega_ld(int K) { return ((K + 1) & ~3) + 2; }
// LAYERS = false (device code): the per-layer arrays are left alone -- filling them in a loop with a run-time index would put the
// whole struct into scratch memory; the kernels get a layer's entries from mg_theta_layer
template <bool LAYERS = true>
__host__ __device__ inline VjfMegaTrialLds vjf_mega_trial_lds(const VjfPlan& P, int lds_limit_floats = 0) {
    VjfMegaTrialLds l;
    const int LD = VJF_MG_LD;
    int o = 0;
    auto take = [&](int nfl) { const int at = o; o += (nfl + 3) & ~3; return at; };
    l.cen = take(((P.n + 3) & ~3) * P.dxu); l.iw = take((P.n + 3) & ~3);   // centroids transposed [dxu][n rounded to 4]
    l.in = take(P.din * LD); l.xu = take(P.dxu * LD); l.phi = take(P.n * LD); l.act = take(P.hsum * LD);
    const bool compact = P.dy >= P.hmax;              // the first delta buffer lives in the (by then dead) decoder-mean rows
    l.nd = compact ? (P.L > 1 ? 1 : 0) : (P.L > 1 ? 2 : 1);
    l.dd = take(l.nd * P.hmax * LD);
    l.mu = take(P.dz * LD); l.lv = take(P.dz * LD); l.xt = take(P.dz * LD); l.e2 = take(P.dz * LD); l.pm = take(P.dz * LD);
    l.dmu = take(P.dz * LD); l.dlv = take(P.dz * LD); l.dx = take(P.dz * LD); l.xn = take(P.dxu * LD);
    l.py = take(P.dy * LD); l.dpy = take(P.dy * LD);
    l.one = take(LD); l.zero = take(LD);
    l.sc = take(VJF_MG_TR * RS_N); l.red = take(VJF_MG_WAVES * VJF_MG_TR); l.plv = take(VJF_MG_TR); l.wg = take(16);
    // partial tiles of the K-split products (heads, pt.mean), VJF_MG_WAVES x 16 rows: in the delta buffers (free until the backward
    // pass) or the dpy rows (free until the losses) when those are large enough, else rows of their own
    const int alias_rows = compact ? P.dy : l.nd * P.hmax;
    l.part = alias_rows >= VJF_MG_WAVES * 16 ? (compact ? l.dpy : l.dd) : take(VJF_MG_WAVES * 16 * LD);
    l.total = o;
    {
        int prev = P.din;
        l.th0 = o;
        if (LAYERS) for (int k = 0; k < VJF_MAX_HIDDEN; ++k) { l.th_w[k] = l.th_ldw[k] = l.th_b[k] = 0; }
        for (int k = 0; k < P.L; ++k) {
            const int ldw = vjf_mega_ld(prev), w = take(P.h[k] * ldw), b = take(P.h[k]);
            if (LAYERS) { l.th_ldw[k] = ldw; l.th_w[k] = w; l.th_b[k] = b; }
            prev = P.h[k];
        }
        l.th_ldh = vjf_mega_ld(prev); l.th_head = take(2 * P.dz * l.th_ldh); l.th_bl = take(P.dz);
        l.th_ldd = vjf_mega_ld(P.dz); l.th_dec = take(P.dy * l.th_ldd); l.th_bd = take(P.dy);
        l.th_len = o - l.th0;
        l.theta = (lds_limit_floats > 0 && o <= lds_limit_floats) ? 1 : 0;
        if (l.theta) l.total = o;
    }
    return l;
}
// the mu / lv / xt / e2 / pm / dmu / dlv / dx rows must be adjacent in this order (the heads write 2 dz rows at mu, the ahead
// features park xs' in the 3 dz rows at dmu): take() pads to 4 floats, so dz * LD must be a multiple of 4 or the code below
// addresses through the struct's offsets only -- it does (no pointer arithmetic across fields except mu -> lv and dmu -> dlv,
// which are handled explicitly).

// Late slab of a trial workgroup: its tiles' gradients, one block per weight tensor, each block TRANSPOSED -- row j = the input
// (activation) index, then the bias row; columns = the output units, padded to a multiple of 4 -- so that the four accumulator
// registers of a lane (four consecutive output units of one input) leave as ONE 16-byte write-through store.  Blocks in the order
// the backward pass produces them: decoder, mean head, log-variance head, recognition layers L-1 .. 0.
struct VjfMegaSlab { int off[VJF_MAX_HIDDEN + 3], ldm[VJF_MAX_HIDDEN + 3], rows[VJF_MAX_HIDDEN + 3], len; };
__host__ __device__ inline VjfMegaSlab vjf_mega_slab_layout(const VjfPlan& P) {
    VjfMegaSlab L;
    int o = 0, k = 0;
    auto blk = [&](int M, int rows) { L.off[k] = o; L.ldm[k] = (M + 3) & ~3; L.rows[k] = rows; o += rows * L.ldm[k]; ++k; };
    const int hL = P.h[P.L - 1];
    blk(P.dy, P.dz + 1);                               // 0: decoder  (dy, dz) + bias
    blk(P.dz, hL);                                     // 1: mean head (dz, hL), no bias
    blk(P.dz, hL + 1);                                 // 2: log-variance head + bias
    for (int l = P.L - 1; l >= 0; --l) blk(P.h[l], (l > 0 ? P.h[l - 1] : P.din) + 1);   // 3 + (L-1-l): layer l + bias
    for (; k < VJF_MAX_HIDDEN + 3; ++k) { L.off[k] = o; L.ldm[k] = 4; L.rows[k] = 0; }
    L.len = o;
    return L;
}

// One entry of the layouts above for a layer / block index that is only known at run time, recomputed from the plan by a short
// scalar loop: indexing the structs' arrays with it would put them into scratch memory (the kernel then needs a scratch buffer
// at launch and pays memory round trips for what is a handful of integer additions).
__device__ __forceinline__ void mg_theta_layer(const VjfPlan& P, int th0, int l, int& w, int& ldw, int& b) {
    int o = th0, prev = P.din;
    w = ldw = b = 0;
    for (int k = 0; k <= l && k < P.L; ++k) {
        ldw = vjf_mega_ld(prev);
        w = o; o += (P.h[k] * ldw + 3) & ~3;
        b = o; o += (P.h[k] + 3) & ~3;
        prev = P.h[k];
    }
}
__device__ __forceinline__ void mg_slab_block(const VjfPlan& P, int blk, int& off, int& ldm, int& rows) {
    const int hL = P.h[P.L - 1];
    int o = 0;
    auto step = [&](int M, int r, bool take_it) { if (take_it) { off = o; ldm = (M + 3) & ~3; rows = r; } o += r * ((M + 3) & ~3); };
    off = 0; ldm = 4; rows = 0;
    step(P.dy, P.dz + 1, blk == 0);
    step(P.dz, hL, blk == 1);
    step(P.dz, hL + 1, blk == 2);
    for (int l = P.L - 1, k = 3; l >= 0; --l, ++k) step(P.h[l], (l > 0 ? P.h[l - 1] : P.din) + 1, blk == k);
}

static inline size_t vjf_mega_gram_lds_floats(const VjfPlan& P) {      // rows of Phi | tile table | centroids^T | -1/(2 w^2) | xs rows
    const size_t npad = (size_t)((P.n + 3) & ~3);
    return (size_t)VJF_MG_GROWS * P.ldE + 64 + npad * P.dxu + npad + (size_t)VJF_MG_GROWS * P.dxu + 16;
}
static inline size_t vjf_mega_prep_lds_floats(const VjfPlan& P) {
    return (size_t)16 * VJF_PREPG_LDP(P.n) + (size_t)P.n * 17 + (size_t)VJF_MG_WAVES * 16 * 17 + 16 * 17 + 64;
}

__device__ __forceinline__ float mg_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void mg_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte sc1 loads (buffer_load_dwordx4 ... sc1): what another workgroup stored write-through, read past this CU's vector L1.
// The descriptor's base must be workgroup-uniform (it lives in scalar registers); the per-lane part is the 32-bit float index.
typedef unsigned mg_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mg_rsrc(const float* uniform_base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(uniform_base), 0, 0x7fffffff, 0x00020000);
}
// the same as a PLAIN load (through this CU's L1): only for bytes that are constants of the launch by the time they are first read
__device__ __forceinline__ float4 mg_ld4_plain(__amdgpu_buffer_rsrc_t r, int float_index) {
    const mg_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, float_index * 4, 0, 0);
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ float4 mg_ld4(__amdgpu_buffer_rsrc_t r, int float_index) {
    const mg_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, float_index * 4, 0, 16);               // aux 16 = sc1
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}

// tanh for the recognition layers (recognition.py:31-42), branch-free: the library's tanhf is ~70 VALU instructions with both of its
// branches taken in every wavefront, eight calls per lane and layer -- 1.2 us of a 32-trial tile's 4.5-us layer on a part whose SIMDs
// run VALU and MFMA instructions one after the other.  |x| < 0.55: x + x^3 P(x^2), the odd series through
// x^15; else 1 - 2 / (e^{2|x|} + 1) on the hardware exp2 and reciprocal (within 1 ulp each): <= ~2-3 ulp of the result, <= 1.1e-7
// absolute (emulated against fp64 over [-12, 12]: 1.8 ulp with exact exp2 / division); saturates to +-1 beyond |x| ~ 9, NaN stays NaN.
__device__ __forceinline__ float mg_tanh(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);          // e^{2|x|}
    const float big = 1.f - 2.f * __builtin_amdgcn_rcpf(t + 1.f);
    const float z = x * x;
    float p = -1.4558343870513183e-3f;                                         // -929569/638512875
    p = fmaf(p, z, 3.5921280365724810e-3f);                                    // 21844/6081075
    p = fmaf(p, z, -8.8632355299021966e-3f);                                   // -1382/155925
    p = fmaf(p, z, 2.1869488536155203e-2f);                                    // 62/2835
    p = fmaf(p, z, -5.3968253968253971e-2f);                                   // -17/315
    p = fmaf(p, z, 1.3333333333333333e-1f);                                    // 2/15
    p = fmaf(p, z, -3.3333333333333331e-1f);                                   // -1/3
    const float small = fmaf(p * z, x, x);
    return ax < 0.55f ? small : copysignf(big, x);
}

// acc_g(row = 4*(lane>>4)+r, col = lane&15) += sum_{kb <= k < ke} Ag[k*lda + m0 + row] * Xs[k*LD + 16 g + col]   (g = 0, 1)
// Rows m0 + i >= M contribute 0 (their A operand is read from a clamped address and masked at use).  kb is a multiple of 4.  The A operands come straight from L2 (k-major matrices: row k contiguous over the output features), 16 k-steps per batch,
// two batches in flight: while one batch's 32 MFMAs issue the next one's loads are on their way (and the SIMD's other wavefront
// fills what latency is left).  The loads are sc1 (they bypass this CU's vector L1): these matrices are rewritten every step by
// other roles, and the waits in front of them do not acquire.
// one batch of mg_mma2 (below) on its own: the 16 A-operand loads of k-steps s0 .. s0 + 15, and their MFMAs -- for a product
// whose loads are issued long before its turn (pt.mean: in front of the variance tiles)
__device__ __forceinline__ void mg_mma2_ld16(float (&a)[16], const float* __restrict__ Ag, int lda, int M, int m0, int kb, int ke, int s0, int lane) {
    const int i = lane & 15, kk = lane >> 4;
    const bool rv = (m0 + i) < M;
    const unsigned row = rv ? (unsigned)(m0 + i) : 0u;
    const int klast = ke - 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int k = min(kb + 4 * (s0 + q) + kk, klast); a[q] = mg_ld(Ag + row + (unsigned)k * (unsigned)lda); }
    // (rows beyond M are masked where the value is USED: a select on a load's destination right behind the load makes the compiler
    //  wait for the load there, and the batch would no longer be in flight beside the previous batch's MFMAs)
}
__device__ __forceinline__ void mg_mma2_mm16(vjf_f32x4& acc0, vjf_f32x4& acc1, const float (&a)[16], const float* Xs, int M, int m0, int kb, int ke, int s0, int lane) {
    constexpr int LD = VJF_MG_LD;
    const int i = lane & 15, kk = lane >> 4;
    const bool rv = (m0 + i) < M;
    const float* xp = Xs + i;
    const int nst = (ke - kb + 3) >> 2, klast = ke - 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (s0 + q < nst) {                            // (uniform)
            const int k = kb + 4 * (s0 + q) + kk;
            const int kc = min(k, klast);
            const float av = (rv && k < ke) ? a[q] : 0.f;
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD + 16], acc1, 0, 0, 0);
        }
    }
}

template <bool DEEP = false>
__device__ __forceinline__ void mg_mma2(vjf_f32x4& acc0, vjf_f32x4& acc1, const float* __restrict__ Ag, int lda, int M, int m0,
                                        const float* Xs, int kb, int ke, int lane) {
    constexpr int LD = VJF_MG_LD;
    const int i = lane & 15, kk = lane >> 4;
    const bool rv = (m0 + i) < M;
    const unsigned row = rv ? (unsigned)(m0 + i) : 0u;
    const unsigned ulda = (unsigned)lda;
    const float* xp = Xs + i;
    const int nst = (ke - kb + 3) >> 2;                // k-steps
    const int klast = ke - 1;
    auto ld16 = [&](float (&a)[16], int s0) {          // steps s0 .. s0 + 15: clamped rows, masked at use
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int k = min(kb + 4 * (s0 + q) + kk, klast); a[q] = mg_ld(Ag + row + (unsigned)k * ulda); }
    };
    auto mm16 = [&](const float (&a)[16], int s0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (s0 + q < nst) {                        // (uniform)
                const int k = kb + 4 * (s0 + q) + kk;
                const int kc = min(k, klast);
                const float av = (rv && k < ke) ? a[q] : 0.f;
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD + 16], acc1, 0, 0, 0);
            }
        }
    };
    if (nst <= 0) return;
    if (DEEP) {
        // up to 64 k-steps (K <= 256): every load of the tile is issued before the first MFMA
        float a0[16], a1[16], a2[16], a3[16];
        ld16(a0, 0);
        if (nst > 16) ld16(a1, 16);
        if (nst > 32) ld16(a2, 32);
        if (nst > 48) ld16(a3, 48);
        mm16(a0, 0);
        if (nst > 16) mm16(a1, 16);
        if (nst > 32) mm16(a2, 32);
        if (nst > 48) mm16(a3, 48);
        for (int s0 = 64; s0 < nst; s0 += 16) { ld16(a0, s0); mm16(a0, s0); }
        return;
    }
    float a0[16], a1[16];
    ld16(a0, 0);
    if (nst > 16) ld16(a1, 16);
    for (int s0 = 0; s0 < nst; s0 += 32) {
        mm16(a0, s0);
        if (s0 + 32 < nst) ld16(a0, s0 + 32);
        if (s0 + 16 < nst) {
            mm16(a1, s0 + 16);
            if (s0 + 48 < nst) ld16(a1, s0 + 48);
        }
    }
}

// The predictive-variance product from the ROW-major inverse factor:  acc_g(row, col) += sum_{k < ke} Xt[(m0 + row) * n + k] * Xs[k * LD + 16 g + col].
// A lane takes 16 bytes along k: lane (i, kk) loads Xt[m0 + i][16 t + 4 kk .. + 3] and feeds component c to the MFMA of step
// (t, c), whose k index is 16 t + 4 kk + c -- any order of the k indices is a valid product as long as both operands use it (the B
// operand reads that row of Xs).  One 16-byte sc1 load per lane and 16 k instead of four 4-byte ones; batches of four loads (64 k),
// two batches in flight.
__device__ __forceinline__ void mg_mma2x(vjf_f32x4& acc0, vjf_f32x4& acc1, __amdgpu_buffer_rsrc_t rx, int n, int M, int m0, const float* Xs, int ke, int lane) {
    constexpr int LD = VJF_MG_LD;
    const int i = lane & 15, kk = lane >> 4;
    const bool rv = (m0 + i) < M;
    const int rowoff = (rv ? m0 + i : 0) * n + 4 * kk;
    const float* xp = Xs + i;
    const int nt = (ke + 15) >> 4;                    // blocks of 16 k
    auto ld4 = [&](float4 (&a)[4], int t0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int kq = 16 * (t0 + q) + 4 * kk; a[q] = mg_ld4(rx, rowoff + (kq + 3 < n ? 16 * (t0 + q) : 0)); }
    };
    auto mm4 = [&](const float4 (&a)[4], int t0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (t0 + q < nt) {                         // (uniform)
                const int k0 = 16 * (t0 + q) + 4 * kk;
                const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = k0 + c, kc = min(k, ke - 1);
                    const float v = (rv && k < ke) ? av[c] : 0.f;      // (masked at use: see mg_mma2)
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v, xp[kc * LD], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v, xp[kc * LD + 16], acc1, 0, 0, 0);
                }
            }
        }
    };
    if (nt <= 0) return;
    float4 a0[4], a1[4];
    ld4(a0, 0);
    if (nt > 4) ld4(a1, 4);
    for (int t0 = 0; t0 < nt; t0 += 8) {
        mm4(a0, t0);
        if (t0 + 8 < nt) ld4(a0, t0 + 8);
        if (t0 + 4 < nt) {
            mm4(a1, t0 + 4);
            if (t0 + 12 < nt) ld4(a1, t0 + 12);
        }
    }
}

// The predictive variance's share of one wavefront: sum over its (at most two) 16-row tiles of L^-1 of the squares of
// (rows j0 .. j0 + 15 of Xt) . Xs, into v2a / v2b (the two 16-trial column groups).  The tiles' k-batches (4 blocks of 16 k each,
// operands as in mg_mma2x) form ONE stream with three batches in flight: the second tile's first loads are out while the first
// tile still multiplies (one call of mg_mma2x per tile drained the pipeline in between: an exposed L2 round trip per tile).
// Same products in the same order as two calls of mg_mma2x followed by the sums of squares: the same bits.
// j0B < 0: no second tile; j0A < 0: none at all.
__device__ __forceinline__ void mg_var2(float& v2a, float& v2b, __amdgpu_buffer_rsrc_t rx, int n, int j0A, int KA, int j0B, int KB,
                                        const float* Xs, int lane) {
    constexpr int LD = VJF_MG_LD;
    if (j0A < 0) return;
    const int i = lane & 15, kk = lane >> 4;
    const bool rvA = (j0A + i) < n, rvB = j0B >= 0 && (j0B + i) < n;
    const int offA = (rvA ? j0A + i : 0) * n + 4 * kk, offB = (rvB ? j0B + i : 0) * n + 4 * kk;
    const float* xp = Xs + i;
    const int ntA = (KA + 15) >> 4, ntB = j0B >= 0 ? (KB + 15) >> 4 : 0;
    const int SA = (ntA + 3) >> 2, SB = (ntB + 3) >> 2, S = SA + SB;
    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    auto fold = [&]() {
        v2a = fmaf(acc0[0], acc0[0], fmaf(acc0[1], acc0[1], fmaf(acc0[2], acc0[2], fmaf(acc0[3], acc0[3], v2a))));
        v2b = fmaf(acc1[0], acc1[0], fmaf(acc1[1], acc1[1], fmaf(acc1[2], acc1[2], fmaf(acc1[3], acc1[3], v2b))));
    };
    auto ldb = [&](float4 (&a)[4], int sb) {
        const bool inB = sb >= SA;                                            // (uniform)
        const int t0 = 4 * (inB ? sb - SA : sb), off = inB ? offB : offA;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int kq = 16 * (t0 + q) + 4 * kk; a[q] = mg_ld4(rx, off + (kq + 3 < n ? 16 * (t0 + q) : 0)); }
    };
    auto mmb = [&](const float4 (&a)[4], int sb) {
        const bool inB = sb >= SA;
        const int t0 = 4 * (inB ? sb - SA : sb), nt = inB ? ntB : ntA, ke = inB ? KB : KA;
        const bool rv = inB ? rvB : rvA;
        if (sb == SA && SA > 0) {                                             // the first batch of the second tile
            fold();
            acc0 = vjf_f32x4{0.f, 0.f, 0.f, 0.f}; acc1 = vjf_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (t0 + q < nt) {                                                // (uniform)
                const int k0 = 16 * (t0 + q) + 4 * kk;
                const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = k0 + c, kc = min(k, ke - 1);
                    const float v = (rv && k < ke) ? av[c] : 0.f;             // (masked at use: see mg_mma2)
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(v, xp[kc * LD], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(v, xp[kc * LD + 16], acc1, 0, 0, 0);
                }
            }
        }
    };
    float4 a0[4], a1[4], a2[4];
    ldb(a0, 0);
    if (S > 1) ldb(a1, 1);
    if (S > 2) ldb(a2, 2);
    for (int sb = 0; sb < S; sb += 3) {
        mmb(a0, sb);
        if (sb + 3 < S) ldb(a0, sb + 3);
        if (sb + 1 < S) { mmb(a1, sb + 1); if (sb + 4 < S) ldb(a1, sb + 4); }
        if (sb + 2 < S) { mmb(a2, sb + 2); if (sb + 5 < S) ldb(a2, sb + 5); }
    }
    fold();
}

// The same product with the A operand in LDS: Ws is a matrix [rows][ldw] as torch stores it.
//   TR = false: A[m][k] = Ws[(m0 + m) * ldw + k]       (out = W x:  forward products)
//   TR = true : A[m][k] = Ws[k * ldw + m0 + m]         (out = W^T x: backward products)
template <bool TRN>
__device__ __forceinline__ void mg_mma2_lds(vjf_f32x4& acc0, vjf_f32x4& acc1, const float* Ws, int ldw, int M, int m0, const float* Xs,
                                            int kb, int ke, int lane) {
    // The shape is the plan's -- run-time values -- and a plain loop over the k-steps (clamped k, masked A, addresses recomputed per
    // step, an LDS round trip per unrolled group) took 2.85 us for a (128, 70) layer where the same loop with the shape as compile-time
    // constants takes 1.6 (tools/lds_mma_bench.hip).  So: chunks of four k-steps whose operands are read with immediate offsets from one
    // base per chunk -- no clamp, no mask: rows beyond M are computed from row 0 and discarded by every caller, only the last, partial
    // k-step is clamped and masked -- and the next chunk's reads are issued before this chunk's MFMAs: 1.67 us.  (A chunk's steps
    // beyond the last full one read LDS behind the operands -- inside the allocation or, past its end, zeros --; their MFMAs are skipped.)
    constexpr int LD = VJF_MG_LD, CH = 4;
    if (ke <= kb) return;                              // (uniform: an empty K slice)
    const int i = lane & 15, kk = lane >> 4;
    const int mi = (m0 + i) < M ? m0 + i : 0;
    const int nf = (ke - kb) >> 2;                     // full k-steps (kb is a multiple of 4)
    const int astep = TRN ? 4 * ldw : 4;               // floats between two k-steps of the A operand
    const float* wp = TRN ? Ws + (size_t)(kb + kk) * ldw + mi : Ws + (size_t)mi * ldw + kb + kk;
    const float* xp = Xs + i + (kb + kk) * LD;
    float a0[CH], p0[CH], q0[CH], a1[CH], p1[CH], q1[CH];
    auto ld = [&](float (&a)[CH], float (&b0)[CH], float (&b1)[CH], int s0) {
        const float* w = wp + s0 * astep; const float* x = xp + 4 * s0 * LD;
#pragma unroll
        for (int q = 0; q < CH; ++q) { a[q] = w[q * astep]; b0[q] = x[4 * q * LD]; b1[q] = x[4 * q * LD + 16]; }
    };
    auto mm = [&](const float (&a)[CH], const float (&b0)[CH], const float (&b1)[CH], int s0) {
#pragma unroll
        for (int q = 0; q < CH; ++q)
            if (s0 + q < nf) {                         // (uniform)
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b0[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b1[q], acc1, 0, 0, 0);
            }
    };
    if (nf > 0) ld(a0, p0, q0, 0);
    for (int s0 = 0; s0 < nf; s0 += 2 * CH) {
        if (s0 + CH < nf) ld(a1, p1, q1, s0 + CH);
        mm(a0, p0, q0, s0);
        if (s0 + 2 * CH < nf) ld(a0, p0, q0, s0 + 2 * CH);
        if (s0 + CH < nf) mm(a1, p1, q1, s0 + CH);
    }
    if ((ke - kb) & 3) {                               // the partial step: clamped row of X, masked A
        const int k = kb + 4 * nf + kk, kc = min(k, ke - 1);
        const float w = TRN ? Ws[(size_t)kc * ldw + mi] : Ws[(size_t)mi * ldw + kc];
        const float av = k < ke ? w : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Xs[kc * LD + i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Xs[kc * LD + i + 16], acc1, 0, 0, 0);
    }
}

// 16-byte write-through store (the asm store is not counted by the compiler: every hand-off drains vmcnt by hand before it signals)
__device__ __forceinline__ void mg_st4(float* p, float x, float y, float z, float w) {
    vjf_f32x4 o = {x, y, z, w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(o) : "memory");
}

// e / d for e d < 2^32 without the ~30-instruction integer division: one v_mul_hi_u32 with m = ceil(2^32 / d) (d >= 2)
__device__ __forceinline__ unsigned mg_magic(unsigned d) { return d < 2 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }
__device__ __forceinline__ int mg_div(int e, unsigned m) { return m ? (int)__umulhi((unsigned)e, m) : e; }

// L2 warm-up.  Parameters that another role has just rewritten (write-through) sit in memory, and the trial workgroups of an
// XCD all walk them in the same order at the same time: every batch of operand loads would be a miss that all of them wait
// for together.  Instead each workgroup first touches one sixteenth of the range (16-byte loads, all in flight, nothing kept):
// between them the 16 trial workgroups that usually share an XCD bring all of it into that XCD's L2 in ONE round trip.
// Which workgroups share an XCD is a placement guess (blockIdx round-robin); a wrong guess costs speed, never correctness.
__device__ __forceinline__ void mg_warm(const float* base, int nfloats, int wg, int tid) {
    const int nq = nfloats >> 2, per = (nq + 15) >> 4, q0 = ((wg >> 3) & 15) * per;
    const __amdgpu_buffer_rsrc_t rb = mg_rsrc(base);
    for (int q = q0 + tid; q < min(nq, q0 + per); q += VJF_MG_THREADS) {
        const float4 v = mg_ld4(rb, q * 4);
        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    }
}

// the same with the loads left in flight (two per thread; a longer range finishes the blocking way): the caller goes on issuing
// its own loads and retires these behind them
__device__ __forceinline__ void mg_warm_issue(const float* base, int nfloats, int wg, int tid, float4 (&r)[2]) {
    const int nq = nfloats >> 2, per = (nq + 15) >> 4, q0 = ((wg >> 3) & 15) * per, q1 = min(nq, q0 + per);
    const __amdgpu_buffer_rsrc_t rb = mg_rsrc(base);
    r[0] = r[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q0 + tid < q1) r[0] = mg_ld4(rb, (q0 + tid) * 4);
    if (q0 + tid + VJF_MG_THREADS < q1) r[1] = mg_ld4(rb, (q0 + tid + VJF_MG_THREADS) * 4);
    for (int q = q0 + tid + 2 * VJF_MG_THREADS; q < q1; q += VJF_MG_THREADS) {
        const float4 v = mg_ld4(rb, q * 4);
        asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    }
}
__device__ __forceinline__ void mg_warm_retire(const float4 (&r)[2]) {
    asm volatile("" ::"v"(r[0].x), "v"(r[0].y), "v"(r[0].z), "v"(r[0].w), "v"(r[1].x), "v"(r[1].y), "v"(r[1].z), "v"(r[1].w));
}

#define VJF_MG_STAMP(i)                                                                     \
    do {                                                                                    \
        if (A.stamps && wg == 0 && tid == 0) {                                              \
            unsigned long long t_;                                                          \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
            A.stamps[(size_t)(t & 31) * 32 + (i)] = t_;                                     \
        }                                                                                   \
    } while (0)

// latest (i) / earliest (j, stored complemented) time over ALL trial workgroups
#define VJF_MG_STAMPX(i, j)                                                                 \
    do {                                                                                    \
        if (A.stamps && tid == 0) {                                                         \
            unsigned long long t_;                                                          \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
            atomicMax(A.stamps + (size_t)(t & 31) * 32 + (i), t_);                          \
            if ((j) >= 0) atomicMax(A.stamps + (size_t)(t & 31) * 32 + (j), ~t_);           \
        }                                                                                   \
    } while (0)

// per-workgroup times of the LAST step of a launch (8 words per trial workgroup behind the 32 x 32 ring)
#define VJF_MG_STAMPW(i)                                                                    \
    do {                                                                                    \
        if (A.stamps && tid == 0 && t == A.T - 1 && !replay) {                              \
            unsigned long long t_;                                                          \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");  \
            A.stamps[1024 + (size_t)wg * 8 + (i)] = t_;                                     \
        }                                                                                   \
    } while (0)

// one 16x16 tile of  G[m][j] = sum_{b<32} D[m0+m][b] * Bop[j0+j][b],  Bop = [Bact (Kin rows) | ones | 0..]  -> block `blk` of the
// late slab ([j][ldm], see vjf_mega_slab_layout): a lane's four registers are G[m .. m+3][j], one 16-byte store
__device__ __forceinline__ void mg_grad_tile(const float* D, int M, int m0, const float* Bact, int Kin, int j0, const float* s_one,
                                             const float* s_zero, float* blk, int ldm, int rows, bool first, int lane) {
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
    const int mq = m0 + 4 * (lane >> 4), j = j0 + (lane & 15);      // (rows m >= M of D are the zero row: the padding columns get 0)
    if (j < rows && mq < ldm) {
        float* p = blk + (size_t)j * ldm + mq;
        if (!first) { acc[0] += mg_ld(p); acc[1] += mg_ld(p + 1); acc[2] += mg_ld(p + 2); acc[3] += mg_ld(p + 3); }   // (a later tile of the workgroup)
        mg_st4(p, acc[0], acc[1], acc[2], acc[3]);
    }
}

typedef __attribute__((address_space(3))) const float mg_lds_cf;    // an LDS pointer by type / a global-memory pointer by type: a value that
typedef __attribute__((address_space(1))) const float mg_glb_cf;    // lives in LDS in one plan and in memory in another is read through one of
                                                                    // these on either side of a select, never through a selected generic pointer
// loss sums of step t over the trial workgroups' late slabs: fp64, 32 strided partial sums per scalar, then a fixed xor tree -> s_sc[RS_*]
// (the residual leaves as the mean square).  A workgroup-wide call (one barrier); read with sc1 loads behind the caller's wait.
__device__ __forceinline__ void mg_sum_losses(const VjfMegaArgs& A, int t, float* s_sc, int tid, float Bf, int dz, bool want_resid) {
    const int ring = 8 * (t % VJF_MG_RING);
    if (tid < 32 * 5) {
        const int sc = tid >> 5, l = tid & 31, slot = sc < RS_SDX2 ? sc : RS_RESID;
        double d = 0.0;
        if (sc < RS_SDX2 || want_resid)
            for (int w = l; w < A.n_trial; w += 32) d += (double)mg_ld(A.slab_late + (size_t)w * A.late_len + A.slab_len + ring + slot);
        d = vjf_sum32(d);
        // (the residual leaves as the mean square: its sum over 32768 x 16 elements has more digits than a float keeps)
        if (l == 0) s_sc[slot] = slot == RS_RESID ? (float)(d / ((double)Bf * (double)dz)) : (float)d;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ trial role
#define MG_PHASE()                                                        \
    do {                                                                  \
        tid = tid0;                                                       \
        asm volatile("" : "+v"(tid));                                     \
        lane = tid & 63;                                                  \
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);                  \
    } while (0)

// RLS = true: the training step (sgd + update, no warm-up) beside the RLS, Gram and operand roles -- every mode switch below is a
// compile-time constant and the code is what it was before the other flag sets existed.  RLS = false (vjf_mega_lite_kernel: trial
// and SGD roles only): warm-up, update=False, sgd=False, read from the launch's flags.
template <bool RLS>
__device__ __forceinline__ void vjf_mega_trial(const VjfPlan& P, const VjfMegaArgs& A, float* smem, const int wg) {
    constexpr int LD = VJF_MG_LD, NW = VJF_MG_WAVES, NT = VJF_MG_THREADS, TR = VJF_MG_TR;
    const int tid0 = threadIdx.x;
    const int dz = P.dz, dy = P.dy, du = P.du, n = P.n, din = P.din, dxu = P.dxu;
    const float* S = A.state;
    float* SCW = A.state + P.off[VJF_SLOT_SCALARS];
    // what the steps of this launch do (vjf/model.py:179-221: the flags of VJF.filter).  mode_rls: the RLS roles, the Gram and the
    // operand role exist; without them (warm-up, update=False) W, w_chol are constants of the launch and sigma -- if it moves at all
    // (warm-up) -- comes from the SGD role with the parameters
    const bool do_sgd = RLS || (A.flags & VJF_FLAG_SGD) != 0u, do_upd = RLS || (A.flags & VJF_FLAG_UPDATE) != 0u;
    const bool warm = !RLS && (A.flags & VJF_FLAG_WARM_UP) != 0u;
    constexpr bool mode_rls = RLS;                     // (the host sends a launch with do_upd && !warm to the full kernel only)
    const bool gated = RLS || do_sgd || do_upd;        // something another role produces changes between steps
    const bool want_resid = !RLS && do_upd && warm;
    // a moments role (vjf_mega_moments) forms the features and the predictive moments of this role's tiles a step ahead: this
    // role then neither forms features nor walks L^-1
    const bool use_mom = !RLS && A.n_mom > 0;
    const unsigned m_dy = mg_magic(dy), m_dz = mg_magic(dz), m_du = mg_magic(du > 0 ? du : 1);
    const VjfMegaTrialLds Lo = vjf_mega_trial_lds<false>(P, A.lds_floats);
    const bool tl = Lo.theta != 0;                    // the optimised parameters are staged in LDS once per step
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
    float* s_part = smem + Lo.part;                    // partial tiles of the K-split products (heads, pt.mean)
    constexpr int part_rows = VJF_MG_WAVES * 16;
    int mean_nsl = 1;
    __shared__ unsigned s_try[2];
    unsigned* cnt = A.cnt;
    const unsigned npost = (unsigned)(A.n_rls - 1);
    float* late = A.slab_late + (size_t)wg * A.late_len;
    const int ldn = (n + 3) & ~3;                      // early slab: [16 columns][ldn] Phi^T dx (transposed), then the scalars
    const size_t sy = (size_t)A.B * dy, su = (size_t)A.B * du, sz = (size_t)A.B * dz;
    int ntl = 0;
    for (int tile = wg; tile < A.ntiles; tile += A.n_trial) ++ntl;
    bool tri_launch = false;                           // a launch without an RLS update: its constant w_chol was SEEN to be upper triangular (below)

    // centroids (transposed: [input dim][centre], 16-byte rows) and -1/(2 w^2): constant for the launch (functional.py:11-22)
    const int npad = (n + 3) & ~3;
    {
        const int tid = tid0;
        const float* cen = S + P.off[VJF_SLOT_CENTROID];
        const float* lw = S + P.off[VJF_SLOT_LOGWIDTH];
        for (int e = tid; e < npad * dxu; e += NT) { const int c = e / npad, k = e - c * npad; s_cen[e] = k < n ? cen[k * dxu + c] : 0.f; }
        for (int e = tid; e < npad; e += NT) { float v = 0.f; if (e < n) { const float w = expf(lw[e]); v = -0.5f / (w * w); } s_iw[e] = v; }
        if (tid < LD) s_zero[tid] = 0.f;
    }
    __syncthreads();

    // A step whose loss has a non-finite component (model.py:138-145) is REPLAYED: the SGD role sees the sums only when every
    // workgroup's backward pass is done, publishes which components to drop and leaves the parameters alone; the trial role finds
    // that word when it fetches the parameters for the next step, runs the flagged step's forward and backward pass again --
    // same parameters, same inputs, the predictive mean / variance it saved, the dropped components' seeds exactly zero --, hands
    // over a second late slab, waits for the SGD role's (unconditional) step on it and only then starts over with the next step.
    // Nothing of this costs the usual step anything but one more word read beside rho.  Step index T is the gate alone.
    float sig_prev = 0.f, rho_prev = 0.f;
    unsigned nredo = 0;
    unsigned ring_seen = 0u;                           // a launch without parameter updates: the count of summed steps as last looked at
    for (int t = 0; t <= A.T; ++t) {
      bool replay = false, replayed = false;
      unsigned rbits = 0;
      for (;;) {
        const int ts = replay ? t - 1 : t;             // the step whose inputs this pass stages
        bool want_replay = false;
        // (the thread index is made opaque at every phase boundary: what the compiler derives from it -- dozens of per-thread LDS and
        //  memory offsets, one set per loop of the step -- is then formed in the phase that uses it instead of at the top of the step,
        //  where it was kept, and spilled to scratch memory, across the whole step)
        int tid = tid0, lane, wave;
        MG_PHASE();
        const int tc = min(ts, A.T - 1);               // (the gate pass of step T stages nothing)
        const float* y_t = A.y + (size_t)tc * sy;
        const float* u_t = A.u ? A.u + (size_t)tc * su : nullptr;
        const float* mu_s = tc ? A.mu + (size_t)(tc - 1) * sz : A.mu0;
        const float* lv_s = tc ? A.lv + (size_t)(tc - 1) * sz : A.lv0;
        const float* eps_s = A.eps + (size_t)tc * 2 * sz;
        const float* eps_t = eps_s + sz;
        float* mu_t = A.mu + (size_t)tc * sz;
        float* lv_t = A.lv + (size_t)tc * sz;
        const bool prior = (mu_s == nullptr);
        const bool m_r = !(rbits & 1u), m_d = !(rbits & 2u), m_h = !(rbits & 4u);   // components kept (all of them unless replaying)
        // early slabs alternate between two sets: the operand role may read step t's long after this workgroup has started
        // step t + 1 (it also waits for the Gram of step t); step t + 2 starts behind the RLS update of step t, which consumed them
        float* early = A.slab_early + ((size_t)(tc & 1) * A.n_trial + wg) * A.early_len;
        VJF_MG_STAMP(0);
        if (tid < 16) s_wg[tid] = 0.f;
        float sig = sig_prev, rho = rho_prev;          // (a replayed pass: the values its step ran with)
        bool tri = false, rls_in = replay;
        // the parameters of step t - 1 (the SGD role's write-through stores) and its verdict on that step's loss
        // One lane polls the SGD role's count; once it is there it looks -- once -- at the RLS roles' count of the same step, and
        // at the verdict word.  No acquire: what the trial role takes from other roles (the parameter image, W, w_chol, sigma, rho)
        // it reads with sc1 loads behind this poll and the workgroup barrier (MI355X guide, "sc1 loads in place of the acquire").
        bool rls_now = false;
        auto gate = [&]() {
            if (t > 0) {
                vjf_chaos(tid, cnt + MG_C_SGD, 1);
                if (tid == 0) {
                    bool there = false;
                    for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
                        if ((int)(__hip_atomic_load(cnt + MG_C_SGD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)t * (unsigned)(RLS ? A.n_sgd : A.n_sgd_live)) >= 0) { there = true; break; }
                        if ((spins & 255u) == 255u && vjf_abort_seen(SCW + VJF_SC_STATUS)) break;
                        __builtin_amdgcn_s_sleep(RLS ? VJF_POLL_SLEEP : VJF_POLL_SLEEP_LITE);
                    }
                    const bool rls = !rls_in && (int)(__hip_atomic_load(cnt + MG_C_PDONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)t * npost) >= 0;
                    const unsigned mw = __hip_atomic_load(cnt + MG_C_MASK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (!tl || (A.flags & VJF_FLAG_HANDOFF_ACQUIRE)) {         // (parameters read from the state with plain loads; or the conservative hand-off)
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    s_try[0] = (there ? 1u : 0u) | (rls ? 2u : 0u);
                    s_try[1] = mw;
                    if (!there) vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE);
                    vjf_s_abort_word = (!there || vjf_abort_seen(SCW + VJF_SC_STATUS)) ? 1 : 0;   // (one verdict for the workgroup: vjf_abort_wg)
                }
                __syncthreads(); MG_PHASE();
                rls_now = (s_try[0] & 2u) != 0u;
                const unsigned mw = s_try[1];
                if (!replayed && (mw >> 8) == (unsigned)t) { rbits = mw & 7u; want_replay = true; }
            }
        };
        if (ts >= A.T && gated) gate();                // (behind the last step: only that)
        int it = 0;
        for (int tile = wg; tile < A.ntiles && ts < A.T; tile += A.n_trial, ++it) {
            const bool first = it == 0, last = it == ntl - 1;
            const int b0 = tile * TR;
            const int nb = min(TR, A.B - b0);
            __syncthreads(); MG_PHASE();                           // (the previous tile's readers of the LDS matrices are done)
            // ---- stage 0: inputs.  A tile's rows of y / u / mu_s / lv_s / eps are contiguous in memory: flat coalesced reads, all of a
            //      thread's loads in flight before its first (transposed) LDS write
            {
                auto cell = [&](const float* src, int d, unsigned md, int e, float& v, int& at) {   // element e of a (TR, d) tile -> value, LDS offset
                    const int b = mg_div(e, md), c2 = e - b * d;
                    v = (src != nullptr && b < nb) ? src[(size_t)b0 * d + e] : 0.f;
                    at = c2 * LD + b;
                };
                float vy[4], vs[4], vu = 0.f; int ay[4], as[4], au = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { vy[q] = 0.f; ay[q] = -1; if (tid + q * NT < TR * dy) cell(y_t, dy, m_dy, tid + q * NT, vy[q], ay[q]); }
                const bool sm = tid < TR * dz;                                  // (dz <= 16: one element of each small tile per thread)
                if (sm) {
                    cell(prior ? nullptr : mu_s, dz, m_dz, tid, vs[0], as[0]);
                    cell(prior ? nullptr : lv_s, dz, m_dz, tid, vs[1], as[1]);
                    cell(eps_s, dz, m_dz, tid, vs[2], as[2]);
                    cell(eps_t, dz, m_dz, tid, vs[3], as[3]);
                    if (prior) {
                        const int j = tid - mg_div(tid, m_dz) * dz;
                        vs[0] = S[P.off[VJF_SLOT_PRIOR_MEAN] + j]; vs[1] = S[P.off[VJF_SLOT_PRIOR_LOGVAR] + j];
                    }
                }
                if (du > 0 && tid < TR * du) cell(u_t, du, m_du, tid, vu, au);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (ay[q] >= 0) s_in[ay[q]] = vy[q];
                if (sm) {
                    s_in[(dy + du) * LD + as[0]] = vs[0];
                    s_in[(dy + du + dz) * LD + as[1]] = vs[1];
                    s_xt[as[2]] = vs[2];                                        // eps_s parked in s_xt
                    s_e2[as[3]] = vs[3];
                }
                if (du > 0 && tid < TR * du) s_in[dy * LD + au] = vu;
                for (int e0 = tid + 4 * NT; e0 < TR * dy; e0 += 4 * NT) {       // (wide observations: further rounds of four)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { vy[q] = 0.f; ay[q] = -1; if (e0 + q * NT < TR * dy) cell(y_t, dy, m_dy, e0 + q * NT, vy[q], ay[q]); }
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (ay[q] >= 0) s_in[ay[q]] = vy[q];
                }
            }
            if (tid < LD) s_one[tid] = tid < nb ? 1.f : 0.f;
            __syncthreads(); MG_PHASE();
            if (first) VJF_MG_STAMP(20);
            {
                for (int e = tid; e < TR * dxu; e += NT) {
                    const int c = e >> 5, b = e & 31;
                    float v;
                    if (c < dz) v = fmaf(s_xt[c * LD + b], expf(0.5f * s_in[(dy + du + dz + c) * LD + b]), s_in[(dy + du + c) * LD + b]);
                    else v = s_in[(dy + c - dz) * LD + b];
                    s_xu[c * LD + b] = v;
                }
                __syncthreads(); MG_PHASE();
                if (first) VJF_MG_STAMP(21);
                // ---- stage 1: RBF features (functional.py:11-22); a replayed pass needs none (its predictive mean / variance are saved)
                if (!replay && !use_mom)
                for (int e = tid; e < TR * n; e += NT) {
                    const int k = e >> 5, b = e & 31;
                    float d2 = 0.f;
                    for (int c = 0; c < dxu; ++c) { const float d = s_xu[c * LD + b] - s_cen[c * npad + k]; d2 = fmaf(d, d, d2); }
                    s_phi[k * LD + b] = expf(d2 * s_iw[k]);
                }
                __syncthreads(); MG_PHASE();
            }
            if (first) VJF_MG_STAMP(2);
            // ---- the RLS update of the previous step (W, w_chol, sigma: write-through stores of the RLS roles), if it is complete
            //      already: its acquire and the L2 warm-up then cost nothing on the path parameters -> forward -> backward.  If not,
            //      the same happens behind the forward pass (below): the values read are the same either way.
            if (first && !replay) {
                rls_in = t == 0 || !mode_rls;
                // (no parameter updates at all: nothing holds this role back between steps but the ring of loss sums -- the role
                //  that sums them must be through with the slot this step will write)
                // (the count only grows: what the last look saw usually covers the next ~30 steps -- a look per step was a memory round
                //  trip and a barrier, 1.2 us of a 20-us step)
                if (!gated && t >= VJF_MG_RING && (int)(ring_seen - (unsigned)(t - VJF_MG_RING + 1) * (unsigned)A.n_sgd_live) < 0) {
                    const unsigned need = (unsigned)(t - VJF_MG_RING + 1) * (unsigned)A.n_sgd_live;
                    vjf_chaos(tid, cnt + MG_C_SGD, 1);
                    if (tid == 0) {
                        bool there = false;
                        unsigned v = 0u;
                        for (unsigned spins = 0; spins < (1u << 21); ++spins) {
                            v = __hip_atomic_load(cnt + MG_C_SGD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((int)(v - need) >= 0) { there = true; break; }
                            if ((spins & 255u) == 255u && ((unsigned)__hip_atomic_load(SCW + VJF_SC_STATUS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & VJF_STATUS_WAIT_MASK)) break;
                            __builtin_amdgcn_s_sleep(RLS ? VJF_POLL_SLEEP : VJF_POLL_SLEEP_LITE);
                        }
                        if (!there) vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE);
                        s_try[1] = v;
                        vjf_s_abort_word = (!there || vjf_abort_seen(SCW + VJF_SC_STATUS)) ? 1 : 0;
                    }
                    __syncthreads();
                    ring_seen = s_try[1];
                    if (vjf_abort_wg()) return;
                }
                if (t > 0 && mode_rls) {
                    if (tid == 0) {
                        const bool there = (int)(__hip_atomic_load(cnt + MG_C_PDONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)t * npost) >= 0;
                        if (there && (A.flags & VJF_FLAG_HANDOFF_ACQUIRE)) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                        s_try[0] = there ? 1u : 0u;
                    }
                    __syncthreads(); MG_PHASE();
                    rls_in = s_try[0] != 0u;
                    if (rls_in) mg_warm(A.xt, P.n * P.n, wg, tid);
                }
                if (rls_in) {
                    sig = mg_ld(S + P.off[VJF_SLOT_TR_LOGVAR]);
                    tri = tri_launch || mg_ld(SCW + VJF_SC_TRI_CLEAN) != 0.f;  // w_chol known upper triangular
                }
                if (t == 0) {                                                 // (the row-major copy of L^-1 of this launch: the inverse loops' first act)
                    if (!mode_rls) {
                        // no RLS roles in this launch: w_chol is a constant of it, and the trial workgroups transpose a share each
                        // (and look at what they move: the state's triangle flag is only set by an RLS update -- a model that has never had
                        //  one, torch.eye (module.py:52), or a state that was just loaded would pay the full square in every variance
                        //  product of the launch although its w_chol is triangular.  A nonzero below the diagonal is counted in the word
                        //  behind the hand-off's own; both travel with the same signal)
                        const float* Wc = S + P.off[VJF_SLOT_W_CHOL];
                        float* xtw = const_cast<float*>(A.xt);
                        bool below = false;
                        for (int e = wg * NT + tid; e < n * n; e += A.n_trial * NT) {
                            const int k = e / n, j = e - k * n;
                            const float v = Wc[e];
                            below = below || (k > j && v != 0.f);
                            mg_st(xtw + (size_t)j * n + k, v);
                        }
                        if (__syncthreads_or(below ? 1 : 0) && tid == 0) __hip_atomic_fetch_add(cnt + MG_C_XT + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        vjf_wg_signal_wt(cnt + MG_C_XT, tid);
                    }
                    if (!vjf_wg_wait_sc1<RLS ? VJF_POLL_SLEEP : VJF_POLL_SLEEP_LITE>(cnt + MG_C_XT, (unsigned)(mode_rls ? A.n_rls - 2 : A.n_trial), tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u))
                        vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_K1);
                    if (vjf_abort_wg()) return;
                    if (!mode_rls) {
                        tri_launch = __hip_atomic_load(cnt + MG_C_XT + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
                        tri = tri || tri_launch;
                    }
                }
            }
            // (the two halves of stage 2 as routines: the training kernel runs them behind the RLS hand-off, where they always were;
            //  a launch without an RLS update has W, w_chol as constants and runs them BEFORE the gate, in the shadow of the SGD role)
            auto moments_a = [&]() {
            // ---- stage 2: predictive variance sum_j (Phi w_chol)_j^2 (module.py:75-76) and pt.mean = xs + Phi W (module.py:77)
            if (!replay) {
                const __amdgpu_buffer_rsrc_t r_xt = mg_rsrc(A.xt);
                const float* Wm = S + P.off[VJF_SLOT_W_MEAN];
                const int ntile = (n + 15) >> 4;
                float v2a = 0.f, v2b = 0.f;
                // pt.mean: dz <= 16 rows = one tile, K = n: every wavefront takes a K slice behind its variance tiles; the slice's
                // operand loads (at most 16 k-steps when n <= 512) go out now, in front of the variance tiles' own
                const int nsl = min(NW, part_rows / 16);
                const int msl = nsl - 1 - wave;                                // (the last wavefronts have the lightest variance shares)
                const int mper = (((n + 3) >> 2) + nsl - 1) / nsl * 4;
                const int mkb = msl * mper, mke = min(n, (msl + 1) * mper);
                const bool mpre = wave < nsl && ((mke - mkb + 3) >> 2) <= 16;
                float am[16];
                if (mpre && mke > mkb) mg_mma2_ld16(am, Wm, dz, dz, 0, mkb, mke, 0, lane);
                // tiles in descending cost, dealt to the wavefronts in a snake so that the triangular work balances; a wavefront's tiles of
                // two rounds go through mg_var2 as one stream of operand batches
                for (int r = 0; r * NW < ntile; r += 2) {
                    int j0p[2], Kp[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int rr = r + h, idx = (rr & 1) ? rr * NW + NW - 1 - wave : rr * NW + wave;
                        const int tt = ntile - 1 - idx;
                        j0p[h] = (idx < ntile) ? tt * 16 : -1;                  // rows j0 .. j0 + 15 of L^-1 = columns of w_chol
                        Kp[h] = tri ? min(n, tt * 16 + 16) : n;
                    }
                    if (j0p[0] < 0) { j0p[0] = j0p[1]; Kp[0] = Kp[1]; j0p[1] = -1; }
                    mg_var2(v2a, v2b, r_xt, n, j0p[0], Kp[0], j0p[1], Kp[1], s_phi, lane);
                }
                if (first) VJF_MG_STAMP(22);
                v2a += __shfl_xor(v2a, 16, 64); v2a += __shfl_xor(v2a, 32, 64);
                v2b += __shfl_xor(v2b, 16, 64); v2b += __shfl_xor(v2b, 32, 64);
                if (lane < 16) { s_red[wave * TR + lane] = v2a; s_red[wave * TR + 16 + lane] = v2b; }
                if (wave < nsl) {
                    const int sl = msl;
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    if (mpre) { if (mke > mkb) mg_mma2_mm16(acc0, acc1, am, s_phi, dz, 0, mkb, mke, 0, lane); }
                    else mg_mma2(acc0, acc1, Wm, dz, dz, 0, s_phi, mkb, mke, lane);
                    float* pr = s_part + (size_t)(sl * 16 + 4 * (lane >> 4)) * LD + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pr[r * LD] = acc0[r]; pr[r * LD + 16] = acc1[r]; }
                }
                mean_nsl = nsl;
            }
            __syncthreads(); MG_PHASE();
            };
            auto moments_b = [&]() {
            if (!replay) {
                if (tid < TR) {
                    float v = 0.f;
                    for (int w = 0; w < NW; ++w) v += s_red[w * TR + tid];
                    s_plv[tid] = logf(v);
                }
                for (int e = tid; e < TR * dz; e += NT) {
                    const int j = e >> 5, b = e & 31;
                    float v = 0.f;
                    for (int sl = 0; sl < mean_nsl; ++sl) v += s_part[(size_t)(sl * 16 + j) * LD + b];
                    s_pm[j * LD + b] = s_xu[j * LD + b] + v;
                    // warm-up: Phi W for the residual dx - Phi W of the state-noise update (model.py:373-374; W is the launch's constant),
                    // parked in the dmu rows until the loss stage, which has dx, sums the squares
                    if (want_resid) s_dmu[j * LD + b] = v;
                }
            }
            __syncthreads(); MG_PHASE();
            };
            // (the slab traffic of a step flows through the same L2s and pushes L^-1 out of some of them: the workgroups of an XCD bring it
            //  back together, a sixteenth each, before they all walk it -- measured without: one XCD's workgroups 12 us late at the gate)
            if (!RLS && !replay && first && !use_mom) { mg_warm(A.xt, P.n * P.n, wg, tid); mg_warm(S + P.off[VJF_SLOT_W_MEAN], (P.n * P.dz) & ~3, wg, tid); }
            if (!RLS && !replay && !use_mom) { moments_a(); moments_b(); }
            if (!RLS && !replay && first) VJF_MG_STAMPW(2);   // (diagnostic: when this workgroup reached the gate)
            // ---- theta of the previous step.  Nothing above depends on it: the inputs and the features of a step are ready before the
            //      parameters are
            if (first && !replay) {
                if (gated) gate();
                if (gated && t > 0 && vjf_abort_wg()) return;
                if (want_replay) break;                                        // (uniform: every thread read the same word)
                if (t > 0 && !tl) {
                    mg_warm(A.aux, P.aux_len, wg, tid);                        // (see mg_warm)
                    mg_warm(S + P.train_off, P.train_len, wg, tid);
                }
            }
            float4 wv[2];
            const bool warm_now = first && !replay && rls_now && !rls_in;      // the RLS update landed while this workgroup waited for the parameters
            if (first && !replay) rho = mg_ld(S + P.off[VJF_SLOT_LIK_LOGVAR]);  // (the SGD role's)
            if (first && !replay && !mode_rls) sig = mg_ld(S + P.off[VJF_SLOT_TR_LOGVAR]);   // (warm-up: the SGD role's too; else a constant)
            if (warm_now) {
                mg_warm_issue(A.xt, P.n * P.n, wg, tid, wv);
                sig = mg_ld(S + P.off[VJF_SLOT_TR_LOGVAR]);
                tri = mg_ld(SCW + VJF_SC_TRI_CLEAN) != 0.f;
                rls_in = true;
            }
            if (first && tl && !replay && t == 0) {                            // (the image of this launch: the SGD role's first act)
                if (!vjf_wg_wait_sc1<RLS ? VJF_POLL_SLEEP : VJF_POLL_SLEEP_LITE>(cnt + MG_C_IMG, (unsigned)A.n_sgd, tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u))
                    vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE);
                if (vjf_abort_wg()) return;
            }
            if (first && tl && !replay && (gated || t == 0)) {                 // (a replayed pass: they are in LDS, untouched since its step; a launch
                                                                               //  that updates nothing: they are the launch's constants, staged once)
                // the parameters of this step into LDS: the image the SGD role keeps has the layout of the region, so this is a flat
                // 16-byte copy with all of a thread's loads in flight -- one round trip
                const __amdgpu_buffer_rsrc_t r_img = mg_rsrc(A.img);
                float4* dst = reinterpret_cast<float4*>(smem + Lo.th0);
                const int n4 = Lo.th_len >> 2;
                for (int q0 = tid; q0 < n4; q0 += 8 * NT) {
                    float4 v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (q0 + q * NT < n4) v[q] = mg_ld4(r_img, (q0 + q * NT) * 4);
#pragma unroll
                    for (int q = 0; q < 8; ++q) if (q0 + q * NT < n4) dst[q0 + q * NT] = v[q];
                }
                __syncthreads(); MG_PHASE();
            }
            if (warm_now) mg_warm_retire(wv);
            if (first) { VJF_MG_STAMP(1); VJF_MG_STAMPX(27, -1); VJF_MG_STAMPW(0); }
            if (first && A.stamps && tid == 0 && t == A.T - 1 && !replay) {
                unsigned xcc;
                unsigned hwid;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
                A.stamps[1024 + (size_t)wg * 8 + 6] = (xcc & 15u) | ((unsigned long long)hwid << 8);
                A.stamps[1024 + (size_t)wg * 8 + 7] = rls_in ? 1u : 0u;
            }
            // ---- stage 3: recognition forward (recognition.py:31-42)
            {
                const float* xin = s_in;
                int kin = din, aoff = 0;
                for (int l = 0; l < P.L; ++l) {
                    const float* WT = A.aux + P.aux_recT[l];                   // (kin, hl)
                    int th_w = 0, th_ldw = 0, th_b = 0;
                    if (tl) mg_theta_layer(P, Lo.th0, l, th_w, th_ldw, th_b);
                    // (bias values through a select of two TYPED loads, never a load through a selected pointer: a pointer that is LDS on
                    //  one side and memory on the other is a generic one, and the aperture test the backend builds for it is the instruction
                    //  this compiler rejects -- "V_CMP_NE_U32_e32 0, $src_shared_base", found with -mllvm -verify-machineinstrs)
                    const float* bias_l = smem + th_b; const float* bias_g = S + P.off[VJF_SLOT_REC_B0 + 2 * l];
                    float* out = s_act + aoff * LD;
                    const int hl = P.h[l], mt = (hl + 15) >> 4;
                    for (int tt = wave; tt < mt; tt += NW) {
                        vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                        if (tl) mg_mma2_lds<false>(acc0, acc1, smem + th_w, th_ldw, hl, tt * 16, xin, 0, kin, lane);
                        else mg_mma2(acc0, acc1, WT, hl, hl, tt * 16, xin, 0, kin, lane);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int f = tt * 16 + 4 * (lane >> 4) + r;
                            if (f < hl) {
                                const float bf = tl ? ((mg_lds_cf*)bias_l)[f] : ((mg_glb_cf*)bias_g)[f];
                                out[f * LD + (lane & 15)] = mg_tanh(acc0[r] + bf);
                                out[f * LD + 16 + (lane & 15)] = mg_tanh(acc1[r] + bf);
                            }
                        }
                    }
                    __syncthreads(); MG_PHASE();
                    xin = out; kin = hl; aoff += hl;
                }
                if (first) VJF_MG_STAMP(23);
                // heads: 2 dz <= 32 rows = at most two 16-row tiles -- the K range is split over the wavefronts, partial tiles meet in
                // LDS (s_part: a region that is free until the losses / the backward pass) and are summed in slice order
                const float* HT = A.aux + P.aux_headT;                         // (hL, 2dz): mean rows then logvar rows
                const int mt = (2 * dz + 15) >> 4;
                const int nsl = min(NW / mt, part_rows / (16 * mt));
                if (wave < mt * nsl) {
                    const int tt = wave / nsl, sl = wave - tt * nsl;
                    const int per = (((kin + 3) >> 2) + nsl - 1) / nsl * 4;
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    if (tl) mg_mma2_lds<false>(acc0, acc1, smem + Lo.th_head, Lo.th_ldh, 2 * dz, tt * 16, xin, sl * per, min(kin, (sl + 1) * per), lane);
                    else mg_mma2(acc0, acc1, HT, 2 * dz, 2 * dz, tt * 16, xin, sl * per, min(kin, (sl + 1) * per), lane);
                    float* pr = s_part + (size_t)((sl * mt + tt) * 16 + 4 * (lane >> 4)) * LD + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { pr[r * LD] = acc0[r]; pr[r * LD + 16] = acc1[r]; }
                }
                __syncthreads(); MG_PHASE();
                if (first) VJF_MG_STAMP(24);
                const float* bl_l = smem + Lo.th_bl; const float* bl_g = S + P.off[VJF_SLOT_LV_B];
                for (int e = tid; e < TR * 2 * dz; e += NT) {
                    const int f = e >> 5, b = e & 31;
                    float v = 0.f;
                    for (int sl = 0; sl < nsl; ++sl) v += s_part[(size_t)((sl * mt + (f >> 4)) * 16 + (f & 15)) * LD + b];
                    if (f < dz) s_mu[f * LD + b] = v; else s_lv[(f - dz) * LD + b] = v + (tl ? ((mg_lds_cf*)bl_l)[f - dz] : ((mg_glb_cf*)bl_g)[f - dz]);
                }
            }
            __syncthreads(); MG_PHASE();
            if (first) VJF_MG_STAMP(3);
            // ---- stage 4: xt, dx, posterior out, py = xt C^T + d (model.py:28-30)
            for (int e = tid; e < TR * dz; e += NT) {
                const int j = e >> 5, b = e & 31;
                const float xt = fmaf(s_e2[j * LD + b], expf(0.5f * s_lv[j * LD + b]), s_mu[j * LD + b]);
                s_xt[j * LD + b] = xt;
                s_dx[j * LD + b] = b < nb ? xt - s_xu[j * LD + b] : 0.f;
            }
            if (!replay) {
                // posterior out (write-through: the Gram role forms the next step's features from it).  The tile's rows are contiguous
                // in memory: four consecutive elements per 16-byte store where the tile starts on a 16-byte boundary (a quarter of
                // the fabric writes: 82 k scalar ones per step at config B before), scalar stores for what is left over
                float* mrow = mu_t + (size_t)b0 * dz;
                float* lrow = lv_t + (size_t)b0 * dz;
                const int ne = nb * dz;
                const int n4 = ((((size_t)mrow | (size_t)lrow) & 15u) == 0) ? (ne >> 2) : 0;
                for (int e4 = tid; e4 < n4; e4 += NT) {
                    float vm[4], vl[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int e = 4 * e4 + c, b = mg_div(e, m_dz), j = e - b * dz;
                        vm[c] = s_mu[j * LD + b]; vl[c] = s_lv[j * LD + b];
                    }
                    mg_st4(mrow + 4 * e4, vm[0], vm[1], vm[2], vm[3]);
                    mg_st4(lrow + 4 * e4, vl[0], vl[1], vl[2], vl[3]);
                }
                for (int e = 4 * n4 + tid; e < ne; e += NT) {
                    const int b = mg_div(e, m_dz), j = e - b * dz;
                    mg_st(mrow + e, s_mu[j * LD + b]);
                    mg_st(lrow + e, s_lv[j * LD + b]);
                }
                // (the moments role's tag for this posterior goes out behind the decoder, below: its stores are acknowledged by then, and
                //  a drain here was 1.5 us on the path of every step; the moments role is a step ahead)
            }
            __syncthreads(); MG_PHASE();
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
                const float* d_l = smem + Lo.th_bd; const float* d_g = S + P.off[VJF_SLOT_DEC_B];
                const int mt = (dy + 15) >> 4;
                for (int tt = wave; tt < mt; tt += NW) {
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    if (tl) mg_mma2_lds<false>(acc0, acc1, smem + Lo.th_dec, Lo.th_ldd, dy, tt * 16, s_xt, 0, dz, lane);
                    else mg_mma2(acc0, acc1, CT, dy, dy, tt * 16, s_xt, 0, dz, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int f = tt * 16 + 4 * (lane >> 4) + r;
                        if (f < dy) { const float df = tl ? ((mg_lds_cf*)d_l)[f] : ((mg_glb_cf*)d_g)[f]; s_py[f * LD + (lane & 15)] = acc0[r] + df; s_py[f * LD + 16 + (lane & 15)] = acc1[r] + df; }
                    }
                }
            }
            if (first) VJF_MG_STAMP(25);
            // early slab: Phi^T dx of this tile (module.py:94), 16 features x 16 columns per MFMA tile, K = 32 trials
            if (!replay && mode_rls) {
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
                    // [dz column][feature]: a lane's four registers are four consecutive features of one column (features >= n: the
                    // zero row of the A operand)
                    const int fq = m0 + 4 * (lane >> 4), col = lane & 15;
                    if (col < dz && fq < ldn) {
                        float* p = early + (size_t)col * ldn + fq;
                        if (!first) { acc[0] += mg_ld(p); acc[1] += mg_ld(p + 1); acc[2] += mg_ld(p + 2); acc[3] += mg_ld(p + 3); }
                        mg_st4(p, acc[0], acc[1], acc[2], acc[3]);
                    }
                }
            }
            // (a look at the RLS hand-off of the previous step by one lane in front of this barrier, where the other wavefronts are
            //  still storing their slab tiles: see fuse_fwd)
            if (tid == 0 && first && last && !replay && !rls_in)
                s_wg[15] = ((int)(__hip_atomic_load(cnt + MG_C_PDONE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)t * npost) >= 0) ? 1.f : 0.f;
            if (use_mom && !replay) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tile's posterior is in memory: its tag follows the barrier)
            __syncthreads(); MG_PHASE();
            if (use_mom && !replay && tid == 0) __hip_atomic_store(cnt + MG_C_TAG_POST + tile, (unsigned)(tc + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // A launch without parameter updates has nothing between its steps to hide the next step's inputs behind (2.5 us of a 20-us
            // step: y, the noise): one load per 128-byte line of them goes out here and is retired behind the moments' own loads --
            // the staging of the next step then finds them in the caches.
            float touch = 0.f;
            if (!gated && !replay && last && tc + 1 < A.T) {
                const int tn = wg;                                   // (the next step starts with this workgroup's first tile)
                const int b0n = tn * TR, nbn = min(TR, A.B - b0n);
                const int ly = (nbn * dy + 31) / 32 + 1, le = (nbn * dz + 31) / 32 + 1;
                const float* yn = A.y + (size_t)(tc + 1) * sy + (size_t)b0n * dy;
                const float* en = A.eps + (size_t)(tc + 1) * 2 * sz + (size_t)b0n * dz;
                const float* tp = nullptr;
                if (tid < ly) tp = yn + min(tid * 32, nbn * dy - 1);
                else if (tid < ly + le) tp = en + min((tid - ly) * 32, nbn * dz - 1);
                else if (tid < ly + 2 * le) tp = en + sz + min((tid - ly - le) * 32, nbn * dz - 1);
                if (tp) touch = *tp;
            }
            if (tid == 0 && !replay && mode_rls) {
                float v = 0.f;
                for (int bb = 0; bb < TR; ++bb) v += s_sc[bb * RS_N + RS_SDX2];
                s_wg[RS_SDX2] += v;
                if (last) mg_st(early + (size_t)16 * ldn + RS_SDX2, s_wg[RS_SDX2]);
            }
            if (first) VJF_MG_STAMP(26);
            // One tile per workgroup and the RLS update of the previous step still to be taken in, but there by now (config B: it
            // lands ~5 us before this point): the early slab's write-through stores are not drained here -- their acknowledgements
            // travel beside the round trips of that hand-off, below, and the "forward done" count follows there (one drain, one
            // barrier less).  If it is NOT there yet (configs whose RLS loop alone bounds the step, e.g. one trial against RBF(100):
            // the wait below lasts ~10 us) the count goes out now -- the Gram role's sums of the next step, and with them the next
            // factorisation, wait for it (measured at configs[0]: 35.2 us a step with the count behind the wait, 30.0 before it).
            const bool fuse_fwd = first && last && !replay && !rls_in && s_wg[15] != 0.f;
            if (last && !replay && !fuse_fwd && mode_rls) vjf_wg_signal_wt(cnt + MG_C_FWD, tid);
            if (first) VJF_MG_STAMP(4);
            if (last) { VJF_MG_STAMPX(28, -1); VJF_MG_STAMPW(1); }
            // ---- the RLS update of the previous step, if it had not landed before the forward pass
            float4 wv_late[2];
            bool warm_late = false;
            if (first && !rls_in) {
                if (!vjf_wg_wait_sc1<RLS ? VJF_POLL_SLEEP : VJF_POLL_SLEEP_LITE>(cnt + MG_C_PDONE, (unsigned)t * npost, tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u))
                    vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_K1);
                if (vjf_abort_wg()) return;
                // (sigma and the triangle flag first, then this workgroup's share of the L2 warm-up with its loads left in flight: the
                //  variance tiles' own operand loads go out behind them instead of waiting a round trip for them)
                sig = mg_ld(S + P.off[VJF_SLOT_TR_LOGVAR]);
                tri = mg_ld(SCW + VJF_SC_TRI_CLEAN) != 0.f;
                if (fuse_fwd) {
                    // every wavefront's stores of the forward pass (posterior, early slab) and these two loads are behind it: the count
                    // the operand and Gram roles wait for
                    vjf_chaos(tid, cnt + MG_C_FWD, 2);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) __hip_atomic_fetch_add(cnt + MG_C_FWD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                mg_warm_issue(A.xt, P.n * P.n, wg, tid, wv_late);
                warm_late = true;
            }
            if (first) { VJF_MG_STAMP(5); if (RLS) VJF_MG_STAMPW(2); }
            if (use_mom && !replay) {
                // the tile's moments of this step from the moments role: its tag, then pt.mean | Phi W | pt.logvar with sc1 loads
                if (!vjf_wg_wait_sc1<RLS ? VJF_POLL_SLEEP : VJF_POLL_SLEEP_LITE>(cnt + MG_C_TAG_MOM + tile, (unsigned)(tc + 1), tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u))
                    vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_K1);
                if (vjf_abort_wg()) return;
                const float* mb = A.mom + ((size_t)tile * 2 + (size_t)(tc & 1)) * (size_t)((2 * dz + 1) * TR);
                for (int e = tid; e < TR * (2 * dz + 1); e += NT) {
                    const int j = e >> 5, b = e & 31;
                    const float v = mg_ld(mb + e);
                    if (j < dz) s_pm[j * LD + b] = v;
                    else if (j < 2 * dz) { if (want_resid) s_dmu[(j - dz) * LD + b] = v; }
                    else s_plv[b] = v;
                }
                __syncthreads(); MG_PHASE();
            }
            asm volatile("" ::"v"(touch));
            if (RLS) moments_a();
            if (warm_late) mg_warm_retire(wv_late);
            if (last && tid == 0 && !replay && mode_rls) __hip_atomic_fetch_add(cnt + MG_C_K1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // W, w_chol, sigma read
            if (RLS) moments_b();
            // pt.mean | pt.logvar of the tile's trials: kept for a replay of this step (by then W and w_chol have moved on)
            if (do_sgd)
            for (int e = tid; e < TR * (dz + 1); e += NT) {
                const int j = e >> 5, b = e & 31;
                if (b < nb) {
                    float* sv = A.pmsave + (size_t)(b0 + b) * (dz + 1) + j;
                    if (!replay) *sv = j < dz ? s_pm[j * LD + b] : s_plv[b];
                    else if (j < dz) s_pm[j * LD + b] = *sv;
                    else s_plv[b] = *sv;
                }
            }
            if (replay) { __syncthreads(); MG_PHASE(); }
            if (first) VJF_MG_STAMP(10);
            if (first) { VJF_MG_STAMP(6); VJF_MG_STAMPW(3); }
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
                        s_dpy[i * LD + b] = (ok && m_r) ? e * r : 0.f;
                    }
                } else {                                                       // likelihood.py:51-62
                    for (int i = s; i < dy; i += LPT) {
                        const float yv = s_in[i * LD + b], pv = s_py[i * LD + b];
                        const float eta = fminf(pv, 10.f), ex = expf(eta);
                        lrec += ex - yv * eta;
                        const float r = pv - yv;
                        ssey = fmaf(r, r, ssey);
                        s_dpy[i * LD + b] = (ok && m_r && pv <= 10.f) ? (ex - yv) : 0.f;
                    }
                }
                lrec = group_sum<LPT>(lrec);
                ssey = group_sum<LPT>(ssey);
                float ldyn = 0.f, ent = 0.f, rsd = 0.f;
                {
                    const float p = expf(-0.5f * sig), e = expf(-sig), plv = s_plv[b];
                    for (int j = s; j < dz; j += LPT) {                         // model.py:390-391, functional.py:62-75
                        const float mp = s_pm[j * LD + b], mu = s_mu[j * LD + b], lv = s_lv[j * LD + b];
                        if (want_resid) { const float r = s_dx[j * LD + b] - s_dmu[j * LD + b]; rsd = fmaf(r, r, rsd); }   // (read before dmu goes there)
                        const float dsc = mp * p - mu * p;
                        const float tr = expf(plv + lv - sig);
                        ldyn += 0.5f * (dsc * dsc + sig) + 0.5f * tr;
                        ent += 0.5f * lv;                                      // functional.py:25-29
                        float dmu = 0.f, dlv = m_h ? -0.5f : 0.f;
                        if (!warm && m_d) { dmu = -e * (mp - mu); dlv += 0.5f * tr; }
                        s_dmu[j * LD + b] = ok ? dmu : 0.f;
                        s_dlv[j * LD + b] = ok ? dlv : 0.f;
                    }
                }
                ldyn = group_sum<LPT>(ldyn);
                ent = group_sum<LPT>(ent);
                if (want_resid) rsd = group_sum<LPT>(rsd);
                if (s == 0) {
                    s_sc[b * RS_N + RS_RESID] = ok ? rsd : 0.f;
                    s_sc[b * RS_N + RS_LRECON] = ok ? lrec : 0.f;
                    s_sc[b * RS_N + RS_LDYN] = ok ? ldyn : 0.f;
                    s_sc[b * RS_N + RS_ENT] = ok ? ent : 0.f;
                    s_sc[b * RS_N + RS_SSEY] = ok ? ssey : 0.f;
                }
            }
            __syncthreads(); MG_PHASE();
            if (first) VJF_MG_STAMP(31);
#ifdef VJF_EXPERIMENT_SLOW_TRIAL   /* sensitivity experiment: every trial workgroup held for this many 10-ns ticks per step */
            { const unsigned long long t0_ = wall_clock64(); while (wall_clock64() - t0_ < VJF_EXPERIMENT_SLOW_TRIAL) __builtin_amdgcn_s_sleep(1); }
#endif
            if ((tid < RS_SDX2 || (tid == RS_RESID && want_resid)) && !replay) {   // (RS_LRECON, RS_LDYN, RS_ENT, RS_SSEY; the residual)
                float v = 0.f;
                for (int bb = 0; bb < TR; ++bb) v += s_sc[bb * RS_N + tid];
                s_wg[tid] += v;
            }
            // ---- stage 6: backward.  dxt = dpy C ; dmu += dxt ; dlv += dxt eps_t exp(lv/2)/2.  Every product whose A
            //      operand comes from memory runs BEFORE the first gradient tile goes out: a load issued behind write-through stores
            //      waits for them to reach memory (vmcnt counts in order).
            if (do_sgd) {
            int gbase = 0;                                                     // running tile count: gradient tiles go round the wavefronts
            auto grad_tensor = [&](const float* D, int M, const float* Bact, int Kin, int blkid) {
                int b_off, b_ldm, b_rows;
                mg_slab_block(P, blkid, b_off, b_ldm, b_rows);
                const int ntm = (M + 15) >> 4, ntj = (Kin + 1 + 15) >> 4;
                const unsigned mj = mg_magic(ntj);
                for (int q = (wave - gbase) & (NW - 1); q < ntm * ntj; q += NW) {                   // this wavefront's tiles of the tensor
                    const int tm = mg_div(q, mj), tj = q - tm * ntj;
                    mg_grad_tile(D, M, tm * 16, Bact, Kin, tj * 16, s_one, s_zero, late + b_off, b_ldm, b_rows, first, lane);
                }
                gbase += ntm * ntj;
            };
            {
                const float* C = S + P.off[VJF_SLOT_DEC_W];                    // (dy, dz): k-major for this product
                const int mt = (dz + 15) >> 4;
                // dz <= 16 rows = ONE tile: the K range (the observations) is split over the wavefronts, as the heads' is -- one wavefront
                // alone took 3.9 us for it while seven waited.  The partial tiles meet in rows that are dead here: the decoder's
                // means (consumed by the losses) or the delta buffers (written from the next stage on).
                float* s_kp = compact ? s_py : s_dd;
                const int rows_av = compact ? dy : Lo.nd * P.hmax;
                const int nslb = min(NW / mt, rows_av / (16 * mt));
                if (nslb >= 2) {
                    if (wave < mt * nslb) {
                        const int tt = wave / nslb, sl = wave - tt * nslb;
                        const int per = (((dy + 3) >> 2) + nslb - 1) / nslb * 4;
                        vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                        if (tl) mg_mma2_lds<true>(acc0, acc1, smem + Lo.th_dec, Lo.th_ldd, dz, tt * 16, s_dpy, sl * per, min(dy, (sl + 1) * per), lane);
                        else mg_mma2(acc0, acc1, C, dz, dz, tt * 16, s_dpy, sl * per, min(dy, (sl + 1) * per), lane);
                        float* pr = s_kp + (size_t)((sl * mt + tt) * 16 + 4 * (lane >> 4)) * LD + (lane & 15);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { pr[r * LD] = acc0[r]; pr[r * LD + 16] = acc1[r]; }
                    }
                    __syncthreads(); MG_PHASE();
                    for (int e = tid; e < TR * dz; e += NT) {
                        const int j = e >> 5, b = e & 31;
                        float a = 0.f;
                        for (int sl = 0; sl < nslb; ++sl) a += s_kp[(size_t)((sl * mt + (j >> 4)) * 16 + (j & 15)) * LD + b];   // (padding trials: dpy = 0, so a = 0)
                        s_dmu[j * LD + b] += a;
                        s_dlv[j * LD + b] = fmaf(a * s_e2[j * LD + b], 0.5f * expf(0.5f * s_lv[j * LD + b]), s_dlv[j * LD + b]);
                    }
                } else
                for (int tt = wave; tt < mt; tt += NW) {
                    vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                    if (tl) mg_mma2_lds<true>(acc0, acc1, smem + Lo.th_dec, Lo.th_ldd, dz, tt * 16, s_dpy, 0, dy, lane);
                    else mg_mma2(acc0, acc1, C, dz, dz, tt * 16, s_dpy, 0, dy, lane);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = tt * 16 + 4 * (lane >> 4) + r;
                        if (j < dz) {
#pragma unroll
                            for (int g = 0; g < 2; ++g) {
                                const int b = 16 * g + (lane & 15);
                                const float a = g ? acc1[r] : acc0[r];         // (padding trials: dpy = 0, so a = 0)
                                s_dmu[j * LD + b] += a;
                                s_dlv[j * LD + b] = fmaf(a * s_e2[j * LD + b], 0.5f * expf(0.5f * s_lv[j * LD + b]), s_dlv[j * LD + b]);
                            }
                        }
                    }
                }
            }
            __syncthreads(); MG_PHASE();
            if (first) VJF_MG_STAMP(7);
            {
                const int hL = P.h[P.L - 1];
                const float* Wm = S + P.off[VJF_SLOT_MEAN_W];                  // (dz, hL): k-major for dh = dmu Wm + dlv Wl
                const float* Wl = S + P.off[VJF_SLOT_LV_W];
                const float* hact = s_act + (P.hsum - hL) * LD;
                // dh_{l-1} = da_l W_l (1 - h_{l-1}^2)  into `dst`   (l = L: the heads)
                auto delta = [&](int l, const float* src, float* dst) {
                    const int hp = P.h[l - 1];
                    int aoff = 0;
                    for (int q = 0; q < l - 1; ++q) aoff += P.h[q];
                    const float* hprev = s_act + aoff * LD;
                    const int mt = (hp + 15) >> 4;
                    int d_w = 0, d_ldw = 0, d_b = 0;
                    if (tl && l < P.L) mg_theta_layer(P, Lo.th0, l, d_w, d_ldw, d_b);
                    for (int tt = wave; tt < mt; tt += NW) {
                        vjf_f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                        if (l == P.L) {
                            if (tl) {
                                mg_mma2_lds<true>(acc0, acc1, smem + Lo.th_head, Lo.th_ldh, hL, tt * 16, s_dmu, 0, dz, lane);
                                mg_mma2_lds<true>(acc0, acc1, smem + Lo.th_head + dz * Lo.th_ldh, Lo.th_ldh, hL, tt * 16, s_dlv, 0, dz, lane);
                            } else {
                                mg_mma2(acc0, acc1, Wm, hL, hL, tt * 16, s_dmu, 0, dz, lane);
                                mg_mma2(acc0, acc1, Wl, hL, hL, tt * 16, s_dlv, 0, dz, lane);
                            }
                        } else if (tl) {
                            mg_mma2_lds<true>(acc0, acc1, smem + d_w, d_ldw, hp, tt * 16, src, 0, P.h[l], lane);
                        } else {
                            mg_mma2(acc0, acc1, S + P.off[VJF_SLOT_REC_W0 + 2 * l], hp, hp, tt * 16, src, 0, P.h[l], lane);   // (h_l, h_{l-1}): k-major
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int k = tt * 16 + 4 * (lane >> 4) + r, b = lane & 15;
                            if (k < hp) {
                                const float h0 = hprev[k * LD + b], h1 = hprev[k * LD + 16 + b];
                                dst[k * LD + b] = acc0[r] * (1.f - h0 * h0);
                                dst[k * LD + 16 + b] = acc1[r] * (1.f - h1 * h1);
                            }
                        }
                    }
                };
                auto layer_grads = [&](int l, const float* da) {               // weights / bias of recognition layer l from da_l and its input
                    int aoff = 0;
                    for (int q = 0; q < l - 1; ++q) aoff += P.h[q];
                    grad_tensor(da, P.h[l], l > 0 ? s_act + aoff * LD : s_in, l > 0 ? P.h[l - 1] : din, 3 + (P.L - 1 - l));
                };
                delta(P.L, nullptr, s_d0);                                     // da_{L-1}
                __syncthreads(); MG_PHASE();
                if (P.L >= 2) { delta(P.L - 1, s_d0, s_d1); __syncthreads(); MG_PHASE(); } // da_{L-2}
                if (first) VJF_MG_STAMP(19);
                // gradient tiles (write-through stores into the workgroup's late slab)
                grad_tensor(s_dpy, dy, s_xt, dz, 0);
                grad_tensor(s_dmu, dz, hact, hL, 1);
                grad_tensor(s_dlv, dz, hact, hL, 2);
                layer_grads(P.L - 1, s_d0);
                if (P.L >= 2) layer_grads(P.L - 2, s_d1);
                float* cur = s_d1; float* nxt = s_d0;                          // deeper networks: the two delta buffers alternate
                for (int l = P.L - 3; l >= 0; --l) {
                    __syncthreads(); MG_PHASE();
                    delta(l + 1, cur, nxt);
                    __syncthreads(); MG_PHASE();
                    layer_grads(l, nxt);
                    float* tmp = cur; cur = nxt; nxt = tmp;
                }
            }
            }
            if (first) { VJF_MG_STAMP(8); VJF_MG_STAMPW(4); }
            if (last) {
                // the workgroup's late slab is complete: loss sums, then the signal the SGD role waits for
                __syncthreads(); MG_PHASE();
                if ((tid < RS_SDX2 || (tid == RS_RESID && want_resid)) && !replay) mg_st(late + A.slab_len + 8 * (tc % VJF_MG_RING) + tid, s_wg[tid]);
                if (gated) vjf_wg_signal_wt(cnt + (replay ? MG_C_REDO_B : MG_C_BWD), tid);
                else {
                    // No gate between the steps of this launch (nothing changes between them): the trial workgroups are not in step
                    // with each other and no role waits for them.  Each counts itself in at the step's word of a ring; the one whose
                    // add comes LAST (told by the value the add returns: every other workgroup's sums are in memory, drained before
                    // its add) sums the step's loss terms in the fixed order, writes the loss, puts the word back to 0 and counts the
                    // step as done -- the count that keeps any workgroup from running a ring's length ahead.
                    vjf_chaos(tid, cnt + MG_C_ARR, 2);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) s_try[0] = __hip_atomic_fetch_add(cnt + MG_C_ARR + (tc % VJF_MG_RING), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned)A.n_trial ? 1u : 0u;
                    __syncthreads(); MG_PHASE();
                    if (s_try[0]) {
                        float* s_tot = s_sc;                                   // (the per-trial terms of this workgroup's tile are summed and stored)
                        mg_sum_losses(A, tc, s_tot, tid, (float)A.B, dz, false);
                        if (tid == 0) {
                            const float invB = 1.0f / (float)A.B;
                            float l_recon = s_tot[RS_LRECON] * invB, l_dyn = s_tot[RS_LDYN] * invB, ent = s_tot[RS_ENT] * invB;
                            const bool ok_r = isfinite(l_recon), ok_d = isfinite(l_dyn), ok_h = isfinite(ent);
                            if (!ok_r) l_recon = 0.f;
                            if (!ok_d) l_dyn = 0.f;
                            if (!ok_h) ent = 0.f;
                            const float loss = warm ? l_recon - ent : l_recon - ent + l_dyn;   // model.py:146-149
                            if (A.loss) { float* l4 = A.loss + 4 * (size_t)tc; l4[0] = loss; l4[1] = -l_recon; l4[2] = -l_dyn; l4[3] = ent; }
                            const unsigned st = (ok_r ? 0u : VJF_STATUS_NONFINITE_RECON) | (ok_d ? 0u : VJF_STATUS_NONFINITE_DYN) | (ok_h ? 0u : VJF_STATUS_NONFINITE_ENT);
                            if (st) vjf_status_or(SCW + VJF_SC_STATUS, st);
                            __hip_atomic_store(cnt + MG_C_ARR + (tc % VJF_MG_RING), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __hip_atomic_fetch_add(cnt + MG_C_SGD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        __syncthreads(); MG_PHASE();
                    }
                }
                VJF_MG_STAMP(9);
                VJF_MG_STAMPX(29, 30);
                VJF_MG_STAMPW(5);
            }
        }
        VJF_MG_STAMP(18);
        if (want_replay) { replay = true; continue; }
        if (replay) {
            // the SGD role's step on the replayed late slabs; then this step starts over (inputs, features, parameters)
            ++nredo;
            if (!(tl ? vjf_wg_wait_sc1<RLS ? VJF_POLL_SLEEP : VJF_POLL_SLEEP_LITE>(cnt + MG_C_REDO_S, nredo * (unsigned)A.n_sgd, tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u)
                     : vjf_wg_wait(cnt + MG_C_REDO_S, nredo * (unsigned)A.n_sgd, tid, SCW + VJF_SC_STATUS)))
                vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE);
            if (vjf_abort_wg()) return;
            replay = false; replayed = true; rbits = 0;
            continue;
        }
        sig_prev = sig; rho_prev = rho;
        break;
      }
    }
}

// ------------------------------------------------------------------------------------------------ moments role
// Launches without an RLS update (vjf_mega_lite_kernel): W and w_chol are constants, and the predictive moments of a tile at step t
// -- pt.mean = xs + Phi W, pt.logvar = log |L^-1 phi|^2 (module.py:64-77) -- depend on nothing but its posterior of step t - 1 and
// the noise.  They are a third of a trial workgroup's serial work per step (RBF features ~7 us, variance and mean ~12 us of ~52), and the
// launch has compute units to spare: these workgroups form them a step ahead, tile by tile, operation for operation what the trial role
// does (the same bits), and hand pt.mean | Phi W | pt.logvar over through memory.  One producer and one consumer per tile: step
// tags in the launch's counter block (MG_C_TAG_POST, MG_C_TAG_MOM), no counts.
static inline size_t vjf_mega_mom_lds_floats(const VjfPlan& P) {
    const size_t npad = (size_t)((P.n + 3) & ~3), LD = 65;                     // (two tiles side by side: 64 columns + 1)
    return npad * P.dxu + npad + (size_t)P.dxu * LD + (size_t)P.n * LD + (size_t)VJF_MG_WAVES * 64 + (size_t)VJF_MG_WAVES * 16 * LD + 64;
}

// mg_var2 / mg_mma2_mm16 for NG column groups of 16 trials (NG = 2: one tile, LD = 33, the trial role's routines instruction for
// instruction; NG = 4: two tiles side by side, LD = 65 -- every operand load of L^-1 and W then feeds twice the multiply-adds).  A
// trial's sums run over k in the same order whatever NG is: the same bits.
template <int NG, int LD>
__device__ __forceinline__ void mg_varN(float (&v2)[NG], __amdgpu_buffer_rsrc_t rx, int n, int j0A, int KA, int j0B, int KB, mg_lds_cf* Xs, int lane,
                                        const bool upper = true) {        // upper = false (uniform): column groups 2, 3 hold no trials, their multiply-adds are skipped
    if (j0A < 0) return;
    const int i = lane & 15, kk = lane >> 4;
    const bool rvA = (j0A + i) < n, rvB = j0B >= 0 && (j0B + i) < n;
    const int offA = (rvA ? j0A + i : 0) * n + 4 * kk, offB = (rvB ? j0B + i : 0) * n + 4 * kk;
    mg_lds_cf* xp = Xs + i;
    const int ntA = (KA + 15) >> 4, ntB = j0B >= 0 ? (KB + 15) >> 4 : 0;
    const int SA = (ntA + 3) >> 2, SB = (ntB + 3) >> 2, S = SA + SB;
    vjf_f32x4 acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = vjf_f32x4{0.f, 0.f, 0.f, 0.f};
    auto fold = [&]() {
#pragma unroll
        for (int g = 0; g < NG; ++g) v2[g] = fmaf(acc[g][0], acc[g][0], fmaf(acc[g][1], acc[g][1], fmaf(acc[g][2], acc[g][2], fmaf(acc[g][3], acc[g][3], v2[g]))));
    };
    auto ldb = [&](float4 (&a)[4], int sb) {
        const bool inB = sb >= SA;
        const int t0 = 4 * (inB ? sb - SA : sb), off = inB ? offB : offA;
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int kq = 16 * (t0 + q) + 4 * kk; a[q] = mg_ld4_plain(rx, off + (kq + 3 < n ? 16 * (t0 + q) : 0)); }   // (plain: L^-1 of a launch without an RLS update is written once, before its first read)
    };
    auto mmb = [&](const float4 (&a)[4], int sb) {
        const bool inB = sb >= SA;
        const int t0 = 4 * (inB ? sb - SA : sb), nt = inB ? ntB : ntA, ke = inB ? KB : KA;
        const bool rv = inB ? rvB : rvA;
        if (sb == SA && SA > 0) {
            fold();
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = vjf_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (t0 + q < nt) {
                const int k0 = 16 * (t0 + q) + 4 * kk;
                const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
                // the four k-steps' B operands first, then their multiply-adds: ONE LDS round trip per block of 16 k instead of one per
                // k-step (the ISA of the trial role's mg_var2 waits on lgkmcnt in front of nearly every pair of MFMAs)
                float bv[4][NG];
                float vv[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int k = k0 + c, kc = min(k, ke - 1);
                    vv[c] = (rv && k < ke) ? av[c] : 0.f;
#pragma unroll
                    for (int g = 0; g < (NG < 2 ? NG : 2); ++g) bv[c][g] = xp[kc * LD + 16 * g];
                    if (NG > 2 && upper) {
#pragma unroll
                        for (int g = 2; g < NG; ++g) bv[c][g] = xp[kc * LD + 16 * g];
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
#pragma unroll
                    for (int g = 0; g < (NG < 2 ? NG : 2); ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[c], bv[c][g], acc[g], 0, 0, 0);
                    if (NG > 2 && upper) {
#pragma unroll
                        for (int g = 2; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[c], bv[c][g], acc[g], 0, 0, 0);
                    }
                }
            }
        }
    };
    float4 a0[4], a1[4], a2[4];
    ldb(a0, 0);
    if (S > 1) ldb(a1, 1);
    if (S > 2) ldb(a2, 2);
    for (int sb = 0; sb < S; sb += 3) {
        mmb(a0, sb);
        if (sb + 3 < S) ldb(a0, sb + 3);
        if (sb + 1 < S) { mmb(a1, sb + 1); if (sb + 4 < S) ldb(a1, sb + 4); }
        if (sb + 2 < S) { mmb(a2, sb + 2); if (sb + 5 < S) ldb(a2, sb + 5); }
    }
    fold();
}
template <int NG, int LD>
__device__ __forceinline__ void mg_mmaN_mm16(vjf_f32x4 (&acc)[NG], const float (&a)[16], const float* Xs, int M, int m0, int kb, int ke, int s0, int lane,
                                             const bool upper = true) {
    const int i = lane & 15, kk = lane >> 4;
    const bool rv = (m0 + i) < M;
    const float* xp = Xs + i;
    const int nst = (ke - kb + 3) >> 2, klast = ke - 1;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (s0 + q < nst) {
            const int k = kb + 4 * (s0 + q) + kk;
            const int kc = min(k, klast);
            const float av = (rv && k < ke) ? a[q] : 0.f;
#pragma unroll
            for (int g = 0; g < (NG < 2 ? NG : 2); ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD + 16 * g], acc[g], 0, 0, 0);
            if (NG > 2 && upper) {
#pragma unroll
                for (int g = 2; g < NG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD + 16 * g], acc[g], 0, 0, 0);
            }
        }
    }
}

// one pass of the moments role: NG / 2 tiles (tile0, and tile1 when NG = 4 and tile1 >= 0) of step t, from their posterior tags to their
// moments tags
template <int NG>
__device__ __forceinline__ bool mg_moments_pass(const VjfPlan& P, const VjfMegaArgs& A, float* smem, const int t, const int tile0, const int tile1, const bool tri) {
    constexpr int LD = 16 * NG + 1, NW = VJF_MG_WAVES, NT = VJF_MG_THREADS, TR = VJF_MG_TR, NC = 16 * NG;   // NC columns = NG / 2 tiles
    const bool two = NG == 4 && tile1 >= 0;
    const int ncol = two ? NC : TR;                    // columns that hold trials (the elementwise loops stop there)
    const int tid0 = threadIdx.x;
    int tid = tid0, lane, wave;
    MG_PHASE();
    const int dz = P.dz, du = P.du, n = P.n, dxu = P.dxu, npad = (n + 3) & ~3;
    const float* S = A.state;
    float* SCW = A.state + P.off[VJF_SLOT_SCALARS];
    float* s_cen = smem; float* s_iw = s_cen + (size_t)npad * dxu;
    float* s_xu = s_iw + npad; float* s_phi = s_xu + (size_t)dxu * LD;
    float* s_red = s_phi + (size_t)n * LD; float* s_part = s_red + NW * NC;
    unsigned* cnt = A.cnt;
    const size_t sz = (size_t)A.B * dz, su = (size_t)A.B * du;
    constexpr int part_rows = VJF_MG_WAVES * 16;
    const float* mu_s = t ? A.mu + (size_t)(t - 1) * sz : A.mu0;
    const float* lv_s = t ? A.lv + (size_t)(t - 1) * sz : A.lv0;
    const float* eps_s = A.eps + (size_t)t * 2 * sz;
    const float* u_t = A.u ? A.u + (size_t)t * su : nullptr;
    // the tiles' posterior of step t - 1 (the trial role's write-through stores, then its tags)
    if (t > 0) {
        bool ok = vjf_wg_wait_sc1<VJF_POLL_SLEEP_LITE>(cnt + MG_C_TAG_POST + tile0, (unsigned)t, tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u);
        bool gone = vjf_abort_wg();
        if (two && !gone) { ok = vjf_wg_wait_sc1<VJF_POLL_SLEEP_LITE>(cnt + MG_C_TAG_POST + tile1, (unsigned)t, tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u) && ok; gone = vjf_abort_wg(); }
        if (!ok) vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE2);
        if (gone) return false;
    } else __syncthreads();
    MG_PHASE();
    const int wg = tile0;                              // (diagnostic stamps: the workgroup that owns tile 0)
    VJF_MG_STAMP(11);
    // xs = mu + eps e^{lv / 2} (util.py:11-13; the prior at the first step of a run) and the inputs u: the trial role's expression
    for (int e = tid; e < NC * dxu; e += NT) {
        const int c = e / NC, col = e - c * NC, b = col & 31;
        if (col >= ncol) { s_xu[c * LD + col] = 0.f; continue; }
        const int b0 = (col < TR ? tile0 : tile1) * TR, nb = min(TR, A.B - b0);
        float v = 0.f;
        if (c < dz) {
            float m, l, ep = 0.f;
            if (mu_s) { m = b < nb ? mg_ld(mu_s + (size_t)(b0 + b) * dz + c) : 0.f; l = b < nb ? mg_ld(lv_s + (size_t)(b0 + b) * dz + c) : 0.f; }
            else { m = S[P.off[VJF_SLOT_PRIOR_MEAN] + c]; l = S[P.off[VJF_SLOT_PRIOR_LOGVAR] + c]; }
            if (b < nb) ep = eps_s[(size_t)(b0 + b) * dz + c];
            v = fmaf(ep, expf(0.5f * l), m);
        } else if (b < nb) v = u_t[(size_t)(b0 + b) * du + c - dz];
        s_xu[c * LD + col] = v;
    }
    __syncthreads(); MG_PHASE();
    VJF_MG_STAMP(12);
    // RBF features (functional.py:11-22): four centres per thread and column (one 16-byte LDS read of the centres per input dimension
    // instead of four 4-byte ones; per element the trial role's operations in the trial role's order: the same bits)
    for (int e = tid; e < NC * (npad >> 2); e += NT) {
        const int k4 = e / NC, col = e - k4 * NC, k = 4 * k4;
        float d2[4] = {0.f, 0.f, 0.f, 0.f};
        if (col < ncol) {
            auto dim = [&](float x, const float4& cc) {                        // (one input dimension: the trial role's order of operations)
                float d;
                d = x - cc.x; d2[0] = fmaf(d, d, d2[0]); d = x - cc.y; d2[1] = fmaf(d, d, d2[1]);
                d = x - cc.z; d2[2] = fmaf(d, d, d2[2]); d = x - cc.w; d2[3] = fmaf(d, d, d2[3]);
            };
            int c = 0;
            for (; c + 3 < dxu; c += 4) {                                      // four dimensions' LDS loads in flight together (a loop of
                const float x0 = s_xu[c * LD + col], x1 = s_xu[(c + 1) * LD + col], x2 = s_xu[(c + 2) * LD + col], x3 = s_xu[(c + 3) * LD + col];   // single loads is a chain of LDS round trips)
                const float4 c0 = *reinterpret_cast<const float4*>(s_cen + c * npad + k);
                const float4 c1 = *reinterpret_cast<const float4*>(s_cen + (c + 1) * npad + k);
                const float4 c2 = *reinterpret_cast<const float4*>(s_cen + (c + 2) * npad + k);
                const float4 c3 = *reinterpret_cast<const float4*>(s_cen + (c + 3) * npad + k);
                dim(x0, c0); dim(x1, c1); dim(x2, c2); dim(x3, c3);
            }
            for (; c < dxu; ++c) dim(s_xu[c * LD + col], *reinterpret_cast<const float4*>(s_cen + c * npad + k));
        }
        const float4 iw = *reinterpret_cast<const float4*>(s_iw + k);
        const float iwv[4] = {iw.x, iw.y, iw.z, iw.w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (k + q < n) s_phi[(k + q) * LD + col] = col < ncol ? expf(d2[q] * iwv[q]) : 0.f;
    }
    __syncthreads(); MG_PHASE();
    VJF_MG_STAMP(13);
    int mean_nsl = 1;
    {   // predictive variance and mean: vjf_mega_trial's stage 2, wavefront for wavefront
        const __amdgpu_buffer_rsrc_t r_xt = mg_rsrc(A.xt);
        const float* Wm = S + P.off[VJF_SLOT_W_MEAN];
        const int ntile = (n + 15) >> 4;
        float v2[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) v2[g] = 0.f;
        const int nsl = min(NW, part_rows / 16);
        const int msl = nsl - 1 - wave;
        const int mper = (((n + 3) >> 2) + nsl - 1) / nsl * 4;
        const int mkb = msl * mper, mke = min(n, (msl + 1) * mper);
        const bool mpre = wave < nsl && ((mke - mkb + 3) >> 2) <= 16;
        float am[16];
        if (mpre && mke > mkb) mg_mma2_ld16(am, Wm, dz, dz, 0, mkb, mke, 0, lane);
        for (int r = 0; r * NW < ntile; r += 2) {
            int j0p[2], Kp[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = r + h, idx = (rr & 1) ? rr * NW + NW - 1 - wave : rr * NW + wave;
                const int tt = ntile - 1 - idx;
                j0p[h] = (idx < ntile) ? tt * 16 : -1;
                Kp[h] = tri ? min(n, tt * 16 + 16) : n;
            }
            if (j0p[0] < 0) { j0p[0] = j0p[1]; Kp[0] = Kp[1]; j0p[1] = -1; }
            mg_varN<NG, LD>(v2, r_xt, n, j0p[0], Kp[0], j0p[1], Kp[1], (mg_lds_cf*)s_phi, lane, two);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            v2[g] += __shfl_xor(v2[g], 16, 64); v2[g] += __shfl_xor(v2[g], 32, 64);
            if (lane < 16) s_red[wave * NC + 16 * g + lane] = v2[g];
        }
        if (wave < nsl) {
            vjf_f32x4 acc[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g] = vjf_f32x4{0.f, 0.f, 0.f, 0.f};
            if (mpre) { if (mke > mkb) mg_mmaN_mm16<NG, LD>(acc, am, s_phi, dz, 0, mkb, mke, 0, lane, two); }
            else {
                // (more than 512 features: the slice in batches of 16 k-steps)
                for (int s0 = 0; 4 * s0 < mke - mkb; s0 += 16) {
                    float a2[16];
                    mg_mma2_ld16(a2, Wm, dz, dz, 0, mkb, mke, s0, lane);
                    mg_mmaN_mm16<NG, LD>(acc, a2, s_phi, dz, 0, mkb, mke, s0, lane, two);
                }
            }
            float* pr = s_part + (size_t)(msl * 16 + 4 * (lane >> 4)) * LD + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int g = 0; g < NG; ++g) pr[r * LD + 16 * g] = acc[g][r];
        }
        mean_nsl = nsl;
    }
    __syncthreads(); MG_PHASE();
    VJF_MG_STAMP(14);
    // out, per tile: [pt.mean (dz x 32) | Phi W (dz x 32) | pt.logvar (32)], write-through; then the tags
    const int mlen = (2 * dz + 1) * TR;
    for (int e = tid; e < NC * dz; e += NT) {
        const int j = e / NC, col = e - j * NC, b = col & 31;
        if (col >= ncol) continue;
        float* mb = A.mom + ((size_t)(col < TR ? tile0 : tile1) * 2 + (size_t)(t & 1)) * (size_t)mlen;
        float v = 0.f;
        for (int sl = 0; sl < mean_nsl; ++sl) v += s_part[(size_t)(sl * 16 + j) * LD + col];
        mg_st(mb + j * TR + b, s_xu[j * LD + col] + v);
        mg_st(mb + TR * dz + j * TR + b, v);
    }
    if (tid < ncol) {
        float* mb = A.mom + ((size_t)(tid < TR ? tile0 : tile1) * 2 + (size_t)(t & 1)) * (size_t)mlen;
        float v = 0.f;
        for (int w = 0; w < NW; ++w) v += s_red[w * NC + tid];
        mg_st(mb + 2 * TR * dz + (tid & 31), logf(v));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(cnt + MG_C_TAG_MOM + tile0, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (two) __hip_atomic_store(cnt + MG_C_TAG_MOM + tile1, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    VJF_MG_STAMP(15);
    return true;
}

__device__ __forceinline__ void vjf_mega_moments(const VjfPlan& P, const VjfMegaArgs& A, float* smem, const int mw) {
    constexpr int NT = VJF_MG_THREADS;
    const int tid0 = threadIdx.x;
    const int n = P.n, dxu = P.dxu, npad = (n + 3) & ~3;
    const float* S = A.state;
    float* SCW = A.state + P.off[VJF_SLOT_SCALARS];
    float* s_cen = smem; float* s_iw = s_cen + (size_t)npad * dxu;
    {
        const float* cen = S + P.off[VJF_SLOT_CENTROID];
        const float* lw = S + P.off[VJF_SLOT_LOGWIDTH];
        for (int e = tid0; e < npad * dxu; e += NT) { const int c = e / npad, k = e - c * npad; s_cen[e] = k < n ? cen[k * dxu + c] : 0.f; }
        for (int e = tid0; e < npad; e += NT) { float v = 0.f; if (e < n) { const float w = expf(lw[e]); v = -0.5f / (w * w); } s_iw[e] = v; }
    }
    // (the row-major L^-1 of this launch: the trial workgroups' first act)
    if (!vjf_wg_wait_sc1<VJF_POLL_SLEEP_LITE>(A.cnt + MG_C_XT, (unsigned)A.n_trial, tid0, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u))
        vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_K1);
    if (vjf_abort_wg()) return;
    // (w_chol upper triangular: the state's flag, or what the trial workgroups saw while they transposed it)
    const bool tri = mg_ld(SCW + VJF_SC_TRI_CLEAN) != 0.f || __hip_atomic_load(A.cnt + MG_C_XT + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
    for (int t = 0; t < A.T; ++t) {
        // this workgroup's tiles mw, mw + n_mom, ..: two at a time side by side (every operand load of L^-1 serves both), a last one alone
        // (ONE instantiation of the pass, four column groups, for both cases -- tile1 < 0: the second half idles.  With a two-group
        //  instantiation beside it hipcc (ROCm 7.2.0) fails in its backend: "Illegal instruction detected ... $src_shared_base",
        //  DESIGN.md section 3 "Toolchain note"; either instantiation alone compiles)
        int tile = mw;
        if (A.n_mom >= A.ntiles) {                       // (uniform) a workgroup per tile: the one-tile layout, the trial role's own
            if (tile < A.ntiles && !mg_moments_pass<2>(P, A, smem, t, tile, -1, tri)) return;
            continue;
        }
        for (; tile < A.ntiles; tile += 2 * A.n_mom)
            if (!mg_moments_pass<4>(P, A, smem, t, tile, tile + A.n_mom < A.ntiles ? tile + A.n_mom : -1, tri)) return;
    }
}

// ------------------------------------------------------------------------------------------------ Gram role
// Phi^T Phi of event e (the features of step e), lower 32x32 tiles.  The rows of Phi are formed here, from the posterior of step
// e - 1 and the noise of step e -- operation for operation what the trial role does for its own tile (stages 0 / 1), so the two
// hold the same bits -- as soon as every trial workgroup has its forward pass of step e - 1 behind it: a step AHEAD of the RLS
// update that consumes the sum.
__device__ __forceinline__ void vjf_mega_gram(const VjfPlan& P, const VjfMegaArgs& A, float* lds, const int hg) {
    constexpr int NT = VJF_MG_THREADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.n, nbl = (n + 31) / 32, ntri = nbl * (nbl + 1) / 2, ldE = P.ldE;
    const unsigned m_l4 = mg_magic(ldE >> 2);
    float* SCW = A.state + P.off[VJF_SLOT_SCALARS];
    const int dz = P.dz, du = P.du, dxu = P.dxu, npad = (n + 3) & ~3;
    float* s_rows = lds;                               // [VJF_MG_GROWS][ldE]
    int* s_tab = reinterpret_cast<int*>(s_rows + (size_t)VJF_MG_GROWS * ldE);   // tile -> (bi << 8) | bj
    float* s_cen = s_rows + (size_t)VJF_MG_GROWS * ldE + 64;                    // [dxu][npad]
    float* s_iw = s_cen + (size_t)npad * dxu;                                   // [npad]
    float* s_x = s_iw + npad;                                                   // [VJF_MG_GROWS][dxu]
    const float* S = A.state;
    {
        const float* cen = S + P.off[VJF_SLOT_CENTROID];
        const float* lw = S + P.off[VJF_SLOT_LOGWIDTH];
        for (int e = tid; e < npad * dxu; e += NT) { const int c = e / npad, k = e - c * npad; s_cen[e] = k < n ? cen[k * dxu + c] : 0.f; }
        for (int e = tid; e < npad; e += NT) { float v = 0.f; if (e < n) { const float w = expf(lw[e]); v = -0.5f / (w * w); } s_iw[e] = v; }
    }
    const size_t sz = (size_t)A.B * dz, su = (size_t)A.B * du;
    const unsigned m_dxu = mg_magic(dxu);
    if (tid < ntri) {
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= tid) ++bi;
        s_tab[tid] = (bi << 8) | (tid - bi * (bi + 1) / 2);
    }
    __syncthreads();
    const int r0 = hg * A.gram_rows, r1 = min(A.B, r0 + A.gram_rows);
    float* myslab = A.gslab + (size_t)hg * ntri * 1024;
    const __amdgpu_buffer_rsrc_t r_gslab = mg_rsrc(A.gslab);
    const int c = lane & 31, kh = lane >> 5;
    for (int e = 0; e < A.T; ++e) {
        float* red = (e & 1) ? A.red1 : A.red0;
        const float* mu_s = e ? A.mu + (size_t)(e - 1) * sz : A.mu0;
        const float* lv_s = e ? A.lv + (size_t)(e - 1) * sz : A.lv0;
        const float* eps_s = A.eps + (size_t)e * 2 * sz;
        const float* u_e = A.u ? A.u + (size_t)e * su : nullptr;
        // (the posterior of step e - 1: write-through stores of the trial role, in memory before its early slab's signal)
        // (the slab of the previous event: every Gram workgroup has summed its share -- nothing is read behind this one: no acquire)
        if (e > 0 && !vjf_wg_wait_sc1(A.cnt + MG_C_STAT, (unsigned)e * (unsigned)A.n_gram, tid, SCW + VJF_SC_STATUS))
            vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE2);
        if (e > 0 && !vjf_wg_wait_sc1(A.cnt + MG_C_FWD, (unsigned)e * (unsigned)A.n_trial, tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u))
            vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE2);
        if (e > 0 && vjf_abort_wg()) return;
        { const int wg = hg, t = e; VJF_MG_STAMP(11); }
        vjf_f32x16 acc[VJF_MG_MAXQ];
#pragma unroll
        for (int q = 0; q < VJF_MG_MAXQ; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
        for (int c0 = r0; c0 < r1; c0 += VJF_MG_GROWS) {
            __syncthreads();
            const int l4 = ldE >> 2;
            const int ks = min(VJF_MG_GROWS / 2, (((min(VJF_MG_GROWS, r1 - c0) + 1) >> 1) + 1) & ~1);   // (a multiple of 2; rows beyond the range are zero)
            // xs = mu + eps e^{lv/2} (model.py:97-99; the prior at the first step of a run: model.py:188-190) and the inputs u
            for (int i = tid; i < VJF_MG_GROWS * dxu; i += NT) {
                const int r = mg_div(i, m_dxu), c2 = i - r * dxu, b = c0 + r;
                float v = 0.f;
                if (b < r1) {
                    if (c2 < dz) {
                        const float m = mu_s ? mg_ld(mu_s + (size_t)b * dz + c2) : S[P.off[VJF_SLOT_PRIOR_MEAN] + c2];   // (sc1: no acquire
                        const float l = mu_s ? mg_ld(lv_s + (size_t)b * dz + c2) : S[P.off[VJF_SLOT_PRIOR_LOGVAR] + c2]; //  behind the waits)
                        v = fmaf(eps_s[(size_t)b * dz + c2], expf(0.5f * l), m);
                    } else {
                        v = u_e[(size_t)b * du + c2 - dz];
                    }
                }
                s_x[i] = v;
            }
            __syncthreads();
            // RBF features (functional.py:11-22), four centres per thread and pass; rows beyond the range and columns beyond n: zero
            for (int i = tid; i < VJF_MG_GROWS * l4; i += NT) {
                const int r = mg_div(i, m_l4), k = (i - r * l4) * 4;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c0 + r < r1 && k < npad) {
                    float d2[4] = {0.f, 0.f, 0.f, 0.f};
                    auto dim = [&](float x, const float4& cc) {                     // (one input dimension: the trial role's order of operations)
                        float d;
                        d = x - cc.x; d2[0] = fmaf(d, d, d2[0]); d = x - cc.y; d2[1] = fmaf(d, d, d2[1]);
                        d = x - cc.z; d2[2] = fmaf(d, d, d2[2]); d = x - cc.w; d2[3] = fmaf(d, d, d2[3]);
                    };
                    int c2 = 0;
                    for (; c2 + 3 < dxu; c2 += 4) {                                  // four dimensions' LDS loads in flight together
                        const float x0 = s_x[r * dxu + c2], x1 = s_x[r * dxu + c2 + 1], x2 = s_x[r * dxu + c2 + 2], x3 = s_x[r * dxu + c2 + 3];
                        const float4 c0v = *reinterpret_cast<const float4*>(s_cen + c2 * npad + k);
                        const float4 c1v = *reinterpret_cast<const float4*>(s_cen + (c2 + 1) * npad + k);
                        const float4 c2v = *reinterpret_cast<const float4*>(s_cen + (c2 + 2) * npad + k);
                        const float4 c3v = *reinterpret_cast<const float4*>(s_cen + (c2 + 3) * npad + k);
                        dim(x0, c0v); dim(x1, c1v); dim(x2, c2v); dim(x3, c3v);
                    }
                    for (; c2 < dxu; ++c2) dim(s_x[r * dxu + c2], *reinterpret_cast<const float4*>(s_cen + c2 * npad + k));
                    const float4 iw = *reinterpret_cast<const float4*>(s_iw + k);
                    o.x = expf(d2[0] * iw.x); o.y = k + 1 < n ? expf(d2[1] * iw.y) : 0.f;
                    o.z = k + 2 < n ? expf(d2[2] * iw.z) : 0.f; o.w = k + 3 < n ? expf(d2[3] * iw.w) : 0.f;
                }
                *reinterpret_cast<float4*>(s_rows + (size_t)r * ldE + k) = o;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < VJF_MG_MAXQ; ++q) {
                const int tt = wave + VJF_MG_WAVES * q;
                if (tt < ntri) {
                    const int code = s_tab[tt], bi = code >> 8, bj = code & 255;
                    const float* pa = s_rows + (size_t)kh * ldE + bi * 32 + c;
                    const float* pb = s_rows + (size_t)kh * ldE + bj * 32 + c;
                    int s = 0;
#pragma unroll 2
                    for (; s + 8 <= ks; s += 8) {                                  // ks = k-steps (row pairs) of this pass that hold rows
                        float a[8], b[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) { a[u] = pa[(size_t)(2 * (s + u)) * ldE]; b[u] = pb[(size_t)(2 * (s + u)) * ldE]; }
#pragma unroll
                        for (int u = 0; u < 8; ++u) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[q], 0, 0, 0);
                    }
                    for (; s < ks; s += 2) {
                        float a[2], b[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) { a[u] = pa[(size_t)(2 * (s + u)) * ldE]; b[u] = pb[(size_t)(2 * (s + u)) * ldE]; }
#pragma unroll
                        for (int u = 0; u < 2; ++u) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc[q], 0, 0, 0);
                    }
                    // the last pass of rows: the tile is final and leaves at once, 16-byte write-through stores, beside the next tile's
                    // multiply-adds.  Accumulator: column = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); slab element
                    // ((j*64 + lane)*4 + r) = register 4 j + r of that lane
                    if (c0 + VJF_MG_GROWS >= r1) {
                        float* sl = myslab + (size_t)tt * 1024;
#pragma unroll
                        for (int j = 0; j < 4; ++j) mg_st4(sl + (j * 64 + lane) * 4, acc[q][4 * j], acc[q][4 * j + 1], acc[q][4 * j + 2], acc[q][4 * j + 3]);
                    }
                }
            }
        }
        if (r0 >= r1) {                                                        // (a workgroup without rows: its slab is zeros)
#pragma unroll
            for (int q = 0; q < VJF_MG_MAXQ; ++q) {
                const int tt = wave + VJF_MG_WAVES * q;
                if (tt < ntri) {
                    float* sl = myslab + (size_t)tt * 1024;
#pragma unroll
                    for (int j = 0; j < 4; ++j) mg_st4(sl + (j * 64 + lane) * 4, 0.f, 0.f, 0.f, 0.f);
                }
            }
        }
        vjf_wg_signal_wt(A.cnt + MG_C_GRAM, tid);
        { const int wg = hg, t = e; VJF_MG_STAMP(12); }
        // The sums of event e go where those of event e - 2 are: the RLS update of step e - 2 must be through with them (the Cholesky
        // loop's operand load, the operand role's P += G / v, the y / W loop's tiles for the state-noise update).  The trial role's
        // forward half of step e - 1, which is all this event waited for, does not wait for that update: without this wait a late
        // RLS role -- the first steps of a process, instruction caches cold -- read sums of the wrong step.  Nothing is read behind it.
#ifndef VJF_CHAOS_OMIT_GRAM_GUARD        /* (diagnostic builds: without the wait tools/chaos_handoffs.py must report deviations) */
        if (e >= 2 && !vjf_wg_wait_sc1(A.cnt + MG_C_PDONE, (unsigned)(e - 1) * (unsigned)(A.n_rls - 1), tid, SCW + VJF_SC_STATUS))
            vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE2);
#endif
        if (!vjf_wg_wait_sc1(A.cnt + MG_C_GRAM, (unsigned)(e + 1) * (unsigned)A.n_gram, tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u))
            vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE2);
        if (vjf_abort_wg()) return;
        // this workgroup's share of the sum over the slabs: a quad of elements per 4 lanes, lane p sums the slabs [p npq, (p+1) npq)
        // (all of them in flight -- for TWO quads at a time: one round trip for the whole share at config B), then
        // (s0 + s1) + (s2 + s3): a fixed order
        {
            const int npq = (A.n_gram + 3) >> 2;
            const int part = tid & 3;
            const int qstride = (A.n_gram * NT) >> 2, nq = ntri * 256;
            const int h1 = min(A.n_gram, (part + 1) * npq);
            auto load16 = [&](float4 (&tq)[16], int quad, int h0) {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    tq[q] = (quad < nq && h0 + q < h1) ? mg_ld4(r_gslab, quad * 4 + (h0 + q) * ntri * 1024) : make_float4(0.f, 0.f, 0.f, 0.f);
            };
            auto add16 = [&](float4& v, const float4 (&tq)[16]) {
#pragma unroll
                for (int q = 0; q < 16; ++q) { v.x += tq[q].x; v.y += tq[q].y; v.z += tq[q].z; v.w += tq[q].w; }
            };
            auto finish = [&](int quad, float4 v) {
                float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vv[r] += __shfl_xor(vv[r], 1, 64);
                    vv[r] += __shfl_xor(vv[r], 2, 64);
                }
                // Four consecutive quads (16 lanes) hold rows gr0 .. gr0 + 3 of four consecutive columns: a 4 x 4 transpose through
                // shuffles gives every one of them a ROW segment too, so that both triangles leave as 16-byte write-through stores
                // (the scalar form was 4 fabric writes per quad: 29 k per step, and the Cholesky loop waits for this sum).  A
                // diagonal block is written in full from both sides: its (i, j) and (j, i) sums are the same bits.
                const int a4 = (lane >> 2) & 3, lb = lane & ~15;
                float o[4];
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2) {
                    const float t0 = __shfl(vv[0], lb + 4 * b2, 64), t1 = __shfl(vv[1], lb + 4 * b2, 64);
                    const float t2 = __shfl(vv[2], lb + 4 * b2, 64), t3 = __shfl(vv[3], lb + 4 * b2, 64);
                    o[b2] = a4 == 0 ? t0 : a4 == 1 ? t1 : a4 == 2 ? t2 : t3;
                }
                if (part == 0 && quad < nq) {
                    const int idx = quad * 4, tt = idx >> 10, el = idx & 1023, code = s_tab[tt];
                    const int j = el >> 8, ln = (el >> 2) & 63;
                    const int gc = (code & 255) * 32 + (ln & 31);
                    const int gr0 = (code >> 8) * 32 + 8 * j + 4 * (ln >> 5);            // the quad: rows gr0 .. gr0 + 3 of column gc
                    if (gr0 + 3 < n && gc < n) {
                        mg_st4(red + P.red_G + (size_t)gc * n + gr0, vv[0], vv[1], vv[2], vv[3]);                // row gc, columns gr0 .. gr0 + 3
                        mg_st4(red + P.red_G + (size_t)(gr0 + a4) * n + (gc - a4), o[0], o[1], o[2], o[3]);        // row gr0 + a4, columns gc - a4 .. + 3
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int gr = gr0 + r;
                            if (gr < n && gc < n && ((code >> 8) != (code & 255) || gc <= gr)) {
                                mg_st(red + P.red_G + (size_t)gr * n + gc, vv[r]);
                                mg_st(red + P.red_G + (size_t)gc * n + gr, vv[r]);
                            }
                        }
                    }
                }
            };
            for (int quad = (hg * NT + tid) >> 2; quad < nq + qstride; quad += 2 * qstride) {   // (uniform trip count over the wavefront: shuffles inside)
                const int quadB = quad + qstride;
                float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vb = va;
                float4 ta[16], tb[16];                                               // (at most 64 Gram workgroups: npq <= 16, one batch per lane)
                load16(ta, quad, part * npq); load16(tb, quadB, part * npq);
                add16(va, ta); add16(vb, tb);
                finish(quad, va); finish(quadB, vb);
            }
        }
        vjf_wg_signal_wt(A.cnt + MG_C_STAT, tid);
        { const int wg = hg, t = e; VJF_MG_STAMP(13); }
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
    // (every byte taken from other roles is read with sc1 loads behind the counts' polls and the workgroup barrier: no acquires)
    const __amdgpu_buffer_rsrc_t r_early = mg_rsrc(A.slab_early);
    for (int t = 0; t < A.T; ++t) {
        float* red = (t & 1) ? A.red1 : A.red0;
        // (four counts, ONE acquire: behind the last of them)
        bool ok = vjf_wg_wait_sc1(A.cnt + MG_C_FWD, (unsigned)(t + 1) * (unsigned)A.n_trial, tid, SCW + VJF_SC_STATUS);
        ok = vjf_wg_wait_sc1(A.cnt + MG_C_STAT, (unsigned)(t + 1) * (unsigned)A.n_gram, tid, SCW + VJF_SC_STATUS) && ok;
        if (t > 0) ok = vjf_wg_wait_sc1(A.cnt + MG_C_PDONE, (unsigned)t * npost, tid, SCW + VJF_SC_STATUS) && ok;
        ok = vjf_wg_wait_sc1(runw, (unsigned)(t + 1), tid, SCW + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u) && ok;      // the Cholesky loop holds its operands (it reads the state's P at step 0)
        if (tid == 0 && !ok) { vjf_status_or(SCW + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_OPERAND); vjf_s_abort_word = 1; }
        __syncthreads();                                                       // (the verdict of lane 0, for every thread alike)
        if (vjf_abort_wg()) return;
        { const int wg = pw; VJF_MG_STAMP(14); }
        // Phi^T dx rows i0 .. i0 + 15 (16 columns x 4 quads of features: the early slabs hold it transposed): 8 lanes per quad, lane p
        // sums the early slabs [p npq, (p+1) npq) (all in flight), then a fixed xor tree
        {
            const int ldn = (n + 3) & ~3;
            const float* base = A.slab_early + (size_t)(t & 1) * A.n_trial * A.early_len;
            const int npq = (A.n_trial + 7) >> 3, part = tid & 7, quad = tid >> 3;      // quad = column * 4 + feature quad
            const int c = quad >> 2, r4 = (quad & 3) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < dz && i0 + r4 < ldn) {
                const int src = (int)(base - A.slab_early) + c * ldn + i0 + r4;
                const int w1 = min(A.n_trial, (part + 1) * npq);
                for (int w0 = part * npq; w0 < w1; w0 += 16) {
                    float4 tq[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        tq[q] = (w0 + q < w1) ? mg_ld4(r_early, src + (w0 + q) * A.early_len) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int q = 0; q < 16; ++q) { v.x += tq[q].x; v.y += tq[q].y; v.z += tq[q].z; v.w += tq[q].w; }
                }
            }
            float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                vv[q] += __shfl_xor(vv[q], 1, 64);
                vv[q] += __shfl_xor(vv[q], 2, 64);
                vv[q] += __shfl_xor(vv[q], 4, 64);
            }
            if (part == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    s_f[(r4 + q) * 17 + c] = vv[q];
                    if (i0 + r4 + q < n && c < dz) mg_st(red + P.red_FDX + (size_t)(i0 + r4 + q) * dz + c, vv[q]);
                }
            }
            if (pw == 0 && tid < 64) {                                             // sum |dx|^2: one wavefront, strided partial sums, xor tree
                float q2 = 0.f;
                for (int w = tid; w < A.n_trial; w += 64) q2 += mg_ld(base + (size_t)w * A.early_len + (size_t)16 * ldn + RS_SDX2);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o, 64);
                if (tid == 0) mg_st(red + P.red_SC + RS_SDX2, q2);
            }
        }
        const float inv_v = expf(-mg_ld(S + P.off[VJF_SLOT_TR_LOGVAR]));
        float* Pm = S + P.off[VJF_SLOT_W_PREC];
        const float* Wm = S + P.off[VJF_SLOT_W_MEAN];
        const float* G = red + P.red_G;
        const __amdgpu_buffer_rsrc_t r_P = mg_rsrc(Pm), r_G = mg_rsrc(G);
        const int n4 = n >> 2;
        const unsigned m_n4 = mg_magic(n4);
        for (int e0 = tid; e0 < 16 * n4; e0 += 4 * NT) {
            float4 p[4], g[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * NT, row = mg_div(e, m_n4), c4 = (e - row * n4) * 4;
                const bool in = e < 16 * n4 && i0 + row < n;
                const size_t off = in ? (size_t)(i0 + row) * n + c4 : 0;
                p[q] = mg_ld4(r_P, (int)off);                                  // (P: this workgroup's own rows -- and the y / W loop's after a failed factorisation)
                g[q] = mg_ld4(r_G, (int)off);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = e0 + q * NT, row = mg_div(e, m_n4), c4 = (e - row * n4) * 4;
                if (e >= 16 * n4) continue;
                const bool in = i0 + row < n;
                float* d = s_p + row * ldp + c4;
                d[0] = in ? p[q].x : 0.f; d[1] = in ? p[q].y : 0.f; d[2] = in ? p[q].z : 0.f; d[3] = in ? p[q].w : 0.f;
                if (in) mg_st4(Pm + (size_t)(i0 + row) * n + c4, fmaf(g[q].x, inv_v, p[q].x), fmaf(g[q].y, inv_v, p[q].y),
                               fmaf(g[q].z, inv_v, p[q].z), fmaf(g[q].w, inv_v, p[q].w));
            }
        }
        for (int e = tid; e < n * 16; e += NT) {
            const int k = e >> 4, cc = e & 15;
            s_w[k * 17 + cc] = cc < dz ? mg_ld(Wm + (size_t)k * dz + cc) : 0.f;
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
        { const int wg = pw; VJF_MG_STAMP(15); }
    }
}

// ------------------------------------------------------------------------------------------------ SGD role
template <bool RLS>
__device__ __forceinline__ void vjf_mega_sgd(const VjfPlan& P, const VjfMegaArgs& A, float* lds, const int sw) {
    constexpr int NT = VJF_MG_THREADS;
    const int tid = threadIdx.x;
    float* s_sc = lds;                                 // RS_N loss sums
    float* S = A.state;
    float* SC = S + P.off[VJF_SLOT_SCALARS];
    const float Bf = (float)A.B, invB = 1.0f / Bf;
    // (set by the host between launches, never inside one)
    const float lr_dec = SC[VJF_SC_LR_DEC], lr_rec = SC[VJF_SC_LR_REC];
    const bool freeze = SC[VJF_SC_FREEZE_DEC] != 0.f;
    const bool tl = vjf_mega_trial_lds<false>(P, A.lds_floats).theta != 0;   // the trial role reads the LDS image (else: the state and its transposed copies)
    // the flags of VJF.filter for the steps of this launch (vjf/model.py:179-221; see vjf_mega_trial)
    const bool do_sgd = RLS || (A.flags & VJF_FLAG_SGD) != 0u, do_upd = RLS || (A.flags & VJF_FLAG_UPDATE) != 0u;
    const bool warm = !RLS && (A.flags & VJF_FLAG_WARM_UP) != 0u;
    constexpr bool mode_rls = RLS;
    const int n_live = RLS ? A.n_sgd : A.n_sgd_live;
    // a quad of the slab (four consecutive output units of one input: vjf_mega_slab_layout) per 8 lanes: lane p sums the late slabs [p npq, (p+1) npq) (16-byte loads, all in flight together with
    // the quad's old values, its table entries and the step's loss sums), then a fixed xor tree; lane 0 of the group clips and
    // steps its four parameters (model.py:210-211)
    const int npq = (A.n_trial + 7) >> 3, part = tid & 7;
    const int nquad = A.slab_len >> 2, qstride = (A.n_sgd * NT) >> 3;
    const int w1 = min(A.n_trial, (part + 1) * npq);
    // Every byte this role takes from the trial role (late slabs, loss sums) is read with sc1 loads behind the count's poll and the
    // workgroup barrier: no agent-scope acquire (vjf_wg_wait_sc1)
    const __amdgpu_buffer_rsrc_t r_late = mg_rsrc(A.slab_late);
    // A lane group serves the same quads in every step: the table entries and the parameters of its first round stay in
    // registers for the whole launch (a longer parameter vector reads the later rounds' from memory each step)
    const int q00 = (sw * NT) >> 3;
    int4 k_pi, k_ci;
    int k_grp;
    float k_w[4];
    auto fetch = [&](int quad, int4& pi, int4& ci, int& grp, float (&w)[4]) {
        pi = make_int4(-1, -1, -1, -1); ci = pi; grp = 0;
        w[0] = w[1] = w[2] = w[3] = 0.f;
        if (quad < nquad && part == 0) {
            pi = *reinterpret_cast<const int4*>(A.sl_pidx + (size_t)quad * 4);
            ci = *reinterpret_cast<const int4*>(A.sl_cidx + (size_t)quad * 4);
            grp = A.sl_grp[quad];
            const float* th = S + P.train_off;
            if (pi.x >= 0) w[0] = mg_ld(th + pi.x);                                // (this lane's own stores of the step before)
            if (pi.y >= 0) w[1] = mg_ld(th + pi.y);
            if (pi.z >= 0) w[2] = mg_ld(th + pi.z);
            if (pi.w >= 0) w[3] = mg_ld(th + pi.w);
        }
    };
    fetch(q00 + (tid >> 3), k_pi, k_ci, k_grp, k_w);
    if (tl) {
        // the parameter image of this launch (the caller may have rewritten the state blob since the last one): every lane group
        // stores the parameters of its quads; the trial role waits for all of them before its first step
        auto put = [&](const int4& pi, const int4& ci, const float (&w)[4]) {
            float* img = const_cast<float*>(A.img);
            if (pi.x >= 0 && ci.x >= 0) mg_st(img + ci.x, w[0]);
            if (pi.y >= 0 && ci.y >= 0) mg_st(img + ci.y, w[1]);
            if (pi.z >= 0 && ci.z >= 0) mg_st(img + ci.z, w[2]);
            if (pi.w >= 0 && ci.w >= 0) mg_st(img + ci.w, w[3]);
        };
        put(k_pi, k_ci, k_w);
        for (int q0 = q00 + qstride; q0 < nquad; q0 += qstride) {
            int4 pi, ci; int grp; float w[4];
            fetch(q0 + (tid >> 3), pi, ci, grp, w);
            put(pi, ci, w);
        }
        vjf_wg_signal_wt(A.cnt + MG_C_IMG, tid);
    }
    if (sw >= n_live) return;                          // (no gradient steps in this launch: one workgroup sums the losses and keeps the scalars)
    unsigned nredo = 0;
    // The scalars this role's first lane keeps -- the likelihood's log-variance and its sample count; in warm-up the state noise and its
    // count -- are its own stores of the step before: read ONCE, kept in registers (a load per step was a chain of two to four
    // memory round trips, 2-4 us, on the path of every gated step: the gate waits for this workgroup too)
    float k_rho = 0.f, k_nlik = 0.f, k_sig = 0.f, k_ntr = 0.f;
    if (sw == 0 && tid == 0) {
        k_rho = mg_ld(S + P.off[VJF_SLOT_LIK_LOGVAR]); k_nlik = mg_ld(SC + VJF_SC_N_LIK);
        if (do_upd && warm) { k_sig = mg_ld(S + P.off[VJF_SLOT_TR_LOGVAR]); k_ntr = mg_ld(SC + VJF_SC_N_TR); }
    }
    for (int t = 0; t < A.T; ++t) {
      float l_recon = 0.f, l_dyn = 0.f, ent = 0.f;
      bool ok_r = true, ok_d = true, ok_h = true, grad_ok = true;
      // pass 0: the step.  A loss with a non-finite component (not all three: then the gradient is zero, model.py:206-214) leaves
      // the parameters alone and publishes which components the trial role is to drop; pass 1 steps on its replayed late slabs
      for (int pass = 0; pass < 2; ++pass) {
        if (pass == 0) {
            if (!vjf_wg_wait_sc1<RLS ? VJF_POLL_SLEEP : VJF_POLL_SLEEP_LITE>(A.cnt + MG_C_BWD, (unsigned)(t + 1) * (unsigned)A.n_trial, tid, SC + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u))
                vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_RESIDENT);
        } else {
            ++nredo;
            if (!vjf_wg_wait_sc1<RLS ? VJF_POLL_SLEEP : VJF_POLL_SLEEP_LITE>(A.cnt + MG_C_REDO_B, nredo * (unsigned)A.n_trial, tid, SC + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u))
                vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_RESIDENT);
            grad_ok = true;
        }
        if (vjf_abort_wg()) return;                                            // (behind one of the two waits above)
        { const int wg = sw; VJF_MG_STAMP(16); }
        bool have_sums = pass == 1;
        // loss sums of the step: fp64, 32 strided partial sums per scalar, then a fixed xor tree (every SGD workgroup, for the guards)
        auto take_sums = [&]() {
            mg_sum_losses(A, t, s_sc, tid, Bf, P.dz, do_upd && warm);
            l_recon = s_sc[RS_LRECON] * invB; l_dyn = s_sc[RS_LDYN] * invB; ent = s_sc[RS_ENT] * invB;
            ok_r = isfinite(l_recon); ok_d = isfinite(l_dyn); ok_h = isfinite(ent);
            grad_ok = ok_r && ok_h && (warm || ok_d);
            have_sums = true;
        };
        // one round: the quads q0 + (tid >> 3).  (Uniform over the workgroup: the first round of a pass holds a barrier.)
        auto round = [&](int q0, const int4& pi, const int4& ci, int grp, float (&wold)[4]) {
            const int quad = q0 + (tid >> 3);
            const bool act = quad < nquad;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 tq[16];
            const int src = (act ? quad : 0) * 4;                                  // (float index into the late slabs)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                tq[q] = (act && part * npq + q < w1) ? mg_ld4(r_late, src + (part * npq + q) * A.late_len) : make_float4(0.f, 0.f, 0.f, 0.f);
            if (!have_sums) take_sums();
            for (int wq = part * npq + 16; wq < w1; wq += 16) {                   // (more than 128 trial workgroups: further rounds)
#pragma unroll
                for (int q = 0; q < 16; ++q) { v.x += tq[q].x; v.y += tq[q].y; v.z += tq[q].z; v.w += tq[q].w; }
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    tq[q] = (act && wq + q < w1) ? mg_ld4(r_late, src + (wq + q) * A.late_len) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) { v.x += tq[q].x; v.y += tq[q].y; v.z += tq[q].z; v.w += tq[q].w; }
            float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vv[r] += __shfl_xor(vv[r], 1, 64);
                vv[r] += __shfl_xor(vv[r], 2, 64);
                vv[r] += __shfl_xor(vv[r], 4, 64);
            }
            if (!act || part != 0 || (grp == 1 && freeze)) return;                 // (frozen decoder)
            const int pidx[4] = {pi.x, pi.y, pi.z, pi.w}, cidx[4] = {ci.x, ci.y, ci.z, ci.w};
            float* cdst = tl ? const_cast<float*>(A.img) : A.aux;
            // the state blob itself: nobody reads these parameters from it during the launch when the trial role has the image and
            // this lane group keeps them in registers -- then it is brought up to date at the last step only
            const bool wst = !tl || q0 != q00 || t == A.T - 1;
            if (!grad_ok) {
                // no step (model.py:206-214 skips optimizer.step() for this step alone): the steps before it, which this lane group
                // has kept in registers, still have to reach the blob when this is the last step of the launch
                if (tl && q0 == q00 && t == A.T - 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (pidx[r] >= 0) mg_st(S + P.train_off + pidx[r], wold[r]);
                }
                return;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (pidx[r] < 0) continue;                                     // (padding of the slab's rows)
                float g = vv[r] * invB;
                g = fminf(fmaxf(g, -1.f), 1.f);                                // clip_grad_value_ (model.py:210)
                const float wn = wold[r] - (grp == 1 ? lr_dec : lr_rec) * g;
                wold[r] = wn;
                if (wst) mg_st(S + P.train_off + pidx[r], wn);
                if (cidx[r] >= 0) mg_st(cdst + cidx[r], wn);
            }
        };
        if (!do_sgd) take_sums();
        else
        for (int q0 = q00; q0 < nquad || q0 == q00; q0 += qstride) {
            int4 pi = k_pi, ci = k_ci; int grp = k_grp;
            float w[4] = {k_w[0], k_w[1], k_w[2], k_w[3]};
            if (q0 != q00) fetch(q0 + (tid >> 3), pi, ci, grp, w);
            round(q0, pi, ci, grp, w);
            if (q0 == q00) { k_w[0] = w[0]; k_w[1] = w[1]; k_w[2] = w[2]; k_w[3] = w[3]; }
        }
        if (pass == 0 && t == 0 && mode_rls && mg_ld(SC + VJF_SC_TRI_CLEAN) == 0.f) {
            // one-time clearing of the halves the inverse loops never write (block-lower part of w_chol, block-upper part of
            // w_pchol): every reader of the dense w_chol of step 0 has signalled its late slab
            float* Wc = S + P.off[VJF_SLOT_W_CHOL];
            float* Lm = S + P.off[VJF_SLOT_W_PCHOL];
            const int n = P.n;
            for (int e = sw * NT + tid; e < n * n; e += n_live * NT) {
                const int i = e / n, j = e - i * n;
                if ((i >> 5) < (j >> 5)) mg_st(Lm + e, 0.f);
                if ((i >> 5) > (j >> 5)) { mg_st(Wc + e, 0.f); mg_st(const_cast<float*>(A.xt) + (size_t)j * n + i, 0.f); }   // (and its row-major transpose)
            }
        }
        const unsigned bad = (ok_r ? 0u : 1u) | (ok_d ? 0u : 2u) | (ok_h ? 0u : 4u);
        // some, not all, of the components IN the loss are non-finite: the reference steps along the gradient of the others
        const bool redo = pass == 0 && do_sgd && !grad_ok && (ok_r || ok_h || (!warm && ok_d));
        if (pass == 0 && sw == 0 && tid == 0) {                                // ---- scalars: loss, likelihood log-variance
            if (redo) __hip_atomic_store(A.cnt + MG_C_MASK, ((unsigned)(t + 1) << 8) | bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!ok_r) l_recon = 0.f;
            if (!ok_d) l_dyn = 0.f;
            if (!ok_h) ent = 0.f;
            const float loss = warm ? l_recon - ent : l_recon - ent + l_dyn;      // model.py:146-149
            if (A.loss) { float* l4 = A.loss + 4 * (size_t)t; l4[0] = loss; l4[1] = -l_recon; l4[2] = -l_dyn; l4[3] = ent; }
            const unsigned st = (ok_r ? 0u : VJF_STATUS_NONFINITE_RECON) | (ok_d ? 0u : VJF_STATUS_NONFINITE_DYN) |
                                (ok_h ? 0u : VJF_STATUS_NONFINITE_ENT);
            if (st) vjf_status_or(SC + VJF_SC_STATUS, st);
            if (P.lik == VJF_LIK_GAUSSIAN) {
                const float sse_y = s_sc[RS_SSEY];
                float rho = k_rho;
                if (do_sgd && ok_r) {                                          // (its gradient comes from the reconstruction term alone)
                    float g = 0.5f * ((float)P.dy - expf(-rho) * sse_y * invB);
                    g = fminf(fmaxf(g, -1.f), 1.f);
                    rho -= SC[VJF_SC_LR_LIK] * g;
                }
                if (do_upd) {                                                  // likelihood.py:28-40
                    const float mse = sse_y / (Bf * (float)P.dy);
                    const float acc = fminf(k_nlik, 1000.f), tot = acc + Bf;
                    rho = logf((acc / tot) * expf(rho) + (Bf / tot) * mse);
                    k_nlik = tot;
                    mg_st(SC + VJF_SC_N_LIK, tot);
                }
                k_rho = rho;
                if (do_sgd || do_upd) mg_st(S + P.off[VJF_SLOT_LIK_LOGVAR], rho);
            }
            if (do_upd && warm) {
                // warm-up: no RLS update, the state-noise running variance from the residual with the launch's W (model.py:370-377)
                const float mse = s_sc[RS_RESID];
                const float acc = fminf(k_ntr, 500.f), tot = acc + Bf;                            // running_var, size_cap=500 (model.py:375)
                k_sig = logf((acc / tot) * expf(k_sig) + (Bf / tot) * mse);
                k_ntr = tot;
                mg_st(S + P.off[VJF_SLOT_TR_LOGVAR], k_sig);
                mg_st(SC + VJF_SC_N_TR, tot);
            }
        }
        __syncthreads();
        vjf_wg_signal_wt(A.cnt + (pass == 0 ? MG_C_SGD : MG_C_REDO_S), tid);
        { const int wg = sw; VJF_MG_STAMP(17); }
        if (!redo) break;
      }
    }
    // (the launch's last act on the triangle flag: set once every SGD workgroup has cleared its share -- they all have signalled
    //  step 0 by then; the kernel boundary makes it visible to the next launch)
    if (sw == 0 && tid == 0 && mode_rls && SC[VJF_SC_TRI_CLEAN] == 0.f) {
        bool there = false;
        for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
            if ((int)(__hip_atomic_load(A.cnt + MG_C_SGD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - (unsigned)n_live) >= 0) { there = true; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (there) mg_st(SC + VJF_SC_TRI_CLEAN, 1.f);
    }
}

// The parameter image of a launch without parameter updates (nothing else for an SGD role to do there): builder `sw` of `nb`
// stores the parameters of its quads of the slab tables at their places in the image (what vjf_mega_sgd does at the start of the
// other launches), then counts itself in at MG_C_IMG.
__device__ __forceinline__ void mg_build_image(const VjfPlan& P, const VjfMegaArgs& A, const int sw, const int nb) {
    constexpr int NT = VJF_MG_THREADS;
    const int tid = threadIdx.x, part = tid & 7;
    if (vjf_mega_trial_lds<false>(P, A.lds_floats).theta == 0) return;    // (the trial role reads the state itself)
    const int nquad = A.slab_len >> 2, qstride = (nb * NT) >> 3;
    const float* th = A.state + P.train_off;
    float* img = const_cast<float*>(A.img);
    for (int q0 = (sw * NT) >> 3; q0 < nquad; q0 += qstride) {
        const int quad = q0 + (tid >> 3);
        if (quad < nquad && part == 0) {
            const int4 pi = *reinterpret_cast<const int4*>(A.sl_pidx + (size_t)quad * 4);
            const int4 ci = *reinterpret_cast<const int4*>(A.sl_cidx + (size_t)quad * 4);
            if (pi.x >= 0 && ci.x >= 0) mg_st(img + ci.x, th[pi.x]);
            if (pi.y >= 0 && ci.y >= 0) mg_st(img + ci.y, th[pi.y]);
            if (pi.z >= 0 && ci.z >= 0) mg_st(img + ci.z, th[pi.z]);
            if (pi.w >= 0 && ci.w >= 0) mg_st(img + ci.w, th[pi.w]);
        }
    }
    vjf_wg_signal_wt(A.cnt + MG_C_IMG, tid);
}

// ------------------------------------------------------------------------------------------------ the kernel
__global__ __launch_bounds__(VJF_MG_THREADS) void vjf_mega_kernel(VjfPlan P, VjfMegaArgs A, VjfCholArgs C, VjfPostArgs Q) {
    static_assert(VJF_CHOL_THREADS == VJF_MG_THREADS && VJF_POST_THREADS == VJF_MG_THREADS, "one workgroup size for every role");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_dead;
    if (threadIdx.x == 0) s_dead = 0;
    __syncthreads();
    int b = (int)blockIdx.x;
    // the NEXT launch's counters are zeroed here, by one workgroup of the operand role, before anything else (that block belongs to the
    // launch before this one, which is complete; the kernel boundary makes the zeros visible to the next launch): no memset in
    // front of a launch
    if (b == A.n_rls + A.n_trial + A.n_gram)
        for (int i = threadIdx.x; i < MG_C_WORDS; i += VJF_MG_THREADS) A.cnt_next[i] = 0u;
    float* stw = A.state + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS;
    if (mg_grid_resident(A.cnt, stw, A.alive_extra)) {
        if (b == 0) vjf_chol_loop<16>(P, C, lds, &s_dead);
        else if (b == 1) vjf_rls_post_loop(P, Q, lds, &s_dead, 2, 0);
        else if (b < A.n_rls) vjf_rls_post_loop(P, Q, lds, &s_dead, 1, b - 2);
        else if ((b -= A.n_rls) < A.n_trial) vjf_mega_trial<true>(P, A, lds, b);
        else if ((b -= A.n_trial) < A.n_gram) vjf_mega_gram(P, A, lds, b);
        else if ((b -= A.n_gram) < A.n_prep) vjf_mega_prep(P, A, lds, b);
        else vjf_mega_sgd<true>(P, A, lds, b - A.n_prep);
    }
    mg_tell_host(stw, A.host_word);
}

// The flag sets of VJF.filter that have no RLS update -- warm_up=True (the first epochs of fit: vjf/model.py:243-259, 148, 370),
// update=False, sgd=False (a deployed filter: model.py:180, 206, 215) -- as ONE launch too: trial and SGD roles only (W, w_chol are
// constants of the launch; in warm-up the SGD role's scalar lane also keeps the state-noise variance, from the residual sums the
// trial role hands over with its loss sums).  Without sgd and update nothing changes between steps and no role waits for another
// except through the ring of loss sums.
__global__ __launch_bounds__(VJF_MG_THREADS) void vjf_mega_lite_kernel(VjfPlan P, VjfMegaArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    int b = (int)blockIdx.x;
    if (b == A.n_trial)                                 // (the next launch's counter block: see vjf_mega_kernel)
        for (int i = threadIdx.x; i < MG_C_WORDS; i += VJF_MG_THREADS) A.cnt_next[i] = 0u;
    float* stw = A.state + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS;
    if (mg_grid_resident(A.cnt, stw, A.alive_extra)) {
        const int idx = b - A.n_trial;
        if (b < A.n_trial) vjf_mega_trial<false>(P, A, lds, b);
        else if (!(A.flags & (VJF_FLAG_SGD | VJF_FLAG_UPDATE))) {
            // nothing changes between the steps: no SGD role -- the first n_sgd of these workgroups build the parameter image, and all
            // n_mom of them (n_mom >= n_sgd, or none) are the moments role; the loss sums are the trial role's own (its last arriver)
            if (idx < A.n_sgd) mg_build_image(P, A, idx, A.n_sgd);
            if (idx < A.n_mom) vjf_mega_moments(P, A, lds, idx);
        }
        else if (idx < A.n_mom) vjf_mega_moments(P, A, lds, idx);
        else vjf_mega_sgd<false>(P, A, lds, idx - A.n_mom);
    }
    mg_tell_host(stw, A.host_word);
}
