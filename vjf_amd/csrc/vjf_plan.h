// vjf_plan.h -- dimensions, state-blob layout and workspace layout shared by host code and kernels.
#pragma once
#include <stdint.h>
#include "../../include/vjf_hip.h"

#define VJF_TILE 32            // Gram tile edge (v_mfma_f32_32x32x2_f32)
#define VJF_MAX_JOBS 4096

static inline int64_t vjf_align(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

// Everything a kernel needs to find its operands.  Passed by value as a kernel argument.
struct VjfPlan {
    int dy, dz, du, n, L, lik;
    int h[VJF_MAX_HIDDEN];
    int din;          // dy + du + 2 dz   (recognition input, recognition.py:20)
    int dxu;          // dz + du          (RBF input, model.py:335)
    int hmax, hsum;
    int off[VJF_N_SLOTS];    // state-blob offsets (fp32 elements)
    int size[VJF_N_SLOTS];
    int n_state;
    // trainable region = [REC_W0 .. DEC_B] contiguous in the blob; gradients mirror it
    int train_off, train_len;
    int dec_off;             // offset of DEC_W inside the blob (decoder group starts here)
    // workspace matrices, one row per trial
    int ldE;                 // E   = [ Phi (n) | dx (dz) | 0 ]           -> RLS statistics
    int ldA;                 // ACT = [ in | 1 | h_1 | 1 | ... | h_L | 1 | xt | 1 | 0 ]
    int ldD;                 // DEL = [ da_1 | ... | da_L | dmu | dlv | dpy | 0 ]
    int colA_act[VJF_MAX_HIDDEN + 1];   // column of acts[k] in ACT (k = 0: recognition input)
    int colA_xt;
    int colD_da[VJF_MAX_HIDDEN];
    int colD_dmu, colD_dlv, colD_dpy;
    // reduce buffer (fp32 elements): [ grad (train_len) | loss sums (4: RS_LRECON .. RS_SSEY) | G (n*n) | FDX (n*dz) | scalars (8) ]
    // -- what the gradient chain sums over ranks ([grad | loss sums]) and what the RLS chain sums ([G | FDX | sum |dx|^2]) are each ONE
    // contiguous range: one all-reduce per chain and step.  The loss sums live at red_SCA + RS_x, sum |dx|^2 at red_SC + RS_SDX2.
    int red_SCA, red_G, red_FDX, red_SC, red_len;
    // "aux": k-major (transposed) copies of the weights whose torch layout is output-major, so that the
    // MFMA A-operand of the forward products is read in 64-byte row segments.  Workspace, fp32 elements.
    //   aux_recT[l] : (h_{l-1}, h_l)      = rec_W[l]^T
    //   aux_headT   : (h_L, 2 dz)         = [mean_W ; lv_W]^T
    //   aux_decT    : (dz, dy)            = dec_W^T
    int aux_recT[VJF_MAX_HIDDEN], aux_headT, aux_decT, aux_len;
    // trainable tensors for the SGD sweep: offset in the blob, rows, cols, aux destination
    int n_train;
    int tr_off[2 * VJF_MAX_HIDDEN + 5], tr_rows[2 * VJF_MAX_HIDDEN + 5], tr_cols[2 * VJF_MAX_HIDDEN + 5];
    int tr_aux[2 * VJF_MAX_HIDDEN + 5];      // -1: none; else aux[tr_aux + c * tr_auxld + tr_auxcol + r]
    int tr_auxld[2 * VJF_MAX_HIDDEN + 5], tr_auxcol[2 * VJF_MAX_HIDDEN + 5];
    int tr_dec[2 * VJF_MAX_HIDDEN + 5];      // 1: decoder group (lr_dec, freeze flag)
};

// reduce-buffer scalars
enum { RS_LRECON = 0, RS_LDYN = 1, RS_ENT = 2, RS_SSEY = 3, RS_SDX2 = 4, RS_N = 8 };

static inline int vjf_make_plan(const vjf_config* c, VjfPlan* p) {
    if (!c || !p) return -1;
    if (c->ydim < 1 || c->xdim < 1 || c->udim < 0 || c->n_rbf < 1) return -2;
    if (c->n_hidden < 1 || c->n_hidden > VJF_MAX_HIDDEN) return -3;
    if (c->likelihood != VJF_LIK_GAUSSIAN && c->likelihood != VJF_LIK_POISSON) return -4;
    p->dy = c->ydim; p->dz = c->xdim; p->du = c->udim; p->n = c->n_rbf; p->L = c->n_hidden;
    p->lik = c->likelihood;
    p->din = p->dy + p->du + 2 * p->dz;
    p->dxu = p->dz + p->du;
    p->hmax = 0; p->hsum = 0;
    for (int k = 0; k < VJF_MAX_HIDDEN; ++k) {
        p->h[k] = k < p->L ? c->hidden[k] : 0;
        if (k < p->L && p->h[k] < 1) return -5;
        if (p->h[k] > p->hmax) p->hmax = p->h[k];
        p->hsum += p->h[k];
    }
    for (int s = 0; s < VJF_N_SLOTS; ++s) p->size[s] = 0;
    p->size[VJF_SLOT_PRIOR_MEAN] = p->dz;
    p->size[VJF_SLOT_PRIOR_LOGVAR] = p->dz;
    p->size[VJF_SLOT_LIK_LOGVAR] = 1;     // kept (unused) for Poisson so layouts agree
    p->size[VJF_SLOT_TR_LOGVAR] = 1;
    p->size[VJF_SLOT_CENTROID] = p->n * p->dxu;
    p->size[VJF_SLOT_LOGWIDTH] = p->n;
    int prev = p->din;
    for (int k = 0; k < p->L; ++k) {
        p->size[VJF_SLOT_REC_W0 + 2 * k] = p->h[k] * prev;
        p->size[VJF_SLOT_REC_B0 + 2 * k] = p->h[k];
        prev = p->h[k];
    }
    p->size[VJF_SLOT_MEAN_W] = p->dz * prev;
    p->size[VJF_SLOT_LV_W] = p->dz * prev;
    p->size[VJF_SLOT_LV_B] = p->dz;
    p->size[VJF_SLOT_DEC_W] = p->dy * p->dz;
    p->size[VJF_SLOT_DEC_B] = p->dy;
    p->size[VJF_SLOT_W_MEAN] = p->n * p->dz;
    p->size[VJF_SLOT_W_CHOL] = p->n * p->n;
    p->size[VJF_SLOT_W_PREC] = p->n * p->n;
    p->size[VJF_SLOT_W_PCHOL] = p->n * p->n;
    p->size[VJF_SLOT_SCALARS] = VJF_N_SCALARS;
    // blob order = state_dict order of the reference, then RLS tensors, then scalars
    int64_t o = 0;
    const int order_head[] = {VJF_SLOT_PRIOR_MEAN, VJF_SLOT_PRIOR_LOGVAR, VJF_SLOT_LIK_LOGVAR,
                              VJF_SLOT_TR_LOGVAR, VJF_SLOT_CENTROID, VJF_SLOT_LOGWIDTH};
    for (int i = 0; i < 6; ++i) { p->off[order_head[i]] = (int)o; o = vjf_align(o + p->size[order_head[i]], 4); }
    p->train_off = (int)o;
    for (int k = 0; k < VJF_MAX_HIDDEN; ++k) {
        p->off[VJF_SLOT_REC_W0 + 2 * k] = (int)o; o = vjf_align(o + p->size[VJF_SLOT_REC_W0 + 2 * k], 4);
        p->off[VJF_SLOT_REC_B0 + 2 * k] = (int)o; o = vjf_align(o + p->size[VJF_SLOT_REC_B0 + 2 * k], 4);
    }
    const int order_tail[] = {VJF_SLOT_MEAN_W, VJF_SLOT_LV_W, VJF_SLOT_LV_B, VJF_SLOT_DEC_W, VJF_SLOT_DEC_B};
    for (int i = 0; i < 5; ++i) { p->off[order_tail[i]] = (int)o; o = vjf_align(o + p->size[order_tail[i]], 4); }
    p->train_len = (int)o - p->train_off;
    p->dec_off = p->off[VJF_SLOT_DEC_W];
    const int order_rls[] = {VJF_SLOT_W_MEAN, VJF_SLOT_W_CHOL, VJF_SLOT_W_PREC, VJF_SLOT_W_PCHOL, VJF_SLOT_SCALARS};
    for (int i = 0; i < 5; ++i) { p->off[order_rls[i]] = (int)o; o = vjf_align(o + p->size[order_rls[i]], 4); }
    if (o > 0x7fffffff) return -6;
    p->n_state = (int)o;
    // workspace matrices
    p->ldE = (int)vjf_align(p->n + p->dz, VJF_TILE);
    int col = 0;
    // (every segment starts on a 16-byte boundary: the GEMMs of the wide route read them 16 bytes at a time; the padding
    //  columns behind a segment's 1 are never read)
    p->colA_act[0] = col; col = (int)vjf_align(col + p->din + 1, 4);
    for (int k = 0; k < p->L; ++k) { p->colA_act[k + 1] = col; col = (int)vjf_align(col + p->h[k] + 1, 4); }
    p->colA_xt = col; col += p->dz + 1;
    p->ldA = (int)vjf_align(col, 4);
    col = 0;
    for (int k = 0; k < p->L; ++k) { p->colD_da[k] = col; col = (int)vjf_align(col + p->h[k], 4); }
    p->colD_dmu = col; col += p->dz;                       // (dlv follows dmu directly: the dxt epilogue writes both)
    p->colD_dlv = col; col = (int)vjf_align(col + p->dz, 4);
    p->colD_dpy = col; col += p->dy;
    p->ldD = (int)vjf_align(col, 4);
    int64_t r = vjf_align(p->train_len, 4);            // (G is read as float4)
    p->red_SCA = (int)r; r += 4;                      // RS_LRECON, RS_LDYN, RS_ENT, RS_SSEY right behind the gradients
    p->red_G = (int)r; r += (int64_t)p->n * p->n;
    p->red_FDX = (int)r; r = vjf_align(r + (int64_t)p->n * p->dz, 4);
    p->red_SC = (int)r; r += RS_N;
    if (r > 0x7fffffff) return -6;
    p->red_len = (int)r;
    // aux layout + trainable-tensor table
    int64_t a = 0;
    int nt = 0;
    auto add = [&](int slot, int rows, int cols, int aux, int auxld, int auxcol, int dec) {
        p->tr_off[nt] = p->off[slot]; p->tr_rows[nt] = rows; p->tr_cols[nt] = cols;
        p->tr_aux[nt] = aux; p->tr_auxld[nt] = auxld; p->tr_auxcol[nt] = auxcol; p->tr_dec[nt] = dec; ++nt;
    };
    int prevw = p->din;
    for (int k = 0; k < VJF_MAX_HIDDEN; ++k) p->aux_recT[k] = 0;
    for (int k = 0; k < p->L; ++k) {
        p->aux_recT[k] = (int)a; a = vjf_align(a + (int64_t)prevw * p->h[k], 4);
        add(VJF_SLOT_REC_W0 + 2 * k, p->h[k], prevw, p->aux_recT[k], p->h[k], 0, 0);
        add(VJF_SLOT_REC_B0 + 2 * k, p->h[k], 1, -1, 0, 0, 0);
        prevw = p->h[k];
    }
    p->aux_headT = (int)a; a = vjf_align(a + (int64_t)prevw * 2 * p->dz, 4);
    add(VJF_SLOT_MEAN_W, p->dz, prevw, p->aux_headT, 2 * p->dz, 0, 0);
    add(VJF_SLOT_LV_W, p->dz, prevw, p->aux_headT, 2 * p->dz, p->dz, 0);
    add(VJF_SLOT_LV_B, p->dz, 1, -1, 0, 0, 0);
    p->aux_decT = (int)a; a = vjf_align(a + (int64_t)p->dz * p->dy, 4);
    add(VJF_SLOT_DEC_W, p->dy, p->dz, p->aux_decT, p->dy, 0, 1);
    add(VJF_SLOT_DEC_B, p->dy, 1, -1, 0, 0, 1);
    p->aux_len = (int)a;
    p->n_train = nt;
    return 0;
}

// One 32x32 output tile of a Gram product  out[i][j] = sum_b X[b][xc+i] * Y[b][yc+j].
struct VjfJob {
    int kind;          // 0: E^T E tile (RLS statistics), 1: gradient tile DEL^T ACT
    int xc, yc;        // first column in X / Y
    int xn, yn;        // valid rows (<=32) / cols (<=32) of the tile
    int ti, tj;        // tile coordinates (kind 0)
    int dst;           // kind 1: offset in the grad region of element (row 0, col 0) of this tile
    int ld;            // kind 1: leading dimension of the weight matrix
    int ncol_w;        // kind 1: columns [0, ncol_w) of the tile are weights, column ncol_w is the bias
    int dst_b;         // kind 1: offset in the grad region of the bias (-1: none)
    int tw, tb;        // kind 1: rows of the plan's trainable-tensor table that the weight / the bias belong to (tb -1: none)
};

// Bound of every wait of one kernel for another (polls; ~2 us each with the sleep between them, VJF_POLL_SLEEP: ~4 s).  Long enough for a
// host that is late with its launches, or a peer rank that is late with its half of a collective; short enough that a sequence
// that really is stuck (a launch held behind a resident kernel's hardware queue) is given up quickly.
#define VJF_WAIT_SPINS (1u << 21)
// s_sleep argument (units of 64 cycles) between two polls of a hand-off word in memory.  The polls of a launch -- 256 workgroups, most
// of them waiting most of the time -- all go to the few memory-side lines of the counter block, and they delay each other AND the
// write-through traffic of the step: with back-to-back polls (1) config B ran 56.3 us a step, with 12: 55.1, 24: 53.4-53.8,
// 40: 52.3-52.8, 64: 53.2 (same box, two runs each; a poll every ~1 us costs less in detection latency than the contention of
// faster ones; four out-of-phase pollers per workgroup: 60.4).  Spreading the counters over 4-KB pages of their own changed nothing.
// (Config C, whose trial role takes its operands from L2, would like 64 better: 74.8 against 76.5 us a step; config B 53.2 against 52.5.)
#ifndef VJF_POLL_SLEEP_LITE
#define VJF_POLL_SLEEP_LITE 16
#endif
#ifndef VJF_POLL_SLEEP
#define VJF_POLL_SLEEP 40
#endif
#ifdef __HIPCC__
// Hand-offs between kernels that run beside each other on different streams (no kernel boundary between producer and
// consumer).  Producer, whole workgroup: every storing wavefront drains its stores, the workgroup barrier, one lane releases
// at agent scope (L2 write-back), drains again, then the relaxed agent-scope count.  Consumer, whole workgroup: one lane
// polls (relaxed, bounded), acquires at agent scope, its vmcnt drained, the workgroup barrier, and only then the plain loads
// (MI355X guide, visibility across XCDs, valid forms).
// -DVJF_CHAOS (diagnostic builds only, tools/chaos_handoffs.sh): one workgroup in eight is held for up to 200 us in front of a wait or
// a signal, so that an access which is ordered by the usual timing of the roles and not by a hand-off shows as a wrong result.
#ifdef VJF_CHAOS
__device__ int vjf_chaos_range[6] = {0, 1 << 30, -1, 0, 20000, 7};          // workgroups [lo, hi) are held (VJF_CHAOS_LO / _HI) at count word [2] (-1: any; VJF_CHAOS_SITE), kind [3] (0 any, 1 waits, 2 signals),
                                                                            // for up to [4] ticks of 10 ns (VJF_CHAOS_TICKS), one time in [5] + 1 (a mask; VJF_CHAOS_MASK)
__device__ const unsigned* vjf_chaos_base = nullptr;
#endif
__device__ __forceinline__ void vjf_chaos(int tid, const unsigned* count, int kind) {
#ifdef VJF_CHAOS
    if (tid == 0 && (int)blockIdx.x >= vjf_chaos_range[0] && (int)blockIdx.x < vjf_chaos_range[1] &&
        (vjf_chaos_range[2] < 0 || count - vjf_chaos_base == vjf_chaos_range[2]) && (vjf_chaos_range[3] == 0 || vjf_chaos_range[3] == kind)) {
        const unsigned long long t0 = wall_clock64();                       // 100 MHz
        unsigned h = ((unsigned)t0 * 2654435761u) ^ (blockIdx.x * 40503u);
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        const unsigned d = (h & (unsigned)vjf_chaos_range[5]) == 0u ? (h >> 8) % (unsigned)vjf_chaos_range[4] : 0u;
        while (wall_clock64() - t0 < d) __builtin_amdgcn_s_sleep(8);
    }
#endif
}
__device__ __forceinline__ void vjf_wg_signal(unsigned* count, int tid) {
    vjf_chaos(tid, count, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
// A wait that ran out somewhere in the sequence (status detail bits 0x1ff00, include/vjf_hip.h) ends every other wait at once:
// the sequence is lost anyway (the host re-runs it), and nothing should sit through its own bound step after step.
__device__ __forceinline__ bool vjf_abort_seen(const float* status) {
    if (!status) return false;
    const float f = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return ((unsigned)f & VJF_STATUS_WAIT_MASK) != 0u;
}
// The same verdict for a whole workgroup: the lane that polled in the wait just before (vjf_wg_wait / vjf_wg_wait_sc1, given a status
// word) read the status word once more behind its poll and left what it saw in this LDS word in front of the wait's barrier --
// every thread of the workgroup takes the SAME decision to leave (a thread-by-thread read could split a workgroup around its later
// barriers when the bits are raised between two threads' loads).  4 bytes of static LDS in the kernels that wait.
__shared__ int vjf_s_abort_word;
__device__ __forceinline__ bool vjf_abort_wg() { return vjf_s_abort_word != 0; }
// The same for a workgroup whose outputs went out as write-through stores (in memory once vmcnt has drained): no L2 write-back
// (an agent-scope release by every workgroup of a kernel that runs beside the trial kernel costs that kernel microseconds).
__device__ __forceinline__ void vjf_wg_signal_wt(unsigned* count, int tid) {
    vjf_chaos(tid, count, 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The wait without the acquire: for consumers that read EVERY handed-off byte with sc1 loads (which bypass this CU's vector L1;
// the producer stored write-through and drained before it signalled) -- MI355X guide, "sc1 loads in place of the acquire".  One
// lane polls, the workgroup barrier, then the sc1 loads.
// `fence` = true adds the acquire (VJF_HANDOFF_ACQUIRE=1; the default of the one-launch route is the sc1 loads alone).
#define VJF_FLAG_HANDOFF_ACQUIRE 0x40000000u      /* internal flag bit of the kernels' `flags` words */
// (SLEEP: the pause between two polls, in units of 64 cycles -- the training launch has ~250 workgroups polling one 1-KB block and
//  wants them a microsecond apart, VJF_POLL_SLEEP; the launches without an RLS update have a third of the pollers and take 16)
template <int SLEEP = VJF_POLL_SLEEP>
__device__ __forceinline__ bool vjf_wg_wait_sc1(const unsigned* count, unsigned target, int tid, const float* status = nullptr, bool fence = false) {
    bool there = true;
    vjf_chaos(tid, count, 1);
    if (tid == 0) {
        there = false;
        for (unsigned spins = 0; spins < (1u << 21); ++spins) {
            if ((int)(__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { there = true; break; }
            if ((spins & 255u) == 255u && status && ((unsigned)__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & VJF_STATUS_WAIT_MASK)) break;
            __builtin_amdgcn_s_sleep(SLEEP);
        }
        if (fence) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        vjf_s_abort_word = (!there || vjf_abort_seen(status)) ? 1 : 0;
    }
    __syncthreads();
    return there;
}
__device__ __forceinline__ void vjf_store_wt(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// returns false (lane 0 only; the others get true) when the count did not arrive within the bound
__device__ __forceinline__ bool vjf_wg_wait(const unsigned* count, unsigned target, int tid, const float* status = nullptr) {
    bool there = true;
    vjf_chaos(tid, count, 1);
    if (tid == 0) {
        there = false;
        for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
            if ((int)(__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { there = true; break; }
            if ((spins & 255u) == 255u && vjf_abort_seen(status)) break;
            __builtin_amdgcn_s_sleep(VJF_POLL_SLEEP);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        vjf_s_abort_word = (!there || vjf_abort_seen(status)) ? 1 : 0;
    }
    __syncthreads();
    return there;
}
#endif

#ifdef __HIPCC__
// sum over the 32 lanes of a half wavefront, fixed xor tree (the loss sums: every kernel that forms them uses this order)
__device__ __forceinline__ double vjf_sum32(double v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
#endif

// A page of pinned host memory (one per process and device, vjf_abi.hip): every workgroup of a one-launch grid looks at the status
// word as it leaves, and when a wait has been given up it sets the word its status scalar hashes to -- the host, which does not
// synchronise between calls, learns of it at the start of the context's next call from a plain load (VJF_MIRROR_SLOT on both
// sides; a collision costs another context one status read, nothing else).  [The store used to sit in vjf_status_or itself, behind
// a __device__ pointer: in the diagnostic build that made this compiler's backend fail -- "Illegal instruction detected: Operand has
// incorrect register class.  V_CMP_NE_U32_e32 0, $src_shared_base" -- whatever form the store took.]
#define VJF_MIRROR_WORDS 256
#define VJF_MIRROR_SLOT(status_ptr) ((unsigned)(((uintptr_t)(status_ptr)) >> 4) & (VJF_MIRROR_WORDS - 1))
// OR status bits into the status scalar (a float holding a small integer).  Kernels of one step may run on two
// streams (vjf_filter_seq), so the read-modify-write is a compare-and-swap loop.
__device__ __forceinline__ void vjf_status_or(float* p, unsigned bits) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = *u, assumed;
    do {
        assumed = old;
        const float nv = (float)((unsigned)__uint_as_float(assumed) | bits);
        old = atomicCAS(u, assumed, __float_as_uint(nv));
    } while (old != assumed);
}
