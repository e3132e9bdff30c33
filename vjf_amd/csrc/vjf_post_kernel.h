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
#include "vjf_chol_kernel.h"         // VJF_CHOL_MAXBLK, vjf_f32x16
#include "vjf_trial_mfma_kernel.h"   // vjf_f32x4

// bound of every in-kernel wait (each poll is an L2 round trip + s_sleep: ~0.5 s in all); time-out -> status bit, never a hang
#define VJF_SPIN_LIMIT VJF_WAIT_SPINS
#define VJF_POST_THREADS 512
#define VJF_POST_KPAR 4                // wavefronts = 2 row tiles x 4 interleaved block sums
#define VJF_POST_LDB 33               // padded leading dimension of a 32x32 block in LDS
#define VJF_POST_LDX 17               // right-hand sides: [row][16 columns + 1 pad]
#define VJF_RESID_BLOCKS 64

struct VjfPostArgs {
    float* state;
    const float* dinv;      // nbl blocks (32x32 row-major) of inverted diagonal blocks, from vjf_chol_lds_kernel
    const float* lscr;      // (n, n) L as vjf_chol_lds_kernel left it (block-lower part valid); copied to w_pchol here
    const float* gbuf;      // (n, dz) g
    const unsigned* flags;  // flags[k] = (epoch << 1) | failed: column k of L (in lscr) and Dinv_k are in memory;
    unsigned epoch;         //   flags[VJF_CHOL_MAXBLK]: the factor as a whole is good.  Written by vjf_chol_lds_kernel, which may
                            //   still be running: the workgroups here consume the columns as they appear
    float* status;          // the status scalar of the state blob (time-out of the wait below)
    const unsigned* k1_done;  // workgroups of the trial kernel that have finished reading W, w_chol, sigma (null: not needed);
    unsigned k1_target;       //   nothing of those is written before the count reaches k1_target
    unsigned* started;        // += 1 per workgroup as it starts (it is resident from then on; see VjfPrepArgs::start_count)
    unsigned* done;           // += 1 per workgroup when its outputs (W, w_chol, sigma: write-through stores) are in memory:
                              //   vjf_gate_kernel on another stream lets the readers of the next step start on it
    const float* red;       // reduce buffer (G, FDX, sum|dx|^2) of this step
    int B_total;
    int role;               // 0: one launch, 2 nbl + 1 workgroups; 1: the 2 nbl inverse workgroups alone -- they keep only the
                            //    current column of L in LDS (48 KB: they fit beside a trial-kernel workgroup on its CU);
                            //    2: the y / W workgroup alone (it needs all of L for the backward substitution)
    // looping form (the y / W and inverse roles of the one-launch route, vjf_mega_kernel.h): nsteps > 0 -> one launch serves nsteps consecutive steps (its workgroups stay on their
    // CUs).  Step `it`: epoch + it, statistics in red / red2 by the parity of step0 + it, k1_target + it * k1_stride, and g
    // (from the operand kernel, another stream) is there when *prep_count has reached prep_target + it * prep_stride.
    int nsteps, step0;
    const float* red2;
    const unsigned* prep_count; unsigned prep_target, prep_stride, k1_stride;
    int undo_P;             // 1: the operand kernel that ran before this one on its stream added Phi^T Phi / v to the state's P without
                            //    knowing whether the factorisation would succeed: on failure the y / W workgroup takes it back
    int fold_sigma;         // 1: the y / W workgroup goes on to the state-noise update (no vjf_resid / vjf_sigma launch)
    unsigned long long* stamps;   // diagnostic only (null in normal runs): s_memrealtime of the y / W workgroup, slots 16..21
    int acquire;                  // 1: every wait acquires at agent scope as well (the one-launch route's default)
    unsigned long long* sig_word; // non-null: the new sigma also goes out as ONE 8-byte word {epoch, bits of sigma} for the Cholesky loop
                                  //   of the next step (it then needs neither this workgroup's exit count nor a second load)
    float* xt;                    // non-null (the one-launch route): (n, n) row-major L^-1 = w_chol^T, kept beside w_chol for the trial role,
                                  //   whose predictive-variance products read it 16 bytes at a time along k (vjf_mega_kernel.h)
    unsigned* xt_count;           //   += 1 per inverse workgroup once, at the start of a launch, its share of xt = w_chol^T (from the state) is in memory
};

#define VJF_POST_STAMP(i)                                                                   \
    do {                                                                                    \
        if (A.stamps && solve && tid == 0) {                                                \
            unsigned long long t_;                                                          \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
            A.stamps[((it_epoch & 7u) << 5) * (A.nsteps > 0 ? 1 : 0) + (i)] = t_;              \
        }                                                                                   \
    } while (0)

static inline size_t vjf_post_inv_lds_bytes(const VjfPlan& P) {      // role 1: (nbl - 1) blocks of one column + Dinv_k | solution | ...
    const int nbl = (P.n + 31) / 32;
    return ((size_t)nbl * 32 * VJF_POST_LDB + (size_t)nbl * 32 * VJF_POST_LDX + 32 * VJF_POST_LDX + 64 + 16 + 16) * 4;
}
static inline size_t vjf_post_lds_bytes(const VjfPlan& P) {
    const int nbl = (P.n + 31) / 32;
    // L blocks + Dinv blocks | solution | block just solved | block table
    return ((size_t)(nbl * (nbl - 1) / 2 + nbl) * 32 * VJF_POST_LDB + (size_t)nbl * 32 * VJF_POST_LDX + 32 * VJF_POST_LDX + 64 + 16 + 16) * 4;
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

// What these workgroups take from workgroups that run beside them (columns of L and the inverted diagonal blocks from the
// Cholesky loop; g, Phi^T dx, the sums from the operand role; Phi^T Phi from the Gram role; sigma and the sample count, their own
// stores of the step before) was stored write-through and drained before the flag / count that announces it, and is read with
// sc1 loads -- 16-byte buffer loads or 4-byte agent-scope loads, which bypass this CU's vector L1 -- behind the poll that matched
// and the workgroup barrier: no agent-scope acquire (an L1 invalidate the whole workgroup would wait ~1.7 us for) per column
// (MI355X guide, "sc1 loads in place of the acquire").  The rare failure path, which reads more, does acquire.
typedef unsigned post_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t post_rsrc(const float* base, size_t nfloats) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(nfloats * 4), 0x00020000);
}
__device__ __forceinline__ float4 post_ld4(__amdgpu_buffer_rsrc_t r, size_t float_off) {
    const post_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, 16);      // aux 16 = sc1
    return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ float post_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Wait (one lane polls, relaxed, bounded) until the Cholesky kernel has published flag word `k` for this epoch; the workgroup
// barrier; then the sc1 loads of the column (see above).  Returns 0 = there, 1 = the factorisation failed, 2 = timed out.
__device__ __forceinline__ int post_wait_column(const unsigned* flags, unsigned epoch, int k, int* s_ctl, int tid, const float* status, bool fence = false) {
    vjf_chaos(tid, flags + k, 1);
    if (tid == 0) {
        int st = 2;
        for (unsigned spins = 0; spins < VJF_SPIN_LIMIT; ++spins) {
            const unsigned v = __hip_atomic_load(flags + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((v >> 1) == epoch) { st = (int)(v & 1u); break; }
            if ((spins & 255u) == 255u && vjf_abort_seen(status)) break;
            __builtin_amdgcn_s_sleep(4);
        }
        if (fence) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        s_ctl[0] = st;
    }
    __syncthreads();
    const int st = s_ctl[0];
    __syncthreads();                                           // (s_ctl is reused by the next wait)
    return st;
}

// The y / W workgroup, whose LDS holds all of L, takes in every column that has been published so far in one go: a look at the
// flags from..kmax without waiting (and at the operand role's count for g),
// s_ctl[0] = 1 if a column reports a failed pivot, s_ctl[1] = the last column found published (from - 1: none new), s_ctl[2] =
// g is there.  With wait_first the first flag is waited for as post_wait_column does (returns its codes in s_ctl[0]).
__device__ __forceinline__ void post_peek_columns(const unsigned* flags, unsigned epoch, int from, int kmax, bool wait_first,
                                                  const unsigned* prep_count, unsigned prep_target, int* s_ctl, int tid, const float* status,
                                                  bool fence = false) {
    if (wait_first) vjf_chaos(tid, flags + from, 1);
    if (tid == 0) {
        int st = 0, kr = from - 1;
        if (wait_first) {
            st = 2;
            for (unsigned spins = 0; spins < VJF_SPIN_LIMIT; ++spins) {
                const unsigned v = __hip_atomic_load(flags + from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 1) == epoch) { st = (int)(v & 1u); break; }
                if ((spins & 255u) == 255u && vjf_abort_seen(status)) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (st == 0) kr = from;
        }
        if (st == 0)
            for (int j = kr + 1; j <= kmax; ++j) {
                const unsigned v = __hip_atomic_load(flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((v >> 1) != epoch) break;
                if (v & 1u) { st = 1; break; }
                kr = j;
            }
        const int g = prep_count ? ((int)(__hip_atomic_load(prep_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - prep_target) >= 0 ? 1 : 0) : 1;
        if (kr < from && !g) __builtin_amdgcn_s_sleep(8);
        else if (fence) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        s_ctl[0] = st; s_ctl[1] = kr; s_ctl[2] = g;
    }
    __syncthreads();
}

__device__ __forceinline__ void vjf_rls_post_body(const VjfPlan& P, const VjfPostArgs& A, float* lds, int* s_dead, const unsigned it_epoch,
                                                  const float* it_red, const unsigned it_k1_target, const unsigned it_prep_target,
                                                  const bool it_first, const int role, const int bix) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                               // (see vjf_chol_body: nothing lane-dependent is hoisted out of the step loop)
    const int lane = tid & 63, wave = tid >> 6;
    const int n = P.n, dz = P.dz, nbl = (n + 31) / 32, ntri = nbl * (nbl + 1) / 2, nlow = ntri - nbl;
    const bool solve = role == 2 || bix == 2 * nbl;            // the y / W workgroup
    const bool col_only = role == 1;                           // LDS holds the current column of L only
    constexpr int LB = VJF_POST_LDB, LX = VJF_POST_LDX;
    float* s_L = lds;                                          // strictly-lower blocks [32][33] of L (bi > bj); col_only: (i - k - 1)
    float* s_D = s_L + (size_t)(col_only ? nbl - 1 : nlow) * 32 * LB;   // nbl blocks [32][33]: inverted diagonal blocks; col_only: one
    float* s_x = s_D + (size_t)(col_only ? 1 : nbl) * 32 * LB; // [npad][17] right-hand sides -> solution
    float* s_y = s_x + (size_t)nbl * 32 * LX;                  // [32][17] the block just solved, for the eager updates
    int* s_tab = reinterpret_cast<int*>(s_y + 32 * LX);        // [32..64): lower tiles incl. diagonal -> (bi << 8) | bj
    int* s_ctl = s_tab + 64;
    auto leave = [&]() {                                       // every workgroup, on every path, exactly once
        vjf_chaos(tid, A.done, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (A.stamps && !solve && tid == 0) {                  // diagnostic: when the LAST inverse loop of the step is done
            unsigned long long t_;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
            atomicMax(A.stamps + ((it_epoch & 7u) << 5) * (A.nsteps > 0 ? 1 : 0) + 22, t_);
        }
        if (tid == 0 && A.done) __hip_atomic_fetch_add(A.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    const float* S = A.state;
    const __amdgpu_buffer_rsrc_t r_L = post_rsrc(A.lscr, (size_t)n * n), r_dinv = post_rsrc(A.dinv, (size_t)nbl * 1024);
    const int j0 = solve ? 0 : bix >> 1;                       // first block row of the substitution
    const int c0 = solve ? 0 : 16 * (bix & 1);
    auto tri = [](int bi, int bj) { return bi * (bi - 1) / 2 + bj; };       // strictly lower: bi > bj
    auto lblk = [&](int bi, int bj) { return s_L + (size_t)(col_only ? bi - bj - 1 : tri(bi, bj)) * 32 * LB; };   // block (bi, bj), bi > bj
    auto dblk = [&](int k) { return s_D + (size_t)(col_only ? 0 : k) * 32 * LB; };

    if (tid == 0 && A.started && it_first) __hip_atomic_fetch_add(A.started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float pre_sdx2 = 0.f, pre_old = 0.f, pre_tot = 1.f;         // sum|dx|^2, old share of the running variance, new count
    double pre_scale = 0.0;                                     // mse -> new share of the running variance
    VJF_POST_STAMP(16);
    if (tid < ntri) {                                          // lower tiles, diagonal included: t = bi (bi + 1) / 2 + bj
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= tid) ++bi;
        s_tab[32 + tid] = (bi << 8) | (tid - bi * (bi + 1) / 2);
    }
    // right-hand side: 16 columns of the identity (inverse); the y / W workgroup takes g after its first wait below (g comes
    // from the prep kernel, which the Cholesky kernel follows in its stream: a column flag of this epoch says it is there)
    if (!solve)
        for (int e = tid; e < nbl * 32 * 16; e += VJF_POST_THREADS) {
            const int r = e >> 4, c = e & 15;
            s_x[r * LX + c] = (r == j0 * 32 + c0 + c) ? 1.f : 0.f;
        }
    __syncthreads();

    float gpre[4][16], fpre[8];
    // weights of this lane's tile entries in tr(W^T G W) = sum_ij G_ij (W W^T)_ij over the lower triangle: 2 below the diagonal,
    // 1 on it, 0 above it and outside the matrix -- two bits per entry, formed here (nothing waits on this workgroup yet) and
    // not in the tail behind W, where every instruction is on the path to the next step's factorisation
    unsigned wbits[4];
    {
        const int c = lane & 31, h = lane >> 5;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = wave + 8 * q;
            unsigned wq = 0u;
            if (solve && t < ntri) {
                const int code = s_tab[32 + t], bi = code >> 8, bj = code & 255;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gi = bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, gj = bj * 32 + c;
                    wq |= ((gi >= n || gj > gi) ? 0u : (gj == gi ? 1u : 2u)) << (2 * r);
                }
            }
            wbits[q] = wq;
        }
    }
    // the tail's operands, fetched early so that it does not wait for them: the tiles of G once the Cholesky loop has started on
    // this step (its first column flag: the Gram role's sums were complete before it began) ...
    auto prefetch_G = [&]() {
        // wavefront w: the lower 32x32 tiles w, w + 8, .. of G in the matrix-core accumulator layout
        // (loads from clamped addresses: the tail masks what lies outside the matrix)
        const float* G = it_red + P.red_G;
        const int c = lane & 31, h = lane >> 5;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int code = s_tab[32 + min(wave + 8 * q, ntri - 1)];
            const int bi = code >> 8, bj = code & 255;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gi = min(bi * 32 + (r & 3) + 8 * (r >> 2) + 4 * h, n - 1), gj = min(bj * 32 + c, n - 1);
                gpre[q][r] = post_ld(G + (size_t)gi * n + gj);
            }
        }
    };
    // ... and, with g, what the operand role leaves beside it (sum |dx|^2, Phi^T dx) and the scalars of the state-noise update
    auto prefetch_rest = [&]() {
        pre_sdx2 = post_ld(it_red + P.red_SC + RS_SDX2);
        const float sig = post_ld(S + P.off[VJF_SLOT_TR_LOGVAR]);
        const float Bf = (float)A.B_total;
        const float acc = fminf(post_ld(S + P.off[VJF_SLOT_SCALARS] + VJF_SC_N_TR), 500.f);   // running_var, size_cap=500 (model.py:375)
        pre_tot = acc + Bf;
        pre_old = (acc / pre_tot) * expf(sig);
        pre_scale = 1.0 / ((double)Bf * (double)P.dz);
        const float* FDX = it_red + P.red_FDX;
#pragma unroll
        for (int q = 0; q < 8; ++q) {                                  // W's [row][16] grid: 224 * 16 <= 8 * 512
            const int e = tid + q * VJF_POST_THREADS, r = min(e >> 4, n - 1), cc = min(e & 15, dz - 1);
            fpre[q] = post_ld(FDX + r * dz + cc);
        }
    };
    auto prefetch_tail = [&]() { prefetch_G(); prefetch_rest(); };

    // Both substitutions run eagerly: as soon as block k of the solution exists (two wavefronts, one 16-row tile each),
    // every wavefront subtracts its contribution from the 16-row tiles of the later blocks it owns, in place in s_x.
    const int tile = wave & 1, grp = wave >> 1;                // owner of tile `tile` of blocks first + grp, first + grp + 4, ..
    const int xr = 4 * (lane >> 4), xc = lane & 15;            // accumulator element (row xr + r, column xc) of a 16x16 tile
    int bad = 0;
    bool g_there = A.prep_count == nullptr;                    // g (and the state's P) from the operand kernel on another stream
    auto wait_g = [&]() {
        if (g_there) return;
        g_there = true;
        if (!vjf_wg_wait(A.prep_count, it_prep_target, tid, A.status)) { vjf_status_or(A.status, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_G); *s_dead = 1; }
    };
    // ---- forward  Y_k = Dinv_k R_k ;  R_i -= L_ik Y_k  (i > k),   k = j0 .. nbl-1, column k of L staged when it appears
    auto stage_column = [&](int k) {   // blocks (i, k), i > k, and Dinv_k; (nbl - k) x 256 float4 chunks, all of a thread's in flight
        const int nb = nbl - k;
        float4 v[4];                                                       // nb * 256 <= 7 * 256 <= 4 * 512
        const int r = (tid >> 3) & 31, c4 = (tid & 7) * 4, bh = tid >> 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int it = 2 * q + bh;                                     // 0: Dinv_k; i = k + it: L block (i, k)
            const int gi = (k + it) * 32 + r, gj = k * 32 + c4;
            const bool real = it < nb && (it == 0 || (gi < n && gj < n));  // (padding rows / columns of L are zero)
            v[q] = it == 0 ? post_ld4(r_dinv, real ? (size_t)k * 1024 + r * 32 + c4 : 0) : post_ld4(r_L, real ? (size_t)gi * n + gj : 0);
        }                                                                  // (zeroed below, where they are stored: a select right behind a
#pragma unroll                                                             //  load waits for it, and the four would go one after the other)
        for (int q = 0; q < 4; ++q) {
            const int it = 2 * q + bh;
            const int gi = (k + it) * 32 + r, gj = k * 32 + c4;
            if (!(it < nb && (it == 0 || (gi < n && gj < n)))) v[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (it < nb) {
                float* dst = (it == 0 ? dblk(k) : lblk(k + it, k)) + (size_t)r * LB + c4;
                dst[0] = v[q].x; dst[1] = v[q].y; dst[2] = v[q].z; dst[3] = v[q].w;
            }
        }
    };
    int staged = j0;                                           // (y / W workgroup) columns < staged are in LDS
    for (int k = j0; k < nbl; ++k) {
        if (!solve) {
            bad = post_wait_column(A.flags, it_epoch, k, s_ctl, tid, A.status, A.acquire != 0);
            if (bad) break;
            stage_column(k);
        } else {
            if (k >= staged) {                                 // wait for column k, take whatever else is there with it
                post_peek_columns(A.flags, it_epoch, k, nbl - 1, true, nullptr, 0u, s_ctl, tid, A.status, A.acquire != 0);
                bad = s_ctl[0];
                const int kr = s_ctl[1];
                __syncthreads();                               // (s_ctl is reused)
                if (bad) break;
                for (int k2 = k; k2 <= kr; ++k2) stage_column(k2);
                staged = kr + 1;
            }
            bool g_pre = false;
            if (k == 0) {
                // g (and with it the RLS statistics) comes from the operand role, usually after the first columns of L: they are
                // taken in as they appear while this workgroup waits for it
                if (!g_there && A.fold_sigma) { prefetch_G(); g_pre = true; }
                if (!g_there) {
                    for (unsigned spins = 0;; ++spins) {
                        post_peek_columns(A.flags, it_epoch, staged, nbl - 1, false, A.prep_count, it_prep_target, s_ctl, tid, A.status, A.acquire != 0);
                        const int st = s_ctl[0], kr = s_ctl[1], g = s_ctl[2];
                        __syncthreads();
                        if (st) { bad = st; break; }
                        for (int k2 = staged; k2 <= kr; ++k2) stage_column(k2);
                        if (kr >= staged) staged = kr + 1;
                        if (g) break;
                        if (spins >= VJF_SPIN_LIMIT / 8 || ((spins & 63u) == 63u && vjf_abort_seen(A.status))) {
                            if (tid == 0) { vjf_status_or(A.status, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_G); *s_dead = 1; }
                            break;
                        }
                    }
                    g_there = true;
                    if (bad) break;
                }
                // Everything this workgroup takes from the kernels that precede the Cholesky kernel in its stream (g, the RLS
                // statistics) is read behind the first column flag of this epoch: the flag says those kernels are complete.
                for (int e = tid; e < nbl * 32 * 16; e += VJF_POST_THREADS) {
                    const int r = e >> 4, c = e & 15;
                    s_x[r * LX + c] = (r < n && c < dz) ? post_ld(A.gbuf + (size_t)r * dz + c) : 0.f;
                }
                if (A.fold_sigma) { if (!g_pre) prefetch_G(); prefetch_rest(); }
            }
        }
        __syncthreads();
        if (k == j0) VJF_POST_STAMP(17);
        vjf_f32x4 y = {0.f, 0.f, 0.f, 0.f};
        if (wave < 2) {
            const float* Db = dblk(k);
            post_mma32(y, s_x + (size_t)k * 32 * LX, lane, [&](int i, int m) { return Db[(16 * wave + i) * LB + m]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) s_y[(16 * wave + xr + r) * LX + xc] = y[r];
        }
        __syncthreads();
        if (wave < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_x[(k * 32 + 16 * wave + xr + r) * LX + xc] = y[r];
        }
        for (int i = k + 1 + grp; i < nbl; i += VJF_POST_THREADS / 128) {
            const float* Lb = lblk(i, k);
            float* xt = s_x + ((size_t)i * 32 + 16 * tile + xr) * LX + xc;
            vjf_f32x4 acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = xt[r * LX];
            post_mma32(acc, s_y, lane, [&](int ii, int m) { return -Lb[(16 * tile + ii) * LB + m]; });
#pragma unroll
            for (int r = 0; r < 4; ++r) xt[r * LX] = acc[r];
        }
        __syncthreads();
    }
    // (every column up to the last one reported good pivots: the factor as a whole is good -- its own flag, which the Cholesky
    //  loop raises right behind the last column's, need not be waited for)
    // readers of W, w_chol, sigma on another stream: all done?  The y / W loop of the one-launch route asks only when it is about to
    // STORE W and sigma (below): the substitutions, the state-noise update and the sigma word for the next factorisation do not
    // touch what those readers read, and the poll's round trip is off the path sigma -> Cholesky -> W -> sigma.
    const bool k1_late = solve && A.fold_sigma && A.sig_word != nullptr;
    if (A.k1_done && !k1_late) {
        if (tid == 0) {
            int st = 2;
            for (unsigned spins = 0; spins < VJF_SPIN_LIMIT; ++spins) {
                const unsigned v = __hip_atomic_load(A.k1_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(v - it_k1_target) >= 0) { st = 0; break; }
                if ((spins & 255u) == 255u && vjf_abort_seen(A.status)) break;
                __builtin_amdgcn_s_sleep(4);
            }
            s_ctl[0] = st;
        }
        __syncthreads();
        if (s_ctl[0] && !bad) bad = 2;
        __syncthreads();
    }
    VJF_POST_STAMP(18);
    if (bad == 2 && tid == 0) { vjf_status_or(A.status, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_COLUMN); *s_dead = 1; }
    const bool failed = bad != 0;                              // factorisation failed: RLS state stays as it was
    if (failed && !(solve && A.fold_sigma)) { leave(); return; }

    if (!failed) {
        if (!solve) {
            // ---- w_pchol = L (module.py:99-100): the factor is good, so the scratch copy goes to the state; 1/(2 nbl) each
            {
                float* Ls = A.state + P.off[VJF_SLOT_W_PCHOL];
                const int nq = n * n / 4, per = (nq + 2 * nbl - 1) / (2 * nbl);
                const int q0 = bix * per, q1 = min(nq, q0 + per);
                for (int q = q0 + tid; q < q1; q += VJF_POST_THREADS) {
                    const int e = 4 * q, i = e / n, j = e - i * n;
                    if ((j >> 5) <= (i >> 5)) *reinterpret_cast<float4*>(Ls + e) = post_ld4(r_L, (size_t)e);
                }
            }
            // ---- w_chol[(j0*32 + c0 + c)][i] = X[i][c]: rows of w_chol, contiguous over i  (module.py:102)
            float* Wc = A.state + P.off[VJF_SLOT_W_CHOL];
            const int first = j0 * 32;
            for (int c = wave; c < 16; c += VJF_POST_THREADS / 64) {
                const int gc = j0 * 32 + c0 + c;
                if (gc >= n) continue;
                // (16-byte write-through stores: n % 4 == 0 on this path and `first` is a multiple of 32 -- a quarter of the fabric
                //  writes of the scalar form, beside the trial role's gradient slabs, which can leave at the same time)
                for (int i = first + 4 * lane; i < n; i += 256) {
                    vjf_f32x4 o = {s_x[i * LX + c], s_x[(i + 1) * LX + c], s_x[(i + 2) * LX + c], s_x[(i + 3) * LX + c]};
                    float* dstp = Wc + (size_t)gc * n + i;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dstp), "v"(o) : "memory");
                }
            }
            if (A.xt) {
                // the same 16 columns of X = L^-1 row by row: xt[i][j0 * 32 + c0 .. + 16) (rows above the block: zeros the trial role
                // never reads; rows inside it above the diagonal: the zeros of the substitution)
                const int gc0 = j0 * 32 + c0;
                for (int e = tid; e < (n - first) * 4; e += VJF_POST_THREADS) {
                    const int i = first + (e >> 2), q = (e & 3) * 4;
                    if (gc0 + q < n) {                                  // (n % 4 == 0)
                        vjf_f32x4 o = {s_x[i * LX + q], s_x[i * LX + q + 1], s_x[i * LX + q + 2], s_x[i * LX + q + 3]};
                        float* dstp = A.xt + (size_t)i * n + gc0 + q;
                        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dstp), "v"(o) : "memory");
                    }
                }
            }
            leave();
            return;
        }
        // ---- backward  W_k = Dinv_k^T R_k ;  R_i -= L_ki^T W_k  (i < k),   k = nbl-1 .. 0   (module.py:101)
        for (int k = nbl - 1; k >= 0; --k) {
            vjf_f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (wave < 2) {
                const float* Db = s_D + (size_t)k * 32 * LB;
                post_mma32(w, s_x + (size_t)k * 32 * LX, lane, [&](int i, int m) { return Db[m * LB + 16 * wave + i]; });
#pragma unroll
                for (int r = 0; r < 4; ++r) s_y[(16 * wave + xr + r) * LX + xc] = w[r];
            }
            __syncthreads();
            if (wave < 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) s_x[(k * 32 + 16 * wave + xr + r) * LX + xc] = w[r];
            }
            for (int i = k - 1 - grp; i >= 0; i -= VJF_POST_THREADS / 128) {
                const float* Lb = s_L + (size_t)tri(k, i) * 32 * LB;
                float* xt = s_x + ((size_t)i * 32 + 16 * tile + xr) * LX + xc;
                vjf_f32x4 acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = xt[r * LX];
                post_mma32(acc, s_y, lane, [&](int ii, int m) { return -Lb[m * LB + 16 * tile + ii]; });
#pragma unroll
                for (int r = 0; r < 4; ++r) xt[r * LX] = acc[r];
            }
            __syncthreads();
        }
        VJF_POST_STAMP(19);
    }   // !failed
    auto store_W = [&]() {
        float* Wm = A.state + P.off[VJF_SLOT_W_MEAN];
        for (int e = tid; e < n * 16; e += VJF_POST_THREADS) {
            const int r = e >> 4, c = e & 15;
            if (c < dz) __hip_atomic_store(Wm + r * dz + c, s_x[r * LX + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1
        }
    };
    if (!failed && !k1_late) store_W();
    if (!A.fold_sigma) { leave(); return; }
    // ---- state-noise update on the new W (model.py:373-377):  q = sum|dx|^2 - 2 tr(W^T FDX) + tr(W^T G W)  over the batch.
    //      tr(W^T G W) = sum_ij G_ij (W W^T)_ij: wavefront w forms the lower 32x32 tiles t = w, w + 8, .. of W W^T on the
    //      f32 matrix cores and contracts them with the tiles of G it prefetched at kernel start (fp64 sums, fixed order).
    {
        double* s_p = reinterpret_cast<double*>(lds);          // one partial per wavefront (over s_L: the substitutions are done)
        if (failed) {                                          // sigma still moves, on the W that stays
            wait_g();
            if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __syncthreads();                                   // (this path reads the state with plain loads)
            const float* Wold = A.state + P.off[VJF_SLOT_W_MEAN];
            for (int e = tid; e < nbl * 32 * 16; e += VJF_POST_THREADS) {
                const int r = e >> 4, c = e & 15;
                s_x[r * LX + c] = (r < n && c < dz) ? Wold[(size_t)r * dz + c] : 0.f;
            }
            prefetch_tail();                                   // (the failure may have come before the first column)
            if (A.undo_P) {                                    // exact up to one rounding, as the Cholesky kernel does on its own path
                float* Pm = A.state + P.off[VJF_SLOT_W_PREC];
                const float* G = it_red + P.red_G;
                const float inv_v = expf(-S[P.off[VJF_SLOT_TR_LOGVAR]]);
                // (write-through: in the one-launch route the next step's operand workgroups, on other CUs, read these rows
                //  with no kernel boundary in between)
                for (int e = tid; e < n * n; e += VJF_POST_THREADS) vjf_store_wt(Pm + e, fmaf(-G[e], inv_v, Pm[e]));
            }
        }
        __syncthreads();
        VJF_POST_STAMP(20);
#ifdef VJF_EXPERIMENT_SLOW_RLS     /* sensitivity experiment: the state-noise tail held for this many 10-ns ticks per step */
        { const unsigned long long t0_ = wall_clock64(); while (wall_clock64() - t0_ < VJF_EXPERIMENT_SLOW_RLS) __builtin_amdgcn_s_sleep(1); }
#endif
        double part = 0.0;
        const int c = lane & 31, h = lane >> 5;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int t = wave + 8 * q;
            if (t < ntri) {
                const int code = s_tab[32 + t], bi = code >> 8, bj = code & 255;
                vjf_f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                float wa[8], wb[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {                  // columns dz..15 and rows n.. of W are zero
                    wa[k] = s_x[(bi * 32 + c) * LX + 2 * k + h];
                    wb[k] = s_x[(bj * 32 + c) * LX + 2 * k + h];
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (2 * k < dz) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[k], wb[k], acc, 0, 0, 0);   // (uniform branch)
                double tp = 0.0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {                 // accumulator: row = (r&3) + 8*(r>>2) + 4*h, column = c
                    const float wgt = (float)((wbits[q] >> (2 * r)) & 3u);
                    tp += (double)((wgt * gpre[q][r]) * acc[r]);   // (one fp32 rounding per product; the SUMS are fp64)
                }
                part += tp;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + q * VJF_POST_THREADS, r = e >> 4, cc = e & 15;
            if (r < n && cc < dz) part -= 2.0 * (double)s_x[r * LX + cc] * (double)fpre[q];
        }

#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (lane == 0) s_p[wave] = part;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < VJF_POST_THREADS / 64; ++w) t += s_p[w];
            t += (double)pre_sdx2;
            if (t < 0.0) t = 0.0;
            float* St = A.state;
            float* SC = St + P.off[VJF_SLOT_SCALARS];
            const float mse = (float)(t * pre_scale);
            const float new_sig = logf(pre_old + ((float)A.B_total / pre_tot) * mse);
            if (A.sig_word)                                    // (first: the next step's factorisation waits for this word alone)
                __hip_atomic_store(A.sig_word, ((unsigned long long)it_epoch << 32) | (unsigned long long)__float_as_uint(new_sig), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            bool ok1 = true;
            if (k1_late && A.k1_done) {                        // sigma, the sample count and W of the state: behind their last readers
                ok1 = false;
                for (unsigned spins = 0; spins < VJF_SPIN_LIMIT; ++spins) {
                    const unsigned v = __hip_atomic_load(A.k1_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int)(v - it_k1_target) >= 0) { ok1 = true; break; }
                    if ((spins & 255u) == 255u && vjf_abort_seen(A.status)) break;
                    __builtin_amdgcn_s_sleep(4);
                }
                if (!ok1) { vjf_status_or(A.status, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_COLUMN); *s_dead = 1; }
            }
            s_ctl[0] = ok1 ? 1 : 0;
            if (ok1) {
                __hip_atomic_store(St + P.off[VJF_SLOT_TR_LOGVAR], new_sig, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(SC + VJF_SC_N_TR, pre_tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (failed) vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED);
        }
        VJF_POST_STAMP(21);
        if (k1_late) {
            __syncthreads();
            if (s_ctl[0] && !failed) store_W();
        }
    }
    leave();
}

// One pass (nsteps <= 0) or the looping form: nsteps steps, one after the other (see VjfPostArgs::nsteps)
__device__ __forceinline__ void vjf_rls_post_loop(const VjfPlan& P, const VjfPostArgs& A, float* lds, int* s_dead, const int role, const int bix) {
    const int steps = A.nsteps > 0 ? A.nsteps : 1;
    if (A.xt && role == 1) {
        // start of a launch: xt = w_chol^T from the state (the caller may have rewritten the blob since the last launch), a share per
        // inverse workgroup; the trial role waits for all of them before its first predictive variance
        const int n = P.n, nwg = 2 * ((n + 31) / 32);
        const float* Wc = A.state + P.off[VJF_SLOT_W_CHOL];
        for (int e = bix * VJF_POST_THREADS + (int)threadIdx.x; e < n * n; e += nwg * VJF_POST_THREADS) {
            const int k = e / n, j = e - k * n;
            vjf_store_wt(A.xt + (size_t)j * n + k, Wc[e]);
        }
        vjf_wg_signal_wt(A.xt_count, (int)threadIdx.x);
    }
    for (int it = 0; it < steps; ++it) {
        const float* red = ((A.step0 + it) & 1) ? A.red2 : A.red;
        vjf_rls_post_body(P, A, lds, s_dead, A.epoch + (unsigned)it, red, A.k1_target + (unsigned)it * A.k1_stride,
                          A.prep_target + (unsigned)it * A.prep_stride, it == 0, role, bix);
        __syncthreads();
        if (*s_dead) break;
    }
}

__global__ __launch_bounds__(VJF_POST_THREADS) void vjf_rls_post_kernel(VjfPlan P, VjfPostArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_dead;                                     // a wait timed out: the looping form stops (status says so)
    if (threadIdx.x == 0) s_dead = 0;
    __syncthreads();
    vjf_rls_post_loop(P, A, lds, &s_dead, A.role, (int)blockIdx.x);
}

// The RLS update of ONE step as one launch (the per-step three-stream route): workgroup 0 is the Cholesky workgroup, workgroup 1
// the y / W workgroup, the rest the inverse workgroups; they hand columns over through the flags as the workgroups of the
// one-launch route do.
template <int DZP>
__global__ __launch_bounds__(VJF_CHOL_THREADS) void vjf_rls_pair_kernel(VjfPlan P, VjfCholArgs C, VjfPostArgs Q) {
    static_assert(VJF_CHOL_THREADS == VJF_POST_THREADS, "one workgroup size for both halves");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_dead;
    if (threadIdx.x == 0) s_dead = 0;
    __syncthreads();
    // (gridDim.x > 2: workgroups 2.. are the inverse workgroups -- the whole RLS update in one launch, one stream)
    if (blockIdx.x == 0) vjf_chol_loop<DZP>(P, C, lds, &s_dead);
    else if (blockIdx.x == 1) vjf_rls_post_loop(P, Q, lds, &s_dead, 2, 0);
    else vjf_rls_post_loop(P, Q, lds, &s_dead, 1, (int)blockIdx.x - 2);
}



// One wavefront that ends when `*count` has reached `target` (bounded): the next kernel of its stream then starts behind the
// producers on another stream without a cross-stream event (6-13 us on this stack).  It holds no LDS and one wave slot, so
// it cannot keep the single-workgroup chain kernels (which need a whole CU's LDS) from being placed.
__global__ __launch_bounds__(64) void vjf_gate_kernel(const unsigned* count, unsigned target, float* status) {
    if (threadIdx.x != 0) return;
    for (unsigned spins = 0; spins < VJF_SPIN_LIMIT; ++spins) {
        const unsigned v = __hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - target) >= 0) return;
        if ((spins & 255u) == 255u && vjf_abort_seen(status)) return;
        __builtin_amdgcn_s_sleep(2);
    }
    vjf_status_or(status, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_GATE);
}

// The other half of such a hand-off: one more arrival at `count`, behind everything its stream has run so far (a kernel of its own,
// so the results of the kernels before it are in memory: a kernel's end writes them back)
__global__ __launch_bounds__(64) void vjf_count_kernel(unsigned* count) {
    if (threadIdx.x == 0) __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

struct VjfResidArgs {
    float* state;
    const float* red;
    double* partial;        // VJF_RESID_BLOCKS partial sums
    int B_total;
    unsigned flags;
};

// q = -2 tr(W^T FDX) + tr(W^T G W) = sum_i W_i . ((G W)_i - 2 FDX_i): workgroup b forms the 16-row strips b, b + 64, .. of
// T = G W as an LDS-tiled product (64 columns of G x 64 columns of W per stage) and contracts them with W on the spot in fp64 (the
// three terms of the residual cancel to a small difference); one fp64 partial per workgroup.
__global__ __launch_bounds__(256) void vjf_resid_kernel(VjfPlan P, VjfResidArgs A) {
    __shared__ float s_g[16][65];
    __shared__ float s_w[64][65];
    __shared__ double s_d[4];
    const int tid = threadIdx.x, n = P.n, dz = P.dz;
    const float* S = A.state;
    const float* W = S + P.off[VJF_SLOT_W_MEAN];
    const float* G = A.red + P.red_G;                            // (both triangles are in the reduce buffer)
    const float* FDX = A.red + P.red_FDX;
    const int r = tid >> 4, cq = (tid & 15) * 4;
    double part = 0.0;
    for (int i0 = blockIdx.x * 16; i0 < n; i0 += VJF_RESID_BLOCKS * 16) {
        for (int c0 = 0; c0 < dz; c0 += 64) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};            // (fp32 products and sums per 16 x 4 patch of T, as the LDS-resident path's matrix-core tiles; fp64 from there)
            for (int j0 = 0; j0 < n; j0 += 64) {
                __syncthreads();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = tid + 256 * q, rr = e >> 6, jj = e & 63;
                    s_g[rr][jj] = (i0 + rr < n && j0 + jj < n) ? G[(size_t)(i0 + rr) * n + j0 + jj] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int e = tid + 256 * q, jj = e >> 6, cc = e & 63;
                    s_w[jj][cc] = (j0 + jj < n && c0 + cc < dz) ? W[(size_t)(j0 + jj) * dz + c0 + cc] : 0.f;
                }
                __syncthreads();
#pragma unroll 8
                for (int jj = 0; jj < 64; ++jj) {
                    const float g = s_g[r][jj];
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[k] = fmaf(g, s_w[jj][cq + k], acc[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + r, c = c0 + cq + k;
                if (i < n && c < dz) {
                    const double w = (double)W[(size_t)i * dz + c];
                    part += w * ((double)acc[k] - 2.0 * (double)FDX[(size_t)i * dz + c]);
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((tid & 63) == 0) s_d[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) A.partial[blockIdx.x] = ((s_d[0] + s_d[1]) + s_d[2]) + s_d[3];
}

// the same partial sums from T = G W given as a matrix (the wide route forms it with its GEMM kernel): fp64 contraction only
__global__ __launch_bounds__(256) void vjf_resid_dot_kernel(VjfPlan P, VjfResidArgs A, const float* T) {
    __shared__ double s_d[4];
    const int tid = threadIdx.x, tot = P.n * P.dz;
    const float* W = A.state + P.off[VJF_SLOT_W_MEAN];
    const float* FDX = A.red + P.red_FDX;
    double part = 0.0;
    for (int e = blockIdx.x * 256 + tid; e < tot; e += VJF_RESID_BLOCKS * 256)
        part += (double)W[e] * ((double)T[e] - 2.0 * (double)FDX[e]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((tid & 63) == 0) s_d[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) A.partial[blockIdx.x] = ((s_d[0] + s_d[1]) + s_d[2]) + s_d[3];
}

// the sum of squares of the residual itself, dx - Phi W (vjf/model.py:373-374), for a rank that holds every trial: R = Phi W comes from
// the GEMM kernel; the difference in fp32 as the reference forms it, the sum in fp64.  Unlike the quadratic form above it does not
// lose the residual under the rounding of G and FDX when the weights reproduce dx almost exactly (B << n).
__global__ __launch_bounds__(256) void vjf_resid_direct_kernel(VjfPlan P, VjfResidArgs A, const float* E, const float* R, int B) {
    __shared__ double s_d[4];
    const int tid = threadIdx.x, dz = P.dz, tot = B * dz;
    double part = 0.0;
    for (int e = blockIdx.x * 256 + tid; e < tot; e += VJF_RESID_BLOCKS * 256) {
        const int b = e / dz, cc = e - b * dz;
        const float r = E[(size_t)b * P.ldE + P.n + cc] - R[e];
        part += (double)r * (double)r;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((tid & 63) == 0) s_d[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) A.partial[blockIdx.x] = ((s_d[0] + s_d[1]) + s_d[2]) + s_d[3];
}

__global__ __launch_bounds__(64) void vjf_sigma_kernel(VjfPlan P, VjfResidArgs A, const int* ok, int direct = 0) {
    // one wavefront: lane b holds partial b (VJF_RESID_BLOCKS == 64), fixed-order xor tree
    double t = A.partial[threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (threadIdx.x != 0) return;
    float* S = A.state;
    float* SC = S + P.off[VJF_SLOT_SCALARS];
    if (!direct) t += (double)A.red[P.red_SC + RS_SDX2];      // (quadratic form: the partials hold -2 tr(W^T FDX) + tr(W^T G W))
    if (t < 0.0) t = 0.0;
    const float Bf = (float)A.B_total;
    const float sig = S[P.off[VJF_SLOT_TR_LOGVAR]];
    const float mse = (float)(t / ((double)Bf * (double)P.dz));
    const float acc = fminf(SC[VJF_SC_N_TR], 500.f), tot = acc + Bf;           // running_var, size_cap=500 (model.py:375)
    S[P.off[VJF_SLOT_TR_LOGVAR]] = logf((acc / tot) * expf(sig) + (Bf / tot) * mse);
    SC[VJF_SC_N_TR] = tot;
    if (ok && ok[0] == 0) vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED);
}
