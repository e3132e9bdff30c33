// vjf_chol_kernel.h -- fast path of the once-per-step serial half for n_rbf <= 224.
//
//   vjf_prep_kernel  (many workgroups): everything element-wise that the step's serial half
//       needs -- finite guards + loss (model.py:138-154), clip + SGD (model.py:210-211),
//       likelihood running variance (likelihood.py:28-40), g = P W + Phi^T dx / v and
//       P += Phi^T Phi / v (module.py:94-96) -- so that ONE compute unit is left with nothing
//       but the dependent chain.
//   vjf_chol_lds_kernel (one workgroup, 8 wavefronts): L = chol(P) (module.py:99),
//       W = P^-1 g (module.py:101), w_chol = L^-T (module.py:102), residual -> state-noise
//       running variance (model.py:373-377).  The matrix lives in LDS as XOR-swizzled 32x32
//       blocks; all block products run on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
//       The two column-sequential pieces (diagonal-block Cholesky, triangular block solve) are
//       written as chains of rank-1 MFMA updates on an accumulator tile: the symmetric tile has
//       row j already spread over the lanes (column index on the lane), which is exactly the
//       A- and B-operand layout, so a column step is  readlane(pivot) -> rsqrt -> scale -> MFMA
//       with no cross-lane data movement.
#pragma once
#include <hip/hip_runtime.h>
#include "vjf_gram_kernel.h"   // vjf_f32x16
#include "vjf_plan.h"

#define VJF_CHOL_THREADS 512
#define VJF_CHOL_MAXBLK 7                 // n <= 224
#define VJF_PREP_ROWS 1                   // rows of P per prep workgroup

// ---------------------------------------------------------------------------------------------
struct VjfPrepArgs {
    float* state;
    const float* red;
    float* gbuf;          // (n, dz) g = P W + FDX / v
    float* aux;           // transposed weight copies, kept in step with the SGD update
    float* loss4;
    int B_total;
    unsigned flags;
    int n_rowblk, n_sgdblk;
    const unsigned* wait_count;   // vjf_prepg_kernel: non-null -> W, sigma come from a kernel on another stream: wait (bounded)
    unsigned wait_target;         //   until *wait_count has reached wait_target, then acquire at agent scope
    int bid0;             // first logical workgroup of this launch: 0 (whole grid, or the RLS-operand rows only)
                          // or n_rowblk (SGD + scalars only) -- the two halves run on different streams in vjf_filter_seq
    // scalar workgroup, vjf_filter_seq only: it ends only when this step's Cholesky kernel (run_word >= run_epoch) and all of its
    // post kernel's workgroups (*start_count >= start_target) are RESIDENT.  The next backward half of the trial kernel waits
    // in-kernel for their results: it must not take the CUs they need before they are placed.
    const unsigned* run_word; unsigned run_epoch;
    const unsigned* start_count; unsigned start_target;
    unsigned* done_count;         // vjf_prepg_kernel: non-null -> += 1 per workgroup once its rows of P and g are in memory
    // Non-finite loss component (vjf/model.py:138-149) on the one-stream route: the first pass leaves the parameters alone and
    // writes the dropped components (bit 0 recon, 1 dynamics, 2 entropy; 0: nothing to replay) and the likelihood log-variance the
    // step started with; the backward half and the gradient sums run again behind it (they return at once on 0), then the second
    // pass (replay_pass) applies the step from the new sums.
    unsigned* replay_mask; float* replay_rho; int replay_pass;
};

// logical grid = n_rowblk + n_sgdblk + 1
__global__ __launch_bounds__(256) void vjf_prep_kernel(VjfPlan P, VjfPrepArgs A) {
    const int tid = threadIdx.x, bid = blockIdx.x + A.bid0;
    float* S = A.state;
    float* SC = S + P.off[VJF_SLOT_SCALARS];
    const float* RSC = A.red + P.red_SCA;                      // the loss sums (RS_LRECON .. RS_SSEY)
    const bool do_sgd = A.flags & VJF_FLAG_SGD, do_upd = A.flags & VJF_FLAG_UPDATE, warm = A.flags & VJF_FLAG_WARM_UP;
    const float Bf = (float)A.B_total, invB = 1.0f / Bf;
    float l_recon = RSC[RS_LRECON] * invB, l_dyn = RSC[RS_LDYN] * invB, ent = RSC[RS_ENT] * invB;
    const bool ok_r = isfinite(l_recon), ok_d = isfinite(l_dyn), ok_h = isfinite(ent);
    const bool grad_ok = ok_r && ok_h && (warm || ok_d);       // see vjf_serial_kernel / DESIGN.md
    // some, not all, of the components in the loss are non-finite: the reference steps along the gradient of the others
    const bool partial = do_sgd && !grad_ok && (ok_r || ok_h || (!warm && ok_d));
    const bool replay = A.replay_mask != nullptr && partial;

    if (bid < A.n_rowblk) {                                    // ---- RLS operands: one row of P per workgroup
        if (!do_upd || warm) return;
        __shared__ float s_part[4 * 32];
        const int n = P.n, dz = P.dz, i = bid;
        const float inv_v = expf(-S[P.off[VJF_SLOT_TR_LOGVAR]]);
        float* Pm = S + P.off[VJF_SLOT_W_PREC];
        const float* Wm = S + P.off[VJF_SLOT_W_MEAN];
        const float* G = A.red + P.red_G;
        const float* FDX = A.red + P.red_FDX;
        // g[i][:] = sum_k P[i][k] W[k][:] : thread k (n <= 224 < 256 on this path) holds one term per output,
        // then wave + workgroup reduction
        const int k = tid;
        float p = 0.f;
        if (k < n) {
            p = Pm[(size_t)i * n + k];
            Pm[(size_t)i * n + k] = p + G[(size_t)i * n + k] * inv_v;      // P += Phi^T Phi / v (module.py:96)
        }
        for (int j = 0; j < dz; ++j) {
            float v = (k < n) ? p * Wm[(size_t)k * dz + j] : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((tid & 63) == 0) s_part[(tid >> 6) * 32 + j] = v;
        }
        __syncthreads();
        if (tid < dz) A.gbuf[(size_t)i * dz + tid] = ((s_part[tid] + s_part[32 + tid]) + s_part[64 + tid]) + s_part[96 + tid] + FDX[(size_t)i * dz + tid] * inv_v;
        return;
    }
    if (bid < A.n_rowblk + A.n_sgdblk) {                       // ---- clip + SGD, tensor by tensor; transposed copies follow
        if (A.replay_pass) { if (__hip_atomic_load(A.replay_mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return; }
        else if (!(do_sgd && grad_ok)) return;
        const float lr_dec = SC[VJF_SC_LR_DEC], lr_rec = SC[VJF_SC_LR_REC];
        const bool freeze = SC[VJF_SC_FREEZE_DEC] != 0.f;
        const int g0 = (bid - A.n_rowblk) * 256 + tid, gs = A.n_sgdblk * 256;
        for (int t = 0; t < P.n_train; ++t) {
            if (P.tr_dec[t] && freeze) continue;
            const float lr = P.tr_dec[t] ? lr_dec : lr_rec;
            const int rows = P.tr_rows[t], cols = P.tr_cols[t], off = P.tr_off[t];
            for (int e = g0; e < rows * cols; e += gs) {
                float g = A.red[off - P.train_off + e] * invB;
                g = fminf(fmaxf(g, -1.f), 1.f);
                const float w = S[off + e] - lr * g;
                S[off + e] = w;
                if (P.tr_aux[t] >= 0) {
                    const int r = e / cols, c = e - r * cols;
                    A.aux[P.tr_aux[t] + (size_t)c * P.tr_auxld[t] + P.tr_auxcol[t] + r] = w;
                }
            }
        }
        return;
    }
    if (A.replay_pass) return;                                 // (the scalars were settled by the first pass)
    if (tid == 0) {                                            // ---- scalars: loss, likelihood log-variance
        if (A.replay_mask) {
            __hip_atomic_store(A.replay_rho, S[P.off[VJF_SLOT_LIK_LOGVAR]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.replay_mask, replay ? ((ok_r ? 0u : 1u) | (ok_d ? 0u : 2u) | (ok_h ? 0u : 4u)) : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!ok_r) l_recon = 0.f;
        if (!ok_d) l_dyn = 0.f;
        if (!ok_h) ent = 0.f;
        float loss = l_recon - ent;
        if (!warm) loss += l_dyn;
        if (A.loss4) { A.loss4[0] = loss; A.loss4[1] = -l_recon; A.loss4[2] = -l_dyn; A.loss4[3] = ent; }
        const unsigned st = (ok_r ? 0u : VJF_STATUS_NONFINITE_RECON) | (ok_d ? 0u : VJF_STATUS_NONFINITE_DYN) |
                            (ok_h ? 0u : VJF_STATUS_NONFINITE_ENT);
        if (st) vjf_status_or(SC + VJF_SC_STATUS, st);
        if (P.lik == VJF_LIK_GAUSSIAN) {
            const float sse_y = RSC[RS_SSEY];
            float rho = S[P.off[VJF_SLOT_LIK_LOGVAR]];
            if (do_sgd && (grad_ok || (replay && ok_r))) {       // (its gradient comes from the reconstruction term alone)
                float g = 0.5f * ((float)P.dy - expf(-rho) * sse_y * invB);
                g = fminf(fmaxf(g, -1.f), 1.f);
                rho -= SC[VJF_SC_LR_LIK] * g;
            }
            if (do_upd) {
                const float mse = sse_y / (Bf * (float)P.dy);
                const float acc = fminf(SC[VJF_SC_N_LIK], 1000.f), tot = acc + Bf;
                rho = logf((acc / tot) * expf(rho) + (Bf / tot) * mse);
                SC[VJF_SC_N_LIK] = tot;
            }
            S[P.off[VJF_SLOT_LIK_LOGVAR]] = rho;
        }
        if (A.run_word) {
            bool there = false;
            for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
                const unsigned r = __hip_atomic_load(A.run_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned q = __hip_atomic_load(A.start_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(r - A.run_epoch) >= 0 && (int)(q - A.start_target) >= 0) { there = true; break; }
                if ((spins & 255u) == 255u && vjf_abort_seen(SC + VJF_SC_STATUS)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (!there) vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_RESIDENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// RLS operands, 16 rows of P per workgroup (replaces the row part of vjf_prep_kernel on the fast path):
//   g[i][:] = sum_k P[i][k] W[k][:] + (Phi^T dx)[i][:] / v     (module.py:94)   on v_mfma_f32_16x16x4_f32, K split over 4 wavefronts
//   P[i][:] += (Phi^T Phi)[i][:] / v                            (module.py:96)   on the rows just read
// grid = ceil(n / 16) workgroups of 256 threads; n % 4 == 0.
#define VJF_PREPG_LDP(n) ((n) + 4)
static inline size_t vjf_prepg_lds_bytes(const VjfPlan& P) { return ((size_t)16 * VJF_PREPG_LDP(P.n) + (size_t)P.n * 17 + 4 * 16 * 17) * 4; }

__global__ __launch_bounds__(256) void vjf_prepg_kernel(VjfPlan P, VjfPrepArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned do_upd = A.flags & VJF_FLAG_UPDATE, warm = A.flags & VJF_FLAG_WARM_UP;
    if (!do_upd || warm) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = P.n, dz = P.dz, i0 = blockIdx.x * 16, ldp = VJF_PREPG_LDP(n);
    float* s_p = lds;                                  // [16][n + 4]  rows of P before the update
    float* s_w = s_p + 16 * ldp;                       // [n][17]      W, columns dz..15 zero
    float* s_r = s_w + (size_t)n * 17;                 // [4][16][17]  per-wavefront partial products
    float* S = A.state;
    if (A.wait_count) {
        if (tid == 0) {
            bool there = false;
            for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
                if ((int)(__hip_atomic_load(A.wait_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - A.wait_target) >= 0) { there = true; break; }
                __builtin_amdgcn_s_sleep(4);
            }
            if (!there) vjf_status_or(S + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_OPERAND);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
    }
    // (a vector load that bypasses L1 / the scalar cache: sigma may have been written while this kernel was already waiting)
    const float inv_v = expf(-__hip_atomic_load(S + P.off[VJF_SLOT_TR_LOGVAR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    float* Pm = S + P.off[VJF_SLOT_W_PREC];
    const float* Wm = S + P.off[VJF_SLOT_W_MEAN];
    const float* G = A.red + P.red_G;
    const float* FDX = A.red + P.red_FDX;
    const int n4 = n >> 2;
    for (int e0 = tid; e0 < 16 * n4; e0 += 4 * 256) {  // 4 float4 of P and of G in flight per thread
        float4 p[4], g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + q * 256, row = e / n4, c4 = (e - row * n4) * 4;
            const bool in = e < 16 * n4 && i0 + row < n;
            const size_t off = in ? (size_t)(i0 + row) * n + c4 : 0;
            p[q] = *reinterpret_cast<const float4*>(Pm + off);
            g[q] = *reinterpret_cast<const float4*>(G + off);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + q * 256, row = e / n4, c4 = (e - row * n4) * 4;
            if (e >= 16 * n4) continue;
            const bool in = i0 + row < n;
            float* d = s_p + row * ldp + c4;
            d[0] = in ? p[q].x : 0.f; d[1] = in ? p[q].y : 0.f; d[2] = in ? p[q].z : 0.f; d[3] = in ? p[q].w : 0.f;
            if (in) {
                float4 o;
                o.x = fmaf(g[q].x, inv_v, p[q].x); o.y = fmaf(g[q].y, inv_v, p[q].y); o.z = fmaf(g[q].z, inv_v, p[q].z); o.w = fmaf(g[q].w, inv_v, p[q].w);
                float* dstp = Pm + (size_t)(i0 + row) * n + c4;
                if (A.done_count) { vjf_store_wt(dstp, o.x); vjf_store_wt(dstp + 1, o.y); vjf_store_wt(dstp + 2, o.z); vjf_store_wt(dstp + 3, o.w); }
                else *reinterpret_cast<float4*>(dstp) = o;
            }
        }
    }
    for (int e = tid; e < n * 16; e += 256) {
        const int k = e >> 4, c = e & 15;
        s_w[k * 17 + c] = c < dz ? Wm[(size_t)k * dz + c] : 0.f;
    }
    __syncthreads();
    {
        const int i = lane & 15, kk = lane >> 4;
        vjf_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s4 = wave; s4 < n4; s4 += 4) {        // k-step s4 covers k = 4 s4 .. 4 s4 + 3
            const float a = s_p[i * ldp + 4 * s4 + kk];
            const float b = s_w[(4 * s4 + kk) * 17 + i];
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) s_r[(wave * 16 + 4 * (lane >> 4) + r) * 17 + (lane & 15)] = acc[r];   // row 4(lane>>4)+r, column lane&15
    }
    __syncthreads();
    for (int e = tid; e < 16 * 16; e += 256) {
        const int r = e >> 4, c = e & 15;
        if (c < dz && i0 + r < n) {
            const float v = ((s_r[r * 17 + c] + s_r[(16 + r) * 17 + c]) + s_r[(32 + r) * 17 + c]) + s_r[(48 + r) * 17 + c];
            const float gv = v + FDX[(size_t)(i0 + r) * dz + c] * inv_v;
            if (A.done_count) vjf_store_wt(A.gbuf + (size_t)(i0 + r) * dz + c, gv); else A.gbuf[(size_t)(i0 + r) * dz + c] = gv;
        }
    }
    if (A.done_count) vjf_wg_signal_wt(A.done_count, tid);
}

// ---------------------------------------------------------------------------------------------
// LDS block helpers.  A 32x32 block is 1024 floats; element (r,c) sits at r*32 + (c ^ r), which
// makes row reads, column reads and the MFMA operand reads bank-conflict free.
__device__ __forceinline__ int vsw(int r, int c) { return r * 32 + (c ^ r); }
__device__ __forceinline__ int vtri(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }
// accumulator layout of v_mfma_f32_32x32x2_f32: column = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
__device__ __forceinline__ int vrow(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

__device__ __forceinline__ float vrl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ float vrsqrt(float d) {               // rsqrt with one Newton step: ~0.5 ulp
    float s = __builtin_amdgcn_rsqf(d);
    return s * fmaf(-0.5f * d * s, s, 1.5f);
}

__device__ __forceinline__ void blk_load(vjf_f32x16& acc, const float* blk, int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = blk[vsw(vrow(r, h), c)];
}
__device__ __forceinline__ void blk_load_t(vjf_f32x16& acc, const float* blk, int lane) {   // acc = blk^T
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = blk[vsw(c, vrow(r, h))];
}
__device__ __forceinline__ void blk_store(const vjf_f32x16& acc, float* blk, int lane) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) blk[vsw(vrow(r, h), c)] = acc[r];
}
// acc += sign * Ab * Bb      (Bt: use Bb^T)
template <bool Bt>
__device__ __forceinline__ void blk_mma(vjf_f32x16& acc, const float* Ab, const float* Bb, float sign, int lane) {
    const int c = lane & 31, h = lane >> 5;
    float a[16], b[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {                   // all 32 operand reads first ...
        const int m = 2 * t + h;
        a[t] = Ab[vsw(c, m)];
        b[t] = Bt ? Bb[vsw(c, m)] : Bb[vsw(m, c)];
    }
    __builtin_amdgcn_sched_barrier(0);               // ... so the 16 MFMAs issue back to back
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sign * a[t], b[t], acc, 0, 0, 0);
}

// acc += A * B with operands given by functors: fa(i, m) = A[i][m], fb(m, j) = B[m][j]  (i, j = lane & 31)
template <class FA, class FB>
__device__ __forceinline__ void blk_mma_f(vjf_f32x16& acc, int lane, FA fa, FB fb) {
    const int c = lane & 31, h = lane >> 5;
    float a[16], b[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int m = 2 * t + h;
        a[t] = fa(c, m);
        b[t] = fb(m, c);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < 16; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
}

// Cholesky of the symmetric tile held in `acc` (one wavefront) together with the inverse of its
// factor: a second accumulator starts as I and receives the same column's rank-1 update
// (R[c][:] -= L[c][j] * X[j][:]), so the two MFMAs of a step overlap in the pipe.
// Writes L (lower part) into `out`, L^-1 (lower) into `inv`.  Returns false on a bad pivot.
// Nothing but the dependent chain sits inside the column loop; the LDS stores follow it.
__device__ __forceinline__ bool potrf_inv_chain(vjf_f32x16& acc, float* out, float* inv, int lane) {
    const int c = lane & 31, h = lane >> 5;
    float lcol[32], xrow[32];
    float dmin = 3.0e38f, slast = 1.f;
    vjf_f32x16 racc;
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[r] = (vrow(r, h) == c) ? 1.f : 0.f;
    // Per column: readlane(pivot) -> rsq -> two scaled rows -> two MFMAs that share the A operand -l.
    // (The inverse's update also clears its own row j -- R[j] -= L[j][j] X[j] = 0 -- which is never read again.)
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int rj = (j & 3) + 4 * (j >> 3), hj = (j >> 2) & 1;      // vrow(rj, hj) == j
        const float d = vrl(acc[rj], j + 32 * hj);
        const float s = __builtin_amdgcn_rsqf(d);
        dmin = fminf(dmin, d);
        slast = s;
        // No per-column lane masks inside the chain: the A operand alone is zeroed on the other k half (which kills that
        // k slot of both products), and entries left of the pivot -- rounding residue of earlier eliminations -- only
        // reach rows / columns < j of the tiles, which are never read again.  The stores below mask.
        const float l = acc[rj] * s;                                    // l[c] = L[c][j] for c >= j on half hj
        const float x = racc[rj] * s;                                   // x[c] = Linv[j][c] on half hj
        const float nl = (h == hj) ? -l : 0.f;
        lcol[j] = l;
        xrow[j] = x;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(nl, l, acc, 0, 0, 0);
        racc = __builtin_amdgcn_mfma_f32_32x32x2f32(nl, x, racc, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int hj = (j >> 2) & 1;
        if (h == hj) {
            if (c >= j) out[vsw(c, j)] = lcol[j];
            inv[vsw(j, c)] = (c <= j) ? xrow[j] : 0.f;
        }
    }
    return (dmin > 0.f) && (slast == slast) && (fabsf(slast) < 3.0e38f);   // positive pivots, no NaN / inf came through
}

// Rank-2 form of the merged chain: two pivots per matrix-core round, so that BOTH k slots of v_mfma_f32_32x32x2 carry a
// column (the rank-1 chain above zeroes one of them) -- 16 rounds of 2 MFMAs instead of 32 steps of 2.
// The tile sits in the accumulator under a symmetric permutation: register r holds logical row 2r on lanes 0..31 and
// logical row 2r + 1 on lanes 32..63 (physical row (r&3) + 8 (r>>2) + 4 half  <->  logical row 2r + half; same map for the
// columns), so the two pivot rows of a round are the two halves of ONE register and land in the A / B operand layout
// (k = lane >> 5) with no data movement; only column 2m scaled by its pivot has to cross to the other half once
// (v_permlane32_swap) to update row 2m + 1 before its own pivot is taken.  A relabelling only: the factor is the lower
// triangular L of the tile in the natural order.  Reads the tile from `dk`, writes L (lower) back and L^-1 (lower) to `inv`.
__device__ __forceinline__ float vlo2both(float v) {            // lanes 0..31 of v on both halves
    const unsigned u = __float_as_uint(v);
    return __uint_as_float(__builtin_amdgcn_permlane32_swap(u, u, false, false)[0]);
}
// `nvalid`: rows / columns of the tile inside the matrix (the rest is the identity padding of the last block): a round whose two
// pivots are padding would scale by 1 and update by 0 -- it is skipped (a uniform branch around the round: the loop stays fully
// unrolled, every register index static), the result is the same bits.
__device__ __forceinline__ bool potrf_inv_chain2(float* dk, float* inv, int lane, int nvalid = 32) {
    const int c = lane & 31, h = lane >> 5;
    const int lc = 2 * ((c & 3) + 4 * (c >> 3)) + ((c >> 2) & 1);   // logical column held by this lane
    vjf_f32x16 acc, racc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = 2 * r + h;                                   // logical row of (register r, half h)
        acc[r] = dk[vsw(lr, lc)];
        racc[r] = (lr == lc) ? 1.f : 0.f;
    }
    float vcol[16], xcol[16];
    float dmin = 3.0e38f, slast = 1.f;
    const int rounds = (nvalid + 1) >> 1;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        vcol[m] = xcol[m] = (lc == 2 * m + h) ? 1.f : 0.f;          // (what a skipped round leaves: the identity)
        if (m < rounds) {                                           // (uniform)
        const int p1 = (m & 3) + 8 * (m >> 2), p2 = p1 + 4;         // physical columns of logical 2m and 2m + 1
        const float d1 = vrl(acc[m], p1);                           // T[2m][2m]
        const float q2 = vrl(acc[m], 32 + p2);                      // T[2m+1][2m+1], before column 2m is eliminated
        const float s1 = __builtin_amdgcn_rsqf(d1);
        const float l1 = acc[m] * s1;                               // lanes 0..31: L[.][2m]
        const float e = vrl(l1, p2);                                // L[2m+1][2m]
        const float t = fmaf(-e, vlo2both(l1), acc[m]);             // lanes 32..63: row 2m+1 with column 2m eliminated
        const float d2 = fmaf(-e, e, q2);
        const float s2 = __builtin_amdgcn_rsqf(d2);
        const float v = h ? t * s2 : l1;                            // L[.][2m] | L[.][2m+1] by half = the k slot
        const float x1 = racc[m] * s1;                              // lanes 0..31: Linv[2m][.]
        const float x2 = fmaf(-e, vlo2both(x1), racc[m]) * s2;      // lanes 32..63: Linv[2m+1][.]
        const float b = h ? x2 : x1;
        const float nv = -v;
        dmin = fminf(dmin, fminf(d1, d2));
        slast = s2;
        vcol[m] = v;
        xcol[m] = b;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(nv, v, acc, 0, 0, 0);
        racc = __builtin_amdgcn_mfma_f32_32x32x2f32(nv, b, racc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int j = 2 * m + h;
        if (lc >= j) dk[vsw(lc, j)] = vcol[m];
        inv[vsw(j, lc)] = (lc <= j) ? xcol[m] : 0.f;
    }
    return (dmin > 0.f) && (slast == slast) && (fabsf(slast) < 3.0e38f);
}

// (Not on the product path either: `tools/chain_follow_bench.hip` and DESIGN.md section 3 "The streamed column chain" -- round 4 built the
//  column loop on these two routines and measured it: correct, and no faster, because the first columns are bound by the helpers' bulk.)
#ifndef VJF_FOLLOW_PHASE
#define VJF_FOLLOW_PHASE 4                // rounds per phase of the follower wavefront (potrf_follow)
#endif
// ---- The streamed column chain: the chain wavefront publishes every round, a FOLLOWER wavefront on another SIMD rides on them.
// A SIMD of this part runs either MFMA or VALU instructions, never both (tools/mfma_valu_overlap.hip: a wavefront's -- or its SIMD
// neighbour's -- VALU work waits while a v_mfma_f32_32x32x2_f32 runs its 16 passes), so whatever else the chain wavefront does is added
// to the dependent chain; what can ride on the chain's rounds belongs on another SIMD.  The follower applies round m of block column k
//   * to T = tile (k+1,k), TRANSPOSED, in the chain's own register layout: the 2x2 step of the inverse's accumulator (same scalars
//     s1, e, s2, same A operand -v): a forward substitution L_{k+1,k} = A_{k+1,k} L_kk^-T that ends with the chain, in place of the
//     16-MFMA product with the finished inverse behind it;
//   * to N = tile (k+1,k+1): the finished column pair of L_{k+1,k} is BOTH operands of one rank-2 update N -= l l^T -- the trailing
//     update of the next diagonal block from registers, in place of a second 16-MFMA product through LDS.
// The chain wavefront is left with its two MFMAs a round and three LDS stores: the pair of scaled columns (64 floats), the three
// scalars, and -- one round later, when an s_waitcnt on them costs nothing -- the count of published rounds.
// (the ring and its count are handed over as LDS-typed pointers: through generic ones hipcc's backend fails on this code with
//  "Illegal instruction detected: Operand has incorrect register class ... $src_shared_base", DESIGN.md section 3 "Toolchain note")
typedef __attribute__((address_space(3))) float vjf_lds_f;
typedef __attribute__((address_space(3))) volatile int vjf_lds_vi;
__device__ __forceinline__ bool potrf_inv_chain2_bcast(float* dk, float* inv, int lane, vjf_lds_f* ring_v, vjf_lds_f* ring_sc, vjf_lds_vi* rnd, const int rnd_base) {
    const int c = lane & 31, h = lane >> 5;
    const int lc = 2 * ((c & 3) + 4 * (c >> 3)) + ((c >> 2) & 1);   // logical column held by this lane
    vjf_f32x16 acc, racc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int lr = 2 * r + h;                                   // logical row of (register r, half h)
        acc[r] = dk[vsw(lr, lc)];
        racc[r] = (lr == lc) ? 1.f : 0.f;
    }
    float vcol[16], xcol[16];
    float dmin = 3.0e38f, slast = 1.f;
    const int l3 = lane % 3;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int p1 = (m & 3) + 8 * (m >> 2), p2 = p1 + 4;         // physical columns of logical 2m and 2m + 1
        const float d1 = vrl(acc[m], p1);                           // T[2m][2m]
        const float q2 = vrl(acc[m], 32 + p2);                      // T[2m+1][2m+1], before column 2m is eliminated
        const float s1 = __builtin_amdgcn_rsqf(d1);
        const float l1 = acc[m] * s1;                               // lanes 0..31: L[.][2m]
        const float e = vrl(l1, p2);                                // L[2m+1][2m]
        const float t = fmaf(-e, vlo2both(l1), acc[m]);             // lanes 32..63: row 2m+1 with column 2m eliminated
        const float d2 = fmaf(-e, e, q2);
        const float s2 = __builtin_amdgcn_rsqf(d2);
        const float v = h ? t * s2 : l1;                            // L[.][2m] | L[.][2m+1] by half = the k slot
        // Publish the round: every lane stores (no exec masks, no branches: lanes that share a word store the same value), and the
        // count of round m - 1 goes out with round m's data -- the LDS executes a wavefront's instructions in order, and by now
        // those stores are hundreds of cycles old anyway; nothing here waits.
        if (m > 0) *rnd = rnd_base + m;
        ring_v[m * 64 + lane] = v;
        ring_sc[m * 3 + l3] = l3 == 0 ? s1 : l3 == 1 ? e : s2;
        const float x1 = racc[m] * s1;                              // lanes 0..31: Linv[2m][.]
        const float x2 = fmaf(-e, vlo2both(x1), racc[m]) * s2;      // lanes 32..63: Linv[2m+1][.]
        const float b = h ? x2 : x1;
        const float nv = -v;
        dmin = fminf(dmin, fminf(d1, d2));
        slast = s2;
        vcol[m] = v;
        xcol[m] = b;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(nv, v, acc, 0, 0, 0);
        racc = __builtin_amdgcn_mfma_f32_32x32x2f32(nv, b, racc, 0, 0, 0);
    }
    *rnd = rnd_base + 16;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int j = 2 * m + h;
        if (lc >= j) dk[vsw(lc, j)] = vcol[m];
        inv[vsw(j, lc)] = (lc <= j) ? xcol[m] : 0.f;
    }
    return (dmin > 0.f) && (slast == slast) && (fabsf(slast) < 3.0e38f);
}
// The follower's side: t1 = tile (k+1,k) (natural order, as panel_tile leaves it; L_{k+1,k} on return), n11 = tile (k+1,k+1) (updated on
// return, where the next chain loads its block from).  Returns false if a published round did not come (`alive()` false or the bound).
template <class AliveFn>
__device__ __forceinline__ bool potrf_follow(float* t1, float* n11, int lane, const vjf_lds_f* ring_v, const vjf_lds_f* ring_sc, vjf_lds_vi* rnd, const int rnd_base, AliveFn alive) {
    const int c = lane & 31, h = lane >> 5;
    const int lc = 2 * ((c & 3) + 4 * (c >> 3)) + ((c >> 2) & 1);
    vjf_f32x16 tacc, nacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        tacc[r] = t1[vsw(lc, 2 * r + h)];                           // T^T: row 2r + h = column 2r + h of the tile
        nacc[r] = n11[vsw(2 * r + h, lc)];
    }
    // The rounds in PHASES of VJF_FOLLOW_PHASE: one poll for the phase's last round, its operands read together (one LDS round trip),
    // then its rounds as straight-line code -- control flow between single rounds made the compiler park both accumulators in VGPRs at
    // every join and sink the operand reads behind the poll (tools/chain_follow_bench.hip: 410 cycles a round against 150 here).  A
    // follower that joins late (the product: its tiles arrive ~2 us into the chain) runs through the published phases without waiting;
    // when the chain ends, at most one phase is left to do.
    constexpr int PH = VJF_FOLLOW_PHASE;
    float tcol[16];
    bool ok = true;
    unsigned spins = 0;
#pragma unroll
    for (int p0 = 0; p0 < 16; p0 += PH) {
        while (ok && *rnd - rnd_base < p0 + PH) {                   // (uniform)
            if ((++spins & 63u) == 0u && (spins > (1u << 22) || !alive())) ok = false;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        float vv[PH], s1v[PH], ev[PH], s2v[PH];
#pragma unroll
        for (int q = 0; q < PH; ++q) { const int j = p0 + q; vv[q] = ring_v[j * 64 + lane]; s1v[q] = ring_sc[j * 3]; ev[q] = ring_sc[j * 3 + 1]; s2v[q] = ring_sc[j * 3 + 2]; }
#pragma unroll
        for (int q = 0; q < PH; ++q) {
            const int j = p0 + q;
            const float y1 = tacc[j] * s1v[q];                              // lanes 0..31: L_{k+1,k}[.][2j]
            const float y2 = fmaf(-ev[q], vlo2both(y1), tacc[j]) * s2v[q];   // lanes 32..63: L_{k+1,k}[.][2j+1]
            const float y = h ? y2 : y1;
            tcol[j] = y;
            tacc = __builtin_amdgcn_mfma_f32_32x32x2f32(-vv[q], y, tacc, 0, 0, 0);
            nacc = __builtin_amdgcn_mfma_f32_32x32x2f32(-y, y, nacc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) t1[vsw(lc, 2 * m + h)] = tcol[m];
#pragma unroll
    for (int r = 0; r < 16; ++r) n11[vsw(2 * r + h, lc)] = nacc[r];
    return ok;
}

// (Not on the product path: kept for `tools/potrf_chain_bench3.hip`, which times the alternatives the rank-2 chain above was chosen
//  against -- potrf alone 5088 cycles, inverse 4868, panel solve 4112, rank-2 chain with the inverse 5292.)
// The same column chain split in three, so that each is a pure one-MFMA-per-step dependent chain (the merged chain
// above pays the compiler's MFMA->VALU wait states twice per step: ~290 cycles/step against ~127 + ~94 + ~94 here)
// and the inverse / the panel solves run on other wavefronts beside it.
//   potrf_chain : L = chol(tile) -> out (lower part), pivot scales 1 / L[j][j] -> piv[0..32)
//   inv_chain   : L^-1 (lower) -> inv, from L and the pivot scales (same arithmetic as the merged chain)
//   trsm_chain  : in place  A_ik -> L_ik = A_ik L_kk^-T  by forward substitution on the transposed tile
__device__ __forceinline__ bool potrf_chain(vjf_f32x16& acc, float* out, float* piv, int lane) {
    const int c = lane & 31, h = lane >> 5;
    float lcol[32], sv[32];
    float dmin = 3.0e38f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int rj = (j & 3) + 4 * (j >> 3), hj = (j >> 2) & 1;      // vrow(rj, hj) == j
        const float d = vrl(acc[rj], j + 32 * hj);
        const float s = __builtin_amdgcn_rsqf(d);
        dmin = fminf(dmin, d);
        sv[j] = s;
        const float l = acc[rj] * s;                                    // l[c] = L[c][j] for c >= j on half hj
        const float nl = (h == hj) ? -l : 0.f;                          // (see potrf_inv_chain on the missing masks)
        lcol[j] = l;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(nl, l, acc, 0, 0, 0);
    }
    float pv = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int hj = (j >> 2) & 1;
        if (h == hj && c >= j) out[vsw(c, j)] = lcol[j];
        pv = (lane == j) ? sv[j] : pv;
    }
    if (lane < 32) piv[lane] = pv;
    const float slast = sv[31];
    return (dmin > 0.f) && (slast == slast) && (fabsf(slast) < 3.0e38f);   // positive pivots, no NaN / inf came through
}

__device__ __forceinline__ void inv_chain(const float* Lk, const float* piv, float* inv, int lane) {
    const int c = lane & 31, h = lane >> 5;
    float la[32], sv[32], xrow[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int hj = (j >> 2) & 1;
        const float l = Lk[vsw(c, j)];
        la[j] = ((h == hj) && (c >= j)) ? -l : 0.f;
        sv[j] = piv[j];
    }
    vjf_f32x16 racc;
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[r] = (vrow(r, h) == c) ? 1.f : 0.f;
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int rj = (j & 3) + 4 * (j >> 3);
        const float x = racc[rj] * sv[j];                               // x[c] = Linv[j][c] on half (j >> 2) & 1 (la is 0 on the other)
        xrow[j] = x;
        racc = __builtin_amdgcn_mfma_f32_32x32x2f32(la[j], x, racc, 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int hj = (j >> 2) & 1;
        if (h == hj) inv[vsw(j, c)] = (c <= j) ? xrow[j] : 0.f;
    }
}

__device__ __forceinline__ void trsm_chain(float* pb, const float* Lk, const float* piv, int lane) {
    const int c = lane & 31, h = lane >> 5;
    float la[32], sv[32], lrow[32];
    vjf_f32x16 acc;
    blk_load_t(acc, pb, lane);                                          // acc = A_ik^T: row j of it is column j of A_ik
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int hj = (j >> 2) & 1;
        const float l = Lk[vsw(c, j)];
        la[j] = ((h == hj) && (c > j)) ? -l : 0.f;
        sv[j] = piv[j];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int rj = (j & 3) + 4 * (j >> 3);
        const float b = acc[rj] * sv[j];                                // b[c] = L_ik[c][j] on half (j >> 2) & 1 (la is 0 on the other)
        lrow[j] = b;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(la[j], b, acc, 0, 0, 0);   // A_ik^T[i][:] -= L_kk[i][j] L_ik[:][j], i > j
    }
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int hj = (j >> 2) & 1;
        if (h == hj) pb[vsw(c, j)] = lrow[j];
    }
}

struct VjfCholArgs {
    float* state;
    const float* red;
    const float* gbuf;     // from vjf_prep_kernel
    int B_total;
    unsigned flags;
    unsigned long long* stamps;   // diagnostic only (null in normal runs): s_memrealtime (100 MHz, one clock for the whole device) at phase boundaries
    float* dinv_out;       // post mode: nbl blocks (32x32 row-major) of inverted diagonal blocks for vjf_rls_post_kernel
    int* ok_out;           // post mode: 1 = factor valid
    int post;              // 1: stop after L and the inverted diagonal blocks; the many-CU post kernels do the rest
    float* lscr;           // post mode: (n, n) scratch that receives L column by column while the factorisation runs
    unsigned* flags_out;   // post mode: flags_out[k] = (epoch << 1) | failed once column k of L and Dinv_k are in global memory:
    unsigned epoch;        //   vjf_rls_post_kernel, launched beside this kernel, consumes the columns as they appear
    float* pscr;           // post mode: lower 32x32 blocks of P (with the identity padding), block after block, row-major: written
                           //   here with P_new; self_prep reads P_old from it (the copy a Cholesky kernel left for the next one)
    int self_prep;         // post mode: 1 = P_new = P_old (pscr) + Phi^T Phi / v is formed HERE, in registers, once sigma of the
                           //   previous step is there (wait_count reaches wait_target): the operand kernel that updates the state's
                           //   P and forms g runs beside this kernel, on the post kernel's stream, instead of before it
    const unsigned* wait_count; unsigned wait_target;
    int src_state;         // self_prep: P_old comes from the state's P (first step of a sequence) instead of pscr
    // looping form (the Cholesky role of the one-launch route, vjf_mega_kernel.h): nsteps > 0 -> ONE launch runs the factorisations of nsteps consecutive steps on its CU
    // (a kernel of this size is not placed while trial-kernel workgroups hold LDS on every CU; resident, it starts the moment
    // sigma arrives).  Step `it`: epoch + it, statistics in red (even step0 + it) or red2 (odd), ready when *stat_count has
    // reached stat_target + it * stat_stride; sigma when *wait_count has reached wait_target + it * wait_stride.
    int nsteps, step0;
    unsigned inject_epoch; // test hook (VJF_DEBUG_INJECT=k): at this epoch the statistics wait is reported as timed out
    const float* red2;
    const unsigned* stat_count; unsigned stat_target, stat_stride, wait_stride;
    const unsigned long long* sig_word;   // non-null (self_prep): sigma of the previous step arrives as ONE 8-byte word {epoch, bits} from the
                                          //   y / W loop; *wait_count is then only awaited before the first column goes out (the scratch
                                          //   copies of L and the inverted diagonal blocks must have been read by everybody)
    int no_triclean;       // post mode: the caller clears the zero halves of w_chol / w_pchol itself (vjf_triclean_kernel)
                           //            (vjf_rls_post_kernel copies it to w_pchol once the factor is known to be good)
};

#define VJF_STAMP(i)                                                                        \
    do {                                                                                    \
        if (A.stamps && tid == 0) {                                                         \
            unsigned long long t_;                                                          \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
            A.stamps[((it_epoch & 7u) << 5) * (A.nsteps > 0 ? 1 : 0) + (i)] = t_;              \
        }                                                                                   \
    } while (0)

static inline int vjf_chol_dzp(int dz) { return dz <= 4 ? 4 : dz <= 8 ? 8 : dz <= 12 ? 12 : dz <= 16 ? 16 : 32; }
static inline size_t vjf_chol_lds_bytes(const VjfPlan& P) {
    const int nbl = (P.n + 31) / 32;
    const size_t blocks = (size_t)(nbl * (nbl + 1) / 2 + nbl) * 1024;
    const int dzp = vjf_chol_dzp(P.dz);
    return (blocks + (size_t)nbl * 32 * dzp * (dzp <= 16 ? 1 : 2) + 192) * 4;   // dzp = 32: y = L^-1 g has a region of its own
}
static inline bool vjf_chol_lds_ok(const VjfPlan& P) {
    return (P.n + 31) / 32 <= VJF_CHOL_MAXBLK && P.n % 4 == 0 && P.dz <= 32 && vjf_chol_lds_bytes(P) <= 160 * 1024 - 512;
}

#define VJF_CHOL_Q 14          // float4 chunks per thread when sweeping the <= 28 lower blocks

// acc[0..DZP) += x * row[0..DZP)   (row: 16-byte aligned LDS, same address on every lane => broadcast)
template <int DZP>
__device__ __forceinline__ void axpy_row(float (&acc)[DZP], float x, const float* row) {
#pragma unroll
    for (int j = 0; j < DZP; j += 4) {
        const float4 w = *reinterpret_cast<const float4*>(row + j);
        acc[j] = fmaf(x, w.x, acc[j]); acc[j + 1] = fmaf(x, w.y, acc[j + 1]);
        acc[j + 2] = fmaf(x, w.z, acc[j + 2]); acc[j + 3] = fmaf(x, w.w, acc[j + 3]);
    }
}

template <int DZP>
__device__ __forceinline__ void vjf_chol_body(const VjfPlan& P, const VjfCholArgs& A, float* lds, int* s_dead, const unsigned it_epoch,
                                              const float* it_red, const unsigned it_wait_target, const unsigned it_stat_target,
                                              const bool it_src_state) {
    int tid = threadIdx.x;
    // (the looping form calls this body once per step: without the barrier the compiler hoists every lane-dependent address
    //  of the body out of that loop and spills hundreds of registers)
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int n = P.n, dz = P.dz;
    const int nbl = (n + 31) / 32, npad = nbl * 32, ntri = nbl * (nbl + 1) / 2;
    float* S = A.state;
    float* SC = S + P.off[VJF_SLOT_SCALARS];
    const bool do_upd = A.flags & VJF_FLAG_UPDATE, warm = A.flags & VJF_FLAG_WARM_UP;
    if (!do_upd) return;
    if (A.post && warm) return;                      // no RLS in warm-up; the residual / sigma kernels run on their own
    // "running": the caller keeps the post kernel (2 nbl + 1 workgroups that each take a whole CU's LDS) behind a one-wavefront
    // gate on this word, so that they do not sit on 15 CUs before there is anything for them to do
    // (self_prep: the word is stored once this step's operands are in registers -- the operand kernel, which overwrites the
    //  state's P, starts behind a gate on it)
    if (A.post && !A.self_prep && tid == 0) __hip_atomic_store(A.flags_out + VJF_CHOL_MAXBLK + 2, it_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float* s_blk = lds;                               // ntri blocks: lower block triangle of P -> L -> L^-1
    float* s_aux = s_blk + (size_t)ntri * 1024;       // nbl blocks: inverted diagonal blocks of L; later scratch
    float* s_g = s_aux + (size_t)nbl * 1024;          // npad x DZP  g, later W
    // (y = L^-1 g overwrites g in place; for DZP = 32 a second npad x DZP region behind g is reserved)
    int* s_flag = (int*)(s_g + (size_t)npad * DZP * (DZP <= 16 ? 1 : 2));   // [0] ok
    double* s_d = (double*)(s_flag + 8);              // 16 doubles for the final reduction
    int* s_bi = s_flag + 64;                          // block-row / block-column of lower block b
    int* s_bj = s_bi + 32;

    float* Wm = S + P.off[VJF_SLOT_W_MEAN];
    float* Wc = S + P.off[VJF_SLOT_W_CHOL];
    float* Pm = S + P.off[VJF_SLOT_W_PREC];
    float* Lm = S + P.off[VJF_SLOT_W_PCHOL];
    const float* G = it_red + P.red_G;
    const float* FDX = it_red + P.red_FDX;
    float sig = S[P.off[VJF_SLOT_TR_LOGVAR]];
    const float Bf = (float)A.B_total;
    unsigned st = 0;
    const bool sp = A.post && A.self_prep;
    VJF_STAMP(0);
    if (tid < ntri) {
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= tid) ++bi;
        s_bi[tid] = bi;
        s_bj[tid] = tid - bi * (bi + 1) / 2;
    }
    if (tid == 0) s_flag[0] = 1;
    if (tid < 16) s_flag[128 + tid] = 0;                // the column loop's hand-off words (below)
    __syncthreads();
    // (the statistics of the step: the Gram role's write-through stores, read with sc1 loads behind their count -- an acquire as
    //  well only in the VJF_HANDOFF_ACQUIRE=1 form.  The wait sits below, between this thread's loads of P, which do not depend on
    //  the statistics, and its loads of G: at the first step of a launch nothing else hides the 108 KB of P)
    auto stat_wait = [&]() {
        if (A.stat_count && (!vjf_wg_wait_sc1(A.stat_count, it_stat_target, tid, SC + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u) || (A.inject_epoch && tid == 0 && it_epoch == A.inject_epoch))) {
            vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_STATS);
            *s_dead = 1;
        }
    };
    if (warm) stat_wait();

    if (!warm) {
        // ---- load the lower block triangle of P_new (vjf_prep_kernel already added Phi^T Phi / v).  Wavefront 0 takes the
        //      first diagonal block alone and starts its column chain; the other seven bring in the rest meanwhile.
        //      All loads of a thread are issued before its first LDS store.
        auto diag_chain = [&](int k) {                                  // L_kk and L_kk^-1 in one chain (one wavefront)
            float* dk = s_blk + (size_t)vtri(k, k) * 1024;
            const bool good = potrf_inv_chain2(dk, s_aux + (size_t)k * 1024, lane, min(32, n - 32 * k));
            if (!good && lane == 0) s_flag[0] = 0;
            return good;
        };
        bool chain_good = true;                                         // wavefront 0: every pivot so far was positive
        auto pad4 = [](int gi, int gj) {                                // identity padding outside the matrix
            return make_float4(gi == gj ? 1.f : 0.f, gi == gj + 1 ? 1.f : 0.f, gi == gj + 2 ? 1.f : 0.f, gi == gj + 3 ? 1.f : 0.f);
        };
        // self_prep: every thread first issues its loads of P_old (pscr) and of G, then the workgroup waits for sigma of the
        // previous step, and P_new = P_old + G / v is formed in the registers (fmaf, as vjf_prepg_kernel forms the state's P).
        // Either way the blocks of P_new go to pscr for the next step's kernel.
        bool lscr_guard = false;                                        // the post workgroups' exit count is still to be checked
        auto sigma_wait = [&]() {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // operands in registers: the state's P may now be overwritten
            __syncthreads();
            if (tid == 0) __hip_atomic_store(A.flags_out + VJF_CHOL_MAXBLK + 2, it_epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            vjf_chaos(tid, A.wait_count, 1);
            if (A.sig_word && it_wait_target != 0u) {
                // sigma inside the hand-off word: one poll, no second load; the exit count of the post workgroups is checked by the
                // wavefront that writes the first column out (below)
                if (tid == 0) {
                    bool there = false;
                    unsigned bits = 0u;
                    for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
                        const unsigned long long v = __hip_atomic_load(A.sig_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((unsigned)(v >> 32) == it_epoch - 1u) { there = true; bits = (unsigned)v; break; }
                        if ((spins & 255u) == 255u && vjf_abort_seen(SC + VJF_SC_STATUS)) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    s_flag[2] = (int)bits; s_flag[3] = there ? 1 : 0;
                }
                __syncthreads();
                sig = __uint_as_float((unsigned)s_flag[2]);
                if (!s_flag[3] && tid == 0) { vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_SIGMA); *s_dead = 1; }
                lscr_guard = true;
                return;
            }
            // (no acquire: the one thing read behind this wait is sigma, with an sc1 load)
            if (!vjf_wg_wait_sc1(A.wait_count, it_wait_target, tid, SC + VJF_SC_STATUS, (A.flags & VJF_FLAG_HANDOFF_ACQUIRE) != 0u)) { vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_SIGMA); *s_dead = 1; }
            sig = __hip_atomic_load(S + P.off[VJF_SLOT_TR_LOGVAR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        typedef unsigned chol_u4 __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t r_G = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(G), 0, 0x7fffffff, 0x00020000);
        auto g4 = [&](int gi, int gj) {                                 // 4 entries of G, zero outside the matrix
            if (!(gi < n && gj < n)) return make_float4(0.f, 0.f, 0.f, 0.f);
            if (A.stat_count) {                                         // (another role's stores of this launch: sc1, past this CU's L1)
                const chol_u4 v4 = __builtin_amdgcn_raw_buffer_load_b128(r_G, (gi * n + gj) * 4, 0, 16);
                return make_float4(__uint_as_float(v4[0]), __uint_as_float(v4[1]), __uint_as_float(v4[2]), __uint_as_float(v4[3]));
            }
            return *reinterpret_cast<const float4*>(G + (size_t)gi * n + gj);
        };
        constexpr int NT = VJF_CHOL_THREADS - 64, NQ = ((VJF_CHOL_MAXBLK * (VJF_CHOL_MAXBLK + 1) / 2 - 1) * 256 + NT - 1) / NT;
        const int t2 = tid - 64;
        float4 v[NQ], g[NQ];                                            // (wavefront 0 uses the first four)
        auto idx_of = [&](int q) { return wave == 0 ? (q < 4 ? lane + 64 * q : ntri * 256) : 256 + t2 + q * NT; };
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = idx_of(q);
            v[q] = make_float4(0.f, 0.f, 0.f, 0.f); g[q] = v[q];
            if (idx < ntri * 256) {
                const int b = idx >> 8, r = (idx >> 3) & 31, c4 = (idx & 7) * 4;
                const int gi = s_bi[b] * 32 + r, gj = s_bj[b] * 32 + c4;
                if (sp && !it_src_state) v[q] = *reinterpret_cast<const float4*>(A.pscr + (size_t)idx * 4);
                else v[q] = (gi < n && gj < n) ? *reinterpret_cast<const float4*>(Pm + (size_t)gi * n + gj) : pad4(gi, gj);
            }
        }
        stat_wait();
        if (sp) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int idx = idx_of(q);
                if (idx < ntri * 256) {
                    const int b = idx >> 8, r = (idx >> 3) & 31, c4 = (idx & 7) * 4;
                    g[q] = g4(s_bi[b] * 32 + r, s_bj[b] * 32 + c4);
                }
            }
        }
        if (sp) {
            sigma_wait();
            const float inv_v = expf(-sig);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                v[q].x = fmaf(g[q].x, inv_v, v[q].x); v[q].y = fmaf(g[q].y, inv_v, v[q].y);
                v[q].z = fmaf(g[q].z, inv_v, v[q].z); v[q].w = fmaf(g[q].w, inv_v, v[q].w);
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = idx_of(q);
            if (idx < ntri * 256) {
                const int b = idx >> 8, r = (idx >> 3) & 31, c4 = (idx & 7) * 4;
                float* blk = s_blk + (size_t)b * 1024;
                blk[vsw(r, c4)] = v[q].x; blk[vsw(r, c4 + 1)] = v[q].y; blk[vsw(r, c4 + 2)] = v[q].z; blk[vsw(r, c4 + 3)] = v[q].w;
                if (A.post) *reinterpret_cast<float4*>(A.pscr + (size_t)idx * 4) = v[q];
            }
        }
        if (wave == 0) chain_good = diag_chain(0);
        else if (!A.post)                                               // (post mode: g goes to vjf_rls_post_kernel, not here)
            for (int e = t2; e < npad * DZP; e += NT) {
                const int r = e / DZP, j = e - r * DZP;
                s_g[e] = (r < n && j < dz) ? A.gbuf[(size_t)r * dz + j] : 0.f;
            }
        __syncthreads();
        VJF_STAMP(1);

        // ---- blocked right-looking Cholesky with look-ahead: while wavefronts 1..7 finish the trailing update of
        //      step k, wavefront 0 updates block (k+1,k+1) first and runs the next diagonal chain
        auto trail = [&](int k, int t) {                                // A_ij -= L_ik L_jk^T for the t-th lower block
            const int bi = k + 1 + s_bi[t], bj = k + 1 + s_bj[t];
            float* cb = s_blk + (size_t)vtri(bi, bj) * 1024;
            vjf_f32x16 acc;
            blk_load(acc, cb, lane);
            blk_mma<true>(acc, s_blk + (size_t)vtri(bi, k) * 1024, s_blk + (size_t)vtri(bj, k) * 1024, -1.f, lane);
            blk_store(acc, cb, lane);
        };
        // post mode: one finished 32x32 block out to global memory (one wavefront, 4 float4 per lane)
        // Write-through (sc1) 16-byte stores: the bytes are in memory, visible to every XCD, once the storing wavefront's vmcnt
        // has drained -- no release fence (cdna guide, Guideline 16 R1).  The asm store is not counted by the compiler: the
        // publishing code below drains it by hand.
        auto put_block = [&](const float* blk, float* dst, int ld, int gi0, int gj0, bool lower_only) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane + 64 * q, r = idx >> 3, c4 = (idx & 7) * 4;
                vjf_f32x4 o;
                o[0] = (!lower_only || c4 <= r) ? blk[vsw(r, c4)] : 0.f;
                o[1] = (!lower_only || c4 + 1 <= r) ? blk[vsw(r, c4 + 1)] : 0.f;
                o[2] = (!lower_only || c4 + 2 <= r) ? blk[vsw(r, c4 + 2)] : 0.f;
                o[3] = (!lower_only || c4 + 3 <= r) ? blk[vsw(r, c4 + 3)] : 0.f;
                if (gi0 + r < ld && gj0 + c4 < ld) {
                    float* p = dst + (size_t)(gi0 + r) * ld + gj0 + c4;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(o) : "memory");
                }
            }
        };
        auto publish = [&](int k0, int k1, unsigned fail) {             // one wavefront: its stores drained, then the flags
            vjf_chaos(lane, A.flags_out + k0, 2);                       // (diagnostic builds: the wavefront is held)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane >= k0 && lane < k1) __hip_atomic_store(A.flags_out + lane, (it_epoch << 1) | fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        // ---- The column loop without a workgroup barrier in it.  The dependent chain -- Dinv_k from the factor chain of block (k,k),
        //      the one panel tile L_{k+1,k} = A_{k+1,k} Dinv_k^T, the last update of block (k+1,k+1), the next chain -- runs on
        //      wavefront 0 alone, back to back (3.4 us a column); it never waits for the bulk of a column's multiply-adds.  Those --
        //      the other panel tiles and the trailing updates of column k -- belong to six helper wavefronts (1-3, 5-7: the three
        //      other SIMDs), which work one column behind it in two stages per column (panels | trailing tiles) separated by a
        //      counter barrier of their own, and which take the two tiles the chain needs next, (k+2,k+1) and (k+2,k+2), first.
        //      Wavefront 4 (the chain's SIMD: no matrix-core work) writes finished columns out.  All hand-offs are words in LDS:
        //        c_chain  = columns whose Dinv is in LDS             (wavefront 0)
        //        c_p1     = columns whose tile L_{k+1,k} is in LDS   (wavefront 0)
        //        c_a, c_b = columns k whose update of tile (k+2,k+1) / (k+2,k+2) is done (helpers)
        //        c_hb[h]  = stages helper h has completed: stage s is complete when every c_hb[h] > s
        //      Every update of a tile is applied in column order by construction (stage barriers), so the bits do not depend on
        //      timing.  Polls are bounded: a logic error shows as a failed factorisation, not as a hang.
        volatile int* s_ctl = s_flag + 128;
        enum { C_CHAIN = 0, C_P1 = 1, C_A = 2, C_B = 3, C_HB = 4 };
        volatile int* v_ok = s_flag;                                   // [0]: 1 while every pivot was positive (wavefront 0 clears it)
        auto lds_wait = [&](int w, int target) {                       // one wavefront: all lanes poll the same word
            for (unsigned spins = 0; spins < (1u << 22); ++spins) {
                if (s_ctl[w] >= target || !v_ok[0]) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (s_ctl[w] < target && v_ok[0]) v_ok[0] = 0;             // (cannot happen: ends the step as a failed factorisation)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        auto lds_post = [&](int w, int v) {                            // this wavefront's LDS writes first, then the word
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) s_ctl[w] = v;
        };
        auto hb_arrive = [&](int hw, int stages_done) {                // a helper's stage is done: its own word (no atomic: one writer)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) s_ctl[C_HB + hw] = stages_done;
        };
        auto hb_wait = [&](int stages_done) {                          // every helper has that many stages behind it
            for (unsigned spins = 0; spins < (1u << 22); ++spins) {
                int lo = s_ctl[C_HB];
#pragma unroll
                for (int h2 = 1; h2 < 6; ++h2) lo = min(lo, (int)s_ctl[C_HB + h2]);
                if (lo >= stages_done || !v_ok[0]) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        auto panel_tile = [&](int bi, int k) {                         // L_ik = A_ik Dinv_k^T, in place (one wavefront reads all of it first)
            float* pb = s_blk + (size_t)vtri(bi, k) * 1024;
            vjf_f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            blk_mma<true>(acc, pb, s_aux + (size_t)k * 1024, 1.f, lane);
            blk_store(acc, pb, lane);
        };
        auto trail_tile = [&](int bi, int bj, int k) {                 // A_ij -= L_ik L_jk^T
            float* cb = s_blk + (size_t)vtri(bi, bj) * 1024;
            vjf_f32x16 acc;
            blk_load(acc, cb, lane);
            blk_mma<true>(acc, s_blk + (size_t)vtri(bi, k) * 1024, s_blk + (size_t)vtri(bj, k) * 1024, -1.f, lane);
            blk_store(acc, cb, lane);
        };
        int kdone = 0;                                                  // wavefront 4: columns written out (their flags stored)
        if (wave == 0) {
            // Nothing but the chain's own work on this wavefront's path: it is the only one that can fail a pivot, so the verdict
            // stays in a register (v_ok[0] re-read from LDS was a round trip at each of three places per column), and the two words
            // it waits for per column -- both posted by the helpers ~2 us earlier -- are read together, once.
            auto wait_ab = [&](int target) {                            // tiles (k+1,k) and (k+1,k+1) carry the updates of columns < k
                for (unsigned spins = 0; spins < (1u << 22); ++spins) {
                    const int a = s_ctl[C_A], b = s_ctl[C_B];
                    if (a >= target && b >= target) { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); return true; }
                    if (!v_ok[0]) return false;
                    __builtin_amdgcn_s_sleep(1);
                }
                v_ok[0] = 0;                                            // (cannot happen: ends the step as a failed factorisation)
                return false;
            };
            bool good = chain_good;
            for (int k = 0; k < nbl && good; ++k) {                     // (chain(0) ran before the barrier above)
                lds_post(C_CHAIN, k + 1);
                if (k + 1 < nbl) {
                    if (k >= 1 && !wait_ab(k)) break;
                    panel_tile(k + 1, k);
                    lds_post(C_P1, k + 1);
                    if (k == 0) VJF_STAMP(8);
                    trail_tile(k + 1, k + 1, k);
                    if (k == 0) VJF_STAMP(4);
                    good = diag_chain(k + 1);
                    if (k == 0) VJF_STAMP(5);
                }
                if (k < 7) VJF_STAMP(9 + k);
            }
        } else if (wave == 4) {
            if (A.post) {
                for (int k = 0; k < nbl; ++k) {
                    // column k of L and Dinv_k out, write-through, as soon as they are final; their flag once the stores have
                    // drained (nobody waits for this wavefront inside the workgroup)
                    lds_wait(C_CHAIN, k + 1);
                    if (k + 1 < nbl) { lds_wait(C_P1, k + 1); hb_wait(2 * k + 1); }
                    if (!v_ok[0]) break;
                    if (k == 0 && lscr_guard) {                         // (this wavefront alone writes the scratch copies)
                        bool there = false;
                        for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
                            if ((int)(__hip_atomic_load(A.wait_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - it_wait_target) >= 0) { there = true; break; }
                            if ((spins & 255u) == 255u && vjf_abort_seen(SC + VJF_SC_STATUS)) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (!there && lane == 0) { vjf_status_or(SC + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_SIGMA); *s_dead = 1; }
                    }
                    for (int it = 0; it < nbl - k; ++it) put_block(s_blk + (size_t)vtri(k + it, k) * 1024, A.lscr, n, (k + it) * 32, k * 32, it == 0);
                    put_block(s_aux + (size_t)k * 1024, A.dinv_out + (size_t)k * 1024, 32, 0, 0, false);
                    publish(k, k + 1, 0u);
                    kdone = k + 1;
                }
            }
        } else {
            const int hw = wave < 4 ? wave - 1 : wave - 2;              // helper 0 .. 5
            int stage = 0;
            for (int k = 0; k + 1 < nbl; ++k) {
                lds_wait(C_CHAIN, k + 1);
                if (!v_ok[0]) break;
                for (int bi = k + 2 + hw; bi < nbl; bi += 6) panel_tile(bi, k);
                hb_arrive(hw, ++stage);
                hb_wait(stage);
                if (!v_ok[0]) break;
                // trailing tiles of column k: (k+1+r, k+1+c), 0 <= c <= r < m, without (0,0) (the chain's own)
                const int m = nbl - 1 - k;
                if (m >= 2) {
                    if (hw == 0) { lds_wait(C_P1, k + 1); if (v_ok[0]) { trail_tile(k + 2, k + 1, k); lds_post(C_A, k + 1); } }
                    if (hw == 1) { trail_tile(k + 2, k + 2, k); lds_post(C_B, k + 1); }
                    for (int r = 2; r < m; ++r)                         // the rest of block column k + 1
                        if (r % 6 == hw) { lds_wait(C_P1, k + 1); if (v_ok[0]) trail_tile(k + 1 + r, k + 1, k); }
                    int q = m;
                    for (int r = 2; r < m; ++r)
                        for (int c2 = 1; c2 <= r; ++c2, ++q)
                            if (q % 6 == hw) trail_tile(k + 1 + r, k + 1 + c2, k);
                }
                hb_arrive(hw, ++stage);
                hb_wait(stage);
            }
        }
        __syncthreads();
        const bool ok = s_flag[0] != 0;
        VJF_STAMP(2);
        if (!ok) {
            // Reference: the fallback calls the removed torch.eig and raises (module.py:104-112).  Here:
            // undo P += G / v (exact up to one rounding) and leave W, w_chol, w_pchol as they were.
            st |= VJF_STATUS_RLS_FAILED;
            const float inv_v = expf(-sig);
            // (self_prep: the state's P is in the hands of the operand kernel on the post kernel's stream; that kernel's y / W
            //  workgroup, which follows it there, takes the update back)
            if (!sp) for (int e = tid; e < n * n; e += VJF_CHOL_THREADS) Pm[e] = fmaf(-G[e], inv_v, Pm[e]);
            if (A.post) {
                for (int idx = tid; idx < ntri * 256; idx += VJF_CHOL_THREADS) {   // the copy for the next kernel, likewise
                    const int b = idx >> 8, r = (idx >> 3) & 31, c4 = (idx & 7) * 4;
                    const int gi = s_bi[b] * 32 + r, gj = s_bj[b] * 32 + c4;
                    if (gi < n && gj < n) {
                        float4 pv = *reinterpret_cast<const float4*>(A.pscr + (size_t)idx * 4);
                        const float4 gv = g4(gi, gj);          // (another role's stores on the one-launch route: an sc1 load there, found by tools/audit_plain_loads.py)
                        pv.x = fmaf(-gv.x, inv_v, pv.x); pv.y = fmaf(-gv.y, inv_v, pv.y); pv.z = fmaf(-gv.z, inv_v, pv.z); pv.w = fmaf(-gv.w, inv_v, pv.w);
                        *reinterpret_cast<float4*>(A.pscr + (size_t)idx * 4) = pv;
                    }
                }
                // columns published so far are those of iterations that completed; every other flag says "failed"
                if (wave == 4) publish(kdone, VJF_CHOL_MAXBLK + 1, 1u);
                if (tid == 0) { A.ok_out[0] = 0; vjf_status_or(SC + VJF_SC_STATUS, st); }
                return;
            }
        } else {
            if (A.post) {
                // L and the inverted diagonal blocks already left column by column; one-time clearing of the zero halves; done
                if (!A.no_triclean && SC[VJF_SC_TRI_CLEAN] == 0.f) {
                    for (int e = tid; e < n * n; e += VJF_CHOL_THREADS) {
                        const int i = e / n, j = e - i * n;
                        if ((i >> 5) < (j >> 5)) Lm[e] = 0.f;
                        if ((i >> 5) > (j >> 5)) Wc[e] = 0.f;
                    }
                    __syncthreads();
                    if (tid == 0) SC[VJF_SC_TRI_CLEAN] = 1.f;
                }
                if (tid == 0) A.ok_out[0] = 1;
                if (wave == 4) publish(VJF_CHOL_MAXBLK, VJF_CHOL_MAXBLK + 1, 0u);   // the factor as a whole is good
                return;
            }
            // ---- w_pchol = L (lower, module.py:99-100)
            {
                float4 v[VJF_CHOL_Q];
#pragma unroll
                for (int q = 0; q < VJF_CHOL_Q; ++q) {
                    const int idx = tid + q * VJF_CHOL_THREADS;
                    if (idx < ntri * 256) {
                        const int b = idx >> 8, r = (idx >> 3) & 31, c4 = (idx & 7) * 4;
                        const float* blk = s_blk + (size_t)b * 1024;
                        const bool dg = s_bi[b] == s_bj[b];
                        v[q].x = (!dg || c4 <= r) ? blk[vsw(r, c4)] : 0.f;
                        v[q].y = (!dg || c4 + 1 <= r) ? blk[vsw(r, c4 + 1)] : 0.f;
                        v[q].z = (!dg || c4 + 2 <= r) ? blk[vsw(r, c4 + 2)] : 0.f;
                        v[q].w = (!dg || c4 + 3 <= r) ? blk[vsw(r, c4 + 3)] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < VJF_CHOL_Q; ++q) {
                    const int idx = tid + q * VJF_CHOL_THREADS;
                    if (idx < ntri * 256) {
                        const int b = idx >> 8, r = (idx >> 3) & 31, c4 = (idx & 7) * 4;
                        const int gi = s_bi[b] * 32 + r, gj = s_bj[b] * 32 + c4;
                        if (gi < n && gj < n) *reinterpret_cast<float4*>(Lm + (size_t)gi * n + gj) = v[q];
                    }
                }
            }
            VJF_STAMP(3);
            VJF_STAMP(4);
            // ---- X = L^-1, block row by block row, in place over L:
            //      X_ij = -Dinv_i * sum_{k=j}^{i-1} L_ik X_kj   (X_jj = Dinv_j)
            for (int bi = 1; bi < nbl; ++bi) {
                vjf_f32x16 xacc;
                const int bj = wave;                                   // bi <= 6 < 8 wavefronts
                if (bj < bi) {
                    vjf_f32x16 t;
#pragma unroll
                    for (int r = 0; r < 16; ++r) t[r] = 0.f;
                    for (int k = bj; k < bi; ++k) {
                        const float* xb = (k == bj) ? s_aux + (size_t)bj * 1024 : s_blk + (size_t)vtri(k, bj) * 1024;
                        blk_mma<false>(t, s_blk + (size_t)vtri(bi, k) * 1024, xb, 1.f, lane);
                    }
                    // X_ij = -Dinv_i * T with T taken straight from the accumulator: register r of T holds
                    // rows vrow(r,0) / vrow(r,1) on the two lane halves = the k pair of MFMA step r.
                    const float* di = s_aux + (size_t)bi * 1024;
                    const int c = lane & 31, h = lane >> 5;
#pragma unroll
                    for (int r = 0; r < 16; ++r) xacc[r] = 0.f;
                    float da[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) da[r] = -di[vsw(c, vrow(r, h))];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) xacc = __builtin_amdgcn_mfma_f32_32x32x2f32(da[r], t[r], xacc, 0, 0, 0);
                }
                __syncthreads();                                       // every reader of row bi of L is done
                if (bj < bi) blk_store(xacc, s_blk + (size_t)vtri(bi, bj) * 1024, lane);
                __syncthreads();
            }
            VJF_STAMP(5);
            // diagonal blocks of X
            for (int e = tid; e < nbl * 1024; e += VJF_CHOL_THREADS) s_blk[(size_t)vtri(e >> 10, e >> 10) * 1024 + (e & 1023)] = s_aux[e];
            __syncthreads();
            // ---- w_chol = X^T (upper, module.py:102): row (bj*32+c) of w_chol, 4 consecutive r per store
            {
                float4 v[VJF_CHOL_Q];
#pragma unroll
                for (int q = 0; q < VJF_CHOL_Q; ++q) {
                    const int idx = tid + q * VJF_CHOL_THREADS;
                    if (idx < ntri * 256) {
                        const int b = idx >> 8, c = (idx >> 3) & 31, r4 = (idx & 7) * 4;
                        const float* blk = s_blk + (size_t)b * 1024;
                        const bool dg = s_bi[b] == s_bj[b];
                        v[q].x = (!dg || c <= r4) ? blk[vsw(r4, c)] : 0.f;
                        v[q].y = (!dg || c <= r4 + 1) ? blk[vsw(r4 + 1, c)] : 0.f;
                        v[q].z = (!dg || c <= r4 + 2) ? blk[vsw(r4 + 2, c)] : 0.f;
                        v[q].w = (!dg || c <= r4 + 3) ? blk[vsw(r4 + 3, c)] : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < VJF_CHOL_Q; ++q) {
                    const int idx = tid + q * VJF_CHOL_THREADS;
                    if (idx < ntri * 256) {
                        const int b = idx >> 8, c = (idx >> 3) & 31, r4 = (idx & 7) * 4;
                        const int gi = s_bi[b] * 32 + r4, gj = s_bj[b] * 32 + c;          // X[gi..gi+3][gj]
                        if (gi < n && gj < n) *reinterpret_cast<float4*>(Wc + (size_t)gj * n + gi) = v[q];
                    }
                }
            }
            // The off-diagonal blocks on the other side of the diagonal (upper for w_pchol, lower for
            // w_chol) are zero and stay zero; they are cleared once per state blob, not every step.
            if (SC[VJF_SC_TRI_CLEAN] == 0.f) {
                for (int e = tid; e < n * n; e += VJF_CHOL_THREADS) {
                    const int i = e / n, j = e - i * n;
                    if ((i >> 5) < (j >> 5)) Lm[e] = 0.f;
                    if ((i >> 5) > (j >> 5)) Wc[e] = 0.f;
                }
                __syncthreads();
                if (tid == 0) SC[VJF_SC_TRI_CLEAN] = 1.f;
            }
            VJF_STAMP(6);
            // ---- y = X g ; W = X^T y  (cholesky_solve, module.py:101) as block products on the matrix cores.
            //      Block row br of y sums br+1 products; wavefront w takes the even-offset products of row w and the
            //      odd-offset products of row nbl-1-w (<= ceil((nbl+1)/2) products each), partial tiles meet in LDS.
            {
                float* s_pe = s_aux;                          // even partials, npad x DZP   (s_aux = Dinv is free now)
                float* s_po = s_aux + (size_t)npad * DZP;     // odd partials
                const int c = lane & 31, h = lane >> 5;
                auto store_part = [&](const vjf_f32x16& acc, float* dst, int blk) {
                    if (c < DZP) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) dst[(blk * 32 + vrow(r, h)) * DZP + c] = acc[r];
                    }
                };
                for (int pass = 0; pass < 2; ++pass) {                      // pass 0: y = X g ; pass 1: W = X^T y
                    const float* src = s_g;                                 // g, then y
                    for (int par = 0; par < 2; ++par) {
                        const int row = par == 0 ? wave : nbl - 1 - wave;   // block row of the output
                        if (wave < nbl && row >= 0 && row < nbl) {
                            vjf_f32x16 acc;
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                            // pass 0: column blocks cb = row - par, row - par - 2, ... >= 0 of X(row, cb)
                            // pass 1: row blocks    rb = row + par, row + par + 2, ... < nbl of X(rb, row)^T
                            for (int o = par; pass == 0 ? (row - o >= 0) : (row + o < nbl); o += 2) {
                                const int ob = pass == 0 ? row - o : row + o;
                                const float* xb = s_blk + (size_t)(pass == 0 ? vtri(row, ob) : vtri(ob, row)) * 1024;
                                const float* sb = src + (size_t)ob * 32 * DZP;
                                if (pass == 0)
                                    blk_mma_f(acc, lane, [&](int i, int m) { return xb[vsw(i, m)]; },
                                              [&](int m, int j) { return j < DZP ? sb[m * DZP + j] : 0.f; });
                                else
                                    blk_mma_f(acc, lane, [&](int i, int m) { return xb[vsw(m, i)]; },
                                              [&](int m, int j) { return j < DZP ? sb[m * DZP + j] : 0.f; });
                            }
                            store_part(acc, par == 0 ? s_pe : s_po, row);
                        } else if (par == 1 && wave < nbl) {
                            // (no odd partner row)
                        }
                    }
                    // rows whose odd partial was not produced (nbl-1-w out of range never happens for w < nbl) are all covered
                    __syncthreads();
                    for (int e = tid; e < npad * DZP; e += VJF_CHOL_THREADS) {
                        const float v = s_pe[e] + s_po[e];
                        s_g[e] = v;                                         // y after pass 0, W after pass 1
                        if (pass == 1) {
                            const int rw = e / DZP, j = e - rw * DZP;
                            if (rw < n && j < dz) Wm[(size_t)rw * dz + j] = v;
                        }
                    }
                    __syncthreads();
                }
            }
        }
    }
    if (warm || st) {                                                  // residual needs W in LDS
        for (int e = tid; e < npad * DZP; e += VJF_CHOL_THREADS) {
            const int r = e / DZP, j = e - r * DZP;
            s_g[e] = (r < n && j < dz) ? Wm[(size_t)r * dz + j] : 0.f;
        }
        __syncthreads();
    }
    VJF_STAMP(7);
    // ---- residual mean square:  sum|dx|^2 - 2 tr(W^T FDX) + tr(W^T G W), fp64 accumulation (model.py:373-374).
    //      thread (i, part) forms half of row i of G W:  G row chunks as float4 (8 in flight), W rows broadcast
    double part_sum = 0.0;
    {
        const int i = tid & 255, part = tid >> 8;
        if (i < n) {
            float acc[DZP];
#pragma unroll
            for (int j = 0; j < DZP; ++j) acc[j] = 0.f;
            const int half = (n / 4 + 1) / 2 * 4;                      // columns [0,half) and [half,n), multiples of 4
            const int c0 = part ? half : 0, c1 = part ? n : half;
            const float* grow = G + (size_t)i * n;
            for (int cb = c0; cb < c1; cb += 32) {
                float4 gv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) gv[q] = (cb + 4 * q < c1) ? *reinterpret_cast<const float4*>(grow + cb + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = cb + 4 * q;
                    if (c < c1) {
                        axpy_row<DZP>(acc, gv[q].x, s_g + c * DZP);
                        axpy_row<DZP>(acc, gv[q].y, s_g + (c + 1) * DZP);
                        axpy_row<DZP>(acc, gv[q].z, s_g + (c + 2) * DZP);
                        axpy_row<DZP>(acc, gv[q].w, s_g + (c + 3) * DZP);
                    }
                }
            }
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < DZP; ++j) q = fmaf(s_g[i * DZP + j], acc[j], q);
            part_sum = (double)q;
            if (part == 0) {
                float f = 0.f;
                for (int j = 0; j < dz; ++j) f = fmaf(s_g[i * DZP + j], FDX[(size_t)i * dz + j], f);
                part_sum -= 2.0 * (double)f;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) part_sum += __shfl_xor(part_sum, o, 64);
    if (lane == 0) s_d[wave] = part_sum;
    __syncthreads();
    if (tid == 0) {
        double t = (double)it_red[P.red_SC + RS_SDX2];
        for (int w = 0; w < VJF_CHOL_THREADS / 64; ++w) t += s_d[w];
        if (t < 0.0) t = 0.0;
        const float mse = (float)(t / ((double)Bf * (double)dz));
        const float acc = fminf(SC[VJF_SC_N_TR], 500.f), tot = acc + Bf;   // running_var, size_cap=500 (model.py:375)
        S[P.off[VJF_SLOT_TR_LOGVAR]] = logf((acc / tot) * expf(sig) + (Bf / tot) * mse);
        SC[VJF_SC_N_TR] = tot;
        if (st) vjf_status_or(SC + VJF_SC_STATUS, st);
    }
    VJF_STAMP(8);
}

// One-time clearing of the halves that the post kernel never writes (block-lower part of w_chol, block-upper part of
// w_pchol), for callers that run the Cholesky kernel beside a reader of w_chol (vjf_filter_seq).  grid-stride.
// One pass (nsteps <= 0) or the looping form: nsteps factorisations, one after the other (see VjfCholArgs::nsteps).  The
// per-step values travel as scalars beside the kernel arguments, which stay in scalar registers.
template <int DZP>
__device__ __forceinline__ void vjf_chol_loop(const VjfPlan& P, const VjfCholArgs& A, float* lds, int* s_dead) {
    const int steps = A.nsteps > 0 ? A.nsteps : 1;
    for (int it = 0; it < steps; ++it) {
        const float* red = ((A.step0 + it) & 1) ? A.red2 : A.red;
        vjf_chol_body<DZP>(P, A, lds, s_dead, A.epoch + (unsigned)it, red, A.wait_target + (unsigned)it * A.wait_stride,
                           A.stat_target + (unsigned)it * A.stat_stride, it == 0 && A.src_state != 0);
        __syncthreads();
        if (*s_dead) break;
    }
}

template <int DZP>
__global__ __launch_bounds__(VJF_CHOL_THREADS) void vjf_chol_lds_kernel(VjfPlan P, VjfCholArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_dead;                             // a wait timed out: the looping form stops (status says so)
    if (threadIdx.x == 0) s_dead = 0;
    __syncthreads();
    vjf_chol_loop<DZP>(P, A, lds, &s_dead);
}

__global__ void vjf_triclean_kernel(VjfPlan P, float* state) {
    float* SC = state + P.off[VJF_SLOT_SCALARS];
    if (SC[VJF_SC_TRI_CLEAN] != 0.f) return;
    float* Wc = state + P.off[VJF_SLOT_W_CHOL];
    float* Lm = state + P.off[VJF_SLOT_W_PCHOL];
    const int n = P.n;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x) {
        const int i = e / n, j = e - i * n;
        if ((i >> 5) < (j >> 5)) Lm[e] = 0.f;
        if ((i >> 5) > (j >> 5)) Wc[e] = 0.f;
    }
}
// (the flag is set by a second, one-thread launch behind it: every workgroup above must have seen it clear)
__global__ void vjf_triclean_done_kernel(VjfPlan P, float* state) { state[P.off[VJF_SLOT_SCALARS] + VJF_SC_TRI_CLEAN] = 1.f; }
