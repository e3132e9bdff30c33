// vjf_rlsb_kernels.h -- the RLS update (module.py:79-112) for feature counts that do not fit one compute unit's LDS
// (n_rbf > 224; BASELINE config E has 1000): a blocked Cholesky, inverse and solves on the matrix in global memory
// (L2-resident: 4 MB at n = 1000), one launch per block column, the tiles of a column spread over the chip.
//
//   GEMM + vjf_rlsb_prep_kernel   g = P W + Phi^T dx / v (module.py:94);  A := P + Phi^T Phi / v into a work buffer (module.py:96)
//   nbl + 1 launches of vjf_rlsc_col_kernel: block column k of L (module.py:99) and block row k - 1 of X = L^-1 (module.py:102)
//   two GEMMs (vjf_skinny_gemm_kernel)      y = X g,  W = X^T y                      (module.py:101)
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
    float* Ld;           // nbl blocks (32x32 row-major): the diagonal blocks of L (the diagonal tiles of Lw keep A_kk: every workgroup
                         // of launch k reads that tile while one of them has the factor ready)
    float* Pacc;         // 2 x nbl tiles (accumulator order): the sums the ahead role leaves for column k (set k & 1)
    int* ok;             // [0]: 1 while every pivot so far was positive
    int k;               // block column (factorisation) or block row (inverse) of this launch
    int absent_wg;       // test hook (VJF_DEBUG_RLSC_ABSENT = w + 1): workgroup w of the resident column loop leaves at once
};

// Loads and stores of the column sequence.  WT = false: plain (one launch per step: the kernel boundary makes them visible);
// WT = true (the resident form, vjf_rlsc_loop_kernel): write-through stores and sc1 loads -- what another workgroup stored and
// drained before it counted itself in at the step barrier is read from memory, past this CU's L1 and this XCD's L2, with no
// cache writeback or invalidate at the barrier (MI355X guide, "sc1 loads in place of the acquire").
template <bool WT> __device__ __forceinline__ float rlsc_ld(const float* p) {
    if (WT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool WT> __device__ __forceinline__ void rlsc_st(float* p, float v) {
    if (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
typedef unsigned rlsc_u4 __attribute__((ext_vector_type(4)));
template <bool WT> __device__ __forceinline__ float4 rlsc_ld4(const float* M, size_t off) {      // M: the same for the whole wavefront
    if (WT) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(M), 0, 0x7fffffff, 0x00020000);
        const rlsc_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(off * 4), 0, 16);       // aux 16 = sc1
        return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
    }
    return *reinterpret_cast<const float4*>(M + off);
}

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
template <bool WT = false>
__device__ __forceinline__ void rlsb_acc_out(const vjf_f32x16& acc, float* M, int n, int bi, int bj, int lane, bool lower_only) {
    const int c = lane & 31, h = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = vrow(r, h), gi = bi * 32 + row, gj = bj * 32 + c;
        if (gi < n && gj < n) rlsc_st<WT>(M + (size_t)gi * n + gj, (!lower_only || c <= row) ? acc[r] : 0.f);
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
    if (gid == 0) { A.ok[0] = 1; A.ok[4] = 0; }             // (ok[4]: the step counter of vjf_rlsc_loop_kernel)
    for (int e = gid; e < n * dz; e += gsz) A.gbuf[e] = A.gbuf[e] + FDX[e] * inv_v;   // gbuf holds P W (the GEMM before this kernel)
    for (int e = gid; e < n * n; e += gsz) Lm[e] = Pm[e] + G[e] * inv_v;
}

// One block column of the factorisation and one block row of the inverse per launch (left-looking, so that a column needs ONE
// launch instead of diagonal / panel / trailing-update launches, and the inverse rides along instead of following as a phase
// of its own).  Launch k, workgroups of eight wavefronts, three roles:
//   column role, block rows i = k .. nbl-1:  D = A_kk - sum_{j<k} L_kj L_kj^T  and  T = A_ik - sum_{j<k} L_ij L_kj^T, of which
//       the terms j <= k-2 were summed by launch k-1 (ahead role) and only j = k-1 -- the column the previous launch produced --
//       is formed here (one wavefront each); L_kk = chol(D) with its inverse (EVERY workgroup: the same bits, no hand-off);
//       i = k: L_kk, L_kk^-1 -> Ld, Dinv, X_kk;   i > k: L_ik = T L_kk^-T -> Lw.
//   ahead role, block rows i = k+1 .. nbl-1:  Pacc_{k+1}[i] = sum_{j<=k-1} L_ij L_{k+1,j}^T  for the NEXT column (columns <= k-1
//       are complete when launch k starts): the j range split over the wavefronts, operands straight from global memory into
//       the MFMA operand layout, partial tiles summed through LDS in a fixed order.  The bulk of the multiply-adds of a column
//       thus runs beside the previous column's serial part (loads, 32x32 factorisation, solve) on other compute units.
//   inverse role, block row r = k-1, tiles j = 0 .. r-1:  X_rj = -L_rr^-1 sum_{m=j}^{r-1} L_rm X_mj   (rows < r of X, row r of L
//       and L_rr^-1 are all complete when launch k starts).
// Launch nbl has the inverse role only (the last block row).
// Operand layout: lane (row, half) of v_mfma_f32_32x32x2_f32 takes k = 8 q + 4 half + e at step (q, e) -- any order of the k
// indices is a valid product as long as both operands use it -- so a k-contiguous operand row is four 16-byte loads.
template <bool WT>
__device__ __forceinline__ void rlsc_ld_kc(float (&v)[16], const float* M, int n, int brow, int bcol, int lane, bool vec) {
    const int row = brow * 32 + (lane & 31), c0 = bcol * 32 + 4 * (lane >> 5);
    if (row < n) {
        const size_t o = (size_t)row * n + c0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (vec) { const float4 x = rlsc_ld4<WT>(M, o + 8 * q); v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 * q + e] = rlsc_ld<WT>(M + o + 8 * q + e);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = 0.f;
    }
}
// B operand from a row-major (k, column) tile: lane (column, half), same k order
template <bool WT>
__device__ __forceinline__ void rlsc_ld_rc(float (&v)[16], const float* M, int n, int brow, int bcol, int lane) {
    const float* p = M + (size_t)(brow * 32 + 4 * (lane >> 5)) * n + bcol * 32 + (lane & 31);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) v[4 * q + e] = rlsc_ld<WT>(p + (size_t)(8 * q + e) * n);
}

#define VJF_RLSC_THREADS 512
#ifndef VJF_RLSC_BARRIER_SLEEP
#define VJF_RLSC_BARRIER_SLEEP 40     /* s_sleep argument between two polls of the step barrier's counter (vjf_rlsc_loop_kernel): 63 pollers beside the trial chain -- with 2 config E ran 763 us a step, with 24 ... 100: 722-739 */
#endif
struct VjfRlscLds {
    float p[4][2][1024];
    float d[1024], i[1024], t[1024];
    int good;
};
// step k of the sequence for workgroup `bid` (see above); every path ends in a return, no thread of the workgroup is left behind
// at a barrier
template <bool WT>
__device__ __forceinline__ void rlsc_col_step(const VjfPlan& P, const VjfRlsbArgs& A, const int k, const int bid, VjfRlscLds& L) {
    float (&s_p)[4][2][1024] = L.p;
    float (&s_d)[1024] = L.d;
    float (&s_i)[1024] = L.i;
    float (&s_t)[1024] = L.t;
    int& s_good = L.good;
    const int n = P.n, nbl = (n + 31) / 32, ncol = nbl - k, nahead = ncol > 1 ? ncol - 1 : 0;
    // (In the resident form the compiler hoists this function's lane-dependent address arithmetic out of the step loop and keeps
    //  it: 256 VGPRs + 208 B per lane of scratch instead of 160 VGPRs and none with the thread index made opaque per step -- and
    //  is the FASTER of the two, 767-771 against 780-787 us per config E step on one box: the hoisted form stays.)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool vec = (n & 3) == 0;
    float* Lm = A.Lw;
    vjf_f32x16 acc0, acc1;
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
    // the eight wavefronts' partial tiles -> s_p[0..3] (wavefront w + 4 onto w, through LDS), fixed order
    auto fold = [&](bool second) {
        if (wave >= 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { s_p[wave - 4][0][q * 64 + lane] = acc0[q]; if (second) s_p[wave - 4][1][q * 64 + lane] = acc1[q]; }
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                s_p[wave][0][q * 64 + lane] += acc0[q];
                if (second) s_p[wave][1][q * 64 + lane] += acc1[q];
            }
        }
        __syncthreads();
    };
    if (bid < ncol) {
        const int i = k + bid;
        const bool below = i != k;
        // the tiles the sums are taken from and the sums launch k - 1 left (this thread's two elements of each), and the operands
        // of the one term formed here, all requested before anything waits
        float dkk[2], aik[2], pkk[2] = {0.f, 0.f}, pik[2] = {0.f, 0.f};
        const float* pc = A.Pacc + (size_t)(k & 1) * nbl * 1024;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 512 * q, row = vrow(e >> 6, (e >> 5) & 1), col = e & 31;
            const int gk = k * 32 + row, gi = i * 32 + row, gj = k * 32 + col;
            dkk[q] = (gk < n && gj < n) ? rlsc_ld<WT>(Lm + (size_t)gk * n + gj) : (gk == gj ? 1.f : 0.f);
            aik[q] = (below && gi < n && gj < n) ? rlsc_ld<WT>(Lm + (size_t)gi * n + gj) : 0.f;
            if (k > 0) { pkk[q] = rlsc_ld<WT>(pc + (size_t)k * 1024 + e); if (below) pik[q] = rlsc_ld<WT>(pc + (size_t)i * 1024 + e); }
        }
        if (k > 0 && wave < 2) {                            // wavefront 0: L_k,k-1 L_k,k-1^T; wavefront 1: L_i,k-1 L_k,k-1^T
            float lk[16], li[16];
            rlsc_ld_kc<WT>(lk, Lm, n, k, k - 1, lane, vec);
            if (wave == 1 && below) rlsc_ld_kc<WT>(li, Lm, n, i, k - 1, lane, vec);
            if (wave == 0) {
#pragma unroll
                for (int t = 0; t < 16; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(lk[t], lk[t], acc0, 0, 0, 0);
            } else if (below) {
#pragma unroll
                for (int t = 0; t < 16; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(li[t], lk[t], acc0, 0, 0, 0);
            }
        }
        if (wave < 2) {
#pragma unroll
            for (int q = 0; q < 16; ++q) s_p[wave][0][q * 64 + lane] = acc0[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 512 * q, row = vrow(e >> 6, (e >> 5) & 1), col = e & 31;
            s_d[vsw(row, col)] = dkk[q] - (pkk[q] + s_p[0][0][e]);
            if (below) s_t[vsw(row, col)] = aik[q] - (pik[q] + s_p[1][0][e]);
        }
        __syncthreads();
        if (wave == 0) {
            const bool good = potrf_inv_chain2(s_d, s_i, lane, min(32, n - 32 * k));
            if (lane == 0) s_good = good ? 1 : 0;
        }
        __syncthreads();
        if (!s_good) { if (!below && tid == 0) __hip_atomic_store(A.ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
        if (!below) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + 512 * q, r = e >> 5, c = e & 31;
                const int gi = k * 32 + r, gj = k * 32 + c;
                const float x = s_i[vsw(r, c)];
                if (gi < n && gj < n) rlsc_st<WT>(A.X + (size_t)gi * n + gj, x);
                rlsc_st<WT>(A.Dinv + (size_t)k * 1024 + e, x);
                rlsc_st<WT>(A.Ld + (size_t)k * 1024 + e, c <= r ? s_d[vsw(r, c)] : 0.f);
            }
        } else if (wave == 0) {
            vjf_f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            blk_mma<true>(acc, s_t, s_i, 1.f, lane);               // T L_kk^-T
            rlsb_acc_out<WT>(acc, Lm, n, i, k, lane, false);
        }
        return;
    }
    if (bid < ncol + nahead) {
        // ahead role: the sums over the finished columns j <= k - 1 for column k + 1, rows i >= k + 1
        const int c1 = k + 1, i = c1 + (bid - ncol);
        // block columns j = wave, wave + 8, ..: the tiles come from other compute units' launches (memory-side latency), so the
        // operands of FOUR block columns are requested before the first MFMA
        for (int jb = wave; jb < k; jb += 32) {
            float lk[4][16], li[4][16];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (jb + 8 * u < k) {
                    rlsc_ld_kc<WT>(lk[u], Lm, n, c1, jb + 8 * u, lane, vec);
                    rlsc_ld_kc<WT>(li[u], Lm, n, i, jb + 8 * u, lane, vec);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (jb + 8 * u < k) {
#pragma unroll
                    for (int t = 0; t < 16; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(li[u][t], lk[u][t], acc0, 0, 0, 0);
                }
        }
        fold(false);
        float* pn = A.Pacc + ((size_t)(c1 & 1) * nbl + i) * 1024;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 512 * q;
            rlsc_st<WT>(pn + e, ((s_p[0][0][e] + s_p[1][0][e]) + s_p[2][0][e]) + s_p[3][0][e]);
        }
        return;
    }
    // inverse role
    const int r = k - 1, j = bid - ncol - nahead;
    if (r < 1 || j >= r) return;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + 512 * q;
        s_i[vsw(e >> 5, e & 31)] = rlsc_ld<WT>(A.Dinv + (size_t)r * 1024 + e);
    }
    for (int mb = j + wave; mb < r; mb += 32) {
        float a[4][16], b[4][16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (mb + 8 * u < r) {
                rlsc_ld_kc<WT>(a[u], Lm, n, r, mb + 8 * u, lane, vec);
                rlsc_ld_rc<WT>(b[u], A.X, n, mb + 8 * u, j, lane);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (mb + 8 * u < r) {
#pragma unroll
                for (int t = 0; t < 16; ++t) acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][t], b[u][t], acc0, 0, 0, 0);
            }
    }
    fold(false);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = tid + 512 * q, row = vrow(e >> 6, (e >> 5) & 1), col = e & 31;
        s_t[vsw(row, col)] = ((s_p[0][0][e] + s_p[1][0][e]) + s_p[2][0][e]) + s_p[3][0][e];
    }
    __syncthreads();
    if (wave == 0) {
        vjf_f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        blk_mma<false>(acc, s_i, s_t, -1.f, lane);                 // -L_rr^-1 S
        rlsb_acc_out<WT>(acc, A.X, n, r, j, lane, false);
    }
}

__global__ __launch_bounds__(VJF_RLSC_THREADS) void vjf_rlsc_col_kernel(VjfPlan P, VjfRlsbArgs A) {
    __shared__ __attribute__((aligned(16))) VjfRlscLds L;
    if (A.ok[0] == 0) return;
    rlsc_col_step<false>(P, A, A.k, (int)blockIdx.x, L);
}

// The whole sequence k = 0 .. nbl as ONE launch of 2 nbl - 1 workgroups (the widest step's count) that stay resident: between two
// steps every workgroup drains its write-through stores, counts itself in at a word in memory and waits for all the others; what
// it reads of the others' results it reads with sc1 loads (see rlsc_ld).  [Fences instead -- a release and an acquire at agent
// scope per step, i.e. a writeback and an invalidate of the XCD's L2 by every wavefront -- cost 12 us a step: measured, dropped.]  A sequence of short launches of few workgroups, each waiting
// for the one before it, is at the mercy of whatever else the chip runs: beside the trial chain's chip-filling kernels on another
// stream (filter_seq_two) every one of them waited for compute units to come free (config E: the update took 770 us beside the
// trial chain, 330 us alone).  Resident workgroups are placed once.  The grid fits the chip many times over (63 workgroups at
// n = 1000) and nothing that is launched before it waits for it, so the workgroups that are placed later than others are placed.
// A failed pivot: every column workgroup finds it (each factors the block itself), one of them clears A.ok before the step's
// barrier, and all workgroups leave behind that barrier.  bar: a word the launch before this one (vjf_rlsb_prep_kernel) zeroed.
__global__ __launch_bounds__(VJF_RLSC_THREADS) void vjf_rlsc_loop_kernel(VjfPlan P, VjfRlsbArgs A, unsigned* bar) {
    __shared__ __attribute__((aligned(16))) VjfRlscLds L;
    __shared__ int s_go;
    if (A.ok[0] == 0) return;
    if (A.absent_wg > 0 && (int)blockIdx.x == A.absent_wg - 1) return;   // (test hook, VJF_DEBUG_RLSC_ABSENT: this workgroup never arrives)
    const int nbl = (P.n + 31) / 32, tid = threadIdx.x;
    for (int k = 0; k <= nbl; ++k) {
        rlsc_col_step<true>(P, A, k, (int)blockIdx.x, L);
        if (k == nbl) break;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (every thread: its write-through stores are in memory)
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(k + 1) * gridDim.x;
            bool there = false;
            for (unsigned spins = 0; spins < VJF_WAIT_SPINS; ++spins) {
                if ((int)(__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { there = true; break; }
                if ((spins & 255u) == 255u && vjf_abort_seen(A.state + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS)) break;
                __builtin_amdgcn_s_sleep(VJF_RLSC_BARRIER_SLEEP);
            }
            s_go = there ? 1 : 0;
        }
        __syncthreads();
        // (a wait that ran out: a workgroup of this launch was never placed -- the update is dropped like one with a failed pivot, and,
        //  unlike a failed pivot, the status word carries a wait bit: check_status() raises, the sequence does not go on unnoticed)
        if (!s_go) {
            if (tid == 0) {
                __hip_atomic_store(A.ok, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vjf_status_or(A.state + P.off[VJF_SLOT_SCALARS] + VJF_SC_STATUS, VJF_STATUS_RLS_FAILED | VJF_STATUS_WAIT_COLUMN);
            }
            return;
        }
        if (__hip_atomic_load(A.ok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
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
        const int bi = i >> 5, bj = j >> 5;                        // w_pchol = L (module.py:99-100)
        Lm[e] = bj > bi ? 0.f : bj == bi ? A.Ld[(size_t)bi * 1024 + (i & 31) * 32 + (j & 31)] : A.Lw[e];
        Pm[e] += G[e] * inv_v;
    }
}
