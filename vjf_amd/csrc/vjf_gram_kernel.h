// vjf_gram_kernel.h -- K1b: every sum over trials that the step needs, as split-K Gram tiles on
// the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, k-ordered fma chain).
//
//   kind 0:  E^T E       -> G = Phi^T Phi (module.py:96) and Phi^T dx (module.py:94)
//   kind 1:  DEL^T ACT   -> the weight/bias gradients that autograd's backward produces for the
//                           recognition layers, heads and decoder (model.py:209)
// A job is one 32x32 output tile; blockIdx.y splits the trial axis.  The 4 wavefronts of a
// workgroup take interleaved 2-trial slices of the split, are summed through LDS in wave order
// and written as one slab; vjf_gram_reduce_kernel adds the slabs in split order (deterministic)
// and scatters to the reduce buffer.
#pragma once
#include <hip/hip_runtime.h>
#include "vjf_plan.h"

typedef float vjf_f32x16 __attribute__((ext_vector_type(16)));
typedef float vjf_f32x4 __attribute__((ext_vector_type(4)));

struct VjfGramArgs {
    const VjfJob* jobs;
    const float* E; const float* ACT; const float* DEL;
    float* slabs;            // (njobs, nsplit, 1024)
    int B, nsplit, rows_per_split;
    int job0;                // first job of this launch (the grid covers a contiguous job range)
    int high_prio;           // raise the wavefronts' issue priority (the statistics Gram that runs beside the trial kernel)
    const unsigned* run_if;  // non-null: nothing is done when the word is 0 (the replay after a non-finite loss component)
};

#define VJF_GRAM_WAVES 4            // wavefronts per workgroup (8 was tried: 1.4 us/step slower at config B)
#define VJF_GRAM_THREADS (64 * VJF_GRAM_WAVES)
__global__ __launch_bounds__(VJF_GRAM_THREADS) void vjf_gram_kernel(VjfPlan P, VjfGramArgs A) {
    constexpr int GW = VJF_GRAM_WAVES;
    __shared__ float s_acc[(GW - 1) * 1024];
    if (A.run_if && __hip_atomic_load(A.run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    if (A.high_prio) __builtin_amdgcn_s_setprio(3);     // beside the trial kernel's older wavefronts: do not starve
    // linear id = job * nsplit + split: workgroups are dealt round-robin over the 8 XCDs, so with nsplit a
    // multiple of 8 every job of one trial range lands on the same XCD and re-reads its rows from that L2
    const int split = blockIdx.x % A.nsplit, jobid = A.job0 + blockIdx.x / A.nsplit;
    const VjfJob job = A.jobs[jobid];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, kh = lane >> 5;
    const float* X; const float* Y; int ldx, ldy;
    if (job.kind == 0) { X = A.E; Y = A.E; ldx = ldy = P.ldE; }
    else { X = A.DEL; Y = A.ACT; ldx = P.ldD; ldy = P.ldA; }
    const bool xok = r < job.xn, yok = r < job.yn;
    const float* xp = X + job.xc + (xok ? r : 0);
    const float* yp = Y + job.yc + (yok ? r : 0);
    const int k_begin = split * A.rows_per_split;
    const int k_end = min(A.B, k_begin + A.rows_per_split);
    vjf_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // wave w takes trials k_begin + 2*GW*j + 2*w + {0,1}; 8 steps' operands (16 loads) are in flight before the first MFMA
    int k = k_begin + 2 * wave + kh;
    for (; k + 14 * GW < k_end + kh; k += 16 * GW) {
        float a[8], b[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kq = k + 2 * GW * q;
            const bool kok = kq < k_end;
            a[q] = (kok && xok) ? xp[(size_t)kq * ldx] : 0.f;
            b[q] = (kok && yok) ? yp[(size_t)kq * ldy] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
    }
    for (; k < k_end + kh; k += 2 * GW) {
        const bool kok = k < k_end;
        const float a = (kok && xok) ? xp[(size_t)k * ldx] : 0.f;
        const float b = (kok && yok) ? yp[(size_t)k * ldy] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    // D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) s_acc[(wave - 1) * 1024 + i * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
        float* slab = A.slabs + ((size_t)jobid * A.nsplit + split) * 1024;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float v = acc[i];
#pragma unroll
            for (int w = 0; w < GW - 1; ++w) v += s_acc[w * 1024 + i * 64 + lane];   // wave order
            const int row = (i & 3) + 8 * (i >> 2) + 4 * kh;
            slab[row * 32 + r] = v;
        }
    }
}

struct VjfReduceArgs {
    const VjfJob* jobs;
    const float* slabs;
    const float* partial;     // (nblocks_k1, RS_N) per-workgroup loss sums from K1
    float* red;               // reduce buffer
    int njobs, nsplit, nblocks_k1;
    int job0;                 // first job of this launch; njobs = jobs in this launch
    unsigned sc_mask;         // which of K1's loss sums the extra workgroup reduces (bit per RS_* index)
    const unsigned* run_if;   // non-null: nothing is done when the word is 0
    unsigned* done_count;     // non-null: the sums leave as write-through stores and every workgroup counts itself in here once its own
                              // are in memory -- a kernel on another stream waits for the count in-kernel (three-stream route)
};

// grid = njobs + 1 workgroups of 1024 threads (one tile element each: all of a thread's slab loads are in flight at once);
// the extra workgroup sums K1's loss partials (sc_mask == 0: grid = njobs).
#define VJF_REDUCE_THREADS 1024
__global__ __launch_bounds__(VJF_REDUCE_THREADS) void vjf_gram_reduce_kernel(VjfPlan P, VjfReduceArgs A) {
    const int tid = threadIdx.x;
    if (A.run_if && __hip_atomic_load(A.run_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) return;
    const bool wt = A.done_count != nullptr;
    auto put = [&](float* p, float v) { if (wt) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v; };
    auto arrive = [&]() {
        if (!wt) return;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(A.done_count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    if ((int)blockIdx.x == A.njobs) {
        // RS_N scalars; 32 threads per scalar accumulate strided partials in double, then a fixed xor tree over the 32
        const int sc = (tid >> 5) & 7, l = tid & 31;
        double v = 0.0;
        if (tid < 256) {
            int b = l;
            for (; b + 7 * 32 < A.nblocks_k1; b += 8 * 32) {       // 8 loads in flight, summed in the same order
                float t[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) t[q] = A.partial[(size_t)(b + 32 * q) * RS_N + sc];
#pragma unroll
                for (int q = 0; q < 8; ++q) v += (double)t[q];
            }
            for (; b < A.nblocks_k1; b += 32) v += (double)A.partial[(size_t)b * RS_N + sc];
            v = vjf_sum32(v);
            if (l == 0 && ((A.sc_mask >> sc) & 1u)) put(A.red + (sc < RS_SDX2 ? P.red_SCA : P.red_SC) + sc, (float)v);   // (loss sums behind the gradients)
        }
        arrive();
        return;
    }
    const VjfJob job = A.jobs[A.job0 + blockIdx.x];
    const float* slab = A.slabs + (size_t)(A.job0 + blockIdx.x) * A.nsplit * 1024;
    for (int e = tid; e < 1024; e += (int)blockDim.x) {
        float v = 0.f;
        int s = 0;
        for (; s + 16 <= A.nsplit; s += 16) {                    // 16 independent loads, summed in split order
            float t[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) t[q] = slab[(size_t)(s + q) * 1024 + e];
#pragma unroll
            for (int q = 0; q < 16; ++q) v += t[q];
        }
        for (; s + 8 <= A.nsplit; s += 8) {
            float t[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) t[q] = slab[(size_t)(s + q) * 1024 + e];
#pragma unroll
            for (int q = 0; q < 8; ++q) v += t[q];
        }
        for (; s < A.nsplit; ++s) v += slab[(size_t)s * 1024 + e];
        const int i = e >> 5, j = e & 31;
        if (i >= job.xn || j >= job.yn) continue;
        if (job.kind == 0) {
            const int gr = job.ti * VJF_TILE + i, gc = job.tj * VJF_TILE + j;   // gr: X column, gc: Y column of E
            if (gr < P.n) {
                if (gc < P.n && gc <= gr) { put(A.red + P.red_G + (size_t)gr * P.n + gc, v); put(A.red + P.red_G + (size_t)gc * P.n + gr, v); }
            } else if (gr < P.n + P.dz && gc < P.n) {
                put(A.red + P.red_FDX + (size_t)gc * P.dz + (gr - P.n), v);
            }
        } else {
            if (j < job.ncol_w) put(A.red + job.dst + (size_t)i * job.ld + j, v);
            else if (j == job.ncol_w && job.dst_b >= 0) put(A.red + job.dst_b + i, v);
        }
    }
    arrive();
}
