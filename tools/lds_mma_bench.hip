// Diagnostic micro-benchmark (not part of the product): the trial role's LDS-fed product out^T = W x^T (vjf_mega_kernel.h: mg_mma2_lds)
// for one recognition layer of config B -- W (128, 70) and 32 trials' activations in LDS, 8 wavefronts, one 16-row tile each, two
// 16-trial column groups -- in the variants below, one workgroup of 512 threads per compute unit, R repetitions.
//   hipcc -O3 --offload-arch=gfx950 -o tools/lds_mma_bench tools/lds_mma_bench.hip && tools/lds_mma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LD = 33, M = 128, K = 70, LDW = 74, NT = 512;

// V0: as the kernel has it (unroll 4, clamped k, masked A)
__device__ __forceinline__ void v0(f32x4& acc0, f32x4& acc1, const float* Ws, const float* Xs, int m0, int lane) {
    const int i = lane & 15, kk = lane >> 4;
    const bool rv = (m0 + i) < M;
    const int mi = rv ? m0 + i : 0;
    const float* xp = Xs + i;
    const int klast = K - 1;
#pragma unroll 4
    for (int ks = 0; ks < K; ks += 4) {
        const int k = ks + kk, kc = min(k, klast);
        const float w = Ws[mi * LDW + kc];
        const float av = (rv && k < K) ? w : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD + 16], acc1, 0, 0, 0);
    }
}
// V4: V0 with the shape as RUN-TIME values (what the product kernel has: K, the row stride and M come from the plan)
__device__ __forceinline__ void v4(f32x4& acc0, f32x4& acc1, const float* Ws, const float* Xs, int m0, int lane, int Kr, int ldw, int Mr) {
    const int i = lane & 15, kk = lane >> 4;
    const bool rv = (m0 + i) < Mr;
    const int mi = rv ? m0 + i : 0;
    const float* xp = Xs + i;
    const int klast = Kr - 1;
#pragma unroll 4
    for (int ks = 0; ks < Kr; ks += 4) {
        const int k = ks + kk, kc = min(k, klast);
        const float w = Ws[mi * ldw + kc];
        const float av = (rv && k < Kr) ? w : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xp[kc * LD + 16], acc1, 0, 0, 0);
    }
}
// V5: run-time shape, chunks of CH k-steps: a chunk's operands are read with immediate offsets from ONE base per chunk (no clamp, no mask:
// rows beyond M are computed and discarded by the caller; only the last, partial k-step is clamped and masked), the next chunk's reads are
// issued before this chunk's MFMAs
template <int CH>
__device__ __forceinline__ void v5(f32x4& acc0, f32x4& acc1, const float* Ws, const float* Xs, int m0, int lane, int Kr, int ldw, int Mr) {
    const int i = lane & 15, kk = lane >> 4;
    const int mi = (m0 + i) < Mr ? m0 + i : 0;
    const float* wp = Ws + mi * ldw + kk;
    const float* xp = Xs + i + kk * LD;
    const int nf = Kr >> 2;                              // full k-steps
    float a0[CH], p0[CH], q0[CH], a1[CH], p1[CH], q1[CH];
    auto ld = [&](float (&a)[CH], float (&b0)[CH], float (&b1)[CH], int s0) {
        const float* w = wp + 4 * s0; const float* x = xp + 4 * s0 * LD;
#pragma unroll
        for (int q = 0; q < CH; ++q) { a[q] = w[4 * q]; b0[q] = x[4 * q * LD]; b1[q] = x[4 * q * LD + 16]; }
    };
    auto mm = [&](const float (&a)[CH], const float (&b0)[CH], const float (&b1)[CH], int s0) {
#pragma unroll
        for (int q = 0; q < CH; ++q)
            if (s0 + q < nf) {                             // (uniform)
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b0[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b1[q], acc1, 0, 0, 0);
            }
    };
    // (reads of a chunk's steps beyond nf touch rows < 4 nf + 4 CH of the operands: inside the LDS allocation, values unused)
    if (nf > 0) ld(a0, p0, q0, 0);
    for (int s0 = 0; s0 < nf; s0 += 2 * CH) {
        if (s0 + CH < nf) ld(a1, p1, q1, s0 + CH);
        mm(a0, p0, q0, s0);
        if (s0 + 2 * CH < nf) ld(a0, p0, q0, s0 + 2 * CH);
        if (s0 + CH < nf) mm(a1, p1, q1, s0 + CH);
    }
    if (Kr & 3) {                                          // the partial step
        const int k = 4 * nf + kk, kc = min(k, Kr - 1);
        const float w = Ws[mi * ldw + kc];
        const float av = k < Kr ? w : 0.f;
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Xs[kc * LD + i], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, Xs[kc * LD + i + 16], acc1, 0, 0, 0);
    }
}
// V1: every operand of the tile read into registers first (18 k-steps: 54 values), then the 36 MFMAs
__device__ __forceinline__ void v1(f32x4& acc0, f32x4& acc1, const float* Ws, const float* Xs, int m0, int lane) {
    const int i = lane & 15, kk = lane >> 4;
    const int mi = m0 + i;
    const float* xp = Xs + i;
    constexpr int NS = (K + 3) / 4;
    float a[NS], b0[NS], b1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * s + kk, kc = min(k, K - 1);
        const float w = Ws[mi * LDW + kc];
        a[s] = k < K ? w : 0.f; b0[s] = xp[kc * LD]; b1[s] = xp[kc * LD + 16];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc1, 0, 0, 0);
    }
}
// V2: four accumulator chains per wavefront (K split in two halves per column group), operands first
__device__ __forceinline__ void v2(f32x4& acc0, f32x4& acc1, const float* Ws, const float* Xs, int m0, int lane) {
    const int i = lane & 15, kk = lane >> 4;
    const int mi = m0 + i;
    const float* xp = Xs + i;
    constexpr int NS = (K + 3) / 4;
    float a[NS], b0[NS], b1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = 4 * s + kk, kc = min(k, K - 1);
        const float w = Ws[mi * LDW + kc];
        a[s] = k < K ? w : 0.f; b0[s] = xp[kc * LD]; b1[s] = xp[kc * LD + 16];
    }
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s + 1 < NS; s += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b0[s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b1[s], acc1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s + 1], b0[s + 1], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s + 1], b1[s + 1], c1, 0, 0, 0);
    }
    if (NS & 1) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[NS - 1], b0[NS - 1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[NS - 1], b1[NS - 1], acc1, 0, 0, 0);
    }
    acc0 += c0; acc1 += c1;
}
// V3: one 32x32x2 MFMA stream: wavefront w forms rows 32 (w & 3) .. + 31 for K half (w >> 2): 4 row tiles x 2 K halves
__device__ __forceinline__ void v3(f32x16& acc, const float* Ws, const float* Xs, int wave, int lane) {
    const int r = lane & 31, h = lane >> 5, m0 = 32 * (wave & 3), kb = (wave >> 2) * 36, ke = min(K, kb + 36);
    constexpr int NS = 18;
    float a[NS], b[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = kb + 2 * s + h, kc = min(k, K - 1);
        const float w = Ws[(m0 + r) * LDW + kc];
        a[s] = k < ke ? w : 0.f; b[s] = Xs[kc * LD + r];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc, 0, 0, 0);
}

template <int V>
__global__ __launch_bounds__(NT) void bench(float* out, unsigned long long* tks, int R, int Kr, int ldw, int Mr) {
    extern __shared__ float lds[];
    float* Ws = lds; float* Xs = lds + M * LDW; float* Os = Xs + K * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < M * LDW; e += NT) Ws[e] = 0.001f * (e % 97);
    for (int e = tid; e < K * LD; e += NT) Xs[e] = 0.01f * (e % 31);
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < R; ++it) {
        if (V != 3) {
            f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            if (V == 0) v0(acc0, acc1, Ws, Xs, wave * 16, lane);
            if (V == 1) v1(acc0, acc1, Ws, Xs, wave * 16, lane);
            if (V == 2) v2(acc0, acc1, Ws, Xs, wave * 16, lane);
            if (V == 4) v4(acc0, acc1, Ws, Xs, wave * 16, lane, Kr, ldw, Mr);
            if (V == 5) v5<8>(acc0, acc1, Ws, Xs, wave * 16, lane, Kr, ldw, Mr);
            if (V == 6) v5<4>(acc0, acc1, Ws, Xs, wave * 16, lane, Kr, ldw, Mr);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = wave * 16 + 4 * (lane >> 4) + r;
                Os[f * LD + (lane & 15)] = acc0[r]; Os[f * LD + 16 + (lane & 15)] = acc1[r];
            }
        } else {
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            v3(acc, Ws, Xs, wave, lane);
            // (partial sums of the two K halves would meet in LDS: one more pass, not timed differently here)
#pragma unroll
            for (int q = 0; q < 16; ++q) Os[(32 * (wave & 3) + (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5)) * LD + (lane & 31)] = acc[q];
        }
        __syncthreads();
        Xs[(it * 7 + tid) % (K * LD)] += 1e-6f * Os[tid % (M * LD)];     // (a dependence between repetitions)
        __syncthreads();
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (tid == 0) tks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * NT + tid] = Os[tid];
}

int main() {
    const int R = 2000, G = 256;
    float* out; unsigned long long* tk;
    hipMalloc(&out, G * NT * 4); hipMalloc(&tk, G * 8);
    const size_t lds = (size_t)(M * LDW + K * LD + M * LD) * 4;
    auto run = [&](auto kern, const char* name) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(G), dim3(NT), lds, 0, out, tk, 10, K, LDW, M);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(G), dim3(NT), lds, 0, out, tk, R, K, LDW, M);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-34s %7.3f us per layer product (2 wavefronts per SIMD x 36 v_mfma_f32_16x16x4_f32 x 32 cycles = 2304 cycles = 0.96 us at 2.4 GHz if the matrix pipe never waited)\n", name, ms * 1e3 / R);
    };
    run(bench<0>, "V0 as in the kernel (unroll 4)");
    run(bench<1>, "V1 operands first, then MFMAs");
    run(bench<2>, "V2 four chains per wavefront");
    run(bench<3>, "V3 32x32x2, K split over wave pairs");
    run(bench<4>, "V4 = V0, shape at run time");
    run(bench<5>, "V5 run-time shape, chunks of 8 steps");
    run(bench<6>, "V5 run-time shape, chunks of 4 steps");
    return 0;
}
