// Diagnostic micro-benchmark (not part of the product): do VALU instructions issue while a v_mfma_f32_* of the same SIMD runs?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/mfma_valu_overlap tools/mfma_valu_overlap.hip && tools/mfma_valu_overlap
// (1) ONE wavefront: a loop of {one MFMA ; N independent v_fma_f32}, four independent accumulators, s_memtime cycles per MFMA:
//     additive (64 + ~4.4 N for 32x32x2 f32) means the fp32 MFMA keeps the wavefront's VALU issue busy for its 16 passes.
// (2) TWO wavefronts of one SIMD (waves 0 and 4 of a 320-thread workgroup): one runs MFMAs only, the other v_fma only; each is timed
//     alone and beside the other.
// The same with v_mfma_f32_32x32x16_bf16 (8 passes on gfx950), which has a datapath of its own.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int N, bool BF16>
__global__ __launch_bounds__(64) void one_wave(float* out, unsigned long long* t, int iters) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = threadIdx.x * 0.001f + r;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    const float av = 0.5f + threadIdx.x, bv = 0.25f;
    bf16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(0.5f + i); bh[i] = (__bf16)0.25f; }
    unsigned long long t1, t2;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            if (BF16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[a], 0, 0, 0);
            else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < N; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], 1.0001f, 0.5f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    float s = 0;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) t[0] = (t2 - t1);
}

// waves 0 and 4 share SIMD 0 (waves of a workgroup go round the four SIMDs); mode bit 0: wave 0 runs MFMAs, bit 1: wave 4 runs v_fma
template <bool BF16>
__global__ __launch_bounds__(320) void two_waves(float* out, unsigned long long* t, int iters, int mode) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long t1, t2;
    float s = 0;
    __syncthreads();
    if (wave == 0 && (mode & 1)) {
        f32x16 acc[4];
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = lane * 0.001f + r;
        const float av = 0.5f + lane, bv = 0.25f;
        bf16x8 ah, bh;
        for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(0.5f + i); bh[i] = (__bf16)0.25f; }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                if (BF16) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[a], 0, 0, 0);
                else acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a], 0, 0, 0);
            }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
        for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
        if (lane == 0) t[0] = t2 - t1;
    }
    if (wave == 4 && (mode & 2)) {
        float v[8];
        for (int i = 0; i < 8; ++i) v[i] = lane + i;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) v[i & 7] = __builtin_fmaf(v[i & 7], 1.0001f, 0.5f);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
        for (int i = 0; i < 8; ++i) s += v[i];
        if (lane == 0) t[1] = t2 - t1;
    }
    out[threadIdx.x] = s;
}

template <int N, bool BF16> void run1(float* out, unsigned long long* t) {
    const int iters = 1000;
    one_wave<N, BF16><<<1, 64>>>(out, t, iters); one_wave<N, BF16><<<1, 64>>>(out, t, iters);
    unsigned long long h; (void)hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
    printf("  one wavefront, %-26s + %2d v_fma behind each: %6.1f cycles per MFMA\n", BF16 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32", N, (double)h / iters / 4);
}
template <bool BF16> void run2(float* out, unsigned long long* t) {
    const int iters = 1000;
    unsigned long long h[2], alone_m, alone_v;
    two_waves<BF16><<<1, 320>>>(out, t, iters, 1); two_waves<BF16><<<1, 320>>>(out, t, iters, 1); (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost); alone_m = h[0];
    two_waves<BF16><<<1, 320>>>(out, t, iters, 2); two_waves<BF16><<<1, 320>>>(out, t, iters, 2); (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost); alone_v = h[1];
    two_waves<BF16><<<1, 320>>>(out, t, iters, 3); two_waves<BF16><<<1, 320>>>(out, t, iters, 3); (void)hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("  two wavefronts of one SIMD, %-26s: MFMA wave %8llu cycles alone, v_fma wave %8llu alone; side by side %8llu and %8llu\n",
           BF16 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32", alone_m, alone_v, h[0], h[1]);
}
int main() {
    float* out; unsigned long long* t; (void)hipMalloc(&out, 4096); (void)hipMalloc(&t, 64);
    run1<0, false>(out, t); run1<8, false>(out, t); run1<16, false>(out, t); run1<32, false>(out, t);
    run1<0, true>(out, t); run1<8, true>(out, t); run1<16, true>(out, t); run1<32, true>(out, t);
    run2<false>(out, t); run2<true>(out, t);
    return 0;
}
