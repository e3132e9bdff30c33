// Diagnostic micro-benchmark (not part of the product): latency of a dependent
// MFMA -> VALU -> MFMA chain for the two f32 shapes, one wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void chain32(float* out, unsigned long long* t, int n) {
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x * 0.001f + i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float x = acc[j] * 0.5f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-x, x, acc, 0, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[threadIdx.x] = s; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void chain16(float* out, unsigned long long* t, int n) {
    f32x4 acc; for (int i = 0; i < 4; ++i) acc[i] = threadIdx.x * 0.001f + i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float x = acc[j & 3] * 0.5f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-x, x, acc, 0, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i];
    out[threadIdx.x] = s; if (threadIdx.x == 0) t[0] = t1 - t0;
}
// with readlane + rsq in the chain (the Cholesky column step)
__global__ void chain16_piv(float* out, unsigned long long* t, int n) {
    f32x4 acc; for (int i = 0; i < 4; ++i) acc[i] = 1.0f + threadIdx.x * 0.001f + i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc[j & 3]), j));
            float s = __builtin_amdgcn_rsqf(fabsf(d) + 1.0f);
            float x = acc[j & 3] * s * 1e-3f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(-x, x, acc, 0, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0; for (int i = 0; i < 4; ++i) s += acc[i];
    out[threadIdx.x] = s; if (threadIdx.x == 0) t[0] = t1 - t0;
}
__global__ void chain32_piv(float* out, unsigned long long* t, int n) {
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 1.0f + threadIdx.x * 0.001f + i;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(acc[j]), j));
            float s = __builtin_amdgcn_rsqf(fabsf(d) + 1.0f);
            float x = acc[j] * s * 1e-3f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-x, x, acc, 0, 0, 0);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
    out[threadIdx.x] = s; if (threadIdx.x == 0) t[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* t; hipMalloc(&out, 1024); hipMalloc(&t, 64);
    unsigned long long h; const int n = 64;
    for (int rep = 0; rep < 2; ++rep) {
        chain32<<<1, 64>>>(out, t, n); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("32x32x2  mfma->mul->mfma      : %.1f cycles/step\n", (double)h / (n * 16));
        chain16<<<1, 64>>>(out, t, n); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("16x16x4  mfma->mul->mfma      : %.1f cycles/step\n", (double)h / (n * 16));
        chain32_piv<<<1, 64>>>(out, t, n); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("32x32x2  +readlane+rsq+2mul   : %.1f cycles/step\n", (double)h / (n * 16));
        chain16_piv<<<1, 64>>>(out, t, n); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("16x16x4  +readlane+rsq+2mul   : %.1f cycles/step\n", (double)h / (n * 16));
    }
    return 0;
}
