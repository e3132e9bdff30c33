// Diagnostic micro-benchmark variants of the Cholesky column chain (one wavefront).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../vjf_amd/csrc/vjf_chol_kernel.h"
template <int VAR>
__device__ __forceinline__ bool chain(vjf_f32x16& acc, float* out, float* inv, int lane) {
    const int c = lane & 31, h = lane >> 5;
    bool ok = true;
    float lcol[32], xrow[32];
    vjf_f32x16 racc;
#pragma unroll
    for (int r = 0; r < 16; ++r) racc[r] = (vrow(r, h) == c) ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int rj = (j & 3) + 4 * (j >> 3), hj = (j >> 2) & 1;
        const float d = vrl(acc[rj], j + 32 * hj);
        if (VAR != 3) if (!(d > 0.f) || !(d < 3.0e38f)) ok = false;
        const float s = __builtin_amdgcn_rsqf(d);
        const bool on = (h == hj);
        const float l = (on && (c >= j)) ? acc[rj] * s : 0.f;
        lcol[j] = l;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(-l, l, acc, 0, 0, 0);
        if (VAR != 1) {
            const float x = on ? racc[rj] * s : 0.f;
            const float lo = (on && (c > j)) ? -l : 0.f;
            xrow[j] = x;
            racc = __builtin_amdgcn_mfma_f32_32x32x2f32(lo, x, racc, 0, 0, 0);
        } else xrow[j] = 0.f;
    }
    if (VAR != 2) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int hj = (j >> 2) & 1;
            if (h == hj) {
                if (c >= j) out[vsw(c, j)] = lcol[j];
                inv[vsw(j, c)] = (c <= j) ? xrow[j] : 0.f;
            }
        }
    } else {
        float s = 0; for (int j = 0; j < 32; ++j) s += lcol[j] + xrow[j];
        out[lane] = s;
    }
    return ok;
}
template <int VAR>
__global__ void k(float* out, unsigned long long* t) {
    __shared__ float blk[1024], inv[1024];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) { int r = e >> 5, c = e & 31; blk[vsw(r, c)] = (r == c ? 40.f : 0.f) + 1.0f / (1 + r + c); }
    __syncthreads();
    unsigned long long t1, t2;
    vjf_f32x16 acc;
    blk_load(acc, blk, lane);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    bool ok = chain<VAR>(acc, blk, inv, lane);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    if (!ok) out[0] = -1;
    __syncthreads();
    out[threadIdx.x] = blk[threadIdx.x] + inv[threadIdx.x];
    if (threadIdx.x == 0) t[0] = t2 - t1;
}
int main() {
    float* out; unsigned long long* t; hipMalloc(&out, 4096); hipMalloc(&t, 64);
    unsigned long long h;
    for (int rep = 0; rep < 2; ++rep) {
        k<0><<<1, 64>>>(out, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("full                 : %llu\n", h);
        k<1><<<1, 64>>>(out, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("no second mfma       : %llu\n", h);
        k<2><<<1, 64>>>(out, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("no lds stores        : %llu\n", h);
        k<3><<<1, 64>>>(out, t); hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost); printf("no ok check          : %llu\n", h);
    }
    return 0;
}
