// Diagnostic micro-benchmark (not part of the product): potrf_inv_chain in isolation, one wavefront.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../vjf_amd/csrc/vjf_chol_kernel.h"
__global__ void k(float* out, unsigned long long* t, int waves_active) {
    __shared__ float blk[1024], inv[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) { int r = e >> 5, c = e & 31; blk[vsw(r, c)] = (r == c ? 40.f : 0.f) + 1.0f / (1 + r + c); }
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0, t2 = 0;
    if (wave == 0) {
        vjf_f32x16 acc;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        blk_load(acc, blk, lane);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        bool ok = potrf_inv_chain(acc, blk, inv, lane);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
        if (!ok) out[0] = -1;
    }
    __syncthreads();
    out[threadIdx.x] = blk[threadIdx.x] + inv[threadIdx.x];
    if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; }
}
int main() {
    float* out; unsigned long long* t; hipMalloc(&out, 4096); hipMalloc(&t, 64);
    unsigned long long h[2];
    for (int threads : {64, 512, 64, 512}) {
        k<<<1, threads>>>(out, t, 0); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("threads=%d: blk_load %llu, potrf_inv_chain(+stores) %llu cycles\n", threads, h[0], h[1]);
    }
    return 0;
}
