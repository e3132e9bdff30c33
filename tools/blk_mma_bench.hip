// Diagnostic micro-benchmark: one trailing-update block product (load C, 16 MFMAs, store C) of the Cholesky kernel,
// alone and with all 8 wavefronts of the workgroup doing the same on their own blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../vjf_amd/csrc/vjf_chol_kernel.h"
#define NREP 16
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* t, int nw) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 24 * 1024; e += 512) lds[e] = 0.001f * (e % 977);
    __syncthreads();
    unsigned long long t1 = 0, t2 = 0;
    if (wave < nw) {
        float* cb = lds + wave * 3 * 1024;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        for (int rep = 0; rep < NREP; ++rep) {
            vjf_f32x16 acc;
            blk_load(acc, cb, lane);
            blk_mma<true>(acc, cb + 1024, cb + 2048, -1.f, lane);
            blk_store(acc, cb, lane);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    }
    __syncthreads();
    out[tid] = lds[tid];
    if (lane == 0) t[wave] = t2 - t1;
}
__global__ void spin(float* out, int n) { float v = threadIdx.x; for (int i = 0; i < n; ++i) v = fmaf(v, 1.0000001f, 0.5f); if (v == 123.f) out[0] = v; }
int main() {
    float* out; unsigned long long* t; hipMalloc(&out, 4096); hipMalloc(&t, 64);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 24 * 4096);
    spin<<<2048, 256>>>(out, 20000000); hipDeviceSynchronize();
    unsigned long long h[8];
    for (int nw : {1, 2, 4, 5, 8}) {
        for (int rep = 0; rep < 2; ++rep) { k<<<1, 512, 24 * 4096>>>(out, t, nw); hipDeviceSynchronize(); }
        hipMemcpy(h, t, 64, hipMemcpyDeviceToHost);
        printf("%d wavefronts: cycles per block product:", nw);
        for (int w = 0; w < nw; ++w) printf(" %llu", h[w] / NREP);
        printf("\n");
    }
    return 0;
}
