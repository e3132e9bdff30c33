// Diagnostic micro-benchmark: the split Cholesky column chains (one wavefront each): potrf-only, inverse-only, panel trsm.
// Checks the three against a host computation and prints their cycle counts next to the merged production chain.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../vjf_amd/csrc/vjf_chol_kernel.h"

__global__ void spin(float* out, int n) {
    float v = threadIdx.x;
    for (int i = 0; i < n; ++i) v = fmaf(v, 1.0000001f, 0.5f);
    if (v == 123.f) out[0] = v;
}
__global__ __launch_bounds__(64) void longrun(const float* A, float* out, unsigned long long* t, int n) {
    __shared__ float blk[1024], piv[32], src[1024];
    const int lane = threadIdx.x & 63;
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) { int r = e >> 5, c = e & 31; src[vsw(r, c)] = A[e]; }
    __syncthreads();
    unsigned long long t1, t2;
    bool ok = true;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    for (int i = 0; i < n; ++i) {
        vjf_f32x16 acc;
        blk_load(acc, src, lane);
        ok = potrf_chain(acc, blk, piv, lane) && ok;
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
    if (!ok) out[0] = -1;
    out[threadIdx.x] = blk[threadIdx.x];
    if (threadIdx.x == 0) t[0] = t2 - t1;
}
#define NREP 12
template <int VAR>
__global__ __launch_bounds__(64) void k(const float* A, const float* Pn, float* out, unsigned long long* t) {
    __shared__ float blk[1024], inv[1024], pan[1024], piv[32];
    const int lane = threadIdx.x & 63;
    bool ok = true;
    for (int rep = 0; rep < NREP; ++rep) {
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += blockDim.x) { int r = e >> 5, c = e & 31; blk[vsw(r, c)] = A[e]; pan[vsw(r, c)] = Pn[e]; }
        __syncthreads();
        unsigned long long t1, t2, t3, t4;
        vjf_f32x16 acc;
        blk_load(acc, blk, lane);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (VAR == 2) {
            ok = potrf_inv_chain2(blk, inv, lane) && ok;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
            t3 = t4 = t2;
        } else if (VAR == 0) {
            ok = potrf_inv_chain(acc, blk, inv, lane) && ok;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
            t3 = t4 = t2;
        } else {
            ok = potrf_chain(acc, blk, piv, lane) && ok;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
            inv_chain(blk, piv, inv, lane);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3)::"memory");
            trsm_chain(pan, blk, piv, lane);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t4)::"memory");
        }
        if (threadIdx.x == 0) { t[rep * 3 + 0] = t2 - t1; t[rep * 3 + 1] = t3 - t2; t[rep * 3 + 2] = t4 - t3; }
    }
    if (!ok) out[0] = -1;
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) { int r = e >> 5, c = e & 31; out[e] = blk[vsw(r, c)]; out[1024 + e] = inv[vsw(r, c)]; out[2048 + e] = pan[vsw(r, c)]; }
}
int main() {
    static float A[1024], Pn[1024], ho[3072];
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) { A[r * 32 + c] = (r == c ? 40.f : 0.f) + 1.0f / (1 + r + c); Pn[r * 32 + c] = sinf(0.37f * r + 1.3f * c) + 0.1f * r; }
    // host: L, L^-1, panel X = Pn L^-T
    static double L[32][32], X[32][32], Li[32][32];
    for (int j = 0; j < 32; ++j) {
        double d = A[j * 32 + j]; for (int m = 0; m < j; ++m) d -= L[j][m] * L[j][m];
        L[j][j] = sqrt(d);
        for (int i = j + 1; i < 32; ++i) { double v = A[i * 32 + j]; for (int m = 0; m < j; ++m) v -= L[i][m] * L[j][m]; L[i][j] = v / L[j][j]; }
    }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double v = Pn[i * 32 + j]; for (int m = 0; m < j; ++m) v -= X[i][m] * L[j][m]; X[i][j] = v / L[j][j]; }
    for (int c = 0; c < 32; ++c) for (int i = 0; i < 32; ++i) { double v = (i == c); for (int m = 0; m < i; ++m) v -= L[i][m] * Li[m][c]; Li[i][c] = v / L[i][i]; }
    float *dA, *dP, *out; unsigned long long* t; hipMalloc(&dA, 4096); hipMalloc(&dP, 4096); hipMalloc(&out, 3 * 4096); hipMalloc(&t, 8 * 3 * NREP);
    hipMemcpy(dA, A, 4096, hipMemcpyHostToDevice); hipMemcpy(dP, Pn, 4096, hipMemcpyHostToDevice);
    unsigned long long h[3 * NREP];
    auto mn = [&](int q) { unsigned long long m = ~0ull; for (int r = 2; r < NREP; ++r) m = h[r * 3 + q] < m ? h[r * 3 + q] : m; return m; };
    spin<<<2048, 256>>>(out, 20000000); hipDeviceSynchronize();
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int n = 4000;
        longrun<<<1, 64>>>(dA, out, t, 10); hipDeviceSynchronize();
        hipEventRecord(e0); longrun<<<1, 64>>>(dA, out, t, n); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, t, 8, hipMemcpyDeviceToHost);
        printf("long run: %d potrf chains: %.3f ms wall = %.1f ns / chain; %llu ticks / chain => %.3f ticks/ns\n", n, ms, ms * 1e6 / n, h[0] / n, h[0] / (ms * 1e6));
    }
    for (int rep = 0; rep < 2; ++rep) {
        k<0><<<1, 64>>>(dA, dP, out, t); hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost);
        printf("merged potrf+inverse : min %llu  (first %llu)\n", mn(0), h[0]);
        k<2><<<1, 64>>>(dA, dP, out, t); hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost);
        {
            double eL = 0, eI = 0;
            for (int r = 0; r < 32; ++r) for (int c = 0; c <= r; ++c) { eL = fmax(eL, fabs(ho[r * 32 + c] - L[r][c])); eI = fmax(eI, fabs(ho[1024 + r * 32 + c] - Li[r][c])); }
            printf("rank-2 merged chain  : min %llu   max err L %.2e  Linv %.2e\n", mn(0), eL, eI);
        }
        k<1><<<1, 64>>>(dA, dP, out, t); hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost);
        double eL = 0, eI = 0, eX = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            if (c <= r) { eL = fmax(eL, fabs(ho[r * 32 + c] - L[r][c])); eI = fmax(eI, fabs(ho[1024 + r * 32 + c] - Li[r][c])); }
            eX = fmax(eX, fabs(ho[2048 + r * 32 + c] - X[r][c]));
        }
        printf("potrf %llu  inverse %llu  trsm %llu (min of 10)  max err L %.2e  Linv %.2e  panel %.2e\n", mn(0), mn(1), mn(2), eL, eI, eX);
    }
    return 0;
}
