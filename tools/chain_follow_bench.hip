// Diagnostic micro-benchmark (not part of the product): the streamed column chain (vjf_chol_kernel.h: potrf_inv_chain2_bcast on
// wavefront 0 = SIMD 0, potrf_follow on wavefront 1 = SIMD 1), checked against a host computation; cycles of the chain alone, of the
// chain while it publishes, and when the follower is through (it starts DELAY cycles after the chain: the late join of the product).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o tools/chain_follow_bench tools/chain_follow_bench.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include "../vjf_amd/csrc/vjf_chol_kernel.h"
#define NREP 12
__global__ __launch_bounds__(128) void k(const float* A, const float* Pn, const float* Nn, float* out, unsigned long long* t, int mode, int delay) {
    extern __shared__ float lds[];                     // (dynamic, carved by offsets as the product does)
    float* blk = lds; float* inv = blk + 1024; float* pan = inv + 1024; float* nxt = pan + 1024; float* ring = nxt + 1024; float* rsc = ring + 1024;
    volatile int* rnd = (volatile int*)(rsc + 48);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool ok = true;
    for (int rep = 0; rep < NREP; ++rep) {
        __syncthreads();
        for (int e = threadIdx.x; e < 1024; e += blockDim.x) { int r = e >> 5, c = e & 31; blk[vsw(r, c)] = A[e]; pan[vsw(r, c)] = Pn[e]; nxt[vsw(r, c)] = Nn[e]; }
        if (threadIdx.x == 0) rnd[0] = 16 * rep;
        __syncthreads();
        unsigned long long t1, t2;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (wave == 0) {
            if (mode == 0) ok = potrf_inv_chain2(blk, inv, lane) && ok;
            else ok = potrf_inv_chain2_bcast(blk, inv, lane, (vjf_lds_f*)ring, (vjf_lds_f*)rsc, (vjf_lds_vi*)rnd, 16 * rep) && ok;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
            if (lane == 0) t[rep * 2] = t2 - t1;
        } else if (mode == 2) {
            unsigned long long tn;
            do { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn)::"memory"); } while ((long long)(tn - t1) < delay);
            ok = potrf_follow(pan, nxt, lane, (vjf_lds_f*)ring, (vjf_lds_f*)rsc, (vjf_lds_vi*)rnd, 16 * rep, []() { return true; }) && ok;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2)::"memory");
            if (lane == 0) t[rep * 2 + 1] = t2 - t1;
        }
    }
    if (!ok) out[0] = -1;
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) { int r = e >> 5, c = e & 31; out[e] = blk[vsw(r, c)]; out[1024 + e] = inv[vsw(r, c)]; out[2048 + e] = pan[vsw(r, c)]; out[3072 + e] = nxt[vsw(r, c)]; }
}
int main() {
    static float A[1024], Pn[1024], Nn[1024], ho[4096];
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
        A[r * 32 + c] = (r == c ? 40.f : 0.f) + 1.0f / (1 + r + c); Pn[r * 32 + c] = sinf(0.37f * r + 1.3f * c) + 0.1f * r;
        Nn[r * 32 + c] = (r == c ? 90.f : 0.f) + cosf(0.1f * (r + c));
    }
    static double L[32][32], X[32][32], Li[32][32], N2[32][32];
    for (int j = 0; j < 32; ++j) {
        double d = A[j * 32 + j]; for (int m = 0; m < j; ++m) d -= L[j][m] * L[j][m];
        L[j][j] = sqrt(d);
        for (int i = j + 1; i < 32; ++i) { double v = A[i * 32 + j]; for (int m = 0; m < j; ++m) v -= L[i][m] * L[j][m]; L[i][j] = v / L[j][j]; }
    }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double v = Pn[i * 32 + j]; for (int m = 0; m < j; ++m) v -= X[i][m] * L[j][m]; X[i][j] = v / L[j][j]; }
    for (int c = 0; c < 32; ++c) for (int i = 0; i < 32; ++i) { double v = (i == c); for (int m = 0; m < i; ++m) v -= L[i][m] * Li[m][c]; Li[i][c] = v / L[i][i]; }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) { double v = Nn[i * 32 + j]; for (int m = 0; m < 32; ++m) v -= X[i][m] * X[j][m]; N2[i][j] = v; }
    float *dA, *dP, *dN, *out; unsigned long long* t;
    (void)hipMalloc(&dA, 4096); (void)hipMalloc(&dP, 4096); (void)hipMalloc(&dN, 4096); (void)hipMalloc(&out, 4 * 4096); (void)hipMalloc(&t, 16 * NREP);
    (void)hipMemcpy(dA, A, 4096, hipMemcpyHostToDevice); (void)hipMemcpy(dP, Pn, 4096, hipMemcpyHostToDevice); (void)hipMemcpy(dN, Nn, 4096, hipMemcpyHostToDevice);
    unsigned long long h[2 * NREP];
    auto mn = [&](int q) { unsigned long long m = ~0ull; for (int r = 2; r < NREP; ++r) m = h[2 * r + q] < m ? h[2 * r + q] : m; return m; };
    auto run = [&](const char* name, int mode, int delay) {
        (void)hipMemset(t, 0, 16 * NREP);
        k<<<1, 128, (5 * 1024 + 64) * 4>>>(dA, dP, dN, out, t, mode, delay);
        (void)hipMemcpy(h, t, sizeof h, hipMemcpyDeviceToHost); (void)hipMemcpy(ho, out, sizeof ho, hipMemcpyDeviceToHost);
        double eL = 0, eI = 0, eX = 0, eN = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) {
            if (c <= r) { eL = fmax(eL, fabs(ho[r * 32 + c] - L[r][c])); eI = fmax(eI, fabs(ho[1024 + r * 32 + c] - Li[r][c])); }
            if (mode == 2) { eX = fmax(eX, fabs(ho[2048 + r * 32 + c] - X[r][c])); eN = fmax(eN, fabs(ho[3072 + r * 32 + c] - N2[r][c])); }
        }
        printf("%-46s chain %6llu cycles, follower through at %6llu   max err L %.1e  Linv %.1e  L_(k+1,k) %.1e  next block %.1e%s\n", name, mn(0), mode == 2 ? mn(1) : 0ull, eL, eI, eX, eN,
               ho[0] == -1 ? "  (FAILED)" : "");
    };
    for (int rep = 0; rep < 2; ++rep) {
        run("plain rank-2 chain (potrf_inv_chain2)", 0, 0);
        run("publishing chain, nobody follows", 1, 0);
        run("publishing chain + follower from the start", 2, 0);
        run("publishing chain + follower 2000 cycles late", 2, 2000);
        run("publishing chain + follower 4000 cycles late", 2, 4000);
        run("publishing chain + follower 6000 cycles late", 2, 6000);
    }
    return 0;
}
