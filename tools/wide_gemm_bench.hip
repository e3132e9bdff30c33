// Diagnostic micro-benchmark of the wide-route GEMM kernels on the shapes of BASELINE configs[4] (run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -o /tmp/wgb tools/wide_gemm_bench.hip && /tmp/wgb
// Every variant is checked against a plain one-thread-per-element product and timed over 20 launches (HIP events).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <functional>
#include "../vjf_amd/csrc/vjf_trial_wide.h"

__global__ void ref_gemm(VjfWideGemm g) {
    const int n = blockIdx.x * 64 + threadIdx.x, m = blockIdx.y;
    if (n >= g.N) return;
    float v = 0.f;
    for (int k = 0; k < g.K; ++k) {
        const float a = g.ta ? g.A[(size_t)k * g.lda + m] : g.A[(size_t)m * g.lda + k];
        const float b = g.nt ? g.Bm[(size_t)n * g.ldb + k] : g.Bm[(size_t)k * g.ldb + n];
        v = fmaf(a, b, v);
    }
    g.C[(size_t)m * g.ldc + n] = v;
}
// calibration: bare v_mfma_f32_32x32x2_f32 loop, 4 accumulators per wavefront, 4 wavefronts per workgroup
__global__ __launch_bounds__(256) void mfma_peak(float* out, int iters) {
    vjf_f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.f - a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float v = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) v += acc[i][r];
    if (v == 123.456f) out[0] = v;
}
struct Shape { const char* name; int M, N, K, nt, ta; };
int main() {
    const Shape shapes[] = {{"Z   = Phi w_chol", 4096, 1000, 1000, 0, 0}, {"h0  = in W0^T", 4096, 512, 640, 1, 0}, {"da0 = da1 W1", 4096, 512, 512, 0, 0},
                            {"py  = xt C^T", 4096, 512, 64, 1, 0}, {"mu  = h Wm^T", 4096, 64, 512, 1, 0}, {"pm  = Phi W", 4096, 64, 1000, 0, 0},
                            {"y   = X g", 1000, 64, 1000, 0, 0}, {"W   = X^T y", 1000, 64, 1000, 0, 1}};
    const size_t cap = (size_t)4096 * 1024;
    float *A, *B, *C, *R;
    hipMalloc(&A, cap * 4); hipMalloc(&B, cap * 4); hipMalloc(&C, cap * 4); hipMalloc(&R, cap * 4);
    std::vector<float> h(cap);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    hipMemcpy(A, h.data(), cap * 4, hipMemcpyHostToDevice);
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    hipMemcpy(B, h.data(), cap * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> hc(cap), hr(cap);
    for (int wgs : {256, 512, 1024}) {
        const int iters = 4096;
        hipLaunchKernelGGL(mfma_peak, dim3(wgs), dim3(256), 0, 0, C, iters);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(mfma_peak, dim3(wgs), dim3(256), 0, 0, C, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("bare MFMA loop, %4d workgroups x 4 waves: %.1f TFLOP/s\n", wgs, 5.0 * wgs * 4 * iters * 4 * 4096.0 / (ms * 1e-3) * 1e-12);
    }
    for (const Shape& sh : shapes) {
        VjfWideGemm g{};
        g.A = A; g.lda = sh.ta ? sh.M : sh.K; g.Bm = B; g.ldb = sh.nt ? sh.K : sh.N; g.C = R; g.ldc = sh.N;
        g.M = sh.M; g.N = sh.N; g.K = sh.K; g.nt = sh.nt; g.ta = sh.ta; g.epi = WEPI_NONE; g.va = g.lda % 4 == 0 && (sh.ta ? sh.M : sh.K) % 4 == 0; g.vb = g.ldb % 4 == 0 && (sh.nt ? sh.K : sh.N) % 4 == 0;
        hipLaunchKernelGGL(ref_gemm, dim3((sh.N + 63) / 64, sh.M), dim3(64), 0, 0, g);
        hipMemcpy(hr.data(), R, (size_t)sh.M * sh.N * 4, hipMemcpyDeviceToHost);
        g.C = C;
        const int tm = (sh.M + 127) / 128;
        struct Var { const char* name; std::function<void()> run; bool ok; };
        std::vector<Var> vars;
        vars.push_back({"64x64 (4 waves)", [&] { hipLaunchKernelGGL(vjf_wide_gemm_kernel, dim3((g.N + 63) / 64, (g.M + 63) / 64), dim3(256), 0, 0, g); }, true});
#define G3(TN, KC, NWR) [&] { const dim3 grid((g.N + TN - 1) / TN, tm); \
            if (g.nt) hipLaunchKernelGGL((vjf_wide_gemm3_kernel<TN, KC, NWR, false, true>), grid, dim3(NWR * 128), 0, 0, g); \
            else hipLaunchKernelGGL((vjf_wide_gemm3_kernel<TN, KC, NWR, false, false>), grid, dim3(NWR * 128), 0, 0, g); }
        vars.push_back({"128x128 k16 8w", G3(128, 16, 4), !sh.ta});
        vars.push_back({"128x64 k32 8w", G3(64, 32, 4), !sh.ta});
        vars.push_back({"128x64 k64 8w", G3(64, 64, 4), !sh.ta});
        vars.push_back({"skinny 8w", [&] { hipLaunchKernelGGL(vjf_skinny_gemm_kernel<8>, dim3((g.M + 31) / 32, (g.N + 31) / 32), dim3(512), 0, 0, g); }, sh.N <= 64});
        vars.push_back({"skinny 4w", [&] { hipLaunchKernelGGL(vjf_skinny_gemm_kernel<4>, dim3((g.M + 31) / 32, (g.N + 31) / 32), dim3(256), 0, 0, g); }, sh.N <= 64});
        printf("%s  M=%d N=%d K=%d nt=%d ta=%d  (%.2f GFLOP)\n", sh.name, sh.M, sh.N, sh.K, sh.nt, sh.ta, 2e-9 * sh.M * sh.N * sh.K);
        for (auto& v : vars) {
            if (!v.ok) continue;
            hipMemset(C, 0, (size_t)sh.M * sh.N * 4);
            v.run();
            if (hipDeviceSynchronize() != hipSuccess) { printf("   %-18s launch failed: %s\n", v.name, hipGetErrorString(hipGetLastError())); continue; }
            hipMemcpy(hc.data(), C, (size_t)sh.M * sh.N * 4, hipMemcpyDeviceToHost);
            double err = 0;
            for (size_t i = 0; i < (size_t)sh.M * sh.N; ++i) err = fmax(err, fabs((double)hc[i] - hr[i]));
            for (int i = 0; i < 3; ++i) v.run();
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) v.run();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("   %-18s %8.1f us  %6.1f TFLOP/s   max |err| %.2e\n", v.name, ms * 50.f, 2e-12 * sh.M * sh.N * sh.K / (ms * 50e-6), err);
        }
    }
    return 0;
}
