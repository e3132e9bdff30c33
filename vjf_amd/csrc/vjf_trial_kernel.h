// vjf_trial_kernel.h -- argument block of the trial-parallel half of one VJF.filter step and small dense helpers shared by
// the stand-alone operators.  The half itself lives in vjf_trial_mfma_kernel.h (16 trials per workgroup, everything in LDS) and
// vjf_trial_wide.h (one GEMM per layer over all trials, for working sets beyond LDS).  Per trial it restates (reference file:line):
//   xs  = mu_s + eps_s * exp(lv_s/2)                                   util.py:11-13, model.py:112
//   Phi = exp(-|[xs,u]-c|^2 / (2 w^2))                                 functional.py:11-22
//   pt  = (xs + Phi W,  log sum_j (Phi w_chol)_j^2)                    module.py:64-77, model.py:334-340
//   qt  = Recognition([y,u,mu_s,lv_s])                                 recognition.py:31-42
//   xt  = mu_t + eps_t * exp(lv_t/2);  py = xt C^T + d                 model.py:119-120, 28-30
//   per-trial loss terms and the hand-derived backward seeds           model.py:124-154, SURVEY 8a-bwd
// and leaves, for the Gram kernel (K1b), one row per trial of
//   E = [Phi | xt-xs], ACT = [in|1|h_1|1|..|h_L|1|xt|1], DEL = [da_1|..|da_L|dmu|dlv|dpy]
// plus per-workgroup partial sums of the loss terms (fixed order => run-to-run deterministic).
// Gradients are emitted WITHOUT the 1/B of the batch mean; K2 applies 1/B_total after the
// (optional) cross-device all-reduce.
#pragma once
#include <hip/hip_runtime.h>
#include "vjf_plan.h"

#define VJF_K1_THREADS 256

struct VjfTrialArgs {
    const float* y; const float* u; const float* mu_s; const float* lv_s;
    const float* eps_s; const float* eps_t;
    float* mu_t; float* lv_t;
    const float* state;
    float* E; float* ACT; float* DEL;
    float* partial;        // (gridDim.x, RS_N)
    int B;
    unsigned flags;
    // Replay of the backward half after a non-finite loss component (vjf/model.py:138-149): the SGD kernel left the parameters alone
    // and published which components the reference drops (bit 0 recon, 1 dynamics, 2 entropy) and the likelihood log-variance the
    // step started with; `replay` launches return at once when the word is 0, else form the seeds without the dropped components'.
    const unsigned* replay_mask; const float* replay_rho; int replay;
};

// out[b][f] = sum_k X[b][k] * W[f][k]      W row-major (N,K): a torch Linear weight.
template <int TB, class Epi>
__device__ __forceinline__ void dense_nt(const float* __restrict__ W, int N, int K, const float* Xs, int ldx, Epi epi) {
    for (int f = threadIdx.x; f < N; f += VJF_K1_THREADS) {
        float acc[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[b] = 0.f;
        const float* w = W + (size_t)f * K;
        for (int k = 0; k < K; ++k) {
            const float wk = w[k];
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[b] = fmaf(Xs[b * ldx + k], wk, acc[b]);
        }
        epi(f, acc);
    }
}

// out[b][j] = sum_k X[b][k] * M[k][j]      M row-major (K,N): coalesced over j.
template <int TB, class Epi>
__device__ __forceinline__ void dense_nn(const float* __restrict__ M, int K, int N, const float* Xs, int ldx, Epi epi) {
    for (int j = threadIdx.x; j < N; j += VJF_K1_THREADS) {
        float acc[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) acc[b] = 0.f;
        for (int k = 0; k < K; ++k) {
            const float m = M[(size_t)k * N + j];
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[b] = fmaf(Xs[b * ldx + k], m, acc[b]);
        }
        epi(j, acc);
    }
}

// sum over the LPT lanes that share a trial (LPT is a power of two <= 64, lanes are contiguous)
template <int LPT>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPT / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

