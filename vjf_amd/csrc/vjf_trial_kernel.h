// vjf_trial_kernel.h -- K1: the trial-parallel half of one VJF.filter step.
//
// One workgroup (256 threads = 4 wavefronts) owns TB consecutive trials and keeps every
// per-trial intermediate in LDS.  It restates, per trial (reference file:line):
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

template <int TB>
__global__ __launch_bounds__(VJF_K1_THREADS) void vjf_trial_kernel(VjfPlan P, VjfTrialArgs A) {
    constexpr int LPT = VJF_K1_THREADS / TB;       // lanes per trial for the per-trial reductions
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * TB;
    const int nb = min(TB, A.B - b0);              // valid trials of this workgroup
    const int dz = P.dz, dy = P.dy, du = P.du, n = P.n, din = P.din, dxu = P.dxu;
    const float* S = A.state;
    const bool prior = (A.mu_s == nullptr);
    const bool warm = (A.flags & VJF_FLAG_WARM_UP) != 0;

    // ---- LDS carve (floats)
    float* s_in = smem;                        // TB x din
    float* s_xu = s_in + TB * din;             // TB x dxu        [xs | u]
    float* s_phi = s_xu + TB * dxu;            // TB x n
    float* s_act = s_phi + TB * n;             // TB x hsum       hidden activations, layer-major per trial
    float* s_d0 = s_act + TB * P.hsum;         // TB x hmax
    float* s_d1 = s_d0 + TB * P.hmax;          // TB x hmax
    float* s_mu = s_d1 + TB * P.hmax;          // TB x dz   mu_t
    float* s_lv = s_mu + TB * dz;              // TB x dz   lv_t
    float* s_xt = s_lv + TB * dz;              // TB x dz
    float* s_e2 = s_xt + TB * dz;              // TB x dz   eps_t
    float* s_pm = s_e2 + TB * dz;              // TB x dz   pt.mean
    float* s_dmu = s_pm + TB * dz;             // TB x dz
    float* s_dlv = s_dmu + TB * dz;            // TB x dz
    float* s_py = s_dlv + TB * dz;             // TB x dy
    float* s_dpy = s_py + TB * dy;             // TB x dy
    float* s_plv = s_dpy + TB * dy;            // TB        pt.logvar (one value per trial)
    float* s_sc = s_plv + TB;                  // TB x RS_N per-trial scalars
    float* s_red = s_sc + TB * RS_N;           // 4 x TB    cross-wave partials of the variance row sum

    // ---- stage 0: inputs, xs
    for (int i = tid; i < TB * din; i += VJF_K1_THREADS) {
        const int b = i / din, c = i - b * din;
        float v = 0.f;
        if (b < nb) {
            const size_t g = (size_t)(b0 + b);
            if (c < dy) v = A.y[g * dy + c];
            else if (c < dy + du) v = A.u[g * du + (c - dy)];
            else if (c < dy + du + dz) { const int j = c - dy - du; v = prior ? S[P.off[VJF_SLOT_PRIOR_MEAN] + j] : A.mu_s[g * dz + j]; }
            else { const int j = c - dy - du - dz; v = prior ? S[P.off[VJF_SLOT_PRIOR_LOGVAR] + j] : A.lv_s[g * dz + j]; }
        }
        s_in[i] = v;
    }
    for (int i = tid; i < TB * dz; i += VJF_K1_THREADS) {
        const int b = i / dz;
        s_e2[i] = (b < nb) ? A.eps_t[(size_t)b0 * dz + i] : 0.f;
    }
    __syncthreads();
    for (int i = tid; i < TB * dxu; i += VJF_K1_THREADS) {
        const int b = i / dxu, c = i - b * dxu;
        float v;
        if (c < dz) {
            const float mu = s_in[b * din + dy + du + c], lv = s_in[b * din + dy + du + dz + c];
            const float e = (b < nb) ? A.eps_s[(size_t)(b0 + b) * dz + c] : 0.f;
            v = fmaf(e, expf(0.5f * lv), mu);
        } else {
            v = s_in[b * din + dy + (c - dz)];
        }
        s_xu[i] = v;
    }
    __syncthreads();

    // ---- stage 1: RBF features (functional.py:11-22), also the Phi part of E
    {
        const float* cen = S + P.off[VJF_SLOT_CENTROID];
        const float* lw = S + P.off[VJF_SLOT_LOGWIDTH];
        for (int i = tid; i < TB * n; i += VJF_K1_THREADS) {
            const int b = i / n, k = i - b * n;
            float d2 = 0.f;
            for (int c = 0; c < dxu; ++c) { const float d = s_xu[b * dxu + c] - cen[k * dxu + c]; d2 = fmaf(d, d, d2); }
            const float w = expf(lw[k]);
            const float ph = expf(-0.5f * d2 / (w * w));
            s_phi[i] = ph;
            if (b < nb) A.E[(size_t)(b0 + b) * P.ldE + k] = ph;
        }
    }
    __syncthreads();

    // ---- stage 2a: pt.mean = xs + Phi W   (module.py:77, model.py:338)
    {
        const float* Wm = S + P.off[VJF_SLOT_W_MEAN];
        dense_nn<TB>(Wm, n, dz, s_phi, n, [&](int j, const float* acc) {
#pragma unroll
            for (int b = 0; b < TB; ++b) s_pm[b * dz + j] = s_xu[b * dxu + j] + acc[b];
        });
    }
    // ---- stage 2b: predictive variance  sum_j (Phi w_chol)[b][j]^2   (module.py:75-76, O(B n^2) form)
    {
        const float* Wc = S + P.off[VJF_SLOT_W_CHOL];
        float v2[TB];
#pragma unroll
        for (int b = 0; b < TB; ++b) v2[b] = 0.f;
        dense_nn<TB>(Wc, n, n, s_phi, n, [&](int, const float* acc) {
#pragma unroll
            for (int b = 0; b < TB; ++b) v2[b] = fmaf(acc[b], acc[b], v2[b]);
        });
        // fixed-order reduction: lanes of a wave (xor tree), then the 4 waves in order
#pragma unroll
        for (int b = 0; b < TB; ++b) {
            float v = v2[b];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if ((tid & 63) == 0) s_red[(tid >> 6) * TB + b] = v;
        }
    }
    __syncthreads();
    if (tid < TB) s_plv[tid] = logf(((s_red[tid] + s_red[TB + tid]) + s_red[2 * TB + tid]) + s_red[3 * TB + tid]);

    // ---- stage 3: recognition forward (recognition.py:31-42)
    {
        const float* xin = s_in; int kin = din, ldx = din; int aoff = 0;
        for (int l = 0; l < P.L; ++l) {
            const float* W = S + P.off[VJF_SLOT_REC_W0 + 2 * l];
            const float* bias = S + P.off[VJF_SLOT_REC_B0 + 2 * l];
            float* out = s_act + TB * aoff;            // TB x h[l]
            const int hl = P.h[l];
            dense_nt<TB>(W, hl, kin, xin, ldx, [&](int f, const float* acc) {
                const float bf = bias[f];
#pragma unroll
                for (int b = 0; b < TB; ++b) out[b * hl + f] = tanhf(acc[b] + bf);
            });
            __syncthreads();
            xin = out; kin = hl; ldx = hl; aoff += hl;
        }
        const float* Wm = S + P.off[VJF_SLOT_MEAN_W];
        const float* Wl = S + P.off[VJF_SLOT_LV_W];
        const float* bl = S + P.off[VJF_SLOT_LV_B];
        // both heads in one pass: f in [0,dz) -> mean, [dz,2dz) -> logvar
        for (int f = tid; f < 2 * dz; f += VJF_K1_THREADS) {
            const bool is_lv = f >= dz;
            const int j = is_lv ? f - dz : f;
            const float* w = (is_lv ? Wl : Wm) + (size_t)j * kin;
            float acc[TB];
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[b] = 0.f;
            for (int k = 0; k < kin; ++k) {
                const float wk = w[k];
#pragma unroll
                for (int b = 0; b < TB; ++b) acc[b] = fmaf(xin[b * ldx + k], wk, acc[b]);
            }
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                if (is_lv) s_lv[b * dz + j] = acc[b] + bl[j];
                else s_mu[b * dz + j] = acc[b];
            }
        }
    }
    __syncthreads();

    // ---- stage 4: xt, py, posterior outputs, dx part of E
    for (int i = tid; i < TB * dz; i += VJF_K1_THREADS) {
        const int b = i / dz, j = i - b * dz;
        const float xt = fmaf(s_e2[i], expf(0.5f * s_lv[i]), s_mu[i]);
        s_xt[i] = xt;
        if (b < nb) {
            const size_t g = (size_t)(b0 + b);
            A.mu_t[g * dz + j] = s_mu[i];
            A.lv_t[g * dz + j] = s_lv[i];
            A.E[g * P.ldE + n + j] = xt - s_xu[b * dxu + j];
        }
    }
    for (int i = tid; i < nb * (P.ldE - n - dz); i += VJF_K1_THREADS) {      // zero padding of E rows
        const int w = P.ldE - n - dz, b = i / w, c = i - b * w;
        A.E[(size_t)(b0 + b) * P.ldE + n + dz + c] = 0.f;
    }
    __syncthreads();
    {
        const float* C = S + P.off[VJF_SLOT_DEC_W];
        const float* d = S + P.off[VJF_SLOT_DEC_B];
        dense_nt<TB>(C, dy, dz, s_xt, dz, [&](int f, const float* acc) {
            const float bf = d[f];
#pragma unroll
            for (int b = 0; b < TB; ++b) s_py[b * dy + f] = acc[b] + bf;
        });
    }
    __syncthreads();

    // ---- stage 5: per-trial loss terms and backward seeds (no 1/B)
    {
        const int b = tid / LPT, s = tid % LPT;
        const float rho = S[P.off[VJF_SLOT_LIK_LOGVAR]];
        const float sig = S[P.off[VJF_SLOT_TR_LOGVAR]];
        float lrec = 0.f, ssey = 0.f;
        if (P.lik == VJF_LIK_GAUSSIAN) {                       // likelihood.py:19-26, functional.py:54-73
            const float p = expf(-0.5f * rho), e = expf(-rho);
            for (int i = s; i < dy; i += LPT) {
                const float yv = s_in[b * din + i], pv = s_py[b * dy + i];
                const float r = pv - yv, dsc = yv * p - pv * p;
                lrec += 0.5f * (dsc * dsc + rho);
                ssey = fmaf(r, r, ssey);
                s_dpy[b * dy + i] = e * r;
            }
        } else {                                               // likelihood.py:51-62
            for (int i = s; i < dy; i += LPT) {
                const float yv = s_in[b * din + i], pv = s_py[b * dy + i];
                const float eta = fminf(pv, 10.f), ex = expf(eta);
                lrec += ex - yv * eta;
                const float r = pv - yv;
                ssey = fmaf(r, r, ssey);
                s_dpy[b * dy + i] = (pv <= 10.f) ? (ex - yv) : 0.f;
            }
        }
        lrec = group_sum<LPT>(lrec);
        ssey = group_sum<LPT>(ssey);
        float ldyn = 0.f, ent = 0.f, sdx2 = 0.f;
        {
            const float p = expf(-0.5f * sig), e = expf(-sig), plv = s_plv[b];
            for (int j = s; j < dz; j += LPT) {                // model.py:390-391, functional.py:62-75
                const float mp = s_pm[b * dz + j], mu = s_mu[b * dz + j], lv = s_lv[b * dz + j];
                const float dsc = mp * p - mu * p;
                const float tr = expf(plv + lv - sig);
                ldyn += 0.5f * (dsc * dsc + sig) + 0.5f * tr;
                ent += 0.5f * lv;                              // functional.py:25-29
                const float dx = s_xt[b * dz + j] - s_xu[b * dxu + j];
                sdx2 = fmaf(dx, dx, sdx2);
                float dmu = 0.f, dlv = -0.5f;
                if (!warm) { dmu = -e * (mp - mu); dlv += 0.5f * tr; }
                s_dmu[b * dz + j] = dmu;
                s_dlv[b * dz + j] = dlv;
            }
        }
        ldyn = group_sum<LPT>(ldyn);
        ent = group_sum<LPT>(ent);
        sdx2 = group_sum<LPT>(sdx2);
        if (s == 0) {
            const bool ok = b < nb;
            s_sc[b * RS_N + RS_LRECON] = ok ? lrec : 0.f;
            s_sc[b * RS_N + RS_LDYN] = ok ? ldyn : 0.f;
            s_sc[b * RS_N + RS_ENT] = ok ? ent : 0.f;
            s_sc[b * RS_N + RS_SSEY] = ok ? ssey : 0.f;
            s_sc[b * RS_N + RS_SDX2] = ok ? sdx2 : 0.f;
        }
    }
    __syncthreads();
    if (tid < RS_N) {
        float v = 0.f;
        if (tid <= RS_SDX2) for (int b = 0; b < TB; ++b) v += s_sc[b * RS_N + tid];
        A.partial[(size_t)blockIdx.x * RS_N + tid] = v;
    }

    // ---- stage 6: backward through decoder, reparametrisation, heads, MLP
    {   // dxt = dpy C ; dmu += dxt ; dlv += dxt * eps_t * exp(lv/2)/2
        const float* C = S + P.off[VJF_SLOT_DEC_W];
        dense_nn<TB>(C, dy, dz, s_dpy, dy, [&](int j, const float* acc) {
#pragma unroll
            for (int b = 0; b < TB; ++b) {
                const int i = b * dz + j;
                s_dmu[i] += acc[b];
                s_dlv[i] = fmaf(acc[b] * s_e2[i], 0.5f * expf(0.5f * s_lv[i]), s_dlv[i]);
            }
        });
    }
    __syncthreads();
    {
        const int hL = P.h[P.L - 1];
        const float* Wm = S + P.off[VJF_SLOT_MEAN_W];
        const float* Wl = S + P.off[VJF_SLOT_LV_W];
        float* dh = s_d0;
        const float* hact = s_act + TB * (P.hsum - hL);
        // dh_L = dmu Wm + dlv Wl, then da_L = dh_L * (1 - h_L^2)
        for (int k = tid; k < hL; k += VJF_K1_THREADS) {
            float acc[TB];
#pragma unroll
            for (int b = 0; b < TB; ++b) acc[b] = 0.f;
            for (int j = 0; j < dz; ++j) {
                const float wm = Wm[(size_t)j * hL + k], wl = Wl[(size_t)j * hL + k];
#pragma unroll
                for (int b = 0; b < TB; ++b) acc[b] = fmaf(s_dmu[b * dz + j], wm, fmaf(s_dlv[b * dz + j], wl, acc[b]));
            }
#pragma unroll
            for (int b = 0; b < TB; ++b) { const float hv = hact[b * hL + k]; dh[b * hL + k] = acc[b] * (1.f - hv * hv); }
        }
        __syncthreads();
        int aoff = P.hsum - hL;
        float* cur = s_d0; float* nxt = s_d1;
        for (int l = P.L - 1; l >= 0; --l) {
            const int hl = P.h[l];
            // cur = da_l (TB x hl): store to DEL
            for (int i = tid; i < nb * hl; i += VJF_K1_THREADS) {
                const int b = i / hl, k = i - b * hl;
                A.DEL[(size_t)(b0 + b) * P.ldD + P.colD_da[l] + k] = cur[b * hl + k];
            }
            if (l > 0) {
                const int hp = P.h[l - 1];
                const float* W = S + P.off[VJF_SLOT_REC_W0 + 2 * l];      // (hl, hp)
                const float* hprev = s_act + TB * (aoff - hp);
                dense_nn<TB>(W, hl, hp, cur, hl, [&](int k, const float* acc) {
#pragma unroll
                    for (int b = 0; b < TB; ++b) { const float hv = hprev[b * hp + k]; nxt[b * hp + k] = acc[b] * (1.f - hv * hv); }
                });
                __syncthreads();
                float* t = cur; cur = nxt; nxt = t;
                aoff -= hp;
            }
        }
    }

    // ---- stage 7: ACT rows and the remaining DEL columns
    for (int i = tid; i < nb * P.ldA; i += VJF_K1_THREADS) {
        const int b = i / P.ldA, c = i - b * P.ldA;
        float v = 0.f;
        if (c < din) v = s_in[b * din + c];
        else if (c == din) v = 1.f;
        else if (c >= P.colA_xt) { const int j = c - P.colA_xt; v = j < dz ? s_xt[b * dz + j] : (j == dz ? 1.f : 0.f); }
        else {
            int l = 0, aoff = 0;
            while (l + 1 < P.L && c >= P.colA_act[l + 2]) { aoff += P.h[l]; ++l; }
            const int k = c - P.colA_act[l + 1];
            v = k < P.h[l] ? s_act[TB * aoff + b * P.h[l] + k] : 1.f;
        }
        A.ACT[(size_t)(b0 + b) * P.ldA + c] = v;
    }
    for (int i = tid; i < nb * (2 * dz + dy); i += VJF_K1_THREADS) {
        const int w = 2 * dz + dy, b = i / w, c = i - b * w;
        float v;
        if (c < dz) v = s_dmu[b * dz + c];
        else if (c < 2 * dz) v = s_dlv[b * dz + (c - dz)];
        else v = s_dpy[b * dy + (c - 2 * dz)];
        A.DEL[(size_t)(b0 + b) * P.ldD + P.colD_dmu + c] = v;
    }
}

static inline size_t vjf_trial_lds_floats(const VjfPlan& P, int TB) {
    return (size_t)TB * (P.din + P.dxu + P.n + P.hsum + 2 * P.hmax + 7 * P.dz + 2 * P.dy + 1 + RS_N + 4);
}
