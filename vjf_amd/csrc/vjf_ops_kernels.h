// vjf_ops_kernels.h -- stand-alone operator kernels behind the vjf.module / vjf.functional surface.
// These serve the reference's operator API one call at a time; the fused step does not use them.
#pragma once
#include <hip/hip_runtime.h>

// functional.rbf (vjf/functional.py:11-22): workgroup = 256 centroids x 16 trials.  A thread runs ITS centroid (its own row, read
// 8 coordinates at a time) against the 16 trials staged in LDS (same address on every lane: broadcast reads, 16 bytes at a
// time when d is a multiple of 4); each distance is the same j-ordered fmaf chain as a plain loop.  grid = (ceil(n / 256), ceil(B / 16)).
__global__ __launch_bounds__(256) void vjf_rbf_kernel(const float* __restrict__ x, const float* __restrict__ c, const float* __restrict__ logw,
                                                      float* __restrict__ out, int B, int n, int d) {
    extern __shared__ __attribute__((aligned(16))) float s_x[];   // 16 x d
    const int tid = threadIdx.x, k = blockIdx.x * 256 + tid, b0 = blockIdx.y * 16, nb = min(16, B - b0);
    for (int e = tid; e < 16 * d; e += 256) s_x[e] = (e / d) < nb ? x[(size_t)b0 * d + e] : 0.f;
    __syncthreads();
    if (k >= n) return;
    const float* cen = c + (size_t)k * d;
    const bool v4 = (d & 3) == 0 && (((size_t)c) & 15) == 0;
    float d2[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) d2[b] = 0.f;
    for (int c0 = 0; c0 < d; c0 += 8) {
        float cv[8];
        if (v4) {
            const float4 ca = *reinterpret_cast<const float4*>(cen + c0);
            const float4 cb = c0 + 4 < d ? *reinterpret_cast<const float4*>(cen + c0 + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            cv[0] = ca.x; cv[1] = ca.y; cv[2] = ca.z; cv[3] = ca.w; cv[4] = cb.x; cv[5] = cb.y; cv[6] = cb.z; cv[7] = cb.w;
#pragma unroll
            for (int b = 0; b < 16; ++b) {
                const float4 x0 = *reinterpret_cast<const float4*>(&s_x[b * d + c0]);
                float t;
                t = x0.x - cv[0]; d2[b] = fmaf(t, t, d2[b]); t = x0.y - cv[1]; d2[b] = fmaf(t, t, d2[b]);
                t = x0.z - cv[2]; d2[b] = fmaf(t, t, d2[b]); t = x0.w - cv[3]; d2[b] = fmaf(t, t, d2[b]);
                if (c0 + 4 < d) {
                    const float4 x1 = *reinterpret_cast<const float4*>(&s_x[b * d + c0 + 4]);
                    t = x1.x - cv[4]; d2[b] = fmaf(t, t, d2[b]); t = x1.y - cv[5]; d2[b] = fmaf(t, t, d2[b]);
                    t = x1.z - cv[6]; d2[b] = fmaf(t, t, d2[b]); t = x1.w - cv[7]; d2[b] = fmaf(t, t, d2[b]);
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) cv[q] = c0 + q < d ? cen[c0 + q] : 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (c0 + q < d) {
#pragma unroll
                    for (int b = 0; b < 16; ++b) { const float t = s_x[b * d + c0 + q] - cv[q]; d2[b] = fmaf(t, t, d2[b]); }
                }
        }
    }
    const float w = expf(logw[k]);
#pragma unroll
    for (int b = 0; b < 16; ++b)
        if (b < nb) out[(size_t)(b0 + b) * n + k] = expf(-0.5f * d2[b] / (w * w));
}

// Scalar losses: VJF_LOSS_BLOCKS workgroups, fp64 accumulation, fixed order (every workgroup sums a fixed share of the elements,
// the last one to finish adds the partial sums in workgroup order: the result does not depend on timing).
//   mode 0: gaussian_loss (functional.py:32-75)  mode 1: gaussian_entropy (:25-29)  mode 2: poisson (likelihood.py:51-62)
// The partial sums live in a small device-resident table of VJF_LOSS_SLOTS slots handed out in turn by the host: up to that
// many loss calls may be in flight at once (on any streams).
#define VJF_LOSS_BLOCKS 64
#define VJF_LOSS_SLOTS 32
__device__ double vjf_loss_part[VJF_LOSS_SLOTS][VJF_LOSS_BLOCKS];
__device__ unsigned vjf_loss_count[VJF_LOSS_SLOTS];
__global__ __launch_bounds__(256) void vjf_loss_kernel(int mode, const float* m1, const float* lv1, const float* m2, const float* lv2,
                                                       const float* logvar, float* out, int B, int d, int slot) {
    __shared__ double s_p[4];
    __shared__ int s_last;
    double acc = 0.0;
    const size_t N = (size_t)B * d;
    const float lvr = (mode == 0) ? logvar[0] : 0.f;
    const float p = expf(-0.5f * lvr);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)VJF_LOSS_BLOCKS * 256) {
        float t;
        if (mode == 0) {
            const float dsc = m1[i] * p - m2[i] * p;
            t = 0.5f * (dsc * dsc + lvr);
            if (lv1 || lv2) t += 0.5f * expf((lv1 ? lv1[i] : 0.f) + (lv2 ? lv2[i] : 0.f) - lvr);
        } else if (mode == 1) {
            t = 0.5f * m1[i];
        } else {
            const float eta = fminf(m1[i], 10.f);
            t = expf(eta) - m2[i] * eta;
        }
        acc += (double)t;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(&vjf_loss_part[slot][blockIdx.x], ((s_p[0] + s_p[1]) + s_p[2]) + s_p[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (release: the partial sum is visible at agent scope before the count)
        const unsigned done = __hip_atomic_fetch_add(&vjf_loss_count[slot], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        s_last = done == VJF_LOSS_BLOCKS - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 64) {                                    // the last workgroup: 64 partial sums, fixed xor tree
        double t = __hip_atomic_load(&vjf_loss_part[slot][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (threadIdx.x == 0) { out[0] = (float)(t / (double)B); __hip_atomic_store(&vjf_loss_count[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
}

// p[0..n) *= f   (test hook of the sharded route: a one-rank communicator standing in for f identical ranks)
__global__ void vjf_scale_kernel(float* p, float f, int n) {
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) p[e] *= f;
}
