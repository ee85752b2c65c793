// Fused Adam over the flat Gaussian parameter buffer.
//
// reference: the per-frame step ends with torch.optim.Adam.step() over five or six parameter groups
// (train/optim.py:11-37, train/iteration.py:58-60) — with the default (foreach) implementation a dozen
// elementwise kernels per step, each streaming the whole state.  Here: ONE pass, 16 B per lane per array,
// 28 B of HBM traffic per parameter (read p, g, m, v; write p, m, v) — the HBM roofline of the update.
// Arithmetic follows torch.optim.Adam (betas, eps outside the sqrt, bias correction as step_size / denom);
// this file is built without FMA contraction so that the update matches torch's to rounding.
#include "fr_common.hpp"

namespace fr {

struct AdamArgs {
    int n_seg;
    unsigned long long seg_end[FR_ADAM_MAX_SEGMENTS];
    float seg_lr[FR_ADAM_MAX_SEGMENTS];
    unsigned seg_period[FR_ADAM_MAX_SEGMENTS], seg_split[FR_ADAM_MAX_SEGMENTS];
    float seg_lr2[FR_ADAM_MAX_SEGMENTS];
    float beta1, beta2, omb1, omb2, eps, grad_scale;  // omb = 1 - beta, rounded from double
};

// state = {step, 1 - beta1^step, 1 - beta2^step, -}: advanced on the device so that the host passes nothing that
// changes from step to step (graph replay).  The bias corrections c_t = 1 - beta^t are carried by the recurrence
// c_{t+1} = (1 - beta) + beta c_t (all terms positive: no cancellation; 1 - 0.999f alone is off by 1.3e-5).
__global__ void k_adam_advance(float* state, float beta1, float omb1, float beta2, float omb2)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        state[0] += 1.0f;
        state[1] = omb1 + beta1 * state[1];
        state[2] = omb2 + beta2 * state[2];
    }
}

__device__ __forceinline__ float adam_one(float& p, float g, float& m, float& v, float lr_over_bc1, float inv_sqrt_bc2,
                                          const AdamArgs& a)
{
    g *= a.grad_scale;
    m = a.beta1 * m + a.omb1 * g;
    v = a.beta2 * v + a.omb2 * (g * g);
    const float denom = sqrtf(v) * inv_sqrt_bc2 + a.eps;
    p -= lr_over_bc1 * (m / denom);
    return p;
}

__global__ void __launch_bounds__(256) k_adam(AdamArgs a, float4* __restrict__ param, const float4* __restrict__ grad,
                                              float4* __restrict__ exp_avg, float4* __restrict__ exp_avg_sq,
                                              unsigned long long n, const float* __restrict__ state)
{
    const float bc1 = state[1], bc2 = state[2];
    const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    const unsigned long long n4 = n / 4, stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (n + 3) / 4; i += stride) {
        const unsigned long long e0 = 4 * i;
        float lr[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {  // a quad may straddle a segment boundary
            int sg = 0;
            for (int q = 0; q + 1 < a.n_seg; q++) sg += (e0 + k >= a.seg_end[q]) ? 1 : 0;
            float l = a.seg_lr[sg];
            if (a.seg_period[sg]) {
                const unsigned long long rel = e0 + k - (sg ? a.seg_end[sg - 1] : 0ull);
                if ((unsigned)(rel % a.seg_period[sg]) >= a.seg_split[sg]) l = a.seg_lr2[sg];
            }
            lr[k] = l / bc1;
        }
        if (i < n4) {
            float4 p = param[i], m = exp_avg[i], v = exp_avg_sq[i];
            const float4 g = grad[i];
            adam_one(p.x, g.x, m.x, v.x, lr[0], inv_sqrt_bc2, a);
            adam_one(p.y, g.y, m.y, v.y, lr[1], inv_sqrt_bc2, a);
            adam_one(p.z, g.z, m.z, v.z, lr[2], inv_sqrt_bc2, a);
            adam_one(p.w, g.w, m.w, v.w, lr[3], inv_sqrt_bc2, a);
            param[i] = p, exp_avg[i] = m, exp_avg_sq[i] = v;
        } else {  // tail of a buffer whose length is not a multiple of 4
            float* ps = reinterpret_cast<float*>(param);
            const float* gs = reinterpret_cast<const float*>(grad);
            float* ms = reinterpret_cast<float*>(exp_avg);
            float* vs = reinterpret_cast<float*>(exp_avg_sq);
            for (int k = 0; k < 4 && e0 + k < n; k++) adam_one(ps[e0 + k], gs[e0 + k], ms[e0 + k], vs[e0 + k], lr[k], inv_sqrt_bc2, a);
        }
    }
}

int launch_adam(const fr_adam_config& cfg, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                unsigned long long n, float* state, hipStream_t s)
{
    if (n == 0) return FR_OK;
    AdamArgs a;
    a.n_seg = cfg.n_segments;
    for (int i = 0; i < FR_ADAM_MAX_SEGMENTS; i++) {
        a.seg_end[i] = i < cfg.n_segments ? cfg.segment_end[i] : 0ull;
        a.seg_lr[i] = i < cfg.n_segments ? cfg.segment_lr[i] : 0.f;
        a.seg_period[i] = i < cfg.n_segments ? cfg.segment_period[i] : 0u;
        a.seg_split[i] = i < cfg.n_segments ? cfg.segment_split[i] : 0u;
        a.seg_lr2[i] = i < cfg.n_segments ? cfg.segment_lr2[i] : 0.f;
    }
    a.beta1 = (float)cfg.beta1, a.beta2 = (float)cfg.beta2, a.eps = (float)cfg.eps, a.grad_scale = cfg.grad_scale;
    a.omb1 = (float)(1.0 - cfg.beta1), a.omb2 = (float)(1.0 - cfg.beta2);
    hipLaunchKernelGGL(k_adam_advance, dim3(1), dim3(64), 0, s, state, a.beta1, a.omb1, a.beta2, a.omb2);
    const unsigned long long quads = (n + 3) / 4;
    unsigned long long blocks = (quads + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, s, a, reinterpret_cast<float4*>(param),
                       reinterpret_cast<const float4*>(grad), reinterpret_cast<float4*>(exp_avg),
                       reinterpret_cast<float4*>(exp_avg_sq), n, state);
    FR_HIP(hipGetLastError());
    return FR_OK;
}

}  // namespace fr
