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

// "Which workgroup of this launch finishes last?" with TWO levels of wrapping counters: atomics on one address are
// served one at a time (about 10 ns each on MI355X: a single counter costs 8 us for 768 workgroups, 75 us for 8192), so
// a workgroup counts itself into one of kDoneGroups counters and only the last of each group into the final one.
// The counters sit in separate 128-byte lines: the L2 serves atomics on one LINE one at a time, too.
// `c`: (kDoneGroups + 1) x kDoneStride zeroed words, left zeroed.  Call from one thread per workgroup.
constexpr unsigned kDoneGroups = 16, kDoneStride = 32;
__device__ __forceinline__ bool last_workgroup(unsigned* c)
{
    const unsigned groups = gridDim.x < kDoneGroups ? gridDim.x : kDoneGroups;
    const unsigned g = blockIdx.x % groups;
    const unsigned members = (gridDim.x - g + groups - 1u) / groups;   // workgroups b with b % groups == g
    if (atomicInc(c + g * kDoneStride, members - 1u) != members - 1u) return false;  // (wraps back to 0)
    return atomicInc(c + kDoneGroups * kDoneStride, groups - 1u) == groups - 1u;
}

struct AdamArgs {
    int n_seg;
    unsigned long long seg_end[FR_ADAM_MAX_SEGMENTS];
    float seg_lr[FR_ADAM_MAX_SEGMENTS];
    unsigned seg_period[FR_ADAM_MAX_SEGMENTS], seg_split[FR_ADAM_MAX_SEGMENTS];
    float seg_lr2[FR_ADAM_MAX_SEGMENTS];
    float beta1, beta2, omb1, omb2, eps, grad_scale;  // omb = 1 - beta, rounded from double
    int n_skip;                                        // fr_adam_config::skip: any non-zero word -> the step does nothing
    const float* skip[FR_ADAM_MAX_GRADS];
};

// state = {step, 1 - beta1^step, 1 - beta2^step, -, ..., done-counters from word 32}: advanced on the device so that the host passes
// nothing that changes from step to step (graph replay).  The bias corrections c_t = 1 - beta^t are carried by the recurrence
// c_{t+1} = (1 - beta) + beta c_t (all terms positive: no cancellation; 1 - 0.999f alone is off by 1.3e-5).  Every
// workgroup of k_adam derives this step's corrections from the OLD state; the workgroup that finishes last
// (last_workgroup) stores the new one — no launch of its own for three floats.
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

// the gradient of the step = the SUM of up to FR_ADAM_MAX_GRADS buffers (the views of a batch rendered in flight
// together, each into its own buffer; grad_scale turns the sum into the mean): no pass of its own to add them up
struct AdamGrads {
    int n;
    const float* g[FR_ADAM_MAX_GRADS];
};

__device__ __forceinline__ float4 adam_grad4(const AdamGrads& gs, unsigned long long i)
{
    float4 r = reinterpret_cast<const float4*>(gs.g[0])[i];
#pragma unroll
    for (int k = 1; k < FR_ADAM_MAX_GRADS; k++) {
        if (k < gs.n) {
            const float4 t = reinterpret_cast<const float4*>(gs.g[k])[i];
            r.x += t.x, r.y += t.y, r.z += t.z, r.w += t.w;
        }
    }
    return r;
}

__global__ void __launch_bounds__(256) k_adam(AdamArgs a, float4* __restrict__ param, AdamGrads grads,
                                              float4* __restrict__ exp_avg, float4* __restrict__ exp_avg_sq,
                                              unsigned long long n, float* state)
{
    // a frame that feeds this step overflowed its binning capacity inside a replayed graph and back-propagated zeros
    // (fr_aux::overflow_out): the whole step is skipped — every workgroup takes the same decision from the same words
    for (int k = 0; k < a.n_skip; k++)
        if (a.skip[k][0] != 0.0f) return;
    const float step_new = state[0] + 1.0f;
    const float bc1 = a.omb1 + a.beta1 * state[1], bc2 = a.omb2 + a.beta2 * state[2];
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
            const float4 g = adam_grad4(grads, i);
            adam_one(p.x, g.x, m.x, v.x, lr[0], inv_sqrt_bc2, a);
            adam_one(p.y, g.y, m.y, v.y, lr[1], inv_sqrt_bc2, a);
            adam_one(p.z, g.z, m.z, v.z, lr[2], inv_sqrt_bc2, a);
            adam_one(p.w, g.w, m.w, v.w, lr[3], inv_sqrt_bc2, a);
            param[i] = p, exp_avg[i] = m, exp_avg_sq[i] = v;
        } else {  // tail of a buffer whose length is not a multiple of 4
            float* ps = reinterpret_cast<float*>(param);
            float* ms = reinterpret_cast<float*>(exp_avg);
            float* vs = reinterpret_cast<float*>(exp_avg_sq);
            for (int k = 0; k < 4 && e0 + k < n; k++) {
                float g1 = grads.g[0][e0 + k];
                for (int q = 1; q < grads.n; q++) g1 += grads.g[q][e0 + k];
                adam_one(ps[e0 + k], g1, ms[e0 + k], vs[e0 + k], lr[k], inv_sqrt_bc2, a);
            }
        }
    }
    // every thread of this workgroup has read the old state (above) before the barrier; the last workgroup to get here
    // knows that all have
    __syncthreads();
    if (threadIdx.x == 0) {
        if (last_workgroup(reinterpret_cast<unsigned*>(state + 32))) state[0] = step_new, state[1] = bc1, state[2] = bc2;
    }
}

int launch_adam(const fr_adam_config& cfg, float* param, const float* const* grad_bufs, int n_grads, float* exp_avg,
                float* exp_avg_sq, unsigned long long n, float* state, hipStream_t s)
{
    if (n == 0) return FR_OK;
    AdamGrads grads;
    grads.n = n_grads;
    for (int k = 0; k < FR_ADAM_MAX_GRADS; k++) grads.g[k] = k < n_grads ? grad_bufs[k] : grad_bufs[0];
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
    a.n_skip = cfg.n_skip < 0 ? 0 : (cfg.n_skip > FR_ADAM_MAX_GRADS ? FR_ADAM_MAX_GRADS : cfg.n_skip);
    for (int k = 0; k < FR_ADAM_MAX_GRADS; k++) a.skip[k] = k < a.n_skip ? cfg.skip[k] : nullptr;
    const unsigned long long quads = (n + 3) / 4;
    unsigned long long blocks = (quads + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, s, a, reinterpret_cast<float4*>(param), grads,
                       reinterpret_cast<float4*>(exp_avg),
                       reinterpret_cast<float4*>(exp_avg_sq), n, state);
    FR_HIP(hipGetLastError());
    return FR_OK;
}

// ---------------------------------------------------------------- L1 image loss and its gradient, one launch
// reference: nn.L1Loss(reduction='mean') on the rendered image (model/loss.py:92) followed by loss.backward() — in
// PyTorch eight launch-bound elementwise / reduction kernels (sub, abs, mean, fill, sign, mul, ...: 41 us of a 205 us
// optimisation step at 512 x 512).  Here: grad = sign(img - gt) / n and per-workgroup partial sums of |img - gt| in one
// pass; the workgroup that finishes last adds the partials up in index order (deterministic) and stores the loss.
constexpr unsigned kL1MaxBlocks = 1024;

struct L1View {   // one image of a (possibly batched) launch
    const float* img;
    const float* gt;
    float* grad;
    float* partial;
    unsigned* counter;
    float* loss;
};

__device__ __forceinline__ void l1_loss_grad_body(const L1View& v, unsigned long long n, float inv_n)
{
    const float* __restrict__ img = v.img;
    const float* __restrict__ gt = v.gt;
    float* __restrict__ grad = v.grad;
    float* const partial = v.partial;
    unsigned* const counter = v.counter;
    float* __restrict__ loss = v.loss;
    __shared__ float s_red[4];
    __shared__ bool s_last;
    const unsigned long long n4 = n / 4, stride = (unsigned long long)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(img)[i], b = reinterpret_cast<const float4*>(gt)[i];
        const float d[4] = {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w};
        float g[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            acc += fabsf(d[k]);
            g[k] = d[k] > 0.f ? inv_n : (d[k] < 0.f ? -inv_n : 0.f);   // torch.sign: 0 at 0
        }
        if (grad) reinterpret_cast<float4*>(grad)[i] = make_float4(g[0], g[1], g[2], g[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - 4 * n4)) {   // tail of a length that is not a multiple of 4
        const unsigned long long e = 4 * n4 + threadIdx.x;
        const float d = img[e] - gt[e];
        acc += fabsf(d);
        if (grad) grad[e] = d > 0.f ? inv_n : (d < 0.f ? -inv_n : 0.f);
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        // write-through store, wait for it, then count this workgroup in (see k_unit_blend_chained on the hand-off)
        __hip_atomic_store(partial + blockIdx.x, (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_last = last_workgroup(counter);
    }
    __syncthreads();
    if (!s_last) return;
    float t = 0.f;
    for (unsigned b = threadIdx.x; b < gridDim.x; b += blockDim.x)
        t += __hip_atomic_load(partial + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) *loss = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * inv_n;
}

__global__ void __launch_bounds__(256) k_l1_loss_grad(L1View v, unsigned long long n, float inv_n) { l1_loss_grad_body(v, n, inv_n); }
// the images of the frames of a batch (same size), one workspace each: grid (x, images)
__global__ void __launch_bounds__(256) k_l1_loss_grad_batch(BatchOf<L1View> b, unsigned long long n, float inv_n)
{
    l1_loss_grad_body(b.v[blockIdx.y], n, inv_n);
}

int launch_l1_loss_grad(unsigned long long n, const float* img, const float* gt, float* grad, float* loss, void* workspace,
                        hipStream_t s)
{
    if (n == 0) return FR_OK;
    unsigned long long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > kL1MaxBlocks ? kL1MaxBlocks : blocks);
    unsigned* counter = static_cast<unsigned*>(workspace);
    float* partial = reinterpret_cast<float*>(counter + (kDoneGroups + 1) * kDoneStride);
    hipLaunchKernelGGL(k_l1_loss_grad, dim3((unsigned)blocks), dim3(256), 0, s, L1View{img, gt, grad, partial, counter, loss}, n,
                       (float)(1.0 / (double)n));
    FR_HIP(hipGetLastError());
    return FR_OK;
}

int launch_l1_loss_grad_batch(int n_images, unsigned long long n, const float* const* img, const float* const* gt, float* const* grad,
                              float* const* loss, void* const* workspace, hipStream_t s)
{
    if (n == 0 || n_images <= 0) return FR_OK;
    unsigned long long blocks = (n / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > kL1MaxBlocks ? kL1MaxBlocks : blocks);
    BatchOf<L1View> b;
    for (int k = 0; k < kMaxBatch; k++) {
        const int j = k < n_images ? k : 0;   // (unused entries: never indexed, blockIdx.y < n_images)
        unsigned* counter = static_cast<unsigned*>(workspace[j]);
        b.v[k] = L1View{img[j], gt[j], grad ? grad[j] : nullptr, reinterpret_cast<float*>(counter + (kDoneGroups + 1) * kDoneStride),
                        counter, loss[j]};
    }
    hipLaunchKernelGGL(k_l1_loss_grad_batch, dim3((unsigned)blocks, (unsigned)n_images), dim3(256), 0, s, b, n, (float)(1.0 / (double)n));
    FR_HIP(hipGetLastError());
    return FR_OK;
}

// ---------------------------------------------------------------- several small device-to-device copies, one launch
// The per-frame inputs of a captured step (camera block, posed vertices, target image) are copied into the buffers the
// graph was captured with: as separate copies each is a launch-bound 5 us dispatch.
struct CopyArgs {
    int n;
    float* dst[FR_COPY_MAX_SEGMENTS];
    const float* src[FR_COPY_MAX_SEGMENTS];
    unsigned long long count[FR_COPY_MAX_SEGMENTS];   // floats
};

__global__ void __launch_bounds__(256) k_multi_copy(CopyArgs a)
{
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (int sg = 0; sg < a.n; sg++) {
        const unsigned long long n = a.count[sg];
        const bool wide = ((reinterpret_cast<uintptr_t>(a.dst[sg]) | reinterpret_cast<uintptr_t>(a.src[sg])) & 15) == 0;
        const unsigned long long n4 = wide ? n / 4 : 0;
        for (unsigned long long i = t0; i < n4; i += stride)
            reinterpret_cast<float4*>(a.dst[sg])[i] = reinterpret_cast<const float4*>(a.src[sg])[i];
        for (unsigned long long e = 4 * n4 + t0; e < n; e += stride) a.dst[sg][e] = a.src[sg][e];
    }
}

int launch_multi_copy(int n_seg, float* const* dst, const float* const* src, const unsigned long long* count, hipStream_t s)
{
    CopyArgs a;
    a.n = n_seg;
    unsigned long long most = 0;
    for (int i = 0; i < FR_COPY_MAX_SEGMENTS; i++) {
        a.dst[i] = i < n_seg ? dst[i] : nullptr, a.src[i] = i < n_seg ? src[i] : nullptr, a.count[i] = i < n_seg ? count[i] : 0ull;
        most = a.count[i] > most ? a.count[i] : most;
    }
    if (most == 0) return FR_OK;
    unsigned long long blocks = (most / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(k_multi_copy, dim3((unsigned)blocks), dim3(256), 0, s, a);
    FR_HIP(hipGetLastError());
    return FR_OK;
}

// ---------------------------------------------------------------- scaled sum of up to four equally long arrays
// dst = scale * (src[0] + ... + src[n-1]): the mean of the gradient buffers of the views a rank rendered in flight
// together, written into the exchange buffer of the all-reduce in one pass (three PyTorch kernels otherwise: 142 MB of
// traffic instead of 94 at 23.6 MB per buffer, on a GPU that is busy rendering the next step's frames).  dst may be one
// of the sources (every element is read before it is written, by the same thread): dst += ... for local accumulation.
struct SumArgs {
    int n;
    const float* src[FR_ADAM_MAX_GRADS];
};

__global__ void __launch_bounds__(256) k_scaled_sum(SumArgs a, float* dst, unsigned long long count, float scale)
{
    const unsigned long long n4 = count / 4, stride = (unsigned long long)gridDim.x * blockDim.x;
    const unsigned long long t0 = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (unsigned long long i = t0; i < n4; i += stride) {
        float4 r = reinterpret_cast<const float4*>(a.src[0])[i];
#pragma unroll
        for (int k = 1; k < FR_ADAM_MAX_GRADS; k++) {
            if (k < a.n) {
                const float4 t = reinterpret_cast<const float4*>(a.src[k])[i];
                r.x += t.x, r.y += t.y, r.z += t.z, r.w += t.w;
            }
        }
        reinterpret_cast<float4*>(dst)[i] = make_float4(r.x * scale, r.y * scale, r.z * scale, r.w * scale);
    }
    for (unsigned long long e = 4 * n4 + t0; e < count; e += stride) {
        float r = a.src[0][e];
        for (int k = 1; k < a.n; k++) r += a.src[k][e];
        dst[e] = r * scale;
    }
}

int launch_scaled_sum(int n_src, const float* const* src, float* dst, unsigned long long count, float scale, hipStream_t s)
{
    if (count == 0) return FR_OK;
    SumArgs a;
    a.n = n_src;
    for (int k = 0; k < FR_ADAM_MAX_GRADS; k++) a.src[k] = k < n_src ? src[k] : src[0];
    unsigned long long blocks = (count / 4 + 255) / 256;
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    hipLaunchKernelGGL(k_scaled_sum, dim3((unsigned)blocks), dim3(256), 0, s, a, dst, count, scale);
    FR_HIP(hipGetLastError());
    return FR_OK;
}

}  // namespace fr
