// microbenchmark: returning u32 atomics to per-XCD private counter arrays, counter stride S words
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k(const unsigned* idx, unsigned n, unsigned* ctr, unsigned tpad, unsigned stride, unsigned* out, int use_xcc)
{
    const unsigned xcc = use_xcc ? (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u) : 0u;
    unsigned acc = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
#ifdef NORET
        atomicAdd(&ctr[((size_t)xcc * tpad + idx[i]) * stride], 1u);
#else
        acc += atomicAdd(&ctr[((size_t)xcc * tpad + idx[i]) * stride], 1u);
#endif
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main(int argc, char** argv)
{
    // argv[1]: number of distinct counters the atomics fall on (default 16384; BASELINE config 2 has 1 248 non-empty tiles)
    const unsigned n = 2u << 20, T = argc > 1 ? (unsigned)atoi(argv[1]) : 16384;
    std::vector<unsigned> h(n);
    srand(1);
#ifdef CLUSTERED
    for (unsigned i = 0; i < n; i += 64) { const unsigned b = (rand() % (T / 64)) * 64; for (unsigned l = 0; l < 64; l++) h[i + l] = b + l; }
#else
    for (auto& x : h) x = rand() % T;
#endif
    unsigned *idx, *ctr, *out;
    hipMalloc(&idx, n * 4); hipMalloc(&ctr, (size_t)8 * T * 32 * 4); hipMalloc(&out, 2048 * 256 * 4);
    hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int use_xcc = 0; use_xcc < 2; use_xcc++)
        for (unsigned stride : {1u, 2u, 4u, 8u, 16u, 32u}) {
            hipMemset(ctr, 0, (size_t)8 * T * 32 * 4);
            hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, idx, n, ctr, T, stride, out, use_xcc);
            hipEventRecord(a);
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, idx, n, ctr, T, stride, out, use_xcc);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("counters=%u xcc_private=%d stride=%2u words: %.1f us per 2M atomics -> %.1f atomics/ns\n", T, use_xcc, stride, ms * 1000 / 5, n / (ms * 1e6 / 5));
        }
    return 0;
}
