// microbenchmark: how fast do the waves of one dispatch START on gfx950, by workgroup size, LDS per workgroup and
// register budget?    hipcc --offload-arch=gfx950 -O3 -w tools/diag/micro_ramp.hip -o gpurun_exp/micro_ramp
// Every wave stamps s_memrealtime (100 MHz) at entry, then spins for ~15 us so that no slot is reused; the entry times'
// spread is the dispatch ramp of a grid that fits the chip at once.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
template <int WAVES, int LDS_BYTES, int VGPRS, int LOADS = 0, int DELAY = 0>
__global__ void __launch_bounds__(64 * WAVES) k(unsigned long long* rt, float* out, const float4* big = nullptr)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // second stamp: behind the wave's first use of a kernel ARGUMENT (the kernarg segment's first touch by this CU)
    unsigned long long t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : "s"(out) : "memory");
    __shared__ char lds[LDS_BYTES > 0 ? LDS_BYTES : 4];
    float a[VGPRS];
    for (int i = 0; i < VGPRS; i++) a[i] = threadIdx.x + i;
    if (DELAY) while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)DELAY) __builtin_amdgcn_s_sleep(4);
    if (LOADS) {   // every wave asks for LOADS KB at once, like a blend unit's first loads
        const float4* src = big + ((size_t)(blockIdx.x * WAVES + (threadIdx.x >> 6)) * LOADS) * 64 + (threadIdx.x & 63);
        float4 acc = make_float4(0, 0, 0, 0);
        for (int i = 0; i < LOADS; i++) { const float4 q = src[i * 64]; acc.x += q.x, acc.y += q.y, acc.z += q.z, acc.w += q.w; }
        a[0] += acc.x + acc.y + acc.z + acc.w;
        asm volatile("" ::"v"(a[0]));
        t1 = __builtin_amdgcn_s_memrealtime();   // (with loads: the second stamp is 'my data is here')
    }
    if (LDS_BYTES > 0) lds[threadIdx.x] = (char)threadIdx.x;
    while (__builtin_amdgcn_s_memrealtime() - t0 < 1500ull) {
        for (int i = 0; i < VGPRS; i++) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(a[i]));
    }
    float s = 0.f;
    for (int i = 0; i < VGPRS; i++) s += a[i];
    if (s == 12345.f) out[0] = s + lds[threadIdx.x & 3];
    if ((threadIdx.x & 63) == 0) rt[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t0, rt[8192 + blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1;
}
template <int WAVES, int LDS_BYTES, int VGPRS, int LOADS = 0, int DELAY = 0>
void run(int total_waves, unsigned long long* d_rt, float* d_out, const float4* big = nullptr)
{
    const int wgs = total_waves / WAVES;
    std::vector<unsigned long long> h(total_waves), h1(total_waves);
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL((k<WAVES, LDS_BYTES, VGPRS, LOADS, DELAY>), dim3(wgs), dim3(64 * WAVES), 0, 0, d_rt, d_out, big);
        hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d_rt, total_waves * 8, hipMemcpyDeviceToHost);
    hipMemcpy(h1.data(), d_rt + 8192, total_waves * 8, hipMemcpyDeviceToHost);
    double dmax = 0, dsum = 0;
    for (int i = 0; i < total_waves; i++) dmax = std::max(dmax, (h1[i] - h[i]) / 100.0), dsum += (h1[i] - h[i]) / 100.0;
    printf("   entry by wave index:");
    { unsigned long long mn = *std::min_element(h.begin(), h.end()); for (int i = 0; i < total_waves; i += total_waves / 16) printf(" %d:%.2f", i, (h[i] - mn) / 100.0); printf("\n"); }
    std::sort(h.begin(), h.end());
    std::sort(h1.begin(), h1.end());
    printf("   [delay %d x10ns] ", DELAY);
    printf("second stamp - entry: mean %.2f max %.2f us; last wave past its kernargs at %.2f us\n", dsum / total_waves, dmax, (h1.back() - h[0]) / 100.0);
    auto us = [&](double q) { return (h[(size_t)(q * (total_waves - 1))] - h[0]) / 100.0; };
    printf("waves/WG %2d  LDS/WG %6d B  regs ~%3d loads %d KB %5d waves: entry p50 %5.2f  p90 %5.2f  p99 %5.2f  max %5.2f us\n", WAVES, LDS_BYTES,
           VGPRS, LOADS, total_waves, us(0.5), us(0.9), us(0.99), us(1.0));
}
int main()
{
    unsigned long long* d_rt;
    float* d_out;
    hipMalloc(&d_rt, 2 * 8192 * 8);
    hipMalloc(&d_out, 64);
    const int N = 3904;   // the unit count of the benchmark frame, below the 4096 slots of 4 waves per SIMD
    float4* big;
    hipMalloc(&big, (size_t)8192 * 8 * 1024);
    hipMemset(big, 0, (size_t)8192 * 8 * 1024);
    run<4, 38912, 120>(N, d_rt, d_out);
    run<4, 38912, 120, 6>(N, d_rt, d_out, big);
    run<4, 12288, 80, 6>(N, d_rt, d_out, big);      // the forward blend's footprint: the chip two thirds full
    run<4, 12288, 120, 6>(N, d_rt, d_out, big);
    run<4, 38912, 80, 6>(N, d_rt, d_out, big);
    run<4, 30000, 120, 6>(N, d_rt, d_out, big);     // 5 workgroups per CU would fit the LDS; registers allow 4
    run<4, 38912, 120, 6>(3072, d_rt, d_out, big);  // three quarters of the slots
    run<4, 38912, 120, 6>(2048, d_rt, d_out, big);
    run<4, 38912, 120, 6, 50>(N, d_rt, d_out, big);
    run<4, 38912, 120, 6, 100>(N, d_rt, d_out, big);
    run<4, 38912, 120, 6, 150>(N, d_rt, d_out, big);
    run<4, 38912, 120, 6, 200>(N, d_rt, d_out, big);
    run<4, 0, 16>(N, d_rt, d_out);
    run<4, 0, 100>(N, d_rt, d_out);
    run<4, 4096, 16>(N, d_rt, d_out);
    run<4, 39000, 16>(N, d_rt, d_out);
    run<4, 39000, 100>(N, d_rt, d_out);
    run<1, 0, 16>(N, d_rt, d_out);
    run<1, 9700, 100>(N, d_rt, d_out);
    run<2, 19400, 100>(N, d_rt, d_out);
    run<8, 78000, 100>(N, d_rt, d_out);
    run<16, 0, 16>(N, d_rt, d_out);
    run<4, 0, 16>(1024, d_rt, d_out);
    run<4, 0, 16>(2048, d_rt, d_out);
    return 0;
}
