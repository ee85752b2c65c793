// microbenchmark: VALU / LDS issue cost on gfx950 by instruction kind and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -w tools/diag/micro_issue.hip -o gpurun_exp/micro_issue
// Each wave runs LOOPS iterations of a block of 32 independent instructions of one kind (8 accumulator chains, so the
// dependent-issue latency is hidden inside one wave); cycles per instruction per SIMD = elapsed shader cycles of the
// slowest wave x (1 / (32 * LOOPS * waves per SIMD)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define LOOPS 2000
#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define R4x8(X) R8(X) R8(X) R8(X) R8(X)
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, unsigned long long* cyc, float seed, unsigned long long* rt)
{
    __shared__ float4 lds[64 * 3 * 4];
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
    for (int i = 0; i < 8; i++) a[i] = seed + i + lane, b[i] = seed * 0.5f + i;
    typedef float v2f __attribute__((ext_vector_type(2)));
    v2f p[8];
    for (int i = 0; i < 8; i++) p[i] = v2f{a[i], b[i]};
    unsigned long long m[4] = {0x123456789abcdefull + lane, 77ull << lane, 5ull, 9ull};
    for (int i = threadIdx.x; i < 64 * 3 * 4; i += 256) lds[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const int rec = (lane * 37 + 11) & 63;   // divergent record index, as in the blend walks
    const float4* L = lds + (threadIdx.x >> 6) * 192;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < LOOPS; it++) {
        if (KIND == 0) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            R4x8(X)
#undef X
        } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
            R4x8(X)
#undef X
        } else if (KIND == 2) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            R4x8(X)
#undef X
        } else if (KIND == 3) {
#define X(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            R4x8(X)
#undef X
        } else if (KIND == 4) {
#define X(i) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(m[i & 3]));
            R4x8(X)
#undef X
        } else if (KIND == 5) {
#define X(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : );
            R4x8(X)
#undef X
        } else if (KIND == 6) {
#define X(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            R4x8(X)
#undef X
        } else if (KIND == 7) {
#define X(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            R4x8(X)
#undef X
        } else if (KIND == 8) {
#define X(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            R4x8(X)
#undef X
        } else if (KIND == 9) {   // divergent 16-byte LDS reads (48-byte records), 8 in flight
            float4 q[8];
#define X(i) q[i] = L[((rec + i * 5 + it) & 63) * 3 + (i % 3)];
            R4x8(X)
#undef X
            for (int i = 0; i < 8; i++) a[i] += q[i].x;
        } else if (KIND == 10) {  // v_ffbh + 64-bit bit clear: the bit walk of the blend loops
#define X(i) { const int j = 63 - __builtin_clzll(m[i & 3] | 1ull); m[i & 3] &= ~(1ull << j); asm volatile("" : "+v"(m[i & 3])); }
            R8(X)
#undef X
        } else if (KIND == 11) {
#define X(i) asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            R4x8(X)
#undef X
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
    s += (float)(m[0] + m[1] + m[2] + m[3]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) { cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0; rt[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = r0; rt[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1; }
}
template <int KIND>
static void run(const char* name, int per_loop)
{
    float* out; unsigned long long *cyc, *rt;
    hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&cyc, 256 * 8 * 4 * 8); hipMalloc(&rt, 256 * 8 * 4 * 16);
    for (int wg_per_cu : {1, 2, 3, 4, 6, 8}) {
        const int grid = 256 * wg_per_cu;
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0f, rt);
        hipDeviceSynchronize();
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(grid), dim3(256), 0, 0, out, cyc, 1.0f, rt);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        std::vector<unsigned long long> h(grid * 4);
        hipMemcpy(h.data(), cyc, grid * 4 * 8, hipMemcpyDeviceToHost);
        double mean = 0; unsigned long long mx = 0;
        for (auto c : h) { mean += c; mx = c > mx ? c : mx; }
        mean /= h.size();
        const double n = (double)per_loop * LOOPS;
        std::vector<unsigned long long> hr(grid * 8);
        hipMemcpy(hr.data(), rt, grid * 8 * 8, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0; double dur = 0;
        for (int i = 0; i < grid * 4; i++) { lo = hr[2 * i] < lo ? hr[2 * i] : lo; hi = hr[2 * i + 1] > hi ? hr[2 * i + 1] : hi; dur += (double)(hr[2 * i + 1] - hr[2 * i]); }
        dur /= grid * 4;   // 100 MHz ticks
        printf("%-30s %d WG/CU: %6.2f ticks/instr/wave; wave %.1f us of span %.1f us (wall %.1f); clock %.2f GHz; %.2f ns/instr/SIMD\n", name,
               wg_per_cu, mean / n, dur / 100, (hi - lo) / 100.0, ms * 1000, mean / (dur * 10), (hi - lo) * 10.0 / (n * wg_per_cu));
    }
    hipFree(out); hipFree(cyc); hipFree(rt);
}
int main()
{
    run<0>("v_fma_f32", 32);
    run<1>("v_pk_fma_f32", 32);
    run<6>("v_mul_f32", 32);
    run<2>("v_exp_f32", 32);
    run<3>("v_rcp_f32", 32);
    run<4>("v_lshlrev_b64", 32);
    run<5>("v_cndmask_b32 (vcc)", 32);
    run<8>("v_cmp + v_cndmask pair", 64);
    run<7>("v_add_f32_dpp quad_perm", 32);
    run<11>("v_mbcnt_lo", 32);
    run<10>("clz64 + clear bit (per step)", 8);
    run<9>("ds_read_b128 divergent (+add)", 32);
    return 0;
}
