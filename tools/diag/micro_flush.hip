// microbenchmark: the blend backward's gradient flush — non-returning f32 atomic adds of 9-float rows into
// [P][16]-float accumulators (one 64-byte line per Gaussian), R rows with random Gaussian ids.
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/diag/micro_flush.hip -o gpurun_exp/micro_flush
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
// MODE 0: 7 records x 9 components per instruction (the kernel's flush)   MODE 1: lane = record, 9 instructions
// MODE 2: 4 records x 16 lanes (whole lines)                              MODE 3: plain stores, 7 x 9 layout (no atomics)
template <int MODE>
__global__ void k(const unsigned* __restrict__ ids, unsigned R, float* __restrict__ acc, const float* __restrict__ val)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
    for (unsigned base = wave * 64; base < R; base += nw * 64) {
        const float v = val[lane];
        if (MODE == 0 || MODE >= 3) {
            const int r = lane / 9, c = lane - r * 9;
            for (int r0 = 0; r0 < 64; r0 += 7) {
                const unsigned j = base + r0 + r;
                if (lane < 63 && r0 + r < 64 && j < R) {
                    float* p = acc + (size_t)ids[j] * 16 + c;
                    if (MODE == 0) unsafeAtomicAdd(p, v);
                    else if (MODE == 4) asm volatile("global_atomic_add_f32 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
                    else if (MODE == 5) asm volatile("global_atomic_add_f32 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
                    else if (MODE == 6) asm volatile("global_atomic_add_f32 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
                    else *p = v;
                }
            }
        } else if (MODE == 1) {
            const unsigned j = base + lane;
            if (j < R) {
                float* p = acc + (size_t)ids[j] * 16;
#pragma unroll
                for (int c = 0; c < 9; c++) unsafeAtomicAdd(p + c, v);
            }
        } else {
            const int r = lane >> 4, c = lane & 15;
            for (int r0 = 0; r0 < 64; r0 += 4) {
                const unsigned j = base + r0 + r;
                if (j < R && c < 9) unsafeAtomicAdd(acc + (size_t)ids[j] * 16 + c, v);
            }
        }
    }
}
template <int MODE>
static void run(const char* name, const unsigned* ids, unsigned R, float* acc, const float* val, int grid)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, ids, R, acc, val);
    hipEventRecord(a);
    for (int r = 0; r < 20; r++) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, ids, R, acc, val);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-44s grid %4d: %6.2f us per %u rows -> %.1f rows/ns\n", name, grid, ms * 1000 / 20, R, R / (ms * 1e6 / 20));
}
int main(int argc, char** argv)
{
    const unsigned P = 100000;
    for (unsigned R : {250000u, 1000000u}) {
        for (int sorted = 0; sorted < 2; sorted++) {
            std::vector<unsigned> h(R);
            srand(1);
            for (auto& x : h) x = rand() % P;
            if (sorted)   // spatially coherent: neighbouring rows hit neighbouring Gaussians
                for (unsigned i = 0; i < R; i++) h[i] = (unsigned)(((unsigned long long)i * P) / R + (rand() % 64)) % P;
            unsigned* ids; float *acc, *val;
            hipMalloc(&ids, R * 4); hipMalloc(&acc, (size_t)P * 64); hipMalloc(&val, 256);
            hipMemcpy(ids, h.data(), R * 4, hipMemcpyHostToDevice);
            hipMemset(acc, 0, (size_t)P * 64); hipMemset(val, 0, 256);
            printf("---- R = %u rows, %s ids\n", R, sorted ? "coherent" : "random");
            for (int grid : {1024}) {
                run<0>("7x9 per instruction (kernel's flush)", ids, R, acc, val, grid);
                run<1>("lane = record, 9 instructions", ids, R, acc, val, grid);
                run<2>("4 records x 16 lanes", ids, R, acc, val, grid);
                run<3>("plain stores, 7x9", ids, R, acc, val, grid);
                run<4>("7x9 atomics sc1", ids, R, acc, val, grid);
                run<5>("7x9 atomics nt", ids, R, acc, val, grid);
                run<6>("7x9 atomics sc1 nt", ids, R, acc, val, grid);
            }
            hipFree(ids); hipFree(acc); hipFree(val);
        }
    }
    return 0;
}
