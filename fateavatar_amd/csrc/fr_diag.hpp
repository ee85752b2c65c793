// Development-only instrumentation of the blend kernels, compiled out of the product build.
//   -DFR_DIAG_STATS   per-launch iteration accounting of k_unit_blend_bwd_sparse in DeviceCounts::pair_hist
//                     (tools/diag/bwd_stats.sh builds and reads it)
// The product translation units only ever see the empty forms of these macros.
#pragma once

#ifdef FR_DIAG_STATS
// slot k of DeviceCounts::pair_hist += v, once per wave
#define FR_STAT_ADD(counts, k, v)                                                                     \
    do {                                                                                              \
        if (lane == 0) atomicAdd(&const_cast<::fr::DeviceCounts*>(counts)->pair_hist[(k)], (uint32_t)(v)); \
    } while (0)
#else
#define FR_STAT_ADD(counts, k, v) \
    do {                          \
    } while (0)
#endif

//   -DFR_DIAG_TRACE   per-unit cycle accounting of k_unit_blend_bwd_sparse (s_memtime stamps at the phase boundaries,
//                     summed over a unit's record ranges): tools/diag/bwd_phases.py builds and reads it
#ifdef FR_DIAG_TRACE
namespace fr {
constexpr unsigned kDiagUnits = 32768, kDiagSlots = 8;
__device__ unsigned long long g_diag_trace[kDiagUnits * kDiagSlots];
}
extern "C" int fr_diag_read_trace(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fr::g_diag_trace), bytes < sizeof(fr::g_diag_trace) ? bytes : sizeof(fr::g_diag_trace));
}
#define FR_TR_DECL unsigned long long tr_acc[::fr::kDiagSlots] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_t = __builtin_readcyclecounter(), tr_rt0 = __builtin_amdgcn_s_memrealtime()
// cycles since the previous stamp go to slot k
#define FR_TR(k)                                                          \
    do {                                                                  \
        const unsigned long long tr_n = __builtin_readcyclecounter();     \
        tr_acc[(k)] += tr_n - tr_t;                                       \
        tr_t = tr_n;                                                      \
    } while (0)
#define FR_TR_STORE(unit)                                                                                   \
    do {                                                                                                    \
        tr_acc[7] = (tr_rt0 << 32) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull);   /* 100 MHz wall clock: start, end */ \
        if (lane == 0 && (unit) < ::fr::kDiagUnits)                                                         \
            for (unsigned tr_k = 0; tr_k < ::fr::kDiagSlots; tr_k++) ::fr::g_diag_trace[(size_t)(unit) * ::fr::kDiagSlots + tr_k] = tr_acc[tr_k]; \
    } while (0)
#else
#define FR_TR_DECL \
    do {           \
    } while (0)
#define FR_TR(k) \
    do {         \
    } while (0)
#define FR_TR_STORE(unit) \
    do {                  \
    } while (0)
#endif

//   -DFR_DIAG_ABLATE=mask   timing experiments on k_unit_blend_bwd_sparse (results are WRONG): bit 2 skips phase A, bit 3
//                     phase B, bit 4 the flush's atomics, bit 5 turns them into plain stores
#ifdef FR_DIAG_ABLATE
#define FR_ABLATE(k) (((FR_DIAG_ABLATE) >> (k)) & 1)
#else
#define FR_ABLATE(k) false
#endif

//   -DFR_DIAG_FWD_TRACE   per-unit time stamps of k_unit_blend_chained's phases (tools/diag/fwd_trace.py)
#ifdef FR_DIAG_FWD_TRACE
namespace fr {
__device__ unsigned long long g_fwd_trace[16384 * 16];
}
extern "C" int fr_debug_read_fwd_trace(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fr::g_fwd_trace), bytes < sizeof(fr::g_fwd_trace) ? bytes : sizeof(fr::g_fwd_trace));
}
#define FW_STAMP(K) do { if (lane == 0 && u < 16384u) ::fr::g_fwd_trace[(size_t)u * 16 + (K)] = __builtin_readcyclecounter(); } while (0)
#define FW_STAMPV(K, V) do { if (lane == 0 && u < 16384u) ::fr::g_fwd_trace[(size_t)u * 16 + (K)] = (unsigned long long)(V); } while (0)
#else
#define FW_STAMP(K) do { } while (0)
#define FW_STAMPV(K, V) do { } while (0)
#endif

// (fr_preprocess.hip has its own timing-experiment switch, -DFR_DIAG_PRE_ABLATE=mask, defined at its head)
