// Per-Gaussian forward stages for gfx950: preprocess (cull, project, covariance, SH colour,
// tile rectangle, tile histogram), tile scan, instance emission.
//
// Compiled with -ffp-contract=off: the per-Gaussian arithmetic below is written in the same
// operation order as the reference (forward.cu:74-152,155-256; auxiliary.h:41-77), so radii,
// pixel centres, conics, depths and colours are bit-comparable with the CPU oracle.  These
// kernels are bandwidth-trivial; the missing FMAs cost nothing measurable.
#include "fr_bind_math.hpp"

// Development only, compiled out of the product build: -DFR_DIAG_PRE_ABLATE=mask — timing experiments on k_preprocess_fwd (results are WRONG): bit 0 no SH colour (no SH loads), bit 1
//                     no counting pass, bit 2 the counting pass without its key stores, bit 3 without its global atomics, bit 4
//                     without the early SH line touches, bit 5 without the geometry (everything culled right after the loads), bit 6 without the d colour / d direction stores, bit 7 without the
//                     blend-record stores
#ifdef FR_DIAG_PRE_ABLATE
#define FR_PRE_ABLATE(k) (((FR_DIAG_PRE_ABLATE) >> (k)) & 1)
#else
#define FR_PRE_ABLATE(k) false
#endif

// -DFR_DIAG_PRE_TRACE: per-wave cycle stamps at the phase boundaries of k_preprocess_fwd (tools/diag/pre_phases.py reads them)
#ifdef FR_DIAG_PRE_TRACE
namespace fr {
constexpr unsigned kPreTraceWaves = 16384, kPreTraceSlots = 12;
__device__ unsigned long long g_pre_trace[kPreTraceWaves * kPreTraceSlots];
}
extern "C" int fr_diag_read_pre_trace(void* dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(fr::g_pre_trace), bytes < sizeof(fr::g_pre_trace) ? bytes : sizeof(fr::g_pre_trace));
}
#define PRE_TR_DECL unsigned long long ptr_acc[::fr::kPreTraceSlots] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ptr_t = __builtin_readcyclecounter(), ptr_rt0 = __builtin_amdgcn_s_memrealtime()
#define PRE_TR(k) do { const unsigned long long n_ = __builtin_readcyclecounter(); ptr_acc[(k)] += n_ - ptr_t; ptr_t = n_; } while (0)
#define PRE_TR_STORE() do { \
        ptr_acc[11] = (ptr_rt0 << 32) | (__builtin_amdgcn_s_memrealtime() & 0xFFFFFFFFull); \
        const unsigned w_ = (blockIdx.x * kPreWG + threadIdx.x) >> 6; \
        if (lane == 0 && w_ < ::fr::kPreTraceWaves && blockIdx.y == 0) \
            for (unsigned k_ = 0; k_ < ::fr::kPreTraceSlots; k_++) ::fr::g_pre_trace[(size_t)w_ * ::fr::kPreTraceSlots + k_] = ptr_acc[k_]; \
    } while (0)
#else
#define PRE_TR_DECL do { } while (0)
#define PRE_TR(k) do { } while (0)
#define PRE_TR_STORE() do { } while (0)
#endif

namespace fr {

__constant__ float kSH_C0 = 0.28209479177387814f;
__constant__ float kSH_C1 = 0.4886025119029199f;
__constant__ float kSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f};
__constant__ float kSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

struct PreArgs {
    int P, D, M, W, H, raw;
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* opacities;
    const float* shs;
    const float* cov3D_precomp;
    const float* colors_precomp;
    const float* view;
    const float* proj;
    const float* campos;
    int* radii;
    uint8_t* visible;       // optional (fr_aux): radii > 0
    int bound;              // fr_aux::binding: means3D / rotations / scales are WRITTEN here, from `bind`
    BindArgs bind;
    GeomView g;
    uint32_t* tile_count;
    uint64_t* buckets;      // key buckets [tiles][8][bucket_cap]
    uint32_t bucket_cap;
    uint32_t tpad;          // row pitch of the per-XCD counter copies
    DeviceCounts* counts;
    int tiles_x, tiles_y;   // 8x8 tiles
    int ref_gx, ref_gy;     // 16x16 tiles (reference grid)
};

// SH -> RGB for one channel; same expression tree as forward.cu:30-63.
__device__ __forceinline__ float sh_channel(const float* sh, int c, int deg, float x, float y, float z)
{
#define SH(k) sh[(k) * 3 + c]
    float result = kSH_C0 * SH(0);
    if (deg > 0) {
        result = result - kSH_C1 * y * SH(1) + kSH_C1 * z * SH(2) - kSH_C1 * x * SH(3);
        if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z;
            float xy = x * y, yz = y * z, xz = x * z;
            result = result + kSH_C2[0] * xy * SH(4) + kSH_C2[1] * yz * SH(5) + kSH_C2[2] * (2.0f * zz - xx - yy) * SH(6) +
                     kSH_C2[3] * xz * SH(7) + kSH_C2[4] * (xx - yy) * SH(8);
            if (deg > 2) {
                result = result + kSH_C3[0] * y * (3.0f * xx - yy) * SH(9) + kSH_C3[1] * xy * z * SH(10) +
                         kSH_C3[2] * y * (4.0f * zz - xx - yy) * SH(11) +
                         kSH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * SH(12) +
                         kSH_C3[4] * x * (4.0f * zz - xx - yy) * SH(13) + kSH_C3[5] * z * (xx - yy) * SH(14) +
                         kSH_C3[6] * x * (xx - 3.0f * yy) * SH(15);
            }
        }
    }
#undef SH
    return result + 0.5f;
}

// d(colour of channel c) / d(direction): the expressions of the reference's SH backward (backward.cu:44-133), evaluated
// HERE, in the forward, where the coefficients are staged in LDS anyway (GeomView::dcolor_ddir).
// (Not part of the forward state the tests hold bit-exact — it only feeds dL/dmeans3D in the backward, which is compared at
// 1e-4 — so its multiply-adds may contract: 120 instructions per wave less than with the file's -ffp-contract=off.)
__device__ __forceinline__ void sh_dchannel_ddir(const float* sh, int c, int deg, float x, float y, float z, float& ddx, float& ddy,
                                                 float& ddz)
{
#pragma clang fp contract(fast)
#define SH(k) sh[(k) * 3 + c]
    ddx = ddy = ddz = 0.f;
    if (deg > 0) {
        ddx = -kSH_C1 * SH(3);
        ddy = -kSH_C1 * SH(1);
        ddz = kSH_C1 * SH(2);
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            const float xy = x * y, yz = y * z, xz = x * z;
            ddx += kSH_C2[0] * y * SH(4) + kSH_C2[2] * 2.f * -x * SH(6) + kSH_C2[3] * z * SH(7) + kSH_C2[4] * 2.f * x * SH(8);
            ddy += kSH_C2[0] * x * SH(4) + kSH_C2[1] * z * SH(5) + kSH_C2[2] * 2.f * -y * SH(6) + kSH_C2[4] * 2.f * -y * SH(8);
            ddz += kSH_C2[1] * y * SH(5) + kSH_C2[2] * 2.f * 2.f * z * SH(6) + kSH_C2[3] * x * SH(7);
            if (deg > 2) {
                ddx += (kSH_C3[0] * SH(9) * 3.f * 2.f * xy + kSH_C3[1] * SH(10) * yz + kSH_C3[2] * SH(11) * -2.f * xy +
                        kSH_C3[3] * SH(12) * -3.f * 2.f * xz + kSH_C3[4] * SH(13) * (-3.f * xx + 4.f * zz - yy) +
                        kSH_C3[5] * SH(14) * 2.f * xz + kSH_C3[6] * SH(15) * 3.f * (xx - yy));
                ddy += (kSH_C3[0] * SH(9) * 3.f * (xx - yy) + kSH_C3[1] * SH(10) * xz + kSH_C3[2] * SH(11) * (-3.f * yy + 4.f * zz - xx) +
                        kSH_C3[3] * SH(12) * -3.f * 2.f * yz + kSH_C3[4] * SH(13) * -2.f * xy + kSH_C3[5] * SH(14) * -2.f * yz +
                        kSH_C3[6] * SH(15) * -3.f * 2.f * xy);
                ddz += (kSH_C3[1] * SH(10) * xy + kSH_C3[2] * SH(11) * 4.f * 2.f * yz + kSH_C3[3] * SH(12) * 3.f * (2.f * zz - xx - yy) +
                        kSH_C3[4] * SH(13) * 4.f * 2.f * xz + kSH_C3[5] * SH(14) * (xx - yy));
            }
        }
    }
#undef SH
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// One thread per Gaussian.  reference: preprocessCUDA, forward.cu:155-256.
// The SH rows ([64 Gaussians][M*3] contiguous floats per wave) are NOT staged through LDS any more (12.5 KB per wave:
// three waves per SIMD): the wave touches every 128-byte line of its block once when it starts, and each thread loads its
// own row (twelve 16-byte loads at M = 16, L2 hits by then) when it evaluates its colour.  The kernel's LDS is the
// counting pass's tables only (5.9 KB per wave), the registers allow five waves per SIMD: 24.1 -> 22.5 us for one frame,
// 57 -> 48 us for the four frames of a batched launch.

// Can any pixel centre of 8x8 tile (tx, ty) lie inside the alpha >= 1/255 footprint of a Gaussian?  The footprint is
// the ellipse q(d) = a dx^2 + 2 b dx dy + c dy^2 <= tau2 around `ctr`; the minimum of the convex q over the tile's
// pixel rectangle is at the centre if that is inside, otherwise on an edge FACING the centre (walking from the
// minimiser towards the centre decreases q and can only leave the rectangle through such an edge).
// fp32 throughout (the double version was over half of this kernel's instruction stream): the answer only has to be
// CONSERVATIVE — a tile kept without need costs one instance whose pixels all fail the blend kernels' exact test, a
// tile dropped wrongly would lose contributions — so q is compared against tau2 plus a bound on its own rounding
// error (a few ulp of the sum of the magnitudes of its terms; q evaluated at a slightly misplaced point of the edge is
// only larger than the minimum by the square of the misplacement).
__device__ __forceinline__ bool footprint_touches_tile(const float4 conic_tau2, const float2 ctr, uint32_t tx, uint32_t ty)
{
    if (!(conic_tau2.w < 3.0e38f)) return true;
    const float A = conic_tau2.x, B = conic_tau2.y, C = conic_tau2.z, t2 = conic_tau2.w;
    const float lox = (float)(tx * kTile) - ctr.x, hix = lox + (float)(kTile - 1);
    const float loy = (float)(ty * kTile) - ctr.y, hiy = loy + (float)(kTile - 1);
    const float qx = lox > 0.f ? lox : (hix < 0.f ? hix : 0.f);
    const float qy = loy > 0.f ? loy : (hiy < 0.f ? hiy : 0.f);
    if (qx == 0.f && qy == 0.f) return true;
    // edge dx = qx: best dy = -b qx / c, clamped to the edge;  edge dy = qy: best dx = -b qy / a
    const float dy = fminf(fmaxf(-B * qx * __builtin_amdgcn_rcpf(C), loy), hiy);
    const float dx = fminf(fmaxf(-B * qy * __builtin_amdgcn_rcpf(A), lox), hix);
    const float ax = A * qx * qx, bx = 2.f * B * qx * dy, cx = C * dy * dy;
    const float ay = A * dx * dx, by = 2.f * B * dx * qy, cy = C * qy * qy;
    const float q1 = (ax + bx) + cx, m1 = (ax + fabsf(bx)) + cx;
    const float q2 = (ay + by) + cy, m2 = (ay + fabsf(by)) + cy;
    const bool in1 = qx != 0.f && q1 <= t2 + (4.0e-6f * m1 + 1.0e-4f);
    const bool in2 = qy != 0.f && q2 <= t2 + (4.0e-6f * m2 + 1.0e-4f);
    return in1 || in2;
}

constexpr int kCountUnroll = 4;                           // candidates per lane and pass of the counting loop
constexpr int kCountLdsBytes = 2816 + 3 * kCountUnroll * 256;  // per-wave tables of the counting pass (see below): 5888 B

// The SH rows of a wave's 64 Gaussians are one contiguous block of 64 x M3 floats.  It is copied into LDS by LDS-DMA
// (global_load_lds_dwordx4: lane l's 16 bytes land at base + 16 l, no staging registers, fully coalesced 1-KB pieces), issued
// when the kernel starts, and every thread reads its own row from there when it evaluates its colour.  (Each thread loading
// its own 192-byte row from memory — twelve 16-byte loads at a 192-byte stride across the wave — asked the L1 for every
// 128-byte line of the block eight times: 25 of config 5's 73 us, tools/diag/ab_pre.sh.)  The counting pass's tables reuse
// the space once the rows have been read.
// The copy is issued from inline assembly (FR_PRE_SH_DMA 2; 1 = the compiler's builtin): with the builtin the compiler knows
// that an LDS-DMA is in flight somewhere and puts an s_waitcnt vmcnt(0) in front of EVERY later LDS access of the kernel —
// each of the counting pass's phases then waited for all of the thread's outstanding stores.  The waits the copy needs are
// written out where its rows are read; an instruction the compiler does not count only makes its own counted waits stricter.
#ifndef FR_PRE_SH_DMA
#define FR_PRE_SH_DMA 2
#endif
__device__ __forceinline__ void dma16_to_lds(const void* g, uint32_t lds_byte_address /* wave-uniform */)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte_address), "v"(g) : "memory");   // (m0 is reserved: the compiler sets it itself before every use of its own)
}
// FR_PRE_ROWS_IN_REGS: at M = 16 every thread takes its row into registers BEFORE the counting tables are written, which then
// reuse the block's space (12 KB per wave instead of 18: three waves per SIMD instead of two) at the price of 48 registers
// held across the counting pass
#ifndef FR_PRE_ROWS_IN_REGS
#define FR_PRE_ROWS_IN_REGS 1
#endif
__host__ __device__ constexpr int pre_wave_lds_bytes(int M3)
{
    // the SH block (it has to be there BEFORE the counting atomics are issued and is read while they are in flight: loads
    // and returning atomics come back in order, a block requested behind the atomics would arrive behind them), then the
    // tables of the counting pass
#if FR_PRE_ROWS_IN_REGS
    if (M3 == 48) return 256 * M3;   // (the rows are in registers by the time the tables are written: same space)
#endif
    return 256 * M3 + kCountLdsBytes;
}

__device__ __forceinline__ void preprocess_fwd_body(const PreArgs& a)
{
    extern __shared__ __attribute__((aligned(16))) float s_sh[];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the frame's device counts (cursors of k_tile_totals) start from zero: the image buffer is caller-owned scratch
    if (blockIdx.x == 0 && threadIdx.x < sizeof(DeviceCounts) / 4) reinterpret_cast<uint32_t*>(a.counts)[threadIdx.x] = 0u;
    if ((int)(blockIdx.x * kPreWG) >= a.P) return;   // (a batched launch's grid is the largest view's; P > 0: block 0 stays)
    const int M3 = a.M * 3;
    PRE_TR_DECL;
    // every input of this thread is requested up front (camera, mean, scale, rotation, opacity — and the SH block
    // below), so that the kernel pays one memory round trip for its inputs instead of one per use
    const CameraRegs cam = load_camera(a.view, a.proj, a.campos, lane);
    const bool live = idx < a.P;
    const int li = live ? idx : 0;
    float3 p_orig;
    float in_sc[3] = {0.f, 0.f, 0.f}, in_rot[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bound) {
        // the frame comes straight from its mesh binding (model/fateavatar.py:225-258): position, rotation and log-scale
        // of this Gaussian are evaluated here (fr_bind_forward's expressions) and stored for the backward and the caller
        float bp[3];
        bind_one_fwd(a.bind, li, bp, in_rot, in_sc);
        p_orig = make_float3(bp[0], bp[1], bp[2]);
        if (live) {
            float* m = const_cast<float*>(a.means3D) + 3 * (size_t)li;
            float* r = const_cast<float*>(a.rotations) + 4 * (size_t)li;
            float* sc = const_cast<float*>(a.scales) + 3 * (size_t)li;
            m[0] = bp[0], m[1] = bp[1], m[2] = bp[2];
            r[0] = in_rot[0], r[1] = in_rot[1], r[2] = in_rot[2], r[3] = in_rot[3];
            sc[0] = in_sc[0], sc[1] = in_sc[1], sc[2] = in_sc[2];
        }
    } else {
        p_orig = make_float3(a.means3D[3 * li], a.means3D[3 * li + 1], a.means3D[3 * li + 2]);
        if (a.scales) in_sc[0] = a.scales[3 * li], in_sc[1] = a.scales[3 * li + 1], in_sc[2] = a.scales[3 * li + 2];
        if (a.rotations)
            in_rot[0] = a.rotations[4 * li], in_rot[1] = a.rotations[4 * li + 1], in_rot[2] = a.rotations[4 * li + 2],
            in_rot[3] = a.rotations[4 * li + 3];
    }
    const float in_opacity = a.opacities[li];
    // (requested with the other inputs: a load issued behind the counting atomics would come back behind them)
    float in_col[3] = {0.f, 0.f, 0.f};
    if (a.colors_precomp) in_col[0] = a.colors_precomp[3 * li], in_col[1] = a.colors_precomp[3 * li + 1], in_col[2] = a.colors_precomp[3 * li + 2];
    // one LDS region per wave: the SH rows of its Gaussians, then the tables of its counting pass —
    // nothing in it is shared between waves, so the kernel needs no workgroup barrier for it
    float* const wave_lds = s_sh + (size_t)wave * (pre_wave_lds_bytes(M3) / 4);   // (launch_forward sizes it the same way)
    const int wave_first = blockIdx.x * kPreWG + wave * 64;
    // The colour stage reads every thread's SH row from LDS.  Full waves get their block by LDS-DMA; the last wave of the
    // launch (the copy moves whole 16-byte pieces and must not read past the array) has every thread copy its own row.
    const bool rows_in_lds = a.shs && !a.colors_precomp && !FR_PRE_ABLATE(0);
    const bool sh_in_lds = rows_in_lds && __builtin_amdgcn_readfirstlane((int)(FR_PRE_SH_DMA && wave_first + 64 <= a.P)) != 0;
#if FR_PRE_SH_DMA
    if (sh_in_lds) {
        const char* blk = reinterpret_cast<const char*>(a.shs + (size_t)wave_first * M3);
        const int bytes = 256 * M3;                                   // 64 rows x M3 floats
#if FR_PRE_SH_DMA == 2
        const uint32_t lds0 = (uint32_t)__builtin_amdgcn_readfirstlane(
            (int)(uintptr_t)(__attribute__((address_space(3))) void*)wave_lds);
#endif
        for (int off = 0; off < bytes; off += 1024)                   // (wave-uniform trip count)
            if (off + lane * 16 < bytes) {
#if FR_PRE_SH_DMA == 2
                dma16_to_lds(blk + off + lane * 16, lds0 + (uint32_t)off);
#else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(blk + off + lane * 16),
                                                 (__attribute__((address_space(3))) void*)(reinterpret_cast<char*>(wave_lds) + off), 16, 0, 0);
#endif
            }
    }
#endif
    if (rows_in_lds && !sh_in_lds && live)
        for (int k = 0; k < M3; k++) wave_lds[lane * M3 + k] = a.shs[(size_t)idx * M3 + k];
    uint32_t ref_tiles = 0;  // tiles_touched in reference semantics (16x16)
    uint2 rect = make_uint2(0u, 0u);
    float4 cull = make_float4(0.f, 0.f, 0.f, __builtin_inff());  // conic + footprint threshold (per-tile culling)
    float2 ctr = make_float2(0.f, 0.f);
    // what the geometry leaves for the colour stage of a Gaussian that passed it (`alive`)
    bool alive = false;
    int mr_out = 0;
    float rec_ca = 0.f, rec_cb = 0.f, rec_cc = 0.f, rec_op = 0.f, rec_z = 0.f;
    PRE_TR(0);   // loads issued
    if (idx < a.P) {
        do {
            // near cull only (auxiliary.h:154)
            const float3 p_view = xform4x3(p_orig, cam.view);
            PRE_TR(1);   // inputs landed (first use)
            if (p_view.z <= 0.2f) break;
            if (FR_PRE_ABLATE(5) && in_opacity > -1e30f) break;

            const float4 p_hom = xform4x4(p_orig, cam.proj);
            const float p_w = 1.0f / (p_hom.w + 0.0000001f);
            const float p_proj_x = p_hom.x * p_w, p_proj_y = p_hom.y * p_w;

            // ---- Sigma3D = R S^2 R^T, 6 upper-triangular floats (forward.cu:118-152)
            float c3[6];
            if (a.cov3D_precomp) {
                for (int k = 0; k < 6; k++) c3[k] = a.cov3D_precomp[6 * (size_t)idx + k];
            } else {
                float sc0 = in_sc[0], sc1 = in_sc[1], sc2 = in_sc[2];
                float r = in_rot[0], x = in_rot[1], y = in_rot[2], z = in_rot[3];
                if (a.raw) {
                    sc0 = act_exp(sc0), sc1 = act_exp(sc1), sc2 = act_exp(sc2);
                    const float inv = act_rot_inv_norm(r, x, y, z);
                    r *= inv, x *= inv, y *= inv, z *= inv;
                }
                cov3d_from_scale_rot(sc0, sc1, sc2, r, x, y, z, a.scale_modifier, c3);   // (not stored: k_preprocess_bwd computes it again)
            }

            // ---- EWA projection to a 2D covariance (forward.cu:74-113)
            float3 t = p_view;  // transformPoint4x3(mean, viewmatrix) again in the reference; same value
            const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
            const float txtz = t.x / t.z, tytz = t.y / t.z;
            t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
            t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
            const float j00 = a.focal_x / t.z, j02 = -(a.focal_x * t.x) / (t.z * t.z);
            const float j11 = a.focal_y / t.z, j12 = -(a.focal_y * t.y) / (t.z * t.z);
            const float* vm = cam.view;
            // T = J * W (2x3): row 0 and row 1; the zero products are kept so the sums round identically
            float T0[3], T1[3];
            for (int w = 0; w < 3; w++) {
                const float W0 = vm[4 * w], W1 = vm[4 * w + 1], W2 = vm[4 * w + 2];
                T0[w] = W0 * j00 + W1 * 0.0f + W2 * j02;
                T1[w] = W0 * 0.0f + W1 * j11 + W2 * j12;
            }
            const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
            float A0[3], A1[3];  // (T * Sigma) rows
            for (int c = 0; c < 3; c++) {
                A0[c] = T0[0] * V[0][c] + T0[1] * V[1][c] + T0[2] * V[2][c];
                A1[c] = T1[0] * V[0][c] + T1[1] * V[1][c] + T1[2] * V[2][c];
            }
            float cov_a = A0[0] * T0[0] + A0[1] * T0[1] + A0[2] * T0[2];
            const float cov_b = A1[0] * T0[0] + A1[1] * T0[1] + A1[2] * T0[2];
            float cov_c = A1[0] * T1[0] + A1[1] * T1[1] + A1[2] * T1[2];
            cov_a += 0.3f;
            cov_c += 0.3f;

            const float det = (cov_a * cov_c - cov_b * cov_b);
            if (det == 0.0f) break;
            const float det_inv = 1.f / det;
            const float conic_a = cov_c * det_inv, conic_b = -cov_b * det_inv, conic_c = cov_a * det_inv;

            const float mid = 0.5f * (cov_a + cov_c);
            const float lambda1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
            const float lambda2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
            const float my_radius = ceilf(3.f * sqrtf(fmaxf(lambda1, lambda2)));
            // ndc2Pix in double (auxiliary.h:41-44)
            const float pix_x = (float)((((double)p_proj_x + 1.0) * a.W - 1.0) * 0.5);
            const float pix_y = (float)((((double)p_proj_y + 1.0) * a.H - 1.0) * 0.5);
            // reference 16x16 rectangle (auxiliary.h:46-56)
            const int mr = (int)my_radius;
            const int rx0 = min(a.ref_gx, max(0, (int)((pix_x - mr) / kRefTile)));
            const int ry0 = min(a.ref_gy, max(0, (int)((pix_y - mr) / kRefTile)));
            const int rx1 = min(a.ref_gx, max(0, (int)((pix_x + mr + kRefTile - 1) / kRefTile)));
            const int ry1 = min(a.ref_gy, max(0, (int)((pix_y + mr + kRefTile - 1) / kRefTile)));
            if ((rx1 - rx0) * (ry1 - ry0) == 0) break;
            ref_tiles = (uint32_t)((rx1 - rx0) * (ry1 - ry0));


            const float opacity = a.raw ? act_sigmoid(in_opacity) : in_opacity;
            // Contract: a Gaussian whose projected state is not finite (NaN / inf position, scale,
            // rotation, opacity or colour) is DROPPED like a culled one — radius 0, no contribution to any pixel, zero gradient
            // rows — instead of spreading NaN over the image and every gradient as the reference's arithmetic would.  The
            // geometry is tested here, the colour when it has been evaluated (below).
            {
                const float chk = ((pix_x + pix_y) + (conic_a + conic_b + conic_c)) + (opacity + p_view.z);
                if (!(fabsf(chk) < 3.0e38f)) break;
            }
            alive = true;
            mr_out = mr;
            rec_ca = conic_a, rec_cb = conic_b, rec_cc = conic_c, rec_op = opacity, rec_z = p_view.z;
            ctr = make_float2(pix_x, pix_y);
            // ---- 8x8-tile rectangle of the alpha >= 1/255 footprint, clipped to the pixels the
            // reference rectangle covers.  A pixel outside it can never pass the blend kernels'
            // alpha test, so dropping those (tile, Gaussian) instances changes no pixel.
            int px0 = rx0 * kRefTile, px1 = min(rx1 * kRefTile, a.W) - 1;
            int py0 = ry0 * kRefTile, py1 = min(ry1 * kRefTile, a.H) - 1;
            if (!(opacity >= 1.0f / 255.0f)) break;  // alpha = opacity * G <= opacity < 1/255 everywhere
            {
                // fp32, conservative (see footprint_touches_tile).  dq = ac - b^2 the accurate way (Kahan): with a plain
                // a * c - b * b its relative error grows with the conditioning 1 / (1 - rho^2) of the conic.
                const float A = conic_a, B = conic_b, C = conic_c;
                const float bb = B * B;
                const float dq = __builtin_fmaf(A, C, -bb) - __builtin_fmaf(B, B, -bb);
                if (dq > 0.f && A > 0.f && C > 0.f && dq < 3.0e38f) {
                    const float inv_dq = 1.0f / dq;
                    const float cond = A * C * inv_dq;                      // 1 / (1 - rho^2)
                    float tau = logf(255.0f * opacity);                     // alpha >= 1/255  <=>  power >= -tau
                    tau = tau * (1.0f + 1.6e-5f * cond + 1e-5f) + 1e-4f;    // fp32 evaluation slack of `power`
                    const float ex = sqrtf(2.0f * tau * C * inv_dq) * (1.0f + 1e-5f) + 2e-3f;
                    const float ey = sqrtf(2.0f * tau * A * inv_dq) * (1.0f + 1e-5f) + 2e-3f;
                    // integer pixel coordinates p with |p - centre| <= extent (extent already carries the slack; the
                    // float sums below round by at most 1e-4 pixel at 4096 pixels, far inside it)
                    const float x_lo = ceilf(pix_x - ex), x_hi = floorf(pix_x + ex);
                    const float y_lo = ceilf(pix_y - ey), y_hi = floorf(pix_y + ey);
                    if (x_lo > (float)px0) px0 = (int)fminf(x_lo, (float)px1 + 1.0f);
                    if (x_hi < (float)px1) px1 = (int)fmaxf(x_hi, (float)px0 - 1.0f);
                    if (y_lo > (float)py0) py0 = (int)fminf(y_lo, (float)py1 + 1.0f);
                    if (y_hi < (float)py1) py1 = (int)fmaxf(y_hi, (float)py0 - 1.0f);
                    cull = make_float4(conic_a, conic_b, conic_c, 2.0f * tau * (1.0f + 1e-6f));
                }
            }
            if (px1 < px0 || py1 < py0) break;
            const int tx0 = px0 / kTile, tx1 = px1 / kTile + 1, ty0 = py0 / kTile, ty1 = py1 / kTile + 1;
            rect = make_uint2((uint32_t)tx0 | ((uint32_t)ty0 << 16), (uint32_t)tx1 | ((uint32_t)ty1 << 16));
        } while (false);
    }
    PRE_TR(2);   // geometry, footprint rectangle (and the wave's reconvergence)

    // ---- count the (tile, Gaussian) instances AND write their keys.  The atomic that counts an instance hands out
    // its slot in the (tile, XCD) key bucket, and the key goes there at once: no second pass over the Gaussians (the
    // reference's duplicateWithKeys, rasterizer_impl.cu:70-111) and no scan in front of it.  The wave spreads its
    // instances over its lanes: one returning atomic round trip per 64 instances instead of one per tile of the
    // widest rectangle.
    // The chip serves about 26 returning atomics per nanosecond whatever their addresses (tools/micro_atomics.hip): the
    // instances of BASELINE config 2 are 10 us of that, config 5's 38 us — so the atomics of a wave's first 256 instances
    // are ISSUED before its colours are evaluated and their keys are stored after: the SH evaluation, the record stores and
    // the SH rows' own trip from memory run while the atomics queue.
    const bool rows_in_regs = FR_PRE_ROWS_IN_REGS && M3 == 48;
    char* cl = reinterpret_cast<char*>(wave_lds) + (rows_in_regs ? 0 : 256 * M3);
    float4* s_cull = reinterpret_cast<float4*>(cl);                                  // [64]   1024 B
    uint2* s_rect = reinterpret_cast<uint2*>(cl + 1024);                              // [64]    512 B
    float2* s_ctr = reinterpret_cast<float2*>(cl + 1536);                             // [64]    512 B
    uint32_t* s_excl = reinterpret_cast<uint32_t*>(cl + 2048);                        // [64]   256 B
    uint2* s_gk = reinterpret_cast<uint2*>(cl + 2304);                                // [64]    512 B  (id, depth bits)
    // (the lanes of the wave talk through these: plain LDS accesses, ordered by wave-level fences between the
    // phases below — as `volatile` pointers they lost their address space and became FLAT loads and stores, whose
    // s_waitcnt vmcnt(0) also waited for every global atomic in flight)
    uint32_t* s_gkey = reinterpret_cast<uint32_t*>(cl + 2816);      // [4][64] 1024 B
    uint32_t* s_gcnt = reinterpret_cast<uint32_t*>(cl + 3840);      // [4][64] 1024 B  -> 4864
    uint32_t* s_gbase = reinterpret_cast<uint32_t*>(cl + 4864);     // [4][64] 1024 B  -> kCountLdsBytes
    const int rw_ = (int)(rect.y & 0xffff) - (int)(rect.x & 0xffff), rh_ = (int)(rect.y >> 16) - (int)(rect.x >> 16);
    const uint32_t n = (rw_ > 0 && rh_ > 0) ? (uint32_t)(rw_ * rh_) : 0u;
    uint32_t incl = n;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
    }
    const uint32_t total = __shfl(incl, 63);
    auto stage_tables = [&]() {
        s_cull[lane] = cull;
        s_ctr[lane] = ctr;
        s_gk[lane] = make_uint2((uint32_t)idx, __float_as_uint(rec_z));
        s_excl[lane] = incl - n;
        s_rect[lane] = rect;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // the SH block has landed (it was requested with the inputs; LDS-DMA counts on vmcnt like any load, and the compiler does
    // not see a dependence between the copy and the LDS reads of the colour stage): waited for HERE, in front of the atomics.
    // (As the builtin, not inline assembly: the compiler's own wait insertion must KNOW that nothing is outstanding here —
    // otherwise the first register it reuses that some never-taken path loaded into costs an s_waitcnt vmcnt(0) in the
    // middle of the colour stage, i.e. a wait for the atomics.)   vmcnt(0), expcnt and lgkmcnt untouched: 0x0F70 on gfx9
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
    float shr[48];
#if FR_PRE_ROWS_IN_REGS
    if (rows_in_regs && rows_in_lds) {
        const float4* row = reinterpret_cast<const float4*>(wave_lds + lane * 48);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const float4 q = row[k];
            shr[4 * k] = q.x, shr[4 * k + 1] = q.y, shr[4 * k + 2] = q.z, shr[4 * k + 3] = q.w;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // every lane has its row: the tables may take the space
        __builtin_amdgcn_wave_barrier();
    }
#endif
    stage_tables();
    const uint32_t xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & (uint32_t)(kXcds - 1);  // HW_REG_XCC_ID[3:0]
    PRE_TR(5);   // counting tables set up (scan)
    // kCountUnroll candidates per lane and pass: their returning atomics are all in flight before the first
    // result is needed (a wave typically has 2-4 x 64 candidates: one round trip instead of several).
    // Wave-level aggregation: lanes of one pass that count into the SAME counter (Gaussians stored in a spatially
    // coherent order, e.g. the reference's UV-raster initialisation, hit a handful of tiles per wave) are grouped
    // through a 64-slot table in LDS keyed by the counter index: the group's first lane issues ONE global atomic
    // for the whole group.  Lanes whose slot is taken by another counter fall back to their own atomic, so with
    // random orders (all counters distinct) nothing is lost but a few LDS operations.
    uint32_t rank[kCountUnroll], key[kCountUnroll], slot[kCountUnroll], tile_id[kCountUnroll], got[kCountUnroll], owner[kCountUnroll];
    uint2 gk[kCountUnroll];
    bool valid[kCountUnroll], grouped[kCountUnroll];
    static_assert(kCountUnroll == 4, "count_finish pins four positions");
    auto count_issue = [&](uint32_t b0) {   // candidates b0 .. b0 + 255: tile, footprint test, grouping, atomics issued
#pragma unroll
        for (int u = 0; u < kCountUnroll; u++) {
            const uint32_t k = b0 + 64u * (uint32_t)u + (uint32_t)lane;
            valid[u] = false, grouped[u] = false;
            rank[u] = 0, key[u] = 0, slot[u] = (uint32_t)lane, tile_id[u] = 0, gk[u] = make_uint2(0u, 0u), owner[u] = 0, got[u] = 0;
            s_gcnt[u * 64 + lane] = 0u;
            if (k < total) {
                // owner of candidate k: the last lane whose first candidate is <= k.  Six fixed steps, no branches: the
                // four candidates of a lane search in lockstep (as a `while (lo < hi)` loop each, the four searches ran one
                // behind the other: 24 dependent LDS round trips per pass instead of 6)
                int lo = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) lo += (s_excl[lo + step] <= k) ? step : 0;
                const uint2 rr = s_rect[lo];
                const uint32_t j = k - s_excl[lo];
                const uint32_t x0 = rr.x & 0xffff, y0 = rr.x >> 16, rw = (rr.y & 0xffff) - x0;
                // j / rw and the remainder through a float reciprocal and one correction either way (j < 2^24: exact in
                // float; the u32 division the compiler emits is thirty instructions per candidate)
                int qd = (int)((float)j * __builtin_amdgcn_rcpf((float)rw));
                int rd = (int)j - qd * (int)rw;
                if (rd < 0) qd -= 1, rd += (int)rw;
                if (rd >= (int)rw) qd += 1, rd -= (int)rw;
                const uint32_t ty = y0 + (uint32_t)qd, tx = x0 + (uint32_t)rd;
                // a one-tile-wide or one-tile-high rectangle is touched everywhere (the footprint is connected and
                // reaches both ends of its bounding box); only wider ones can miss a corner tile
                const uint32_t rh = (rr.y >> 16) - y0;
                if (!(rw > 1 && rh > 1) || footprint_touches_tile(s_cull[lo], s_ctr[lo], tx, ty)) {
                    valid[u] = true;
                    const uint32_t bx4 = (uint32_t)(a.tiles_x + 3) / 4;   // ImageView::counter_index
                    key[u] = xcc * a.tpad + ((ty >> 2) * bx4 + (tx >> 2)) * 16u + ((ty & 3u) << 2 | (tx & 3u));
                    tile_id[u] = ty * (uint32_t)a.tiles_x + tx;
                    gk[u] = s_gk[lo];
                    owner[u] = (uint32_t)lo;
                    slot[u] = (key[u] * 2654435761u) >> 26;
                    s_gkey[u * 64 + slot[u]] = key[u];   // several lanes may write: one of them wins the slot
                }
            }
        }
        PRE_TR(6);   // candidates: owner search, tile, footprint test, hash slot
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the slots' winners are in LDS
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < kCountUnroll; u++) {
            grouped[u] = valid[u] && s_gkey[u * 64 + slot[u]] == key[u];
            if (grouped[u]) rank[u] = atomicAdd(&s_gcnt[u * 64 + slot[u]], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // ... the groups' sizes
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int u = 0; u < kCountUnroll; u++) {
            if (valid[u] && (!grouped[u] || rank[u] == 0)) {
                const uint32_t n_add = grouped[u] ? s_gcnt[u * 64 + slot[u]] : 1u;
                got[u] = FR_PRE_ABLATE(3) ? (uint32_t)lane : atomicAdd(&a.tile_count[key[u]], n_add);
            }
        }
        PRE_TR(7);   // grouping, atomics issued
    };
    // `dead` (wave mask): Gaussians whose colour turned out not to be finite AFTER their instances had been counted — their
    // keys get depth +inf: they sort behind every real entry of their tile (no unit boundary of the real entries moves, the
    // image is what it is without them) and point at a record of opacity 0
    auto count_finish = [&](unsigned long long dead) {   // the atomics' results -> positions -> the four key stores
#pragma unroll
        for (int u = 0; u < kCountUnroll; u++)
            if (grouped[u] && rank[u] == 0) s_gbase[u * 64 + slot[u]] = got[u];
        PRE_TR(8);   // atomics returned
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // ... the groups' bases
        __builtin_amdgcn_wave_barrier();
        // every position first, then the four key stores one behind the other.  (As one loop — position, test, store —
        // the compiler put an s_waitcnt vmcnt(0) in front of every store: each waited for the one before it to complete,
        // three store round trips per pass.)
        uint32_t pos[kCountUnroll];
#pragma unroll
        for (int u = 0; u < kCountUnroll; u++) {
            const uint32_t base = grouped[u] ? s_gbase[u * 64 + slot[u]] : got[u];
            pos[u] = base + rank[u];
        }
        asm volatile("" : "+v"(pos[0]), "+v"(pos[1]), "+v"(pos[2]), "+v"(pos[3]));
#pragma unroll
        for (int u = 0; u < kCountUnroll; u++) {
            const uint32_t depth_bits = ((dead >> owner[u]) & 1ull) ? 0x7F800000u : gk[u].y;
            // (a position beyond the bucket is dropped: k_tile_totals sees the count and flags the frame)
            if (valid[u] && pos[u] < a.bucket_cap && !FR_PRE_ABLATE(2))
                a.buckets[((size_t)tile_id[u] * kXcds + xcc) * a.bucket_cap + pos[u]] = ((uint64_t)depth_bits << 32) | gk[u].x;
        }
        PRE_TR(9);   // key stores issued
    };
    const bool counting = total > 0u && !FR_PRE_ABLATE(1);
    if (counting) count_issue(0u);

    // ---- colour (forward.cu:20-71), with the atomics above in flight
    bool dead_colour = false;
    if (idx < a.P) {
        int radius_out = 0;
        if (alive) {
            float col[3];
            float raw_sum = 0.f;   // of the colour BEFORE the clamp at 0 (fmaxf would turn a NaN into 0)
            uint8_t clamp_bits = 0;
            float dd[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            const bool from_sh = !FR_PRE_ABLATE(0) && !a.colors_precomp;
            if (FR_PRE_ABLATE(0)) {
                col[0] = col[1] = col[2] = 0.5f, raw_sum = 1.5f;
            } else if (a.colors_precomp) {
                col[0] = in_col[0], col[1] = in_col[1], col[2] = in_col[2];
                raw_sum = (col[0] + col[1]) + col[2];
            } else {
                float dx = p_orig.x - cam.campos[0], dy = p_orig.y - cam.campos[1], dz = p_orig.z - cam.campos[2];
                const float len = sqrtf(dx * dx + dy * dy + dz * dz);
                dx = dx / len, dy = dy / len, dz = dz / len;
                // (one body for both sources of the coefficients; fully unrolled so that every index is a constant)
                auto eval_colour = [&](const float* sh) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float v = sh_channel(sh, c, a.D, dx, dy, dz);
                        if (v < 0) clamp_bits |= (uint8_t)(1u << c);
                        col[c] = fmaxf(v, 0.0f);
                        raw_sum += v;
                        sh_dchannel_ddir(sh, c, a.D, dx, dy, dz, dd[c], dd[3 + c], dd[6 + c]);
                    }
                };
                const float* lrow = wave_lds + lane * M3;
                if (M3 == 48) {   // twelve 16-byte LDS reads
                    if (!rows_in_regs) {
                        const float4* row = reinterpret_cast<const float4*>(lrow);
#pragma unroll
                        for (int k = 0; k < 12; k++) {
                            const float4 q = row[k];
                            shr[4 * k] = q.x, shr[4 * k + 1] = q.y, shr[4 * k + 2] = q.z, shr[4 * k + 3] = q.w;
                        }
                    }
                    eval_colour(shr);
                } else {
                    eval_colour(lrow);
                }
            }
            PRE_TR(3);   // colour
            dead_colour = !(fabsf(raw_sum) < 3.0e38f);
            if (!dead_colour) {
                radius_out = mr_out;
                if (from_sh) {
                    float* o = a.g.dcolor_ddir + (size_t)idx * 9;
                    if (!FR_PRE_ABLATE(6))
                        for (int k = 0; k < 9; k++) o[k] = dd[k];
                    else asm volatile("" ::"v"(dd[0]), "v"(dd[1]), "v"(dd[2]), "v"(dd[3]), "v"(dd[4]), "v"(dd[5]), "v"(dd[6]), "v"(dd[7]), "v"(dd[8]));
                }
                a.g.opacity_act[idx] = rec_op;
                a.g.clamped[idx] = clamp_bits;
            }
            {
                // the blend record (GeomView::rec_tmpl): the conic as (-0.5 a, -b, -0.5 c) — exact scalings — so that the blend
                // loops get the reference's power = a'dx^2 + c'dy^2 + b'dxdy (forward.cu:340) in three multiply-adds, and the
                // blend backward the reference's conic itself.  A Gaussian whose colour is not finite leaves a record of
                // opacity 0 and colour 0 behind its counted instances: no pixel passes its alpha test
                float4* t = a.g.rec_tmpl + (size_t)idx * 3;
                if (!FR_PRE_ABLATE(7)) {
                    t[0] = make_float4(ctr.x, ctr.y, rec_ca * -0.5f, -rec_cb);
                    t[1] = make_float4(rec_cc * -0.5f, dead_colour ? 0.f : rec_op, dead_colour ? 0.f : col[0], dead_colour ? 0.f : col[1]);
                    t[2] = make_float4(dead_colour ? 0.f : col[2], __uint_as_float((uint32_t)idx), rec_z, 0.f);   // (.z: view-space depth, for diagnostics)
                } else asm volatile("" ::"v"(col[0]), "v"(col[1]), "v"(col[2]));
            }
        }
        a.radii[idx] = radius_out;
        if (a.visible) a.visible[idx] = radius_out > 0 ? 1 : 0;
    }
    PRE_TR(4);   // record stores (and the wave's reconvergence)
    const unsigned long long dead = __ballot(dead_colour);
    if (counting) {
        count_finish(dead);
        if (total > 64u * kCountUnroll) {   // (rare: more than 256 instances in a wave) the remaining passes, one behind the other
            for (uint32_t b0 = 64 * kCountUnroll; b0 < total; b0 += 64 * kCountUnroll) {   // wave-uniform trip count
                count_issue(b0);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                count_finish(dead);
            }
        }
    }
    PRE_TR(10);
    PRE_TR_STORE();
    // num_rendered in reference semantics: per-workgroup partial sums, added up by k_tile_totals (a single
    // counter would serialise one device-scope atomic per wave, ~11 ns each)
    __shared__ uint32_t s_ref[4];
    uint32_t s = ref_tiles;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if (lane == 0) s_ref[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kPreWG / 64; w++) t += s_ref[w];
        a.g.block_ref_tiles[blockIdx.x] = t;
    }
}

__global__ void __launch_bounds__(kPreWG) k_preprocess_fwd(PreArgs a) { preprocess_fwd_body(a); }
// batched frames: view blockIdx.y
__global__ void __launch_bounds__(kPreWG) k_preprocess_fwd_batch(BatchOf<PreArgs> b) { preprocess_fwd_body(b.v[blockIdx.y]); }

// Everything between the counting pass and the sort, in ONE wide launch (it used to be a totals kernel plus a
// single-workgroup scan): per tile, add up the eight per-XCD counter copies, write the sub-list table and leave the
// counters zeroed for the next frame; then ALLOCATE the tile's record range and blend-unit range.  Nothing needs the
// tiles' lists in tile order, only each list contiguous, so a workgroup scans its own 1024 tiles in LDS and takes its
// block of records / units / list entries with one returning atomic per cursor (4 workgroups at 512 x 512): no
// grid-wide prefix, no second launch.  The frame counts (instances, units, longest list, largest bucket, the
// reference-semantics num_rendered) are reduced with a few atomics per workgroup; k_tile_sort's first thread turns
// them into the overflow verdict and the host-visible counts.
// reference counterparts: InclusiveSum + the blocking count read-back (rasterizer_impl.cu:277-281), identifyTileRanges.
struct TotalsArgs {
    ImageView v;
    uint32_t T;
    uint64_t capacity;
    const uint32_t* block_ref_tiles;
    uint32_t n_blocks;
};

#ifndef FR_TOT_WAVES
#define FR_TOT_WAVES 1
#endif
constexpr int kTotWaves = FR_TOT_WAVES;   // waves per workgroup of k_tile_totals: 256 tiles per wave.  Measured at 512 x 512 (4 096 tiles,
                                          // rocprofv3, one box): 1 wave per workgroup 5.4 us, 2: 6.1, 4: 6.6, 8: 8.2, 16 (ONE workgroup, which
                                          // then needs no cursor atomics): 11.4 — the pass is a chain of round trips per workgroup, and the
                                          // more CUs share its loads the shorter each link
__device__ __forceinline__ void tile_totals_body(const TotalsArgs& a)
{
    const ImageView v = a.v;
    const uint32_t T = a.T, n_blocks = a.n_blocks;
    const uint64_t capacity = a.capacity;
    const uint32_t* __restrict__ block_ref_tiles = a.block_ref_tiles;
    __shared__ uint32_t s_wave[2][kTotWaves];   // per-wave totals: instances, units
    __shared__ uint32_t s_base[8];        // workgroup bases: instances, units, medium / big / large list heads
    __shared__ uint32_t s_heads[3];       // workgroup-local list counters
    if (threadIdx.x < 3) s_heads[threadIdx.x] = 0;
    DeviceCounts* c = v.counts;
    // every thread: four consecutive counters = one row of a 4x4-tile block
    const uint32_t i = (blockIdx.x * (uint32_t)(64 * kTotWaves) + threadIdx.x) * 4u;
    uint32_t sub[4][kSubWords];
    uint32_t acc[4] = {0u, 0u, 0u, 0u};
    uint32_t biggest = 0;
    if (i < v.tpad) {
#pragma unroll
        for (int x = 0; x < kXcds; x++) {
            uint4* p1 = reinterpret_cast<uint4*>(v.tile_count + (size_t)x * v.tpad + i);
            const uint4 c1 = *p1;
            *p1 = make_uint4(0u, 0u, 0u, 0u);
            const uint32_t a1[4] = {c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
                sub[q][x] = acc[q];                  // start of XCD x's sub-list inside the tile's list
                acc[q] += a1[q];
                biggest = max(biggest, a1[q]);
            }
        }
    }
    const uint32_t bx = (uint32_t)(v.tiles_x + 3) / 4, blk = i >> 4, row = (i >> 2) & 3u;
    const uint32_t ty = (blk / bx) * 4 + row, tx0 = (blk % bx) * 4;
    const bool row_ok = i < v.tpad && ty < (uint32_t)v.tiles_y;   // (padding rows of the block grid are never counted into)
    uint32_t n_sum = 0, u_sum = 0, longest = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (!(row_ok && tx0 + q < (uint32_t)v.tiles_x)) acc[q] = 0;
        n_sum += acc[q];
        u_sum += (acc[q] + kUnit - 1) / kUnit;
        longest = max(longest, acc[q]);
    }
    // ---- workgroup scan of (instances, units): wave scan by shuffles, wave totals through LDS
    const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t n_inc = n_sum, u_inc = u_sum;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t tn = __shfl_up(n_inc, off), tu = __shfl_up(u_inc, off);
        if (ln >= off) n_inc += tn, u_inc += tu;
    }
    for (int off = 32; off > 0; off >>= 1) {
        biggest = max(biggest, (uint32_t)__shfl_down(biggest, off));
        longest = max(longest, (uint32_t)__shfl_down(longest, off));
    }
    // reference-semantics num_rendered: this workgroup adds up its slice of the preprocess workgroups' partial sums
    uint32_t ref = 0;
    for (uint32_t k = blockIdx.x * (uint32_t)(64 * kTotWaves) + threadIdx.x; k < n_blocks; k += gridDim.x * (uint32_t)(64 * kTotWaves)) ref += block_ref_tiles[k];
    for (int off = 32; off > 0; off >>= 1) ref += __shfl_down(ref, off);
    if (ln == 63) s_wave[0][wv] = n_inc, s_wave[1][wv] = u_inc;
    if (ln == 0) {
        if (biggest) atomicMax(&c->max_bucket, biggest);
        if (longest) atomicMax(&c->max_tile_list, longest);
        if (ref) atomicAdd(&c->num_rendered, ref);
    }
    // list membership of this thread's tiles (workgroup-local rank first, one global atomic per list and workgroup)
    uint32_t cls[4], lrank[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t n = acc[q];
        cls[q] = n > (uint32_t)kSortRegMax ? 2u : (n > (uint32_t)kSortGroupMax ? 1u : (n > (uint32_t)kSortWaveMax ? 0u : 3u));
        lrank[q] = 0;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++)
        if (cls[q] < 3u) lrank[q] = atomicAdd(&s_heads[cls[q]], 1u);
    __syncthreads();
    if (threadIdx.x < 5) {   // five returning atomics, one per lane: a single round trip, not five in a row
        const uint32_t k = threadIdx.x;
        uint32_t amount = 0;
        if (k < 2) for (int w = 0; w < kTotWaves; w++) amount += s_wave[k][w];
        else amount = s_heads[k - 2];
        uint32_t* cursor = k == 0 ? &c->num_instances : k == 1 ? &c->num_units : k == 2 ? &c->medium_tiles
                           : k == 3 ? &c->big_tiles : &c->large_tiles;
        // (the launch's only workgroup — every image up to 4 096 tiles, 512 x 512 — owns the cursors: plain stores, no
        // returning-atomic round trip in front of the tables)
        if (gridDim.x == 1) *cursor = amount, s_base[k] = 0u;
        else s_base[k] = amount ? atomicAdd(cursor, amount) : 0u;
        if (blockIdx.x == 0 && k == 0) c->capacity = (uint32_t)capacity;
    }
    __syncthreads();
    uint32_t n_run = s_base[0] + n_inc - n_sum, u_run = s_base[1] + u_inc - u_sum;
    for (int w = 0; w < wv; w++) n_run += s_wave[0][w], u_run += s_wave[1][w];
    if (!row_ok) return;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (tx0 + q >= (uint32_t)v.tiles_x) break;
        const uint32_t tile = ty * (uint32_t)v.tiles_x + tx0 + q;
        uint4* dst = reinterpret_cast<uint4*>(v.tile_sub + (size_t)tile * kSubWords);
#pragma unroll
        for (int w4 = 0; w4 < kSubWords / 4; w4++)
            dst[w4] = make_uint4(sub[q][4 * w4], sub[q][4 * w4 + 1], sub[q][4 * w4 + 2], sub[q][4 * w4 + 3]);
        v.tile_total[tile] = acc[q];
        v.tile_offset[tile] = n_run;
        v.unit_offset[tile] = u_run;
        n_run += acc[q];
        u_run += (acc[q] + kUnit - 1) / kUnit;
        if (cls[q] == 0u) v.medium_list[s_base[2] + lrank[q]] = tile;
        else if (cls[q] == 1u) v.big_list[s_base[3] + lrank[q]] = tile;
        else if (cls[q] == 2u) v.large_list[s_base[4] + lrank[q]] = tile;
    }
    (void)T;
}

__global__ void __launch_bounds__(64 * kTotWaves) k_tile_totals(TotalsArgs a) { tile_totals_body(a); }
__global__ void __launch_bounds__(64 * kTotWaves) k_tile_totals_batch(BatchOf<TotalsArgs> b) { tile_totals_body(b.v[blockIdx.y]); }

// reference: checkFrustum, rasterizer_impl.cu:54-66
__global__ void __launch_bounds__(256) k_mark_visible(int P, const float* means3D, const float* view, uint8_t* present)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P) return;
    const float3 p = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    const float3 pv = xform4x3(p, view);
    present[idx] = (pv.z <= 0.2f) ? 0 : 1;
}

int launch_mark_visible(int P, const float* means3D, const float* view, uint8_t* present, hipStream_t s)
{
    if (P <= 0) return FR_OK;
    hipLaunchKernelGGL(k_mark_visible, dim3((P + 255) / 256), dim3(256), 0, s, P, means3D, view, present);
    FR_HIP(hipGetLastError());
    return FR_OK;
}

// implemented in fr_blend.hip
int launch_sort_and_blend(int n, const FrameView* f, hipStream_t s, bool debug);

static int debug_sync(bool debug, hipStream_t s, const char* stage)
{
    if (!debug) return FR_OK;
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail_hip(e, stage, __FILE__, __LINE__);
    return FR_OK;
}

// The handle's buffers (gradient accumulators, per-tile counters, key buckets) grow with the scene and the tile grid.
// Growing means hipMalloc (and freeing or retiring the old buffer), which a capturing stream does not allow — and trying
// would invalidate the caller's capture.  Asked for EVERY view of a call before any of them touches the stream or a handle.
static bool capture_would_grow(const ForwardCall& c)
{
    const fr_handle_impl* h = c.h;
    const ImageView v = ImageView::make(nullptr, c.prm->W, c.prm->H);
    const size_t T = (size_t)v.tiles_x * v.tiles_y;
    const uint32_t need = h->counts_seen ? reinterpret_cast<const uint32_t*>(h->host_counts)[4] : 0u;
    return (size_t)c.prm->P > h->accum_rows || v.tpad > h->tile_counter_tiles || !h->key_buckets || T > h->bucket_tiles ||
           h->bucket_cap < need + need / 4;
}

// Everything one view's forward needs done on the host before its kernels can be enqueued: handle buffers sized, stream
// ordered behind the handle's previous frame, views of the caller's buffers, the kernels' argument blocks.
static int prepare_forward(const ForwardCall& c, hipStream_t s, bool capturing, FrameView& f, PreArgs& a, TotalsArgs& tot,
                           size_t& pre_lds)
{
    fr_handle_impl* h = c.h;
    const fr_params& prm = *c.prm;
    const fr_inputs& in = *c.in;
    const int P = prm.P, W = prm.W, H = prm.H;
    GeomView g = GeomView::make(c.geometry, (size_t)P);
    ImageView v = ImageView::make(c.image, W, H);
    const uint32_t T = (uint32_t)v.tiles_x * v.tiles_y;
    BinningView b = BinningView::make(c.binning, (size_t)c.cap, (size_t)T);
    int rc;
    // (launch_forward has already refused a capturing stream whose frame would need handle buffers grown: capture_would_grow)
    // the backward's gradient accumulators live in the handle (zero between backward passes): size them here, where an
    // allocation is still allowed (a backward may be part of a captured graph)
    if ((rc = ensure_accum(h, (size_t)P, s))) return rc;
    g.accum = h->accum;

    // frames of one handle share the per-tile counters: order this frame behind the previous one if that was
    // enqueued on a different stream (fr_common.hpp, fr_handle_impl::frame_done)
    if (!capturing && h->have_last && h->last_stream != s) FR_HIP(hipStreamWaitEvent(s, h->frame_done, 0));

    // per-tile counters: handle-owned, zero between frames (k_tile_totals restores the zeros), so a frame normally
    // starts without any zeroing launch.  They are (re)allocated when the tile grid grows — not possible while the
    // stream is being captured into a graph: run one eager frame of the same size first.
    const size_t counter_words = (size_t)kXcds * v.tpad;   // 8 XCD copies
    if (v.tpad > h->tile_counter_tiles) {   // (tile_counter_tiles holds the largest pitch allocated so far)
        if ((rc = release_buffer(h, h->tile_counters, s))) return rc;
        h->tile_counters = nullptr, h->tile_counter_tiles = 0;
        FR_HIP(hipMalloc(&h->tile_counters, counter_words * sizeof(uint32_t)));
        h->tile_counter_tiles = v.tpad;
        h->counters_clean = false;
    }
    if (!h->counters_clean)
        if ((rc = launch_zero(h->tile_counters, (size_t)kXcds * h->tile_counter_tiles * sizeof(uint32_t), s)))
            return rc;
    h->counters_clean = false;  // until every stage of this frame has been enqueued
    v.tile_count = h->tile_counters;

    // key buckets: handle-owned, [tiles][8 XCDs][bucket_cap].  They grow when the tile grid grows, and their capacity
    // doubles when the last frame that reported back needed more (word 4 of the pinned count slot): that frame was
    // flagged as overflowed, and the caller is repeating it right now.
    {
        uint32_t want = h->bucket_cap ? h->bucket_cap : kBucketCapInit;
        const uint32_t need = h->counts_seen ? reinterpret_cast<const uint32_t*>(h->host_counts)[4] : 0u;
        while (want < need + need / 4) want <<= 1;
        if (want != h->bucket_cap || (size_t)T > h->bucket_tiles) {
            if ((rc = release_buffer(h, h->key_buckets, s))) return rc;
            h->key_buckets = nullptr;
            const size_t tiles = (size_t)T > h->bucket_tiles ? (size_t)T : h->bucket_tiles;
            FR_HIP(hipMalloc(reinterpret_cast<void**>(&h->key_buckets), tiles * kXcds * want * sizeof(uint64_t)));
            h->bucket_tiles = tiles, h->bucket_cap = want;
        }
        v.buckets = h->key_buckets, v.bucket_cap = h->bucket_cap;
    }

    a.P = P, a.D = prm.D, a.M = prm.M, a.W = W, a.H = H;
    a.raw = (prm.flags & FR_FLAG_RAW_ACTIVATIONS) ? 1 : 0;
    a.tan_fovx = prm.tan_fovx, a.tan_fovy = prm.tan_fovy;
    a.focal_y = H / (2.0f * prm.tan_fovy);  // rasterizer_impl.cu:222-223
    a.focal_x = W / (2.0f * prm.tan_fovx);
    a.scale_modifier = prm.scale_modifier;
    a.means3D = in.means3D, a.scales = in.scales, a.rotations = in.rotations, a.opacities = in.opacities;
    a.shs = in.shs, a.cov3D_precomp = in.cov3D_precomp, a.colors_precomp = in.colors_precomp;
    a.view = in.viewmatrix, a.proj = in.projmatrix, a.campos = in.campos;
    a.visible = prm.aux ? prm.aux->visible : nullptr;
    a.bound = (prm.aux && prm.aux->binding) ? 1 : 0;
    if (a.bound) a.bind = bind_args(*prm.aux->binding);
    else a.bind = BindArgs{};
    a.radii = c.radii, a.g = g, a.tile_count = v.tile_count, a.buckets = v.buckets, a.bucket_cap = v.bucket_cap;
    a.tpad = v.tpad, a.counts = v.counts;
    a.tiles_x = v.tiles_x, a.tiles_y = v.tiles_y;
    a.ref_gx = (W + kRefTile - 1) / kRefTile, a.ref_gy = (H + kRefTile - 1) / kRefTile;
    pre_lds = (kPreWG / 64) * (size_t)pre_wave_lds_bytes(prm.M * 3);
    tot.v = v, tot.T = T, tot.capacity = c.cap, tot.block_ref_tiles = g.block_ref_tiles;
    tot.n_blocks = (uint32_t)((P + kPreWG - 1) / kPreWG);
    f.h = h, f.prm = c.prm, f.in = c.in, f.g = g, f.v = v, f.b = b, f.out_color = c.out_color;
    return FR_OK;
}

int launch_forward(int n, const ForwardCall* calls, hipStream_t s)
{
    FrameView f[kMaxBatch];
    PreArgs pre[kMaxBatch];
    TotalsArgs tot[kMaxBatch];
    int rc;
    bool capturing = false;
    for (int k = 0; k < n; k++) capturing = note_capture(calls[k].h, s) || capturing;
    if (capturing)   // (first pass over ALL views: nothing has been enqueued, no handle modified, when this returns)
        for (int k = 0; k < n; k++)
            if (capture_would_grow(calls[k]))
                return fail_msg(FR_ERR_UNSUPPORTED,
                                "this frame needs handle buffers (re)allocated, which cannot happen while the stream is "
                                "being captured: run one eager frame of this size on this handle first");
    size_t pre_lds = 0;
    uint32_t pre_blocks = 0, tot_blocks = 0;
    bool debug = false, no_wait = true;
    for (int k = 0; k < n; k++) {
        size_t lds;
        if ((rc = prepare_forward(calls[k], s, capturing, f[k], pre[k], tot[k], lds))) return rc;
        pre_lds = lds > pre_lds ? lds : pre_lds;
        pre_blocks = max(pre_blocks, (uint32_t)((calls[k].prm->P + kPreWG - 1) / kPreWG));
        tot_blocks = max(tot_blocks, (f[k].v.tpad + (uint32_t)(256 * kTotWaves) - 1u) / (uint32_t)(256 * kTotWaves));
        debug = debug || calls[k].prm->debug != 0;
        no_wait = no_wait && (calls[k].prm->flags & FR_FLAG_NO_WAIT) != 0;
    }
    fr_handle_impl* h0 = calls[0].h;   // (stage profiling of a batch goes to the first view's handle)
    if (pre_blocks > 0) {
        {
            StageScope sc(h0, ST_PREPROCESS_FWD, s);
            launch_views(k_preprocess_fwd, k_preprocess_fwd_batch, n, pre, pre_blocks, kPreWG, pre_lds, s);
        }
        FR_HIP(hipGetLastError());
        if ((rc = debug_sync(debug, s, "preprocess_fwd"))) return rc;
    }
    {
        StageScope sc(h0, ST_SCAN, s);
        for (int k = 0; k < n; k++)
            if (calls[k].prm->P <= 0 || pre_blocks == 0)   // (k_preprocess_fwd, which zeroes the frame's device counts, did not run for it)
                if ((rc = launch_zero(f[k].v.counts, sizeof(DeviceCounts), s))) return rc;
        launch_views(k_tile_totals, k_tile_totals_batch, n, tot, tot_blocks, 64 * kTotWaves, 0, s);
    }
    FR_HIP(hipGetLastError());
    if ((rc = debug_sync(debug, s, "scan_tiles"))) return rc;
    if ((rc = launch_sort_and_blend(n, f, s, debug))) return rc;
    for (int k = 0; k < n; k++) {
        fr_handle_impl* h = calls[k].h;
        h->counters_clean = true;
        if (!capturing) {
            FR_HIP(hipEventRecord(h->frame_done, s));
            h->last_stream = s, h->have_last = true;
        }
    }
    if (no_wait) return FR_OK;
    // The whole frame is enqueued; only now wait for the counts (GPU keeps working meanwhile).
    int result = FR_OK;
    for (int k = 0; k < n; k++) {
        fr_handle_impl* h = calls[k].h;
        if (calls[k].prm->flags & FR_FLAG_NO_WAIT) continue;
        FR_HIP(hipEventSynchronize(h->counts_ready));
        h->counts_seen = true;
        fr_counts c = *h->host_counts;
        if (calls[k].counts) *calls[k].counts = c;
        if (c.overflow) result = FR_ERR_BINNING_CAPACITY;
    }
    return result;
}

}  // namespace fr
