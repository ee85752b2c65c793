// Per-Gaussian backward for gfx950: one kernel that turns the blend-backward accumulators
// (dL/dmean2D, dL/dconic, dL/dopacity, dL/dcolour) into every gradient the API returns.
// reference: computeCov2DCUDA (backward.cu:144-274) + preprocessCUDA (backward.cu:346-396) +
// SH backward (backward.cu:20-139) + Sigma3D backward (backward.cu:278-341), fused.
//
// Compiled with -ffp-contract=off and written in the reference's operation order (see
// fr_preprocess.hip).  Every output row is written (zeros for culled Gaussians), so callers
// need not pre-zero anything — or, per array, ADDED to what the array holds (FR_FLAG_ACCUMULATE:
// gradient accumulation over the frames of a batch without a second buffer and an add kernel).
#include "fr_bind_math.hpp"

namespace fr {

__constant__ float bSH_C0 = 0.28209479177387814f;
__constant__ float bSH_C1 = 0.4886025119029199f;
__constant__ float bSH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                                -1.0925484305920792f, 0.5462742152960396f};
__constant__ float bSH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                                -0.4570457994644658f, 1.445305721320277f,  -0.5900435899266435f};

struct PreBwdArgs {
    int P, D, M, W, H, raw;
    float tan_fovx, tan_fovy, focal_x, focal_y, scale_modifier;
    const float* means3D;
    const float* scales;
    const float* rotations;
    const float* shs;
    const float* cov3D_precomp;
    const float* view;
    const float* proj;
    const float* campos;
    const int* radii;
    GeomView g;
    fr_grads out;
    float* grad_accum;  // optional (fr_aux): += ||dL_dmeans2D[:, :2]|| of visible Gaussians
    float* denom;       // optional (fr_aux): += 1 for visible Gaussians
    float* overflow_out;  // optional (fr_aux): 1.0f if the frame overflowed its binning capacity (all gradients zero), else 0.0f
    const DeviceCounts* counts;   // the frame's counts: an overflowed frame (nothing was blended) adds nothing to the statistics
    uint32_t acc;       // bit k: ADD into the k-th array of fr_grads instead of overwriting it (FR_FLAG_ACCUMULATE)
    int bound;          // fr_aux::binding: the gradients of mean / rotation / scale continue through `bind` into `bg`
    BindArgs bind;
    BindGrads bg;
};

// bit positions of `acc` = position of the pointer in fr_grads
enum { G_MEANS2D = 0, G_COLORS, G_OPACITY, G_MEANS3D, G_COV3D, G_SH, G_SCALES, G_ROTATIONS };

// one gradient element: overwritten, or added to what the array holds (wave-uniform choice per array)
__device__ __forceinline__ void put(float* p, float v, bool add) { *p = add ? *p + v : v; }

__device__ __forceinline__ void store3(float* p, size_t i, float a, float b, float c, bool add)
{
    if (!p) return;
    if (add) a += p[3 * i], b += p[3 * i + 1], c += p[3 * i + 2];
    p[3 * i] = a, p[3 * i + 1] = b, p[3 * i + 2] = c;
}

// Everything for ONE Gaussian.  `row` (LDS, may be null) receives this Gaussian's dL_dsh row (the kernel writes the rows
// of a wave through LDS for coalesced HBM access).
__device__ __forceinline__ void preprocess_bwd_one(const PreBwdArgs& a, const CameraRegs& cam, const int radius,
                                                   const int idx, float* row)
{
    const size_t i = (size_t)idx;
    const int Mc = a.M;
    const auto adds = [&](int k) { return ((a.acc >> k) & 1u) != 0u; };
    if (!(radius > 0)) {   // no gradient: zeros where the arrays are overwritten, nothing where they accumulate
        if (row)
            for (int k = 0; k < Mc * 3; k++) row[k] = 0.f;
        if (!adds(G_MEANS2D)) store3(a.out.dL_dmeans2D, i, 0.f, 0.f, 0.f, false);
        if (!adds(G_COLORS)) store3(a.out.dL_dcolors, i, 0.f, 0.f, 0.f, false);
        if (a.out.dL_dopacity && !adds(G_OPACITY)) a.out.dL_dopacity[i] = 0.f;
        if (!adds(G_MEANS3D)) store3(a.out.dL_dmeans3D, i, 0.f, 0.f, 0.f, false);
        if (a.out.dL_dcov3D && !adds(G_COV3D))
            for (int k = 0; k < 6; k++) a.out.dL_dcov3D[6 * i + k] = 0.f;
        if (a.out.dL_dsh && !row && !adds(G_SH))
            for (int k = 0; k < Mc * 3; k++) a.out.dL_dsh[i * Mc * 3 + k] = 0.f;
        if (!adds(G_SCALES)) store3(a.out.dL_dscales, i, 0.f, 0.f, 0.f, false);
        if (a.out.dL_drotations && !adds(G_ROTATIONS))
            for (int k = 0; k < 4; k++) a.out.dL_drotations[4 * i + k] = 0.f;
        if (a.bound) bind_one_bwd_zero(idx, a.bg);
        return;
    }
    // accumulating arrays: what they hold is requested NOW, so that the round trip runs under the arithmetic below
    // (a load next to its store would sit behind the stores in front of it)
    float old_m3[3] = {0.f, 0.f, 0.f}, old_sc[3] = {0.f, 0.f, 0.f}, old_q[4] = {0.f, 0.f, 0.f, 0.f}, old_op = 0.f;
    if (a.acc) {
        if (adds(G_MEANS3D) && a.out.dL_dmeans3D)
            for (int k = 0; k < 3; k++) old_m3[k] = a.out.dL_dmeans3D[3 * i + k];
        if (adds(G_SCALES) && a.out.dL_dscales)
            for (int k = 0; k < 3; k++) old_sc[k] = a.out.dL_dscales[3 * i + k];
        if (adds(G_ROTATIONS) && a.out.dL_drotations)
            for (int k = 0; k < 4; k++) old_q[k] = a.out.dL_drotations[4 * i + k];
        if (adds(G_OPACITY) && a.out.dL_dopacity) old_op = a.out.dL_dopacity[i];
    }
    // read the accumulator row and leave it zeroed for the next backward (the rows are zero between backward
    // passes: no zeroing launch, and no zeroing writes in the forward)
    float acc[12];
    {
        float4* row4 = reinterpret_cast<float4*>(a.g.accum + i * kAccumStride);
        const float4 r0 = row4[0], r1 = row4[1], r2 = row4[2];
        row4[0] = row4[1] = row4[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        acc[0] = r0.x, acc[1] = r0.y, acc[2] = r0.z, acc[3] = r0.w, acc[4] = r1.x, acc[5] = r1.y, acc[6] = r1.z,
        acc[7] = r1.w, acc[8] = r2.x, acc[9] = r2.y, acc[10] = r2.z, acc[11] = r2.w;
    }
    const float3 mean = make_float3(a.means3D[3 * idx], a.means3D[3 * idx + 1], a.means3D[3 * idx + 2]);
    const float* vm = cam.view;
    // Sigma3D: the caller's, or computed again from the scales and rotations (the same function and bits as the forward)
    float q_r = 0.f, q_x = 0.f, q_y = 0.f, q_z = 0.f, rot_inv = 1.0f;
    float sc[3] = {0.f, 0.f, 0.f};
    float c3[6];
    if (a.scales) {
        q_r = a.rotations[4 * idx], q_x = a.rotations[4 * idx + 1], q_y = a.rotations[4 * idx + 2], q_z = a.rotations[4 * idx + 3];
        sc[0] = a.scales[3 * idx], sc[1] = a.scales[3 * idx + 1], sc[2] = a.scales[3 * idx + 2];
        if (a.raw) {
            sc[0] = act_exp(sc[0]), sc[1] = act_exp(sc[1]), sc[2] = act_exp(sc[2]);
            rot_inv = act_rot_inv_norm(q_r, q_x, q_y, q_z);
            q_r *= rot_inv, q_x *= rot_inv, q_y *= rot_inv, q_z *= rot_inv;
        }
    }
    if (a.cov3D_precomp) {
        for (int k = 0; k < 6; k++) c3[k] = a.cov3D_precomp[6 * i + k];
    } else {
        cov3d_from_scale_rot(sc[0], sc[1], sc[2], q_r, q_x, q_y, q_z, a.scale_modifier, c3);
    }

    // ---------------- conic -> Sigma2D -> Sigma3D and mean (backward.cu:144-274)
    float3 t = xform4x3(mean, vm);
    const float limx = 1.3f * a.tan_fovx, limy = 1.3f * a.tan_fovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const float x_grad_mul = txtz < -limx || txtz > limx ? 0 : 1;
    const float y_grad_mul = tytz < -limy || tytz > limy ? 0 : 1;
    const float h_x = a.focal_x, h_y = a.focal_y;
    const float j00 = h_x / t.z, j02 = -(h_x * t.x) / (t.z * t.z);
    const float j11 = h_y / t.z, j12 = -(h_y * t.y) / (t.z * t.z);
    float T0[3], T1[3];  // T0[w] = T[0][w], T1[w] = T[1][w] of the reference's column-major T
    for (int w = 0; w < 3; w++) {
        const float W0 = vm[4 * w], W1 = vm[4 * w + 1], W2 = vm[4 * w + 2];
        T0[w] = W0 * j00 + W1 * 0.0f + W2 * j02;
        T1[w] = W0 * 0.0f + W1 * j11 + W2 * j12;
    }
    const float V[3][3] = {{c3[0], c3[1], c3[2]}, {c3[1], c3[3], c3[4]}, {c3[2], c3[4], c3[5]}};
    float A0[3], A1[3];
    for (int c = 0; c < 3; c++) {
        A0[c] = T0[0] * V[0][c] + T0[1] * V[1][c] + T0[2] * V[2][c];
        A1[c] = T1[0] * V[0][c] + T1[1] * V[1][c] + T1[2] * V[2][c];
    }
    float ca = A0[0] * T0[0] + A0[1] * T0[1] + A0[2] * T0[2];
    const float cb = A1[0] * T0[0] + A1[1] * T0[1] + A1[2] * T0[2];
    float cc = A1[0] * T1[0] + A1[1] * T1[1] + A1[2] * T1[2];
    ca += 0.3f;
    cc += 0.3f;
    const float denom = ca * cc - cb * cb;
    // moments of q = dL/dG * G accumulated by the blend backward -> the reference's per-Gaussian sums
    // (backward.cu:538-554): dG/ddelx = -G*(dx*A + dy*B), dG/ddely = -G*(dy*C + dx*B)
    // The conic is not read back: it is the inverse of the 2D covariance this kernel has just computed again, with the
    // forward's expressions (forward.cu:74-113, 207-219) — the same bits; only the activated opacity comes from the state.
    const float det_inv = 1.f / denom;
    const float4 co = make_float4(cc * det_inv, -cb * det_inv, ca * det_inv, a.g.opacity_act[idx]);
    // (ACC_MX / ACC_MY arrive combined with the conic per pixel: sum of q * dG/d(centre) / G, fr_common.hpp)
    const float g2x = acc[ACC_MX] * (0.5f * a.W);
    const float g2y = acc[ACC_MY] * (0.5f * a.H);
    const float dcx = -0.5f * acc[ACC_CA], dcy = -0.5f * acc[ACC_CB], dcz = -0.5f * acc[ACC_CC];
    const float dop = (co.w != 0.f) ? acc[ACC_OP] / co.w : 0.f;
    float dcol[3] = {acc[ACC_R], acc[ACC_G], acc[ACC_B]};
    store3(a.out.dL_dmeans2D, i, g2x, g2y, 0.f, adds(G_MEANS2D));
    // fused _add_densification_stats (model/fateavatar.py:734-737); this branch is radii > 0
    // (a replayed frame that overflowed its captured binning capacity back-propagates zeros: it must not count as a view)
    if (!a.counts->overflow) {
        if (a.grad_accum) a.grad_accum[i] += sqrtf(g2x * g2x + g2y * g2y);
        if (a.denom) a.denom[i] += 1.0f;
    }
    store3(a.out.dL_dcolors, i, dcol[0], dcol[1], dcol[2], adds(G_COLORS));
    // raw-parameter mode: d sigmoid = o (1 - o); co.w is the activated opacity the forward stored
    if (a.out.dL_dopacity) a.out.dL_dopacity[i] = old_op + (a.raw ? dop * co.w * (1.0f - co.w) : dop);

    float dL_da = 0, dL_db = 0, dL_dc = 0;
    const float denom2inv = 1.0f / ((denom * denom) + 0.0000001f);
    float dcov[6] = {0, 0, 0, 0, 0, 0};
    if (denom2inv != 0) {
        dL_da = denom2inv * (-cc * cc * dcx + 2 * cb * cc * dcy + (denom - ca * cc) * dcz);
        dL_dc = denom2inv * (-ca * ca * dcz + 2 * ca * cb * dcy + (denom - ca * cc) * dcx);
        dL_db = denom2inv * 2 * (cb * cc * dcx - (denom + 2 * cb * cb) * dcy + ca * cb * dcz);
        dcov[0] = (T0[0] * T0[0] * dL_da + T0[0] * T1[0] * dL_db + T1[0] * T1[0] * dL_dc);
        dcov[3] = (T0[1] * T0[1] * dL_da + T0[1] * T1[1] * dL_db + T1[1] * T1[1] * dL_dc);
        dcov[5] = (T0[2] * T0[2] * dL_da + T0[2] * T1[2] * dL_db + T1[2] * T1[2] * dL_dc);
        dcov[1] = 2 * T0[0] * T0[1] * dL_da + (T0[0] * T1[1] + T0[1] * T1[0]) * dL_db + 2 * T1[0] * T1[1] * dL_dc;
        dcov[2] = 2 * T0[0] * T0[2] * dL_da + (T0[0] * T1[2] + T0[2] * T1[0]) * dL_db + 2 * T1[0] * T1[2] * dL_dc;
        dcov[4] = 2 * T0[2] * T0[1] * dL_da + (T0[1] * T1[2] + T0[2] * T1[1]) * dL_db + 2 * T1[1] * T1[2] * dL_dc;
    }
    if (a.out.dL_dcov3D)
        for (int k = 0; k < 6; k++) put(a.out.dL_dcov3D + 6 * i + k, dcov[k], adds(G_COV3D));

    // dL/dT (upper 2x3), backward.cu:237-248.  Vrk[c][r] is symmetric.
    float dT0[3], dT1[3];
    for (int c = 0; c < 3; c++) {
        const float tv0 = T0[0] * V[c][0] + T0[1] * V[c][1] + T0[2] * V[c][2];
        const float tv1 = T1[0] * V[c][0] + T1[1] * V[c][1] + T1[2] * V[c][2];
        dT0[c] = 2 * tv0 * dL_da + tv1 * dL_db;
        dT1[c] = 2 * tv1 * dL_dc + tv0 * dL_db;
    }
    // dL/dJ, backward.cu:252-255 (W[c][r] = vm[c + 4r])
    const float dL_dJ00 = vm[0] * dT0[0] + vm[4] * dT0[1] + vm[8] * dT0[2];
    const float dL_dJ02 = vm[2] * dT0[0] + vm[6] * dT0[1] + vm[10] * dT0[2];
    const float dL_dJ11 = vm[1] * dT1[0] + vm[5] * dT1[1] + vm[9] * dT1[2];
    const float dL_dJ12 = vm[2] * dT1[0] + vm[6] * dT1[1] + vm[10] * dT1[2];
    const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    // NB the clamped t.x / t.y are treated as constants here (stop-gradient quirk of the reference)
    const float dL_dtx = x_grad_mul * -h_x * tz2 * dL_dJ02;
    const float dL_dty = y_grad_mul * -h_y * tz2 * dL_dJ12;
    const float dL_dtz = -h_x * tz2 * dL_dJ00 - h_y * tz2 * dL_dJ11 + (2 * h_x * t.x) * tz3 * dL_dJ02 + (2 * h_y * t.y) * tz3 * dL_dJ12;
    float dmx = vm[0] * dL_dtx + vm[1] * dL_dty + vm[2] * dL_dtz;
    float dmy = vm[4] * dL_dtx + vm[5] * dL_dty + vm[6] * dL_dtz;
    float dmz = vm[8] * dL_dtx + vm[9] * dL_dty + vm[10] * dL_dtz;

    // ---------------- projection part of dL/dmean3D (backward.cu:370-387)
    {
        const float* proj = cam.proj;
        const float4 m_hom = xform4x4(mean, proj);
        const float m_w = 1.0f / (m_hom.w + 0.0000001f);
        const float mul1 = (proj[0] * mean.x + proj[4] * mean.y + proj[8] * mean.z + proj[12]) * m_w * m_w;
        const float mul2 = (proj[1] * mean.x + proj[5] * mean.y + proj[9] * mean.z + proj[13]) * m_w * m_w;
        const float px = (proj[0] * m_w - proj[3] * mul1) * g2x + (proj[1] * m_w - proj[3] * mul2) * g2y;
        const float py = (proj[4] * m_w - proj[7] * mul1) * g2x + (proj[5] * m_w - proj[7] * mul2) * g2y;
        const float pz = (proj[8] * m_w - proj[11] * mul1) * g2x + (proj[9] * m_w - proj[11] * mul2) * g2y;
        dmx += px, dmy += py, dmz += pz;
    }

    // ---------------- SH backward (backward.cu:20-139)
    if (a.shs) {
        const float dox = mean.x - cam.campos[0], doy = mean.y - cam.campos[1], doz = mean.z - cam.campos[2];
        const float len = sqrtf(dox * dox + doy * doy + doz * doz);
        const float x = dox / len, y = doy / len, z = doz / len;
        // (the derivative of the colour with respect to the direction comes from the forward, GeomView::dcolor_ddir: this
        // kernel does not read the SH coefficients at all; `row`, if staged, only collects the dL_dsh row)
        const uint8_t cl = a.g.clamped[idx];
        float dRGB[3];
        for (int c = 0; c < 3; c++) dRGB[c] = dcol[c] * (((cl >> c) & 1) ? 0 : 1);
        const float* dd = a.g.dcolor_ddir + i * 9;
        const float dRGBdx[3] = {dd[0], dd[1], dd[2]}, dRGBdy[3] = {dd[3], dd[4], dd[5]}, dRGBdz[3] = {dd[6], dd[7], dd[8]};
        const int deg = a.D;
        const int used = (deg + 1) * (deg + 1);
        // the dL_dsh row: into the wave's LDS block (coalesced to HBM afterwards) or straight into the array.  One body,
        // called once per destination: a pointer that may be either made every access a FLAT one.
        auto write_row = [&](float* dsh, const bool dsh_adds) {
#define DSH(k, c, v_) put(dsh + (k) * 3 + (c), (v_), dsh_adds)
            for (int c = 0; c < 3; c++) {
                DSH(0, c, bSH_C0 * dRGB[c]);
                if (deg > 0) {
                    DSH(1, c, (-bSH_C1 * y) * dRGB[c]);
                    DSH(2, c, (bSH_C1 * z) * dRGB[c]);
                    DSH(3, c, (-bSH_C1 * x) * dRGB[c]);
                    if (deg > 1) {
                        const float xx = x * x, yy = y * y, zz = z * z;
                        const float xy = x * y, yz = y * z, xz = x * z;
                        DSH(4, c, (bSH_C2[0] * xy) * dRGB[c]);
                        DSH(5, c, (bSH_C2[1] * yz) * dRGB[c]);
                        DSH(6, c, (bSH_C2[2] * (2.f * zz - xx - yy)) * dRGB[c]);
                        DSH(7, c, (bSH_C2[3] * xz) * dRGB[c]);
                        DSH(8, c, (bSH_C2[4] * (xx - yy)) * dRGB[c]);
                        if (deg > 2) {
                            DSH(9, c, (bSH_C3[0] * y * (3.f * xx - yy)) * dRGB[c]);
                            DSH(10, c, (bSH_C3[1] * xy * z) * dRGB[c]);
                            DSH(11, c, (bSH_C3[2] * y * (4.f * zz - xx - yy)) * dRGB[c]);
                            DSH(12, c, (bSH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy)) * dRGB[c]);
                            DSH(13, c, (bSH_C3[4] * x * (4.f * zz - xx - yy)) * dRGB[c]);
                            DSH(14, c, (bSH_C3[5] * z * (xx - yy)) * dRGB[c]);
                            DSH(15, c, (bSH_C3[6] * x * (xx - 3.f * yy)) * dRGB[c]);
                        }
                    }
                }
            }
            // coefficients above the active degree receive no gradient (the reference leaves its
            // zero-initialised rows untouched)
            if (!dsh_adds)
                for (int k = used; k < Mc; k++) dsh[k * 3] = 0.f, dsh[k * 3 + 1] = 0.f, dsh[k * 3 + 2] = 0.f;
#undef DSH
        };
        if (row) write_row(row, false);                                   // (added to the array by unstage_rows)
        else if (a.out.dL_dsh) write_row(a.out.dL_dsh + i * Mc * 3, adds(G_SH));
        const float ddx = dRGBdx[0] * dRGB[0] + dRGBdx[1] * dRGB[1] + dRGBdx[2] * dRGB[2];
        const float ddy = dRGBdy[0] * dRGB[0] + dRGBdy[1] * dRGB[1] + dRGBdy[2] * dRGB[2];
        const float ddz = dRGBdz[0] * dRGB[0] + dRGBdz[1] * dRGB[1] + dRGBdz[2] * dRGB[2];
        // dnormvdv (auxiliary.h:107-117)
        const float sum2 = dox * dox + doy * doy + doz * doz;
        const float invsum32 = 1.0f / sqrtf(sum2 * sum2 * sum2);
        dmx += ((+sum2 - dox * dox) * ddx - doy * dox * ddy - doz * dox * ddz) * invsum32;
        dmy += (-dox * doy * ddx + (sum2 - doy * doy) * ddy - doz * doy * ddz) * invsum32;
        dmz += (-dox * doz * ddx - doy * doz * ddy + (sum2 - doz * doz) * ddz) * invsum32;
    } else if (a.out.dL_dsh && !row && !adds(G_SH)) {
        for (int k = 0; k < Mc * 3; k++) a.out.dL_dsh[i * Mc * 3 + k] = 0.f;
    }
    store3(a.out.dL_dmeans3D, i, old_m3[0] + dmx, old_m3[1] + dmy, old_m3[2] + dmz, false);
    const float g_mean[3] = {dmx, dmy, dmz};   // (this frame's, for the binding)
    float g_scl[3] = {0.f, 0.f, 0.f}, g_rot[4] = {0.f, 0.f, 0.f, 0.f};

    // ---------------- Sigma3D -> scale, quaternion (backward.cu:278-341)
    if (a.scales) {
        const float r = q_r, x = q_x, y = q_y, z = q_z;   // (read and activated above, for Sigma3D)
        const float s[3] = {a.scale_modifier * sc[0], a.scale_modifier * sc[1], a.scale_modifier * sc[2]};
        // Rc[c][w]: the reference's column-major R (its column c is row c of the usual rotation matrix)
        const float Rc[3][3] = {{1.f - 2.f * (y * y + z * z), 2.f * (x * y - r * z), 2.f * (x * z + r * y)},
                                {2.f * (x * y + r * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - r * x)},
                                {2.f * (x * z - r * y), 2.f * (y * z + r * x), 1.f - 2.f * (x * x + y * y)}};
        // M = S * R : M[c][w] = s_w * Rc[c][w];  dL_dSigma symmetric with halved off-diagonals
        const float dS[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]},
                                {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]},
                                {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
        // dL_dM = (2M) * dL_dSigma : dM[c][w] = sum_k (2 M[k][w]) * dS[c][k]
        float dM[3][3];
        for (int c = 0; c < 3; c++)
            for (int w = 0; w < 3; w++)
                dM[c][w] = (s[w] * Rc[0][w] * 2.0f) * dS[c][0] + (s[w] * Rc[1][w] * 2.0f) * dS[c][1] + (s[w] * Rc[2][w] * 2.0f) * dS[c][2];
        // Rt[c][w] = Rc[w][c], dMt[c][w] = dM[w][c];  dL_dscale_c = dot(Rt[c], dMt[c])
        float dMt[3][3];
        for (int c = 0; c < 3; c++)
            for (int w = 0; w < 3; w++) dMt[c][w] = dM[w][c];
        const float dsx = Rc[0][0] * dMt[0][0] + Rc[1][0] * dMt[0][1] + Rc[2][0] * dMt[0][2];
        const float dsy = Rc[0][1] * dMt[1][0] + Rc[1][1] * dMt[1][1] + Rc[2][1] * dMt[1][2];
        const float dsz = Rc[0][2] * dMt[2][0] + Rc[1][2] * dMt[2][1] + Rc[2][2] * dMt[2][2];
        // raw-parameter mode: d exp = the activated scale
        g_scl[0] = a.raw ? dsx * sc[0] : dsx, g_scl[1] = a.raw ? dsy * sc[1] : dsy, g_scl[2] = a.raw ? dsz * sc[2] : dsz;
        store3(a.out.dL_dscales, i, old_sc[0] + g_scl[0], old_sc[1] + g_scl[1], old_sc[2] + g_scl[2], false);
        for (int w = 0; w < 3; w++) dMt[0][w] *= s[0], dMt[1][w] *= s[1], dMt[2][w] *= s[2];
#define Dm(c_, r_) dMt[c_][r_]
        if (a.out.dL_drotations || a.bound) {
            float dq[4];
            dq[0] = 2 * z * (Dm(0, 1) - Dm(1, 0)) + 2 * y * (Dm(2, 0) - Dm(0, 2)) + 2 * x * (Dm(1, 2) - Dm(2, 1));
            dq[1] = 2 * y * (Dm(1, 0) + Dm(0, 1)) + 2 * z * (Dm(2, 0) + Dm(0, 2)) + 2 * r * (Dm(1, 2) - Dm(2, 1)) - 4 * x * (Dm(2, 2) + Dm(1, 1));
            dq[2] = 2 * x * (Dm(1, 0) + Dm(0, 1)) + 2 * r * (Dm(2, 0) - Dm(0, 2)) + 2 * z * (Dm(1, 2) + Dm(2, 1)) - 4 * y * (Dm(2, 2) + Dm(0, 0));
            dq[3] = 2 * r * (Dm(0, 1) - Dm(1, 0)) + 2 * x * (Dm(2, 0) + Dm(0, 2)) + 2 * y * (Dm(1, 2) + Dm(2, 1)) - 4 * z * (Dm(1, 1) + Dm(0, 0));
            if (a.raw) {
                // through q_hat = q / |q|:  dL/dq = (g - q_hat (q_hat . g)) / |q|
                const float dot = r * dq[0] + x * dq[1] + y * dq[2] + z * dq[3];
                dq[0] = (dq[0] - r * dot) * rot_inv;
                dq[1] = (dq[1] - x * dot) * rot_inv;
                dq[2] = (dq[2] - y * dot) * rot_inv;
                dq[3] = (dq[3] - z * dot) * rot_inv;
            }
            if (a.out.dL_drotations)
                for (int k = 0; k < 4; k++) a.out.dL_drotations[4 * i + k] = old_q[k] + dq[k];
            for (int k = 0; k < 4; k++) g_rot[k] = dq[k];
        }
#undef Dm
    } else {
        if (!adds(G_SCALES)) store3(a.out.dL_dscales, i, 0.f, 0.f, 0.f, false);
        if (a.out.dL_drotations && !adds(G_ROTATIONS))
            for (int k = 0; k < 4; k++) a.out.dL_drotations[4 * i + k] = 0.f;
    }
    // ---------------- through the mesh binding (model/fateavatar.py:225-258): fr_bind_backward's expressions
    if (a.bound) bind_one_bwd(a.bind, idx, g_mean, g_rot, g_scl, a.bg);
}

// A wave's [64][row_len] block of dL_dsh rows goes from LDS to HBM in coalesced 16-byte stores (the inverse of the
// forward's staging, fr_preprocess.hip).
template <bool ADD>
__device__ __forceinline__ void unstage_rows_t(float* __restrict__ dst, const float* src, int stride, int rows, int row_len, int lane)
{
    const int total = rows * row_len;
    const unsigned magic = (unsigned)((0x100000000ull + (unsigned)row_len - 1u) / (unsigned)row_len);
    // ADD: the array's old values are requested kBatch 16-byte loads at a time before the first one is used
    constexpr int kBatch = ADD ? 6 : 1;
    for (int base = lane * 4; base < total; base += 64 * 4 * kBatch) {
        float4 o[kBatch];
        if (ADD) {
#pragma unroll
            for (int u = 0; u < kBatch; u++) {
                const int c = base + u * 64 * 4;
                o[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (c + 3 < total) {
                    o[u] = *reinterpret_cast<const float4*>(dst + c);
                } else {
                    if (c < total) o[u].x = dst[c];
                    if (c + 1 < total) o[u].y = dst[c + 1];
                    if (c + 2 < total) o[u].z = dst[c + 2];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            const int c = base + u * 64 * 4;
            if (c >= total) break;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int e = c + k;
                const int r = (int)__umulhi((unsigned)e, magic);  // e / row_len (exact: e < 2^16)
                v[k] = (e < total) ? src[r * stride + (e - r * row_len)] : 0.f;
            }
            if (ADD) v[0] += o[u].x, v[1] += o[u].y, v[2] += o[u].z, v[3] += o[u].w;
            if (c + 3 < total) *reinterpret_cast<float4*>(dst + c) = make_float4(v[0], v[1], v[2], v[3]);
            else
                for (int k = 0; k < 4; k++)
                    if (c + k < total) dst[c + k] = v[k];
        }
    }
}
__device__ __forceinline__ void unstage_rows(float* __restrict__ dst, const float* src, int stride, int rows, int row_len, int lane,
                                             bool add)
{
    if (add) unstage_rows_t<true>(dst, src, stride, rows, row_len, lane);
    else unstage_rows_t<false>(dst, src, stride, rows, row_len, lane);
}

#ifndef FR_PREBWD_WAVES
#define FR_PREBWD_WAVES 4
#endif
constexpr int kPreBwdWaves = FR_PREBWD_WAVES;  // waves per workgroup of k_preprocess_bwd

__device__ __forceinline__ void preprocess_bwd_body(const PreBwdArgs& a)
{
    extern __shared__ __attribute__((aligned(16))) float s_rows[];
    if ((int)(blockIdx.x * blockDim.x) >= a.P) return;   // (a batched launch's grid is the largest view's)
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // (fr_aux::overflow_out: the step's optimizer kernel reads it — fr_adam_config::skip)
    if (a.overflow_out && blockIdx.x == 0 && threadIdx.x == 0) *a.overflow_out = a.counts->overflow ? 1.0f : 0.0f;
    const int M3 = a.M * 3, stride = M3 | 1;
    const int wave_first = blockIdx.x * (64 * kPreBwdWaves) + wave * 64;
    const int rows = min(64, a.P - wave_first);
    const bool staged = a.shs != nullptr && a.out.dL_dsh != nullptr;
    float* w_rows = s_rows + (size_t)wave * 64 * stride;
    // camera and radius are requested before the SH block is staged: one round trip for all of them
    const CameraRegs cam = load_camera(a.view, a.proj, a.campos, lane);
    const int radius = idx < a.P ? a.radii[idx] : 0;
    if (idx < a.P) preprocess_bwd_one(a, cam, radius, idx, staged ? w_rows + lane * stride : nullptr);
    __syncthreads();
    if (staged && rows > 0) unstage_rows(a.out.dL_dsh + (size_t)wave_first * M3, w_rows, stride, rows, M3, lane, ((a.acc >> G_SH) & 1u) != 0u);
}

__global__ void __launch_bounds__(64 * kPreBwdWaves) k_preprocess_bwd(PreBwdArgs a) { preprocess_bwd_body(a); }
__global__ void __launch_bounds__(64 * kPreBwdWaves) k_preprocess_bwd_batch(BatchOf<PreBwdArgs> b) { preprocess_bwd_body(b.v[blockIdx.y]); }

// implemented in fr_blend.hip: the blend backward of n views (their accumulators in g[k].accum)
int launch_blend_backward(int n, const BackwardCall* calls, const GeomView* g, const ImageView* v, hipStream_t s, bool debug);

int launch_backward(int n, const BackwardCall* calls, hipStream_t s)
{
    GeomView g[kMaxBatch];
    ImageView v[kMaxBatch];
    PreBwdArgs args[kMaxBatch];
    bool capturing = false, debug = false;
    for (int k = 0; k < n; k++) capturing = note_capture(calls[k].h, s) || capturing;
    size_t lds = 0;
    uint32_t blocks = 0;
    const int wg = 64 * kPreBwdWaves;
    for (int k = 0; k < n; k++) {
        fr_handle_impl* h = calls[k].h;
        const fr_params& prm = *calls[k].prm;
        const fr_inputs& in = *calls[k].in;
        const int P = prm.P;
        g[k] = GeomView::make(calls[k].geometry, (size_t)P);
        // backward passes of one handle share its gradient accumulators (the blend backward adds rows, k_preprocess_bwd
        // reads and re-zeroes them): order this one behind the previous one if that ran on another stream.  (Not inside a
        // capture: a capture is ordered by its own stream, replays by whoever launches them — as for the forward.)
        if (!capturing && h->have_last_bwd && h->last_bwd_stream != s) FR_HIP(hipStreamWaitEvent(s, h->bwd_done, 0));
        {   // gradient accumulators: handle-owned, zero between backward passes (normally sized by the forward already)
            int rc0 = ensure_accum(h, (size_t)P, s);
            if (rc0) return rc0;
            g[k].accum = h->accum;
        }
        v[k] = ImageView::make(const_cast<void*>(calls[k].image), prm.W, prm.H);
        debug = debug || prm.debug != 0;
        PreBwdArgs& a = args[k];
        a.P = P, a.D = prm.D, a.M = prm.M, a.W = prm.W, a.H = prm.H;
        a.raw = (prm.flags & FR_FLAG_RAW_ACTIVATIONS) ? 1 : 0;
        a.tan_fovx = prm.tan_fovx, a.tan_fovy = prm.tan_fovy;
        a.focal_y = prm.H / (2.0f * prm.tan_fovy);
        a.focal_x = prm.W / (2.0f * prm.tan_fovx);
        a.scale_modifier = prm.scale_modifier;
        a.means3D = in.means3D, a.scales = in.scales, a.rotations = in.rotations, a.shs = in.shs;
        a.cov3D_precomp = in.cov3D_precomp, a.view = in.viewmatrix, a.proj = in.projmatrix, a.campos = in.campos;
        a.radii = calls[k].radii, a.g = g[k], a.out = *calls[k].grads;
        a.grad_accum = prm.aux ? prm.aux->grad_accum : nullptr;
        a.denom = prm.aux ? prm.aux->denom : nullptr;
        a.overflow_out = prm.aux ? prm.aux->overflow_out : nullptr;
        a.counts = v[k].counts;
        a.acc = ((uint32_t)prm.flags >> FR_FLAG_ACCUMULATE_SHIFT) & 0xFFu;
        a.bound = (prm.aux && prm.aux->binding) ? 1 : 0;
        if (a.bound) {
            a.bind = bind_args(*prm.aux->binding);
            a.bg = BindGrads{prm.aux->d_verts, prm.aux->d_offset, prm.aux->d_rotation, prm.aux->d_scaling};
        } else {
            a.bind = BindArgs{};
            a.bg = BindGrads{};
        }
        const size_t l = (in.shs && calls[k].grads->dL_dsh) ? (size_t)kPreBwdWaves * 64 * ((prm.M * 3) | 1) * sizeof(float) : 0;
        lds = l > lds ? l : lds;
        blocks = max(blocks, (uint32_t)((P + wg - 1) / wg));
    }
    // g.accum is all zero here: zeroed when allocated, and again by k_preprocess_bwd after every backward
    int rc = launch_blend_backward(n, calls, g, v, s, debug);
    if (rc) return rc;
    {
        StageScope sc(calls[0].h, ST_PREPROCESS_BWD, s);
        launch_views(k_preprocess_bwd, k_preprocess_bwd_batch, n, args, blocks, (uint32_t)wg, lds, s);
    }
    FR_HIP(hipGetLastError());
    if (!capturing)
        for (int k = 0; k < n; k++) {
            fr_handle_impl* h = calls[k].h;
            FR_HIP(hipEventRecord(h->bwd_done, s));
            h->last_bwd_stream = s, h->have_last_bwd = true;
        }
    if (debug) FR_HIP(hipStreamSynchronize(s));
    return FR_OK;
}

}  // namespace fr
