// FateAvatar's mesh binding, per Gaussian: the arithmetic shared by the stand-alone binding kernels
// (fr_binding.hip) and by the per-Gaussian kernels of the rasterizer when a frame is rendered straight from its binding
// (fr_aux::binding: fr_preprocess.hip evaluates bind_one_fwd in front of its own work, fr_preprocess_bwd.hip continues
// through bind_one_bwd).  One source of the expressions = the same bits on both routes.
//
// reference: model/fateavatar.py:225-258 with volume_rendering/mesh_compute.py:27-59 and pytorch3d 0.7.7's
// matrix_to_quaternion / quaternion_multiply / standardize_quaternion.  Every translation unit that includes this is
// built with -ffp-contract=off: the expressions are in the oracle's operation order.
#pragma once
#include "fr_common.hpp"

namespace fr {

struct BindArgs {
    int N;
    const float* verts;       // [V,3]
    const int* faces;         // [F,3]
    const int* face_index;    // [N]
    const float* bary;        // [N,3]
    const float* canon;       // [F] face scale of the canonical mesh
    float shell_len;
    int resize_scale;
    const float* offset;      // [N]
    const float* rotation;    // [N,4]
    const float* scaling;     // [N,3]
};

struct Vec3 {
    float x, y, z;
};
__device__ __forceinline__ Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ Vec3 add(Vec3 a, Vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ Vec3 mul(Vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot3(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ Vec3 cross3(Vec3 a, Vec3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ Vec3 load3(const float* p, size_t i) { return {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; }

constexpr float kLenEps = 1e-20f;  // mesh_compute.py:17

// x / sqrt(max(x.x, eps)); `clamped` tells the backward that the length did not depend on x
__device__ __forceinline__ Vec3 safe_normalize(Vec3 x, float& len, bool& clamped)
{
    const float d = dot3(x, x);
    clamped = !(d > kLenEps);
    len = sqrtf(fmaxf(d, kLenEps));
    return {x.x / len, x.y / len, x.z / len};
}
// gradient of y = x / len w.r.t. x
__device__ __forceinline__ Vec3 safe_normalize_bwd(Vec3 g, Vec3 y, float len, bool clamped)
{
    if (clamped) return {g.x / len, g.y / len, g.z / len};
    const float t = dot3(y, g);
    return {(g.x - y.x * t) / len, (g.y - y.y * t) / len, (g.z - y.z * t) / len};
}

struct FaceFrame {
    Vec3 e1, e2, a0, a1, a2, c1, c2;
    float l1, lc1, lc2, d, scale;
    bool k1, kc1, kc2;  // clamps
};

__device__ __forceinline__ FaceFrame face_frame(Vec3 v0, Vec3 v1, Vec3 v2)
{
    FaceFrame f;
    f.e1 = sub(v1, v0), f.e2 = sub(v2, v0);
    f.a0 = safe_normalize(f.e1, f.l1, f.k1);
    f.c1 = cross3(f.a0, f.e2);
    f.a1 = safe_normalize(f.c1, f.lc1, f.kc1);
    f.c2 = cross3(f.a1, f.a0);
    const Vec3 y = safe_normalize(f.c2, f.lc2, f.kc2);
    f.a2 = {-y.x, -y.y, -y.z};
    f.d = dot3(f.a2, f.e2);
    f.scale = (f.l1 + fabsf(f.d)) / 2.0f;
    return f;
}

// pytorch3d matrix_to_quaternion of R = [a0 a1 a2] (columns); returns the selected candidate index and sign
struct QuatSel {
    float q[4];
    float num[4], den, qa;
    int sel;
    float sgn;
};
__device__ __forceinline__ QuatSel frame_to_quaternion(const FaceFrame& f)
{
    const float m00 = f.a0.x, m10 = f.a0.y, m20 = f.a0.z;
    const float m01 = f.a1.x, m11 = f.a1.y, m21 = f.a1.z;
    const float m02 = f.a2.x, m12 = f.a2.y, m22 = f.a2.z;
    const float x[4] = {1.0f + m00 + m11 + m22, 1.0f + m00 - m11 - m22, 1.0f - m00 + m11 - m22, 1.0f - m00 - m11 + m22};
    float qa[4];
    int sel = 0;
    for (int k = 0; k < 4; k++) {
        qa[k] = x[k] > 0.f ? sqrtf(x[k]) : 0.f;
        if (qa[k] > qa[sel]) sel = k;
    }
    QuatSel s;
    s.sel = sel, s.qa = qa[sel];
    const float diag = qa[sel] * qa[sel];
    switch (sel) {
        case 0: s.num[0] = diag, s.num[1] = m21 - m12, s.num[2] = m02 - m20, s.num[3] = m10 - m01; break;
        case 1: s.num[0] = m21 - m12, s.num[1] = diag, s.num[2] = m10 + m01, s.num[3] = m02 + m20; break;
        case 2: s.num[0] = m02 - m20, s.num[1] = m10 + m01, s.num[2] = diag, s.num[3] = m12 + m21; break;
        default: s.num[0] = m10 - m01, s.num[1] = m20 + m02, s.num[2] = m21 + m12, s.num[3] = diag; break;
    }
    s.den = 2.0f * fmaxf(qa[sel], 0.1f);
    for (int k = 0; k < 4; k++) s.q[k] = s.num[k] / s.den;
    s.sgn = s.q[0] < 0.f ? -1.f : 1.f;
    for (int k = 0; k < 4; k++) s.q[k] *= s.sgn;
    return s;
}

// forward of one Gaussian: what the reference assigns to gaussian._xyz / _rotation / _scaling before render()
__device__ __forceinline__ void bind_one_fwd(const BindArgs& a, int n, float xyz[3], float rot[4], float scl[3])
{
    const int fi = a.face_index[n];
    const int i0 = a.faces[3 * fi], i1 = a.faces[3 * fi + 1], i2 = a.faces[3 * fi + 2];
    const Vec3 v0 = load3(a.verts, i0), v1 = load3(a.verts, i1), v2 = load3(a.verts, i2);
    const FaceFrame f = face_frame(v0, v1, v2);
    // position: barycentric point + shell offset along the (unnormalised) face normal
    const float b0 = a.bary[3 * n], b1 = a.bary[3 * n + 1], b2 = a.bary[3 * n + 2];
    const Vec3 pos = {b0 * v0.x + b1 * v1.x + b2 * v2.x, b0 * v0.y + b1 * v1.y + b2 * v2.y, b0 * v0.z + b1 * v1.z + b2 * v2.z};
    const Vec3 nrm = cross3(f.e1, f.e2);
    const float t = tanhf(a.offset[n]);
    xyz[0] = pos.x + nrm.x * a.shell_len * t;
    xyz[1] = pos.y + nrm.y * a.shell_len * t;
    xyz[2] = pos.z + nrm.z * a.shell_len * t;
    // rotation: face quaternion (x) own quaternion, real part made non-negative
    const QuatSel qs = frame_to_quaternion(f);
    const float aw = qs.q[0], ax = qs.q[1], ay = qs.q[2], az = qs.q[3];
    const float bw = a.rotation[4 * n], bx = a.rotation[4 * n + 1], by = a.rotation[4 * n + 2], bz = a.rotation[4 * n + 3];
    const float o[4] = {aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw};
    const float sg = o[0] < 0.f ? -1.f : 1.f;
    for (int k = 0; k < 4; k++) rot[k] = sg * o[k];
    // scale: log of the face's stretch relative to the canonical mesh
    const float ls = a.resize_scale ? logf(f.scale / a.canon[fi]) : 0.f;
    for (int k = 0; k < 3; k++) scl[k] = a.scaling[3 * n + k] + ls;
}

// where the backward of one Gaussian puts its results; any member may be null
struct BindGrads {
    float* d_verts;     // [V,3] ADDED with float atomics
    float* d_offset;    // [N]   written
    float* d_rotation;  // [N,4] written
    float* d_scaling;   // [N,3] written
};

// a Gaussian without any gradient (culled by the frame): zero rows, nothing for the vertices
__device__ __forceinline__ void bind_one_bwd_zero(int n, const BindGrads& o)
{
    if (o.d_offset) o.d_offset[n] = 0.f;
    if (o.d_rotation)
        for (int k = 0; k < 4; k++) o.d_rotation[4 * n + k] = 0.f;
    if (o.d_scaling)
        for (int k = 0; k < 3; k++) o.d_scaling[3 * n + k] = 0.f;
}

// backward of one Gaussian: gradients of its three bound values in, gradients of offset / rotation / scaling written,
// dL/dverts of its face's three vertices added
__device__ __forceinline__ void bind_one_bwd(const BindArgs& a, int n, const float g_xyz[3], const float g_rot[4],
                                             const float g_scl[3], const BindGrads& o)
{
    const int fi = a.face_index[n];
    const int i0 = a.faces[3 * fi], i1 = a.faces[3 * fi + 1], i2 = a.faces[3 * fi + 2];
    const Vec3 v0 = load3(a.verts, i0), v1 = load3(a.verts, i1), v2 = load3(a.verts, i2);
    const FaceFrame f = face_frame(v0, v1, v2);
    Vec3 de1 = {0, 0, 0}, de2 = {0, 0, 0}, da0 = {0, 0, 0}, da1 = {0, 0, 0}, da2 = {0, 0, 0};
    float dl1 = 0.f;

    // ---- scaling
    float gs = 0.f;
    for (int k = 0; k < 3; k++) {
        const float g = g_scl[k];
        if (o.d_scaling) o.d_scaling[3 * n + k] = g;
        gs += g;
    }
    if (a.resize_scale) {
        const float dscale = gs / f.scale;           // d log(scale / canon) / d scale
        dl1 += 0.5f * dscale;
        const float dd = 0.5f * dscale * (f.d < 0.f ? -1.f : (f.d > 0.f ? 1.f : 0.f));
        da2 = add(da2, mul(f.e2, dd));
        de2 = add(de2, mul(f.a2, dd));
    }

    // ---- position
    const Vec3 gx = {g_xyz[0], g_xyz[1], g_xyz[2]};
    const Vec3 nrm = cross3(f.e1, f.e2);
    const float t = tanhf(a.offset[n]);
    if (o.d_offset) o.d_offset[n] = a.shell_len * dot3(gx, nrm) * (1.0f - t * t);
    const Vec3 dnrm = mul(gx, a.shell_len * t);
    de1 = add(de1, cross3(f.e2, dnrm));              // d(e1 x e2)/de1 . g = e2 x g
    de2 = add(de2, cross3(dnrm, f.e1));
    const float b0 = a.bary[3 * n], b1 = a.bary[3 * n + 1], b2 = a.bary[3 * n + 2];
    Vec3 dv0 = mul(gx, b0), dv1 = mul(gx, b1), dv2 = mul(gx, b2);

    // ---- rotation: out = sg * (qf (x) r)
    const QuatSel qs = frame_to_quaternion(f);
    const float aw = qs.q[0], ax = qs.q[1], ay = qs.q[2], az = qs.q[3];
    const float bw = a.rotation[4 * n], bx = a.rotation[4 * n + 1], by = a.rotation[4 * n + 2], bz = a.rotation[4 * n + 3];
    const float ow = aw * bw - ax * bx - ay * by - az * bz;
    const float sg = ow < 0.f ? -1.f : 1.f;
    float g[4];
    for (int k = 0; k < 4; k++) g[k] = sg * g_rot[k];
    const float gw = g[0], gxq = g[1], gy = g[2], gz = g[3];
    if (o.d_rotation) {
        o.d_rotation[4 * n] = gw * aw + gxq * ax + gy * ay + gz * az;
        o.d_rotation[4 * n + 1] = -gw * ax + gxq * aw + gy * az - gz * ay;
        o.d_rotation[4 * n + 2] = -gw * ay - gxq * az + gy * aw + gz * ax;
        o.d_rotation[4 * n + 3] = -gw * az + gxq * ay - gy * ax + gz * aw;
    }
    float dq[4] = {gw * bw + gxq * bx + gy * by + gz * bz, -gw * bx + gxq * bw - gy * bz + gz * by,
                   -gw * by + gxq * bz + gy * bw - gz * bx, -gw * bz - gxq * by + gy * bx + gz * bw};
    // through standardize + the selected candidate row: q_k = sgn * num_k / den
    float dnum[4], dden = 0.f;
    for (int k = 0; k < 4; k++) {
        dq[k] *= qs.sgn;
        dnum[k] = dq[k] / qs.den;
        dden -= dq[k] * qs.num[k] / (qs.den * qs.den);
    }
    float dqa = (qs.qa > 0.1f) ? 2.0f * dden : 0.f;   // den = 2 max(qa, 0.1)
    dqa += 2.0f * qs.qa * dnum[qs.sel];               // diagonal numerator qa^2
    const float dx = qs.qa > 0.f ? dqa / (2.0f * qs.qa) : 0.f;   // qa = sqrt(x), zero subgradient at x <= 0
    // x_sel = 1 +- m00 +- m11 +- m22
    const float s00 = (qs.sel == 0 || qs.sel == 1) ? 1.f : -1.f;
    const float s11 = (qs.sel == 0 || qs.sel == 2) ? 1.f : -1.f;
    const float s22 = (qs.sel == 0 || qs.sel == 3) ? 1.f : -1.f;
    float dm[3][3] = {{s00 * dx, 0, 0}, {0, s11 * dx, 0}, {0, 0, s22 * dx}};  // dm[r][c]
    switch (qs.sel) {
        case 0:  // num = (diag, m21 - m12, m02 - m20, m10 - m01)
            dm[2][1] += dnum[1], dm[1][2] -= dnum[1], dm[0][2] += dnum[2], dm[2][0] -= dnum[2], dm[1][0] += dnum[3], dm[0][1] -= dnum[3];
            break;
        case 1:  // (m21 - m12, diag, m10 + m01, m02 + m20)
            dm[2][1] += dnum[0], dm[1][2] -= dnum[0], dm[1][0] += dnum[2], dm[0][1] += dnum[2], dm[0][2] += dnum[3], dm[2][0] += dnum[3];
            break;
        case 2:  // (m02 - m20, m10 + m01, diag, m12 + m21)
            dm[0][2] += dnum[0], dm[2][0] -= dnum[0], dm[1][0] += dnum[1], dm[0][1] += dnum[1], dm[1][2] += dnum[3], dm[2][1] += dnum[3];
            break;
        default:  // (m10 - m01, m20 + m02, m21 + m12, diag)
            dm[1][0] += dnum[0], dm[0][1] -= dnum[0], dm[2][0] += dnum[1], dm[0][2] += dnum[1], dm[2][1] += dnum[2], dm[1][2] += dnum[2];
            break;
    }
    // m[r][c] = a_c[r]
    da0 = add(da0, Vec3{dm[0][0], dm[1][0], dm[2][0]});
    da1 = add(da1, Vec3{dm[0][1], dm[1][1], dm[2][1]});
    da2 = add(da2, Vec3{dm[0][2], dm[1][2], dm[2][2]});

    // ---- face frame (mesh_compute.py:45-47), last to first
    {   // a2 = -normalize(c2), c2 = a1 x a0
        const Vec3 y = {-f.a2.x, -f.a2.y, -f.a2.z};
        const Vec3 dc2 = safe_normalize_bwd(Vec3{-da2.x, -da2.y, -da2.z}, y, f.lc2, f.kc2);
        da1 = add(da1, cross3(f.a0, dc2));
        da0 = add(da0, cross3(dc2, f.a1));
    }
    {   // a1 = normalize(c1), c1 = a0 x e2
        const Vec3 dc1 = safe_normalize_bwd(da1, f.a1, f.lc1, f.kc1);
        da0 = add(da0, cross3(f.e2, dc1));
        de2 = add(de2, cross3(dc1, f.a0));
    }
    {   // a0 = normalize(e1); s0 = length(e1)
        de1 = add(de1, safe_normalize_bwd(da0, f.a0, f.l1, f.k1));
        if (!f.k1) de1 = add(de1, mul(f.a0, dl1));
    }
    dv1 = add(dv1, de1);
    dv2 = add(dv2, de2);
    dv0 = sub(dv0, add(de1, de2));
    if (o.d_verts) {
        atomic_add_f32(o.d_verts + 3 * i0, dv0.x), atomic_add_f32(o.d_verts + 3 * i0 + 1, dv0.y), atomic_add_f32(o.d_verts + 3 * i0 + 2, dv0.z);
        atomic_add_f32(o.d_verts + 3 * i1, dv1.x), atomic_add_f32(o.d_verts + 3 * i1 + 1, dv1.y), atomic_add_f32(o.d_verts + 3 * i1 + 2, dv1.z);
        atomic_add_f32(o.d_verts + 3 * i2, dv2.x), atomic_add_f32(o.d_verts + 3 * i2 + 1, dv2.y), atomic_add_f32(o.d_verts + 3 * i2 + 2, dv2.z);
    }
}

}  // namespace fr
