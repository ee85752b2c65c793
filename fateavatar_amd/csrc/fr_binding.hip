// FateAvatar's mesh binding on gfx950: the step right before the rasterizer.
//
// reference: model/fateavatar.py:225-258 — per frame, every Gaussian n bound to face f = face_index[n] with
// barycentrics bary[n] gets
//     xyz      = sum_k bary_k v_k + (e1 x e2) * shell_len * tanh(offset_n)          (:236-241, :258; the normal is NOT normalised)
//     rotation = standardize(q_face (x) rotation_n),  q_face = matrix_to_quaternion([a0 a1 a2])   (:231-232, :257)
//     scaling  = scaling_n + log(face_scale / face_scale_canonical)                 (:228-229, :256)
// with the face frame / scale of volume_rendering/mesh_compute.py:36-59 and pytorch3d 0.7.7's quaternion conversion
// and product.  The reference evaluates this as ~40 PyTorch kernels over [F] and [N] temporaries (plus their autograd
// twins); here ONE kernel per direction, one thread per Gaussian, recomputing the ~150 flops of its face's frame
// instead of gathering per-face temporaries.  The backward scatters dL/dverts with float atomics (a vertex is shared
// by the Gaussians of ~6 faces x ~10 Gaussians each).  Built without FMA contraction, in the oracle's operation order.
#include "fr_bind_math.hpp"

namespace fr {

__global__ void __launch_bounds__(256) k_bind_fwd(BindArgs a, float* xyz, float* rot_out, float* scale_out)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    float p[3], q[4], s[3];
    bind_one_fwd(a, n, p, q, s);
    for (int k = 0; k < 3; k++) xyz[3 * n + k] = p[k];
    for (int k = 0; k < 4; k++) rot_out[4 * n + k] = q[k];
    for (int k = 0; k < 3; k++) scale_out[3 * n + k] = s[k];
}

__global__ void __launch_bounds__(256) k_bind_bwd(BindArgs a, const float* g_xyz, const float* g_rot, const float* g_scale,
                                                  BindGrads o)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.N) return;
    float gp[3] = {0.f, 0.f, 0.f}, gq[4] = {0.f, 0.f, 0.f, 0.f}, gs[3] = {0.f, 0.f, 0.f};
    if (g_xyz)
        for (int k = 0; k < 3; k++) gp[k] = g_xyz[3 * n + k];
    if (g_rot)
        for (int k = 0; k < 4; k++) gq[k] = g_rot[4 * n + k];
    if (g_scale)
        for (int k = 0; k < 3; k++) gs[k] = g_scale[3 * n + k];
    bind_one_bwd(a, n, gp, gq, gs, o);
}

// per-face scale of a mesh (the canonical one, computed once): mesh_compute.py:51-56
__global__ void __launch_bounds__(256) k_face_scale(int F, const float* verts, const int* faces, float* out)
{
    const int fi = blockIdx.x * blockDim.x + threadIdx.x;
    if (fi >= F) return;
    const FaceFrame f = face_frame(load3(verts, faces[3 * fi]), load3(verts, faces[3 * fi + 1]), load3(verts, faces[3 * fi + 2]));
    out[fi] = f.scale;
}

BindArgs bind_args(const fr_binding& b)
{
    BindArgs a;
    a.N = b.N, a.verts = b.verts, a.faces = b.faces, a.face_index = b.face_index, a.bary = b.bary;
    a.canon = b.face_scale_canonical, a.shell_len = b.shell_len, a.resize_scale = b.resize_scale;
    a.offset = b.offset, a.rotation = b.rotation, a.scaling = b.scaling;
    return a;
}

int launch_bind_forward(const fr_binding& b, float* xyz, float* rot, float* scale, hipStream_t s)
{
    if (b.N <= 0) return FR_OK;
    hipLaunchKernelGGL(k_bind_fwd, dim3((b.N + 255) / 256), dim3(256), 0, s, bind_args(b), xyz, rot, scale);
    FR_HIP(hipGetLastError());
    return FR_OK;
}

int launch_bind_backward(const fr_binding& b, const float* g_xyz, const float* g_rot, const float* g_scale, float* d_verts,
                         float* d_offset, float* d_rotation, float* d_scaling, hipStream_t s)
{
    if (b.N <= 0) return FR_OK;
    hipLaunchKernelGGL(k_bind_bwd, dim3((b.N + 255) / 256), dim3(256), 0, s, bind_args(b), g_xyz, g_rot, g_scale,
                       BindGrads{d_verts, d_offset, d_rotation, d_scaling});
    FR_HIP(hipGetLastError());
    return FR_OK;
}

int launch_face_scale(int F, const float* verts, const int* faces, float* out, hipStream_t s)
{
    if (F <= 0) return FR_OK;
    hipLaunchKernelGGL(k_face_scale, dim3((F + 255) / 256), dim3(256), 0, s, F, verts, faces, out);
    FR_HIP(hipGetLastError());
    return FR_OK;
}

}  // namespace fr
