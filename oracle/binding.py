"""TEST INFRASTRUCTURE ONLY — CPU restatement (stock PyTorch, any dtype) of FateAvatar's mesh binding: the step that
turns posed mesh vertices + per-Gaussian binding (face index, barycentrics) + raw per-Gaussian parameters into the
Gaussians handed to the rasterizer.  Only tests may import this.

reference: model/fateavatar.py:225-258 (forward_train), built from
  compute_face_orientation / compute_face_normals   volume_rendering/mesh_compute.py:27-59
  reweight_verts_by_barycoords                       volume_rendering/mesh_sampling.py:171-200
  matrix_to_quaternion / quaternion_multiply         pytorch3d 0.7.7 (README.md:40) — NOT in /root/reference and not
      installed here: restated from the published algorithm (pytorch3d/transforms/rotation_conversions.py:
      _sqrt_positive_part, the four-candidate division with floor 0.1, argmax selection, standardize_quaternion;
      quaternion_raw_multiply + standardize_quaternion).  Parity of these two is therefore UNPINNED; the
      mesh_compute / mesh_sampling parts are pinned by tests/golden/golden_binding.npz, generated from the
      reference's own functions (tests/golden/make_golden.py).
"""
import torch


def _length(x, eps=1e-20):
    return torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))  # mesh_compute.py:17-18


def _safe_normalize(x, eps=1e-20):
    return x / _length(x, eps)  # mesh_compute.py:20-21


def face_orientation(verts, faces):
    """compute_face_orientation(..., return_scale=True), mesh_compute.py:36-59.  verts [V,3], faces [F,3] ->
    orientation [F,3,3] (columns a0, a1, a2), scale [F,1]."""
    i0, i1, i2 = faces[:, 0].long(), faces[:, 1].long(), faces[:, 2].long()
    v0, v1, v2 = verts[i0], verts[i1], verts[i2]
    a0 = _safe_normalize(v1 - v0)
    a1 = _safe_normalize(torch.cross(a0, v2 - v0, dim=-1))
    a2 = -_safe_normalize(torch.cross(a1, a0, dim=-1))
    orientation = torch.cat([a0[..., None], a1[..., None], a2[..., None]], dim=-1)
    s0 = _length(v1 - v0)
    s1 = (a2 * (v2 - v0)).sum(-1, keepdim=True).abs()
    return orientation, (s0 + s1) / 2


def face_normals(verts, faces):
    """compute_face_normals, mesh_compute.py:26-34 (not normalised)."""
    i0, i1, i2 = faces[:, 0].long(), faces[:, 1].long(), faces[:, 2].long()
    return torch.cross(verts[i1] - verts[i0], verts[i2] - verts[i0], dim=-1)


def standardize_quaternion(q):
    return torch.where(q[..., 0:1] < 0, -q, q)


def matrix_to_quaternion(m):
    """pytorch3d.transforms.matrix_to_quaternion (0.7.7).  m [...,3,3] -> [...,4] (r, i, j, k), real part >= 0."""
    m00, m01, m02 = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    m10, m11, m12 = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    m20, m21, m22 = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    x = torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22, 1.0 - m00 - m11 + m22], dim=-1)
    q_abs = torch.where(x > 0, torch.sqrt(torch.clamp(x, min=0)), torch.zeros_like(x))   # _sqrt_positive_part
    by_rijk = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = by_rijk / (2.0 * torch.clamp(q_abs[..., None], min=0.1))
    sel = q_abs.argmax(dim=-1)
    out = torch.gather(cand, -2, sel[..., None, None].expand(*sel.shape, 1, 4)).squeeze(-2)
    return standardize_quaternion(out)


def quaternion_multiply(a, b):
    """pytorch3d.transforms.quaternion_multiply: Hamilton product, then real part >= 0."""
    aw, ax, ay, az = a.unbind(-1)
    bw, bx, by, bz = b.unbind(-1)
    o = torch.stack([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw], dim=-1)
    return standardize_quaternion(o)


def bind(verts, faces, face_index, bary, face_scale_canonical, offset, rotation, scaling, shell_len, resize_scale=True):
    """model/fateavatar.py:225-258 for ONE frame.  verts [V,3]; faces [F,3]; face_index [N]; bary [N,3];
    face_scale_canonical [F,1]; offset [N,1]; rotation [N,4]; scaling [N,3] (raw parameters).
    Returns (xyz [N,3], rotation [N,4], scaling [N,3]) — what the reference assigns to gaussian._xyz / _rotation /
    _scaling before render()."""
    orien, fscale = face_orientation(verts, faces)
    fi = face_index.long()
    ratio = (fscale / face_scale_canonical)[fi]                      # :228-229
    quat = matrix_to_quaternion(orien[fi])                           # :231-232
    nrm = face_normals(verts, faces)[fi]                             # :226,233
    tri = verts[faces[fi].long()]                                    # [N,3,3]
    pos = (bary[..., None] * tri).sum(-2)                            # mesh_sampling.py:171-200
    out_scaling = scaling + torch.log(ratio) if resize_scale else scaling   # :256
    out_rotation = quaternion_multiply(quat, rotation)               # :257
    xyz = pos + nrm * shell_len * torch.tanh(offset)                 # :258
    return xyz, out_rotation, out_scaling
