"""FateAvatar's mesh binding as ONE fused op per direction.

reference: model/fateavatar.py:225-258 — face frame / scale / normal of the posed mesh
(volume_rendering/mesh_compute.py:27-59), barycentric point (volume_rendering/mesh_sampling.py:171-200),
pytorch3d's matrix_to_quaternion and quaternion_multiply, then
    gaussian._scaling  = _scaling + log(face_scale / face_scale_canonical)      (:256, cfg_model.resize_scale)
    gaussian._rotation = quaternion_multiply(face_quaternion, _rotation)         (:257)
    gaussian._xyz      = position + face_normal * shell_len * tanh(_offset)      (:258)
About forty PyTorch kernels (and their autograd twins) there; `bind_gaussians` is one HIP kernel forward and one
backward, differentiable w.r.t. verts (the delta blendshapes train through it), offset, rotation and scaling.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _chk(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError(f"bind_gaussians: {name} must be on a HIP device (there is no CPU path)")
    return t.to(dtype).contiguous()


def face_scale(verts: torch.Tensor, faces: torch.Tensor) -> torch.Tensor:
    """compute_face_orientation(..., return_scale=True)[1] (mesh_compute.py:51-56) of one mesh: [F,1].  Used once, on
    the canonical mesh (model/fateavatar.py:84-85)."""
    verts, faces = _chk(verts, torch.float32, "verts"), _chk(faces, torch.int32, "faces")
    out = torch.empty((faces.shape[0], 1), dtype=torch.float32, device=verts.device)
    with torch.cuda.device(verts.device):
        rc = _lib.lib().fr_face_scale(verts.shape[0], faces.shape[0], verts.data_ptr(), faces.data_ptr(), out.data_ptr(),
                                      torch.cuda.current_stream(verts.device).cuda_stream)
    if rc != _lib.FR_OK:
        raise RuntimeError(f"fr_face_scale failed: {_lib.last_error()}")
    return out


def _desc(verts, faces, face_index, bary, canon, offset, rotation, scaling, shell_len, resize_scale):
    b = _lib.fr_binding()
    b.N, b.V, b.F = face_index.shape[0], verts.shape[0], faces.shape[0]
    b.verts, b.faces, b.face_index, b.bary = verts.data_ptr(), faces.data_ptr(), face_index.data_ptr(), bary.data_ptr()
    b.face_scale_canonical = canon.data_ptr() if canon is not None else None
    b.shell_len, b.resize_scale = float(shell_len), int(bool(resize_scale))
    b.offset, b.rotation, b.scaling = offset.data_ptr(), rotation.data_ptr(), scaling.data_ptr()
    return b


class _Bind(torch.autograd.Function):
    @staticmethod
    def forward(ctx, verts, offset, rotation, scaling, faces, face_index, bary, canon, shell_len, resize_scale):
        verts, offset = _chk(verts, torch.float32, "verts"), _chk(offset, torch.float32, "offset")
        rotation, scaling = _chk(rotation, torch.float32, "rotation"), _chk(scaling, torch.float32, "scaling")
        faces, face_index = _chk(faces, torch.int32, "faces"), _chk(face_index, torch.int32, "face_index")
        bary = _chk(bary, torch.float32, "bary_coords")
        canon = _chk(canon, torch.float32, "face_scale_canonical") if canon is not None else None
        N, dev = face_index.shape[0], verts.device
        if verts.dim() != 2 or offset.numel() != N or rotation.shape != (N, 4) or scaling.shape != (N, 3) or bary.shape != (N, 3):
            raise RuntimeError("bind_gaussians: verts [V,3], offset [N,1], rotation [N,4], scaling [N,3], bary [N,3]")
        if resize_scale and (canon is None or canon.numel() != faces.shape[0]):
            raise RuntimeError("bind_gaussians: resize_scale needs face_scale_canonical [F,1]")
        xyz = torch.empty((N, 3), dtype=torch.float32, device=dev)
        rot = torch.empty((N, 4), dtype=torch.float32, device=dev)
        scl = torch.empty((N, 3), dtype=torch.float32, device=dev)
        b = _desc(verts, faces, face_index, bary, canon, offset, rotation, scaling, shell_len, resize_scale)
        with torch.cuda.device(dev):
            rc = _lib.lib().fr_bind_forward(C.byref(b), xyz.data_ptr(), rot.data_ptr(), scl.data_ptr(),
                                            torch.cuda.current_stream(dev).cuda_stream)
        if rc != _lib.FR_OK:
            raise RuntimeError(f"fr_bind_forward failed: {_lib.last_error()}")
        ctx.save_for_backward(verts, offset, rotation, scaling, faces, face_index, bary, canon)
        ctx.consts = (float(shell_len), bool(resize_scale), offset.shape)
        # optional extension (see rasterizer.GradOut): a raw parameter may carry `_fr_grad_out`, a slot whose preallocated
        # buffer (a view into a flat gradient buffer) receives its gradient without a copy
        from .rasterizer import GradOut
        ctx.grad_slots = (GradOut.of(offset), GradOut.of(rotation), GradOut.of(scaling))
        return xyz, rot, scl

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_scl):
        verts, offset, rotation, scaling, faces, face_index, bary, canon = ctx.saved_tensors
        shell_len, resize_scale, offset_shape = ctx.consts
        dev, N = verts.device, face_index.shape[0]
        need_v, need_o, need_r, need_s = ctx.needs_input_grad[:4]
        c = lambda g: g.contiguous().float() if g is not None else None  # noqa: E731
        g_xyz, g_rot, g_scl = c(g_xyz), c(g_rot), c(g_scl)
        d_verts = torch.zeros_like(verts) if need_v else None

        def out(need, slot, shape):
            if not need:
                return None
            buf = slot.claim()[0] if slot is not None else None   # first backward of the step writes the slot in place
            if buf is not None and buf.numel() == int(torch.Size(shape).numel()) and buf.is_contiguous():
                return buf.view(shape)
            return torch.empty(shape, dtype=torch.float32, device=dev)

        d_off = out(need_o, ctx.grad_slots[0], (N,))
        d_rot = out(need_r, ctx.grad_slots[1], (N, 4))
        d_scl = out(need_s, ctx.grad_slots[2], (N, 3))
        p = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        b = _desc(verts, faces, face_index, bary, canon, offset, rotation, scaling, shell_len, resize_scale)
        with torch.cuda.device(dev):
            rc = _lib.lib().fr_bind_backward(C.byref(b), p(g_xyz), p(g_rot), p(g_scl), p(d_verts), p(d_off), p(d_rot), p(d_scl),
                                             torch.cuda.current_stream(dev).cuda_stream)
        if rc != _lib.FR_OK:
            raise RuntimeError(f"fr_bind_backward failed: {_lib.last_error()}")
        return (d_verts, d_off.view(offset_shape) if d_off is not None else None, d_rot, d_scl, None, None, None, None, None,
                None)


def bind_gaussians(verts, faces, face_index, bary_coords, face_scale_canonical, offset, rotation, scaling,
                   shell_len: float, resize_scale: bool = True):
    """One frame of model/fateavatar.py:225-258.  verts [V,3] (posed), faces [F,3], face_index [N], bary_coords [N,3],
    face_scale_canonical [F,1] (`face_scale` of the canonical mesh), raw offset [N,1] / rotation [N,4] / scaling [N,3].
    Returns (xyz [N,3], rotation [N,4], scaling [N,3]): the values the reference assigns to gaussian._xyz /
    gaussian._rotation / gaussian._scaling before render()."""
    return _Bind.apply(verts, offset, rotation, scaling, faces, face_index, bary_coords, face_scale_canonical, shell_len,
                       resize_scale)
