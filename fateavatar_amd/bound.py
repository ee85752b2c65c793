"""Frames rendered STRAIGHT FROM THEIR MESH BINDING: `bind_gaussians` + `render` as one differentiable op per batch of
views, with the binding evaluated inside the rasterizer's per-Gaussian kernels (include/fr_rasterizer.h, fr_aux::binding).

reference: model/fateavatar.py:225-276 — per frame the Gaussians are bound to the posed mesh (about forty PyTorch
kernels and their autograd twins, :225-258), assigned to gaussian._xyz / _rotation / _scaling, and rendered
(volume_rendering/render_3dgs.py:7-81).  `binding.bind_gaussians` makes the binding one kernel per direction;
here it is no kernel at all: the preprocess kernel computes a Gaussian's bound position / rotation / log-scale in front of
its own work (and stores them for the backward), and the per-Gaussian backward kernel carries its gradients on through
the binding to offset / rotation / scaling (and the posed vertices).  Same expressions from one header
(csrc/fr_bind_math.hpp), so the results are those of `bind_gaussians` followed by `render_batch`, bit for bit in the
forward and to atomic-summation order in the backward.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

from . import _lib
from .binding import _chk, _desc
from .rasterizer import (NUM_CHANNELS, GaussianRasterizationSettings, GradOut, rasterize_gaussians_backward_batch,
                         rasterize_gaussians_batch)
from .render import _zero_points


class MeshBinding(NamedTuple):
    """What does not change from frame to frame: model/fateavatar.py:120-164 (face_index, bary_coords), :84-85
    (face_scale_canonical), the mesh topology and the two configuration values of :256-258."""
    faces: torch.Tensor                   # [F,3] int32
    face_index: torch.Tensor              # [N]   int32
    bary_coords: torch.Tensor             # [N,3]
    face_scale_canonical: Optional[torch.Tensor]   # [F,1] (`binding.face_scale` of the canonical mesh), None without resize_scale
    shell_len: float
    resize_scale: bool = True


class _RasterizeBoundBatch(torch.autograd.Function):
    """Tensor arguments per view: (verts, offset, rotation, scaling, means2D, sh, opacities) — the RAW parameters, as
    render() hands them over with `fused_activations`.  Outputs per view: (color, radii)."""
    PER_VIEW = 7

    @staticmethod
    def forward(ctx, settings, bindings, slots, *tensors):
        K, n = len(settings), _RasterizeBoundBatch.PER_VIEW
        assert len(tensors) == n * K and len(bindings) == K
        ctx.K, ctx.settings, ctx.bindings, ctx.slots = K, settings, bindings, slots
        ctx.set_materialize_grads(False)
        empty = torch.Tensor([])
        views, viss, descs, bound, checked = [], [], [], [], []
        for k, (rs, mb) in enumerate(zip(settings, bindings)):
            verts, offset, rotation, scaling, means2D, sh, opacities = tensors[n * k:n * k + n]
            verts, offset = _chk(verts, torch.float32, "verts"), _chk(offset, torch.float32, "offset")
            rotation, scaling = _chk(rotation, torch.float32, "rotation"), _chk(scaling, torch.float32, "scaling")
            checked.append((verts, offset, rotation, scaling))   # (what the descriptor points at: alive until the launch, saved for the backward)
            N, dev = mb.face_index.shape[0], verts.device
            if verts.dim() != 2 or offset.numel() != N or rotation.shape != (N, 4) or scaling.shape != (N, 3) or \
                    mb.bary_coords.shape != (N, 3):
                raise RuntimeError("render_bound_batch: verts [V,3], offset [N,1], rotation [N,4], scaling [N,3], bary [N,3]")
            if mb.resize_scale and (mb.face_scale_canonical is None or mb.face_scale_canonical.numel() != mb.faces.shape[0]):
                raise RuntimeError("render_bound_batch: resize_scale needs face_scale_canonical [F,1]")
            descs.append(_desc(verts, mb.faces, mb.face_index, mb.bary_coords, mb.face_scale_canonical, offset, rotation, scaling,
                               mb.shell_len, mb.resize_scale))
            # the bound values: written by the preprocess kernel, read again by the backward
            xyz = torch.empty((N, 3), dtype=torch.float32, device=dev)
            rot = torch.empty((N, 4), dtype=torch.float32, device=dev)
            scl = torch.empty((N, 3), dtype=torch.float32, device=dev)
            bound.append((xyz, rot, scl))
            views.append((rs.bg, xyz, empty, opacities, scl, rot, rs.scale_modifier, empty, rs.viewmatrix, rs.projmatrix,
                          rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh, rs.sh_degree, rs.campos, rs.prefiltered,
                          rs.debug))
            viss.append(torch.empty((N,), dtype=torch.bool, device=dev))
        res = rasterize_gaussians_batch(views, slots=slots, raw=True, visibles=viss, bindings=descs)
        ctx.stats, ctx.num_rendered, ctx.grad_slots, ctx.grad_owners, ctx.offset_shapes = [], [], [], [], []
        saved, outs = [], []
        for k in range(K):
            verts, offset, rotation, scaling, means2D, sh, opacities = tensors[n * k:n * k + n]
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = res[k]
            radii._fr_visible = viss[k]
            radii._fr_bound = bound[k]           # (xyz, rotation, scaling) as bind_gaussians returns them
            ctx.stats.append(getattr(means2D, "_fr_densification_stats", None))
            ctx.num_rendered.append(num_rendered)
            ctx.offset_shapes.append(tuple(offset.shape))
            slots_k = {"dL_dsh": GradOut.of(sh) if sh.numel() else None, "dL_dopacity": GradOut.of(opacities),
                       "d_offset": GradOut.of(offset), "d_rotation": GradOut.of(rotation), "d_scaling": GradOut.of(scaling)}
            owners = {"dL_dsh": sh, "dL_dopacity": opacities, "d_offset": offset, "d_rotation": rotation, "d_scaling": scaling}
            ctx.grad_slots.append(slots_k)
            ctx.grad_owners.append({m: t for m, t in owners.items() if slots_k.get(m) is not None and t.is_leaf})
            saved += [*checked[k], sh, radii, geomBuffer, binningBuffer, imgBuffer, *bound[k]]
            outs += [color, radii]
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(*outs[1::2])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        K, n = ctx.K, _RasterizeBoundBatch.PER_VIEW
        none = (None, None, None) + (None,) * (n * K)
        grad_colors = grad_outs[0::2]
        if all(g is None for g in grad_colors):
            return none
        empty = torch.Tensor([])
        views, wants, outs, descs, bgrads = [], [], [], [], []
        for k, (rs, mb) in enumerate(zip(ctx.settings, ctx.bindings)):
            verts, offset, rotation, scaling, sh, radii, geomBuffer, binningBuffer, imgBuffer, xyz, rot, scl = \
                ctx.saved_tensors[12 * k:12 * k + 12]
            dev, N = verts.device, xyz.shape[0]
            g = grad_colors[k]
            if g is None:
                g = torch.zeros((NUM_CHANNELS, rs.image_height, rs.image_width), dtype=torch.float32, device=dev)
            views.append((rs.bg, xyz, radii, empty, scl, rot, rs.scale_modifier, empty, rs.viewmatrix, rs.projmatrix, rs.tanfovx,
                          rs.tanfovy, g, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered[k], binningBuffer, imgBuffer,
                          rs.debug))
            want = {"dL_dmeans2D", "dL_dopacity"}
            if sh.numel():
                want.add("dL_dsh")
            claims = {m: slot.claim(ctx.grad_owners[k].get(m)) for m, slot in ctx.grad_slots[k].items() if slot is not None}
            wants.append(want)
            outs.append({m: c[0] for m, c in claims.items() if not c[1] and m.startswith("dL_")})
            need_v, need_o, need_r, need_s = ctx.needs_input_grad[3 + n * k:3 + n * k + 4]

            def buf(need, name, shape):
                if not need:
                    return None
                c = claims.get(name)
                b = c[0] if (c is not None and not c[1]) else None
                if b is not None and b.numel() == int(torch.Size(shape).numel()) and b.is_contiguous():
                    return b.view(shape)
                return torch.empty(shape, dtype=torch.float32, device=dev)

            descs.append(_desc(verts, mb.faces, mb.face_index, mb.bary_coords, mb.face_scale_canonical, offset, rotation, scaling,
                               mb.shell_len, mb.resize_scale))
            bgrads.append({"d_verts": torch.zeros_like(verts) if need_v else None, "d_offset": buf(need_o, "d_offset", (N,)),
                           "d_rotation": buf(need_r, "d_rotation", (N, 4)), "d_scaling": buf(need_s, "d_scaling", (N, 3))})
        res = rasterize_gaussians_backward_batch(views, slots=ctx.slots, raw=True, wants=wants, outs=outs, stats=ctx.stats,
                                                 bindings=descs, bind_grads=bgrads)
        flat = [None, None, None]
        for k in range(K):
            grad_means2D, _, grad_opacities, _, _, grad_sh, _, _ = res[k]
            b = bgrads[k]
            d_off = b["d_offset"].view(ctx.offset_shapes[k]) if b["d_offset"] is not None else None
            # (fresh view objects: AccumulateGrad adopts a gradient without a copy only if nobody else references it)
            fresh = lambda t: t.view(t.shape) if t is not None else None  # noqa: E731
            flat += [b["d_verts"], d_off, fresh(b["d_rotation"]), fresh(b["d_scaling"]), grad_means2D, grad_sh, grad_opacities]
        return tuple(flat)


def render_bound_batch(viewpoint_cameras, pcs, posed_verts, binding: MeshBinding, bg_colors, scaling_modifier=1.0, slots=None):
    """`bind_gaussians` + `render_batch` for K views in one launch chain without binding kernels.

    `pcs`: per view (or one for all) a holder with the raw parameters `_opacity` [N,1], `_offset` [N,1], `_rotation` [N,4],
    `_scaling` [N,3], the features `get_features` [N,M,3], `max_sh_degree` — and optionally `fused_densification_stats`;
    `posed_verts`: per view the posed mesh [V,3].  Returns the list of render() dicts; `out["bound"]` holds the
    (xyz, rotation, scaling) the reference assigns to the Gaussians before render() (model/fateavatar.py:256-258) as plain
    kernel OUTPUTS, DETACHED from autograd: gradients reach offset / rotation / scaling / verts through the image only.  A
    regulariser on the bound values themselves needs the differentiable stand-alone op (`binding.bind_gaussians`, what
    `AvatarStep(fold_binding=False)` renders through)."""
    K = len(viewpoint_cameras)
    if not 1 <= K <= _lib.FR_MAX_BATCH:
        raise RuntimeError(f"render_bound_batch: 1 .. {_lib.FR_MAX_BATCH} views")
    if not isinstance(pcs, (list, tuple)):
        pcs = [pcs] * K
    if isinstance(bg_colors, torch.Tensor):
        bg_colors = [bg_colors] * K
    if isinstance(posed_verts, torch.Tensor):
        posed_verts = [posed_verts] * K
    mb = MeshBinding(_chk(binding.faces, torch.int32, "faces"), _chk(binding.face_index, torch.int32, "face_index"),
                     _chk(binding.bary_coords, torch.float32, "bary_coords"),
                     _chk(binding.face_scale_canonical, torch.float32, "face_scale_canonical")
                     if binding.face_scale_canonical is not None else None, float(binding.shell_len), bool(binding.resize_scale))
    settings, tensors, points = [], [], []
    for cam, pc, bg, verts in zip(viewpoint_cameras, pcs, bg_colors, posed_verts):
        sp = _zero_points(pc._scaling)
        stats = getattr(pc, "fused_densification_stats", None)
        if stats is not None:
            sp._fr_densification_stats = stats
        try:
            sp.retain_grad()
        except Exception:
            pass
        settings.append(GaussianRasterizationSettings(
            image_height=int(cam.image_height), image_width=int(cam.image_width), tanfovx=math.tan(cam.FoVx * 0.5),
            tanfovy=math.tan(cam.FoVy * 0.5), bg=bg, scale_modifier=scaling_modifier, viewmatrix=cam.world_view_transform,
            projmatrix=cam.full_proj_transform, sh_degree=pc.max_sh_degree, campos=cam.camera_center, prefiltered=False,
            debug=False))
        tensors += [verts, pc._offset, pc._rotation, pc._scaling, sp, pc.get_features, pc._opacity]
        points.append(sp)
    res = _RasterizeBoundBatch.apply(settings, [mb] * K, list(range(K)) if slots is None else list(slots), *tensors)
    out = []
    for k, sp in enumerate(points):
        image, radii = res[2 * k], res[2 * k + 1]
        out.append({"render": image, "viewspace_points": sp, "visibility_filter": radii._fr_visible, "radii": radii,
                    "bound": tuple(t.detach() for t in radii._fr_bound) if radii._fr_bound is not None else None})
    return out
