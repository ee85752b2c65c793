"""`render()` — the caller-facing entry of the path (reference: volume_rendering/render_3dgs.py:7-81).

Same signature, same returned dict.  `pc` is anything with the reference GaussianModel's getters
(get_xyz, get_opacity, get_scaling, get_rotation, get_features, max_sh_degree;
volume_rendering/gaussian_model.py:105-128); `viewpoint_camera` anything with FoVx, FoVy, image_height,
image_width, world_view_transform, full_proj_transform, camera_center (volume_rendering/camera_3dgs.py:22-72).
"""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views_autograd


_zero_cache = {}


def _zero_points(means3D: torch.Tensor) -> torch.Tensor:
    """A fresh leaf tensor of zeros shaped like means3D that requires grad.  Nothing ever writes its values (the
    rasterizer only uses it as the slot whose .grad receives dL/dmeans2D), so the zeros themselves are a cached
    constant per (shape, dtype, device) and every call returns a new leaf aliasing them: no fill kernel per frame."""
    key = (tuple(means3D.shape), means3D.dtype, means3D.device)
    z = _zero_cache.get(key)
    if z is None:
        if len(_zero_cache) > 8:
            _zero_cache.clear()
        z = _zero_cache[key] = torch.zeros_like(means3D, requires_grad=False)
    return z.detach().requires_grad_(True)


def render(viewpoint_camera, pc, bg_color: torch.Tensor, scaling_modifier=1.0, override_color: torch.Tensor = None,
           device='cuda'):
    means3D = pc.get_xyz
    # zero tensor whose .grad receives the screen-space mean gradients (render_3dgs.py:21-27).  The reference adds
    # `+ 0` and retains the gradient of the resulting non-leaf; a leaf collects the same .grad with one kernel less
    # and lets autograd adopt the rasterizer's gradient buffer instead of copying it.
    screenspace_points = _zero_points(means3D)
    stats = getattr(pc, "fused_densification_stats", None)
    if stats is not None:  # extension: (xyz_gradient_accum, denom) updated inside the backward kernel
        screenspace_points._fr_densification_stats = stats
    if screenspace_points.requires_grad:
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass

    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height),
        image_width=int(viewpoint_camera.image_width),
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg_color,
        scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.max_sh_degree,
        campos=viewpoint_camera.camera_center,
        prefiltered=False,
        debug=False,
    )
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)
    means2D = screenspace_points
    # Extension: a holder that sets `fused_activations` hands over its RAW opacity / scaling /
    # rotation and the HIP kernels apply sigmoid / exp / normalize (and their derivatives) themselves, instead of
    # ~15 small PyTorch kernels per frame.  The default is the reference behaviour.
    fused = bool(getattr(pc, "fused_activations", False))
    if fused:
        opacity, scales, rotations = pc._opacity, pc._scaling, pc._rotation
    else:
        opacity = pc.get_opacity
        scales = pc.get_scaling
        rotations = pc.get_rotation
    cov3D_precomp = None
    shs = pc.get_features
    if override_color is None:
        colors_precomp = None
    else:
        colors_precomp = override_color
        shs = None
    rendered_image, radii = rasterizer(means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp,
                                       opacities=opacity, scales=scales, rotations=rotations,
                                       cov3D_precomp=cov3D_precomp, **({"raw_activations": True} if fused else {}))
    visible = getattr(radii, "_fr_visible", None)  # written by the preprocess kernel (same values as radii > 0)
    return {"render": rendered_image, "viewspace_points": screenspace_points,
            "visibility_filter": visible if visible is not None else radii > 0, "radii": radii}


def render_batch(viewpoint_cameras, pcs, bg_colors, scaling_modifier=1.0, slots=None):
    """`render()` for K views IN ONE LAUNCH CHAIN (include/fr_rasterizer.h: fr_forward_batch / fr_backward_batch): the
    reference renders the frames of a batch one after the other (model/fateavatar.py:251-276), and one frame's kernels
    leave most of an MI355X idle; here every kernel of the frame is launched once for all K views, with no stream or
    hardware-queue arrangement on the caller's side.  `pcs` / `bg_colors`: one per view, or a single holder / tensor
    for all of them (shared Gaussians: autograd then sums the views' gradients).  Returns the list of render() dicts."""
    import torch as _torch
    K = len(viewpoint_cameras)
    if not isinstance(pcs, (list, tuple)):
        pcs = [pcs] * K
    if isinstance(bg_colors, _torch.Tensor):
        bg_colors = [bg_colors] * K
    settings, tensors, points = [], [], []
    fused = bool(getattr(pcs[0], "fused_activations", False))
    empty = _torch.Tensor([])
    for cam, pc, bg in zip(viewpoint_cameras, pcs, bg_colors):
        if bool(getattr(pc, "fused_activations", False)) != fused:
            raise RuntimeError("render_batch: the views' holders must agree on fused_activations")
        means3D = pc.get_xyz
        sp = _zero_points(means3D)
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
        if fused:
            opacity, scales, rotations = pc._opacity, pc._scaling, pc._rotation
        else:
            opacity, scales, rotations = pc.get_opacity, pc.get_scaling, pc.get_rotation
        tensors.append((means3D, sp, pc.get_features, empty, opacity, scales, rotations, empty))
        points.append(sp)
    res = rasterize_views_autograd(settings, tensors, raw_activations=fused, slots=slots)
    out = []
    for (image, radii), sp in zip(res, points):
        visible = getattr(radii, "_fr_visible", None)
        out.append({"render": image, "viewspace_points": sp,
                    "visibility_filter": visible if visible is not None else radii > 0, "radii": radii})
    return out
