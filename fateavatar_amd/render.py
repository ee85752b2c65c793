"""`render()` — the caller-facing entry of the path (reference: volume_rendering/render_3dgs.py:7-81).

Same signature, same returned dict.  `pc` is anything with the reference GaussianModel's getters
(get_xyz, get_opacity, get_scaling, get_rotation, get_features, max_sh_degree;
volume_rendering/gaussian_model.py:105-128); `viewpoint_camera` anything with FoVx, FoVy, image_height,
image_width, world_view_transform, full_proj_transform, camera_center (volume_rendering/camera_3dgs.py:22-72).
"""
from __future__ import annotations

import math

import torch

from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


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
