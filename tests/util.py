"""Shared helpers of the parity tests: run a scene through the CPU oracle and through the HIP path
(via the `_C`-level mirror, i.e. through the C ABI) and compare."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from fateavatar_amd import _lib, rasterizer
from fateavatar_amd.scenes import GaussianScene
from oracle import oracle


def oracle_forward(s: GaussianScene, colors_precomp=None, cov3D_precomp=None, scale_modifier=1.0):
    c = s.camera
    return oracle.forward(
        bg=s.bg, means3D=s.means3D, opacities=s.opacities, viewmatrix=c.world_view_transform,
        projmatrix=c.full_proj_transform, campos=c.camera_center, tanfovx=c.tanfovx, tanfovy=c.tanfovy,
        H=c.image_height, W=c.image_width, shs=None if colors_precomp is not None else s.shs,
        sh_degree=s.sh_degree, colors_precomp=colors_precomp, scales=None if cov3D_precomp is not None else s.scales,
        rotations=None if cov3D_precomp is not None else s.rotations, cov3D_precomp=cov3D_precomp,
        scale_modifier=scale_modifier)


class HipFrame:
    """One forward (+ optional backward) through the C ABI."""

    def __init__(self, s: GaussianScene, dev, colors_precomp=None, cov3D_precomp=None, scale_modifier=1.0, debug=False):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        e = torch.empty(0)
        c = s.camera
        self.s, self.dev = s, dev
        self.bg, self.means3D, self.op = t(s.bg), t(s.means3D), t(s.opacities)
        self.colors = t(colors_precomp) if colors_precomp is not None else e
        self.sh = e if colors_precomp is not None else t(s.shs)
        self.cov = t(cov3D_precomp) if cov3D_precomp is not None else e
        self.scales = e if cov3D_precomp is not None else t(s.scales)
        self.rots = e if cov3D_precomp is not None else t(s.rotations)
        self.view, self.proj, self.campos = t(c.world_view_transform), t(c.full_proj_transform), t(c.camera_center)
        self.mod, self.debug = scale_modifier, debug
        self.H, self.W = c.image_height, c.image_width
        (self.num_rendered, self.color, self.radii, self.geom, self.binning, self.img) = rasterizer.rasterize_gaussians(
            self.bg, self.means3D, self.colors, self.op, self.scales, self.rots, scale_modifier, self.cov, self.view,
            self.proj, c.tanfovx, c.tanfovy, self.H, self.W, self.sh, s.sh_degree, self.campos, False, debug)
        self.counts = rasterizer.last_counts[dev.index or 0]
        self.final_T, self.n_contrib = rasterizer.image_aux(self.img, self.H, self.W)

    def geometry(self, field: int, cols: int, dtype=torch.float32):
        L = _lib.lib()
        P = self.means3D.shape[0]
        base = self.geom.data_ptr()
        off = L.fr_debug_geometry_field(base, P, field) - base
        nbytes = P * cols * torch.empty(0, dtype=dtype).element_size()
        return self.geom[off:off + nbytes].view(dtype).view(P, cols).cpu().numpy()

    def backward(self, dL_dpix: np.ndarray):
        c = self.s.camera
        g = torch.from_numpy(np.ascontiguousarray(dL_dpix)).to(self.dev)
        out = rasterizer.rasterize_gaussians_backward(
            self.bg, self.means3D, self.radii, self.colors, self.scales, self.rots, self.mod, self.cov, self.view,
            self.proj, c.tanfovx, c.tanfovy, g, self.sh, self.s.sh_degree, self.campos, self.geom, self.num_rendered,
            self.binning, self.img, self.debug)
        names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
                 "dL_drotations"]
        return {k: v.cpu().numpy() for k, v in zip(names, out)}


def frac_close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 1.0
    return float(np.mean(np.abs(a - b) <= atol + rtol * np.abs(b)))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a - b))
