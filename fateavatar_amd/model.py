"""Minimal Gaussian parameter holder with the reference GaussianModel's getters
(volume_rendering/gaussian_model.py:39-50,105-128): raw parameters + exp / sigmoid / normalize
activations in stock PyTorch.  It exists so that `render()` can be driven exactly like the
reference drives it; densification, PLY I/O and optimizer surgery are out of scope.

All parameters live in ONE flat fp32 buffer (and their gradients in one flat buffer), so the
data-parallel exchange is a single all-reduce (fateavatar_amd/dp.py).
"""
from __future__ import annotations

import numpy as np
import torch

from .rasterizer import GradOut


class FlatGaussians(torch.nn.Module):
    FIELDS = (("_xyz", 3), ("_features", None), ("_opacity", 1), ("_scaling", 3), ("_rotation", 4))

    def __init__(self, means3D, shs, opacities, scales, rotations, sh_degree: int, device, fused_activations=False):
        """Arguments are ACTIVATED values (numpy): they are inverted into raw parameters like
        create_from_pcd does (gaussian_model.py:137-160)."""
        super().__init__()
        P, M = means3D.shape[0], shs.shape[1]
        self.max_sh_degree = sh_degree
        self.P, self.M = P, M
        # True: render() passes the raw parameters and the rasterizer kernels apply the activations themselves
        self.fused_activations = bool(fused_activations)
        # one flat value buffer and one flat gradient buffer; every parameter (and its .grad) is a VIEW into
        # them, so autograd accumulates in place and the data-parallel exchange is a single all-reduce
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
        op = t(opacities).reshape(-1).clamp(1e-6, 1 - 1e-6)
        self._bind([t(means3D), t(shs), torch.log(op / (1 - op)).reshape(-1, 1), torch.log(t(scales)), t(rotations)])

    @classmethod
    def from_raw(cls, xyz, features, opacity, scaling, rotation, sh_degree: int, device, fused_activations=False):
        """From RAW parameters (logit opacity, log scale, un-normalised quaternion), e.g. `ply.load_ply()`."""
        self = cls.__new__(cls)
        torch.nn.Module.__init__(self)
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(device)  # noqa: E731
        self.max_sh_degree, self.M = sh_degree, int(np.shape(features)[1])
        self.fused_activations = bool(fused_activations)
        self._bind([t(xyz), t(features), t(opacity).reshape(-1, 1), t(scaling), t(rotation)])
        return self

    def save_ply(self, path: str) -> None:
        """GaussianModel.save_ply (gaussian_model.py:205-223)."""
        from . import ply
        g = lambda n: getattr(self, n).detach().cpu().numpy()  # noqa: E731
        ply.save_ply(path, g("_xyz"), g("_features"), g("_opacity"), g("_scaling"), g("_rotation"))

    @classmethod
    def load_ply(cls, path: str, device, max_sh_degree=None, fused_activations=False):
        """GaussianModel.load_ply (gaussian_model.py:230-269)."""
        from . import ply
        d = ply.load_ply(path, max_sh_degree)
        return cls.from_raw(d["xyz"], d["features"], d["opacity"], d["scaling"], d["rotation"], d["sh_degree"], device,
                            fused_activations)

    def widths(self):
        """Floats per Gaussian of each field, in flat-buffer order."""
        return [3, self.M * 3, 1, 3, 4]

    def _bind(self, raw):
        """(Re)build the flat value / gradient buffers from one raw tensor per field ([P, ...] each) and make every
        parameter, and its gradient slot, a view into them."""
        P = raw[0].shape[0]
        device = raw[0].device
        self.P = P
        sizes = [P * w for w in self.widths()]
        shapes = [(P, 3), (P, self.M, 3), (P, 1), (P, 3), (P, 4)]
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
        # the gradient buffer, and behind it (same allocation, so that ONE all-reduce carries both) the step's OVERFLOW WORD:
        # the rasterizer's backward sets it to 1 when its frame overflowed the binning capacity inside a replayed graph (all
        # its gradients are zero then), and the fused Adam skips a step whose word — summed over lanes and ranks — is not 0
        self._grad_store = torch.zeros(sum(sizes) + 4, dtype=torch.float32, device=device)
        self.flat_grad = self._grad_store[:sum(sizes)]
        self.overflow_word = self._grad_store[sum(sizes):sum(sizes) + 1]
        self._grad_views = {}
        off = 0
        for (name, _), n, shp, r in zip(self.FIELDS, sizes, shapes, raw):
            self.flat[off:off + n].copy_(r.detach().reshape(-1))
            p = torch.nn.Parameter(self.flat[off:off + n].view(shp))
            gv = self.flat_grad[off:off + n].view(shp)
            self._grad_views[name] = gv
            if name in ("_xyz", "_features") or self.fused_activations:
                # these reach the rasterizer untouched: it writes their gradient straight into the flat
                # buffer (rasterizer.py `_fr_grad_out`), no accumulation kernel, no zero-fill
                p._fr_grad_out = GradOut(gv)
            setattr(self, name, p)
            off += n

    @torch.no_grad()
    def resize(self, keep_mask=None, new_rows=None):
        """Prune and / or append Gaussians (reference: _prune_low_opacity_points / _uv_densify,
        model/fateavatar.py:610-711): rows where `keep_mask` is False are dropped, then `new_rows` — one raw tensor
        [n_new, ...] per field, in FIELDS order — are appended.  The flat buffers are rebuilt and every parameter is a
        new nn.Parameter (as in the reference); returns the row map `old_index` (int64 [P_new], -1 for appended rows)
        that optimizer state has to follow (FusedAdam.remap_rows)."""
        dev = self.flat.device
        P_old = self.P
        keep = torch.ones(P_old, dtype=torch.bool, device=dev) if keep_mask is None else keep_mask.to(dev).bool().reshape(-1)
        if keep.numel() != P_old:
            raise ValueError("keep_mask must have one entry per Gaussian")
        old_index = torch.nonzero(keep).reshape(-1)
        raw = [getattr(self, name).detach()[old_index] for name, _ in self.FIELDS]
        n_new = 0
        if new_rows is not None:
            n_new = int(new_rows[0].shape[0])
            for i, (r, add) in enumerate(zip(raw, new_rows)):
                add = add.to(dev, torch.float32).reshape((n_new,) + tuple(r.shape[1:]))
                raw[i] = torch.cat([r, add], dim=0)
        self._bind(raw)
        return torch.cat([old_index, torch.full((n_new,), -1, dtype=torch.int64, device=dev)])

    def _p(self, name):
        return getattr(self, name)

    # ---- the reference getters
    @property
    def get_xyz(self):
        return self._p("_xyz")

    @property
    def get_features(self):
        return self._p("_features")

    @property
    def get_opacity(self):
        return torch.sigmoid(self._p("_opacity"))

    @property
    def get_scaling(self):
        return torch.exp(self._p("_scaling"))

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._p("_rotation"))

    def grad_of(self, name):
        return getattr(self, name).grad

    def begin_step(self):
        """Drop the previous gradients (set_to_none, like the reference's zero_grad(set_to_none=True),
        train/iteration.py:49): the next backward ASSIGNS instead of accumulating."""
        for name, _ in self.FIELDS:
            getattr(self, name).grad = None
        self.accumulate_into_kept_grads(False)

    def accumulate_into_kept_grads(self, on: bool = True) -> None:
        """Gradient accumulation over several `.backward()` calls without `begin_step()` in between (one frame at a
        time, gradients kept): with `on`, the rasterizer's backward ADDS each further frame's gradients to the flat
        gradient buffer inside its kernel (FR_FLAG_ACCUMULATE; rasterizer.GradOut) instead of handing autograd a
        temporary to add.  Only for `.backward()`; `begin_step()` switches it off again."""
        for name, _ in self.FIELDS:
            slot = getattr(getattr(self, name), "_fr_grad_out", None)
            if slot is not None:
                slot.add_to_kept = bool(on)

    def exchange_buffer(self) -> torch.Tensor:
        """`collect_grads()` + the overflow word behind it: what a data-parallel step all-reduces (SUM)."""
        self.collect_grads()
        return self._grad_store

    def collect_grads(self) -> torch.Tensor:
        """After backward: make `flat_grad` hold every parameter's gradient (xyz / features are already there;
        the three activated parameters are copied in).  Returns the flat buffer for the data-parallel exchange."""
        for name, _ in self.FIELDS:
            g, view = getattr(self, name).grad, self._grad_views[name]
            if g is None:
                view.zero_()
            elif g.data_ptr() != view.data_ptr():
                view.copy_(g)
        return self.flat_grad


class TorchCamera:
    """Device-side camera with the attribute names render() reads (camera_3dgs.py:22-72)."""

    def __init__(self, cam, device):
        self.image_height, self.image_width = cam.image_height, cam.image_width
        self.FoVx, self.FoVy = cam.FoVx, cam.FoVy
        # the three tensors are views into ONE 35-float buffer, so that copy_from is a single device copy
        self._packed = torch.from_numpy(np.concatenate([
            np.asarray(cam.world_view_transform, np.float32).reshape(-1), np.asarray(cam.full_proj_transform, np.float32).reshape(-1),
            np.asarray(cam.camera_center, np.float32).reshape(-1)])).to(device)
        self.world_view_transform = self._packed[0:16].view(4, 4)
        self.full_proj_transform = self._packed[16:32].view(4, 4)
        self.camera_center = self._packed[32:35]

    def clone(self) -> "TorchCamera":
        """A camera with its own matrix block (same intrinsics): the static input of another captured frame."""
        o = TorchCamera.__new__(TorchCamera)
        o.__dict__.update(self.__dict__)
        o._packed = self._packed.clone()
        o.world_view_transform = o._packed[0:16].view(4, 4)
        o.full_proj_transform = o._packed[16:32].view(4, 4)
        o.camera_center = o._packed[32:35]
        return o

    def copy_from(self, other: "TorchCamera") -> None:
        """Overwrite the matrices in place (same intrinsics): lets a captured HIP graph render a new view."""
        self.check_same_intrinsics(other)
        self._packed.copy_(other._packed, non_blocking=True)

    def check_same_intrinsics(self, other: "TorchCamera") -> None:
        if (other.image_height, other.image_width, other.FoVx, other.FoVy) != \
                (self.image_height, self.image_width, self.FoVx, self.FoVy):
            raise ValueError("copy_from needs a camera with the same image size and field of view")
