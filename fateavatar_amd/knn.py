"""`simple_knn._C.distCUDA2` on MI355X (reference: submodules/simple-knn/spatial.cu:14-25, ext.cpp)."""
from __future__ import annotations

import torch

from . import _lib


def _knn(points: torch.Tensor, fn_name: str) -> torch.Tensor:
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be on a HIP device (torch device 'cuda'); there is no CPU path")
    dev = points.device.index if points.device.index is not None else torch.cuda.current_device()
    P = points.size(0)
    pts = points.float().contiguous()
    means = torch.full((P,), 0.0, dtype=torch.float32, device=points.device)
    if P == 0:
        return means
    L = _lib.lib()
    ws_bytes = L.fr_knn_workspace_bytes(P)
    ws = torch.empty((ws_bytes,), dtype=torch.uint8, device=points.device)
    with torch.cuda.device(dev):
        rc = getattr(L, fn_name)(P, pts.data_ptr(), means.data_ptr(), ws.data_ptr(), ws_bytes,
                                 torch.cuda.current_stream(dev).cuda_stream)
    if rc != _lib.FR_OK:
        raise RuntimeError(f"{fn_name} failed (code {rc}): {_lib.last_error()}")
    return means


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] float32 on a HIP device -> [P] mean squared distance to the 3 nearest other points."""
    return _knn(points, "fr_knn_mean_dist2")


def nearest_dist2(points: torch.Tensor) -> torch.Tensor:
    """points [P,3] -> [P] squared distance to the nearest other point (= knn_points(p, p, K>=2).dists[..., 1])."""
    return _knn(points, "fr_knn_nearest_dist2")


def init_scale_by_knn(points: torch.Tensor):
    """FateAvatar.get_init_scale_by_knn (model/fateavatar.py:597-608) without pytorch3d: (mean_scaling, max_scaling,
    scale_init) = (mean nearest-neighbour distance, 10x that, its log on the CPU)."""
    mean_scaling = torch.sqrt(nearest_dist2(points)).mean()
    return mean_scaling, 10 * mean_scaling, torch.log(mean_scaling).cpu()
