"""The reference's pybind module surface (submodules/diff-gaussian-rasterization/ext.cpp:15-19):
rasterize_gaussians, rasterize_gaussians_backward, mark_visible.

Default: the Python/ctypes host (fateavatar_amd/rasterizer.py).  With FR_USE_TORCH_EXT=1 in the environment the names
resolve to the COMPILED torch extension (fateavatar_amd/csrc/torch_ext.cpp, built by __graft_entry__.build()), which
has the reference's exact C++ signatures; both sit on the same C ABI (include/fr_rasterizer.h)."""
import os

if os.environ.get("FR_USE_TORCH_EXT") == "1":
    from fateavatar_amd import torch_ext as _te
    _m = _te.load()
    rasterize_gaussians = _m.rasterize_gaussians
    rasterize_gaussians_backward = _m.rasterize_gaussians_backward
    mark_visible = _m.mark_visible
else:
    from fateavatar_amd.rasterizer import (mark_visible, rasterize_gaussians,  # noqa: F401
                                           rasterize_gaussians_backward)
