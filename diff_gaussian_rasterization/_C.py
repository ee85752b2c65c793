"""The reference's pybind module surface (submodules/diff-gaussian-rasterization/ext.cpp:15-19)."""
from fateavatar_amd.rasterizer import (mark_visible, rasterize_gaussians,  # noqa: F401
                                       rasterize_gaussians_backward)
