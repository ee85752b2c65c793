"""Import-name alias so that the reference caller works unchanged
(`from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer`,
reference volume_rendering/render_3dgs.py:3, model/baseline/monogaussianavatar.py:13)."""
from fateavatar_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                       _RasterizeGaussians, cpu_deep_copy_tuple)
from fateavatar_amd.rasterizer import rasterize_gaussians_autograd as rasterize_gaussians  # noqa: F401
from . import _C  # noqa: F401
