"""fateavatar_amd — MI355X (gfx950) native 3D-Gaussian-splat rasterizer, a drop-in for the render path of
zjwfufu/FateAvatar (volume_rendering/render_3dgs.py -> diff_gaussian_rasterization, simple_knn).

Product modules: rasterizer (operator interface), render (caller-facing `render()`), knn (`distCUDA2`),
dp (data-parallel frame sharding), scenes (synthetic inputs).  All compute runs in hand-written HIP kernels
behind the C ABI of include/fr_rasterizer.h (fateavatar_amd/libfr_hip.so); there is no CPU fallback.
"""
__version__ = "0.1.0"
