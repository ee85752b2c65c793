"""fateavatar_amd — MI355X (gfx950) native 3D-Gaussian-splat rasterizer, a drop-in for the render path of
zjwfufu/FateAvatar (volume_rendering/render_3dgs.py -> diff_gaussian_rasterization, simple_knn).

Product modules: rasterizer (operator interface), render (caller-facing `render()`), knn (`distCUDA2`, initial scale),
binding (mesh binding of the Gaussians), optim (fused Adam), train (the per-frame optimisation step), dp (data-parallel
frame sharding), model (flat parameter holder), ply / obj / mesh_sampling (formats and init-time sampling), scenes
(synthetic inputs).  All device compute runs in hand-written HIP kernels behind the C ABI of include/fr_rasterizer.h
(fateavatar_amd/libfr_hip.so); there is no CPU fallback.
"""
__version__ = "0.1.0"
