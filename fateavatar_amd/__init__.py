"""fateavatar_amd — MI355X (gfx950) native 3D-Gaussian-splat rasterizer, a drop-in for the render path of
zjwfufu/FateAvatar (volume_rendering/render_3dgs.py -> diff_gaussian_rasterization, simple_knn).

Product modules: rasterizer (operator interface), render (caller-facing `render()`), knn (`distCUDA2`, initial scale),
binding (mesh binding of the Gaussians), optim (fused Adam), train (the per-frame optimisation step), dp (data-parallel
frame sharding), model (flat parameter holder), ply / obj / mesh_sampling (formats and init-time sampling), scenes
(synthetic inputs).  All device compute runs in hand-written HIP kernels behind the C ABI of include/fr_rasterizer.h
(fateavatar_amd/libfr_hip.so); there is no CPU fallback.
"""
import os as _os

# Kernel arguments of eager launches in DEVICE memory (where a captured graph's are): every wave's first scalar load then
# stays on the GPU instead of going to host memory — 2 us off an eager blend launch at BASELINE config 2, +7 % frames/s for
# frames that are not replayed from a graph.  The ROCm runtime reads the variable when it initialises, so it is set here, at
# import, unless the caller has decided otherwise; replayed graphs are unaffected.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
