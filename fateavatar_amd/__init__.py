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

# Replayed graphs: ROCm 7 replays a captured graph from AQL packets it recorded at instantiation ("graph packet capture");
# on this stack a replay then ends with ~8.6 us before the next thing on the stream starts, whatever that is.  With the
# recording off the runtime enqueues the graph's kernel nodes like ordinary launches (the ROCm 6 path): ~40 us more HOST time
# per replay of nine nodes, but 4 us less on the device per replay — +4.4 % frames/s one frame at a time, +3.5 % FateAvatar
# steps/s, +0.8 % with twelve frames in flight (EXPERIMENTS.md).  Read by the runtime when it initialises, like the above;
# set DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 in the environment to keep the runtime's default.
_os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")

__version__ = "0.1.0"
