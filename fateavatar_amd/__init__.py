"""fateavatar_amd — MI355X (gfx950) native 3D-Gaussian-splat rasterizer, a drop-in for the render path of
zjwfufu/FateAvatar (volume_rendering/render_3dgs.py -> diff_gaussian_rasterization, simple_knn).

Product modules: rasterizer (operator interface), render (caller-facing `render()`), knn (`distCUDA2`, initial scale),
binding (mesh binding of the Gaussians), optim (fused Adam), train (the per-frame optimisation step), dp (data-parallel
frame sharding), model (flat parameter holder), ply / obj / mesh_sampling (formats and init-time sampling), scenes
(synthetic inputs).  All device compute runs in hand-written HIP kernels behind the C ABI of include/fr_rasterizer.h
(fateavatar_amd/libfr_hip.so); there is no CPU fallback.
"""
import os as _os

# Two process-wide switches of the ROCm runtime make this package's launches cheaper.  They are NOT applied by importing the
# package (they change how every HIP user of the process behaves, PyTorch's own graphs and RCCL included, and one of them is
# an undocumented debug knob of the runtime): call `fateavatar_amd.tune_runtime()` BEFORE anything initialises HIP, or set
# FR_TUNE_RUNTIME=1 in the environment to have the import do it.  bench.py and the tools call it explicitly and print the
# effective values (`config.hip_env`); `runtime_env()` returns them for any caller's own records.
#   HIP_FORCE_DEV_KERNARG=1            kernel arguments of eager launches in DEVICE memory (where a captured graph's are):
#       every wave's first scalar load stays on the GPU — 2 us off an eager blend launch at BASELINE config 2, +7 % frames/s
#       for frames that are not replayed from a graph; replayed graphs are unaffected.
#   DEBUG_CLR_GRAPH_PACKET_CAPTURE=0   ROCm 7 replays a captured graph from AQL packets recorded at instantiation; on this
#       stack a replay then ends with ~8.6 us before the next thing on the stream starts.  With the recording off the runtime
#       enqueues the graph's kernel nodes like ordinary launches (the ROCm 6 path): ~40 us more HOST time per replay of nine
#       nodes, 4 us less on the device — +4.4 % frames/s one frame at a time, +3.5 % FateAvatar steps/s, +0.8 % with twelve
#       frames in flight (EXPERIMENTS.md).  A point release of the runtime may drop the knob; nothing depends on it.
_RUNTIME_SWITCHES = {"HIP_FORCE_DEV_KERNARG": "1", "DEBUG_CLR_GRAPH_PACKET_CAPTURE": "0"}


def runtime_env() -> dict:
    """The runtime switches as this process's environment holds them (None = the runtime's default)."""
    return {k: _os.environ.get(k) for k in _RUNTIME_SWITCHES}


def tune_runtime(warn: bool = True) -> dict:
    """Set the two ROCm runtime switches above unless the environment already holds a value for them.  The runtime reads them
    when it initialises: called after that (torch.cuda already initialised) the call changes nothing for this process, says
    so once in a warning, and `late` in the returned record is True.  Returns {"env": runtime_env(), "late": bool}."""
    late = False
    try:
        import sys as _sys
        _t = _sys.modules.get("torch")
        late = bool(_t is not None and _t.cuda.is_initialized())
    except Exception:
        late = False
    for k, v in _RUNTIME_SWITCHES.items():
        _os.environ.setdefault(k, v)
    if late and warn:
        import warnings as _w
        _w.warn("fateavatar_amd.tune_runtime() was called after the ROCm runtime initialised: HIP_FORCE_DEV_KERNARG / "
                "DEBUG_CLR_GRAPH_PACKET_CAPTURE take effect only in processes started from here on", RuntimeWarning, stacklevel=2)
    return {"env": runtime_env(), "late": late}


if _os.environ.get("FR_TUNE_RUNTIME") == "1":
    tune_runtime()

__version__ = "0.1.0"
