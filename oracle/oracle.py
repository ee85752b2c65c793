"""ctypes/numpy front end of the CPU oracle (oracle/fr_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from the product packages.  Parity status and
citations are in the header of fr_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libfr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "fr_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        fp = C.POINTER(C.c_float)
        L.fro_create.restype = C.c_void_p
        L.fro_destroy.argtypes = [C.c_void_p]
        L.fro_forward.restype = C.c_int
        L.fro_forward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp, fp,
                                  C.c_float, fp, fp, fp, fp, fp, C.c_float, C.c_float, fp, C.POINTER(C.c_int)]
        L.fro_backward.restype = None
        L.fro_backward.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, fp, C.c_int, C.c_int, fp, fp, fp, fp,
                                   C.c_float, fp, fp, fp, fp, fp, C.c_float, C.c_float, C.POINTER(C.c_int), fp] + [fp] * 9
        L.fro_mark_visible.argtypes = [C.c_int, fp, fp, fp, C.POINTER(C.c_uint8)]
        L.fro_knn_mean_dist2.argtypes = [C.c_int, fp, fp]
        L.fro_num_rendered.argtypes = [C.c_void_p]
        L.fro_num_rendered.restype = C.c_int
        L.fro_eval_sh.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp, C.POINTER(C.c_uint8)]
        L.fro_cov3d.argtypes = [fp, C.c_float, fp, fp]
        L.fro_num_threads.restype = C.c_int
        L.fro_set_num_threads.argtypes = [C.c_int]
        L.fro_set_bwd_float_order.argtypes = [C.c_int]
        L.fro_set_bwd_contract.argtypes = [C.c_int]
        for name, ty in [("depths", C.c_float), ("clamped", C.c_uint8), ("means2D", C.c_float), ("cov3D", C.c_float),
                         ("conic_opacity", C.c_float), ("rgb", C.c_float), ("tiles_touched", C.c_uint32),
                         ("final_T", C.c_float), ("n_contrib", C.c_uint32), ("ranges", C.c_uint32),
                         ("point_list", C.c_uint32), ("point_keys", C.c_uint64)]:
            f = getattr(L, "fro_" + name)
            f.argtypes = [C.c_void_p]
            f.restype = C.POINTER(ty)
        _lib = L
    return _lib


def _f32(a):
    if a is None:
        return None
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _p(a):
    if a is None or a.size == 0:
        return None
    return a.ctypes.data_as(C.POINTER(C.c_float))


@dataclass
class ForwardResult:
    num_rendered: int
    color: np.ndarray          # [3,H,W]
    radii: np.ndarray          # [P] int32
    final_T: np.ndarray        # [H,W]
    n_contrib: np.ndarray      # [H,W] uint32 (reference semantics: index in the 16x16 tile list)
    means2D: np.ndarray        # [P,2]
    depths: np.ndarray
    cov3D: np.ndarray
    conic_opacity: np.ndarray
    rgb: np.ndarray
    clamped: np.ndarray
    tiles_touched: np.ndarray
    ranges: np.ndarray         # [T,2]
    point_list: np.ndarray     # [R]
    _ctx: object = field(default=None, repr=False)
    _inputs: dict = field(default=None, repr=False)


class _Ctx:
    def __init__(self):
        self.h = lib().fro_create()

    def __del__(self):
        try:
            lib().fro_destroy(self.h)
        except Exception:
            pass


def forward(*, bg, means3D, opacities, viewmatrix, projmatrix, campos, tanfovx, tanfovy, H, W, shs=None, sh_degree=0,
            colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0) -> ForwardResult:
    """Mirror of `_C.rasterize_gaussians` (DGR/rasterize_points.cu:35-115) on numpy arrays."""
    L = lib()
    means3D = _f32(means3D)
    P = means3D.shape[0]
    inp = dict(bg=_f32(bg), means3D=means3D, opacities=_f32(opacities).reshape(-1), viewmatrix=_f32(viewmatrix),
               projmatrix=_f32(projmatrix), campos=_f32(campos), shs=_f32(shs), colors_precomp=_f32(colors_precomp),
               scales=_f32(scales), rotations=_f32(rotations), cov3D_precomp=_f32(cov3D_precomp),
               tanfovx=float(tanfovx), tanfovy=float(tanfovy), H=int(H), W=int(W), D=int(sh_degree),
               scale_modifier=float(scale_modifier))
    M = 0 if inp["shs"] is None or inp["shs"].size == 0 else inp["shs"].shape[1]
    inp["M"] = M
    color = np.zeros((3, H, W), np.float32)
    radii = np.zeros((P,), np.int32)
    ctx = _Ctx()
    R = 0
    if P:
        R = L.fro_forward(ctx.h, P, inp["D"], M, _p(inp["bg"]), W, H, _p(means3D), _p(inp["shs"]),
                          _p(inp["colors_precomp"]), _p(inp["opacities"]), _p(inp["scales"]), inp["scale_modifier"],
                          _p(inp["rotations"]), _p(inp["cov3D_precomp"]), _p(inp["viewmatrix"]), _p(inp["projmatrix"]),
                          _p(inp["campos"]), inp["tanfovx"], inp["tanfovy"], _p(color),
                          radii.ctypes.data_as(C.POINTER(C.c_int)))
    T = ((W + 15) // 16) * ((H + 15) // 16)

    def arr(name, shape, dtype):
        if P == 0 or int(np.prod(shape)) == 0:
            return np.zeros(shape, dtype)
        ptr = getattr(L, "fro_" + name)(ctx.h)
        return np.ctypeslib.as_array(ptr, shape=(int(np.prod(shape)),)).astype(dtype).reshape(shape)

    return ForwardResult(
        num_rendered=R, color=color, radii=radii,
        final_T=arr("final_T", (H, W), np.float32), n_contrib=arr("n_contrib", (H, W), np.uint32),
        means2D=arr("means2D", (P, 2), np.float32), depths=arr("depths", (P,), np.float32),
        cov3D=arr("cov3D", (P, 6), np.float32), conic_opacity=arr("conic_opacity", (P, 4), np.float32),
        rgb=arr("rgb", (P, 3), np.float32), clamped=arr("clamped", (P, 3), np.uint8),
        tiles_touched=arr("tiles_touched", (P,), np.uint32), ranges=arr("ranges", (T, 2), np.uint32),
        point_list=arr("point_list", (R,), np.uint32), _ctx=ctx, _inputs=inp)


@dataclass
class BackwardResult:
    dL_dmeans2D: np.ndarray   # [P,3]
    dL_dcolors: np.ndarray    # [P,3]
    dL_dopacity: np.ndarray   # [P,1]
    dL_dmeans3D: np.ndarray   # [P,3]
    dL_dcov3D: np.ndarray     # [P,6]
    dL_dsh: np.ndarray        # [P,M,3]
    dL_dscales: np.ndarray    # [P,3]
    dL_drotations: np.ndarray  # [P,4]
    dL_dconic: np.ndarray     # [P,2,2]


def backward(fwd: ForwardResult, dL_dout_color) -> BackwardResult:
    """Mirror of `_C.rasterize_gaussians_backward` (DGR/rasterize_points.cu:117-196)."""
    L = lib()
    i = fwd._inputs
    P = i["means3D"].shape[0]
    M = i["M"]
    g = _f32(dL_dout_color)
    out = BackwardResult(
        dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dcolors=np.zeros((P, 3), np.float32),
        dL_dopacity=np.zeros((P, 1), np.float32), dL_dmeans3D=np.zeros((P, 3), np.float32),
        dL_dcov3D=np.zeros((P, 6), np.float32), dL_dsh=np.zeros((P, M, 3), np.float32),
        dL_dscales=np.zeros((P, 3), np.float32), dL_drotations=np.zeros((P, 4), np.float32),
        dL_dconic=np.zeros((P, 2, 2), np.float32))
    if P:
        radii = np.ascontiguousarray(fwd.radii, dtype=np.int32)
        L.fro_backward(fwd._ctx.h, P, i["D"], M, _p(i["bg"]), i["W"], i["H"], _p(i["means3D"]), _p(i["shs"]),
                       _p(i["colors_precomp"]), _p(i["scales"]), i["scale_modifier"], _p(i["rotations"]),
                       _p(i["cov3D_precomp"]), _p(i["viewmatrix"]), _p(i["projmatrix"]), _p(i["campos"]),
                       i["tanfovx"], i["tanfovy"], radii.ctypes.data_as(C.POINTER(C.c_int)), _p(g),
                       _p(out.dL_dmeans2D), _p(out.dL_dconic), _p(out.dL_dopacity), _p(out.dL_dcolors),
                       _p(out.dL_dmeans3D), _p(out.dL_dcov3D), _p(out.dL_dsh) if M else None,
                       _p(out.dL_dscales), _p(out.dL_drotations))
    return out


def mark_visible(means3D, viewmatrix, projmatrix) -> np.ndarray:
    means3D = _f32(means3D)
    P = means3D.shape[0]
    out = np.zeros((P,), np.uint8)
    if P:
        lib().fro_mark_visible(P, _p(means3D), _p(_f32(viewmatrix)), _p(_f32(projmatrix)),
                               out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out.astype(bool)


def knn_mean_dist2(points) -> np.ndarray:
    """Mirror of `simple_knn._C.distCUDA2` (KNN/spatial.cu:14-25)."""
    pts = _f32(points)
    P = pts.shape[0]
    out = np.zeros((P,), np.float32)
    if P:
        lib().fro_knn_mean_dist2(P, _p(pts), _p(out))
    return out


def num_threads() -> int:
    return lib().fro_num_threads()


def cpu_quota_cores() -> int:
    """Cores this process may actually use: the scheduler affinity, capped by the container's CPU quota (cgroup v2
    `cpu.max`, v1 `cpu.cfs_quota_us`).  os.cpu_count() reports the host's hardware threads (256 on the GPU boxes) while the
    quota there is 16 CPUs: threads beyond it only time-slice against each other."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, math.ceil(q / per)))
        except Exception:
            pass
    return n


def set_num_threads(n: int) -> None:
    lib().fro_set_num_threads(int(n))


def set_bwd_float_order(seed: int) -> None:
    """Diagnostic (tests/test_gpu_parity.py's noise floor): seed != 0 makes backward() add every Gaussian's per-pixel terms in
    FLOAT and in a seeded order, as the reference's atomics do (fr_oracle.c: fro_set_bwd_float_order); 0 restores the
    double sums."""
    lib().fro_set_bwd_float_order(int(seed))


def set_bwd_contract(on: bool) -> None:
    """Diagnostic: evaluate the backward's Gaussian exponent with an FMA contraction, as nvcc's default would (fr_oracle.c:
    fro_set_bwd_contract)."""
    lib().fro_set_bwd_contract(1 if on else 0)


def eval_sh(deg: int, sh, mean, campos):
    """Clamped RGB of one Gaussian (forward.cu:20-71). sh [M,3]."""
    sh, mean, campos = _f32(sh), _f32(mean), _f32(campos)
    out = np.zeros(3, np.float32)
    cl = np.zeros(3, np.uint8)
    lib().fro_eval_sh(int(deg), sh.shape[0], _p(mean), _p(campos), _p(sh), _p(out), cl.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out, cl.astype(bool)


def cov3d(scale, mod: float, rot):
    """6 upper-triangular floats of Sigma = R S^2 R^T (forward.cu:118-152)."""
    out = np.zeros(6, np.float32)
    lib().fro_cov3d(_p(_f32(scale)), float(mod), _p(_f32(rot)), _p(out))
    return out
