"""ctypes binding of libfr_hip.so (the C ABI of include/fr_rasterizer.h).

The library is the product: if it is missing or cannot be loaded this module raises —
there is no CPU or PyTorch fallback anywhere in the package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("FR_HIP_LIB", os.path.join(_HERE, "libfr_hip.so"))  # FR_HIP_LIB: experiments only

FR_OK = 0
FR_ERR_INVALID_ARGUMENT = 1
FR_ERR_BINNING_CAPACITY = 2
FR_ERR_HIP = 3
FR_ERR_UNSUPPORTED = 4
FR_FLAG_NO_WAIT = 1
FR_FLAG_RAW_ACTIVATIONS = 2
FR_FLAG_ACCUMULATE_SHIFT = 8   # fr_backward: bit (8 + k) = add into the k-th array of fr_grads

_fp = C.c_void_p  # device pointers travel as integers


class fr_binding(C.Structure):
    _fields_ = [("N", C.c_int32), ("V", C.c_int32), ("F", C.c_int32), ("verts", C.c_void_p), ("faces", C.c_void_p),
                ("face_index", C.c_void_p), ("bary", C.c_void_p), ("face_scale_canonical", C.c_void_p),
                ("shell_len", C.c_float), ("resize_scale", C.c_int32), ("offset", C.c_void_p), ("rotation", C.c_void_p),
                ("scaling", C.c_void_p)]


class fr_aux(C.Structure):
    _fields_ = [("visible", C.c_void_p), ("grad_accum", C.c_void_p), ("denom", C.c_void_p),
                ("binding", C.POINTER(fr_binding)), ("d_verts", C.c_void_p), ("d_offset", C.c_void_p),
                ("d_rotation", C.c_void_p), ("d_scaling", C.c_void_p), ("overflow_out", C.c_void_p)]


class fr_params(C.Structure):
    _fields_ = [("P", C.c_int32), ("D", C.c_int32), ("M", C.c_int32), ("W", C.c_int32), ("H", C.c_int32),
                ("tan_fovx", C.c_float), ("tan_fovy", C.c_float), ("scale_modifier", C.c_float),
                ("prefiltered", C.c_int32), ("debug", C.c_int32), ("flags", C.c_int32),
                ("aux", C.POINTER(fr_aux))]


FR_ADAM_MAX_SEGMENTS = 16
FR_ADAM_STATE_FLOATS = 576
FR_ADAM_MAX_GRADS = 4
FR_MAX_BATCH = 4


class fr_adam_config(C.Structure):
    _fields_ = [("n_segments", C.c_int32), ("segment_end", C.c_uint64 * FR_ADAM_MAX_SEGMENTS),
                ("segment_lr", C.c_float * FR_ADAM_MAX_SEGMENTS), ("segment_period", C.c_uint32 * FR_ADAM_MAX_SEGMENTS),
                ("segment_split", C.c_uint32 * FR_ADAM_MAX_SEGMENTS), ("segment_lr2", C.c_float * FR_ADAM_MAX_SEGMENTS),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("grad_scale", C.c_float),
                ("skip", C.c_void_p * FR_ADAM_MAX_GRADS), ("n_skip", C.c_int32)]


class fr_inputs(C.Structure):
    _fields_ = [(n, _fp) for n in ("background", "means3D", "shs", "colors_precomp", "opacities", "scales",
                                   "rotations", "cov3D_precomp", "viewmatrix", "projmatrix", "campos")]


class fr_grads(C.Structure):
    _fields_ = [(n, _fp) for n in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh",
                                   "dL_dscales", "dL_drotations")]


class fr_counts(C.Structure):
    _fields_ = [("num_rendered", C.c_uint32), ("num_instances", C.c_uint32), ("max_tile_list", C.c_uint32),
                ("overflow", C.c_uint32)]


EXPORTS = ["fr_create", "fr_destroy", "fr_last_error", "fr_version", "fr_profile_enable", "fr_profile_read", "fr_geometry_bytes", "fr_image_bytes",
           "fr_binning_bytes", "fr_forward", "fr_forward_batch", "fr_read_counts", "fr_backward", "fr_backward_batch", "fr_mark_visible", "fr_image_final_T",
           "fr_image_n_contrib", "fr_debug_geometry_field", "fr_debug_selftest_reduce", "fr_knn_workspace_bytes", "fr_knn_mean_dist2", "fr_knn_nearest_dist2", "fr_adam_step", "fr_adam_step_multi", "fr_l1_workspace_bytes", "fr_l1_loss_grad", "fr_l1_loss_grad_batch", "fr_multi_copy", "fr_scaled_sum", "fr_face_scale",
           "fr_bind_forward", "fr_bind_backward"]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j8"]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return SO_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise RuntimeError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C fateavatar_amd/csrc`). There is no fallback path.")
    # PyTorch-ROCm ships its own libamdhip64: import it FIRST so that this library binds to the same
    # HIP runtime instance (two runtimes in one process cannot share streams or device pointers).
    import torch  # noqa: F401

    # RTLD_GLOBAL: the compiled `_C` extension (fateavatar_amd/torch_ext.py) links against this library by name
    L = C.CDLL(SO_PATH, mode=C.RTLD_GLOBAL)
    L.fr_create.argtypes = [C.POINTER(C.c_void_p)]
    L.fr_create.restype = C.c_int
    L.fr_destroy.argtypes = [C.c_void_p]
    L.fr_last_error.restype = C.c_char_p
    L.fr_version.restype = C.c_char_p
    L.fr_profile_enable.argtypes = [C.c_void_p, C.c_int32]
    L.fr_profile_enable.restype = C.c_int
    L.fr_profile_read.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint32)]
    L.fr_profile_read.restype = C.c_int
    L.fr_geometry_bytes.argtypes = [C.c_int32]
    L.fr_geometry_bytes.restype = C.c_size_t
    L.fr_image_bytes.argtypes = [C.c_int32, C.c_int32]
    L.fr_image_bytes.restype = C.c_size_t
    L.fr_binning_bytes.argtypes = [C.c_uint64, C.c_int32, C.c_int32]
    L.fr_binning_bytes.restype = C.c_size_t
    L.fr_forward.argtypes = [C.c_void_p, C.POINTER(fr_params), C.POINTER(fr_inputs), _fp, _fp, _fp, _fp, _fp,
                             C.c_uint64, C.POINTER(fr_counts), C.c_void_p]
    L.fr_forward.restype = C.c_int
    # batched frames: arrays (one entry per view) of what fr_forward / fr_backward take
    _pp = C.POINTER(C.c_void_p)
    L.fr_forward_batch.argtypes = [C.c_int32, _pp, C.POINTER(C.POINTER(fr_params)), C.POINTER(C.POINTER(fr_inputs)), _pp, _pp,
                                   _pp, _pp, _pp, C.POINTER(C.c_uint64), C.POINTER(fr_counts), C.c_void_p]
    L.fr_forward_batch.restype = C.c_int
    L.fr_backward_batch.argtypes = [C.c_int32, _pp, C.POINTER(C.POINTER(fr_params)), C.POINTER(C.POINTER(fr_inputs)), _pp, _pp,
                                    _pp, _pp, _pp, C.POINTER(C.POINTER(fr_grads)), C.c_void_p]
    L.fr_backward_batch.restype = C.c_int
    L.fr_read_counts.argtypes = [C.c_void_p, C.POINTER(fr_counts)]
    L.fr_read_counts.restype = C.c_int
    L.fr_backward.argtypes = [C.c_void_p, C.POINTER(fr_params), C.POINTER(fr_inputs), _fp, _fp, _fp, _fp, _fp,
                              C.POINTER(fr_grads), C.c_void_p]
    L.fr_backward.restype = C.c_int
    L.fr_mark_visible.argtypes = [C.c_int32, _fp, _fp, _fp, _fp, C.c_void_p]
    L.fr_mark_visible.restype = C.c_int
    L.fr_image_final_T.argtypes = [_fp, C.c_int32, C.c_int32]
    L.fr_image_final_T.restype = C.c_void_p
    L.fr_image_n_contrib.argtypes = [_fp, C.c_int32, C.c_int32]
    L.fr_image_n_contrib.restype = C.c_void_p
    L.fr_debug_geometry_field.argtypes = [_fp, C.c_int32, C.c_int32]
    L.fr_debug_geometry_field.restype = C.c_void_p
    L.fr_debug_selftest_reduce.argtypes = [_fp, _fp, C.c_void_p]
    L.fr_debug_selftest_reduce.restype = C.c_int
    L.fr_knn_workspace_bytes.argtypes = [C.c_int32]
    L.fr_knn_workspace_bytes.restype = C.c_size_t
    L.fr_knn_mean_dist2.argtypes = [C.c_int32, _fp, _fp, _fp, C.c_size_t, C.c_void_p]
    L.fr_knn_mean_dist2.restype = C.c_int
    L.fr_knn_nearest_dist2.argtypes = [C.c_int32, _fp, _fp, _fp, C.c_size_t, C.c_void_p]
    L.fr_knn_nearest_dist2.restype = C.c_int
    L.fr_face_scale.argtypes = [C.c_int32, C.c_int32, _fp, _fp, _fp, C.c_void_p]
    L.fr_face_scale.restype = C.c_int
    L.fr_bind_forward.argtypes = [C.POINTER(fr_binding), _fp, _fp, _fp, C.c_void_p]
    L.fr_bind_forward.restype = C.c_int
    L.fr_bind_backward.argtypes = [C.POINTER(fr_binding), _fp, _fp, _fp, _fp, _fp, _fp, _fp, C.c_void_p]
    L.fr_bind_backward.restype = C.c_int
    L.fr_adam_step.argtypes = [C.POINTER(fr_adam_config), _fp, _fp, _fp, _fp, C.c_uint64, _fp, C.c_void_p]
    L.fr_adam_step.restype = C.c_int
    L.fr_adam_step_multi.argtypes = [C.POINTER(fr_adam_config), _fp, C.POINTER(C.c_void_p), C.c_int32, _fp, _fp, C.c_uint64, _fp,
                                     C.c_void_p]
    L.fr_adam_step_multi.restype = C.c_int
    L.fr_l1_workspace_bytes.argtypes = []
    L.fr_l1_workspace_bytes.restype = C.c_size_t
    L.fr_l1_loss_grad.argtypes = [C.c_uint64, _fp, _fp, _fp, _fp, C.c_void_p, C.c_void_p]
    L.fr_l1_loss_grad.restype = C.c_int
    L.fr_l1_loss_grad_batch.argtypes = [C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.fr_l1_loss_grad_batch.restype = C.c_int
    L.fr_multi_copy.argtypes = [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_void_p]
    L.fr_multi_copy.restype = C.c_int
    L.fr_scaled_sum.argtypes = [C.c_int32, C.POINTER(C.c_void_p), _fp, C.c_uint64, C.c_float, C.c_void_p]
    L.fr_scaled_sum.restype = C.c_int
    _lib = L
    return L


STAGES = ("preprocess_fwd", "scan", "emit", "tile_sort", "blend_fwd", "blend_bwd", "preprocess_bwd")


def profile_enable(device_index: int, on: bool) -> None:
    rc = lib().fr_profile_enable(handle(device_index), int(on))
    if rc != FR_OK:
        raise RuntimeError(last_error())


def profile_read(device_index: int) -> dict:
    """{stage: (total_ms, launches)} since profile_enable(True); synchronise the device first."""
    out = {}
    for i, name in enumerate(STAGES):
        ms, n = C.c_double(), C.c_uint32()
        rc = lib().fr_profile_read(handle(device_index), i, C.byref(ms), C.byref(n))
        if rc != FR_OK:
            raise RuntimeError(last_error())
        out[name] = (ms.value, n.value)
    return out


def last_error() -> str:
    return lib().fr_last_error().decode("utf-8", "replace")


_handles: dict = {}


def handle(device_index: int, slot: int = 0):
    """One fr_handle per (device, slot), created with that device current.  Frames of ONE handle are ordered (the handle
    owns per-frame state); frames that are to overlap on the device — several views in flight on several streams — use
    different slots (rasterizer.handle_slot)."""
    key = device_index if slot == 0 else (device_index, slot)
    h = _handles.get(key)
    if h is None:
        import torch

        if torch.cuda.is_current_stream_capturing():
            # fr_create allocates (device + pinned memory, a stream, events): not allowed under capture, and trying
            # would invalidate the caller's capture
            raise RuntimeError(f"rasterizer handle (device {device_index}, slot {slot}) does not exist yet and cannot be "
                               "created while a stream is being captured: run one eager frame through it first")
        with torch.cuda.device(device_index):
            out = C.c_void_p()
            rc = lib().fr_create(C.byref(out))
            if rc != FR_OK:
                raise RuntimeError(f"fr_create failed: {last_error()}")
        h = out
        _handles[key] = h
    return h
