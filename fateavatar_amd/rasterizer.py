"""Python host of the MI355X rasterizer: the reference's operator interface, same names,
argument meaning and error behaviour, over the C ABI in include/fr_rasterizer.h.

Mirrors (paths relative to /root/reference/submodules/diff-gaussian-rasterization/):
  GaussianRasterizationSettings   diff_gaussian_rasterization/__init__.py:157-169
  GaussianRasterizer              diff_gaussian_rasterization/__init__.py:171-220
  _RasterizeGaussians             diff_gaussian_rasterization/__init__.py:44-155
  rasterize_gaussians / rasterize_gaussians_backward / mark_visible  (the `_C` module)
                                  ext.cpp:15-19, rasterize_points.cu:35-217

All tensors must live on a HIP device ("cuda" in PyTorch-ROCm).  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib

NUM_CHANNELS = 3  # cuda_rasterizer/config.h:15

# per-device guess of the binning capacity (instances); grows when a frame overflows it
_capacity_hint: dict = {}
last_counts: dict = {}  # device index -> fr_counts of the most recent forward (diagnostics)


def _dev_index(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise RuntimeError("fateavatar_amd rasterizer: tensors must be on a HIP device (torch device 'cuda'); "
                           "there is no CPU path")
    return t.device.index if t.device.index is not None else torch.cuda.current_device()


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: torch.Tensor | None) -> int:
    return 0 if t is None or t.numel() == 0 else t.data_ptr()


def _check(rc: int, what: str):
    if rc != _lib.FR_OK:
        raise RuntimeError(f"{what} failed (code {rc}): {_lib.last_error()}")


# When True, forwards never wait for their instance counts (graph-capturable, zero host syncs): the binning
# capacity is the current high-water mark and overflow is only detected by `check_async_overflow()`.
_no_wait = False


def set_no_wait(on: bool) -> None:
    """Capture mode: make rasterize_gaussians free of host synchronisation (see FR_FLAG_NO_WAIT).  Process-wide:
    prefer the scoped form `with no_wait(): ...` around a graph capture, so that renders outside the capture (a
    validation view, a second model) keep their overflow check and capacity regrow."""
    global _no_wait
    _no_wait = bool(on)


class no_wait:
    """`with rasterizer.no_wait():` — FR_FLAG_NO_WAIT for the forwards issued inside the block only."""

    def __enter__(self):
        global _no_wait
        self._prev = _no_wait
        _no_wait = True
        return self

    def __exit__(self, *exc):
        global _no_wait
        _no_wait = self._prev
        return False


_slot = 0   # which fr_handle of the device the calls of this thread of control use (see handle_slot)


class handle_slot:
    """`with rasterizer.handle_slot(k):` — render through the k-th handle of the device.  Frames of one handle are
    ordered; views that should be IN FLIGHT TOGETHER (one stream and one captured graph each) take one slot each."""

    def __init__(self, slot: int):
        self.slot = int(slot)

    def __enter__(self):
        global _slot
        self._prev, _slot = _slot, self.slot
        return self

    def __exit__(self, *exc):
        global _slot
        _slot = self._prev
        return False


def read_counts(device_index: int = 0, slot: int | None = None):
    """fr_counts of the most recent frame on this device (synchronise first)."""
    c = _lib.fr_counts()
    _check(_lib.lib().fr_read_counts(_lib.handle(device_index, _slot if slot is None else slot), C.byref(c)), "fr_read_counts")
    return c


def check_async_overflow(device_index: int = 0) -> bool:
    """After synchronising: True if the last no-wait frame overflowed its binning capacity (its outputs are
    then invalid); the capacity hint is raised so that the next frame fits."""
    c = read_counts(device_index)
    last_counts[device_index] = c
    if c.overflow:
        _capacity_hint[device_index] = max(_capacity_hint.get(device_index, 0), int(c.num_instances * 1.25) + 1024)
    return bool(c.overflow)


def _params(P, degree, M, W, H, tan_fovx, tan_fovy, scale_modifier, prefiltered, debug, raw=False,
            aux=None, extra_flags=0) -> _lib.fr_params:
    flags = (_lib.FR_FLAG_NO_WAIT if _no_wait else 0) | (_lib.FR_FLAG_RAW_ACTIVATIONS if raw else 0) | int(extra_flags)
    prm = _lib.fr_params(int(P), int(degree), int(M), int(W), int(H), float(tan_fovx), float(tan_fovy),
                         float(scale_modifier), int(bool(prefiltered)), int(bool(debug)), flags)
    if aux is not None:
        prm._aux_keepalive = aux
        prm.aux = C.pointer(aux)
    return prm


def _stats3(stats):
    """(xyz_gradient_accum, denom[, overflow word]) -> the three of them (None where absent)."""
    if stats is None:
        return None, None, None
    return stats[0], stats[1], (stats[2] if len(stats) > 2 else None)


def _aux(visible=None, grad_accum=None, denom=None, binding=None, bind_grads=None, overflow_out=None):
    """fr_aux (optional fused side inputs / outputs) from torch tensors, or None if nothing is asked for.  `binding`: an
    `_lib.fr_binding` descriptor (the frame is rendered straight from its mesh binding); `bind_grads`: dict with the
    backward's d_verts / d_offset / d_rotation / d_scaling tensors (any may be None)."""
    if visible is None and grad_accum is None and denom is None and binding is None and overflow_out is None:
        return None
    for t, dt in ((visible, (torch.bool, torch.uint8)), (grad_accum, (torch.float32,)), (denom, (torch.float32,))):
        if t is not None and (t.dtype not in dt or not t.is_contiguous() or not t.is_cuda):
            raise RuntimeError("fused side outputs must be contiguous device tensors (bool/uint8 mask, float32 stats)")
    aux = _lib.fr_aux(*(t.data_ptr() if t is not None else None for t in (visible, grad_accum, denom)))
    if overflow_out is not None:
        if not (overflow_out.is_cuda and overflow_out.dtype == torch.float32 and overflow_out.numel() >= 1 and overflow_out.is_contiguous()):
            raise RuntimeError("the overflow word must be a contiguous float32 device tensor")
        aux._overflow_keepalive = overflow_out
        aux.overflow_out = overflow_out.data_ptr()
    if binding is not None:
        aux._binding_keepalive = binding
        aux.binding = C.pointer(binding)
        for n in ("d_verts", "d_offset", "d_rotation", "d_scaling"):
            t = (bind_grads or {}).get(n)
            if t is not None:
                if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                    raise RuntimeError(f"binding gradient {n} must be a contiguous float32 device tensor")
                setattr(aux, n, t.data_ptr())
    return aux


def _inputs(bg, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos):
    return _lib.fr_inputs(_ptr(bg), _ptr(means3D), _ptr(sh), _ptr(colors), _ptr(opacity), _ptr(scales),
                          _ptr(rotations), _ptr(cov3D_precomp), _ptr(viewmatrix), _ptr(projmatrix), _ptr(campos))


def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                        prefiltered, debug, _raw=False, _visible=None):
    """`_C.rasterize_gaussians` (rasterize_points.cu:35-115).  `_raw=True` (extension, FR_FLAG_RAW_ACTIVATIONS):
    opacity / scales / rotations are the RAW parameters and the kernels apply sigmoid / exp / normalize.

    Returns (num_rendered, out_color[3,H,W], radii[P] int32, geomBuffer, binningBuffer, imgBuffer); the three
    byte buffers are opaque and must be handed back to `rasterize_gaussians_backward`."""
    if means3D.dim() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    dev = _dev_index(means3D)
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    opts = dict(device=means3D.device)
    out_color = torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, **opts)
    radii = torch.empty((P,), dtype=torch.int32, **opts)
    if P == 0:  # rasterize_points.cu:81 skips the rasterizer entirely
        empty = torch.empty((0,), dtype=torch.uint8, **opts)
        return 0, out_color.zero_(), radii, empty, empty.clone(), empty.clone()

    background, means3D, opacity = _f32c(background), _f32c(means3D), _f32c(opacity)
    colors, scales, rotations, cov3D_precomp, sh = (_f32c(t) for t in (colors, scales, rotations, cov3D_precomp, sh))
    viewmatrix, projmatrix, campos = _f32c(viewmatrix), _f32c(projmatrix), _f32c(campos)
    M = sh.size(1) if sh.numel() != 0 else 0

    L = _lib.lib()
    h = _lib.handle(dev, _slot)
    prm = _params(P, degree, M, W, H, tan_fovx, tan_fovy, scale_modifier, prefiltered, debug, _raw,
                  _aux(visible=_visible))
    inp = _inputs(background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                  campos)
    geom = torch.empty((L.fr_geometry_bytes(P),), dtype=torch.uint8, **opts)
    img = torch.empty((L.fr_image_bytes(W, H),), dtype=torch.uint8, **opts)
    cap = max(_capacity_hint.get(dev, 0), 4 * P + 65536)
    counts = _lib.fr_counts()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        while True:
            binning = torch.empty((L.fr_binning_bytes(cap, W, H),), dtype=torch.uint8, **opts)
            rc = L.fr_forward(h, C.byref(prm), C.byref(inp), out_color.data_ptr(), radii.data_ptr(), geom.data_ptr(),
                              img.data_ptr(), binning.data_ptr(), cap, C.byref(counts), stream)
            if rc == _lib.FR_ERR_BINNING_CAPACITY:
                cap = int(counts.num_instances * 1.25) + 1024
                continue
            _check(rc, "fr_forward")
            break
    if _no_wait:  # counts arrive later (read_counts / check_async_overflow)
        return 0, out_color, radii, geom, binning, img
    _capacity_hint[dev] = max(_capacity_hint.get(dev, 0), int(counts.num_instances * 1.25) + 1024)
    last_counts[dev] = counts
    return int(counts.num_rendered), out_color, radii, geom, binning, img


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                 viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos,
                                 geomBuffer, R, binningBuffer, imageBuffer, debug, _want=None, _out=None, _raw=False,
                                 _stats=None, _accumulate=()):
    """`_C.rasterize_gaussians_backward` (rasterize_points.cu:117-196).

    Returns (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6], dL_dsh[P,M,3],
    dL_dscales[P,3], dL_drotations[P,4]).  `_stats=(xyz_gradient_accum[P,1], denom[P,1])` (extension): the kernel
    also does `_add_densification_stats` (model/fateavatar.py:734-737) for the Gaussians with radii > 0.
    `_accumulate` (extension): names of `_out` buffers the frame's gradient is ADDED to (FR_FLAG_ACCUMULATE)."""
    dev = _dev_index(means3D)
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if sh.numel() != 0 else 0
    opts = dict(device=means3D.device, dtype=torch.float32)
    shapes = dict(dL_dmeans2D=(P, 3), dL_dcolors=(P, NUM_CHANNELS), dL_dopacity=(P, 1), dL_dmeans3D=(P, 3),
                  dL_dcov3D=(P, 6), dL_dsh=(P, M, 3), dL_dscales=(P, 3), dL_drotations=(P, 4))
    names = tuple(shapes)
    for k in _accumulate:
        if not (_out and _out.get(k) is not None):
            raise RuntimeError(f"rasterize_gaussians_backward: cannot accumulate into {k}: no buffer was given for it")
    if P == 0:
        return tuple((_out[k] if k in _accumulate else torch.zeros(s, **opts)) for k, s in shapes.items())
    acc_flags = sum(1 << (_lib.FR_FLAG_ACCUMULATE_SHIFT + names.index(k)) for k in _accumulate)
    # the kernel writes every row of every array it is given, so uninitialised memory is fine
    g = {k: (torch.empty(s, **opts) if (_want is None or k in _want) else None) for k, s in shapes.items()}
    if _out:  # caller-provided gradient buffers (e.g. views into a flat gradient buffer): written in place
        for k, buf in _out.items():
            if buf is not None:
                assert buf.shape == shapes[k] and buf.is_contiguous() and buf.dtype == torch.float32, k
                # a FRESH view object: autograd's AccumulateGrad only adopts an incoming gradient without
                # cloning it when nobody else holds a reference to that tensor object
                g[k] = buf.view(buf.shape)

    background, means3D = _f32c(background), _f32c(means3D)
    colors, scales, rotations, cov3D_precomp, sh = (_f32c(t) for t in (colors, scales, rotations, cov3D_precomp, sh))
    viewmatrix, projmatrix, campos = _f32c(viewmatrix), _f32c(projmatrix), _f32c(campos)
    dL_dout_color = _f32c(dL_dout_color)
    radii = radii.contiguous()

    L = _lib.lib()
    h = _lib.handle(dev, _slot)
    aux = _aux(grad_accum=_stats3(_stats)[0], denom=_stats3(_stats)[1], overflow_out=_stats3(_stats)[2]) if _stats is not None else None
    prm = _params(P, degree, M, W, H, tan_fovx, tan_fovy, scale_modifier, False, debug, _raw, aux, acc_flags)
    inp = _inputs(background, means3D, sh, colors, None, scales, rotations, cov3D_precomp, viewmatrix, projmatrix,
                  campos)
    grads = _lib.fr_grads(*[_ptr(g[k]) for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D",
                                                  "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")])
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = L.fr_backward(h, C.byref(prm), C.byref(inp), radii.data_ptr(), geomBuffer.data_ptr(),
                           imageBuffer.data_ptr(), binningBuffer.data_ptr(), dL_dout_color.data_ptr(),
                           C.byref(grads), stream)
    _check(rc, "fr_backward")
    return tuple(g.values())


# ------------------------------------------------------------------ batched frames (fr_forward_batch / fr_backward_batch)
def rasterize_gaussians_batch(views, slots=None, raw=False, visibles=None, bindings=None):
    """K views through ONE launch chain (include/fr_rasterizer.h, fr_forward_batch): `views` is a list of the positional
    argument tuples of `rasterize_gaussians` (background ... debug), one per view; view k uses the device's handle
    `slots[k]` (default k).  Returns the list of `rasterize_gaussians` result tuples.  The results are those of K separate
    calls; what changes is that every kernel of the frame is launched once for all views.
    `bindings` (extension, fr_aux::binding): per view an `_lib.fr_binding` or None — the view's means3D / scales / rotations
    tensors are then OUTPUTS (written by the preprocess kernel from the mesh binding)."""
    K = len(views)
    if not 1 <= K <= _lib.FR_MAX_BATCH:
        raise RuntimeError(f"rasterize_gaussians_batch: 1 .. {_lib.FR_MAX_BATCH} views")
    slots = list(range(K)) if slots is None else [int(x) for x in slots]
    if len(set(slots)) != K:
        raise RuntimeError("rasterize_gaussians_batch: the views of a batch need a handle slot each")
    visibles = visibles or [None] * K
    bindings = bindings or [None] * K
    L = _lib.lib()
    st = []
    dev = None
    for k, a in enumerate(views):
        (background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix,
         tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug) = a
        if bindings[k] is not None and not all(t.is_contiguous() and t.dtype == torch.float32 for t in (means3D, scales, rotations)):
            raise RuntimeError("rasterize_gaussians_batch: a bound view's means3D / scales / rotations are written in place")
        if means3D.dim() != 2 or means3D.size(1) != 3:
            raise RuntimeError("means3D must have dimensions (num_points, 3)")
        d = _dev_index(means3D)
        if dev is None:
            dev = d
        elif d != dev:
            raise RuntimeError("rasterize_gaussians_batch: the views of a batch live on one device")
        P, H, W = means3D.size(0), int(image_height), int(image_width)
        if P == 0:
            raise RuntimeError("rasterize_gaussians_batch: batched views need at least one Gaussian")
        opts = dict(device=means3D.device)
        background, means3D, opacity = _f32c(background), _f32c(means3D), _f32c(opacity)
        colors, scales, rotations, cov3D_precomp, sh = (_f32c(t) for t in (colors, scales, rotations, cov3D_precomp, sh))
        viewmatrix, projmatrix, campos = _f32c(viewmatrix), _f32c(projmatrix), _f32c(campos)
        M = sh.size(1) if sh.numel() != 0 else 0
        v = dict(P=P, H=H, W=W, opts=opts,
                 keep=(background, means3D, opacity, colors, scales, rotations, cov3D_precomp, sh, viewmatrix, projmatrix, campos),
                 prm=_params(P, degree, M, W, H, tan_fovx, tan_fovy, scale_modifier, prefiltered, debug, raw,
                             _aux(visible=visibles[k], binding=bindings[k])),
                 inp=_inputs(background, means3D, sh, colors, opacity, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos),
                 out_color=torch.empty((NUM_CHANNELS, H, W), dtype=torch.float32, **opts),
                 radii=torch.empty((P,), dtype=torch.int32, **opts),
                 geom=torch.empty((L.fr_geometry_bytes(P),), dtype=torch.uint8, **opts),
                 img=torch.empty((L.fr_image_bytes(W, H),), dtype=torch.uint8, **opts),
                 cap=max(_capacity_hint.get(d, 0), 4 * P + 65536))
        st.append(v)
    handles = (C.c_void_p * K)(*[_lib.handle(dev, sl) for sl in slots])
    prm_p = (C.POINTER(_lib.fr_params) * K)(*[C.pointer(v["prm"]) for v in st])
    inp_p = (C.POINTER(_lib.fr_inputs) * K)(*[C.pointer(v["inp"]) for v in st])
    arr = lambda key: (C.c_void_p * K)(*[v[key].data_ptr() for v in st])  # noqa: E731
    counts = (_lib.fr_counts * K)()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        while True:
            for v in st:
                v["binning"] = torch.empty((L.fr_binning_bytes(v["cap"], v["W"], v["H"]),), dtype=torch.uint8, **v["opts"])
            caps = (C.c_uint64 * K)(*[v["cap"] for v in st])
            rc = L.fr_forward_batch(K, handles, prm_p, inp_p, arr("out_color"), arr("radii"), arr("geom"), arr("img"),
                                    arr("binning"), caps, counts, stream)
            if rc == _lib.FR_ERR_BINNING_CAPACITY:
                for k, v in enumerate(st):
                    if counts[k].overflow:
                        v["cap"] = int(counts[k].num_instances * 1.25) + 1024
                continue
            _check(rc, "fr_forward_batch")
            break
    out = []
    for k, v in enumerate(st):
        if _no_wait:
            out.append((0, v["out_color"], v["radii"], v["geom"], v["binning"], v["img"]))
            continue
        c = _lib.fr_counts(counts[k].num_rendered, counts[k].num_instances, counts[k].max_tile_list, counts[k].overflow)
        _capacity_hint[dev] = max(_capacity_hint.get(dev, 0), int(c.num_instances * 1.25) + 1024)
        last_counts[dev] = c
        out.append((int(c.num_rendered), v["out_color"], v["radii"], v["geom"], v["binning"], v["img"]))
    return out


def rasterize_gaussians_backward_batch(views, slots=None, raw=False, wants=None, outs=None, stats=None, accumulates=None,
                                       bindings=None, bind_grads=None):
    """`rasterize_gaussians_backward` for K views in ONE launch chain (fr_backward_batch): `views` is a list of its
    positional argument tuples (background ... debug); `wants` / `outs` / `stats` / `accumulates`: per-view lists of the
    corresponding keyword arguments.  Returns the list of gradient tuples.  `bindings` / `bind_grads` (extension,
    fr_aux::binding): per view the descriptor the forward was given and a dict of the d_verts / d_offset / d_rotation /
    d_scaling tensors the kernel writes (d_verts: adds)."""
    K = len(views)
    if not 1 <= K <= _lib.FR_MAX_BATCH:
        raise RuntimeError(f"rasterize_gaussians_backward_batch: 1 .. {_lib.FR_MAX_BATCH} views")
    slots = list(range(K)) if slots is None else [int(x) for x in slots]
    wants, outs, stats, accumulates, bindings, bind_grads = (x or [None] * K for x in (wants, outs, stats, accumulates, bindings,
                                                                                      bind_grads))
    L = _lib.lib()
    st = []
    dev = None
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    for k, a in enumerate(views):
        (background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix, projmatrix, tan_fovx,
         tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer, imageBuffer, debug) = a
        d = _dev_index(means3D)
        dev = d if dev is None else dev
        P = means3D.size(0)
        H, W = dL_dout_color.size(1), dL_dout_color.size(2)
        M = sh.size(1) if sh.numel() != 0 else 0
        opts = dict(device=means3D.device, dtype=torch.float32)
        shapes = dict(dL_dmeans2D=(P, 3), dL_dcolors=(P, NUM_CHANNELS), dL_dopacity=(P, 1), dL_dmeans3D=(P, 3),
                      dL_dcov3D=(P, 6), dL_dsh=(P, M, 3), dL_dscales=(P, 3), dL_drotations=(P, 4))
        acc = tuple(accumulates[k] or ())
        out_k = outs[k] or {}
        for n in acc:
            if out_k.get(n) is None:
                raise RuntimeError(f"rasterize_gaussians_backward_batch: cannot accumulate into {n}: no buffer was given for it")
        acc_flags = sum(1 << (_lib.FR_FLAG_ACCUMULATE_SHIFT + names.index(n)) for n in acc)
        g = {n: (torch.empty(sh_, **opts) if (wants[k] is None or n in wants[k]) else None) for n, sh_ in shapes.items()}
        for n, buf in out_k.items():
            if buf is not None:
                assert buf.shape == shapes[n] and buf.is_contiguous() and buf.dtype == torch.float32, n
                g[n] = buf.view(buf.shape)
        background, means3D = _f32c(background), _f32c(means3D)
        colors, scales, rotations, cov3D_precomp, sh = (_f32c(t) for t in (colors, scales, rotations, cov3D_precomp, sh))
        viewmatrix, projmatrix, campos = _f32c(viewmatrix), _f32c(projmatrix), _f32c(campos)
        dL_dout_color = _f32c(dL_dout_color)
        radii = radii.contiguous()
        aux = _aux(grad_accum=_stats3(stats[k])[0], denom=_stats3(stats[k])[1], overflow_out=_stats3(stats[k])[2],
                   binding=bindings[k], bind_grads=bind_grads[k])
        v = dict(g=g, keep=(background, means3D, colors, scales, rotations, cov3D_precomp, sh, viewmatrix, projmatrix, campos,
                            dL_dout_color, radii, geomBuffer, binningBuffer, imageBuffer),
                 prm=_params(P, degree, M, W, H, tan_fovx, tan_fovy, scale_modifier, False, debug, raw, aux, acc_flags),
                 inp=_inputs(background, means3D, sh, colors, None, scales, rotations, cov3D_precomp, viewmatrix, projmatrix, campos),
                 grads=_lib.fr_grads(*[_ptr(g[n]) for n in names]),
                 radii=radii, geom=geomBuffer, img=imageBuffer, binning=binningBuffer, dpix=dL_dout_color)
        st.append(v)
    # fr_aux::overflow_out is OVERWRITTEN (0 or 1) by every backward: views of one launch that shared a word would race, and a
    # view that did not overflow could clear the flag of one that did (the optimizer would step on a partly zero gradient)
    words = [int(_stats3(stats[k])[2].data_ptr()) for k in range(K) if _stats3(stats[k])[2] is not None]
    if len(set(words)) != len(words):
        raise RuntimeError("rasterize_gaussians_backward_batch: the views of a batch need ONE overflow word EACH "
                           "(fused_densification_stats[2]); give every view its own statistics tuple")
    handles = (C.c_void_p * K)(*[_lib.handle(dev, sl) for sl in slots])
    prm_p = (C.POINTER(_lib.fr_params) * K)(*[C.pointer(v["prm"]) for v in st])
    inp_p = (C.POINTER(_lib.fr_inputs) * K)(*[C.pointer(v["inp"]) for v in st])
    grd_p = (C.POINTER(_lib.fr_grads) * K)(*[C.pointer(v["grads"]) for v in st])
    arr = lambda key: (C.c_void_p * K)(*[v[key].data_ptr() for v in st])  # noqa: E731
    with torch.cuda.device(dev):
        rc = L.fr_backward_batch(K, handles, prm_p, inp_p, arr("radii"), arr("geom"), arr("img"), arr("binning"), arr("dpix"),
                                 grd_p, torch.cuda.current_stream(dev).cuda_stream)
    _check(rc, "fr_backward_batch")
    return [tuple(v["g"].values()) for v in st]


class _RasterizeGaussiansBatch(torch.autograd.Function):
    """`_RasterizeGaussians` for K views rendered together.  Tensor arguments: per view (means3D, means2D, sh,
    colors_precomp, opacities, scales, rotations, cov3Ds_precomp); outputs: per view (color, radii)."""

    @staticmethod
    def forward(ctx, settings, raw_activations, slots, *tensors):
        K = len(settings)
        assert len(tensors) == 8 * K
        ctx.raw, ctx.K, ctx.settings, ctx.slots = bool(raw_activations), K, settings, slots
        ctx.set_materialize_grads(False)
        views, viss = [], []
        for k, rs in enumerate(settings):
            means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp = tensors[8 * k:8 * k + 8]
            views.append((rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                          rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                          rs.sh_degree, rs.campos, rs.prefiltered, rs.debug))
            viss.append(torch.empty((means3D.shape[0],), dtype=torch.bool, device=means3D.device))
        res = rasterize_gaussians_batch(views, slots=slots, raw=ctx.raw, visibles=viss)
        ctx.stats, ctx.num_rendered, ctx.grad_slots, ctx.grad_owners = [], [], [], []
        saved, outs = [], []
        for k in range(K):
            means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp = tensors[8 * k:8 * k + 8]
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = res[k]
            radii._fr_visible = viss[k]
            ctx.stats.append(getattr(means2D, "_fr_densification_stats", None))
            ctx.num_rendered.append(num_rendered)
            slots_k = {"dL_dmeans3D": GradOut.of(means3D), "dL_dsh": GradOut.of(sh) if sh.numel() else None}
            owners = {"dL_dmeans3D": means3D, "dL_dsh": sh}
            if ctx.raw:
                slots_k.update(dL_dopacity=GradOut.of(opacities), dL_dscales=GradOut.of(scales), dL_drotations=GradOut.of(rotations))
                owners.update(dL_dopacity=opacities, dL_dscales=scales, dL_drotations=rotations)
            ctx.grad_slots.append(slots_k)
            ctx.grad_owners.append({n: t for n, t in owners.items() if slots_k.get(n) is not None and t.is_leaf})
            saved += [colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer]
            outs += [color, radii]
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(*outs[1::2])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        K = ctx.K
        none = (None, None, None) + (None,) * (8 * K)
        grad_colors = grad_outs[0::2]
        if all(g is None for g in grad_colors):
            return none
        views, wants, outs, accs = [], [], [], []
        for k, rs in enumerate(ctx.settings):
            colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = \
                ctx.saved_tensors[10 * k:10 * k + 10]
            g = grad_colors[k]
            if g is None:   # (a view nobody differentiated: its frame still runs with a zero image gradient)
                g = torch.zeros((NUM_CHANNELS, rs.image_height, rs.image_width), dtype=torch.float32, device=means3D.device)
            views.append((rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp, rs.viewmatrix,
                          rs.projmatrix, rs.tanfovx, rs.tanfovy, g, sh, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered[k],
                          binningBuffer, imgBuffer, rs.debug))
            want = {"dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dscales", "dL_drotations"}
            if colors_precomp.numel():
                want.add("dL_dcolors")
            if cov3Ds_precomp.numel():
                want.add("dL_dcov3D")
            if sh.numel():
                want.add("dL_dsh")
            claims = {n: slot.claim(ctx.grad_owners[k].get(n)) for n, slot in ctx.grad_slots[k].items() if slot is not None}
            # (in-kernel accumulation across the views of ONE batch would race: a later view of the same parameters gets a
            # fresh tensor, which autograd adds)
            wants.append(want)
            outs.append({n: c[0] for n, c in claims.items() if not c[1]})
            accs.append(())
        res = rasterize_gaussians_backward_batch(views, slots=ctx.slots, raw=ctx.raw, wants=wants, outs=outs, stats=ctx.stats,
                                                 accumulates=accs)
        flat = [None, None, None]
        for k in range(K):
            (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
             grad_rotations) = res[k]
            flat += [grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                     grad_cov3Ds_precomp]
        return tuple(flat)


def rasterize_views_autograd(settings, per_view_tensors, raw_activations=False, slots=None):
    """K views through one launch chain, differentiable: `settings` a list of GaussianRasterizationSettings,
    `per_view_tensors` a list of (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp) with
    empty tensors for what a view does not use.  Returns [(color, radii), ...]."""
    K = len(settings)
    flat = [t for v in per_view_tensors for t in v]
    out = _RasterizeGaussiansBatch.apply(list(settings), bool(raw_activations), list(range(K)) if slots is None else list(slots), *flat)
    return [(out[2 * k], out[2 * k + 1]) for k in range(K)]


def mark_visible(means3D, viewmatrix, projmatrix):
    """`_C.mark_visible` (rasterize_points.cu:198-217)."""
    dev = _dev_index(means3D)
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        means3D, viewmatrix, projmatrix = _f32c(means3D), _f32c(viewmatrix), _f32c(projmatrix)
        with torch.cuda.device(dev):
            rc = _lib.lib().fr_mark_visible(P, means3D.data_ptr(), viewmatrix.data_ptr(), projmatrix.data_ptr(),
                                            present.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        _check(rc, "fr_mark_visible")
    return present


def image_aux(imgBuffer: torch.Tensor, H: int, W: int):
    """(final_T[H,W] float32, n_contrib[H,W] int32) views into an image buffer: the per-pixel alpha channel
    (alpha = 1 - final_T; reference accum_alpha, forward.cu:369) and contributor count."""
    L = _lib.lib()
    base = imgBuffer.data_ptr()
    oT = L.fr_image_final_T(base, W, H) - base
    oN = L.fr_image_n_contrib(base, W, H) - base
    T = imgBuffer[oT:oT + 4 * H * W].view(torch.float32).view(H, W)
    N = imgBuffer[oN:oN + 4 * H * W].view(torch.int32).view(H, W)
    return T, N


def cpu_deep_copy_tuple(input_tuple):
    return tuple(item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple)


class GradOut:
    """A gradient slot a parameter tensor can carry as `tensor._fr_grad_out`: the rasterizer's backward writes the
    parameter's gradient straight into `buf` instead of allocating a tensor that autograd then copies or adds.
    The kernel OVERWRITES for the first backward after the parameter's .grad was cleared (forward() re-arms the slot
    when it sees `.grad is None`); any further backward gets a fresh tensor, which autograd adds — several renders
    from the same parameters, as model/fateavatar.py:251-276 does, accumulate correctly.

    Opt-in (`add_to_kept = True`, what FlatGaussians.accumulate_into_kept_grads() sets): a further backward is ADDED
    to the buffer by the kernel itself (FR_FLAG_ACCUMULATE) when that is provably where the parameter's gradient lives
    (`param.grad` is a view of `buf`: the gradients of an earlier `.backward()` were kept), and autograd is handed
    nothing for that parameter.  It is opt-in because the backward cannot tell `.backward()` from
    `torch.autograd.grad()`, which must not touch `param.grad`; and it never applies to the frames of ONE backward
    pass of a summed loss, whose first gradient still sits in autograd's input buffer (param.grad is None)."""

    def __init__(self, buf: torch.Tensor):
        self.buf = buf
        self.claimed = False
        self.add_to_kept = False

    @staticmethod
    def of(t):
        slot = getattr(t, "_fr_grad_out", None)
        if slot is None:
            return None
        if t.is_leaf and t.grad is None:
            slot.claimed = False   # gradients were cleared (zero_grad(set_to_none=True)): a new accumulation starts
        return slot

    def claim(self, param=None):
        """-> (buffer or None, accumulate)."""
        if not self.claimed:
            self.claimed = True
            return self.buf, False
        g = param.grad if (self.add_to_kept and param is not None and param.is_leaf) else None
        if (g is not None and g.data_ptr() == self.buf.data_ptr() and g.shape == self.buf.shape and g.is_contiguous()
                and g.dtype == self.buf.dtype):
            return self.buf, True
        return None, False


class _RasterizeGaussians(torch.autograd.Function):
    """diff_gaussian_rasterization/__init__.py:44-155."""

    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, raw_activations=False):
        rs = raster_settings
        ctx.raw = bool(raw_activations)
        ctx.fr_slot = _slot   # the backward goes through the handle the forward used
        # the gradient slot of the int32 `radii` output would otherwise be materialised as a zero tensor per backward
        ctx.set_materialize_grads(False)
        args = (rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
                rs.sh_degree, rs.campos, rs.prefiltered, rs.debug)
        # extension: the visibility mask (radii > 0) comes out of the preprocess kernel; render() picks it up from
        # `radii._fr_visible` instead of launching a compare kernel
        vis = torch.empty((means3D.shape[0],), dtype=torch.bool, device=means3D.device) if means3D.is_cuda else None
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = \
                    rasterize_gaussians(*args, _raw=ctx.raw, _visible=vis)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = \
                rasterize_gaussians(*args, _raw=ctx.raw, _visible=vis)
        if vis is not None and means3D.shape[0] > 0:
            radii._fr_visible = vis
        # extension: `means2D._fr_densification_stats = (xyz_gradient_accum, denom)` makes the backward kernel
        # accumulate the densification statistics itself
        ctx.stats = getattr(means2D, "_fr_densification_stats", None)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        # optional extension: an input tensor may carry `_fr_grad_out`, a GradOut slot whose preallocated buffer
        # receives its gradient (zero-copy into e.g. a flat data-parallel gradient buffer)
        ctx.grad_slots = {"dL_dmeans3D": GradOut.of(means3D), "dL_dsh": GradOut.of(sh) if sh.numel() else None}
        owners = {"dL_dmeans3D": means3D, "dL_dsh": sh}
        if ctx.raw:  # raw parameters reach the kernels directly: their gradients can be written in place too
            ctx.grad_slots.update(dL_dopacity=GradOut.of(opacities), dL_dscales=GradOut.of(scales),
                                  dL_drotations=GradOut.of(rotations))
            owners.update(dL_dopacity=opacities, dL_dscales=scales, dL_drotations=rotations)
        # (leaves only: a reference to a non-leaf input from its own grad_fn's context would be a cycle)
        ctx.grad_owners = {k: t for k, t in owners.items() if ctx.grad_slots.get(k) is not None and t.is_leaf}
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer,
                              binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        num_rendered = ctx.num_rendered
        rs = ctx.raster_settings
        if grad_out_color is None:
            return (None,) * 10
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = \
            ctx.saved_tensors
        args = (rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_out_color, sh, rs.sh_degree, rs.campos,
                geomBuffer, num_rendered, binningBuffer, imgBuffer, rs.debug)
        # gradients nobody can receive are not computed: dL_dcolors without colors_precomp, dL_dcov3D without
        # cov3D_precomp, dL_dsh without sh (the reference fills them in and autograd drops them)
        want = {"dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dscales", "dL_drotations"}
        if colors_precomp.numel():
            want.add("dL_dcolors")
        if cov3Ds_precomp.numel():
            want.add("dL_dcov3D")
        if sh.numel():
            want.add("dL_dsh")
        if rs.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                with handle_slot(ctx.fr_slot):
                    grads = rasterize_gaussians_backward(*args, _raw=ctx.raw, _stats=ctx.stats, _want=want)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            # the FIRST backward of a step may write a gradient straight into its slot's buffer; any further backward
            # of the same step (several frames rendered from the same parameters) gets a fresh tensor, which autograd
            # then adds to the first
            claims = {k: slot.claim(ctx.grad_owners.get(k)) for k, slot in ctx.grad_slots.items() if slot is not None}
            out = {k: c[0] for k, c in claims.items()}
            added = tuple(k for k, c in claims.items() if c[1])
            with handle_slot(ctx.fr_slot):
                grads = rasterize_gaussians_backward(*args, _out=out, _raw=ctx.raw, _stats=ctx.stats, _want=want,
                                                     _accumulate=added)
            if added:   # already in the parameter's .grad: nothing for autograd to add
                names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
                         "dL_drotations")
                grads = tuple(None if n in added else g for n, g in zip(names, grads))
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales,
         grad_rotations) = grads
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales,
                grad_rotations, grad_cov3Ds_precomp, None, None)


def rasterize_gaussians_autograd(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                 raster_settings, raw_activations=False):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, raw_activations)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    """diff_gaussian_rasterization/__init__.py:171-220."""

    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, raw_activations=False):
        """`raw_activations=True` (extension, not in the reference): opacities / scales / rotations are the RAW
        parameters; sigmoid / exp / normalize and their derivatives run inside the HIP kernels."""
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])
        if shs is None:
            shs = empty
        if colors_precomp is None:
            colors_precomp = empty
        if scales is None:
            scales = empty
        if rotations is None:
            rotations = empty
        if cov3D_precomp is None:
            cov3D_precomp = empty
        if raw_activations and (scales is None or scales.numel() == 0):
            raise Exception('raw_activations needs the scale/rotation pair')
        return rasterize_gaussians_autograd(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                            cov3D_precomp, rs, raw_activations)
