"""Image loss of the optimisation step.

reference: `nn.L1Loss(reduction='mean')` on the rendered image (model/loss.py:92) followed by `loss.backward()` — eight
launch-bound PyTorch kernels between the rasterizer's forward and its backward.  `l1_loss_and_grad` produces the loss
and the gradient autograd would hand to the rasterizer (`sign(img - gt) / n`, for a unit upstream gradient) in one
launch of the HIP library (`fr_l1_loss_grad`, include/fr_rasterizer.h); the caller continues with
`render.backward(grad)`.  There is no CPU path."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

# (device index, stream handle) -> zeroed scratch (the kernel leaves it zeroed).  One workspace PER STREAM: the kernel
# elects its last workgroup through counters in the workspace and sums per-workgroup partials left there, so two launches
# that overlap on the device — the lanes of AvatarBatchStep run on their own streams, each from its own captured graph —
# must not share one (the loss scalars would mix; include/fr_rasterizer.h says the same of fr_l1_loss_grad).
_workspace = {}


def l1_workspace(dev: torch.device) -> torch.Tensor:
    """A fresh zeroed workspace for `l1_loss_and_grad(..., workspace=)`: for callers that launch from several streams or
    graphs at once and want to own the scratch explicitly."""
    return torch.zeros((_lib.lib().fr_l1_workspace_bytes(),), dtype=torch.uint8, device=dev)


def l1_loss_and_grad(img: torch.Tensor, gt: torch.Tensor, loss_out: Optional[torch.Tensor] = None,
                     grad_out: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor]:
    """mean |img - gt| (0-dim device tensor) and its gradient with respect to `img`.  `loss_out` / `grad_out`: write into
    these tensors instead of fresh ones (buffers of a captured step).  `workspace`: scratch from `l1_workspace()`; by
    default one is kept per (device, current stream) — launches that may overlap must not share one."""
    if not (img.is_cuda and gt.is_cuda):
        raise RuntimeError("l1_loss_and_grad needs device tensors (there is no CPU path)")
    if img.shape != gt.shape:
        raise RuntimeError(f"l1_loss_and_grad: shapes differ: {tuple(img.shape)} vs {tuple(gt.shape)}")
    img = img.detach()
    if img.dtype != torch.float32 or not img.is_contiguous():
        img = img.float().contiguous()
    if gt.dtype != torch.float32 or not gt.is_contiguous():
        gt = gt.float().contiguous()
    dev = img.device
    grad = grad_out if grad_out is not None else torch.empty_like(img)
    loss = loss_out if loss_out is not None else torch.empty((), dtype=torch.float32, device=dev)
    if grad.shape != img.shape or grad.dtype != torch.float32 or not grad.is_contiguous() or loss.numel() != 1:
        raise RuntimeError("l1_loss_and_grad: bad output buffers")
    L = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = workspace
    if ws is None:
        key = (dev.index, stream)
        ws = _workspace.get(key)
        if ws is None:
            # (allocated on a side stream-independent path: torch.zeros inside a capture would become part of the graph)
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("l1_loss_and_grad: first call on this stream happens inside a graph capture; call it once "
                                   "eagerly on the stream first, or pass workspace=l1_workspace(device)")
            ws = _workspace[key] = l1_workspace(dev)
    elif not (ws.is_cuda and ws.device == dev and ws.dtype == torch.uint8 and ws.numel() >= L.fr_l1_workspace_bytes()):
        raise RuntimeError("l1_loss_and_grad: workspace must come from l1_workspace() on the image's device")
    with torch.cuda.device(dev):
        rc = L.fr_l1_loss_grad(img.numel(), img.data_ptr(), gt.data_ptr(), grad.data_ptr(), loss.data_ptr(), ws.data_ptr(),
                               stream)
    if rc != _lib.FR_OK:
        raise RuntimeError(f"fr_l1_loss_grad failed: {_lib.last_error()}")
    return loss, grad


def l1_loss_and_grad_batch(imgs, gts, loss_outs, grad_outs, workspaces):
    """`l1_loss_and_grad` for the images of the 1 .. 4 frames of a batch in ONE launch (`fr_l1_loss_grad_batch`): lists of
    equally-sized contiguous float32 device tensors; every image has its own loss scalar, gradient buffer and workspace
    (`l1_workspace()`).  Returns (loss_outs, grad_outs)."""
    import ctypes as C
    K = len(imgs)
    if not (1 <= K <= _lib.FR_MAX_BATCH and len(gts) == len(loss_outs) == len(grad_outs) == len(workspaces) == K):
        raise RuntimeError(f"l1_loss_and_grad_batch: 1 .. {_lib.FR_MAX_BATCH} images, one gt / loss / grad / workspace each")
    imgs = [i.detach() for i in imgs]
    dev, n = imgs[0].device, imgs[0].numel()
    for t in list(imgs) + list(gts) + list(grad_outs):
        if not t.is_cuda:
            raise RuntimeError("l1_loss_and_grad_batch needs device tensors (there is no CPU path)")
        if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != n:
            raise RuntimeError("l1_loss_and_grad_batch: contiguous float32 tensors of one device and one size")
    L = _lib.lib()
    for w, l in zip(workspaces, loss_outs):
        if not (w.is_cuda and w.device == dev and w.dtype == torch.uint8 and w.numel() >= L.fr_l1_workspace_bytes()) or l.numel() != 1:
            raise RuntimeError("l1_loss_and_grad_batch: workspaces from l1_workspace(), one-element loss tensors")
    arr = lambda ts: (C.c_void_p * K)(*[t.data_ptr() for t in ts])  # noqa: E731
    with torch.cuda.device(dev):
        rc = L.fr_l1_loss_grad_batch(K, n, arr(imgs), arr(gts), arr(grad_outs), arr(loss_outs), arr(workspaces),
                                     torch.cuda.current_stream(dev).cuda_stream)
    if rc != _lib.FR_OK:
        raise RuntimeError(f"fr_l1_loss_grad_batch failed: {_lib.last_error()}")
    return loss_outs, grad_outs


_copy_calls = {}   # (dst ptr, src ptr, floats) per pair -> the prepared argument arrays of fr_multi_copy


def multi_copy(pairs) -> None:
    """`dst.copy_(src)` for up to twelve (dst, src) pairs of contiguous float32 device tensors in ONE launch
    (`fr_multi_copy`): the per-frame inputs of a captured step (of every frame of a batch).  A step calls this with the
    same few sets of tensors over and over (its static buffers, the frames of a resident sequence): the ctypes argument
    arrays of a set are made once and looked up by the tensors' addresses afterwards — this call is on the host's critical
    path of a 130 us step."""
    import ctypes as C
    pairs = [(d, s) for d, s in pairs if d.numel()]
    if not pairs:
        return
    if len(pairs) > 12:
        raise RuntimeError("multi_copy: at most twelve pairs")
    dev = pairs[0][0].device
    f32 = torch.float32
    for d, s in pairs:
        if not (d.is_cuda and s.is_cuda and d.device == dev and s.device == dev):
            raise RuntimeError("multi_copy needs tensors of one device")
        if d.dtype is not f32 or s.dtype is not f32 or not d.is_contiguous() or not s.is_contiguous() or d.numel() != s.numel():
            raise RuntimeError("multi_copy: contiguous float32 tensors of equal size")
    key = tuple((d.data_ptr(), s.data_ptr(), d.numel()) for d, s in pairs)
    call = _copy_calls.get(key)
    if call is None:
        n = len(pairs)
        call = (n, (C.c_void_p * n)(*[d.data_ptr() for d, _ in pairs]), (C.c_void_p * n)(*[s.data_ptr() for _, s in pairs]),
                (C.c_uint64 * n)(*[d.numel() for d, _ in pairs]))
        if len(_copy_calls) > 4096:     # (addresses are only a key while their tensors live: bounded, rebuilt on demand)
            _copy_calls.clear()
        _copy_calls[key] = call
    n, dst, src, cnt = call
    # (the raw handle of the device's current stream without building a torch.cuda.Stream object: 5 us of a host path that
    # has to stay under the step's 130 us)
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    stream = raw(dev.index) if raw is not None else torch.cuda.current_stream(dev).cuda_stream
    if torch.cuda.current_device() == dev.index:
        rc = _lib.lib().fr_multi_copy(n, dst, src, cnt, stream)
    else:
        with torch.cuda.device(dev):
            rc = _lib.lib().fr_multi_copy(n, dst, src, cnt, stream)
    if rc != _lib.FR_OK:
        raise RuntimeError(f"fr_multi_copy failed: {_lib.last_error()}")


def scaled_sum(dst: torch.Tensor, srcs, scale: float) -> torch.Tensor:
    """dst = scale * sum(srcs) for 1 .. 4 contiguous float32 device tensors of dst's size, in ONE pass (`fr_scaled_sum`):
    the mean of the gradient buffers of the views a rank rendered in flight together."""
    import ctypes as C
    srcs = list(srcs)
    if not 1 <= len(srcs) <= _lib.FR_ADAM_MAX_GRADS:
        raise RuntimeError(f"scaled_sum: 1..{_lib.FR_ADAM_MAX_GRADS} sources")
    dev = dst.device
    for t in [dst] + srcs:
        if not t.is_cuda or t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != dst.numel():
            raise RuntimeError("scaled_sum: contiguous float32 tensors of one device and one size")
    ptrs = (C.c_void_p * len(srcs))(*[t.data_ptr() for t in srcs])
    with torch.cuda.device(dev):
        rc = _lib.lib().fr_scaled_sum(len(srcs), ptrs, dst.data_ptr(), dst.numel(), float(scale),
                                      torch.cuda.current_stream(dev).cuda_stream)
    if rc != _lib.FR_OK:
        raise RuntimeError(f"fr_scaled_sum failed: {_lib.last_error()}")
    return dst
