"""Fused Adam over a flat parameter buffer.

Replaces `torch.optim.Adam(groups, lr=0.0).step()` over the Gaussian parameter groups
(reference: train/optim.py:11-37, called at train/iteration.py:58-60): same update rule and defaults
(betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad), ONE kernel over the flat buffer instead of
a dozen `foreach` kernels, and nothing step-dependent on the host, so the step can sit inside a HIP graph.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib


class FusedAdam:
    """`segments`: consecutive (n_elements, lr) or (n_elements, lr, period, split, lr2) runs of the flat buffer —
    the reference's param groups.  The 5-tuple form gives element e of the run lr if e % period < split else lr2
    (SH coefficients stored [P, M, 3]: DC at lr, the rest at lr / 20, train/optim.py:49-50)."""

    def __init__(self, flat_param: torch.Tensor, flat_grad: torch.Tensor, segments: Sequence[tuple],
                 betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0):
        if not (flat_param.is_cuda and flat_grad.is_cuda):
            raise RuntimeError("FusedAdam needs device tensors (there is no CPU path)")
        for t in (flat_param, flat_grad):
            if t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 1:
                raise RuntimeError("FusedAdam works on flat contiguous float32 buffers")
        if flat_param.shape != flat_grad.shape:
            raise RuntimeError("parameter and gradient buffers differ in size")
        if not 1 <= len(segments) <= _lib.FR_ADAM_MAX_SEGMENTS:
            raise RuntimeError(f"1..{_lib.FR_ADAM_MAX_SEGMENTS} segments")
        self.param, self.grad = flat_param, flat_grad
        self.exp_avg = torch.zeros_like(flat_param)
        self.exp_avg_sq = torch.zeros_like(flat_param)
        # step, 1 - beta1^t, 1 - beta2^t, -, kernel bookkeeping (FR_ADAM_STATE_FLOATS)
        self.state = torch.zeros(_lib.FR_ADAM_STATE_FLOATS, dtype=torch.float32, device=flat_param.device)
        cfg = _lib.fr_adam_config()
        cfg.n_segments = len(segments)
        end = 0
        for i, seg in enumerate(segments):
            n, lr = int(seg[0]), float(seg[1])
            end += n
            cfg.segment_end[i] = end
            cfg.segment_lr[i] = lr
            if len(seg) == 5:
                cfg.segment_period[i], cfg.segment_split[i], cfg.segment_lr2[i] = int(seg[2]), int(seg[3]), float(seg[4])
        if end != flat_param.numel():
            raise RuntimeError(f"segments cover {end} elements, the buffer has {flat_param.numel()}")
        cfg.beta1, cfg.beta2, cfg.eps, cfg.grad_scale = float(betas[0]), float(betas[1]), float(eps), float(grad_scale)
        self.cfg = cfg

    @torch.no_grad()
    def remap_rows(self, flat_param: torch.Tensor, flat_grad: torch.Tensor, old_index: torch.Tensor,
                   widths: Sequence[int], old_rows: int, zero_fields: Sequence[int] = ()) -> None:
        """Optimizer-state surgery after a prune / densify of row-structured parameters (reference:
        model/fateavatar.py:640-651,688-694): the buffer holds one [rows, width] block per field, `old_index[r]` is
        the old row new row r came from, or -1 for an appended row, whose moments start at zero (`torch.zeros_like`
        in the reference).  Rebinds the optimizer to the new flat buffers; the step count is kept, as the reference
        keeps `state["step"]`.  Fields listed in `zero_fields` get zero moments everywhere (_reset_opacity,
        model/fateavatar.py:713-731)."""
        new_rows = int(old_index.numel())
        if sum(widths) * new_rows != flat_param.numel() or flat_param.shape != flat_grad.shape:
            raise RuntimeError("remap_rows: buffer size does not match rows x widths")
        src = old_index.clamp(min=0)
        fresh = (old_index < 0)
        new_m, new_v = torch.zeros_like(flat_param), torch.zeros_like(flat_param)
        o_old = o_new = 0
        for f, w in enumerate(widths):
            for old, new in ((self.exp_avg, new_m), (self.exp_avg_sq, new_v)):
                blk = old[o_old:o_old + old_rows * w].view(old_rows, w)[src]
                blk[fresh] = 0.0
                if f in zero_fields:
                    blk.zero_()
                new[o_new:o_new + new_rows * w].view(new_rows, w).copy_(blk)
            o_old += old_rows * w
            o_new += new_rows * w
        self.param, self.grad, self.exp_avg, self.exp_avg_sq = flat_param, flat_grad, new_m, new_v
        end = 0
        for i, w in enumerate(widths):
            end += new_rows * w
            self.cfg.segment_end[i] = end

    @torch.no_grad()
    def zero_field_moments(self, widths: Sequence[int], rows: int, fields: Sequence[int]) -> None:
        """Zero the moments of whole fields IN PLACE (_reset_opacity, model/fateavatar.py:713-731): the buffers keep
        their addresses, so a HIP graph that captured fr_adam_step stays valid."""
        off = 0
        for f, w in enumerate(widths):
            if f in fields:
                self.exp_avg[off:off + rows * w].zero_()
                self.exp_avg_sq[off:off + rows * w].zero_()
            off += rows * w

    def set_grad_scale(self, s: float) -> None:
        self.cfg.grad_scale = float(s)

    def set_skip_words(self, words: Sequence[torch.Tensor]) -> None:
        """Device floats (0 .. FR_ADAM_MAX_GRADS one-element tensors) that make a step a no-op when any of them is non-zero
        — the overflow words the rasterizer's backward writes (fr_aux::overflow_out): a frame that overflowed its binning
        capacity inside a replayed graph back-propagated zeros, and a step on such a gradient is skipped (parameters,
        moments and step count untouched) instead of moving on momentum alone."""
        words = list(words)
        if len(words) > _lib.FR_ADAM_MAX_GRADS:
            raise RuntimeError(f"at most {_lib.FR_ADAM_MAX_GRADS} skip words")
        for w in words:
            if not (w.is_cuda and w.dtype == torch.float32 and w.numel() >= 1):
                raise RuntimeError("skip words are float32 device tensors")
        self._skip_words = words      # (keeps them alive)
        for k in range(_lib.FR_ADAM_MAX_GRADS):
            self.cfg.skip[k] = words[k].data_ptr() if k < len(words) else None
        self.cfg.n_skip = len(words)

    @torch.no_grad()
    def step(self, grads: Optional[Sequence[torch.Tensor]] = None) -> None:
        """One Adam step on `self.grad` — or, given `grads` (1 .. FR_ADAM_MAX_GRADS flat buffers laid out like the
        parameters), on their SUM: the views of a batch back-propagate into a buffer each, and grad_scale makes the sum
        the batch mean."""
        dev = self.param.device
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            if grads is None:
                rc = _lib.lib().fr_adam_step(C.byref(self.cfg), self.param.data_ptr(), self.grad.data_ptr(),
                                             self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.param.numel(),
                                             self.state.data_ptr(), stream)
            else:
                for g in grads:
                    if g.shape != self.param.shape or g.dtype != torch.float32 or not g.is_contiguous() or g.device != dev:
                        raise RuntimeError("FusedAdam.step: gradient buffers must look like the parameter buffer")
                ptrs = (C.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
                rc = _lib.lib().fr_adam_step_multi(C.byref(self.cfg), self.param.data_ptr(), ptrs, len(grads),
                                                   self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.param.numel(),
                                                   self.state.data_ptr(), stream)
        if rc != _lib.FR_OK:
            raise RuntimeError(f"fr_adam_step failed: {_lib.last_error()}")

    @property
    def step_count(self) -> int:
        return int(self.state[0].item())
