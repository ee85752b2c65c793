"""The per-frame optimisation step around the path.

reference loop: `iteration_step_fateavatar`, train/iteration.py:21-89 —
    zero_grad(set_to_none) -> render -> L1(mean) -> backward -> _add_densification_stats -> Adam.step()
with the Adam groups of train/optim.py:11-37 and the learning rates of config/fateavatar.yaml:34-39
(generic-3DGS variant for the positions).  Densify / prune / opacity reset are
reference model surgery outside the path and are not reproduced.

What is fused here compared with the reference: the activations and the densification statistics run inside
the rasterizer kernels (FR_FLAG_RAW_ACTIVATIONS, fr_aux), Adam is one kernel over the flat parameter buffer
(fr_adam_step), and the whole step — about 20 kernel launches instead of about 90 — is replayed as ONE HIP graph.
Data-parallel (one frame per rank): the flat gradient is summed over ranks with one RCCL
all-reduce between the backward and the Adam kernel (which applies 1/world); the densification statistics are
plain sums over frames, so every rank accumulates its own and `reduce_densification_stats()` adds them up
when they are needed.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import dp
from .loss import l1_loss_and_grad, l1_workspace, multi_copy
from .model import FlatGaussians, TorchCamera
from .optim import FusedAdam
from .render import render

# config/fateavatar.yaml:34-39 (+ the generic 3DGS position rate; FateAvatar optimises a mesh offset instead)
DEFAULT_LRS = dict(xyz=1.6e-4, feature_dc=2.5e-3, feature_rest=2.5e-3 / 20, opacity=0.05, scaling=5e-3, rotation=1e-3)


class TrainStep:
    def __init__(self, pc: FlatGaussians, camera: TorchCamera, bg: torch.Tensor, lrs: Optional[dict] = None,
                 use_graph: bool = True):
        if not pc.fused_activations:
            raise ValueError("TrainStep drives the fused path: build FlatGaussians(..., fused_activations=True)")
        self.pc, self.bg = pc, bg
        self.dev = pc.flat.device
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        # RCCL ("nccl") collectives can be captured into a HIP graph: the whole step — render, backward, all-reduce of
        # the flat gradient buffer, Adam — is then ONE replay, with no host work between the backward and the update.
        # (gloo cannot be captured: the CPU tests keep the eager exchange.)
        self.exchange = torch.distributed.is_initialized() and (self.world > 1 or dp.group_of_one())
        self.exchange_in_graph = self.exchange and torch.distributed.get_backend() == "nccl"
        if use_graph and self.exchange and not self.exchange_in_graph:
            import warnings
            warnings.warn(f"TrainStep: the {torch.distributed.get_backend()} exchange cannot be captured; the all-reduce and Adam "
                          "run eagerly behind the captured frame")
        lr = dict(DEFAULT_LRS, **(lrs or {}))
        P, M = pc.P, pc.M
        self.adam = FusedAdam(pc.flat, pc.flat_grad, [
            (P * 3, lr["xyz"]),
            (P * M * 3, lr["feature_dc"], M * 3, 3, lr["feature_rest"]),
            (P, lr["opacity"]), (P * 3, lr["scaling"]), (P * 4, lr["rotation"])], grad_scale=1.0 / self.world)
        # a replayed frame that overflowed its binning capacity back-propagates zeros: the rasterizer's backward says so in
        # the word behind the gradient buffer (summed over ranks by the same all-reduce) and the update skips that step
        self.adam.set_skip_words([pc.overflow_word])
        # _add_densification_stats accumulators (model/fateavatar.py:186-188,734-737), updated by the backward kernel
        self.xyz_gradient_accum = torch.zeros((P, 1), device=self.dev)
        self.denom = torch.zeros((P, 1), device=self.dev)
        pc.fused_densification_stats = (self.xyz_gradient_accum, self.denom, pc.overflow_word)
        # static inputs of the captured step
        self.cam = camera
        self.gt = torch.zeros((3, camera.image_height, camera.image_width), device=self.dev)
        self.loss = torch.zeros((), device=self.dev)
        self._dimage = torch.zeros_like(self.gt)   # dL/dimage of the step
        self._l1_ws = l1_workspace(self.dev)       # scratch of this step's loss kernel (not shared with launches that may overlap)
        self.out = None
        self.use_graph = bool(use_graph)
        self._graph = None       # render .. backward (.. Adam when world == 1)
        self._eager_steps = 0
        self.overflows = 0       # replayed frames that overflowed the captured binning capacity (see _poll_overflow)
        self.host_steps = 0      # step() calls; the device's own count of APPLIED updates is adam.step_count (skipped_steps)

    # -- the step body: everything between zero_grad and the gradient exchange
    def _forward_backward(self):
        self.pc.begin_step()                                   # zero_grad(set_to_none=True), iteration.py:48-49
        out = render(self.cam, self.pc, self.bg)               # activations + rasterizer (fused)
        # nn.L1Loss(reduction='mean') (loss.py:92) + loss.backward(): the loss and the gradient autograd would hand to the
        # rasterizer in one launch, written straight into the step's buffers; then the rasterizer backward (stats fused)
        _, g = l1_loss_and_grad(out["render"], self.gt, loss_out=self.loss, grad_out=self._dimage, workspace=self._l1_ws)
        out["render"].backward(g)
        # keep the step's outputs WITHOUT their autograd graph: a graph kept alive across steps keeps its
        # AccumulateGrad nodes (and the stream they were created on) alive, which breaks a later stream capture
        self.out = {"render": out["render"].detach(), "radii": out["radii"], "visibility_filter": out["visibility_filter"]}

    def _exchange_and_update(self):
        dp.allreduce_sum_(self.pc.exchange_buffer())  # gradients + overflow word; Adam applies grad_scale = 1 / world
        self.adam.step()

    def _body(self):
        self._forward_backward()
        if not self.exchange:
            self.adam.step()
        elif self.exchange_in_graph:
            self._exchange_and_update()

    def _capture(self):
        from . import rasterizer
        # nothing in the captured frame may wait on the host: FR_FLAG_NO_WAIT, scoped to the capture (a replay does not
        # go through rasterize_gaussians at all), so that other renders of the process keep their overflow handling
        with rasterizer.no_wait():
            # one frame on a side stream warms the allocator pools of the capture path; it is not a step (no Adam), so
            # the statistics it accumulated are put back
            acc, den = self.xyz_gradient_accum.clone(), self.denom.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._forward_backward()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.xyz_gradient_accum.copy_(acc)
            self.denom.copy_(den)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):  # thread_local: an RCCL watchdog thread may exist
                self._body()
            torch.cuda.synchronize()
        self._graph = g

    def step(self, camera: TorchCamera, gt_image: torch.Tensor) -> torch.Tensor:
        """One optimisation step on this rank's frame.  Returns the (device) loss scalar of the step."""
        self.host_steps += 1
        self._load_inputs(camera, gt_image)
        if self.use_graph and self._graph is None and self._eager_steps >= 2:
            self._capture()
        if self._graph is not None:
            self._poll_overflow()     # may drop the graph
        if self._graph is not None:
            self._graph.replay()
        else:
            self._body()              # eager: first steps size the binning capacity (high-water mark)
            self._eager_steps += 1
        if self.exchange and not self.exchange_in_graph:
            self._exchange_and_update()
        return self.loss

    def _load_inputs(self, camera: TorchCamera, gt_image: torch.Tensor, extra=()) -> None:
        """The frame's inputs go into the buffers the step was captured with — camera block, target image and whatever
        a subclass adds — in one launch when they are device tensors already."""
        pairs = list(extra)
        if camera is not self.cam:
            self.cam.check_same_intrinsics(camera)
            pairs.append((self.cam._packed, camera._packed))
        pairs.append((self.gt, gt_image))
        if all(s.is_cuda and s.dtype == torch.float32 and s.is_contiguous() and s.shape == d.shape for d, s in pairs):
            multi_copy(pairs)
        else:
            for d, s in pairs:
                d.copy_(s, non_blocking=True)

    @property
    def skipped_steps(self) -> int:
        """Steps whose update the device skipped because a captured frame overflowed its binning capacity (the overflow word,
        FusedAdam.set_skip_words): step() calls minus applied Adam updates.  The host-side schedules (learning-rate decay,
        densify / prune / reset intervals) advance with step() calls; a caller that wants the lost iterations back re-runs this
        many.  Reads device state: synchronises."""
        return self.host_steps - self.adam.step_count

    def _poll_overflow(self):
        """The sort kernel of every frame writes its counts to pinned host memory; reading them costs nothing and
        needs no synchronisation (they belong to the most recent frame that has got that far).  A replayed frame that
        overflowed the capacity the graph was captured with produced no image and no gradients — and its Adam launch did
        nothing (the overflow word the backward sets, FusedAdam.set_skip_words): raise the capacity, drop the graph (the
        next steps run eagerly with the overflow check, then re-capture) and count the event."""
        from . import rasterizer
        if rasterizer.check_async_overflow(self.dev.index or 0):
            self.overflows += 1
            self._graph, self._eager_steps = None, 0
            # every occurrence warns.  The replays since the overflow SKIPPED their update on the device: the Adam step count
            # (`self.adam.step_count`, device state) stays behind the host's iteration counters by the number of skipped steps —
            # the `skipped_steps` property is that difference, for callers whose schedules should re-run the lost iterations
            import warnings
            warnings.warn(f"{type(self).__name__}: the binning capacity overflowed inside the captured step (occurrence "
                          f"{self.overflows}); the affected replays back-propagated zeros and their optimizer launch skipped the "
                          "step (overflow word; `skipped_steps` counts them), the step runs eagerly and is captured again")

    # -- Gaussian maintenance (reference: train/iteration.py:62-86 -> model/fateavatar.py:610-731), generic-3DGS flavour:
    #    the FateAvatar versions additionally carry the mesh binding (face index, barycentrics) of every row
    @torch.no_grad()
    def _after_resize(self, old_index, old_rows, zero_fields=()):
        pc = self.pc
        self.adam.remap_rows(pc.flat, pc.flat_grad, old_index, pc.widths(), old_rows, zero_fields)
        # statistics restart from zero after a change of the point set (model/fateavatar.py:667-672)
        self.xyz_gradient_accum = torch.zeros((pc.P, 1), device=self.dev)
        self.denom = torch.zeros((pc.P, 1), device=self.dev)
        pc.fused_densification_stats = (self.xyz_gradient_accum, self.denom, pc.overflow_word)
        self.adam.set_skip_words([pc.overflow_word])
        self._graph, self._eager_steps = None, 0   # buffers moved: the captured step is stale

    @torch.no_grad()
    def prune_low_opacity(self, min_opacity: float = 0.005) -> int:
        """_prune_low_opacity_points (model/fateavatar.py:674-711).  Returns the number of Gaussians removed."""
        pc = self.pc
        keep = ~(torch.sigmoid(pc._opacity) < min_opacity).reshape(-1)
        old_rows = pc.P
        old_index = pc.resize(keep_mask=keep)
        self._after_resize(old_index, old_rows)
        return old_rows - pc.P

    @torch.no_grad()
    def densify_by_gradient(self, increase_num: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """The sampling and cloning rule of _uv_densify (model/fateavatar.py:610-672) without the mesh re-binding:
        `increase_num` rows drawn with probability proportional to xyz_gradient_accum (multinomial, with
        replacement), cloned with their scale multiplied by 0.75; appended rows start with zero Adam moments.
        Data-parallel: the statistics are per-view sums (model/fateavatar.py:734-737), so the draw is made on rank 0 from
        the sum over all ranks and broadcast — every replica appends the same rows.  Returns the sampled row indices."""
        pc = self.pc
        acc, _ = self.reduce_densification_stats()
        w = acc.reshape(-1)
        idx = torch.zeros(increase_num, dtype=torch.int64, device=self.dev)
        if float(w.sum()) <= 0:       # (the summed statistics are identical on every rank: all of them raise, none is left
            raise RuntimeError("no densification statistics accumulated yet")   # waiting in the broadcast below)
        if not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0:
            idx = torch.multinomial(w, increase_num, replacement=True, generator=generator)   # (`generator`: rank 0's only)
        dp.broadcast_(idx)
        rows = [getattr(pc, name).detach()[idx].clone() for name, _ in pc.FIELDS]
        rows[3] = torch.log(torch.exp(rows[3]) * 0.75)   # _scaling
        old_rows = pc.P
        old_index = pc.resize(new_rows=rows)
        self._after_resize(old_index, old_rows)
        return idx

    @torch.no_grad()
    def reset_opacity(self) -> None:
        """_reset_opacity (model/fateavatar.py:713-731): opacity <- min(opacity, 0.01), its Adam moments <- 0."""
        pc = self.pc
        cur = torch.sigmoid(pc._opacity)
        new = torch.minimum(cur, torch.full_like(cur, 0.01))
        pc._opacity.data.copy_(torch.log(new / (1 - new)))
        # in place: the captured graph keeps pointing at the same parameter and moment buffers
        self.adam.zero_field_moments(pc.widths(), pc.P, fields=(2,))

    # -- checkpoint / resume (reference layout: Trainer.save_checkpoint, train/trainer.py:396-435 — a dict with
    #    'global_step' and 'model' = the Gaussian parameters under the GaussianModel names and shapes; the reference
    #    does not save optimizer state, `optimizer` and `densification` here are additions a resume needs)
    @torch.no_grad()
    def state_dict(self) -> dict:
        pc = self.pc
        f = pc._features.detach()
        model = {"_xyz": pc._xyz.detach().clone(), "_features_dc": f[:, :1, :].clone(), "_features_rest": f[:, 1:, :].clone(),
                 "_opacity": pc._opacity.detach().clone(), "_scaling": pc._scaling.detach().clone(),
                 "_rotation": pc._rotation.detach().clone()}
        return {"global_step": self.adam.step_count, "model": model,
                "optimizer": {"exp_avg": self.adam.exp_avg.clone(), "exp_avg_sq": self.adam.exp_avg_sq.clone(),
                              "state": self.adam.state[:4].clone()},
                "densification": {"xyz_gradient_accum": self.xyz_gradient_accum.clone(), "denom": self.denom.clone()}}

    @torch.no_grad()
    def load_state_dict(self, sd: dict) -> None:
        pc, m = self.pc, sd["model"]
        rows = [m["_xyz"], torch.cat([m["_features_dc"], m["_features_rest"]], dim=1), m["_opacity"], m["_scaling"],
                m["_rotation"]]
        if int(rows[1].shape[1]) != pc.M:
            raise ValueError("checkpoint holds a different number of SH coefficients")
        old_rows = pc.P
        pc._bind([r.to(self.dev, torch.float32) for r in rows])
        # rebind the optimizer to the new buffers (row map irrelevant: every moment is overwritten below)
        self._after_resize(torch.full((pc.P,), -1, dtype=torch.int64, device=self.dev), old_rows)
        opt = sd.get("optimizer")
        if opt is not None:
            self.adam.exp_avg.copy_(opt["exp_avg"])
            self.adam.exp_avg_sq.copy_(opt["exp_avg_sq"])
            self.adam.state.zero_()
            self.adam.state[:4].copy_(opt["state"][:4])
        self.host_steps = self.adam.step_count          # (skipped_steps counts from the restored state on)
        dens = sd.get("densification")
        if dens is not None:
            self.xyz_gradient_accum.copy_(dens["xyz_gradient_accum"])
            self.denom.copy_(dens["denom"])

    def check(self) -> None:
        """After synchronising: raise if a captured (no-wait) frame overflowed its binning capacity."""
        from . import rasterizer
        if self._graph is not None and rasterizer.check_async_overflow(self.dev.index or 0):
            raise RuntimeError("binning capacity overflowed inside the captured step; re-create the TrainStep")

    def reduce_densification_stats(self):
        """(xyz_gradient_accum, denom) summed over all ranks (they are sums over the frames each rank has seen)."""
        acc, den = self.xyz_gradient_accum.clone(), self.denom.clone()
        if self.world > 1:
            dp.allreduce_sum_(acc)
            dp.allreduce_sum_(den)
        return acc, den
