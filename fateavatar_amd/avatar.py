"""FateAvatar's own per-frame optimisation loop around the path (BASELINE.json configs[2]).

reference:
  * parameters and their Adam groups — `_opacity, _offset, _features_dc, _rotation, _scaling` in that order
    (train/optim.py:15-21), learning rates of config/fateavatar.yaml:34-39, Adam(eps 1e-8, default betas)
  * a frame — model/fateavatar.py:225-276: the Gaussians are BOUND to the posed mesh (`bind_gaussians`: position =
    barycentric point + normal * shell_len * tanh(offset); rotation = face quaternion (x) rotation; scaling += log of
    the face's scale ratio), rendered with SH degree 0, one L1 term here (train/loss.py:92)
  * the step — train/iteration.py:21-60: zero_grad(set_to_none) -> render -> loss -> backward -> densification
    statistics -> Adam
  * maintenance — train/iteration.py:62-86 with model/fateavatar.py:610-731: `_uv_densify` (multinomial draw by
    accumulated screen-space gradient, NEW barycentrics on the sampled rows' faces, scale * 0.75, zero Adam moments,
    statistics reset), `_prune_low_opacity_points` (statistics kept for the surviving rows), `_reset_opacity`.
What is fused: binding forward / backward are one kernel each, activations and statistics run inside the rasterizer
kernels, Adam is one kernel over the flat buffer, and the whole step replays as ONE HIP graph.  Data-parallel: one frame
per rank, one flat-gradient all-reduce; the random draws of a densification are made on rank 0 from the SUMMED
statistics and broadcast, so every rank appends identical rows.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import dp
from .binding import bind_gaussians, face_scale
from .bound import MeshBinding, render_bound_batch
from .model import TorchCamera
from .optim import FusedAdam
from .rasterizer import GradOut
from .render import render
from .loss import l1_loss_and_grad, l1_workspace
from .train import TrainStep

# config/fateavatar.yaml:34-39 (group names of train/optim.py:15-21)
FATE_LRS = dict(opacity=0.05, offset=0.0016, color=0.0025, rotation=0.001, scaling=0.005)
# config/fateavatar.yaml:41-47
FATE_MAINTAIN = dict(opacity_reset_interval=60000, densify_interval=3000, prune_interval=2000, min_opacity=0.005,
                     increase_num=1000, max_points_num=200000)


class AvatarGaussians(torch.nn.Module):
    """The mesh-bound Gaussian parameters of FateAvatar (model/fateavatar.py:166-190) in ONE flat buffer, in the order
    of the optimizer groups.  `face_index` / `bary_coords` are the binding (model/fateavatar.py:120-164)."""
    FIELDS = (("_opacity", 1), ("_offset", 1), ("_features_dc", 3), ("_rotation", 4), ("_scaling", 3))
    SHAPES = {"_opacity": (1,), "_offset": (1,), "_features_dc": (1, 3), "_rotation": (4,), "_scaling": (3,)}
    max_sh_degree = 0   # model/fateavatar.py:54,244
    fused_activations = True

    def __init__(self, face_index, bary_coords, scale_init: float, device):
        """_register_init_gaussian (model/fateavatar.py:166-190): grey colour (DC = inverse_sigmoid(0.5) = 0), log
        scale_init on all axes, identity rotation, opacity 0.1, zero offset."""
        super().__init__()
        P = int(len(face_index))
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=device)  # noqa: E731
        rot = z(P, 4)
        rot[:, 0] = 1
        op = torch.full((P, 1), float(np.log(0.1 / 0.9)), dtype=torch.float32, device=device)
        self.face_index = torch.as_tensor(np.asarray(face_index), dtype=torch.int32, device=device).contiguous()
        self.bary_coords = torch.as_tensor(np.asarray(bary_coords), dtype=torch.float32, device=device).contiguous()
        self._bind([op, z(P, 1), z(P, 1, 3), rot, torch.full((P, 3), float(scale_init), dtype=torch.float32, device=device)])

    @classmethod
    def from_template(cls, device, uv_resolution: int = 256, num_points: Optional[int] = None, sampling: str = "uv",
                      rng: Optional[np.random.Generator] = None) -> "AvatarGaussians":
        """The reference's own initialisation on the head template: `_register_template_mesh` + `_register_init_gaussian`
        (model/fateavatar.py:120-190, 597-608).  `sampling="uv"` (the reference): binding points = the texel centres the
        template's UV layout covers at `uv_resolution` x `uv_resolution` (config/fateavatar.yaml:28 `tex_size: 256`), in
        row-major texel order, padded to uv_resolution^2 rows with random points on sampled faces
        (volume_rendering/mesh_sampling.py:86-138); `num_points` (default uv_resolution^2) asks for another row count the way
        the reference's `num_points` argument does (raster side int(sqrt(num_points))).  `sampling="random"`: the
        area-weighted draw the reference keeps commented out (model/fateavatar.py:135-137, mesh_sampling.py:140-169).
        scale_init = the reference's knn estimate on the canonical template points."""
        from . import mesh_sampling, scenes
        from .knn import init_scale_by_knn
        n = int(num_points) if num_points is not None else int(uv_resolution) * int(uv_resolution)
        rng = rng or np.random.default_rng(0)
        verts, faces, _ = scenes.head_geometry()
        if sampling == "uv":
            uv = scenes.head_uv()
            if uv is None:
                raise RuntimeError("the head template's UV layout is not in fateavatar_amd/data/head_template_geom.npz")
            fi, bc = mesh_sampling.uniform_sampling_barycoords(n, uv[0], uv[1], rng=rng)
        elif sampling == "random":
            fi, bc = mesh_sampling.random_sampling_barycoords(n, verts, faces, rng)
        else:
            raise ValueError(f"sampling must be 'uv' or 'random', not {sampling!r}")
        pts = (verts[faces[fi]] * bc[:, :, None]).sum(1).astype(np.float32)      # reweight_verts_by_barycoords
        scale_init = float(init_scale_by_knn(torch.from_numpy(pts).to(device))[2])
        return cls(fi, bc, scale_init, device)

    @property
    def P(self):
        return int(self.face_index.shape[0])

    def widths(self):
        return [w for _, w in self.FIELDS]

    def _bind(self, raw):
        P = raw[0].shape[0]
        dev = raw[0].device
        sizes = [P * w for _, w in self.FIELDS]
        self.flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        # (gradient buffer + the step's overflow word behind it: model.FlatGaussians._bind)
        self._grad_store = torch.zeros(sum(sizes) + 4, dtype=torch.float32, device=dev)
        self.flat_grad = self._grad_store[:sum(sizes)]
        self.overflow_word = self._grad_store[sum(sizes):sum(sizes) + 1]
        off = 0
        for (name, w), n, r in zip(self.FIELDS, sizes, raw):
            shp = (P,) + self.SHAPES[name]
            self.flat[off:off + n].copy_(r.detach().reshape(-1))
            p = torch.nn.Parameter(self.flat[off:off + n].view(shp))
            # every parameter reaches a HIP kernel raw (the rasterizer or the binding op): its gradient is written straight
            # into the flat gradient buffer
            p._fr_grad_out = GradOut(self.flat_grad[off:off + n].view(shp))
            setattr(self, name, p)
            off += n

    def lane(self) -> "AvatarGaussians":
        """A second set of leaves over the SAME parameter storage with a gradient buffer of its own: what another view of
        the batch, rendered in flight together with this one, back-propagates into (AvatarBatchStep)."""
        o = AvatarGaussians.__new__(AvatarGaussians)
        torch.nn.Module.__init__(o)
        o.face_index, o.bary_coords, o.flat = self.face_index, self.bary_coords, self.flat
        o._grad_store = torch.zeros_like(self._grad_store)
        o.flat_grad = o._grad_store[:self.flat.numel()]
        o.overflow_word = o._grad_store[self.flat.numel():self.flat.numel() + 1]
        off = 0
        for name, w in self.FIELDS:
            n = self.P * w
            shp = (self.P,) + self.SHAPES[name]
            p = torch.nn.Parameter(self.flat[off:off + n].view(shp))
            p._fr_grad_out = GradOut(o.flat_grad[off:off + n].view(shp))
            setattr(o, name, p)
            off += n
        return o

    def begin_step(self):
        for name, _ in self.FIELDS:
            getattr(self, name).grad = None

    def exchange_buffer(self) -> torch.Tensor:
        """`collect_grads()` + the overflow word behind it: what a data-parallel step all-reduces (SUM)."""
        self.collect_grads()
        return self._grad_store

    def collect_grads(self) -> torch.Tensor:
        off = 0
        for name, w in self.FIELDS:
            n = self.P * w
            g, view = getattr(self, name).grad, self.flat_grad[off:off + n]
            if g is None:
                view.zero_()
            elif g.data_ptr() != view.data_ptr():
                view.copy_(g.reshape(-1))
            off += n
        return self.flat_grad

    @torch.no_grad()
    def resize(self, keep_mask=None, new_rows=None, new_face_index=None, new_bary=None, order=None):
        """Drop the rows where keep_mask is False — or take the rows in the sequence `order` (row indices: a permutation
        re-stores the set in another order) —, then append `new_rows` (one tensor per field) with their binding.
        Returns the row map (old row of every new row, -1 for appended ones) the optimizer state has to follow."""
        dev = self.flat.device
        if order is not None:
            if keep_mask is not None:
                raise ValueError("resize: keep_mask or order, not both")
            old_index = order.to(dev, torch.int64).reshape(-1)
        else:
            keep = torch.ones(self.P, dtype=torch.bool, device=dev) if keep_mask is None else keep_mask.reshape(-1).bool()
            old_index = torch.nonzero(keep).reshape(-1)
        raw = [getattr(self, name).detach()[old_index] for name, _ in self.FIELDS]
        fi, bc = self.face_index[old_index], self.bary_coords[old_index]
        n_new = 0
        if new_rows is not None:
            n_new = int(new_rows[0].shape[0])
            raw = [torch.cat([r, a.to(dev, torch.float32).reshape((n_new,) + tuple(r.shape[1:]))]) for r, a in zip(raw, new_rows)]
            fi = torch.cat([fi, new_face_index.to(dev, torch.int32)])
            bc = torch.cat([bc, new_bary.to(dev, torch.float32)])
        self.face_index, self.bary_coords = fi.contiguous(), bc.contiguous()
        self._bind(raw)
        return torch.cat([old_index, torch.full((n_new,), -1, dtype=torch.int64, device=dev)])


class _BoundFrame:
    """What render() reads of a Gaussian holder (render_3dgs.py:19-63), for one frame's bound values."""
    max_sh_degree = 0
    fused_activations = True

    def __init__(self, xyz, pc: AvatarGaussians, rot, scl, stats):
        self.get_xyz, self._opacity, self._scaling, self._rotation = xyz, pc._opacity, scl, rot
        self.get_features = pc._features_dc            # cat(features_dc, features_rest) with an empty rest (M = 1)
        self.fused_densification_stats = stats


class _RawFrame:
    """What render_bound_batch() reads: the RAW parameters of a frame whose binding the rasterizer evaluates itself.  (Not a
    holder for render(): it has no bound positions.)"""
    max_sh_degree = 0
    fused_activations = True

    def __init__(self, pc: AvatarGaussians, stats):
        self._opacity, self._offset, self._rotation, self._scaling = pc._opacity, pc._offset, pc._rotation, pc._scaling
        self.get_features = pc._features_dc
        self.fused_densification_stats = stats


class AvatarStep(TrainStep):
    """One optimisation step of FateAvatar per call: `step(camera, posed_verts, gt_image)`."""

    def __init__(self, pc: AvatarGaussians, faces: torch.Tensor, canonical_verts: torch.Tensor, camera: TorchCamera,
                 bg: torch.Tensor, lrs: Optional[dict] = None, shell_len: float = 0.05, resize_scale: bool = True,
                 use_graph: bool = True, fold_binding: bool = True, keep_coherent: bool = False):
        """`keep_coherent`: after every `uv_densify` the rows are re-stored in a spatially coherent order (`sort_coherent`):
        the reference appends the new rows at the end (model/fateavatar.py:640-665), so a set that started in UV-raster
        order (mesh_sampling.py:86-138: neighbours in storage are neighbours on the mesh) grows an unordered tail; the
        results do not depend on the order, the rasterizer's speed does.  Off by default: row i then stays
        row i as in the reference.
        `fold_binding` (default): the binding is evaluated inside the rasterizer's per-Gaussian kernels (bound.py,
        fr_aux::binding) — no binding launches, no bound arrays written by one kernel to be read by the next.  False: the
        stand-alone `bind_gaussians` op in front of `render()` (same results; kept as the A/B and as the op's own user)."""
        self.pc, self.bg = pc, bg
        self.fold_binding = bool(fold_binding)
        self.keep_coherent = bool(keep_coherent)
        self.dev = pc.flat.device
        self.world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        self.exchange = torch.distributed.is_initialized() and (self.world > 1 or dp.group_of_one())
        self.exchange_in_graph = self.exchange and torch.distributed.get_backend() == "nccl"   # (see TrainStep)
        self.lr = dict(FATE_LRS, **(lrs or {}))
        self.faces = faces.to(self.dev, torch.int32).contiguous()
        self.shell_len, self.resize_scale = float(shell_len), bool(resize_scale)
        # compute_face_orientation(canonical verts, return_scale=True)[1] (model/fateavatar.py:84-85)
        self.face_scale_canonical = face_scale(canonical_verts.to(self.dev, torch.float32), self.faces)
        self._make_adam()
        self.xyz_gradient_accum = torch.zeros((pc.P, 1), device=self.dev)
        self.denom = torch.zeros((pc.P, 1), device=self.dev)
        self.cam = camera
        self.canon_verts = canonical_verts.to(self.dev, torch.float32).clone().contiguous()
        self.verts = canonical_verts.to(self.dev, torch.float32).clone().contiguous()   # static input of the captured step
        self.gt = torch.zeros((3, camera.image_height, camera.image_width), device=self.dev)
        self.loss = torch.zeros((), device=self.dev)
        self._dimage = torch.zeros_like(self.gt)   # dL/dimage of the step
        self._l1_ws = l1_workspace(self.dev)       # scratch of this frame's loss kernel (one per lane: loss.py)
        self.out = None
        self.use_graph = bool(use_graph)
        self._graph, self._eager_steps, self.overflows = None, 0, 0
        self.host_steps = 0      # (TrainStep.skipped_steps)

    def adam_segments(self):
        """The optimizer groups (train/optim.py:15-21 with config/fateavatar.yaml:34-39) as runs of the flat buffer."""
        lr, P = self.lr, self.pc.P
        return [(P * 1, lr["opacity"]), (P * 1, lr["offset"]), (P * 3, lr["color"]), (P * 4, lr["rotation"]), (P * 3, lr["scaling"])]

    def _make_adam(self):
        pc = self.pc
        self.adam = FusedAdam(pc.flat, pc.flat_grad, self.adam_segments(), grad_scale=1.0 / self.world)
        self.adam.set_skip_words([pc.overflow_word])

    def _forward_backward(self):
        self._forward_backward_on(self)

    def _forward_backward_on(self, L):
        """Bind, render, L1, backward for one frame.  `L` holds what the frame reads and writes — pc, verts, cam, gt, loss,
        _dimage, xyz_gradient_accum, denom, out: this object itself, or one lane of an AvatarBatchStep."""
        pc = L.pc
        pc.begin_step()                                             # zero_grad(set_to_none=True), iteration.py:48-49
        if self.fold_binding:
            from . import rasterizer
            out = render_bound_batch([L.cam], [_RawFrame(pc, (L.xyz_gradient_accum, L.denom, pc.overflow_word))], [L.verts], self._binding(pc),
                                     self.bg, slots=[rasterizer._slot])[0]
        else:
            xyz, rot, scl = bind_gaussians(L.verts, self.faces, pc.face_index, pc.bary_coords, self.face_scale_canonical,
                                           pc._offset, pc._rotation, pc._scaling, self.shell_len, self.resize_scale)
            frame = _BoundFrame(xyz, pc, rot, scl, (L.xyz_gradient_accum, L.denom, pc.overflow_word))
            out = render(L.cam, frame, self.bg)
        _, g = l1_loss_and_grad(out["render"], L.gt, loss_out=L.loss, grad_out=L._dimage, workspace=L._l1_ws)   # see TrainStep
        out["render"].backward(g)
        L.out = {"render": out["render"].detach(), "radii": out["radii"], "visibility_filter": out["visibility_filter"]}

    def _binding(self, pc: AvatarGaussians) -> MeshBinding:
        return MeshBinding(self.faces, pc.face_index, pc.bary_coords, self.face_scale_canonical, self.shell_len, self.resize_scale)

    def step(self, camera: TorchCamera, posed_verts: torch.Tensor, gt_image: torch.Tensor) -> torch.Tensor:
        self._extra_inputs = [(self.verts, posed_verts)]
        return super().step(camera, gt_image)

    def _load_inputs(self, camera, gt_image, extra=()):
        super()._load_inputs(camera, gt_image, extra=self._extra_inputs)

    # ---- maintenance (train/iteration.py:62-86)
    def maintain(self, global_step: int, cfg: Optional[dict] = None) -> dict:
        """The reference's schedule after the optimizer step of `global_step`.  Returns what was done."""
        c = dict(FATE_MAINTAIN, **(cfg or {}))
        did = {}
        if global_step % c["densify_interval"] == 0 and self.pc.P < c["max_points_num"]:
            did["densified"] = self.uv_densify(min(c["max_points_num"] - self.pc.P, c["increase_num"]))
        if global_step % c["prune_interval"] == 0:
            did["pruned"] = self.prune_low_opacity(c["min_opacity"])
        if global_step % c["opacity_reset_interval"] == 0 and global_step != 0:
            self.reset_opacity()
            did["opacity_reset"] = True
        return did

    @torch.no_grad()
    def _rebind_optimizer(self, old_index, old_rows):
        pc = self.pc
        self.adam.remap_rows(pc.flat, pc.flat_grad, old_index, pc.widths(), old_rows)
        self.adam.set_skip_words([pc.overflow_word])
        self._graph, self._eager_steps = None, 0   # buffers moved: the captured step is stale

    @torch.no_grad()
    def uv_densify(self, increase_num: int, generator: Optional[torch.Generator] = None) -> int:
        """_uv_densify (model/fateavatar.py:610-672).  The two random draws (multinomial over xyz_gradient_accum, with
        replacement; uniform barycentrics) are made on rank 0 from the statistics summed over all ranks and broadcast."""
        pc = self.pc
        acc, _ = self.reduce_densification_stats()
        w = acc.reshape(-1)
        idx = torch.zeros(increase_num, dtype=torch.int64, device=self.dev)
        uvw = torch.zeros((increase_num, 3), dtype=torch.float32, device=self.dev)
        if float(w.sum()) <= 0:       # (the summed statistics are identical on every rank: all of them raise, none is left
            raise RuntimeError("no densification statistics accumulated yet")   # waiting in the broadcast below)
        if not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0:
            idx = torch.multinomial(w, increase_num, replacement=True, generator=generator)   # (`generator`: rank 0's only)
            uvw = torch.rand((increase_num, 3), device=self.dev, generator=generator)
        dp.broadcast_(idx)
        dp.broadcast_(uvw)
        new_bary = uvw / uvw.sum(dim=-1, keepdim=True)
        rows = [getattr(pc, name).detach()[idx].clone() for name, _ in pc.FIELDS]
        rows[4] = torch.log(torch.exp(rows[4]) * 0.75)          # new_scaling (:624)
        old_rows = pc.P
        old_index = pc.resize(new_rows=rows, new_face_index=pc.face_index[idx], new_bary=new_bary)
        self._rebind_optimizer(old_index, old_rows)
        # statistics restart from zero (:667-669)
        self.xyz_gradient_accum = torch.zeros((pc.P, 1), device=self.dev)
        self.denom = torch.zeros((pc.P, 1), device=self.dev)
        self.last_densify = (idx, new_bary)
        if self.keep_coherent:
            AvatarStep.sort_coherent(self)     # (a subclass's lanes are rebuilt by its own uv_densify)
        return increase_num

    @torch.no_grad()
    def coherent_order(self, cells: int = 32) -> torch.Tensor:
        """Row indices in a spatially coherent sequence: grid cells (`cells` per axis, x fastest) of the Gaussians' points
        on the CANONICAL mesh (barycentric points: the posed offsets are small against a cell) — scenes.spatial_order on
        the device, deterministic (stable sort), so identical on every rank of a data-parallel run."""
        pc = self.pc
        tri = self.canon_verts[self.faces.long()[pc.face_index.long()]]                 # [P,3,3]
        pts = (tri * pc.bary_coords.unsqueeze(-1)).sum(1).double()
        lo, hi = pts.min(0).values, pts.max(0).values
        c = ((pts - lo) / (hi - lo).clamp_min(1e-12) * cells).long().clamp_(max=cells - 1)
        return torch.argsort((c[:, 2] * cells + c[:, 1]) * cells + c[:, 0], stable=True)

    @torch.no_grad()
    def sort_coherent(self, cells: int = 32) -> torch.Tensor:
        """Re-store the Gaussians (parameters, binding, Adam moments, densification statistics) in `coherent_order`.
        Returns the order applied (old row of every new row)."""
        pc = self.pc
        order = self.coherent_order(cells)
        old_rows = pc.P
        acc, den = self.xyz_gradient_accum[order], self.denom[order]
        old_index = pc.resize(order=order)
        self._rebind_optimizer(old_index, old_rows)
        self.xyz_gradient_accum, self.denom = acc.contiguous(), den.contiguous()
        return order

    @torch.no_grad()
    def prune_low_opacity(self, min_opacity: float = 0.005) -> int:
        """_prune_low_opacity_points (model/fateavatar.py:674-711): the statistics of the surviving rows are kept."""
        pc = self.pc
        keep = ~(torch.sigmoid(pc._opacity) < min_opacity).reshape(-1)
        old_rows = pc.P
        acc, den = self.xyz_gradient_accum[keep], self.denom[keep]
        old_index = pc.resize(keep_mask=keep)
        self._rebind_optimizer(old_index, old_rows)
        self.xyz_gradient_accum, self.denom = acc.contiguous(), den.contiguous()
        return old_rows - pc.P

    @torch.no_grad()
    def reset_opacity(self) -> None:
        """_reset_opacity (model/fateavatar.py:713-731), in place (the captured graph stays valid)."""
        pc = self.pc
        cur = torch.sigmoid(pc._opacity)
        new = torch.minimum(cur, torch.full_like(cur, 0.01))
        pc._opacity.data.copy_(torch.log(new / (1 - new)))
        self.adam.zero_field_moments(pc.widths(), pc.P, fields=(0,))

    # ---- checkpoints in the reference's layout (Trainer.save_checkpoint, train/trainer.py:396-435: a dict with 'epoch',
    #      'global_step' and 'model' = model.state_dict(), which for FateAvatar holds the six Gaussian parameters and the
    #      two binding buffers next to the FLAME / blendshape entries)
    GAUSSIAN_ATTRIBUTES = ['_offset', '_features_dc', '_features_rest', '_scaling', '_rotation', '_opacity', 'face_index',
                           'bary_coords']   # train/deserialize.py:10-12

    def state_dict(self) -> dict:
        pc = self.pc
        model = {name: getattr(pc, name).detach().clone() for name, _ in pc.FIELDS}
        model["_features_rest"] = torch.zeros((pc.P, 0, 3), device=self.dev)   # max_sh_degree 0: empty (fateavatar.py:172-183)
        model["face_index"], model["bary_coords"] = pc.face_index.clone(), pc.bary_coords.clone()
        # 'optimizer' and 'densification' are additions a seamless resume needs; the reference saves neither
        return {"global_step": self.adam.step_count, "model": model,
                "optimizer": {"exp_avg": self.adam.exp_avg.clone(), "exp_avg_sq": self.adam.exp_avg_sq.clone(),
                              "state": self.adam.state[:4].clone()},
                "densification": {"xyz_gradient_accum": self.xyz_gradient_accum.clone(), "denom": self.denom.clone()}}

    @torch.no_grad()
    def load_state_dict(self, sd: dict) -> list:
        """deserialize_checkpoints_fateavatar (train/deserialize.py:7-40): the Gaussian attributes are POPPED from
        sd['model'] (their row count differs from the freshly built model's), re-attached as parameters / buffers, and the
        per-point statistics restart from zero; whatever else 'model' holds (FLAME, blendshape deltas: outside this
        path) is returned as the list of ignored keys.  A checkpoint written by the reference itself loads the same way."""
        model = dict(sd["model"])
        missing = [k for k in self.GAUSSIAN_ATTRIBUTES if k not in model]
        if missing:
            raise KeyError(f"checkpoint lacks Gaussian attributes {missing}")
        g = {k: model.pop(k) for k in self.GAUSSIAN_ATTRIBUTES}
        if int(g["_features_rest"].shape[1]) != 0:
            raise ValueError("FateAvatar renders SH degree 0: _features_rest must be empty")
        pc = self.pc
        old_rows = pc.P
        pc.face_index = g["face_index"].to(self.dev, torch.int32).contiguous()
        pc.bary_coords = g["bary_coords"].to(self.dev, torch.float32).contiguous()
        P = int(pc.face_index.shape[0])
        pc._bind([g[name].to(self.dev, torch.float32).reshape((P,) + pc.SHAPES[name]) for name, _ in pc.FIELDS])
        self._rebind_optimizer(torch.full((P,), -1, dtype=torch.int64, device=self.dev), old_rows)   # fresh (zero) moments
        self.xyz_gradient_accum = torch.zeros((P, 1), device=self.dev)
        self.denom = torch.zeros((P, 1), device=self.dev)
        opt, dens = sd.get("optimizer"), sd.get("densification")
        if opt is not None:
            self.adam.exp_avg.copy_(opt["exp_avg"])
            self.adam.exp_avg_sq.copy_(opt["exp_avg_sq"])
            self.adam.state.zero_()
            self.adam.state[:4].copy_(opt["state"][:4])
        self.host_steps = self.adam.step_count          # (skipped_steps counts from the restored state on)
        if dens is not None:
            self.xyz_gradient_accum.copy_(dens["xyz_gradient_accum"])
            self.denom.copy_(dens["denom"])
        return sorted(model.keys())


class _Lane:
    """One view of a batch: everything its frame reads and writes next to the shared parameters."""


class AvatarBatchStep(AvatarStep):
    """K frames per optimisation step, IN FLIGHT TOGETHER: `step(cameras, posed_verts, gt_images)` with K of each.

    reference: the model renders the frames of a batch one after the other with shared Gaussians
    (`for bs_ in range(bs)`, model/fateavatar.py:251-276) and the loss is the mean over the batch (train/loss.py:92-105).
    Here every frame of the batch is a LANE — its own leaves over the shared parameter storage with its own gradient
    buffer (AvatarGaussians.lane), static inputs, densification statistics, stream, fr_handle slot and captured graph
    (bind -> render -> L1 -> backward) — and the lanes run concurrently: one frame's kernels leave most of the chip idle
    (DESIGN.md §4).  One fused Adam then steps on the SUM of the lanes' gradients (fr_adam_step_multi; grad_scale
    = 1 / (K x ranks) makes it the batch mean)."""

    def __init__(self, pc: AvatarGaussians, faces, canonical_verts, camera: TorchCamera, bg, views_per_step: int = 3,
                 chain: bool = True, **kw):
        """`chain` (default): the K frames go through ONE launch chain on ONE stream instead of K lanes on K streams — bind x K,
        `render_batch` (fr_forward_batch: every rasterizer kernel launched once for the K views), L1 x K, one batched
        backward (fr_backward_batch), bind backward x K, Adam: the whole step one captured graph, no stream or
        hardware-queue arrangement and no cross-stream events (10.0 k against 8.2 k frames/s at K = 4, 100 k Gaussians,
        512^2).  `chain=False` keeps the lanes: a stream, a handle and a captured graph per frame."""
        from . import _lib
        if not 1 <= int(views_per_step) <= _lib.FR_ADAM_MAX_GRADS:
            raise ValueError(f"views_per_step must be 1..{_lib.FR_ADAM_MAX_GRADS}")
        self.K = int(views_per_step)
        self.chain = bool(chain)
        self._chain_graph = None
        super().__init__(pc, faces, canonical_verts, camera, bg, **kw)
        self._build_lanes()

    def _make_adam(self):
        super()._make_adam()
        self.adam.set_grad_scale(1.0 / (self.world * self.K))

    def _build_lanes(self):
        from .streams import concurrent_streams
        # (streams that land on one hardware queue are serialised: pick lanes' streams that overlap with each other and
        # with the caller's — fateavatar_amd/streams.py)
        # (the launch chain runs on the caller's stream: no lane streams, no probing)
        streams = ([None] * self.K if self.chain else
                   concurrent_streams(self.K, self.dev, also_with=[torch.cuda.current_stream(self.dev)]))
        self.lanes = []
        for k in range(self.K):
            L = _Lane()
            L.k = k
            if k == 0:     # lane 0 IS this object's own frame state
                L.pc, L.cam, L.verts, L.gt, L.loss, L._dimage = self.pc, self.cam, self.verts, self.gt, self.loss, self._dimage
                L.xyz_gradient_accum, L.denom, L._l1_ws = self.xyz_gradient_accum, self.denom, self._l1_ws
            else:
                L.pc, L.cam, L.verts = self.pc.lane(), self.cam.clone(), self.verts.clone()
                L.gt, L.loss, L._dimage = torch.zeros_like(self.gt), torch.zeros_like(self.loss), torch.zeros_like(self._dimage)
                L.xyz_gradient_accum, L.denom = torch.zeros_like(self.xyz_gradient_accum), torch.zeros_like(self.denom)
                L._l1_ws = l1_workspace(self.dev)   # (the lanes' loss kernels overlap: one scratch each, loss.py)
            L.out, L.graph = None, None
            L.stream, L.done = streams[k], torch.cuda.Event()
            self.lanes.append(L)
        self._eager_steps = 0
        self._chain_graph = None           # (captured over the old lanes' buffers)
        self._ready = torch.cuda.Event()
        # one overflowed view skips the whole step (its zero gradient would otherwise be averaged in)
        self.adam.set_skip_words([L.pc.overflow_word for L in self.lanes])

    # ---- the step
    def _capture_lanes(self):
        from . import rasterizer
        for L in self.lanes:
            with rasterizer.handle_slot(L.k), rasterizer.no_wait():
                acc, den = L.xyz_gradient_accum.clone(), L.denom.clone()
                L.stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(L.stream):
                    self._forward_backward_on(L)          # warms the allocator pools of the capture path: not a step
                torch.cuda.synchronize()
                L.xyz_gradient_accum.copy_(acc)
                L.denom.copy_(den)
                L.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(L.graph, stream=L.stream, capture_error_mode="thread_local"):
                    self._forward_backward_on(L)
                torch.cuda.synchronize()

    def _drop_graphs(self):
        for L in self.lanes:
            L.graph = None
        self._chain_graph = None
        self._eager_steps = 0

    # ---- the step as one launch chain
    def _chain_body(self):
        from .render import render_batch
        frames = []
        for L in self.lanes:
            L.pc.begin_step()
            if self.fold_binding:
                frames.append(_RawFrame(L.pc, (L.xyz_gradient_accum, L.denom, L.pc.overflow_word)))
                continue
            xyz, rot, scl = bind_gaussians(L.verts, self.faces, L.pc.face_index, L.pc.bary_coords, self.face_scale_canonical,
                                           L.pc._offset, L.pc._rotation, L.pc._scaling, self.shell_len, self.resize_scale)
            frames.append(_BoundFrame(xyz, L.pc, rot, scl, (L.xyz_gradient_accum, L.denom, L.pc.overflow_word)))
        if self.fold_binding:
            outs = render_bound_batch([L.cam for L in self.lanes], frames, [L.verts for L in self.lanes], self._binding(self.pc),
                                      self.bg, slots=[L.k for L in self.lanes])
        else:
            outs = render_batch([L.cam for L in self.lanes], frames, self.bg, slots=[L.k for L in self.lanes])
        from .loss import l1_loss_and_grad_batch
        images = [out["render"] for out in outs]
        _, grads = l1_loss_and_grad_batch(images, [L.gt for L in self.lanes], [L.loss for L in self.lanes],
                                          [L._dimage for L in self.lanes], [L._l1_ws for L in self.lanes])
        for L, out in zip(self.lanes, outs):
            L.out = {"render": out["render"].detach(), "radii": out["radii"], "visibility_filter": out["visibility_filter"]}
        torch.autograd.backward(images, grad_tensors=grads)
        if self.exchange:                          # sum of the local lanes, then the sum over the ranks; Adam scales
            flat = [L.pc.exchange_buffer() for L in self.lanes]   # (gradients + overflow words: the words add up too)
            for g in flat[1:]:
                flat[0].add_(g)
            if self.exchange_in_graph or not torch.cuda.is_current_stream_capturing():
                dp.allreduce_sum_(flat[0])
            self.adam.step()
        else:
            self.adam.step([L.pc.collect_grads() for L in self.lanes])

    def _capture_chain(self):
        from . import rasterizer
        with rasterizer.no_wait():
            stats = [(L.xyz_gradient_accum.clone(), L.denom.clone()) for L in self.lanes]
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream(self.dev))
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                self._chain_body()
            torch.cuda.synchronize()
            for L, (acc, den) in zip(self.lanes, stats):   # (capturing does not execute: nothing to put back, but keep the
                L.xyz_gradient_accum.copy_(acc)            # statistics exactly as they were)
                L.denom.copy_(den)
        self._chain_graph = g

    def step(self, cameras, posed_verts, gt_images):
        """One optimisation step on K frames.  Returns the K (device) loss scalars of the step."""
        from . import rasterizer
        from .loss import multi_copy
        if not (len(cameras) == len(posed_verts) == len(gt_images) == self.K):
            raise ValueError(f"AvatarBatchStep.step needs {self.K} cameras, vertex sets and images")
        self.host_steps += 1
        main = torch.cuda.current_stream(self.dev)
        pairs = []
        for L, cam, verts, gt in zip(self.lanes, cameras, posed_verts, gt_images):
            pairs += [(L.verts, verts), (L.gt, gt)]
            if cam is not L.cam:
                L.cam.check_same_intrinsics(cam)
                pairs.append((L.cam._packed, cam._packed))
        try:
            multi_copy(pairs)                      # the inputs of all K frames: one launch
        except RuntimeError:                       # (host tensors, other dtypes, more than twelve pairs: plain copies)
            for d, s in pairs:
                d.copy_(s, non_blocking=True)
        if self.chain:
            return self._step_chain()
        captured = self.lanes[0].graph is not None
        if self.use_graph and not captured and self._eager_steps >= 2:
            self._capture_lanes()
            captured = True
        self._steps_since_poll = getattr(self, "_steps_since_poll", 0) + 1
        if captured and self._steps_since_poll >= 8:
            # a replayed frame that overflowed its binning capacity (pinned counts, no synchronisation): back to eager
            self._steps_since_poll = 0
            for L in self.lanes:
                with rasterizer.handle_slot(L.k):
                    if rasterizer.check_async_overflow(self.dev.index or 0):
                        self.overflows += 1
                        self._drop_graphs()
                        captured = False
                        break
        self._ready.record(main)                   # inputs loaded; the previous step's Adam has been enqueued before it
        for L in self.lanes:
            with torch.cuda.stream(L.stream):
                L.stream.wait_event(self._ready)
                if captured:
                    L.graph.replay()
                else:
                    with rasterizer.handle_slot(L.k):
                        self._forward_backward_on(L)
                        L.pc.collect_grads()
                L.done.record(L.stream)
        if not captured:
            self._eager_steps += 1
        for L in self.lanes:
            main.wait_event(L.done)
        grads = [L.pc.flat_grad for L in self.lanes]
        if self.exchange:                          # sum of the local lanes, then the sum over the ranks; Adam scales
            grads = [L.pc._grad_store for L in self.lanes]        # (gradients + overflow words: the words add up too)
            for g in grads[1:]:
                grads[0].add_(g)
            dp.allreduce_sum_(grads[0])
            self.adam.step()
        else:
            self.adam.step(grads)
        self.out = self.lanes[0].out
        return tuple(L.loss for L in self.lanes)

    def _step_chain(self):
        from . import rasterizer
        captured = self._chain_graph is not None
        # (a gloo exchange cannot be captured: such steps stay eager)
        if self.use_graph and not captured and self._eager_steps >= 2 and (not self.exchange or self.exchange_in_graph):
            self._capture_chain()
            captured = True
        # Every step polls the lanes' pinned count slots (a host read, no synchronisation): the counts are those of the
        # most recent replay that has FINISHED, so an overflow inside the captured chain is seen one or two steps late.
        # Those replays back-propagated zeros for the overflowed view — and their Adam launch SKIPPED the step on the device
        # (the view's overflow word, FusedAdam.set_skip_words): nothing moved on a partly zero gradient.  They are counted
        # and named — `overflow_steps` — and the chain is captured again with the raised capacity.
        self._step_no = getattr(self, "_step_no", 0) + 1
        if captured:
            for L in self.lanes:
                with rasterizer.handle_slot(L.k):
                    if rasterizer.check_async_overflow(self.dev.index or 0):
                        self.overflows += 1
                        self.overflow_steps = getattr(self, "overflow_steps", []) + [self._step_no - 1]
                        import warnings   # (every occurrence: each one is a step whose view contributed a zero gradient)
                        warnings.warn(f"AvatarBatchStep: the binning capacity overflowed inside the captured chain around step "
                                      f"{self._step_no - 1} (view {L.k}, occurrence {self.overflows}); that view back-propagated zeros and "
                                      "the optimizer skipped the affected step(s) on every rank; the chain is captured again with a "
                                      "larger capacity")
                        self._drop_graphs()
                        captured = False
                        break
        if captured:
            self._chain_graph.replay()
        else:
            self._chain_body()
            self._eager_steps += 1
        self.out = self.lanes[0].out
        return tuple(L.loss for L in self.lanes)

    def check(self) -> None:
        from . import rasterizer
        for L in self.lanes:
            with rasterizer.handle_slot(L.k):
                if (L.graph is not None or self._chain_graph is not None) and rasterizer.check_async_overflow(self.dev.index or 0):
                    raise RuntimeError("binning capacity overflowed inside a captured frame; re-create the step")

    # ---- maintenance: the lanes' statistics are folded into this object's before anything reads them, and the lanes are
    #      rebuilt whenever the point set (and with it every buffer) changes
    @torch.no_grad()
    def _fold_stats(self):
        for L in self.lanes[1:]:
            self.xyz_gradient_accum.add_(L.xyz_gradient_accum)
            self.denom.add_(L.denom)
            L.xyz_gradient_accum.zero_()
            L.denom.zero_()

    def reduce_densification_stats(self):
        self._fold_stats()
        return super().reduce_densification_stats()

    def uv_densify(self, increase_num: int, generator=None) -> int:
        torch.cuda.synchronize()
        n = super().uv_densify(increase_num, generator)
        self._build_lanes()
        return n

    def prune_low_opacity(self, min_opacity: float = 0.005) -> int:
        torch.cuda.synchronize()
        self._fold_stats()
        n = super().prune_low_opacity(min_opacity)
        self._build_lanes()
        return n

    def reset_opacity(self) -> None:
        torch.cuda.synchronize()
        super().reset_opacity()          # in place: the lanes' leaves see it (shared storage), their graphs stay valid

    def sort_coherent(self, cells: int = 32):
        torch.cuda.synchronize()
        self._fold_stats()
        order = super().sort_coherent(cells)
        self._build_lanes()
        return order

    def load_state_dict(self, sd: dict) -> list:
        torch.cuda.synchronize()
        ignored = super().load_state_dict(sd)
        self._build_lanes()
        return ignored
