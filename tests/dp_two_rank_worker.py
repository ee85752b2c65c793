"""Worker of tests/test_gpu_dp_two_ranks.py: ONE rank of a world_size-2 data-parallel run of the REAL optimisation step.

Launched by `python -m torch.distributed.run --nproc-per-node 2 ... tests/dp_two_rank_worker.py --mode avatar|train --out DIR`
with FR_DIST_BACKEND=gloo, both ranks on cuda:0 (a 1-GPU box): every rank builds the same Gaussians, renders its OWN frame
of every step, exchanges the flat gradient (dp.allreduce_sum_, Adam applies 1 / world), and goes through the reference's
maintenance schedule — `_uv_densify` (rank-0 draws from the SUMMED statistics, broadcast), prune, opacity reset
(model/fateavatar.py:251-276,610-737; train/iteration.py:62-86; train/loss.py:92-105).  Each rank writes what the test
compares to DIR/rank{r}.pt.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from fateavatar_amd import dp  # noqa: E402


def avatar_setup(dev, P, res, n_frames, seed=0):
    """The synthetic FateAvatar set-up of tests/test_gpu_avatar.py (INSTA-layout sequence, mesh-bound Gaussians, targets
    rendered from a hidden ground-truth avatar) — deterministic, so identical in every process that builds it."""
    from fateavatar_amd import insta, mesh_sampling, scenes
    from fateavatar_amd.avatar import AvatarGaussians, AvatarStep, _BoundFrame
    from fateavatar_amd.binding import bind_gaussians
    from fateavatar_amd.knn import init_scale_by_knn
    from fateavatar_amd.model import TorchCamera
    from fateavatar_amd.render import render
    transform, posed, faces = insta.synthetic_sequence(n_frames, res, seed)
    verts, _, _ = scenes.head_geometry()
    fi, bc = mesh_sampling.random_sampling_barycoords(P, verts, faces, np.random.default_rng(seed))
    pts = (verts[faces[fi]] * bc[:, :, None]).sum(1).astype(np.float32)
    _, _, scale_init = init_scale_by_knn(torch.from_numpy(pts).to(dev))
    cams = [TorchCamera(c, dev) for c in insta.camera_arrays(transform)]
    make = lambda: AvatarGaussians(fi, bc, float(scale_init), dev)  # noqa: E731
    S = dict(posed=torch.from_numpy(posed).to(dev), faces=torch.from_numpy(faces).to(dev), canon=torch.from_numpy(verts).to(dev),
             cams=cams, make=make)
    bg = torch.ones(3, device=dev)
    gt = make()
    g = torch.Generator(device="cpu").manual_seed(5)
    with torch.no_grad():
        gt._features_dc.copy_((torch.rand(gt.P, 1, 3, generator=g) * 2.0 - 1.0).to(dev))
        gt._opacity.fill_(float(np.log(0.6 / 0.4)))
        gt._offset.copy_((0.3 * torch.randn(gt.P, 1, generator=g)).to(dev))
    # (the targets are rendered BEFORE torch.distributed exists in the callers below: this step must not exchange)
    st = AvatarStep(gt, S["faces"], S["canon"], cams[0].clone(), bg, use_graph=False)
    imgs = []
    for f, cam in enumerate(cams):
        st.cam.copy_from(cam)
        st.verts.copy_(S["posed"][f])
        with torch.no_grad():
            xyz, rot, scl = bind_gaussians(st.verts, st.faces, gt.face_index, gt.bary_coords, st.face_scale_canonical, gt._offset,
                                           gt._rotation, gt._scaling, st.shell_len, True)
            imgs.append(render(st.cam, _BoundFrame(xyz, gt, rot, scl, None), bg)["render"].clone())
    S["gts"], S["bg"] = imgs, bg
    return S


def train_setup(dev, P, res, n_views, seed=0):
    """The generic-3DGS step's set-up (TrainStep: xyz optimised directly): config-2-style head scenes from `n_views` cameras,
    targets from a hidden appearance."""
    from fateavatar_amd import scenes
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    sc = [scenes.head_scene(P=P, res=res, sh_degree=1, seed=seed, view=v, n_views=n_views) for v in range(n_views)]
    s0 = sc[0]
    cams = [TorchCamera(s.camera, dev) for s in sc]
    bg = torch.from_numpy(s0.bg).to(dev)
    make = lambda: FlatGaussians(s0.means3D, s0.shs, s0.opacities, s0.scales, s0.rotations, s0.sh_degree, dev,  # noqa: E731
                                 fused_activations=True)
    gt = make()
    g = torch.Generator(device="cpu").manual_seed(7)
    with torch.no_grad():
        gt._features.copy_((0.6 * (torch.rand(tuple(gt._features.shape), generator=g) - 0.5)).to(dev))
        gt._opacity.fill_(float(np.log(0.5 / 0.5)))
        imgs = [render(c, gt, bg)["render"].clone() for c in cams]
    return dict(cams=cams, make=make, gts=imgs, bg=bg)


MAINTAIN = dict(densify_interval=10, prune_interval=15, opacity_reset_interval=25, min_opacity=0.005, increase_num=400,
                max_points_num=10 ** 6)


def local_gradient(st, load):
    """This rank's OWN gradient of the frame `load` puts into the step's buffers, at the current parameters: the step's
    forward + backward without the exchange and without Adam; the densification statistics it added are taken back."""
    acc, den = st.xyz_gradient_accum.clone(), st.denom.clone()
    load()
    st._forward_backward()
    g = st.pc.collect_grads().clone()
    st.xyz_gradient_accum.copy_(acc)
    st.denom.copy_(den)
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("avatar", "train"), required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=33)
    ap.add_argument("--P", type=int, default=20000)
    ap.add_argument("--res", type=int, default=192)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    torch.cuda.set_device(dev)
    n_frames = 8
    # everything that must NOT exchange (the targets' renders) happens before the process group exists
    S = avatar_setup(dev, a.P, a.res, n_frames) if a.mode == "avatar" else train_setup(dev, a.P, a.res, n_frames)
    r, w, _ = dp.init_from_env()
    assert (r, w) == (rank, world)
    pc = S["make"]()
    if a.mode == "avatar":
        from fateavatar_amd.avatar import AvatarStep
        st = AvatarStep(pc, S["faces"], S["canon"], S["cams"][0].clone(), S["bg"], use_graph=True)
        frame_of = lambda it: (world * it + rank) % n_frames  # noqa: E731
        step = lambda f: st.step(S["cams"][f], S["posed"][f], S["gts"][f])  # noqa: E731
        load = lambda f: (st.cam.copy_from(S["cams"][f]), st.verts.copy_(S["posed"][f]), st.gt.copy_(S["gts"][f]))  # noqa: E731
    else:
        from fateavatar_amd.train import TrainStep
        st = TrainStep(pc, S["cams"][0].clone(), S["bg"], use_graph=True)
        frame_of = lambda it: (world * it + rank) % n_frames  # noqa: E731
        step = lambda f: st.step(S["cams"][f], S["gts"][f])  # noqa: E731
        load = lambda f: (st.cam.copy_from(S["cams"][f]), st.gt.copy_(S["gts"][f]))  # noqa: E731
    assert st.world == world and (st.exchange or world == 1)
    out = {"P0": pc.P, "world": world, "grad_scale": float(st.adam.cfg.grad_scale)}
    # ---- step 1 by hand, in step()'s own order, keeping what the exchange sees: own gradient -> all-reduce(SUM) -> Adam
    f = frame_of(0)
    g_local = local_gradient(st, lambda: load(f))
    out["g_local"] = g_local.cpu()
    out["params0"] = pc.flat.detach().clone().cpu()
    load(f)
    st._forward_backward()
    g_step = st.pc.collect_grads().clone()           # this evaluation's own gradient (atomic order: not g_local's bits)
    if world > 1:
        parts = [torch.empty_like(g_step) for _ in range(world)]
        torch.distributed.all_gather(parts, g_step)
        st._exchange_and_update()
        out["g_exchanged"] = pc.flat_grad.detach().clone().cpu()
        out["g_parts_sum"] = sum(parts[1:], parts[0].clone()).cpu()
    else:
        st.adam.step()
    st._eager_steps += 1
    did_log = []
    # ---- the rest through step() + the reference's maintenance schedule (shortened intervals)
    losses = []
    for it in range(1, a.steps):
        f = frame_of(it)
        losses.append(step(f).clone())
        gs = it + 1
        # the prune threshold: the 10 % quantile of the current opacities (a fixed one prunes nothing or everything, depending
        # on where 15 steps of this synthetic problem have taken them) — a function of the parameters, so the same on every
        # rank exactly when the replicas are identical
        thr = float(torch.sigmoid(pc._opacity.detach()).reshape(-1).quantile(0.1)) if gs % MAINTAIN["prune_interval"] == 0 else 0.0
        if a.mode == "avatar":
            did = st.maintain(gs, dict(MAINTAIN, min_opacity=thr))
        else:
            did = {}
            if gs % MAINTAIN["densify_interval"] == 0:
                did["densified"] = int(st.densify_by_gradient(MAINTAIN["increase_num"]).numel())
            if gs % MAINTAIN["prune_interval"] == 0:
                did["pruned"] = st.prune_low_opacity(thr)
            if gs % MAINTAIN["opacity_reset_interval"] == 0:
                st.reset_opacity()
                did["opacity_reset"] = True
        if did:
            did_log.append((gs, {k: int(v) for k, v in did.items()}))
    torch.cuda.synchronize()
    st.check()
    acc, den = st.reduce_densification_stats()
    out.update(
        did=did_log, P_end=pc.P, losses=[float(x) for x in losses], overflows=int(st.overflows), graph=st._graph is not None,
        flat=pc.flat.detach().cpu(), exp_avg=st.adam.exp_avg.cpu(), exp_avg_sq=st.adam.exp_avg_sq.cpu(),
        adam_state=st.adam.state[:4].cpu(), acc_sum=acc.cpu(), den_sum=den.cpu(),
        acc_local=st.xyz_gradient_accum.cpu(), den_local=st.denom.cpu())
    if a.mode == "avatar":
        out.update(face_index=pc.face_index.cpu(), bary=pc.bary_coords.cpu())
    os.makedirs(a.out, exist_ok=True)
    torch.save(out, os.path.join(a.out, f"rank{rank}.pt"))
    if torch.distributed.is_initialized():
        dp.barrier()
        torch.distributed.destroy_process_group()
    print(f"rank {rank} done: P {out['P0']} -> {out['P_end']}, did {did_log}", flush=True)


if __name__ == "__main__":
    main()
