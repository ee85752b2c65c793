#!/usr/bin/env python
"""the per-frame optimisation loop on a synthetic multi-view sequence.

    python tools/train_synthetic.py [--P 100000 --res 512 --views 8 --steps 300]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/train_synthetic.py ...

Targets are renders of a hidden "ground-truth" Gaussian set on the head template from `--views` cameras (the FLAME
weights are absent: declared stand-in for the INSTA sequence); the trained set starts from perturbed appearance and
opacity.  Prints one JSON line: optimisation steps/s (render + L1 + backward + densification statistics + Adam, one
frame per rank per step, one flat-gradient all-reduce when N > 1) and the loss before / after.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
os.environ.setdefault("FR_TUNE_RUNTIME", "1")   # (fateavatar_amd.tune_runtime() at import: the runtime switches the measurements are quoted under)
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd import dp, scenes  # noqa: E402
from fateavatar_amd.model import FlatGaussians, TorchCamera  # noqa: E402
from fateavatar_amd.render import render  # noqa: E402
from fateavatar_amd.train import TrainStep  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--P", type=int, default=100_000)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--lanes", dest="chain", action="store_false",
                    help="--views-per-step K: K lanes on K streams (a graph per frame) instead of ONE launch chain on one "
                         "stream (render_batch: the default)")
    ap.add_argument("--random-order", action="store_true",
                    help="FateAvatar step: area-weighted random binding points, stored as drawn, instead of the reference's "
                         "UV-raster initialisation (model/fateavatar.py:128-133)")
    ap.add_argument("--keep-coherent", action="store_true",
                    help="FateAvatar step: re-store the rows in a spatially coherent order after every densification (an "
                         "extension; the reference appends)")
    ap.add_argument("--binding-op", action="store_true",
                    help="FateAvatar step: the stand-alone binding kernels instead of the binding inside the rasterizer's kernels")
    ap.add_argument("--views-per-step", type=int, default=1,
                    help="--fateavatar: frames per optimisation step, rendered in flight together (the reference's batch)")
    ap.add_argument("--fateavatar", action="store_true",
                    help="FateAvatar's own loop: mesh-bound parameters (offset / rotation / scaling / colour / opacity), "
                         "synthetic INSTA-layout sequence with per-frame posed mesh, SH degree 0")
    a = ap.parse_args()
    rank, world, local = dp.init_from_env()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if a.fateavatar:
        return main_fateavatar(a, rank, world, dev)
    truth = scenes.head_scene(P=a.P, res=a.res, sh_degree=a.sh_degree, seed=0, opacity=0.5)
    cams = [TorchCamera(scenes.head_scene(P=8, res=a.res, sh_degree=a.sh_degree, seed=0, view=v, n_views=a.views).camera, dev)
            for v in range(a.views)]
    bg = torch.from_numpy(truth.bg).to(dev)
    pc_true = FlatGaussians(truth.means3D, truth.shs, truth.opacities, truth.scales, truth.rotations, a.sh_degree, dev,
                            fused_activations=True)
    with torch.no_grad():
        gts = [render(c, pc_true, bg)["render"].clone() for c in cams]
    rng = np.random.default_rng(1)
    shs0 = (truth.shs + 0.2 * rng.standard_normal(truth.shs.shape)).astype(np.float32)
    pc = FlatGaussians(truth.means3D, shs0, truth.opacities * 0.6, truth.scales, truth.rotations, a.sh_degree, dev,
                       fused_activations=True)
    cam = TorchCamera(scenes.head_scene(P=8, res=a.res, sh_degree=a.sh_degree, seed=0, view=0, n_views=a.views).camera, dev)
    ts = TrainStep(pc, cam, bg, use_graph=not a.no_graph)
    losses = []
    warm = 10
    for it in range(warm):
        v = (it * world + rank) % a.views
        losses.append(ts.step(cams[v], gts[v]).clone())
    torch.cuda.synchronize()
    dp.barrier()
    t0 = time.perf_counter()
    for it in range(warm, warm + a.steps):
        v = (it * world + rank) % a.views
        loss = ts.step(cams[v], gts[v])
        if it >= warm + a.steps - 4:     # (the loss lives in the step's static buffer: a trainer reads it now and then,
            losses.append(loss.clone())  # and a 4-byte device copy per step costs 14 us of a 180 us step)
    torch.cuda.synchronize()
    dp.barrier()
    dt = time.perf_counter() - t0
    ts.check()
    if rank == 0:
        l = [float(x) for x in losses]
        print(json.dumps({"metric": "optimisation steps/s (render + L1 + backward + stats + Adam)", "value": round(a.steps / dt, 1),
                          "frames_per_s": round(world * a.steps / dt, 1), "n_gpus": world, "ms_per_step": round(dt / a.steps * 1e3, 4),
                          "P": a.P, "res": a.res, "views": a.views, "graph": not a.no_graph,
                          "loss_first": round(float(np.mean(l[:4])), 6), "loss_last": round(float(np.mean(l[-4:])), 6)}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def fateavatar_setup(P, res, dev, views=8, views_per_step=1, use_graph=True, chain=True, fold_binding=True, order="uv",
                     keep_coherent=False):
    """FateAvatar's optimisation step on the synthetic INSTA-layout sequence: the mesh-bound Gaussian
    set, its step object (AvatarStep, or AvatarBatchStep for K > 1 frames per step), cameras, posed meshes and targets
    rendered from a hidden ground-truth set.  Used by this script and by bench.py's `avatar` mode.
    `order="uv"`: the reference's own initialisation — `uniform_sampling_barycoords(P, ...)` on the template's UV raster
    (model/fateavatar.py:128-133), rows in row-major texel order; `order="random"`: the area-weighted draw as drawn (the A/B)."""
    from fateavatar_amd import insta
    from fateavatar_amd.avatar import AvatarGaussians, AvatarStep, _BoundFrame
    from fateavatar_amd.binding import bind_gaussians
    n_frames = max(views, 8)
    transform, posed, faces = insta.synthetic_sequence(n_frames, res, seed=0)
    verts, _, _ = scenes.head_geometry()
    pc = AvatarGaussians.from_template(dev, num_points=P, sampling=order, rng=np.random.default_rng(0))
    fi, bc = pc.face_index.cpu().numpy(), pc.bary_coords.cpu().numpy()
    scale_init = float(pc._scaling[0, 0])
    cams = [TorchCamera(c, dev) for c in insta.camera_arrays(transform)]
    posed_t, faces_t, canon = torch.from_numpy(posed).to(dev), torch.from_numpy(faces).to(dev), torch.from_numpy(verts).to(dev)
    bg = torch.ones(3, device=dev)
    # hidden ground truth: same binding, other appearance
    gt = AvatarGaussians(fi, bc, scale_init, dev)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        gt._features_dc.copy_((torch.rand(gt.P, 1, 3, generator=g) * 2.0 - 1.0).to(dev))
        gt._opacity.fill_(float(np.log(0.6 / 0.4)))
        gt._offset.copy_((0.3 * torch.randn(gt.P, 1, generator=g)).to(dev))
    ref = AvatarStep(gt, faces_t, canon, TorchCamera(insta.camera_arrays(transform)[0], dev), bg, use_graph=False)
    gts = []
    with torch.no_grad():
        for f in range(n_frames):
            xyz, rot, scl = bind_gaussians(posed_t[f], ref.faces, gt.face_index, gt.bary_coords, ref.face_scale_canonical, gt._offset,
                                           gt._rotation, gt._scaling, ref.shell_len, True)
            gts.append(render(cams[f], _BoundFrame(xyz, gt, rot, scl, None), bg)["render"].clone())
    K = max(1, views_per_step)
    cam0 = TorchCamera(insta.camera_arrays(transform)[0], dev)
    if K == 1:
        st = AvatarStep(pc, faces_t, canon, cam0, bg, use_graph=use_graph, fold_binding=fold_binding, keep_coherent=keep_coherent)
    else:   # the reference's batch of K frames per step (model/fateavatar.py:251-276), in flight together
        from fateavatar_amd.avatar import AvatarBatchStep
        st = AvatarBatchStep(pc, faces_t, canon, cam0, bg, views_per_step=K, use_graph=use_graph, chain=chain,
                             fold_binding=fold_binding, keep_coherent=keep_coherent)
    return dict(st=st, cams=cams, posed=posed_t, gts=gts, n_frames=n_frames, K=K)


def main_fateavatar(a, rank, world, dev):
    su = fateavatar_setup(a.P, a.res, dev, views=a.views, views_per_step=a.views_per_step, use_graph=not a.no_graph, chain=a.chain,
                           fold_binding=not a.binding_op, order="random" if a.random_order else "uv", keep_coherent=a.keep_coherent)
    st, cams, posed_t, gts, n_frames, K = su["st"], su["cams"], su["posed"], su["gts"], su["n_frames"], su["K"]

    def one_step(it, keep=True):
        if K == 1:
            f = (it * world + rank) % n_frames
            loss = st.step(cams[f], posed_t[f], gts[f])
        else:
            fs = [((it * world + rank) * K + k) % n_frames for k in range(K)]
            loss = st.step([cams[f] for f in fs], [posed_t[f] for f in fs], [gts[f] for f in fs])[0]   # (first lane's loss)
        # the loss lives in the step's static buffer: a trainer reads it now and then (a 4-byte device copy per step costs
        # 14 us of a 180 us step)
        return loss.clone() if keep else None

    losses, warm = [], 10
    for it in range(warm):
        losses.append(one_step(it))
    torch.cuda.synchronize()
    dp.barrier()
    t0 = time.perf_counter()
    for it in range(warm, warm + a.steps):
        if it >= warm + a.steps - 4:
            losses.append(one_step(it))
        else:
            one_step(it, keep=False)
    t_host = time.perf_counter() - t0           # the host's share: everything enqueued
    torch.cuda.synchronize()
    dp.barrier()
    dt = time.perf_counter() - t0
    st.check()
    if rank == 0:
        l = [float(x) for x in losses]
        print(json.dumps({"host_enqueue_ms_per_step": round(t_host / a.steps * 1e3, 4),
                          "metric": "FateAvatar optimisation steps/s (bind + render + L1 + backward + stats + Adam)",
                          "storage_order": "random (area-weighted draw)" if a.random_order else "the reference's UV-raster initialisation (row-major texels)",
                          "binding": "stand-alone kernels" if a.binding_op else "inside the per-Gaussian kernels (fr_aux::binding)",
                          "value": round(a.steps / dt, 1), "frames_per_s": round(world * K * a.steps / dt, 1), "n_gpus": world,
                          "views_per_step": K, "launch_chain": bool(a.chain) if K > 1 else None,
                          "ms_per_step": round(dt / a.steps * 1e3, 4), "P": a.P, "res": a.res, "frames": n_frames, "sh_degree": 0,
                          "graph": not a.no_graph, "overflows": st.overflows,
                          "loss_first": round(float(np.mean(l[:4])), 6), "loss_last": round(float(np.mean(l[-4:])), 6)}))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
