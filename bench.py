#!/usr/bin/env python
"""bench.py — render+backward frames/s of the MI355X rasterizer on BASELINE.json's metric config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = ONE frame of the hot path per GPU: `render(camera, gaussians, bg)` through the reference's Python
API (activations inside the HIP preprocess kernels by default, `--torch-activations` for the reference's
stock PyTorch ops) followed by the backward pass down to the raw Gaussian parameters; at N > 1 each rank
renders its own view of the replicated Gaussians and the step ends with ONE RCCL all-reduce(AVG) of the flat
gradient buffer.
Workload (config.workload): BASELINE.json configs[1] — 100 000 Gaussians sampled on the head
template, 512x512, SH degree 3 (M=16), synthetic data, random-init appearance.  Inputs are resident
in HBM before the timed region.

The JSON line carries
  roofline     — the blend backward (the graded kernel): algorithmic bytes 76*R + 20*H*W + 8*T
                 (R = num_rendered and T = 16x16 tiles in reference semantics) divided by
                 that kernel's mean launch duration, measured with HIP events on the launch stream over
                 eager launches of the same frame right after the timed region (events recorded inside a
                 replayed graph cannot be read back), against the 8 TB/s HBM peak.
  cpu_baseline — the CPU oracle (oracle/fr_oracle.c, OpenMP, kind "port": the reference has no CPU
                 rasterizer) timed on the host cores on a bounded sample of the same frames.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from fateavatar_amd import _lib, dp, scenes  # noqa: E402
from fateavatar_amd.model import FlatGaussians, TorchCamera  # noqa: E402
from fateavatar_amd.render import render  # noqa: E402
from fateavatar_amd import rasterizer  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s achievable)


def cpu_baseline(scene, n_frames: int):
    """Oracle forward+backward on the host, all cores.  Returns (frames/s, threads, sample text)."""
    from oracle import oracle
    c = scene.camera
    H, W = c.image_height, c.image_width
    dpix = np.full((3, H, W), 1.0 / (3 * H * W), np.float32)
    kw = dict(bg=scene.bg, means3D=scene.means3D, opacities=scene.opacities, viewmatrix=c.world_view_transform,
              projmatrix=c.full_proj_transform, campos=c.camera_center, tanfovx=c.tanfovx, tanfovy=c.tanfovy, H=H,
              W=W, shs=scene.shs, sh_degree=scene.sh_degree, scales=scene.scales, rotations=scene.rotations)
    f = oracle.forward(**kw)  # warm-up (page-in, thread pool)
    oracle.backward(f, dpix)
    t0 = time.perf_counter()
    for _ in range(n_frames):
        f = oracle.forward(**kw)
        oracle.backward(f, dpix)
    dt = time.perf_counter() - t0
    return n_frames / dt, oracle.num_threads(), f"{n_frames} frames fwd+bwd of the same workload ({dt:.1f} s)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--P", type=int, default=100_000)
    ap.add_argument("--res", type=int, default=512)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames of the CPU baseline sample (0 = skip)")
    ap.add_argument("--graph", dest="graph", action="store_true", default=True, help="replay the frame as a HIP graph")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--fused-activations", dest="fused_activations", action="store_true", default=True,
                    help="sigmoid/exp/normalize inside the HIP kernels (SURVEY.md §8f row 1)")
    ap.add_argument("--torch-activations", dest="fused_activations", action="store_false",
                    help="reference behaviour: activations as stock PyTorch ops")
    args = ap.parse_args()

    rank, world, local = dp.init_from_env()
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # replicated Gaussians (same seed on every rank), one view per rank
    scene = scenes.head_scene(P=args.P, res=args.res, sh_degree=args.sh_degree, seed=0, view=rank, n_views=max(world, 1))
    pc = FlatGaussians(scene.means3D, scene.shs, scene.opacities, scene.scales, scene.rotations, scene.sh_degree, dev,
                       fused_activations=args.fused_activations)
    cam = TorchCamera(scene.camera, dev)
    bg = torch.from_numpy(scene.bg).to(dev)
    H = W = args.res
    # upstream gradient of the image: d(L1-mean against a fixed random target)/d(pixel) has magnitude 1/(3HW) and
    # a random sign; it is fixed, so the step is exactly render + backward
    dL_dpix = ((torch.rand((3, H, W), generator=torch.Generator().manual_seed(1 + rank)) < 0.5).float() * 2 - 1)
    dL_dpix = (dL_dpix / (3 * H * W)).to(dev)

    def frame():
        pc.begin_step()                       # grads set to None: backward assigns (zero_grad(set_to_none=True))
        out = render(cam, pc, bg)             # activations + HIP rasterizer forward
        torch.autograd.backward(out["render"], grad_tensors=dL_dpix)  # HIP rasterizer backward + activation backward

    def eager_step():
        frame()
        if world > 1:
            dp.allreduce_mean_(pc.collect_grads())

    # eager warm-up: sizes the binning capacity (high-water mark) and fills the allocator pools
    for _ in range(max(3, args.warmup // 2)):
        eager_step()
    torch.cuda.synchronize()
    counts = rasterizer.last_counts[local]

    graph = None
    if args.graph:
        # The per-frame work is launch-bound on the host (~40 small launches): capture ONE frame (activations,
        # rasterizer forward, loss, backward) into a HIP graph and replay it.  The rasterizer runs in no-wait
        # mode inside the graph (no host synchronisation at all); overflow of the binning capacity is checked
        # after the timed region.
        rasterizer.set_no_wait(True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                frame()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread of an N > 1 run must not trip the capture
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            frame()
        torch.cuda.synchronize()

    def step():
        if graph is not None:
            graph.replay()
            if world > 1:
                dp.allreduce_mean_(pc.collect_grads())
        else:
            eager_step()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dp.barrier()
    elapsed = time.perf_counter() - t0
    if graph is not None:
        if rasterizer.check_async_overflow(local):
            c = rasterizer.last_counts[local]
            raise SystemExit(f"rank {rank}: binning capacity overflowed inside the captured graph "
                             f"(instances {c.num_instances}, num_rendered {c.num_rendered}, max list {c.max_tile_list}); "
                             "rerun (capacity hint was raised)")
        rasterizer.set_no_wait(False)
    # per-kernel durations: HIP events around every stage launch, on the launch stream, over eager replays of
    # the same frame right after the timed region (event records inside a replayed graph cannot be read back)
    _lib.profile_enable(local, True)
    for _ in range(min(args.steps, 50)):
        eager_step()
    torch.cuda.synchronize()
    prof = _lib.profile_read(local)
    _lib.profile_enable(local, False)
    counts = rasterizer.last_counts[local]

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        fps = world * args.steps / elapsed
        R = int(counts.num_rendered)
        T16 = ((W + 15) // 16) * ((H + 15) // 16)
        bytes_bwd = 76 * R + 20 * H * W + 8 * T16
        ms, n = prof["blend_bwd"]
        roof = None
        if n:
            avg_s = ms / n * 1e-3
            ach = bytes_bwd / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "r01_blend_bwd_traffic.json")
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    if tj.get("P") == args.P and tj.get("res") == args.res:
                        traffic = tj.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            roof = {"bound": "hbm", "kernel": "k_blend_bwd", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "algorithmic_bytes": bytes_bwd, "avg_launch_us": round(avg_s * 1e6, 2), "launches": n}
        stages = {k: round(v[0] / v[1] * 1e3, 2) for k, v in prof.items() if v[1]}
        cpu = None
        if args.cpu_frames > 0 and world == 1:
            v, cores, sample = cpu_baseline(scene, args.cpu_frames)
            cpu = {"value": round(v, 3), "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample}
        cfg_name = ("BASELINE.json configs[1]" if (args.P, args.res) == (100_000, 512) else
                    "SURVEY.md §8d config 5 (not the metric's configuration)" if (args.P, args.res) == (500_000, 1024)
                    else "custom size (not the metric's configuration)")
        line = {
            "metric": "render+backward frames/sec at 512^2, 100k Gaussians", "value": round(fps, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: {args.P} Gaussians on the head template, "
                                   f"{args.res}x{args.res}, SH deg {args.sh_degree} (M={(args.sh_degree + 1) ** 2}), "
                                   "forward+backward through render() with a fixed dL/dpixel",
                       "frames_per_step_per_gpu": 1,
                       "activations": "fused in the HIP preprocess kernels" if args.fused_activations else "stock PyTorch",
                       "launch": "hipgraph replay" if args.graph else "eager",
                       "parallelism": f"dp{world} (one view per GPU, flat-grad all-reduce)",
                       "num_rendered": R, "tile_instances_8x8": int(counts.num_instances),
                       "max_tile_list": int(counts.max_tile_list)},
            "roofline": roof, "cpu_baseline": cpu, "stage_us": stages,
        }
        print(json.dumps(line), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
