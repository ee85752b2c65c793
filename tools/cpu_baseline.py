#!/usr/bin/env python
"""CPU baseline table: the oracle (oracle/fr_oracle.c, OpenMP; kind "port" — the reference has no CPU
rasterizer) timed on the host cores of the box it runs on, for BASELINE.json configs 1, 2 and 5, on all cores and on
one core.  Prints one JSON line per row.  Test/measurement infrastructure: it times the oracle, nothing else."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402


def run(name, s, backward, threads, budget_s):
    c = s.camera
    H, W = c.image_height, c.image_width
    kw = dict(bg=s.bg, means3D=s.means3D, opacities=s.opacities, viewmatrix=c.world_view_transform,
              projmatrix=c.full_proj_transform, campos=c.camera_center, tanfovx=c.tanfovx, tanfovy=c.tanfovy, H=H, W=W,
              shs=s.shs, sh_degree=s.sh_degree, scales=s.scales, rotations=s.rotations)
    dpix = np.full((3, H, W), 1.0 / (3 * H * W), np.float32)
    oracle.set_num_threads(threads)
    f = oracle.forward(**kw)
    if backward:
        oracle.backward(f, dpix)
    n, t0 = 0, time.perf_counter()
    while True:
        f = oracle.forward(**kw)
        if backward:
            oracle.backward(f, dpix)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 50:
            break
    dt = time.perf_counter() - t0
    print(json.dumps({"config": name, "pass": "forward+backward" if backward else "forward", "threads": threads,
                      "frames": n, "frames_per_s": round(n / dt, 4), "ms_per_frame": round(dt / n * 1e3, 2)}), flush=True)


if __name__ == "__main__":
    host_threads = os.cpu_count() or 1
    all_cores = oracle.cpu_quota_cores()   # what the container may use: "all cores" below means these
    cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][:1]
    print(json.dumps({"host_hardware_threads": host_threads, "cpu_quota_cores": all_cores, "cpu": cpu[0] if cpu else "?"}), flush=True)
    cfgs = [("config 1: 10k random, 256^2, SH deg 0", scenes.random_scene(10000, 256, 256, sh_degree=0, seed=0), False),
            ("config 2: 100k head template, 512^2, SH deg 3", scenes.head_scene(), True),
            ("config 5: 500k head template, 1024^2, SH deg 3", scenes.head_scene(P=500000, res=1024), True)]
    for name, s, bwd in cfgs:
        run(name, s, bwd, all_cores, 6.0)
    # config 2 over the thread counts: where does the port stop scaling on this host?
    name, s, bwd = cfgs[1]
    for t in sorted({t for t in (2, 4, 8, 16, 32, 64, 128, 256) if t != all_cores and t <= host_threads}):
        run(name, s, bwd, t, 4.0)   # (beyond the quota the threads time-slice: the rows are there to show it)
    for name, s, bwd in cfgs[:2]:
        run(name, s, bwd, 1, 12.0)
