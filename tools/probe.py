"""Quick timing probe (GPU box): forward / backward wall time of one config, through the C ABI mirror."""
import argparse
import sys
import os
import time

import numpy as np
os.environ.setdefault("FR_TUNE_RUNTIME", "1")   # (fateavatar_amd.tune_runtime() at import: the runtime switches the measurements are quoted under)
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd import scenes, rasterizer  # noqa: E402
from tests.util import HipFrame  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=100000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--scale", type=float, default=None)
ap.add_argument("--opacity", type=float, default=0.1)
a = ap.parse_args()
dev = torch.device("cuda:0")
s = scenes.head_scene(P=a.P, res=a.res, scale=a.scale, opacity=a.opacity)
f = HipFrame(s, dev)
c = f.counts
print(f"P={a.P} res={a.res} num_rendered={c.num_rendered} instances={c.num_instances} max_list={c.max_tile_list}")
H = W = a.res
dpix = (np.random.default_rng(0).uniform(-1, 1, (3, H, W)) / (H * W)).astype(np.float32)
g = torch.from_numpy(dpix).to(dev)
cam = s.camera


def fwd():
    return rasterizer.rasterize_gaussians(f.bg, f.means3D, f.colors, f.op, f.scales, f.rots, 1.0, f.cov, f.view, f.proj,
                                          cam.tanfovx, cam.tanfovy, H, W, f.sh, s.sh_degree, f.campos, False, False)


def bwd(r):
    return rasterizer.rasterize_gaussians_backward(f.bg, f.means3D, r[2], f.colors, f.scales, f.rots, 1.0, f.cov, f.view,
                                                   f.proj, cam.tanfovx, cam.tanfovy, g, f.sh, s.sh_degree, f.campos,
                                                   r[3], r[0], r[4], r[5], False)


for _ in range(5):
    r = fwd(); bwd(r)
torch.cuda.synchronize()
if os.environ.get("FR_DEBUG_PAIR_HIST") == "1":  # DeviceCounts sits at the head of the image buffer (fr_common.hpp)
    w = r[5][:64].view(torch.int32).cpu().numpy()
    print(f"units={w[14]} has_dense={w[8]} units by pairs named <=500/<=1000/<=1500/<=2500/more: {w[9:14].tolist()}")
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
tf = tb = 0.0
t0 = time.perf_counter()
for _ in range(a.iters):
    e0.record(); r = fwd(); e1.record(); bwd(r); e2.record()
    torch.cuda.synchronize()
    tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
wall = (time.perf_counter() - t0) / a.iters
print(f"fwd {tf/a.iters*1e3:.1f} us  bwd {tb/a.iters*1e3:.1f} us  wall/iter {wall*1e6:.1f} us (sync each iter)")
t0 = time.perf_counter()
for _ in range(a.iters):
    r = fwd(); bwd(r)
torch.cuda.synchronize()
print(f"pipelined wall/iter {(time.perf_counter()-t0)/a.iters*1e6:.1f} us -> {a.iters/(time.perf_counter()-t0):.0f} frames/s")
