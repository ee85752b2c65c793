"""Cost of an oversized binning capacity (the chained blend kernel launches one workgroup per four unit SLOTS)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes, rasterizer  # noqa: E402
from tests.util import HipFrame  # noqa: E402

dev = torch.device("cuda:0")
s = scenes.head_scene()
for cap in (0, 2_000_000, 10_000_000, 50_000_000):
    rasterizer._capacity_hint[dev] = cap
    f = HipFrame(s, dev)
    cam = s.camera
    H = W = 512

    def fwd():
        return rasterizer.rasterize_gaussians(f.bg, f.means3D, f.colors, f.op, f.scales, f.rots, 1.0, f.cov, f.view, f.proj,
                                              cam.tanfovx, cam.tanfovy, H, W, f.sh, s.sh_degree, f.campos, False, False)
    for _ in range(5):
        fwd()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        fwd()
    torch.cuda.synchronize()
    print(f"capacity hint {cap}: forward {(time.perf_counter() - t0) / 50 * 1e6:.1f} us")
