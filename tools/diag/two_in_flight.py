"""GPU box: render+backward of K views IN FLIGHT TOGETHER on one GPU — one stream, one fr_handle and one captured graph
per view — against one view at a time.  The frame's kernels are latency-bound at BASELINE config 2 (the chip is not
full): how much of the idle machine does a second frame pick up?
    python tools/diag/two_in_flight.py [K=2] [--P 100000 --res 512]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import rasterizer, scenes  # noqa: E402
from fateavatar_amd.model import FlatGaussians, TorchCamera  # noqa: E402
from fateavatar_amd.render import render  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("K", type=int, nargs="?", default=2)
ap.add_argument("--P", type=int, default=100_000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--probe", action="store_true", help="pick the view streams with fateavatar_amd.streams.concurrent_streams")
ap.add_argument("--extra", action="store_true", help="a fourth stream moves 2 x 23.6 MB per step, as a gradient exchange would")
a = ap.parse_args()
dev = torch.device("cuda:0")


class View:
    def __init__(self, k):
        s = scenes.head_scene(P=a.P, res=a.res, sh_degree=3, seed=0, view=k, n_views=max(a.K, 1))
        self.k = k
        self.pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, s.sh_degree, dev, fused_activations=True)
        self.cam = TorchCamera(s.camera, dev)
        self.bg = torch.from_numpy(s.bg).to(dev)
        g = ((torch.rand((3, a.res, a.res), generator=torch.Generator().manual_seed(1 + k)) < 0.5).float() * 2 - 1)
        self.dL = (g / (3 * a.res * a.res)).to(dev)
        self.stream = torch.cuda.Stream()
        self.graph = None

    def frame(self):
        self.pc.begin_step()
        out = render(self.cam, self.pc, self.bg)
        torch.autograd.backward(out["render"], grad_tensors=self.dL)

    def capture(self):
        with rasterizer.handle_slot(self.k):
            for _ in range(3):
                self.frame()
            torch.cuda.synchronize()
            with rasterizer.no_wait():
                with torch.cuda.stream(self.stream):
                    for _ in range(3):
                        self.frame()
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self.stream):
                    self.frame()
            torch.cuda.synchronize()


views = [View(k) for k in range(a.K)]
if a.probe:
    from fateavatar_amd.streams import concurrent_streams
    for v, st in zip(views, concurrent_streams(a.K, dev, also_with=[torch.cuda.current_stream()])):
        v.stream = st
for v in views:
    v.capture()


xs = torch.cuda.Stream()
xa, xb = torch.zeros(5_900_000, device=dev), torch.zeros(5_900_000, device=dev)


def run(active, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for v in active:
            with torch.cuda.stream(v.stream):
                v.graph.replay()
        if a.extra:
            with torch.cuda.stream(xs):
                xb.copy_(xa, non_blocking=True)
                xa.add_(xb)
    torch.cuda.synchronize()
    return steps * len(active) / (time.perf_counter() - t0)


run(views, 20)
one = run(views[:1], a.steps)
many = run(views, a.steps)
print(f"1 view in flight: {one:.0f} frames/s;  {a.K} in flight: {many:.0f} frames/s ({many / one:.2f}x)")
