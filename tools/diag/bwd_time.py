"""Isolated duration of the blend backward launch (GPU box): stage events of the C ABI's profiler (dispatch-tied), config 2 by
default.  FR_HIP_LIB selects a variant build, FR_BLEND_BWD the kernel."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes, _lib  # noqa: E402
from tests.util import HipFrame  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=100000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--opacity", type=float, default=0.1)
ap.add_argument("--scale", type=float, default=None)
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
s = scenes.head_scene(P=a.P, res=a.res, opacity=a.opacity, scale=a.scale)
f = HipFrame(s, dev)
g = (np.random.default_rng(0).uniform(-1, 1, (3, a.res, a.res)) / (a.res * a.res)).astype(np.float32)
for _ in range(5):
    f.backward(g)
torch.cuda.synchronize()
_lib.profile_enable(0, True)
for _ in range(a.iters):
    f.backward(g)
torch.cuda.synchronize()
t = _lib.profile_read(0)
_lib.profile_enable(0, False)
print(f"{os.environ.get('FR_HIP_LIB', 'default lib').split('/')[-1]} FR_BLEND_BWD={os.environ.get('FR_BLEND_BWD', '')} P={a.P} res={a.res} "
      f"opacity={a.opacity} scale={a.scale}: instances={f.counts.num_instances} stage us: { {k: round(v[0] * 1e3 / max(v[1], 1), 2) for k, v in t.items() if v[1]} }")
