"""Dump the task lengths of the blend backward (GPU box; -DFR_BWD_TRACE build): per unit the 64 phase-A chain lengths
(records per pixel) and the 64 phase-B walk lengths (pixels per record) -> gpurun_out/<name>.npy  (uint8 [units, 128])"""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes, _lib  # noqa: E402
from tests.util import HipFrame  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=100000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--opacity", type=float, default=0.1)
ap.add_argument("--scale", type=float, default=None)
ap.add_argument("--out", default="lens_c2")
a = ap.parse_args()
dev = torch.device("cuda:0")
s = scenes.head_scene(P=a.P, res=a.res, opacity=a.opacity, scale=a.scale)
f = HipFrame(s, dev)
g = (np.random.default_rng(0).uniform(-1, 1, (3, a.res, a.res)) / (a.res * a.res)).astype(np.float32)
f.backward(g)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(8192 * 128, np.uint8)
L.fr_debug_read_bwd_lens.argtypes = [C.c_void_p, C.c_size_t]
assert L.fr_debug_read_bwd_lens(buf.ctypes.data, buf.nbytes) == 0
w = np.frombuffer(f.img[:64].cpu().numpy().tobytes(), np.uint32)
nu = min(int(w[14]), 8192)
lens = buf.reshape(8192, 128)[:nu]
os.makedirs(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out"), exist_ok=True)
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", a.out + ".npy"), lens)
print(a.out, "units", nu, "pairs", int(lens[:, :64].sum()), int(lens[:, 64:].sum()), "max A", lens[:, :64].max(), "max B", lens[:, 64:].max())
