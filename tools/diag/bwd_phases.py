"""Where the cycles of k_unit_blend_bwd_sparse go, per phase (GPU box).  Needs the -DFR_DIAG_TRACE build:
    tools/diag/build_variant.sh trace -DFR_DIAG_TRACE        (here)
    FR_HIP_LIB=$PWD/.ab/libfr_trace.so python tools/diag/bwd_phases.py [--P 100000 --res 512 --opacity 0.1]
Prints mean / p90 / max cycles per unit of: loads, staging, range set-up, phase A, phase B, flush, and the wall-clock
spread of the waves' starts and ends."""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes, rasterizer, _lib  # noqa: E402
from tests.util import HipFrame  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=100000); ap.add_argument("--res", type=int, default=512)
ap.add_argument("--opacity", type=float, default=0.1); ap.add_argument("--scale", type=float, default=None)
a = ap.parse_args()
dev = torch.device("cuda:0")
s = scenes.head_scene(P=a.P, res=a.res, scale=a.scale, opacity=a.opacity)
f = HipFrame(s, dev); cam = s.camera; H = W = a.res
g = torch.ones(3, H, W, device=dev) / (3 * H * W)
for _ in range(3):
    r = rasterizer.rasterize_gaussians(f.bg, f.means3D, f.colors, f.op, f.scales, f.rots, 1.0, f.cov, f.view, f.proj, cam.tanfovx, cam.tanfovy, H, W, f.sh, s.sh_degree, f.campos, False, False)
    rasterizer.rasterize_gaussians_backward(f.bg, f.means3D, r[2], f.colors, f.scales, f.rots, 1.0, f.cov, f.view, f.proj, cam.tanfovx, cam.tanfovy, g, f.sh, s.sh_degree, f.campos, r[3], r[0], r[4], r[5], False)
torch.cuda.synchronize()
nu = int(r[5][:64].view(torch.int32).cpu().numpy()[14])
L = _lib.lib()
buf = np.zeros((32768, 8), np.uint64)
L.fr_diag_read_trace.argtypes = [C.c_void_p, C.c_size_t]
assert L.fr_diag_read_trace(buf.ctypes.data, buf.nbytes) == 0
t = buf[:min(nu, 32768)].astype(np.float64)
names = ["loads", "staging", "-", "range set-up", "phase A", "phase B", "flush"]
print(f"units {nu}; cycles per unit (mean / p90 / max), share of the mean total")
tot = t[:, :7].sum(1)
for k, n in enumerate(names):
    c = t[:, k]
    print(f"  {n:14s} {c.mean():9.0f} {np.percentile(c, 90):9.0f} {c.max():9.0f}   {100 * c.mean() / tot.mean():5.1f} %")
print(f"  {'total':14s} {tot.mean():9.0f} {np.percentile(tot, 90):9.0f} {tot.max():9.0f}")
rt = buf[:min(nu, 32768), 7]
rt = rt[rt != 0]
st = (rt >> np.uint64(32)).astype(np.int64) & 0xFFFFFFFF
en = (rt & np.uint64(0xFFFFFFFF)).astype(np.int64)
last = en > en.max() - 20000          # (rows of the LAST launch only: units it skipped keep an earlier launch's stamps)
st, en = st[last], en[last]
t0 = st.min()
st, en = (st - t0) * 0.01, (en - t0) * 0.01   # us (100 MHz)
print("waves of the last launch: %d" % st.size)
print("wave start after the first wave (us): p1 %.2f p10 %.2f median %.2f p90 %.2f max %.2f" % (np.percentile(st, 1), np.percentile(st, 10), np.median(st), np.percentile(st, 90), st.max()))
print("wave end   after the first wave's start (us): median %.2f p90 %.2f max %.2f" % (np.median(en), np.percentile(en, 90), en.max()))
print("wave life (us): mean %.2f p90 %.2f max %.2f" % ((en - st).mean(), np.percentile(en - st, 90), (en - st).max()))
