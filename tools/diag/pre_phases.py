"""Where the cycles of k_preprocess_fwd go, per phase and wave (GPU box).  Needs the -DFR_DIAG_PRE_TRACE build:
    tools/diag/build_variant_file.sh pretrace fr_preprocess -DFR_DIAG_PRE_TRACE        (here)
    FR_HIP_LIB=$PWD/.ab/libfr_pretrace.so python tools/diag/pre_phases.py [--P 100000 --res 512]
Prints mean / p90 / max cycles per wave of each phase and the wall-clock spread of the waves' starts and ends."""
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
for _ in range(3):
    r = rasterizer.rasterize_gaussians(f.bg, f.means3D, f.colors, f.op, f.scales, f.rots, 1.0, f.cov, f.view, f.proj, cam.tanfovx, cam.tanfovy, H, W, f.sh, s.sh_degree, f.campos, False, False)
torch.cuda.synchronize()
L = _lib.lib()
nw = min((a.P + 63) // 64, 16384)
buf = np.zeros((16384, 12), np.uint64)
L.fr_diag_read_pre_trace.argtypes = [C.c_void_p, C.c_size_t]
assert L.fr_diag_read_pre_trace(buf.ctypes.data, buf.nbytes) == 0
t = buf[:nw].astype(np.float64)
names = {0: "loads issued", 1: "inputs landed", 2: "geometry + rectangle", 5: "count: tables", 6: "count: candidates", 7: "count: group + issue",
         3: "colour (SH)", 4: "record stores", 8: "count: atomics back", 9: "count: key stores", 10: "tail"}
order = [0, 1, 2, 5, 6, 7, 3, 4, 8, 9, 10]     # (program order: the colour is evaluated behind the issued atomics)
tot = t[:, :11].sum(1)
print(f"waves {nw}; cycles per wave (mean / p90 / max), share of the mean total")
for k in order:
    n = names[k]
    c = t[:, k]
    print(f"  {n:22s} {c.mean():9.0f} {np.percentile(c, 90):9.0f} {c.max():9.0f}   {100 * c.mean() / tot.mean():5.1f} %")
print(f"  {'total':22s} {tot.mean():9.0f} {np.percentile(tot, 90):9.0f} {tot.max():9.0f}")
rt = buf[:nw, 11]
st = (rt >> np.uint64(32)).astype(np.int64) & 0xFFFFFFFF
en = (rt & np.uint64(0xFFFFFFFF)).astype(np.int64)
t0 = st.min()
st, en = (st - t0) * 0.01, (en - t0) * 0.01   # us (100 MHz)
print("wave start after the first wave (us): p10 %.2f median %.2f p90 %.2f max %.2f" % (np.percentile(st, 10), np.median(st), np.percentile(st, 90), st.max()))
print("wave end after the first wave's start (us): median %.2f p90 %.2f max %.2f" % (np.median(en), np.percentile(en, 90), en.max()))
print("wave life (us): mean %.2f p90 %.2f max %.2f" % ((en - st).mean(), np.percentile(en - st, 90), (en - st).max()))
