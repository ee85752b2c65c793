"""Which units end k_unit_blend_bwd_sparse (GPU box; -DFR_DIAG_TRACE build: FR_HIP_LIB=$PWD/.ab/libfr_trace.so).  Per class of
units (pairs named, from the work list the forward leaves): how many, their waves' start / life / end, and who the last 2 %
of the waves to finish are."""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes, rasterizer, _lib  # noqa: E402
from tests.util import HipFrame  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=100000); ap.add_argument("--res", type=int, default=512)
ap.add_argument("--opacity", type=float, default=0.1)
a = ap.parse_args()
dev = torch.device("cuda:0")
s = scenes.head_scene(P=a.P, res=a.res, opacity=a.opacity)
f = HipFrame(s, dev); cam = s.camera; H = W = a.res
g = torch.ones(3, H, W, device=dev) / (3 * H * W)
for _ in range(3):
    r = rasterizer.rasterize_gaussians(f.bg, f.means3D, f.colors, f.op, f.scales, f.rots, 1.0, f.cov, f.view, f.proj, cam.tanfovx, cam.tanfovy, H, W, f.sh, s.sh_degree, f.campos, False, False)
    rasterizer.rasterize_gaussians_backward(f.bg, f.means3D, r[2], f.colors, f.scales, f.rots, 1.0, f.cov, f.view, f.proj, cam.tanfovx, cam.tanfovy, g, f.sh, s.sh_degree, f.campos, r[3], r[0], r[4], r[5], False)
torch.cuda.synchronize()
nu = int(r[5][:64].view(torch.int32).cpu().numpy()[14])
binning = r[4]
off = (-binning.data_ptr()) & 255
wl = binning[off:off + 32 * nu].view(torch.int32).view(-1, 8).cpu().numpy()   # BwdUnit: d(4), u, pad[0]=pairs, pad[1]=walk trips
pairs = np.zeros(nu, np.int64); slot_of = np.zeros(nu, np.int64)
pairs[wl[:, 4]] = wl[:, 5]; slot_of[wl[:, 4]] = np.arange(nu)
L = _lib.lib()
buf = np.zeros((32768, 8), np.uint64)
L.fr_diag_read_trace.argtypes = [C.c_void_p, C.c_size_t]
assert L.fr_diag_read_trace(buf.ctypes.data, buf.nbytes) == 0
rt = buf[:nu, 7]
st = ((rt >> np.uint64(32)).astype(np.int64) & 0xFFFFFFFF) * 0.01
en = (rt & np.uint64(0xFFFFFFFF)).astype(np.int64) * 0.01
ok = en > en.max() - 200
t0 = st[ok].min()
st, en = st - t0, en - t0
print(f"units {nu}; kernel ~ {en[ok].max():.2f} us after the first wave's start")
for lo, hi in ((0, 300), (300, 500), (500, 640), (640, 800), (800, 1000), (1000, 10 ** 6)):
    m = ok & (pairs >= lo) & (pairs < hi)
    if m.any():
        print(f"  pairs [{lo:4d},{hi:6d}): {int(m.sum()):5d} units  slot median {np.median(slot_of[m]):7.0f}  start {np.median(st[m]):5.2f}  life mean {np.mean(en[m]-st[m]):5.2f} max {np.max(en[m]-st[m]):5.2f}  end median {np.median(en[m]):5.2f} max {np.max(en[m]):5.2f}")
last = np.argsort(en * ok)[-max(1, nu // 50):]
print("last 2 % of the waves to finish: pairs", np.percentile(pairs[last], [0, 25, 50, 75, 100]).astype(int).tolist(),
      " slots", np.percentile(slot_of[last], [0, 50, 100]).astype(int).tolist(), " life", np.round(np.percentile((en - st)[last], [0, 50, 100]), 2).tolist(),
      " start", np.round(np.percentile(st[last], [0, 50, 100]), 2).tolist())
tot = buf[:nu, :7].astype(np.float64).sum(1)
print("cycles by phase of those:", np.round(buf[last, :7].astype(np.float64).mean(0)).astype(int).tolist(), " of all:", np.round(buf[:nu, :7][ok].astype(np.float64).mean(0)).astype(int).tolist())
