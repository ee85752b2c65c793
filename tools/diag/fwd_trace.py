"""Per-unit phase timeline of the forward blend (GPU box).  Needs the -DFR_DIAG_FWD_TRACE build:
    tools/diag/build_variant.sh ftrace -DFR_DIAG_FWD_TRACE        (here)
    FR_HIP_LIB=$PWD/.ab/libfr_ftrace.so python tools/diag/fwd_trace.py [--P 100000 --res 512]   (GPU box)
Stamps (shader cycles): 0 entry, 1 records in registers (counts -> descriptor -> ids -> records), 2 staged + masks + transpose,
3 local walk done, 4 entering transmittance known (look-back), 5 row written, 6 (a tile's last unit) the others' rows are in,
7 gathered; values: 8 / 12 s_memrealtime at entry / exit, 9 pairs, 10 list length, 11 segment index."""
import argparse, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes, _lib  # noqa: E402
from tests.util import HipFrame  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=100000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--opacity", type=float, default=0.1)
a = ap.parse_args()
dev = torch.device("cuda:0")
s = scenes.head_scene(P=a.P, res=a.res, opacity=a.opacity)
f = HipFrame(s, dev)
f2 = HipFrame(s, dev)   # (the traced launch: a second frame, warm)
torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(16384 * 16, np.uint64)
L.fr_debug_read_fwd_trace.argtypes = [C.c_void_p, C.c_size_t]
assert L.fr_debug_read_fwd_trace(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(16384, 16).astype(np.int64)
work = t[:, 5] > 0
# (units are traced by SLOT, and the slots a frame uses differ from frame to frame: keep the stamps of the last launch only)
work &= t[:, 8] > t[work, 8].max() - 20000
w = t[work]
last = w[:, 7] > 0
print(f"units {work.sum()}, of which last units of their tile (gatherers) {last.sum()}; instances {f2.counts.num_instances}")
rt0 = w[:, 8].min()
ent, ext = (w[:, 8] - rt0) / 100.0, (w[:, 12] - rt0) / 100.0
print(f"kernel span first entry -> last exit {ext.max():.2f} us; entries p50 {np.percentile(ent, 50):.2f} p90 {np.percentile(ent, 90):.2f} max {ent.max():.2f}; "
      f"exits p50 {np.percentile(ext, 50):.2f} p90 {np.percentile(ext, 90):.2f} p99 {np.percentile(ext, 99):.2f}")
names = ["entry -> records in registers", "-> staged, masks, transpose", "-> local walk done", "-> entering T known (look-back)", "-> row written"]
d = np.diff(w[:, :6], axis=1)
print(f"{'segment':36s} {'mean':>8s} {'p50':>8s} {'p90':>8s} {'max':>8s}  (shader cycles)")
for k, n in enumerate(names):
    x = d[:, k]
    print(f"{n:36s} {x.mean():8.0f} {np.percentile(x, 50):8.0f} {np.percentile(x, 90):8.0f} {x.max():8.0f}")
g = w[last]
for n, x in (("(last units) row -> others' rows in", g[:, 6] - g[:, 5]), ("(last units) gather", g[:, 7] - g[:, 6])):
    print(f"{n:36s} {x.mean():8.0f} {np.percentile(x, 50):8.0f} {np.percentile(x, 90):8.0f} {x.max():8.0f}")
tot = np.where(last, w[:, 7], w[:, 5]) - w[:, 0]
print(f"{'whole unit':36s} {tot.mean():8.0f} {np.percentile(tot, 50):8.0f} {np.percentile(tot, 90):8.0f} {tot.max():8.0f}")
for i in np.argsort(-ext)[:8]:
    print(f"late unit: entry {ent[i]:5.2f} exit {ext[i]:5.2f} us, cycles {tot[i]}, segments {d[i].tolist()}, gather {(w[i, 6] - w[i, 5], w[i, 7] - w[i, 6]) if last[i] else None}, "
          f"pairs {w[i, 9]}, list {w[i, 10]}, segment {w[i, 11]}")
nseg = (w[:, 10] + 63) // 64
print("units by list length (units per tile): " + ", ".join(f"{k}: {int((nseg == k).sum())}" for k in sorted(set(nseg.tolist()))[:12]))
idx = np.nonzero(work)[0]
late = ent > 1.5
print(f"late entries (> 1.5 us): {late.sum()}; slots {idx[late][:24].tolist()}; entry us {np.round(ent[late][:24], 2).tolist()}")
print("entry time by slot decile:", [round(float(np.mean(ent[(idx >= np.percentile(idx, q)) & (idx <= np.percentile(idx, q + 10))])), 2) for q in range(0, 100, 10)])
