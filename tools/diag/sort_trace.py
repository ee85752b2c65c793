"""Per-wave phase timeline of k_tile_sort (GPU box).  Needs the -DFR_DIAG_SORT_TRACE build:
    tools/diag/build_variant.sh sotrace -DFR_DIAG_SORT_TRACE        (here)
    FR_HIP_LIB=$PWD/.ab/libfr_sotrace.so python tools/diag/sort_trace.py [--P 100000 --res 512]   (GPU box)
Stamps (s_memrealtime, 10 ns): 0 entry, 1 counters known, 2 behind the workgroup's barrier, 3 cooperative lists done, 4 own list
sorted, 6 allocation known, 7 stores issued; 5 = list length."""
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
buf = np.zeros(16384 * 8, np.uint64)
L.fr_debug_read_sort_trace.argtypes = [C.c_void_p, C.c_size_t]
assert L.fr_debug_read_sort_trace(buf.ctypes.data, buf.nbytes) == 0
t = buf.reshape(16384, 8).astype(np.int64)
G = 1024 * 4
t0 = t[t[:, 0] > 0, 0].min()
def show(name, w, labs):
    print(f"-- {name}: {len(w)} waves")
    for k, lab in labs:
        x = w[:, k]
        ok = x > 0
        if not ok.any():
            continue
        y = (x[ok] - t0) / 100.0
        print(f"   {lab:22s} n {ok.sum():5d} mean {y.mean():6.2f}  p50 {np.percentile(y, 50):6.2f}  p90 {np.percentile(y, 90):6.2f}  max {y.max():6.2f} us")
g = t[:G]
g = g[g[:, 0] > 0]
show("group sorters without a list", g[g[:, 7] == 0], ((0, "entry"), (1, "bitmap searched")))
show("group sorters with a list", g[g[:, 7] > 0], ((0, "entry"), (1, "bitmap searched"), (7, "list done")))
o = t[G:]
o = o[o[:, 0] > 0]
n = o[:, 5]
labs = ((0, "entry"), (1, "counters known"), (2, "keys here"), (3, "sorted"), (4, "place known"), (7, "stores issued"))
show("owners, empty or a group sorter's tile", o[o[:, 7] == 0], labs)
for lo, hi in ((1, 64), (65, 128), (129, 256)):
    show(f"owners, {lo}..{hi} keys", o[(o[:, 7] > 0) & (n >= lo) & (n <= hi)], labs)
