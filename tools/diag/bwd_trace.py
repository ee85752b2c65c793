"""Per-wave phase timeline of the blend backward (GPU box).  Needs the -DFR_BWD_TRACE build:
    tools/diag/build_variant.sh trace -DFR_BWD_TRACE        (here)
    FR_HIP_LIB=$PWD/.ab/libfr_trace.so python tools/diag/bwd_trace.py [--P 100000 --res 512]   (GPU box)
Stamps (shader cycles, s_memtime): 0 entry, 1 every load of the unit landed, 2 the tile has live pixels, 3 staged + scanned +
transposed, 4 phase A of the first range done, 5 phase B done, 6 flush done, 7 unit done; values: 8/12 s_memrealtime at
entry / exit (100 MHz), 9 pair slots of the unit."""
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
g = (np.random.default_rng(0).uniform(-1, 1, (3, a.res, a.res)) / (a.res * a.res)).astype(np.float32)
for _ in range(3):
    f.backward(g)
torch.cuda.synchronize()
if os.environ.get("FR_TRACE_IDLE"):   # the traced launch on an idle GPU, nothing queued in front of it
    import time
    time.sleep(0.05)
    f.backward(g)
    torch.cuda.synchronize()
L = _lib.lib()
buf = np.zeros(8192 * 16, np.uint64)
L.fr_debug_read_bwd_trace.argtypes = [C.c_void_p, C.c_size_t]
rc = L.fr_debug_read_bwd_trace(buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
t = buf.reshape(8192, 16).astype(np.int64)
work = t[:, 7] > 0
w = t[work]
print(f"waves with a unit: {work.sum()} of 8192; instances {f.counts.num_instances}")
rt0 = t[t[:, 8] > 0, 8].min()
print(f"kernel span by s_memrealtime: first entry -> last unit exit {(w[:, 12].max() - rt0) / 100:.2f} us; "
      f"entries spread over {(t[t[:, 8] > 0, 8].max() - rt0) / 100:.2f} us")
names = ["entry->all loads of the unit landed", "loads->tile has live pixels", "live->staged+transposed", "staged->phase A done",
         "phase A->phase B done", "phase B->flush done", "flush->unit done (further ranges)"]
d = np.diff(w[:, :8], axis=1)
print(f"{'segment':38s} {'mean':>8s} {'p50':>8s} {'p90':>8s} {'max':>8s}  (shader cycles)")
for k, n in enumerate(names):
    x = d[:, k]
    print(f"{n:38s} {x.mean():8.0f} {np.percentile(x, 50):8.0f} {np.percentile(x, 90):8.0f} {x.max():8.0f}")
tot = w[:, 7] - w[:, 0]
print(f"{'whole unit':38s} {tot.mean():8.0f} {np.percentile(tot, 50):8.0f} {np.percentile(tot, 90):8.0f} {tot.max():8.0f}")
print(f"pairs/unit mean {w[:, 9].mean():.0f} p90 {np.percentile(w[:, 9], 90):.0f} max {w[:, 9].max()}")
# the slowest units: what made them slow
idx = np.argsort(-tot)[:8]
for i in idx:
    print("slow unit: total", tot[i], "segments", d[i].tolist(), "pairs", w[i, 9])
# correlation of unit duration with pairs
print("corr(total, pairs) =", np.corrcoef(tot, w[:, 9])[0, 1])
# who finishes last: entry / exit (us from the first entry), duration in cycles, and the unit index
ent = (w[:, 8] - rt0) / 100.0
ext = (w[:, 12] - rt0) / 100.0
uidx = np.nonzero(work)[0]
print("exit time percentiles (us): p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(ext, [50, 90, 99, 100])))
print("entry time percentiles (us): p50 %.2f p90 %.2f p99 %.2f max %.2f" % tuple(np.percentile(ent, [50, 90, 99, 100])))
for i in np.argsort(-ext)[:10]:
    print(f"late wave {uidx[i]:5d}: entry {ent[i]:6.2f} exit {ext[i]:6.2f} us, cycles {tot[i]}, segments {d[i].tolist()}, pairs {w[i, 9]}")
print("cycles per us over the slowest waves: %.0f" % np.median(tot[np.argsort(-tot)[:50]] / (ext - ent)[np.argsort(-tot)[:50]]))
# dispatch-order simulation: the waves enter along the measured ramp whatever they carry; kernel end = max(entry + duration)
dur = ext - ent
ramp = np.sort(ent)
n_list, base = w[:, 10], w[:, 11]
def sim(order, label):
    end = (ramp + dur[order]).max()
    print(f"  order {label:34s}: kernel end {end:6.2f} us")
print("dispatch-order simulation (measured ramp, measured durations):")
sim(np.arange(len(dur)), "as dispatched (unit index)")
sim(np.argsort(-dur), "longest first (oracle)")
sim(np.argsort(-w[:, 9]), "most pairs first")
sim(np.argsort(-(n_list.astype(np.int64) * 64 - base), kind="stable"), "longest list first, front units first")
sim(np.argsort(-n_list, kind="stable"), "longest list first")
sim(np.argsort(-np.minimum(n_list - base, 64) * 1000 - n_list, kind="stable"), "full units first, then by list")
print("corr(duration, n) %.3f corr(duration, pairs) %.3f corr(duration, min(n-base,64)) %.3f" % (
    np.corrcoef(dur, n_list)[0, 1], np.corrcoef(dur, w[:, 9])[0, 1], np.corrcoef(dur, np.minimum(n_list - base, 64))[0, 1]))

print("entry time by unit index (us):", " ".join(f"{uidx[i]}:{ent[i]:.2f}" for i in range(0, len(ent), max(1, len(ent) // 24))))
print("entry -> first stamp (loads landed), us: p50 %.2f p90 %.2f max %.2f" % tuple(np.percentile(d[:, 0] / 2230.0, [50, 90, 100])))

raw = (w[:, 13] - w[:, 13].min()) / 100.0
print("RAW entry (first instruction) percentiles (us): p50 %.2f p90 %.2f p99 %.2f max %.2f; kernargs+first code -> 'entry' stamp: mean %.2f max %.2f us"
      % (*np.percentile(raw, [50, 90, 99, 100]), ((w[:, 8] - w[:, 13]) / 100.0).mean(), ((w[:, 8] - w[:, 13]) / 100.0).max()))
print("raw entry by unit index (us):", " ".join(f"{uidx[i]}:{raw[i]:.2f}" for i in range(0, len(ent), max(1, len(ent) // 24))))
# two-class dispatch (heavy units first, arrival order inside a class): what a forward-side classification could give
rng = np.random.default_rng(0)
def two_class(score, frac, label):
    thr = np.quantile(score, 1.0 - frac)
    heavy = score >= thr
    order = np.concatenate([rng.permutation(np.nonzero(heavy)[0]), rng.permutation(np.nonzero(~heavy)[0])])
    sim(order, f"2 classes, top {int(frac * 100)}% by {label}")
for fr in (0.1, 0.25, 0.5):
    two_class(w[:, 9].astype(float), fr, "pairs")
    two_class(d[:, 3].astype(float), fr, "phase-A cycles")
    two_class(dur, fr, "duration (oracle)")
sim(rng.permutation(len(dur)), "random order")

it = w[:, 9] * 0 + 1   # (slot 14 of the trace is the unit index now)
print("forward walk iterations of the units: p10 %d p50 %d p75 %d p90 %d p99 %d max %d; corr(duration, iters) %.3f, corr(phase A cycles, iters) %.3f"
      % (*np.percentile(it, [10, 50, 75, 90, 99, 100]), np.corrcoef(dur, it)[0, 1], np.corrcoef(d[:, 3], it)[0, 1]))
for thr in (8, 10, 11, 12, 14, 16):
    print(f"  iters >= {thr}: {100.0 * (it >= thr).mean():.0f} % of the units")
two_class(it.astype(float), 0.1, "forward iterations")
two_class(it.astype(float), 0.25, "forward iterations")
sim(np.argsort(-it, kind="stable"), "most forward iterations first")
slot = np.nonzero(work)[0]
print("mean duration (us) by slot decile:", " ".join(f"{dur[(slot >= a) & (slot < b)].mean():.1f}" for a, b in zip(np.linspace(0, slot.max() + 1, 11)[:-1], np.linspace(0, slot.max() + 1, 11)[1:])))
print("mean entry (us) by slot decile:   ", " ".join(f"{ent[(slot >= a) & (slot < b)].mean():.1f}" for a, b in zip(np.linspace(0, slot.max() + 1, 11)[:-1], np.linspace(0, slot.max() + 1, 11)[1:])))

# where the waves ran: HW_ID (wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13) + XCC_ID
hw = w[:, 15]
xcc, hwid = hw >> 32, hw & 0xFFFFFFFF
simd, cu, sh, se = (hwid >> 4) & 3, (hwid >> 8) & 15, (hwid >> 12) & 1, (hwid >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = cu_key * 4 + simd
print(f"distinct XCCs {len(np.unique(xcc))}, CUs {len(np.unique(cu_key))}, SIMDs {len(np.unique(simd_key))}")
cyc = tot.astype(float)
busy = cyc - d[:, 0]   # cycles after the loads landed
def per(key, val):
    ks, inv = np.unique(key, return_inverse=True)
    return np.bincount(inv, weights=val), np.bincount(inv)
for name, key in (("SIMD", simd_key), ("CU", cu_key), ("XCC", xcc)):
    ssum, cnt = per(key, busy)
    last, _ = per(key, np.zeros(len(key)))
    ends = np.array([ext[key == k].max() for k in np.unique(key)])
    print(f"per {name}: waves min {cnt.min()} max {cnt.max()}; sum of post-load cycles mean {ssum.mean():.0f} max {ssum.max():.0f} (max/mean {ssum.max() / ssum.mean():.2f}); "
          f"last exit mean {ends.mean():.2f} p90 {np.percentile(ends, 90):.2f} max {ends.max():.2f} us; corr(sum, last exit) {np.corrcoef(ssum, ends)[0, 1]:.2f}")
print("slot -> (xcc, se, sh, cu, simd) of the first 24 slots:", [(int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i])) for i in range(24)])
# slots sharing a SIMD
ks = np.unique(simd_key)
ex = [sorted(slot[simd_key == k].tolist()) for k in ks[:6]]
print("slots on the first SIMDs:", ex)

# per-unit work counters (indexed by unit) for tools/diag/bwd_order.py
wk = np.zeros(8192 * 4, np.uint32)
L.fr_debug_read_bwd_work.argtypes = [C.c_void_p, C.c_size_t]
assert L.fr_debug_read_bwd_work(wk.ctypes.data, wk.nbytes) == 0
wk = wk.reshape(8192, 4)
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/bwd_work.npy", wk)
uw = w[:, 14]  # (class or pairs; not the unit) -- the unit index of a slot is not in the trace: use the counters by unit
n_units = int((wk[:, 2] > 0).sum())
print(f"work counters of {n_units} units: phase A iterations mean {wk[:n_units, 0].mean():.1f} max {wk[:n_units, 0].max()}, phase B {wk[:n_units, 1].mean():.1f} max {wk[:n_units, 1].max()}, ranges mean {wk[:n_units, 2].mean():.2f}")

du = np.zeros(8192)
du[w[:, 14]] = dur
np.save("gpurun_out/bwd_dur.npy", du)
pa = np.zeros(8192)
pa[w[:, 14]] = d[:, 3]
np.save("gpurun_out/bwd_phase_a.npy", pa)
