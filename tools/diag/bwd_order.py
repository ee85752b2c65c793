"""What can the ORDER of the blend backward's work list give (GPU box)?  Renders one frame, reads the list the forward blend
left (BinningView::bwd_units, 32 B per slot: descriptor, unit, pairs named, forward walk iterations), rewrites it under
several policies and times the backward launch for each (dispatch-tied stage events, mean of --iters launches).
The hardware puts slots s, s + 1024, s + 2048, s + 3072 on one SIMD and starts the waves of the last ~10 % of a
3 900-slot list about 2.5 us after the first."""
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
ap.add_argument("--iters", type=int, default=150)
ap.add_argument("--dur", default=None, help="npy of per-unit durations from a trace run (oracle orders)")
a = ap.parse_args()
dev = torch.device("cuda:0")
s = scenes.head_scene(P=a.P, res=a.res, opacity=a.opacity, scale=a.scale)
f = HipFrame(s, dev)
g = (np.random.default_rng(0).uniform(-1, 1, (3, a.res, a.res)) / (a.res * a.res)).astype(np.float32)
torch.cuda.synchronize()
img_off = (-f.img.data_ptr()) & 255
counts = f.img[img_off:img_off + 64].view(torch.int32).cpu().numpy()
nu = int(counts[14])
off = (-f.binning.data_ptr()) & 255
lst = f.binning[off:off + 32 * nu]
base = lst.view(torch.int32).view(nu, 8).cpu().numpy().copy()
pairs, iters, unit = base[:, 5].astype(np.int64), base[:, 6].astype(np.int64), base[:, 4].astype(np.int64)
assert sorted(unit.tolist()) == list(range(nu)), "the list is not a permutation of the units"
print(f"units {nu}; pairs p25 {np.percentile(pairs, 25):.0f} p50 {np.percentile(pairs, 50):.0f} p75 {np.percentile(pairs, 75):.0f} p90 {np.percentile(pairs, 90):.0f}; "
      f"multi-range (> 640) {100.0 * (pairs > 640).mean():.1f} %")


def run(order, label):
    lst.copy_(torch.from_numpy(base[order].copy()).view(torch.uint8).view(-1).to(dev))
    for _ in range(5):
        f.backward(g)
    torch.cuda.synchronize()
    _lib.profile_enable(0, True)
    for _ in range(a.iters):
        f.backward(g)
    torch.cuda.synchronize()
    t = _lib.profile_read(0)
    _lib.profile_enable(0, False)
    v = t["blend_bwd"]
    print(f"  {label:58s} {v[0] * 1e3 / max(v[1], 1):6.2f} us")


rng = np.random.default_rng(1)


def affine(order):
    """The same priority order, but every unit stays on the XCD its forward ran on (workgroup = 4 slots, XCD = workgroup
    % 8): the units of XCD x, in priority order, take the slots of XCD x in slot order."""
    out = np.empty(nu, np.int64)
    xs = (np.arange(nu) // 4) % 8
    xu = (unit[order] // 4) % 8
    for x in range(8):
        sl = np.nonzero(xs == x)[0]
        us = order[xu == x]
        n = min(len(sl), len(us))
        out[sl[:n]] = us[:n]
        assert len(sl) == len(us), (x, len(sl), len(us))
    return out


nat = np.argsort(unit, kind="stable")                       # slot = unit
cost = pairs + 12 * iters                                   # a guess at the work: pairs and the length of the longest walk
by_pairs = np.argsort(-pairs, kind="stable")
by_cost = np.argsort(-cost, kind="stable")
run(np.arange(nu), "as the forward left it (641+ pairs first, rest reversed)")
run(nat, "unit order")
run(nat[::-1], "reversed unit order")
run(rng.permutation(nu), "random")
run(by_pairs, "most pairs first")
run(by_pairs[::-1], "fewest pairs first")
run(by_cost, "pairs + 12 x forward iterations, descending")
run(np.argsort(-iters, kind="stable"), "most forward iterations first")
# heavy first, the lightest in the late-starting tail, the middle as it comes
hv = by_pairs[: int(0.17 * nu)]
lt = by_pairs[int(0.85 * nu):]
md = rng.permutation(by_pairs[int(0.17 * nu): int(0.85 * nu)])
run(np.concatenate([hv, md, lt]), "17 % heaviest, random middle, 15 % lightest last")
run(np.concatenate([rng.permutation(hv), md, rng.permutation(lt)]), "the same, random inside the classes")


def snake(o):   # descending list dealt so that the four slots of a SIMD (s + 1024 k) hold one unit of every quarter, the
    q = [o[k::4] for k in range(4)]   # heaviest next to the lightest
    out = np.full(max(nu, 4096), -1, np.int64)
    n = 1024
    parts = [q[0], q[3][::-1], q[1], q[2][::-1]]
    res = np.concatenate([p_ for p_ in parts])
    return res[:nu]


run(snake(by_pairs), "quarters by pairs: q0, q3 reversed, q1, q2 reversed")
if a.dur:
    du = np.load(a.dur)[unit]
    run(np.argsort(-du, kind="stable"), "longest measured duration first (oracle)")
    run(np.argsort(du, kind="stable"), "shortest measured duration first")
    o = np.argsort(-du, kind="stable")
    run(np.concatenate([o[: int(0.22 * nu)], o[int(0.22 * nu):]]), "the same (check)")
    hv2 = np.nonzero(pairs >= 641)[0]
    lt2 = np.nonzero(pairs < 641)[0]
    run(np.concatenate([hv2[np.argsort(-du[hv2], kind="stable")], lt2[np.argsort(-du[lt2], kind="stable")]]), "641+ pairs first, each class by measured duration descending")
if os.path.exists("gpurun_out/bwd_work.npy"):   # per-unit iteration counts from a trace run of the same scene (tools/diag/bwd_trace.py)
    wk = np.load("gpurun_out/bwd_work.npy").astype(np.int64)
    for wa, wb, wr, label in ((88, 45, 600, "88 A + 45 B + 600 per range"), (88, 0, 0, "phase A iterations only"), (60, 60, 1500, "60 A + 60 B + 1500 per range")):
        work = (wa * wk[:, 0] + wb * wk[:, 1] + wr * wk[:, 2] + 800)[unit]     # per list entry
        o = np.argsort(-work, kind="stable")
        run(o, f"work = {label}: descending")
        # LPT into 1024 SIMD bins (bin b owns slots b, b + 1024, b + 2048, b + 3072 where they exist), heaviest bins'
        # first slots first
        nb = 1024
        cap = np.array([len(range(b, nu, nb)) for b in range(nb)])
        load = np.zeros(nb)
        fill = np.zeros(nb, np.int64)
        out = np.full(nu, -1, np.int64)
        import heapq
        heap = [(0.0, b) for b in range(nb)]
        heapq.heapify(heap)
        for e in o:
            while True:
                l, b = heapq.heappop(heap)
                if fill[b] < cap[b]:
                    break
            out[b + nb * fill[b]] = e
            fill[b] += 1
            load[b] += work[e]
            if fill[b] < cap[b]:
                heapq.heappush(heap, (load[b], b))
        assert (out >= 0).all()
        run(out, f"work = {label}: LPT over the SIMDs' four slots (max/mean {load.max() / load.mean():.3f})")
print("the same policies, every unit kept on the XCD of its forward:")
run(affine(rng.permutation(nu)), "random")
run(affine(by_pairs), "most pairs first")
run(affine(by_pairs[::-1]), "fewest pairs first")
run(affine(by_cost), "pairs + 12 x forward iterations, descending")
run(affine(np.argsort(-iters, kind="stable")), "most forward iterations first")
run(affine(np.concatenate([hv, md, lt])), "17 % heaviest, random middle, 15 % lightest last")
run(affine(np.concatenate([by_pairs[: int(0.22 * nu)], rng.permutation(by_pairs[int(0.22 * nu):])])), "22 % heaviest first, rest random")
run(affine(np.concatenate([by_pairs[: int(0.22 * nu)], by_pairs[int(0.22 * nu):][::-1]])), "22 % heaviest first, rest ascending")
run(affine(snake(by_pairs)), "quarters by pairs: q0, q3 reversed, q1, q2 reversed")
run(affine(snake(by_cost)), "quarters by cost: q0, q3 reversed, q1, q2 reversed")
# variants of the forward's own order (position inside a stripe = arrival rank at the stripe's cursor: heavy units in
# arrival order from the front, the others in reverse arrival order behind them)
v1 = np.arange(nu)
v2 = np.arange(nu)
for j in range(64):
    sl = np.arange(j, nu, 64)
    hv_j = sl[pairs[sl] >= 641]
    lt_j = sl[pairs[sl] < 641]
    v1[sl] = np.concatenate([hv_j[::-1], lt_j])            # heavy: last arrival first
    v2[sl] = np.concatenate([hv_j, lt_j[::-1]])            # light: first arrival first
run(v1, "forward's order, heavy class reversed (late arrivals first)")
run(v2, "forward's order, light class in arrival order")
v3 = np.arange(nu)
for j in range(64):
    sl = np.arange(j, nu, 64)
    hv_j = sl[pairs[sl] >= 641]
    lt_j = sl[pairs[sl] < 641]
    arr = np.concatenate([hv_j, lt_j[::-1]])   # arrival order is lost between the classes: interleave is unknown; classes kept
    v3[sl] = np.concatenate([lt_j, hv_j])      # light class first
run(v3, "forward's order, light class in front of the heavy class")
# classes by pairs named, each class in unit (= spatial) order, heaviest class first
def classes(thr, dirs):
    parts = []
    edges = [1 << 30] + list(thr) + [-1]
    for k in range(len(edges) - 1):
        sel = np.nonzero((pairs < edges[k]) & (pairs >= edges[k + 1]))[0]
        sel = sel[np.argsort(unit[sel], kind="stable")]
        parts.append(sel if dirs[k % len(dirs)] > 0 else sel[::-1])
    return np.concatenate(parts)
run(classes([641], [1, -1]), "classes 641+ | rest, unit order up / down")
run(classes([641], [1, 1]), "classes 641+ | rest, unit order up / up")
run(classes([641], [-1, 1]), "classes 641+ | rest, unit order down / up")
run(classes([641], [-1, -1]), "classes 641+ | rest, unit order down / down")
run(classes([641, 480], [1, -1, 1]), "classes 641+ | 480+ | rest, up / down / up")
run(classes([641, 480], [1, -1, -1]), "classes 641+ | 480+ | rest, up / down / down")
run(classes([641, 480, 320], [1, -1, 1, -1]), "classes 641+ | 480+ | 320+ | rest, alternating")
run(classes([641, 480, 320], [1, -1, -1, -1]), "classes 641+ | 480+ | 320+ | rest, up then down")
run(classes([800, 641, 480, 320], [1, -1]), "classes 800+ | 641+ | 480+ | 320+ | rest, alternating")
run(classes([641, 560, 480, 400, 320, 240], [1, -1]), "seven classes, alternating")
# which property of the forward's order matters?
p2 = np.arange(nu)
for j in range(64):
    sl = np.arange(j, nu, 64)
    h = pairs[sl] >= 641
    p2[sl] = np.concatenate([rng.permutation(sl[h]), rng.permutation(sl[~h])])
run(p2, "forward's order, shuffled inside each stripe's classes (arrival order lost)")
wg = np.arange(nu // 4 * 4).reshape(-1, 4)
nh = int((pairs >= 641).sum()) // 4
p1 = np.concatenate([wg[rng.permutation(nh)].ravel(), wg[nh + rng.permutation(len(wg) - nh)].ravel(), np.arange(nu // 4 * 4, nu)])
run(p1, "forward's order, workgroups (4 slots) shuffled inside the class regions")
p3 = np.arange(nu)
k = nu // 64 * 64
p3[:k] = (np.arange(k).reshape(-1, 64)[:, (np.arange(64) + 4) % 64]).ravel()
run(p3, "forward's order, stripes rotated by 4 (next XCD, same rows)")
p4 = np.arange(nu)
p4[:k] = (np.arange(k).reshape(-1, 64)[:, (np.arange(64) + 32) % 64]).ravel()
run(p4, "forward's order, stripes rotated by 32 (same XCD, other workgroup)")
p5 = np.arange(nu)
p5[:k] = (np.arange(k).reshape(-1, 64)[:, rng.permutation(64)]).ravel()
run(p5, "forward's order, stripes permuted (rows kept)")
run(np.arange(nu), "as the forward left it (again)")
