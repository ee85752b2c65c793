"""GPU box: replay iteration K of `tools/fuzz_parity.py N SEED [big]` and look at its forward outliers.
    python tools/diag/fuzz_replay.py K SEED [big]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes  # noqa: E402
from tests import util  # noqa: E402

K, seed, big = int(sys.argv[1]), int(sys.argv[2]), len(sys.argv) > 3
_, P, H, W, kw, dpix, _ = util.fuzz_case(seed, K, big)
print(P, H, W, kw)
s = scenes.random_scene(P, H, W, **kw)
o = util.oracle_forward(s)
h = util.HipFrame(s, torch.device("cuda:0"))
col, fT = h.color.cpu().numpy(), h.final_T.cpu().numpy()
n_out, unexpl = util.unexplained_outliers(o, col, fT)
print("instances", h.counts.num_instances, "max list", h.counts.max_tile_list, "outliers", n_out, "unexplained", unexpl[:8])
bad = (np.abs(col - o.color) > 1e-5 + 1e-4 * np.abs(o.color)).any(0) | (np.abs(fT - o.final_T) > 1e-5 + 1e-4 * np.abs(o.final_T))
nc = h.n_contrib.cpu().numpy()
for y, x in list(zip(*np.nonzero(bad)))[:12]:
    print(f"px ({x},{y}): margin {util.explain_pixel(o, x, y):.3f}  colour hip {col[:, y, x]} oracle {o.color[:, y, x]}  "
          f"T hip {fT[y, x]:.6e} oracle {o.final_T[y, x]:.6e}  n_contrib hip {nc[y, x]} oracle {o.n_contrib[y, x]}")
if os.environ.get("FUZZ_BWD"):
    from oracle import oracle
    ob, hb = oracle.backward(o, dpix), h.backward(dpix)
    k = os.environ["FUZZ_BWD"]
    ref, got = getattr(ob, k).reshape(P, -1), hb[k].reshape(P, -1)
    scale = np.abs(ref).max()
    err = np.abs(got - ref)
    bad = err > 1e-4 * np.abs(ref) + 1e-6 * scale
    print(k, "entries off:", int(bad.sum()), "of", bad.size, "scale", scale)
    # what the reference's own float order does to the same entries (the oracle with seeded summation orders / nvcc-style
    # contraction against its double sums: the floors of tests/test_gpu_parity.py::_check_backward)
    alts = []
    for sd, contract in ((1, False), (2, True), (3, False), (4, True)):
        oracle.set_bwd_float_order(sd); oracle.set_bwd_contract(contract)
        try:
            alts.append(getattr(oracle.backward(o, dpix), k).reshape(P, -1))
        finally:
            oracle.set_bwd_float_order(0); oracle.set_bwd_contract(False)
    for i in np.argwhere(bad)[:10]:
        i = tuple(i)
        print("  id", i, "ref", ref[i], "got", got[i], "err/|ref|", err[i] / abs(ref[i]), "err/scale", err[i] / scale,
              "radius", o.radii[i[0]], "opacity", o.conic_opacity[i[0], 3],
              "| oracle float orders err/|ref|:", [float(abs(a[i] - ref[i]) / abs(ref[i])) for a in alts])
    # which rows carry the L2 error (flip-affected rows are excluded from the test's rel_l2)
    from tests.test_gpu_parity import _flip_affected_gaussians
    skip, n_flips = _flip_affected_gaussians(o, h)
    e2 = ((got - ref) ** 2).sum(1)
    keep = ~skip
    print("flips", n_flips, "rows skipped", int(skip.sum()), "rel_l2 kept rows", util.rel_l2(got[keep], ref[keep]),
          "rel_l2 all rows", util.rel_l2(got, ref))
    order = np.argsort(-np.where(keep, e2, 0.0))[:10]
    tot = float(e2[keep].sum())
    for i in order:
        print(f"  row {i}: share of kept err^2 {e2[i] / tot:.3f}  ref {ref[i]}  got {got[i]}  radius {o.radii[i]}  "
              f"opacity {o.conic_opacity[i, 3]:.3f}  |ref|/scale {np.abs(ref[i]).max() / scale:.3g}")
