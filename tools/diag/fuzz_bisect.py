"""GPU box: replay iteration K of `tools/fuzz_parity.py N SEED [big]` with the detected flip pixels masked out of dL/dpixel
(the test's no-exemption mode) and find WHICH PIXELS carry the remaining error of one gradient array's worst row — the
gradients are linear in dL/dpixel, so the error of a pixel set is the error of the run with dL/dpixel zeroed outside it.
    FUZZ_BWD=dL_dcov3D python tools/diag/fuzz_bisect.py K SEED [big]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import util  # noqa: E402
from tests.test_gpu_parity import _flip_pixels  # noqa: E402

K, seed, big = int(sys.argv[1]), int(sys.argv[2]), len(sys.argv) > 3
key = os.environ.get("FUZZ_BWD", "dL_dcov3D")
_, P, H, W, kw, dpix, _ = util.fuzz_case(seed, K, big)
print(P, H, W, kw)
s = scenes.random_scene(P, H, W, **kw)
o = util.oracle_forward(s)
h = util.HipFrame(s, torch.device("cuda:0"))
bad = _flip_pixels(o, h)
print("detected flip pixels:", [(int(x), int(y)) for y, x in zip(*np.nonzero(bad))])
dpix[:, bad] = 0.0


def err_rows(d):
    ref, got = getattr(oracle.backward(o, d), key).reshape(P, -1), h.backward(d)[key].reshape(P, -1)
    return ref, got


ref, got = err_rows(dpix)
e2 = ((got - ref).astype(np.float64) ** 2).sum(1)
print(key, "rel_l2 all rows", util.rel_l2(got, ref))
row = int(np.argmax(e2))
print("worst row", row, "share", e2[row] / e2.sum(), "ref", ref[row], "got", got[row], "radius", o.radii[row],
      "mean2D", o.means2D[row], "conic_opacity", o.conic_opacity[row])
for k2 in ("dL_dmeans2D", "dL_dopacity", "dL_dcolors", "dL_dcov3D", "dL_dmeans3D"):
    r2, g2 = getattr(oracle.backward(o, dpix), k2).reshape(P, -1)[row], h.backward(dpix)[k2].reshape(P, -1)[row]
    print(f"  {k2}: ref {r2} got {g2}")

# bisect the row's error over pixel rectangles
y0, y1, x0, x1 = 0, H, 0, W
tot = float(np.sqrt(e2[row]))
while (y1 - y0) * (x1 - x0) > 1:
    if (y1 - y0) >= (x1 - x0):
        ym = (y0 + y1) // 2; halves = [(y0, ym, x0, x1), (ym, y1, x0, x1)]
    else:
        xm = (x0 + x1) // 2; halves = [(y0, y1, x0, xm), (y0, y1, xm, x1)]
    errs = []
    for (a, b, c, d_) in halves:
        dm = np.zeros_like(dpix); dm[:, a:b, c:d_] = dpix[:, a:b, c:d_]
        r, g = err_rows(dm)
        errs.append(float(np.linalg.norm((g[row] - r[row]).astype(np.float64))))
    k = int(np.argmax(errs))
    print(f"  rect y[{y0},{y1}) x[{x0},{x1}): halves' error {errs[0]:.3e} {errs[1]:.3e} (row total {tot:.3e})")
    y0, y1, x0, x1 = halves[k]
print("pixel", (x0, y0), "margin", util.explain_pixel(o, x0, y0), "colour hip", h.color.cpu().numpy()[:, y0, x0], "oracle", o.color[:, y0, x0],
      "T hip", float(h.final_T.cpu().numpy()[y0, x0]), "oracle", float(o.final_T[y0, x0]),
      "n_contrib hip", int(h.n_contrib.cpu().numpy()[y0, x0]), "oracle", int(o.n_contrib[y0, x0]))
