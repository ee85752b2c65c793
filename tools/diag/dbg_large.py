import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes
from tests import util
s = scenes.random_scene(5000, 1000, 1536, sh_degree=1, seed=21, spread=0.28, scale_lo=0.004, scale_hi=0.03, opacity_lo=0.2, opacity_hi=0.9)
o = util.oracle_forward(s)
for rep in range(2):
    h = util.HipFrame(s, torch.device("cuda:0"))
    c = h.counts
    print("counts", c.num_rendered, c.num_instances, c.max_tile_list, c.overflow, c.capacity_required, "oracle num_rendered", o.num_rendered)
    col = h.color.cpu().numpy(); fT = h.final_T.cpu().numpy()
    bad = ~np.isclose(fT, o.final_T, rtol=1e-4, atol=1e-5)
    print("bad T pixels", bad.sum(), "of", bad.size)
    if bad.any():
        ys, xs = np.nonzero(bad)
        tiles = set(zip((ys // 8).tolist(), (xs // 8).tolist()))
        print("bad tiles", len(tiles), sorted(tiles)[:20])
        ty, tx = sorted(tiles)[0]
        print("regions of bad tiles", sorted({(tx + 3 * ty) & 7 for ty, tx in tiles}))
        print("sample", fT[ys[0], xs[0]], o.final_T[ys[0], xs[0]], "n_contrib", h.n_contrib.cpu().numpy()[ys[0], xs[0]] if hasattr(h, "n_contrib") else None)
