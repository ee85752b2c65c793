"""GPU box: one seeded fuzz configuration of tests/test_gpu_configs.py in detail — where do HIP and oracle gradients differ?"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes
from tests import util
from oracle import oracle

seed, big = int(sys.argv[1]), bool(int(sys.argv[2]))
rng = np.random.default_rng(seed)
P = int(rng.integers(1, 60000 if big else 6000))
H, W = int(rng.integers(8, 900 if big else 300)), int(rng.integers(8, 900 if big else 300))
deg = int(rng.integers(0, 4))
slo = float(10 ** rng.uniform(-3.5, -1.5)); shi = slo * float(rng.uniform(1, 20))
olo = float(rng.uniform(0.001, 0.5)); ohi = float(rng.uniform(olo, 1.0))
kw = dict(sh_degree=deg, seed=int(rng.integers(1 << 30)), spread=float(rng.uniform(0.05, 1.5)), scale_lo=slo,
          scale_hi=shi, opacity_lo=olo, opacity_hi=ohi, behind_fraction=float(rng.choice([0.0, 0.1])),
          M=int(rng.choice([(deg + 1) ** 2, 16])), bg=tuple(rng.uniform(0, 1, 3)))
print(P, H, W, kw)
s = scenes.random_scene(P, H, W, **kw)
o = util.oracle_forward(s)
h = util.HipFrame(s, torch.device("cuda:0"))
rng2 = np.random.default_rng(seed)
dpix = (rng2.uniform(-1, 1, (3, H, W)) / (H * W)).astype(np.float32)
ob = oracle.backward(o, dpix)
hb = h.backward(dpix)
col, fT = h.color.cpu().numpy(), h.final_T.cpu().numpy()
n_out, unexpl = util.unexplained_outliers(o, col, fT)
bad = (np.abs(col - o.color) > 1e-5 + 1e-4 * np.abs(o.color)).any(0)
ys, xs = np.nonzero(bad)
print("forward outlier pixels:", n_out, list(zip(xs.tolist(), ys.tolist()))[:20], "unexplained", unexpl[:5])
ncd = (h.n_contrib.cpu().numpy() != 0) != (o.n_contrib != 0)
print("n_contrib zero/nonzero mismatch px:", int(ncd.sum()))
for k in ["dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations"]:
    ref, got = getattr(ob, k), hb[k]
    d = np.abs(got - ref).reshape(P, -1).max(1)
    top = np.argsort(-d)[:6]
    print(k, "rel_l2", util.rel_l2(got, ref), "max|ref|", np.abs(ref).max())
    for i in top:
        print("   id", i, "diff", d[i], "ref", ref[i].ravel()[:4], "got", got[i].ravel()[:4], "xy", o.means2D[i], "r", o.radii[i],
              "conic_op", o.conic_opacity[i])
