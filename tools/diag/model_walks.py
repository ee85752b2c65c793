"""CPU model of the blend backward's walks (no GPU): how many trips phase A (lane = pixel, over its records) and phase B
(lane = record, over its pixels) need per 64-record unit of an 8x8 tile, and what other assignments of lanes would need.
    python tools/diag/model_walks.py [--P 100000] [--res 512] [--opacity 0.1]
The pairs are the (pixel, record) pairs with opacity * G >= 1/255 and power <= 0, from the ORACLE's per-Gaussian state (the
kernels walk a superset that is a few per cent larger: footprint_mask); termination is ignored (config 2: no pixel ends early).
Validate against profiles/r05_b_bwd_stats.txt (the instrumented kernel): units, pairs per unit, phase-A and phase-B trips."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fateavatar_amd import scenes  # noqa: E402
from tests import util  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=100000)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--opacity", type=float, default=0.1)
a = ap.parse_args()

s = scenes.head_scene(P=a.P, res=a.res, sh_degree=3, seed=0, opacity=a.opacity)
o = util.oracle_forward(s)
H = W = a.res
vis = np.nonzero(o.radii > 0)[0]
xy, co, dep, rad = o.means2D[vis].astype(np.float64), o.conic_opacity[vis].astype(np.float64), o.depths[vis], o.radii[vis]

# candidate pixels of every Gaussian: its radius square (the reference's rectangle), clipped to the image
R = int(rad.max())
off = np.arange(-R - 1, R + 2)
oy, ox = np.meshgrid(off, off, indexing="ij")
cx, cy = np.round(xy[:, 0]).astype(np.int64), np.round(xy[:, 1]).astype(np.int64)
px = (cx[:, None] + ox.ravel()[None, :]).ravel()
py = (cy[:, None] + oy.ravel()[None, :]).ravel()
g = np.repeat(np.arange(len(vis)), off.size ** 2)
ok = (px >= 0) & (px < W) & (py >= 0) & (py < H)
px, py, g = px[ok], py[ok], g[ok]
dx, dy = xy[g, 0] - px, xy[g, 1] - py
power = -0.5 * (co[g, 0] * dx * dx + co[g, 2] * dy * dy) - co[g, 1] * dx * dy
alpha = co[g, 3] * np.exp(np.minimum(power, 0.0))
keep = (power <= 0) & (alpha >= 1.0 / 255.0)
px, py, g = px[keep], py[keep], g[keep]
print(f"pairs {len(g)}  ({len(g) / len(vis):.1f} pixels per visible Gaussian)")

# instances: (8x8 tile, Gaussian) with at least one pair; per tile sorted by (depth, id); units of 64
tile = (py // 8) * (W // 8) + (px // 8)
inst_key = tile.astype(np.int64) * len(vis) + g
uk, inv = np.unique(inst_key, return_inverse=True)
it, ig = uk // len(vis), uk % len(vis)
order = np.lexsort((vis[ig], dep[ig], it))          # tile, then depth, then id
rank_in_tile = np.empty(len(uk), np.int64)
it_s = it[order]
starts = np.r_[0, np.nonzero(np.diff(it_s))[0] + 1]
pos = np.arange(len(uk)) - np.repeat(starts, np.diff(np.r_[starts, len(uk)]))
rank_in_tile[order] = pos
unit_of_inst = rank_in_tile // 64
rec_in_unit = rank_in_tile % 64
n_tiles_used = len(starts)
tile_len = np.diff(np.r_[starts, len(uk)])
units_per_tile = (tile_len + 63) // 64
print(f"instances {len(uk)}  tiles {n_tiles_used}  units {units_per_tile.sum()}  (tiles with one unit: {(units_per_tile == 1).mean():.1%})")

# unit id of every pair, its pixel within the tile, its record within the unit
tile_index = {t: i for i, t in enumerate(it_s[starts])}
tile_base = np.r_[0, np.cumsum(units_per_tile)][:-1]
tidx = np.searchsorted(it_s[starts], it)             # tile slot of each instance
unit_global = tile_base[tidx] + unit_of_inst          # per instance
pu = unit_global[inv]                                 # per pair
pp = ((py % 8) * 8 + (px % 8))[...]                   # pixel within tile
pr = rec_in_unit[inv]
U = int(units_per_tile.sum())
N = np.zeros((U, 64, 64), bool)                       # [unit][pixel][record]
N[pu, pp, pr] = True
per_pixel = N.sum(2)                                   # [U,64] records per pixel
per_rec = N.sum(1)                                     # [U,64] pixels per record
pairs = per_pixel.sum(1)


def trips(x, per_trip=2):
    return np.ceil(x / per_trip)


A_now, B_now = trips(per_pixel.max(1)), trips(per_rec.max(1))
bal = trips(np.ceil(pairs / 64.0))
print(f"pairs per unit {pairs.mean():.0f}; phase A trips {A_now.mean():.2f} (balanced {bal.mean():.2f}); phase B trips {B_now.mean():.2f}; "
      f"lane utilisation A {pairs.sum() / (A_now.sum() * 128):.1%}  B {pairs.sum() / (B_now.sum() * 128):.1%}")
act_px = (per_pixel > 0).sum(1)
print(f"pixels with any pair per unit: mean {act_px.mean():.1f}; units with <= 32 active pixels: {(act_px <= 32).mean():.1%}, <= 16: {(act_px <= 16).mean():.1%}")

# (1) two lanes per pixel where <= 32 pixels are active: the alpha evaluations of a pixel are split over two lanes (the serial
# recurrence is not): model the expensive half (30 of 42 instructions per pair) at half the pixel's count, the rest at the full count
two = np.where(act_px <= 32, 0.7 * trips(np.ceil(per_pixel.max(1) / 2.0)) + 0.3 * A_now, A_now)
print(f"(1) two lanes per pixel on units with <= 32 active pixels: phase-A cost {two.sum() / A_now.sum():.3f} of now")
# (2) four lanes per pixel where <= 16 are active
four = np.where(act_px <= 16, 0.7 * trips(np.ceil(per_pixel.max(1) / 4.0)) + 0.3 * A_now, two)
print(f"(2) ... and four lanes per pixel on units with <= 16: {four.sum() / A_now.sum():.3f}")
# (3) phase B with two lanes per record where <= 32 records have pairs
act_rec = (per_rec > 0).sum(1)
twoB = np.where(act_rec <= 32, trips(np.ceil(per_rec.max(1) / 2.0)), B_now)
print(f"(3) records with any pair per unit: mean {act_rec.mean():.1f}; two lanes per record where <= 32 are active: phase-B cost {twoB.sum() / B_now.sum():.3f} of now")
# (4) the busiest pixel's share: how much of phase A is decided by ONE pixel
srt = np.sort(per_pixel, 1)
print(f"(4) per unit: busiest pixel {srt[:, -1].mean():.1f} records, second {srt[:, -2].mean():.1f}, 8th {srt[:, -8].mean():.1f}, median {srt[:, 32].mean():.1f}, mean {per_pixel.mean():.1f}")
# (5) units in pairs of the same tile (128 records per wave) where a tile has several
print(f"(5) tiles with >= 2 units: {(units_per_tile >= 2).mean():.1%} of the tiles, {units_per_tile[units_per_tile >= 2].sum() / U:.1%} of the units")
# (5b) what merging consecutive units of a tile would do to phase A (lane = pixel: trips = the busiest pixel's records / 2)
first_unit = tile_base                       # per tile slot
now_total = A_now.sum()
pair2 = whole = 0.0
for t in range(n_tiles_used):
    u0, k = int(first_unit[t]), int(units_per_tile[t])
    pp_t = per_pixel[u0:u0 + k]              # [k,64]
    whole += np.ceil(pp_t.sum(0).max() / 2.0)
    for j in range(0, k, 2):
        pair2 += np.ceil(pp_t[j:j + 2].sum(0).max() / 2.0)
print(f"(5b) phase-A trips with 128-record units (pairs of consecutive units): {pair2 / now_total:.3f} of now; with a tile's whole list in one wave: {whole / now_total:.3f}")
# the same for phase B does not apply (lane = record: more records need more lanes)
# (6) a wide tail: once <= K pixels still have records left, the wave's 64 lanes are dealt out floor(64 / active) per pixel, each
# lane evaluating ONE record's alpha, the pixel's T / accum recurrence done as a prefix product over its lanes.  A wide step is
# charged `wide_cost` normal trips (two records per lane now; a wide step adds a gather of the pixel's state and a DPP scan).
for K, wide_cost in ((32, 1.0), (32, 1.5), (16, 1.5), (8, 1.5)):
    total = 0.0
    for u in range(U):
        left = np.sort(per_pixel[u])[::-1].astype(np.int64).copy()
        cost = 0.0
        while left[0] > 0:
            active = int((left > 0).sum())
            if active > K:
                left = np.maximum(left - 2, 0)
                cost += 1.0
            else:
                m = 64 // active
                left = np.maximum(left - m, 0)
                cost += wide_cost
        total += cost
    print(f"(6) wide tail once <= {K} pixels are left, a wide step charged {wide_cost} trips: phase-A cost {total / now_total:.3f} of now")

# (7) the SIMPLE wide tail (round 6): once few pixels are left, lane = record finishes ONE pixel's remaining records per step (every
# lane evaluates its own record's alpha for that pixel, the T / accum recurrences as prefix products over the lanes): a step costs
# `c` normal trips per pixel whatever it has left; taken when active * c < ceil(deepest / 2)
for c in (0.6, 0.8, 1.0):
    total, costs = 0.0, []
    for u in range(U):
        left = np.sort(per_pixel[u])[::-1].astype(np.int64).copy()
        cost = 0.0
        while left[0] > 0:
            active = int((left > 0).sum())
            if active * c < np.ceil(left[0] / 2.0):
                cost += active * c
                break
            left = np.maximum(left - 2, 0)
            cost += 1.0
        total += cost
        costs.append(cost)
    costs = np.asarray(costs)
    print(f"(7) record-wide finish at {c} trips per pixel: phase-A cost {total / now_total:.3f} of now; deepest unit {costs.max():.1f} trips "
          f"(now {A_now.max():.1f}), p99 {np.percentile(costs, 99):.1f} ({np.percentile(A_now, 99):.1f}), p90 {np.percentile(costs, 90):.1f} ({np.percentile(A_now, 90):.1f})")
