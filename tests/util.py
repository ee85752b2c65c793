"""Shared helpers of the parity tests: run a scene through the CPU oracle and through the HIP path
(via the `_C`-level mirror, i.e. through the C ABI) and compare."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from fateavatar_amd import _lib, rasterizer
from fateavatar_amd.scenes import GaussianScene
from oracle import oracle


def oracle_forward(s: GaussianScene, colors_precomp=None, cov3D_precomp=None, scale_modifier=1.0):
    c = s.camera
    return oracle.forward(
        bg=s.bg, means3D=s.means3D, opacities=s.opacities, viewmatrix=c.world_view_transform,
        projmatrix=c.full_proj_transform, campos=c.camera_center, tanfovx=c.tanfovx, tanfovy=c.tanfovy,
        H=c.image_height, W=c.image_width, shs=None if colors_precomp is not None else s.shs,
        sh_degree=s.sh_degree, colors_precomp=colors_precomp, scales=None if cov3D_precomp is not None else s.scales,
        rotations=None if cov3D_precomp is not None else s.rotations, cov3D_precomp=cov3D_precomp,
        scale_modifier=scale_modifier)


class HipFrame:
    """One forward (+ optional backward) through the C ABI."""

    def __init__(self, s: GaussianScene, dev, colors_precomp=None, cov3D_precomp=None, scale_modifier=1.0, debug=False):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        e = torch.empty(0)
        c = s.camera
        self.s, self.dev = s, dev
        self.bg, self.means3D, self.op = t(s.bg), t(s.means3D), t(s.opacities)
        self.colors = t(colors_precomp) if colors_precomp is not None else e
        self.sh = e if colors_precomp is not None else t(s.shs)
        self.cov = t(cov3D_precomp) if cov3D_precomp is not None else e
        self.scales = e if cov3D_precomp is not None else t(s.scales)
        self.rots = e if cov3D_precomp is not None else t(s.rotations)
        self.view, self.proj, self.campos = t(c.world_view_transform), t(c.full_proj_transform), t(c.camera_center)
        self.mod, self.debug = scale_modifier, debug
        self.H, self.W = c.image_height, c.image_width
        (self.num_rendered, self.color, self.radii, self.geom, self.binning, self.img) = rasterizer.rasterize_gaussians(
            self.bg, self.means3D, self.colors, self.op, self.scales, self.rots, scale_modifier, self.cov, self.view,
            self.proj, c.tanfovx, c.tanfovy, self.H, self.W, self.sh, s.sh_degree, self.campos, False, debug)
        self.counts = rasterizer.last_counts[dev.index or 0]
        self.final_T, self.n_contrib = rasterizer.image_aux(self.img, self.H, self.W)

    def geometry(self, field: int, cols: int, dtype=torch.float32):
        """A per-Gaussian array of the forward state.  Fields 0 (pixel-space centre), 1 (depth) and 3 (colour) live inside the
        blend-record template (field 8, GeomView::rec_tmpl)."""
        if field == 0:
            return np.ascontiguousarray(self.geometry(8, 12)[:, 0:2])
        if field == 1:
            return np.ascontiguousarray(self.geometry(8, 12)[:, 10:11])
        if field == 3:
            t = self.geometry(8, 12)
            return np.ascontiguousarray(np.concatenate([t[:, 6:9], np.zeros((t.shape[0], 1), np.float32)], axis=1))
        L = _lib.lib()
        P = self.means3D.shape[0]
        base = self.geom.data_ptr()
        off = L.fr_debug_geometry_field(base, P, field) - base
        nbytes = P * cols * torch.empty(0, dtype=dtype).element_size()
        return self.geom[off:off + nbytes].view(dtype).view(P, cols).cpu().numpy()

    def backward(self, dL_dpix: np.ndarray):
        c = self.s.camera
        g = torch.from_numpy(np.ascontiguousarray(dL_dpix)).to(self.dev)
        out = rasterizer.rasterize_gaussians_backward(
            self.bg, self.means3D, self.radii, self.colors, self.scales, self.rots, self.mod, self.cov, self.view,
            self.proj, c.tanfovx, c.tanfovy, g, self.sh, self.s.sh_degree, self.campos, self.geom, self.num_rendered,
            self.binning, self.img, self.debug)
        names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
                 "dL_drotations"]
        return {k: v.cpu().numpy() for k, v in zip(names, out)}


def frac_close(a, b, rtol, atol):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.size == 0:
        return 1.0
    return float(np.mean(np.abs(a - b) <= atol + rtol * np.abs(b)))


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    d = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / d) if d > 0 else float(np.linalg.norm(a - b))


def explain_pixel(o, x: int, y: int):
    """Why may the pixel (x, y) legitimately differ between two fp32 implementations of the reference's blend loop
    (forward.cu:330-361)?  Walks the pixel's 16x16-tile list in the ORACLE and returns the smallest normalised margin by
    which any of the three discrete tests was decided:
        power > 0            margin |power| / (1e-5 + 4e-7 m)
        alpha < 1/255        margin |alpha - 1/255| / (1/255) / (2e-5 + 4e-7 m)
        T (1 - alpha) < 1e-4 margin |T (1 - alpha) - 1e-4| / 1e-4 / (2e-5 + 2.4e-7 n)
    (m = |a| dx^2 / 2 + |c| dy^2 / 2 + |b dx dy|, the magnitude of the terms `power` is the sum of: an elongated splat far
    from its centre cancels terms of size 50 to a power of -5, and an absolute error of m ulps in the power is a relative
    error of m ulps in alpha; n = entries blended so far: T is a product of n factors, and the two implementations
    associate it differently).
    A value <= 1 means some splat sits on a threshold within rounding: the pixel is a threshold flip."""
    i = o._inputs
    W = i["W"]
    gx = (W + 15) // 16
    t = (y // 16) * gx + (x // 16)
    lo, hi = int(o.ranges[t, 0]), int(o.ranges[t, 1])
    ids = o.point_list[lo:hi].astype(np.int64)
    if ids.size == 0:
        return np.inf
    f32 = np.float32
    xy = o.means2D[ids]
    co = o.conic_opacity[ids]
    dx = (xy[:, 0] - f32(x)).astype(f32)
    dy = (xy[:, 1] - f32(y)).astype(f32)
    power = (f32(-0.5) * (co[:, 0] * dx * dx + co[:, 2] * dy * dy) - co[:, 1] * dx * dy).astype(f32)
    alpha = np.minimum(f32(0.99), co[:, 3] * np.exp(np.minimum(power, f32(0)))).astype(f32)
    mag = (0.5 * (np.abs(co[:, 0]) * dx * dx + np.abs(co[:, 2]) * dy * dy) + np.abs(co[:, 1] * dx * dy)).astype(np.float64)
    best = float(np.min(np.abs(power.astype(np.float64)) / (1e-5 + 4e-7 * mag)))
    T = 1.0
    n = 0
    for k in range(ids.size):
        if power[k] > 0:
            continue
        a = float(alpha[k])
        best = min(best, abs(a - 1.0 / 255.0) * 255.0 / (2e-5 + 4e-7 * float(mag[k])))
        if a < 1.0 / 255.0:
            continue
        tt = T * (1.0 - a)
        best = min(best, abs(tt - 1e-4) / 1e-4 / (2e-5 + 2.4e-7 * n))
        if tt < 1e-4:
            break
        T = tt
        n += 1
    return best


def flip_pixels(o, got_color, got_T, rtol=1e-4, atol=1e-5):
    """Pixels whose blend sequence took a different discrete decision than the oracle's: the image or the final transmittance
    outside the forward tolerance — or the final transmittance off by more than 0.1 % RELATIVE.  Every one of the reference's
    three tests (power > 0, alpha < 1/255, T (1 - alpha) < 1e-4) decides whether an entry of alpha >= 1/255 is blended, so a flip
    moves T_final by a factor (1 - alpha) <= 1 - 1/255, i.e. by >= 0.39 %, whereas two fp32 evaluations of the same sequence
    differ by ~1e-7 per factor (<= 2e-4 after 2 600 entries).  The relative test finds the flips the tolerance cannot see: a
    pixel that ends near T = 1e-4 moves its colour by < 1e-5 when it blends ninety entries more or fewer, but the entries it
    does or does not reach still get gradient from it (fuzz case 918 of seed 9002: one such pixel, 0.2 % of dL/dcov3D of a
    splat 637 pixels in radius)."""
    bad = (np.abs(got_color - o.color) > atol + rtol * np.abs(o.color)).any(0) | \
          (np.abs(got_T - o.final_T) > atol + rtol * np.abs(o.final_T))
    return bad | (np.abs(got_T - o.final_T) > 1e-3 * np.abs(o.final_T))


def unexplained_outliers(o, got_color, got_T, rtol=1e-4, atol=1e-5, limit=2000):
    """Flip pixels (flip_pixels) where NO splat of the pixel's list sits on one of the reference's thresholds
    (explain_pixel > 1).  Returns (n_flip_pixels, [(x, y, margin), ...])."""
    bad = flip_pixels(o, got_color, got_T, rtol, atol)
    ys, xs = np.nonzero(bad)
    out = []
    for x, y in list(zip(xs.tolist(), ys.tolist()))[:limit]:
        m = explain_pixel(o, x, y)
        if not (m <= 1.0):
            out.append((x, y, m))
    return int(bad.sum()), out


def assert_same_trajectory(a, b, what="", tight=5e-3, lr_max=0.05):
    """Two optimisation runs that differ only in the summation order of float atomics (eager against graph replay, lanes
    against a launch chain ...).  Adam turns a gradient that is pure rounding noise into a full step of either sign
    (m / sqrt(v) is +-1 whatever the size), so a handful of parameters whose true gradient is zero may end up a learning
    rate apart; everything else must agree closely.  Held to: rel-L2 <= 1e-3 of the whole parameter vector, 99.99 % of
    the entries within `tight` — and never more than 64 entries outside it, however long the vector (a regression confined
    to one tile list must not hide in the fraction) — and no entry further apart than two steps of the largest learning
    rate."""
    import torch
    d = (a - b).abs()
    rl = float(torch.linalg.norm((a - b).double()) / torch.linalg.norm(b.double()))
    frac = float((d < tight).double().mean())
    n_out = int((d >= tight).sum())
    assert rl <= 1e-3 and frac >= 0.9999 and n_out <= 64 and float(d.max()) <= 2 * lr_max, (what, rl, frac, n_out, float(d.max()))


def fuzz_stream(seed, big=False):
    """The configurations of `tools/fuzz_parity.py N SEED [big]`, in order: yields (k, P, H, W, kw, dL_dpixel, name) with
    `scenes.random_scene(P, H, W, **kw)` the scene of iteration k.  One stream of draws, so that a campaign's case k can be
    replayed (tools/diag/fuzz_replay.py, fuzz_bisect.py) and pinned as a regression test (tests/test_gpu_configs.py)."""
    rng = np.random.default_rng(seed)
    k = 0
    huge = big == "huge"   # config-5-sized scenes: lists beyond 1 024 entries (k_tile_sort_big), regrown buffers; small splats only
    while True:
        if huge:
            P = int(rng.integers(60000, 400000))
            H, W = int(rng.integers(512, 1400)), int(rng.integers(512, 1400))
        else:
            P = int(rng.integers(1, 60000 if big else 6000))
            H, W = int(rng.integers(8, 900 if big else 300)), int(rng.integers(8, 900 if big else 300))
        deg = int(rng.integers(0, 4))
        slo = float(10 ** (rng.uniform(-3.5, -2.7) if huge else rng.uniform(-3.5, -1.5)))
        shi = slo * float(rng.uniform(1, 6) if huge else rng.uniform(1, 20))
        olo = float(rng.uniform(0.001, 0.5))
        ohi = float(rng.uniform(olo, 1.0))
        spread = float(rng.uniform(0.05, 1.5))
        kw = dict(sh_degree=deg, seed=int(rng.integers(1 << 30)), spread=spread, scale_lo=slo, scale_hi=shi, opacity_lo=olo,
                  opacity_hi=ohi, behind_fraction=float(rng.choice([0.0, 0.1])), M=int(rng.choice([(deg + 1) ** 2, 16])),
                  bg=tuple(rng.uniform(0, 1, 3)))
        dpix = (rng.uniform(-1, 1, (3, H, W)) / (H * W)).astype(np.float32)
        name = f"fuzz{k}: P={P} {H}x{W} deg={deg} scale=[{slo:.4f},{shi:.4f}] op=[{olo:.3f},{ohi:.3f}] spread={spread:.2f}"
        yield k, P, H, W, kw, dpix, name
        k += 1


def fuzz_case(seed, k, big=False):
    """Iteration k of fuzz_stream(seed, big)."""
    for c in fuzz_stream(seed, big):
        if c[0] == k:
            return c


def fuzz_camera(seed, k, H, W):
    """A random look-at camera for case k of a fuzz stream (its own generator: the stream's draws are untouched): eye 0.5 .. 1.6 from
    the centre of random_scene's cube, looking at a point near it, any roll, tan(fov_x / 2) 0.1 .. 0.7 — general view and
    projection matrices, Gaussians beside and behind the camera, the 1.3 x tan(fov) clamp of the covariance Jacobian."""
    import math
    from fateavatar_amd import scenes
    rng = np.random.default_rng([seed, k, 77])
    d = rng.normal(size=3)
    eye = np.array([0.0, 0.0, 1.0]) + d / np.linalg.norm(d) * rng.uniform(0.5, 1.6)
    target = np.array([0.0, 0.0, 1.0]) + rng.uniform(-0.15, 0.15, 3)
    up = rng.normal(size=3)
    tanfov = float(rng.uniform(0.1, 0.7))
    fov = 2 * math.atan(tanfov)
    fovy = 2 * math.atan(tanfov * H / W)
    return scenes.look_at_camera(eye, target, up, fov, fovy, H, W)


def fuzz_inputs(seed, k, s):
    """The API's optional inputs for case k of a fuzz stream, drawn per case (their own generator): colors_precomp instead of
    the SH coefficients (35 %), cov3D_precomp — the scene's own covariance, from a first oracle pass — (35 %), scale_modifier
    0.4 .. 1.6 (50 %).  Keyword arguments of oracle_forward / HipFrame."""
    r = np.random.default_rng([seed, k, 78])
    extra = {}
    if r.random() < 0.35:
        extra["colors_precomp"] = r.uniform(0, 1, (s.P, 3)).astype(np.float32)
    if r.random() < 0.35:
        extra["cov3D_precomp"] = oracle_forward(s).cov3D.copy()
    if r.random() < 0.5:
        extra["scale_modifier"] = float(r.uniform(0.4, 1.6))
    return extra
