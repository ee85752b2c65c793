"""Known-answer vectors for the rasterizer proper that NEITHER oracle/fr_oracle.c NOR the HIP kernels produced.

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_known_answers.py        (build container only)

The reference ships no test or fixture for its rasterizer, and its CUDA sources cannot be built in this image, so the
blend / binning / backward arithmetic has no reference-held pin.  This script is the substitute VERDICT r1 asked for:

  * forward: a float64 numpy evaluation of the rasterizer's mathematics for tiny analytic scenes, written from the
    formulas the reference cites (EWA splatting projection, forward.cu:74-113; Sigma = R S S R^T, forward.cu:118-152;
    near cull / 1e-7 homogeneous guard, auxiliary.h:139-164; ndc2Pix and getRect, auxiliary.h:41-56; conic / radius,
    forward.cu:216-232; the per-pixel loop and its three tests, forward.cu:330-361; output, forward.cu:367-373).  The
    SH colour comes from the REFERENCE's own `eval_sh` (tools/gs_utils/sh_utils.py, imported here), the camera
    matrices from the reference's `getWorld2View2` / `getProjectionMatrix` (tools/gs_utils/graphics_utils.py).
  * gradients: CENTRAL FINITE DIFFERENCES of that float64 forward (loss = sum(out * dL_dpix)), i.e. no restatement
    of backward.cu at all.  Two places where the reference's backward is deliberately NOT the derivative of its forward
    are modelled in the forward that is differentiated:
      - guard-band clamp (backward.cu:168-176,262-264): the clamped t.x / t.y are constants (stop-gradient);
      - dL_dmeans2D is the derivative w.r.t. the NDC-scaled pixel centre in the blend only (backward.cu:460-461,545-546):
        obtained by perturbing the 2D centres with everything else fixed, times 0.5 W / 0.5 H.
      - dL_dscales is the derivative with respect to scale_modifier * scale (backward.cu:295,322-325: the chain rule's factor
        `mod` is not applied): the finite difference divided by the modifier (scenes "random_inputs_*", the only ones with a
        modifier other than 1 — found BY those scenes: the oracle, which restates backward.cu, disagreed with the plain finite
        difference by exactly the modifier).
      - the 0.99 alpha clamp (forward.cu:343) whose gradient the reference passes straight through (backward.cu:499-534:
        dL_dalpha goes on to opacity and G as if alpha = opacity * G): a pair clamped at the base point is evaluated as
        alpha = 0.99 + (opacity * G - (opacity * G at the base point)) — the clamped VALUE, the unclamped DERIVATIVE
        (scene "alpha_clamp_gradient"; the older forward-only clamp scene is kept).

Two scenes go beyond a single tile: "big_lists" (450 Gaussians on 48 x 48 pixels = 3 x 3 reference tiles, 6 x 6 tiles of the
8 x 8 implementation; a dense cluster gives lists of several hundred entries per tile, so the image depends on the 64-bit
tile|depth key order across tiles, on lists longer than one 64-record blend unit and longer than the 256-key register sort)
with its image, transmittance, radii and contributor counts, and finite-difference gradients of a sample of its
Gaussians.  They are evaluated by a pixel-vectorised copy of the same forward (forward64v), which this script first
checks against the scalar one on every small scene.

Six more ("random_camera_k", round 5) are drawn at random under general cameras — a rotated, translated view, images that are no
multiple of a tile, 70 - 110 anisotropic Gaussians from sub-pixel to a quarter of the image, SH degree 0 .. 3, Gaussians behind the
camera and off screen — with finite-difference gradients of twelve Gaussians each; four of the same family go through the
API's optional inputs (colors_precomp, cov3D_precomp, both, scale_modifier 0.6 / 1.5), three sit at the edges of the tiling
(7 x 5, 16 x 16, 17 x 33), "random_big_600" puts 600 Gaussians on 64 x 64 (lists of hundreds of entries per tile) and
"head_like_1500" is the regime of BASELINE config 2 in small (this one takes 25 minutes of finite differences).

Every scene is checked to sit far from the discrete decisions (power > 0, alpha < 1/255, T < 1e-4, radius ceil, tile
rectangle, depth order), so that an fp32 implementation takes the same ones.  Output: tests/golden/known_answers.npz
(inputs + expected outputs; data only).
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
torch.Tensor.cuda = lambda self, *a, **k: self

from tools.gs_utils import sh_utils  # noqa: E402
from tools.gs_utils.graphics_utils import getProjectionMatrix, getWorld2View2  # noqa: E402

TILE = 16  # config.h:16-17


def camera(R, T, fovx, fovy):
    """camera_3dgs.py:53-72 with the reference's own helpers."""
    wvt = torch.tensor(getWorld2View2(np.asarray(R, np.float64), np.asarray(T, np.float64))).float().transpose(0, 1)
    proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    return wvt.numpy().astype(np.float32), full.numpy().astype(np.float32), center.numpy().astype(np.float32)


def sh_colour(D, sh, direction):
    """max(eval_sh + 0.5, 0) (forward.cu:63-70) through the reference's eval_sh, float64."""
    s = torch.from_numpy(np.ascontiguousarray(sh.T))[None]          # [1, 3, M]
    d = torch.from_numpy(direction)[None]
    v = sh_utils.eval_sh(D, s, d)[0].numpy() + 0.5
    return np.maximum(v, 0.0)


def per_gaussian(sc, frozen=None, margins=None):
    """Float64 preprocess.  Returns a dict of per-Gaussian arrays; `frozen` = (flag_x, tx, flag_y, ty) of the baseline."""
    f64 = np.float64
    P = sc["means3D"].shape[0]
    W, H = sc["W"], sc["H"]
    view, proj, campos = sc["viewmatrix"].astype(f64), sc["projmatrix"].astype(f64), sc["campos"].astype(f64)
    fx, fy = W / (2.0 * sc["tanfovx"]), H / (2.0 * sc["tanfovy"])
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    out = dict(radii=np.zeros(P, np.int32), xy=np.zeros((P, 2)), depth=np.zeros(P), conic=np.zeros((P, 3)),
               rgb=np.zeros((P, 3)), rect=np.zeros((P, 4), np.int64), clampx=np.zeros(P, bool), clampy=np.zeros(P, bool),
               tx=np.zeros(P), ty=np.zeros(P))
    for i in range(P):
        p = sc["means3D"][i].astype(f64)
        pv = np.append(p, 1.0) @ view
        if pv[2] <= 0.2:
            continue
        ph = np.append(p, 1.0) @ proj
        pw = 1.0 / (ph[3] + 0.0000001)
        ndc = ph[:2] * pw
        if sc.get("cov3D_precomp") is not None:
            c6 = sc["cov3D_precomp"][i].astype(f64)
            Sig = np.array([[c6[0], c6[1], c6[2]], [c6[1], c6[3], c6[4]], [c6[2], c6[4], c6[5]]])
        else:
            r, x, y, z = sc["rotations"][i].astype(f64)
            Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                           [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                           [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
            Mm = Rm @ np.diag(sc["scale_modifier"] * sc["scales"][i].astype(f64))
            Sig = Mm @ Mm.T
        limx, limy = 1.3 * sc["tanfovx"], 1.3 * sc["tanfovy"]
        tz = pv[2]
        cx, cy = abs(pv[0] / tz) > limx, abs(pv[1] / tz) > limy
        tx = min(limx, max(-limx, pv[0] / tz)) * tz
        ty = min(limy, max(-limy, pv[1] / tz)) * tz
        if frozen is not None:                       # stop-gradient of the clamped values (backward.cu:168-176)
            if frozen[0][i]:
                tx = frozen[1][i]
            if frozen[2][i]:
                ty = frozen[3][i]
        out["clampx"][i], out["clampy"][i], out["tx"][i], out["ty"][i] = cx, cy, tx, ty
        J = np.array([[fx / tz, 0.0, -fx * tx / (tz * tz)], [0.0, fy / tz, -fy * ty / (tz * tz)]])
        Wm = view[:3, :3].T                          # world -> view rotation
        cov = J @ Wm @ Sig @ Wm.T @ J.T
        a, b, c = cov[0, 0] + 0.3, cov[0, 1], cov[1, 1] + 0.3
        det = a * c - b * b
        if det == 0.0:
            continue
        mid = 0.5 * (a + c)
        lam = mid + math.sqrt(max(0.1, mid * mid - det))
        lam2 = mid - math.sqrt(max(0.1, mid * mid - det))
        rr = 3.0 * math.sqrt(max(lam, lam2))
        radius = int(math.ceil(rr))
        px, py = ((ndc[0] + 1.0) * W - 1.0) * 0.5, ((ndc[1] + 1.0) * H - 1.0) * 0.5
        e = [(px - radius) / TILE, (py - radius) / TILE, (px + radius + TILE - 1) / TILE, (py + radius + TILE - 1) / TILE]
        x0, y0 = min(gx, max(0, int(e[0]))), min(gy, max(0, int(e[1])))     # int(): C truncation
        x1, y1 = min(gx, max(0, int(e[2]))), min(gy, max(0, int(e[3])))
        if margins is not None:
            margins.append(("radius", abs(rr - round(rr)), i))
            margins.extend(("rect", abs(v - round(v)), i) for v in e if -0.5 < v < max(gx, gy) + 0.5)
        if (x1 - x0) * (y1 - y0) == 0:
            continue
        if sc.get("colors_precomp") is not None:
            rgb = sc["colors_precomp"][i].astype(f64)
        else:
            d = p - campos
            rgb = sh_colour(sc["D"], sc["shs"][i].astype(f64), d / np.linalg.norm(d))
        out["radii"][i], out["xy"][i], out["depth"][i] = radius, (px, py), pv[2]
        out["conic"][i], out["rgb"][i], out["rect"][i] = (c / det, -b / det, a / det), rgb, (x0, y0, x1, y1)
    return out


def forward64(sc, frozen=None, d_xy=None, order=None, margins=None):
    g = per_gaussian(sc, frozen, margins)
    W, H = sc["W"], sc["H"]
    P = sc["means3D"].shape[0]
    xy = g["xy"] + (d_xy if d_xy is not None else 0.0)
    op = sc["opacities"].astype(np.float64).reshape(-1)
    bg = sc["bg"].astype(np.float64)
    vis = np.nonzero(g["radii"] > 0)[0]
    if order is None:   # sort on the float32 depth bits, stable in the Gaussian index (rasterizer_impl.cu:88-108,300-308)
        order = sorted(vis.tolist(), key=lambda i: (np.float32(g["depth"][i]), i))
    color = np.zeros((3, H, W))
    final_T = np.ones((H, W))
    n_contrib = np.zeros((H, W), np.int64)
    decisions = []
    for y in range(H):
        for x in range(W):
            tx_, ty_ = x // TILE, y // TILE
            T, C, contributor, last = 1.0, np.zeros(3), 0, 0
            taken = 0   # bit k set: list entry k was blended (the discrete decisions of this pixel)
            for i in order:
                x0, y0, x1, y1 = g["rect"][i]
                if not (x0 <= tx_ < x1 and y0 <= ty_ < y1):
                    continue
                contributor += 1
                dx, dy = xy[i, 0] - x, xy[i, 1] - y
                A, B, Cc = g["conic"][i]
                power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
                if margins is not None and A * Cc - B * B <= 0:   # a positive-definite conic cannot give power > 0
                    margins.append(("power", abs(power), i))
                if power > 0.0:
                    continue
                alpha = min(0.99, op[i] * math.exp(power))
                if margins is not None:
                    margins.append(("alpha", abs(alpha * 255.0 - 1.0), i))
                if alpha < 1.0 / 255.0:
                    continue
                test_T = T * (1.0 - alpha)
                if margins is not None:
                    margins.append(("T", abs(test_T / 1e-4 - 1.0), i))
                if test_T < 0.0001:
                    break
                C += g["rgb"][i] * alpha * T
                T = test_T
                last = contributor
                taken |= 1 << contributor
            decisions.append(taken)
            final_T[y, x], n_contrib[y, x] = T, last
            color[:, y, x] = C + T * bg
    sig = (tuple(decisions), g["radii"].tobytes(), g["rect"].tobytes(), (g["rgb"] > 0).tobytes())
    return dict(color=color, final_T=final_T, n_contrib=n_contrib, radii=g["radii"], order=order, g=g, sig=sig)


def per_gaussian_all(sc, frozen=None, margins=None, base=None, only=None):
    """per_gaussian() for every Gaussian, or (base given) only for Gaussian `only` on top of the cached `base` — a finite
    difference moves one Gaussian at a time."""
    if base is None or only is None:
        return per_gaussian(sc, frozen, margins)
    one = {k: (np.asarray(v)[only:only + 1] if (isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == sc["means3D"].shape[0]
                                               and k not in ("viewmatrix", "projmatrix", "campos", "bg", "dL_dpix")) else v)
           for k, v in sc.items()}
    fr = None if frozen is None else tuple(np.asarray(a)[only:only + 1] for a in frozen)
    g1 = per_gaussian(one, fr)
    g = {k: v.copy() for k, v in base.items()}
    for k in g:
        g[k][only] = g1[k][0]
    return g


def forward64v(sc, frozen=None, d_xy=None, order=None, margins=None, gcache=None, only=None, clamp_ref=None):
    """forward64 with the per-pixel loop vectorised over the pixels (identical arithmetic per pixel, float64).
    `clamp_ref`: (clamped [n, H, W] bool, raw0 [n, H, W]) of the base point — pairs clamped there take
    alpha = 0.99 + (raw - raw0), see the module docstring.  Returns also `blended` [n, H, W] and `raw` [n, H, W]."""
    g = per_gaussian_all(sc, frozen, margins, gcache, only)
    W, H = sc["W"], sc["H"]
    xy = g["xy"] + (d_xy if d_xy is not None else 0.0)
    op = sc["opacities"].astype(np.float64).reshape(-1)
    bg = sc["bg"].astype(np.float64)
    vis = np.nonzero(g["radii"] > 0)[0]
    if order is None:
        order = sorted(vis.tolist(), key=lambda i: (np.float32(g["depth"][i]), i))
    ys, xs = np.mgrid[0:H, 0:W]
    tx_, ty_ = xs // TILE, ys // TILE
    T = np.ones((H, W))
    C = np.zeros((3, H, W))
    contributor = np.zeros((H, W), np.int64)
    last = np.zeros((H, W), np.int64)
    done = np.zeros((H, W), bool)
    n = len(order)
    blended = np.zeros((n, H, W), bool)
    raws = np.zeros((n, H, W))
    for k, i in enumerate(order):
        x0, y0, x1, y1 = g["rect"][i]
        inrect = (tx_ >= x0) & (tx_ < x1) & (ty_ >= y0) & (ty_ < y1) & ~done
        contributor = contributor + inrect
        dx, dy = xy[i, 0] - xs, xy[i, 1] - ys
        A, B, Cc = g["conic"][i]
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
        ok = inrect & ~(power > 0.0)
        raw = op[i] * np.exp(np.minimum(power, 0.0))
        raws[k] = raw
        alpha = np.minimum(0.99, raw)
        if clamp_ref is not None:
            alpha = np.where(clamp_ref[0][k], 0.99 + (raw - clamp_ref[1][k]), alpha)
        if margins is not None:
            if A * Cc - B * B <= 0:
                margins.extend(("power", float(v), i) for v in np.abs(power[inrect]))
            margins.extend(("alpha", float(v), i) for v in np.abs(alpha[ok] * 255.0 - 1.0))
        ok2 = ok & ~(alpha < 1.0 / 255.0)
        test_T = T * (1.0 - alpha)
        if margins is not None:
            margins.extend(("T", float(v), i) for v in np.abs(test_T[ok2] / 1e-4 - 1.0))
        stop = ok2 & (test_T < 0.0001)
        done = done | stop
        take = ok2 & ~stop
        C = C + np.where(take, alpha * T, 0.0)[None] * g["rgb"][i][:, None, None]
        T = np.where(take, test_T, T)
        last = np.where(take, contributor, last)
        blended[k] = take
    color = C + T[None] * bg[:, None, None]
    sig = (blended.tobytes(), g["radii"].tobytes(), g["rect"].tobytes(), (g["rgb"] > 0).tobytes())
    return dict(color=color, final_T=T, n_contrib=last, radii=g["radii"], order=order, g=g, sig=sig, blended=blended, raw=raws)


def known_answer_big(sc, sample, clamp_model=False):
    """Known answers of a scene too big for the scalar loops: forward by forward64v; finite differences for the
    Gaussians in `sample` only (every parameter of those), one Gaussian re-projected per evaluation."""
    margins = []
    base = forward64v(sc, margins=margins)
    need = dict(power=1e-4, alpha=1e-4, T=1e-4, radius=2e-3, rect=2e-4)
    for m in margins:
        assert m[1] > need[m[0]], (sc["name"], "scene sits on a discrete decision", m)
    dep = [base["g"]["depth"][i] for i in base["order"]]
    for a, b in zip(dep, dep[1:]):
        assert a == b or b - a > 1e-5, (sc["name"], "depths too close", a, b)
    res = dict(color=base["color"], final_T=base["final_T"], n_contrib=base["n_contrib"].astype(np.int32), radii=base["radii"])
    if sample is None:
        return res, base
    g0 = base["g"]
    frozen = (g0["clampx"], g0["tx"], g0["clampy"], g0["ty"])
    G = sc["dL_dpix"].astype(np.float64)
    order = base["order"]
    clamp_ref = None
    if clamp_model:
        clamped = base["raw"] > 0.99
        # (only pairs that are evaluated at all matter; a clamped pair must not sit on the clamp's edge either)
        edge = np.abs(base["raw"][base["blended"]] - 0.99)
        assert edge.min() > 1e-3, (sc["name"], "a blended pair sits on the 0.99 clamp", edge.min())
        assert (clamped & base["blended"]).sum() > 0, (sc["name"], "no clamped pair")
        clamp_ref = (clamped, base["raw"])
        again = forward64v(sc, frozen, None, order, gcache=g0, clamp_ref=clamp_ref)
        assert np.abs(again["color"] - base["color"]).max() < 1e-14
    base_sig = forward64v(sc, frozen, None, order, gcache=g0, clamp_ref=clamp_ref)["sig"]
    assert base_sig == base["sig"]

    def loss(scene, only, d_xy=None):
        f = forward64v(scene, frozen, d_xy, order, gcache=g0, only=only, clamp_ref=clamp_ref)
        return float((f["color"] * G).sum()), f["sig"]

    def central(make, scale):
        for h in (1e-6 * scale, 1e-7 * scale, 1e-8 * scale, 1e-9 * scale):
            (lp, sp), (lm, sm) = loss(*make(+h)), loss(*make(-h))
            if sp == base["sig"] and sm == base["sig"]:
                return (lp - lm) / (2 * h)
        raise AssertionError((sc["name"], "no step keeps the discrete decisions"))

    def fd_rows(key):
        arr = sc[key].astype(np.float64)
        a2 = arr.reshape(arr.shape[0], -1)
        out = np.zeros((len(sample), a2.shape[1]))
        for r, i in enumerate(sample):
            for c in range(a2.shape[1]):
                def make(h, i=i, c=c):
                    a = a2.copy()
                    a[i, c] += h
                    return dict(sc, **{key: a.reshape(arr.shape)}), i
                out[r, c] = central(make, max(1.0, abs(float(a2[i, c]))))
        return out.reshape((len(sample),) + arr.shape[1:])

    P = sc["means3D"].shape[0]
    res["sample"] = np.asarray(sample, np.int32)
    res["dL_dmeans3D"] = fd_rows("means3D")
    res["dL_dopacity"] = fd_rows("opacities").reshape(len(sample), 1)
    if sc.get("colors_precomp") is not None:
        res["dL_dcolors"] = fd_rows("colors_precomp")
    else:
        res["dL_dsh"] = fd_rows("shs")
    if sc.get("cov3D_precomp") is not None:
        res["dL_dcov3D"] = fd_rows("cov3D_precomp")   # NB: the 6 stored floats; off-diagonals enter the matrix twice
    else:
        # (the reference differentiates with respect to the MODIFIED scale: see known_answer below)
        res["dL_dscales"] = fd_rows("scales") / float(sc["scale_modifier"])
        res["dL_drotations"] = fd_rows("rotations")
    d2 = np.zeros((len(sample), 3))
    for r, i in enumerate(sample):
        for k, half in ((0, 0.5 * sc["W"]), (1, 0.5 * sc["H"])):
            def make(h, i=i, k=k):
                d = np.zeros((P, 2))
                d[i, k] = h
                return sc, None, d
            d2[r, k] = central(make, 1.0) * half
    res["dL_dmeans2D"] = d2
    return res, base


def known_answer(sc, grads=True):
    margins = []
    base = forward64(sc, margins=margins)
    worst = {}
    for k, v, _ in margins:
        worst[k] = min(worst.get(k, np.inf), v)
    need = dict(power=1e-4, alpha=1e-4, T=1e-4, radius=2e-3, rect=2e-4)
    for k, v in worst.items():
        assert v > need[k], (sc["name"], "scene sits on a discrete decision", k, v)
    # depth order must not hinge on fp32 rounding: distinct depths differ by > 1e-5, or are exactly equal
    dep = [base["g"]["depth"][i] for i in base["order"]]
    for a, b in zip(dep, dep[1:]):
        assert a == b or b - a > 1e-5, (sc["name"], "depths too close", a, b)
    res = dict(color=base["color"], final_T=base["final_T"], n_contrib=base["n_contrib"].astype(np.int32), radii=base["radii"])
    if not grads:
        return res
    g0 = base["g"]
    frozen = (g0["clampx"], g0["tx"], g0["clampy"], g0["ty"])
    G = sc["dL_dpix"].astype(np.float64)
    order = base["order"]

    def loss(scene, d_xy=None):
        f = forward64(scene, frozen, d_xy, order)
        return float((f["color"] * G).sum()), f["sig"]

    def central(make, scale):
        """Central difference whose two evaluations take the SAME discrete decisions as the base point (blended
        entries per pixel, radii, rectangles, SH clamps): the step shrinks until they do."""
        for h in (1e-6 * scale, 1e-7 * scale, 1e-8 * scale, 1e-9 * scale):
            (lp, sp), (lm, sm) = loss(*make(+h)), loss(*make(-h))
            if sp == base["sig"] and sm == base["sig"]:
                return (lp - lm) / (2 * h)
        raise AssertionError((sc["name"], "no step keeps the discrete decisions"))

    def fd(key):
        arr = sc[key].astype(np.float64)
        out = np.zeros_like(arr)
        it = np.nditer(arr, flags=["multi_index"])
        for v in it:
            idx = it.multi_index

            def make(h, idx=idx):
                a = arr.copy()
                a[idx] += h
                return (dict(sc, **{key: a}),)
            out[idx] = central(make, max(1.0, abs(float(v))))
        return out

    P = sc["means3D"].shape[0]
    res["dL_dmeans3D"] = fd("means3D")
    res["dL_dopacity"] = fd("opacities").reshape(P, 1)
    if sc.get("colors_precomp") is not None:
        res["dL_dcolors"] = fd("colors_precomp")
    else:
        res["dL_dsh"] = fd("shs")
    if sc.get("cov3D_precomp") is not None:
        res["dL_dcov3D"] = fd("cov3D_precomp")   # NB: the 6 stored floats; off-diagonals enter the matrix twice
    else:
        # A third place where the reference's backward is not the derivative of its forward: Sigma is built from s = mod * scale
        # (backward.cu:295) and dL_dscale is dot(Rt[i], dL_dMt[i]) (backward.cu:322-325) — the derivative with respect to s, the
        # factor `mod` of the chain rule is not applied.  With scale_modifier = 1 (every call of FateAvatar's render path) the two are
        # the same; the expected value here is the finite difference divided by the modifier.
        res["dL_dscales"] = fd("scales") / float(sc["scale_modifier"])
        res["dL_drotations"] = fd("rotations")
    d2 = np.zeros((P, 3))
    for i in range(P):
        for k, half in ((0, 0.5 * sc["W"]), (1, 0.5 * sc["H"])):
            def make(h, i=i, k=k):
                d = np.zeros((P, 2))
                d[i, k] = h
                return sc, d
            d2[i, k] = central(make, 1.0) * half
    res["dL_dmeans2D"] = d2
    return res


def quat(axis, deg):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    h = math.radians(deg) / 2
    return np.concatenate([[math.cos(h)], math.sin(h) * a]).astype(np.float32)


def scenes():
    rng = np.random.default_rng(20260928)
    f32 = np.float32
    out = []

    def base(name, W, H, tan, R=np.eye(3), T=(0, 0, 0), bg=(0.2, 0.5, 0.9)):
        fov = 2 * math.atan(tan)
        fovy = 2 * math.atan(tan * H / W)
        wvt, full, ctr = camera(R, T, fov, fovy)
        return dict(name=name, W=W, H=H, tanfovx=math.tan(fov / 2), tanfovy=math.tan(fovy / 2), viewmatrix=wvt, projmatrix=full,
                    campos=ctr, bg=np.asarray(bg, f32), scale_modifier=1.0,
                    dL_dpix=(rng.uniform(-1, 1, (3, H, W)) / (H * W)).astype(f32))

    def sh(P, D, M):
        s = np.zeros((P, M, 3), f32)
        s[:, :(D + 1) ** 2] = rng.uniform(-0.6, 0.9, (P, (D + 1) ** 2, 3))
        return s

    # A: one isotropic Gaussian whose centre projects onto a pixel centre (10, 7) of a 32 x 32 image
    s = base("single_centred", 32, 32, 0.25)
    z = 2.0
    ndc = lambda p, S: (2 * p + 1) / S - 1  # noqa: E731
    # (3e-3 px off the exact centre, so that no implementation can round the centre pixel's power to +0)
    s.update(means3D=np.asarray([[ndc(10.003, 32) * s["tanfovx"] * z, ndc(6.998, 32) * s["tanfovy"] * z, z]], f32),
             scales=np.full((1, 3), 0.035, f32), rotations=quat((0, 0, 1), 0)[None], opacities=np.asarray([0.6], f32),
             shs=sh(1, 0, 1), D=0)
    out.append(s)

    # B: two anisotropic Gaussians at EXACTLY the same depth (order = index), a third behind them, one behind the camera
    #    (culled), one far off screen (empty rectangle); rotated camera; 40 x 24 image = partial edge tiles; SH degree 1
    #    (unrotated camera: view z = z + T_z is then computed identically for both, an exact tie in any precision)
    s = base("depth_tie", 40, 24, 0.3, T=(0.02, -0.01, 0.1), bg=(0.0, 0.0, 0.0))
    m = np.asarray([[0.03, 0.02, 1.5], [-0.05, 0.03, 1.5], [0.0, 0.0, 1.9], [0.0, 0.0, -1.0], [5.0, 0.1, 1.5]], np.float64)
    s.update(means3D=m.astype(f32),
             scales=np.asarray([[0.05, 0.02, 0.03], [0.02, 0.06, 0.02], [0.12, 0.1, 0.05], [0.05, 0.05, 0.05], [0.05, 0.05, 0.05]], f32),
             rotations=np.stack([quat((1, 2, 3), 40), quat((3, -1, 1), 75), quat((0, 1, 0), 20), quat((1, 0, 0), 0), quat((1, 0, 0), 0)]),
             opacities=np.asarray([0.5, 0.7, 0.9, 0.5, 0.5], f32), shs=sh(5, 1, 4), D=1)
    out.append(s)

    # C: a large Gaussian whose centre lies OUTSIDE the 1.3 tan(fov) guard band (clamped Jacobian, stop-gradient quirk)
    #    but whose footprint reaches into the image, plus an ordinary one; SH degree 2
    s = base("guard_band", 32, 32, 0.3, bg=(1.0, 1.0, 1.0))
    s.update(means3D=np.asarray([[0.62, 0.05, 1.4], [-0.02, 0.49, 1.2], [0.05, -0.04, 1.0]], f32),
             scales=np.asarray([[0.16, 0.09, 0.1], [0.07, 0.15, 0.08], [0.04, 0.03, 0.05]], f32),
             rotations=np.stack([quat((0, 0, 1), 25), quat((1, 1, 0), 50), quat((1, 0, 1), 10)]),
             opacities=np.asarray([0.8, 0.6, 0.4], f32), shs=sh(3, 2, 9), D=2)
    out.append(s)

    # D: colors_precomp + cov3D_precomp; the second covariance is INDEFINITE, so the conic is too and part of its
    #    pixels have power > 0 (skipped, forward.cu:336-337)
    s = base("indefinite_precomp", 32, 32, 0.25)
    s.update(means3D=np.asarray([[0.03, 0.0, 1.3], [-0.04, 0.02, 1.6]], f32),
             cov3D_precomp=np.asarray([[0.004, 0.001, 0.0, 0.006, 0.0005, 0.003],
                                       [0.004, 0.009, 0.0, -0.002, 0.0, 0.003]], f32),
             colors_precomp=np.asarray([[0.9, 0.2, 0.1], [0.1, 0.6, 0.8]], f32), opacities=np.asarray([0.7, 0.9], f32),
             scales=None, rotations=None, shs=None, D=0)
    out.append(s)

    # E: eight almost coincident, fairly opaque Gaussians: pixels near the centre TERMINATE (T < 1e-4) part-way through the
    #    list, pixels further out blend all of them; SH degree 3
    ang = math.radians(7)
    Rb = np.array([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
    s = base("terminates", 32, 32, 0.25, R=Rb, T=(0.15, 0.01, 0.05), bg=(0.3, 0.1, 0.7))
    n = 8
    m = np.zeros((n, 3), np.float64)
    m[:, 0] = 0.01 + 0.004 * rng.standard_normal(n)
    m[:, 1] = -0.02 + 0.004 * rng.standard_normal(n)
    m[:, 2] = 1.2 + 0.05 * np.arange(n)
    s.update(means3D=m.astype(f32), scales=rng.uniform(0.03, 0.06, (n, 3)).astype(f32),
             rotations=np.stack([quat(rng.standard_normal(3), float(rng.uniform(0, 180))) for _ in range(n)]),
             opacities=np.full(n, 0.85, f32), shs=sh(n, 3, 16), D=3)
    out.append(s)

    # F (forward only): opacity 1 -> alpha is clamped at 0.99 near the centre (forward.cu:343)
    s = base("alpha_clamp_forward_only", 32, 32, 0.25)
    s.update(means3D=np.asarray([[0.0, 0.0, 1.5], [0.02, 0.01, 2.0]], f32), scales=np.asarray([[0.06] * 3, [0.1] * 3], f32),
             rotations=np.stack([quat((0, 0, 1), 0)] * 2), opacities=np.asarray([1.0, 0.9], f32), shs=sh(2, 0, 1), D=0,
             forward_only=True)
    out.append(s)

    # G: the 0.99 clamp WITH gradients (frozen-offset model, see the module docstring): an opaque Gaussian (opacity 1:
    #    clamped within ~0.14 sigma of its centre) between a translucent one in front and a wide one behind
    s = base("alpha_clamp_gradient", 32, 32, 0.25, bg=(0.1, 0.4, 0.2))
    zc = 1.5   # (centre 3e-3 px off pixel (15, 15): that pixel and its four neighbours have opacity * G > 0.99)
    s.update(means3D=np.asarray([[0.012, -0.01, 1.2], [ndc(15.003, 32) * s["tanfovx"] * zc, ndc(14.998, 32) * s["tanfovy"] * zc, zc],
                                 [0.03, 0.02, 2.0]], f32),
             scales=np.asarray([[0.05, 0.03, 0.04], [0.2, 0.2, 0.2], [0.12, 0.1, 0.1]], f32),
             rotations=np.stack([quat((1, 1, 0), 30), quat((0, 0, 1), 0), quat((0, 1, 1), 60)]),
             opacities=np.asarray([0.45, 1.0, 0.8], f32), shs=sh(3, 1, 4), D=1, big=True, clamp_model=True, sample=[0, 1, 2])
    out.append(s)

    # H: lists at the scale of a real frame (see the module docstring).  48 x 48 pixels; a dense cluster of 300 small,
    #    faint splats over one 8 x 8 tile; 110 medium ones everywhere; 25 wide ones across several tiles; 15 opaque ones at
    #    the back of the cluster, so that some pixels cross T < 1e-4 only hundreds of entries into their list.
    s = base("big_lists", 48, 48, 0.3, T=(0.01, -0.02, 0.0), bg=(0.05, 0.1, 0.15))
    P = 450
    px_w = lambda zz: 2.0 * s["tanfovx"] * zz / 48.0     # noqa: E731  (world size of a pixel at depth zz)
    slots = rng.permutation(P)
    zs = 1.0 + 0.004 * slots + rng.uniform(0.0, 0.001, P)            # distinct depths, gaps > 1e-3

    def draw(i):
        zz = zs[i]
        if i < 300:      # cluster over pixels 16..23
            cx, cy, sig, opa = 19.5 + rng.uniform(-3.5, 3.5), 19.5 + rng.uniform(-3.5, 3.5), rng.uniform(0.8, 1.6), rng.uniform(0.008, 0.04)
        elif i < 410:    # everywhere
            cx, cy, sig, opa = rng.uniform(1, 47), rng.uniform(1, 47), rng.uniform(1.2, 3.5), rng.uniform(0.1, 0.6)
        elif i < 435:    # wide
            cx, cy, sig, opa = rng.uniform(4, 44), rng.uniform(4, 44), rng.uniform(5.0, 9.0), rng.uniform(0.04, 0.15)
        else:            # opaque, at the back of the cluster
            cx, cy, sig, opa = 21.0 + rng.uniform(-2.5, 2.5), 20.0 + rng.uniform(-2.5, 2.5), rng.uniform(1.0, 2.0), rng.uniform(0.85, 0.97)
        if i >= 435:
            zz = 3.0 + 0.01 * (i - 435) + rng.uniform(0.0, 0.002)
        x = ((2 * cx + 1) / 48.0 - 1) * s["tanfovx"] * zz - 0.01
        y = ((2 * cy + 1) / 48.0 - 1) * s["tanfovy"] * zz + 0.02
        sc3 = sig * px_w(zz) * rng.uniform(0.6, 1.4, 3)
        return (np.asarray([x, y, zz], f32), sc3.astype(f32), quat(rng.standard_normal(3), float(rng.uniform(0, 180))), f32(opa))

    m3, sc3, rot, opa = np.zeros((P, 3), f32), np.zeros((P, 3), f32), np.zeros((P, 4), f32), np.zeros(P, f32)
    for i in range(P):
        m3[i], sc3[i], rot[i], opa[i] = draw(i)
    s.update(means3D=m3, scales=sc3, rotations=rot, opacities=opa, shs=sh(P, 1, 4), D=1, big=True)
    # keep every (pixel, Gaussian) pair and every radius / rectangle away from the discrete decisions: redraw the few
    # Gaussians that sit on one
    need = dict(power=1e-4, alpha=1e-4, T=1e-4, radius=2e-3, rect=2e-4)
    for attempt in range(200):
        margins = []
        forward64v(s, margins=margins)
        bad = sorted({int(mm[2]) for mm in margins if mm[1] <= 2.0 * need[mm[0]]})
        if not bad:
            break
        for i in bad:
            s["means3D"][i], s["scales"][i], s["rotations"][i], s["opacities"][i] = draw(i)
    else:
        raise AssertionError("big_lists: could not move every pair off the decisions")
    pick = np.concatenate([rng.choice(300, 16, replace=False), 300 + rng.choice(110, 10, replace=False),
                           410 + rng.choice(25, 5, replace=False), 435 + rng.choice(15, 5, replace=False)])
    s["sample"] = sorted(int(v) for v in pick)
    out.append(s)

    # I: a family of random scenes under GENERAL cameras (their own generators: the draws above are untouched): a rotated and
    #    translated view, images that are no multiple of a tile, 70 - 110 anisotropic Gaussians of every size from sub-pixel to a
    #    quarter of the image, opacities 0.03 .. 0.95, SH degree 0 .. 3 with 16 stored coefficients or exactly (D + 1)^2, a few
    #    Gaussians behind the camera and far off screen; finite-difference gradients of a sample of twelve each.
    for k, (W, H) in enumerate([(40, 28), (33, 47), (56, 24), (48, 40), (27, 27), (64, 36)]):
        out.append(random_camera_scene(k, W, H, base, sh))
    # J: the same family through the API's optional inputs: colors_precomp, cov3D_precomp (valid covariances, of random scales and
    #    rotations), both, and scale_modifier != 1 with near-opaque Gaussians (pixels that terminate inside the list)
    out.append(random_camera_scene(10, 44, 30, base, sh, name="random_inputs_colors", colors=True, mod=0.6))
    out.append(random_camera_scene(11, 36, 42, base, sh, name="random_inputs_cov3d", cov=True))
    out.append(random_camera_scene(12, 50, 26, base, sh, name="random_inputs_both", colors=True, cov=True))
    out.append(random_camera_scene(13, 38, 38, base, sh, name="random_inputs_modifier_opaque", mod=1.5, opa=(0.5, 0.97)))
    # K: image shapes at the edges of the tiling — smaller than a tile, exactly one reference tile, one pixel past a tile in both
    #    directions — and one scene at the scale of "big_lists" under a general camera (600 Gaussians on 64 x 64: lists of hundreds
    #    of entries in every tile, pixels that terminate deep inside them)
    out.append(random_camera_scene(20, 7, 5, base, sh, name="random_tiny_7x5", P=12))
    out.append(random_camera_scene(21, 16, 16, base, sh, name="random_one_tile_16x16", P=24))
    out.append(random_camera_scene(22, 17, 33, base, sh, name="random_past_a_tile_17x33", P=40))
    out.append(random_camera_scene(23, 64, 64, base, sh, name="random_big_600", P=600, n_sample=16))
    # L: the regime of BASELINE config 2 in small: 1 500 identical isotropic splats of opacity 0.1 on an ellipsoid with the head
    #    template's bounding box, the benchmark's camera (R = diag(1, -1, -1), T = (0, 1.47, 0.98), tan(fov / 2) = 0.2), SH degree 3
    #    with DC = 0 and a small rest, white background, 96 x 96 pixels: every footprint carries its alpha = 1/255 ring INSIDE the
    #    3-sigma rectangle, front and back surface overlap, ~25 splats per pixel
    out.append(head_like_scene(base))
    # ... and where training takes it (config/fateavatar.yaml: opacities driven towards 1): the same surface with 700 splats of
    #    opacity 0.9 on 64 x 64 — pixels terminate after a handful of entries, most of every list is never blended
    out.append(head_like_scene(base, name="head_like_opaque_700", P=700, res=64, opacity=0.9, tag=98))
    return out


def head_like_scene(base, name="head_like_1500", P=1500, res=96, opacity=0.1, tag=99):
    r = np.random.default_rng([20260929, tag])
    f32 = np.float32
    W = H = res
    s = base(name, W, H, 0.2, R=np.diag([1.0, -1.0, -1.0]), T=(0.0, 1.47, 0.98), bg=(1.0, 1.0, 1.0))
    s["dL_dpix"] = (r.uniform(-1, 1, (3, H, W)) / (H * W)).astype(f32)
    scale = 6.085e-4 * math.sqrt(100_000 / P)          # mean nearest-neighbour spacing at 100 k, scaled

    def draw(i):
        u, v = math.acos(r.uniform(-1, 1)), r.uniform(0, 2 * math.pi)
        return np.asarray([0.1036 * math.sin(u) * math.cos(v), 1.470 + 0.157 * math.cos(u), -0.021 + 0.111 * math.sin(u) * math.sin(v)], f32)

    m3 = np.stack([draw(i) for i in range(P)])
    shs = np.zeros((P, 16, 3), f32)
    shs[:, 1:] = 0.1 * r.uniform(-1, 1, (P, 15, 3))
    s.update(means3D=m3, scales=np.full((P, 3), scale, f32), rotations=np.tile(np.asarray([1, 0, 0, 0], f32), (P, 1)),
             opacities=np.full(P, opacity, f32), shs=shs, D=3, big=True)
    need = dict(power=1e-4, alpha=1e-4, T=1e-4, radius=2e-3, rect=2e-4)
    for attempt in range(400):
        margins = []
        base0 = forward64v(s, margins=margins)
        dep = np.sort(base0["g"]["depth"][base0["g"]["radii"] > 0])
        close = set()
        if len(dep) > 1 and np.diff(dep).min() <= 2e-5:          # distinct depths: the order must not hinge on fp32 rounding
            d_all = base0["g"]["depth"]
            for a in np.nonzero(np.diff(dep) <= 2e-5)[0]:
                close.update(int(j) for j in np.nonzero(d_all == dep[a])[0])
        bad = sorted({int(mm[2]) for mm in margins if mm[1] <= 2.0 * need[mm[0]]} | close)
        if not bad:
            break
        for i in bad:
            s["means3D"][i] = draw(i)
    else:
        raise AssertionError(name + ": could not move every pair off the decisions")
    s["sample"] = sorted(int(v) for v in r.choice(P, 12, replace=False))
    return s


def rotation_matrix(axis, deg):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    t = math.radians(deg)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(t) * K + (1 - math.cos(t)) * (K @ K)


def random_camera_scene(k, W, H, base, sh, name=None, colors=False, cov=False, mod=1.0, opa=(0.03, 0.95), P=None, n_sample=12):
    r = np.random.default_rng([20260929, k])
    f32 = np.float32
    R = rotation_matrix(r.standard_normal(3), float(r.uniform(10, 70)))
    T = r.uniform(-0.3, 0.3, 3)
    tan = float(r.uniform(0.2, 0.6))
    s = base(name or f"random_camera_{k}", W, H, tan, R=R, T=T, bg=tuple(r.uniform(0, 1, 3)))
    s["scale_modifier"] = mod
    s["dL_dpix"] = (r.uniform(-1, 1, (3, H, W)) / (H * W)).astype(f32)
    P = int(r.integers(70, 111)) if P is None else P
    D = k % 4
    M = 16 if k % 2 == 0 else (D + 1) ** 2
    zs = 1.0 + 0.02 * r.permutation(P) + r.uniform(0.0, 0.005, P)          # distinct view depths, gaps > 1e-2

    def draw(i):
        zz = zs[i]
        kind = i % 10
        if kind == 0 and i < 30:     # behind the camera / far off screen
            cam = np.array([r.uniform(-0.2, 0.2), r.uniform(-0.2, 0.2), -r.uniform(0.3, 2.0)]) if i % 20 == 0 else \
                np.array([zz * tan * r.uniform(4, 6), 0.1, zz])
            sig = 1.5
        else:
            cx, cy = r.uniform(-3, W + 3), r.uniform(-3, H + 3)
            sig = float(np.exp(r.uniform(np.log(0.5), np.log(0.25 * max(W, H)))))
            cam = np.array([((2 * cx + 1) / W - 1) * s["tanfovx"] * zz, ((2 * cy + 1) / H - 1) * s["tanfovy"] * zz, zz])
        world = R @ (cam - T)                                                    # x_view = R^T x_world + T (getWorld2View2)
        pxw = 2.0 * s["tanfovx"] * zz / W
        sc3 = sig * pxw * r.uniform(0.5, 1.5, 3) / (1.0 if cov else mod)
        return world.astype(f32), sc3.astype(f32), quat(r.standard_normal(3), float(r.uniform(0, 180))), f32(r.uniform(*opa))

    m3, sc3, rot, opa_v = np.zeros((P, 3), f32), np.zeros((P, 3), f32), np.zeros((P, 4), f32), np.zeros(P, f32)
    for i in range(P):
        m3[i], sc3[i], rot[i], opa_v[i] = draw(i)
    shs = np.zeros((P, M, 3), f32)
    shs[:, :(D + 1) ** 2] = r.uniform(-0.6, 0.9, (P, (D + 1) ** 2, 3))
    opa_arr = opa_v
    s.update(means3D=m3, scales=sc3, rotations=rot, opacities=opa_arr, shs=shs, D=D, big=True)
    if colors:
        s.update(colors_precomp=r.uniform(0, 1, (P, 3)).astype(f32), shs=None, D=0)

    def covariances():
        c = np.zeros((P, 6), f32)
        for i in range(P):
            q0, x, y, z = s["rotations"][i].astype(np.float64)
            Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - q0 * z), 2 * (x * z + q0 * y)],
                           [2 * (x * y + q0 * z), 1 - 2 * (x * x + z * z), 2 * (y * z - q0 * x)],
                           [2 * (x * z - q0 * y), 2 * (y * z + q0 * x), 1 - 2 * (x * x + y * y)]])
            Mm = Rm @ np.diag(s["scales"][i].astype(np.float64))
            S = Mm @ Mm.T
            c[i] = (S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2])
        return c

    need = dict(power=1e-4, alpha=1e-4, T=1e-4, radius=2e-3, rect=2e-4)
    for attempt in range(400):
        if cov:
            s["cov3D_precomp"] = covariances()
        margins = []
        forward64v(s, margins=margins)
        bad = sorted({int(mm[2]) for mm in margins if mm[1] <= 2.0 * need[mm[0]]})
        if not bad:
            break
        for i in bad:
            s["means3D"][i], s["scales"][i], s["rotations"][i], s["opacities"][i] = draw(i)
    else:
        raise AssertionError(s["name"] + ": could not move every pair off the decisions")
    if cov:
        s.update(scales=None, rotations=None)
    s["sample"] = sorted(int(v) for v in r.choice(P, min(P, n_sample), replace=False))
    return s


def main():
    blob = {}
    names = []
    # KA_REUSE=1: a scene whose inputs are byte-identical to those in the existing file keeps its expected outputs (the finite
    # differences of the large scenes take half an hour); without it everything is recomputed
    old = np.load(os.path.join(OUT, "known_answers.npz")) if os.environ.get("KA_REUSE") and os.path.exists(os.path.join(OUT, "known_answers.npz")) else None
    for sc in scenes():
        n = sc["name"]
        ins = {k: np.asarray(v) for k, v in sc.items() if k not in ("name", "forward_only", "big", "clamp_model", "sample") and v is not None}
        if old is not None and n in [str(x) for x in old["names"]]:
            old_in = {k.split("/", 2)[2]: old[k] for k in old.files if k.startswith(n + "/in/")}
            if set(old_in) == set(ins) and all(old_in[k].dtype == ins[k].dtype and np.array_equal(old_in[k], ins[k]) for k in ins):
                names.append(n)
                for k in old.files:
                    if k.startswith(n + "/"):
                        blob[k] = old[k]
                print(n, "(inputs unchanged: expected outputs kept)", flush=True)
                continue
        if sc.get("big"):
            ka, base = known_answer_big(sc, sc.get("sample"), clamp_model=sc.get("clamp_model", False))
            if sc.get("clamp_model"):
                print(n, "clamped blended pairs", int(((base["raw"] > 0.99) & base["blended"]).sum()))
        else:
            ka = known_answer(sc, grads=not sc.get("forward_only", False))
            # the pixel-vectorised forward used for the big scenes is the same function
            a, b = forward64(sc), forward64v(sc)
            assert np.abs(a["color"] - b["color"]).max() < 1e-13 and np.array_equal(a["n_contrib"], b["n_contrib"]), n
            assert np.abs(a["final_T"] - b["final_T"]).max() < 1e-14, n
        names.append(n)
        for k, v in sc.items():
            if k in ("name", "forward_only", "big", "clamp_model", "sample") or v is None:
                continue
            blob[f"{n}/in/{k}"] = np.asarray(v)
        for k, v in ka.items():
            blob[f"{n}/out/{k}"] = np.asarray(v)
        print(n, "P", sc["means3D"].shape[0], "radii max", int(ka["radii"].max()), "min final_T", float(ka["final_T"].min()),
              "max n_contrib", int(ka["n_contrib"].max()), "terminated px", int(((ka["final_T"] < 1e-3)).sum()), flush=True)
    blob["names"] = np.asarray(names)
    np.savez_compressed(os.path.join(OUT, "known_answers.npz"), **blob)


if __name__ == "__main__":
    main()
