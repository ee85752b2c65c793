"""Known-answer vectors that neither the CPU oracle nor the HIP kernels produced (tests/golden/known_answers.npz,
written by tests/golden/make_known_answers.py): a float64 evaluation of the rasterizer's formulas for tiny analytic
scenes — SH colour through the reference's own eval_sh, cameras through its own graphics_utils — and CENTRAL FINITE
DIFFERENCES of it for every gradient, with the reference's two deliberate non-derivatives (guard-band stop-gradient,
NDC-scaled screen-space gradient) modelled in the differentiated forward.

Both implementations are held to them: the oracle here on the CPU, the HIP path under -m gpu.
Scenes: one Gaussian centred on a pixel; two at exactly the same depth + one behind + one culled + one off screen;
one guard-band-clamped; colors_precomp + an indefinite cov3D_precomp (power > 0 branch); eight stacked ones whose
centre pixels terminate at T < 1e-4; alpha clamped at 0.99 (forward only, and — the reference passes the gradient straight
through the clamp, backward.cu:499-534 — with gradients of the frozen-offset model); and "big_lists": 450 Gaussians on
3 x 3 reference tiles with lists of hundreds of entries per tile (cross-tile key order, lists longer than one 64-record
blend unit and than the 256-key register sort, pixels that terminate hundreds of entries deep), its image / transmittance /
radii / contributor counts and the finite-difference gradients of a sample of 36 of its Gaussians; and six "random_camera_k"
scenes (round 5): 70 - 110 anisotropic Gaussians from sub-pixel to a quarter of the image under a ROTATED, translated camera,
images that are no multiple of a tile (40 x 28 ... 64 x 36), SH degree 0 .. 3, Gaussians behind the camera and off screen, lists
of 40 - 70 entries, with the finite-difference gradients of twelve Gaussians each; and four "random_inputs_*" scenes of the same
family through the API's optional inputs: colors_precomp (scale_modifier 0.6), cov3D_precomp, both, and scale_modifier 1.5 with
near-opaque Gaussians (69 pixels terminate inside their list); three at the edges of the tiling (7 x 5, 16 x 16, 17 x 33) and
"random_big_600": 600 Gaussians on 64 x 64 under a general camera, lists of hundreds of entries in every tile, 4 031 of the 4 096
pixels terminating inside them; and "head_like_1500", the regime of BASELINE config 2 in small: 1 500 identical isotropic splats of
opacity 0.1 on an ellipsoid with the head template's bounding box under the benchmark's camera, 96 x 96, ~25 splats per pixel — and
"head_like_opaque_700", where training takes that scene: opacity 0.9, 515 of 4 096 pixels terminate after a handful of entries.  The modifier scenes found a third place where the reference's
backward is not the derivative of its forward: dL_dscales is the derivative with respect to scale_modifier * scale
(backward.cu:295,322-325) — the oracle, which restates backward.cu, and the plain finite difference differed by exactly the
modifier; the expected value is the finite difference divided by it."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "known_answers.npz")
Z = np.load(G)
NAMES = [str(n) for n in Z["names"]]
LONG_LISTS = ("big_lists", "random_big_600")   # lists of hundreds of entries per tile: looser bounds for fp32 sums of that length
GRADS = ["dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dsh", "dL_dcolors", "dL_dscales", "dL_drotations", "dL_dcov3D"]


def _scene(name):
    i = {k.split("/", 2)[2]: Z[k] for k in Z.files if k.startswith(name + "/in/")}
    o = {k.split("/", 2)[2]: Z[k] for k in Z.files if k.startswith(name + "/out/")}
    return i, o


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _compare(name, got_fwd, got_bwd, want, tol_img, tol_grad):
    color, final_T, n_contrib, radii = got_fwd
    np.testing.assert_array_equal(radii, want["radii"], err_msg=name)
    np.testing.assert_array_equal(n_contrib, want["n_contrib"], err_msg=name)
    assert np.abs(color - want["color"]).max() <= tol_img, (name, np.abs(color - want["color"]).max())
    assert np.abs(final_T - want["final_T"]).max() <= tol_img, name
    if got_bwd is None:
        assert "dL_dmeans3D" not in want
        return
    checked = 0
    rows = want.get("sample")     # gradients known for these Gaussians only (the big scene)
    for k in GRADS:
        if k not in want:
            continue
        w = want[k]
        g = np.asarray(got_bwd[k], np.float64)
        if rows is not None:
            g = g.reshape((g.shape[0], -1))[rows]
        g = g.reshape(w.shape)
        if np.abs(w).max() == 0:   # e.g. the rotation gradient of an isotropic Gaussian: exactly zero to first order
            assert np.abs(g).max() <= 1e-7, (name, k, np.abs(g).max())
            continue
        assert _rel_l2(g, w) <= tol_grad, (name, k, _rel_l2(g, w), g.ravel()[:6], w.ravel()[:6])
        # entry-wise too: relative to the largest entry of the same Gaussian's gradient
        assert np.abs(g - w).max() <= 5 * tol_grad * np.abs(w).max(), (name, k)
        checked += 1
    assert checked >= 5, (name, checked)


def _oracle_kwargs(i):
    g = lambda k: i[k] if k in i else None  # noqa: E731
    return dict(bg=i["bg"], means3D=i["means3D"], opacities=i["opacities"], viewmatrix=i["viewmatrix"],
                projmatrix=i["projmatrix"], campos=i["campos"], tanfovx=float(i["tanfovx"]), tanfovy=float(i["tanfovy"]),
                H=int(i["H"]), W=int(i["W"]), shs=g("shs"), sh_degree=int(i["D"]), colors_precomp=g("colors_precomp"),
                scales=g("scales"), rotations=g("rotations"), cov3D_precomp=g("cov3D_precomp"),
                scale_modifier=float(i["scale_modifier"]))


def test_known_answers_cover_the_branches():
    """The fixtures really exercise what they claim (so that a regenerated file cannot silently lose a case)."""
    i, o = _scene("terminates")
    assert int((o["n_contrib"] < 8).sum()) > 0 and float(o["final_T"].min()) < 2e-4     # early termination happened
    i, o = _scene("depth_tie")
    assert list(o["radii"][3:]) == [0, 0] and int(o["n_contrib"].max()) == 3
    assert np.abs(o["dL_dmeans3D"][3:]).max() == 0
    i, o = _scene("guard_band")
    tz = (np.c_[i["means3D"], np.ones(3)] @ i["viewmatrix"].astype(np.float64))
    assert abs(tz[0, 0] / tz[0, 2]) > 1.3 * float(i["tanfovx"]) and abs(tz[1, 1] / tz[1, 2]) > 1.3 * float(i["tanfovy"])
    i, o = _scene("alpha_clamp_forward_only")
    assert "dL_dmeans3D" not in o
    i, o = _scene("alpha_clamp_gradient")
    assert float(i["opacities"].max()) == 1.0 and np.abs(o["dL_dopacity"][1]).max() > 0
    i, o = _scene("big_lists")
    assert i["means3D"].shape[0] == 450 and (int(i["W"]), int(i["H"])) == (48, 48)          # 3 x 3 reference tiles
    assert int(o["n_contrib"].max()) > 256 and len(o["sample"]) == 36
    assert int((o["final_T"] < 2e-4).sum()) > 0                                              # pixels terminated deep in the list
    deep = o["n_contrib"][o["final_T"] < 2e-4]
    assert int(deep.min()) > 128, deep.min()
    degrees = set()
    for k in range(6):
        i, o = _scene(f"random_camera_{k}")
        rot = i["viewmatrix"][:3, :3].astype(np.float64)
        assert np.abs(rot - np.eye(3)).max() > 0.1 and abs(np.linalg.det(rot) - 1.0) < 1e-5        # a genuinely rotated view
        assert 70 <= i["means3D"].shape[0] <= 110 and len(o["sample"]) == 12 and int(o["n_contrib"].max()) >= 40
        assert int((o["radii"] == 0).sum()) >= 2                                                    # culled / off-screen Gaussians
        assert (int(i["W"]) % 16, int(i["H"]) % 16) != (0, 0)
        degrees.add(int(i["D"]))
    assert degrees == {0, 1, 2, 3}
    i, o = _scene("random_big_600")
    assert i["means3D"].shape[0] == 600 and int(o["n_contrib"].max()) > 200 and int((o["final_T"] < 1e-3).sum()) > 3500 and len(o["sample"]) == 16
    for n, wh in (("random_tiny_7x5", (7, 5)), ("random_one_tile_16x16", (16, 16)), ("random_past_a_tile_17x33", (17, 33))):
        i, o = _scene(n)
        assert (int(i["W"]), int(i["H"])) == wh and int((o["radii"] > 0).sum()) >= 6
    i, o = _scene("head_like_1500")
    assert i["means3D"].shape[0] == 1500 and float(i["opacities"].max()) == pytest.approx(0.1) and int((o["radii"] > 0).sum()) == 1500
    assert np.allclose(i["viewmatrix"][:3, :3], np.diag([1.0, -1.0, -1.0])) and int(o["n_contrib"].max()) > 200   # the benchmark's camera
    i, o = _scene("head_like_opaque_700")
    assert float(i["opacities"].min()) == pytest.approx(0.9) and int((o["final_T"] < 1e-3).sum()) > 400
    assert int(o["n_contrib"][o["final_T"] < 1e-3].min()) < 40 and int(o["n_contrib"].max()) > 200   # ... some of them early in lists of 200
    i, o = _scene("random_inputs_colors")
    assert "shs" not in i and "dL_dcolors" in o and "dL_dsh" not in o and float(i["scale_modifier"]) == pytest.approx(0.6)
    i, o = _scene("random_inputs_cov3d")
    assert "scales" not in i and "dL_dcov3D" in o and "dL_dscales" not in o
    i, o = _scene("random_inputs_both")
    assert "shs" not in i and "scales" not in i and {"dL_dcolors", "dL_dcov3D"} <= set(o)
    i, o = _scene("random_inputs_modifier_opaque")
    assert float(i["scale_modifier"]) == pytest.approx(1.5) and int((o["final_T"] < 1e-3).sum()) > 50 and np.abs(o["dL_dscales"]).max() > 0


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_known_answers(name):
    from oracle import oracle
    i, want = _scene(name)
    f = oracle.forward(**_oracle_kwargs(i))
    b = None
    if "dL_dmeans3D" in want:
        bb = oracle.backward(f, i["dL_dpix"])
        b = {k: getattr(bb, k) for k in GRADS}
    # (image / transmittance: 3e-6, or 1e-5 where a pixel blends more than 128 entries — an fp32 product and sum of that length;
    # gradients: 2e-5, or 1e-4 for the scenes whose every tile list holds hundreds of entries)
    deep = int(want["n_contrib"].max()) > 128
    _compare(name, (f.color, f.final_T, f.n_contrib.astype(np.int64), f.radii), b, want,
             tol_img=1e-5 if deep else 3e-6, tol_grad=1e-4 if name in LONG_LISTS else 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_matches_known_answers(name, gpu_device):
    import torch
    from fateavatar_amd import rasterizer
    i, want = _scene(name)
    t = lambda k: torch.from_numpy(np.ascontiguousarray(i[k], dtype=np.float32)).to(gpu_device) if k in i else torch.empty(0)  # noqa: E731
    H, W = int(i["H"]), int(i["W"])
    args = (t("bg"), t("means3D"), t("colors_precomp"), t("opacities"), t("scales"), t("rotations"), float(i["scale_modifier"]),
            t("cov3D_precomp"), t("viewmatrix"), t("projmatrix"), float(i["tanfovx"]), float(i["tanfovy"]), H, W, t("shs"),
            int(i["D"]), t("campos"), False, False)
    R, color, radii, geom, binning, img = rasterizer.rasterize_gaussians(*args)
    fT, _ = rasterizer.image_aux(img, H, W)
    # n_contrib in reference (16x16-list) semantics is not what the 8x8 implementation stores; the image, the
    # transmittance and the radii pin the same decisions
    b = None
    if "dL_dmeans3D" in want:
        out = rasterizer.rasterize_gaussians_backward(
            t("bg"), t("means3D"), radii, t("colors_precomp"), t("scales"), t("rotations"), float(i["scale_modifier"]),
            t("cov3D_precomp"), t("viewmatrix"), t("projmatrix"), float(i["tanfovx"]), float(i["tanfovy"]), t("dL_dpix"),
            t("shs"), int(i["D"]), t("campos"), geom, R, binning, img, False)
        names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
        b = {k: v.cpu().numpy() for k, v in zip(names, out)}
    if name in LONG_LISTS:   # the fixture is only worth its name if the HIP path really took its long-list machinery
        c = rasterizer.last_counts[gpu_device.index or 0]
        assert c.max_tile_list > (256 if name == "big_lists" else 128), c.max_tile_list   # (8 x 8 lists: several blend units each)
    deep = int(want["n_contrib"].max()) > 128
    _compare(name, (color.cpu().numpy(), fT.cpu().numpy(), want["n_contrib"], radii.cpu().numpy()), b, want,
             tol_img=2e-5 if deep else 1e-5, tol_grad=2e-4 if name in LONG_LISTS else 1e-4)
