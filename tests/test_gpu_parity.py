"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star: rendered RGBA and per-Gaussian grads within 1e-4 rel):
  * integer outputs (radii, num_rendered) and the per-Gaussian forward state (pixel centre, depth,
    conic, colour, Sigma3D): BIT-EXACT — the preprocess kernels run without FMA contraction in the
    reference's operation order;
  * blended outputs (RGB, final transmittance = the alpha channel): |d| <= 1e-5 + 1e-4*|ref| on
    >= 99.99 % of the pixels; the rest must be explainable as threshold flips (alpha < 1/255,
    T < 1e-4, power > 0), i.e. bounded by one splat's contribution;
  * gradients: |d| <= 1e-4*|ref| + 1e-6*max|ref| on >= 99.9 % of the entries and rel-L2 <= max(1e-4, 8 x the measured
    float-order / contraction floor)
    (float atomics make the reference itself order-dependent at the 1e-6 level).
"""
import numpy as np
import pytest

from fateavatar_amd import scenes
from tests import util

pytestmark = pytest.mark.gpu


def _scene_list():
    S = {}
    S["rand_deg0"] = scenes.random_scene(2000, 128, 128, sh_degree=0, seed=1)
    S["rand_deg1"] = scenes.random_scene(1500, 72, 56, sh_degree=1, seed=2, M=16, bg=(0.2, 0.5, 0.9))
    S["rand_deg2"] = scenes.random_scene(1500, 100, 130, sh_degree=2, seed=3, M=16, bg=(0.0, 0.0, 0.0))
    S["rand_deg3_behind"] = scenes.random_scene(3000, 128, 96, sh_degree=3, seed=4, behind_fraction=0.2,
                                                bg=(0.3, 0.1, 0.7))
    S["dense_opaque"] = scenes.random_scene(4000, 64, 64, sh_degree=0, seed=5, opacity_lo=0.6, opacity_hi=0.99,
                                            scale_lo=0.02, scale_hi=0.08)
    S["wide_offscreen"] = scenes.random_scene(2500, 96, 96, sh_degree=3, seed=6, spread=1.2, scale_lo=0.005,
                                              scale_hi=0.2, opacity_lo=0.001, opacity_hi=1.0)
    S["head_small"] = scenes.head_scene(P=20000, res=256, sh_degree=3, seed=0)
    return S


SCENES = _scene_list()


def _check_forward(o, h, name):
    assert h.num_rendered == o.num_rendered, name
    np.testing.assert_array_equal(h.radii.cpu().numpy(), o.radii)
    vis = o.radii > 0
    m2 = h.geometry(0, 2)
    np.testing.assert_array_equal(m2[vis], o.means2D[vis])
    np.testing.assert_array_equal(h.geometry(1, 1)[vis, 0], o.depths[vis])
    # conic + opacity: kept only inside the 48-byte blend record, the conic as (-0.5 a, -b, -0.5 c): exact scalings
    rec = h.geometry(8, 12)
    co = o.conic_opacity.astype(np.float32)
    want = np.stack([co[:, 0] * np.float32(-0.5), -co[:, 1], co[:, 2] * np.float32(-0.5), co[:, 3]], axis=1)
    np.testing.assert_array_equal(rec[vis, 2:6], want[vis])
    np.testing.assert_array_equal(h.geometry(3, 4)[vis, :3], o.rgb[vis] if o._inputs["colors_precomp"] is None
                                  else o._inputs["colors_precomp"][vis])
    # (Sigma3D itself is no longer stored: both per-Gaussian kernels compute it with one function, and the conic above,
    # bit-equal to the oracle's, is a function of it; the backward's copy is covered by the gradient comparisons)
    col = h.color.cpu().numpy()
    fT = h.final_T.cpu().numpy()
    fc = util.frac_close(col, o.color, 1e-4, 1e-5)
    ft = util.frac_close(fT, o.final_T, 1e-4, 1e-5)
    # (99.99 % of the values — but a single threshold flip must not fail an image of a few hundred pixels: one value
    # per colour plane may always be out; that it IS a flip is checked next)
    n_px = fT.size
    assert (fc >= 0.9999 or round((1.0 - fc) * col.size) <= 3) and (ft >= 0.9999 or round((1.0 - ft) * n_px) <= 1), (name, fc, ft)
    assert np.isfinite(col).all()
    # every pixel outside the tolerance must be EXPLAINED: some splat of its list sits on one of the reference's
    # three thresholds (power > 0, alpha < 1/255, T (1 - alpha) < 1e-4) within fp32 rounding (util.explain_pixel)
    n_out, unexplained = util.unexplained_outliers(o, col, fT)
    assert not unexplained, (name, n_out, unexplained[:5])
    # ... and a flip moves a pixel by at most one splat's contribution (alpha <= 0.99, colour <= ~2)
    assert np.abs(col - o.color).max() < 0.05, name


def _flip_affected_gaussians(o, h):
    """Gaussians whose gradient a threshold flip can legitimately change: everything in the 16x16-tile list of a flip pixel
    (util.flip_pixels; such pixels are checked to BE threshold flips by _check_forward).  A flip changes the transmittance
    of every later entry of that pixel, hence the gradients of all of them."""
    bad = _flip_pixels(o, h)
    mask = np.zeros(o.radii.shape[0], bool)
    if bad.any():
        H, W = bad.shape
        gx = (W + 15) // 16
        ys, xs = np.nonzero(bad)
        for t in np.unique((ys // 16) * gx + xs // 16):
            mask[o.point_list[int(o.ranges[t, 0]):int(o.ranges[t, 1])]] = True
    return mask, int(bad.sum())


SKIPPED = {}   # scene -> (flip pixels, rows exempted from the tight test, fraction of the visible rows): printed with -rP


# aggregate rel-L2 the flip-exempt rows of a scene are still held to (per array).  A flip changes what ONE pixel gives to every
# later entry of its list: of 12 160 fuzz configurations (tools/fuzz_parity.py, round 5) all but two stay below 7.2e-3 — 2.7e-2 on
# dL_dmeans2D for a scene whose opacities (0.004 .. 0.023) sit on the alpha >= 1/255 threshold itself, 6.5e-2 on dL_dcov3D for
# 40 146 sub-pixel Gaussians (scale 0.0003 .. 0.0015) of opacity 0.05 .. 0.12
SKIP_ROWS_BOUND = 1e-1
NOISE_K = 8.0   # the aggregate gradient bound is max(1e-4, NOISE_K x the float-order noise floor of the array)
ACHIEVED = {}   # name -> {array: (rel-L2 of the kept rows, floor, rel-L2 of the flip-exempt rows)}


def _rel_l2_kept(got, ref, keep):
    """L2 error of the rows without a threshold flip, relative to their norm — but to no less than 0.1 % of the whole
    array's: in a scene of image-sized splats nearly every Gaussian shares a pixel with a flip, and what is left are
    rows whose gradients are 1e-6 of the array's scale, i.e. fp32 noise of sums that cancel."""
    return float(np.linalg.norm((got[keep] - ref[keep]).astype(np.float64))
                 / max(np.linalg.norm(ref[keep].astype(np.float64)), 1e-3 * np.linalg.norm(ref.astype(np.float64)), 1e-300))


def _check_backward(o, h, dpix, name, max_skip_frac=0.02, agg_bound=1e-4):
    """Gradients against the oracle.  north_star's bound is 1e-4 relative; the reference itself is not reproducible to
    better than the order of its float atomics and the FMA contractions of its compiler, so the aggregate bound per array
    is max(agg_bound, NOISE_K x floor), where `floor` is MEASURED on this scene: the oracle's backward with float sums in
    seeded orders, and with nvcc-style contractions (two results the reference could produce), against its double sums.
    At every BASELINE configuration the floor is ~1e-7, the bound 1e-4, and the HIP path sits at 3e-7 .. 9e-7.
    `agg_bound`: 1e-4 for every scene (round 4 let scenes of image-sized splats state 3e-4: the blend backward then summed a
    Gaussian's q dx and q dy per tile and combined them with the conic afterwards; it combines per pixel now, as
    backward.cu:540-546 does — test_fuzz_regression_image_sized_splats).
    `max_skip_frac`: the largest share of the VISIBLE Gaussians that threshold flips may exempt from the tight test
    (they are still held to SKIP_ROWS_BOUND in aggregate, and what they achieve is printed).  A flip in a 16x16 list of thousands of entries
    exempts thousands of rows, so dense stress scenes state a larger bound — but every scene states one; a scene whose flips would
    exempt more is checked by `_check_backward_no_exemptions` instead: the flips masked out of dL/dpixel, every row held to the
    tight test."""
    from oracle import oracle
    skip, n_flips = _flip_affected_gaussians(o, h)
    # (flips are rare events of the (pixel, entry) tests: one in 1e4 pixels — or, where every pixel walks thousands of entries,
    # one in 1e7 tests: 23 flip pixels of 220 350 in a fuzz scene of 50 530 image-sized splats whose opacities, 0.029 .. 0.031, put
    # a threshold ring inside every footprint; never fewer than three: 2 of 16 440 pixels in a fuzz scene of opaque splats.  Each
    # of them is checked to BE a flip by _check_forward)
    assert n_flips <= max(3, int(1e-4 * dpix.shape[1] * dpix.shape[2]), int(1e-7 * 256 * o.num_rendered)), (name, n_flips)
    n_vis = max(1, int((o.radii > 0).sum()))
    frac = float(skip.sum()) / n_vis
    if frac > max_skip_frac and not isinstance(h, _Agree):
        # more rows than this scene may hide behind an exemption: no row is exempt then (the flips are masked out of dL/dpixel)
        print(f"[skipped rows] {name}: {n_flips} flip pixel(s) would exempt {frac:.2%} > {max_skip_frac:.2%} of the visible Gaussians")
        return _check_backward_no_exemptions(o, h, dpix, name)
    ob = oracle.backward(o, dpix)
    floors = []
    for seed, contract in ((1, False), (2, True)):
        oracle.set_bwd_float_order(seed)
        oracle.set_bwd_contract(contract)
        try:
            floors.append(oracle.backward(o, dpix))
        finally:
            oracle.set_bwd_float_order(0)
            oracle.set_bwd_contract(False)
    hb = h.backward(dpix)
    SKIPPED[name] = (n_flips, int(skip.sum()), round(frac, 5))
    print(f"[skipped rows] {name}: {n_flips} flip pixel(s) exempt {int(skip.sum())} of {n_vis} visible Gaussians ({frac:.4%}) from the tight test")
    keep = ~skip
    ACHIEVED[name] = {}
    for k in ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
              "dL_drotations"]:
        ref, got = getattr(ob, k), hb[k]
        assert got.shape == ref.shape, (name, k, got.shape, ref.shape)
        assert np.isfinite(got).all(), (name, k)
        if ref.size == 0:   # (dL_dsh under colors_precomp: no coefficients were passed)
            continue
        scale = np.abs(ref).max()
        if scale == 0:
            assert np.abs(got).max() == 0, (name, k)
            continue
        # (absolute floor 5 ppm of the largest entry: a splat that covers thousands of pixels sums thousands of signed
        # terms to a small net gradient, and the two implementations add them in different orders)
        fr = util.frac_close(got[keep], ref[keep], 1e-4, 5e-6 * scale)
        rl = _rel_l2_kept(got, ref, keep)
        floor = max(_rel_l2_kept(getattr(f, k), ref, keep) for f in floors)
        bound = max(agg_bound, NOISE_K * floor)
        rl_skip = util.rel_l2(got[skip], ref[skip]) if skip.any() else 0.0
        ACHIEVED[name][k] = (rl, floor, rl_skip)
        print(f"[gradient] {name} {k}: rel-L2 {rl:.2e} (float-order floor {floor:.1e}, bound {bound:.1e}); flip-exempt rows {rl_skip:.2e}")
        # (0.1 % of the entries, but never fewer than three: a scene of 100 Gaussians has 400 quaternion entries, and an
        # entry that is the small difference of large terms misses a 1e-4 relative test in fp32 either way)
        n_off = round((1.0 - fr) * got[keep].size)
        assert (fr >= 0.999 or n_off <= 3) and rl <= bound, (name, k, fr, rl, floor, bound, n_flips, int(skip.sum()))
        # Gaussians that share a pixel with a threshold flip (a flip changes the transmittance of every later entry of that
        # ONE pixel; a row sums over all its pixels): within a few per cent in aggregate, not merely "same sign and size"
        if skip.any():
            assert rl_skip <= SKIP_ROWS_BOUND, (name, k, "flip-affected rows", rl_skip)
    # (last, so that a real mismatch is reported as such and not as a scene that exempts too much)
    assert frac <= max_skip_frac, (name, "threshold flips exempt too many rows from the tight gradient test", n_flips, int(skip.sum()), n_vis, frac)


@pytest.mark.parametrize("name", list(SCENES))
def test_forward_backward_vs_oracle(name, gpu_device):
    s = SCENES[name]
    o = util.oracle_forward(s)
    h = util.HipFrame(s, gpu_device)
    _check_forward(o, h, name)
    rng = np.random.default_rng(11)
    H, W = s.camera.image_height, s.camera.image_width
    dpix = (rng.uniform(-1, 1, (3, H, W)) / (H * W)).astype(np.float32)
    _check_backward(o, h, dpix, name)


def _flip_pixels(o, h):
    """util.flip_pixels: outside the forward tolerance, or T_final off by more than 0.1 % relative."""
    return util.flip_pixels(o, h.color.cpu().numpy(), h.final_T.cpu().numpy())


@pytest.mark.parametrize("name", ["dense_opaque", "wide_offscreen"])
def test_backward_without_exemptions_on_dense_scenes(name, gpu_device):
    """The two scenes whose threshold flips exempt the most rows, with NO row exempt: the upstream gradient is zero at the
    pixels whose forward value flipped (a pixel with dL/dpixel = 0 contributes nothing to any gradient, whatever its
    blend sequence was), so every Gaussian's gradient must pass the tight test — a regression confined to a few tile
    lists cannot hide behind an exemption here."""
    s = SCENES[name]
    o = util.oracle_forward(s)
    h = util.HipFrame(s, gpu_device)
    _check_forward(o, h, name)
    H, W = s.camera.image_height, s.camera.image_width
    dpix = (np.random.default_rng(13).uniform(-1, 1, (3, H, W)) / (H * W)).astype(np.float32)
    _check_backward_no_exemptions(o, h, dpix, name)


class _Agree:
    """A frame whose forward outputs are the oracle's: `_check_backward` then finds no flip pixel and exempts no row."""

    def __init__(self, h, o):
        import torch
        self._h, self.color, self.final_T = h, torch.from_numpy(o.color.copy()), torch.from_numpy(o.final_T.copy())

    def backward(self, d):
        return self._h.backward(d)


def _check_backward_no_exemptions(o, h, dpix, name):
    """`_check_backward` with NO row exempt: dL/dpixel is zeroed at the pixels whose forward value flipped (such a pixel then
    contributes nothing to any gradient, whatever its blend sequence was), and every Gaussian is held to the tight test."""
    bad = _flip_pixels(o, h)
    dpix = dpix.copy()
    dpix[:, bad] = 0.0
    print(f"[no exemptions] {name}: {int(bad.sum())} flip pixel(s) masked out of dL/dpixel")
    _check_backward(o, _Agree(h, o), dpix, name + "-no-exemptions", max_skip_frac=0.0)
    assert SKIPPED[name + "-no-exemptions"][1] == 0


def _check_backward_capped(o, h, dpix, name, max_skip_frac):
    """`_check_backward` with the scene's cap stated: threshold flips exempt at most `max_skip_frac` of the visible rows —
    otherwise (scenes of image-sized splats: one flipped pixel's 16x16 list holds a third of the scene) the flips are masked
    out of dL/dpixel and NO row is exempt.  Either way no scene hides more than `max_skip_frac` of its rows behind an exemption."""
    _check_backward(o, h, dpix, name, max_skip_frac=max_skip_frac)


def test_colors_precomp_and_cov3d_precomp(gpu_device):
    s = SCENES["rand_deg0"]
    rng = np.random.default_rng(5)
    cols = rng.uniform(0, 1, (s.P, 3)).astype(np.float32)
    o0 = util.oracle_forward(s)
    cov = o0.cov3D.copy()
    o = util.oracle_forward(s, colors_precomp=cols, cov3D_precomp=cov)
    h = util.HipFrame(s, gpu_device, colors_precomp=cols, cov3D_precomp=cov)
    _check_forward(o, h, "precomp")
    H, W = s.camera.image_height, s.camera.image_width
    dpix = (rng.uniform(-1, 1, (3, H, W)) / (H * W)).astype(np.float32)
    from oracle import oracle
    ob = oracle.backward(o, dpix)
    hb = h.backward(dpix)
    for k in ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D"]:
        ref, got = getattr(ob, k), hb[k]
        scale = np.abs(ref).max()
        assert util.frac_close(got, ref, 1e-4, 1e-6 * scale) >= 0.999, k
        assert util.rel_l2(got, ref) <= 1e-4, k
    assert np.abs(hb["dL_dscales"]).max() == 0 and np.abs(hb["dL_drotations"]).max() == 0


def test_scale_modifier_and_debug_mode(gpu_device):
    s = SCENES["rand_deg1"]
    o = util.oracle_forward(s, scale_modifier=0.6)
    h = util.HipFrame(s, gpu_device, scale_modifier=0.6, debug=True)
    _check_forward(o, h, "scale_modifier")


def test_binning_capacity_regrow(gpu_device):
    """A capacity that is too small must be reported and the retry must give the same image."""
    from fateavatar_amd import rasterizer
    s = SCENES["dense_opaque"]
    rasterizer._capacity_hint.clear()
    h1 = util.HipFrame(s, gpu_device)
    need = h1.counts.num_instances
    assert need > 0
    import ctypes as C
    import torch
    from fateavatar_amd import _lib
    L = _lib.lib()
    # call the ABI directly with half the needed capacity
    cap = max(1, need // 2)
    c = s.camera
    binning = torch.empty((L.fr_binning_bytes(cap, c.image_width, c.image_height),), dtype=torch.uint8, device=gpu_device)
    out = torch.empty_like(h1.color)
    radii = torch.empty_like(h1.radii)
    geom = torch.empty_like(h1.geom)
    img = torch.empty_like(h1.img)
    prm = rasterizer._params(s.P, s.sh_degree, s.shs.shape[1], c.image_width, c.image_height, c.tanfovx, c.tanfovy,
                             1.0, False, False)
    inp = rasterizer._inputs(h1.bg, h1.means3D, h1.sh, None, h1.op, h1.scales, h1.rots, None, h1.view, h1.proj,
                             h1.campos)
    counts = _lib.fr_counts()
    rc = L.fr_forward(_lib.handle(0), C.byref(prm), C.byref(inp), out.data_ptr(), radii.data_ptr(), geom.data_ptr(),
                      img.data_ptr(), binning.data_ptr(), cap, C.byref(counts), torch.cuda.current_stream().cuda_stream)
    assert rc == _lib.FR_ERR_BINNING_CAPACITY
    assert counts.num_instances == need and counts.overflow == 1
    torch.cuda.synchronize()


@pytest.mark.parametrize("P,lo,hi", [(700, 256, 1024), (1400, 1024, 2048), (3000, 2048, 4096), (6000, 4096, 1 << 30)])
def test_long_tile_lists_take_the_multi_wave_and_fallback_sorts(P, lo, hi, gpu_device):
    """Many Gaussians on one 8x8 tile: the 4-wave register sorts (<= 1024, <= 2048 in k_tile_sort, <= 4096 keys in the big-list
    sorter) and the global-memory
    fallback (> 4096) must give the same blend order as the oracle (equal depths included: ties break by id)."""
    rng = np.random.default_rng(3)
    s = scenes.random_scene(P, 32, 32, sh_degree=0, seed=9, spread=0.004, scale_lo=0.002, scale_hi=0.004,
                            opacity_lo=0.02, opacity_hi=0.05)
    s.means3D[:, 2] = 1.0 + rng.uniform(0, 0.5, s.P).astype(np.float32)
    s.means3D[::7, 2] = 1.25  # exact depth ties
    o = util.oracle_forward(s)
    h = util.HipFrame(s, gpu_device)
    assert lo < h.counts.max_tile_list <= hi, h.counts.max_tile_list
    _check_forward(o, h, "long_lists")
    dpix = (rng.uniform(-1, 1, (3, 32, 32)) / (32 * 32)).astype(np.float32)
    from oracle import oracle
    ob, hb = oracle.backward(o, dpix), h.backward(dpix)
    assert util.rel_l2(hb["dL_dmeans2D"], ob.dL_dmeans2D) < 1e-4
    assert util.rel_l2(hb["dL_dopacity"], ob.dL_dopacity) < 1e-4


def test_large_non_square_image(gpu_device):
    """1536 x 1000 pixels = 192 x 125 tiles: the totals kernel's multi-workgroup path, partial edge tiles in y,
    and the XCD-private counter pitch for a tile count that is not a multiple of 16."""
    s = scenes.random_scene(5000, 1000, 1536, sh_degree=1, seed=21, spread=0.28, scale_lo=0.004, scale_hi=0.03,
                            opacity_lo=0.2, opacity_hi=0.9)
    o = util.oracle_forward(s)
    h = util.HipFrame(s, gpu_device)
    _check_forward(o, h, "large_image")
    rng = np.random.default_rng(2)
    dpix = (rng.uniform(-1, 1, (3, 1000, 1536)) / (1000 * 1536)).astype(np.float32)
    from oracle import oracle
    ob, hb = oracle.backward(o, dpix), h.backward(dpix)
    assert util.rel_l2(hb["dL_dmeans2D"], ob.dL_dmeans2D) < 1e-4
    assert util.rel_l2(hb["dL_dsh"], ob.dL_dsh) < 1e-4


def test_mark_visible(gpu_device):
    import torch
    from fateavatar_amd import rasterizer
    from oracle import oracle
    s = SCENES["rand_deg3_behind"]
    c = s.camera
    got = rasterizer.mark_visible(torch.from_numpy(s.means3D).to(gpu_device),
                                  torch.from_numpy(c.world_view_transform).to(gpu_device),
                                  torch.from_numpy(c.full_proj_transform).to(gpu_device)).cpu().numpy()
    ref = oracle.mark_visible(s.means3D, c.world_view_transform, c.full_proj_transform)
    np.testing.assert_array_equal(got, ref)
    assert 0 < ref.sum() < s.P


def test_reduce_scatter_selftest(gpu_device):
    """The 36-value wave reduce-scatter of the blend backward, in isolation."""
    import torch
    from fateavatar_amd import _lib
    rng = np.random.default_rng(0)
    x = rng.normal(size=(64, 36)).astype(np.float32)
    tin = torch.from_numpy(x).to(gpu_device)
    tout = torch.zeros(64, dtype=torch.float32, device=gpu_device)
    rc = _lib.lib().fr_debug_selftest_reduce(tin.data_ptr(), tout.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    got = tout.cpu().numpy()
    ref = x.astype(np.float64).sum(0)
    for l in range(64):
        v = int(format(l, "06b")[::-1], 2)
        if v < 36:
            assert abs(got[l] - ref[v]) <= 1e-4 * max(1.0, abs(ref[v])), (l, v, got[l], ref[v])


def test_knn_bit_exact_vs_oracle(gpu_device):
    import torch
    from oracle import oracle
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(2)
    verts, faces, _ = scenes.head_geometry()
    for pts in (scenes.sample_mesh(verts, faces, 20000, seed=3), rng.normal(size=(5000, 3)).astype(np.float32),
                rng.uniform(-1, 1, (37, 3)).astype(np.float32),
                np.repeat(rng.normal(size=(50, 3)).astype(np.float32), 5, axis=0)):  # exact duplicates
        got = distCUDA2(torch.from_numpy(pts).to(gpu_device)).cpu().numpy()
        ref = oracle.knn_mean_dist2(pts)
        np.testing.assert_array_equal(got, ref)


def test_nearest_neighbour_distance_and_init_scale(gpu_device):
    """fr_knn_nearest_dist2 == brute-force nearest-neighbour distance; init_scale_by_knn reproduces the config-2 spacing
    SURVEY.md §8d measured with the reference's pytorch3d path (6.085e-4 on 100 k template samples)."""
    import torch
    from scipy.spatial import cKDTree
    from fateavatar_amd.knn import init_scale_by_knn, nearest_dist2
    rng = np.random.default_rng(5)
    pts = rng.normal(size=(4000, 3)).astype(np.float32)
    got = nearest_dist2(torch.from_numpy(pts).to(gpu_device)).cpu().numpy()
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=2)
    assert np.allclose(got, d[:, 1] ** 2, rtol=1e-4, atol=1e-12)
    verts, faces, from_fixture = scenes.head_geometry()
    mean_s, max_s, log_s = init_scale_by_knn(torch.from_numpy(scenes.sample_mesh(verts, faces, 100_000, seed=0)).to(gpu_device))
    assert abs(float(max_s) - 10 * float(mean_s)) < 1e-9 and abs(float(log_s) - np.log(float(mean_s))) < 1e-5
    if from_fixture == "head_template":
        assert abs(float(mean_s) - 6.085e-4) < 0.05e-4, float(mean_s)


def test_render_api_autograd_path(gpu_device):
    """render() + autograd down to the raw parameters == oracle grads chained through the activations."""
    import torch
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    from oracle import oracle
    s = scenes.head_scene(P=6000, res=96, sh_degree=3, seed=1)
    pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 3, gpu_device)
    cam = TorchCamera(s.camera, gpu_device)
    out = render(cam, pc, torch.from_numpy(s.bg).to(gpu_device))
    img = out["render"]
    rng = np.random.default_rng(0)
    w = (rng.uniform(-1, 1, img.shape) / img.numel()).astype(np.float32)
    (img * torch.from_numpy(w).to(gpu_device)).sum().backward()
    c = s.camera
    f = oracle.forward(bg=s.bg, means3D=s.means3D, opacities=pc.get_opacity.detach().cpu().numpy(),
                       viewmatrix=c.world_view_transform, projmatrix=c.full_proj_transform, campos=c.camera_center,
                       tanfovx=c.tanfovx, tanfovy=c.tanfovy, H=96, W=96, shs=s.shs, sh_degree=3,
                       scales=pc.get_scaling.detach().cpu().numpy(), rotations=pc.get_rotation.detach().cpu().numpy())
    b = oracle.backward(f, w)
    assert util.frac_close(img.detach().cpu().numpy(), f.color, 1e-4, 1e-5) >= 0.9999
    assert np.array_equal(out["visibility_filter"].cpu().numpy(), f.radii > 0)
    assert util.rel_l2(out["viewspace_points"].grad.cpu().numpy(), b.dL_dmeans2D) < 1e-4
    assert util.rel_l2(pc.grad_of("_xyz").cpu().numpy(), b.dL_dmeans3D) < 1e-4
    assert util.rel_l2(pc.grad_of("_features").cpu().numpy(), b.dL_dsh) < 1e-4
    # chain rule through sigmoid / exp in stock PyTorch
    op = pc.get_opacity.detach().cpu().numpy()
    assert util.rel_l2(pc.grad_of("_opacity").cpu().numpy(), b.dL_dopacity * op * (1 - op)) < 1e-4
    sc = pc.get_scaling.detach().cpu().numpy()
    assert util.rel_l2(pc.grad_of("_scaling").cpu().numpy(), b.dL_dscales * sc) < 1e-4


def test_fused_visibility_mask_and_densification_stats(gpu_device):
    """fr_aux: the bool mask written by the preprocess kernel == radii > 0, and the statistics the backward kernel
    accumulates == _add_densification_stats (model/fateavatar.py:734-737) done with torch on the same gradients."""
    import torch
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    P = 8000
    s0 = scenes.head_scene(P=P, res=128, sh_degree=1, seed=2)
    s0.means3D[::5, 0] += 50.0  # every fifth Gaussian far off screen: culled, radii == 0
    pc = FlatGaussians(s0.means3D, s0.shs, s0.opacities, s0.scales, s0.rotations, 1, gpu_device)
    accum = torch.zeros((P, 1), device=gpu_device)
    denom = torch.zeros((P, 1), device=gpu_device)
    ref_accum, ref_denom = torch.zeros_like(accum), torch.zeros_like(denom)
    pc.fused_densification_stats = (accum, denom)
    rng = np.random.default_rng(5)
    for view in range(3):
        s = scenes.head_scene(P=P, res=128, sh_degree=1, seed=2, view=view, n_views=3)
        out = render(TorchCamera(s.camera, gpu_device), pc, torch.from_numpy(s.bg).to(gpu_device))
        w = torch.from_numpy((rng.uniform(-1, 1, (3, 128, 128)) / (128 * 128)).astype(np.float32)).to(gpu_device)
        pc.begin_step()
        torch.autograd.backward(out["render"], grad_tensors=w)
        vis = out["visibility_filter"]
        assert vis.dtype == torch.bool and torch.equal(vis, out["radii"] > 0) and 0 < int(vis.sum()) < P
        g = out["viewspace_points"].grad
        ref_accum[vis] += torch.norm(g[vis, :2], dim=-1, keepdim=True)
        ref_denom[vis] += 1
    assert torch.equal(denom, ref_denom) and float(denom.max()) == 3.0
    assert torch.allclose(accum, ref_accum, rtol=1e-5, atol=0.0) and float(accum.max()) > 0


def test_views_of_a_batch_must_not_share_an_overflow_word(gpu_device):
    """fr_aux::overflow_out is overwritten (0 or 1) by every backward: two views of ONE batched launch that shared the word
    would race, and a view that did not overflow could clear the flag of one that did.  The batched backward refuses a
    shared word; a word per view (or none) is fine."""
    import torch
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render_batch
    P = 2000
    s = [scenes.head_scene(P=P, res=96, sh_degree=1, seed=2, view=v, n_views=2) for v in range(2)]
    cams = [TorchCamera(x.camera, gpu_device) for x in s]
    bg = torch.from_numpy(s[0].bg).to(gpu_device)
    w = torch.ones((3, 96, 96), device=gpu_device) / (3 * 96 * 96)

    def holders(words):
        out = []
        for k in range(2):
            pc = FlatGaussians(s[0].means3D, s[0].shs, s[0].opacities, s[0].scales, s[0].rotations, 1, gpu_device)
            pc.fused_densification_stats = (torch.zeros((P, 1), device=gpu_device), torch.zeros((P, 1), device=gpu_device), words[k])
            out.append(pc)
        return out

    shared = torch.zeros(1, device=gpu_device)
    outs = render_batch(cams, holders([shared, shared]), bg)
    with pytest.raises(RuntimeError, match="ONE overflow word EACH"):
        torch.autograd.backward([o["render"] for o in outs], grad_tensors=[w, w])
    own = [torch.zeros(1, device=gpu_device), torch.zeros(1, device=gpu_device)]
    pcs = holders(own)
    outs = render_batch(cams, pcs, bg)
    torch.autograd.backward([o["render"] for o in outs], grad_tensors=[w, w])
    torch.cuda.synchronize()
    assert float(own[0]) == 0.0 and float(own[1]) == 0.0
    assert all(float(pc.fused_densification_stats[1].max()) == 1.0 for pc in pcs)


def test_fused_activations_match_torch_activations(gpu_device):
    """render() with raw parameters + in-kernel sigmoid/exp/normalize == render() with PyTorch activations
    (which the oracle tests pin), forward and every raw-parameter gradient."""
    import torch
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    s = scenes.random_scene(3000, 96, 80, sh_degree=2, seed=12, M=16, scale_lo=0.01, scale_hi=0.06, opacity_lo=0.05,
                            opacity_hi=0.95, bg=(0.1, 0.6, 0.3))
    rng = np.random.default_rng(4)
    rot_raw = (s.rotations * rng.uniform(0.3, 3.0, (s.P, 1))).astype(np.float32)  # un-normalised raw quaternions
    cam = TorchCamera(s.camera, gpu_device)
    bg = torch.from_numpy(s.bg).to(gpu_device)
    w = torch.from_numpy((rng.uniform(-1, 1, (3, 96, 80)) / (3 * 96 * 80)).astype(np.float32)).to(gpu_device)
    res = {}
    for fused in (False, True):
        pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, rot_raw, 2, gpu_device, fused_activations=fused)
        out = render(cam, pc, bg)
        torch.autograd.backward(out["render"], grad_tensors=w)
        res[fused] = dict(img=out["render"].detach().cpu().numpy(), radii=out["radii"].cpu().numpy(),
                          g2=out["viewspace_points"].grad.cpu().numpy(),
                          **{n: pc.grad_of(n).cpu().numpy() for n, _ in pc.FIELDS})
    a, b = res[False], res[True]
    assert np.mean(a["radii"] == b["radii"]) > 0.999
    assert util.frac_close(b["img"], a["img"], 1e-4, 1e-5) >= 0.9999
    for k in ["g2", "_xyz", "_features", "_opacity", "_scaling", "_rotation"]:
        assert util.rel_l2(b[k], a[k]) < 1e-4, (k, util.rel_l2(b[k], a[k]))
        assert np.abs(a[k]).max() > 0


def test_graph_replay_survives_interleaved_eager_kernels(gpu_device):
    """A frame captured in a HIP graph (FR_FLAG_NO_WAIT) must give the same image and counts on every replay,
    also when eager kernels run between replays (regression: hipMemsetAsync as a graph node did not)."""
    import torch
    from fateavatar_amd import rasterizer
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    s = scenes.head_scene(P=20000, res=256, sh_degree=3, seed=0)
    pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 3, gpu_device, fused_activations=True)
    cam = TorchCamera(s.camera, gpu_device)
    bg = torch.from_numpy(s.bg).to(gpu_device)
    g = torch.rand(3, 256, 256, device=gpu_device) / (3 * 256 * 256)
    static_img = torch.zeros(3, 256, 256, device=gpu_device)

    def frame():
        pc.begin_step()
        out = render(cam, pc, bg)
        torch.autograd.backward(out["render"], grad_tensors=g)
        static_img.copy_(out["render"].detach())

    for _ in range(3):
        frame()
    torch.cuda.synchronize()
    ref_img = static_img.clone()
    ref_grad = pc.collect_grads().clone()
    ref_counts = rasterizer.last_counts[0]
    try:
        rasterizer.set_no_wait(True)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            frame()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            frame()
        other = torch.ones(4096, device=gpu_device)
        for i in range(6):
            graph.replay()
            other.mul_(1.5)               # eager kernels between replays
            pc.collect_grads().mul_(1.0)
        torch.cuda.synchronize()
        c = rasterizer.read_counts(0)
        assert not rasterizer.check_async_overflow(0)
    finally:
        rasterizer.set_no_wait(False)
    assert (c.num_instances, c.num_rendered, c.max_tile_list) == (ref_counts.num_instances, ref_counts.num_rendered,
                                                                  ref_counts.max_tile_list)
    assert torch.equal(static_img, ref_img)
    assert util.rel_l2(pc.flat_grad.cpu().numpy(), ref_grad.cpu().numpy()) < 1e-5


def test_graph_replay_survives_an_eager_frame_that_outgrows_the_handle_buffers(gpu_device):
    """A captured frame bakes the handle's buffer pointers and bucket capacity into its kernels.  An eager frame on the
    SAME handle that needs bigger key buckets / more accumulator rows (a denser pose, a bigger model) must not free what
    the graph replays on: outgrown buffers of a handle that has been captured are retired until fr_destroy
    (fr_handle_impl::captured).  Runs in a subprocess with its own handle state."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "from fateavatar_amd import scenes, rasterizer\n"
        "from fateavatar_amd.model import FlatGaussians, TorchCamera\n"
        "from fateavatar_amd.render import render\n"
        "dev = torch.device('cuda:0')\n"
        "def setup(P, scale, seed):\n"
        "    s = scenes.head_scene(P=P, res=256, sh_degree=1, seed=seed, scale=scale)\n"
        "    pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 1, dev, fused_activations=True)\n"
        "    return pc, TorchCamera(s.camera, dev), torch.from_numpy(s.bg).to(dev)\n"
        "pc, cam, bg = setup(20000, None, 0)\n"
        "g = torch.rand(3, 256, 256, device=dev) / (3 * 256 * 256)\n"
        "img = torch.zeros(3, 256, 256, device=dev)\n"
        "def frame():\n"
        "    pc.begin_step(); out = render(cam, pc, bg)\n"
        "    torch.autograd.backward(out['render'], grad_tensors=g); img.copy_(out['render'].detach())\n"
        "for _ in range(2): frame()\n"
        "torch.cuda.synchronize(); ref_img = img.clone(); ref_grad = pc.collect_grads().clone()\n"
        "rasterizer.set_no_wait(True)\n"
        "side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())\n"
        "with torch.cuda.stream(side): frame()\n"
        "torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()\n"
        "graph = torch.cuda.CUDAGraph()\n"
        "with torch.cuda.graph(graph): frame()\n"
        "graph.replay(); torch.cuda.synchronize()\n"
        "assert torch.equal(img, ref_img)\n"
        "rasterizer.set_no_wait(False)\n"
        "# an eager frame of a bigger, much denser model on the same handle: more accumulator rows, longer (tile, XCD) lists\n"
        "pc2, cam2, bg2 = setup(60000, 6e-3, 1)\n"
        "for _ in range(3):\n"
        "    out2 = render(cam2, pc2, bg2); out2['render'].sum().backward(); pc2.begin_step()\n"
        "torch.cuda.synchronize()\n"
        "big = rasterizer.last_counts[0]\n"
        "assert big.max_tile_list > 512, big.max_tile_list\n"
        "junk = [torch.full((1 << 22,), float('nan'), device=dev) for _ in range(8)]   # whatever was freed gets reused\n"
        "rasterizer.set_no_wait(True)\n"
        "for _ in range(3): graph.replay()\n"
        "torch.cuda.synchronize()\n"
        "assert not rasterizer.check_async_overflow(0)\n"
        "rasterizer.set_no_wait(False)\n"
        "assert torch.equal(img, ref_img), float((img - ref_img).abs().max())\n"
        "d = (pc.flat_grad - ref_grad).norm() / ref_grad.norm()\n"
        "assert float(d) < 1e-5, float(d)\n"
        "print('replay-ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "replay-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_capturing_a_size_the_handle_has_not_seen_is_refused_cleanly(gpu_device):
    """A handle grows its own buffers (accumulators, tile counters, key buckets) with the scene and the tile grid; that
    needs hipMalloc/hipFree, which a capturing stream does not allow.  fr_forward says so BEFORE touching the stream
    (FR_ERR_UNSUPPORTED), the capture ends without a HIP error, and after one eager frame the same capture works."""
    import torch
    from fateavatar_amd import rasterizer
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    s = scenes.head_scene(P=6000, res=200, sh_degree=1, seed=3)
    pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 1, gpu_device, fused_activations=True)
    cam = TorchCamera(s.camera, gpu_device)
    bg = torch.from_numpy(s.bg).to(gpu_device)
    keep = torch.zeros(3, 200, 200, device=gpu_device)

    def frame():
        with torch.no_grad():
            keep.copy_(render(cam, pc, bg)["render"])

    with rasterizer.handle_slot(0):   # sizes the torch-side buffers and the capacity hint
        frame()
        torch.cuda.synchronize()
        want = keep.clone()
    side = torch.cuda.Stream()
    with rasterizer.handle_slot(12), rasterizer.no_wait():   # a handle that does not exist yet: the wrapper refuses
        graph = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="eager frame"):
            with torch.cuda.graph(graph, stream=side):
                frame()
    from fateavatar_amd import _lib
    _lib.handle(0, 11)
    with rasterizer.handle_slot(11), rasterizer.no_wait():   # one that exists but has not seen a frame: fr_forward does
        graph = torch.cuda.CUDAGraph()
        with pytest.raises(RuntimeError, match="eager frame of this size"):
            with torch.cuda.graph(graph, stream=side):
                frame()
        torch.cuda.synchronize()   # (no sticky HIP error)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            frame()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(keep, want)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            frame()
        keep.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(keep, want)
        assert not rasterizer.read_counts(0).overflow


def test_fused_adam_matches_torch_adam(gpu_device):
    """fr_adam_step over a flat buffer with per-segment (and two-rate) learning rates == torch.optim.Adam over the
    equivalent parameter groups (train/optim.py:11-37), step by step, including the bias correction."""
    import torch
    from fateavatar_amd.optim import FusedAdam
    torch.manual_seed(0)
    P, M = 1001, 4  # 1001 * (3 + 12 + 1) = 16016 elements; segment boundaries not multiples of 4
    sizes = [P * 3, P * M * 3, P]
    n = sum(sizes)
    flat = torch.randn(n, device=gpu_device)
    grad = torch.zeros_like(flat)
    lrs = dict(xyz=1.6e-4, dc=2.5e-3, rest=2.5e-3 / 20, op=0.05)
    opt = FusedAdam(flat, grad, [(sizes[0], lrs["xyz"]), (sizes[1], lrs["dc"], M * 3, 3, lrs["rest"]), (sizes[2], lrs["op"])],
                    grad_scale=0.5)
    o0, o1 = sizes[0], sizes[0] + sizes[1]
    feat = flat[o0:o1].view(P, M * 3)
    ref = [flat[:o0].clone().requires_grad_(), feat[:, :3].clone().requires_grad_(), feat[:, 3:].clone().requires_grad_(),
           flat[o1:].clone().requires_grad_()]
    topt = torch.optim.Adam([dict(params=[ref[0]], lr=lrs["xyz"]), dict(params=[ref[1]], lr=lrs["dc"]),
                             dict(params=[ref[2]], lr=lrs["rest"]), dict(params=[ref[3]], lr=lrs["op"])], lr=0.0)
    for step in range(6):
        g = torch.randn(n, device=gpu_device) * (10.0 ** (step - 3))
        g[::7] = 0.0
        grad.copy_(g)
        gs = 0.5 * g  # grad_scale
        gf = gs[o0:o1].view(P, M * 3)
        for p, gg in zip(ref, (gs[:o0], gf[:, :3], gf[:, 3:], gs[o1:])):
            p.grad = gg.clone().contiguous()
        opt.step()
        topt.step()
        got = flat.clone()
        want = torch.cat([ref[0].detach(), torch.cat([ref[1].detach(), ref[2].detach()], dim=1).reshape(-1), ref[3].detach()])
        assert torch.allclose(got, want, rtol=2e-6, atol=1e-7), (step, float((got - want).abs().max()))
    assert opt.step_count == 6


def test_train_step_graph_equals_eager_and_fits(gpu_device):
    """Row H harness: zero_grad -> render -> L1 -> backward -> stats -> Adam.  The HIP-graph replay of the step must
    track the eager step, and a short fit against images of a hidden Gaussian set must reduce the loss."""
    import torch
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    from fateavatar_amd.train import TrainStep
    P, res, views = 4000, 96, 4
    truth = scenes.head_scene(P=P, res=res, sh_degree=1, seed=3, opacity=0.6)
    cams = [TorchCamera(scenes.head_scene(P=8, res=res, sh_degree=1, seed=3, view=v, n_views=views).camera, gpu_device)
            for v in range(views)]
    bg = torch.from_numpy(truth.bg).to(gpu_device)
    pc_true = FlatGaussians(truth.means3D, truth.shs, truth.opacities, truth.scales, truth.rotations, 1, gpu_device,
                            fused_activations=True)
    with torch.no_grad():
        gts = [render(c, pc_true, bg)["render"].clone() for c in cams]
    rng = np.random.default_rng(0)
    shs0 = (truth.shs + 0.3 * rng.standard_normal(truth.shs.shape)).astype(np.float32)

    def run(use_graph):
        pc = FlatGaussians(truth.means3D, shs0, truth.opacities * 0.7, truth.scales, truth.rotations, 1, gpu_device,
                           fused_activations=True)
        cam = TorchCamera(scenes.head_scene(P=8, res=res, sh_degree=1, seed=3, view=0, n_views=views).camera, gpu_device)
        ts = TrainStep(pc, cam, bg, use_graph=use_graph)
        losses = []
        for it in range(24):
            losses.append(ts.step(cams[it % views], gts[it % views]).clone())
        torch.cuda.synchronize()
        ts.check()
        return pc.flat.clone(), [float(x) for x in losses], ts

    flat_e, loss_e, ts_e = run(False)
    flat_g, loss_g, ts_g = run(True)
    assert ts_g._graph is not None and ts_e._graph is None
    assert np.mean(loss_e[-4:]) < 0.8 * np.mean(loss_e[:4]), loss_e
    # same trajectory up to the summation order of the gradient atomics
    assert np.allclose(loss_g, loss_e, rtol=2e-3), (loss_g, loss_e)
    util.assert_same_trajectory(flat_g, flat_e, "graph vs eager")
    assert torch.equal(ts_g.denom, ts_e.denom) and float(ts_e.denom.max()) == 24.0
    assert torch.allclose(ts_g.xyz_gradient_accum, ts_e.xyz_gradient_accum, rtol=5e-2, atol=1e-7)


def test_prune_densify_reset_follow_reference_optimizer_surgery(gpu_device):
    """TrainStep.prune_low_opacity / densify_by_gradient / reset_opacity: parameters AND Adam moments follow the row
    surgery of model/fateavatar.py:610-731 (kept rows keep their moments, appended rows start at zero, the step count
    is kept), and the (re-captured) step keeps training afterwards."""
    import torch
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    from fateavatar_amd.train import TrainStep
    P, res = 3000, 96
    truth = scenes.head_scene(P=P, res=res, sh_degree=1, seed=4, opacity=0.5)
    cam = TorchCamera(truth.camera, gpu_device)
    bg = torch.from_numpy(truth.bg).to(gpu_device)
    pc_true = FlatGaussians(truth.means3D, truth.shs, truth.opacities, truth.scales, truth.rotations, 1, gpu_device,
                            fused_activations=True)
    with torch.no_grad():
        gt = render(cam, pc_true, bg)["render"].clone()
    op0 = truth.opacities.copy()
    op0[::9] = 0.002  # below the pruning threshold
    pc = FlatGaussians(truth.means3D, truth.shs * 0.5, op0, truth.scales, truth.rotations, 1, gpu_device,
                       fused_activations=True)
    ts = TrainStep(pc, TorchCamera(truth.camera, gpu_device), bg, lrs=dict(opacity=1e-4))
    for _ in range(5):
        ts.step(cam, gt)
    torch.cuda.synchronize()
    assert ts._graph is not None

    def blocks(flat, rows):
        out, o = [], 0
        for w in pc.widths():
            out.append(flat[o:o + rows * w].view(rows, w).clone())
            o += rows * w
        return out

    # ---- prune
    rows0 = pc.P
    keep = ~(torch.sigmoid(pc._opacity) < 0.005).reshape(-1)
    p0, m0, v0 = blocks(pc.flat, rows0), blocks(ts.adam.exp_avg, rows0), blocks(ts.adam.exp_avg_sq, rows0)
    removed = ts.prune_low_opacity(0.005)
    assert removed == int((~keep).sum()) and removed >= P // 9 - 1 and pc.P == rows0 - removed
    for a, b in zip(blocks(pc.flat, pc.P) + blocks(ts.adam.exp_avg, pc.P) + blocks(ts.adam.exp_avg_sq, pc.P), p0 + m0 + v0):
        assert torch.equal(a, b[keep])
    assert ts.adam.step_count == 5 and ts._graph is None
    for _ in range(4):
        ts.step(cam, gt)
    torch.cuda.synchronize()
    assert ts._graph is not None and ts.adam.step_count == 9

    # ---- densify
    rows1 = pc.P
    p1, m1 = blocks(pc.flat, rows1), blocks(ts.adam.exp_avg, rows1)
    idx = ts.densify_by_gradient(200, generator=torch.Generator(device=gpu_device).manual_seed(0))
    assert pc.P == rows1 + 200 and int(idx.max()) < rows1
    p2, m2, v2 = blocks(pc.flat, pc.P), blocks(ts.adam.exp_avg, pc.P), blocks(ts.adam.exp_avg_sq, pc.P)
    for f in range(5):
        assert torch.equal(p2[f][:rows1], p1[f]) and torch.equal(m2[f][:rows1], m1[f])
        assert float(m2[f][rows1:].abs().max()) == 0.0 and float(v2[f][rows1:].abs().max()) == 0.0
        want = torch.log(torch.exp(p1[f][idx]) * 0.75) if f == 3 else p1[f][idx]
        assert torch.equal(p2[f][rows1:], want)
    assert float(ts.denom.abs().max()) == 0.0 and ts.denom.shape == (pc.P, 1)
    l0 = float(ts.step(cam, gt))
    for _ in range(6):
        l1 = float(ts.step(cam, gt))
    assert l1 < l0

    # ---- opacity reset
    ts.reset_opacity()
    assert float(torch.sigmoid(pc._opacity.detach()).max()) <= 0.01 + 1e-6
    mo = blocks(ts.adam.exp_avg, pc.P)
    assert float(mo[2].abs().max()) == 0.0 and float(mo[0].abs().max()) > 0.0
    ts.step(cam, gt)
    torch.cuda.synchronize()
    ts.check()


def test_checkpoint_resume_continues_the_same_trajectory(gpu_device, tmp_path):
    """TrainStep.state_dict -> torch.save -> a fresh TrainStep.load_state_dict: the resumed run follows the
    uninterrupted one (up to the summation order of the gradient atomics); the saved `model` uses the GaussianModel
    parameter names and shapes."""
    import torch
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    from fateavatar_amd.train import TrainStep
    truth = scenes.head_scene(P=2500, res=80, sh_degree=1, seed=6, opacity=0.5)
    cam = TorchCamera(truth.camera, gpu_device)
    bg = torch.from_numpy(truth.bg).to(gpu_device)
    mk = lambda shs: FlatGaussians(truth.means3D, shs, truth.opacities, truth.scales, truth.rotations, 1, gpu_device,  # noqa: E731
                                   fused_activations=True)
    with torch.no_grad():
        gt = render(cam, mk(truth.shs), bg)["render"].clone()
    a = TrainStep(mk(truth.shs * 0.3), TorchCamera(truth.camera, gpu_device), bg)
    for _ in range(6):
        a.step(cam, gt)
    sd = a.state_dict()
    assert sd["global_step"] == 6 and sd["model"]["_features_dc"].shape == (2500, 1, 3)
    assert sd["model"]["_features_rest"].shape == (2500, 3, 3) and sd["model"]["_opacity"].shape == (2500, 1)
    torch.save(sd, tmp_path / "ck.pth")
    for _ in range(5):
        a.step(cam, gt)
    b = TrainStep(mk(truth.shs * 0.0), TorchCamera(truth.camera, gpu_device), bg)
    b.load_state_dict(torch.load(tmp_path / "ck.pth", map_location=gpu_device))
    assert b.adam.step_count == 6
    for _ in range(5):
        b.step(cam, gt)
    torch.cuda.synchronize()
    assert b.adam.step_count == a.adam.step_count == 11
    util.assert_same_trajectory(a.pc.flat, b.pc.flat, "checkpoint round trip", tight=2e-3)
    assert torch.equal(a.denom, b.denom)


def _binding_case(device, N=20000, seed=0, dtype=None):
    import torch
    verts, faces, _ = scenes.head_geometry()
    g = torch.Generator().manual_seed(seed)
    V, F = verts.shape[0], faces.shape[0]
    canon = torch.from_numpy(verts)
    posed = canon * (1.0 + 0.05 * torch.randn(V, 1, generator=g)) + 0.002 * torch.randn(V, 3, generator=g)
    fi = torch.randint(0, F, (N,), generator=g, dtype=torch.int32)
    uvw = torch.rand(N, 3, generator=g)
    bary = uvw / uvw.sum(-1, keepdim=True)
    offset = torch.randn(N, 1, generator=g)
    rot = torch.randn(N, 4, generator=g)
    scl = torch.randn(N, 3, generator=g) - 6.0
    t = lambda x: x.to(device) if dtype is None or not x.is_floating_point() else x.to(device, dtype)  # noqa: E731
    return dict(canon=t(canon), posed=t(posed), faces=t(torch.from_numpy(faces)), fi=t(fi), bary=t(bary), offset=t(offset),
                rot=t(rot), scl=t(scl))


def test_mesh_binding_forward_and_backward_vs_oracle(gpu_device):
    """fr_bind_forward / fr_bind_backward (one kernel each) == the oracle's restatement of model/fateavatar.py:225-258:
    forward against the same formulas in fp32, gradients against float64 autograd of them."""
    import torch
    from fateavatar_amd.binding import bind_gaussians, face_scale
    from oracle import binding as B
    c = _binding_case(gpu_device)
    shell = 0.01
    canon_scale = face_scale(c["canon"], c["faces"])
    ref_scale = B.face_orientation(c["canon"].cpu(), c["faces"].cpu())[1]
    assert torch.allclose(canon_scale.cpu(), ref_scale, rtol=2e-6, atol=0)
    leaves = [c[k].clone().requires_grad_(True) for k in ("posed", "offset", "rot", "scl")]
    xyz, rot, scl = bind_gaussians(leaves[0], c["faces"], c["fi"], c["bary"], canon_scale, leaves[1], leaves[2], leaves[3], shell)
    # ---- forward vs fp32 oracle
    o32 = B.bind(c["posed"].cpu(), c["faces"].cpu(), c["fi"].cpu(), c["bary"].cpu(), ref_scale, c["offset"].cpu(),
                 c["rot"].cpu(), c["scl"].cpu(), shell)
    for got, want, name in zip((xyz, rot, scl), o32, ("xyz", "rotation", "scaling")):
        assert torch.allclose(got.detach().cpu(), want, rtol=2e-5, atol=2e-6), (name, float((got.detach().cpu() - want).abs().max()))
    assert bool((rot[:, 0] >= 0).all())
    # ---- backward vs float64 autograd of the oracle
    g = torch.Generator().manual_seed(9)
    w = [torch.randn(x.shape, generator=g) for x in (xyz, rot, scl)]
    torch.autograd.backward([xyz, rot, scl], [x.to(gpu_device) for x in w])
    d = lambda k: c[k].cpu().double()  # noqa: E731
    l64 = [d("posed").requires_grad_(True), d("offset").requires_grad_(True), d("rot").requires_grad_(True),
           d("scl").requires_grad_(True)]
    o64 = B.bind(l64[0], c["faces"].cpu(), c["fi"].cpu(), d("bary"), ref_scale.double(), l64[1], l64[2], l64[3], shell)
    torch.autograd.backward(list(o64), [x.double() for x in w])
    for got, want, name in zip(leaves, l64, ("d_verts", "d_offset", "d_rotation", "d_scaling")):
        ref = want.grad.float()
        err = float((got.grad.cpu() - ref).norm() / ref.norm())
        assert err < 2e-5, (name, err)
        assert got.grad.shape == got.shape


def test_mesh_binding_options_and_degenerate_face(gpu_device):
    """resize_scale off leaves the scaling untouched; a zero-area face goes through the eps clamps without NaN/inf;
    only the requested gradients are produced."""
    import torch
    from fateavatar_amd.binding import bind_gaussians, face_scale
    from oracle import binding as B
    c = _binding_case(gpu_device, N=3000, seed=3)
    f0 = c["faces"][7].long()
    c["posed"][f0[2]] = c["posed"][f0[1]]              # degenerate face 7
    c["fi"][:50] = 7
    canon_scale = face_scale(c["canon"], c["faces"])
    off = c["offset"].clone().requires_grad_(True)
    xyz, rot, scl = bind_gaussians(c["posed"], c["faces"], c["fi"], c["bary"], None, off, c["rot"], c["scl"], 0.02,
                                   resize_scale=False)
    assert torch.equal(scl, c["scl"]) and torch.isfinite(xyz).all() and torch.isfinite(rot).all()
    xyz.sum().backward()
    assert off.grad is not None and torch.isfinite(off.grad).all()
    o = B.bind(c["posed"].cpu(), c["faces"].cpu(), c["fi"].cpu(), c["bary"].cpu(), canon_scale.cpu(), c["offset"].cpu(),
               c["rot"].cpu(), c["scl"].cpu(), 0.02, resize_scale=False)
    assert torch.allclose(xyz.detach().cpu(), o[0], rtol=2e-5, atol=2e-6)
    ok = torch.ones(3000, dtype=torch.bool)
    ok[:50] = False                                     # the quaternion of a degenerate frame is not well defined
    assert torch.allclose(rot.detach().cpu()[ok], o[1][ok], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("fused", [False, True])
def test_two_frames_in_flight_like_the_reference_batch_loop(gpu_device, fused, monkeypatch):
    """model/fateavatar.py:251-276 renders every frame of the batch first and back-propagates one summed loss
    afterwards: two forwards in flight, then both backwards.  Gradients must equal the sum of the per-frame ones.
    One frame at a time with the gradients kept, the second backward is ADDED to the gradient buffers by the kernel
    (FR_FLAG_ACCUMULATE) instead of going through a temporary and autograd's add."""
    import torch
    from fateavatar_amd import rasterizer
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    accumulated = []
    inner = rasterizer.rasterize_gaussians_backward

    def spy(*a, **k):
        accumulated.append(tuple(sorted(k.get("_accumulate", ()))))
        return inner(*a, **k)

    monkeypatch.setattr(rasterizer, "rasterize_gaussians_backward", spy)
    P, res = 5000, 112
    sa = scenes.head_scene(P=P, res=res, sh_degree=2, seed=8, view=0, n_views=3, opacity=0.4)
    sb = scenes.head_scene(P=P, res=res, sh_degree=2, seed=8, view=1, n_views=3, opacity=0.4)
    bg = torch.from_numpy(sa.bg).to(gpu_device)
    cams = [TorchCamera(sa.camera, gpu_device), TorchCamera(sb.camera, gpu_device)]
    rng = np.random.default_rng(3)
    ws = [torch.from_numpy((rng.uniform(-1, 1, (3, res, res)) / res ** 2).astype(np.float32)).to(gpu_device) for _ in range(2)]

    def grads(order):
        pc = FlatGaussians(sa.means3D, sa.shs, sa.opacities, sa.scales, sa.rotations, 2, gpu_device, fused_activations=fused)
        if order == "batched":
            outs = [render(c, pc, bg) for c in cams]                      # both forwards first
            (sum((o["render"] * w).sum() for o, w in zip(outs, ws))).backward()
            vs = [o["viewspace_points"].grad.clone() for o in outs]
        else:
            vs = []
            for c, w in zip(cams, ws):                                    # one frame at a time, gradients accumulate
                o = render(c, pc, bg)
                (o["render"] * w).sum().backward()
                vs.append(o["viewspace_points"].grad.clone())
                if order == "sequential, added in the kernel":
                    pc.accumulate_into_kept_grads()
        return {n: getattr(pc, n).grad.clone() for n, _ in pc.FIELDS}, vs

    ga, va = grads("batched")
    assert accumulated == [(), ()]
    del accumulated[:]
    gb, vb = grads("sequential")
    assert accumulated == [(), ()]       # (not asked for: autograd adds, as it must for torch.autograd.grad users)
    del accumulated[:]
    gc, vc = grads("sequential, added in the kernel")
    in_place = ("dL_dmeans3D", "dL_dsh") if not fused else ("dL_dmeans3D", "dL_dopacity", "dL_drotations", "dL_dscales", "dL_dsh")
    assert accumulated == [(), tuple(sorted(in_place))], accumulated
    # the truth: each frame alone, from its own copy of the parameters, added up
    want = None
    for c, w in zip(cams, ws):
        pc = FlatGaussians(sa.means3D, sa.shs, sa.opacities, sa.scales, sa.rotations, 2, gpu_device, fused_activations=fused)
        (render(c, pc, bg)["render"] * w).sum().backward()
        g1 = {n: getattr(pc, n).grad.clone() for n, _ in pc.FIELDS}
        want = g1 if want is None else {n: want[n] + g1[n] for n in want}
    for n in ga:
        assert util.rel_l2(ga[n].cpu().numpy(), want[n].cpu().numpy()) < 1e-5, ("batched", n)
        assert util.rel_l2(gb[n].cpu().numpy(), want[n].cpu().numpy()) < 1e-5, ("sequential", n)
        assert util.rel_l2(gc[n].cpu().numpy(), want[n].cpu().numpy()) < 1e-5, ("sequential, added in the kernel", n)
    assert float(ga["_xyz"].abs().max()) > 0 and float(ga["_features"].abs().max()) > 0
    for x, y in zip(va, vb):
        assert util.rel_l2(x.cpu().numpy(), y.cpu().numpy()) < 1e-5
    assert not torch.equal(va[0], va[1])


@pytest.mark.parametrize("mode", ["FR_BLEND_FWD=gather",
                                  "FR_DENSE_PAIRS_FWD=0,FR_DENSE_PAIRS_BWD=0", "FR_DENSE_PAIRS_FWD=9999,FR_DENSE_PAIRS_BWD=9999",
                                  "FR_CHAIN_SPINS=0", "FR_CHAIN_SPINS=3", "FR_HEAVY_PAIRS=0", "FR_HEAVY_PAIRS=99999", "FR_HEAVY_PAIRS=60"])
def test_selectable_blend_paths_stay_correct(gpu_device, mode):
    """The blend paths the environment can select (the gather as its own launch, and the per-unit
    all-pairs / sparse choice forced either way — 0: every unit takes the in-kernel all-pairs loops, 9999: none does, units
    with more pairs than slots are walked in record ranges) must keep matching the oracle for as long as they stay in
    the tree.  FR_CHAIN_SPINS=0 / 3: the units of the one-launch forward give up waiting for each other at
    once (after three polls) and compute the missing products and rows themselves — the path that makes the launch
    independent of dispatch order.  FR_HEAVY_PAIRS: the order of the backward's work list (every unit from the front,
    every unit from the back, most units in front) must not matter for the result.  Run in a subprocess: the switches
    are read at handle creation."""
    import os
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "from fateavatar_amd import scenes\n"
        "from tests import util\n"
        "from tests.test_gpu_parity import _check_forward, _check_backward\n"
        "dev = torch.device('cuda:0')\n"
        "rng = np.random.default_rng(0)\n"
        "for s in (scenes.head_scene(P=20000, res=256, sh_degree=1, seed=0, opacity=0.5),\n"
        "          scenes.random_scene(4000, 64, 64, sh_degree=0, seed=5, opacity_lo=0.6, opacity_hi=0.99, scale_lo=0.02, scale_hi=0.08)):\n"
        "    o = util.oracle_forward(s); h = util.HipFrame(s, dev); _check_forward(o, h, 'mode')\n"
        "    assert h.counts.max_tile_list > 64\n"
        "    H, W = s.camera.image_height, s.camera.image_width\n"
        "    dpix = (rng.uniform(-1, 1, (3, H, W)) / (H * W)).astype(np.float32)\n"
        "    _check_backward(o, h, dpix, 'mode')\n"
        "print('mode-ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ)
    for kv in mode.split(","):
        k, v = kv.split("=")
        env[k] = v
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "mode-ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


def test_backward_work_list_is_a_permutation_of_the_units(gpu_device):
    """k_unit_blend_chained leaves the blend backward a work list (BinningView::bwd_units, the first array of the binning
    buffer: 32 bytes per slot = descriptor, unit): every unit exactly once, whatever the stripes' cursors did — long
    lists (several units per tile, units in both classes) and a scene of short ones."""
    import torch
    for s, want_heavy in ((scenes.random_scene(4000, 64, 64, sh_degree=0, seed=5, opacity_lo=0.6, opacity_hi=0.99, scale_lo=0.02, scale_hi=0.08), True),
                          (scenes.head_scene(P=20000, res=256, sh_degree=0, seed=1), False)):
        h = util.HipFrame(s, gpu_device)
        torch.cuda.synchronize()
        n_tiles8 = ((s.camera.image_width + 7) // 8) * ((s.camera.image_height + 7) // 8)
        off = (-h.binning.data_ptr()) & 255
        # the number of units: every tile list of n entries has ceil(n / 64) of them; the descriptors carry (list start,
        # list length), so count the distinct lists
        raw = h.binning[off:off + 32 * (h.counts.num_instances // 64 + n_tiles8 + 1)].view(torch.int32).view(-1, 8).cpu().numpy()
        # slots are filled densely from 0: find nu as the point where the units stop being a permutation prefix
        starts = {}
        nu = 0
        for (xy, seg, start, n, u, *_pad) in raw.tolist():
            if n <= 0 or seg * 64 >= n:
                break
            nu += 1
        units = raw[:nu, 4]
        lists = {(int(r[2]), int(r[3])) for r in raw[:nu]}
        assert nu == sum((n + 63) // 64 for _, n in lists), (nu, len(lists))
        assert sorted(units.tolist()) == list(range(nu))
        # every slot's descriptor is its unit's: segment index in range, and the (start, n) pair names a list with that many units
        for xy, seg, start, n, u, *_pad in raw[:nu].tolist():
            assert 0 <= seg < (n + 63) // 64
        if want_heavy:
            assert h.counts.max_tile_list > 64


def test_forward_chain_makes_progress_while_other_work_holds_the_cus(gpu_device):
    """k_unit_blend_chained resolves each tile's unit chain INSIDE one launch (units wait for the products of the units
    in front of them).  Its progress argument needs no residency: a workgroup only waits for lower-numbered ones and
    exits when done.  Render long-list frames while a side stream keeps every CU busy with large GEMMs: the frames must
    complete (the test would hang otherwise) and match the oracle."""
    import torch
    s = scenes.random_scene(6000, 96, 96, sh_degree=0, seed=17, spread=0.01, scale_lo=0.004, scale_hi=0.01,
                            opacity_lo=0.05, opacity_hi=0.4)
    o = util.oracle_forward(s)
    side = torch.cuda.Stream(device=gpu_device)
    a = torch.randn((4096, 4096), device=gpu_device)
    stop_after = 40
    with torch.cuda.stream(side):
        for _ in range(stop_after):
            a = torch.nn.functional.normalize(a @ a, dim=1)
    frames = []
    for _ in range(12):
        frames.append(util.HipFrame(s, gpu_device))
    torch.cuda.synchronize()
    assert frames[0].counts.max_tile_list > 8 * 64       # chains of more than eight units
    for k in (0, 5, 11):
        _check_forward(o, frames[k], f"busy-{k}")
    assert torch.isfinite(a).all()


def test_views_in_flight_together_match_views_rendered_alone(gpu_device):
    """Three views of one model rendered and back-propagated CONCURRENTLY — a stream, an fr_handle slot and a captured
    graph each (bench.py --in-flight) — give the images and gradients the same views give one after the other."""
    import torch
    from fateavatar_amd import rasterizer
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    K, res = 3, 192
    bg = torch.ones(3, device=gpu_device)
    views = []
    for k in range(K):
        s = scenes.head_scene(P=30000, res=res, sh_degree=2, seed=0, view=k, n_views=K, opacity=0.4)
        g = torch.Generator().manual_seed(k)
        views.append(dict(k=k, s=s, cam=TorchCamera(s.camera, gpu_device),
                          pc=FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, s.sh_degree, gpu_device,
                                           fused_activations=True),
                          dL=((torch.rand((3, res, res), generator=g) - 0.5) / (res * res)).to(gpu_device),
                          stream=torch.cuda.Stream(device=gpu_device)))

    def frame(v):
        v["pc"].begin_step()
        out = render(v["cam"], v["pc"], bg)
        torch.autograd.backward(out["render"], grad_tensors=v["dL"])
        return out["render"]

    alone = []
    for v in views:                                   # one after the other, default handle
        img = frame(v).detach().clone()
        torch.cuda.synchronize()
        alone.append((img, v["pc"].collect_grads().clone()))
    outs = {}
    for v in views:                                   # capture one graph per view on its own handle and stream
        with rasterizer.handle_slot(v["k"] + 1):
            for _ in range(2):
                frame(v)
            torch.cuda.synchronize()
            with rasterizer.no_wait():
                with torch.cuda.stream(v["stream"]):
                    frame(v)
                torch.cuda.synchronize()
                v["graph"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(v["graph"], stream=v["stream"]):
                    outs[v["k"]] = frame(v)
        torch.cuda.synchronize()
    for _ in range(25):                               # in flight together
        for v in views:
            with torch.cuda.stream(v["stream"]):
                v["graph"].replay()
    torch.cuda.synchronize()
    for v, (img, grad) in zip(views, alone):
        with rasterizer.handle_slot(v["k"] + 1):
            assert not rasterizer.check_async_overflow(gpu_device.index or 0)
        assert torch.allclose(outs[v["k"]], img, rtol=1e-5, atol=1e-6), v["k"]
        got = v["pc"].collect_grads()
        assert float(grad.abs().max()) > 0
        # (float atomics in the blend backward: two runs agree to rounding, not bit for bit)
        assert util.rel_l2(got.cpu().numpy(), grad.cpu().numpy()) < 1e-5, v["k"]


@pytest.mark.parametrize("opacity", [0.1, 0.9])
def test_forward_is_bit_reproducible_across_runs_and_in_flight(gpu_device, opacity):
    """The forward has no order-dependent arithmetic: binning order varies from run to run (atomics hand out the bucket
    slots), the per-tile sort restores (depth, id) order, and the units of the one-launch blend exchange their products
    inside the launch.  150 renders of the same 100 k-Gaussian frame — the last 100 with a second view in flight on
    another stream and handle — must all give the SAME BITS: a hand-off that ever delivered a stale or half-written
    product would show here."""
    import torch
    from fateavatar_amd import rasterizer
    s = scenes.head_scene(P=100_000, res=512, sh_degree=1, seed=0, opacity=opacity)
    s2 = scenes.head_scene(P=100_000, res=512, sh_degree=1, seed=0, opacity=opacity, view=1, n_views=3)
    h0 = util.HipFrame(s, gpu_device)
    ref_img, ref_T, ref_n = h0.color.clone(), h0.final_T.clone(), h0.n_contrib.clone()
    side = torch.cuda.Stream(device=gpu_device)
    for it in range(150):
        if it >= 50:
            with rasterizer.handle_slot(1), torch.cuda.stream(side):
                util.HipFrame(s2, gpu_device)
        h = util.HipFrame(s, gpu_device)
        assert torch.equal(h.color, ref_img) and torch.equal(h.final_T, ref_T) and torch.equal(h.n_contrib, ref_n), it
    torch.cuda.synchronize()


def test_long_lists_without_the_big_sorter_launch(gpu_device):
    """The big-list sorter is only launched when the previous frame had a list longer than 2048; a frame whose long
    lists come as a surprise is sorted by the slow path inside k_tile_sort and must be just as correct.  Frame order:
    short lists -> long lists (slow path) -> long lists again (big sorter) -> short lists."""
    rng = np.random.default_rng(3)
    short = SCENES["rand_deg0"]
    long_ = scenes.random_scene(3000, 32, 32, sh_degree=0, seed=9, spread=0.004, scale_lo=0.002, scale_hi=0.004,
                                opacity_lo=0.02, opacity_hi=0.05)
    long_.means3D[:, 2] = 1.0 + rng.uniform(0, 0.5, long_.P).astype(np.float32)
    o_short, o_long = util.oracle_forward(short), util.oracle_forward(long_)
    h = util.HipFrame(short, gpu_device)
    assert h.counts.max_tile_list <= 2048
    _check_forward(o_short, h, "short-1")
    for tag in ("long-surprise", "long-again"):
        h = util.HipFrame(long_, gpu_device)
        assert h.counts.max_tile_list > 2048
        _check_forward(o_long, h, tag)
        dpix = (rng.uniform(-1, 1, (3, 32, 32)) / (32 * 32)).astype(np.float32)
        _check_backward(o_long, h, dpix, tag)
    _check_forward(o_short, util.HipFrame(short, gpu_device), "short-2")


def test_non_finite_gaussians_are_dropped_not_propagated(gpu_device):
    """Contract: a Gaussian with a non-finite opacity, scale, position or SH coefficient is dropped — no
    instance, no contribution, radius 0, zero gradient rows — instead of poisoning the image as the reference's
    arithmetic would; every other Gaussian renders exactly as if the bad ones were not there."""
    s = scenes.random_scene(1500, 64, 80, sh_degree=1, seed=31)
    bad = np.zeros(s.P, bool)
    bad[[3, 400, 777, 1200, 55, 910]] = True
    clean = scenes.GaussianScene(s.means3D[~bad], s.scales[~bad], s.rotations[~bad], s.opacities[~bad], s.shs[~bad],
                                 s.sh_degree, s.bg, s.camera)
    s.opacities[3, 0] = np.nan
    s.scales[400, 1] = np.inf
    s.means3D[777, 0] = np.nan
    s.opacities[1200, 0] = -np.inf
    s.shs[55, 2, 1] = np.nan
    s.shs[910, 0, 0] = np.inf
    h = util.HipFrame(s, gpu_device)
    hc = util.HipFrame(clean, gpu_device)
    assert (h.radii.cpu().numpy()[bad] == 0).all()
    col = h.color.cpu().numpy()
    assert np.isfinite(col).all()
    np.testing.assert_array_equal(col, hc.color.cpu().numpy())
    np.testing.assert_array_equal(h.final_T.cpu().numpy(), hc.final_T.cpu().numpy())
    rng = np.random.default_rng(1)
    dpix = (rng.uniform(-1, 1, (3, 64, 80)) / (64 * 80)).astype(np.float32)
    g, gc = h.backward(dpix), hc.backward(dpix)
    for k in ("dL_dmeans2D", "dL_dopacity", "dL_dmeans3D", "dL_dsh", "dL_dscales", "dL_drotations"):
        assert np.isfinite(g[k]).all(), k
        assert np.abs(g[k][bad]).max() == 0, k
        assert util.rel_l2(g[k][~bad], gc[k]) < 1e-5, k


def test_backward_accumulates_into_the_arrays_it_is_told_to(gpu_device):
    """FR_FLAG_ACCUMULATE(k): the k-th gradient array receives old + gradient (the same fp32 add torch does), culled
    Gaussians leave it untouched, and arrays without the flag are overwritten as always."""
    import torch
    from fateavatar_amd import rasterizer as R
    s = scenes.head_scene(P=3000, res=96, sh_degree=3, seed=12, opacity=0.5)
    m = s.means3D.copy()
    m[::7] += 100.0          # out of the frustum: culled
    dev = gpu_device
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
    cam = s.camera
    means, shs, op, sc, rot = t(m), t(s.shs), t(s.opacities), t(s.scales), t(s.rotations)
    bg = t(s.bg)
    empty = torch.empty(0, device=dev)
    view, proj, campos = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
    fw = R.rasterize_gaussians(bg, means, empty, op, sc, rot, 1.0, empty, view, proj, cam.tanfovx, cam.tanfovy, 96, 96, shs, 3,
                               campos, False, False)
    num_rendered, color, radii, geom, binning, img = fw
    assert int((radii == 0).sum()) >= 3000 // 7
    g = t(np.random.default_rng(0).uniform(-1, 1, (3, 96, 96)) / 96 ** 2)
    args = (bg, means, radii, empty, sc, rot, 1.0, empty, view, proj, cam.tanfovx, cam.tanfovy, g, shs, 3, campos, geom,
            num_rendered, binning, img, False)
    names = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")
    plain = dict(zip(names, R.rasterize_gaussians_backward(*args)))
    gen = torch.Generator().manual_seed(5)
    for added in (names, ("dL_dsh", "dL_dopacity"), ("dL_dmeans3D",)):
        old = {k: torch.randn(plain[k].shape, generator=gen).to(dev) * float(plain[k].abs().max()) for k in names}
        out = {k: old[k].clone() for k in names}
        got = dict(zip(names, R.rasterize_gaussians_backward(*args, _out=out, _accumulate=added)))
        for k in names:
            assert got[k].data_ptr() == out[k].data_ptr()
            want = old[k] + plain[k] if k in added else plain[k]
            # (two backward passes differ in the last bits: the blend backward adds with atomics)
            tol = 2e-6 * max(float(plain[k].abs().max()), 1e-30)
            assert float((got[k] - want).abs().max()) <= tol, (added, k, float((got[k] - want).abs().max()), tol)
            culled = (radii == 0)
            if k in added:
                assert torch.equal(got[k][culled], old[k][culled]), (added, k)
            else:
                assert float(got[k][culled].abs().max()) == 0.0, (added, k)
    with pytest.raises(RuntimeError):
        R.rasterize_gaussians_backward(*args, _accumulate=("dL_dsh",))


def test_render_batch_equals_separate_renders(gpu_device):
    """fr_forward_batch / fr_backward_batch (render_batch): K views through ONE launch chain must give what K separate
    render() calls give — images, radii and visibility bit for bit, gradients to atomic-summation order — also when the
    views differ in resolution and Gaussian count (a batched launch's grid is the largest view's), and with ONE holder
    shared by all views (autograd sums the views' gradients)."""
    import torch
    from fateavatar_amd import rasterizer
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render, render_batch
    specs = [dict(P=20000, res=256, view=0), dict(P=20000, res=256, view=1), dict(P=12000, res=192, view=2)]
    sc = [scenes.head_scene(P=d["P"], res=d["res"], sh_degree=2, seed=0, view=d["view"], n_views=3, opacity=0.5) for d in specs]
    cams = [TorchCamera(s.camera, gpu_device) for s in sc]
    bgs = [torch.from_numpy(s.bg).to(gpu_device) for s in sc]
    gs = [torch.rand(3, d["res"], d["res"], device=gpu_device, generator=None) / (3 * d["res"] ** 2) for d in specs]

    def holders(fused):
        return [FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 2, gpu_device, fused_activations=fused) for s in sc]

    for fused in (True, False):
        a, b = holders(fused), holders(fused)
        outs_a = []
        for k in range(3):
            with rasterizer.handle_slot(k):
                a[k].begin_step()
                o = render(cams[k], a[k], bgs[k])
                torch.autograd.backward(o["render"], grad_tensors=gs[k])
                outs_a.append(o)
        for pc in b:
            pc.begin_step()
        outs_b = render_batch(cams, b, bgs)
        torch.autograd.backward([o["render"] for o in outs_b], grad_tensors=gs)
        torch.cuda.synchronize()
        for k in range(3):
            assert torch.equal(outs_a[k]["render"], outs_b[k]["render"]), (fused, k)
            assert torch.equal(outs_a[k]["radii"], outs_b[k]["radii"]) and torch.equal(outs_a[k]["visibility_filter"], outs_b[k]["visibility_filter"])
            ga, gb = a[k].collect_grads().cpu().numpy(), b[k].collect_grads().cpu().numpy()
            assert util.rel_l2(gb, ga) < 1e-5, (fused, k, util.rel_l2(gb, ga))
            va, vb = outs_a[k]["viewspace_points"].grad.cpu().numpy(), outs_b[k]["viewspace_points"].grad.cpu().numpy()
            assert util.rel_l2(vb, va) < 1e-5 and float(abs(va).max()) > 0
    # ---- one holder for all views (same Gaussians, two cameras): the views' gradients add up
    s0 = sc[0]
    pc1 = FlatGaussians(s0.means3D, s0.shs, s0.opacities, s0.scales, s0.rotations, 2, gpu_device, fused_activations=True)
    pc2 = FlatGaussians(s0.means3D, s0.shs, s0.opacities, s0.scales, s0.rotations, 2, gpu_device, fused_activations=True)
    pc1.begin_step()
    loss = 0
    for k in range(2):
        with rasterizer.handle_slot(k):
            loss = loss + (render(cams[k], pc1, bgs[k])["render"] * gs[k]).sum()
    loss.backward()
    pc2.begin_step()
    o2 = render_batch(cams[:2], pc2, bgs[0])
    sum((o["render"] * g).sum() for o, g in zip(o2, gs)).backward()
    torch.cuda.synchronize()
    g1, g2 = pc1.collect_grads().cpu().numpy(), pc2.collect_grads().cpu().numpy()
    assert util.rel_l2(g2, g1) < 1e-5 and float(abs(g1).max()) > 0


def test_batch_refuses_shared_handles_and_too_many_views(gpu_device):
    import torch
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render_batch
    s = scenes.head_scene(P=2000, res=64, sh_degree=0, seed=0)
    pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 0, gpu_device, fused_activations=True)
    cam, bg = TorchCamera(s.camera, gpu_device), torch.from_numpy(s.bg).to(gpu_device)
    with pytest.raises(RuntimeError, match="handle slot each"):
        render_batch([cam, cam], pc, bg, slots=[1, 1])
    with pytest.raises(RuntimeError, match="1 .. 4 views"):
        render_batch([cam] * 5, pc, bg)
