"""UV-raster / area-weighted sampling of binding points (volume_rendering/mesh_sampling.py:86-169 stand-ins)."""
import os

import numpy as np
import pytest

from fateavatar_amd import mesh_sampling as ms


def test_uv_raster_of_two_triangles():
    # unit square split along the diagonal (0,0)-(1,1); 4x4 texels; both faces counter-clockwise in (u, v)
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    faces = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    p2f, bary = ms.rasterize_uv(uv, faces, 4)
    on_edge = np.fliplr(np.eye(4, dtype=bool))                # texel centres ON the shared edge u = v (xi = 3 - yi)
    # pytorch3d's inside test is strict (every barycentric > 0): a centre exactly on an edge belongs to NEITHER face
    assert (p2f[on_edge] == -1).all() and (p2f[~on_edge] >= 0).all()
    assert np.allclose(bary[~on_edge].sum(-1), 1.0, atol=1e-6) and (bary >= 0).all()
    # texel centre -> uv -> reconstruct from barycentrics
    for yi in range(4):
        for xi in range(4):
            if on_edge[yi, xi]:
                continue
            u, v = (2 * xi + 1) / 8, 1 - (2 * yi + 1) / 8
            rec = (bary[yi, xi][:, None] * uv[faces[p2f[yi, xi]]]).sum(0)
            assert np.allclose(rec, [u, v], atol=1e-6)
    assert p2f[3, 3] == 0 and p2f[0, 0] == 1                  # below the diagonal (u > v): face 0; above: face 1
    # 5 x 5 texels: no centre on the diagonal's... centre (2, 2) IS (u = v = 0.5): empty; an overlapping duplicate of face 0
    # with a higher index never wins (equal z: the lowest face index stays)
    p2f5, _ = ms.rasterize_uv(uv, np.concatenate([faces, faces[:1]]), 5)
    assert p2f5[2, 2] == -1 and set(np.unique(p2f5).tolist()) == {-1, 0, 1}


def test_uv_raster_texel_convention_hand_computed():
    """Pins the texel convention and the face tests against numbers worked out by hand from pytorch3d 0.7.7's rasterizer
    (csrc/rasterize_meshes: RasterizeMeshesNaive, CheckPixelInsideFace, BarycentricCoordsForward) and the reference's two sign
    flips (volume_rendering/mesh_sampling.py:31-33,114-116):
      * the reference hands pytorch3d the vertices (x, y) = (-(2u - 1), -(-(2v - 1))) = (1 - 2u, 2v - 1);
      * `rasterize_meshes` samples output pixel (yi, xi) at the NDC point (1 - (2 xi + 1) / W, 1 - (2 yi + 1) / H): pixel
        CENTRES, +X to the left, +Y up, image row 0 at the top;
      hence texel (yi, xi) <-> u = (2 xi + 1) / (2 S), v = 1 - (2 yi + 1) / (2 S);
      * `cull_backfaces=True` (mesh_sampling.py:27) drops a face whose NDC area (v0 - v1) x (v2 - v1) is negative: with the flips
        above, 4 x its signed area in (u, v) — faces wound CLOCKWISE in the UV square get no samples.
    Three tiny triangles, each containing exactly ONE texel centre of a 4 x 4 raster and no texel corner: corner sampling
    would leave the raster empty, a bottom-left origin or a missing x flip would put them in other texels."""
    S = 4
    # triangle 0 around the centre of texel (row 0, column 0): u = 1/8, v = 7/8 (top-left of the UV square); counter-clockwise
    # triangle 1 around the centre of texel (row 3, column 2): u = 5/8, v = 1/8; counter-clockwise
    # triangle 2 around the centre of texel (row 1, column 3): u = 7/8, v = 5/8; CLOCKWISE: culled
    uv = np.array([[0.075, 0.85], [0.2, 0.85], [0.125, 0.95],
                   [0.575, 0.10], [0.70, 0.10], [0.625, 0.20],
                   [0.825, 0.60], [0.875, 0.70], [0.95, 0.60]], np.float64)
    faces = np.array([[0, 1, 2], [3, 4, 5], [6, 7, 8]], np.int32)
    p2f, bary = ms.rasterize_uv(uv, faces, S)
    want = np.full((S, S), -1, np.int32)
    want[0, 0], want[3, 2] = 0, 1
    assert np.array_equal(p2f, want), p2f
    # ... and the same triangle wound the other way round is sampled
    p2f_ccw, _ = ms.rasterize_uv(uv, np.array([[6, 8, 7]], np.int32), S)
    assert p2f_ccw[1, 3] == 0 and (p2f_ccw >= 0).sum() == 1
    # barycentrics of the texel centre, by hand: P = (0.125, 0.875): v: 0.85 + 0.1 wC = 0.875 -> wC = 1/4;
    # u: 0.075 wA + 0.2 wB + 0.125 / 4 = 0.125 with wA + wB = 3/4 -> wB = 0.3, wA = 0.45
    assert np.allclose(bary[0, 0], [0.45, 0.3, 0.25], atol=1e-6), bary[0, 0]
    # P = (0.625, 0.125): v: 0.10 + 0.10 wC = 0.125 -> wC = 1/4; u: 0.575 wA + 0.70 wB + 0.625 / 4 = 0.625 with
    # wA + wB = 3/4 -> 0.43125 + 0.125 wB = 0.46875 -> wB = 0.3, wA = 0.45
    assert np.allclose(bary[3, 2], [0.45, 0.3, 0.25], atol=1e-6), bary[3, 2]
    # the sampler lists covered texels in row-major order: the top-left triangle first
    fi, bc = ms.uniform_sampling_barycoords(S * S, uv, faces, strict=False)
    assert fi.tolist() == [0, 1] and np.allclose(bc, [[0.45, 0.3, 0.25], [0.45, 0.3, 0.25]], atol=1e-6)
    # a face whose NDC area is within pytorch3d's kEpsilon (1e-8) of zero is skipped
    sliver = np.array([[0.1, 0.1], [0.9, 0.1], [0.5, 0.1 + 1e-9]], np.float64)   # NDC area 4 x 0.8e-9 / 2 ... < 1e-8
    assert (ms.rasterize_uv(sliver, np.array([[0, 1, 2]], np.int32), 64)[0] == -1).all()


def test_uniform_sampling_counts_and_order():
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    faces = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    fi, bc = ms.uniform_sampling_barycoords(100, uv, faces)
    assert fi.shape == (100,) and bc.shape == (100, 3) and np.allclose(bc.sum(-1), 1, atol=1e-6)
    fi2, bc2 = ms.uniform_sampling_barycoords(120, uv, faces)   # 10x10 texels = 100 < 120: padded on sampled faces
    assert fi2.shape == (120,) and set(fi2.tolist()) <= {0, 1}
    fi3, _ = ms.uniform_sampling_barycoords(100, uv, faces, d_size=1.5)   # 15x15 = 225 > 100: truncated, row-major
    assert fi3.shape == (100,)
    fi4, _ = ms.uniform_sampling_barycoords(100, uv, faces, d_size=1.5, strict=False)
    # 225 texels, 15 of their centres on the diagonal both faces share: those whose edge function rounds to exactly zero
    # belong to neither face (strict inside test)
    assert 210 <= fi4.shape[0] <= 225


def test_random_sampling_is_area_weighted():
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [3, 0, 0], [0, 3, 0]], np.float32)
    f = np.array([[0, 1, 2], [0, 3, 4]], np.int32)             # areas 0.5 and 4.5
    fi, bc = ms.random_sampling_barycoords(20000, v, f, np.random.default_rng(1))
    assert abs((fi == 1).mean() - 0.9) < 0.01 and np.allclose(bc.sum(-1), 1, atol=1e-6)


@pytest.mark.skipif(not os.path.exists("/root/reference/weights/head_template_mouth_close.obj"),
                    reason="the reference checkout is only present in the build container")
def test_head_template_uv_sampling():
    from fateavatar_amd.obj import load_obj
    m = load_obj("/root/reference/weights/head_template_mouth_close.obj")
    fi, bc = ms.uniform_sampling_barycoords(10000, m["verts_uvs"], m["faces_uvs"])
    assert fi.shape == (10000,) and fi.max() < len(m["faces"]) and (bc >= 0).all()
    # the template's UV faces are wound counter-clockwise but for five (the faces pytorch3d's cull_backfaces=True drops):
    # the sign convention of rasterize_uv keeps the mesh, not the five
    t = m["verts_uvs"].astype(np.float64)[m["faces_uvs"]]
    A = (t[:, 1, 0] - t[:, 0, 0]) * (t[:, 2, 1] - t[:, 0, 1]) - (t[:, 2, 0] - t[:, 0, 0]) * (t[:, 1, 1] - t[:, 0, 1])
    assert (A < 0).sum() == 5 and not np.isin(fi, np.nonzero(A < 0)[0]).any()
    # neighbours in the list are neighbours on the mesh (row-major texel order): mean 3D distance of consecutive samples
    pos = (bc[:, :, None] * m["verts"][m["faces"][fi]]).sum(1)
    order = np.linalg.norm(np.diff(pos, axis=0), axis=1).mean()
    shuffled = np.linalg.norm(np.diff(pos[np.random.default_rng(0).permutation(len(pos))], axis=0), axis=1).mean()
    assert order < 0.2 * shuffled


def test_reference_initialisation_on_the_committed_template():
    """The reference's `_register_template_mesh` call (model/fateavatar.py:128-133: uniform_sampling_barycoords(tex_size^2, uvcoords,
    uvfaces), config/fateavatar.yaml:28 tex_size 256) on the template data the repository ships (vt / per-corner vt indices of
    weights/head_template_mouth_close.obj, exported by tests/golden/make_golden.py:gen_head).  Pins the covered-texel count and
    the first rows, and checks every texel sample INDEPENDENTLY of the rasterizer: its barycentrics must reproduce the texel
    centre it stands for, the texels must come in row-major order, and no sample may lie on a culled (clockwise) face."""
    from fateavatar_amd import scenes
    got = scenes.head_uv()
    assert got is not None
    uv, fuv = got
    assert uv.shape == (5150, 2) and fuv.shape == (10006, 3) and fuv.min() >= 0 and fuv.max() == 5149
    S = 256
    fi, bc = ms.uniform_sampling_barycoords(S * S, uv, fuv, rng=np.random.default_rng(0))
    assert fi.shape == (S * S,) and bc.shape == (S * S, 3) and bc.dtype == np.float32
    n_tex = 59_099                                            # texel centres the UV layout covers at 256 x 256 (90.2 %)
    p2f, _ = ms.rasterize_uv(uv, fuv, S)
    assert int((p2f >= 0).sum()) == n_tex
    assert fi[:6].tolist() == [9990, 9990, 9990, 10000, 9993, 9993]
    assert np.allclose(bc[:2], [[0.7098801, 0.28616953, 0.00394932], [0.8041375, 0.18685095, 0.00901049]], atol=2e-6)
    # every texel sample: barycentrics (strictly positive) x the face's UV corners = a texel CENTRE, centres row-major
    # (pytorch3d divides the edge functions by area + 1e-8: the barycentrics of a face of NDC area A sum to A / (A + 1e-8),
    # 0.994 on the template's smallest faces — the reference uses them as they come, so does the restatement)
    t = uv.astype(np.float64)[fuv]
    A = (t[:, 1, 0] - t[:, 0, 0]) * (t[:, 2, 1] - t[:, 0, 1]) - (t[:, 2, 0] - t[:, 0, 0]) * (t[:, 1, 1] - t[:, 0, 1])
    sums = bc[:n_tex].sum(-1).astype(np.float64)
    assert np.allclose(sums, 4 * A[fi[:n_tex]] / (4 * A[fi[:n_tex]] + 1e-8), atol=2e-6)
    rec = (bc[:n_tex, :, None].astype(np.float64) * uv[fuv[fi[:n_tex]]]).sum(1) / sums[:, None]
    xi = rec[:, 0] * S - 0.5
    yi = (1.0 - rec[:, 1]) * S - 0.5
    assert np.abs(xi - np.round(xi)).max() < 1e-4 and np.abs(yi - np.round(yi)).max() < 1e-4
    lin = np.round(yi).astype(np.int64) * S + np.round(xi).astype(np.int64)
    assert (np.diff(lin) > 0).all() and lin.min() >= 0 and lin.max() < S * S
    assert (bc[:n_tex] > 0).all() and np.allclose(bc[n_tex:].sum(-1), 1.0, atol=1e-5)
    assert (A < 0).sum() == 5 and not np.isin(fi, np.nonzero(A < 0)[0]).any()
    # the padding (strict=True: mesh_sampling.py:123-131): random points on faces that already carry a texel
    assert np.isin(fi[n_tex:], fi[:n_tex]).all()
    # another row count goes through the same call (raster side int(sqrt(n))): BASELINE config 3's 100 000
    fi2, _ = ms.uniform_sampling_barycoords(100_000, uv, fuv, rng=np.random.default_rng(0))
    assert fi2.shape == (100_000,) and int((ms.rasterize_uv(uv, fuv, 316)[0] >= 0).sum()) == 90_611
