"""UV-raster / area-weighted sampling of binding points (volume_rendering/mesh_sampling.py:86-169 stand-ins)."""
import os

import numpy as np
import pytest

from fateavatar_amd import mesh_sampling as ms


def test_uv_raster_of_two_triangles():
    # unit square split along the diagonal (0,0)-(1,1); 4x4 texels
    uv = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32)
    faces = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    p2f, bary = ms.rasterize_uv(uv, faces, 4)
    assert (p2f >= 0).all()                                   # the square covers every texel centre
    assert np.allclose(bary.sum(-1), 1.0, atol=1e-6) and (bary >= 0).all()
    # texel centre -> uv -> reconstruct from barycentrics
    for yi in range(4):
        for xi in range(4):
            u, v = (2 * xi + 1) / 8, 1 - (2 * yi + 1) / 8
            rec = (bary[yi, xi][:, None] * uv[faces[p2f[yi, xi]]]).sum(0)
            assert np.allclose(rec, [u, v], atol=1e-6)
    assert p2f[3, 3] == 0 and p2f[0, 0] == 1                  # below the diagonal (u > v): face 0; above: face 1
    assert (np.fliplr(p2f).diagonal() == 0).all()             # texel centres ON the shared edge u = v: the lowest index wins


def test_uv_raster_texel_convention_hand_computed():
    """Pins the texel convention against numbers worked out by hand from pytorch3d 0.7.7's documented rasterizer
    semantics and the reference's two sign flips (volume_rendering/mesh_sampling.py:31-33,114-116):
      * the reference hands pytorch3d the vertices (x, y) = (-(2u - 1), -(-(2v - 1))) = (1 - 2u, 2v - 1);
      * `rasterize_meshes` samples output pixel (yi, xi) at the NDC point (1 - (2 xi + 1) / W, 1 - (2 yi + 1) / H): pixel
        CENTRES, +X to the left, +Y up, image row 0 at the top;
      hence texel (yi, xi) <-> u = (2 xi + 1) / (2 S), v = 1 - (2 yi + 1) / (2 S).
    Two tiny triangles, each containing exactly ONE texel centre of a 4 x 4 raster and no texel corner: corner sampling
    would leave the raster empty, a bottom-left origin or a missing x flip would put them in other texels."""
    S = 4
    # triangle 0 around the centre of texel (row 0, column 0): u = 1/8, v = 7/8 (top-left of the UV square); counter-clockwise
    # triangle 1 around the centre of texel (row 3, column 2): u = 5/8, v = 1/8; CLOCKWISE (no back-face culling in the UV raster)
    uv = np.array([[0.075, 0.85], [0.2, 0.85], [0.125, 0.95],
                   [0.575, 0.10], [0.625, 0.20], [0.70, 0.10]], np.float64)
    faces = np.array([[0, 1, 2], [3, 4, 5]], np.int32)
    p2f, bary = ms.rasterize_uv(uv, faces, S)
    want = np.full((S, S), -1, np.int32)
    want[0, 0], want[3, 2] = 0, 1
    assert np.array_equal(p2f, want), p2f
    # barycentrics of the texel centre, by hand: P = (0.125, 0.875): v: 0.85 + 0.1 wC = 0.875 -> wC = 1/4;
    # u: 0.075 wA + 0.2 wB + 0.125 / 4 = 0.125 with wA + wB = 3/4 -> wB = 0.3, wA = 0.45
    assert np.allclose(bary[0, 0], [0.45, 0.3, 0.25], atol=1e-6), bary[0, 0]
    # P = (0.625, 0.125): v: 0.10 + 0.10 wB = 0.125 -> wB = 1/4; u: 0.575 wA + 0.625 / 4 + 0.70 wC = 0.625 with
    # wA + wC = 3/4 -> 0.575 * 0.75 + 0.125 wC = 0.46875 -> wC = 0.3, wA = 0.45
    assert np.allclose(bary[3, 2], [0.45, 0.25, 0.3], atol=1e-6), bary[3, 2]
    # the sampler lists covered texels in row-major order: the top-left triangle first
    fi, bc = ms.uniform_sampling_barycoords(S * S, uv, faces, strict=False)
    assert fi.tolist() == [0, 1] and np.allclose(bc, [[0.45, 0.3, 0.25], [0.45, 0.25, 0.3]], atol=1e-6)


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
    assert fi4.shape == (225,)


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
    # neighbours in the list are neighbours on the mesh (row-major texel order): mean 3D distance of consecutive samples
    pos = (bc[:, :, None] * m["verts"][m["faces"][fi]]).sum(1)
    order = np.linalg.norm(np.diff(pos, axis=0), axis=1).mean()
    shuffled = np.linalg.norm(np.diff(pos[np.random.default_rng(0).permutation(len(pos))], axis=0), axis=1).mean()
    assert order < 0.2 * shuffled
