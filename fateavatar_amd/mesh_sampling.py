"""Sampling Gaussian binding points on the template mesh without pytorch3d.

reference: volume_rendering/mesh_sampling.py:86-169 — `uniform_sampling_barycoords` rasterizes the UV layout of the
template at sqrt(num_points) x sqrt(num_points) texels with pytorch3d's `rasterize_meshes` and keeps, for every covered
texel, the face under its centre and the barycentric coordinates of the centre; `random_sampling_barycoords` draws
faces area-weighted.  These run once at model construction; they are host-side numpy here.

pytorch3d (0.7.7, README.md:40) is not available in this image, so `rasterize_uv` restates what its rasterizer does for the
reference's call (mesh_sampling.py:13-57: blur_radius 0, faces_per_pixel 1, perspective_correct False, cull_backfaces TRUE)
— `RasterizeMeshesNaive` / `CheckPixelInsideFace` and `BarycentricCoordsForward` of pytorch3d's csrc/rasterize_meshes —, parity
unpinned by running it:
  * the reference hands it the vertices (x, y) = (1 - 2u, 2v - 1) (two sign flips, mesh_sampling.py:31-33, 114-116); output
    pixel (row yi, column xi) is sampled at its CENTRE, NDC (1 - (2 xi + 1) / S, 1 - (2 yi + 1) / S) (+X left, +Y up, row 0 at
    the top): texel (yi, xi) <-> u = (2 xi + 1) / (2 S), v = 1 - (2 yi + 1) / (2 S);
  * a face is skipped if it faces away — signed NDC area (v0 - v1) x (v2 - v1) < 0, which with the flips above is a face wound
    CLOCKWISE in (u, v); 5 of the head template's 10 006 UV faces are — or if that area is within 1e-8 of zero;
  * a texel centre is inside a face iff all three barycentrics, edge function / (area + 1e-8), are STRICTLY positive: a centre
    exactly on an edge belongs to no face (the texel stays empty);
  * of several faces over one centre the smallest interpolated z wins and, at equal z (the UV layout is flat), the lowest face
    index.
"""
from __future__ import annotations

import math

import numpy as np

_K_EPS = 1e-8   # pytorch3d's kEpsilon


def rasterize_uv(tex_coord: np.ndarray, uv_faces: np.ndarray, size: int):
    """pix_to_face [S,S] int32 (-1 = empty) and bary [S,S,3] float32 of the UV layout at S x S texels."""
    S = int(size)
    uv = np.asarray(tex_coord, np.float64)
    tri = uv[np.asarray(uv_faces, np.int64)]                      # [F,3,2]
    nx = 1.0 - 2.0 * tri[..., 0]                                  # NDC as the reference hands it over
    ny = 2.0 * tri[..., 1] - 1.0
    # texel-centre coordinates of the corners (bounding boxes only): column = u*S - 0.5, row = (1 - v)*S - 0.5
    px = tri[..., 0] * S - 0.5
    py = (1.0 - tri[..., 1]) * S - 0.5
    pix_to_face = np.full((S, S), -1, np.int32)
    bary = np.zeros((S, S, 3), np.float32)
    x0 = np.clip(np.ceil(px.min(1)).astype(np.int64), 0, S)
    x1 = np.clip(np.floor(px.max(1)).astype(np.int64), -1, S - 1)
    y0 = np.clip(np.ceil(py.min(1)).astype(np.int64), 0, S)
    y1 = np.clip(np.floor(py.max(1)).astype(np.int64), -1, S - 1)

    def edge(qx, qy, ax, ay, bx, by):                             # EdgeFunctionForward(q, a, b)
        return (qx - ax) * (by - ay) - (qy - ay) * (bx - ax)

    for f in range(tri.shape[0] - 1, -1, -1):                     # descending: the lowest index is written last
        if x1[f] < x0[f] or y1[f] < y0[f]:
            continue
        ax, ay, bx, by, cx, cy = nx[f, 0], ny[f, 0], nx[f, 1], ny[f, 1], nx[f, 2], ny[f, 2]
        area = edge(ax, ay, bx, by, cx, cy)                       # face_area = EdgeFunctionForward(v0, v1, v2)
        if area < 0.0 or abs(area) <= _K_EPS:                     # cull_backfaces / zero_face_area
            continue
        ys, xs = np.mgrid[y0[f]:y1[f] + 1, x0[f]:x1[f] + 1]
        qx = 1.0 - (2.0 * xs + 1.0) / S
        qy = 1.0 - (2.0 * ys + 1.0) / S
        den = edge(cx, cy, ax, ay, bx, by) + _K_EPS               # BarycentricCoordsForward
        w0 = edge(qx, qy, bx, by, cx, cy) / den
        w1 = edge(qx, qy, cx, cy, ax, ay) / den
        w2 = edge(qx, qy, ax, ay, bx, by) / den
        inside = (w0 > 0) & (w1 > 0) & (w2 > 0)
        if inside.any():
            yy, xx = ys[inside], xs[inside]
            pix_to_face[yy, xx] = f
            bary[yy, xx] = np.stack([w0[inside], w1[inside], w2[inside]], -1)
    return pix_to_face, bary


def uniform_sampling_barycoords(num_points: int, tex_coord, uv_faces, d_size: float = 1.0, strict: bool = True,
                                rng: np.random.Generator | None = None):
    """mesh_sampling.py:86-138.  Returns (face_index [n] int64, bary_coords [n,3] float32) in row-major texel order
    (so consecutive Gaussians are neighbours on the mesh); with `strict` the result is padded with random samples on
    already sampled faces or truncated to exactly `num_points`."""
    rng = rng or np.random.default_rng(0)
    uv_size = int(math.sqrt(num_points) * d_size)
    pix_to_face, bary = rasterize_uv(tex_coord, uv_faces, uv_size)
    mask = pix_to_face >= 0
    face_index = pix_to_face[mask].astype(np.int64)
    bary_coords = bary[mask]
    cur = face_index.shape[0]
    if strict:
        if cur < num_points:
            pad = num_points - cur
            extra_faces = face_index[rng.integers(0, cur, pad)]
            w = rng.random((pad, 3)).astype(np.float32)
            face_index = np.concatenate([face_index, extra_faces])
            bary_coords = np.concatenate([bary_coords, w / w.sum(-1, keepdims=True)])
        elif cur > num_points:
            face_index, bary_coords = face_index[:num_points], bary_coords[:num_points]
    return face_index, bary_coords


def random_sampling_barycoords(num_points: int, vertices, faces, rng: np.random.Generator | None = None):
    """mesh_sampling.py:140-169: faces drawn with probability proportional to their area (with replacement), barycentric
    coordinates rand(3) / sum."""
    rng = rng or np.random.default_rng(0)
    v = np.asarray(vertices, np.float64)
    t = v[np.asarray(faces, np.int64)]
    area = 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1)
    face_index = rng.choice(len(area), size=num_points, replace=True, p=area / area.sum())
    w = rng.random((num_points, 3)).astype(np.float32)
    return face_index.astype(np.int64), w / w.sum(-1, keepdims=True)
