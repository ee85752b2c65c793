"""Sampling Gaussian binding points on the template mesh without pytorch3d.

reference: volume_rendering/mesh_sampling.py:86-169 — `uniform_sampling_barycoords` rasterizes the UV layout of the
template at sqrt(num_points) x sqrt(num_points) texels with pytorch3d's `rasterize_meshes` and keeps, for every covered
texel, the face under its centre and the barycentric coordinates of the centre; `random_sampling_barycoords` draws
faces area-weighted.  These run once at model construction; they are host-side numpy here.  Texel (row yi, column xi)
has its centre at u = (2 xi + 1) / (2 S), v = 1 - (2 yi + 1) / (2 S) — the convention that results from the reference's
NDC flips (mesh_sampling.py:31-33, 114-116) and pytorch3d's pixel centres; overlapping UV faces resolve to the lowest
face index; back-face culling is not applied (the FLAME UV layout is consistently wound).  pytorch3d is not available
in this image, so the convention cannot be pinned by running it; it is derived from pytorch3d 0.7.7's documented pixel ->
NDC mapping (`rasterize_meshes` samples output pixel (yi, xi) at NDC (1 - (2 xi + 1) / W, 1 - (2 yi + 1) / H): pixel
centres, +X left, +Y up, row 0 at the top) and pinned by a hand-computed two-triangle case that tells pixel-centre from
pixel-corner sampling and a top-left from a bottom-left origin (tests/test_mesh_sampling.py).
"""
from __future__ import annotations

import math

import numpy as np


def rasterize_uv(tex_coord: np.ndarray, uv_faces: np.ndarray, size: int):
    """pix_to_face [S,S] int32 (-1 = empty) and bary [S,S,3] float32 of the UV layout at S x S texels."""
    S = int(size)
    uv = np.asarray(tex_coord, np.float64)
    tri = uv[np.asarray(uv_faces, np.int64)]                      # [F,3,2]
    # texel-centre coordinates of the corners: column = u*S - 0.5, row = (1 - v)*S - 0.5
    px = tri[..., 0] * S - 0.5
    py = (1.0 - tri[..., 1]) * S - 0.5
    pix_to_face = np.full((S, S), -1, np.int32)
    bary = np.zeros((S, S, 3), np.float32)
    x0 = np.clip(np.ceil(px.min(1)).astype(np.int64), 0, S)
    x1 = np.clip(np.floor(px.max(1)).astype(np.int64), -1, S - 1)
    y0 = np.clip(np.ceil(py.min(1)).astype(np.int64), 0, S)
    y1 = np.clip(np.floor(py.max(1)).astype(np.int64), -1, S - 1)
    for f in range(tri.shape[0] - 1, -1, -1):                     # descending: the lowest index is written last
        if x1[f] < x0[f] or y1[f] < y0[f]:
            continue
        ax, ay, bx, by, cx, cy = px[f, 0], py[f, 0], px[f, 1], py[f, 1], px[f, 2], py[f, 2]
        area = (bx - ax) * (cy - ay) - (cx - ax) * (by - ay)
        if area == 0.0:
            continue
        ys, xs = np.mgrid[y0[f]:y1[f] + 1, x0[f]:x1[f] + 1]
        w0 = ((bx - xs) * (cy - ys) - (cx - xs) * (by - ys)) / area
        w1 = ((cx - xs) * (ay - ys) - (ax - xs) * (cy - ys)) / area
        w2 = 1.0 - w0 - w1
        inside = (w0 >= 0) & (w1 >= 0) & (w2 >= 0)
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
