"""Wavefront OBJ reader for the FLAME head template: what the reference gets from
`pytorch3d.io.load_obj` (model/fateavatar.py:120-135, volume_rendering/mesh_sampling.py:6): vertex positions, per-corner
position indices, UV coordinates and per-corner UV indices of a triangle (or fan-triangulated polygon) mesh.
numpy only."""
from __future__ import annotations

import numpy as np


def load_obj(path: str) -> dict:
    """Returns float32 `verts` [V,3], int32 `faces` [F,3] (0-based position indices), float32 `verts_uvs` [T,2] and
    int32 `faces_uvs` [F,3] (0-based; -1 where a corner has no texture index; [0,2] / [0,3] arrays when the file has no
    `vt`).  Polygons with more than three corners are fan-triangulated like pytorch3d does; negative (relative) indices
    are resolved."""
    verts, uvs, faces, faces_uv = [], [], [], []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if not tok or tok[0].startswith("#"):
                continue
            if tok[0] == "v":
                verts.append([float(t) for t in tok[1:4]])
            elif tok[0] == "vt":
                uvs.append([float(t) for t in tok[1:3]])
            elif tok[0] == "f":
                vi, ti = [], []
                for corner in tok[1:]:
                    parts = corner.split("/")
                    i = int(parts[0])
                    vi.append(i - 1 if i > 0 else len(verts) + i)
                    if len(parts) > 1 and parts[1]:
                        t = int(parts[1])
                        ti.append(t - 1 if t > 0 else len(uvs) + t)
                    else:
                        ti.append(-1)
                for k in range(1, len(vi) - 1):
                    faces.append([vi[0], vi[k], vi[k + 1]])
                    faces_uv.append([ti[0], ti[k], ti[k + 1]])
    return {"verts": np.asarray(verts, np.float32).reshape(-1, 3), "faces": np.asarray(faces, np.int32).reshape(-1, 3),
            "verts_uvs": np.asarray(uvs, np.float32).reshape(-1, 2), "faces_uvs": np.asarray(faces_uv, np.int32).reshape(-1, 3)}
