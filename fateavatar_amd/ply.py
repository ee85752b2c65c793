"""The Inria 3DGS point-cloud format the reference saves and loads.

reference: GaussianModel.save_ply / load_ply / construct_list_of_attributes,
volume_rendering/gaussian_model.py:190-269 (written there through the `plyfile` package, which this image does not
have: the binary little-endian PLY container is small enough to read and write with numpy alone).

One `vertex` element of float32 properties, in this order:
    x y z  nx ny nz  f_dc_0..2  f_rest_0..(3(M-1)-1)  opacity  scale_0..2  rot_0..3
All values are the RAW parameters (logit opacity, log scale, un-normalised quaternion r,x,y,z); normals are zeros.
SH coefficients are stored CHANNEL-major — f_rest holds all red coefficients, then green, then blue
(`features.transpose(1, 2).flatten(1)`, gaussian_model.py:213-214) — while the rasterizer and FlatGaussians keep them
coefficient-major [P, M, 3]; the functions below convert.
"""
from __future__ import annotations

import os

import numpy as np


def attribute_names(M: int) -> list[str]:
    """construct_list_of_attributes (gaussian_model.py:190-203) for M SH coefficients per channel."""
    names = ["x", "y", "z", "nx", "ny", "nz"]
    names += [f"f_dc_{i}" for i in range(3)]
    names += [f"f_rest_{i}" for i in range(3 * (M - 1))]
    names.append("opacity")
    names += [f"scale_{i}" for i in range(3)]
    names += [f"rot_{i}" for i in range(4)]
    return names


def save_ply(path: str, xyz, features, opacity, scaling, rotation) -> None:
    """Write raw Gaussian parameters: xyz [P,3], features [P,M,3] (coefficient-major), opacity [P,1] or [P],
    scaling [P,3], rotation [P,4]."""
    xyz = np.asarray(xyz, np.float32)
    features = np.asarray(features, np.float32)
    P, M = features.shape[0], features.shape[1]
    if features.shape != (P, M, 3) or xyz.shape != (P, 3):
        raise ValueError("features must be [P, M, 3] and xyz [P, 3]")
    f_dc = features[:, :1, :].transpose(0, 2, 1).reshape(P, 3)
    f_rest = features[:, 1:, :].transpose(0, 2, 1).reshape(P, 3 * (M - 1))
    cols = [xyz, np.zeros_like(xyz), f_dc, f_rest, np.asarray(opacity, np.float32).reshape(P, 1),
            np.asarray(scaling, np.float32).reshape(P, 3), np.asarray(rotation, np.float32).reshape(P, 4)]
    table = np.ascontiguousarray(np.concatenate(cols, axis=1), dtype="<f4")
    names = attribute_names(M)
    assert table.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {P}\n" + \
             "".join(f"property float {n}\n" for n in names) + "end_header\n"
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(table.tobytes())


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def _read_vertex_table(path: str):
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, count, props, in_vertex = None, None, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            tok = line.decode("ascii").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                if in_vertex:
                    count = int(tok[2])
                elif count is None:
                    raise ValueError(f"{path}: an element precedes `vertex`; not a Gaussian point cloud")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property in the vertex element")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if count is None:
            raise ValueError(f"{path}: no vertex element")
        if fmt == "ascii":
            data = np.loadtxt(f, dtype=np.float64, max_rows=count, ndmin=2)
            return {n: data[:, i] for i, (n, _) in enumerate(props)}, count
        order = {"binary_little_endian": "<", "binary_big_endian": ">"}.get(fmt)
        if order is None:
            raise ValueError(f"{path}: unsupported PLY format {fmt}")
        rec = np.dtype([(n, order + t) for n, t in props])
        arr = np.frombuffer(f.read(rec.itemsize * count), dtype=rec, count=count)
        return {n: arr[n] for n, _ in props}, count


def load_ply(path: str, max_sh_degree: int | None = None) -> dict:
    """Read a Gaussian PLY.  Returns raw parameters as float32 arrays: xyz [P,3], features [P,M,3]
    (coefficient-major: DC first), opacity [P,1], scaling [P,3], rotation [P,4], and sh_degree.  With
    `max_sh_degree` the f_rest count is checked like the reference's assert (gaussian_model.py:243)."""
    col, P = _read_vertex_table(path)
    by_index = lambda prefix: sorted((n for n in col if n.startswith(prefix)), key=lambda n: int(n.split("_")[-1]))  # noqa: E731
    rest = by_index("f_rest_")
    if len(rest) % 3:
        raise ValueError(f"{path}: {len(rest)} f_rest properties, not a multiple of 3")
    M = len(rest) // 3 + 1
    deg = int(round(M ** 0.5)) - 1
    if (deg + 1) ** 2 != M:
        raise ValueError(f"{path}: {M} SH coefficients per channel is not a square")
    if max_sh_degree is not None and len(rest) != 3 * (max_sh_degree + 1) ** 2 - 3:
        raise ValueError(f"{path}: holds SH degree {deg}, expected {max_sh_degree}")
    f32 = lambda names: np.stack([np.asarray(col[n], np.float32) for n in names], axis=1)  # noqa: E731
    feats = np.empty((P, M, 3), np.float32)
    feats[:, 0, :] = f32(["f_dc_0", "f_dc_1", "f_dc_2"])
    if M > 1:
        feats[:, 1:, :] = f32(rest).reshape(P, 3, M - 1).transpose(0, 2, 1)
    return {"xyz": f32(["x", "y", "z"]), "features": feats, "opacity": f32(["opacity"]),
            "scaling": f32(by_index("scale_")), "rotation": f32(by_index("rot_")), "sh_degree": deg}
