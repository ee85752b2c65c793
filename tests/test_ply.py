"""PLY point-cloud format of the reference (volume_rendering/gaussian_model.py:190-269), read / written without plyfile."""
import numpy as np
import pytest
import torch

from fateavatar_amd import ply
from fateavatar_amd.model import FlatGaussians


def _rand(P, M, seed=0):
    r = np.random.default_rng(seed)
    return (r.normal(size=(P, 3)).astype(np.float32), r.normal(size=(P, M, 3)).astype(np.float32),
            r.normal(size=(P, 1)).astype(np.float32), r.normal(size=(P, 3)).astype(np.float32),
            r.normal(size=(P, 4)).astype(np.float32))


def test_header_and_layout_are_the_reference_format(tmp_path):
    xyz, f, op, sc, rot = _rand(2, 4)
    path = tmp_path / "sub" / "pc.ply"
    ply.save_ply(str(path), xyz, f, op, sc, rot)
    raw = path.read_bytes()
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(9)] + \
            ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    header = ("ply\nformat binary_little_endian 1.0\nelement vertex 2\n" + "".join(f"property float {n}\n" for n in names) +
              "end_header\n").encode()
    assert raw.startswith(header) and len(raw) == len(header) + 2 * len(names) * 4
    row0 = np.frombuffer(raw[len(header):], "<f4").reshape(2, len(names))[0]
    assert np.array_equal(row0[0:3], xyz[0]) and np.all(row0[3:6] == 0)          # normals are zeros
    assert np.array_equal(row0[6:9], f[0, 0, :])                                 # DC: r g b
    # f_rest is channel-major: all red coefficients, then green, then blue (transpose(1, 2).flatten)
    assert np.array_equal(row0[9:12], f[0, 1:, 0]) and np.array_equal(row0[12:15], f[0, 1:, 1])
    assert np.array_equal(row0[15:18], f[0, 1:, 2])
    assert row0[18] == op[0, 0] and np.array_equal(row0[19:22], sc[0]) and np.array_equal(row0[22:26], rot[0])


@pytest.mark.parametrize("M", [1, 4, 16])
def test_round_trip_is_exact(tmp_path, M):
    xyz, f, op, sc, rot = _rand(37, M, seed=M)
    p = str(tmp_path / "pc.ply")
    ply.save_ply(p, xyz, f, op, sc, rot)
    d = ply.load_ply(p)
    assert d["sh_degree"] == int(M ** 0.5) - 1
    for got, want in ((d["xyz"], xyz), (d["features"], f), (d["opacity"], op), (d["scaling"], sc), (d["rotation"], rot)):
        assert got.dtype == np.float32 and np.array_equal(got, want)
    with pytest.raises(ValueError, match="expected"):
        ply.load_ply(p, max_sh_degree=int(M ** 0.5))


def test_flat_gaussians_save_and_load(tmp_path):
    xyz, f, op, sc, rot = _rand(20, 4, seed=3)
    pc = FlatGaussians.from_raw(xyz, f, op, sc, rot, 1, torch.device("cpu"))
    p = str(tmp_path / "a.ply")
    pc.save_ply(p)
    pc2 = FlatGaussians.load_ply(p, torch.device("cpu"), max_sh_degree=1)
    assert pc2.P == 20 and pc2.M == 4 and torch.equal(pc2.flat, pc.flat)
    assert torch.allclose(pc2.get_opacity, torch.sigmoid(torch.from_numpy(op)))


def test_reads_ascii_and_rejects_other_files(tmp_path):
    p = tmp_path / "t.ply"
    names = ply.attribute_names(1)
    rows = np.arange(2 * len(names), dtype=np.float32).reshape(2, len(names))
    p.write_text("ply\nformat ascii 1.0\ncomment hand-written\nelement vertex 2\n" +
                 "".join(f"property float {n}\n" for n in names) + "end_header\n" +
                 "\n".join(" ".join(str(float(v)) for v in r) for r in rows) + "\n")
    d = ply.load_ply(str(p))
    assert d["features"].shape == (2, 1, 3) and np.array_equal(d["xyz"], rows[:, 0:3]) and np.array_equal(d["rotation"], rows[:, -4:])
    q = tmp_path / "bad.ply"
    q.write_text("not a ply\n")
    with pytest.raises(ValueError, match="not a PLY"):
        ply.load_ply(str(q))
