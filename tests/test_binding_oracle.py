"""The mesh-binding oracle (oracle/binding.py) against vectors generated from the reference's own mesh_compute
functions, plus self-consistency of the pytorch3d restatements (quaternion conversion / product)."""
import os

import numpy as np
import torch

from oracle import binding as B

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_binding.npz"))


def test_face_frame_scale_and_normals_match_the_reference():
    faces = torch.from_numpy(G["faces"])
    for b in range(G["verts"].shape[0]):
        verts = torch.from_numpy(G["verts"][b])
        orien, scale = B.face_orientation(verts, faces)
        assert torch.equal(orien, torch.from_numpy(G["orientation"][b]))     # same ops, same order: bit-identical
        assert torch.equal(scale, torch.from_numpy(G["scale"][b]))
        assert torch.equal(B.face_normals(verts, faces), torch.from_numpy(G["normals"][b]))
    assert np.isfinite(G["orientation"]).all()                              # the degenerate face went through the eps clamp


def test_matrix_to_quaternion_covers_all_branches_and_inverts():
    g = torch.Generator().manual_seed(0)
    q = torch.randn(4000, 4, generator=g, dtype=torch.float64)
    q = q / q.norm(dim=-1, keepdim=True)
    # force every candidate to be selected: dominate one component
    for k in range(4):
        q[k * 100:(k + 1) * 100, k] += 3.0
    q = B.standardize_quaternion(q / q.norm(dim=-1, keepdim=True))
    w, x, y, z = q.unbind(-1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    q2 = B.matrix_to_quaternion(R)
    sel = torch.stack([1 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2], 1 + R[:, 0, 0] - R[:, 1, 1] - R[:, 2, 2],
                       1 - R[:, 0, 0] + R[:, 1, 1] - R[:, 2, 2], 1 - R[:, 0, 0] - R[:, 1, 1] + R[:, 2, 2]], -1).argmax(-1)
    assert set(sel.tolist()) == {0, 1, 2, 3}
    assert float((q2 - q).abs().max()) < 1e-12 and bool((q2[:, 0] >= 0).all())


def test_quaternion_multiply_is_rotation_composition():
    g = torch.Generator().manual_seed(1)
    a = torch.randn(500, 4, generator=g, dtype=torch.float64)
    b = torch.randn(500, 4, generator=g, dtype=torch.float64)
    a, b = a / a.norm(dim=-1, keepdim=True), b / b.norm(dim=-1, keepdim=True)

    def rot(q):
        w, x, y, z = q.unbind(-1)
        return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                            2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                            2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    ab = B.quaternion_multiply(a, b)
    assert float((rot(ab) - rot(a) @ rot(b)).abs().max()) < 1e-12 and bool((ab[:, 0] >= 0).all())
