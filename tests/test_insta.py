"""INSTA-layout sequences (fateavatar_amd/insta.py): the camera conversion of train/dataset.py:474-548 and the synthetic
stand-in sequence."""
import math

import numpy as np

from fateavatar_amd import insta, scenes


def test_frame0_is_the_config2_camera_and_conversion_inverts_c2w():
    tr, posed, faces = insta.synthetic_sequence(6, 256, seed=1)
    assert set(tr) >= {"fl_x", "fl_y", "cx", "cy", "w", "h", "frames"} and len(tr["frames"]) == 6
    poses, fovx, fovy, (h, w) = insta.cameras_from_transforms(tr)
    assert (h, w) == (256, 256) and abs(math.tan(fovx / 2) - 0.2) < 1e-9 and abs(fovx - fovy) < 1e-12
    for (R, T), frame in zip(poses, tr["frames"]):
        c2w = np.asarray(frame["transform_matrix"], np.float64)
        w2c = np.eye(4)
        w2c[:3, :3], w2c[:3, 3] = R.T, T            # getWorld2View2: Rt[:3,:3] = R^T, Rt[:3,3] = T
        np.testing.assert_allclose(w2c @ c2w, np.eye(4), atol=2e-6)
    cams = insta.camera_arrays(tr)
    ref = scenes.head_scene(P=8, res=256).camera
    np.testing.assert_array_equal(cams[0].world_view_transform, ref.world_view_transform)
    np.testing.assert_array_equal(cams[0].full_proj_transform, ref.full_proj_transform)
    assert not np.allclose(cams[2].world_view_transform, cams[0].world_view_transform)


def test_synthetic_motion_is_rigid_plus_jaw():
    tr, posed, faces = insta.synthetic_sequence(9, 64)
    verts, f0, _ = scenes.head_geometry()
    assert posed.shape == (9,) + verts.shape and np.array_equal(faces, f0)
    d = np.linalg.norm(posed - posed[0], axis=-1)           # [frames, V]
    assert 0.002 < d.max() < 0.05                           # millimetres to centimetres, head-sized units
    # edge lengths away from the jaw region are preserved by the rigid part
    upper = verts[:, 1] > 1.53
    e = faces[np.all(upper[faces], axis=1)][:500]
    l0 = np.linalg.norm(verts[e[:, 0]] - verts[e[:, 1]], axis=1)
    l4 = np.linalg.norm(posed[4][e[:, 0]] - posed[4][e[:, 1]], axis=1)
    np.testing.assert_allclose(l4, l0, rtol=1e-4, atol=1e-7)
    # ... and the lower face moves RELATIVE to the skull (the jaw-like part): chin-to-forehead distance varies over
    # the sequence, forehead-to-crown distance does not
    lower = (verts[:, 1] < 1.44) & (verts[:, 2] > 0.05)
    crown = verts[:, 1] > 1.60
    assert lower.sum() > 20 and crown.sum() > 20
    chin = posed[:, lower].mean(1)
    brow = posed[:, upper & (verts[:, 2] > 0.05)].mean(1)
    top = posed[:, crown].mean(1)
    assert np.ptp(np.linalg.norm(chin - brow, axis=1)) > 2e-3
    assert np.ptp(np.linalg.norm(top - brow, axis=1)) < 1e-5
