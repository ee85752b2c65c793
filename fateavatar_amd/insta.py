"""INSTA-layout sequences (reference: train/dataset.py:395-575, the `insta` dataset).

A sequence is a `transforms_*.json`-style dict — global intrinsics `fl_x, fl_y, cx, cy, w, h` and per frame a 4x4
camera-to-world `transform_matrix` — plus, per frame, the posed mesh the Gaussians are bound to.  The reference gets
the posed mesh from FLAME (expression / jaw / eye parameters per frame); the FLAME weights are not in the repository,
so `synthetic_sequence` poses the head template with an expression-free rigid + jaw-like motion instead (a declared
stand-in).  `cameras_from_transforms` is the reference's conversion of the file to what
`Camera(R, T, FoVx, FoVy)` takes (dataset.py:474-480 with rot_camera=True, :528-548).
"""
from __future__ import annotations

import math

import numpy as np

from . import scenes


def cameras_from_transforms(transform: dict):
    """[(R, T)], fovx, fovy, (H, W): world_mat = inv(c2w); R = world_mat[:3, :3]^T; T = world_mat[:3, 3]
    (dataset.py:474-480); fov from the normalised focal length and principal point (dataset.py:528-548)."""
    w, h = int(transform["w"]), int(transform["h"])
    focal_cxcy = [0.5 * transform["fl_x"] / transform["cx"], 0.5 * transform["fl_y"] / transform["cy"],
                  transform["cx"] / w, transform["cy"] / h]
    fovx = 2 * math.atan2(focal_cxcy[2], focal_cxcy[0])
    fovy = 2 * math.atan2(focal_cxcy[3], focal_cxcy[1])
    poses = []
    for frame in transform["frames"]:
        c2w = np.asarray(frame["transform_matrix"], np.float32)
        world_mat = np.linalg.inv(c2w)
        poses.append((world_mat[:3, :3].T.copy(), world_mat[:3, 3].copy()))
    return poses, fovx, fovy, (h, w)


def camera_arrays(transform: dict):
    """One scenes.CameraArrays per frame (what render() reads)."""
    poses, fovx, fovy, (h, w) = cameras_from_transforms(transform)
    return [scenes.make_camera(R, T, fovx, fovy, h, w) for R, T in poses]


def _rot(axis, ang):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)


def synthetic_sequence(n_frames: int = 32, res: int = 512, seed: int = 0):
    """(transform dict in INSTA layout, posed vertices [n_frames, V, 3] float32, faces [F, 3] int32).

    Camera: the config-2 camera (R = diag(1,-1,-1), T = (0, 1.47, 0.98), tan(fov/2) = 0.2) orbiting the head by up to
    +-12 degrees, written as camera-to-world matrices.  Mesh: the head template under a rigid nod / turn of a few
    degrees about the neck and a jaw-like opening (vertices of the lower face rotate about a hinge through the
    temporomandibular region, with a smooth weight), expression-free."""
    rng = np.random.default_rng(seed)
    verts, faces, _ = scenes.head_geometry()
    V = verts.astype(np.float64)
    ctr = np.array([0.0, 1.47, 0.0])
    fl = 0.5 * res / 0.2
    frames = []
    posed = np.zeros((n_frames, V.shape[0], 3), np.float32)
    # jaw-like weight: below the mouth line and towards the front of the face
    y, z = V[:, 1], V[:, 2]
    # (template: chin at y ~ 1.41, mouth line ~ 1.465, eyes ~ 1.53, face front at z ~ 0.09, neck down to y = 1.31)
    wj = np.clip((1.468 - y) / 0.03, 0, 1) * np.clip((z - 0.0) / 0.04, 0, 1)
    hinge = np.array([0.0, 1.50, -0.01])
    neck = np.array([0.0, 1.36, -0.04])
    for f in range(n_frames):
        t = f / max(1, n_frames - 1)
        # ---- camera orbit (world_to_view(x) = R^T x + T as in scenes.make_camera)
        ang = math.radians(12.0) * math.sin(2 * math.pi * t)
        Ry = _rot((0, 1, 0), ang)
        R0 = np.diag([1.0, -1.0, -1.0])
        T0 = np.array([0.0, 1.47, 0.98])
        Rn = Ry.T @ R0
        Tn = R0.T @ (ctr - Ry @ ctr) + T0
        w2c = np.eye(4)
        w2c[:3, :3] = Rn.T
        w2c[:3, 3] = Tn
        frames.append({"transform_matrix": np.linalg.inv(w2c).tolist(), "file_path": f"images/{f:05d}.png"})
        # ---- mesh motion
        jaw = math.radians(9.0) * (0.5 - 0.5 * math.cos(2 * math.pi * (2 * t + 0.1))) + math.radians(0.5) * rng.standard_normal()
        Rj = _rot((1, 0, 0), jaw)
        P = V + wj[:, None] * (((V - hinge) @ Rj.T + hinge) - V)
        Rg = _rot((0, 1, 0), math.radians(4.0) * math.sin(2 * math.pi * t + 0.7)) @ _rot((1, 0, 0), math.radians(2.5) * math.cos(2 * math.pi * t))
        P = (P - neck) @ Rg.T + neck + np.array([0.002, 0.001, 0.0]) * math.sin(4 * math.pi * t)
        posed[f] = P.astype(np.float32)
    transform = {"fl_x": fl, "fl_y": fl, "cx": res / 2, "cy": res / 2, "w": res, "h": res, "frames": frames}
    return transform, posed, faces.astype(np.int32)
