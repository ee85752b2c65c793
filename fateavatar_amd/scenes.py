"""Synthetic inputs for the rasterizer hot path (tests, smoke, bench).

Host-side numpy only.  Nothing here reads /root/reference at run time: the head
template geometry used by BASELINE.json configs 2/5 travels as the fixture
fateavatar_amd/data/head_template_geom.npz (package data, written by tests/golden/make_golden.py).

`make_camera` mirrors what the caller of the path hands to the rasterizer
(reference volume_rendering/camera_3dgs.py:53-72 and
tools/gs_utils/graphics_utils.py:51-80): transposed world->view matrix,
transposed projection, their product and the camera centre.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import numpy as np

_REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEAD_GEOM = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "head_template_geom.npz")


@dataclass
class CameraArrays:
    """What render() reads from a camera (reference render_3dgs.py:30-46)."""
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    world_view_transform: np.ndarray  # [4,4] f32, transposed (row-vector convention)
    full_proj_transform: np.ndarray   # [4,4] f32
    camera_center: np.ndarray         # [3] f32

    @property
    def tanfovx(self) -> float:
        return math.tan(self.FoVx * 0.5)

    @property
    def tanfovy(self) -> float:
        return math.tan(self.FoVy * 0.5)


def projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> np.ndarray:
    """graphics_utils.py:64-84 (getProjectionMatrix), float32 like the torch original."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def world_to_view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """graphics_utils.py:51-62 (getWorld2View2_torch with translate=0, scale=1)."""
    Rt = np.zeros((4, 4), np.float32)
    Rt[:3, :3] = np.asarray(R, np.float32).T
    Rt[:3, 3] = np.asarray(t, np.float32)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    Rt = np.linalg.inv(C2W)
    return Rt.astype(np.float32)


def make_camera(R, T, FoVx: float, FoVy: float, height: int, width: int, znear: float = 0.01,
                zfar: float = 100.0) -> CameraArrays:
    wvt = world_to_view(np.asarray(R, np.float32), np.asarray(T, np.float32)).T.copy()
    proj = projection_matrix(znear, zfar, FoVx, FoVy).T.copy()
    full = (wvt @ proj).astype(np.float32)
    center = np.linalg.inv(wvt)[3, :3].astype(np.float32)
    return CameraArrays(height, width, FoVx, FoVy, wvt, full, center)


def look_at_camera(eye, target, up, FoVx, FoVy, height, width) -> CameraArrays:
    """Convenience: camera at `eye` looking at `target` (+z forward, 3DGS/COLMAP convention)."""
    eye = np.asarray(eye, np.float64)
    f = np.asarray(target, np.float64) - eye
    f /= np.linalg.norm(f)
    r = np.cross(f, np.asarray(up, np.float64))  # x right (y down)
    r /= np.linalg.norm(r)
    d = np.cross(f, r)
    R = np.stack([r, d, f], axis=1)  # camera-to-world rotation (columns = camera axes)
    T = -R.T @ eye
    return make_camera(R.astype(np.float32), T.astype(np.float32), FoVx, FoVy, height, width)


@dataclass
class GaussianScene:
    """Activated Gaussian attributes exactly as render() passes them (render_3dgs.py:19-63)."""
    means3D: np.ndarray    # [P,3]
    scales: np.ndarray     # [P,3]  (after exp)
    rotations: np.ndarray  # [P,4]  (normalised, r,x,y,z)
    opacities: np.ndarray  # [P,1]  (after sigmoid)
    shs: np.ndarray        # [P,M,3]
    sh_degree: int
    bg: np.ndarray         # [3]
    camera: CameraArrays

    @property
    def P(self) -> int:
        return self.means3D.shape[0]


def random_scene(P: int, H: int, W: int, sh_degree: int = 0, seed: int = 0, tanfov: float = 0.2,
                 bg=(1.0, 1.0, 1.0), M: int | None = None, spread: float = 0.3, scale_lo: float = 0.01,
                 scale_hi: float = 0.03, opacity_lo: float = 0.1, opacity_hi: float = 0.5,
                 behind_fraction: float = 0.0) -> GaussianScene:
    """BASELINE.json config 1 family: random anisotropic Gaussians in a cube in front of the camera."""
    rng = np.random.default_rng(seed)
    M = (sh_degree + 1) ** 2 if M is None else M
    means = rng.uniform(-spread, spread, (P, 3)).astype(np.float32)
    means[:, 2] += 1.0
    if behind_fraction > 0:
        nb = int(P * behind_fraction)
        means[:nb, 2] = rng.uniform(-1.0, 0.25, nb).astype(np.float32)
    scales = rng.uniform(scale_lo, scale_hi, (P, 3)).astype(np.float32)
    q = rng.normal(size=(P, 4)).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    op = rng.uniform(opacity_lo, opacity_hi, (P, 1)).astype(np.float32)
    shs = rng.uniform(-0.5, 0.5, (P, M, 3)).astype(np.float32)
    fov = 2 * math.atan(tanfov)
    fovy = 2 * math.atan(tanfov * H / W) if H != W else fov
    cam = make_camera(np.eye(3, dtype=np.float32), np.zeros(3, np.float32), fov, fovy, H, W)
    return GaussianScene(means, scales, q.astype(np.float32), op, shs, sh_degree, np.asarray(bg, np.float32), cam)


def _procedural_head(n_lat: int = 96, n_lon: int = 128):
    """Fallback geometry when the head fixture is absent: an ellipsoid with the template's bounding box
    (x +-0.1036, y 1.313..1.627, z -0.132..0.090)."""
    u = np.linspace(0.02, math.pi - 0.02, n_lat)
    v = np.linspace(0, 2 * math.pi, n_lon, endpoint=False)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    x = 0.1036 * np.sin(uu) * np.cos(vv)
    y = 1.470 + 0.157 * np.cos(uu)
    z = -0.021 + 0.111 * np.sin(uu) * np.sin(vv)
    verts = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    faces = []
    for i in range(n_lat - 1):
        for j in range(n_lon):
            a = i * n_lon + j
            b = i * n_lon + (j + 1) % n_lon
            c = a + n_lon
            d = b + n_lon
            faces.append((a, c, b))
            faces.append((b, c, d))
    return verts, np.asarray(faces, np.int32)


def head_geometry():
    if os.path.exists(HEAD_GEOM):
        z = np.load(HEAD_GEOM)
        return z["verts"].astype(np.float32), z["faces"].astype(np.int32), "head_template"
    v, f = _procedural_head()
    return v, f, "procedural_ellipsoid"


def head_uv():
    """UV layout of the head template: (verts_uvs [5150,2] float32, faces_uvs [10006,3] int32) — `aux.verts_uvs` and
    `faces.textures_idx` of the reference's load_obj call (model/fateavatar.py:120-127), the inputs of its UV-raster
    initialisation.  None when only the procedural stand-in geometry is available."""
    if os.path.exists(HEAD_GEOM):
        z = np.load(HEAD_GEOM)
        if "verts_uvs" in z.files:
            return z["verts_uvs"].astype(np.float32), z["faces_uvs"].astype(np.int32)
    return None


def sample_mesh(verts, faces, n: int, seed: int = 0) -> np.ndarray:
    """Area-weighted face choice + barycentric = rand(3)/sum (mesh_sampling.py:166-167 style)."""
    rng = np.random.default_rng(seed)
    tri = verts[faces]  # [F,3,3]
    area = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    fid = rng.choice(len(faces), size=n, p=area / area.sum())
    bary = rng.random((n, 3))
    bary /= bary.sum(1, keepdims=True)
    pts = (tri[fid] * bary[:, :, None]).sum(1)
    return pts.astype(np.float32)


def head_scene(P: int = 100_000, res: int = 512, sh_degree: int = 3, seed: int = 0, scale: float | None = None,
               opacity: float = 0.1, view: int = 0, n_views: int = 1) -> GaussianScene:
    """BASELINE.json config 2 (P=100k, 512^2) / config 5 (P=500k, 1024^2): Gaussians sampled on the head
    template, isotropic scale = mean nearest-neighbour spacing, identity rotation, opacity 0.1, SH deg 3 with
    DC = 0 and rest ~ 0.1*U(-1,1), white background, camera R=diag(1,-1,-1), T=(0,1.47,0.98), tanfov 0.2
    (SURVEY.md §8d).  `view`/`n_views` orbit the camera around the head for the data-parallel batch (config 4)."""
    verts, faces, _ = head_geometry()
    means = sample_mesh(verts, faces, P, seed)
    if scale is None:
        # mean NN spacing measured in SURVEY.md Appendix B; scales with sqrt(area/P)
        scale = 6.085e-4 * math.sqrt(100_000 / P)
    rng = np.random.default_rng(seed + 1)
    M = (sh_degree + 1) ** 2
    shs = np.zeros((P, M, 3), np.float32)
    if M > 1:
        shs[:, 1:, :] = (0.1 * rng.uniform(-1, 1, (P, M - 1, 3))).astype(np.float32)
    scales = np.full((P, 3), scale, np.float32)
    rots = np.zeros((P, 4), np.float32)
    rots[:, 0] = 1
    op = np.full((P, 1), opacity, np.float32)
    fov = 2 * math.atan(0.2)
    R = np.diag([1.0, -1.0, -1.0]).astype(np.float32)
    T = np.asarray([0.0, 1.47, 0.98], np.float32)
    if n_views > 1 and view > 0:
        # rotate the camera about the head's vertical axis through (0, 1.47, 0): world' = Ry * (world - c) + c
        # view 0 frontal, then alternating +/- steps out to +-30 degrees
        step = (math.pi / 6) / max(1, (n_views) // 2)
        ang = ((view + 1) // 2) * step * (1.0 if view % 2 else -1.0)
        c, s = math.cos(ang), math.sin(ang)
        Ry = np.asarray([[c, 0, s], [0, 1, 0], [-s, 0, c]], np.float32)
        ctr = np.asarray([0.0, 1.47, 0.0], np.float32)
        # view(x) = R^T x + T  ->  view'(x) = R^T (Ry (x - ctr) + ctr) + T
        Rn = (Ry.T @ R).astype(np.float32)
        Tn = (R.T @ (ctr - Ry @ ctr) + T).astype(np.float32)
        R, T = Rn, Tn
    cam = make_camera(R, T, fov, fov, res, res)
    return GaussianScene(means, scales, rots, op, shs, sh_degree, np.ones(3, np.float32), cam)


def spatial_order(means3D: np.ndarray, cells: int = 32) -> np.ndarray:
    """A permutation that stores Gaussians in a spatially coherent order: grid cells of the bounding box (`cells` per axis),
    x fastest.  The rasterizer's results do not depend on the order of its inputs, its speed does: a wave of the forward
    preprocess handles 64 consecutive Gaussians, and when those are neighbours their (tile, Gaussian) instances share
    tiles — the counting pass groups their atomics and their key stores coalesce (+6.5 % frames/s at BASELINE config 2
    against a random order, EXPERIMENTS.md).  The reference's UV-raster initialisation (mesh_sampling.py:86-138) already
    produces such an order; a caller with randomly ordered Gaussians can permute its parameter arrays once with this."""
    m = np.asarray(means3D, np.float64)
    lo, hi = m.min(0), m.max(0)
    c = np.minimum(((m - lo) / np.maximum(hi - lo, 1e-12) * cells).astype(np.int64), cells - 1)
    return np.argsort((c[:, 2] * cells + c[:, 1]) * cells + c[:, 0], kind="stable")

