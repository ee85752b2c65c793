"""CPU: cross-check the oracle's BACKWARD formulas against float64 autograd of an independent,
differentiable torch restatement of the forward.

The restatement reuses the oracle's discrete decisions (sorted per-tile lists, n_contrib) and includes
the reference's stop-gradient on guard-band-clamped view-space x/y (backward.cu:168-176,262-264): the
reference's dL_dmeans3D is NOT the true derivative for those Gaussians, and parity means reproducing it."""
import math

import numpy as np
import pytest
import torch

from fateavatar_amd import scenes
from oracle import oracle

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_rgb(deg, sh, d):
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6] + \
            C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8]
    if deg > 2:
        r = r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10] + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + \
            C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12] + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + \
            C3[5] * z * (xx - yy) * sh[:, 14] + C3[6] * x * (xx - 3 * yy) * sh[:, 15]
    return torch.clamp_min(r + 0.5, 0.0)


def torch_render(s, f, P):
    """float64 differentiable forward.  P: dict of leaf tensors.  Returns image [3,H,W]."""
    c = s.camera
    H, W = c.image_height, c.image_width
    dd = torch.float64
    V = torch.tensor(c.world_view_transform, dtype=dd)   # row-vector convention: p' = [x y z 1] @ V
    F = torch.tensor(c.full_proj_transform, dtype=dd)
    cam = torch.tensor(c.camera_center, dtype=dd)
    m = P["means3D"]
    hom = torch.cat([m, torch.ones_like(m[:, :1])], 1)
    pv = hom @ V
    ph = hom @ F
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None] + P["ndc_delta"]
    pix = torch.stack([((ndc[:, 0] + 1) * W - 1) * 0.5, ((ndc[:, 1] + 1) * H - 1) * 0.5], 1)
    # Sigma3D
    q = P["rotations"]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)
    S = torch.diag_embed(P["scales"])
    M = R @ S
    Sig = M @ M.transpose(1, 2)
    # EWA with the reference's stop-gradient on clamped t.x / t.y
    tx, ty, tz = pv[:, 0], pv[:, 1], pv[:, 2]
    limx, limy = 1.3 * c.tanfovx, 1.3 * c.tanfovy
    cx = torch.where((tx / tz).abs() > limx, (torch.clamp(tx / tz, -limx, limx) * tz).detach(), tx)
    cy = torch.where((ty / tz).abs() > limy, (torch.clamp(ty / tz, -limy, limy) * tz).detach(), ty)
    fx, fy = W / (2 * c.tanfovx), H / (2 * c.tanfovy)
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * cx / (tz * tz), zero, fy / tz, -fy * cy / (tz * tz)], 1).view(-1, 2, 3)
    Wm = V[:3, :3].T  # view rotation acting on column vectors
    T = J @ Wm
    cov = T @ Sig @ T.transpose(1, 2)
    a, b, cc = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * cc - b * b
    conic = torch.stack([cc / det, -b / det, a / det], 1)
    d = m - cam
    d = d / d.norm(dim=1, keepdim=True)
    rgb = sh_rgb(s.sh_degree, P["shs"], d)
    op = P["opacities"][:, 0]

    img = torch.zeros((3, H, W), dtype=dd)
    bg = torch.tensor(s.bg, dtype=dd)
    gx = (W + 15) // 16
    ncon = torch.tensor(f.n_contrib.astype(np.int64))
    for tile in range(f.ranges.shape[0]):
        r0, r1 = int(f.ranges[tile, 0]), int(f.ranges[tile, 1])
        tx0, ty0 = (tile % gx) * 16, (tile // gx) * 16
        ys, xs = torch.meshgrid(torch.arange(ty0, min(ty0 + 16, H)), torch.arange(tx0, min(tx0 + 16, W)), indexing="ij")
        Tt = torch.ones(ys.shape, dtype=dd)
        Cc = torch.zeros((3,) + ys.shape, dtype=dd)
        last = ncon[ys, xs]
        for k in range(r0, r1):
            g = int(f.point_list[k])
            dx, dy = pix[g, 0] - xs, pix[g, 1] - ys
            power = -0.5 * (conic[g, 0] * dx * dx + conic[g, 2] * dy * dy) - conic[g, 1] * dx * dy
            alpha = torch.clamp_max(op[g] * torch.exp(power), 0.99)
            use = ((k - r0) < last) & (power <= 0) & (alpha >= 1.0 / 255.0)
            # entries before the last contributor that would have terminated the pixel cannot exist, so
            # `use` reproduces exactly the set of blended pairs of the float32 run
            w = torch.where(use, alpha * Tt, torch.zeros_like(Tt))
            Cc = Cc + rgb[g][:, None, None] * w
            Tt = torch.where(use, Tt * (1 - alpha), Tt)
        img[:, ys, xs] = Cc + Tt * bg[:, None, None]
    return img


@pytest.mark.parametrize("seed,deg", [(3, 3), (4, 1)])
def test_backward_matches_float64_autograd(seed, deg):
    s = scenes.random_scene(260, 48, 40, sh_degree=deg, seed=seed, M=16, spread=0.45, scale_lo=0.01, scale_hi=0.06,
                            opacity_lo=0.2, opacity_hi=0.9, bg=(0.2, 0.5, 0.9), tanfov=0.25)
    c = s.camera
    H, W = c.image_height, c.image_width
    f = oracle.forward(bg=s.bg, means3D=s.means3D, opacities=s.opacities, viewmatrix=c.world_view_transform,
                       projmatrix=c.full_proj_transform, campos=c.camera_center, tanfovx=c.tanfovx,
                       tanfovy=c.tanfovy, H=H, W=W, shs=s.shs, sh_degree=deg, scales=s.scales, rotations=s.rotations)
    rng = np.random.default_rng(1)
    dpix = rng.uniform(-1, 1, (3, H, W)).astype(np.float32)
    b = oracle.backward(f, dpix)

    dd = torch.float64
    P = {k: torch.tensor(v, dtype=dd, requires_grad=True) for k, v in
         dict(means3D=s.means3D, scales=s.scales, rotations=s.rotations, opacities=s.opacities, shs=s.shs).items()}
    P["ndc_delta"] = torch.zeros((s.P, 2), dtype=dd, requires_grad=True)
    img = torch_render(s, f, P)
    assert np.abs(img.detach().numpy() - f.color).max() < 5e-5
    (img * torch.tensor(dpix, dtype=dd)).sum().backward()

    def rel(a, ref):
        return np.linalg.norm(a - ref) / max(np.linalg.norm(ref), 1e-30)

    vis = f.radii > 0
    lim = np.abs(np.stack([(s.means3D @ c.world_view_transform[:3, 0] + c.world_view_transform[3, 0]),
                           (s.means3D @ c.world_view_transform[:3, 1] + c.world_view_transform[3, 1])], 1)
                 / (s.means3D @ c.world_view_transform[:3, 2] + c.world_view_transform[3, 2])[:, None])
    n_clamped = int(((lim[:, 0] > 1.3 * c.tanfovx) | (lim[:, 1] > 1.3 * c.tanfovy))[vis].sum())
    assert n_clamped > 0, "scene must contain guard-band-clamped Gaussians to exercise the stop-gradient"
    assert rel(b.dL_dmeans2D[:, :2], P["ndc_delta"].grad.numpy()) < 2e-5
    assert np.abs(b.dL_dmeans2D[:, 2]).max() == 0
    assert rel(b.dL_dopacity, P["opacities"].grad.numpy()) < 2e-5
    assert rel(b.dL_dsh, P["shs"].grad.numpy()) < 2e-5
    assert rel(b.dL_dscales, P["scales"].grad.numpy()) < 2e-5
    assert rel(b.dL_drotations, P["rotations"].grad.numpy()) < 2e-5
    assert rel(b.dL_dmeans3D, P["means3D"].grad.numpy()) < 2e-5
    # culled Gaussians get exactly zero
    assert np.abs(b.dL_dmeans3D[~vis]).max() == 0 if (~vis).any() else True
