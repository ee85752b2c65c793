"""Generate the golden fixtures in tests/golden/ from the reference's importable Python.

Runs ONLY in the build container (needs /root/reference); the fixtures it writes are
plain data (inputs + expected outputs) and are committed.  Run as

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/make_golden.py

What is pinned (these are the only parts of the path the reference can execute here —
its rasterizer is CUDA and cannot be built in this image):
  golden_sh.npz      tools/gs_utils/sh_utils.py:57-117   eval_sh  (SH basis constants/signs, deg 0..3)
  golden_cov3d.npz   tools/gs_utils/general_utils.py:78-113 + volume_rendering/gaussian_model.py:30-33
                     Sigma = (R S)(R S)^T from (scale, unit quaternion r,x,y,z), 6 upper-tri floats
  golden_camera.npz  volume_rendering/camera_3dgs.py:22-72, tools/gs_utils/graphics_utils.py:51-84
                     world_view_transform / projection / full_proj_transform / camera_center
  golden_proj.npz    tools/gs_utils/graphics_utils.py:22-29 geom_transform_points: NDC / view-space coordinates of points
                     in front of the golden cameras (the 1 / (w + 1e-7) projection of forward.cu:196-200)
  golden_misc.npz    tools/gs_utils/general_utils.py:18-19 inverse_sigmoid
  golden_binding.npz volume_rendering/mesh_compute.py:27-59 compute_face_orientation (+ scale) / compute_face_normals
  ../../fateavatar_amd/data/head_template_geom.npz  vertices, triangle indices, UV coordinates (`vt`) and per-corner UV indices of
                     weights/head_template_mouth_close.obj (input geometry of BASELINE.json configs 2, 3 and 5; data, not code)
"""
import math
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

# the reference hard-codes .cuda()/device="cuda"; run it on CPU
torch.Tensor.cuda = lambda self, *a, **k: self
_zeros = torch.zeros


def _zeros_cpu(*a, **k):
    k.pop("device", None)
    return _zeros(*a, **k)


torch.zeros = _zeros_cpu

from tools.gs_utils import sh_utils, general_utils  # noqa: E402
from volume_rendering.camera_3dgs import Camera  # noqa: E402
from tools.gs_utils.graphics_utils import getProjectionMatrix  # noqa: E402


def gen_sh():
    g = torch.Generator().manual_seed(1234)
    N = 257
    dirs = torch.randn(N, 3, generator=g)
    dirs = dirs / dirs.norm(dim=1, keepdim=True)
    sh = torch.rand(N, 16, 3, generator=g) * 2 - 1  # our layout [N, M, 3]
    out = {}
    for deg in range(4):
        res = sh_utils.eval_sh(deg, sh.permute(0, 2, 1), dirs)  # [N,3]
        out[f"deg{deg}"] = res.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "golden_sh.npz"), dirs=dirs.numpy(), sh=sh.numpy(), **out)


def gen_cov3d():
    g = torch.Generator().manual_seed(99)
    N = 301
    s = torch.exp(torch.randn(N, 3, generator=g) * 0.7 - 3.0)
    q = torch.randn(N, 4, generator=g)
    q = q / q.norm(dim=1, keepdim=True)
    for mod in (1.0, 0.37):
        L = general_utils.build_scaling_rotation(mod * s, q)
        cov = L @ L.transpose(1, 2)
        sym = general_utils.strip_symmetric(cov)
        if mod == 1.0:
            c1 = sym.numpy().astype(np.float32)
        else:
            c2 = sym.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(OUT, "golden_cov3d.npz"), scales=s.numpy(), quats=q.numpy(), cov_mod1=c1,
                        cov_mod037=c2)


def gen_camera():
    cams = {}
    fov = 2 * math.atan(0.2)
    # config 2/5 camera
    specs = [("cfg2", torch.diag(torch.tensor([1.0, -1.0, -1.0])), torch.tensor([0.0, 1.47, 0.98]), fov, fov, (512, 512))]
    g = torch.Generator().manual_seed(7)
    for k in range(4):
        A = torch.randn(3, 3, generator=g)
        Q, _ = torch.linalg.qr(A)
        if torch.det(Q) < 0:
            Q[:, 0] = -Q[:, 0]
        T = torch.randn(3, generator=g)
        fx = 0.3 + 0.5 * torch.rand(1, generator=g).item()
        fy = 0.3 + 0.5 * torch.rand(1, generator=g).item()
        specs.append((f"rand{k}", Q, T, fx, fy, (240 + 16 * k, 320 - 8 * k)))
    for name, R, T, fx, fy, res in specs:
        cam = Camera(R.unsqueeze(0), T.unsqueeze(0), fx, fy, res, data_device="cpu")
        cams[name + "_R"] = R.numpy()
        cams[name + "_T"] = T.numpy()
        cams[name + "_fov"] = np.asarray([fx, fy], np.float64)
        cams[name + "_res"] = np.asarray(res, np.int32)
        cams[name + "_wvt"] = cam.world_view_transform.numpy()
        cams[name + "_proj"] = cam.projection_matrix.numpy()
        cams[name + "_full"] = cam.full_proj_transform.numpy()
        cams[name + "_center"] = cam.camera_center.numpy()
    cams["names"] = np.asarray([s[0] for s in specs])
    np.savez_compressed(os.path.join(OUT, "golden_camera.npz"), **cams)


def gen_proj():
    """tools/gs_utils/graphics_utils.py:22-29 geom_transform_points — the reference's own Python statement of the projection the
    rasterizer starts with (forward.cu:196-200: p_hom = transformPoint4x4(p, projmatrix), p_proj = p_hom / (p_hom.w + 1e-7)) —
    on points in front of every golden camera: NDC through full_proj_transform, view space through world_view_transform."""
    from tools.gs_utils.graphics_utils import geom_transform_points
    z = np.load(os.path.join(OUT, "golden_camera.npz"))
    g = torch.Generator().manual_seed(11)
    out = {"names": z["names"]}
    for name in z["names"]:
        R, T = torch.from_numpy(z[f"{name}_R"]), torch.from_numpy(z[f"{name}_T"])
        fx, fy = (float(v) for v in z[f"{name}_fov"])
        n = 256
        vz = 0.3 + 4.7 * torch.rand(n, generator=g)
        vx = (2 * torch.rand(n, generator=g) - 1) * 0.9 * math.tan(fx / 2) * vz
        vy = (2 * torch.rand(n, generator=g) - 1) * 0.9 * math.tan(fy / 2) * vz
        view = torch.stack([vx, vy, vz], 1)
        pts = (view - T[None]) @ R.T.float()            # view = R^T x + T  (getWorld2View2)  ->  x = R (view - T)
        pts = pts.float().contiguous()
        out[f"{name}_points"] = pts.numpy()
        out[f"{name}_ndc"] = geom_transform_points(pts, torch.from_numpy(z[f"{name}_full"])).numpy()
        out[f"{name}_view"] = geom_transform_points(pts, torch.from_numpy(z[f"{name}_wvt"])).numpy()
    np.savez_compressed(os.path.join(OUT, "golden_proj.npz"), **out)


def gen_misc():
    """inverse_sigmoid, RGB2SH / SH2RGB (sh_utils.py:114-118), get_expon_lr_func (general_utils.py:29-62) and the numpy
    getWorld2View2 with scene translate / scale (graphics_utils.py:38-49)."""
    from tools.gs_utils.graphics_utils import getWorld2View2
    x = torch.linspace(0.01, 0.99, 50)
    rgb = torch.linspace(-0.3, 1.4, 37)
    steps = np.asarray([-5, 0, 1, 10, 100, 999, 1000, 5000, 29999, 30000, 45000], np.float64)
    lr_a = general_utils.get_expon_lr_func(1.6e-4, 1.6e-6, max_steps=30000)
    lr_b = general_utils.get_expon_lr_func(1e-2, 1e-4, lr_delay_steps=1000, lr_delay_mult=0.01, max_steps=30000)
    lr_c = general_utils.get_expon_lr_func(0.0, 0.0)
    g = torch.Generator().manual_seed(3)
    A = torch.randn(3, 3, generator=g)
    Q, _ = torch.linalg.qr(A)
    R, t = Q.numpy().astype(np.float64), torch.randn(3, generator=g).numpy().astype(np.float64)
    tr = np.asarray([0.1, -0.2, 0.05])
    np.savez_compressed(os.path.join(OUT, "golden_misc.npz"), x=x.numpy(),
                        inverse_sigmoid=general_utils.inverse_sigmoid(x).numpy(),
                        rgb=rgb.numpy(), RGB2SH=sh_utils.RGB2SH(rgb).numpy(), SH2RGB=sh_utils.SH2RGB(rgb).numpy(),
                        lr_steps=steps, lr_a=np.asarray([lr_a(s) for s in steps]), lr_b=np.asarray([lr_b(s) for s in steps]),
                        lr_c=np.asarray([lr_c(s) for s in steps]),
                        w2v_R=R, w2v_t=t, w2v_translate=tr, w2v_plain=getWorld2View2(R, t),
                        w2v_moved=getWorld2View2(R, t, tr, 1.7))


def gen_head():
    """Geometry AND UV layout of the head template: what the reference reads with pytorch3d's load_obj
    (model/fateavatar.py:120-127: verts, faces.verts_idx, aux.verts_uvs, faces.textures_idx).  Parsed with the repository's
    own OBJ reader; numeric arrays only."""
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from fateavatar_amd.obj import load_obj
    m = load_obj(os.path.join(REF, "weights", "head_template_mouth_close.obj"))
    assert m["verts"].shape == (5023, 3) and m["faces"].shape == (10006, 3)
    assert m["verts_uvs"].shape == (5150, 2) and m["faces_uvs"].shape == (10006, 3) and m["faces_uvs"].min() >= 0
    np.savez_compressed(os.path.join(OUT, "..", "..", "fateavatar_amd", "data", "head_template_geom.npz"), verts=m["verts"],
                        faces=m["faces"], verts_uvs=m["verts_uvs"], faces_uvs=m["faces_uvs"])


def gen_binding():
    """volume_rendering/mesh_compute.py:27-59: per-face frame (a0, a1, a2), face scale and unnormalised face normals —
    the mesh part of FateAvatar's Gaussian binding (model/fateavatar.py:225-233)."""
    from volume_rendering import mesh_compute
    g = torch.Generator().manual_seed(77)
    V, F = 60, 90
    verts = torch.randn(2, V, 3, generator=g) * 0.1                      # [bs, V, 3], two frames
    faces = torch.stack([torch.randperm(V, generator=g)[:3] for _ in range(F)]).int()
    verts[:, faces[5, 2].long()] = verts[:, faces[5, 1].long()]           # a degenerate (zero-area) face: eps clamps
    orien, scale = mesh_compute.compute_face_orientation(verts, faces, return_scale=True)
    normals = mesh_compute.compute_face_normals(verts, faces)
    np.savez_compressed(os.path.join(OUT, "golden_binding.npz"), verts=verts.numpy(), faces=faces.numpy(),
                        orientation=orien.numpy(), scale=scale.numpy(), normals=normals.numpy())


if __name__ == "__main__":
    gen_binding()
    gen_sh()
    gen_cov3d()
    gen_camera()
    gen_proj()
    gen_misc()
    gen_head()
    print("golden fixtures written to", OUT)
