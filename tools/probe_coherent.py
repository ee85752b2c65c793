"""Stage timings of the head scene with a spatially COHERENT Gaussian order (sorted by screen tile), as the reference's
UV-raster initialisation produces, vs the random order of the bench scene."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd import scenes, rasterizer, _lib
from tests.util import HipFrame
dev = torch.device("cuda:0")
for name in ("random", "coherent"):
    s = scenes.head_scene()
    if name == "coherent":
        c = s.camera
        p = np.concatenate([s.means3D, np.ones((s.P, 1), np.float32)], 1) @ c.full_proj_transform
        xy = p[:, :2] / (p[:, 3:4] + 1e-7)
        key = (np.floor((xy[:, 1] + 1) * 32).astype(np.int64) * 64 + np.floor((xy[:, 0] + 1) * 32).astype(np.int64))
        o = np.argsort(key, kind="stable")
        for a in ("means3D", "scales", "rotations", "opacities", "shs"):
            setattr(s, a, np.ascontiguousarray(getattr(s, a)[o]))
    f = HipFrame(s, dev)
    cam = s.camera
    g = torch.full((3, 512, 512), 1.0 / (3 * 512 * 512), device=dev)
    def fwd():
        return rasterizer.rasterize_gaussians(f.bg, f.means3D, f.colors, f.op, f.scales, f.rots, 1.0, f.cov, f.view, f.proj,
                                              cam.tanfovx, cam.tanfovy, 512, 512, f.sh, s.sh_degree, f.campos, False, False)
    def bwd(r):
        return rasterizer.rasterize_gaussians_backward(f.bg, f.means3D, r[2], f.colors, f.scales, f.rots, 1.0, f.cov, f.view,
                                                       f.proj, cam.tanfovx, cam.tanfovy, g, f.sh, s.sh_degree, f.campos,
                                                       r[3], r[0], r[4], r[5], False)
    for _ in range(5):
        bwd(fwd())
    torch.cuda.synchronize()
    _lib.profile_enable(0, True)
    for _ in range(30):
        bwd(fwd())
    torch.cuda.synchronize()
    prof = _lib.profile_read(0)
    _lib.profile_enable(0, False)
    print(name, f.counts.num_instances, {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items() if v[1]})
