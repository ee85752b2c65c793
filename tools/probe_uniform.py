"""Stage timings of a spatially UNIFORM scene vs the head scene at the same P (is binning contention-bound?)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd import scenes, rasterizer, _lib
from tests.util import HipFrame
dev = torch.device("cuda:0")
for name, s in (("head", scenes.head_scene()),
                ("uniform", scenes.random_scene(100000, 512, 512, sh_degree=3, seed=1, spread=0.19, scale_lo=0.0015, scale_hi=0.0025,
                                                opacity_lo=0.1, opacity_hi=0.1))):
    s.means3D[:, 2] = np.where(name == "uniform", 1.0 + 0.0 * s.means3D[:, 2], s.means3D[:, 2])
    f = HipFrame(s, dev)
    c = f.counts
    cam = s.camera
    H = W = 512
    g = torch.full((3, H, W), 1.0 / (3 * H * W), device=dev)
    def fwd():
        return rasterizer.rasterize_gaussians(f.bg, f.means3D, f.colors, f.op, f.scales, f.rots, 1.0, f.cov, f.view, f.proj,
                                              cam.tanfovx, cam.tanfovy, H, W, f.sh, s.sh_degree, f.campos, False, False)
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
    print(name, "num_rendered", c.num_rendered, "instances", c.num_instances, "max_list", c.max_tile_list,
          {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items() if v[1]})
