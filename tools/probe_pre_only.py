"""preprocess_fwd stage time for config 2 and the stress scene (variant libraries via FR_HIP_LIB; later stages may see garbage)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd import scenes, rasterizer, _lib
from tests.util import HipFrame
dev = torch.device("cuda:0")
for P, res, sc, op in ((100000, 512, None, 0.1), (500000, 1024, 6.085e-4, 0.5)):
    s = scenes.head_scene(P=P, res=res, scale=sc, opacity=op)
    f = HipFrame(s, dev)
    cam = s.camera
    def fwd():
        return rasterizer.rasterize_gaussians(f.bg, f.means3D, f.colors, f.op, f.scales, f.rots, 1.0, f.cov, f.view, f.proj,
                                              cam.tanfovx, cam.tanfovy, res, res, f.sh, s.sh_degree, f.campos, False, False)
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    _lib.profile_enable(0, True)
    for _ in range(20):
        fwd()
    torch.cuda.synchronize()
    prof = _lib.profile_read(0)
    _lib.profile_enable(0, False)
    print(os.environ.get("FR_HIP_LIB", "default")[-8:], P, round(prof["preprocess_fwd"][0] / prof["preprocess_fwd"][1] * 1e3, 1))
