"""Forward-only stage timings of the head scene (variant libraries via FR_HIP_LIB; outputs may be invalid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd import scenes, rasterizer, _lib
from tests.util import HipFrame
dev = torch.device("cuda:0")
s = scenes.head_scene()
f = HipFrame(s, dev)
cam = s.camera
def fwd():
    return rasterizer.rasterize_gaussians(f.bg, f.means3D, f.colors, f.op, f.scales, f.rots, 1.0, f.cov, f.view, f.proj,
                                          cam.tanfovx, cam.tanfovy, 512, 512, f.sh, s.sh_degree, f.campos, False, False)
for _ in range(5):
    fwd()
torch.cuda.synchronize()
_lib.profile_enable(0, True)
for _ in range(30):
    fwd()
torch.cuda.synchronize()
prof = _lib.profile_read(0)
print(os.environ.get("FR_HIP_LIB", "default")[-10:], {k: round(v[0] / v[1] * 1e3, 1) for k, v in prof.items() if v[1]})
