"""Where does the HOST time of an eager frame go?  (cProfile over 200 eager frames.)"""
import sys, os, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd import scenes
from fateavatar_amd.model import FlatGaussians, TorchCamera
from fateavatar_amd.render import render
dev = torch.device("cuda:0")
s = scenes.head_scene()
pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 3, dev, fused_activations=True)
cam = TorchCamera(s.camera, dev); bg = torch.from_numpy(s.bg).to(dev)
g = torch.full((3, 512, 512), 1.0 / (3 * 512 * 512), device=dev)
def frame():
    pc.begin_step(); out = render(cam, pc, bg); torch.autograd.backward(out["render"], grad_tensors=g)
for _ in range(20): frame()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): frame()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(18)
