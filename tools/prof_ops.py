import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fateavatar_amd import scenes
from fateavatar_amd.model import FlatGaussians, TorchCamera
from fateavatar_amd.render import render
dev = torch.device("cuda:0")
s = scenes.head_scene()
pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 3, dev, fused_activations=True)
cam = TorchCamera(s.camera, dev); bg = torch.from_numpy(s.bg).to(dev)
g = torch.rand(3, 512, 512, device=dev)
def frame():
    pc.begin_step(); out = render(cam, pc, bg); torch.autograd.backward(out["render"], grad_tensors=g)
for _ in range(5): frame()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as p:
    for _ in range(5): frame()
    torch.cuda.synchronize()
print(p.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
