import sys, os, faulthandler
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from fateavatar_amd import scenes
from fateavatar_amd.model import FlatGaussians, TorchCamera
from fateavatar_amd.render import render
from fateavatar_amd.train import TrainStep
dev = torch.device("cuda:0")
P, res, views = 4000, 96, 4
truth = scenes.head_scene(P=P, res=res, sh_degree=1, seed=3, opacity=0.6)
print("scene", flush=True)
cams = [TorchCamera(scenes.head_scene(P=8, res=res, sh_degree=1, seed=3, view=v, n_views=views).camera, dev) for v in range(views)]
bg = torch.from_numpy(truth.bg).to(dev)
pc_true = FlatGaussians(truth.means3D, truth.shs, truth.opacities, truth.scales, truth.rotations, 1, dev, fused_activations=True)
with torch.no_grad():
    gts = [render(c, pc_true, bg)["render"].clone() for c in cams]
print("gts", flush=True)
pc = FlatGaussians(truth.means3D, truth.shs, truth.opacities * 0.7, truth.scales, truth.rotations, 1, dev, fused_activations=True)
cam = TorchCamera(scenes.head_scene(P=8, res=res, sh_degree=1, seed=3, view=0, n_views=views).camera, dev)
ts = TrainStep(pc, cam, bg, use_graph=(len(sys.argv) > 1))
for it in range(8):
    l = ts.step(cams[it % views], gts[it % views])
    torch.cuda.synchronize()
    print(it, float(l), flush=True)
