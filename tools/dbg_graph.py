import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from fateavatar_amd import scenes, rasterizer
from fateavatar_amd.model import FlatGaussians, TorchCamera
from fateavatar_amd.render import render
dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
s = scenes.head_scene(view=int(os.environ.get("VIEW", "0")), n_views=2)
pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 3, dev, fused_activations=True)
cam = TorchCamera(s.camera, dev); bg = torch.from_numpy(s.bg).to(dev)
g = torch.rand(3, 512, 512, device=dev)
def frame():
    pc.begin_step(); out = render(cam, pc, bg); torch.autograd.backward(out["render"], grad_tensors=g)
for _ in range(5): frame()
torch.cuda.synchronize()
rasterizer.set_no_wait(True)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3): frame()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
G = torch.cuda.CUDAGraph()
with torch.cuda.graph(G): frame()
torch.cuda.synchronize()
other = torch.ones(1000, device=dev)
for i in range(20):
    G.replay()
    if mode == "hostcopy":
        h = pc.collect_grads().cpu(); pc.flat_grad.copy_(h.to(dev)); pc.flat_grad.div_(2)
    elif mode == "div":
        pc.collect_grads().div_(2)
    elif mode == "other":
        other.div_(2)
    elif mode == "othersync":
        other.div_(2); torch.cuda.synchronize()
    elif mode == "divsync":
        pc.collect_grads().div_(2); torch.cuda.synchronize()
    elif mode == "sync":
        torch.cuda.synchronize()
    torch.cuda.synchronize() if i % 5 == 4 else None
    if i % 5 == 4:
        c = rasterizer.read_counts(0); print(mode, i, c.num_rendered, c.num_instances, c.max_tile_list, c.overflow)
