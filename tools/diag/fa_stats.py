import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_gpu_avatar import _setup, _targets
from fateavatar_amd.avatar import AvatarStep
from fateavatar_amd import rasterizer
dev = torch.device("cuda:0")
S = _setup(dev, 100_000, 512, 16)
bg = torch.ones(3, device=dev)
gts = _targets(S, dev, bg)
st = AvatarStep(S["make"](), S["faces"], S["canon"], S["cams"][0].clone(), bg, use_graph=False)
for it in range(4):
    st.step(S["cams"][it], S["posed"][it], gts[it])
torch.cuda.synchronize()
c = rasterizer.last_counts[0]
print("num_rendered", c.num_rendered, "instances", c.num_instances, "max_list", c.max_tile_list)
