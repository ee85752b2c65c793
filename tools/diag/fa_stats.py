"""The blend workload of FateAvatar's own step (GPU box): instances, units, pairs per unit of the synthetic avatar frame after a
few steps, beside BASELINE config 2's — with the -DFR_DIAG_STATS build (FR_HIP_LIB=$PWD/.ab/libfr_stats.so) also the backward's
record ranges and trips."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("FR_DEBUG_PAIR_HIST", "1")
from tests.test_gpu_avatar import _setup, _targets
from fateavatar_amd.avatar import AvatarStep
from fateavatar_amd import rasterizer
dev = torch.device("cuda:0")
bg = torch.ones(3, device=dev)
for P, order in ((100_000, "random"), (100_000, "uv"), (256 * 256, "uv")):     # (uv: the reference's own initialisation)
    S = _setup(dev, P, 512, 16, order=order)
    gts = _targets(S, dev, bg)
    st = AvatarStep(S["make"](), S["faces"], S["canon"], S["cams"][0].clone(), bg, use_graph=False)
    print(f"== P {P} binding points: {order}; scale_init {np.exp(S['scale_init']):.3e}")
    for steps in (1, 3, 60):
        for it in range(steps):
            st.step(S["cams"][it % 16], S["posed"][it % 16], gts[it % 16])
        torch.cuda.synchronize()
        c = rasterizer.last_counts[0]
        pc = st.pc
        print(f"after {st.adam.step_count} steps: num_rendered {c.num_rendered} instances {c.num_instances} max_list {c.max_tile_list}; "
              f"opacity mean {float(torch.sigmoid(pc._opacity).mean()):.3f} scale mean {float(torch.exp(pc._scaling).mean()):.2e} "
              f"max {float(torch.exp(pc._scaling).max()):.2e}")
