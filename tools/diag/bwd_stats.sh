#!/bin/bash
# Iteration accounting of k_unit_blend_bwd_sparse.  Build the instrumented library HERE first:
#   tools/diag/build_variant.sh stats -DFR_DIAG_STATS
# then on the GPU box: tools/diag/bwd_stats.sh
cd "${GRAFT_REPO_ROOT:-.}"
FR_HIP_LIB=$PWD/.ab/libfr_stats.so python - <<'PY'
import numpy as np, torch
from fateavatar_amd import rasterizer as R, scenes
dev = torch.device("cuda", 0)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)
for P, res, kw in ((100_000, 512, {}), (100_000, 512, dict(opacity=0.9)), (500_000, 1024, {})):
    s = scenes.head_scene(P=P, res=res, sh_degree=3, seed=0, **kw)
    cam = s.camera
    means, shs, op, sc, rot, bg = t(s.means3D), t(s.shs), t(s.opacities), t(s.scales), t(s.rotations), t(s.bg)
    view, proj, campos = t(cam.world_view_transform), t(cam.full_proj_transform), t(cam.camera_center)
    empty = torch.empty(0, device=dev)
    g = torch.ones(3, res, res, device=dev) / (3 * res * res)
    for _ in range(2):
        nr, color, radii, geom, binning, img = R.rasterize_gaussians(bg, means, empty, op, sc, rot, 1.0, empty, view, proj, cam.tanfovx,
                                                                   cam.tanfovy, res, res, shs, 3, campos, False, False)
        torch.cuda.synchronize()
        img[:64].view(torch.int32)[9:14] = 0   # DeviceCounts::pair_hist (fr_common.hpp)
        R.rasterize_gaussians_backward(bg, means, radii, empty, sc, rot, 1.0, empty, view, proj, cam.tanfovx, cam.tanfovy, g, shs, 3,
                                       campos, geom, nr, binning, img, False)
    torch.cuda.synchronize()
    w = img[:64].view(torch.int32).cpu().numpy()   # DeviceCounts sits at the head of the image buffer (fr_common.hpp)
    c = R.read_counts(0)
    u = max(int(w[14]), 1)
    print(f"P={P} res={res} {kw}: instances={c.num_instances} units={w[14]} ranges={w[9]} ({w[9]/u:.2f}/unit) "
          f"A_trips={w[10]} ({w[10]/u:.1f}/unit, two records each) pair_slots={w[12]} ({w[12]/u:.0f}/unit) "
          f"B_trips={w[13]} ({w[13]/u:.1f}/unit, two pixels each)")
PY
