#!/bin/bash
# Iteration accounting of k_unit_blend_bwd_sparse at config 2.  Build the instrumented library HERE first:
#   (cd fateavatar_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics \
#      -ffp-contract=fast -fno-slp-vectorize -DFR_BWD_STATS -c fr_blend.hip -o /tmp/fr_blend_stats.o && \
#    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../.ab/libfr_stats.so /tmp/fr_blend_stats.o \
#      _obj/fr_preprocess.o _obj/fr_preprocess_bwd.o _obj/fr_knn.o _obj/fr_optim.o _obj/fr_binding.o _obj/fr_api.o)
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
    g = torch.rand(3, res, res, device=dev) / (3 * res * res)
    for _ in range(2):
        nr, color, radii, geom, binning, img = R.rasterize_gaussians(bg, means, empty, op, sc, rot, 1.0, empty, view, proj, cam.tanfovx,
                                                                   cam.tanfovy, res, res, shs, 3, campos, False, False)
        R.rasterize_gaussians_backward(bg, means, radii, empty, sc, rot, 1.0, empty, view, proj, cam.tanfovx, cam.tanfovy, g, shs, 3,
                                       campos, geom, nr, binning, img, False)
    torch.cuda.synchronize()
    w = img[:64].view(torch.int32).cpu().numpy()   # DeviceCounts sits at the head of the image buffer (fr_common.hpp)
    c = R.read_counts(0)
    print(f"P={P} res={res} {kw}: instances={c.num_instances} units={w[14]} ranges={w[9]} A_iters={w[10]} B_iters={w[11]} "
          f"pair_slots={w[12]} all_pairs_units={w[13]}")
PY
