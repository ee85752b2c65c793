"""Fuzz campaign: random scene configurations, HIP vs oracle (forward state, image, gradients).  GPU box.
    python tools/fuzz_parity.py N SEED [big|huge]   (FR_FUZZ_ONLY=k: only iteration k; FR_FUZZ_CAMERA=1: random look-at cameras; FR_FUZZ_INPUTS=1: colors_precomp / cov3D_precomp / scale_modifier drawn per case; huge: 60 k - 400 k Gaussians, 512 - 1400 pixels a side)
The configurations come from tests/util.fuzz_stream; a failing iteration k is replayed with tools/diag/fuzz_replay.py /
fuzz_bisect.py and pinned in tests/test_gpu_configs.py (test_fuzz_regression_*)."""
import itertools
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd import scenes  # noqa: E402
from tests import util  # noqa: E402
from tests.test_gpu_parity import _check_backward_capped, _check_forward  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1234
big = ("huge" if sys.argv[3] == "huge" else True) if len(sys.argv) > 3 else False
bad = 0
kinds = {}
for it, P, H, W, kw, dpix, name in itertools.islice(util.fuzz_stream(seed, big), n):
    if os.environ.get("FR_FUZZ_ONLY") and it != int(os.environ["FR_FUZZ_ONLY"]):
        continue
    if os.environ.get("FR_FUZZ_PRINT_KW"):
        print("kw", it, dict(P=P, H=H, W=W, **kw), flush=True)
    try:
        s = scenes.random_scene(P, H, W, **kw)
        if os.environ.get("FR_FUZZ_CAMERA"):   # a random look-at camera instead of the identity view (util.fuzz_camera)
            s.camera = util.fuzz_camera(seed, it, H, W)
        extra = util.fuzz_inputs(seed, it, s) if os.environ.get("FR_FUZZ_INPUTS") else {}   # the API's optional inputs, drawn per case
        name += (" inputs=" + ",".join(sorted(extra))) if extra else ""
        o = util.oracle_forward(s, **extra)
        h = util.HipFrame(s, dev, **extra)
        _check_forward(o, h, name)
        # (as tests/test_gpu_configs.py: at most 5 % of the rows exempt by threshold flips, else the flips are masked out of
        # dL/dpixel and no row is exempt; aggregate bound 1e-4 for every scene)
        _check_backward_capped(o, h, dpix, name, max_skip_frac=0.05)
        print("ok  ", name, "inst", h.counts.num_instances, "maxlist", h.counts.max_tile_list, flush=True)
    except Exception as e:
        bad += 1
        kind = "skipped-row bound" if "exempt too many rows" in repr(e) else "PARITY"
        kinds[kind] = kinds.get(kind, 0) + 1
        print("FAIL", kind, name, repr(e)[:300], flush=True)
        traceback.print_exc(limit=2)
print("failures:", bad, kinds)
