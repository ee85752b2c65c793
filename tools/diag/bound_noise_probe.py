"""GPU box: how far apart are two runs of the SAME route (the order of float atomics), against the binding inside the
kernels vs the binding op?  Re-runs tests/test_gpu_avatar.py::test_binding_inside_the_kernels_on_random_configurations
with its comparison recording the aggregate errors: worst of six runs per seed for (folded, op), (op, op), (folded, folded).
Round 4: the three columns agree — up to 2.5e-5 for a few hundred image-sized splats, 1e-8 .. 1e-5 otherwise — which is where
the test's 1e-4 bound comes from."""
import sys, os, types
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import tests.test_gpu_avatar as T
src = open(T.__file__).read()
# reuse the test body: exec it with a hook that records the errors instead of asserting
body = src[src.index("def test_binding_inside_the_kernels_on_random_configurations"):]
body = body[:body.index("\n\n\ndef ", 10)] if "\n\n\ndef " in body[10:] else body
body = body.replace("    o_f, b_f, l_f, v_f = run(True)\n    o_u, b_u, l_u, v_u = run(False)",
                    "    o_f, b_f, l_f, v_f = run(MODE[0])\n    o_u, b_u, l_u, v_u = run(MODE[1])")
body = body.replace("close = lambda a, b: float((a - b).norm()) <= 1e-4 * float(b.norm()) + 1e-12",
                    "close = lambda a, b: (ERR.append(float((a - b).norm()) / max(float(b.norm()), 1e-30)), True)[1]")
ns = dict(np=np, pytest=T.pytest, MODE=[True, False], ERR=[])
exec(body, ns)
dev = torch.device("cuda:0")
for seed in range(8):
    for mode in ((True, False), (False, False), (True, True)):
        worst = 0.0
        for rep in range(6):
            ns["MODE"][:] = mode; ns["ERR"].clear()
            ns["test_binding_inside_the_kernels_on_random_configurations"](dev, seed)
            worst = max(worst, max(ns["ERR"]))
        print(seed, mode, f"{worst:.2e}", flush=True)
