"""Timing of the fused mesh binding (fwd + bwd) at N = 100k Gaussians on the head template vs the PyTorch op chain."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fateavatar_amd.binding import bind_gaussians, face_scale
from oracle import binding as B          # the reference's op chain in stock PyTorch (here only as the thing timed against)
from tests.test_gpu_parity import _binding_case
dev = torch.device("cuda:0")
c = _binding_case(dev, N=100000)
cs = face_scale(c["canon"], c["faces"])
def run(fn):
    leaves = [c[k].clone().requires_grad_(True) for k in ("posed", "offset", "rot", "scl")]
    def step():
        for l in leaves: l.grad = None
        out = fn(leaves)
        (out[0].sum() + out[1].sum() + out[2].sum()).backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 50 * 1e6
t_hip = run(lambda l: bind_gaussians(l[0], c["faces"], c["fi"], c["bary"], cs, l[1], l[2], l[3], 0.01))
t_pt = run(lambda l: B.bind(l[0], c["faces"], c["fi"], c["bary"], cs, l[1], l[2], l[3], 0.01))
print(f"binding fwd+bwd (incl. the 3 sums): HIP {t_hip:.1f} us   PyTorch op chain {t_pt:.1f} us")
