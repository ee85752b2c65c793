"""The compiled `_C` module (fateavatar_amd/csrc/torch_ext.cpp): builds against the installed PyTorch-ROCm, exports
the reference's three functions (ext.cpp:15-19) with their C++ signatures, and — on a GPU — renders the same frame
as the ctypes host."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_extension_builds_and_exports_the_reference_names():
    from fateavatar_amd import torch_ext
    torch_ext.build()          # no-op when __graft_entry__.build() already made it
    m = torch_ext.load()
    doc = m.rasterize_gaussians.__doc__
    # 19 positional arguments, (int, 5 tensors) out — rasterize_points.h:18-38
    assert doc.count("arg") == 19 and "-> tuple[int, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]" in doc
    assert m.rasterize_gaussians_backward.__doc__.count("arg") == 21      # rasterize_points.h:40-62
    assert m.rasterize_gaussians_backward.__doc__.count("torch.Tensor") == 15 + 8
    assert m.mark_visible.__doc__.count("arg") == 3 and m.distCUDA2.__doc__.count("arg") == 1


def test_alias_package_resolves_to_the_extension_on_request():
    code = ("import diff_gaussian_rasterization._C as c, simple_knn._C as k; "
            "print(type(c.rasterize_gaussians).__name__, type(k.distCUDA2).__name__)")
    env = dict(os.environ, FR_USE_TORCH_EXT="1", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.stdout.split() == ["builtin_function_or_method", "builtin_function_or_method"], (r.stdout, r.stderr[-2000:])


def test_cpu_tensors_are_refused():
    import torch
    from fateavatar_amd import torch_ext
    m = torch_ext.load()
    e = torch.empty(0)
    with pytest.raises(RuntimeError, match="HIP device"):
        m.rasterize_gaussians(torch.ones(3), torch.zeros(4, 3), e, torch.ones(4, 1), torch.ones(4, 3), torch.ones(4, 4), 1.0, e,
                              torch.eye(4), torch.eye(4), 0.2, 0.2, 16, 16, torch.zeros(4, 1, 3), 0, torch.zeros(3), False, False)


@pytest.mark.gpu
def test_extension_frame_equals_ctypes_frame(gpu_device):
    import torch
    from fateavatar_amd import rasterizer, scenes, torch_ext
    from simple_knn._C import distCUDA2
    m = torch_ext.load()
    s = scenes.head_scene(P=20000, res=256, sh_degree=3, seed=0, opacity=0.4)
    c = s.camera
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)  # noqa: E731
    e = torch.empty(0)
    args = (t(s.bg), t(s.means3D), e, t(s.opacities), t(s.scales), t(s.rotations), 1.0, e, t(c.world_view_transform),
            t(c.full_proj_transform), c.tanfovx, c.tanfovy, 256, 256, t(s.shs), 3, t(c.camera_center), False, False)
    R1, col1, rad1, g1, b1, i1 = m.rasterize_gaussians(*args)
    R2, col2, rad2, g2, b2, i2 = rasterizer.rasterize_gaussians(*args)
    assert R1 == R2 > 0 and torch.equal(rad1, rad2) and torch.equal(col1, col2)
    dpix = t((np.random.default_rng(0).uniform(-1, 1, (3, 256, 256)) / 65536).astype(np.float32))
    bargs = lambda R, rad, g, b, i: (args[0], args[1], rad, e, args[4], args[5], 1.0, e, args[8], args[9], c.tanfovx, c.tanfovy,  # noqa: E731
                                     dpix, args[14], 3, args[16], g, R, b, i, False)
    o1 = m.rasterize_gaussians_backward(*bargs(R1, rad1, g1, b1, i1))
    o2 = rasterizer.rasterize_gaussians_backward(*bargs(R2, rad2, g2, b2, i2))
    assert len(o1) == 8
    for a, b in zip(o1, o2):
        assert a.shape == b.shape
        d = float((a - b).norm() / b.norm().clamp_min(1e-30))
        assert d < 1e-5, d      # same kernels; only the order of the float atomics differs
    assert torch.equal(m.mark_visible(args[1], args[8], args[9]), rasterizer.mark_visible(args[1], args[8], args[9]))
    pts = t(s.means3D[:5000])
    assert torch.equal(m.distCUDA2(pts), distCUDA2(pts))
    # P == 0 short-circuit (rasterize_points.cu:81)
    z = m.rasterize_gaussians(args[0], torch.zeros(0, 3, device=gpu_device), e, e, e, e, 1.0, e, args[8], args[9], c.tanfovx,
                              c.tanfovy, 32, 32, e, 0, args[16], False, False)
    assert z[0] == 0 and float(z[1].abs().max()) == 0.0
