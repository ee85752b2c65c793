"""CPU: the Python host mirrors the reference operator interface (names, argument order, errors)."""
import numpy as np
import pytest
import torch

import diff_gaussian_rasterization as dgr
from fateavatar_amd import rasterizer, scenes
from fateavatar_amd.model import FlatGaussians


def _settings(**kw):
    base = dict(image_height=32, image_width=32, tanfovx=0.2, tanfovy=0.2, bg=torch.ones(3), scale_modifier=1.0,
                viewmatrix=torch.eye(4), projmatrix=torch.eye(4), sh_degree=0, campos=torch.zeros(3),
                prefiltered=False, debug=False)
    base.update(kw)
    return dgr.GaussianRasterizationSettings(**base)


def test_settings_fields_match_reference_order():
    # reference diff_gaussian_rasterization/__init__.py:157-169
    assert dgr.GaussianRasterizationSettings._fields == (
        "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "debug")


def test_alias_packages_expose_reference_names():
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer, _C  # noqa: F401
    from simple_knn._C import distCUDA2  # noqa: F401
    assert callable(_C.rasterize_gaussians) and callable(_C.rasterize_gaussians_backward) and callable(_C.mark_visible)


def test_argument_validation_messages_match_reference():
    r = dgr.GaussianRasterizer(_settings())
    P = 4
    m3, m2, op = torch.zeros(P, 3), torch.zeros(P, 3), torch.ones(P, 1)
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(m3, m2, op, scales=torch.ones(P, 3), rotations=torch.ones(P, 4))
    with pytest.raises(Exception, match="Please provide excatly one of either SHs or precomputed colors!"):
        r(m3, m2, op, shs=torch.zeros(P, 1, 3), colors_precomp=torch.zeros(P, 3), scales=torch.ones(P, 3),
          rotations=torch.ones(P, 4))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m3, m2, op, shs=torch.zeros(P, 1, 3))
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair or precomputed 3D covariance"):
        r(m3, m2, op, shs=torch.zeros(P, 1, 3), scales=torch.ones(P, 3), rotations=torch.ones(P, 4),
          cov3D_precomp=torch.zeros(P, 6))


def test_cpu_tensors_are_rejected_loudly():
    """No silent CPU fallback: the product path needs the HIP device."""
    r = dgr.GaussianRasterizer(_settings())
    P = 4
    with pytest.raises(RuntimeError, match="no CPU path"):
        r(torch.zeros(P, 3), torch.zeros(P, 3), torch.ones(P, 1), shs=torch.zeros(P, 1, 3), scales=torch.ones(P, 3),
          rotations=torch.ones(P, 4))
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError, match="no CPU path"):
        distCUDA2(torch.zeros(8, 3))


def test_means3d_shape_check_matches_reference_glue():
    # rasterize_points.cu:57-59
    with pytest.raises(RuntimeError, match=r"means3D must have dimensions \(num_points, 3\)"):
        rasterizer.rasterize_gaussians(torch.ones(3), torch.zeros(5, 2), torch.empty(0), torch.ones(5, 1),
                                       torch.ones(5, 3), torch.ones(5, 4), 1.0, torch.empty(0), torch.eye(4),
                                       torch.eye(4), 0.2, 0.2, 8, 8, torch.zeros(5, 1, 3), 0, torch.zeros(3), False,
                                       False)


def test_flat_gaussians_getters_reproduce_activated_values():
    s = scenes.random_scene(50, 16, 16, sh_degree=1, seed=0)
    pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 1, "cpu")
    np.testing.assert_allclose(pc.get_xyz.detach().numpy(), s.means3D, rtol=0, atol=0)
    np.testing.assert_allclose(pc.get_opacity.detach().numpy(), s.opacities, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pc.get_scaling.detach().numpy(), s.scales, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pc.get_rotation.detach().numpy(), s.rotations, rtol=1e-5, atol=1e-6)
    assert pc.get_features.shape == (50, 4, 3)
    # gradients of every field land in ONE flat buffer
    (pc.get_opacity.sum() + pc.get_scaling.sum() + pc.get_xyz.sum()).backward()
    flat = pc.collect_grads()
    assert flat.shape == pc.flat.shape and flat.abs().sum() > 0
    assert torch.equal(pc.grad_of("_xyz"), torch.ones(50, 3))
    assert torch.equal(flat[:150].view(50, 3), torch.ones(50, 3))
    assert flat[150:150 + 50 * 4 * 3].abs().sum() == 0  # features had no gradient in this graph
    pc.begin_step()
    assert pc.grad_of("_scaling") is None


def test_head_scene_and_view_orbit():
    s0 = scenes.head_scene(P=2000, res=64, view=0, n_views=4)
    s1 = scenes.head_scene(P=2000, res=64, view=2, n_views=4)
    assert s0.means3D.shape == (2000, 3) and s0.shs.shape == (2000, 16, 3)
    np.testing.assert_array_equal(s0.means3D, s1.means3D)  # replicated Gaussians, different cameras
    assert not np.allclose(s0.camera.world_view_transform, s1.camera.world_view_transform)
    # both cameras look at the head: the head centre projects near the image centre
    for s in (s0, s1):
        c = np.array([0.0, 1.47, 0.0, 1.0], np.float32) @ s.camera.full_proj_transform
        assert abs(c[0] / c[3]) < 0.3 and abs(c[1] / c[3]) < 0.3 and c[3] > 0.2


def test_fused_adam_and_train_step_refuse_cpu_tensors():
    """No CPU fallback for the fused optimizer either: host-side argument checks fail loudly."""
    import pytest
    import torch
    from fateavatar_amd.optim import FusedAdam
    p, g = torch.zeros(16), torch.zeros(16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        FusedAdam(p, g, [(16, 1e-3)])


def test_flat_gaussians_resize_row_map():
    """Prune + append on the flat parameter holder (model/fateavatar.py:610-711 semantics): surviving rows keep their
    values and order, appended rows follow, and the returned row map is what optimizer state must follow."""
    import numpy as np
    import torch
    from fateavatar_amd.model import FlatGaussians
    P, M = 10, 4
    rng = np.random.default_rng(0)
    pc = FlatGaussians(rng.normal(size=(P, 3)).astype(np.float32), rng.normal(size=(P, M, 3)).astype(np.float32),
                       rng.uniform(0.1, 0.9, (P, 1)).astype(np.float32), rng.uniform(0.1, 0.2, (P, 3)).astype(np.float32),
                       rng.normal(size=(P, 4)).astype(np.float32), 1, torch.device("cpu"))
    old = {n: getattr(pc, n).detach().clone() for n, _ in pc.FIELDS}
    keep = torch.tensor([1, 0, 1, 1, 0, 1, 1, 1, 0, 1], dtype=torch.bool)
    new_rows = [old[n][[2, 5]].clone() for n, _ in pc.FIELDS]
    row_map = pc.resize(keep, new_rows)
    assert pc.P == 9 and row_map.tolist() == [0, 2, 3, 5, 6, 7, 9, -1, -1]
    assert pc.flat.numel() == sum(pc.widths()) * pc.P == pc.flat_grad.numel()
    for n, _ in pc.FIELDS:
        p = getattr(pc, n)
        assert isinstance(p, torch.nn.Parameter) and p.data_ptr() >= pc.flat.data_ptr()
        assert torch.equal(p[:7], old[n][keep]) and torch.equal(p[7:], old[n][[2, 5]])


def test_avatar_gaussians_can_be_restored_in_another_order():
    """AvatarGaussians.resize(order=...): a permutation re-stores parameters AND binding; the row map is the order."""
    import numpy as np
    import pytest
    import torch
    from fateavatar_amd.avatar import AvatarGaussians
    rng = np.random.default_rng(1)
    P = 11
    pc = AvatarGaussians(rng.integers(0, 50, P), rng.uniform(0, 1, (P, 3)).astype(np.float32), -6.0, torch.device("cpu"))
    with torch.no_grad():
        for n, _ in pc.FIELDS:
            getattr(pc, n).copy_(torch.from_numpy(rng.normal(size=tuple(getattr(pc, n).shape)).astype(np.float32)))
    old = {n: getattr(pc, n).detach().clone() for n, _ in pc.FIELDS}
    fi, bc = pc.face_index.clone(), pc.bary_coords.clone()
    order = torch.from_numpy(rng.permutation(P))
    row_map = pc.resize(order=order)
    assert torch.equal(row_map, order) and pc.P == P
    assert torch.equal(pc.face_index, fi[order]) and torch.equal(pc.bary_coords, bc[order])
    for n, _ in pc.FIELDS:
        assert torch.equal(getattr(pc, n).detach(), old[n][order])
        assert getattr(pc, n)._fr_grad_out.buf.shape == getattr(pc, n).shape
    with pytest.raises(ValueError, match="not both"):
        pc.resize(keep_mask=torch.ones(P, dtype=torch.bool), order=order)


def test_mesh_binding_refuses_cpu_tensors():
    import pytest
    import torch
    from fateavatar_amd.binding import bind_gaussians
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU path"):
        bind_gaussians(z(4, 3), z(2, 3, dtype=torch.int32), z(5, dtype=torch.int32), z(5, 3), z(2, 1), z(5, 1), z(5, 4),
                       z(5, 3), 0.01)


def test_frames_rendered_from_their_binding_refuse_cpu_tensors_and_bad_shapes():
    """bound.render_bound_batch (fr_aux::binding) has no CPU path either, and counts its views."""
    import types
    import pytest
    import torch
    from fateavatar_amd.bound import MeshBinding, render_bound_batch
    from fateavatar_amd.scenes import look_at_camera
    from fateavatar_amd.model import TorchCamera
    z = torch.zeros
    pc = types.SimpleNamespace(_opacity=z(5, 1), _offset=z(5, 1), _rotation=z(5, 4), _scaling=z(5, 3), get_features=z(5, 1, 3),
                               max_sh_degree=0)
    mb = MeshBinding(z(2, 3, dtype=torch.int32), z(5, dtype=torch.int32), z(5, 3), z(2, 1), 0.01, True)
    cam = TorchCamera(look_at_camera((0, 0, 2.0), (0, 0, 0), (0, 1, 0), 0.6, 0.6, 32, 32), torch.device("cpu"))
    with pytest.raises(RuntimeError, match="no CPU path"):
        render_bound_batch([cam], pc, z(4, 3), mb, torch.ones(3))
    with pytest.raises(RuntimeError, match="1 .. 4 views"):
        render_bound_batch([cam] * 5, pc, z(4, 3), mb, torch.ones(3))


def test_reference_render_runs_against_the_alias_packages_up_to_the_device_boundary():
    """The reference's OWN volume_rendering/render_3dgs.py (imported from the read-only checkout, present only in the
    build container) must import against the alias package and drive it with its keyword arguments; on CPU tensors the
    call has to get as far as the C-ABI boundary and stop there with the no-CPU-path error (nothing earlier may fail)."""
    import importlib
    import os
    import sys
    import pytest
    import torch
    if not os.path.exists("/root/reference/volume_rendering/render_3dgs.py"):
        pytest.skip("the reference checkout is only present in the build container")
    sys.dont_write_bytecode = True
    sys.path.insert(0, "/root/reference")
    try:
        ref = importlib.import_module("volume_rendering.render_3dgs")
    finally:
        sys.path.remove("/root/reference")
    from fateavatar_amd import scenes
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    s = scenes.random_scene(50, 32, 32, sh_degree=1, seed=0)
    pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, 1, torch.device("cpu"))
    cam = TorchCamera(s.camera, torch.device("cpu"))
    with pytest.raises(RuntimeError, match="HIP device|no CPU"):
        ref.render(cam, pc, torch.from_numpy(s.bg), device="cpu")


def test_handle_slot_is_scoped():
    """`with rasterizer.handle_slot(k):` selects the k-th fr_handle of the device for the calls inside it (frames that are
    to be in flight together take one slot each) and restores the previous slot on the way out, also on an exception."""
    from fateavatar_amd import rasterizer
    assert rasterizer._slot == 0
    with rasterizer.handle_slot(2):
        assert rasterizer._slot == 2
        with rasterizer.handle_slot(1):
            assert rasterizer._slot == 1
        assert rasterizer._slot == 2
    assert rasterizer._slot == 0
    try:
        with rasterizer.handle_slot(3):
            raise KeyError("x")
    except KeyError:
        pass
    assert rasterizer._slot == 0


def test_backward_work_list_stripe_assignment_is_exact():
    """The blend backward's work list (fateavatar_amd/csrc/fr_blend.hip, k_unit_blend_chained: "Which stripe?") deals the
    units out to 64 stripes so that the eight stripes whose slots run on XCD x hold a CONTIGUOUS run of units, and fills
    every stripe exactly: stripe j owns slots j, j + 64, ... < nu, and the two-ended cursors only work if exactly that
    many units are sent to it.  The arithmetic, restated here, must be a perfect deal for every unit count."""
    def stripe_of(u, nu):
        q, rem = divmod(nu, 64)
        x, first = 0, 0
        while x < 7:
            lo4, hi4 = 4 * x, 32 + 4 * x
            n_x = 8 * q + (min(rem - lo4, 4) if rem > lo4 else 0) + (min(rem - hi4, 4) if rem > hi4 else 0)
            if u < first + n_x:
                break
            first += n_x
            x += 1
        t = (u - first) & 7
        return (4 * x + t) if t < 4 else (32 + 4 * x + (t - 4)), x

    for nu in list(range(1, 700)) + [1023, 1024, 1025, 3914, 3916, 4096, 4097, 16383, 50001]:
        counts = [0] * 64
        last_x = 0
        for u in range(nu):
            j, x = stripe_of(u, nu)
            assert 0 <= j < 64 and (j // 4) % 8 == x          # the stripe's slots run on XCD x (slot w: XCD (w / 4) % 8)
            assert x >= last_x                                 # contiguous runs: the XCD never goes back
            last_x = x
            counts[j] += 1
        want = [(nu - j + 63) // 64 for j in range(64)]        # slots j, j + 64, ... below nu
        assert counts == want, (nu, [(j, counts[j], want[j]) for j in range(64) if counts[j] != want[j]][:4])


def test_runtime_switches_are_opt_in():
    """Importing the package leaves the process environment alone; `tune_runtime()` (or FR_TUNE_RUNTIME=1 at import) sets the two
    ROCm runtime switches unless the environment already holds a value; `runtime_env()` reports them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    keys = ("HIP_FORCE_DEV_KERNARG", "DEBUG_CLR_GRAPH_PACKET_CAPTURE")
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "import fateavatar_amd as fa\n"
            "print('A', [os.environ.get(k) for k in %r])\n"
            "r = fa.tune_runtime()\n"
            "print('B', [os.environ.get(k) for k in %r], r['late'], r['env'] == fa.runtime_env())\n" % (root, keys, keys))
    base = {k: v for k, v in os.environ.items() if k not in keys and k != "FR_TUNE_RUNTIME"}
    out = subprocess.run([sys.executable, "-c", code], env=base, capture_output=True, text=True, timeout=120).stdout
    assert "A [None, None]" in out and "B ['1', '0'] False True" in out, out
    out = subprocess.run([sys.executable, "-c", code], env=dict(base, FR_TUNE_RUNTIME="1"), capture_output=True, text=True, timeout=120).stdout
    assert "A ['1', '0']" in out, out
    out = subprocess.run([sys.executable, "-c", code], env=dict(base, DEBUG_CLR_GRAPH_PACKET_CAPTURE="1"), capture_output=True, text=True,
                         timeout=120).stdout
    assert "A [None, '1']" in out and "B ['1', '1'] False True" in out, out


def test_fuzz_stream_is_pinned():
    """The fuzz campaigns' configurations are a seeded stream (tests/util.fuzz_stream); the regression tests of
    tests/test_gpu_configs.py name cases of it by (seed, iteration).  The stream must not drift."""
    from tests import util
    _, P, H, W, kw, dpix, name = util.fuzz_case(991, 61, big=True)
    assert (P, H, W, kw["seed"], kw["sh_degree"]) == (29994, 502, 715, 94317314, 1) and dpix.shape == (3, 502, 715)
    assert name.startswith("fuzz61: P=29994 502x715 deg=1 scale=[0.0239,0.1873]")
    _, P, H, W, kw, dpix, _ = util.fuzz_case(9001, 4055)
    assert (P, H, W, kw["seed"], kw["M"]) == (2979, 79, 122, 896468609, 1)
    a, b = util.fuzz_camera(9401, 3, 100, 80), util.fuzz_camera(9401, 3, 100, 80)
    import numpy as np
    assert np.array_equal(a.world_view_transform, b.world_view_transform) and a.image_height == 100


def test_flip_pixels_finds_sub_tolerance_transmittance_changes():
    """util.flip_pixels (what the GPU parity tests treat as a threshold flip): outside the forward tolerance, OR T_final off by
    more than 0.1 % relative — a pixel near T = 1e-4 that blends one entry of alpha = 1/255 more changes by 4e-7 absolute, far
    inside 1e-5 + 1e-4 |T|, and must still be found; fp32 noise of a long product (1e-5 relative) must not."""
    import numpy as np
    from types import SimpleNamespace
    from tests import util
    rng = np.random.default_rng(0)
    col = rng.uniform(0, 1, (3, 8, 8)).astype(np.float32)
    T = np.full((8, 8), 1.0e-4, np.float32)
    T[0, :] = 0.5
    o = SimpleNamespace(color=col, final_T=T)
    got_T = T.copy()
    got_T[3, 3] *= np.float32(1.0 - 1.0 / 255.0)      # one more entry at the alpha threshold: 0.39 %
    got_T[4, 4] *= np.float32(1.0 + 1e-5)             # rounding noise
    got_T[0, 2] *= np.float32(1.0 - 1.0 / 255.0)      # the same flip at T = 0.5: also outside the plain tolerance
    bad = util.flip_pixels(o, col.copy(), got_T)
    assert bad[3, 3] and bad[0, 2] and not bad[4, 4] and int(bad.sum()) == 2
    assert not (np.abs(got_T[3, 3] - T[3, 3]) > 1e-5 + 1e-4 * T[3, 3])   # (the plain tolerance does not see the first one)
    col2 = col.copy()
    col2[1, 5, 5] += np.float32(1e-3)
    assert util.flip_pixels(o, col2, T.copy())[5, 5]
