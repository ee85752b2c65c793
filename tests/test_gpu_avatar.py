"""-m gpu: FateAvatar's own optimisation loop (BASELINE.json configs[2]) at full size — 100 k mesh-bound Gaussians,
512x512, synthetic INSTA-layout sequence — and its maintenance operations.

reference: model/fateavatar.py:225-276 (bound frame), train/iteration.py:21-89 (step + schedule), train/optim.py:15-21
(groups), config/fateavatar.yaml:34-47, model/fateavatar.py:610-731 (_uv_densify / _prune_low_opacity_points /
_reset_opacity), train/dataset.py:474-548 (INSTA cameras)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(dev, P, res, n_frames, seed=0, order="random"):
    import torch
    from fateavatar_amd import insta, mesh_sampling, scenes
    from fateavatar_amd.avatar import AvatarGaussians
    from fateavatar_amd.knn import init_scale_by_knn
    from fateavatar_amd.model import TorchCamera
    transform, posed, faces = insta.synthetic_sequence(n_frames, res, seed)
    verts, _, _ = scenes.head_geometry()
    if order == "uv":     # the reference's own initialisation (model/fateavatar.py:128-133)
        uv, fuv = scenes.head_uv()
        fi, bc = mesh_sampling.uniform_sampling_barycoords(P, uv, fuv, rng=np.random.default_rng(seed))
    else:
        fi, bc = mesh_sampling.random_sampling_barycoords(P, verts, faces, np.random.default_rng(seed))
    pts = (verts[faces[fi]] * bc[:, :, None]).sum(1).astype(np.float32)
    _, _, scale_init = init_scale_by_knn(torch.from_numpy(pts).to(dev))      # model/fateavatar.py:597-608
    cams = [TorchCamera(c, dev) for c in insta.camera_arrays(transform)]
    mk = lambda: AvatarGaussians(fi, bc, float(scale_init), dev)  # noqa: E731
    return dict(transform=transform, posed=torch.from_numpy(posed).to(dev), faces=torch.from_numpy(faces).to(dev),
                canon=torch.from_numpy(verts).to(dev), cams=cams, make=mk, scale_init=float(scale_init), fi=fi, bc=bc)


def _targets(S, dev, bg):
    """Images of a hidden ground-truth avatar: the same binding, coloured / opaque / offset differently."""
    import torch
    from fateavatar_amd.avatar import AvatarStep
    gt = S["make"]()
    g = torch.Generator(device="cpu").manual_seed(5)
    with torch.no_grad():
        gt._features_dc.copy_((torch.rand(gt.P, 1, 3, generator=g) * 2.0 - 1.0).to(dev))
        gt._opacity.fill_(float(np.log(0.6 / 0.4)))
        gt._offset.copy_((0.3 * torch.randn(gt.P, 1, generator=g)).to(dev))
    st = AvatarStep(gt, S["faces"], S["canon"], S["cams"][0], bg, use_graph=False)
    imgs = []
    for f, cam in enumerate(S["cams"]):
        st.cam.copy_from(cam)
        st.verts.copy_(S["posed"][f])
        with torch.no_grad():
            from fateavatar_amd.avatar import _BoundFrame
            from fateavatar_amd.binding import bind_gaussians
            from fateavatar_amd.render import render
            xyz, rot, scl = bind_gaussians(st.verts, st.faces, gt.face_index, gt.bary_coords, st.face_scale_canonical, gt._offset,
                                           gt._rotation, gt._scaling, st.shell_len, True)
            imgs.append(render(st.cam, _BoundFrame(xyz, gt, rot, scl, None), bg)["render"].clone())
    return imgs


def test_config3_fateavatar_loop_100k_512(gpu_device):
    """60 steps of the FateAvatar step at 100 k / 512x512 over a 16-frame synthetic INSTA sequence: the HIP-graph replay
    follows the eager step, the loss falls, gradients reach every parameter group through the binding op, nothing
    overflows."""
    import torch
    from fateavatar_amd.avatar import FATE_LRS, AvatarStep
    dev = gpu_device
    P, res, n_frames, steps = 100_000, 512, 16, 60
    S = _setup(dev, P, res, n_frames)
    assert abs(S["scale_init"] - np.log(6.085e-4)) < 0.02          # SURVEY.md §8d config 2 spacing
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)

    def run(use_graph):
        pc = S["make"]()
        cam0 = type(S["cams"][0]).__new__(type(S["cams"][0]))
        cam0.__dict__.update(S["cams"][0].__dict__)
        cam0._packed = S["cams"][0]._packed.clone()
        cam0.world_view_transform = cam0._packed[0:16].view(4, 4)
        cam0.full_proj_transform = cam0._packed[16:32].view(4, 4)
        cam0.camera_center = cam0._packed[32:35]
        st = AvatarStep(pc, S["faces"], S["canon"], cam0, bg, use_graph=use_graph)
        losses = []
        for it in range(steps):
            f = it % n_frames
            losses.append(st.step(S["cams"][f], S["posed"][f], gts[f]).clone())
        torch.cuda.synchronize()
        st.check()
        return pc, [float(x) for x in losses], st

    pc_e, loss_e, st_e = run(False)
    pc_g, loss_g, st_g = run(True)
    assert st_g._graph is not None and st_e._graph is None and st_g.overflows == 0
    assert st_g.adam.step_count == steps == st_e.adam.step_count
    assert np.mean(loss_e[-8:]) < 0.85 * np.mean(loss_e[:8]), (loss_e[:4], loss_e[-4:])
    assert np.allclose(loss_g[:16], loss_e[:16], rtol=5e-3), (loss_g[:16], loss_e[:16])
    assert np.allclose(loss_g, loss_e, rtol=3e-2), (loss_g[-4:], loss_e[-4:])
    assert torch.equal(st_g.denom, st_e.denom) and float(st_e.denom.max()) > 0
    # every group moved (gradients flow through the binding op to offset / rotation / scaling)
    fresh = S["make"]()
    for name, _ in pc_e.FIELDS:
        d = float((getattr(pc_e, name).detach() - getattr(fresh, name).detach()).abs().max())
        assert d > 0, name
    # the learning rates are the reference's (config/fateavatar.yaml:34-39): first Adam step moves a parameter by ~lr
    assert FATE_LRS == dict(opacity=0.05, offset=0.0016, color=0.0025, rotation=0.001, scaling=0.005)
    from tests import util as _u
    _u.assert_same_trajectory(pc_g.flat, pc_e.flat, "graph vs eager", tight=2e-2)


def test_config3_loop_on_the_reference_uv_raster_initialisation(gpu_device):
    """Row H on the reference's OWN initialisation (model/fateavatar.py:120-190, config/fateavatar.yaml:28 tex_size 256):
    65 536 Gaussians bound to the texel centres of the template's UV layout in row-major texel order, padded with random
    points on sampled faces.  The set renders (the head covers the image's centre), the step trains on it (loss falls),
    the captured step follows the eager one, nothing overflows — and `AvatarGaussians.from_template` builds that set."""
    import torch
    from fateavatar_amd.avatar import AvatarGaussians, AvatarStep, _BoundFrame
    from fateavatar_amd.binding import bind_gaussians
    from fateavatar_amd.render import render
    dev = gpu_device
    P, res, n_frames, steps = 256 * 256, 512, 8, 40
    S = _setup(dev, P, res, n_frames, order="uv")
    pc_t = AvatarGaussians.from_template(dev, uv_resolution=256)
    assert pc_t.P == P and np.array_equal(pc_t.face_index.cpu().numpy(), S["fi"]) and np.array_equal(pc_t.bary_coords.cpu().numpy(), S["bc"])
    assert abs(float(pc_t._scaling[0, 0]) - S["scale_init"]) < 1e-6
    # the first 59 099 rows are texels (a face per covered texel centre, strictly inside it), the rest padding on sampled faces
    assert (S["bc"][:59_099] > 0).all() and np.isin(S["fi"][59_099:], S["fi"][:59_099]).all()
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)
    # it renders: the initial (grey, opacity 0.1) set leaves a head-shaped footprint in the middle of the image
    st0 = AvatarStep(S["make"](), S["faces"], S["canon"], S["cams"][0], bg, use_graph=False)
    with torch.no_grad():
        pc0 = st0.pc
        xyz, rot, scl = bind_gaussians(S["posed"][0], st0.faces, pc0.face_index, pc0.bary_coords, st0.face_scale_canonical, pc0._offset,
                                       pc0._rotation, pc0._scaling, st0.shell_len, True)
        out = render(S["cams"][0], _BoundFrame(xyz, pc0, rot, scl, None), bg)
    img = out["render"]
    covered = (img.mean(0) < 0.999)
    assert 0.05 < float(covered.float().mean()) < 0.6 and bool(covered[res // 2, res // 2])
    assert int((out["radii"] > 0).sum()) > 0.9 * P

    def run(use_graph):
        pc = S["make"]()
        st = AvatarStep(pc, S["faces"], S["canon"], S["cams"][0].clone(), bg, use_graph=use_graph)
        losses = [st.step(S["cams"][it % n_frames], S["posed"][it % n_frames], gts[it % n_frames]).clone() for it in range(steps)]
        torch.cuda.synchronize()
        st.check()
        return pc, [float(x) for x in losses], st

    pc_e, loss_e, st_e = run(False)
    pc_g, loss_g, st_g = run(True)
    assert st_g._graph is not None and st_g.overflows == 0 and st_g.adam.step_count == steps
    assert np.mean(loss_e[-8:]) < 0.9 * np.mean(loss_e[:8]), (loss_e[:4], loss_e[-4:])
    assert np.allclose(loss_g[:16], loss_e[:16], rtol=5e-3), (loss_g[:16], loss_e[:16])
    assert torch.equal(st_g.denom, st_e.denom) and float(st_e.denom.max()) > 0


def test_binding_gradients_match_autograd_of_the_torch_ops(gpu_device):
    """The gradients the step writes into the flat buffer == autograd through stock PyTorch activations + the bound
    frame (fused activations off, no gradient slots): the fused path changes no number beyond float rounding."""
    import torch
    from fateavatar_amd.avatar import AvatarStep, _BoundFrame
    from fateavatar_amd.binding import bind_gaussians
    from fateavatar_amd.render import render
    dev = gpu_device
    S = _setup(dev, 6000, 96, 4, seed=2)
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)
    pc = S["make"]()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        pc._features_dc.add_(0.2)
        pc._offset.add_(0.1)
        # anisotropic, rotated splats: with the isotropic initial state the rotation gradient is exactly zero
        pc._scaling.add_((0.5 * torch.randn(pc.P, 3, generator=g)).to(dev))
        pc._rotation.add_((0.5 * torch.randn(pc.P, 4, generator=g)).to(dev))
    st = AvatarStep(pc, S["faces"], S["canon"], S["cams"][1], bg, use_graph=False)
    st.verts.copy_(S["posed"][1])
    st.gt.copy_(gts[1])
    st._forward_backward()
    got = {n: getattr(pc, n).grad.clone() for n, _ in pc.FIELDS}
    assert all(getattr(pc, n).grad.data_ptr() == getattr(pc, n)._fr_grad_out.buf.data_ptr() for n, _ in pc.FIELDS)
    # reference path: same values, plain tensors, PyTorch activations
    leaves = {n: getattr(pc, n).detach().clone().requires_grad_(True) for n, _ in pc.FIELDS}
    xyz, rot, scl = bind_gaussians(st.verts, st.faces, pc.face_index, pc.bary_coords, st.face_scale_canonical, leaves["_offset"],
                                   leaves["_rotation"], leaves["_scaling"], st.shell_len, True)

    class Plain:
        max_sh_degree = 0
        get_xyz = xyz
        get_opacity = torch.sigmoid(leaves["_opacity"])
        get_scaling = torch.exp(scl)
        get_rotation = torch.nn.functional.normalize(rot)
        get_features = leaves["_features_dc"]
    out = render(st.cam, Plain, bg)
    torch.nn.functional.l1_loss(out["render"], st.gt).backward()
    for n in got:
        ref = leaves[n].grad
        err = float((got[n] - ref).norm() / ref.norm().clamp_min(1e-30))
        assert err < 2e-4, (n, err)
        assert float(ref.abs().max()) > 0


def test_uv_densify_prune_reset_follow_the_reference(gpu_device):
    """_uv_densify: rows appended on the sampled rows' FACES with fresh barycentrics, copied parameters, scale * 0.75,
    zero moments, statistics reset; _prune_low_opacity_points: rows, binding, moments AND statistics masked together;
    the schedule of train/iteration.py:62-86; the step keeps training on the re-bound set."""
    import torch
    from fateavatar_amd.avatar import AvatarStep
    dev = gpu_device
    S = _setup(dev, 5000, 96, 4, seed=3)
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)
    pc = S["make"]()
    st = AvatarStep(pc, S["faces"], S["canon"], S["cams"][0], bg)
    for it in range(6):
        st.step(S["cams"][it % 4], S["posed"][it % 4], gts[it % 4])
    torch.cuda.synchronize()
    assert st._graph is not None

    def blocks(flat, rows):
        out, o = [], 0
        for w in pc.widths():
            out.append(flat[o:o + rows * w].view(rows, w).clone())
            o += rows * w
        return out

    # ---- densify (the schedule fires at global_step % 3000 == 0, i.e. also at step 0 in the reference)
    rows0 = pc.P
    p0, m0 = blocks(pc.flat, rows0), blocks(st.adam.exp_avg, rows0)
    fi0, bc0 = pc.face_index.clone(), pc.bary_coords.clone()
    did = st.maintain(3000, dict(increase_num=300, prune_interval=10 ** 9))
    assert did == {"densified": 300} and pc.P == rows0 + 300 and st._graph is None
    idx, new_bary = st.last_densify
    assert torch.equal(pc.face_index[:rows0], fi0) and torch.equal(pc.bary_coords[:rows0], bc0)
    assert torch.equal(pc.face_index[rows0:], fi0[idx])                      # same faces ...
    assert torch.equal(pc.bary_coords[rows0:], new_bary)                     # ... fresh barycentrics
    assert torch.allclose(new_bary.sum(-1), torch.ones(300, device=dev), atol=1e-6) and float(new_bary.min()) >= 0
    assert not torch.equal(new_bary, bc0[idx])
    p1, m1, v1 = blocks(pc.flat, pc.P), blocks(st.adam.exp_avg, pc.P), blocks(st.adam.exp_avg_sq, pc.P)
    for f in range(5):
        assert torch.equal(p1[f][:rows0], p0[f]) and torch.equal(m1[f][:rows0], m0[f])
        assert float(m1[f][rows0:].abs().max()) == 0.0 and float(v1[f][rows0:].abs().max()) == 0.0
        want = torch.log(torch.exp(p0[f][idx]) * 0.75) if f == 4 else p0[f][idx]
        assert torch.equal(p1[f][rows0:], want)
    assert float(st.denom.abs().max()) == 0.0 and st.denom.shape == (pc.P, 1)
    assert st.adam.step_count == 6
    l0 = float(st.step(S["cams"][0], S["posed"][0], gts[0]))
    for it in range(8):
        l1 = float(st.step(S["cams"][it % 4], S["posed"][it % 4], gts[it % 4]))
    torch.cuda.synchronize()
    assert st._graph is not None and np.isfinite(l1)

    # ---- prune
    with torch.no_grad():
        pc._opacity[::7] = -8.0                      # sigmoid(-8) < 0.005
    keep = ~(torch.sigmoid(pc._opacity) < 0.005).reshape(-1)
    rows1 = pc.P
    p2, m2 = blocks(pc.flat, rows1), blocks(st.adam.exp_avg, rows1)
    fi1, acc1, den1 = pc.face_index.clone(), st.xyz_gradient_accum.clone(), st.denom.clone()
    did = st.maintain(2000, dict(densify_interval=10 ** 9))
    assert did["pruned"] == int((~keep).sum()) > 0 and pc.P == int(keep.sum())
    for a, b in zip(blocks(pc.flat, pc.P) + blocks(st.adam.exp_avg, pc.P), p2 + m2):
        assert torch.equal(a, b[keep])
    assert torch.equal(pc.face_index, fi1[keep])
    assert torch.equal(st.xyz_gradient_accum, acc1[keep]) and torch.equal(st.denom, den1[keep]) and float(den1.max()) > 0

    # ---- opacity reset: in place, graph stays valid
    for it in range(4):
        st.step(S["cams"][it % 4], S["posed"][it % 4], gts[it % 4])
    g_before = st._graph
    did = st.maintain(60000, dict(densify_interval=10 ** 9, prune_interval=10 ** 9))
    assert did == {"opacity_reset": True} and st._graph is g_before
    assert float(torch.sigmoid(pc._opacity.detach()).max()) <= 0.01 + 1e-6
    assert float(blocks(st.adam.exp_avg, pc.P)[0].abs().max()) == 0.0
    st.step(S["cams"][0], S["posed"][0], gts[0])
    torch.cuda.synchronize()
    st.check()
    assert st.maintain(7, None) == {}


def test_checkpoint_layout_and_reference_style_load(gpu_device, tmp_path):
    """state_dict() has the reference's keys / shapes (train/trainer.py:396-435 + FateAvatar.state_dict), and a
    checkpoint in the REFERENCE's form — Gaussian attributes with a different row count next to unrelated model
    entries — loads the way deserialize_checkpoints_fateavatar does (train/deserialize.py:7-40)."""
    import torch
    from fateavatar_amd.avatar import AvatarStep
    dev = gpu_device
    S = _setup(dev, 3000, 64, 4, seed=4)
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)
    st = AvatarStep(S["make"](), S["faces"], S["canon"], S["cams"][0], bg)
    for it in range(5):
        st.step(S["cams"][it % 4], S["posed"][it % 4], gts[it % 4])
    sd = st.state_dict()
    m = sd["model"]
    assert sorted(m) == sorted(AvatarStep.GAUSSIAN_ATTRIBUTES) and sd["global_step"] == 5
    assert m["_offset"].shape == (3000, 1) and m["_features_dc"].shape == (3000, 1, 3) and m["_features_rest"].shape == (3000, 0, 3)
    assert m["_scaling"].shape == (3000, 3) and m["_rotation"].shape == (3000, 4) and m["_opacity"].shape == (3000, 1)
    assert m["face_index"].shape == (3000,) and m["bary_coords"].shape == (3000, 3)
    torch.save(sd, tmp_path / "a.pth")
    # resume: same trajectory
    for it in range(5, 9):
        st.step(S["cams"][it % 4], S["posed"][it % 4], gts[it % 4])
    st2 = AvatarStep(S["make"](), S["faces"], S["canon"], S["cams"][0], bg)
    assert st2.load_state_dict(torch.load(tmp_path / "a.pth", map_location=dev)) == []
    for it in range(5, 9):
        st2.step(S["cams"][it % 4], S["posed"][it % 4], gts[it % 4])
    torch.cuda.synchronize()
    assert st2.adam.step_count == st.adam.step_count == 9
    from tests import util as _u
    _u.assert_same_trajectory(st2.pc.flat, st.pc.flat, "checkpoint round trip", tight=2e-3)
    # a reference-style file: 'epoch', other model entries, MORE points than the fresh model, no optimizer state
    n = 3500
    g = torch.Generator().manual_seed(0)
    ref = {"epoch": 3, "global_step": 1234, "model": {
        "_offset": torch.randn(n, 1, generator=g), "_features_dc": torch.randn(n, 1, 3, generator=g),
        "_features_rest": torch.zeros(n, 0, 3), "_scaling": torch.randn(n, 3, generator=g) - 7.0,
        "_rotation": torch.randn(n, 4, generator=g), "_opacity": torch.randn(n, 1, generator=g),
        "face_index": torch.randint(0, S["faces"].shape[0], (n,), generator=g), "bary_coords": torch.rand(n, 3, generator=g),
        "flame.shapedirs": torch.zeros(4), "delta_vertex": torch.zeros(5, 3)}}
    ignored = st2.load_state_dict(ref)
    assert ignored == ["delta_vertex", "flame.shapedirs"] and st2.pc.P == n and st2._graph is None
    assert torch.equal(st2.pc._rotation.detach().cpu(), ref["model"]["_rotation"])
    assert float(st2.adam.exp_avg.abs().max()) == 0.0 and st2.denom.shape == (n, 1) and float(st2.denom.max()) == 0.0
    l = st2.step(S["cams"][0], S["posed"][0], gts[0])
    torch.cuda.synchronize()
    assert np.isfinite(float(l))


def test_batch_step_is_the_mean_gradient_step_of_its_frames(gpu_device):
    """AvatarBatchStep (K frames per step, in flight together) against what it stands for: the K frames' gradients,
    each computed alone, averaged, and ONE Adam step on the mean (the reference's batch loop, model/fateavatar.py:251-276,
    with the loss averaged over the batch, train/loss.py:92-105).  Checked for the eager lanes and the captured ones,
    over several steps, including the per-lane densification statistics."""
    import torch
    from fateavatar_amd.avatar import AvatarBatchStep, AvatarStep
    from fateavatar_amd.optim import FusedAdam
    dev = gpu_device
    K, P, res, n_frames, steps = 3, 20_000, 256, 6, 8
    S = _setup(dev, P, res, n_frames)
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)

    # ---- the reference procedure: one frame at a time, no optimizer inside; Adam by hand on the mean gradient
    pc_r = S["make"]()
    ref = AvatarStep(pc_r, S["faces"], S["canon"], S["cams"][0].clone(), bg, use_graph=False)
    adam = FusedAdam(pc_r.flat, torch.zeros_like(pc_r.flat), ref.adam_segments(), grad_scale=1.0)
    losses_ref = []
    for it in range(steps):
        mean = torch.zeros_like(pc_r.flat)
        ls = []
        for k in range(K):
            f = (it * K + k) % n_frames
            ref.cam.copy_from(S["cams"][f])
            ref.verts.copy_(S["posed"][f])
            ref.gt.copy_(gts[f])
            ref._forward_backward()
            mean.add_(pc_r.collect_grads(), alpha=1.0 / K)
            ls.append(float(ref.loss))
        adam.grad.copy_(mean)
        adam.step()
        losses_ref.append(ls)
    torch.cuda.synchronize()
    stats_ref = (ref.xyz_gradient_accum.clone(), ref.denom.clone())

    for use_graph, chain in ((False, False), (True, False), (False, True), (True, True)):
        # chain: the K frames through ONE launch chain on one stream (render_batch) instead of K lanes on K streams
        pc = S["make"]()
        st = AvatarBatchStep(pc, S["faces"], S["canon"], S["cams"][0].clone(), bg, views_per_step=K, use_graph=use_graph, chain=chain)
        losses = []
        for it in range(steps):
            fs = [(it * K + k) % n_frames for k in range(K)]
            out = st.step([S["cams"][f] for f in fs], [S["posed"][f] for f in fs], [gts[f] for f in fs])
            losses.append([float(x) for x in out])
        torch.cuda.synchronize()
        st.check()
        assert ((st._chain_graph if chain else st.lanes[0].graph) is not None) == use_graph and st.overflows == 0
        assert st.adam.step_count == steps
        assert np.allclose(losses, losses_ref, rtol=2e-3), (use_graph, chain, losses[-1], losses_ref[-1])
        from tests import util as _u
        _u.assert_same_trajectory(pc.flat, pc_r.flat, (use_graph, chain))
        assert util_rel_l2(pc.flat, pc_r.flat) < 1e-3, (use_graph, chain)
        acc, den = st.reduce_densification_stats()
        assert torch.equal(den, stats_ref[1]) and float(den.max()) == steps * K
        assert util_rel_l2(acc, stats_ref[0]) < 1e-3


def util_rel_l2(a, b):
    import torch
    return float(torch.linalg.norm((a - b).double()) / torch.linalg.norm(b.double()))


@pytest.mark.parametrize("chain", [True, False])
def test_batch_step_survives_the_maintenance_schedule(gpu_device, chain):
    """AvatarBatchStep under densify / prune / opacity reset / checkpoint: the lanes' densification statistics are folded
    together before anything reads them (every frame of every lane counts), the lanes are rebuilt over the re-bound
    point set (fresh graphs, same shared parameter storage), the opacity reset keeps the captured lanes, and a step
    restored from a checkpoint continues like the one that wrote it."""
    import torch
    from fateavatar_amd.avatar import AvatarBatchStep
    dev = gpu_device
    K = 2
    S = _setup(dev, 5000, 96, 4, seed=3)
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)
    pc = S["make"]()
    st = AvatarBatchStep(pc, S["faces"], S["canon"], S["cams"][0].clone(), bg, views_per_step=K, chain=chain)

    def graphs():   # the captured step: one graph for the launch chain, or one per lane
        return [st._chain_graph] if chain else [L.graph for L in st.lanes]

    def run(n, start=0):
        out = None
        for it in range(start, start + n):
            fs = [(it * K + k) % 4 for k in range(K)]
            out = st.step([S["cams"][f] for f in fs], [S["posed"][f] for f in fs], [gts[f] for f in fs])
        torch.cuda.synchronize()
        return [float(x) for x in out]

    run(6)
    assert all(g is not None for g in graphs())
    assert st.lanes[1].pc.flat.data_ptr() == pc.flat.data_ptr() and st.lanes[1].pc.flat_grad.data_ptr() != pc.flat_grad.data_ptr()
    acc, den = st.reduce_densification_stats()              # folds the lanes
    assert float(den.max()) == 6 * K and float(st.lanes[1].denom.abs().max()) == 0.0
    rows0 = pc.P
    did = st.maintain(3000, dict(increase_num=200, prune_interval=10 ** 9))
    assert did == {"densified": 200} and pc.P == rows0 + 200
    assert all(g is None for g in graphs()) and all(L.pc.P == pc.P and L.denom.shape == (pc.P, 1) for L in st.lanes)
    assert st.lanes[1].pc.flat.data_ptr() == pc.flat.data_ptr()
    l = run(6, 6)
    assert all(np.isfinite(l)) and all(g is not None for g in graphs())
    with torch.no_grad():
        pc._opacity[::5] = -8.0
    keep = ~(torch.sigmoid(pc._opacity) < 0.005).reshape(-1)
    den_before = st.reduce_densification_stats()[1]
    did = st.maintain(2000, dict(densify_interval=10 ** 9))
    assert did["pruned"] == int((~keep).sum()) > 0 and pc.P == int(keep.sum())
    assert torch.equal(st.denom, den_before[keep]) and float(st.denom.max()) == 6 * K
    run(4, 12)
    before = graphs()
    did = st.maintain(60000, dict(densify_interval=10 ** 9, prune_interval=10 ** 9))
    assert did == {"opacity_reset": True} and graphs() == before
    assert float(torch.sigmoid(st.lanes[1].pc._opacity.detach()).max()) <= 0.01 + 1e-6      # the lanes share the storage
    l = run(3, 16)
    st.check()
    assert all(np.isfinite(l)) and st.overflows == 0
    # checkpoint round trip: a fresh step loaded from the state continues with the same losses
    sd = st.state_dict()
    pc2 = S["make"]()
    st2 = AvatarBatchStep(pc2, S["faces"], S["canon"], S["cams"][0].clone(), bg, views_per_step=K, use_graph=False, chain=chain)
    st2.load_state_dict(sd)
    assert pc2.P == pc.P and st2.lanes[1].pc.flat.data_ptr() == pc2.flat.data_ptr()
    fs = [0, 1]
    a = [float(x) for x in st.step([S["cams"][f] for f in fs], [S["posed"][f] for f in fs], [gts[f] for f in fs])]
    b = [float(x) for x in st2.step([S["cams"][f] for f in fs], [S["posed"][f] for f in fs], [gts[f] for f in fs])]
    assert np.allclose(a, b, rtol=1e-4), (a, b)


@pytest.mark.parametrize("K", [1, 3])
def test_binding_inside_the_rasterizer_kernels_equals_the_binding_op(gpu_device, K):
    """fr_aux::binding (bound.render_bound_batch: the preprocess kernels evaluate the binding, the per-Gaussian backward
    continues through it) against `bind_gaussians` + `render_batch`: same expressions from one header, so the image, the
    radii and the bound arrays are the same BITS, and every gradient — offset, rotation, scaling, opacity, colour, the
    screen-space points, the densification statistics and dL/dverts — agrees to the order of float atomics (observed
    ~1e-6 in aggregate at this size; held to 5e-5)."""
    import torch
    from fateavatar_amd.avatar import _BoundFrame, _RawFrame
    from fateavatar_amd.binding import bind_gaussians, face_scale
    from fateavatar_amd.bound import MeshBinding, render_bound_batch
    from fateavatar_amd.render import render_batch
    dev = gpu_device
    S = _setup(dev, 20_000, 128, 4, seed=3)
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)
    canon = face_scale(S["canon"], S["faces"].to(torch.int32))
    g = torch.Generator().manual_seed(4)
    base = S["make"]()
    with torch.no_grad():
        base._features_dc.add_(0.2)
        base._offset.add_((0.2 * torch.randn(base.P, 1, generator=g)).to(dev))
        base._scaling.add_((0.5 * torch.randn(base.P, 3, generator=g)).to(dev))
        base._rotation.add_((0.5 * torch.randn(base.P, 4, generator=g)).to(dev))
        base._opacity.add_(2.0)
    faces = S["faces"].to(torch.int32).contiguous()

    def run(folded):
        leaves = [{n: getattr(base, n).detach().clone().requires_grad_(True) for n, _ in base.FIELDS} for _ in range(K)]
        verts = [S["posed"][k].clone().requires_grad_(True) for k in range(K)]
        stats = [(torch.zeros(base.P, 1, device=dev), torch.zeros(base.P, 1, device=dev)) for _ in range(K)]

        class PC:
            pass
        pcs = []
        for k in range(K):
            pc = PC()
            for n in leaves[k]:
                setattr(pc, n, leaves[k][n])
            pc.face_index, pc.bary_coords = base.face_index, base.bary_coords
            pcs.append(pc)
        cams = S["cams"][:K]
        if folded:
            outs = render_bound_batch(cams, [_RawFrame(pc, st) for pc, st in zip(pcs, stats)], verts,
                                      MeshBinding(faces, base.face_index, base.bary_coords, canon, 0.05, True), bg)
            bound = [o["bound"] for o in outs]
        else:
            frames, bound = [], []
            for k in range(K):
                xyz, rot, scl = bind_gaussians(verts[k], faces, base.face_index, base.bary_coords, canon, leaves[k]["_offset"],
                                               leaves[k]["_rotation"], leaves[k]["_scaling"], 0.05, True)
                frames.append(_BoundFrame(xyz, pcs[k], rot, scl, stats[k]))
                bound.append((xyz.detach(), rot.detach(), scl.detach()))
            outs = render_batch(cams, frames, bg)
        loss = sum(torch.nn.functional.l1_loss(o["render"], gts[k]) for k, o in enumerate(outs))
        loss.backward()
        torch.cuda.synchronize()
        return outs, bound, leaves, verts, stats

    o_f, b_f, l_f, v_f, s_f = run(True)
    o_u, b_u, l_u, v_u, s_u = run(False)
    for k in range(K):
        assert torch.equal(o_f[k]["render"], o_u[k]["render"]) and torch.equal(o_f[k]["radii"], o_u[k]["radii"])
        assert torch.equal(o_f[k]["visibility_filter"], o_u[k]["visibility_filter"])
        assert int((o_f[k]["radii"] > 0).sum()) > 1000
        for a, b in zip(b_f[k], b_u[k]):
            assert torch.equal(a, b)
        for n in l_f[k]:
            a, b = l_f[k][n].grad, l_u[k][n].grad
            assert a is not None and b is not None and a.shape == b.shape, n
            err = float((a - b).norm() / b.norm().clamp_min(1e-30))
            assert err < 5e-5 and float(b.abs().max()) > 0, (n, err)
        a, b = v_f[k].grad, v_u[k].grad
        assert float((a - b).norm() / b.norm()) < 5e-5 and float(b.abs().max()) > 0
        a, b = o_f[k]["viewspace_points"].grad, o_u[k]["viewspace_points"].grad
        assert float((a - b).norm() / b.norm()) < 5e-5
        assert torch.equal(s_f[k][1], s_u[k][1]) and float((s_f[k][0] - s_u[k][0]).abs().max()) <= 1e-4 * float(s_u[k][0].abs().max())


def test_fateavatar_step_with_and_without_the_folded_binding(gpu_device):
    """AvatarStep(fold_binding=True) (the default) follows AvatarStep(fold_binding=False) — eager and as a replayed graph."""
    import torch
    from fateavatar_amd.avatar import AvatarStep
    from tests import util as _u
    dev = gpu_device
    S = _setup(dev, 30_000, 192, 6, seed=6)
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)

    def run(fold, use_graph):
        pc = S["make"]()
        st = AvatarStep(pc, S["faces"], S["canon"], S["cams"][0].clone(), bg, use_graph=use_graph, fold_binding=fold)
        losses = [float(st.step(S["cams"][it % 6], S["posed"][it % 6], gts[it % 6])) for it in range(24)]
        torch.cuda.synchronize()
        st.check()
        return pc, losses, st

    pc_u, loss_u, st_u = run(False, False)
    for use_graph in (False, True):
        pc_f, loss_f, st_f = run(True, use_graph)
        assert (st_f._graph is not None) == use_graph
        assert np.allclose(loss_f, loss_u, rtol=2e-2), (loss_f[-4:], loss_u[-4:])
        assert torch.equal(st_f.denom, st_u.denom)
        _u.assert_same_trajectory(pc_f.flat, pc_u.flat, f"folded binding, graph={use_graph}", tight=2e-2)


@pytest.mark.parametrize("seed", range(8))
def test_binding_inside_the_kernels_on_random_configurations(gpu_device, seed):
    """The same equality (test_binding_inside_the_rasterizer_kernels_equals_the_binding_op) over random sizes: Gaussian
    counts that do not fill a wave or a workgroup (down to ONE), odd image sizes, 1 - 4 views, with and without
    resize_scale, perturbed meshes (thin and nearly degenerate faces), random shell lengths."""
    import torch
    from fateavatar_amd import mesh_sampling, scenes
    from fateavatar_amd.avatar import _BoundFrame, _RawFrame
    from fateavatar_amd.binding import bind_gaussians, face_scale
    from fateavatar_amd.bound import MeshBinding, render_bound_batch
    from fateavatar_amd.model import TorchCamera
    from fateavatar_amd.render import render_batch
    dev = gpu_device
    rng = np.random.default_rng(1000 + seed)
    P = int([1, 63, 257, 4097, 900, 12000, 30000, 2][seed])
    H, W = int(rng.integers(9, 200)), int(rng.integers(9, 200))
    K = int(rng.integers(1, 5))
    resize = bool(seed % 3 != 1)
    shell = float(rng.uniform(0.0, 0.2))
    verts, faces, _ = scenes.head_geometry()
    faces = faces.astype(np.int32)
    fi, bc = mesh_sampling.random_sampling_barycoords(P, verts, faces, rng)
    canon_v = torch.from_numpy(verts).to(dev)
    faces_t = torch.from_numpy(faces).to(dev)
    canon = face_scale(canon_v, faces_t) if resize else None
    fi_t = torch.from_numpy(np.asarray(fi, np.int32)).to(dev)
    bc_t = torch.from_numpy(np.asarray(bc, np.float32)).to(dev)
    g = torch.Generator().manual_seed(seed)
    base = dict(_opacity=torch.randn(P, 1, generator=g) + 1.0, _offset=0.5 * torch.randn(P, 1, generator=g),
                _features_dc=torch.rand(P, 1, 3, generator=g) * 2 - 1, _rotation=torch.randn(P, 4, generator=g),
                _scaling=float(np.log(0.004)) + 0.7 * torch.randn(P, 3, generator=g))
    cams, posed = [], []
    for k in range(K):
        ctr = verts.mean(0)
        eye = ctr + np.array([0.3 * rng.uniform(-1, 1), 0.2 * rng.uniform(-1, 1), 0.9 + 0.2 * rng.uniform(-1, 1)], np.float32)
        cams.append(TorchCamera(scenes.look_at_camera(eye, ctr, (0, 1, 0), 0.6, 0.6 * H / W, H, W), dev))
        v = verts * (1.0 + 0.1 * rng.uniform(-1, 1)) + 0.03 * rng.standard_normal(verts.shape).astype(np.float32)
        if seed % 2:   # squash: thin, nearly degenerate faces
            v[:, 2] *= 1e-3
        posed.append(torch.from_numpy(v.astype(np.float32)).to(dev))
    bg = torch.tensor(rng.uniform(0, 1, 3).astype(np.float32), device=dev)
    dimg = [torch.from_numpy((rng.uniform(-1, 1, (3, H, W)) / (H * W)).astype(np.float32)).to(dev) for _ in range(K)]

    def run(folded):
        leaves = [{n: t.clone().to(dev).requires_grad_(True) for n, t in base.items()} for _ in range(K)]
        vs = [p.clone().requires_grad_(True) for p in posed]

        class PC:
            pass
        pcs = []
        for k in range(K):
            pc = PC()
            for n in leaves[k]:
                setattr(pc, n, leaves[k][n])
            pcs.append(pc)
        if folded:
            outs = render_bound_batch(cams, [_RawFrame(pc, None) for pc in pcs], vs, MeshBinding(faces_t, fi_t, bc_t, canon, shell, resize), bg)
            bound = [o["bound"] for o in outs]
        else:
            frames, bound = [], []
            for k in range(K):
                xyz, rot, scl = bind_gaussians(vs[k], faces_t, fi_t, bc_t, canon, leaves[k]["_offset"], leaves[k]["_rotation"],
                                               leaves[k]["_scaling"], shell, resize)
                frames.append(_BoundFrame(xyz, pcs[k], rot, scl, None))
                bound.append((xyz.detach(), rot.detach(), scl.detach()))
            outs = render_batch(cams, frames, bg)
        torch.autograd.backward([o["render"] for o in outs], grad_tensors=dimg)
        torch.cuda.synchronize()
        return outs, bound, leaves, vs

    o_f, b_f, l_f, v_f = run(True)
    o_u, b_u, l_u, v_u = run(False)
    # (both routes end in float atomics — the blend backward's rows, dL/dverts — whose order differs from run to run: with a
    # few hundred image-sized splats two runs of the SAME route differ by up to ~3e-5 in aggregate; the bound is the
    # project's gradient tolerance)
    close = lambda a, b: float((a - b).norm()) <= 1e-4 * float(b.norm()) + 1e-12  # noqa: E731
    for k in range(K):
        assert torch.equal(o_f[k]["render"], o_u[k]["render"]) and torch.equal(o_f[k]["radii"], o_u[k]["radii"]), (seed, k)
        for a, b in zip(b_f[k], b_u[k]):
            assert torch.equal(a, b), (seed, k)
        for n in l_f[k]:
            assert close(l_f[k][n].grad, l_u[k][n].grad), (seed, k, n)
        assert close(v_f[k].grad, v_u[k].grad), (seed, k, "verts")
        assert close(o_f[k]["viewspace_points"].grad, o_u[k]["viewspace_points"].grad), (seed, k)
    if P >= 63:   # (the views do look at the Gaussians: the equalities above are not vacuous)
        assert all(int((o["radii"] > 0).sum()) > P // 4 for o in o_u)
        assert all(float(l["_rotation"].grad.abs().max()) > 0 and float(v.grad.abs().max()) > 0 for l, v in zip(l_u, v_u))


def test_restoring_the_gaussians_in_a_coherent_order_changes_no_result(gpu_device):
    """AvatarStep.sort_coherent(): parameters, binding, Adam moments and densification statistics are permuted together, so
    a run that re-stores its Gaussians half-way ends where the untouched run ends (row for row through the permutation, to
    the order of float atomics) — and `keep_coherent` does it after every uv_densify."""
    import torch
    from fateavatar_amd.avatar import AvatarStep
    from tests import util as _u
    dev = gpu_device
    S = _setup(dev, 20_000, 160, 6, seed=9)
    bg = torch.ones(3, device=dev)
    gts = _targets(S, dev, bg)

    def run(sort_at, use_graph):
        pc = S["make"]()
        st = AvatarStep(pc, S["faces"], S["canon"], S["cams"][0].clone(), bg, use_graph=use_graph)
        order = None
        for it in range(12):
            if it == sort_at:
                order = st.sort_coherent()
            st.step(S["cams"][it % 6], S["posed"][it % 6], gts[it % 6])
        torch.cuda.synchronize()
        st.check()
        return pc, st, order

    pc_a, st_a, _ = run(None, False)
    for use_graph in (False, True):
        pc_b, st_b, order = run(5, use_graph)
        assert sorted(order.tolist()) == list(range(pc_a.P)) and not torch.equal(order, torch.arange(pc_a.P, device=dev))
        assert torch.equal(st_b.coherent_order(), torch.arange(pc_b.P, device=dev))          # stored sorted now
        assert torch.equal(pc_b.face_index, pc_a.face_index[order]) and torch.equal(pc_b.bary_coords, pc_a.bary_coords[order])
        assert torch.equal(st_b.denom, st_a.denom[order])
        for n, _ in pc_a.FIELDS:
            a, b = getattr(pc_a, n).detach()[order], getattr(pc_b, n).detach()
            _u.assert_same_trajectory(b.reshape(-1), a.reshape(-1), f"{n}, graph={use_graph}", tight=2e-2)
        assert st_b.adam.step_count == st_a.adam.step_count == 12

    # keep_coherent: the set after a densification is the reference's set, stored sorted
    g1, g2 = torch.Generator(device=dev).manual_seed(3), torch.Generator(device=dev).manual_seed(3)
    pcs, sts = [], []
    for keep, g in ((False, g1), (True, g2)):
        pc = S["make"]()
        st = AvatarStep(pc, S["faces"], S["canon"], S["cams"][0].clone(), bg, keep_coherent=keep)
        for it in range(4):
            st.step(S["cams"][it % 6], S["posed"][it % 6], gts[it % 6])
        # (the draw is a multinomial over the accumulated statistics, which two runs only share to the order of float
        # atomics: give both the same weights, so that they draw the same rows)
        st.xyz_gradient_accum.copy_(torch.linspace(1.0, 2.0, pc.P, device=dev).reshape(-1, 1))
        st.uv_densify(1500, generator=g)
        for it in range(4):
            st.step(S["cams"][it % 6], S["posed"][it % 6], gts[it % 6])
        torch.cuda.synchronize()
        st.check()
        pcs.append(pc)
        sts.append(st)
    assert pcs[0].P == pcs[1].P == 21_500
    key = lambda pc: torch.stack([pc.face_index.double(), pc.bary_coords[:, 0].double(), pc.bary_coords[:, 1].double()], 1)  # noqa: E731
    ka, kb = key(pcs[0]), key(pcs[1])
    ia = torch.from_numpy(np.lexsort(ka.cpu().numpy().T[::-1])).to(dev)
    ib = torch.from_numpy(np.lexsort(kb.cpu().numpy().T[::-1])).to(dev)
    assert torch.equal(ka[ia], kb[ib])                                                        # the same bound set
    assert torch.equal(sts[1].coherent_order(), torch.arange(21_500, device=dev))
    assert not torch.equal(sts[0].coherent_order(), torch.arange(21_500, device=dev))
    _u.assert_same_trajectory(pcs[1]._opacity.detach().reshape(-1)[ib], pcs[0]._opacity.detach().reshape(-1)[ia], "densified",
                              tight=2e-2)
