"""The step helpers around the rasterizer: fused L1 loss + gradient, one-launch input copies."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(3, 512, 512), (3, 37, 53), (1, 1, 3), (3, 1024, 1024)])
def test_l1_loss_and_grad_matches_autograd(gpu_device, shape):
    """reference: nn.L1Loss(reduction='mean') + backward (model/loss.py:92).  Loss to 1e-6 relative (different summation
    order), gradient exactly sign(img - gt) / n — including 0 where the two agree."""
    from fateavatar_amd.loss import l1_loss_and_grad
    g = torch.Generator(device="cpu").manual_seed(sum(shape))
    img = torch.rand(shape, generator=g).to(gpu_device)
    gt = torch.rand(shape, generator=g).to(gpu_device)
    gt.view(-1)[::7] = img.view(-1)[::7]          # exact ties: sign(0) = 0
    ref_in = img.clone().requires_grad_(True)
    ref = torch.nn.functional.l1_loss(ref_in, gt)
    ref.backward()
    for _ in range(3):                             # the workspace must come back zeroed: repeat
        loss, grad = l1_loss_and_grad(img, gt)
        torch.cuda.synchronize()
        assert abs(float(loss) - float(ref.detach())) <= 1e-6 * abs(float(ref.detach())) + 1e-9
        assert torch.equal(grad, ref_in.grad)
    # into caller-owned buffers
    lo, go = torch.zeros((), device=gpu_device), torch.zeros_like(img)
    l2, g2 = l1_loss_and_grad(img, gt, loss_out=lo, grad_out=go)
    assert l2.data_ptr() == lo.data_ptr() and g2.data_ptr() == go.data_ptr() and torch.equal(go, ref_in.grad)


@pytest.mark.parametrize("K,shape", [(1, (3, 64, 64)), (3, (3, 512, 512)), (4, (3, 37, 53))])
def test_l1_of_the_frames_of_a_batch_in_one_launch(gpu_device, K, shape):
    """fr_l1_loss_grad_batch == K calls of fr_l1_loss_grad: the same loss BITS (same partial sums, same order) and the same
    gradients, repeatedly (the workspaces come back zeroed), and distinct workspaces are enforced."""
    from fateavatar_amd.loss import l1_loss_and_grad, l1_loss_and_grad_batch, l1_workspace
    g = torch.Generator(device="cpu").manual_seed(K + sum(shape))
    imgs = [torch.rand(shape, generator=g).to(gpu_device) for _ in range(K)]
    gts = [torch.rand(shape, generator=g).to(gpu_device) for _ in range(K)]
    for a, b in zip(imgs, gts):
        b.view(-1)[::5] = a.view(-1)[::5]
    ref = [l1_loss_and_grad(a, b) for a, b in zip(imgs, gts)]
    losses = [torch.zeros((), device=gpu_device) for _ in range(K)]
    grads = [torch.full(shape, 7.0, device=gpu_device) for _ in range(K)]
    wss = [l1_workspace(gpu_device) for _ in range(K)]
    for _ in range(3):
        l1_loss_and_grad_batch(imgs, gts, losses, grads, wss)
        torch.cuda.synchronize()
        for k in range(K):
            assert torch.equal(losses[k], ref[k][0]) and torch.equal(grads[k], ref[k][1])
    if K > 1:
        with pytest.raises(RuntimeError, match="one workspace per image"):
            l1_loss_and_grad_batch(imgs, gts, losses, grads, [wss[0]] * K)


def test_l1_gradient_drives_the_rasterizer_backward(gpu_device):
    """render.backward(grad) with the fused gradient = l1_loss(render, gt).backward(): same parameter gradients."""
    from fateavatar_amd import scenes
    from fateavatar_amd.loss import l1_loss_and_grad
    from fateavatar_amd.model import FlatGaussians, TorchCamera
    from fateavatar_amd.render import render
    s = scenes.head_scene(P=5000, res=128, sh_degree=1, seed=2, opacity=0.5)
    cam = TorchCamera(s.camera, gpu_device)
    gt = torch.rand((3, 128, 128), device=gpu_device)
    bg = torch.ones(3, device=gpu_device)
    grads = []
    for fused in (False, True):
        pc = FlatGaussians(s.means3D, s.shs, s.opacities, s.scales, s.rotations, s.sh_degree, gpu_device)
        pc.begin_step()
        out = render(cam, pc, bg)
        if fused:
            _, g = l1_loss_and_grad(out["render"], gt)
            out["render"].backward(g)
        else:
            torch.nn.functional.l1_loss(out["render"], gt).backward()
        grads.append(pc.collect_grads().clone())
    assert float(grads[0].abs().max()) > 0
    # (the blend backward sums with float atomics: two runs agree to rounding, not bit for bit)
    assert torch.allclose(grads[0], grads[1], rtol=1e-4, atol=1e-9)


def test_multi_copy(gpu_device):
    from fateavatar_amd.loss import multi_copy
    rng = np.random.default_rng(0)
    sizes = [35, 3 * 64 * 64, 5023 * 3, 1]
    src = [torch.from_numpy(rng.standard_normal(n).astype(np.float32)).to(gpu_device) for n in sizes]
    dst = [torch.zeros(n, device=gpu_device) for n in sizes]
    multi_copy(list(zip(dst, src)))
    for d, s in zip(dst, src):
        assert torch.equal(d, s)
    # unaligned views fall back to scalar copies inside the kernel
    big = torch.zeros(1000, device=gpu_device)
    multi_copy([(big[1:36], src[0])])
    assert torch.equal(big[1:36], src[0]) and float(big[0]) == 0 and float(big[36]) == 0
    with pytest.raises(RuntimeError):
        multi_copy([(dst[0], src[1])])


def test_scaled_sum(gpu_device):
    from fateavatar_amd.loss import scaled_sum
    g = torch.Generator().manual_seed(3)
    for n in (1 << 20, 1001):
        srcs = [torch.randn(n, generator=g).to(gpu_device) for _ in range(3)]
        dst = torch.empty(n, device=gpu_device)
        for k in (1, 2, 3):
            scaled_sum(dst, srcs[:k], 1.0 / k)
            want = srcs[0].clone()
            for t in srcs[1:k]:
                want = want + t
            assert torch.equal(dst, want * (1.0 / k))
        a = srcs[0].clone()
        scaled_sum(a, [a, srcs[1]], 0.5)            # in place
        assert torch.equal(a, (srcs[0] + srcs[1]) * 0.5)
    with pytest.raises(RuntimeError):
        scaled_sum(dst, [srcs[0][:10]], 1.0)


def test_concurrent_streams_overlap(gpu_device):
    """fateavatar_amd.streams: the runtime maps streams onto a few hardware queues; streams on one queue are serialised.
    concurrent_streams() returns streams that overlap pairwise and with the caller's stream."""
    from fateavatar_amd.streams import _overlap, concurrent_streams
    cur = torch.cuda.current_stream(gpu_device)
    ss = concurrent_streams(3, gpu_device, also_with=[cur])
    assert len(ss) == 3 and len({s.cuda_stream for s in ss}) == 3
    for i, a in enumerate(ss):
        assert _overlap(cur, a, gpu_device, 400_000)
        for b in ss[i + 1:]:
            assert _overlap(a, b, gpu_device, 400_000) and _overlap(b, a, gpu_device, 400_000)


@pytest.mark.gpu
def test_l1_losses_of_overlapping_launches_do_not_mix(gpu_device):
    """The loss kernel elects its last workgroup and sums per-workgroup partials through its workspace: launches that
    overlap (the lanes of AvatarBatchStep, one stream each) must each have their own (loss.py keeps one per stream)."""
    import torch
    from fateavatar_amd.loss import l1_loss_and_grad, l1_workspace
    torch.manual_seed(0)
    K, n_rep = 3, 40
    imgs = [torch.rand(3, 512, 512, device=gpu_device) for _ in range(K)]
    gts = [torch.rand(3, 512, 512, device=gpu_device) * (k + 1) for k in range(K)]
    want = [float((a - b).abs().mean()) for a, b in zip(imgs, gts)]
    streams = [torch.cuda.Stream(device=gpu_device) for _ in range(K)]
    losses = [[torch.zeros((), device=gpu_device) for _ in range(n_rep)] for _ in range(K)]
    explicit = [l1_workspace(gpu_device) for _ in range(K)]
    for mode in ("per-stream default", "explicit"):
        for s in streams:
            s.wait_stream(torch.cuda.current_stream(gpu_device))
        for r in range(n_rep):
            for k in range(K):
                with torch.cuda.stream(streams[k]):
                    l1_loss_and_grad(imgs[k], gts[k], loss_out=losses[k][r], workspace=explicit[k] if mode == "explicit" else None)
        torch.cuda.synchronize()
        for k in range(K):
            got = torch.stack(losses[k]).cpu().numpy()
            assert np.allclose(got, want[k], rtol=1e-5), (mode, k, got.min(), got.max(), want[k])
