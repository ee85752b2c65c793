"""-m gpu: TWO real ranks through the REAL optimisation step (BASELINE.json configs[3] on the hardware a 1-GPU box has).

Two processes (torchrun, world_size 2, gloo, both on cuda:0) each run `AvatarStep` / `TrainStep` on their own frame of
every step for 33 steps across `_uv_densify`, prune and the opacity reset (tests/dp_two_rank_worker.py), and must end as
bit-identical replicas.

reference: the frames of a batch are rendered one after the other with shared Gaussians and the loss is their mean
(model/fateavatar.py:251-276, train/loss.py:92-105) — here one frame per rank and one flat-gradient all-reduce; the
densification statistics are per-view sums (model/fateavatar.py:734-737); `_uv_densify`'s multinomial / barycentric draws
(model/fateavatar.py:610-672) are made once, on rank 0, from the summed statistics.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P, RES, STEPS = 20000, 192, 33


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_two_ranks(mode, out_dir):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(FR_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_two_rank_worker.py"), "--mode", mode, "--out", out_dir,
           "--steps", str(STEPS), "--P", str(P), "--res", str(RES)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, "\n".join([r.stdout[-2000:], r.stderr[-8000:]])
    import torch
    return [torch.load(os.path.join(out_dir, f"rank{k}.pt"), weights_only=False) for k in range(2)]


def _rel_l2(a, b):
    import torch
    return float(torch.linalg.norm((a - b).double()) / torch.linalg.norm(b.double()).clamp_min(1e-30))


def _check_replicas(r0, r1, mode):
    import torch
    # (i) every parameter, Adam moment, Adam step state and the binding: bit-identical replicas after 33 steps
    keys = ["flat", "exp_avg", "exp_avg_sq", "adam_state", "acc_sum", "den_sum"] + (["face_index", "bary"] if mode == "avatar" else [])
    for k in keys:
        assert r0[k].shape == r1[k].shape, (k, r0[k].shape, r1[k].shape)
        assert torch.equal(r0[k], r1[k]), (k, float((r0[k].double() - r1[k].double()).abs().max()))
    # the statistics each rank holds are its OWN views' sums (they differ: the ranks saw different frames); what the
    # maintenance reads is their sum, identical everywhere and equal to the parts added up
    assert float(r0["den_local"].sum()) > 0 and not torch.equal(r0["acc_local"], r1["acc_local"])
    assert torch.equal(r0["den_local"] + r1["den_local"], r0["den_sum"])
    assert torch.allclose(r0["acc_local"] + r1["acc_local"], r0["acc_sum"], rtol=1e-6, atol=0)
    # (iii) the same maintenance happened on both ranks and the row count moved by exactly what it reports
    assert r0["did"] == r1["did"] and r0["P_end"] == r1["P_end"], (r0["did"], r1["did"])
    dens = sum(d.get("densified", 0) for _, d in r0["did"])
    pruned = sum(d.get("pruned", 0) for _, d in r0["did"])
    assert dens >= 2 * 400 and any("opacity_reset" in d for _, d in r0["did"]), r0["did"]
    assert pruned > 0, r0["did"]
    assert r0["P_end"] == r0["P0"] + dens - pruned
    assert r0["flat"].numel() % r0["P_end"] == 0
    assert r0["overflows"] == 0 and r1["overflows"] == 0
    assert int(r0["adam_state"][0]) == STEPS
    assert np.isfinite(r0["losses"]).all() and np.isfinite(r1["losses"]).all()
    # the two ranks optimise the same parameters towards their own frames: both losses fall — up to the opacity reset
    # of step 25 (losses[k] is step k + 2), which takes every opacity to 0.01 and the image with it
    for r in (r0, r1):
        assert np.mean(r["losses"][17:22]) < np.mean(r["losses"][:5]), r["losses"]


def _check_first_exchange(r0, r1, single_rank_grads):
    import torch
    # (ii) step 1: what the all-reduce left in the flat gradient buffer is the SUM of the two ranks' own gradients of that
    # very evaluation, to the bit (two addends: no order to differ in), identical on both ranks, and Adam scales it by 1 / 2
    assert torch.equal(r0["g_exchanged"], r1["g_exchanged"])
    assert torch.equal(r0["g_exchanged"], r0["g_parts_sum"])
    assert r0["grad_scale"] == 0.5 == r1["grad_scale"]
    assert torch.equal(r0["params0"], r1["params0"])
    # ... and the MEAN of the gradients two single-rank processes compute for the two frames at the same parameters, to
    # float-order noise (the blend backward's atomics add in a different order every launch)
    mean_single = 0.5 * (single_rank_grads[0] + single_rank_grads[1])
    e = _rel_l2(0.5 * r0["g_exchanged"], mean_single)
    assert e < 2e-5, e
    for k, r in enumerate((r0, r1)):       # each rank's own gradient is the single-rank gradient of ITS frame
        assert _rel_l2(r["g_local"], single_rank_grads[k]) < 2e-5
    assert _rel_l2(single_rank_grads[0], single_rank_grads[1]) > 1e-2   # (the two frames' gradients do differ)


def test_two_ranks_avatar_step_stay_bit_identical(gpu_device, tmp_path):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_two_rank_worker as W
    r0, r1 = _run_two_ranks("avatar", str(tmp_path))
    _check_replicas(r0, r1, "avatar")
    # the single-rank gradients of frames 0 and 1 at the initial parameters, in THIS process (no process group)
    from fateavatar_amd.avatar import AvatarStep
    assert not torch.distributed.is_initialized()
    S = W.avatar_setup(gpu_device, P, RES, 8)
    singles = []
    for f in (0, 1):
        pc = S["make"]()
        st = AvatarStep(pc, S["faces"], S["canon"], S["cams"][0].clone(), S["bg"], use_graph=False)
        assert not st.exchange
        assert torch.equal(pc.flat.detach().cpu(), r0["params0"])
        singles.append(W.local_gradient(st, lambda: (st.cam.copy_from(S["cams"][f]), st.verts.copy_(S["posed"][f]),
                                                     st.gt.copy_(S["gts"][f]))).cpu())
    _check_first_exchange(r0, r1, singles)


def test_two_ranks_train_step_stay_bit_identical(gpu_device, tmp_path):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_two_rank_worker as W
    r0, r1 = _run_two_ranks("train", str(tmp_path))
    _check_replicas(r0, r1, "train")
    from fateavatar_amd.train import TrainStep
    S = W.train_setup(gpu_device, P, RES, 8)
    singles = []
    for f in (0, 1):
        pc = S["make"]()
        st = TrainStep(pc, S["cams"][0].clone(), S["bg"], use_graph=False)
        assert not st.exchange
        singles.append(W.local_gradient(st, lambda: (st.cam.copy_from(S["cams"][f]), st.gt.copy_(S["gts"][f]))).cpu())
    _check_first_exchange(r0, r1, singles)
