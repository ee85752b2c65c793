"""The small helpers the reference's Gaussian callers take from tools/gs_utils (sh_utils.py:114-118,
general_utils.py:18-19,29-62, graphics_utils.py:38-49), for callers that use this package instead: colour <-> SH DC
term, inverse sigmoid, the exponential learning-rate schedule (GaussianModel.training_setup of the 3DGS baselines) and
the world-to-view matrix with the scene translate / scale.  Each is pinned on the reference's own function
(tests/golden/golden_misc.npz, tests/test_oracle_golden.py)."""
from __future__ import annotations

import numpy as np

C0 = 0.28209479177387814


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5


def inverse_sigmoid(x):
    return np.log(x / (1 - x)) if isinstance(x, (np.ndarray, float)) else (x / (1 - x)).log()


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation from lr_init (step 0) to lr_final (step max_steps), optionally eased in over
    lr_delay_steps by a sine ramp starting at lr_delay_mult; 0 for negative steps or when both rates are 0."""

    def helper(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay_rate = 1.0
        if lr_delay_steps > 0:
            delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)

    return helper


def getWorld2View2(R, t, translate=np.array([.0, .0, .0]), scale=1.0):
    """World-to-view matrix [4,4] float32 from the camera-to-world rotation R and the view translation t, with the camera
    centre moved by `translate` and scaled by `scale`."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))
