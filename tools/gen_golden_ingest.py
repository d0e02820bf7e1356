"""tests/golden/ingest_shift.npz: the reference's RandomShiftsAug (hulc/utils/transforms.py:8-29, unmodified) on random uint8-valued
frames, with the integer shifts it drew recorded (a torch.randint wrapper).  Run in the build container only."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
from hulc.utils.transforms import RandomShiftsAug  # noqa: E402

fx = {}
torch.manual_seed(7)
for name, h, pad, n in (("gripper", 84, 4, 3), ("small", 40, 10, 4)):
    x = torch.randint(0, 256, (n, 3, h, h), dtype=torch.uint8)
    draws = []
    orig = torch.randint

    def rec(*a, **k):
        t = orig(*a, **k)
        draws.append(t.clone())
        return t
    torch.randint = rec
    try:
        y = RandomShiftsAug(pad)(x)
    finally:
        torch.randint = orig
    assert len(draws) == 1
    fx[f"in_{name}"] = x.numpy()
    fx[f"shift_{name}"] = draws[0].reshape(n, 2).numpy().astype(np.int32)
    fx[f"out_{name}"] = y.numpy().astype(np.float32)
    fx[f"pad_{name}"] = np.int32(pad)
# RelativeActions (transforms.py:32-56, unmodified): absolute targets around a random robot state, including clipped entries and angle
# differences that wrap through +-pi
from hulc.utils.transforms import RelativeActions  # noqa: E402
rng = np.random.default_rng(11)
n = 64
ro = np.concatenate([rng.uniform(-0.5, 0.5, (n, 3)), rng.uniform(-np.pi, np.pi, (n, 3)), rng.standard_normal((n, 9))], 1).astype(np.float32)
act = np.concatenate([ro[:, :3] + rng.uniform(-0.04, 0.04, (n, 3)), ro[:, 3:6] + rng.uniform(-0.1, 0.1, (n, 3)), rng.choice([-1.0, 1.0], (n, 1))], 1).astype(np.float32)
act[:8, 3:6] = np.where(act[:8, 3:6] > 0, act[:8, 3:6] - 2 * np.pi, act[:8, 3:6] + 2 * np.pi)     # same orientation, other branch
fx["rel_robot_obs"], fx["rel_actions_abs"] = ro, act
fx["rel_max"] = np.array([0.02, 0.05], np.float32)
fx["rel_out"] = RelativeActions(0.02, 0.05)((act, ro)).astype(np.float32)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ingest_shift.npz"), **fx)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hulc_oracle as O  # noqa: E402
for name in ("gripper", "small"):
    o = O.random_shifts_aug(fx[f"in_{name}"].astype(np.float32), fx[f"shift_{name}"], int(fx[f"pad_{name}"]))
    print(name, "max |oracle - reference| on 0..255 values:", np.abs(o - fx[f"out_{name}"]).max())
print("relative actions max |oracle - reference|:", np.abs(O.relative_actions(fx["rel_actions_abs"], fx["rel_robot_obs"], 0.02, 0.05) - fx["rel_out"]).max())
