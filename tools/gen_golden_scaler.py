"""tests/golden/grad_scaler.npz: the scale / growth-tracker trajectory of torch's GradScaler (the scaler Lightning's native-AMP plugin
drives at the reference's `precision: 16`, conf/trainer/play_trainer.yaml:3) for a recorded sequence of steps with and without
non-finite gradients, plus the parameter / Adam-state values of a tiny torch.optim.Adam problem stepped through it (skipped steps
leave parameters, moments AND Adam's step count untouched).  CPU GradScaler = the same state machine as the CUDA one
(torch/amp/grad_scaler.py: _amp_foreach_non_finite_check_and_unscale_, _amp_update_scale_).
Run in the build container only:  python tools/gen_golden_scaler.py"""
from __future__ import annotations

import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(init_scale, growth_factor, backoff_factor, growth_interval, inf_steps, nsteps, seed):
    torch.manual_seed(seed)
    p = torch.nn.Parameter(torch.randn(257))
    opt = torch.optim.Adam([p], lr=2e-4)
    sc = torch.amp.GradScaler("cpu", init_scale=init_scale, growth_factor=growth_factor, backoff_factor=backoff_factor, growth_interval=growth_interval)
    g = torch.Generator().manual_seed(seed + 1)
    grads = torch.randn(nsteps, 257, generator=g) * 1e-3
    scales, trackers, params = [], [], []
    for t in range(nsteps):
        opt.zero_grad()
        sc.scale(torch.zeros(1))          # GradScaler initialises its device state lazily on the first scale() call
        # what the backward of (loss * scale) leaves in .grad: the true gradient times the scale, inf where an fp16 intermediate overflowed
        gr = grads[t] * sc.get_scale()
        if t in inf_steps:
            gr = gr.clone()
            gr[(7 * t) % 257] = float("inf") if t % 2 == 0 else float("nan")
        p.grad = gr
        sc.step(opt)
        sc.update()
        scales.append(sc.get_scale())
        trackers.append(int(sc._growth_tracker.item()))
        params.append(p.detach().clone().numpy())
    return dict(scales=np.array(scales, np.float64), trackers=np.array(trackers, np.int32), params=np.stack(params), grads=grads.numpy())


if __name__ == "__main__":
    out = {}
    cases = {
        # name: (init_scale, growth, backoff, interval, inf step set, nsteps, seed)
        "default_overflow_start": (65536.0, 2.0, 0.5, 2000, {0, 1, 2, 9}, 16, 1),      # the usual first steps of an fp16 run: scale halves until finite
        "short_interval": (1024.0, 2.0, 0.5, 4, {5, 6, 13}, 24, 2),                     # growth after 4 good steps, trackers reset by a skip
        "odd_factors": (300.0, 3.0, 0.25, 3, {2, 3, 4, 11}, 20, 3),
    }
    for name, (s0, gf, bf, gi, infs, n, seed) in cases.items():
        torch.manual_seed(seed)
        p0 = torch.randn(257).numpy()
        r = run(s0, gf, bf, gi, infs, n, seed)
        out[f"{name}/cfg"] = np.array([s0, gf, bf, gi, n, seed], np.float64)
        out[f"{name}/inf_steps"] = np.array(sorted(infs), np.int32)
        out[f"{name}/p0"] = p0
        for k in ("scales", "trackers", "params", "grads"):
            out[f"{name}/{k}"] = r[k]
        print(name, "scales", r["scales"][:12], "trackers", r["trackers"][:12])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "grad_scaler.npz"), **out)
