"""Full-size (BASELINE.json configs[1]: B=64 windows, seq_len 32) checks through the C-ABI, via size-independent properties — the
numpy oracle needs minutes at this size, so instead of a second implementation the step is checked against ITSELF:

  * window additivity (what data parallelism relies on): every loss term is a mean over windows, so
    grad(64 windows) == mean(grad(first 32), grad(last 32)) and the same for the losses — fp32 (parity) mode, dropout off,
    injected plan sample;
  * bf16 (bench) mode against fp32 mode on the same batch: loss within 3e-3 relative, global gradient cosine > 0.995;
  * one Adam step at lr 2e-4 on the same batch lowers the loss; two identical steps are bit-identical in fp32 mode.
"""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import synth_batch  # noqa: E402
from hulc_amd import spec  # noqa: E402
from hulc_amd.engine import StepEngine  # noqa: E402

B, S = 64, 32


def _batch(dev):
    mb = synth_batch(B, S, dev, seed=7)
    g = torch.Generator(device=dev); g.manual_seed(11)
    mb["plan_idx"] = torch.randint(0, 32, (B, 32), device=dev, generator=g, dtype=torch.int32)   # injected categorical sample
    return mb


def _slice(mb, lo, hi):
    return {k: (v[lo:hi].contiguous() if torch.is_tensor(v) else v) for k, v in mb.items()}


def _step(eng, mb, weight=1.0):
    eng.zero_grads()
    losses = eng.forward_loss(mb, False, weight, 3.0, step=0)
    eng.backward()
    torch.cuda.synchronize()
    return losses, eng.flat_grads.clone()


def _engine(dims, batch, dtype):
    eng = StepEngine(dims, batch, S, dtype=dtype, device="cuda:0", dropout_p=0.0, seed=3)
    eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
    return eng


def test_window_additivity_fp32_full_size():
    dev = torch.device("cuda:0")
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    mb = _batch(dev)
    eng = _engine(dims, B, "fp32")
    l_full, g_full = _step(eng, mb)
    l_again, g_again = _step(eng, mb)
    assert l_full == l_again and torch.equal(g_full, g_again)            # fp32 mode is deterministic (no atomics)
    l1, g1 = _step(eng, _slice(mb, 0, B // 2))
    l2, g2 = _step(eng, _slice(mb, B // 2, B))
    eng.close()
    for k in ("total_mod", "kl", "action"):
        assert abs(0.5 * (l1[k] + l2[k]) - l_full[k]) <= 2e-5 * abs(l_full[k]) + 1e-7, (k, l1[k], l2[k], l_full[k])
    g_half = 0.5 * (g1 + g2)
    rel = ((g_half - g_full).double().norm() / g_full.double().norm()).item()
    assert rel < 1e-4, rel
    assert np.isfinite(g_full.cpu().numpy()).all() and g_full.abs().max().item() > 0


def test_bf16_mode_tracks_fp32_mode_full_size():
    dev = torch.device("cuda:0")
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    mb = _batch(dev)
    e32 = _engine(dims, B, "fp32")
    l32, g32 = _step(e32, mb)
    e32.close()
    e16 = _engine(dims, B, "bf16")
    l16, g16 = _step(e16, mb)
    assert abs(l16["total_mod"] - l32["total_mod"]) <= 3e-3 * abs(l32["total_mod"]), (l16, l32)
    cos = (torch.dot(g16.double(), g32.double()) / (g16.double().norm() * g32.double().norm())).item()
    assert cos > 0.995, cos
    # one Adam step on the same batch lowers the loss
    e16.adam_step(lr=2e-4)
    l_after, _ = _step(e16, mb)
    e16.close()
    assert l_after["total_mod"] < l16["total_mod"], (l_after, l16)
