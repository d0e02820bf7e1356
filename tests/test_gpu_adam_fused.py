"""The optimizer step that writes the transposed 16-bit weight copies itself (csrc/kernels.h adam_tiled_kernel, VERDICT r4 #8 i): after three
training steps the parameters, both moments, and the gradients of a fourth backward — which reads every transposed copy — must be BIT-identical to
the flat Adam pass followed by the batched transpose (hulc_set_option "adam_fused_transposes" 0)."""
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


@pytest.mark.parametrize("kind,rnn_type,dtype,opt", [("hulc", "rnn", "bf16", "adam"), ("hulc", "rnn", "fp16", "adam"), ("hulc", "rnn", "bf16", "adamw"),
                                                     ("mcil", "gru", "bf16", "adam"), ("gcbc", "rnn", "bf16", "adam")])
def test_adam_that_writes_the_transposed_copies_equals_flat_pass_plus_transpose(kind, rnn_type, dtype, opt):
    """Three optimizer steps on INJECTED gradients (the backward's own atomics are not run-to-run deterministic, and Adam's first steps turn a 1e-10
    difference of a near-zero gradient into a sign flip of its update): parameters and both moments bit-identical; then one forward + backward, which
    reads every transposed copy: a stale or wrongly indexed W^T is an O(1) error there, run-to-run noise is ~1e-6."""
    B, S = 8, 8
    mcil = kind == "mcil"
    dims = spec.ModelDims(kind=kind, max_window=32, use_clip=False, rnn_type=rnn_type)
    dev = torch.device("cuda:0")
    mb = synth_batch(B, S, dev, 1, False)
    g = torch.Generator(device=dev); g.manual_seed(5)
    if mcil:
        mb["plan_eps"] = torch.randn(B, 256, device=dev, generator=g)
    else:
        mb["plan_idx"] = torch.randint(0, 32, (B, 32), device=dev, generator=g, dtype=torch.int32)
    out = {}
    for fuse in (0, 1):
        eng = StepEngine(dims, B, S, dtype=dtype, device="cuda:0", dropout_p=0.1, seed=1, num_classes=dims.mix_classes)
        eng.set_option("adam_fused_transposes", fuse)
        gs = 1.0
        if dtype == "fp16":
            gs = 256.0
            eng.scaler_enable(init_scale=gs)
        eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
        gg = torch.Generator(device=dev); gg.manual_seed(11)
        for step in range(3):
            eng.zero_grads()
            eng.flat_grads.copy_(torch.randn(eng.numel, device=dev, generator=gg) * (1e-3 * gs))       # hulc_grads is the caller's buffer: an external writer
            if opt == "adam":
                eng.adam_step(lr=1e-3)
            else:
                eng.optimizer_step(kind="adamw", lr=1e-3, weight_decay=0.01)
        eng.zero_grads()
        loss = eng.forward_loss(mb, False, 1.0, 3.0, step=3)["total_mod"]
        eng.backward()
        torch.cuda.synchronize()
        out[fuse] = dict(p=eng.flat_params.clone(), m=eng.adam_m.clone(), v=eng.adam_v.clone(), g=eng.flat_grads.clone() / gs, loss=loss, views=eng.views)
        lay = dict(eng.layout)
        eng.close()
    a, b = out[0], out[1]
    for k in ("p", "m", "v"):
        assert torch.isfinite(b[k]).all(), k
        assert torch.equal(a[k], b[k]), (k, int((a[k] != b[k]).sum()), float((a[k] - b[k]).abs().max()))
    assert float((a["p"] - torch.from_numpy(spec.init_all(dims, seed=0, ln_jitter=True)["plan_proposal.fc_model.2.weight" if kind != "gcbc" else "action_decoder.rnn.weight_hh_l0"].reshape(-1)).to(dev).mean()).abs().max()) > 0
    assert abs(a["loss"] - b["loss"]) <= 1e-5 * abs(a["loss"]), (a["loss"], b["loss"])
    bad = []
    for n, (off, shape) in lay.items():
        sz = int(np.prod(shape)) if len(shape) else 1
        x, y = a["g"][off:off + sz].double(), b["g"][off:off + sz].double()
        den = float(x.norm())
        if den > 1e-12 and float((x - y).norm()) / den > 2e-3:
            bad.append((n, float((x - y).norm()) / den))
    assert not bad, bad[:5]


def test_paired_layer1_weight_gradient_gemm_equals_two_launches():
    """hulc_set_option "gemm_pair": dW_hh1 and dW_ih1 of the action decoder as ONE launch that streams dZ1^T once (gemm.h gemm_glds_pair_kernel; B % 64 == 0)
    against the two gemm_glds launches: the same products in the same k order — the two weight gradients must agree to fp32 round-off of the accumulation
    (the pair kernel multiplies the same 64-token k-steps, so bit-identical is expected; 1e-6 relative is the gate), everything else bit-identical or noise."""
    B, S = 64, 6
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    dev = torch.device("cuda:0")
    mb = synth_batch(B, S, dev, 1, False)
    g = torch.Generator(device=dev); g.manual_seed(5)
    mb["plan_idx"] = torch.randint(0, 32, (B, 32), device=dev, generator=g, dtype=torch.int32)
    out = {}
    for pair in (0, 1):
        eng = StepEngine(dims, B, S, dtype="bf16", device="cuda:0", dropout_p=0.0, seed=1, num_classes=dims.mix_classes)
        eng.set_option("gemm_pair", pair)
        eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
        eng.zero_grads()
        eng.forward_loss(mb, False, 1.0, 3.0, step=0)
        eng.backward()
        torch.cuda.synchronize()
        v = eng.views(eng.flat_grads)
        out[pair] = {n: v[n].clone() for n in ("action_decoder.rnn.weight_hh_l1", "action_decoder.rnn.weight_ih_l1", "action_decoder.rnn.weight_hh_l0")}
        eng.close()
    for n in out[0]:
        a, b = out[0][n].double(), out[1][n].double()
        assert float(a.norm()) > 0
        rel = float((a - b).norm() / a.norm())
        assert rel < 1e-6, (n, rel)


def test_grouped_decoder_backward_gemms_equal_three_launches():
    """hulc_set_option "gemm_group" (default 0: measured equal, profiles/r06_ab_gemm_group.txt): dW_hh1, dW_ih1 and dH0 = dZ1 W_ih1 of the action decoder's backward as ONE grouped launch (gemm.h
    gemm_glds_group_kernel: 3 x 256 tiles in one grid) against the three gemm_glds launches.  Same tile kernel body, same k order per tile -> the two weight gradients
    are bit-identical; dH0 feeds layer 0's BPTT, so weight_hh_l0 / weight_ih_l0 agree bit for bit as well (S = 32: K = 2048 and 1984, and S = 6: K = 384 / 320)."""
    for B, S in ((64, 32), (64, 6)):
        dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
        dev = torch.device("cuda:0")
        mb = synth_batch(B, S, dev, 1, False)
        g = torch.Generator(device=dev); g.manual_seed(5)
        mb["plan_idx"] = torch.randint(0, 32, (B, 32), device=dev, generator=g, dtype=torch.int32)
        out = {}
        for grp in (0, 1):
            eng = StepEngine(dims, B, S, dtype="bf16", device="cuda:0", dropout_p=0.0, seed=1, num_classes=dims.mix_classes)
            eng.set_option("gemm_group", grp)
            eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
            eng.zero_grads()
            eng.forward_loss(mb, False, 1.0, 3.0, step=0)
            eng.backward()
            torch.cuda.synchronize()
            v = eng.views(eng.flat_grads)
            out[grp] = {n: v[n].clone() for n in ("action_decoder.rnn.weight_hh_l1", "action_decoder.rnn.weight_ih_l1", "action_decoder.rnn.weight_hh_l0", "action_decoder.rnn.weight_ih_l0")}
            eng.close()
        for n in out[0]:
            assert float(out[0][n].double().norm()) > 0
            rel = float((out[0][n] - out[1][n]).double().norm() / out[0][n].double().norm())
            if n.endswith("_l1"):
                assert torch.equal(out[0][n], out[1][n]), (B, S, n, rel)          # the grouped products themselves: plain stores of the same tile sums
            else:
                assert rel < 1e-6, (B, S, n, rel)                                  # downstream of dH0 (layer 0's BPTT and its weight gradients)


def test_compact_gemm_epilogues_equal_the_generic_one():
    """hulc_set_option "epilogue_fast": the compact epilogue paths of the GEMM kernels (plain fp32 store / accumulate; 16-bit output with bias, residual, ReLU, ReLU mask
    and the second store — gemm.h epi_plain4 / epi_fast16_*) against the generic epi_store4 path on a step whose decoder GEMMs take the 128 x 128 LDS-DMA kernel
    (B = 64, S = 16: 1024 token rows).  Same arithmetic in the same order: loss and every gradient tensor agree to the run-to-run noise of the backward's atomics."""
    B, S = 64, 16
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    dev = torch.device("cuda:0")
    mb = synth_batch(B, S, dev, 3, False)
    g = torch.Generator(device=dev); g.manual_seed(7)
    mb["plan_idx"] = torch.randint(0, 32, (B, 32), device=dev, generator=g, dtype=torch.int32)
    out = {}
    for fast in (0, 1, 2):                               # 2: the compact paths again = the run-to-run noise of the backward (fp32 atomics in front of 16-bit roundings)
        eng = StepEngine(dims, B, S, dtype="bf16", device="cuda:0", dropout_p=0.1, seed=1, num_classes=dims.mix_classes)
        eng.set_option("epilogue_fast", min(fast, 1))
        eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
        eng.zero_grads()
        loss = eng.forward_loss(mb, False, 1.0, 3.0, step=2)["total_mod"]
        eng.backward()
        torch.cuda.synchronize()
        out[fast] = dict(loss=loss, g=eng.flat_grads.clone())
        lay = dict(eng.layout)
        eng.close()
    assert abs(out[0]["loss"] - out[1]["loss"]) <= 1e-6 * abs(out[0]["loss"]), (out[0]["loss"], out[1]["loss"])
    bad, noise_max, diff_max = [], 0.0, 0.0
    for n, (off, shape) in lay.items():
        sz = int(np.prod(shape)) if len(shape) else 1
        x, y, z = out[0]["g"][off:off + sz].double(), out[1]["g"][off:off + sz].double(), out[2]["g"][off:off + sz].double()
        den = float(y.norm())
        if den <= 1e-12:
            continue
        d, nz = float((x - y).norm()) / den, float((z - y).norm()) / den
        noise_max, diff_max = max(noise_max, nz), max(diff_max, d)
        if d > 3.0 * nz + 2e-4:
            bad.append((n, d, nz))
    print("generic vs compact: worst tensor %.2e; compact vs compact again: %.2e" % (diff_max, noise_max))
    assert not bad, bad[:5]
