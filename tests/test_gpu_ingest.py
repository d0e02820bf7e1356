"""GPU, through the C-ABI: the uint8 (B,S,H,W,C) ingest path (SURVEY.md §8(f) row 1 — ScaleImageTensor + Normalize + RandomShiftsAug
fused into conv1's load) gives the same training step as the reference boundary fed with the transformed fp32 NCHW frames."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hulc_oracle as O  # noqa: E402
from golden_util import load_case  # noqa: E402
from hulc_amd.engine import StepEngine  # noqa: E402


def _inputs(B, S, seed):
    rng = np.random.default_rng(seed)
    fs = rng.integers(0, 256, (B, S, 200, 200, 3), dtype=np.uint8)
    fg = rng.integers(0, 256, (B, S, 84, 84, 3), dtype=np.uint8)
    sh_s = rng.integers(0, 21, (B * S, 2)).astype(np.int32)
    sh_g = rng.integers(0, 9, (B * S, 2)).astype(np.int32)
    sh_s[0] = (0, 20); sh_g[0] = (8, 0)                      # extreme shifts: the replicate clamp on both sides
    return fs, fg, sh_s, sh_g


def _run(dims, P, B, S, dtype, mb, fold=None):
    eng = StepEngine(dims, B, S, dtype=dtype, device="cuda:0", dropout_p=0.0, seed=1)
    if fold is not None:
        eng.set_option("u8_fold", fold)
    eng.load_numpy(P)
    eng.zero_grads()
    l = eng.forward_loss(mb, False, 1.0, 3.0, step=0)
    eng.backward()
    torch.cuda.synchronize()
    g = eng.flat_grads.clone()
    views = {n: (off, int(np.prod(sh)) if len(sh) else 1) for n, (off, sh) in eng.layout.items()}
    eng.close()
    return l, g, views


@pytest.mark.parametrize("dtype,tol_loss,tol_grad", [("fp32", 1e-6, 2e-5), ("bf16", 2e-4, 3e-3)])
def test_u8_ingest_step_equals_fp32_boundary(dtype, tol_loss, tol_grad):
    """uint8 frames + shifts through the fused ingest path == the transformed fp32 NCHW frames through the reference boundary.  fp32 engine: exact
    conversion (ingest_u8_kernel).  bf16 engine: the path that stages x = u (2/255) - 1 itself (`u8_fold` 0: the same 16-bit rounded operand as the
    fp32 boundary, tight tolerance) AND the production path with the affine folded out of the data path (Conv1Src::fold: exact byte operands,
    scale + folded bias in conv1's epilogue, (2/255) S - db in its weight gradient) — which differs from the fp32 boundary by that boundary's own
    input rounding, so it is held (i) to the boundary within the 16-bit rounding level and (ii) to the fp32 ENGINE at least as closely as the
    fp32-boundary run of the bf16 engine is."""
    dims, P, batch, fx = load_case("hulc_tiny")
    mb0 = batch["vis"]
    B, S = mb0["actions"].shape[:2]
    fs, fg, sh_s, sh_g = _inputs(B, S, 5)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).cuda() if dt is None else torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
    common = dict(actions=t(mb0["actions"], np.float32), robot_obs=t(mb0["robot_obs"], np.float32), plan_idx=t(mb0["plan_idx"], np.int32))
    ref = dict(common, rgb_static=t(O.ingest_u8(fs, sh_s, 10)), rgb_gripper=t(O.ingest_u8(fg, sh_g, 4)))
    u8 = dict(common, rgb_static=t(fs), rgb_gripper=t(fg), shift_static=t(sh_s), shift_gripper=t(sh_g), pad_static=10, pad_gripper=4)
    l0, g0, views = _run(dims, P, B, S, dtype, ref)
    l1, g1, _ = _run(dims, P, B, S, dtype, u8, fold=0)
    assert abs(l0["total_mod"] - l1["total_mod"]) <= tol_loss * abs(l0["total_mod"]), (l0, l1)
    rel = ((g0 - g1).double().norm() / g0.double().norm()).item()
    assert rel <= tol_grad, rel
    if dtype == "fp32":
        return
    l2, g2, _ = _run(dims, P, B, S, dtype, u8, fold=1)             # the production path
    lt, gt, _ = _run(dims, P, B, S, "fp32", ref)                   # the parity engine on the reference boundary
    rel_l = lambda a, b: abs(a["total_mod"] - b["total_mod"]) / abs(b["total_mod"])
    rel_g = lambda a, b: ((a - b).double().norm() / b.double().norm()).item()
    print(f"[u8 fold] loss vs fp32 boundary {rel_l(l2, l0):.2e} (no fold {rel_l(l1, l0):.2e}); vs fp32 engine: fold {rel_l(l2, lt):.2e} boundary {rel_l(l0, lt):.2e}; "
          f"gradient vs fp32 engine: fold {rel_g(g2, gt):.3e} boundary {rel_g(g0, gt):.3e}, fold vs boundary {rel_g(g2, g0):.3e}")
    assert rel_l(l2, l0) <= 3e-3 and rel_g(g2, g0) <= 5e-2
    assert rel_l(l2, lt) <= max(2.0 * rel_l(l0, lt), 1e-3)
    assert rel_g(g2, gt) <= 1.15 * rel_g(g0, gt) + 1e-3
    # the first-layer tensors — the only ones the fold computes differently — one by one against the fp32 engine
    for n, (off, k) in views.items():
        if "conv_model.0." in n:
            e_fold = rel_g(g2[off:off + k], gt[off:off + k]); e_ref = rel_g(g0[off:off + k], gt[off:off + k])
            assert e_fold <= 1.5 * e_ref + 5e-3, (n, e_fold, e_ref)


def test_u8_ingest_validation_without_augmentation():
    """`val` transforms (no RandomShiftsAug): hulc_validate on uint8 frames == on their scaled / normalised fp32 version."""
    dims, P, batch, fx = load_case("hulc_tiny")
    mb0 = batch["vis"]
    B, S = mb0["actions"].shape[:2]
    fs, fg, _, _ = _inputs(B, S, 9)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).cuda() if dt is None else torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
    common = dict(actions=t(mb0["actions"], np.float32), robot_obs=t(mb0["robot_obs"], np.float32))
    eng = StepEngine(dims, B, S, dtype="bf16", device="cuda:0", seed=2)
    eng.load_numpy(P)
    a = eng.validate(dict(common, rgb_static=t(O.ingest_u8(fs)), rgb_gripper=t(O.ingest_u8(fg))), False, None)
    b = eng.validate(dict(common, rgb_static=t(fs), rgb_gripper=t(fg)), False, None)
    # a RandomShiftsAug pad beyond the kernels' staged replicate margin (16 pixels) is refused, in training and in validation
    sh = torch.zeros(B * S, 2, dtype=torch.int32, device="cuda")
    bad = dict(common, rgb_static=t(fs), rgb_gripper=t(fg), shift_static=sh, shift_gripper=sh, pad_static=20, pad_gripper=4)
    with pytest.raises(RuntimeError):
        eng.forward_loss(bad, False, 1.0, 3.0)
    with pytest.raises(RuntimeError):
        eng.validate(bad, False, None)
    eng.close()
    assert abs(a["action_loss_pp"] - b["action_loss_pp"]) <= 2e-4 * abs(a["action_loss_pp"])
    assert torch.equal(a["sampled_plan_idx_pp"], b["sampled_plan_idx_pp"])


def test_absolute_action_ingest_equals_relative_boundary():
    """hulc_batch.actions_absolute: RelativeActions (transforms.py:32-56) applied on the device == the step fed with the reference's
    transformed actions (the committed reference fixture), training step and validation."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "ingest_shift.npz"))
    dims, P, batch, _ = load_case("hulc_tiny")
    mb0 = batch["vis"]
    B, S = mb0["actions"].shape[:2]
    n = B * S
    t = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
    ro, act_abs, act_rel = (fx[k][:n].reshape(B, S, -1) for k in ("rel_robot_obs", "rel_actions_abs", "rel_out"))
    mp, mo = (float(v) for v in fx["rel_max"])
    common = dict(rgb_static=t(mb0["rgb_static"]), rgb_gripper=t(mb0["rgb_gripper"]), robot_obs=t(ro), plan_idx=t(mb0["plan_idx"], np.int32))
    eng = StepEngine(dims, B, S, dtype="fp32", device="cuda:0", dropout_p=0.0, seed=1)
    eng.load_numpy(P)
    res = []
    for mb in (dict(common, actions=t(act_rel)), dict(common, actions=t(act_abs), actions_absolute=True, max_rel_pos=mp, max_rel_orn=mo)):
        eng.zero_grads()
        l = eng.forward_loss(mb, False, 1.0, 3.0, step=0)
        eng.backward()
        torch.cuda.synchronize()
        v = eng.validate(dict(mb, step=3), False, None)
        res.append((l, eng.flat_grads.clone(), v))
    (l0, g0, v0), (l1, g1, v1) = res
    assert abs(l0["action"] - l1["action"]) <= 2e-6 * abs(l0["action"]), (l0, l1)
    assert ((g0 - g1).double().norm() / g0.double().norm()).item() <= 2e-5
    assert abs(v0["action_loss_pp"] - v1["action_loss_pp"]) <= 1e-5 * abs(v0["action_loss_pp"]) and np.allclose(v0["mae_pp"], v1["mae_pp"], atol=1e-5)
    with pytest.raises(RuntimeError):
        eng.forward_loss(dict(common, actions=t(act_abs), actions_absolute=True, max_rel_pos=0.0), False, 1.0, 3.0)
    eng.close()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_frame_store_windows_equal_the_materialised_batch(dtype):
    """hulc_batch::window_start: the step on windows GATHERED by index from a device-resident uint8 frame store == the step on the same
    windows passed as a materialised (B,S,H,W,C) batch — overlapping windows, the first and the last S frames of the store, RandomShiftsAug shifts per
    batch frame, an out-of-range start (clamped to the last window, never an out-of-bounds read).  fp32 engine: bit for bit (deterministic);
    bf16 engine: the loss bit for bit (the forward is deterministic), the gradients to the backward's run-to-run atomics noise.  Validation too."""
    dims, P, batch, fx = load_case("hulc_tiny")
    mb0 = batch["vis"]
    B, S = mb0["actions"].shape[:2]
    rng = np.random.default_rng(11)
    F = 3 * S + 5
    store_s = rng.integers(0, 256, (F, 200, 200, 3), dtype=np.uint8)
    store_g = rng.integers(0, 256, (F, 84, 84, 3), dtype=np.uint8)
    starts = np.array(([0, F - S, 2, 3] * B)[:B], np.int64)                # first / last window of the store, two overlapping ones
    sh_s = rng.integers(0, 21, (B * S, 2)).astype(np.int32)
    sh_g = rng.integers(0, 9, (B * S, 2)).astype(np.int32)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a)).cuda() if dt is None else torch.from_numpy(np.ascontiguousarray(a, dt)).cuda()
    common = dict(actions=t(mb0["actions"], np.float32), robot_obs=t(mb0["robot_obs"], np.float32), plan_idx=t(mb0["plan_idx"], np.int32),
                  shift_static=t(sh_s), shift_gripper=t(sh_g), pad_static=10, pad_gripper=4)
    idx = (starts[:, None] + np.arange(S)[None, :]).reshape(-1)
    mat = dict(common, rgb_static=t(store_s[idx].reshape(B, S, 200, 200, 3)), rgb_gripper=t(store_g[idx].reshape(B, S, 84, 84, 3)))
    sto = dict(common, rgb_static=t(store_s), rgb_gripper=t(store_g), window_start=t(starts))
    l0, g0, _ = _run(dims, P, B, S, dtype, mat)
    l1, g1, _ = _run(dims, P, B, S, dtype, sto)
    same = lambda a, b: all(a[k] == b[k] for k in a) if dtype == "fp32" else all(abs(a[k] - b[k]) <= 1e-5 * abs(a[k]) + 1e-7 for k in a)   # (bf16: fp32 atomics in the transformer)
    assert same(l0, l1), (l0, l1)
    if dtype == "fp32":
        assert torch.equal(g0, g1)
    else:
        l2, g2, _ = _run(dims, P, B, S, dtype, mat)                       # the materialised step again = the backward's own noise
        noise = ((g0 - g2).double().norm() / g0.double().norm()).item()
        rel = ((g0 - g1).double().norm() / g0.double().norm()).item()
        assert rel <= 3.0 * noise + 1e-5, (rel, noise)
    # a start beyond the store is clamped to the last window; validation reads the store the same way
    bad = starts.copy(); bad[1] = F + 100
    eng = StepEngine(dims, B, S, dtype=dtype, device="cuda:0", dropout_p=0.0, seed=1)
    eng.load_numpy(P)
    la = eng.forward_loss(dict(sto, window_start=t(bad)), False, 1.0, 3.0, step=0)
    assert same(la, l1), (la, l1)                                          # starts[1] was already F - S
    va = eng.validate({k: v for k, v in mat.items() if not k.startswith("shift") and k != "plan_idx"}, False, None)
    vb = eng.validate({k: v for k, v in sto.items() if not k.startswith("shift") and k != "plan_idx"}, False, None)
    assert abs(va["action_loss_pp"] - vb["action_loss_pp"]) <= 1e-5 * abs(va["action_loss_pp"]) and torch.equal(va["sampled_plan_idx_pp"], vb["sampled_plan_idx_pp"])
    with pytest.raises(RuntimeError):                                      # a store shorter than one window
        eng.forward_loss(dict(sto, rgb_static=t(store_s[:S - 1]), rgb_gripper=t(store_g[:S - 1])), False, 1.0, 3.0)
    with pytest.raises(ValueError):                                        # stores of different lengths
        eng.forward_loss(dict(sto, rgb_gripper=t(store_g[:F - 1])), False, 1.0, 3.0)
    eng.close()


@pytest.mark.parametrize("IH,pad", [(200, 10), (84, 4)])
def test_u8_conv1_forward_at_launch_scale_matches_a_direct_convolution_and_itself(IH, pad):
    """Round 6: conv1's uint8 forward on 2048 frames with random RandomShiftsAug shifts — the LDS-DMA kernel the step uses, twice, and the register-staged kernel
    (16-byte windows, dbg bit 8) — against torch's convolution of the shifted / clamped / normalised frames.  The small ingest fixtures (8 frames) never showed
    what this size does: a row's last DMA piece spills 8 bytes of the next row onto the row's right margin, and the margin fill raced with slower waves' pieces
    (positive column shifts: the three rightmost output columns of some rows wrong by O(0.1), differently per run).  Gates: both kernels within bf16 rounding of
    torch on sampled frames incl. the buffer's last, the two DMA runs and the register kernel bit-identical."""
    from hulc_amd import lib as L
    lib = L.load()
    Nf, OH = 2048, (IH - 8) // 4 + 1
    g = torch.Generator(device="cuda").manual_seed(IH)
    w = (torch.randn(32, 192, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(32, device="cuda", generator=g) * 0.1
    x = torch.randint(0, 256, (Nf, IH, IH, 3), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    sh = torch.randint(0, 2 * pad + 1, (Nf, 2), device="cuda", generator=g, dtype=torch.int32)
    sh[0] = torch.tensor([0, 2 * pad]); sh[1] = torch.tensor([2 * pad, 0]); sh[Nf - 1] = torch.tensor([2 * pad, 2 * pad])
    outs = [torch.zeros(Nf, OH, OH, 32, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
    for o, dbg in zip(outs, (0, 256, 0)):
        L.check(lib.hulc_k_conv_tile(6, x.data_ptr(), w.data_ptr(), b.data_ptr(), sh.data_ptr(), o.data_ptr(), Nf, IH, OH, dbg, None))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[2]), int((outs[0] != outs[2]).sum())           # the same launch twice
    assert torch.equal(outs[0], outs[1]), int((outs[0] != outs[1]).sum())           # same arithmetic on the same 16-bit operands
    idx = torch.arange(IH, device="cuda")
    W = w.float().reshape(32, 3, 8, 8)
    right = [int(f) for f in torch.nonzero(sh[:, 0] > pad).reshape(-1)[:6]]          # positive column shifts: the frames the race hit
    for f in [0, 1, Nf - 1] + right:
        dx, dy = int(sh[f, 0]) - pad, int(sh[f, 1]) - pad
        fr = (x[f].float()[(idx + dy).clamp(0, IH - 1)][:, (idx + dx).clamp(0, IH - 1)] * (2 / 255) - 1).permute(2, 0, 1)
        ref = torch.relu(torch.nn.functional.conv2d(fr.to(torch.bfloat16).float()[None], W, b, stride=4))[0].permute(1, 2, 0)
        err = float((outs[0][f].float() - ref).abs().max())
        assert err < 2e-2, (f, sh[f].tolist(), err)                                  # bf16 output rounding of values up to ~3


@pytest.mark.parametrize("IH,pad", [(200, 10), (84, 4)])
def test_u8_conv1_weight_gradient_at_launch_scale_matches_torch(IH, pad):
    """Round 6: conv1's weight / bias gradient from uint8 frames on 2048 frames with random RandomShiftsAug shifts — the kernel the step uses (conversion from the
    prefetch registers, conv1_wgrad_tr2r_kernel, with interior / row-end slots), the same kernel with one general slot kind, round 5's form (raw rows through LDS, tr2u) and both without the affine fold — against an fp64 torch reference
    dW[o][c][kh][kw] = sum dY[n][oh][ow][o] x[n][c][4 oh + kh][4 ow + kw] over the shifted / clamped / normalised frames.  With the fold the MFMA operand is the exact
    byte, so the only error is the fp32 accumulation (and atomics' order); without it x is rounded to bf16 first and the reference rounds the same way."""
    from hulc_amd import lib as L
    lib = L.load()
    Nf, OH = 2048, (IH - 8) // 4 + 1
    g = torch.Generator(device="cuda").manual_seed(IH + 1)
    x = torch.randint(0, 256, (Nf, IH, IH, 3), device="cuda", generator=g, dtype=torch.int32).to(torch.uint8)
    sh = torch.randint(0, 2 * pad + 1, (Nf, 2), device="cuda", generator=g, dtype=torch.int32)
    sh[0] = torch.tensor([0, 2 * pad]); sh[1] = torch.tensor([2 * pad, 0]); sh[Nf - 1] = torch.tensor([2 * pad, 2 * pad])
    dY = (torch.randn(Nf, OH, OH, 32, device="cuda", generator=g) * (torch.arange(32, device="cuda") % 5 + 1)).to(torch.bfloat16)
    idx = torch.arange(IH, device="cuda")
    refs = {1: torch.zeros(32, 3, 8, 8, dtype=torch.float64, device="cuda"), 0: torch.zeros(32, 3, 8, 8, dtype=torch.float64, device="cuda")}
    for f0 in range(0, Nf, 256):
        rows = (idx[None, :] + (sh[f0:f0 + 256, 1:2] - pad)).clamp(0, IH - 1).long()          # [n][y]
        cols = (idx[None, :] + (sh[f0:f0 + 256, 0:1] - pad)).clamp(0, IH - 1).long()          # [n][x]
        xs = x[f0:f0 + 256][torch.arange(256, device="cuda")[:, None, None], rows[:, :, None], cols[:, None, :]]      # (n, y, x, c)
        xe = (xs.double() * (2 / 255) - 1).permute(0, 3, 1, 2)
        xr = (xs.float() * (2 / 255) - 1).to(torch.bfloat16).double().permute(0, 3, 1, 2)
        d = dY[f0:f0 + 256].double()
        for fold, xx in ((1, xe), (0, xr)):
            for kh in range(8):
                for kw in range(8):
                    refs[fold][:, :, kh, kw] += torch.einsum("nhwd,nchw->dc", d, xx[:, :, kh:kh + 4 * OH:4, kw:kw + 4 * OH:4])
    refb = dY.double().sum(dim=(0, 1, 2))
    got = {}
    for form in (2, 1, 0):
        for fold in (1, 0):
            gw = torch.zeros(32, 192, device="cuda"); gb = torch.zeros(32, device="cuda")
            L.check(lib.hulc_k_conv1_wgrad_u8(x.data_ptr(), sh.data_ptr(), pad, dY.data_ptr(), gw.data_ptr(), gb.data_ptr(), Nf, IH, form, fold, None))
            torch.cuda.synchronize()
            ref = refs[fold].reshape(32, -1)
            err = float((gw.double() - ref).abs().max() / ref.abs().max())
            errb = float((gb.double() - refb).abs().max() / refb.abs().max())
            got[(form, fold)] = gw
            assert err < (2e-5 if fold else 1e-4), (form, fold, err)
            assert errb < 1e-5, (form, fold, errb)
    # the two forms feed the MFMAs the same operands in the same order; only the slab atomics' order differs
    assert float((got[(1, 1)] - got[(0, 1)]).abs().max() / got[(1, 1)].abs().max()) < 2e-6
    assert float((got[(2, 1)] - got[(1, 1)]).abs().max() / got[(1, 1)].abs().max()) < 2e-6
    assert float((got[(2, 0)] - got[(1, 0)]).abs().max() / got[(1, 0)].abs().max()) < 2e-6


@pytest.mark.parametrize("ingest", ["fp32", "u8"])
def test_encoder_forward_is_run_to_run_identical_at_benchmark_size(ingest):
    """The perceptual encoders' forward has no atomics: at the benchmark's size (B = 64, S = 32: 2048 frames per camera, bf16) two forwards of the same batch must give
    the same `emb` bit for bit, with the fp32 boundary and with uint8 frames + RandomShiftsAug shifts.  (A race in a staging path shows up here long before it moves a
    loss: round 6's margin race of the uint8 forward changed ~4 000 of 157 M conv1 outputs per launch.)"""
    sys.path.insert(0, ROOT)
    from bench import synth_batch
    from hulc_amd import spec
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    dev = torch.device("cuda:0")
    B, S = 64, 32
    mb = synth_batch(B, S, dev, 5, False, ingest)
    eng = StepEngine(dims, B, S, dtype="bf16", device="cuda:0", dropout_p=0.0, seed=1)
    eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
    embs = []
    for _ in range(3):
        eng.zero_grads()
        eng.forward_loss(mb, False, 1.0, 3.0, step=0, sync_losses=False)
        torch.cuda.synchronize()
        embs.append(eng.get_tensor("emb", B * S * 128).copy())
    eng.close()
    assert np.array_equal(embs[0], embs[1]) and np.array_equal(embs[0], embs[2]), (int((embs[0] != embs[1]).sum()), int((embs[0] != embs[2]).sum()))
