"""GPU (-m gpu): the HIP training step (through the C-ABI, libhulc_hip.so) against the numpy oracle and the
reference-generated golden fixtures.  fp32 mode is the parity mode (north_star: forward/loss within 1e-3 fp32);
bf16 mode is the bench mode and is checked with bf16-sized tolerances."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import hulc_oracle as O  # noqa: E402
from golden_util import CASES, FP32_NOISY, adam_close, check_grads, check_grads64, grad_entries, load_case, rel_l2, sample_idx  # noqa: E402


def _engine(dims, B, S, dtype, dropout=0.0, **kw):
    from hulc_amd.engine import StepEngine
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible — the product path has no CPU fallback")
    return StepEngine(dims, B, S, dtype=dtype, dropout_p=dropout, **kw)


def to_dev(mb, inject_plan=True):
    out = {}
    for k, v in mb.items():
        if k == "use_for_aux":
            out["aux_rows"] = np.nonzero(v)[0].astype(np.int32)
        elif k == "plan_idx":
            if inject_plan:
                out[k] = torch.from_numpy(v.astype(np.int32)).cuda()
        else:
            out[k] = torch.from_numpy(v).cuda()
    return out


def run_step(eng, batch, clip_beta=3.0, step=0, inject_plan=True):
    eng.zero_grads()
    nmod = len(batch)
    tot, per = 0.0, {}
    for sc, mb in batch.items():
        l = eng.forward_loss(to_dev(mb, inject_plan), "lang" in sc, 1.0 / nmod, clip_beta, step=step)
        eng.backward()
        per[sc] = l
        tot += l["total_mod"] / nmod + (clip_beta * l["clip"] if eng.dims.use_clip else 0.0)
    return tot, per


def grads_np(eng):
    return {n: t.detach().cpu().numpy() for n, t in eng.views(eng.flat_grads).items()}


@pytest.mark.parametrize("name", list(CASES))
def test_fp32_step_matches_oracle_and_reference(name):
    dims, P, batch, fx = load_case(name)
    Bmax = max(mb["actions"].shape[0] for mb in batch.values())
    S = next(iter(batch.values()))["actions"].shape[1]
    eng = _engine(dims, Bmax, S, "fp32")
    eng.load_numpy(P)
    losses_o, G, caches = O.training_step(P, dims, batch, keep_cache=True)
    tot, per = run_step(eng, batch)
    ref = float(fx["loss_total"])
    assert abs(tot - ref) <= 1e-3 * abs(ref), (tot, ref)                 # north_star tolerance vs the REFERENCE
    assert abs(tot - float(losses_o["total"])) <= 2e-5 * abs(ref)        # and much tighter vs the oracle
    for sc in batch:
        assert abs(per[sc]["action"] - float(fx[f"log/train/action_loss_{sc}"])) <= 1e-3 * abs(ref)
        if dims.kind == "hulc":
            assert abs(per[sc]["kl"] - float(fx[f"log/train/kl_loss_scaled_{sc}"])) <= 1e-5
    # intermediates of the last modality still live in the workspace
    sc = list(batch)[-1]
    c = caches[sc]
    B = batch[sc]["actions"].shape[0]
    N = B * S
    assert rel_l2(eng.get_tensor("emb", N * 128).reshape(B, S, 128), fx[f"emb_{sc}"]) < 1e-4
    assert rel_l2(eng.get_tensor("seq_feat", B * 4096).reshape(B, 4096), fx[f"seq_feat_{sc}"]) < 1e-4
    assert rel_l2(eng.get_tensor("pr_logits", B * 1024).reshape(B, 1024), fx[f"pr_logits_{sc}"]) < 1e-4
    assert np.abs(eng.get_tensor("a_tcp", N * 7).reshape(B, S, 7) - fx[f"a_tcp_{sc}"]).max() < 3e-4
    a3 = eng.get_tensor("s_a3", N * 441 * 64).reshape(N, 21, 21, 64).transpose(0, 3, 1, 2)
    assert rel_l2(a3, c["enc_s"]["a3"]) < 1e-5
    if dims.kind == "hulc":
        assert np.array_equal(eng.plan_idx(B), batch[sc]["plan_idx"])
    # gradients: vs oracle (tight) and vs the reference fixture entries
    Gg = grads_np(eng)
    errs = {n: rel_l2(Gg[n], G[n]) for n in G if np.linalg.norm(G[n]) > 1e-6}
    noisy = lambda n: any(n.endswith(x) for x in FP32_NOISY)      # conv weight / bias sums: fp32 accumulation-order + ReLU-flip limited (golden_util)
    worst = max((e, n) for n, e in errs.items() if not noisy(n))
    worst_noisy = max((e, n) for n, e in errs.items() if noisy(n))
    assert worst[0] < 1e-3, worst
    assert worst_noisy[0] < 5e-3, worst_noisy
    assert np.median(list(errs.values())) < 2e-4
    check_grads(Gg, fx, tol_l2=1e-2, tol_norm=5e-3, label=name)          # the reference's own fp32 gradients (noisy themselves)
    w64, wn64 = check_grads64(Gg, fx, label=name)                         # its float64 gradients: 1e-3 / 5e-3 for the named conv tensors
    print(f"[{name}] HIP fp32 grads: vs oracle worst {worst[0]:.2e} ({worst[1]}) noisy {worst_noisy[0]:.2e}; vs fp64 reference worst {w64[0]:.2e} ({w64[1]}) noisy {wn64[0]:.2e} ({wn64[1]})")
    for key in fx.files:
        if key.startswith("gradnone/"):
            assert not np.any(Gg[key[len("gradnone/"):]])
    # Adam (fused flat kernel) vs the reference's torch.optim.Adam
    eng.adam_step()
    pv = eng.views(eng.flat_params)
    for key in fx.files:
        if key.startswith("adam1/"):
            n = key[len("adam1/"):]
            flat = pv[n].detach().cpu().numpy().reshape(-1)
            got = flat if flat.size <= 4096 else flat[sample_idx(n, flat.size)]
            assert adam_close(got, fx[key], grad_entries(fx, n)), n
    eng.close()


@pytest.mark.parametrize("name,dtype", [("hulc_tiny", "fp32"), ("hulc_s32", "fp32"), ("hulc_s64", "fp32"), ("hulc_s32", "bf16")])
def test_train_mode_step_matches_oracle_run_with_the_engines_dropout_masks(name, dtype):
    """the TIMED path runs with dropout 0.1 and was pinned only statistically.  The engine's masks are counter-based (csrc/common.h
    hash_uniform of a per-site seed, csrc/engine.h site_seed); the oracle restates that hash (hulc_oracle.engine_keep_mask) and runs the SAME train-mode step —
    the nine dropout sites of the plan-recognition transformer, forward and backward — so train mode is held to the oracle exactly like eval mode: fp32 engine
    loss 2e-5 and every gradient tensor 1e-3 (5e-3 for the conv sums); bf16 (fused transformer kernels) against the rounding-aware oracle, cosine > 0.995 and
    the transformer's own tensors within 5e-2.  A wrong site index, seed or element order shows up as an O(0.3) error in the transformer's gradients."""
    dims, P, batch, fx = load_case(name)
    Bmax = max(mb["actions"].shape[0] for mb in batch.values())
    S = next(iter(batch.values()))["actions"].shape[1]
    pdrop, seed, step = 0.1, 7, 3
    eng = _engine(dims, Bmax, S, dtype, dropout=pdrop, seed=seed)
    eng.load_numpy(P)
    tot, per = run_step(eng, batch, step=step)
    Gg = grads_np(eng)
    eng.close()
    try:
        O.TRAIN_DROPOUT = (pdrop, seed, step)
        if dtype != "fp32":
            O.set_operand_rounding(dtype)
        losses_t, G = O.training_step(P, dims, batch)
        O.TRAIN_DROPOUT = None
        losses_e, _ = O.training_step(P, dims, batch, want_grads=False)
    finally:
        O.TRAIN_DROPOUT = None
        O.set_operand_rounding(None)
    assert abs(float(losses_t["total"]) - float(losses_e["total"])) > 1e-4 * abs(float(losses_e["total"]))      # the masks do something
    errs = {n: rel_l2(Gg[n], G[n]) for n in G if np.linalg.norm(G[n]) > 1e-6}
    tr = {n: e for n, e in errs.items() if n.startswith("plan_recognition.")}
    if dtype == "fp32":
        assert abs(tot - float(losses_t["total"])) <= 2e-5 * abs(float(losses_t["total"])), (tot, losses_t["total"])
        noisy = lambda n: any(n.endswith(x) for x in FP32_NOISY)
        worst = max((e, n) for n, e in errs.items() if not noisy(n))
        worst_noisy = max((e, n) for n, e in errs.items() if noisy(n))
        assert worst[0] < 1e-3, worst
        assert worst_noisy[0] < 5e-3, worst_noisy
    else:
        assert abs(tot - float(losses_t["total"])) <= 2e-3 * abs(float(losses_t["total"])), (tot, losses_t["total"])
        a = np.concatenate([Gg[n].reshape(-1) for n in G]).astype(np.float64)
        b = np.concatenate([G[n].reshape(-1) for n in G]).astype(np.float64)
        assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.995
        assert max(tr.values()) < 5e-2, sorted(tr.items(), key=lambda kv: -kv[1])[:4]
    print(f"[train mode {name} {dtype}] loss {tot:.6f} vs oracle {float(losses_t['total']):.6f} (eval {float(losses_e['total']):.6f}); transformer tensors worst {max(tr.values()):.2e}")


def test_bf16_step_close_to_oracle():
    dims, P, batch, fx = load_case("hulc_s32")
    eng = _engine(dims, 3, 32, "bf16")
    eng.load_numpy(P)
    losses_o, G = O.training_step(P, dims, batch)
    tot, _ = run_step(eng, batch)
    assert abs(tot - float(losses_o["total"])) <= 3e-3 * abs(float(losses_o["total"])), (tot, losses_o["total"])
    Gg = grads_np(eng)
    a = np.concatenate([Gg[n].reshape(-1) for n in G]).astype(np.float64)
    b = np.concatenate([G[n].reshape(-1) for n in G]).astype(np.float64)
    cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
    assert cos > 0.995, cos
    assert abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) < 0.03
    eng.close()


def test_gemm_kernel_asymmetric():
    """C-ABI hulc_k_gemm_nt with asymmetric operands and ragged sizes (transposes / tails would show)."""
    import ctypes as C
    from hulc_amd import lib as L
    lib = L.load()
    rng = np.random.default_rng(0)
    for (M, N, K) in [(70, 50, 72), (300, 200, 136), (1030, 260, 64), (64, 2048, 2048), (5, 16, 128)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        Bm = (rng.standard_normal((N, K)) + np.arange(N)[:, None] * 0.01).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        ref = np.maximum(A.astype(np.float64) @ Bm.astype(np.float64).T + bias, 0)
        for dt, tol in (("fp32", 2e-6), ("bf16", 1.2e-2)):
            a = torch.from_numpy(A).cuda(); b = torch.from_numpy(Bm).cuda()
            if dt == "bf16":
                a = a.to(torch.bfloat16).contiguous(); b = b.to(torch.bfloat16).contiguous()
            c = torch.zeros(M, N, device="cuda")
            bd = torch.from_numpy(bias).cuda()
            L.check(lib.hulc_k_gemm_nt(L.DTYPE[dt], a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, K, K, N, bd.data_ptr(), 1, None))
            torch.cuda.synchronize()
            err = np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max()
            assert err < tol, (dt, M, N, K, err)


def test_device_sampling_and_dropout_run():
    """Train-mode path: on-device categorical sample (no injected plan) + dropout 0.1; loss finite, sample valid,
    same seed/step -> identical loss, different step -> different dropout masks."""
    dims, P, batch, fx = load_case("hulc_tiny")
    eng = _engine(dims, 2, 4, "fp32", dropout=0.1, seed=123)
    eng.load_numpy(P)
    t1, _ = run_step(eng, batch, step=5, inject_plan=False)
    idx = eng.plan_idx(2)
    assert idx.min() >= 0 and idx.max() < 32
    t2, _ = run_step(eng, batch, step=5, inject_plan=False)
    t3, _ = run_step(eng, batch, step=6, inject_plan=False)
    assert np.isfinite(t1) and t1 == t2 and t1 != t3
    g = np.concatenate([v.reshape(-1) for v in grads_np(eng).values()])
    assert np.isfinite(g).all()
    eng.close()


def test_dropout_gradient_consistency():
    """Dropout on: analytic gradient == central finite difference of the same-mask loss.  Uses GCBC + the CLIP loss,
    the only fully differentiable path through the transformer (HULC's KL balancing / straight-through sample use
    stop-gradients, so finite differences do not apply there)."""
    dims, P, batch, fx = load_case("gcbc_s16")
    eng = _engine(dims, 2, 16, "fp32", dropout=0.1, seed=9)
    eng.load_numpy(P)
    CB = 300.0
    base, _ = run_step(eng, batch, clip_beta=CB, step=3)
    G = grads_np(eng)
    worst = 0.0
    for name in ("plan_recognition.transformer_encoder.layers.0.linear2.bias",
                 "plan_recognition.transformer_encoder.layers.0.self_attn.in_proj_bias",
                 "plan_recognition.position_embeddings.weight"):
        g = G[name]
        d = np.sign(g).astype(np.float32) * 1e-3
        P2 = dict(P); P2[name] = P[name] + d
        eng.load_numpy(P2)
        up, _ = run_step(eng, batch, clip_beta=CB, step=3)
        P2[name] = P[name] - d
        eng.load_numpy(P2)
        dn, _ = run_step(eng, batch, clip_beta=CB, step=3)
        fd = 0.5 * (up - dn)
        an = float((g * d).sum())
        print("dropout FD check", name, "fd", fd, "analytic", an)
        assert abs(an) > 2e-4 and abs(fd - an) <= 0.1 * abs(an) + 2e-5, (name, fd, an)
    eng.close()


def test_error_paths():
    dims, P, batch, fx = load_case("hulc_tiny")
    eng = _engine(dims, 2, 4, "fp32")
    with pytest.raises(RuntimeError):
        eng.backward()                                   # before bind / forward
    eng.load_numpy(P)
    big = to_dev(batch["vis"])
    big = {k: (torch.cat([v, v, v]) if torch.is_tensor(v) else v) for k, v in big.items()}
    with pytest.raises(RuntimeError, match="exceeds workspace"):
        eng.forward_loss(big, False, 1.0, 3.0)
    with pytest.raises(RuntimeError):
        eng.backward()                                   # no forward kept
    eng.close()


def test_backward_in_two_parts_equals_whole():
    """hulc_backward_part(0) + (1) == hulc_backward (the split exists so the all-reduce can overlap the encoder backward);
    after part 0 every non-encoder gradient is already final."""
    dims, P, batch, fx = load_case("hulc_tiny")
    eng = _engine(dims, 2, 4, "fp32")
    eng.load_numpy(P)
    run_step(eng, batch)
    whole = eng.flat_grads.clone()
    eng.zero_grads()
    nmod = len(batch)
    n_enc = eng.encoder_numel
    for k, (sc, mb) in enumerate(batch.items()):
        eng.forward_loss(to_dev(mb), "lang" in sc, 1.0 / nmod, 3.0)
        eng.backward(0)
        if k == nmod - 1:
            assert torch.equal(eng.flat_grads[n_enc:], whole[n_enc:])
            with pytest.raises(RuntimeError):
                eng.backward()                      # encoder part still pending
        eng.backward(1)
    assert torch.equal(eng.flat_grads, whole)
    names_enc = [n for n, (off, _) in eng.layout.items() if off < n_enc]
    assert names_enc and all(n.startswith("perceptual_encoder.") for n in names_enc)
    eng.close()


@pytest.mark.parametrize("name", ["hulc_tiny", "gcbc_s16", "mcil_s6"])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_paired_pass_equals_two_modality_passes(name, dtype):
    """hulc_forward_loss_pair (vis + lang windows as ONE 2B-window pass) == the reference's order, one pass per modality: per-modality
    losses, every gradient tensor, and the reference fixture's total loss."""
    if name.startswith("mcil"):
        from golden_util import load_mcil_case
        dims, P, batch, fx = load_mcil_case(name)
    else:
        dims, P, batch, fx = load_case(name)
    if set(batch) != {"vis", "lang"} or batch["vis"]["actions"].shape[0] != batch["lang"]["actions"].shape[0]:
        pytest.skip("needs both modalities with the same number of windows")
    B, S = batch["vis"]["actions"].shape[:2]
    kw = dict(num_classes=dims.mix_classes)
    eng = _engine(dims, B, S, dtype, **kw)
    eng.load_numpy(P)
    tot, per = run_step(eng, batch)
    g_seq = eng.flat_grads.clone()
    eng.close()
    eng = _engine(dims, 2 * B, S, dtype, **kw)
    eng.load_numpy(P)
    eng.zero_grads()
    lv, ll = eng.forward_loss_pair(to_dev(batch["vis"]), to_dev(batch["lang"]), 0.5, 3.0, step=0)
    eng.backward()
    torch.cuda.synchronize()
    tol = 2e-6 if dtype == "fp32" else 2e-3
    for got, sc in ((lv, "vis"), (ll, "lang")):
        for k in ("total_mod", "kl", "action", "clip"):
            assert abs(got[k] - per[sc][k]) <= tol * max(1.0, abs(per[sc][k])), (sc, k, got[k], per[sc][k])
    tot_pair = (lv["total_mod"] + ll["total_mod"]) / 2 + (3.0 * ll["clip"] if dims.use_clip else 0.0)
    ref = float(fx["loss_total"])
    assert abs(tot_pair - ref) <= (1e-3 if dtype == "fp32" else 5e-3) * abs(ref)
    g_pair = eng.flat_grads
    if dtype == "fp32":
        views_a, views_b = eng.views(g_seq), eng.views(g_pair)
        worst = max(rel_l2(views_b[n].cpu().numpy(), views_a[n].cpu().numpy()) for n in views_a if float(views_a[n].abs().max()) > 1e-7)
        assert worst < 2e-5, worst
    else:
        cos = float((g_seq.double() @ g_pair.double()) / (g_seq.double().norm() * g_pair.double().norm()))
        assert cos > 0.999, cos
    # the pair is followed by ONE backward; a second one has nothing to run on
    with pytest.raises(RuntimeError):
        eng.backward()
    eng.close()
