"""GPU (-m gpu): the fp16 engine (HULC_DTYPE_F16, the reference's `precision: 16`, conf/trainer/play_trainer.yaml:3) and its on-device
dynamic loss scaler (torch.cuda.amp.GradScaler semantics) through the C-ABI.

* the step in fp16 against the reference fixtures (loss within 3e-3, gradient cosine > 0.995 after unscaling), including the S = 64
  fixture of BASELINE config 5's window length;
* the scaler state machine against a trajectory recorded from torch's GradScaler (tests/golden/grad_scaler.npz,
  tools/gen_golden_scaler.py): scale / growth tracker / skipped steps / Adam's own step count;
* overflow injection: a loss scale large enough to overflow the fp16 gradient intermediates -> the step is skipped, the scale backs off
  until the gradients are finite again;
* `precision=16` selects fp16 (not bf16)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import hulc_oracle as O  # noqa: E402
from golden_util import ROOT, load_case  # noqa: E402
from test_gpu_parity import _engine, grads_np, run_step  # noqa: E402


def _cos(Gg, G):
    a = np.concatenate([Gg[n].reshape(-1) for n in G]).astype(np.float64)
    b = np.concatenate([G[n].reshape(-1) for n in G]).astype(np.float64)
    return float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))), float(np.linalg.norm(a) / np.linalg.norm(b))


@pytest.mark.parametrize("name,scale", [("hulc_s32", 1024.0), ("hulc_s64", 4096.0), ("hulc_edge", 512.0), ("gcbc_s16", 1024.0)])
def test_fp16_step_close_to_reference(name, scale):
    dims, P, batch, fx = load_case(name)
    Bmax = max(mb["actions"].shape[0] for mb in batch.values())
    S = next(iter(batch.values()))["actions"].shape[1]
    eng = _engine(dims, Bmax, S, "fp16")
    st = eng.scaler_state()
    assert st["scale"] == 65536.0 and st["growth_tracker"] == 0          # GradScaler() defaults at context creation
    eng.scaler_enable(init_scale=scale)
    eng.load_numpy(P)
    losses_o, G = O.training_step(P, dims, batch)
    tot, per = run_step(eng, batch)
    ref = float(fx["loss_total"])
    assert abs(tot - ref) <= 3e-3 * abs(ref), (tot, ref)                   # losses are reported UNscaled
    Gg = {n: g / scale for n, g in grads_np(eng).items()}                  # the gradient buffer holds gradients x scale
    assert all(np.isfinite(g).all() for g in Gg.values())
    cos, ratio = _cos(Gg, G)
    assert cos > 0.995, cos
    assert abs(ratio - 1) < 0.02, ratio
    # the reference's fp64 gradient entries (sampled / full): same cosine gate on what the fixture holds
    a, b = [], []
    for key in fx.files:
        if key.startswith("grad64/") or key.startswith("gradsamp64/"):
            n = key.split("/", 1)[1]
            g = Gg[n].reshape(-1)
            if key.startswith("gradsamp64/"):
                from golden_util import sample_idx
                g = g[sample_idx(n, g.size)]
            a.append(g.astype(np.float64)); b.append(np.asarray(fx[key], np.float64).reshape(-1))
    a, b = np.concatenate(a), np.concatenate(b)
    assert float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b))) > 0.995
    # one optimizer step with finite gradients: taken, scale unchanged, tracker 1, parameters close to the fixture's torch.optim.Adam step
    p_before = eng.flat_params.clone()
    eng.adam_step()
    st = eng.scaler_state()
    assert st == dict(scale=scale, growth_tracker=1, skipped_steps=0, last_found_inf=0, taken_steps=1), st
    assert not torch.equal(p_before, eng.flat_params)
    pv = eng.views(eng.flat_params)
    n = "action_decoder.rnn.bias_hh_l1"
    if "adam1/" + n in fx.files:
        d_ref = fx["adam1/" + n].reshape(-1) - P[n].reshape(-1)
        d_got = pv[n].detach().cpu().numpy().reshape(-1) - P[n].reshape(-1)
        assert np.mean(np.sign(d_ref) == np.sign(d_got)) > 0.97           # Adam's first step is lr * sign(g): sign agreement
    eng.close()


def test_grad_scaler_state_machine_matches_torch():
    """Write gradients x scale (+ inf / nan on the recorded steps) into the bound gradient buffer and step: scale, growth tracker and the
    first 257 parameters must follow torch.amp.GradScaler + torch.optim.Adam exactly (skipped steps do not advance Adam's step count)."""
    fx = np.load(os.path.join(ROOT, "tests", "golden", "grad_scaler.npz"))
    dims, P, batch, _ = load_case("hulc_tiny")
    eng = _engine(dims, 2, 4, "fp32")             # the scaler is independent of the compute type; fp32 keeps the Adam comparison tight
    eng.load_numpy(P)
    names = sorted({k.split("/")[0] for k in fx.files})
    assert len(names) == 3
    for name in names:
        s0, gf, bf, gi, n, _seed = fx[f"{name}/cfg"]
        eng.scaler_enable(init_scale=float(s0), growth_factor=float(gf), backoff_factor=float(bf), growth_interval=int(gi))
        eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_t = 0
        eng.flat_params[:257] = torch.from_numpy(fx[f"{name}/p0"]).cuda()
        infs = set(int(i) for i in fx[f"{name}/inf_steps"])
        skipped = 0
        for t in range(int(n)):
            scale = eng.scaler_state()["scale"]
            eng.flat_grads.zero_()
            g = torch.from_numpy(fx[f"{name}/grads"][t]).cuda() * scale
            if t in infs:
                g[(7 * t) % 257] = float("inf") if t % 2 == 0 else float("nan")
                skipped += 1
            eng.flat_grads[:257] = g
            eng.adam_step()
            st = eng.scaler_state()
            assert st["scale"] == float(fx[f"{name}/scales"][t]), (name, t, st)
            assert st["growth_tracker"] == int(fx[f"{name}/trackers"][t]), (name, t, st)
            assert st["skipped_steps"] == skipped and st["last_found_inf"] == int(t in infs)
            got = eng.flat_params[:257].cpu().numpy()
            np.testing.assert_allclose(got, fx[f"{name}/params"][t], rtol=2e-6, atol=2e-7, err_msg=f"{name} step {t}")
    eng.scaler_enable(init_scale=0.0)             # off again: plain Adam
    assert eng.scaler_state()["scale"] == 1.0
    eng.close()


def test_fp16_overflow_skips_step_and_backs_off():
    dims, P, batch, fx = load_case("hulc_tiny")
    eng = _engine(dims, 2, 4, "fp16")
    eng.load_numpy(P)
    eng.scaler_enable(init_scale=2.0 ** 40)       # far beyond fp16 range for the decoder gradients: every step overflows at first
    p0 = eng.flat_params.clone()
    m0 = eng.adam_m.clone()
    nskip = 0
    for it in range(48):
        tot, _ = run_step(eng, batch)
        assert abs(tot - float(fx["loss_total"])) <= 3e-3 * abs(float(fx["loss_total"]))   # the forward never sees the scale
        eng.adam_step()
        st = eng.scaler_state()
        if st["last_found_inf"]:
            nskip += 1
            assert torch.equal(p0, eng.flat_params) and torch.equal(m0, eng.adam_m)        # skipped: nothing moved
            assert st["scale"] == 2.0 ** (40 - nskip) and st["growth_tracker"] == 0
        else:
            break
    assert 1 <= nskip < 48 and st["skipped_steps"] == nskip
    assert not torch.equal(p0, eng.flat_params)                                            # first finite step was taken
    assert np.isfinite(eng.flat_params.cpu().numpy()).all()
    eng.close()


def test_fp16_gemm_kernels():
    """hulc_k_gemm_nt with HULC_DTYPE_F16: the fp16 build of the LDS-DMA and register-staged GEMM kernels vs fp64."""
    from hulc_amd import lib as L
    lib = L.load()
    rng = np.random.default_rng(3)
    for (M, N, K, force) in [(70, 50, 72, 0), (1030, 260, 64, 0), (1024, 256, 512, 0), (1024, 256, 512, 4), (64, 2048, 2048, 0)]:
        A = rng.standard_normal((M, K)).astype(np.float32)
        Bm = (rng.standard_normal((N, K)) + np.arange(N)[:, None] * 0.01).astype(np.float32)
        bias = rng.standard_normal(N).astype(np.float32)
        a = torch.from_numpy(A).cuda().to(torch.float16).contiguous()
        b = torch.from_numpy(Bm).cuda().to(torch.float16).contiguous()
        ref = np.maximum(a.float().cpu().numpy().astype(np.float64) @ b.float().cpu().numpy().astype(np.float64).T + bias, 0)
        c = torch.zeros(M, N, device="cuda")
        bd = torch.from_numpy(bias).cuda()
        L.check(lib.hulc_k_gemm_nt(L.DTYPE["fp16"], a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, K, K, N, bd.data_ptr(), 1 | force, None))
        torch.cuda.synchronize()
        err = np.abs(c.cpu().numpy() - ref).max() / np.abs(ref).max()
        assert err < 2e-5, (M, N, K, force, err)          # operands already fp16-exact: only fp32 accumulation order differs


def test_module_precision_16_selects_fp16():
    from hulc_amd.hulc import Hulc
    m = Hulc(precision=16, max_batch_size=2, max_seq_len=4, use_clip_auxiliary_loss=False)
    assert m.precision == "fp16" and m.engine.dtype == "fp16"
    assert m.engine.scaler_state()["scale"] == 65536.0
    sd = m.configure_optimizers()["optimizer"].state_dict()
    assert sd["grad_scaler"] == dict(scale=65536.0, _growth_tracker=0, skipped_steps=0, calls=0) and sd["step"] == 0
    m.engine.close()
    m2 = Hulc(precision="bf16", max_batch_size=2, max_seq_len=4, use_clip_auxiliary_loss=False)
    assert m2.engine.dtype == "bf16" and "grad_scaler" not in m2.configure_optimizers()["optimizer"].state_dict()
    m2.engine.close()


def test_library_rccl_allreduce_world1_and_bucket_plan():
    """hulc_comm_* / hulc_backward_allreduce through the C-ABI with a 1-rank RCCL communicator (all this box offers): RCCL is resolved,
    the communicator initialises, every bucket is issued on the private stream in reverse-forward order, the event ordering lets the
    following Adam step see the reduced buffer, and a 1-rank SUM leaves the gradients bit-identical (fp32 buckets) / bf16-rounded
    (bf16 buckets).  The library's bucket ranges equal the host-side mirror hulc_amd.parallel.bucket_schedule."""
    from hulc_amd import parallel, spec
    dims, P, batch, fx = load_case("hulc_tiny")
    for dtype, bucket in (("fp32", "fp32"), ("bf16", "fp32"), ("bf16", "bf16")):
        eng = _engine(dims, 2, 4, dtype)
        eng.load_numpy(P)
        lay, numel = spec.layout(dims)
        assert eng.comm_buckets() == parallel.bucket_schedule(lay, numel)
        bk = eng.comm_buckets()
        assert bk[0][1] == numel and bk[-1][0] == 0                                               # decoder (.. end of buffer) first, encoders last
        t = sorted(bk)
        assert sum(hi - lo for lo, hi in bk) == numel and all(t[i][1] == t[i + 1][0] for i in range(len(t) - 1))      # a partition of the buffer
        tot_ref, _ = run_step(eng, batch)
        g_ref = eng.flat_grads.clone()
        eng.comm_init(eng.comm_unique_id(), 0, 1)
        assert eng.comm_size() == (0, 1)                                                          # ncclCommUserRank / ncclCommCount of the LIVE communicator (bench.py's N > 1 gate)
        with pytest.raises(RuntimeError):
            eng.comm_init(eng.comm_unique_id(), 0, 1)                                             # one communicator per context
        # same step, last backward with the overlapped bucketed all-reduce
        eng.zero_grads()
        scopes = list(batch)
        for i, sc in enumerate(scopes):
            from test_gpu_parity import to_dev
            eng.forward_loss(to_dev(batch[sc]), "lang" in sc, 1.0 / len(scopes), 3.0, step=0)
            if i == len(scopes) - 1:
                eng.backward_allreduce(bucket)
            else:
                eng.backward()
        torch.cuda.synchronize()
        st = eng.comm_stats()
        assert st["collectives"] == 5 and st["bytes"] == numel * (4 if bucket == "fp32" else 2)
        if bucket == "fp32":
            if dtype == "fp32":
                assert torch.equal(eng.flat_grads, g_ref)                                          # fp32 engine is deterministic: bit-identical
            else:
                # bf16 engine: atomics reorder sums, and now and then (2 of 30 evaluations, tools/det_probe.py) one reordered sum crosses a 16-bit
                # rounding boundary and moves everything downstream of it by ~1e-3 of the tensor's scale; a lost or doubled bucket is an O(0.3) error
                rel = ((eng.flat_grads - g_ref).double().norm() / g_ref.double().norm()).item()
                assert rel < 2e-2, rel
        else:
            rel = ((eng.flat_grads - g_ref).double().norm() / g_ref.double().norm()).item()
            assert 1e-5 < rel < 4e-3, rel                                                          # went through bf16 on the wire
        # whole-buffer form + Adam ordered after it (no host sync in between)
        p0 = eng.flat_params.clone()
        eng.allreduce_grads(bucket)
        eng.adam_step()
        torch.cuda.synchronize()
        assert eng.comm_stats()["collectives"] == 6 and not torch.equal(p0, eng.flat_params)
        if dtype == "fp32":
            with pytest.raises(RuntimeError):
                eng.allreduce_grads("bf16")                                                        # no 16-bit kernels in the fp32 unit
        eng.close()


def test_fp16_checkpoint_resume_continues_the_trajectory():
    """in fp16 mode Adam's bias corrections run on the device-side count of TAKEN steps; a checkpoint must carry it.  An
    uninterrupted run of 6 optimizer steps (one of them skipped by an injected overflow) equals: 3 steps -> state_dict -> a NEW module
    -> load_state_dict -> 3 more steps.  Without the restored count the resumed run would restart at t = 1 on warm moments and its
    first update would be ~3x too small (bias correction 1 - 0.9 = 0.1 against 1 - 0.9^3 = 0.27)."""
    from hulc_amd.hulc import Hulc
    from hulc_amd.utils import synthetic

    from test_gpu_parity import to_dev

    def make():
        m = Hulc(precision=16, max_batch_size=2, max_seq_len=4, use_clip_auxiliary_loss=False, seed=11)
        m.engine.set_dropout(0.0)
        m.engine.scaler_enable(init_scale=1024.0)        # well inside fp16's range for this model: the only skipped step is the injected one
        return m, m.configure_optimizers()["optimizer"]

    def steps(m, opt, lo, hi, overflow_at=-1):
        eng = m.engine
        for i in range(lo, hi):
            opt.zero_grad()
            for sc, mb in synthetic.make_batch(2, 0, 4, seed=100 + i).items():
                eng.forward_loss(to_dev(mb), "lang" in sc, 1.0, 3.0, step=i)        # injected plan draw: both runs see the same samples
                eng.backward()
            if i == overflow_at:
                eng.flat_grads[5] = float("inf")
            m._grads_reduced = True       # world = 1: nothing to reduce
            opt.step()

    m, opt = make()
    p0 = m.engine.flat_params.clone()
    steps(m, opt, 0, 6, overflow_at=1)
    ref = m.engine.flat_params.clone()
    st_ref = m.engine.scaler_state()
    assert st_ref["taken_steps"] == 5 and st_ref["skipped_steps"] == 1
    m.engine.close()

    m1, opt1 = make()
    assert torch.equal(m1.engine.flat_params, p0)
    steps(m1, opt1, 0, 3, overflow_at=1)
    sd_opt = opt1.state_dict()
    assert sd_opt["step"] == 2 and sd_opt["grad_scaler"]["calls"] == 3 and sd_opt["grad_scaler"]["skipped_steps"] == 1
    params = m1.engine.flat_params.clone()
    m1.engine.close()

    m2, opt2 = make()
    m2.engine.flat_params.copy_(params)
    m2.engine.prepare_weights()
    opt2.load_state_dict(sd_opt)
    st = m2.engine.scaler_state()
    assert st["taken_steps"] == 2 and st["scale"] == sd_opt["grad_scaler"]["scale"]
    steps(m2, opt2, 3, 6)
    got = m2.engine.flat_params
    rel = ((got - ref).double().norm() / (ref - p0).double().norm()).item()
    # same trajectory up to what reordered fp32 atomics grow into over three steps of this tiny (B = 2, S = 4) model: two IDENTICAL fresh runs
    # differ by 3e-4 in their step-0 gradients and by 0.5 - 1.5 % of the distance travelled after six steps (tools/resume_probe.py), and a
    # stream synchronisation in the resumed context's first step (the persistent recurrence's one-time device check) is enough to change
    # the order.  A restarted bias correction would be off by > 0.3.
    assert rel < 0.1, rel
    assert m2.engine.scaler_state()["taken_steps"] == 5
    m2.engine.close()


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_tightly_packed_layout_keeps_every_gradient_element_and_votes_outside_the_buffer(dtype):
    """hulc_bind_params accepts any 4-aligned table.  With a TIGHTLY packed one (no 64-element padding) the element behind a tensor is
    the first gradient element of the next tensor: the data-parallel skip vote must not ride there (it votes through a 4-byte all-reduce of its
    own: 6 collectives instead of 5), the lazily zeroed ranges must not round into a neighbour, and the bucket plan must still partition the
    buffer.  Gradients of the packed layout == gradients of the padded one, tensor by tensor; the step after it too (Adam + a second backward)."""
    from hulc_amd import parallel, spec
    from hulc_amd.engine import StepEngine
    dims, P, batch, fx = load_case("hulc_tiny")
    res = {}
    for pad in (64, 4):
        eng = StepEngine(dims, 2, 4, dtype=dtype, dropout_p=0.0, device="cuda:0", layout_pad=pad)
        eng.load_numpy(P)
        lay, numel = spec.layout(dims, pad)
        assert eng.numel == numel and eng.comm_buckets() == parallel.bucket_schedule(lay, numel)
        parallel.check_bucket_plan(eng.comm_buckets(), numel)
        eng.comm_init(eng.comm_unique_id(), 0, 1)
        if pad == 4:
            eng.set_option("debug_vote_word", 1)      # (the 1-element spatial_softmax.temperature leaves padding even at pad 4: force the no-padding vote path)
        out = []
        for it in range(2):
            eng.zero_grads()
            scopes = list(batch)
            for i, sc in enumerate(scopes):
                from test_gpu_parity import to_dev
                eng.forward_loss(to_dev(batch[sc]), "lang" in sc, 1.0 / len(scopes), 3.0, step=0)
                if i == len(scopes) - 1:
                    eng.backward_allreduce("fp32")
                else:
                    eng.backward()
            torch.cuda.synchronize()
            out.append({n: g.copy() for n, g in grads_np(eng).items()})
            eng.adam_step()
        torch.cuda.synchronize()
        res[pad] = (out, eng.comm_stats()["collectives"], {n: t.detach().cpu().numpy().copy() for n, t in eng.views(eng.flat_params).items()})
    assert res[64][1] == 10 and res[4][1] == 12, (res[64][1], res[4][1])                         # the packed layout's vote is one extra 4-byte collective per backward
    tol = 0.0 if dtype == "fp32" else 2e-2                              # fp32 engine: deterministic -> bit-identical; bf16: run-to-run atomics order
    for it in range(2):
        for n, g in res[64][0][it].items():
            h = res[4][0][it][n]
            assert np.isfinite(h).all() and (dtype != "fp32" or np.array_equal(g, h)), (it, n)
        a = np.concatenate([res[64][0][it][n].reshape(-1) for n in res[64][0][it]]).astype(np.float64)
        b = np.concatenate([res[4][0][it][n].reshape(-1) for n in res[64][0][it]]).astype(np.float64)
        assert np.linalg.norm(a - b) <= tol * np.linalg.norm(a), (it, np.linalg.norm(a - b) / np.linalg.norm(a))
    for n, p in res[64][2].items():
        assert np.allclose(p, res[4][2][n], rtol=0, atol=0 if dtype == "fp32" else 1e-3), n
