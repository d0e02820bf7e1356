"""The optimizers and LR schedules the reference's conf tree ships besides Adam + constant (hulc/models/hulc.py:189-252,
conf/model/optimizer/{adam,adamw,sgd}.yaml, conf/model/lr_scheduler/{constant,linear,cosine}_schedule_with_warmup.yaml).

CPU: the host-side schedules against `transformers.get_*_schedule` on a real torch optimizer (the functions hydra instantiates in the
reference), compute_warmup / num_training_steps against the reference's arithmetic, the conf options through the mini-Hydra.
GPU (-m gpu): hulc_optimizer_step (one fused pass over the flat buffers) against torch.optim.Adam(weight_decay) / AdamW / SGD, 10 steps,
with the schedule's learning rates fed per call."""
import os
import types

import numpy as np
import pytest
import torch

from hulc_amd import hulc as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FakeModule:
    def parameters(self):
        return []


def _traj(sched, opt, n):
    out = [opt.param_groups[0]["lr"]]
    for _ in range(n):
        sched.step()
        out.append(opt.param_groups[0]["lr"])
    return np.array(out)


@pytest.mark.parametrize("kind,kw", [("constant", {}), ("linear", dict(num_warmup_steps=7, num_training_steps=40)), ("linear", dict(num_warmup_steps=0, num_training_steps=13)),
                                      ("cosine", dict(num_warmup_steps=5, num_training_steps=50, num_cycles=0.5)), ("cosine", dict(num_warmup_steps=3, num_training_steps=20, num_cycles=1.5))])
def test_schedules_match_transformers(kind, kw):
    import transformers
    lr = 2e-4
    ref_opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=lr)
    ref = {"constant": transformers.get_constant_schedule, "linear": transformers.get_linear_schedule_with_warmup,
           "cosine": transformers.get_cosine_schedule_with_warmup}[kind](ref_opt, **kw)
    want = [ref_opt.param_groups[0]["lr"]]
    for _ in range(60):
        ref_opt.step()
        ref.step()
        want.append(ref_opt.param_groups[0]["lr"])
    opt = H.FusedAdam(_FakeModule(), lr=lr)
    sched = {"constant": H.ConstantSchedule, "linear": H.LinearWarmupSchedule, "cosine": H.CosineWarmupSchedule}[kind](opt, **kw)
    got = _traj(sched, opt, 60)
    # exact up to num_training_steps; past the end the cosine factor here stays at its final value (transformers' lambda climbs the
    # cosine again once progress > 1, which a mis-inferred run length would turn into a second warm-up)
    n_end = kw.get("num_training_steps", 60) if kind == "cosine" else 60
    assert np.allclose(got[:n_end + 1], np.array(want)[:n_end + 1], rtol=1e-12, atol=0), (got[:10], want[:10])
    assert np.allclose(got[n_end:], got[n_end], rtol=0, atol=1e-18)
    assert got[0] == (0.0 if kw.get("num_warmup_steps", 0) > 0 else lr)        # LambdaLR applies f(0) at construction
    # state round trip (checkpoint resume keeps the position on the curve)
    opt2 = H.FusedAdam(_FakeModule(), lr=lr)
    s2 = type(sched)(opt2, **kw)
    s2.load_state_dict(sched.state_dict())
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"] and s2.last_epoch == 60


def test_compute_warmup_and_training_steps_follow_the_reference():
    m = H.Hulc.__new__(H.Hulc)                     # only the host arithmetic is under test: no engine
    tr = types.SimpleNamespace(datamodule=types.SimpleNamespace(steps_per_epoch=50), limit_train_batches=None, world=2, accumulate_grad_batches=1, max_epochs=4, max_steps=-1)
    object.__setattr__(m, "trainer", tr)
    # a datamodule that states its length PER RANK (steps_per_epoch: what every rank's loader yields) is not divided by the devices again
    assert m.num_training_steps == 50 * 4
    tr.max_steps = 30
    assert m.num_training_steps == 30                         # hulc.py:213-214
    tr.max_steps = -1
    tr.limit_train_batches = 10
    assert m.num_training_steps == 10 * 4                     # hulc.py:201-202
    tr.limit_train_batches = 0.5
    assert m.num_training_steps == 25 * 4                     # hulc.py:203-205
    assert m.compute_warmup(-1, 0.1) == (100, 10)             # hulc.py:229-236: fraction of the inferred steps, truncated
    assert m.compute_warmup(200, 25) == (200, 25)
    # an un-sharded loader (what the reference measures, hulc.py:197-199): (len // (accumulation x devices)) x max_epochs, hulc.py:209-211
    class _DM:
        def train_dataloader(self):
            return {"vis": list(range(50)), "lang": list(range(37))}
    tr.datamodule, tr.limit_train_batches = _DM(), None
    assert m.num_training_steps == (50 // 2) * 4
    tr.limit_train_batches = 0.5
    assert m.num_training_steps == (25 // 2) * 4


def test_trainer_takes_exactly_the_inferred_number_of_optimizer_steps():
    """Trainer.fit honours limit_train_batches (and rejects accumulate_grad_batches > 1), so the warm-up schedules built from
    Hulc.num_training_steps end where the run ends; the cosine factor does not rise again past the end."""
    from hulc_amd.trainer import Trainer
    with pytest.raises(NotImplementedError):
        Trainer(accumulate_grad_batches=2)
    dm = types.SimpleNamespace(steps_per_epoch=50)
    assert Trainer(limit_train_batches=7)._train_batches(dm) == 7
    assert Trainer(limit_train_batches=0.5)._train_batches(dm) == 25
    assert Trainer()._train_batches(dm) == 50
    # a datamodule WITHOUT steps_per_epoch hands out an un-sharded loader — the fit loop and num_training_steps share epoch_batches():
    # a float limit scales len(loader), the ranks split what is left (batch i -> rank i % world), an int limit is split the same way
    from hulc_amd.trainer import epoch_batches
    class _DM:
        def train_dataloader(self):
            return {"vis": list(range(50)), "lang": list(range(37))}
    assert epoch_batches(_DM(), None, 2) == (25, False) and epoch_batches(_DM(), 0.5, 2) == (12, False) and epoch_batches(_DM(), 9, 2) == (4, False)
    assert epoch_batches(dm, 0.5, 2) == (25, True)
    class _Gen:
        def train_dataloader(self):
            return (i for i in range(5))
    assert epoch_batches(_Gen(), None, 1) == (float("inf"), False) and epoch_batches(_Gen(), 3, 1) == (3, False)
    m = H.Hulc.__new__(H.Hulc)
    for world, ltb in ((1, None), (2, 0.5), (2, 9)):
        t = Trainer(limit_train_batches=ltb, max_epochs=3)
        t.world, t.datamodule = world, _DM()
        object.__setattr__(m, "trainer", t)
        assert m.num_training_steps == t._train_batches(_DM()) * 3
    opt = H.FusedAdam(_FakeModule(), lr=1.0)
    sched = H.CosineWarmupSchedule(opt, 2, 10)
    for _ in range(25):
        sched.step()
    assert opt.param_groups[0]["lr"] == 0.0                   # clamped at the end of the curve (transformers' lambda would be back at 1.0 by step 18)


@pytest.mark.parametrize("opt_name,sched_name", [("adamw", "cosine_schedule_with_warmup"), ("sgd", "linear_schedule_with_warmup"), ("adam", "constant")])
def test_conf_options_compose(opt_name, sched_name):
    from hulc_amd import config
    cfg = config.compose(os.path.join(ROOT, "conf"), "config", [f"model/optimizer={opt_name}", f"model/lr_scheduler={sched_name}"])
    oc, ls = cfg.model.optimizer, cfg.model.lr_scheduler
    assert oc["_target_"] == {"adamw": "torch.optim.AdamW", "sgd": "torch.optim.SGD", "adam": "torch.optim.Adam"}[opt_name]
    assert float(oc["lr"]) == 2e-4
    if opt_name == "adamw":
        assert float(oc["weight_decay"]) == 1e-6              # conf/model/optimizer/adamw.yaml:3
    if opt_name == "sgd":
        assert float(oc["momentum"]) == 0.9                   # conf/model/optimizer/sgd.yaml:3
    if "warmup" in sched_name:
        assert ls["num_training_steps"] == -1 and ls["num_warmup_steps"] == 0.1
    # configure_optimizers wiring without an engine: a stub module carrying only what the method reads
    m = H.Hulc.__new__(H.Hulc)
    object.__setattr__(m, "optimizer_config", dict(oc))
    object.__setattr__(m, "lr_scheduler", dict(ls))
    object.__setattr__(m, "trainer", types.SimpleNamespace(datamodule=types.SimpleNamespace(steps_per_epoch=50), limit_train_batches=None, world=1,
                                                           accumulate_grad_batches=1, max_epochs=2, max_steps=-1))
    m.parameters = lambda: []
    out = m.configure_optimizers()
    opt, sched = out["optimizer"], out["lr_scheduler"]["scheduler"]
    assert opt.kind == opt_name and out["lr_scheduler"]["interval"] == "step"
    if "warmup" in sched_name:
        assert opt.param_groups[0]["lr"] == 0.0               # step 0 of a warm-up
        for _ in range(10):
            sched.step()
        assert abs(opt.param_groups[0]["lr"] - 2e-4) < 1e-12  # 10 % of 100 steps


@pytest.mark.gpu
@pytest.mark.parametrize("kind,kw", [("adam", dict(weight_decay=0.0)), ("adam", dict(weight_decay=1e-2)), ("adamw", dict(weight_decay=1e-6)), ("adamw", dict(weight_decay=0.1)),
                                      ("sgd", dict(momentum=0.9)), ("sgd", dict(momentum=0.9, weight_decay=5e-4, nesterov=True)), ("sgd", dict(momentum=0.0))])
def test_fused_optimizers_match_torch(kind, kw):
    from hulc_amd import spec
    from hulc_amd.engine import StepEngine
    dims = spec.ModelDims(kind="gcbc", max_window=16, use_clip=False)
    eng = StepEngine(dims, 2, 4, dtype="fp32", dropout_p=0.0)
    eng.load_numpy(spec.init_all(dims, seed=3))
    g = torch.Generator(device="cuda").manual_seed(5)
    p_ref = torch.nn.Parameter(eng.flat_params.detach().clone().double())
    lr0 = 1e-3
    if kind == "adam":
        ref = torch.optim.Adam([p_ref], lr=lr0, **kw)
    elif kind == "adamw":
        ref = torch.optim.AdamW([p_ref], lr=lr0, **kw)
    else:
        ref = torch.optim.SGD([p_ref], lr=lr0, **kw)
    fake = types.SimpleNamespace(parameters=lambda: [], engine=eng, _grads_reduced=True)
    opt = H.FusedAdam(fake, lr=lr0, kind=kind, **kw)
    sched = H.CosineWarmupSchedule(opt, 3, 10)
    import transformers
    rsched = transformers.get_cosine_schedule_with_warmup(ref, 3, 10)
    for step in range(10):
        grad = torch.randn(eng.numel, device="cuda", generator=g) * (0.1 + step)
        eng.flat_grads.copy_(grad)
        p_ref.grad = grad.double()
        assert abs(opt.param_groups[0]["lr"] - ref.param_groups[0]["lr"]) < 1e-15
        opt.step()
        ref.step()
        sched.step()
        rsched.step()
    torch.cuda.synchronize()
    err = (eng.flat_params.double() - p_ref.detach()).abs().max().item()
    assert err < 2e-6, (kind, kw, err)                        # fp32 kernel against a float64 torch trajectory
    if kind == "sgd" and kw.get("momentum"):
        mb = ref.state[p_ref]["momentum_buffer"]
        assert (eng.adam_m.double() - mb).abs().max().item() < 1e-4 * mb.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("kind,kw", [("adamw", dict(weight_decay=0.05)), ("sgd", dict(momentum=0.9, weight_decay=1e-3))])
def test_fused_optimizers_under_the_fp16_loss_scaler(kind, kw):
    """hulc_optimizer_step in an fp16 context: gradients arrive multiplied by the loss scale, a step with a non-finite gradient is skipped
    (parameters and moments bit-unchanged, scale halves), and the steps that are taken match torch's optimizer on the unscaled gradients — the
    GradScaler contract, for the optimizers beyond Adam (SGD's `first step` = the first TAKEN step)."""
    from hulc_amd import spec
    from hulc_amd.engine import StepEngine
    dims = spec.ModelDims(kind="gcbc", max_window=16, use_clip=False)
    eng = StepEngine(dims, 2, 4, dtype="fp16", dropout_p=0.0)
    eng.load_numpy(spec.init_all(dims, seed=3))
    scale0 = 1024.0
    eng.scaler_enable(init_scale=scale0)
    g = torch.Generator(device="cuda").manual_seed(5)
    p_ref = torch.nn.Parameter(eng.flat_params.detach().clone().double())
    ref = (torch.optim.AdamW if kind == "adamw" else torch.optim.SGD)([p_ref], lr=1e-3, **kw)
    fake = types.SimpleNamespace(parameters=lambda: [], engine=eng, _grads_reduced=True)
    opt = H.FusedAdam(fake, lr=1e-3, kind=kind, **kw)
    scale = scale0
    for step in range(6):
        grad = torch.randn(eng.numel, device="cuda", generator=g) * 0.1
        overflow = step in (0, 3)                      # incl. the very first step: SGD's momentum buffer must start at the first TAKEN step
        eng.flat_grads.copy_(grad * scale)
        if overflow:
            eng.flat_grads[12345] = float("inf")
        before = (eng.flat_params.clone(), eng.adam_m.clone())
        opt.step()
        torch.cuda.synchronize()
        st = eng.scaler_state()
        if overflow:
            assert torch.equal(before[0], eng.flat_params) and torch.equal(before[1], eng.adam_m)
            scale *= 0.5
            assert st["scale"] == scale and st["last_found_inf"] == 1
        else:
            p_ref.grad = grad.double()
            ref.step()
            assert st["scale"] == scale and st["last_found_inf"] == 0
    err = (eng.flat_params.double() - p_ref.detach()).abs().max().item()
    assert err < 5e-6, (kind, err)
    assert eng.scaler_state()["skipped_steps"] == 2 and eng.scaler_state()["taken_steps"] == 4
