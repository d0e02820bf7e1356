"""GPU, through the C-ABI: validation forward (hulc_validate) and the stateful rollout (hulc_rollout_plan / hulc_rollout_act)
against fixtures produced by the unmodified reference (tools/gen_golden_val.py) with the reference's own draws injected."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_util import VAL_CASES, load_rollout_case, load_val_case  # noqa: E402
from hulc_amd.engine import StepEngine  # noqa: E402


def _dev(mb):
    out = {}
    for k, v in mb.items():
        if k in ("rgb_static", "rgb_gripper", "actions", "robot_obs", "lang"):
            out[k] = torch.from_numpy(np.ascontiguousarray(v, np.float32)).cuda()
    return out


@pytest.mark.parametrize("name", list(VAL_CASES))
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_validate_matches_reference(name, dtype, tol):
    dims, P, batch, noise, fx = load_val_case(name)
    for sc, mb in batch.items():
        B, S = mb["actions"].shape[:2]
        eng = StepEngine(dims, B, S, dtype=dtype, device="cuda:0", dropout_p=0.1, seed=5, num_classes=dims.mix_classes)
        eng.load_numpy(P)
        d = _dev(mb)
        if "use_for_aux" in mb:
            d["aux_rows"] = np.nonzero(mb["use_for_aux"])[0].astype(np.int32)
        o = eng.validate(d, "lang" in sc, noise[sc], want_pred=True)
        eng.close()
        if dims.kind == "gcbc":
            ref = float(fx[f"action_loss_pp_{sc}"])
            assert abs(o["action_loss_pp"] - ref) <= tol * abs(ref), (sc, o["action_loss_pp"], ref)
            if dtype == "fp32":
                assert np.abs(o["mae_pp"] - fx[f"mae_pp_{sc}"].mean(0)).max() <= 2e-3 and abs(o["gripper_sr_pp"] - float(fx[f"gripper_sr_pp_{sc}"])) <= 1e-6
            continue
        for k in ("action_loss_pp", "action_loss_pr", "kl_loss"):
            ref = float(fx[f"{k}_{sc}"])
            assert abs(o[k] - ref) <= tol * abs(ref) + 1e-6, (sc, k, o[k], ref)
        if f"val_pred_clip_loss_{sc}" in fx.files:          # hulc.py:804-808, lang modality with the CLIP auxiliary loss
            ref = float(fx[f"val_pred_clip_loss_{sc}"])
            assert abs(o["val_pred_clip_loss"] - ref) <= tol * abs(ref), (sc, o["val_pred_clip_loss"], ref)
        if dims.kind == "mcil":         # continuous plans: the injected draws come back as the sampled plans
            assert np.array_equal(o["sampled_plan_pp"].cpu().numpy(), noise[sc]["plan_pp"]) and np.array_equal(o["sampled_plan_pr"].cpu().numpy(), noise[sc]["plan_pr"])
        else:
            assert np.array_equal(o["sampled_plan_idx_pp"].cpu().numpy(), noise[sc]["plan_idx_pp"])
            assert np.array_equal(o["sampled_plan_idx_pr"].cpu().numpy(), noise[sc]["plan_idx_pr"])
        if dtype == "fp32":             # the sampled action is a discontinuous function of the logits (argmax): exact draws only in fp32
            for k in ("mae_pp", "mae_pr"):
                assert np.abs(o[k] - fx[f"{k}_{sc}"].mean(0)).max() <= 2e-3, (sc, k, o[k], fx[f"{k}_{sc}"].mean(0))
            for k in ("gripper_sr_pp", "gripper_sr_pr"):
                assert abs(o[k] - float(fx[f"{k}_{sc}"])) <= 1e-6, (sc, k)
        assert np.isfinite(o["pred_pp"].cpu().numpy()).all()


def test_validate_device_draws_are_valid_and_reproducible():
    dims, P, batch, noise, fx = load_val_case("val_hulc_tiny")
    mb = batch["vis"]
    B, S = mb["actions"].shape[:2]
    eng = StepEngine(dims, B, S, dtype="fp32", device="cuda:0", seed=9)
    eng.load_numpy(P)
    d = _dev(mb)
    a = eng.validate(d, False, None, want_pred=True)
    b = eng.validate(d, False, None, want_pred=True)
    eng.close()
    idx = a["sampled_plan_idx_pp"].cpu().numpy()
    assert idx.min() >= 0 and idx.max() < 32
    assert torch.equal(a["pred_pp"], b["pred_pp"]) and a["action_loss_pp"] == b["action_loss_pp"]      # counter RNG: same step -> same draws
    assert 0.0 <= a["gripper_sr_pp"] <= 1.0 and np.isfinite(a["mae_pr"]).all()


@pytest.mark.parametrize("case", ["rollout_hulc", "rollout_mcil"])
def test_rollout_matches_reference_step(case):
    dims, P, frames, nsteps, replan_freq, fx = load_rollout_case(case)
    pkey = "plan" if dims.kind == "mcil" else "plan_idx"
    eng = StepEngine(dims, 1, 2, dtype="fp32", device="cuda:0", seed=3, num_classes=dims.mix_classes)
    eng.load_numpy(P)
    for mode in ("vis", "lang"):
        mb = frames[mode]
        t_ = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()
        goal = dict(rgb_static=t_(mb["rgb_static"][:, nsteps:nsteps + 1]), rgb_gripper=t_(mb["rgb_gripper"][:, nsteps:nsteps + 1])) if mode == "vis" \
            else t_(frames["lang"]["lang"][0])
        eng.rollout_reset()
        for t in range(nsteps):
            obs = dict(rgb_static=t_(mb["rgb_static"][:, t:t + 1]), rgb_gripper=t_(mb["rgb_gripper"][:, t:t + 1]), robot_obs_raw=t_(mb["robot_obs"][0, t]))
            if t % replan_freq == 0:
                plan = eng.rollout_plan(obs, goal, plan_idx=fx[f"{pkey}_{mode}"][t][0])
                assert np.array_equal(plan.cpu().numpy(), fx[f"{pkey}_{mode}"][t][0])
            a = eng.rollout_act(obs, u_mix=fx[f"u_mix_{mode}"][t][0, 0], u_act=fx[f"u_act_{mode}"][t][0, 0])
            ref = fx[f"actions_{mode}"][0, t]
            assert np.abs(a - ref).max() <= 2e-3, (mode, t, a, ref)
    # error paths
    eng.rollout_reset()
    with pytest.raises(RuntimeError):
        eng.rollout_act(obs)
    eng.close()


def test_mcil_validate_device_draws_and_module_validation_step():
    """mcil without injected draws: on-device Normal samples are finite / reproducible; the module's validation_step logs the reference's
    metric names and returns the continuous plans (B,256)."""
    from hulc_amd import config
    dims, P, batch, noise, fx = load_val_case("val_mcil_s8")
    mb = batch["vis"]
    B, S = mb["actions"].shape[:2]
    eng = StepEngine(dims, B, S, dtype="fp32", device="cuda:0", seed=9, num_classes=dims.mix_classes)
    eng.load_numpy(P)
    d = _dev(mb)
    a = eng.validate(d, False, None, want_pred=True)
    b = eng.validate(d, False, None, want_pred=True)
    eng.close()
    pl = a["sampled_plan_pp"].cpu().numpy()
    assert pl.shape == (B, 256) and np.isfinite(pl).all() and pl.std() > 0 and not np.array_equal(pl, a["sampled_plan_pr"].cpu().numpy())
    assert torch.equal(a["pred_pp"], b["pred_pp"]) and a["kl_loss"] == b["kl_loss"] and a["kl_loss"] > 0
    cfg = config.compose(os.path.join(ROOT, "conf"), "config", ["model=mcil", "trainer.precision=fp32", "datamodule.batch_size=4"])
    model = config.instantiate(cfg.model, device="cuda:0", max_seq_len=32)
    model.load_state_dict({n: torch.from_numpy(P[n]) for n in P}, strict=False)
    model.eval()
    from test_gpu_module import ref_style_batch
    rb = ref_style_batch({sc: dict(m, plan_idx=np.zeros((1, 1), np.int32)) for sc, m in batch.items()})
    for sc in rb:
        rb[sc].pop("plan_idx")
    out = model.validation_step(rb, 0, noise)
    for sc in batch:
        assert np.array_equal(out[f"sampled_plan_pp_{sc}"].cpu().numpy(), noise[sc]["plan_pp"])
        for k, lk in (("action_loss_pp", f"val_act/{sc}_act_loss_pp"), ("action_loss_pr", f"val_act/{sc}_act_loss_pr"), ("kl_loss", f"val_kl/{sc}_kl_loss"),
                      ("gripper_sr_pp", f"val_grip/{sc}_grip_sr_pp")):
            ref = float(fx[f"{k}_{sc}"])
            assert abs(model.logged[lk] - ref) <= 1e-3 * abs(ref) + 1e-6, (lk, model.logged[lk], ref)
    model.engine.close()
