"""GPU (-m gpu): the drop-in `Hulc` / `GCBC` module surface, the fit loop, checkpoints, and the Hydra-style entry point."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import hulc_oracle as O  # noqa: E402
from golden_util import load_case  # noqa: E402
from hulc_amd import config, spec  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ref_style_batch(batch, dev="cuda"):
    out = {}
    for sc, mb in batch.items():
        d = dict(rgb_obs=dict(rgb_static=torch.from_numpy(mb["rgb_static"]).to(dev), rgb_gripper=torch.from_numpy(mb["rgb_gripper"]).to(dev)),
                 depth_obs={}, robot_obs=torch.zeros(mb["actions"].shape[:2] + (8,), device=dev), actions=torch.from_numpy(mb["actions"]).to(dev),
                 state_info=dict(robot_obs=torch.from_numpy(mb["robot_obs"]).to(dev)), idx=torch.arange(mb["actions"].shape[0], device=dev),
                 plan_idx=torch.from_numpy(mb["plan_idx"].astype(np.int32)).to(dev))
        if "lang" in mb:
            d["lang"] = torch.from_numpy(mb["lang"]).to(dev)
            d["use_for_aux_lang_loss"] = torch.from_numpy(mb["use_for_aux"]).to(dev)
        out[sc] = d
    return out


def build(kind="hulc", precision="fp32", overrides=()):
    cfg = config.compose(os.path.join(ROOT, "conf"), "config", [f"model={kind}", f"trainer.precision={precision}", "datamodule.batch_size=4", *overrides])
    return config.instantiate(cfg.model, device="cuda:0", max_seq_len=32), cfg


def test_module_training_step_matches_reference_fixture():
    dims, P, batch, fx = load_case("hulc_tiny")
    model, cfg = build("hulc")
    # state_dict contract: every reference parameter name, same shapes (fixture keys come from the reference's named_parameters)
    names = {k[len("adam1/"):] for k in fx.files if k.startswith("adam1/")}
    sd = model.state_dict()
    assert names <= set(sd.keys())
    assert set(n for n, _ in model.named_parameters()) == names
    for n, (off, shape) in model.engine.layout.items():
        assert tuple(sd[n].shape) == tuple(shape)
    for extra in ("perceptual_encoder.rgb_static_encoder.spatial_softmax.x_map", "action_decoder.action_max_bound"):
        assert extra in sd
    model.load_state_dict({n: torch.from_numpy(P[n]) for n in P}, strict=False)
    model.eval()                                       # dropout off, like the fixture
    loss = model.training_step(ref_style_batch(batch), 0)
    ref = float(fx["loss_total"])
    assert abs(float(loss) - ref) <= 1e-3 * abs(ref)
    for k in ("train/kl_loss", "train/action_loss", "train/total_loss", "train/lang_clip_loss", "train/action_loss_vis", "train/kl_loss_scaled_lang"):
        assert abs(model.logged[k] - float(fx["log/" + k])) <= 1e-3 * max(1.0, abs(float(fx["log/" + k]))), k
    # gradients are exposed through .grad views of the flat buffer
    p = dict(model.named_parameters())["action_decoder.mean_fc.weight"]
    assert p.grad is not None and p.grad.data_ptr() != 0 and float(p.grad.abs().sum()) > 0
    opt = model.configure_optimizers()["optimizer"]
    w0 = p.detach().clone()
    opt.step()
    assert not torch.equal(w0, p.detach())
    model.engine.close()


def test_vision_only_without_lang_raises_like_reference():
    model, cfg = build("hulc")
    dims, P, batch, fx = load_case("hulc_visonly")
    with pytest.raises(KeyError):                      # SURVEY trap T4: KeyError 'aux_lang' (hulc.py:531)
        model.training_step(ref_style_batch(batch), 0)
    model.engine.close()


def test_fit_loop_reduces_loss_and_checkpoint_roundtrip(tmp_path):
    from hulc_amd.trainer import SyntheticDataModule, Trainer, get_last_checkpoint, ModelCheckpoint
    model, cfg = build("hulc", "bf16", ["model.plan_recognition.dropout_p=0.1"])
    dm = SyntheticDataModule(batch_size=4, max_window_size=8, modalities=["vis", "lang"], steps_per_epoch=1, seed=3)
    one = list(dm.train_dataloader(0))

    class Fixed:
        def train_dataloader(self, rank=0):
            for _ in range(12):
                yield one[0]
        def val_dataloader(self, rank=0):
            yield one[0]
    tr = Trainer(max_epochs=2, log_dir=str(tmp_path), callbacks=[ModelCheckpoint()], log_every=1)
    hist = tr.fit(model, Fixed())
    assert len(tr.val_history) == 2 and "val_act/action_loss_pp" in tr.val_history[0]                 # validation loop ran each epoch
    assert tr.val_history[1]["val_act/action_loss_pp"] < tr.val_history[0]["val_act/action_loss_pp"]     # ... and the fit improved it
    losses = [h["loss"] for h in hist]
    assert np.isfinite(losses).all() and losses[-1] < losses[0] - 0.5, losses       # Adam on a fixed batch must descend
    ck = get_last_checkpoint(str(tmp_path))
    assert ck and ck.endswith("epoch=1.ckpt")
    sd = torch.load(ck, map_location="cpu", weights_only=False)
    assert {"state_dict", "optimizer_states", "epoch", "global_step"} <= set(sd)
    w = model.state_dict()["plan_proposal.fc_model.2.weight"].cpu()
    assert torch.equal(w, sd["state_dict"]["plan_proposal.fc_model.2.weight"])
    model2, _ = build("hulc", "bf16")
    tr2 = Trainer(max_epochs=3, log_dir=str(tmp_path), log_every=1)
    tr2.fit(model2, Fixed(), ckpt_path=ck)
    assert tr2.current_epoch == 3 and model2.engine.adam_t == 36
    model.engine.close(); model2.engine.close()


def test_unsupported_options_fail_loudly():
    with pytest.raises(NotImplementedError):
        build("hulc", overrides=["model.state_recons=true"])
    with pytest.raises(NotImplementedError):
        build("hulc", overrides=["model.action_decoder.rnn_model=gru_decoder"])


def test_module_validation_step_and_rollout_match_reference():
    """Hulc.validation_step logs the reference's metric names with the reference's values (injected draws); Hulc.reset/step
    reproduce the reference rollout (fixtures from tools/gen_golden_val.py)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_util import load_rollout_case, load_val_case
    from hulc_amd.hulc import Hulc
    dims, P, batch, noise, fx = load_val_case("val_hulc_tiny")
    m = Hulc(precision="fp32", max_batch_size=2, max_seq_len=4, use_clip_auxiliary_loss=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    rb = {}
    for sc, mb in batch.items():
        d = dict(rgb_obs=dict(rgb_static=torch.from_numpy(mb["rgb_static"]), rgb_gripper=torch.from_numpy(mb["rgb_gripper"])), depth_obs={},
                 robot_obs=torch.zeros(mb["actions"].shape[:2] + (8,)), actions=torch.from_numpy(mb["actions"]),
                 state_info=dict(robot_obs=torch.from_numpy(mb["robot_obs"])), idx=torch.arange(mb["actions"].shape[0]))
        if "lang" in mb:
            d["lang"] = torch.from_numpy(mb["lang"]); d["use_for_aux_lang_loss"] = torch.from_numpy(mb["use_for_aux"])
        rb[sc] = d
    m.eval()
    out = m.validation_step(rb, 0, noise=noise)
    for sc in batch:
        assert abs(m.logged[f"val_act/{sc}_act_loss_pp"] - float(fx[f"action_loss_pp_{sc}"])) < 1e-3 * abs(float(fx[f"action_loss_pp_{sc}"]))
        assert abs(m.logged[f"val_kl/{sc}_kl_loss"] - float(fx[f"kl_loss_{sc}"])) < 1e-3 * abs(float(fx[f"kl_loss_{sc}"])) + 1e-7
        assert abs(m.logged[f"val_total_mae/{sc}_total_mae_pp"] - float(fx[f"mae_pp_{sc}"].mean())) < 2e-3
        assert abs(m.logged[f"val_grip/{sc}_grip_sr_pr"] - float(fx[f"gripper_sr_pr_{sc}"])) < 1e-6
        assert out[f"sampled_plan_pp_{sc}"].shape == (batch[sc]["actions"].shape[0], 1024)
        assert np.array_equal(out[f"sampled_plan_pp_{sc}"].reshape(-1, 32, 32).argmax(-1).cpu().numpy(), noise[sc]["plan_idx_pp"])
    # rollout: same module class, weights of the rollout fixture
    dims, P, frames, nsteps, replan_freq, rfx = load_rollout_case()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()})
    m.replan_freq = replan_freq
    m.lang_embeddings = {"do the task": frames["lang"]["lang"][0:1]}
    for mode in ("vis", "lang"):
        mb = frames[mode]
        m.reset()
        goal = "do the task" if mode == "lang" else dict(rgb_obs=dict(rgb_static=torch.from_numpy(mb["rgb_static"][:, nsteps:nsteps + 1]),
                                                                        rgb_gripper=torch.from_numpy(mb["rgb_gripper"][:, nsteps:nsteps + 1])))
        for t in range(nsteps):
            obs = dict(rgb_obs=dict(rgb_static=torch.from_numpy(mb["rgb_static"][:, t:t + 1]), rgb_gripper=torch.from_numpy(mb["rgb_gripper"][:, t:t + 1])),
                       depth_obs={}, robot_obs=torch.zeros(1, 1, 8), robot_obs_raw=torch.from_numpy(mb["robot_obs"][:, t:t + 1]))
            a = m.step(obs, goal, noise=dict(plan_idx=rfx[f"plan_idx_{mode}"][t][0], u_mix=rfx[f"u_mix_{mode}"][t][0, 0], u_act=rfx[f"u_act_{mode}"][t][0, 0]))
            assert a.shape == (1, 1, 7)
            assert np.abs(a.numpy()[0, 0] - rfx[f"actions_{mode}"][0, t]).max() <= 2e-3, (mode, t)
