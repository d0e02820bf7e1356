"""Generate tests/golden/val_*.npz and rollout_*.npz by running the UNMODIFIED reference (/root/reference) on CPU:
validation forward (Hulc.validation_step body -> lmp_val, hulc/models/hulc.py:770-797, :301-388) and the stateful rollout
(reset / step, :843-957).  The stochastic draws the reference makes are RECORDED (torch.rand wrapper for the two draws of
LogisticDecoderRNN._sample; the categorical plan samples are outputs) and stored as inputs of the fixture.

Run in the build container only:  python tools/gen_golden_val.py
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
warnings.filterwarnings("ignore")

from hulc_amd import spec  # noqa: E402
from hulc_amd.utils import portable_rng as prng  # noqa: E402
from hulc_amd.utils import synthetic  # noqa: E402
import ref_harness  # noqa: E402
from gen_golden import to_ref_batch  # noqa: E402

VAL_CASES = {
    # name: (Bv, Bl, S, use_clip, seed[, kind])
    "val_hulc_tiny": (2, 2, 4, True, 11),
    "val_hulc_s16": (3, 0, 16, False, 12),
    "val_gcbc_s8": (2, 2, 8, True, 13, "gcbc"),
    "val_mcil_s8": (2, 2, 8, False, 14, "mcil"),      # conf/model/mcil.yaml: continuous plans are stored as drawn (B,256)
}


class RandRecorder:
    """Wraps torch.rand: the reference's own generator draws, we only keep a copy of what it drew."""

    def __init__(self):
        self.draws = []
        self.orig = torch.rand

    def __enter__(self):
        def rand(*a, **k):
            t = self.orig(*a, **k)
            self.draws.append(t.detach().cpu().numpy().copy())
            return t
        torch.rand = rand
        return self

    def __exit__(self, *exc):
        torch.rand = self.orig


def load_params(model, P):
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.copy_(torch.from_numpy(P[n]).reshape(p.shape))


def run_val(name, case, outdir):
    Bv, Bl, S, use_clip, seed = case[:5]
    kind = case[5] if len(case) > 5 else "hulc"
    dims = spec.ModelDims(kind=kind, max_window=32, use_clip=use_clip)
    P = spec.init_all(dims, seed=seed, ln_jitter=True)
    batch = synthetic.make_batch(Bv, Bl, S, seed=seed, edge_frac=0.05, aux_mask="all")
    model = ref_harness.build_reference(kind, max_window=32, use_clip=use_clip)
    model.eval()
    load_params(model, P)
    rb = to_ref_batch(batch)
    fx = {"meta": np.array([Bv, Bl, S, int(use_clip), seed], np.int64)}
    torch.manual_seed(4321 + seed)
    with torch.no_grad():
        for sc, db in rb.items():                      # the body of validation_step (hulc.py:770-797) for one modality
            emb = model.perceptual_encoder(db["rgb_obs"], db["depth_obs"], db["robot_obs"])
            goal = model.language_goal(db["lang"]) if "lang" in sc else model.visual_goal(emb[:, -1])
            if kind == "gcbc":                          # gcbc.py:226-246 (validation_step body)
                empty_plan = torch.empty((db["actions"].shape[0]), 0)
                with RandRecorder() as rr:
                    loss, sample_act = model.action_decoder.loss_and_act(empty_plan, emb, goal, db["actions"], db["state_info"]["robot_obs"])
                assert len(rr.draws) == 2
                mae = torch.nn.functional.l1_loss(sample_act[..., :-1], db["actions"][..., :-1], reduction="none").mean(1)
                gd = sample_act[..., -1]
                m = gd > 0
                gd[m] = 1
                gd[~m] = -1
                fx[f"u_mix_pp_{sc}"], fx[f"u_act_pp_{sc}"] = rr.draws
                fx[f"action_loss_pp_{sc}"] = np.float32(loss.item())
                fx[f"mae_pp_{sc}"] = mae.numpy()
                fx[f"gripper_sr_pp_{sc}"] = np.float32(torch.mean((db["actions"][..., -1] == gd).float()).item())
                continue
            with RandRecorder() as rr:
                (plan_pp, loss_pp, plan_pr, loss_pr, kl, mae_pp, mae_pr, sr_pp, sr_pr, seq_feat) = model.lmp_val(
                    emb, goal, db["actions"], db["state_info"]["robot_obs"])
            assert len(rr.draws) == 4, len(rr.draws)
            B = emb.shape[0]
            if kind == "mcil":
                fx[f"plan_pp_{sc}"] = plan_pp.numpy().astype(np.float32)
                fx[f"plan_pr_{sc}"] = plan_pr.numpy().astype(np.float32)
            else:
                fx[f"plan_idx_pp_{sc}"] = plan_pp.reshape(B, 32, 32).argmax(-1).numpy().astype(np.int32)
                fx[f"plan_idx_pr_{sc}"] = plan_pr.reshape(B, 32, 32).argmax(-1).numpy().astype(np.int32)
            fx[f"u_mix_pp_{sc}"], fx[f"u_act_pp_{sc}"], fx[f"u_mix_pr_{sc}"], fx[f"u_act_pr_{sc}"] = rr.draws
            for k, v in (("action_loss_pp", loss_pp), ("action_loss_pr", loss_pr), ("kl_loss", kl), ("gripper_sr_pp", sr_pp), ("gripper_sr_pr", sr_pr)):
                fx[f"{k}_{sc}"] = np.float32(v.item())
            fx[f"mae_pp_{sc}"] = mae_pp.numpy()
            fx[f"mae_pr_{sc}"] = mae_pr.numpy()
            fx[f"seq_feat_{sc}"] = seq_feat.numpy()
            if "lang" in sc and use_clip:               # hulc.py:804-808
                fx[f"val_pred_clip_loss_{sc}"] = np.float32(model.clip_auxiliary_loss(seq_feat, goal, db["use_for_aux_lang_loss"]).item())
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **fx)
    # sanity: oracle vs reference (report only; the committed test does the check)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hulc_oracle as O
    for sc, mb in batch.items():
        noise = {k: fx[f"{k}_{sc}"] for k in ("plan_idx_pp", "plan_idx_pr", "plan_pp", "plan_pr", "u_mix_pp", "u_act_pp", "u_mix_pr", "u_act_pr") if f"{k}_{sc}" in fx}
        o = O.validation_forward(P, dims, mb, "lang" in sc, noise)
        if kind == "gcbc":
            print(f"[{name}/{sc}] loss ref {fx[f'action_loss_pp_{sc}']:.6f} oracle {o['action_loss_pp']:.6f} mae err {np.abs(o['mae_pp'] - fx[f'mae_pp_{sc}']).max():.2e} "
                  f"sr {fx[f'gripper_sr_pp_{sc}']:.3f}/{o['gripper_sr_pp']:.3f}")
            continue
        print(f"[{name}/{sc}] loss_pp ref {fx[f'action_loss_pp_{sc}']:.6f} oracle {o['action_loss_pp']:.6f} | kl {fx[f'kl_loss_{sc}']:.6f} {o['kl_loss']:.6f} | "
              f"mae_pp err {np.abs(o['mae_pp'] - fx[f'mae_pp_{sc}']).max():.2e} mae_pr err {np.abs(o['mae_pr'] - fx[f'mae_pr_{sc}']).max():.2e} "
              f"sr {fx[f'gripper_sr_pp_{sc}']:.3f}/{o['gripper_sr_pp']:.3f} {fx[f'gripper_sr_pr_{sc}']:.3f}/{o['gripper_sr_pr']:.3f}"
              + (f" | clip {fx[f'val_pred_clip_loss_{sc}']:.6f} {float(o['val_pred_clip_loss']):.6f}" if f"val_pred_clip_loss_{sc}" in fx else ""))


def run_rollout(name, outdir, seed=21, nsteps=5, replan_freq=2, kind="hulc"):
    """Two rollouts with the same weights: vision goal (nsteps steps, replan every replan_freq) then language goal."""
    mcil = kind == "mcil"
    dims = spec.ModelDims(kind=kind, max_window=32, use_clip=not mcil)
    P = spec.init_all(dims, seed=seed, ln_jitter=True)
    model = ref_harness.build_reference(kind, max_window=32, use_clip=not mcil)
    model.eval()
    load_params(model, P)
    model.replan_freq = replan_freq
    frames = synthetic.make_batch(1, 1, nsteps + 1, seed=seed, edge_frac=0.0, aux_mask="all")
    vis, lang = frames["vis"], frames["lang"]
    fx = {"meta": np.array([nsteps, replan_freq, seed], np.int64)}
    model.lang_embeddings = {"do the task": lang["lang"][0:1].reshape(1, 1, 384)}
    torch.manual_seed(99 + seed)
    for mode, mb in (("vis", vis), ("lang", lang)):
        model.reset()
        goal = None
        if mode == "vis":
            goal = dict(rgb_obs=dict(rgb_static=torch.from_numpy(mb["rgb_static"][:, nsteps:nsteps + 1]), rgb_gripper=torch.from_numpy(mb["rgb_gripper"][:, nsteps:nsteps + 1])),
                        depth_obs={}, robot_obs=torch.zeros(1, 1, 8))
        else:
            goal = "do the task"
        acts, plans, umix, uact = [], [], [], []
        for t in range(nsteps):
            obs = dict(rgb_obs=dict(rgb_static=torch.from_numpy(mb["rgb_static"][:, t:t + 1]), rgb_gripper=torch.from_numpy(mb["rgb_gripper"][:, t:t + 1])),
                       depth_obs={}, robot_obs=torch.zeros(1, 1, 8), robot_obs_raw=torch.from_numpy(mb["robot_obs"][:, t:t + 1]))
            with RandRecorder() as rr:
                a = model.step(obs, goal)
            assert len(rr.draws) == 2
            umix.append(rr.draws[0]); uact.append(rr.draws[1])
            plans.append(model.plan.numpy().astype(np.float32).copy() if mcil else model.plan.reshape(1, 32, 32).argmax(-1).numpy().astype(np.int32))
            acts.append(a.detach().numpy().copy())
        fx[f"actions_{mode}"] = np.concatenate(acts, 1)          # (1, nsteps, 7)
        fx[f"plan_{mode}" if mcil else f"plan_idx_{mode}"] = np.stack(plans, 0)   # (nsteps, 1, 32 | 256): the plan in force at each step
        fx[f"u_mix_{mode}"] = np.stack(umix, 0)
        fx[f"u_act_{mode}"] = np.stack(uact, 0)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **fx)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hulc_oracle as O
    for mode, mb in (("vis", vis), ("lang", lang)):
        ro = O.Rollout(P, dims, replan_freq)
        goal = dict(rgb_static=mb["rgb_static"][:, nsteps:nsteps + 1], rgb_gripper=mb["rgb_gripper"][:, nsteps:nsteps + 1]) if mode == "vis" else lang["lang"][0:1]
        worst = 0.0
        for t in range(nsteps):
            obs = dict(rgb_static=mb["rgb_static"][:, t:t + 1], rgb_gripper=mb["rgb_gripper"][:, t:t + 1], robot_obs_raw=mb["robot_obs"][:, t:t + 1])
            pk = dict(plan=fx[f"plan_{mode}"][t]) if mcil else dict(plan_idx=fx[f"plan_idx_{mode}"][t])
            a = ro.step(obs, goal, dict(pk, u_mix=fx[f"u_mix_{mode}"][t], u_act=fx[f"u_act_{mode}"][t]))
            worst = max(worst, np.abs(a - fx[f"actions_{mode}"][:, t:t + 1]).max())
        print(f"[{name}/{mode}] rollout oracle-vs-reference max |action diff| {worst:.2e}")


def run_rollout_gcbc(name, outdir, seed=23, nvis=3, nlang=3):
    """GCBC.reset / step (gcbc.py:281-320): a vision-goal rollout, reset(), then a language-goal rollout with the SAME model object —
    the reference's GCBC never clears the decoder's hidden state, so the second rollout starts from the first one's last state."""
    dims = spec.ModelDims(kind="gcbc", max_window=32, use_clip=True)
    P = spec.init_all(dims, seed=seed, ln_jitter=True)
    model = ref_harness.build_reference("gcbc", max_window=32, use_clip=True)
    model.eval()
    load_params(model, P)
    n = max(nvis, nlang)
    frames = synthetic.make_batch(1, 1, n + 1, seed=seed, edge_frac=0.0, aux_mask="all")
    vis, lang = frames["vis"], frames["lang"]
    fx = {"meta": np.array([nvis, nlang, seed], np.int64)}
    model.lang_embeddings = {"do the task": lang["lang"][0:1].reshape(1, 1, 384)}
    torch.manual_seed(77 + seed)
    for mode, mb, ns in (("vis", vis, nvis), ("lang", lang, nlang)):
        model.reset()
        if mode == "vis":
            goal = dict(rgb_obs=dict(rgb_static=torch.from_numpy(mb["rgb_static"][:, n:n + 1]), rgb_gripper=torch.from_numpy(mb["rgb_gripper"][:, n:n + 1])),
                        depth_obs={}, robot_obs=torch.zeros(1, 1, 8))
        else:
            goal = "do the task"
        acts, umix, uact = [], [], []
        for t in range(ns):
            obs = dict(rgb_obs=dict(rgb_static=torch.from_numpy(mb["rgb_static"][:, t:t + 1]), rgb_gripper=torch.from_numpy(mb["rgb_gripper"][:, t:t + 1])),
                       depth_obs={}, robot_obs=torch.zeros(1, 1, 8), robot_obs_raw=torch.from_numpy(mb["robot_obs"][:, t:t + 1]))
            with RandRecorder() as rr:
                a = model.step(obs, goal)
            assert len(rr.draws) == 2
            umix.append(rr.draws[0]); uact.append(rr.draws[1])
            acts.append(a.detach().numpy().copy())
        fx[f"actions_{mode}"] = np.concatenate(acts, 1)
        fx[f"u_mix_{mode}"] = np.stack(umix, 0)
        fx[f"u_act_{mode}"] = np.stack(uact, 0)
    np.savez_compressed(os.path.join(outdir, name + ".npz"), **fx)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import hulc_oracle as O
    ro = O.Rollout(P, dims, 30)
    for mode, mb, ns in (("vis", vis, nvis), ("lang", lang, nlang)):
        ro.reset()
        goal = dict(rgb_static=mb["rgb_static"][:, n:n + 1], rgb_gripper=mb["rgb_gripper"][:, n:n + 1]) if mode == "vis" else lang["lang"][0:1]
        worst = 0.0
        for t in range(ns):
            obs = dict(rgb_static=mb["rgb_static"][:, t:t + 1], rgb_gripper=mb["rgb_gripper"][:, t:t + 1], robot_obs_raw=mb["robot_obs"][:, t:t + 1])
            a = ro.step(obs, goal, dict(u_mix=fx[f"u_mix_{mode}"][t], u_act=fx[f"u_act_{mode}"][t]))
            worst = max(worst, np.abs(a - fx[f"actions_{mode}"][:, t:t + 1]).max())
        print(f"[{name}/{mode}] rollout oracle-vs-reference max |action diff| {worst:.2e}")


if __name__ == "__main__":
    out = os.path.join(ROOT, "tests", "golden")
    only = sys.argv[1:]
    for name, case in VAL_CASES.items():
        if not only or name in only:
            run_val(name, case, out)
    if not only or "rollout_hulc" in only:
        run_rollout("rollout_hulc", out)
    if not only or "rollout_mcil" in only:
        run_rollout("rollout_mcil", out, seed=22, kind="mcil")
    if not only or "rollout_gcbc" in only:
        run_rollout_gcbc("rollout_gcbc", out)
