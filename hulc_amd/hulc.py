"""`Hulc` / `GCBC` — the reference's LightningModule surface over the MI355X-native step engine.

Drop-in for `hulc.models.hulc.Hulc` (reference hulc/models/hulc.py:27-153) and `hulc.models.gcbc.GCBC`
(hulc/models/gcbc.py:11-48): same constructor keywords (the Hydra `conf/model/*.yaml` tree instantiates it unchanged),
same hooks (`training_step`, `configure_optimizers`, `set_kl_beta`, `on_fit_start`, `log`), same `state_dict` keys and
shapes, same logged metric names (hulc.py:470-536).  Differences forced by the design, both documented in INTEGRATION.md:

* the step's forward AND backward run inside `training_step` (C-ABI `hulc_forward_loss` + `hulc_backward` per
  modality): the returned loss is a detached scalar tensor, gradients are already in `.grad` (views of one flat
  buffer) — Lightning users set `automatic_optimization = False`-style manual optimisation, or use
  `hulc_amd.trainer.fit`;
* `configure_optimizers()` returns a `FusedAdam` handle (one HIP kernel over the flat buffer, DP mean folded in).

There is no torch/CPU fallback: constructing the module without the HIP library or without a GPU raises.
"""
from __future__ import annotations

import logging
import math
import os
from typing import Any, Dict, Iterator, Optional, Tuple

import numpy as np
import torch

from . import parallel, spec
from .engine import StepEngine

logger = logging.getLogger(__name__)

def _get(cfg, key, default=None):
    if cfg is None:
        return default
    if isinstance(cfg, dict):
        return cfg.get(key, default)
    return getattr(cfg, key, default)


class FusedAdam:
    """Optimizer handle with the torch.optim surface Lightning touches (step / zero_grad / param_groups / state_dict).

    `kind` selects what the reference's conf tree can instantiate at hulc.py:239-240: "adam" (torch.optim.Adam, conf/model/optimizer/adam.yaml;
    weight_decay = L2), "adamw" (torch.optim.AdamW, adamw.yaml: decoupled decay), "sgd" (torch.optim.SGD, sgd.yaml: momentum buffer = the
    first-moment buffer).  One fused pass over the flat buffers either way (hulc_optimizer_step)."""

    def __init__(self, module: "Hulc", lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0, kind: str = "adam",
                 momentum: float = 0.0, dampening: float = 0.0, nesterov: bool = False):
        if kind not in ("adam", "adamw", "sgd"):
            raise NotImplementedError(f"optimizer kind {kind!r}")
        self.module, self.kind = module, kind
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=float(weight_decay), momentum=float(momentum), dampening=float(dampening),
                                  nesterov=bool(nesterov), params=list(module.parameters()))]
        self.param_groups[0]["initial_lr"] = lr

    def zero_grad(self, set_to_none: bool = False):
        self.module.engine.zero_grads()

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        g = self.param_groups[0]
        world = parallel.world_size()
        if not self.module._grads_reduced:               # training_step already overlapped the all-reduce with the backward
            if getattr(self.module.engine, "has_comm", False):
                self.module.engine.allreduce_grads(getattr(self.module.engine, "comm_bucket_dtype", "fp32"))
            else:
                parallel.allreduce_sum_(self.module.engine.flat_grads)
        self.module._grads_reduced = False
        if self.kind == "adam" and not g["weight_decay"]:
            self.module.engine.adam_step(lr=g["lr"], b1=g["betas"][0], b2=g["betas"][1], eps=g["eps"], grad_scale=1.0 / world)
        else:
            self.module.engine.optimizer_step(self.kind, lr=g["lr"], b1=g["betas"][0], b2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"],
                                              momentum=g["momentum"], dampening=g["dampening"], nesterov=g["nesterov"], grad_scale=1.0 / world)
        return loss

    def state_dict(self):
        e = self.module.engine
        sd = dict(step=e.adam_t, exp_avg=e.adam_m.clone(), exp_avg_sq=e.adam_v.clone(),
                  param_groups=[{k: v for k, v in self.param_groups[0].items() if k != "params"}])
        if e.dtype == "fp16":       # Lightning keeps GradScaler.state_dict() next to the optimizer states ("native_amp_scaling_state")
            st = e.scaler_state()
            # torch's optimizer `step` only counts the steps GradScaler let through; the fp16 Adam kernel takes its bias corrections from that
            # (device-side) count, so it is what a checkpoint must carry — `calls` keeps the number of optimizer.step() calls made
            sd["step"] = int(st["taken_steps"])
            sd["grad_scaler"] = dict(scale=st["scale"], _growth_tracker=st["growth_tracker"], skipped_steps=int(st["skipped_steps"]), calls=int(e.adam_t))
        return sd

    def load_state_dict(self, sd):
        e = self.module.engine
        e.adam_t = int(sd["step"])
        e.adam_m.copy_(sd["exp_avg"])
        e.adam_v.copy_(sd["exp_avg_sq"])
        if e.dtype == "fp16":       # the taken-step count travels as `step` (torch semantics); without it Adam would restart its bias corrections at t = 1 on warm moments
            gs = sd.get("grad_scaler") or {}
            e.adam_t = int(gs.get("calls", sd["step"]))
            if gs:
                e.scaler_load(float(gs["scale"]), int(gs.get("_growth_tracker", 0)), int(sd["step"]))
            else:                   # a checkpoint without scaler state (e.g. saved by an fp32 / bf16 run): keep the current scale, restore the count
                st = e.scaler_state()
                e.scaler_load(float(st["scale"]), int(st["growth_tracker"]), int(sd["step"]))


class LambdaSchedule:
    """torch.optim.lr_scheduler.LambdaLR as transformers' schedule factories build it: lr = initial_lr * f(last_epoch), evaluated once at
    construction (last_epoch 0) and after every .step().  The learning rate is a per-call argument of hulc_adam_step / hulc_optimizer_step,
    so a schedule is host-side arithmetic only."""

    def __init__(self, optimizer: FusedAdam, fn):
        self.optimizer, self.fn = optimizer, fn
        self.base_lrs = [g.setdefault("initial_lr", g["lr"]) for g in optimizer.param_groups]
        self.last_epoch = 0
        self._apply()

    def _apply(self):
        for g, b in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = b * self.fn(self.last_epoch)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs)}

    def load_state_dict(self, sd):
        self.last_epoch = int(sd["last_epoch"])
        self.base_lrs = list(sd.get("base_lrs", self.base_lrs))
        self._apply()


class ConstantSchedule(LambdaSchedule):
    """transformers.get_constant_schedule (conf/model/lr_scheduler/constant.yaml:1) == LambdaLR(lambda _: 1.0)."""

    def __init__(self, optimizer: FusedAdam):
        super().__init__(optimizer, lambda _step: 1.0)


class LinearWarmupSchedule(LambdaSchedule):
    """transformers.get_linear_schedule_with_warmup (conf/model/lr_scheduler/linear_schedule_with_warmup.yaml): linear 0 -> 1 over the warm-up,
    then linear 1 -> 0 at num_training_steps."""

    def __init__(self, optimizer: FusedAdam, num_warmup_steps: int, num_training_steps: int):
        w, n = int(num_warmup_steps), int(num_training_steps)

        def fn(step):
            if step < w:
                return float(step) / float(max(1, w))
            return max(0.0, float(n - step) / float(max(1, n - w)))

        super().__init__(optimizer, fn)


class CosineWarmupSchedule(LambdaSchedule):
    """transformers.get_cosine_schedule_with_warmup (conf/model/lr_scheduler/cosine_schedule_with_warmup.yaml: num_cycles 0.5)."""

    def __init__(self, optimizer: FusedAdam, num_warmup_steps: int, num_training_steps: int, num_cycles: float = 0.5):
        import math
        w, n, c = int(num_warmup_steps), int(num_training_steps), float(num_cycles)

        def fn(step):
            if step < w:
                return float(step) / float(max(1, w))
            # clamped at the end of the curve: a run that steps past num_training_steps (a mis-inferred length) stays at the final
            # factor instead of climbing the cosine again (identical to transformers' lambda for every step <= num_training_steps)
            progress = min(1.0, float(step - w) / float(max(1, n - w)))
            return max(0.0, 0.5 * (1.0 + math.cos(math.pi * c * 2.0 * progress)))

        super().__init__(optimizer, fn)


class Hulc(torch.nn.Module):
    KIND = "hulc"

    def __init__(
        self,
        perceptual_encoder=None,
        plan_proposal=None,
        plan_recognition=None,
        language_goal=None,
        visual_goal=None,
        action_decoder=None,
        kl_beta: float = 0.01,
        kl_balancing_mix: float = 0.8,
        state_recons: bool = False,
        state_recon_beta: float = 0.5,
        use_bc_z_auxiliary_loss: bool = False,
        bc_z_auxiliary_loss_beta: float = 1.0,
        use_mia_auxiliary_loss: bool = False,
        mia_auxiliary_loss_beta: float = 1.0,
        optimizer=None,
        lr_scheduler=None,
        distribution=None,
        val_instructions=None,
        use_clip_auxiliary_loss: bool = True,
        clip_auxiliary_loss_beta: float = 3.0,
        replan_freq: int = 30,
        bc_z_lang_decoder=None,
        mia_lang_discriminator=None,
        proj_vis_lang=None,
        # ---- engine options (not in the reference; defaults keep the reference behaviour)
        precision: str = "bf16",
        max_batch_size: int = 64,
        max_seq_len: Optional[int] = None,
        device: str = "cuda:0",
        seed: int = 42,
    ):
        super().__init__()
        # ---- options outside the hot path are rejected loudly instead of silently ignored (SURVEY §2.1 OUT OF SCOPE rows)
        if state_recons or use_bc_z_auxiliary_loss or use_mia_auxiliary_loss or bc_z_lang_decoder or mia_lang_discriminator:
            raise NotImplementedError("state_recons / bc_z / mia auxiliary losses are disabled in every BASELINE config and not built")
        pr = plan_recognition
        ad = action_decoder
        self.kind = self.KIND
        rnn_type = "nn.RNN"
        # conf/model/mcil.yaml: the same Hulc class with a BiRNN plan recognition net, a continuous plan distribution
        # and the mcil decoder options; the three go together (any other mix has no built engine path)
        mcil_flags = [_get(distribution, "dist", "discrete") == "continuous", "BiRNN" in str(_get(pr, "_target_", "Transformers")),
                      not _get(ad, "gripper_control", True), not _get(ad, "discrete_gripper", True)]
        if any(mcil_flags):
            if not all(mcil_flags) or self.KIND != "hulc" or _get(ad, "perceptual_emb_slice", None) is not None:
                raise NotImplementedError("continuous distribution, BiRNN plan recognition and the mcil decoder options (gripper_control / "
                                          "discrete_gripper false, no perceptual_emb_slice) are only built together (conf/model/mcil.yaml)")
            if use_clip_auxiliary_loss:
                raise NotImplementedError("conf/model/mcil.yaml trains without the CLIP auxiliary loss (proj_vis_lang: none)")
            rnn_type = str(_get(pr, "rnn_type", "nn.RNN"))
            if rnn_type not in ("nn.RNN", "nn.GRU"):
                raise NotImplementedError(f"plan_recognition.rnn_type {rnn_type!r}: nn.RNN (birnn.yaml default) and nn.GRU are built")
            chk = [(_get(pr, "plan_features", 256), 256), (_get(pr, "birnn_dropout_p", 0.0), 0.0),
                   (_get(distribution, "plan_features", 256), 256)]
            of = _get(ad, "out_features", 7)
            if not isinstance(of, str) and int(of) != 7:
                raise NotImplementedError(f"action_decoder.out_features {of!r}: the built mixture has 7 dimensions")
            self.kind = "mcil"
        else:
            if pr is not None and "Transformers" not in str(_get(pr, "_target_", "Transformers")):
                raise NotImplementedError("plan recognition: the transformer (hulc / gcbc) and the BiRNN of conf/model/mcil.yaml are built")
            chk = [(_get(ad, "gripper_control", True), True), (_get(ad, "discrete_gripper", True), True),
                   (_get(pr, "num_heads", 8), 8), (_get(pr, "num_layers", 2), 2), (_get(pr, "encoder_hidden_size", 2048), 2048),
                   (_get(pr, "fc_hidden_size", 4096), 4096), (_get(distribution, "category_size", 32), 32),
                   (_get(distribution, "class_size", 32), 32)]
        chk += [(_get(ad, "n_mixtures", 10), 10), (_get(ad, "hidden_size", 2048), 2048), (_get(ad, "num_layers", 2), 2),
                (_get(ad, "rnn_model", "rnn_decoder"), "rnn_decoder"), (_get(ad, "policy_rnn_dropout_p", 0.0), 0.0),
                (_get(visual_goal, "latent_goal_features", 32), 32),
                # options that change the maths and have no built path: action bounds other than +-1 (the engine's bin width is
                # 1/(num_classes-1), logistic_decoder_rnn.py:157-182), loaded statistics.yaml bounds, the plan-recognition normalisation flags
                (bool(_get(ad, "load_action_bounds", False)), False), (bool(_get(pr, "encoder_normalize", False)), False),
                (bool(_get(pr, "positional_normalize", False)), False)]
        for key, want in (("act_max_bound", 1.0), ("act_min_bound", -1.0)):
            b = _get(ad, key, None)
            if b is not None and not isinstance(b, str) and any(float(x) != want for x in np.asarray(b, np.float64).reshape(-1)):
                raise NotImplementedError(f"action_decoder.{key} {b!r}: the built loss uses bounds of {want:+.0f} on every dimension (conf/datamodule/default.yaml)")
        if self.KIND != "mcil" and not any(mcil_flags):
            chk += [(bool(_get(pr, "position_embedding", True)), True)]
            sl = _get(ad, "perceptual_emb_slice", [64, 128])
            if sl is not None and not isinstance(sl, str) and [int(x) for x in sl] != [64, 128]:
                raise NotImplementedError(f"action_decoder.perceptual_emb_slice {sl!r}: the decoder is built on the gripper half [64, 128] (hulc_default.yaml:15)")
        for got, want in chk:
            if got != want:
                raise NotImplementedError(f"configuration value {got!r} differs from the built architecture ({want!r})")
        self.use_clip_auxiliary_loss = bool(use_clip_auxiliary_loss)
        self.clip_auxiliary_loss_beta = float(clip_auxiliary_loss_beta)
        self.kl_beta = float(kl_beta)
        self.kl_balancing_mix = float(kl_balancing_mix)
        self.replan_freq = replan_freq
        self.val_instructions = dict(val_instructions) if val_instructions else {}      # task -> [instruction] (conf/annotations/new_playtable_validation.yaml)
        self._clip_gt = None
        self.modality_scope = "vis"
        self.optimizer_config = optimizer
        self.lr_scheduler = lr_scheduler
        mw = _get(pr, "max_position_embeddings", 32) or 32
        max_window = 32 if isinstance(mw, str) else int(mw)      # a dangling ${...} (vision_only datasets) falls back to 32
        self.dims = spec.ModelDims(kind=self.kind, max_window=max_window, use_clip=self.use_clip_auxiliary_loss,
                                   rnn_type="gru" if (self.kind == "mcil" and rnn_type == "nn.GRU") else "rnn")
        # Lightning precision flags: 16 / "16-mixed" = native AMP fp16 + GradScaler (the reference's conf/trainer/play_trainer.yaml:3) ->
        # the fp16 engine with its on-device loss scaler; bf16 needs none; 32 = the fp32 parity engine
        pmap = {"16": "fp16", "fp16": "fp16", "16-mixed": "fp16", "bf16": "bf16", "bf16-mixed": "bf16", "32": "fp32", "fp32": "fp32", "32-true": "fp32"}
        if str(precision) not in pmap:
            raise ValueError(f"precision {precision!r}: one of {sorted(pmap)}")
        self.precision = pmap[str(precision)]
        self.dropout_p = 0.0 if self.kind == "mcil" else float(_get(pr, "dropout_p", 0.1))
        self.pair_modalities = os.environ.get("HULC_PAIR", "1") != "0"      # vis + lang of a step as one paired pass (see training_step)
        self._engine_kw = dict(max_batch=int(max_batch_size) * (2 if self.pair_modalities else 1), max_seq=int(max_seq_len or max_window), dtype=self.precision, device=device,
                               kl_beta=self.kl_beta, kl_balancing_mix=self.kl_balancing_mix,
                               num_classes=int(_get(ad, "num_classes", 256 if self.kind == "mcil" else 10)), gripper_alpha=float(_get(ad, "gripper_alpha", 1.0)),
                               log_scale_min=float(_get(ad, "log_scale_min", -7.0)), seed=int(seed))
        self.engine = StepEngine(self.dims, dropout_p=self.dropout_p, **self._engine_kw)
        self._train_mode = True
        self._params = {n: torch.nn.Parameter(v, requires_grad=True) for n, v in self.engine.views(self.engine.flat_params).items()}
        for n, g in self.engine.views(self.engine.flat_grads).items():
            self._params[n].grad = g
        self.reset_parameters(seed)
        self.logged: Dict[str, float] = {}
        self._epoch_acc: Dict[str, list] = {}
        self._grads_reduced = False
        self._comm_tried = False
        self.global_step = 0
        self.rollout_step_counter = 0
        self.latent_goal = None
        self.plan = None
        self.lang_embeddings = None

    # ---- parameters / state dict (reference key names) -------------------------------------------
    def reset_parameters(self, seed: int = 0):
        self.engine.load_numpy(spec.init_all(self.dims, seed=seed))
        parallel.broadcast_(self.engine.flat_params)
        self.engine.prepare_weights()

    def named_parameters(self, prefix: str = "", recurse: bool = True) -> Iterator[Tuple[str, torch.nn.Parameter]]:  # type: ignore[override]
        for n, p in self._params.items():
            yield (prefix + ("." if prefix else "") + n, p)

    def parameters(self, recurse: bool = True):  # type: ignore[override]
        for _, p in self.named_parameters():
            yield p

    def _buffers_dict(self) -> Dict[str, torch.Tensor]:
        lin = torch.linspace(-1.0, 1.0, 21)
        ad = "action_decoder."
        ss = "perceptual_encoder.rgb_static_encoder.spatial_softmax."
        nd = self.dims.mix_dims                       # logistic_decoder_rnn.py:61: out_features - 1 with the discrete gripper head
        buf = {ss + "x_map": lin.repeat_interleave(21), ss + "y_map": lin.repeat(21), ss + "temperature": torch.ones(1),
               ad + "one_hot_embedding_eye": torch.eye(10), ad + "ones": torch.ones(1, 1, 10),
               ad + "action_max_bound": torch.ones(1, 1, nd, 10), ad + "action_min_bound": -torch.ones(1, 1, nd, 10)}
        if self.kind != "mcil":
            buf[ad + "gripper_bounds"] = torch.tensor([-1.0, 1.0])     # :169-170, discrete_gripper only
        return buf

    def state_dict(self, *args, **kwargs):  # type: ignore[override]
        sd = {n: p.detach().clone() for n, p in self._params.items()}
        sd.update(self._buffers_dict())
        return sd

    def load_state_dict(self, state_dict, strict: bool = True):  # type: ignore[override]
        missing = [n for n in self._params if n not in state_dict]
        unexpected = [k for k in state_dict if k not in self._params and k not in self._buffers_dict()]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]} unexpected {unexpected[:5]}")
        # buffers that parametrise the loss: a checkpoint trained with other bounds would load and run with wrong bin widths
        for k, want in self._buffers_dict().items():
            if k in state_dict and any(k.endswith(x) for x in ("action_max_bound", "action_min_bound", "gripper_bounds", "temperature")):
                got = state_dict[k].detach().to("cpu", torch.float32).reshape(-1)
                if got.numel() != want.numel() or not torch.allclose(got, want.reshape(-1), atol=1e-6):
                    raise RuntimeError(f"load_state_dict: buffer {k} = {got[:4].tolist()}... differs from the built-in value {want.reshape(-1)[:4].tolist()}... "
                                       "(the engine hard-codes action bounds +-1 and spatial-softmax temperature 1)")
        with torch.no_grad():
            for n, p in self._params.items():
                if n in state_dict:
                    src = state_dict[n]
                    if n.endswith("position_embeddings.weight") and src.shape[0] > p.shape[0]:
                        src = src[: p.shape[0]]          # hulc/utils/utils.py:9-11 trims position-embedding rows
                    p.copy_(src.to(p.device, torch.float32).reshape(p.shape))
        self.engine.prepare_weights()
        return missing, unexpected

    # ---- Lightning-style hooks -------------------------------------------------------------------------------
    @property
    def device(self):
        return self.engine.device

    def train(self, mode: bool = True):  # type: ignore[override]
        self._train_mode = bool(mode)
        self.engine.set_dropout(self.dropout_p if self._train_mode else 0.0)
        return self

    def eval(self):  # type: ignore[override]
        return self.train(False)

    def log(self, name: str, value, on_step: bool = False, on_epoch: bool = True, batch_size: Optional[int] = None, **kw):
        """LightningModule.log as the reference uses it (hulc.py:470-536: on_step=False, on_epoch=True, batch_size=b): `logged` keeps the
        latest value, `epoch_metrics()` the batch-size-weighted mean over the epoch that Lightning reduces at epoch end."""
        v = float(value)
        self.logged[name] = v
        if on_epoch:
            w = float(batch_size) if batch_size else 1.0
            acc = self._epoch_acc.setdefault(name, [0.0, 0.0])
            acc[0] += v * w
            acc[1] += w

    def epoch_metrics(self, reset: bool = False) -> Dict[str, float]:
        """Weighted epoch means of every metric logged with on_epoch=True since the last reset, averaged over ranks (sync_dist)."""
        names = sorted(self._epoch_acc)
        if not names:
            return {}
        if parallel.world_size() > 1:
            t = torch.tensor([[self._epoch_acc[n][0], self._epoch_acc[n][1]] for n in names], dtype=torch.float64, device=self.device)
            torch.distributed.all_reduce(t)
            sums = {n: (float(t[i, 0]), float(t[i, 1])) for i, n in enumerate(names)}
        else:
            sums = {n: tuple(self._epoch_acc[n]) for n in names}
        if reset:
            self._epoch_acc = {}
        return {n: s / max(w, 1e-30) for n, (s, w) in sums.items()}

    def set_kl_beta(self, kl_beta):
        """hulc.py:563-565 — called by the KL-schedule callbacks."""
        self.kl_beta = float(kl_beta)
        self.engine.set_kl_beta(self.kl_beta)

    def on_fit_start(self) -> None:
        """hulc.py:697-737: preprocessing for the CLIP ground-truth metrics (`lang_gt/*`, validation only, never used for training).  Reads
        <train>/lang_annotations/auto_lang_ann.npy (instructions, tasks, embeddings), <val>/.../auto_lang_ann.npy (task of every annotated
        validation episode) and <val>/.../embeddings.npy (one embedding per validation task) through the datamodule's lang datasets.
        The reference orders the unique instructions and the task ids by Python set iteration; here first occurrence decides — every logged
        metric is invariant to that order.  A datamodule without `train_datasets` (SyntheticDataModule) has no annotations: the metrics are
        then not logged."""
        self._clip_gt = None
        if not self.use_clip_auxiliary_loss:
            return
        dm = getattr(getattr(self, "trainer", None), "datamodule", None)
        if dm is None or not hasattr(dm, "train_datasets"):
            return
        import pathlib
        train_dataset, val_dataset = dm.train_datasets["lang"], dm.val_datasets["lang"]
        load = lambda ds, f: np.load(pathlib.Path(ds.abs_datasets_dir) / ds.lang_folder / f, allow_pickle=True).item()
        lang_data_train = load(train_dataset, "auto_lang_ann.npy")
        lang_data_val = load(val_dataset, "auto_lang_ann.npy")
        lang_embeddings_val = load(val_dataset, "embeddings.npy")
        ann = list(lang_data_train["language"]["ann"])
        first: Dict[str, int] = {}
        for i, a in enumerate(ann):
            first.setdefault(a, i)
        train_lang_ids = list(first.values())                                            # one row per distinct instruction (hulc.py:714-717)
        emb = np.asarray(lang_data_train["language"]["emb"])[train_lang_ids]
        train_lang_emb = emb.reshape(len(train_lang_ids), -1).astype(np.float32)         # (m,1,384) -> (m,384) (.squeeze(), :719)
        train_lang_tasks = [lang_data_train["language"]["task"][i] for i in train_lang_ids]
        task_to_id: Dict[str, int] = {}
        for t in train_lang_tasks:
            task_to_id.setdefault(t, len(task_to_id))
        val_lang_tasks, val_lang_emb = [], []
        for val_task in (self.val_instructions or {}):                                   # hulc.py:729-735
            if val_task not in task_to_id:
                continue
            val_lang_tasks.append(val_task)
            val_lang_emb.append(np.asarray(lang_embeddings_val[val_task]["emb"][0], np.float32).reshape(1, -1))
        if not val_lang_emb:
            raise RuntimeError("on_fit_start: no task of model.val_instructions appears in the training annotations (the reference fails in torch.cat, hulc.py:736)")
        self._clip_gt = dict(train_emb=train_lang_emb, val_emb=np.concatenate(val_lang_emb).astype(np.float32),
                             train_task_ids=np.array([task_to_id[t] for t in train_lang_tasks]), val_task_ids=np.array([task_to_id[t] for t in val_lang_tasks]),
                             task_to_id=task_to_id, val_tasks=list(lang_data_val["language"]["task"]), lang_lookup=val_dataset.lang_lookup, encoded=False)

    def on_validation_epoch_start(self) -> None:
        """hulc.py:967-974: encoded_lang_train / encoded_lang_val = language_goal(instruction embeddings) with the current weights; kept on the
        device together with their proj_vis_lang projection (include/hulc_hip.h hulc_clip_gt_encode)."""
        gt = getattr(self, "_clip_gt", None)
        if gt is None:
            return
        self.engine.clip_gt_encode(gt["train_emb"], 0)
        self.engine.clip_gt_encode(gt["val_emb"], 1)
        gt["encoded"] = True

    @staticmethod
    def _clip_groundtruth_loss(logits: np.ndarray, task_ids: np.ndarray, gt_tasks: np.ndarray):
        """hulc.py:1031-1043 on the (n,m) logits_per_image the engine returns: per-row min-max normalised scores, sum over the instructions
        of the ground-truth task minus sum over the others, mean over rows; success rate of the arg-max instruction's task."""
        scores = logits.astype(np.float32) - logits.min(1, keepdims=True)
        scores = scores / (scores.max(1, keepdims=True) - scores.min(1, keepdims=True))
        pos = task_ids[None, :] == gt_tasks[:, None]
        loss = np.float32(np.mean(np.where(pos, scores, 0).sum(1, dtype=np.float32) - np.where(pos, 0, scores).sum(1, dtype=np.float32)))
        sr = float(np.mean(task_ids[np.argmax(scores, 1)] == gt_tasks))
        return float(loss), sr

    def clip_groundtruth(self, idx, use_for_aux_loss) -> None:
        """hulc.py:980-1005: CLIP ground-truth metric of the lang batch the engine validated last.  idx: episode indices of the batch;
        use_for_aux_loss: bool mask (None = every row, like the reference's None check)."""
        gt = getattr(self, "_clip_gt", None)
        if gt is None:
            return
        idx = np.asarray(idx.cpu() if torch.is_tensor(idx) else idx).reshape(-1)
        mask = np.ones(len(idx), bool) if use_for_aux_loss is None else np.asarray(use_for_aux_loss.cpu() if torch.is_tensor(use_for_aux_loss) else use_for_aux_loss, bool).reshape(-1)
        if not mask.any():
            return
        if not gt["encoded"]:
            raise RuntimeError("clip_groundtruth before on_validation_epoch_start (the reference fails on self.encoded_lang_train, hulc.py:996)")
        gt_tasks = np.array([gt["task_to_id"][gt["val_tasks"][gt["lang_lookup"][int(i)]]] for i in idx])[mask]
        train_score, train_sr = self._clip_groundtruth_loss(self.engine.clip_gt_scores(0), gt["train_task_ids"], gt_tasks)
        val_score, val_sr = self._clip_groundtruth_loss(self.engine.clip_gt_scores(1), gt["val_task_ids"], gt_tasks)
        self.log("lang_gt/train_gt", train_score, sync_dist=True)
        self.log("lang_gt/val_gt", val_score, sync_dist=True)
        self.log("lang_gt/train_sr", train_sr, sync_dist=True)
        self.log("lang_gt/val_sr", val_sr, sync_dist=True)

    @property
    def num_training_steps(self) -> int:
        """hulc.py:189-216: total optimizer steps inferred from the trainer and its datamodule — (batches per epoch // (accumulation x devices))
        x max_epochs, capped by max_steps; = the number of optimizer steps Trainer.fit really takes (it honours limit_train_batches and
        rejects accumulate_grad_batches > 1).  This package's Trainer attaches itself as `module.trainer` (with `.datamodule`) before it calls
        configure_optimizers, like Lightning does."""
        tr = getattr(self, "trainer", None)
        if tr is None:
            raise RuntimeError("num_training_steps needs module.trainer (set by Trainer.fit); pass lr_scheduler.num_training_steps >= 0 otherwise")
        from .trainer import epoch_batches      # the fit loop's own arithmetic: per-rank `steps_per_epoch` is not divided by the devices again, an
        dm = getattr(tr, "datamodule", None)    # un-sharded loader is (the reference measures the un-sharded loader, hulc.py:197-199, 209-211)
        num_devices = max(1, int(getattr(tr, "world", 1)))
        per_epoch, _ = epoch_batches(dm, getattr(tr, "limit_train_batches", None), num_devices)
        if per_epoch == float("inf"):
            raise RuntimeError("num_training_steps: the training loader has no len(); set lr_scheduler.num_training_steps or an int limit_train_batches")
        max_estimated = (per_epoch // int(getattr(tr, "accumulate_grad_batches", 1))) * int(tr.max_epochs)
        if tr.max_steps and 0 < tr.max_steps < max_estimated:
            return int(tr.max_steps)
        return max_estimated

    def compute_warmup(self, num_training_steps: int, num_warmup_steps):
        """hulc.py:218-237: num_training_steps < 0 -> inferred; a float num_warmup_steps is a fraction of the training steps."""
        if num_training_steps < 0:
            num_training_steps = self.num_training_steps
        if isinstance(num_warmup_steps, float):
            num_warmup_steps *= num_training_steps
        return int(num_training_steps), int(num_warmup_steps)

    def configure_optimizers(self):
        """hulc.py:239-252: the optimizer of conf/model/optimizer/*.yaml over all parameters + the per-step scheduler of conf/model/lr_scheduler/*.yaml."""
        oc = self.optimizer_config
        tgt = str(_get(oc, "_target_", "torch.optim.Adam"))
        lr, wd = float(_get(oc, "lr", 2e-4)), _get(oc, "weight_decay", None)
        if tgt.endswith(".AdamW") or tgt == "AdamW":
            opt = FusedAdam(self, lr=lr, betas=tuple(_get(oc, "betas", (0.9, 0.999))), eps=float(_get(oc, "eps", 1e-8)),
                            weight_decay=float(1e-2 if wd is None else wd), kind="adamw")          # torch.optim.AdamW's default decay is 1e-2
        elif tgt.endswith(".Adam") or tgt == "Adam":
            opt = FusedAdam(self, lr=lr, betas=tuple(_get(oc, "betas", (0.9, 0.999))), eps=float(_get(oc, "eps", 1e-8)), weight_decay=float(wd or 0.0))
        elif tgt.endswith(".SGD") or tgt == "SGD":
            opt = FusedAdam(self, lr=lr, weight_decay=float(wd or 0.0), kind="sgd", momentum=float(_get(oc, "momentum", 0.0) or 0.0),
                            dampening=float(_get(oc, "dampening", 0.0) or 0.0), nesterov=bool(_get(oc, "nesterov", False)))
        else:
            raise NotImplementedError(f"optimizer {tgt}: the reference ships torch.optim.Adam / AdamW / SGD (conf/model/optimizer/*.yaml)")
        ls = self.lr_scheduler
        sc = str(_get(ls, "_target_", "transformers.get_constant_schedule"))
        if _get(ls, "num_warmup_steps", None) is not None:
            n, w = self.compute_warmup(int(_get(ls, "num_training_steps", -1)), _get(ls, "num_warmup_steps", 0))
            if parallel.rank() == 0:
                print(f"[hulc_amd] Inferring number of training steps, set to {n}; warm-up steps {w}", flush=True)
            if "cosine_schedule_with_warmup" in sc:
                sched = CosineWarmupSchedule(opt, w, n, float(_get(ls, "num_cycles", 0.5)))
            elif "linear_schedule_with_warmup" in sc:
                sched = LinearWarmupSchedule(opt, w, n)
            else:
                raise NotImplementedError(f"lr scheduler {sc}: the reference ships the constant, linear-warmup and cosine-warmup schedules")
        elif "constant_schedule" in sc and "warmup" not in sc:
            sched = ConstantSchedule(opt)
        else:
            raise NotImplementedError(f"lr scheduler {sc}: the reference ships the constant, linear-warmup and cosine-warmup schedules")
        return {"optimizer": opt, "lr_scheduler": {"scheduler": sched, "interval": "step", "frequency": 1}}

    @staticmethod
    def _modality_batch(dataset_batch: Dict[str, Any], is_lang: bool, device) -> Dict[str, Any]:
        """Reference batch dict (hulc.py:395-414) -> engine inputs."""
        f = lambda t: t.to(device=device, dtype=torch.float32, non_blocking=True)
        # uint8 (B,S,H,W,3) dataset frames take the fused ingest path (scale / normalise / RandomShiftsAug inside conv1, include/hulc_hip.h
        # frames_u8) and must NOT be cast: a float copy of 0..255 values would be fed unnormalised
        img = lambda t: t.to(device=device, non_blocking=True) if t.dtype == torch.uint8 else f(t)
        mb = dict(rgb_static=img(dataset_batch["rgb_obs"]["rgb_static"]), rgb_gripper=img(dataset_batch["rgb_obs"]["rgb_gripper"]),
                  actions=f(dataset_batch["actions"]), robot_obs=f(dataset_batch["state_info"]["robot_obs"]))
        if dataset_batch.get("window_start") is not None:      # HBM-resident frame store (hulc_amd.utils.frame_store.FrameStore.batch): rgb_obs are the stores
            mb["window_start"] = dataset_batch["window_start"].to(device=device, dtype=torch.int64)
        for k in ("shift_static", "shift_gripper", "pad_static", "pad_gripper"):       # optional RandomShiftsAug draws of the ingest path
            if dataset_batch.get(k) is not None:
                mb[k] = dataset_batch[k].to(device=device) if torch.is_tensor(dataset_batch[k]) else dataset_batch[k]
        if is_lang:
            mb["lang"] = f(dataset_batch["lang"])            # KeyError 'lang' like the reference (hulc.py:440)
            m = dataset_batch["use_for_aux_lang_loss"]
            mb["aux_rows"] = torch.nonzero(m.reshape(-1)).reshape(-1).to("cpu", torch.int32).numpy()
        if "plan_idx" in dataset_batch and dataset_batch["plan_idx"] is not None:
            mb["plan_idx"] = dataset_batch["plan_idx"].to(device=device, dtype=torch.int32)
        if dataset_batch.get("plan_eps") is not None:         # mcil: injected N(0,1) draw (parity tests)
            mb["plan_eps"] = dataset_batch["plan_eps"].to(device=device, dtype=torch.float32)
        return mb

    def training_step(self, batch: Dict[str, Dict], batch_idx: int) -> torch.Tensor:
        """hulc.py:390-537.  Computes the loss AND accumulates its gradients (see module docstring)."""
        eng = self.engine
        if not self._comm_tried and parallel.world_size() > 1:      # first step of a multi-GPU run: the library's own RCCL communicator
            self._comm_tried = True
            parallel.configure_shared_gpu(eng)                      # several ranks on one device: no persistent recurrences
            parallel.setup_comm(eng, os.environ.get("HULC_BUCKET_DTYPE", "fp32"))
        eng.zero_grads()
        nmod = len(batch)
        if self.use_clip_auxiliary_loss and not any("lang" in s for s in batch):
            raise KeyError("aux_lang")                       # SURVEY trap T4: reference raises at hulc.py:531
        kl = act = tot = clip = 0.0
        total_bs = 0
        bs: Dict[str, int] = {}
        scopes = list(batch)
        # the reference's usual batch {"vis": ..., "lang": ...} with equal window counts runs as ONE paired pass (hulc_forward_loss_pair):
        # same losses and gradients as one pass per modality, the latency-bound part of the step only once
        paired = None
        if (self.pair_modalities and nmod == 2 and "lang" not in scopes[0] and "lang" in scopes[1]
                and batch[scopes[0]]["actions"].shape[:2] == batch[scopes[1]]["actions"].shape[:2]
                and 2 * batch[scopes[0]]["actions"].shape[0] <= self._engine_kw["max_batch"]):
            mbs = [self._modality_batch(batch[sc], "lang" in sc, eng.device) for sc in scopes]
            paired = eng.forward_loss_pair(mbs[0], mbs[1], 0.5, self.clip_auxiliary_loss_beta, step=self.global_step)
            if parallel.world_size() > 1:
                parallel.backward_overlapped(eng)
                self._grads_reduced = True
            else:
                eng.backward()
        for imod, (self.modality_scope, dataset_batch) in enumerate(batch.items()):
            is_lang = "lang" in self.modality_scope
            if paired is not None:
                mb, l = mbs[imod], paired[imod]
            else:
                mb = self._modality_batch(dataset_batch, is_lang, eng.device)
                l = eng.forward_loss(mb, is_lang, 1.0 / nmod, self.clip_auxiliary_loss_beta, step=self.global_step)
                if imod == nmod - 1 and parallel.world_size() > 1:
                    parallel.backward_overlapped(eng)          # RCCL all-reduce of the finished 98 % under the encoder backward
                    self._grads_reduced = True
                else:
                    eng.backward()
            b = mb["actions"].shape[0]
            bs[self.modality_scope] = b
            total_bs += b
            if self.kind != "gcbc":
                self.log(f"train/kl_loss_scaled_{self.modality_scope}", l["kl"], on_step=False, on_epoch=True, batch_size=b)
            self.log(f"train/action_loss_{self.modality_scope}", l["action"], on_step=False, on_epoch=True, batch_size=b)
            if self.kind != "gcbc":
                self.log(f"train/total_loss_{self.modality_scope}", l["total_mod"], on_step=False, on_epoch=True, batch_size=b)
            kl += l["kl"]; act += l["action"]; tot += l["total_mod"]
            if is_lang and self.use_clip_auxiliary_loss:
                clip += l["clip"]
        total = tot / nmod
        if self.use_clip_auxiliary_loss:
            total = total + self.clip_auxiliary_loss_beta * clip
            self.log("train/lang_clip_loss", parallel.mean_scalar(self.clip_auxiliary_loss_beta * clip, device=eng.device), on_step=False, on_epoch=True, sync_dist=True)
        if self.kind != "gcbc":
            self.log("train/kl_loss", kl / nmod, on_step=False, on_epoch=True, batch_size=total_bs)
        self.log("train/action_loss", act / nmod, on_step=False, on_epoch=True, batch_size=total_bs)
        self.log("train/total_loss", total, on_step=False, on_epoch=True, batch_size=total_bs)
        self.global_step += 1
        return torch.tensor(total, dtype=torch.float32, device=eng.device)

    def validation_step(self, batch: Dict[str, Dict], batch_idx: int, noise: Optional[Dict[str, Dict]] = None) -> Dict[str, torch.Tensor]:
        """hulc.py:739-841: per modality lmp_val (:301-388) + the logged reductions; eval-mode forward, no gradients.
        `noise` (optional, tests): {scope: {plan_idx_pp, plan_idx_pr, u_mix_pp, u_act_pp, u_mix_pr, u_act_pr}} injected draws."""
        if self.kind == "gcbc":
            return self._validation_step_gcbc(batch, batch_idx, noise)
        eng = self.engine
        output: Dict[str, torch.Tensor] = {}
        val_total_act_loss_pp = 0.0
        nmod = len(batch)
        for self.modality_scope, dataset_batch in batch.items():
            sc = self.modality_scope
            is_lang = "lang" in sc
            mb = self._modality_batch(dataset_batch, is_lang, eng.device)
            mb["step"] = self.global_step * 131 + batch_idx
            r = eng.validate(mb, is_lang, (noise or {}).get(sc))
            val_total_act_loss_pp += r["action_loss_pp"]
            mae_pp, mae_pr = r["mae_pp"], r["mae_pr"]
            self.log(f"val_total_mae/{sc}_total_mae_pr", float(mae_pr.mean()), sync_dist=True)
            self.log(f"val_total_mae/{sc}_total_mae_pp", float(mae_pp.mean()), sync_dist=True)
            self.log(f"val_pos_mae/{sc}_pos_mae_pr", float(mae_pr[:3].mean()), sync_dist=True)
            self.log(f"val_pos_mae/{sc}_pos_mae_pp", float(mae_pp[:3].mean()), sync_dist=True)
            self.log(f"val_orn_mae/{sc}_orn_mae_pr", float(mae_pr[3:6].mean()), sync_dist=True)
            self.log(f"val_orn_mae/{sc}_orn_mae_pp", float(mae_pp[3:6].mean()), sync_dist=True)
            if is_lang and self.use_clip_auxiliary_loss:
                self.log("val/val_pred_clip_loss", r["val_pred_clip_loss"], sync_dist=True)       # hulc.py:804-808
                self.clip_groundtruth(dataset_batch.get("idx"), dataset_batch.get("use_for_aux_lang_loss"))
            self.log(f"val_kl/{sc}_kl_loss", r["kl_loss"], sync_dist=True)
            self.log(f"val_act/{sc}_act_loss_pp", r["action_loss_pp"], sync_dist=True)
            self.log(f"val_act/{sc}_act_loss_pr", r["action_loss_pr"], sync_dist=True)
            self.log(f"val_grip/{sc}_grip_sr_pr", r["gripper_sr_pr"], sync_dist=True)
            self.log(f"val_grip/{sc}_grip_sr_pp", r["gripper_sr_pp"], sync_dist=True)
            self.log("val_act/action_loss_pp", val_total_act_loss_pp / nmod, sync_dist=True)
            if self.kind == "mcil":       # continuous plans (B, 256) as drawn
                output[f"sampled_plan_pp_{sc}"], output[f"sampled_plan_pr_{sc}"] = r["sampled_plan_pp"], r["sampled_plan_pr"]
            else:
                one_hot = lambda idx: torch.nn.functional.one_hot(idx.long(), 32).to(torch.float32).reshape(idx.shape[0], -1)
                output[f"sampled_plan_pp_{sc}"] = one_hot(r["sampled_plan_idx_pp"])     # (B, 1024) like distributions.py:37-41
                output[f"sampled_plan_pr_{sc}"] = one_hot(r["sampled_plan_idx_pr"])
            output[f"idx_{sc}"] = dataset_batch.get("idx")
        return output

    def _validation_step_gcbc(self, batch, batch_idx, noise=None):
        """gcbc.py:183-270: one decoder pass without a plan (loss_and_act), mae / gripper success rate, the reference's metric names."""
        eng = self.engine
        val_total = 0.0
        nmod = len(batch)
        output: Dict[str, torch.Tensor] = {}
        for self.modality_scope, dataset_batch in batch.items():
            sc = self.modality_scope
            is_lang = "lang" in sc
            mb = self._modality_batch(dataset_batch, is_lang, eng.device)
            mb["step"] = self.global_step * 131 + batch_idx
            r = eng.validate(mb, is_lang, (noise or {}).get(sc))
            mae = r["mae_pp"]
            val_total += r["action_loss_pp"]
            self.log(f"val_total_mae/{sc}_total_mae", float(mae.mean()), sync_dist=True)
            self.log(f"val_pos_mae/{sc}_pos_mae", float(mae[:3].mean()), sync_dist=True)
            self.log(f"val_orn_mae/{sc}_orn_mae", float(mae[3:6].mean()), sync_dist=True)
            self.log(f"val_act/{sc}_act_loss", r["action_loss_pp"], sync_dist=True)
            self.log(f"val_grip/{sc}_grip_sr", r["gripper_sr_pp"], sync_dist=True)
            self.log("val_act/action_loss", val_total / nmod, sync_dist=True)
            output[f"idx_{sc}"] = dataset_batch.get("idx")
        return output

    # ---- rollout (hulc.py:843-957) ---------------------------------------------------------------------------------
    def reset(self):
        """Call this at the beginning of a new rollout when doing inference (hulc.py:843-849)."""
        self.plan = None
        self.latent_goal = None
        self.rollout_step_counter = 0
        self.engine.rollout_reset()

    def load_lang_embeddings(self, embeddings_path):
        """hulc.py:871-879: <dataset>/validation/embeddings.npy -> {annotation sentence: 384-d embedding}."""
        import numpy as np
        embeddings = np.load(embeddings_path, allow_pickle=True).item()
        self.lang_embeddings = {v["ann"][0]: v["emb"] for k, v in embeddings.items()}

    @staticmethod
    def _rollout_obs(obs: Dict[str, Any]) -> Dict[str, torch.Tensor]:
        return dict(rgb_static=obs["rgb_obs"]["rgb_static"], rgb_gripper=obs["rgb_obs"]["rgb_gripper"], robot_obs_raw=obs.get("robot_obs_raw"))

    def _plan_value(self, plan_dev) -> torch.Tensor:
        """The engine's plan -> the reference's VALUE: (1, 1024) one-hot of the 32 category indices (distributions.py:37-41), (1, 256) for mcil."""
        if self.kind == "mcil":
            return plan_dev.to(torch.float32).reshape(1, -1)
        return torch.nn.functional.one_hot(plan_dev.long(), 32).to(torch.float32).reshape(1, -1)

    def get_pp_plan_vision(self, obs: dict, goal: dict, noise: Optional[Dict] = None):
        """hulc.py:905-927: obs + goal frame as one 2-frame window -> visual goal encoder -> plan proposal -> a sampled plan; clears the
        decoder's hidden state.  Returns (sampled_plan, latent_goal) as values like the reference: (1,1024) one-hot / (1,256), (1,32)."""
        assert len(obs["rgb_obs"]) == len(goal["rgb_obs"])
        noise = noise or {}
        g = dict(rgb_static=goal["rgb_obs"]["rgb_static"], rgb_gripper=goal["rgb_obs"]["rgb_gripper"])
        plan = self.engine.rollout_plan(self._rollout_obs(obs), g, plan_idx=noise.get("plan") if self.kind == "mcil" else noise.get("plan_idx"))
        return self._plan_value(plan), torch.from_numpy(self.engine.rollout_get_goal()).reshape(1, -1).to(plan.device)

    def get_pp_plan_lang(self, obs: dict, goal, noise: Optional[Dict] = None):
        """hulc.py:929-948: `goal` = the embedded language instruction (384-d) -> language goal encoder -> plan proposal -> a sampled plan;
        clears the decoder's hidden state.  Returns (sampled_plan, latent_goal)."""
        noise = noise or {}
        g = torch.as_tensor(goal, dtype=torch.float32).reshape(-1)
        plan = self.engine.rollout_plan(self._rollout_obs(obs), g, plan_idx=noise.get("plan") if self.kind == "mcil" else noise.get("plan_idx"))
        return self._plan_value(plan), torch.from_numpy(self.engine.rollout_get_goal()).reshape(1, -1).to(plan.device)

    def predict_with_plan(self, obs: Dict[str, Any], latent_goal: torch.Tensor, sampled_plan: torch.Tensor, noise: Optional[Dict] = None) -> torch.Tensor:
        """hulc.py:881-903: encode the current frame, one stateful decoder step with the GIVEN latent goal and plan, sample, tcp -> world.
        The plan and goal are installed as values (hulc_rollout_set_state), so a caller may hold, swap or replay them like with the reference."""
        noise = noise or {}
        p = None
        if self.kind != "gcbc":
            p = sampled_plan.detach().reshape(-1).cpu()
            p = p.numpy().astype(np.float32) if self.kind == "mcil" else p.reshape(32, 32).argmax(-1).numpy().astype(np.int32)
        self.engine.rollout_set_state(p, latent_goal.detach().reshape(-1).cpu().numpy())
        action = self.engine.rollout_act(self._rollout_obs(obs), u_mix=noise.get("u_mix"), u_act=noise.get("u_act"))
        return torch.from_numpy(action).reshape(1, 1, 7)

    def step(self, obs, goal, noise: Optional[Dict] = None):
        """One step of inference (hulc.py:851-869): replan every replan_freq steps from the plan proposal, then act.
        obs: rgb_obs {rgb_static (1,1,3,200,200), rgb_gripper (1,1,3,84,84)}, robot_obs_raw (1,1,15); goal: a sentence (key of
        load_lang_embeddings) or a dict with rgb_obs goal images.  Returns the (1,1,7) world-frame action."""
        noise = noise or {}
        if self.rollout_step_counter % self.replan_freq == 0:
            if isinstance(goal, str):
                if self.lang_embeddings is None:
                    raise RuntimeError("call load_lang_embeddings() before stepping with a language goal (hulc.py:871)")
                embedded_lang = torch.from_numpy(np.asarray(self.lang_embeddings[goal], np.float32)).reshape(-1)
                self.plan, self.latent_goal = self.get_pp_plan_lang(obs, embedded_lang, noise)
            else:
                self.plan, self.latent_goal = self.get_pp_plan_vision(obs, goal, noise)
        action = self.predict_with_plan(obs, self.latent_goal, self.plan, noise)
        self.rollout_step_counter += 1
        return action

    # ---- epoch hooks (hulc.py:959-978; rank-zero log lines like the reference's logger.info / log_rank_0) ----------------------
    @property
    def current_epoch(self) -> int:
        return int(getattr(getattr(self, "trainer", None), "current_epoch", 0))

    def on_train_epoch_start(self) -> None:
        if parallel.rank() == 0:
            logger.info(f"Start training epoch {self.current_epoch}")

    def on_train_epoch_end(self, unused=None) -> None:
        if parallel.rank() == 0:
            logger.info(f"Finished training epoch {self.current_epoch}")

    def on_validation_epoch_end(self) -> None:
        if parallel.rank() == 0:
            logger.info(f"Finished validation epoch {self.current_epoch}")


class GCBC(Hulc):
    """Goal-conditioned behaviour cloning ablation (reference hulc/models/gcbc.py:11-181): no latent plan in the decoder,
    loss = action loss (+ CLIP aux), plan_proposal.* and plan_recognition.fc_state.* never receive a gradient."""

    KIND = "gcbc"

    def reset(self):
        """gcbc.py:281-285: only the latent goal is dropped — the decoder's hidden state is NOT cleared by the reference's GCBC
        (LogisticDecoderRNN.act keeps self.hidden_state, and nothing in gcbc.py calls clear_hidden_state); mirrored here."""
        self.latent_goal = None

    def step(self, obs, goal, noise: Optional[Dict] = None):
        """gcbc.py:287-320: encode the goal once per rollout (language embedding, or obs + goal frame as one 2-frame window through the
        visual goal encoder), then one stateful decoder step without a plan per call."""
        noise = noise or {}
        o = self._rollout_obs(obs)
        if self.latent_goal is None:
            if isinstance(goal, str):
                if self.lang_embeddings is None:
                    raise RuntimeError("call load_lang_embeddings() before stepping with a language goal (hulc.py:871)")
                g = torch.from_numpy(np.asarray(self.lang_embeddings[goal], np.float32)).reshape(-1)
            else:
                g = dict(rgb_static=goal["rgb_obs"]["rgb_static"], rgb_gripper=goal["rgb_obs"]["rgb_gripper"])
            self.engine.rollout_plan(o, g)
            self.latent_goal = True
        action = self.engine.rollout_act(o, u_mix=noise.get("u_mix"), u_act=noise.get("u_act"))
        return torch.from_numpy(action).reshape(1, 1, 7)


def initialize_pretrained_weights(model: Hulc, cfg) -> None:
    """hulc/utils/utils.py:7-16: load `cfg.pretrain_chk` non-strictly; the position-embedding table is trimmed to this model's window
    (load_state_dict does the row trimming), `pretrain_exclude_pr` drops every plan_recognition.* tensor first."""
    ck = torch.load(str(_get(cfg, "pretrain_chk")), map_location="cpu", weights_only=False)
    sd = dict(ck["state_dict"])
    if _get(cfg, "pretrain_exclude_pr", False):
        for key in list(sd.keys()):
            if key.startswith("plan_recognition"):
                del sd[key]
    model.load_state_dict(sd, strict=False)
