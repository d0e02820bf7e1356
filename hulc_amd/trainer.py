"""Minimal fit loop standing in for `pytorch_lightning.Trainer.fit` (reference hulc/training.py:27-74), plus the callbacks the
hot path touches: KL-beta schedules (hulc/utils/kl_callbacks.py), epoch checkpoints (conf/callbacks/checkpoint/all.yaml,
Lightning-style `{"state_dict", "hyper_parameters", ...}` files, resume via the newest checkpoint as in training.py:38-46),
and a synthetic CALVIN-shaped datamodule (the real CalvinDataModule lives in the absent calvin_agent package).
One process per GPU; gradients are combined by hulc_amd.parallel (RCCL).
"""
from __future__ import annotations

import glob
import math
import os
import time
from typing import Dict, List, Optional

import numpy as np
import torch

from . import parallel


# ---------------------------------------------------------------------------------------------------------------------
class SyntheticDataModule:
    """Yields reference-shaped batch dicts (hulc/models/hulc.py:395-414) of random windows resident on the device."""

    def __init__(self, batch_size: int = 32, max_window_size: int = 32, min_window_size: int = 20, modalities=("vis", "lang"),
                 steps_per_epoch: int = 50, device: str = "cuda:0", seed: int = 0, **_unused):
        self.batch_size, self.S = int(batch_size), int(max_window_size)
        self.modalities = list(modalities)
        self.steps_per_epoch = int(steps_per_epoch)
        self.device = torch.device(device)
        self.seed = seed

    def _modality(self, lang: bool, g: torch.Generator):
        B, S, dev = self.batch_size, self.S, self.device

        def img(h):
            u = torch.randint(0, 256, (B, S, 3, h, h), device=dev, generator=g, dtype=torch.int32).float()
            return (u / 255.0 - 0.5) / 0.5

        act = torch.rand(B, S, 7, device=dev, generator=g) * 2 - 1
        act[..., 6] = torch.where(torch.rand(B, S, device=dev, generator=g) < 0.5, -1.0, 1.0)
        ro = torch.randn(B, S, 15, device=dev, generator=g) * 0.3
        ro[..., 3:6] = torch.rand(B, S, 3, device=dev, generator=g) * 2 - 1
        d = dict(rgb_obs=dict(rgb_static=img(200), rgb_gripper=img(84)), depth_obs={}, robot_obs=torch.zeros(B, S, 8, device=dev),
                 actions=act, state_info=dict(robot_obs=ro), idx=torch.arange(B, device=dev))
        if lang:
            l = torch.randn(B, 384, device=dev, generator=g)
            d["lang"] = l / l.norm(dim=-1, keepdim=True)
            d["use_for_aux_lang_loss"] = torch.ones(B, dtype=torch.bool, device=dev)
        return d

    def train_dataloader(self, rank: int = 0):
        g = torch.Generator(device=self.device)
        for i in range(self.steps_per_epoch):
            g.manual_seed(self.seed + 1000 * rank + i)
            yield {m: self._modality("lang" in m, g) for m in self.modalities}

    val_batches = 1

    def val_dataloader(self, rank: int = 0):
        """Validation batches (validation_step, hulc.py:739-841): same shapes, a fixed seed disjoint from the training ones."""
        g = torch.Generator(device=self.device)
        for i in range(self.val_batches):
            g.manual_seed(self.seed * 7919 + 17 + 1000 * rank + i)
            yield {m: self._modality("lang" in m, g) for m in self.modalities}


# ---------------------------------------------------------------------------------------------------------------------
class KLConstantSchedule:
    """kl_callbacks.py:28-36 — constant beta: nothing to do."""

    def on_train_epoch_start(self, trainer, module):
        pass


class KLLinearSchedule:
    """kl_callbacks.py:62-80."""

    def __init__(self, start_epoch: int, end_epoch: int, max_kl_beta: float):
        self.start, self.end, self.max = start_epoch, end_epoch, max_kl_beta

    def _beta(self, epoch):
        if epoch < self.start:
            return 0.0
        if epoch > self.end:
            return self.max
        return self.max * (epoch - self.start) / max(1, (self.end - self.start))

    def on_train_epoch_start(self, trainer, module):
        module.set_kl_beta(self._beta(trainer.current_epoch))


class KLSigmoidSchedule(KLLinearSchedule):
    """kl_callbacks.py:39-59: sigmoid ramp between start and end epoch."""

    def _beta(self, epoch):
        if epoch < self.start:                 # kl_callbacks.py:41-44: exactly 0 before the ramp, exactly max after it
            return 0.0
        if epoch > self.end:
            return self.max
        x = (epoch - self.start) / max(1, (self.end - self.start))
        return self.max / (1.0 + math.exp(-(12.0 * x - 6.0)))


class ModelCheckpoint:
    def __init__(self, dirpath: str = "saved_models", filename: str = "{epoch}", save_top_k: int = -1, verbose: bool = False, **_):
        self.dirpath, self.filename = dirpath, filename

    def on_train_epoch_end(self, trainer, module):
        if trainer.rank != 0:
            return
        os.makedirs(os.path.join(trainer.log_dir, self.dirpath), exist_ok=True)
        path = os.path.join(trainer.log_dir, self.dirpath, self.filename.format(epoch=f"epoch={trainer.current_epoch}") + ".ckpt")
        save_checkpoint(path, module, trainer.optimizer, trainer.current_epoch, trainer.global_step, getattr(trainer, "lr_scheduler", None))


def save_checkpoint(path, module, optimizer, epoch, global_step, lr_scheduler=None):
    """Lightning-style checkpoint dict; `state_dict` keys are the reference's."""
    torch.save({"epoch": epoch, "global_step": global_step, "state_dict": {k: v.cpu() for k, v in module.state_dict().items()},
                "optimizer_states": [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in optimizer.state_dict().items()}],
                "lr_schedulers": [lr_scheduler.state_dict()] if lr_scheduler is not None and hasattr(lr_scheduler, "state_dict") else [],
                "hyper_parameters": {"kind": module.kind, "use_clip_auxiliary_loss": module.use_clip_auxiliary_loss, "precision": module.precision,
                                     "rnn_type": module.dims.rnn_type, "max_window": module.dims.max_window}}, path)


def get_last_checkpoint(log_dir: str) -> Optional[str]:
    """training.py:38-46 / calvin_agent.utils.utils.get_last_checkpoint: newest *.ckpt under saved_models."""
    files = sorted(glob.glob(os.path.join(log_dir, "saved_models", "*.ckpt")), key=os.path.getmtime)
    return files[-1] if files else None


# ---------------------------------------------------------------------------------------------------------------------
def _unsharded_loader(datamodule):
    """The loader of a datamodule that does not shard by rank itself: train_dataloader() -> an iterable, or a {modality: iterable} dict whose
    per-step batch is {modality: batch} (the reference's combined loader, hulc.py:433-469: the step iterates the modalities)."""
    loaders = datamodule.train_dataloader()
    if isinstance(loaders, dict):
        keys = list(loaders)
        return ({k: b for k, b in zip(keys, bs)} for bs in zip(*[loaders[k] for k in keys]))
    return loaders


def epoch_batches(datamodule, limit_train_batches, world: int):
    """(steps each rank takes per epoch, per_rank) — the ONE place the epoch length is derived (hulc.py:189-212; ADVICE r5: the fit loop and
    Hulc.num_training_steps disagreed for loaders without `steps_per_epoch`, so a warm-up schedule could reach its end before the run did).

    * a datamodule with `steps_per_epoch` states its length PER RANK (every rank draws that many batches of its own): limit_train_batches applies
      to it directly, nothing is divided by the world size (per_rank = True);
    * any other datamodule hands out the UN-SHARDED loader the reference measures (hulc.py:197-199): size = the longest loader's len(), an int
      limit replaces it (:201-202), a float scales it (:203-205), and each of the `world` ranks takes size // world of those batches (:209-211) —
      Trainer.fit strides such a loader by rank (batch i goes to rank i % world), which is what Lightning's DistributedSampler does to it;
    * a loader without len() and no int limit: (inf, False) — the epoch ends when the loader does.
    """
    ltb = limit_train_batches
    ltb_int = isinstance(ltb, int) and not isinstance(ltb, bool) and ltb != 0
    if hasattr(datamodule, "steps_per_epoch"):
        size, per_rank = int(datamodule.steps_per_epoch), True
    else:
        per_rank, size = False, None
        try:
            loaders = datamodule.train_dataloader()
            size = max(len(loaders[k]) for k in loaders) if isinstance(loaders, dict) else len(loaders)
        except TypeError:
            size = None
    if ltb_int:
        size = ltb
    elif size is None:
        return float("inf"), per_rank
    elif isinstance(ltb, float):
        size = int(size * ltb)
    return (size if per_rank else size // max(1, int(world))), per_rank


class Trainer:
    def __init__(self, max_epochs: int = 1, max_steps: int = -1, log_dir: str = "./runs", callbacks: Optional[List] = None,
                 log_every: int = 10, limit_val_batches: Optional[int] = None, check_val_every_n_epoch: int = 1, limit_train_batches=None,
                 accumulate_grad_batches: int = 1, **_unused):
        self.max_epochs, self.max_steps, self.log_dir = int(max_epochs), int(max_steps), log_dir
        self.limit_train_batches, self.accumulate_grad_batches = limit_train_batches, max(1, int(accumulate_grad_batches or 1))
        if self.accumulate_grad_batches != 1:
            # training_step runs forward AND backward and the optimizer steps on every batch; a schedule inferred for fewer optimizer steps than
            # are taken would reach lr 0 early.  No reference configuration sets it (conf/trainer/*.yaml): rejected, not ignored
            raise NotImplementedError("accumulate_grad_batches > 1 is not supported by hulc_amd.trainer.Trainer (the reference's confs never set it)")
        self.datamodule = None
        self.limit_val_batches, self.check_val_every_n_epoch = limit_val_batches, max(1, int(check_val_every_n_epoch or 1))
        self.val_history: List[Dict[str, float]] = []
        self.callbacks = callbacks or []
        self.log_every = log_every
        self.current_epoch = 0
        self.global_step = 0
        self.rank, self.world, self.local = 0, 1, 0
        self.optimizer = None
        self.history: List[Dict[str, float]] = []
        self.epoch_history: List[Dict[str, float]] = []

    def _train_batches(self, datamodule) -> float:
        """Optimizer steps THIS rank takes per training epoch = epoch_batches(...)[0]: ONE arithmetic for the fit loop and Hulc.num_training_steps."""
        return epoch_batches(datamodule, self.limit_train_batches, self.world)[0]

    def validate(self, module, datamodule) -> Dict[str, float]:
        """Lightning's validation loop for this module: eval mode, validation_step over the val batches, mean of every `val*` metric."""
        if not hasattr(datamodule, "val_dataloader"):
            return {}
        module.eval()
        if hasattr(module, "on_validation_epoch_start"):
            module.on_validation_epoch_start()
        sums: Dict[str, float] = {}
        counts: Dict[str, int] = {}
        n = 0
        for bi, batch in enumerate(datamodule.val_dataloader(self.rank)):
            if self.limit_val_batches is not None and bi >= self.limit_val_batches:
                break
            for k in [k for k in module.logged if k.startswith("lang_gt/")]:
                del module.logged[k]                 # logged only by batches with masked lang rows (hulc.py:988-989): mean over those batches
            module.validation_step(batch, bi)
            for k, v in module.logged.items():
                if k.startswith("val") or k.startswith("lang_gt/"):
                    sums[k] = sums.get(k, 0.0) + float(v)
                    counts[k] = counts.get(k, 0) + 1
            n += 1
        module.train()
        if hasattr(module, "on_validation_epoch_end"):
            module.on_validation_epoch_end()
        out = parallel.mean_metrics(sums, counts, device=module.device)
        self.val_history.append(out)
        return out

    def fit(self, module, datamodule, ckpt_path: Optional[str] = None):
        self.rank, self.world, self.local = parallel.init_from_env()
        self.datamodule = datamodule
        module.trainer = self                      # Lightning attaches the trainer before configure_optimizers (Hulc.num_training_steps reads it)
        oc = module.configure_optimizers()
        self.optimizer, sched = oc["optimizer"], oc["lr_scheduler"]["scheduler"]
        self.lr_scheduler = sched
        if ckpt_path:
            ck = torch.load(ckpt_path, map_location="cpu", weights_only=False)
            hp = ck.get("hyper_parameters") or {}
            if hp.get("kind", module.kind) != module.kind or hp.get("rnn_type", module.dims.rnn_type) != module.dims.rnn_type:
                raise RuntimeError(f"checkpoint {ckpt_path} holds a {hp.get('kind')!r} ({hp.get('rnn_type')}) model, this run builds {module.kind!r} "
                                   f"({module.dims.rnn_type}): refusing to resume (pass a fresh log_dir or the matching model= option)")
            if self.rank == 0:
                print(f"[hulc_amd] resuming from {ckpt_path} (epoch {ck.get('epoch')}, global_step {ck.get('global_step')})", flush=True)
            module.load_state_dict(ck["state_dict"])
            self.optimizer.load_state_dict({k: (v.to(module.device) if torch.is_tensor(v) else v) for k, v in ck["optimizer_states"][0].items()})
            if ck.get("lr_schedulers") and hasattr(sched, "load_state_dict"):
                sched.load_state_dict(ck["lr_schedulers"][0])
            self.current_epoch, self.global_step = ck["epoch"] + 1, ck["global_step"]
            module.global_step = self.global_step
        module.on_fit_start()
        module.train()
        t0 = time.time()
        done = False
        while self.current_epoch < self.max_epochs and not done:
            for cb in self.callbacks:
                if hasattr(cb, "on_train_epoch_start"):
                    cb.on_train_epoch_start(self, module)
            if hasattr(module, "on_train_epoch_start"):
                module.on_train_epoch_start()
            n_batches, per_rank = epoch_batches(datamodule, self.limit_train_batches, self.world)
            taken = 0
            for bi, batch in enumerate(datamodule.train_dataloader(self.rank) if per_rank else _unsharded_loader(datamodule)):
                if not per_rank and bi % self.world != self.rank:         # an un-sharded loader: batch i belongs to rank i % world (DistributedSampler's stride)
                    continue
                if taken >= n_batches:                                     # Lightning's limit_train_batches (int: batches, float: fraction of the epoch)
                    break
                taken += 1
                loss = module.training_step(batch, self.global_step)      # forward + loss + backward (grads accumulated)
                self.optimizer.step()                                      # RCCL all-reduce (mean) + fused Adam
                sched.step()
                self.global_step += 1
                if self.rank == 0 and self.global_step % self.log_every == 0:
                    rec = dict(step=self.global_step, epoch=self.current_epoch, loss=float(loss), time=time.time() - t0, **module.logged)
                    self.history.append(rec)
                    print(f"[hulc_amd] epoch {self.current_epoch} step {self.global_step} loss {float(loss):.4f} "
                          f"action {module.logged.get('train/action_loss', float('nan')):.4f}", flush=True)
                if 0 < self.max_steps <= self.global_step:
                    done = True
                    break
            if (self.current_epoch + 1) % self.check_val_every_n_epoch == 0 and self.limit_val_batches != 0 and not done:
                self.validate(module, datamodule)                             # Lightning: validation loop at the end of the epoch
            # Lightning reduces `on_step=False, on_epoch=True` metrics at the end of the epoch: batch-size weighted means (hulc.py:470-536)
            if hasattr(module, "epoch_metrics"):
                em = module.epoch_metrics(reset=True)
                self.epoch_history.append(dict(epoch=self.current_epoch, **em))
                if self.rank == 0 and em:
                    print(f"[hulc_amd] epoch {self.current_epoch} means: " + ", ".join(f"{k} {v:.4f}" for k, v in sorted(em.items()) if k.startswith("train/")), flush=True)
            if hasattr(module, "on_train_epoch_end"):
                module.on_train_epoch_end()
            for cb in self.callbacks:
                if hasattr(cb, "on_train_epoch_end"):
                    cb.on_train_epoch_end(self, module)
            self.current_epoch += 1
        torch.cuda.synchronize()
        return self.history
