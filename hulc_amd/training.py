"""CLI mirroring `python hulc/training.py group=option key=value ...` (reference hulc/training.py:27-74):

    python -m hulc_amd.training trainer.max_steps=20 datamodule.batch_size=8 [model=gcbc] [trainer.precision=fp32]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m hulc_amd.training ...
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import config, parallel
from .trainer import Trainer, get_last_checkpoint

CONF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "conf")


def newest_run_with_checkpoint(pattern: str):
    """The most recently written run directory matching `pattern` ("{now}" = any date/time pair) that holds a checkpoint, or None."""
    import glob
    runs = [d for d in glob.glob(pattern.replace("{now}", os.path.join("*", "*"))) if os.path.isdir(d) and get_last_checkpoint(d)]
    return max(runs, key=lambda d: os.path.getmtime(get_last_checkpoint(d))) if runs else None


def resolve_log_dir(log_dir: str, resume: bool, rank: int = 0, world: int = 1) -> str:
    """`{now}` in log_dir expands to <date>/<time> — a fresh run directory (the reference's hydra.run.dir), decided on rank 0 and shared.
    `resume=true` instead RE-ENTERS the newest existing run directory of that pattern that holds a checkpoint (a requeued / preempted job
    continues instead of starting over; the reference gets the same effect from SLURM re-running inside the old Hydra directory); with no
    such run it starts a fresh one.  An explicit `log_dir=<existing run>` (no `{now}`) is used as is and resumes whenever it holds a
    checkpoint (training.py:38-46)."""
    if "{now}" not in log_dir:
        return log_dir
    choice = None
    if rank == 0:
        choice = newest_run_with_checkpoint(log_dir) if resume else None
        if choice is None:
            import time as _t
            choice = log_dir.replace("{now}", _t.strftime("%Y-%m-%d/%H-%M-%S"))
        elif resume:
            print(f"[hulc_amd] resume=true: re-entering {choice}", flush=True)
    if world > 1:
        import torch.distributed as dist
        box = [choice]
        dist.broadcast_object_list(box, src=0)
        choice = box[0]
    return choice


def train(overrides=None, conf_dir: str = CONF_DIR):
    cfg = config.compose(conf_dir, "config", overrides or [])
    rank, world, local = parallel.init_from_env()
    torch.manual_seed(cfg.seed)                        # seed_everything(cfg.seed) (training.py:36)
    np.random.seed(cfg.seed)
    device = f"cuda:{local}"
    dm = config.instantiate({k: v for k, v in cfg.datamodule.items() if k not in ("root_data_dir", "action_space", "action_max", "action_min")},
                            device=device, seed=cfg.seed)
    # The reference runs inside a fresh timestamped Hydra directory (conf/config.yaml hydra.run.dir) and resumes from the newest
    # checkpoint found THERE (training.py:38-46), i.e. only when the same run directory is re-entered.  Here: log_dir may contain
    # "{now}" (expanded once on rank 0, then shared); the default conf writes runs/<date>/<time>; `resume=true` re-enters the newest run
    # of that pattern that holds a checkpoint, an explicit `log_dir=<existing run>` re-enters that run (resolve_log_dir).
    log_dir = resolve_log_dir(str(cfg.log_dir), bool(cfg.get("resume", False)), rank, world)
    cfg.log_dir = log_dir
    chk = get_last_checkpoint(log_dir)                   # resume like training.py:38-46 (a fresh {now} directory holds none)
    if "lang" not in cfg.datamodule.get("modalities", ["vis", "lang"]):
        cfg.model.use_clip_auxiliary_loss = False        # SURVEY trap T4
    model = config.instantiate(cfg.model, device=device, max_seq_len=cfg.datamodule.max_window_size)
    if chk is None and cfg.get("pretrain_chk"):          # training.py:45-46 -> hulc/utils/utils.py:7-16
        from .hulc import initialize_pretrained_weights
        initialize_pretrained_weights(model, cfg)
    callbacks = [config.instantiate(c) for c in cfg.callbacks.values() if isinstance(c, dict) and "_target_" in c]
    tr = Trainer(max_epochs=cfg.trainer.max_epochs, max_steps=cfg.trainer.get("max_steps", -1), log_dir=cfg.log_dir, callbacks=callbacks)
    hist = tr.fit(model, dm, ckpt_path=chk)
    return model, hist


if __name__ == "__main__":
    train(sys.argv[1:])
