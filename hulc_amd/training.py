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


def train(overrides=None, conf_dir: str = CONF_DIR):
    cfg = config.compose(conf_dir, "config", overrides or [])
    rank, world, local = parallel.init_from_env()
    torch.manual_seed(cfg.seed)                        # seed_everything(cfg.seed) (training.py:36)
    np.random.seed(cfg.seed)
    device = f"cuda:{local}"
    dm = config.instantiate({k: v for k, v in cfg.datamodule.items() if k not in ("root_data_dir", "action_space", "action_max", "action_min")},
                            device=device, seed=cfg.seed)
    chk = get_last_checkpoint(cfg.log_dir)               # resume like training.py:38-46
    if "lang" not in cfg.datamodule.get("modalities", ["vis", "lang"]):
        cfg.model.use_clip_auxiliary_loss = False        # SURVEY trap T4
    model = config.instantiate(cfg.model, device=device, max_seq_len=cfg.datamodule.max_window_size)
    callbacks = [config.instantiate(c) for c in cfg.callbacks.values() if isinstance(c, dict) and "_target_" in c]
    tr = Trainer(max_epochs=cfg.trainer.max_epochs, max_steps=cfg.trainer.get("max_steps", -1), log_dir=cfg.log_dir, callbacks=callbacks)
    hist = tr.fit(model, dm, ckpt_path=chk)
    return model, hist


if __name__ == "__main__":
    train(sys.argv[1:])
