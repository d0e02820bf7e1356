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


RUN_CONFIG = "run_config.json"      # the resolved configuration a run directory was started with (written by rank 0 next to the checkpoints)


def config_fingerprint(cfg) -> dict:
    """What must agree before a run directory is RE-ENTERED: the model, loss, training and datamodule groups — everything that
    shapes the parameters, the optimizer state and the data — not the run bookkeeping (log_dir, resume, trainer.max_epochs / max_steps,
    callbacks: a requeued job may legitimately extend those)."""
    import json

    def plain(x):
        if isinstance(x, dict):
            return {str(k): plain(v) for k, v in sorted(x.items())}
        if isinstance(x, (list, tuple)):
            return [plain(v) for v in x]
        return x if isinstance(x, (int, float, str, bool)) or x is None else str(x)
    keep = {k: plain(cfg.get(k)) for k in ("model", "loss", "training", "datamodule", "seed")}
    keep["trainer.precision"] = plain(cfg.get("trainer", {}).get("precision"))
    return json.loads(json.dumps(keep))


def run_config_matches(run_dir: str, fp: dict):
    """(True, None) if `run_dir` was started with the same fingerprint (or predates fingerprints), else (False, first differing key path)."""
    import json
    path = os.path.join(run_dir, RUN_CONFIG)
    if not os.path.exists(path):
        return True, None
    try:
        old = json.load(open(path))
    except Exception:
        return False, "unreadable " + RUN_CONFIG

    def diff(a, b, at=""):
        if isinstance(a, dict) and isinstance(b, dict):
            for k in sorted(set(a) | set(b)):
                d = diff(a.get(k), b.get(k), f"{at}.{k}" if at else k)
                if d:
                    return d
            return None
        return None if a == b else f"{at}: {a!r} (run) != {b!r} (now)"
    d = diff(old, fp)
    return d is None, d


def newest_run_with_checkpoint(pattern: str, fp: dict = None):
    """The most recently written run directory matching `pattern` ("{now}" = any date/time pair) that holds a checkpoint AND was started with
    the same configuration fingerprint, or None.  Runs of other configurations are reported and skipped, never continued."""
    import glob
    runs = [d for d in glob.glob(pattern.replace("{now}", os.path.join("*", "*"))) if os.path.isdir(d) and get_last_checkpoint(d)]
    ok = []
    for d in sorted(runs, key=lambda d: os.path.getmtime(get_last_checkpoint(d)), reverse=True):
        same, why = run_config_matches(d, fp) if fp is not None else (True, None)
        if same:
            ok.append(d)
        else:
            print(f"[hulc_amd] resume=true: skipping {d} — started with a different configuration ({why})", flush=True)
    return ok[0] if ok else None


def resolve_log_dir(log_dir: str, resume: bool, rank: int = 0, world: int = 1, fp: dict = None) -> str:
    """`{now}` in log_dir expands to <date>/<time> — a fresh run directory (the reference's hydra.run.dir), decided on rank 0 and shared.
    `resume=true` instead RE-ENTERS the newest existing run directory of that pattern that holds a checkpoint (a requeued / preempted job
    continues instead of starting over; the reference gets the same effect from SLURM re-running inside the old Hydra directory); with no
    such run it starts a fresh one.  An explicit `log_dir=<existing run>` (no `{now}`) is used as is and resumes whenever it holds a
    checkpoint (training.py:38-46)."""
    if "{now}" not in log_dir:
        return log_dir
    choice = None
    if rank == 0:
        choice = newest_run_with_checkpoint(log_dir, fp) if resume else None
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
    fp = config_fingerprint(cfg)
    log_dir = resolve_log_dir(str(cfg.log_dir), bool(cfg.get("resume", False)), rank, world, fp)
    cfg.log_dir = log_dir
    chk = get_last_checkpoint(log_dir)                   # resume like training.py:38-46 (a fresh {now} directory holds none)
    if chk is not None:                                  # an explicit log_dir=<run> of another configuration: refuse instead of continuing it
        same, why = run_config_matches(log_dir, fp)
        if not same:
            raise RuntimeError(f"{log_dir} holds checkpoints of a run with a different configuration ({why}); pass a fresh log_dir or the matching options")
    if rank == 0:
        import json
        os.makedirs(log_dir, exist_ok=True)
        if not os.path.exists(os.path.join(log_dir, RUN_CONFIG)):
            json.dump(fp, open(os.path.join(log_dir, RUN_CONFIG), "w"), indent=1, sort_keys=True)
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
