"""Synthetic CALVIN-shaped batches (SURVEY.md §8d input spec), numpy, portable (counter-based RNG).

Shapes follow the batch contract documented at reference hulc/models/hulc.py:395-414: float NCHW frames
already scaled to [-1,1] (conf/datamodule/transforms/rand_shift.yaml:2-10), relative actions in [-1,1] with
a +-1 gripper channel, raw 15-d robot_obs (euler angles in [3:6]).
"""
from __future__ import annotations

import numpy as np

from . import portable_rng as prng


def make_modality(tag: str, B: int, S: int, seed: int = 0, lang: bool = False, edge_frac: float = 0.05,
                  n_cat: int = 32, n_cls: int = 32, aux_mask: str = "all"):
    u8 = np.floor(prng.uniform01(f"{tag}.rgb_static", (B, S, 3, 200, 200), seed) * 256.0)
    rs = ((u8 / 255.0 - 0.5) / 0.5).astype(np.float32)
    u8 = np.floor(prng.uniform01(f"{tag}.rgb_gripper", (B, S, 3, 84, 84), seed) * 256.0)
    rg = ((u8 / 255.0 - 0.5) / 0.5).astype(np.float32)
    act = prng.uniform(f"{tag}.actions", (B, S, 7), -1.0, 1.0, seed)
    e = prng.uniform01(f"{tag}.actions.edge", (B, S, 7), seed)
    act = np.where(e < edge_frac / 2, -1.0, np.where(e > 1 - edge_frac / 2, 1.0, act)).astype(np.float32)
    act[..., 6] = np.where(prng.uniform01(f"{tag}.grip", (B, S), seed) < 0.5, -1.0, 1.0)
    ro = prng.normal(f"{tag}.robot_obs", (B, S, 15), 0.3, seed)
    ro[..., 3:6] = prng.uniform(f"{tag}.euler", (B, S, 3), -1.0, 1.0, seed)
    mb = dict(rgb_static=rs, rgb_gripper=rg, actions=act.astype(np.float32), robot_obs=ro.astype(np.float32),
              plan_idx=prng.randint(f"{tag}.plan_idx", (B, n_cat), n_cls, seed))
    if lang:
        l = prng.normal(f"{tag}.lang", (B, 384), 1.0, seed)
        mb["lang"] = (l / np.linalg.norm(l, axis=-1, keepdims=True)).astype(np.float32)
        if aux_mask == "all":
            mb["use_for_aux"] = np.ones((B,), bool)
        elif aux_mask == "none":
            mb["use_for_aux"] = np.zeros((B,), bool)
        else:
            mb["use_for_aux"] = prng.uniform01(f"{tag}.aux", (B,), seed) < 0.6
    return mb


def make_batch(B_vis: int, B_lang: int, S: int, seed: int = 0, **kw):
    batch = {}
    if B_vis > 0:
        batch["vis"] = make_modality("vis", B_vis, S, seed, lang=False, **kw)
    if B_lang > 0:
        batch["lang"] = make_modality("lang", B_lang, S, seed, lang=True, **kw)
    return batch
