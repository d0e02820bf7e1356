"""HBM-resident frame store: the MI355X-first replacement for the reference's host-side shared-memory frame cache.

The reference keeps the CALVIN episodes' frames in host shared memory (README.md:85-86: ~20 minutes to fill; dataset/README.md:55-56) and every step
converts uint8 -> fp32, applies the transforms on the CPU and copies a (B,S,3,H,W) fp32 batch to the GPU.  A MI355X holds 288 GB: the uint8 frames of a
training split live ON the device (`FrameStore`), a batch is B window starts, and conv1 gathers the windows by index (include/hulc_hip.h:
hulc_batch::window_start; scale / normalise / RandomShiftsAug run inside conv1's load path as for any uint8 batch).  Per step nothing but the indices,
the (B,S,7) actions and (B,S,15) robot_obs cross PCIe — or nothing at all if those live on the device too (`actions` / `robot_obs` arguments).

    store = FrameStore(rgb_static_u8, rgb_gripper_u8, episode_ends=[...], device="cuda:0")       # (F,200,200,3), (F,84,84,3) uint8, once
    starts = store.sample_starts(B, S, generator)                                                 # (B,) int64: every window inside ONE episode
    batch = {"vis": store.batch(starts, S, actions, robot_obs, shifts=True, generator=g)}         # reference-shaped dict for Hulc.training_step
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch


class FrameStore:
    def __init__(self, rgb_static: torch.Tensor, rgb_gripper: torch.Tensor, episode_ends: Optional[Sequence[int]] = None, device="cuda:0",
                 actions: Optional[torch.Tensor] = None, robot_obs: Optional[torch.Tensor] = None, pad_static: int = 10, pad_gripper: int = 4):
        """rgb_static (F,H,W,3) / rgb_gripper (F,h,w,3): uint8, the frames of all episodes back to back; episode_ends: exclusive end index of every
        episode (ascending, last == F; default: one episode).  actions (F,7) / robot_obs (F,15): optional per-frame fp32 tables kept on the device too."""
        if rgb_static.dtype != torch.uint8 or rgb_gripper.dtype != torch.uint8 or rgb_static.dim() != 4 or rgb_gripper.dim() != 4:
            raise ValueError("FrameStore expects uint8 (F,H,W,3) tensors")
        if rgb_static.shape[0] != rgb_gripper.shape[0] or rgb_static.shape[-1] != 3 or rgb_gripper.shape[-1] != 3:
            raise ValueError("both cameras must hold the same F frames, channels last")
        self.device = torch.device(device)
        self.rgb_static = rgb_static.to(self.device).contiguous()
        self.rgb_gripper = rgb_gripper.to(self.device).contiguous()
        self.F = int(rgb_static.shape[0])
        ends = np.asarray([self.F] if episode_ends is None else list(episode_ends), np.int64)
        if ends.size == 0 or ends[-1] != self.F or np.any(np.diff(np.concatenate([[0], ends])) <= 0):
            raise ValueError("episode_ends must be ascending exclusive end indices whose last entry is F")
        self.episode_ends = ends
        self.episode_starts = np.concatenate([[0], ends[:-1]])
        self.actions = None if actions is None else actions.to(self.device, torch.float32).contiguous()
        self.robot_obs = None if robot_obs is None else robot_obs.to(self.device, torch.float32).contiguous()
        self.pad_static, self.pad_gripper = int(pad_static), int(pad_gripper)

    def bytes(self) -> int:
        return self.rgb_static.numel() + self.rgb_gripper.numel()

    def valid_starts(self, S: int) -> np.ndarray:
        """Every start index whose S frames lie inside ONE episode (host array; the sampling population, hulc's disk datasets index the same way)."""
        parts = [np.arange(a, b - S + 1, dtype=np.int64) for a, b in zip(self.episode_starts, self.episode_ends) if b - a >= S]
        return np.concatenate(parts) if parts else np.zeros((0,), np.int64)

    def sample_starts(self, B: int, S: int, generator: Optional[np.random.Generator] = None) -> torch.Tensor:
        """B window starts drawn uniformly from valid_starts(S) -> (B,) int64 on the device."""
        pop = self.valid_starts(S)
        if pop.size == 0:
            raise ValueError(f"no episode of the store holds {S} frames")
        g = generator or np.random.default_rng()
        return torch.from_numpy(pop[g.integers(0, pop.size, size=B)]).to(self.device)

    def batch(self, starts: torch.Tensor, S: int, actions: Optional[torch.Tensor] = None, robot_obs: Optional[torch.Tensor] = None, shifts: bool = False,
              generator: Optional[torch.Generator] = None, lang: Optional[torch.Tensor] = None, use_for_aux: Optional[torch.Tensor] = None) -> Dict:
        """The reference-shaped batch dict of one modality (hulc/models/hulc.py:395-414) for `Hulc.training_step` / `validation_step`: the stores stand in
        for rgb_obs, `window_start` names the windows.  actions / robot_obs: (B,S,7) / (B,S,15) tensors, or None to gather them from the store's own
        per-frame tables.  shifts=True draws the per-frame RandomShiftsAug offsets (transforms.py:8-29) on the device."""
        starts = starts.to(self.device, torch.int64)
        B = int(starts.shape[0])
        if actions is None or robot_obs is None:
            if self.actions is None or self.robot_obs is None:
                raise ValueError("pass actions / robot_obs or build the store with its per-frame tables")
            idx = (starts[:, None] + torch.arange(S, device=self.device)[None, :]).clamp_(0, self.F - 1)
            actions = self.actions[idx] if actions is None else actions
            robot_obs = self.robot_obs[idx] if robot_obs is None else robot_obs
        d = dict(rgb_obs=dict(rgb_static=self.rgb_static, rgb_gripper=self.rgb_gripper), window_start=starts, depth_obs={},
                 actions=actions.to(self.device, torch.float32), state_info=dict(robot_obs=robot_obs.to(self.device, torch.float32)),
                 robot_obs=torch.zeros(B, S, 8, device=self.device), idx=torch.arange(B, device=self.device),
                 pad_static=self.pad_static, pad_gripper=self.pad_gripper)
        if shifts:
            d["shift_static"] = torch.randint(0, 2 * self.pad_static + 1, (B * S, 2), device=self.device, generator=generator, dtype=torch.int32)
            d["shift_gripper"] = torch.randint(0, 2 * self.pad_gripper + 1, (B * S, 2), device=self.device, generator=generator, dtype=torch.int32)
        if lang is not None:
            d["lang"] = lang.to(self.device, torch.float32)
            d["use_for_aux_lang_loss"] = (torch.ones(B, dtype=torch.bool, device=self.device) if use_for_aux is None else use_for_aux.to(self.device))
        return d

    def materialise(self, starts: torch.Tensor, S: int):
        """The same windows as (B,S,H,W,3) uint8 tensors (tests; the path the store exists to avoid)."""
        idx = (starts.to(self.device, torch.int64)[:, None] + torch.arange(S, device=self.device)[None, :]).reshape(-1)
        B = int(starts.shape[0])
        return (self.rgb_static[idx].reshape(B, S, *self.rgb_static.shape[1:]).contiguous(), self.rgb_gripper[idx].reshape(B, S, *self.rgb_gripper.shape[1:]).contiguous())
