"""StepEngine — thin Python owner of one libhulc_hip context + the flat fp32 parameter/gradient/Adam buffers.

torch is plumbing here (device memory, streams, torch.distributed); every FLOP of the step runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import lib as L
from . import spec


class StepEngine:
    def __init__(self, dims: spec.ModelDims, max_batch: int, max_seq: int, dtype: str = "bf16", device: str = "cuda:0",
                 kl_beta: float = 0.01, kl_balancing_mix: float = 0.8, dropout_p: float = 0.1, num_classes: int = 10,
                 gripper_alpha: float = 1.0, log_scale_min: float = -7.0, seed: int = 42, layout_pad: int = 64):
        self.lib = L.load()
        if not torch.cuda.is_available():
            raise RuntimeError("hulc_amd.StepEngine needs a HIP device (torch.cuda.is_available() is False); no CPU fallback")
        self.dims = dims
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.dtype = dtype
        self.layout, self.numel = spec.layout(dims, layout_pad)      # layout_pad 4: the tightly packed table a C caller may bind (tests)
        dev = self.device
        self.flat_params = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        cfg = L.HulcConfig(kind=L.KIND["mcil_gru" if (dims.kind == "mcil" and dims.rnn_type == "gru") else dims.kind], dtype=L.DTYPE[dtype], max_batch=max_batch, max_seq=max_seq,
                           max_window=dims.max_window, use_clip=int(dims.use_clip), kl_beta=kl_beta,
                           kl_balancing_mix=kl_balancing_mix, dropout_p=dropout_p, num_classes=num_classes,
                           gripper_alpha=gripper_alpha, log_scale_min=log_scale_min, seed=seed)
        self.cfg = cfg
        self.ctx = C.c_void_p()
        L.check(self.lib.hulc_ctx_create(C.byref(cfg), C.byref(self.ctx)))
        L.check(self.lib.hulc_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
        names = list(self.layout.keys())
        self._names = (C.c_char_p * len(names))(*[n.encode() for n in names])
        self._offs = (C.c_int64 * len(names))(*[self.layout[n][0] for n in names])
        self._nums = (C.c_int64 * len(names))(*[int(np.prod(self.layout[n][1])) if len(self.layout[n][1]) else 1 for n in names])
        self._bound = False
        self._keep = []
        self.adam_t = 0

    # ---- parameters -------------------------------------------------------------------------------------------
    def views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        for n, (off, shape) in self.layout.items():
            k = int(np.prod(shape)) if len(shape) else 1
            out[n] = flat[off:off + k].view(*shape) if len(shape) else flat[off:off + 1].view(())
        return out

    def bind(self):
        L.check(self.lib.hulc_bind_params(self.ctx, self.flat_params.data_ptr(), self.flat_grads.data_ptr(), self.adam_m.data_ptr(),
                                          self.adam_v.data_ptr(), self.numel, len(self.layout), self._names, self._offs, self._nums))
        self._bound = True

    def load_numpy(self, params: Dict[str, np.ndarray]):
        v = self.views(self.flat_params)
        for n, t in v.items():
            t.copy_(torch.from_numpy(np.asarray(params[n], np.float32)).reshape(t.shape))
        if self._bound:
            self.prepare_weights()
        else:
            self.bind()

    def prepare_weights(self):
        L.check(self.lib.hulc_prepare_weights(self.ctx))

    def zero_grads(self):
        """hulc_zero_grads.  16-bit engines (option lazy_zero_grads, default 1): the large store-first Linear weight gradients keep the previous
        step's values until the next backward writes them / the library reads them (include/hulc_hip.h); call `flush_grads()` before reading
        `flat_grads` yourself between zero_grads() and the end of the next backward."""
        L.check(self.lib.hulc_zero_grads(self.ctx))

    def flush_grads(self):
        """hulc_flush_grads: zero whatever zero_grads() left marked stale (no-op otherwise)."""
        L.check(self.lib.hulc_flush_grads(self.ctx))

    # ---- step pieces ------------------------------------------------------------------------------------------
    @staticmethod
    def _ingest_fields(mb: Dict, ptr) -> Dict:
        """uint8 (B,S,H,W,C) frames select the fused ingest path (include/hulc_hip.h: frames_u8); optional per-frame RandomShiftsAug
        shifts `shift_static` / `shift_gripper` (B*S,2) int32 in [0, 2*pad] with pads `pad_static` (10) / `pad_gripper` (4).
        `window_start` (B,) int64 selects the FRAME STORE form (hulc_batch::window_start): rgb_static / rgb_gripper are then the device-resident
        stores (F,H,W,3) uint8 and window b = store frames [window_start[b], window_start[b] + S) — nothing is materialised per step."""
        rel = {}
        if mb.get("actions_absolute"):        # RelativeActions (transforms.py:32-56) applied on the device
            rel = dict(actions_absolute=1, max_rel_pos=float(mb.get("max_rel_pos", 0.02)), max_rel_orn=float(mb.get("max_rel_orn", 0.05)))
        if mb["rgb_static"].dtype != torch.uint8:
            return rel
        if mb["rgb_gripper"].dtype != torch.uint8 or mb["rgb_static"].shape[-1] != 3 or mb["rgb_gripper"].shape[-1] != 3:
            raise ValueError("uint8 ingest expects both cameras as uint8 (B,S,H,W,3) tensors")
        f = dict(rel, frames_u8=1, pad_static=int(mb.get("pad_static", 10)), pad_gripper=int(mb.get("pad_gripper", 4)))
        for k in ("shift_static", "shift_gripper"):
            if mb.get(k) is not None:
                f[k] = ptr(mb[k].to(torch.int32))
        if mb.get("window_start") is not None:
            if mb["rgb_static"].dim() != 4 or mb["rgb_gripper"].dim() != 4 or mb["rgb_static"].shape[0] != mb["rgb_gripper"].shape[0]:
                raise ValueError("frame store: rgb_static / rgb_gripper must be (F,H,W,3) uint8 stores of the same F next to window_start (B,)")
            ws = mb["window_start"]
            if not ws.is_cuda:
                raise ValueError("frame store: window_start must live on the device")
            f["window_start"] = ptr(ws.to(torch.int64))
            f["store_frames"] = int(mb["rgb_static"].shape[0])
        return f

    def _batch_struct(self, mb: Dict, is_lang: bool, step: int, keep: list):
        B, S = mb["actions"].shape[:2]

        def ptr(t):
            t = t.contiguous()
            keep.append(t)
            return t.data_ptr()

        b = L.HulcBatch(B=B, S=S, is_lang=int(is_lang), rgb_static=ptr(mb["rgb_static"]), rgb_gripper=ptr(mb["rgb_gripper"]),
                        actions=ptr(mb["actions"]), robot_obs=ptr(mb["robot_obs"]), lang=ptr(mb["lang"]) if is_lang else None,
                        plan_idx=ptr(mb["plan_idx"]) if mb.get("plan_idx") is not None else None, aux_rows=None, n_aux=0, step=step,
                        **self._ingest_fields(mb, ptr))
        if mb.get("plan_eps") is not None:          # mcil: injected N(0,1) draw of the reparametrised plan sample
            b.plan_eps = ptr(mb["plan_eps"].to(torch.float32))
        if is_lang and mb.get("aux_rows") is not None and len(mb["aux_rows"]) > 0:
            rows = np.ascontiguousarray(mb["aux_rows"], np.int32)
            keep.append(rows)
            b.aux_rows = rows.ctypes.data
            b.n_aux = len(rows)
        return b

    def forward_loss_pair(self, mb_vis: Dict, mb_lang: Dict, loss_weight: float, clip_weight: float, step: int = 0, sync_losses: bool = True):
        """Both modalities of a step as ONE pass over 2B windows (hulc_forward_loss_pair; needs equal B and S).  Returns the two loss
        dicts (vis, lang), or the (8,) device tensor [vis x4, lang x4] with sync_losses=False.  One backward() follows for the pair."""
        keep = []
        bv = self._batch_struct(mb_vis, False, step, keep)
        bl = self._batch_struct(mb_lang, True, step, keep)
        self._keep = keep
        if sync_losses:
            out = (C.c_float * 8)()
            L.check(self.lib.hulc_forward_loss_pair(self.ctx, C.byref(bv), C.byref(bl), loss_weight, clip_weight, out, 1))
            return (dict(total_mod=out[0], kl=out[1], action=out[2], clip=out[3]), dict(total_mod=out[4], kl=out[5], action=out[6], clip=out[7]))
        if not hasattr(self, "_loss_dev8"):
            self._loss_dev8 = torch.zeros(8, dtype=torch.float32, device=self.device)
        L.check(self.lib.hulc_forward_loss_pair(self.ctx, C.byref(bv), C.byref(bl), loss_weight, clip_weight, self._loss_dev8.data_ptr(), 0))
        self._loss_dev = self._loss_dev8[4:]
        return self._loss_dev8

    def forward_loss(self, mb: Dict, is_lang: bool, loss_weight: float, clip_weight: float, step: int = 0,
                     sync_losses: bool = True):
        """mb: device tensors rgb_static (B,S,3,200,200) f32, rgb_gripper, actions, robot_obs(15), [lang], [plan_idx int32],
        [aux_rows: host int32 numpy]."""
        keep = []
        b = self._batch_struct(mb, is_lang, step, keep)
        self._keep = keep
        if sync_losses:
            out = (C.c_float * 4)()
            L.check(self.lib.hulc_forward_loss(self.ctx, C.byref(b), loss_weight, clip_weight, out, 1))
            return dict(total_mod=out[0], kl=out[1], action=out[2], clip=out[3])
        if not hasattr(self, "_loss_dev"):
            self._loss_dev = torch.zeros(4, dtype=torch.float32, device=self.device)
        L.check(self.lib.hulc_forward_loss(self.ctx, C.byref(b), loss_weight, clip_weight, self._loss_dev.data_ptr(), 0))
        return self._loss_dev

    # ------------------------------------------------------------------ validation / rollout (forward only)
    @staticmethod
    def _dev_or_host_ptr(x, keep, dtype):
        """Noise arrays may be torch (device/host) tensors or numpy arrays; returns a raw pointer and keeps the buffer alive."""
        if x is None:
            return None
        if torch.is_tensor(x):
            t = x.to(dtype=torch.int32 if dtype == np.int32 else torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()
        a = np.ascontiguousarray(x, dtype)
        keep.append(a)
        return a.ctypes.data

    def validate(self, mb: Dict, is_lang: bool, noise: Optional[Dict] = None, want_pred: bool = False) -> Dict:
        """One modality of Hulc.validation_step (hulc.py:770-797 -> lmp_val :301-388), eval mode.  noise (optional, parity):
        plan_idx_pp / plan_idx_pr (B,32) int, u_mix_pp / u_mix_pr (B,S,6,10), u_act_pp / u_act_pr (B,S,6) uniform [0,1) draws."""
        B, S = mb["actions"].shape[:2]
        keep = []

        def ptr(t):
            t = t.contiguous()
            keep.append(t)
            return t.data_ptr()

        b = L.HulcBatch(B=B, S=S, is_lang=int(is_lang), rgb_static=ptr(mb["rgb_static"]), rgb_gripper=ptr(mb["rgb_gripper"]),
                        actions=ptr(mb["actions"]), robot_obs=ptr(mb["robot_obs"]), lang=ptr(mb["lang"]) if is_lang else None,
                        plan_idx=None, aux_rows=None, n_aux=0, step=int(mb.get("step", 0)), **self._ingest_fields(mb, ptr))
        noise = dict(noise or {})
        mcil = self.dims.kind == "mcil"
        if mcil:       # continuous plans travel as (B,256) fp32 in the plan slots (include/hulc_hip.h: hulc_val_noise)
            noise["plan_idx_pp"], noise["plan_idx_pr"] = noise.get("plan_pp"), noise.get("plan_pr")
        nz = L.HulcValNoise(**{k: self._dev_or_host_ptr(noise.get(k), keep, np.int32 if (k.startswith("plan") and not mcil) else np.float32)
                               for k in ("plan_idx_pp", "plan_idx_pr", "u_mix_pp", "u_act_pp", "u_mix_pr", "u_act_pr")})
        if is_lang and mb.get("aux_rows") is not None and len(mb["aux_rows"]) > 0:      # rows of the CLIP validation loss (hulc.py:804-808)
            rows = np.ascontiguousarray(mb["aux_rows"], np.int32)
            keep.append(rows)
            b.aux_rows = rows.ctypes.data
            b.n_aux = len(rows)
        out = (C.c_float * 18)()
        ppp = torch.zeros(B, 256 if mcil else 32, dtype=torch.float32 if mcil else torch.int32, device=self.device)
        ppr = torch.zeros(B, 256 if mcil else 32, dtype=torch.float32 if mcil else torch.int32, device=self.device)
        pred_pp = torch.zeros(B, S, 7, device=self.device) if want_pred else None
        pred_pr = torch.zeros(B, S, 7, device=self.device) if want_pred else None
        L.check(self.lib.hulc_validate(self.ctx, C.byref(b), C.byref(nz), out, ppp.data_ptr(), ppr.data_ptr(),
                                       pred_pp.data_ptr() if want_pred else None, pred_pr.data_ptr() if want_pred else None))
        o = list(out)
        res = dict(action_loss_pp=o[0], action_loss_pr=o[1], kl_loss=o[2], gripper_sr_pp=o[3], gripper_sr_pr=o[4],
                   mae_pp=np.array(o[5:11], np.float32), mae_pr=np.array(o[11:17], np.float32), sampled_plan_idx_pp=ppp, sampled_plan_idx_pr=ppr, val_pred_clip_loss=o[17])
        if mcil:
            res.update(sampled_plan_pp=ppp, sampled_plan_pr=ppr)
        if want_pred:
            res.update(pred_pp=pred_pp, pred_pr=pred_pr)
        return res

    def clip_gt_encode(self, lang_emb, slot: int) -> None:
        """encoded_lang_{train,val} of Hulc.on_validation_epoch_start (hulc.py:967-974) kept on the device: slot 0 = training instructions,
        1 = validation instructions.  lang_emb: (m,384) fp32, numpy or tensor."""
        keep = []
        if isinstance(lang_emb, torch.Tensor):
            lang_emb = lang_emb.to(dtype=torch.float32).contiguous()
        p = self._dev_or_host_ptr(lang_emb, keep, np.float32)
        m = int(lang_emb.shape[0])
        if lang_emb.ndim != 2 or lang_emb.shape[1] != 384:
            raise ValueError(f"lang_emb must be (m,384), got {tuple(lang_emb.shape)}")
        L.check(self.lib.hulc_clip_gt_encode(self.ctx, p, m, int(slot)))

    def clip_gt_scores(self, slot: int) -> np.ndarray:
        """logits_per_image (n,m) of Hulc._clip_groundtruth_loss (hulc.py:1024-1029) for the masked rows of the last lang `validate`."""
        n, m = C.c_int32(0), C.c_int32(0)
        L.check(self.lib.hulc_clip_gt_scores(self.ctx, int(slot), None, 0, C.byref(n), C.byref(m)))       # shape query
        out = np.empty((n.value, m.value), np.float32)
        L.check(self.lib.hulc_clip_gt_scores(self.ctx, int(slot), out.ctypes.data, out.size, C.byref(n), C.byref(m)))
        return out

    def rollout_reset(self):
        L.check(self.lib.hulc_rollout_reset(self.ctx))

    def _obs(self, obs, keep):
        def ptr(t):
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()
        return L.HulcRolloutObs(rgb_static=ptr(obs["rgb_static"]), rgb_gripper=ptr(obs["rgb_gripper"]),
                                robot_obs_raw=ptr(obs["robot_obs_raw"]) if obs.get("robot_obs_raw") is not None else None)

    def rollout_plan(self, obs: Dict, goal, plan_idx=None):
        """goal: dict(rgb_static, rgb_gripper) (1,1,3,H,W) for a visual goal or a (384,) language embedding tensor.  Returns the plan (32,) int32."""
        keep = []
        o = self._obs(obs, keep)
        gs = gg = gl = None
        if isinstance(goal, dict):
            gs = goal["rgb_static"].to(device=self.device, dtype=torch.float32).contiguous(); keep.append(gs)
            gg = goal["rgb_gripper"].to(device=self.device, dtype=torch.float32).contiguous(); keep.append(gg)
        else:
            gl = torch.as_tensor(goal).to(device=self.device, dtype=torch.float32).reshape(-1).contiguous(); keep.append(gl)
        mcil = self.dims.kind == "mcil"      # continuous (256,) fp32 plan instead of (32,) int32 category indices
        out = torch.zeros(256 if mcil else 32, dtype=torch.float32 if mcil else torch.int32, device=self.device)
        L.check(self.lib.hulc_rollout_plan(self.ctx, C.byref(o), gs.data_ptr() if gs is not None else None, gg.data_ptr() if gg is not None else None,
                                           gl.data_ptr() if gl is not None else None, self._dev_or_host_ptr(plan_idx, keep, np.float32 if mcil else np.int32), out.data_ptr()))
        return out

    def rollout_act(self, obs: Dict, u_mix=None, u_act=None) -> np.ndarray:
        keep = []
        o = self._obs(obs, keep)
        out = (C.c_float * 7)()
        L.check(self.lib.hulc_rollout_act(self.ctx, C.byref(o), self._dev_or_host_ptr(u_mix, keep, np.float32),
                                          self._dev_or_host_ptr(u_act, keep, np.float32), out))
        return np.array(list(out), np.float32)

    def rollout_get_goal(self) -> np.ndarray:
        """The (32,) fp32 latent goal the last rollout_plan encoded (second return value of get_pp_plan_vision / get_pp_plan_lang)."""
        out = np.empty(32, np.float32)
        L.check(self.lib.hulc_rollout_get_goal(self.ctx, out.ctypes.data))
        return out

    def rollout_set_state(self, plan, latent_goal) -> None:
        """Install a caller-held plan ((32,) int32 indices; (256,) fp32 for mcil; None for gcbc) and (32,) latent goal for the next rollout_act calls."""
        g = np.ascontiguousarray(np.asarray(latent_goal, np.float32).reshape(-1))
        if g.size != 32:
            raise ValueError("latent_goal must have 32 elements")
        p = None
        if plan is not None:
            mcil = self.dims.kind == "mcil"
            p = np.ascontiguousarray(np.asarray(plan, np.float32 if mcil else np.int32).reshape(-1))
            if p.size != (256 if mcil else 32):
                raise ValueError("plan must have %d elements" % (256 if mcil else 32))
        L.check(self.lib.hulc_rollout_set_state(self.ctx, p.ctypes.data if p is not None else None, g.ctypes.data))

    def set_kl_beta(self, kl_beta: float):
        L.check(self.lib.hulc_set_kl_beta(self.ctx, float(kl_beta)))

    def set_dropout(self, p: float):
        L.check(self.lib.hulc_set_dropout(self.ctx, float(p)))

    def set_option(self, name: str, value: int):
        """Runtime options of the context (include/hulc_hip.h: hulc_set_option), e.g. ``persistent_rnn`` = 0 when processes share one GPU."""
        L.check(self.lib.hulc_set_option(self.ctx, name.encode(), int(value)))

    def timers_enable(self, on: bool = True, only: str = ""):
        L.check(self.lib.hulc_timers_enable(self.ctx, int(on), only.encode()))

    def timers_read(self, reset: bool = True) -> dict:
        import json
        buf = C.create_string_buffer(1 << 16)
        L.check(self.lib.hulc_timers_read(self.ctx, buf, len(buf), int(reset)))
        return json.loads(buf.value.decode())

    def backward(self, part: int = -1):
        """part -1: whole backward; 0: everything but the perceptual encoders; 1: the encoders (must follow 0)."""
        if part < 0:
            L.check(self.lib.hulc_backward(self.ctx))
        else:
            L.check(self.lib.hulc_backward_part(self.ctx, part))

    @property
    def encoder_numel(self) -> int:
        """Flat-buffer elements [0, encoder_numel) belong to perceptual_encoder.*; the rest is final after backward(part=0)."""
        return min(off for n, (off, _) in self.layout.items() if not n.startswith("perceptual_encoder."))

    def adam_step(self, lr=2e-4, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
        self.adam_t += 1
        L.check(self.lib.hulc_adam_step(self.ctx, lr, b1, b2, eps, self.adam_t, grad_scale))

    def optimizer_step(self, kind="adam", lr=2e-4, b1=0.9, b2=0.999, eps=1e-8, weight_decay=0.0, momentum=0.0, dampening=0.0, nesterov=False, grad_scale=1.0):
        """hulc_optimizer_step: torch.optim.Adam / AdamW / SGD (conf/model/optimizer/*.yaml) over the flat buffers; `adam_t` counts the calls."""
        self.adam_t += 1
        o = L.HulcOptim(kind=L.OPTIM[kind], lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=weight_decay, momentum=momentum, dampening=dampening,
                        nesterov=int(bool(nesterov)), step=self.adam_t, grad_scale=grad_scale)
        L.check(self.lib.hulc_optimizer_step(self.ctx, C.byref(o)))

    # ---- data-parallel gradient all-reduce owned by the library (RCCL over xGMI, include/hulc_hip.h: hulc_comm_*) ---------------
    @staticmethod
    def comm_unique_id() -> bytes:
        """rank 0: the 128-byte ncclUniqueId every rank must pass to comm_init (distribute it with any store / torch.distributed)."""
        buf = C.create_string_buffer(128)
        L.check(L.load().hulc_comm_unique_id(buf, 128))
        return buf.raw

    def comm_prepare(self):
        """Phase 1 of comm_init (RCCL resolved, private stream created): can fail on one rank alone, so agree on it before comm_init blocks."""
        L.check(self.lib.hulc_comm_prepare(self.ctx))

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        L.check(self.lib.hulc_comm_init(self.ctx, buf, int(rank), int(world)))
        self.has_comm = True

    def comm_destroy(self):
        if getattr(self, "has_comm", False):
            self.lib.hulc_comm_destroy(self.ctx)
            self.has_comm = False

    def comm_buckets(self):
        """[(lo, hi)] element ranges of the flat gradient buffer in the order hulc_backward_allreduce reduces them."""
        lo, hi = (C.c_int64 * 8)(), (C.c_int64 * 8)()
        n = self.lib.hulc_comm_buckets(self.ctx, lo, hi, 8)
        if n < 0:
            L.check(1)
        return [(int(lo[i]), int(hi[i])) for i in range(n)]

    def comm_size(self):
        """(rank, world) of the live library communicator as RCCL itself reports them (hulc_comm_size)."""
        r, w = C.c_int32(-1), C.c_int32(-1)
        L.check(self.lib.hulc_comm_size(self.ctx, C.byref(r), C.byref(w)))
        return r.value, w.value

    def comm_stats(self) -> Dict:
        n, b = C.c_int64(), C.c_double()
        L.check(self.lib.hulc_comm_stats(self.ctx, C.byref(n), C.byref(b)))
        return dict(collectives=n.value, bytes=b.value)

    def comm_timeline(self) -> Dict:
        """hulc_comm_timeline (after a hulc_backward_allreduce with set_option('comm_timing', 1)): per bucket, when its collective started / ended
        relative to the END of the backward on the engine stream (us; negative = hidden under the backward), and the backward's own duration."""
        out = (C.c_double * 32)()
        bwd = C.c_double()
        n = self.lib.hulc_comm_timeline(self.ctx, out, 8, C.byref(bwd))
        if n < 0:
            L.check(1)
        b = [dict(bucket=int(out[4 * i + 3]), issued_at_us=round(out[4 * i], 1), done_at_us=round(out[4 * i + 1], 1), bytes=int(out[4 * i + 2])) for i in range(n)]
        exposed = max([x["done_at_us"] for x in b] + [0.0])
        return dict(backward_us=round(bwd.value, 1), buckets=b, exposed_after_backward_us=round(exposed, 1))

    def get_option(self, name: str) -> int:
        v = C.c_int64()
        L.check(self.lib.hulc_get_option(self.ctx, name.encode(), C.byref(v)))
        return int(v.value)

    def allreduce_grads(self, bucket_dtype: str = "fp32"):
        """One SUM all-reduce of the whole gradient buffer (after backward()); stream-ordered, no host sync."""
        L.check(self.lib.hulc_allreduce_grads(self.ctx, L.DTYPE[bucket_dtype]))

    def backward_allreduce(self, bucket_dtype: str = "fp32"):
        """backward() of the step's last forward with the bucketed SUM all-reduce overlapped (reverse-forward order)."""
        L.check(self.lib.hulc_backward_allreduce(self.ctx, L.DTYPE[bucket_dtype]))

    # ---- dynamic loss scaling (fp16 mode; torch.cuda.amp.GradScaler semantics, include/hulc_hip.h: hulc_scaler_*) ----------
    def scaler_enable(self, init_scale: float = 65536.0, growth_factor: float = 2.0, backoff_factor: float = 0.5, growth_interval: int = 2000):
        """init_scale <= 0 switches scaling off.  fp16 engines start with GradScaler's defaults, fp32 / bf16 engines with it off."""
        L.check(self.lib.hulc_scaler_enable(self.ctx, float(init_scale), float(growth_factor), float(backoff_factor), int(growth_interval)))

    def scaler_state(self) -> Dict:
        """{"scale", "growth_tracker", "skipped_steps", "last_found_inf", "taken_steps"} — synchronises the stream.  flat_grads holds gradients x scale."""
        sc, tr, sk, fi, tk = C.c_float(), C.c_int32(), C.c_int64(), C.c_int32(), C.c_int64()
        L.check(self.lib.hulc_scaler_get(self.ctx, C.byref(sc), C.byref(tr), C.byref(sk), C.byref(fi), C.byref(tk)))
        return dict(scale=sc.value, growth_tracker=tr.value, skipped_steps=sk.value, last_found_inf=fi.value, taken_steps=tk.value)

    def scaler_load(self, scale: float, growth_tracker: int = 0, taken_steps: int = -1):
        """taken_steps >= 0 also restores the device-side count of optimizer steps taken (Adam's bias corrections in fp16 mode)."""
        L.check(self.lib.hulc_scaler_set(self.ctx, float(scale), int(growth_tracker), int(taken_steps)))

    def get_tensor(self, name: str, n: int) -> np.ndarray:
        out = np.zeros(n, np.float32)
        got = C.c_int64()
        L.check(self.lib.hulc_get_tensor(self.ctx, name.encode(), out.ctypes.data, n, C.byref(got)))
        return out[:got.value]

    def plan_idx(self, B: int) -> np.ndarray:
        out = np.zeros(B * 32, np.int32)
        L.check(self.lib.hulc_get_plan_idx(self.ctx, out.ctypes.data, out.size))
        return out.reshape(B, 32)

    def workspace_bytes(self) -> int:
        return int(self.lib.hulc_workspace_bytes(self.ctx))

    def close(self):
        if getattr(self, "ctx", None):
            self.comm_destroy()
            self.lib.hulc_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
