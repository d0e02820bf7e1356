"""ctypes binding of libhulc_hip.so (include/hulc_hip.h).  No CPU fallback: a missing library raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libhulc_hip.so")

KIND = {"hulc": 0, "gcbc": 1, "mcil": 2, "mcil_gru": 3}
DTYPE = {"fp32": 0, "bf16": 1, "fp16": 2}


class HulcConfig(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("max_batch", C.c_int32), ("max_seq", C.c_int32),
                ("max_window", C.c_int32), ("use_clip", C.c_int32), ("kl_beta", C.c_float),
                ("kl_balancing_mix", C.c_float), ("dropout_p", C.c_float), ("num_classes", C.c_int32),
                ("gripper_alpha", C.c_float), ("log_scale_min", C.c_float), ("seed", C.c_uint64)]


class HulcBatch(C.Structure):
    _fields_ = [("B", C.c_int32), ("S", C.c_int32), ("is_lang", C.c_int32), ("rgb_static", C.c_void_p),
                ("rgb_gripper", C.c_void_p), ("actions", C.c_void_p), ("robot_obs", C.c_void_p), ("lang", C.c_void_p),
                ("plan_idx", C.c_void_p), ("aux_rows", C.c_void_p), ("n_aux", C.c_int32), ("step", C.c_uint64),
                ("frames_u8", C.c_int32), ("pad_static", C.c_int32), ("pad_gripper", C.c_int32), ("shift_static", C.c_void_p),
                ("shift_gripper", C.c_void_p), ("plan_eps", C.c_void_p), ("actions_absolute", C.c_int32), ("max_rel_pos", C.c_float),
                ("max_rel_orn", C.c_float), ("window_start", C.c_void_p), ("store_frames", C.c_int64)]


class HulcValNoise(C.Structure):
    _fields_ = [("plan_idx_pp", C.c_void_p), ("plan_idx_pr", C.c_void_p), ("u_mix_pp", C.c_void_p), ("u_act_pp", C.c_void_p),
                ("u_mix_pr", C.c_void_p), ("u_act_pr", C.c_void_p)]


class HulcRolloutObs(C.Structure):
    _fields_ = [("rgb_static", C.c_void_p), ("rgb_gripper", C.c_void_p), ("robot_obs_raw", C.c_void_p)]


class HulcOptim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("momentum", C.c_float), ("dampening", C.c_float), ("nesterov", C.c_int32), ("step", C.c_int64), ("grad_scale", C.c_float)]


OPTIM = {"adam": 0, "adamw": 1, "sgd": 2}


class HulcSbertConfig(C.Structure):
    _fields_ = [("layers", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32), ("intermediate", C.c_int32), ("vocab", C.c_int32),
                ("max_position", C.c_int32), ("max_sentences", C.c_int32), ("max_tokens", C.c_int32), ("normalize", C.c_int32), ("ln_eps", C.c_float)]


EXPORTS = ["hulc_last_error", "hulc_ctx_create", "hulc_ctx_destroy", "hulc_set_stream", "hulc_workspace_bytes",
           "hulc_bind_params", "hulc_prepare_weights", "hulc_zero_grads", "hulc_flush_grads", "hulc_forward_loss", "hulc_forward_loss_pair", "hulc_backward", "hulc_backward_part",
           "hulc_adam_step", "hulc_optimizer_step", "hulc_comm_unique_id", "hulc_comm_prepare", "hulc_comm_init", "hulc_comm_destroy", "hulc_comm_buckets", "hulc_comm_stats", "hulc_comm_size", "hulc_comm_timeline", "hulc_allreduce_grads", "hulc_backward_allreduce", "hulc_scaler_enable", "hulc_scaler_get", "hulc_scaler_set", "hulc_validate", "hulc_clip_gt_encode", "hulc_clip_gt_scores", "hulc_rollout_reset", "hulc_rollout_plan", "hulc_rollout_act", "hulc_rollout_get_goal", "hulc_rollout_set_state", "hulc_sbert_create", "hulc_sbert_destroy", "hulc_sbert_set_stream", "hulc_sbert_bind", "hulc_sbert_encode", "hulc_set_kl_beta", "hulc_set_dropout", "hulc_set_option", "hulc_get_option", "hulc_timers_enable", "hulc_timers_read", "hulc_get_tensor", "hulc_get_plan_idx", "hulc_k_gemm_nt", "hulc_k_cast", "hulc_k_trread_probe", "hulc_k_conv_wgrad", "hulc_k_conv1_wgrad_u8", "hulc_k_conv1_interior_groups", "hulc_k_conv_tile", "hulc_k_skinny", "hulc_k_attention", "hulc_k_rnn_persist", "hulc_k_rnn_persist_flag_words"]

_lib = None


def load():
    """Load the HIP library; raises RuntimeError (never falls back) when it is absent or unloadable."""
    global _lib
    if _lib is not None:
        return _lib
    # HULC_LIB_PATH: another BUILD of the same library (an older commit's, an experiment's) for same-box A/B runs of whole builds (tools/ab_lib.sh); never a fallback —
    # a path that does not exist is an error like a missing in-tree build
    path = os.environ.get("HULC_LIB_PATH") or LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(f"hulc_amd: {path} not built — run `python -c 'import __graft_entry__ as g; g.build()'`; "
                           "there is no CPU fallback for the product path")
    # torch ships its own libamdhip64; it must be the HIP runtime of the process.  Loading this library first would pull in
    # /opt/rocm's copy and a later `import torch` then sees no device ("hulc_ctx_create: no HIP device visible").
    import torch  # noqa: F401
    lib = C.CDLL(path)
    lib.hulc_last_error.restype = C.c_char_p
    lib.hulc_workspace_bytes.restype = C.c_int64
    lib.hulc_ctx_create.argtypes = [C.POINTER(HulcConfig), C.POINTER(C.c_void_p)]
    lib.hulc_ctx_destroy.argtypes = [C.c_void_p]
    lib.hulc_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.hulc_workspace_bytes.argtypes = [C.c_void_p]
    lib.hulc_bind_params.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32,
                                     C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.hulc_prepare_weights.argtypes = [C.c_void_p]
    lib.hulc_zero_grads.argtypes = [C.c_void_p]
    if os.environ.get("HULC_LIB_PATH") and not hasattr(lib, "hulc_flush_grads"):      # an older build under A/B: the entry points it predates stay unbound
        lib.hulc_zero_grads.argtypes = [C.c_void_p]
    else:
        lib.hulc_flush_grads.argtypes = [C.c_void_p]
    lib.hulc_forward_loss.argtypes = [C.c_void_p, C.POINTER(HulcBatch), C.c_float, C.c_float, C.c_void_p, C.c_int32]
    lib.hulc_forward_loss_pair.argtypes = [C.c_void_p, C.POINTER(HulcBatch), C.POINTER(HulcBatch), C.c_float, C.c_float, C.c_void_p, C.c_int32]
    lib.hulc_backward.argtypes = [C.c_void_p]
    lib.hulc_backward_part.argtypes = [C.c_void_p, C.c_int32]
    lib.hulc_adam_step.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_float]
    lib.hulc_optimizer_step.argtypes = [C.c_void_p, C.POINTER(HulcOptim)]
    lib.hulc_comm_unique_id.argtypes = [C.c_void_p, C.c_int64]
    lib.hulc_comm_prepare.argtypes = [C.c_void_p]
    lib.hulc_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.hulc_comm_destroy.argtypes = [C.c_void_p]
    lib.hulc_comm_buckets.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]
    lib.hulc_comm_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    if hasattr(lib, "hulc_comm_size") or not os.environ.get("HULC_LIB_PATH"):
        lib.hulc_comm_size.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.hulc_comm_timeline.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double)]
    lib.hulc_allreduce_grads.argtypes = [C.c_void_p, C.c_int32]
    lib.hulc_backward_allreduce.argtypes = [C.c_void_p, C.c_int32]
    lib.hulc_scaler_enable.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_int32]
    lib.hulc_scaler_get.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    lib.hulc_scaler_set.argtypes = [C.c_void_p, C.c_float, C.c_int32, C.c_int64]
    lib.hulc_clip_gt_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.hulc_clip_gt_scores.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.hulc_validate.argtypes = [C.c_void_p, C.POINTER(HulcBatch), C.POINTER(HulcValNoise), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hulc_rollout_reset.argtypes = [C.c_void_p]
    lib.hulc_rollout_plan.argtypes = [C.c_void_p, C.POINTER(HulcRolloutObs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hulc_rollout_act.argtypes = [C.c_void_p, C.POINTER(HulcRolloutObs), C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hulc_rollout_get_goal.argtypes = [C.c_void_p, C.c_void_p]
    lib.hulc_rollout_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hulc_sbert_create.argtypes = [C.POINTER(HulcSbertConfig), C.POINTER(C.c_void_p)]
    lib.hulc_sbert_destroy.argtypes = [C.c_void_p]
    lib.hulc_sbert_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.hulc_sbert_bind.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.hulc_sbert_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.hulc_set_kl_beta.argtypes = [C.c_void_p, C.c_float]
    lib.hulc_set_dropout.argtypes = [C.c_void_p, C.c_float]
    lib.hulc_timers_enable.argtypes = [C.c_void_p, C.c_int32, C.c_char_p]
    lib.hulc_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    lib.hulc_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    lib.hulc_k_rnn_persist.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    lib.hulc_k_rnn_persist_flag_words.restype = C.c_int32
    lib.hulc_timers_read.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.c_int32]
    lib.hulc_get_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.hulc_get_plan_idx.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.hulc_k_gemm_nt.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64,
                                   C.c_int64, C.c_int64, C.c_void_p, C.c_int32, C.c_void_p]
    lib.hulc_k_conv_wgrad.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    if hasattr(lib, "hulc_k_conv1_wgrad_u8") or not os.environ.get("HULC_LIB_PATH"):
        lib.hulc_k_conv1_interior_groups.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        lib.hulc_k_conv1_wgrad_u8.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.hulc_k_conv_tile.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p]
    lib.hulc_k_skinny.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.hulc_k_trread_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.hulc_k_cast.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.hulc_k_attention.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64, C.c_void_p]
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise RuntimeError("libhulc_hip: " + load().hulc_last_error().decode("utf-8", "replace"))
