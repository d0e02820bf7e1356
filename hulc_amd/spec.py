"""Parameter table of the HULC / GCBC training step: names, shapes, init — the state-dict contract.

Names and shapes are exactly those of the reference's ``state_dict`` (reference modules
hulc/models/hulc.py:86-121 and the sub-network constructors they instantiate), so reference checkpoints
load by name.  The order is the flat-buffer order used by ``hulc_amd`` (one contiguous fp32 parameter
buffer + one contiguous gradient buffer; every ``nn.Parameter`` is a view into it).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Tuple

import numpy as np

from .utils import portable_rng as prng


@dataclass(frozen=True)
class ModelDims:
    kind: str = "hulc"            # "hulc" | "gcbc" | "mcil" (conf/model/mcil.yaml: BiRNN plan encoder, continuous latent; use_clip must be False)
    max_window: int = 32          # rows of plan_recognition.position_embeddings
    use_clip: bool = True         # use_clip_auxiliary_loss (creates proj_vis_lang + logit_scale)
    emb: int = 128                # perceptual latent size (64 static + 64 gripper)
    vf: int = 64
    goal: int = 32
    lang: int = 384
    hidden: int = 2048
    n_cat: int = 32               # category_size
    n_cls: int = 32               # class_size
    heads: int = 8
    ff: int = 2048
    fc_hidden: int = 4096
    n_mix: int = 10
    act_dims: int = 6             # out_features - 1 (discrete gripper)
    num_classes: int = 10

    cont_plan: int = 256          # mcil: distribution.plan_features (conf/model/distribution/continuous.yaml)
    rnn_type: str = "rnn"         # mcil: plan_recognition.rnn_type — "rnn" = nn.RNN (tanh; birnn.yaml default), "gru" = nn.GRU (BASELINE config 4)

    @property
    def plan(self) -> int:
        return self.cont_plan if self.kind == "mcil" else self.n_cat * self.n_cls

    @property
    def state(self) -> int:       # width of the fc_state outputs: logits (discrete) or mean | var (continuous, distributions.py:55-59)
        return 2 * self.cont_plan if self.kind == "mcil" else self.n_cat * self.n_cls

    @property
    def dec_plan(self) -> int:    # plan features seen by the decoder (gcbc.py:44 sets 0)
        return 0 if self.kind == "gcbc" else self.plan

    @property
    def dec_emb(self) -> int:     # perceptual features seen by the decoder: perceptual_emb_slice [64,128] (hulc_default.yaml:15); all 128 for mcil
        return self.emb if self.kind == "mcil" else 64

    @property
    def dec_in(self) -> int:      # logistic_decoder_rnn.py:56-59
        return self.dec_plan + self.dec_emb + self.goal

    @property
    def mix_dims(self) -> int:    # dimensions modelled by the logistic mixture: 6 + discrete gripper head, or all 7 (mcil_default.yaml)
        return 7 if self.kind == "mcil" else self.act_dims

    @property
    def mix_classes(self) -> int:
        return 256 if self.kind == "mcil" else self.num_classes


# (name, shape, init) ; init = ("u", fan_in) uniform(+-1/sqrt(fan_in)) | ("n",) N(0,1) | ("xav", fi, fo)
# | ("one",) | ("zero",) | ("const", v)
def param_table(d: ModelDims) -> List[Tuple[str, Tuple[int, ...], tuple]]:
    t: List[Tuple[str, Tuple[int, ...], tuple]] = []

    def lin(name, out_f, in_f, fan=None):
        fan = in_f if fan is None else fan
        t.append((name + ".weight", (out_f, in_f), ("u", fan)))
        t.append((name + ".bias", (out_f,), ("u", fan)))

    def conv(name, co, ci, k):
        fan = ci * k * k
        t.append((name + ".weight", (co, ci, k, k), ("u", fan)))
        t.append((name + ".bias", (co,), ("u", fan)))

    def ln(name, n):
        t.append((name + ".weight", (n,), ("one",)))
        t.append((name + ".bias", (n,), ("zero",)))

    pe = "perceptual_encoder.rgb_static_encoder."
    conv(pe + "conv_model.0", 32, 3, 8)
    conv(pe + "conv_model.2", 64, 32, 4)
    conv(pe + "conv_model.4", 64, 64, 3)
    lin(pe + "fc1.0", 512, 128)
    lin(pe + "fc2", d.vf, 512)
    ln(pe + "ln", d.vf)
    pg = "perceptual_encoder.rgb_gripper_encoder."
    conv(pg + "conv_model.0", 32, 3, 8)
    conv(pg + "conv_model.2", 64, 32, 4)
    conv(pg + "conv_model.4", 64, 64, 3)
    lin(pg + "conv_model.7", 128, 3136)
    lin(pg + "fc1.0", 512, 128)
    lin(pg + "fc2", d.vf, 512)
    ln(pg + "ln", d.vf)

    H = d.hidden
    lin("plan_proposal.fc_model.0", H, d.emb + d.goal)
    lin("plan_proposal.fc_model.2", H, H)
    lin("plan_proposal.fc_model.4", H, H)
    lin("plan_proposal.fc_model.6", H, H)
    lin("plan_proposal.fc_state.0", d.state, H)

    pr = "plan_recognition."
    if d.kind == "mcil":          # PlanRecognitionBiRNNNetwork (plan_recognition_net.py:12-42): nn.RNN(tanh), 2 layers, bidirectional
        ng = 3 if d.rnn_type == "gru" else 1      # nn.GRU stacks the reset | update | new gate blocks along dim 0
        for l, kin in ((0, d.emb), (1, 2 * H)):
            for sfx in ("", "_reverse"):
                t.append((f"{pr}birnn_model.weight_ih_l{l}{sfx}", (ng * H, kin), ("u", H)))
                t.append((f"{pr}birnn_model.weight_hh_l{l}{sfx}", (ng * H, H), ("u", H)))
                t.append((f"{pr}birnn_model.bias_ih_l{l}{sfx}", (ng * H,), ("u", H)))
                t.append((f"{pr}birnn_model.bias_hh_l{l}{sfx}", (ng * H,), ("u", H)))
        lin(pr + "fc_state.0", d.state, 2 * H)
    else:
        t.append((pr + "position_embeddings.weight", (d.max_window, d.emb), ("n",)))
    for l in range(0 if d.kind == "mcil" else 2):
        L = f"{pr}transformer_encoder.layers.{l}."
        t.append((L + "self_attn.in_proj_weight", (3 * d.emb, d.emb), ("xav", d.emb, 3 * d.emb)))
        t.append((L + "self_attn.in_proj_bias", (3 * d.emb,), ("zero",)))
        t.append((L + "self_attn.out_proj.weight", (d.emb, d.emb), ("u", d.emb)))
        t.append((L + "self_attn.out_proj.bias", (d.emb,), ("zero",)))
        lin(L + "linear1", d.ff, d.emb)
        lin(L + "linear2", d.emb, d.ff)
        ln(L + "norm1", d.emb)
        ln(L + "norm2", d.emb)
    if d.kind != "mcil":
        lin(pr + "fc", d.fc_hidden, d.emb)
        lin(pr + "fc_state.0", d.plan, d.fc_hidden)

    lin("visual_goal.mlp.0", H, d.emb)
    lin("visual_goal.mlp.2", H, H)
    lin("visual_goal.mlp.4", d.goal, H)
    ln("visual_goal.ln", d.goal)
    lin("language_goal.mlp.1", H, d.lang)
    lin("language_goal.mlp.3", H, H)
    lin("language_goal.mlp.5", d.goal, H)
    ln("language_goal.ln", d.goal)

    ad = "action_decoder."
    for l, kin in ((0, d.dec_in), (1, H)):
        t.append((f"{ad}rnn.weight_ih_l{l}", (H, kin), ("u", H)))
        t.append((f"{ad}rnn.weight_hh_l{l}", (H, H), ("u", H)))
        t.append((f"{ad}rnn.bias_ih_l{l}", (H,), ("u", H)))
        t.append((f"{ad}rnn.bias_hh_l{l}", (H,), ("u", H)))
    no = d.mix_dims * d.n_mix
    lin(ad + "mean_fc", no, H)
    lin(ad + "log_scale_fc", no, H)
    lin(ad + "prob_fc", no, H)
    if d.kind != "mcil":
        lin(ad + "gripper_fc", 2, H)

    if d.use_clip:
        lin("proj_vis_lang.mlp_im.0", 128, d.fc_hidden)
        lin("proj_vis_lang.mlp_im.2", d.goal, 128)
        lin("proj_vis_lang.mlp_lang.0", 128, d.goal)
        lin("proj_vis_lang.mlp_lang.2", d.goal, 128)
        t.append(("logit_scale", (), ("const", math.log(1.0 / 0.07))))
    return t


def layout(d: ModelDims, pad: int = 64):
    """name -> (offset, shape); every tensor starts on a `pad`-element boundary: 64 (256 B, the default) or 4 — the tightest table
    hulc_bind_params accepts (offsets are multiples of 4), what a C caller with its own packed buffers would bind."""
    assert pad >= 4 and pad % 4 == 0
    off = 0
    out = {}
    for name, shape, _ in param_table(d):
        n = int(np.prod(shape)) if len(shape) else 1
        out[name] = (off, shape)
        off += (n + pad - 1) // pad * pad
    return out, off


def n_params(d: ModelDims) -> int:
    return sum(int(np.prod(s)) if len(s) else 1 for _, s, _ in param_table(d))


def init_param(name: str, shape, init: tuple, seed: int = 0, ln_jitter: bool = False) -> np.ndarray:
    """Portable default initialisation (distribution shapes of torch's defaults).

    ``ln_jitter`` perturbs LayerNorm affine params / zero-biases so parity tests exercise them.
    """
    kind = init[0]
    if kind == "u":
        b = 1.0 / math.sqrt(init[1])
        return prng.uniform(name, shape, -b, b, seed)
    if kind == "n":
        return prng.normal(name, shape, 1.0, seed)
    if kind == "xav":
        b = math.sqrt(6.0 / (init[1] + init[2]))
        return prng.uniform(name, shape, -b, b, seed)
    if kind == "one":
        return (1.0 + (prng.uniform(name, shape, -0.2, 0.2, seed) if ln_jitter else 0.0)) * np.ones(shape, np.float32)
    if kind == "zero":
        return prng.uniform(name, shape, -0.1, 0.1, seed) if ln_jitter else np.zeros(shape, np.float32)
    if kind == "const":
        return np.full(shape, init[1], np.float32)
    raise ValueError(kind)


def init_all(d: ModelDims, seed: int = 0, ln_jitter: bool = False):
    return {n: init_param(n, s, i, seed, ln_jitter).astype(np.float32) for n, s, i in param_table(d)}
