"""CPU ORACLE (test infrastructure, NOT the product): numpy restatement of one HULC / GCBC training step.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product path (``hulc_amd``) never does and fails loudly if its HIP library is missing.

It restates, in plain numpy fp32 (float64 only inside reductions where noted), the algorithm of the
reference's ``Hulc.training_step`` + autograd backward + ``torch.optim.Adam`` step.  Every function cites
the reference file:line it follows (paths relative to /root/reference).  Layouts are the reference's
(NCHW images, (out,in) Linear weights), parameters are addressed by their reference ``state_dict`` names.

Parity pin: ``tools/gen_golden.py`` imports the unmodified reference in the build container and writes
losses / intermediate activations / per-parameter gradients / post-Adam parameters for several small
configurations into ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this oracle against
them (fp32, <=2e-5 relative).  Stochastic draws (the categorical plan sample) are injected as inputs
(``plan_idx``); dropout is off (``model.eval()``-style) in all fixtures — see DESIGN.md "parity".
"""
from __future__ import annotations

import math

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------------
# ---- rounding-aware mode (tests of the bf16 / fp16 engines).  The half-precision engines keep fp32 master weights and fp32
# accumulation but every GEMM / convolution OPERAND — weights, stored activations, stored gradients — is a 16-bit value.  With
# set_operand_rounding("bf16" | "fp16") this oracle rounds exactly those operands (round-to-nearest-even, what v_cvt / the engine's pack2h do)
# and the handful of 16-bit tensors that non-GEMM kernels read (q(): conv3's output under the spatial softmax, qkv, the attention output
# gradient, the recurrent pre-activations / states / state gradients, emb, goal), so its ReLU masks and pre-activations follow the engine's to
# fp32 summation-order noise instead of differing by 2^-9 relative — per-tensor gradient gates tighten from 0.2 to 2e-2.  Default (None) is the
# plain fp32 restatement every golden-vector test uses; nothing in the product imports this file.
_QMODE = None
_GSCALE = 1.0


def set_operand_rounding(mode, grad_scale=1.0):
    """None (fp32, default) | "bf16" | "fp16".  grad_scale (a power of two): the fp16 engine's loss scale — its 16-bit GRADIENT tensors hold
    gradient x scale (GradScaler semantics), so their rounding / underflow happens at that magnitude (qg)."""
    global _QMODE, _GSCALE
    if mode in ("fp32", ""):
        mode = None
    assert mode in (None, "bf16", "fp16"), mode
    _QMODE, _GSCALE = mode, float(grad_scale)


def q(x):
    """x as the 16-bit engines store it (values rounded to the mode's format, returned as float32); identity in fp32 mode."""
    if _QMODE is None:
        return x
    x = np.ascontiguousarray(x, F32)
    if _QMODE == "fp16":
        return x.astype(np.float16).astype(F32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)                 # round to nearest even on the 16 dropped bits
    out = ((u + r) & np.uint32(0xFFFF0000)).view(F32)
    return np.where(np.isfinite(x), out, x).astype(F32)


def qg(x):
    """a stored 16-bit GRADIENT tensor: like q(), at the loss-scaled magnitude in fp16 mode."""
    if _QMODE != "fp16" or _GSCALE == 1.0:
        return q(x)
    return (q(np.asarray(x, F32) * F32(_GSCALE)) / F32(_GSCALE)).astype(F32)


def mm(a, b):
    """a @ b with both operands as the engine holds them (16-bit in rounding-aware mode), fp32 accumulate."""
    return q(a) @ q(b)


def relu(x):
    return np.maximum(x, 0)


def linear(x, w, b=None):
    y = mm(x, w.T)
    if b is not None:
        y = y + b
    return y.astype(F32)


def linear_bwd(x, w, dy):
    """returns dx, dw, db for y = x w^T + b ; x (M,K), w (N,K), dy (M,N)"""
    dy = qg(dy)
    return (dy @ q(w)).astype(F32), (dy.T @ q(x)).astype(F32), dy.sum(0).astype(F32)


def _im2col(x, kh, kw, s):
    n, c, h, w = x.shape
    oh, ow = (h - kh) // s + 1, (w - kw) // s + 1
    st = x.strides
    v = np.lib.stride_tricks.as_strided(
        x, (n, oh, ow, c, kh, kw), (st[0], st[2] * s, st[3] * s, st[1], st[2], st[3]), writeable=False)
    return v.reshape(n * oh * ow, c * kh * kw), oh, ow


def conv2d(x, w, b, s):
    """nn.Conv2d, no padding (vision_network.py:38-45, vision_network_gripper.py:12-17)."""
    o, c, kh, kw = w.shape
    col, oh, ow = _im2col(np.ascontiguousarray(q(x)), kh, kw, s)
    y = col @ q(w).reshape(o, -1).T + b
    return np.ascontiguousarray(y.reshape(x.shape[0], oh, ow, o).transpose(0, 3, 1, 2)).astype(F32)


def conv2d_bwd(x, w, dy, s, need_dx=True):
    o, c, kh, kw = w.shape
    n = x.shape[0]
    col, oh, ow = _im2col(np.ascontiguousarray(q(x)), kh, kw, s)
    dy2 = qg(dy.transpose(0, 2, 3, 1).reshape(-1, o))
    dw = (dy2.T @ col).reshape(w.shape).astype(F32)
    db = dy2.sum(0).astype(F32)
    dx = None
    if need_dx:
        dcol = (dy2 @ q(w).reshape(o, -1)).reshape(n, oh, ow, c, kh, kw)
        dx = np.zeros_like(x, dtype=F32)
        for i in range(kh):
            for j in range(kw):
                dx[:, :, i:i + s * oh:s, j:j + s * ow:s] += dcol[:, :, :, :, i, j].transpose(0, 3, 1, 2)
    return dx, dw, db


def layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xh = (x - mu) * rstd
    return (xh * g + b).astype(F32), (xh.astype(F32), rstd.astype(F32))


def layer_norm_bwd(dy, g, cache):
    xh, rstd = cache
    dxh = dy * g
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xh * (dxh * xh).mean(-1, keepdims=True))
    red = tuple(range(dy.ndim - 1))
    return dx.astype(F32), (dy * xh).sum(red).astype(F32), dy.sum(red).astype(F32)


def softmax(x, axis=-1):
    m = x.max(axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis, keepdims=True)


def log_softmax(x, axis=-1):
    m = x.max(axis, keepdims=True)
    z = x - m
    return z - np.log(np.exp(z).sum(axis, keepdims=True))


def softplus(x):
    return np.logaddexp(0.0, x)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# ----------------------------------------------------------------------------------------------------
# perceptual encoders  (perceptual_encoders/vision_network.py, vision_network_gripper.py, concat_encoders.py)
# ----------------------------------------------------------------------------------------------------
def spatial_softmax(f, temp=1.0):
    """vision_network.py:100-108 ; f (N,C,H,W) -> (N,2C) interleaved [ex_c, ey_c]; x follows the ROW index."""
    n, c, h, w = f.shape
    p = softmax(f.reshape(n, c, h * w) / temp, -1)
    lin_h = np.linspace(-1.0, 1.0, h, dtype=F32)
    lin_w = np.linspace(-1.0, 1.0, w, dtype=F32)
    xm = np.repeat(lin_h, w)            # x_map[i*w+j] = lin[i]   (meshgrid indexing="ij", :88-92)
    ym = np.tile(lin_w, h)              # y_map[i*w+j] = lin[j]
    ex = (p * xm).sum(-1)
    ey = (p * ym).sum(-1)
    out = np.stack([ex, ey], -1).reshape(n, 2 * c).astype(F32)
    return out, (p.astype(F32), xm, ym, ex, ey)


def spatial_softmax_bwd(dout, cache, shape, temp=1.0):
    p, xm, ym, ex, ey = cache
    n, c, h, w = shape
    d = dout.reshape(n, c, 2)
    dex, dey = d[..., 0:1], d[..., 1:2]
    ds = p * (dex * (xm - ex[..., None]) + dey * (ym - ey[..., None])) / temp
    return ds.reshape(n, c, h, w).astype(F32)


def static_encoder_fwd(P, pre, x):
    """vision_network.py:55-65"""
    c = {}
    c["x"] = x
    c["a1"] = relu(conv2d(x, P[pre + "conv_model.0.weight"], P[pre + "conv_model.0.bias"], 4))
    c["a2"] = relu(conv2d(c["a1"], P[pre + "conv_model.2.weight"], P[pre + "conv_model.2.bias"], 2))
    c["a3"] = q(relu(conv2d(c["a2"], P[pre + "conv_model.4.weight"], P[pre + "conv_model.4.bias"], 1)))   # q: the spatial softmax reads the stored 16-bit map
    c["ss"], c["ss_cache"] = spatial_softmax(c["a3"])
    c["f1"] = relu(linear(c["ss"], P[pre + "fc1.0.weight"], P[pre + "fc1.0.bias"]))
    c["f2"] = linear(c["f1"], P[pre + "fc2.weight"], P[pre + "fc2.bias"])
    c["out"], c["ln_cache"] = layer_norm(c["f2"], P[pre + "ln.weight"], P[pre + "ln.bias"])
    c["out"] = q(c["out"])                    # perceptual_emb is a stored 16-bit tensor (read by the position add, the goal / plan MLPs, the decoder)
    return c["out"], c


def static_encoder_bwd(P, G, pre, c, dout):
    d, dg, db = layer_norm_bwd(dout, P[pre + "ln.weight"], c["ln_cache"])
    _acc(G, pre + "ln.weight", dg)
    _acc(G, pre + "ln.bias", db)
    _cond_lin(G, pre + "fc2", c["f1"], d)
    d, dw, db = linear_bwd(c["f1"], P[pre + "fc2.weight"], d)
    _acc(G, pre + "fc2.weight", dw)
    _acc(G, pre + "fc2.bias", db)
    d = d * (c["f1"] > 0)
    _cond_lin(G, pre + "fc1.0", c["ss"], d)
    d, dw, db = linear_bwd(c["ss"], P[pre + "fc1.0.weight"], d)
    _acc(G, pre + "fc1.0.weight", dw)
    _acc(G, pre + "fc1.0.bias", db)
    d = spatial_softmax_bwd(d, c["ss_cache"], c["a3"].shape) * (c["a3"] > 0)
    _conv_stack_bwd(P, G, pre, c, d)


CONDITION_SUMS = False      # tests only: also accumulate G["abs/" + name] = sum |dY|^T |X| (weights) / sum |dY| (biases) of the six conv layers


def _cond(G, name, x, w, dy, s):
    """The CONDITION of a convolution's weight / bias gradient as a sum (tests/test_gpu_fullsize.py): every element of dW is a sum of N x OH x OW
    products dY x, and kappa = || sum |dY| |x| || / || sum dY x || says by how much a relative perturbation eps of the summands (one 16-bit rounding of
    dY = 2^-9) can move the result: ||delta dW|| <= kappa eps ||dW||.  Encoder gradients cancel heavily (kappa 10 - 40), which is what bounds the
    full-size parity gates from below — not the kernels' own arithmetic."""
    if not CONDITION_SUMS:
        return
    _, dwa, dba = conv2d_bwd(np.abs(x), w, np.abs(dy), s, need_dx=False)
    _acc(G, "abs/" + name + ".weight", dwa)
    _acc(G, "abs/" + name + ".bias", dba)


def _cond_lin(G, name, x, dy):
    """the same for a Linear layer's gradients: sum_rows |dY|^T |x| and sum_rows |dY| (the encoder tails: 2048 frame rows per step)"""
    if not CONDITION_SUMS:
        return
    dy = np.abs(qg(dy))
    _acc(G, "abs/" + name + ".weight", (dy.T @ np.abs(q(x))).astype(F32))
    _acc(G, "abs/" + name + ".bias", dy.sum(0).astype(F32))


def _conv_stack_bwd(P, G, pre, c, d3):
    d, dw, db = conv2d_bwd(c["a2"], P[pre + "conv_model.4.weight"], d3, 1)
    _acc(G, pre + "conv_model.4.weight", dw)
    _acc(G, pre + "conv_model.4.bias", db)
    _cond(G, pre + "conv_model.4", c["a2"], P[pre + "conv_model.4.weight"], d3, 1)
    d = d * (c["a2"] > 0)
    d2 = d
    d, dw, db = conv2d_bwd(c["a1"], P[pre + "conv_model.2.weight"], d, 2)
    _acc(G, pre + "conv_model.2.weight", dw)
    _acc(G, pre + "conv_model.2.bias", db)
    _cond(G, pre + "conv_model.2", c["a1"], P[pre + "conv_model.2.weight"], d2, 2)
    d = d * (c["a1"] > 0)
    _, dw, db = conv2d_bwd(c["x"], P[pre + "conv_model.0.weight"], d, 4, need_dx=False)
    _acc(G, pre + "conv_model.0.weight", dw)
    _acc(G, pre + "conv_model.0.bias", db)
    _cond(G, pre + "conv_model.0", c["x"], P[pre + "conv_model.0.weight"], d, 4)


def gripper_encoder_fwd(P, pre, x):
    """vision_network_gripper.py:10-20,49-57 ; Flatten is (C,H,W) order."""
    c = {}
    c["x"] = x
    c["a1"] = relu(conv2d(x, P[pre + "conv_model.0.weight"], P[pre + "conv_model.0.bias"], 4))
    c["a2"] = relu(conv2d(c["a1"], P[pre + "conv_model.2.weight"], P[pre + "conv_model.2.bias"], 2))
    c["a3"] = relu(conv2d(c["a2"], P[pre + "conv_model.4.weight"], P[pre + "conv_model.4.bias"], 1))
    c["flat"] = c["a3"].reshape(x.shape[0], -1)
    c["g0"] = relu(linear(c["flat"], P[pre + "conv_model.7.weight"], P[pre + "conv_model.7.bias"]))
    c["f1"] = relu(linear(c["g0"], P[pre + "fc1.0.weight"], P[pre + "fc1.0.bias"]))
    c["f2"] = linear(c["f1"], P[pre + "fc2.weight"], P[pre + "fc2.bias"])
    c["out"], c["ln_cache"] = layer_norm(c["f2"], P[pre + "ln.weight"], P[pre + "ln.bias"])
    c["out"] = q(c["out"])
    return c["out"], c


def gripper_encoder_bwd(P, G, pre, c, dout):
    d, dg, db = layer_norm_bwd(dout, P[pre + "ln.weight"], c["ln_cache"])
    _acc(G, pre + "ln.weight", dg)
    _acc(G, pre + "ln.bias", db)
    _cond_lin(G, pre + "fc2", c["f1"], d)
    d, dw, db = linear_bwd(c["f1"], P[pre + "fc2.weight"], d)
    _acc(G, pre + "fc2.weight", dw)
    _acc(G, pre + "fc2.bias", db)
    d = d * (c["f1"] > 0)
    _cond_lin(G, pre + "fc1.0", c["g0"], d)
    d, dw, db = linear_bwd(c["g0"], P[pre + "fc1.0.weight"], d)
    _acc(G, pre + "fc1.0.weight", dw)
    _acc(G, pre + "fc1.0.bias", db)
    d = d * (c["g0"] > 0)
    _cond_lin(G, pre + "conv_model.7", c["flat"], d)
    d, dw, db = linear_bwd(c["flat"], P[pre + "conv_model.7.weight"], d)
    _acc(G, pre + "conv_model.7.weight", dw)
    _acc(G, pre + "conv_model.7.bias", db)
    d = d.reshape(c["a3"].shape) * (c["a3"] > 0)
    _conv_stack_bwd(P, G, pre, c, d)


def _acc(G, name, g):
    g = np.asarray(g, F32)
    if name in G:
        G[name] = G[name] + g
    else:
        G[name] = g.copy()


# ----------------------------------------------------------------------------------------------------
# MLP helpers (goal encoders goal_encoders.py:31-36/64-69, plan proposal plan_proposal_net.py:42-47)
# ----------------------------------------------------------------------------------------------------
def mlp_fwd(P, names, x, last_relu):
    acts = [x]
    for i, n in enumerate(names):
        y = linear(acts[-1], P[n + ".weight"], P[n + ".bias"])
        if i < len(names) - 1 or last_relu:
            y = relu(y)
        acts.append(y)
    return acts[-1], acts


def mlp_bwd(P, G, names, acts, d, last_relu):
    for i in reversed(range(len(names))):
        n = names[i]
        if i < len(names) - 1 or last_relu:
            d = d * (acts[i + 1] > 0)
        d, dw, db = linear_bwd(acts[i], P[n + ".weight"], d)
        _acc(G, n + ".weight", dw)
        _acc(G, n + ".bias", db)
    return d


# ----------------------------------------------------------------------------------------------------
# plan recognition transformer (plan_recognition_net.py:94-117 ; nn.TransformerEncoderLayer post-LN, relu)
# ----------------------------------------------------------------------------------------------------
# ---- TRAIN mode with the ENGINE's dropout masks (tests; VERDICT r5 weak #2: the timed path runs with dropout 0.1 and was only checked statistically) ----------
# The engine's masks are counter-based: element `idx` of site `k` is kept iff hash_uniform(site_seed(k), idx) >= p (csrc/common.h hash_u32 / hash_uniform,
# csrc/engine.h site_seed: splitmix64 of (context seed, optimizer-step index, modality, site)).  Restated here in numpy so that the oracle can run the SAME
# train-mode step — forward and backward — as the library: sites 0 (emb + pos), 1 + 4 l (attention weights), 2 + 4 l (after out_proj), 3 + 4 l (after the FFN
# activation), 4 + 4 l (after linear2) of layer l; the element index is the row-major offset of the tensor the mask multiplies ((B S, 128), (B, 8, S, S), (B S, 2048)).
TRAIN_DROPOUT = None      # None (eval mode) | (p, seed, step[, first_window]): modality_fwd / modality_bwd then run in train mode with the engine's masks;
                          # first_window: the batch index of this call's window 0 inside the engine's batch (chunked evaluation, tests/oracle_pool.py)


def _mix64(z):
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def engine_site_seed(seed, step, is_lang, site):
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * np.uint64(step * 64 + (32 if is_lang else 0) + site + 1)
        return _mix64(z)


def engine_keep_mask(site_seed, shape, p, first=0):
    """keep[idx] = hash_uniform(site_seed, idx) >= p for idx = first + row-major offsets of `shape` (fp32 arithmetic as on the device)."""
    with np.errstate(over="ignore"):
        idx = np.arange(int(np.prod(shape)), dtype=np.uint64) + np.uint64(first)
        z = _mix64(np.uint64(site_seed) + (idx + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15))
    h = (z >> np.uint64(32)).astype(np.uint32)
    u = ((h >> np.uint32(8)).astype(F32) + F32(0.5)) * F32(1.0 / 16777216.0)
    return (u >= F32(p)).reshape(shape)


def plan_recognition_fwd(P, emb, heads=8, drop=None):
    """drop = (p, numpy Generator): TRAIN mode — inverted dropout with the oracle's OWN masks at the five kinds of site the reference has
    (plan_recognition_net.py:111 on emb + pos; nn.TransformerEncoderLayer(dropout=p): attention weights, after out_proj, after the FFN
    activation, after linear2); or drop = (p, callable(site, shape) -> keep mask): the masks of somebody else (the engine's, TRAIN_DROPOUT).
    The multipliers keep / (1 - p) are kept in the cache (plan_recognition_bwd applies them)."""
    pr = "plan_recognition."
    B, S, D = emb.shape
    hd = D // heads
    c = {"layers": []}
    mult = {}

    def _drop(t, site=None):
        if drop is None:
            return t
        pdrop, src = drop
        keep = src(site, t.shape) if callable(src) else (src.random(t.shape) >= pdrop)
        mult[site] = (keep / F32(1.0 - pdrop)).astype(F32)
        return (t * mult[site]).astype(F32)
    x = _drop((emb + P[pr + "position_embeddings.weight"][:S][None]).astype(F32), 0)      # :101-105, :111
    for l in range(2):
        L = f"{pr}transformer_encoder.layers.{l}."
        lc = {"x_in": x}
        qkv = q(linear(x.reshape(B * S, D), P[L + "self_attn.in_proj_weight"], P[L + "self_attn.in_proj_bias"]))   # q: the attention kernel reads the stored qkv
        qkv = qkv.reshape(B, S, 3, heads, hd).transpose(2, 0, 3, 1, 4)          # (3,B,H,S,hd)
        qh, k, v = qkv[0], qkv[1], qkv[2]
        sc = (qh * F32(1.0 / math.sqrt(hd))) @ k.transpose(0, 1, 3, 2)
        psm = softmax(sc, -1).astype(F32)
        pa = _drop(psm, 1 + 4 * l)
        ao = (pa @ v).transpose(0, 2, 1, 3).reshape(B * S, D).astype(F32)
        lc.update(q=qh, k=k, v=v, pa=pa, psm=psm, ao=ao)
        sa = _drop(linear(ao, P[L + "self_attn.out_proj.weight"], P[L + "self_attn.out_proj.bias"]), 2 + 4 * l).reshape(B, S, D)
        x1, lc["ln1"] = layer_norm(x + sa, P[L + "norm1.weight"], P[L + "norm1.bias"])
        lc["x1"] = x1
        h = _drop(relu(linear(x1.reshape(B * S, D), P[L + "linear1.weight"], P[L + "linear1.bias"])), 3 + 4 * l)
        lc["h"] = h
        ff = _drop(linear(h, P[L + "linear2.weight"], P[L + "linear2.bias"]), 4 + 4 * l).reshape(B, S, D)
        x, lc["ln2"] = layer_norm(x1 + ff, P[L + "norm2.weight"], P[L + "norm2.bias"])
        lc["x_out"] = x
        c["layers"].append(lc)
    c["x_final"] = x
    c["drop_mult"] = mult
    if _QMODE is None:
        y = linear(x.reshape(B * S, D), P[pr + "fc.weight"], P[pr + "fc.bias"]).reshape(B, S, -1)   # :113
        seq_feat = y.mean(1).astype(F32)                                                          # :114
    else:           # the engine hoists the mean over S before the affine fc (exact in real arithmetic): its GEMM operand is the 16-bit mean
        seq_feat = linear(x.mean(1).astype(F32), P[pr + "fc.weight"], P[pr + "fc.bias"])
    logits = linear(seq_feat, P[pr + "fc_state.0.weight"], P[pr + "fc_state.0.bias"])          # :115
    c["seq_feat"] = seq_feat
    return logits, seq_feat, c


def plan_recognition_bwd(P, G, c, dlogits, dseq_feat, heads=8, fc_state_used=True):
    pr = "plan_recognition."
    x = c["x_final"]
    B, S, D = x.shape
    hd = D // heads
    dsf = np.zeros_like(c["seq_feat"]) if dseq_feat is None else dseq_feat.astype(F32).copy()
    if fc_state_used and dlogits is not None:
        d, dw, db = linear_bwd(c["seq_feat"], P[pr + "fc_state.0.weight"], dlogits)
        _acc(G, pr + "fc_state.0.weight", dw)
        _acc(G, pr + "fc_state.0.bias", db)
        dsf = dsf + d
    if _QMODE is None:
        dy = np.repeat((dsf / S)[:, None, :], S, 1).reshape(B * S, -1).astype(F32)
        d, dw, db = linear_bwd(x.reshape(B * S, D), P[pr + "fc.weight"], dy)
        dx = d.reshape(B, S, D)
    else:           # mirror of the hoisted mean: fc's gradients from the (B, 128) mean, d x = d mean / S on every token
        d, dw, db = linear_bwd(x.mean(1).astype(F32), P[pr + "fc.weight"], dsf)
        dx = np.repeat((d / S)[:, None, :], S, 1).astype(F32)
    _acc(G, pr + "fc.weight", dw)
    _acc(G, pr + "fc.bias", db)
    mult = c.get("drop_mult") or {}
    dm = lambda t, site: t if site not in mult else (t * mult[site].reshape(t.shape)).astype(F32)      # a site's dropout multiplier applied to its gradient
    for l in (1, 0):
        L = f"{pr}transformer_encoder.layers.{l}."
        lc = c["layers"][l]
        dr, dg, db = layer_norm_bwd(dx, P[L + "norm2.weight"], lc["ln2"])
        _acc(G, L + "norm2.weight", dg)
        _acc(G, L + "norm2.bias", db)
        dh, dw, db = linear_bwd(lc["h"], P[L + "linear2.weight"], dm(dr.reshape(B * S, D), 4 + 4 * l))
        _acc(G, L + "linear2.weight", dw)
        _acc(G, L + "linear2.bias", db)
        dh = dm(dh, 3 + 4 * l) * (lc["h"] > 0)
        dx1, dw, db = linear_bwd(lc["x1"].reshape(B * S, D), P[L + "linear1.weight"], dh)
        _acc(G, L + "linear1.weight", dw)
        _acc(G, L + "linear1.bias", db)
        dx1 = dx1.reshape(B, S, D) + dr
        dr1, dg, db = layer_norm_bwd(dx1, P[L + "norm1.weight"], lc["ln1"])
        _acc(G, L + "norm1.weight", dg)
        _acc(G, L + "norm1.bias", db)
        dao, dw, db = linear_bwd(lc["ao"], P[L + "self_attn.out_proj.weight"], dm(dr1.reshape(B * S, D), 2 + 4 * l))
        _acc(G, L + "self_attn.out_proj.weight", dw)
        _acc(G, L + "self_attn.out_proj.bias", db)
        dao = qg(dao).reshape(B, S, heads, hd).transpose(0, 2, 1, 3)               # (B,H,S,hd); q: the attention backward reads the stored 16-bit gradient
        pa, psm, qh, k, v = lc["pa"], lc["psm"], lc["q"], lc["k"], lc["v"]      # pa: the (dropped) weights that multiplied v; psm: the softmax itself
        dv = pa.transpose(0, 1, 3, 2) @ dao
        dpa = dm(dao @ v.transpose(0, 1, 3, 2), 1 + 4 * l)
        dsc = psm * (dpa - (dpa * psm).sum(-1, keepdims=True))
        scale = F32(1.0 / math.sqrt(hd))
        dq = (dsc @ k) * scale
        dk = dsc.transpose(0, 1, 3, 2) @ (qh * scale)
        dqkv = np.stack([dq, dk, dv], 0).transpose(1, 3, 0, 2, 4).reshape(B * S, 3 * D).astype(F32)
        dxin, dw, db = linear_bwd(lc["x_in"].reshape(B * S, D), P[L + "self_attn.in_proj_weight"], dqkv)
        _acc(G, L + "self_attn.in_proj_weight", dw)
        _acc(G, L + "self_attn.in_proj_bias", db)
        dx = dxin.reshape(B, S, D) + dr1
    dx = dm(dx, 0)
    dpos = np.zeros_like(P[pr + "position_embeddings.weight"])
    dpos[:S] = dx.sum(0)
    _acc(G, pr + "position_embeddings.weight", dpos)
    return dx.astype(F32)      # grad w.r.t. perceptual_emb


# ----------------------------------------------------------------------------------------------------
# action decoder (decoders/logistic_decoder_rnn.py, utils/rnn.py, utils/gripper_control.py)
# ----------------------------------------------------------------------------------------------------
def euler_xyz_to_matrix(e):
    """pytorch3d_transforms.py:162-218, convention "XYZ": R = Rx(a) Ry(b) Rz(c)."""
    a, b, c = e[..., 0], e[..., 1], e[..., 2]
    ca, sa, cb, sb, cc, sc = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(c), np.sin(c)
    R = np.empty(e.shape[:-1] + (3, 3), F32)
    R[..., 0, 0] = cb * cc
    R[..., 0, 1] = -cb * sc
    R[..., 0, 2] = sb
    R[..., 1, 0] = ca * sc + sa * sb * cc
    R[..., 1, 1] = ca * cc - sa * sb * sc
    R[..., 1, 2] = -sa * cb
    R[..., 2, 0] = sa * sc - ca * sb * cc
    R[..., 2, 1] = sa * cc + ca * sb * sc
    R[..., 2, 2] = ca * cb
    return R


def world_to_tcp_frame(action, robot_obs):
    """gripper_control.py:16-36 (fp32); inverse(R) == R^T."""
    act = action.astype(F32)
    e = robot_obs[..., 3:6].astype(F32)
    R = euler_xyz_to_matrix(e)
    Rt = np.swapaxes(R, -1, -2)
    pos = (Rt @ act[..., :3, None])[..., 0]
    Rn = euler_xyz_to_matrix(e + act[..., 3:6] * F32(0.01))
    M = np.swapaxes(Rn, -1, -2) @ R
    # matrix_to_euler_angles "XYZ" (pytorch3d_transforms.py:221-303)
    o0 = np.arctan2(-M[..., 1, 2], M[..., 2, 2])
    o1 = np.arcsin(M[..., 0, 2])
    o2 = np.arctan2(-M[..., 0, 1], M[..., 0, 0])
    orn = np.stack([o0, o1, o2], -1).astype(F32)
    orn = np.where(orn < -np.pi, orn + 2 * np.pi, orn)
    orn = np.where(orn > np.pi, orn - 2 * np.pi, orn)
    orn = (orn * 100).astype(F32)
    return np.concatenate([pos, orn, act[..., -1:]], -1).astype(F32)


def rnn_fwd(P, pre, x, h0=None):
    """utils/rnn.py:5-14 : 2-layer ReLU nn.RNN, batch_first; h0 = 0 unless given as (2, B, H) (stateful rollout,
    logistic_decoder_rnn.py:107-111).  The cache carries the final hidden state as c["h_n"]."""
    B, S, _ = x.shape
    c = {"x": x}
    inp = x
    hn = []
    for l in range(2):
        wih, whh = P[f"{pre}weight_ih_l{l}"], P[f"{pre}weight_hh_l{l}"]
        b = P[f"{pre}bias_ih_l{l}"] + P[f"{pre}bias_hh_l{l}"]
        zx = q((mm(inp.reshape(B * S, -1), wih.T) + b).reshape(B, S, -1))     # q: the hoisted input projection is stored (16-bit) and added in the step's epilogue
        H = np.zeros((B, S, whh.shape[0]), F32)
        h = np.zeros((B, whh.shape[0]), F32) if h0 is None else h0[l].astype(F32)
        whhT = q(whh).T                                  # rounded once, not once per time step (q of a 2048 x 2048 matrix is ~30 ms)
        for t in range(S):
            h = q(relu(zx[:, t] + q(h) @ whhT).astype(F32))
            H[:, t] = h
        c[f"H{l}"] = H
        hn.append(h)
        inp = H
    c["h_n"] = np.stack(hn, 0)
    return inp, c


def rnn_bwd(P, G, pre, c, dH1):
    B, S, Hn = dH1.shape
    dout = dH1
    for l in (1, 0):
        wih, whh = P[f"{pre}weight_ih_l{l}"], P[f"{pre}weight_hh_l{l}"]
        H = c[f"H{l}"]
        inp = c["H0"] if l == 1 else c["x"]
        dZ = np.zeros((B, S, Hn), F32)
        carry = np.zeros((B, Hn), F32)
        dout = qg(dout)                                  # dH of the layer is a stored 16-bit tensor (residual operand of the BPTT step)
        whh_q = q(whh)
        for t in reversed(range(S)):
            dz = qg((dout[:, t] + carry) * (H[:, t] > 0))
            dZ[:, t] = dz
            carry = dz @ whh_q
        dz2 = dZ.reshape(B * S, Hn)
        Hprev = np.concatenate([np.zeros((B, 1, Hn), F32), H[:, :-1]], 1).reshape(B * S, Hn)
        _acc(G, f"{pre}weight_hh_l{l}", dz2.T @ q(Hprev))
        _acc(G, f"{pre}weight_ih_l{l}", dz2.T @ q(inp.reshape(B * S, -1)))
        _acc(G, f"{pre}bias_ih_l{l}", dz2.sum(0))
        _acc(G, f"{pre}bias_hh_l{l}", dz2.sum(0))
        dout = (dz2 @ q(wih)).reshape(B, S, -1).astype(F32)
    return dout       # grad w.r.t. decoder input x


def logistic_loss(logit_probs, log_scales_raw, means, gripper_logits, actions_tcp,
                  num_classes=10, log_scale_min=-7.0, gripper_alpha=1.0, amin=-1.0, amax=1.0):
    """logistic_decoder_rnn.py:136-152 (_loss), :184-231 (_logistic_loss), :19-24 (log_sum_exp).

    Returns loss and grads w.r.t. (logit_probs, log_scales_raw, means, gripper_logits)."""
    B, S, Dd, K = means.shape
    a = actions_tcp[..., :Dd, None].astype(F32) * np.ones((1, 1, 1, K), F32)
    ls = np.maximum(log_scales_raw, F32(log_scale_min))
    inv = np.exp(-ls)
    hb = F32(((amax - amin) / 2.0) / (num_classes - 1))
    cen = a - means
    plus = inv * (cen + hb)
    minus = inv * (cen - hb)
    mid = inv * cen
    sp, sm = sigmoid(plus), sigmoid(minus)
    delta = sp - sm
    caseA = a < amin + 1e-3
    caseB = (~caseA) & (a > amax - 1e-3)
    caseC = (~caseA) & (~caseB) & (delta > 1e-5)
    caseD = ~(caseA | caseB | caseC)
    logp = np.where(caseA, plus - softplus(plus),
                    np.where(caseB, -softplus(minus),
                             np.where(caseC, np.log(np.maximum(delta, 1e-12)),
                                      mid - ls - 2.0 * softplus(mid) - np.log((num_classes - 1) / 2.0))))
    lsm = log_softmax(logit_probs, -1)
    lp = logp + lsm
    m = lp.max(-1, keepdims=True)
    lse = m[..., 0] + np.log(np.exp(lp - m).sum(-1))
    logistics = -(lse.sum(-1)).mean()
    # gripper cross entropy (:144-151) ; labels: -1 -> 0, else long(value).  discrete_gripper=False (mcil_default.yaml): no head, :152
    if gripper_logits is not None:
        g = actions_tcp[..., -1]
        lab = np.where(g == -1, 0, g).astype(np.int64)
        glsm = log_softmax(gripper_logits, -1)
        ce = -np.take_along_axis(glsm, lab[..., None], -1)[..., 0].mean()
        loss = F32(logistics + gripper_alpha * ce)
    else:
        loss = F32(logistics)
    # ---- backward
    n = B * S
    wk = np.exp(lp - lse[..., None])
    dlogp = -wk / n
    dlogit = -(wk - np.exp(lsm)) / n
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        g_plus = np.where(caseA, sigmoid(-plus), np.where(caseC, sp * (1 - sp) / np.where(caseC, delta, 1.0), 0.0))
        g_minus = np.where(caseB, -sm, np.where(caseC, -sm * (1 - sm) / np.where(caseC, delta, 1.0), 0.0))
    g_mid = np.where(caseD, 1.0 - 2.0 * sigmoid(mid), 0.0)
    dmean = dlogp * (-inv) * (g_plus + g_minus + g_mid)
    dls = dlogp * (-(g_plus * plus + g_minus * minus + g_mid * mid) - np.where(caseD, 1.0, 0.0))
    dls_raw = np.where(log_scales_raw >= log_scale_min, dls, 0.0)
    if gripper_logits is None:
        return loss, (dlogit.astype(F32), dls_raw.astype(F32), dmean.astype(F32), None)
    dgl = np.exp(glsm)
    np.put_along_axis(dgl, lab[..., None], np.take_along_axis(dgl, lab[..., None], -1) - 1.0, -1)
    dgl = gripper_alpha * dgl / n
    return loss, (dlogit.astype(F32), dls_raw.astype(F32), dmean.astype(F32), dgl.astype(F32))


def decoder_loss_fwd(P, plan, emb, goal, actions, robot_obs, dims):
    """LogisticDecoderRNN.loss :121-134 -> forward :260-287."""
    ad = "action_decoder."
    B, S, _ = emb.shape
    mcil = dims.kind == "mcil"
    pe = emb[..., dims.emb - dims.dec_emb:dims.emb]               # perceptual_emb_slice [64,128]; mcil: no slice (mcil_default.yaml)
    parts = []
    if plan is not None and plan.shape[-1] > 0:
        parts.append(np.repeat(plan[:, None, :], S, 1))
    parts += [pe, np.repeat(goal[:, None, :], S, 1)]
    x = np.concatenate(parts, -1).astype(F32)
    H1, rc = rnn_fwd(P, ad + "rnn.", x)
    h2 = H1.reshape(B * S, -1)
    K, Dd = dims.n_mix, dims.mix_dims
    probs = linear(h2, P[ad + "prob_fc.weight"], P[ad + "prob_fc.bias"]).reshape(B, S, Dd, K)
    means = linear(h2, P[ad + "mean_fc.weight"], P[ad + "mean_fc.bias"]).reshape(B, S, Dd, K)
    lsr = linear(h2, P[ad + "log_scale_fc.weight"], P[ad + "log_scale_fc.bias"]).reshape(B, S, Dd, K)
    grip = None if mcil else linear(h2, P[ad + "gripper_fc.weight"], P[ad + "gripper_fc.bias"]).reshape(B, S, 2)
    a_tcp = actions.astype(F32) if mcil else world_to_tcp_frame(actions, robot_obs)      # gripper_control: false (logistic_decoder_rnn.py:133-134)
    loss, grads = logistic_loss(probs, lsr, means, grip, a_tcp, num_classes=dims.mix_classes)
    c = dict(rc=rc, x=x, H1=H1, probs=probs, means=means, log_scales=lsr, gripper=grip, a_tcp=a_tcp, grads=grads)
    return loss, c


def decoder_loss_bwd(P, G, c, dims, scale):
    ad = "action_decoder."
    H1 = c["H1"]
    B, S, Hn = H1.shape
    h2 = H1.reshape(B * S, Hn)
    dlogit, dls, dmean, dgl = [None if g is None else g * F32(scale) for g in c["grads"]]
    dH = np.zeros((B * S, Hn), F32)
    for nm, d in (("prob_fc", dlogit), ("mean_fc", dmean), ("log_scale_fc", dls), ("gripper_fc", dgl)):
        if d is None:
            continue
        dx, dw, db = linear_bwd(h2, P[f"{ad}{nm}.weight"], d.reshape(B * S, -1))
        _acc(G, f"{ad}{nm}.weight", dw)
        _acc(G, f"{ad}{nm}.bias", db)
        dH += dx
    dx = rnn_bwd(P, G, ad + "rnn.", c["rc"], dH.reshape(B, S, Hn))
    np_ = dims.dec_plan
    dplan = dx[..., :np_].sum(1) if np_ > 0 else None
    dpe = dx[..., np_:np_ + dims.dec_emb]
    dgoal = dx[..., np_ + dims.dec_emb:].sum(1)
    return dplan, dpe, dgoal


# ----------------------------------------------------------------------------------------------------
# KL (hulc.py:539-561), straight-through sample (distributions.py:23-27), CLIP aux (hulc.py:650-695)
# ----------------------------------------------------------------------------------------------------
# ----------------------------------------------------------------------------------------------------
# mcil variant (conf/model/mcil.yaml): bidirectional tanh-RNN plan recognition, continuous latent plan
# ----------------------------------------------------------------------------------------------------
def _birnn_zx(inp2d, wih, b, l):
    """The hoisted input projection of one direction of a plan-encoder layer as the engines hold it: fp32 in the plain mode; in the
    rounding-aware mode a stored 16-bit tensor — layer 1's input is [H0 forward | H0 reverse], projected as two K = 2048 GEMMs of which the
    first result is stored (16-bit) and re-read as the residual of the second (engine.h birnn_fwd / bigru_fwd)."""
    if _QMODE is None:
        return (inp2d @ wih.T + b).astype(F32)
    if l == 0:
        return q(mm(inp2d, wih.T) + b)
    Hn = inp2d.shape[1] // 2
    return q(q(mm(inp2d[:, :Hn], wih[:, :Hn].T) + b) + mm(inp2d[:, Hn:], wih[:, Hn:].T))


def bigru_fwd(P, emb):
    """PlanRecognitionBiRNNNetwork.forward with rnn_type nn.GRU (plan_recognition_net.py:27-42; torch.nn.GRU cell equations):
    r = sig(W_ir x + b_ir + W_hr h + b_hr), z = sig(W_iz x + b_iz + W_hz h + b_hz), n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),
    h' = (1 - z) * n + z * h; gate blocks stacked r | z | n along dim 0 of weight_ih / weight_hh."""
    pre = "plan_recognition.birnn_model."
    B, S, _ = emb.shape
    c = {"inp0": emb}
    inp = emb
    for l in range(2):
        outs = []
        for sfx, order in (("", range(S)), ("_reverse", range(S - 1, -1, -1))):
            wih, whh = P[f"{pre}weight_ih_l{l}{sfx}"], P[f"{pre}weight_hh_l{l}{sfx}"]
            bih, bhh = P[f"{pre}bias_ih_l{l}{sfx}"], P[f"{pre}bias_hh_l{l}{sfx}"]
            Hn = whh.shape[1]
            zx = _birnn_zx(inp.reshape(B * S, -1), wih, bih, l).reshape(B, S, 3 * Hn)
            Hs, R, Z, N, GN, HP = (np.zeros((B, S, Hn), F32) for _ in range(6))
            h = np.zeros((B, Hn), F32)
            whhT = q(whh).T
            for t in order:
                g = (q(h) @ whhT + bhh).astype(F32)           # fp32 accumulators; the gates are computed on them in the same launch
                r = sigmoid(zx[:, t, :Hn] + g[:, :Hn]).astype(F32)
                z = sigmoid(zx[:, t, Hn:2 * Hn] + g[:, Hn:2 * Hn]).astype(F32)
                n = np.tanh(zx[:, t, 2 * Hn:] + r * g[:, 2 * Hn:]).astype(F32)
                HP[:, t] = h
                h = q(((1.0 - z) * n + z * h).astype(F32))     # the state is a stored 16-bit tensor; r, z, n, W_hn h + b_hn are kept (16-bit) for the backward
                Hs[:, t], R[:, t], Z[:, t], N[:, t], GN[:, t] = h, q(r), q(z), q(n), q(g[:, 2 * Hn:])
            c[f"g{l}{sfx}"] = (Hs, R, Z, N, GN, HP)
            outs.append(Hs)
        inp = np.concatenate(outs, -1)
        c[f"out{l}"] = inp
    x = inp[:, -1]
    c["x"] = x
    state = linear(x, P["plan_recognition.fc_state.0.weight"], P["plan_recognition.fc_state.0.bias"])
    return state, x, c


def bigru_bwd(P, G, c, dstate):
    pre = "plan_recognition.birnn_model."
    dx, dw, db = linear_bwd(c["x"], P["plan_recognition.fc_state.0.weight"], dstate)
    _acc(G, "plan_recognition.fc_state.0.weight", dw)
    _acc(G, "plan_recognition.fc_state.0.bias", db)
    emb = c["inp0"]
    B, S, _ = emb.shape
    Hn = c["g0"][0].shape[-1]
    dout = np.zeros((B, S, 2 * Hn), F32)
    dout[:, -1] = qg(dx)                                 # d x is a stored 16-bit tensor (as is every dH below)
    for l in (1, 0):
        inp = c["out0"] if l == 1 else emb
        dinp = np.zeros_like(inp)
        for k, (sfx, order) in enumerate((("", list(range(S))), ("_reverse", list(range(S - 1, -1, -1))))):
            wih, whh = P[f"{pre}weight_ih_l{l}{sfx}"], P[f"{pre}weight_hh_l{l}{sfx}"]
            Hs, R, Z, N, GN, HP = c[f"g{l}{sfx}"]
            dH = dout[..., k * Hn:(k + 1) * Hn]
            dZx = np.zeros((B, S, 3 * Hn), F32)          # grads of the input-side pre-activations (r | z | n)
            dGh = np.zeros((B, S, 3 * Hn), F32)          # grads of the hidden-side pre-activations (n block scaled by r)
            carry = np.zeros((B, Hn), F32)
            whh_q = q(whh)
            for i in reversed(range(S)):
                t = order[i]
                dh = dH[:, t] + carry
                r, z, n, gn, hp = R[:, t], Z[:, t], N[:, t], GN[:, t], HP[:, t]
                dn = dh * (1.0 - z) * (1.0 - n * n)
                dz = dh * (hp - n) * z * (1.0 - z)
                dr = dn * gn * r * (1.0 - r)
                dZx[:, t] = qg(np.concatenate([dr, dz, dn], -1))       # both pre-activation gradients are stored 16-bit GEMM operands
                dGh[:, t] = qg(np.concatenate([dr, dz, dn * r], -1))
                carry = qg(dh * z) + dGh[:, t] @ whh_q                 # the direct path is stored (16-bit), the GEMM part stays in fp32 accumulators
            _acc(G, f"{pre}weight_hh_l{l}{sfx}", dGh.reshape(B * S, -1).T @ q(HP.reshape(B * S, Hn)))
            _acc(G, f"{pre}bias_hh_l{l}{sfx}", dGh.reshape(B * S, -1).sum(0))
            _acc(G, f"{pre}weight_ih_l{l}{sfx}", dZx.reshape(B * S, -1).T @ q(inp.reshape(B * S, -1)))
            _acc(G, f"{pre}bias_ih_l{l}{sfx}", dZx.reshape(B * S, -1).sum(0))
            dinp += (dZx.reshape(B * S, -1) @ q(wih)).reshape(B, S, -1)
        dout = (qg(dinp) if l == 1 else dinp).astype(F32)       # layer 0's output gradient is stored (16-bit); d emb accumulates in fp32
    return dout


def birnn_fwd(P, emb):
    """PlanRecognitionBiRNNNetwork.forward (plan_recognition_net.py:37-42): nn.RNN(tanh, 2 layers, bidirectional, batch_first);
    x = output[:, -1] = [forward hidden after the last step | reverse hidden at the last position (= its FIRST step)]."""
    pre = "plan_recognition.birnn_model."
    B, S, _ = emb.shape
    c = {"inp0": emb}
    inp = emb
    for l in range(2):
        outs = []
        for sfx, order in (("", range(S)), ("_reverse", range(S - 1, -1, -1))):
            wih, whh = P[f"{pre}weight_ih_l{l}{sfx}"], P[f"{pre}weight_hh_l{l}{sfx}"]
            b = P[f"{pre}bias_ih_l{l}{sfx}"] + P[f"{pre}bias_hh_l{l}{sfx}"]
            zx = _birnn_zx(inp.reshape(B * S, -1), wih, b, l).reshape(B, S, -1)
            Hs = np.zeros((B, S, whh.shape[0]), F32)
            h = np.zeros((B, whh.shape[0]), F32)
            whhT = q(whh).T
            for t in order:
                h = q(np.tanh(zx[:, t] + q(h) @ whhT).astype(F32))
                Hs[:, t] = h
            c[f"H{l}{sfx}"] = Hs
            outs.append(Hs)
        inp = np.concatenate(outs, -1)
        c[f"out{l}"] = inp
    x = inp[:, -1]
    c["x"] = x
    state = linear(x, P["plan_recognition.fc_state.0.weight"], P["plan_recognition.fc_state.0.bias"])
    return state, x, c


def birnn_bwd(P, G, c, dstate):
    pre = "plan_recognition.birnn_model."
    dx, dw, db = linear_bwd(c["x"], P["plan_recognition.fc_state.0.weight"], dstate)
    _acc(G, "plan_recognition.fc_state.0.weight", dw)
    _acc(G, "plan_recognition.fc_state.0.bias", db)
    emb = c["inp0"]
    B, S, _ = emb.shape
    Hn = c["H0"].shape[-1]
    dout = np.zeros((B, S, 2 * Hn), F32)
    dout[:, -1] = qg(dx)                                 # d x is a stored 16-bit tensor (as is every dH below)
    for l in (1, 0):
        inp = c["out0"] if l == 1 else emb
        dinp = np.zeros_like(inp)
        for k, (sfx, order) in enumerate((("", list(range(S))), ("_reverse", list(range(S - 1, -1, -1))))):
            wih, whh = P[f"{pre}weight_ih_l{l}{sfx}"], P[f"{pre}weight_hh_l{l}{sfx}"]
            Hs = c[f"H{l}{sfx}"]
            dH = dout[..., k * Hn:(k + 1) * Hn]
            dZ = np.zeros((B, S, Hn), F32)
            carry = np.zeros((B, Hn), F32)
            Hprev = np.zeros((B, S, Hn), F32)
            whh_q = q(whh)
            for i in reversed(range(S)):                     # reverse of the processing order
                t = order[i]
                dz = qg((dH[:, t] + carry) * (1.0 - Hs[:, t] ** 2))
                dZ[:, t] = dz
                carry = dz @ whh_q
                if i > 0:
                    Hprev[:, t] = Hs[:, order[i - 1]]
            dz2 = dZ.reshape(B * S, Hn)
            _acc(G, f"{pre}weight_hh_l{l}{sfx}", dz2.T @ q(Hprev.reshape(B * S, Hn)))
            _acc(G, f"{pre}weight_ih_l{l}{sfx}", dz2.T @ q(inp.reshape(B * S, -1)))
            _acc(G, f"{pre}bias_ih_l{l}{sfx}", dz2.sum(0))
            _acc(G, f"{pre}bias_hh_l{l}{sfx}", dz2.sum(0))
            dinp += (dz2 @ q(wih)).reshape(B, S, -1)
        dout = (qg(dinp) if l == 1 else dinp).astype(F32)       # layer 0's output gradient is stored (16-bit); d emb accumulates in fp32
    return dout              # grad w.r.t. perceptual_emb (B,S,128)


def cont_state(state):
    """Distribution.forward_dist, continuous (distributions.py:55-59): mean, std = softplus(var) + 1e-4."""
    n = state.shape[-1] // 2
    mean, var = state[..., :n], state[..., n:]
    return mean.astype(F32), (softplus(var) + F32(1e-4)).astype(F32), var


def kl_normal_balanced(pp_state, pr_state, beta=0.01, alpha=0.8):
    """Hulc.compute_kl_loss (hulc.py:539-561) for Independent(Normal): KL(pr || pp) summed over the plan dims, mean over the batch,
    balanced with stop-gradients.  Returns loss and the gradients w.r.t. the two fc_state outputs (mean | var)."""
    B = pp_state.shape[0]
    m2, s2, v2 = cont_state(pp_state)
    m1, s1, v1 = cont_state(pr_state)
    kl = (np.log(s2 / s1) + (s1 ** 2 + (m1 - m2) ** 2) / (2 * s2 ** 2) - 0.5).sum(-1).mean()
    loss = F32(beta * (alpha * kl + (1 - alpha) * kl))
    # lhs: gradient to the prior (pp); rhs: gradient to the posterior (pr)
    dm2 = -(m1 - m2) / s2 ** 2
    ds2 = 1.0 / s2 - (s1 ** 2 + (m1 - m2) ** 2) / s2 ** 3
    dm1 = (m1 - m2) / s2 ** 2
    ds1 = -1.0 / s1 + s1 / s2 ** 2
    dpp = np.concatenate([dm2, ds2 * sigmoid(v2)], -1) * (beta * alpha / B)
    dpr = np.concatenate([dm1, ds1 * sigmoid(v1)], -1) * (beta * (1 - alpha) / B)
    return loss, dpp.astype(F32), dpr.astype(F32)


# ----------------------------------------------------------------------------------------------------
# dataloader image transforms: uint8 HWC frames -> the fp32 NCHW tensors the step consumes
# ----------------------------------------------------------------------------------------------------
def random_shifts_aug(x, shift, pad):
    """hulc/utils/transforms.py:8-29 (RandomShiftsAug.forward) for given integer draws.  x (n,c,h,w) float, shift (n,2) = the
    torch.randint(0, 2*pad+1) draws (column 0 shifts x / width, column 1 shifts y / height).  The replicate-padded image is sampled
    by grid_sample on a grid that lands on pixel centres, so out[y][x] = padded[y + sy][x + sx] = in[clamp(y+sy-pad)][clamp(x+sx-pad)]."""
    n, c, h, w = x.shape
    out = np.empty_like(x)
    ys, xs = np.arange(h), np.arange(w)
    for i in range(n):
        sy = np.clip(ys + int(shift[i, 1]) - pad, 0, h - 1)
        sx = np.clip(xs + int(shift[i, 0]) - pad, 0, w - 1)
        out[i] = x[i][:, sy][:, :, sx]
    return out


def relative_actions(actions_abs, robot_obs, max_pos, max_orn):
    """hulc/utils/transforms.py:32-56 (RelativeActions.__call__), batched over leading dims: absolute tcp targets (..., 7) and the raw
    robot state (..., >= 6) -> clipped / scaled relative position and wrapped orientation differences, gripper action unchanged."""
    a = np.asarray(actions_abs)
    ro = np.asarray(robot_obs)
    rel_pos = np.clip(a[..., :3] - ro[..., :3], -max_pos, max_pos) / max_pos
    diff = a[..., 3:6] - ro[..., 3:6]
    rel_orn = (diff + np.pi) % (2 * np.pi) - np.pi                              # batch_angle_between(robot_obs, actions) :41-44
    rel_orn = np.clip(rel_orn, -max_orn, max_orn) / max_orn
    return np.concatenate([rel_pos, rel_orn, a[..., 6:7]], -1)


def ingest_u8(frames, shift=None, pad=0):
    """conf/datamodule/transforms/rand_shift.yaml (train: rgb_static / rgb_gripper): uint8 (B,S,H,W,C) ->
    RandomShiftsAug(pad) [skipped when shift is None: the `val` transforms] -> ScaleImageTensor (x/255) -> Normalize(0.5, 0.5);
    returns fp32 (B,S,C,H,W) in [-1,1]."""
    B, S, H, W, C = frames.shape
    x = np.transpose(frames.reshape(B * S, H, W, C), (0, 3, 1, 2)).astype(F32)
    if shift is not None:
        x = random_shifts_aug(x, np.asarray(shift).reshape(B * S, 2), pad)
    x = ((x / F32(255.0) - F32(0.5)) / F32(0.5)).astype(F32)
    return x.reshape(B, S, C, H, W)


# ----------------------------------------------------------------------------------------------------
# validation / rollout forward: forward-only, stochastic draws injected
# ----------------------------------------------------------------------------------------------------
def tcp_to_world_frame(action, robot_obs):
    """gripper_control.py:39-63 (fp32); inverse(R(e)) == R(e)^T; the NaN fallback branch (:52-56) is not restated."""
    act = action.astype(F32)
    e = robot_obs[..., 3:6].astype(F32)
    R = euler_xyz_to_matrix(e)
    pos = (R @ act[..., :3, None])[..., 0]
    Rrel = euler_xyz_to_matrix(act[..., 3:6] * F32(0.01))
    M = R @ np.swapaxes(Rrel, -1, -2)
    o0 = np.arctan2(-M[..., 1, 2], M[..., 2, 2])
    o1 = np.arcsin(M[..., 0, 2])
    o2 = np.arctan2(-M[..., 0, 1], M[..., 0, 0])
    orn = np.stack([o0, o1, o2], -1).astype(F32) - e
    orn = np.where(orn < -np.pi, orn + 2 * np.pi, orn)
    orn = np.where(orn > np.pi, orn - 2 * np.pi, orn)
    orn = (orn * 100).astype(F32)
    return np.concatenate([pos, orn, act[..., -1:]], -1).astype(F32)


def logistic_sample(logit_probs, log_scales_raw, means, gripper_logits, u_mix, u_act, log_scale_min=-7.0):
    """logistic_decoder_rnn.py:234-258 (_sample).  u_mix (B,S,D,K) and u_act (B,S,D) are the two torch.rand draws in [0,1)
    BEFORE the affine map to [1e-5, 1-1e-5] (:238-239, :250); gripper command = bounds[argmax] with bounds (-1, 1) (:59)."""
    r1, r2 = F32(1e-5), F32(1.0 - 1e-5)
    ls = np.maximum(log_scales_raw, F32(log_scale_min))           # forward() clamps before _sample sees them (:279)
    t = ((r1 - r2) * u_mix.astype(F32) + r2).astype(F32)
    g = logit_probs - np.log(-np.log(t))
    k = g.argmax(-1)
    sel_ls = np.take_along_axis(ls, k[..., None], -1)[..., 0]
    sel_mu = np.take_along_axis(means, k[..., None], -1)[..., 0]
    u = ((r1 - r2) * u_act.astype(F32) + r2).astype(F32)
    act = (sel_mu + np.exp(sel_ls) * (np.log(u) - np.log(1.0 - u))).astype(F32)
    if gripper_logits is None:                                     # discrete_gripper false (mcil_default.yaml): :257-258, all 7 dims sampled
        return act
    grip = np.where(gripper_logits.argmax(-1) == 0, F32(-1.0), F32(1.0))
    return np.concatenate([act, grip[..., None]], -1).astype(F32)


def decoder_heads(P, plan, emb, goal, dims, h0=None):
    """LogisticDecoderRNN.forward :260-287 -> (logit_probs, log_scales_raw, means, gripper_logits, h_n)."""
    ad = "action_decoder."
    B, S, _ = emb.shape
    parts = []
    if plan is not None and plan.shape[-1] > 0:
        parts.append(np.repeat(plan[:, None, :], S, 1))
    parts += [emb[..., dims.emb - dims.dec_emb:dims.emb], np.repeat(goal[:, None, :], S, 1)]
    x = np.concatenate(parts, -1).astype(F32)
    H1, rc = rnn_fwd(P, ad + "rnn.", x, h0)
    h2 = H1.reshape(B * S, -1)
    K, Dd = dims.n_mix, dims.mix_dims
    probs = linear(h2, P[ad + "prob_fc.weight"], P[ad + "prob_fc.bias"]).reshape(B, S, Dd, K)
    means = linear(h2, P[ad + "mean_fc.weight"], P[ad + "mean_fc.bias"]).reshape(B, S, Dd, K)
    lsr = linear(h2, P[ad + "log_scale_fc.weight"], P[ad + "log_scale_fc.bias"]).reshape(B, S, Dd, K)
    grip = None if dims.kind == "mcil" else linear(h2, P[ad + "gripper_fc.weight"], P[ad + "gripper_fc.bias"]).reshape(B, S, 2)
    return probs, lsr, means, grip, rc["h_n"]


def encode(P, rgb_static, rgb_gripper):
    """ConcatEncoders.forward (concat_encoders.py:59-109) on (B,S,3,H,W) frames -> (B,S,128)."""
    B, S = rgb_static.shape[:2]
    es, _ = static_encoder_fwd(P, "perceptual_encoder.rgb_static_encoder.", rgb_static.reshape((B * S,) + rgb_static.shape[2:]).astype(F32))
    eg, _ = gripper_encoder_fwd(P, "perceptual_encoder.rgb_gripper_encoder.", rgb_gripper.reshape((B * S,) + rgb_gripper.shape[2:]).astype(F32))
    return np.concatenate([es.reshape(B, S, -1), eg.reshape(B, S, -1)], -1).astype(F32)


def goal_encode(P, emb_last_or_lang, is_lang):
    names, ln = (LG_NAMES, "language_goal.ln") if is_lang else (VG_NAMES, "visual_goal.ln")
    gpre, _ = mlp_fwd(P, names, emb_last_or_lang.astype(F32), False)
    goal, _ = layer_norm(gpre, P[ln + ".weight"], P[ln + ".bias"])
    return goal


def onehot_plan(idx, dims):
    B = idx.shape[0]
    oh = np.zeros((B, dims.n_cat, dims.n_cls), F32)
    np.put_along_axis(oh, idx[..., None].astype(np.int64), 1.0, -1)
    return oh.reshape(B, -1)


def validation_forward(P, dims, mb, is_lang, noise):
    """Hulc.validation_step body for one modality (hulc.py:770-797) -> lmp_val (:301-388) + the logged reductions (:816-833).

    noise: plan_idx_pp / plan_idx_pr (B, n_cat) — the categorical samples (distributions.py:37-41), u_mix_pp/u_act_pp and
    u_mix_pr/u_act_pr — the torch.rand draws of the two _sample calls."""
    B, S = mb["actions"].shape[:2]
    emb = encode(P, mb["rgb_static"], mb["rgb_gripper"])
    goal = goal_encode(P, mb["lang"] if is_lang else emb[:, -1], is_lang)
    if dims.kind == "mcil":
        # continuous plans: noise["plan_pp"] / ["plan_pr"] (B,256) are the draws of Independent(Normal).sample() (distributions.py:37-38);
        # gripper_control false: loss and sample stay in the world frame (logistic_decoder_rnn.py:99-100)
        pp_state = mlp_fwd(P, PP_NAMES, np.concatenate([emb[:, 0], goal], -1), False)[0]
        pr_state, seq_feat, _ = (bigru_fwd if dims.rnn_type == "gru" else birnn_fwd)(P, emb)
        out = {"seq_feat": seq_feat, "pp_logits": pp_state, "pr_logits": pr_state}
        for tag in ("pp", "pr"):
            probs, lsr, means, _, _ = decoder_heads(P, noise[f"plan_{tag}"].astype(F32), emb, goal, dims)
            loss, _ = logistic_loss(probs, lsr, means, None, mb["actions"].astype(F32), num_classes=dims.mix_classes)
            pred = logistic_sample(probs, lsr, means, None, noise[f"u_mix_{tag}"], noise[f"u_act_{tag}"])
            mae = np.abs(pred[..., :-1] - mb["actions"][..., :-1]).mean(1)
            sr = F32((np.where(pred[..., -1] > 0, 1.0, -1.0) == mb["actions"][..., -1]).mean())
            out.update({f"action_loss_{tag}": loss, f"mae_{tag}": mae.astype(F32), f"gripper_sr_{tag}": sr, f"pred_{tag}": pred})
        out["kl_loss"], _, _ = kl_normal_balanced(pp_state, pr_state)
        return out
    pp_logits = mlp_fwd(P, PP_NAMES, np.concatenate([emb[:, 0], goal], -1), False)[0] if dims.kind == "hulc" else None
    pr_logits, seq_feat, _ = plan_recognition_fwd(P, emb, dims.heads)
    a_tcp = world_to_tcp_frame(mb["actions"], mb["robot_obs"])
    out = {"seq_feat": seq_feat, "pp_logits": pp_logits, "pr_logits": pr_logits}
    for tag in (("pp", "pr") if dims.kind == "hulc" else ("pp",)):     # GCBC (gcbc.py:214-246): one pass, no plan — reported under "pp"
        plan = onehot_plan(noise[f"plan_idx_{tag}"], dims) if dims.kind == "hulc" else None
        probs, lsr, means, grip, _ = decoder_heads(P, plan, emb, goal, dims)
        loss, _ = logistic_loss(probs, lsr, means, grip, a_tcp, num_classes=dims.num_classes)
        pred = logistic_sample(probs, lsr, means, grip, noise[f"u_mix_{tag}"], noise[f"u_act_{tag}"])
        pred_w = tcp_to_world_frame(pred, mb["robot_obs"])
        mae = np.abs(pred_w[..., :-1] - mb["actions"][..., :-1]).mean(1)          # (B, 6)  hulc.py:347-350
        sr = F32((np.where(pred_w[..., -1] > 0, 1.0, -1.0) == mb["actions"][..., -1]).mean())
        out.update({f"action_loss_{tag}": loss, f"mae_{tag}": mae.astype(F32), f"gripper_sr_{tag}": sr, f"pred_{tag}": pred_w})
    if is_lang and dims.use_clip and mb.get("use_for_aux") is not None:      # val/val_pred_clip_loss (hulc.py:804-808)
        out["val_pred_clip_loss"] = clip_loss(P, seq_feat, goal, mb["use_for_aux"].astype(bool))[0]
    if dims.kind != "hulc":
        return out
    out["kl_loss"], _, _ = kl_loss(pp_logits, pr_logits, dims)
    return out


class Rollout:
    """Hulc.reset / step / get_pp_plan_vision / get_pp_plan_lang / predict_with_plan (hulc.py:843-957) + the stateful
    LogisticDecoderRNN.act (logistic_decoder_rnn.py:102-116).  Draws are injected per call."""

    def __init__(self, P, dims, replan_freq=30):
        self.P, self.dims, self.replan_freq = P, dims, replan_freq
        self.reset()

    def reset(self):
        if self.dims.kind == "gcbc" and hasattr(self, "h"):      # gcbc.py:281-285: `self.latent_goal = None` and nothing else
            self.goal = None
            return
        self.plan = None
        self.goal = None
        self.h = None
        self.counter = 0

    def step(self, obs, goal, noise):
        """obs: rgb_static (1,1,3,200,200), rgb_gripper (1,1,3,84,84), robot_obs_raw (1,1,15); goal: same two images (vision)
        or a (1,384) language embedding; noise: plan_idx (1,n_cat) on replan steps, u_mix (1,1,6,10), u_act (1,1,6)."""
        P, dims = self.P, self.dims
        if dims.kind == "gcbc":
            # GCBC.reset / step (gcbc.py:281-320): the goal is encoded once per rollout (reset() only drops the goal); the decoder runs
            # without a plan and its hidden state is never cleared — LogisticDecoderRNN.act keeps it and gcbc.py has no clear_hidden_state call
            if self.goal is None:
                if isinstance(goal, dict):
                    emb = encode(P, np.concatenate([obs["rgb_static"], goal["rgb_static"]], 1), np.concatenate([obs["rgb_gripper"], goal["rgb_gripper"]], 1))
                    self.goal = goal_encode(P, emb[:, -1], False)
                else:
                    self.goal = goal_encode(P, goal, True)
            emb = encode(P, obs["rgb_static"], obs["rgb_gripper"])
            probs, lsr, means, grip, self.h = decoder_heads(P, None, emb, self.goal, dims, self.h)
            pred = logistic_sample(probs, lsr, means, grip, noise["u_mix"], noise["u_act"])
            self.counter += 1
            return tcp_to_world_frame(pred, obs["robot_obs_raw"])
        if self.counter % self.replan_freq == 0:
            if isinstance(goal, dict):
                emb = encode(P, np.concatenate([obs["rgb_static"], goal["rgb_static"]], 1), np.concatenate([obs["rgb_gripper"], goal["rgb_gripper"]], 1))
                self.goal = goal_encode(P, emb[:, -1], False)
            else:
                emb = encode(P, obs["rgb_static"], obs["rgb_gripper"])
                self.goal = goal_encode(P, goal, True)
            pp_logits, _ = mlp_fwd(P, PP_NAMES, np.concatenate([emb[:, 0], self.goal], -1), False)
            self.pp_logits = pp_logits
            self.plan = noise["plan"].astype(F32) if dims.kind == "mcil" else onehot_plan(noise["plan_idx"], dims)
            self.h = None                                          # clear_hidden_state (hulc.py:925 / :946)
        emb = encode(P, obs["rgb_static"], obs["rgb_gripper"])
        probs, lsr, means, grip, self.h = decoder_heads(P, self.plan, emb, self.goal, dims, self.h)
        pred = logistic_sample(probs, lsr, means, grip, noise["u_mix"], noise["u_act"])
        self.counter += 1
        return pred if dims.kind == "mcil" else tcp_to_world_frame(pred, obs["robot_obs_raw"])


def kl_loss(pp_logits, pr_logits, dims, beta=0.01, alpha=0.8):
    B = pp_logits.shape[0]
    a = log_softmax(pr_logits.reshape(B, dims.n_cat, dims.n_cls), -1)
    b = log_softmax(pp_logits.reshape(B, dims.n_cat, dims.n_cls), -1)
    p, q = np.exp(a), np.exp(b)
    klc = (p * (a - b)).sum(-1)                 # (B, n_cat)
    kl = klc.sum(-1).mean()
    loss = F32(beta * (alpha * kl + (1 - alpha) * kl))
    dpp = (beta * alpha / B) * (q - p)
    dpr = (beta * (1 - alpha) / B) * p * ((a - b) - klc[..., None])
    return loss, dpp.reshape(B, -1).astype(F32), dpr.reshape(B, -1).astype(F32)


def clip_loss(P, seq_feat, goal, mask):
    if not mask.any():
        return F32(0.0), None
    sf, lg = seq_feat[mask], goal[mask]
    n = sf.shape[0]
    names_im = ["proj_vis_lang.mlp_im.0", "proj_vis_lang.mlp_im.2"]
    names_la = ["proj_vis_lang.mlp_lang.0", "proj_vis_lang.mlp_lang.2"]
    img, acts_i = mlp_fwd(P, names_im, sf, False)
    txt, acts_t = mlp_fwd(P, names_la, lg, False)
    ni = np.linalg.norm(img, axis=-1, keepdims=True)
    nt = np.linalg.norm(txt, axis=-1, keepdims=True)
    i_n, t_n = img / ni, txt / nt
    s = np.exp(P["logit_scale"])
    cos = i_n @ t_n.T
    logits = s * cos
    lr, lc = log_softmax(logits, 1), log_softmax(logits, 0)
    idx = np.arange(n)
    loss = F32((-lr[idx, idx].mean() - lc[idx, idx].mean()) / 2)
    eye = np.eye(n, dtype=F32)
    dlog = ((np.exp(lr) - eye) + (np.exp(lc) - eye)) / (2 * n)
    cache = dict(mask=mask, acts_i=acts_i, acts_t=acts_t, names_im=names_im, names_la=names_la,
                 i_n=i_n, t_n=t_n, ni=ni, nt=nt, s=s, cos=cos, dlog=dlog)
    return loss, cache


def clip_gt_setup(train_ann, train_task, train_emb, val_instr_tasks, val_emb):
    """Hulc.on_fit_start (hulc.py:697-737) on in-memory annotation data.  train_ann / train_task: per training annotation; train_emb (N,1,384);
    val_instr_tasks: the keys of model.val_instructions in order, val_emb (K,1,384) their embeddings.  The reference orders distinct
    instructions / task ids by set iteration; first occurrence is used here (the metrics do not depend on the order)."""
    first = {}
    for i, a in enumerate(train_ann):
        first.setdefault(str(a), i)
    ids = list(first.values())
    tasks = [str(train_task[i]) for i in ids]
    task_to_id = {}
    for t in tasks:
        task_to_id.setdefault(t, len(task_to_id))
    keep = [k for k, t in enumerate(val_instr_tasks) if str(t) in task_to_id]            # hulc.py:730-731
    return dict(train_emb=np.asarray(train_emb)[ids].reshape(len(ids), -1).astype(F32), train_task_ids=np.array([task_to_id[t] for t in tasks]),
                val_emb=np.asarray(val_emb)[keep].reshape(len(keep), -1).astype(F32), val_task_ids=np.array([task_to_id[str(val_instr_tasks[k])] for k in keep]),
                task_to_id=task_to_id)


def clip_gt_loss(P, seq_feat_masked, encoded_lang, task_ids, gt_tasks):
    """Hulc._clip_groundtruth_loss (hulc.py:1007-1043)."""
    img, _ = mlp_fwd(P, ["proj_vis_lang.mlp_im.0", "proj_vis_lang.mlp_im.2"], seq_feat_masked, False)
    txt, _ = mlp_fwd(P, ["proj_vis_lang.mlp_lang.0", "proj_vis_lang.mlp_lang.2"], encoded_lang, False)
    i_n = img / np.linalg.norm(img, axis=-1, keepdims=True)
    t_n = txt / np.linalg.norm(txt, axis=-1, keepdims=True)
    logits = (np.exp(P["logit_scale"]) * i_n) @ t_n.T
    scores = logits - logits.min(1, keepdims=True)
    scores = scores / (scores.max(1, keepdims=True) - scores.min(1, keepdims=True))
    loss = []
    for score, gt in zip(scores, gt_tasks):
        loss.append(score[task_ids == gt].sum(dtype=F32) - score[task_ids != gt].sum(dtype=F32))
    sr = float(np.mean(task_ids[np.argmax(scores, 1)] == gt_tasks))
    return F32(np.mean(loss)), sr, logits.astype(F32)


def clip_groundtruth(P, setup, seq_feat, mask, gt_tasks_all):
    """Hulc.on_validation_epoch_start + clip_groundtruth (hulc.py:967-974, 980-1005): the four `lang_gt/*` values, or None when no row is masked in."""
    mask = np.asarray(mask, bool)
    if not mask.any():
        return None
    gt = np.asarray(gt_tasks_all)[mask]
    sf = seq_feat[mask]
    out = {}
    for tag in ("train", "val"):
        enc = goal_encode(P, setup[f"{tag}_emb"], True)
        out[f"lang_gt/{tag}_gt"], out[f"lang_gt/{tag}_sr"], out[f"logits_{tag}"] = clip_gt_loss(P, sf, enc, setup[f"{tag}_task_ids"], gt)
    return out


def clip_loss_bwd(P, G, c, scale, B):
    dlog = c["dlog"] * F32(scale)
    s = c["s"]
    _acc(G, "logit_scale", np.asarray((dlog * c["cos"]).sum() * s, F32).reshape(()))
    din = s * (dlog @ c["t_n"])
    dtn = s * (dlog.T @ c["i_n"])
    dimg = (din - c["i_n"] * (c["i_n"] * din).sum(-1, keepdims=True)) / c["ni"]
    dtxt = (dtn - c["t_n"] * (c["t_n"] * dtn).sum(-1, keepdims=True)) / c["nt"]
    dsf_m = mlp_bwd(P, G, c["names_im"], c["acts_i"], dimg.astype(F32), False)
    dg_m = mlp_bwd(P, G, c["names_la"], c["acts_t"], dtxt.astype(F32), False)
    dsf = np.zeros((B, dsf_m.shape[1]), F32)
    dg = np.zeros((B, dg_m.shape[1]), F32)
    dsf[c["mask"]] = dsf_m
    dg[c["mask"]] = dg_m
    return dsf, dg


# ----------------------------------------------------------------------------------------------------
# one modality pass + whole step  (hulc.py:390-537, gcbc.py:50-181)
# ----------------------------------------------------------------------------------------------------
PP_NAMES = ["plan_proposal.fc_model.0", "plan_proposal.fc_model.2", "plan_proposal.fc_model.4",
            "plan_proposal.fc_model.6", "plan_proposal.fc_state.0"]
VG_NAMES = ["visual_goal.mlp.0", "visual_goal.mlp.2", "visual_goal.mlp.4"]
LG_NAMES = ["language_goal.mlp.1", "language_goal.mlp.3", "language_goal.mlp.5"]


def modality_fwd(P, dims, mb, is_lang):
    """mb: dict(rgb_static (B,S,3,200,200), rgb_gripper (B,S,3,84,84), actions (B,S,7), robot_obs (B,S,15),
    plan_idx (B,n_cat) int [hulc], lang (B,384), use_for_aux (B,) bool [lang])."""
    c = {}
    B, S = mb["actions"].shape[:2]
    xs = mb["rgb_static"].reshape((B * S,) + mb["rgb_static"].shape[2:]).astype(F32)
    xg = mb["rgb_gripper"].reshape((B * S,) + mb["rgb_gripper"].shape[2:]).astype(F32)
    es, c["enc_s"] = static_encoder_fwd(P, "perceptual_encoder.rgb_static_encoder.", xs)
    eg, c["enc_g"] = gripper_encoder_fwd(P, "perceptual_encoder.rgb_gripper_encoder.", xg)
    emb = np.concatenate([es.reshape(B, S, -1), eg.reshape(B, S, -1)], -1).astype(F32)   # concat_encoders.py:86
    c["emb"] = emb
    if is_lang:
        gpre, c["goal_acts"] = mlp_fwd(P, LG_NAMES, mb["lang"].astype(F32), False)
        goal, c["goal_ln"] = layer_norm(gpre, P["language_goal.ln.weight"], P["language_goal.ln.bias"])
    else:
        gpre, c["goal_acts"] = mlp_fwd(P, VG_NAMES, emb[:, -1], False)
        goal, c["goal_ln"] = layer_norm(gpre, P["visual_goal.ln.weight"], P["visual_goal.ln.bias"])
    goal = q(goal)                          # latent_goal is a stored 16-bit tensor in the half-precision engines
    c["goal"] = goal
    if dims.kind == "mcil":
        pr_state, seq_feat, c["pr"] = (bigru_fwd if dims.rnn_type == "gru" else birnn_fwd)(P, emb)
        pp_state, c["pp_acts"] = mlp_fwd(P, PP_NAMES, np.concatenate([emb[:, 0], goal], -1), False)
        mean, std, _ = cont_state(pr_state)
        plan = (mean + std * mb["plan_eps"].astype(F32)).astype(F32)          # pr_dist.rsample() (hulc.py:289) with the injected N(0,1) draw
        c.update(pr_logits=pr_state, pp_logits=pp_state, seq_feat=seq_feat, plan=plan, pr_std=std)
        act, c["dec"] = decoder_loss_fwd(P, plan, emb, goal, mb["actions"], mb["robot_obs"], dims)
        kl, c["dpp_kl"], c["dpr_kl"] = kl_normal_balanced(pp_state, pr_state)
        c["clip"] = None
        return dict(kl=kl, action=act, total=F32(act + kl), clip=F32(0)), c
    drop = None
    if TRAIN_DROPOUT is not None and TRAIN_DROPOUT[0] > 0:
        pdrop, dseed, dstep = TRAIN_DROPOUT[:3]
        b0 = TRAIN_DROPOUT[3] if len(TRAIN_DROPOUT) > 3 else 0           # every masked tensor has the window index as its slowest dimension
        drop = (pdrop, lambda site, shape: engine_keep_mask(engine_site_seed(dseed, dstep, is_lang, site), shape, pdrop, first=b0 * (int(np.prod(shape)) // B)))
    pr_logits, seq_feat, c["pr"] = plan_recognition_fwd(P, emb, dims.heads, drop=drop)
    c["pr_logits"], c["seq_feat"] = pr_logits, seq_feat
    out = {}
    if dims.kind == "hulc":
        ppx = np.concatenate([emb[:, 0], goal], -1)
        pp_logits, c["pp_acts"] = mlp_fwd(P, PP_NAMES, ppx, False)
        # fc_model layers all have ReLU, fc_state has none: names[-1] is fc_state (plan_proposal_net.py:26-40)
        c["pp_logits"] = pp_logits
        idx = mb["plan_idx"]
        probs = softmax(pr_logits.reshape(B, dims.n_cat, dims.n_cls), -1).astype(F32)
        onehot = np.zeros_like(probs)
        np.put_along_axis(onehot, idx[..., None], 1.0, -1)
        plan = (onehot + probs - probs).reshape(B, -1).astype(F32)      # distributions.py:27 / hulc.py:289-291
        c["pr_probs"] = probs
        c["plan"] = plan
        act, c["dec"] = decoder_loss_fwd(P, plan, emb, goal, mb["actions"], mb["robot_obs"], dims)
        kl, c["dpp_kl"], c["dpr_kl"] = kl_loss(pp_logits, pr_logits, dims)
        out.update(kl=kl, action=act, total=F32(act + kl))
    else:
        act, c["dec"] = decoder_loss_fwd(P, None, emb, goal, mb["actions"], mb["robot_obs"], dims)
        out.update(kl=F32(0), action=act, total=act)
    out["clip"] = F32(0)
    c["clip"] = None
    if is_lang and dims.use_clip:
        out["clip"], c["clip"] = clip_loss(P, seq_feat, goal, mb["use_for_aux"].astype(bool))
    return out, c


def modality_bwd(P, G, dims, c, is_lang, w_mod, w_clip):
    emb = c["emb"]
    B, S, _ = emb.shape
    demb = np.zeros_like(emb)
    dplan, dpe, dgoal = decoder_loss_bwd(P, G, c["dec"], dims, w_mod)
    demb[..., dims.emb - dims.dec_emb:] += dpe
    dsf = None
    if dims.kind == "mcil":
        n = dims.plan
        _, _, var = cont_state(c["pr_logits"])
        # plan = mean + std * eps -> d mean = dplan, d var = dplan * eps * sigmoid(var)
        dpr = np.concatenate([dplan, dplan * ((c["plan"] - c["pr_logits"][:, :n]) / c["pr_std"]) * sigmoid(var)], -1) + c["dpr_kl"] * F32(w_mod)
        demb += (bigru_bwd if dims.rnn_type == "gru" else birnn_bwd)(P, G, c["pr"], dpr.astype(F32))
        dppx = mlp_bwd(P, G, PP_NAMES, c["pp_acts"], (c["dpp_kl"] * F32(w_mod)).astype(F32), False)
        demb[:, 0] += dppx[:, :dims.emb]
        dgoal = dgoal + dppx[:, dims.emb:]
    if c["clip"] is not None:
        dsf, dg_clip = clip_loss_bwd(P, G, c["clip"], w_clip, B)
        dgoal = dgoal + dg_clip
    dpr_logits = None
    if dims.kind == "hulc":
        probs = c["pr_probs"]
        dpl = dplan.reshape(B, dims.n_cat, dims.n_cls)
        dst = probs * (dpl - (dpl * probs).sum(-1, keepdims=True))      # straight-through softmax Jacobian
        dpr_logits = (dst.reshape(B, -1) + c["dpr_kl"] * F32(w_mod)).astype(F32)
        dpp = (c["dpp_kl"] * F32(w_mod)).astype(F32)
        dppx = mlp_bwd(P, G, PP_NAMES, c["pp_acts"], dpp, False)
        demb[:, 0] += dppx[:, :dims.emb]
        dgoal = dgoal + dppx[:, dims.emb:]
    if dims.kind != "mcil" and (dpr_logits is not None or dsf is not None):
        demb += plan_recognition_bwd(P, G, c["pr"], dpr_logits, dsf, dims.heads,
                                     fc_state_used=dims.kind == "hulc")
    lnn = "language_goal.ln" if is_lang else "visual_goal.ln"
    dgp, dg, db = layer_norm_bwd(dgoal.astype(F32), P[lnn + ".weight"], c["goal_ln"])
    _acc(G, lnn + ".weight", dg)
    _acc(G, lnn + ".bias", db)
    din = mlp_bwd(P, G, LG_NAMES if is_lang else VG_NAMES, c["goal_acts"], dgp, False)
    if not is_lang:
        demb[:, -1] += din
    static_encoder_bwd(P, G, "perceptual_encoder.rgb_static_encoder.", c["enc_s"],
                       np.ascontiguousarray(demb[..., :64]).reshape(B * S, 64))
    gripper_encoder_bwd(P, G, "perceptual_encoder.rgb_gripper_encoder.", c["enc_g"],
                        np.ascontiguousarray(demb[..., 64:]).reshape(B * S, 64))


def training_step(P, dims, batch, clip_beta=3.0, want_grads=True, keep_cache=False):
    """batch: {"vis": mb, "lang": mb} (either may be absent), iterated in that insertion order.

    Returns (losses dict, grads dict or None[, caches])."""
    losses = {}
    caches = {}
    nmod = len(batch)
    tot = F32(0)
    for scope, mb in batch.items():
        o, c = modality_fwd(P, dims, mb, "lang" in scope)
        caches[scope] = c
        for k, v in o.items():
            losses[f"{k}_{scope}"] = F32(v)
        tot = tot + o["total"]
    tot = tot / nmod
    clip = sum(losses.get(f"clip_{s}", 0.0) for s in batch)
    if dims.use_clip:
        tot = tot + clip_beta * clip
    losses["total"] = F32(tot)
    losses["kl"] = F32(sum(losses[f"kl_{s}"] for s in batch) / nmod)
    losses["action"] = F32(sum(losses[f"action_{s}"] for s in batch) / nmod)
    losses["clip"] = F32(clip_beta * clip)
    G = None
    if want_grads:
        G = {}
        for scope in batch:
            modality_bwd(P, G, dims, caches[scope], "lang" in scope, 1.0 / nmod, clip_beta)
        for n in P:                       # params untouched this step get an explicit zero gradient
            G.setdefault(n, np.zeros_like(P[n]))
    if keep_cache:
        return losses, G, caches
    return losses, G


def train_mode_losses(P, dims, mb, rng, dropout_p=0.1, enc_cache=None):
    """One TRAIN-mode forward of the vision-goal HULC step with the oracle's own stochastic draws: dropout masks in the
    plan-recognition transformer (plan_recognition_net.py:111 + the encoder layers) and the categorical plan sample drawn from the posterior
    (hulc.py:289 rsample()).  Returns (dict(action, kl, total), enc_cache); the perceptual encoders are deterministic, so their output is
    computed once and passed back in through enc_cache."""
    B, S = mb["actions"].shape[:2]
    if enc_cache is None:
        xs = mb["rgb_static"].reshape((B * S,) + mb["rgb_static"].shape[2:]).astype(F32)
        xg = mb["rgb_gripper"].reshape((B * S,) + mb["rgb_gripper"].shape[2:]).astype(F32)
        es, _ = static_encoder_fwd(P, "perceptual_encoder.rgb_static_encoder.", xs)
        eg, _ = gripper_encoder_fwd(P, "perceptual_encoder.rgb_gripper_encoder.", xg)
        enc_cache = np.concatenate([es.reshape(B, S, -1), eg.reshape(B, S, -1)], -1).astype(F32)
    emb = enc_cache
    gpre, _ = mlp_fwd(P, VG_NAMES, emb[:, -1], False)
    goal, _ = layer_norm(gpre, P["visual_goal.ln.weight"], P["visual_goal.ln.bias"])
    pr_logits, _, _ = plan_recognition_fwd(P, emb, dims.heads, drop=(dropout_p, rng) if dropout_p > 0 else None)
    pp_logits, _ = mlp_fwd(P, PP_NAMES, np.concatenate([emb[:, 0], goal], -1), False)
    probs = softmax(pr_logits.reshape(B, dims.n_cat, dims.n_cls).astype(np.float64), -1)
    u = rng.random((B, dims.n_cat, 1))
    idx = np.minimum((np.cumsum(probs, -1) < u).sum(-1), dims.n_cls - 1)          # inverse-CDF categorical draw per (window, category)
    onehot = np.zeros((B, dims.n_cat, dims.n_cls), F32)
    np.put_along_axis(onehot, idx[..., None], 1.0, -1)
    act, _ = decoder_loss_fwd(P, onehot.reshape(B, -1), emb, goal, mb["actions"], mb["robot_obs"], dims)
    kl, _, _ = kl_loss(pp_logits, pr_logits, dims)
    return dict(action=float(act), kl=float(kl), total=float(act + kl)), enc_cache


def adam_step(P, G, state, step, lr=2e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (conf/model/optimizer/adam.yaml:1-2; hulc.py:239-252). step counts from 1."""
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for n, g in G.items():
        m = state.setdefault("m." + n, np.zeros_like(P[n]))
        v = state.setdefault("v." + n, np.zeros_like(P[n]))
        m[...] = b1 * m + (1 - b1) * g
        v[...] = b2 * v + (1 - b2) * g * g
        denom = np.sqrt(v) / math.sqrt(bc2) + eps
        P[n] = (P[n] - (lr / bc1) * m / denom).astype(F32)


# ----------------------------------------------------------------------------------------------------
# MiniLM sentence encoder: sentence_transformers "all-MiniLM-L6-v2" = transformers BertModel (pinned version in
# this image: transformers 5.x, models/bert/modeling_bert.py: BertEmbeddings, BertSelfAttention, BertSelfOutput, BertIntermediate,
# BertOutput) -> Pooling(mean, attention-mask weighted, clamp(min=1e-9)) -> Normalize (F.normalize p=2, eps 1e-12).
# The library is a third-party dependency absent from /root/reference; parity is pinned on transformers.BertModel run in the build
# container with seeded weights (tools/gen_golden_sbert.py -> tests/golden/sbert_minilm.npz).
# ----------------------------------------------------------------------------------------------------
def _erf(x):
    # Abramowitz-Stegun 7.1.26 is too coarse (1.5e-7 abs) for a 2e-5 parity bar on 6 layers -> use math.erf elementwise
    return np.vectorize(math.erf, otypes=[np.float64])(x)


def sbert_forward(W, ids, mask, heads=12, eps=1e-12, normalize=True):
    """W: BertModel state_dict (numpy); ids, mask: (B,L) ints.  Returns (sentence_embeddings (B,H), last_hidden_state (B,L,H))."""
    B, Ln = ids.shape
    H = W["embeddings.word_embeddings.weight"].shape[1]
    x = (W["embeddings.word_embeddings.weight"][ids] + W["embeddings.position_embeddings.weight"][None, :Ln]
         + W["embeddings.token_type_embeddings.weight"][0][None, None]).astype(F32)
    x = layer_norm(x.reshape(B * Ln, H), W["embeddings.LayerNorm.weight"], W["embeddings.LayerNorm.bias"], eps)[0].reshape(B, Ln, H)
    D = H // heads
    bias = np.where(mask[:, None, None, :] != 0, 0.0, -np.inf).astype(F32)           # extended attention mask
    nl = 1 + max(int(k.split(".")[2]) for k in W if k.startswith("encoder.layer."))
    for l in range(nl):
        p = f"encoder.layer.{l}."
        x2 = x.reshape(B * Ln, H)
        q, k, v = (linear(x2, W[p + f"attention.self.{n}.weight"], W[p + f"attention.self.{n}.bias"]).reshape(B, Ln, heads, D).transpose(0, 2, 1, 3)
                   for n in ("query", "key", "value"))
        sc = (q @ k.transpose(0, 1, 3, 2)) / F32(math.sqrt(D)) + bias
        ctx = (softmax(sc, -1).astype(F32) @ v).transpose(0, 2, 1, 3).reshape(B * Ln, H)
        y = linear(ctx, W[p + "attention.output.dense.weight"], W[p + "attention.output.dense.bias"]) + x2
        x1 = layer_norm(y, W[p + "attention.output.LayerNorm.weight"], W[p + "attention.output.LayerNorm.bias"], eps)[0]
        h = linear(x1, W[p + "intermediate.dense.weight"], W[p + "intermediate.dense.bias"])
        h = (0.5 * h * (1.0 + _erf(h.astype(np.float64) / math.sqrt(2.0)))).astype(F32)
        y2 = linear(h, W[p + "output.dense.weight"], W[p + "output.dense.bias"]) + x1
        x = layer_norm(y2, W[p + "output.LayerNorm.weight"], W[p + "output.LayerNorm.bias"], eps)[0].reshape(B, Ln, H)
    m = (mask != 0).astype(F32)[..., None]
    emb = (x * m).sum(1) / np.maximum(m.sum(1), 1e-9)
    if normalize:
        emb = emb / np.maximum(np.linalg.norm(emb, axis=-1, keepdims=True), 1e-12)
    return emb.astype(F32), x
