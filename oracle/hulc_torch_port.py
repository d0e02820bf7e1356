"""TEST INFRASTRUCTURE ONLY — a second CPU restatement of the HULC / GCBC training step, on torch's CPU library kernels with autograd.

Why it exists: `bench.py`'s `cpu_baseline` leg has to time "the reference's CPU path" on the GPU box's host cores, and the reference
itself never travels there.  The numpy oracle (oracle/hulc_oracle.py) is the parity checker, but as a BASELINE it is unfair: its im2col +
OpenBLAS convolutions are ~5x slower than the mkldnn / ATen kernels the reference runs on.  This file restates the same step
(forward + loss, autograd backward, torch.optim.Adam) with torch.nn.functional ops — i.e. on the very library kernels the reference's
CPU path executes — so that the baseline is the reference's arithmetic at the reference's library speed.  It is NOT the reference's code:
every function below is written from the closed forms of SURVEY.md Appendix A / the cited reference lines, takes the flat state_dict of
numpy arrays the oracle takes, and is pinned against the same reference fixtures (tests/test_oracle_golden.py::test_torch_port_*).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product path never does.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def _t(P, dtype=torch.float32):
    return {k: torch.tensor(np.asarray(v), dtype=dtype).requires_grad_(True) for k, v in P.items()}


def _mlp(W, names, x, last_relu=False):
    for i, n in enumerate(names):
        x = F.linear(x, W[n + ".weight"], W[n + ".bias"])
        if i < len(names) - 1 or last_relu:
            x = F.relu(x)
    return x


def _static_encoder(W, pre, x):
    """vision_network.py:11-66: conv8s4 -> conv4s2 -> conv3s1 (ReLU each) -> SpatialSoftmax (:74-108) -> fc1+ReLU -> fc2 -> LayerNorm."""
    x = F.relu(F.conv2d(x, W[pre + "conv_model.0.weight"], W[pre + "conv_model.0.bias"], stride=4))
    x = F.relu(F.conv2d(x, W[pre + "conv_model.2.weight"], W[pre + "conv_model.2.bias"], stride=2))
    x = F.relu(F.conv2d(x, W[pre + "conv_model.4.weight"], W[pre + "conv_model.4.bias"], stride=1))
    n, c, h, w = x.shape
    p = F.softmax(x.reshape(n, c, h * w), dim=-1).reshape(n, c, h, w)
    lin_h = torch.linspace(-1.0, 1.0, h, dtype=x.dtype)
    lin_w = torch.linspace(-1.0, 1.0, w, dtype=x.dtype)
    ex = (p.sum(3) * lin_h).sum(2)                      # "x" follows the ROW index (meshgrid indexing="ij", :88-92)
    ey = (p.sum(2) * lin_w).sum(2)
    feat = torch.stack([ex, ey], -1).reshape(n, 2 * c)
    x = F.relu(F.linear(feat, W[pre + "fc1.0.weight"], W[pre + "fc1.0.bias"]))
    x = F.linear(x, W[pre + "fc2.weight"], W[pre + "fc2.bias"])
    return F.layer_norm(x, (x.shape[-1],), W[pre + "ln.weight"], W[pre + "ln.bias"], 1e-5)


def _gripper_encoder(W, pre, x):
    """vision_network_gripper.py:10-57: nature_cnn (3 convs + Flatten(C,H,W) + Linear 3136->128, ReLU) -> fc1+ReLU -> fc2 -> LayerNorm."""
    x = F.relu(F.conv2d(x, W[pre + "conv_model.0.weight"], W[pre + "conv_model.0.bias"], stride=4))
    x = F.relu(F.conv2d(x, W[pre + "conv_model.2.weight"], W[pre + "conv_model.2.bias"], stride=2))
    x = F.relu(F.conv2d(x, W[pre + "conv_model.4.weight"], W[pre + "conv_model.4.bias"], stride=1))
    x = F.relu(F.linear(x.flatten(1), W[pre + "conv_model.7.weight"], W[pre + "conv_model.7.bias"]))
    x = F.relu(F.linear(x, W[pre + "fc1.0.weight"], W[pre + "fc1.0.bias"]))
    x = F.linear(x, W[pre + "fc2.weight"], W[pre + "fc2.bias"])
    return F.layer_norm(x, (x.shape[-1],), W[pre + "ln.weight"], W[pre + "ln.bias"], 1e-5)


def _encode(W, rs, rg):
    B, S = rs.shape[:2]
    es = _static_encoder(W, "perceptual_encoder.rgb_static_encoder.", rs.reshape((B * S,) + rs.shape[2:]))
    eg = _gripper_encoder(W, "perceptual_encoder.rgb_gripper_encoder.", rg.reshape((B * S,) + rg.shape[2:]))
    return torch.cat([es.reshape(B, S, -1), eg.reshape(B, S, -1)], -1)       # concat_encoders.py:59-109


def _plan_recognition(W, emb, heads=8):
    """plan_recognition_net.py:94-117, eval mode (dropout off)."""
    pr = "plan_recognition."
    B, S, D = emb.shape
    x = emb + W[pr + "position_embeddings.weight"][:S]
    for l in range(2):
        L = f"{pr}transformer_encoder.layers.{l}."
        qkv = F.linear(x, W[L + "self_attn.in_proj_weight"], W[L + "self_attn.in_proj_bias"])
        q, k, v = (t.reshape(B, S, heads, D // heads).transpose(1, 2) for t in qkv.chunk(3, -1))
        a = F.softmax((q * (1.0 / math.sqrt(D // heads))) @ k.transpose(-1, -2), -1) @ v
        a = a.transpose(1, 2).reshape(B, S, D)
        x = F.layer_norm(x + F.linear(a, W[L + "self_attn.out_proj.weight"], W[L + "self_attn.out_proj.bias"]), (D,), W[L + "norm1.weight"], W[L + "norm1.bias"], 1e-5)
        f = F.linear(F.relu(F.linear(x, W[L + "linear1.weight"], W[L + "linear1.bias"])), W[L + "linear2.weight"], W[L + "linear2.bias"])
        x = F.layer_norm(x + f, (D,), W[L + "norm2.weight"], W[L + "norm2.bias"], 1e-5)
    seq_feat = F.linear(x, W[pr + "fc.weight"], W[pr + "fc.bias"]).mean(1)
    return F.linear(seq_feat, W[pr + "fc_state.0.weight"], W[pr + "fc_state.0.bias"]), seq_feat


def _euler(e):
    """pytorch3d_transforms.py:162-218, convention XYZ: R = Rx(a) Ry(b) Rz(c)  (SURVEY appendix A2)."""
    a, b, c = e[..., 0], e[..., 1], e[..., 2]
    ca, sa, cb, sb, cc, sc = a.cos(), a.sin(), b.cos(), b.sin(), c.cos(), c.sin()
    return torch.stack([torch.stack([cb * cc, -cb * sc, sb], -1),
                        torch.stack([ca * sc + sa * sb * cc, ca * cc - sa * sb * sc, -sa * cb], -1),
                        torch.stack([sa * sc - ca * sb * cc, sa * cc + ca * sb * sc, ca * cb], -1)], -2)


def _world_to_tcp(act, ro):
    """gripper_control.py:16-36 (fp32 island): pos = R^T act[:3]; orn = XYZ-euler(R'^T R) * 100 with R' = R(e + 0.01 act[3:6])."""
    act, ro = act.float(), ro.float()
    R = _euler(ro[..., 3:6])
    Rn = _euler(ro[..., 3:6] + 0.01 * act[..., 3:6])
    pos = (R.transpose(-1, -2) @ act[..., :3, None])[..., 0]
    M = Rn.transpose(-1, -2) @ R
    orn = torch.stack([torch.atan2(-M[..., 1, 2], M[..., 2, 2]), torch.asin(M[..., 0, 2].clamp(-1, 1)), torch.atan2(-M[..., 0, 1], M[..., 0, 0])], -1)
    orn = torch.where(orn < -math.pi, orn + 2 * math.pi, orn)
    orn = torch.where(orn > math.pi, orn - 2 * math.pi, orn)
    return torch.cat([pos, orn * 100.0, act[..., 6:7]], -1)


def _logistic_loss(probs, lsr, means, grip, a, num_classes=10, log_scale_min=-7.0, gripper_alpha=1.0):
    """logistic_decoder_rnn.py:136-152, 184-231; bounds +-1."""
    ls = lsr.clamp(min=log_scale_min)
    act = a[..., :6, None].to(means.dtype)
    inv = torch.exp(-ls)
    h = 1.0 / (num_classes - 1)
    cen = act - means
    plus, minus, mid = inv * (cen + h), inv * (cen - h), inv * cen
    delta = torch.sigmoid(plus) - torch.sigmoid(minus)
    logp = torch.where(act < -1 + 1e-3, plus - F.softplus(plus),
                       torch.where(act > 1 - 1e-3, -F.softplus(minus),
                                   torch.where(delta > 1e-5, torch.log(delta.clamp(min=1e-12)), mid - ls - 2.0 * F.softplus(mid) - math.log((num_classes - 1) / 2))))
    lp = logp + F.log_softmax(probs, -1)
    loss = -torch.logsumexp(lp, -1).sum(-1).mean()
    labels = torch.where(a[..., 6] == -1, torch.zeros_like(a[..., 6]), a[..., 6]).long()
    return loss + gripper_alpha * F.cross_entropy(grip.reshape(-1, 2), labels.reshape(-1))


def _decoder_loss(W, plan, emb, goal, actions, robot_obs, num_classes=10):
    """logistic_decoder_rnn.py:121-134, 260-287; rnn_decoder = nn.RNN(relu, 2 layers) (decoders/utils/rnn.py:5-14, appendix A7)."""
    ad = "action_decoder."
    B, S, _ = emb.shape
    parts = ([plan[:, None, :].expand(B, S, -1)] if plan is not None and plan.shape[-1] > 0 else []) + [emb[..., 64:128], goal[:, None, :].expand(B, S, -1)]
    x = torch.cat(parts, -1)
    flat = [W[ad + f"rnn.{n}_l{l}"] for l in range(2) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    h0 = torch.zeros(2, B, 2048, dtype=x.dtype)
    out, _ = torch._VF.rnn_relu(x, h0, flat, True, 2, 0.0, False, False, True)
    K = 10
    probs = F.linear(out, W[ad + "prob_fc.weight"], W[ad + "prob_fc.bias"]).reshape(B, S, 6, K)
    means = F.linear(out, W[ad + "mean_fc.weight"], W[ad + "mean_fc.bias"]).reshape(B, S, 6, K)
    lsr = F.linear(out, W[ad + "log_scale_fc.weight"], W[ad + "log_scale_fc.bias"]).reshape(B, S, 6, K)
    grip = F.linear(out, W[ad + "gripper_fc.weight"], W[ad + "gripper_fc.bias"])
    return _logistic_loss(probs, lsr, means, grip, _world_to_tcp(actions, robot_obs), num_classes)


def _kl(pp, pr, beta=0.01, alpha=0.8):
    """hulc.py:539-561 (appendix A5): balanced KL between two 32x32 categorical states."""
    B = pp.shape[0]
    a, b = F.log_softmax(pr.reshape(B, 32, 32), -1), F.log_softmax(pp.reshape(B, 32, 32), -1)
    kl = lambda x, y: (x.exp() * (x - y)).sum((1, 2)).mean()
    return beta * (alpha * kl(a.detach(), b) + (1 - alpha) * kl(a, b.detach()))


def _clip(W, seq_feat, goal, mask, logit_scale):
    """hulc.py:650-695 + proj_vis_lang.py:7-27."""
    if not bool(mask.any()):
        return None
    img = _mlp(W, ["proj_vis_lang.mlp_im.0", "proj_vis_lang.mlp_im.2"], seq_feat[mask])
    txt = _mlp(W, ["proj_vis_lang.mlp_lang.0", "proj_vis_lang.mlp_lang.2"], goal[mask])
    img, txt = img / img.norm(dim=-1, keepdim=True), txt / txt.norm(dim=-1, keepdim=True)
    logits = logit_scale.exp() * img @ txt.t()
    lab = torch.arange(logits.shape[0])
    return (F.cross_entropy(logits, lab) + F.cross_entropy(logits.t(), lab)) / 2


def training_step(W, kind, batch, use_clip=True, clip_beta=3.0):
    """hulc.py:390-537 / gcbc.py:50-181 on a dict of torch parameters W (requires_grad); returns the total loss tensor and the logged parts.
    batch = {"vis": mb, "lang": mb} of numpy arrays like hulc_amd.utils.synthetic.make_batch; plan_idx = the injected categorical sample."""
    dt = next(iter(W.values())).dtype
    tot, parts, nmod = 0.0, {}, len(batch)
    clip_total = None
    for scope, mb in batch.items():
        is_lang = "lang" in scope
        emb = _encode(W, torch.tensor(mb["rgb_static"], dtype=dt), torch.tensor(mb["rgb_gripper"], dtype=dt))
        if is_lang:
            g = _mlp(W, ["language_goal.mlp.1", "language_goal.mlp.3", "language_goal.mlp.5"], torch.tensor(mb["lang"], dtype=dt))
            goal = F.layer_norm(g, (32,), W["language_goal.ln.weight"], W["language_goal.ln.bias"], 1e-5)
        else:
            g = _mlp(W, ["visual_goal.mlp.0", "visual_goal.mlp.2", "visual_goal.mlp.4"], emb[:, -1])
            goal = F.layer_norm(g, (32,), W["visual_goal.ln.weight"], W["visual_goal.ln.bias"], 1e-5)
        pr_logits, seq_feat = _plan_recognition(W, emb)
        acts, ro = torch.tensor(mb["actions"]), torch.tensor(mb["robot_obs"])
        if kind == "hulc":
            pp = _mlp(W, [f"plan_proposal.fc_model.{i}" for i in (0, 2, 4, 6)], torch.cat([emb[:, 0], goal], -1), last_relu=True)
            pp_logits = F.linear(pp, W["plan_proposal.fc_state.0.weight"], W["plan_proposal.fc_state.0.bias"])
            B = emb.shape[0]
            probs = F.softmax(pr_logits.reshape(B, 32, 32), -1)
            onehot = F.one_hot(torch.tensor(np.asarray(mb["plan_idx"])).long(), 32).to(dt)
            plan = (onehot + probs - probs.detach()).reshape(B, -1)            # straight-through sample (distributions.py:27, appendix A6)
            kl = _kl(pp_logits, pr_logits)
            act = _decoder_loss(W, plan, emb, goal, acts, ro)
            mod = act + kl
            parts[f"kl_{scope}"] = float(kl.detach())
        else:
            act = _decoder_loss(W, None, emb, goal, acts, ro)
            mod = act
        parts[f"action_{scope}"] = float(act.detach())
        tot = tot + mod / nmod
        if is_lang and use_clip:
            c = _clip(W, seq_feat, goal, torch.tensor(np.asarray(mb["use_for_aux"], bool)), W["logit_scale"])
            if c is not None:
                clip_total = c if clip_total is None else clip_total + c
    if use_clip and clip_total is not None:
        tot = tot + clip_beta * clip_total
        parts["clip"] = float((clip_beta * clip_total).detach())
    return tot, parts


class Stepper:
    """One optimizer step = training_step + backward + torch.optim.Adam.step (hulc.py:239-252), for timing and for the Adam fixtures."""

    def __init__(self, P, kind="hulc", use_clip=True, dtype=torch.float32):
        self.W = _t(P, dtype)
        self.kind, self.use_clip = kind, use_clip
        self.opt = torch.optim.Adam(list(self.W.values()), lr=2e-4)

    def step(self, batch):
        self.opt.zero_grad(set_to_none=True)
        loss, parts = training_step(self.W, self.kind, batch, self.use_clip)
        loss.backward()
        self.opt.step()
        return float(loss.detach()), parts

    def grads(self, batch):
        for w in self.W.values():
            w.grad = None
        loss, parts = training_step(self.W, self.kind, batch, self.use_clip)
        loss.backward()
        return float(loss.detach()), parts, {k: (w.grad.detach().numpy().copy() if w.grad is not None else np.zeros(w.shape, np.float32)) for k, w in self.W.items()}
