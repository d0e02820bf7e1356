"""Shared helpers for the golden-fixture tests (fixtures come from tools/gen_golden.py = the reference)."""
import os

import numpy as np

from hulc_amd import spec
from hulc_amd.utils import portable_rng as prng
from hulc_amd.utils import synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NSAMP = 64
# name: (kind, Bv, Bl, S, use_clip, aux_mask, edge_frac, seed)  -- must mirror tools/gen_golden.py CASES
CASES = {
    "hulc_tiny": ("hulc", 2, 2, 4, True, "all", 0.05, 1),
    "hulc_s32": ("hulc", 2, 3, 32, True, "some", 0.05, 2),
    "hulc_visonly": ("hulc", 3, 0, 8, False, "all", 0.05, 3),
    "gcbc_s16": ("gcbc", 2, 2, 16, True, "all", 0.05, 4),
    "hulc_edge": ("hulc", 1, 2, 5, True, "none", 0.6, 5),
    "hulc_s64": ("hulc", 2, 2, 64, True, "all", 0.05, 6, 64),        # BASELINE config 5's window length; 9th field = position-table rows
}


def load_case(name):
    kind, Bv, Bl, S, use_clip, aux_mask, edge_frac, seed = CASES[name][:8]
    dims = spec.ModelDims(kind=kind, max_window=CASES[name][8] if len(CASES[name]) > 8 else 32, use_clip=use_clip)
    P = spec.init_all(dims, seed=seed, ln_jitter=True)
    batch = synthetic.make_batch(Bv, Bl, S, seed=seed, edge_frac=edge_frac, aux_mask=aux_mask)
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    for sc in batch:
        if f"plan_idx_{sc}" in fx.files:
            batch[sc]["plan_idx"] = fx[f"plan_idx_{sc}"]
    return dims, P, batch, fx


def sample_idx(name, n):
    return prng.randint("sample." + name, (NSAMP,), n, 0)


def rel_l2(a, b):
    a = np.asarray(a, np.float64).reshape(-1)
    b = np.asarray(b, np.float64).reshape(-1)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def check_grads(G, fx, tol_l2=5e-3, tol_norm=2e-3, label=""):
    """G: name -> full gradient array (reference layout). Compares with the reference's fixture entries."""
    bad = []
    for key in fx.files:
        if key.startswith("gradnorm/"):
            n = key[len("gradnorm/"):]
            ref_norm = float(fx[key])
            g = np.asarray(G[n], np.float64)
            norm = float(np.sqrt((g ** 2).sum()))
            if ref_norm < 1e-12:
                if norm > 1e-7:
                    bad.append((n, "norm-nonzero", norm))
                continue
            if abs(norm - ref_norm) / ref_norm > tol_norm:
                bad.append((n, "norm", norm, ref_norm))
            if "grad/" + n in fx.files:
                e = rel_l2(g, fx["grad/" + n])
            else:
                e = rel_l2(g.reshape(-1)[sample_idx(n, g.size)], fx["gradsamp/" + n])
            if e > tol_l2:
                bad.append((n, "rel_l2", e))
    assert not bad, f"{label} gradient mismatches: {bad[:8]} (+{max(0, len(bad) - 8)} more)"


# Tensors whose fp32 evaluation — the reference's own as much as ours — sits above 1e-3 of the fp64 truth: the conv gradients sum
# O(10^5..10^6) per-pixel terms in fp32 (accumulation order), and a ReLU pre-activation within fp32 noise of zero flips its mask.
# Everything NOT named here is held to 1e-3 rel-L2 against the reference's float64 gradients.
FP32_NOISY = ("conv_model.0.bias", "conv_model.2.bias", "conv_model.4.bias", "conv_model.0.weight", "conv_model.2.weight", "conv_model.4.weight")


def check_grads64(G, fx, tol_l2=1e-3, tol_noisy=5e-3, label="", noisy=FP32_NOISY, scale=1.0):
    """G: name -> full gradient array.  Against the reference's FLOAT64 gradient entries (grad64/ gradsamp64/ gradnorm64/, tools/gen_golden.py):
    rel-L2 <= tol_l2 for every tensor, tol_noisy for the named fp32-noise-limited ones.  Returns the worst (error, name) per class."""
    bad, worst, worst_noisy = [], (0.0, ""), (0.0, "")
    for key in fx.files:
        if not key.startswith("gradnorm64/"):
            continue
        n = key[len("gradnorm64/"):]
        ref_norm = float(fx[key])
        g = np.asarray(G[n], np.float64) / scale
        if ref_norm < 1e-12:
            if float(np.sqrt((g ** 2).sum())) > 1e-7:
                bad.append((n, "norm-nonzero"))
            continue
        if "grad64/" + n in fx.files:
            e = rel_l2(g, fx["grad64/" + n])
        else:
            e = rel_l2(g.reshape(-1)[sample_idx(n, g.size)], fx["gradsamp64/" + n])
        e = max(e, abs(float(np.sqrt((g ** 2).sum())) - ref_norm) / ref_norm)
        is_noisy = any(n.endswith(x) for x in noisy)
        if is_noisy:
            worst_noisy = max(worst_noisy, (e, n))
        else:
            worst = max(worst, (e, n))
        if e > (tol_noisy if is_noisy else tol_l2):
            bad.append((n, round(e, 6)))
    assert not bad, f"{label} gradient mismatches vs the fp64 reference: {bad[:10]} (+{max(0, len(bad) - 10)} more)"
    return worst, worst_noisy


def grad_entries(fx, n):
    """the reference gradient at the same entries the adam1/adam2 fixtures hold (None if grad was None)."""
    if "grad/" + n in fx.files:
        return fx["grad/" + n].reshape(-1)
    if "gradsamp/" + n in fx.files:
        return fx["gradsamp/" + n]
    return None


def adam_close(got, ref, g, lr=2e-4, steps=1):
    """Adam's update is lr*g/(|g|+1e-8)-like: entries whose gradient is at fp32-noise level (|g| ~ eps) may
    legitimately differ by up to 2*lr per step (sign flip); everywhere else the parameters must agree tightly."""
    err = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))
    if g is None:
        return bool((err <= 2 * steps * lr * 1.05 + 1e-6).all()) and float(np.median(err)) < 1e-5
    big = np.abs(g) > 1e-5
    return bool((err[big] <= 4e-6 + 2e-3 * lr).all()) and bool((err[~big] <= 2 * lr * 1.05 + 1e-6).all())


# ---- validation / rollout fixtures (tools/gen_golden_val.py = the reference's lmp_val / step)
VAL_CASES = {"val_hulc_tiny": (2, 2, 4, True, 11), "val_hulc_s16": (3, 0, 16, False, 12), "val_gcbc_s8": (2, 2, 8, True, 13, "gcbc"),
             "val_mcil_s8": (2, 2, 8, False, 14, "mcil")}   # (Bv, Bl, S, use_clip, seed[, kind])
VAL_NOISE_KEYS = ("plan_idx_pp", "plan_idx_pr", "plan_pp", "plan_pr", "u_mix_pp", "u_act_pp", "u_mix_pr", "u_act_pr")


def load_val_case(name):
    Bv, Bl, S, use_clip, seed = VAL_CASES[name][:5]
    kind = VAL_CASES[name][5] if len(VAL_CASES[name]) > 5 else "hulc"
    dims = spec.ModelDims(kind=kind, max_window=32, use_clip=use_clip)
    P = spec.init_all(dims, seed=seed, ln_jitter=True)
    batch = synthetic.make_batch(Bv, Bl, S, seed=seed, edge_frac=0.05, aux_mask="all")
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    noise = {sc: {k: fx[f"{k}_{sc}"] for k in VAL_NOISE_KEYS if f"{k}_{sc}" in fx.files} for sc in batch}
    return dims, P, batch, noise, fx


def load_rollout_case(name="rollout_hulc"):
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    nsteps, replan_freq, seed = (int(v) for v in fx["meta"])
    mcil = name.endswith("mcil")
    dims = spec.ModelDims(kind="mcil" if mcil else "hulc", max_window=32, use_clip=not mcil)
    P = spec.init_all(dims, seed=seed, ln_jitter=True)
    frames = synthetic.make_batch(1, 1, nsteps + 1, seed=seed, edge_frac=0.0, aux_mask="all")
    return dims, P, frames, nsteps, replan_freq, fx


# ---- mcil variant (tools/gen_golden_mcil.py)
MCIL_CASES = {"mcil_s6": (2, 2, 6, 41), "mcil_s12": (3, 0, 12, 32), "mcil_gru_s6": (2, 2, 6, 43, "gru")}       # name: (Bv, Bl, S, seed[, rnn_type])


def load_mcil_case(name):
    Bv, Bl, S, seed = MCIL_CASES[name][:4]
    dims = spec.ModelDims(kind="mcil", max_window=32, use_clip=False, rnn_type=MCIL_CASES[name][4] if len(MCIL_CASES[name]) > 4 else "rnn")
    P = spec.init_all(dims, seed=seed, ln_jitter=True)
    batch = synthetic.make_batch(Bv, Bl, S, seed=seed, edge_frac=0.05, aux_mask="all")
    fx = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    for sc in batch:
        batch[sc]["plan_eps"] = fx[f"plan_eps_{sc}"]
    return dims, P, batch, fx
