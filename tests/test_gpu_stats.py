"""GPU (-m gpu): the DEVICE random draws of the training step as distributions.

Parity fixtures inject the stochastic inputs (plan sample, eval-mode dropout); the benchmarked configuration draws them on the device with a
counter-based generator keyed by (context seed, step, site, element).  These tests hold those draws to what the reference's draws ARE:
* `OneHotCategoricalStraightThrough.rsample()` (hulc/models/hulc.py:289, distributions.py:27): >= 10^5 categorical draws against the posterior's
  own probabilities, chi-square over every (window, category) distribution;
* `Independent(Normal).rsample()` of the mcil configuration (distributions.py:55-59): the implied eps = (plan - mean) / std, Kolmogorov-Smirnov
  against N(0, 1) over >= 10^5 draws, plus mean / variance / lag-1 correlation;
* `nn.Dropout(p=0.1)` (plan_recognition_net.py:89,111 and the encoder layers): keep rate 0.9 within 4 sigma, kept values scaled by exactly
  1 / (1 - p), masks change with the step and repeat for the same step (element-wise site; every other site uses the same hash test);
* the mean TRAIN-mode loss over 64 steps (64 seeds) against the oracle's expectation under its own numpy masks and draws: within 1 %
  (SURVEY's bound; measured 1e-4) — the end-to-end check that dropout scaling and sampling feed the loss the way the reference's do."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import hulc_oracle as O  # noqa: E402
from hulc_amd import spec  # noqa: E402
from hulc_amd.utils import synthetic  # noqa: E402
from test_gpu_parity import _engine, to_dev  # noqa: E402


def _softmax(x):
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def test_device_categorical_sampler_follows_the_posterior():
    B, S, K = 8, 4, 420
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    P = spec.init_all(dims, seed=5, ln_jitter=True)
    # sharpen the posterior: at init the logits are ~uniform, and a sampler that ignored them would pass a uniform test
    rng = np.random.default_rng(0)
    P["plan_recognition.fc_state.0.bias"] = (rng.standard_normal(1024) * 0.6).astype(np.float32)
    eng = _engine(dims, B, S, "fp32", dropout=0.0, seed=77)
    eng.load_numpy(P)
    mb = to_dev(synthetic.make_batch(B, 0, S, seed=9)["vis"], inject_plan=False)
    counts = np.zeros((B, 32, 32), np.int64)
    draws = []
    for i in range(K):
        eng.forward_loss(mb, False, 1.0, 3.0, step=i)
        idx = eng.plan_idx(B)
        assert idx.min() >= 0 and idx.max() < 32
        np.add.at(counts, (np.arange(B)[:, None], np.arange(32)[None, :], idx), 1)
        draws.append(idx.copy())
    probs = _softmax(eng.get_tensor("pr_logits", B * 1024).reshape(B, 32, 32).astype(np.float64))
    eng.close()
    assert counts.sum() == B * 32 * K >= 100000
    exp = probs * K
    assert exp.min() > 0.5, exp.min()
    chi2 = float(((counts - exp) ** 2 / exp).sum())
    dof = B * 32 * 31
    z = (chi2 - dof) / np.sqrt(2.0 * dof)
    assert abs(z) < 5.0, (chi2, dof, z)
    # the posterior matters: against the uniform distribution the same counts must be wildly off
    chi2_u = float(((counts - K / 32.0) ** 2 / (K / 32.0)).sum())
    assert (chi2_u - dof) / np.sqrt(2.0 * dof) > 50.0
    # draws of consecutive steps and of neighbouring categories are not copies of each other
    d = np.stack(draws)                               # (K, B, 32)
    assert np.mean(d[1:] == d[:-1]) < 0.2 and np.mean(d[:, :, 1:] == d[:, :, :-1]) < 0.2
    # the same step repeats its draw (counter-based: reproducible training runs)
    eng2 = _engine(dims, B, S, "fp32", dropout=0.0, seed=77)
    eng2.load_numpy(P)
    eng2.forward_loss(mb, False, 1.0, 3.0, step=3)
    assert np.array_equal(eng2.plan_idx(B), draws[3])
    eng2.close()


def test_device_normal_sampler_is_standard_normal():
    from scipy import stats
    B, S, K = 8, 4, 56
    dims = spec.ModelDims(kind="mcil", max_window=32, use_clip=False)
    P = spec.init_all(dims, seed=6, ln_jitter=True)
    eng = _engine(dims, B, S, "fp32", dropout=0.0, seed=123, num_classes=dims.mix_classes)
    eng.load_numpy(P)
    mb = {k: v for k, v in to_dev(synthetic.make_batch(B, 0, S, seed=4)["vis"], inject_plan=False).items()}
    eps = []
    for i in range(K):
        eng.forward_loss(mb, False, 1.0, 3.0, step=i)
        plan = eng.get_tensor("plan", B * 256).reshape(B, 256).astype(np.float64)
        st = eng.get_tensor("pr_logits", B * 512).reshape(B, 512)
        mean, std, _ = O.cont_state(st)
        eps.append((plan - mean.astype(np.float64)) / std.astype(np.float64))
    eng.close()
    e = np.stack(eps)                                 # (K, B, 256)
    x = e.reshape(-1)
    assert x.size >= 100000
    ks = stats.kstest(x, "norm")
    assert ks.pvalue > 1e-3, ks
    n = x.size
    assert abs(x.mean()) < 5.0 / np.sqrt(n) and abs(x.var() - 1.0) < 5.0 * np.sqrt(2.0 / n)
    assert abs(stats.kurtosis(x)) < 0.1 and abs(stats.skew(x)) < 0.05
    lag = np.corrcoef(x[:-1], x[1:])[0, 1]
    step_corr = np.corrcoef(e[:-1].reshape(-1), e[1:].reshape(-1))[0, 1]
    assert abs(lag) < 0.02 and abs(step_corr) < 0.02, (lag, step_corr)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_dropout_keep_rate_scale_and_reproducibility(dtype):
    B, S, K, p = 8, 16, 24, 0.1
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    P = spec.init_all(dims, seed=7, ln_jitter=True)
    eng = _engine(dims, B, S, dtype, dropout=p, seed=31)
    eng.load_numpy(P)
    mb = to_dev(synthetic.make_batch(B, 0, S, seed=2)["vis"])
    pos = P["plan_recognition.position_embeddings.weight"][:S]
    kept_total, n_total, masks = 0, 0, []
    for i in range(K):
        eng.forward_loss(mb, False, 1.0, 3.0, step=i)
        emb = eng.get_tensor("emb", B * S * 128).reshape(B, S, 128)
        x0 = eng.get_tensor("pr_x0", B * S * 128).reshape(B, S, 128)
        full = (emb + pos[None]).astype(np.float64)
        keep = x0 != 0
        # element-wise site (plan_recognition_net.py:111): kept elements carry (emb + pos) / (1 - p) exactly (fp32 arithmetic in both engines)
        sel = keep & (np.abs(full) > 1e-6)
        assert np.allclose(x0[sel], (full[sel] / (1.0 - p)), rtol=2e-6, atol=1e-7)
        dropped_nonzero = (~keep) & (np.abs(full) > 1e-6)
        kept_total += int(sel.sum()); n_total += int(sel.sum() + dropped_nonzero.sum())
        masks.append(keep)
        # (the attention kernels store the softmax weights BEFORE their dropout and re-derive the mask in the backward from the same counter
        # hash — the attention / GEMM-epilogue sites use the identical hash_uniform(seed, element) < p test and are covered end to end by the
        # mean-loss test below and by the finite-difference gradient test of tests/test_gpu_parity.py)
    rate = kept_total / n_total
    sig = np.sqrt(p * (1 - p) / n_total)
    assert abs(rate - (1 - p)) < 4 * sig, (rate, sig, n_total)
    # masks differ from step to step (about 1 - 2p(1-p) = 82 % agreement for independent masks) and repeat for the same step
    agree = np.mean(masks[0] == masks[1])
    assert 0.78 < agree < 0.86, agree
    eng.forward_loss(mb, False, 1.0, 3.0, step=0)
    x0 = eng.get_tensor("pr_x0", B * S * 128).reshape(B, S, 128)
    assert np.array_equal(x0 != 0, masks[0])
    eng.close()


def test_train_mode_mean_loss_matches_the_oracle_expectation():
    """64 train-mode forwards (64 device seeds = 64 steps) against 64 oracle forwards with numpy masks / draws.  The two means estimate the
    same expectation; their difference must vanish within the sampling error and within SURVEY's 1 % bound."""
    B, S, K, p = 4, 8, 64, 0.1
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    P = spec.init_all(dims, seed=21, ln_jitter=True)
    mb = synthetic.make_batch(B, 0, S, seed=3)["vis"]
    cache, ref = None, []
    for s in range(K):
        r, cache = O.train_mode_losses(P, dims, mb, np.random.default_rng(1000 + s), p, cache)
        ref.append((r["total"], r["kl"]))
    ref = np.array(ref)
    eng = _engine(dims, B, S, "fp32", dropout=p, seed=5)
    eng.load_numpy(P)
    dmb = to_dev(mb, inject_plan=False)
    got = []
    for i in range(K):
        l = eng.forward_loss(dmb, False, 1.0, 3.0, step=i)
        got.append((l["total_mod"], l["kl"]))
    got = np.array(got)
    eng.close()
    for c, name in ((0, "total"), (1, "kl")):
        m_ref, m_got = ref[:, c].mean(), got[:, c].mean()
        se = np.sqrt(ref[:, c].var() / K + got[:, c].var() / K)
        assert abs(m_got - m_ref) <= 0.01 * abs(m_ref), (name, m_got, m_ref)
        assert abs(m_got - m_ref) <= max(5.0 * se, 2e-5 * abs(m_ref)), (name, m_got, m_ref, se)
        # the spread of the stochastic loss is the same order on both sides (dropout really perturbs the device step)
        assert 0.3 < (got[:, c].std() + 1e-12) / (ref[:, c].std() + 1e-12) < 3.0, (name, got[:, c].std(), ref[:, c].std())
