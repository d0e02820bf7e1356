"""The chunked / pooled oracle evaluation (tests/oracle_pool.py) that the full-size GPU parity tests rely on, checked on CPU at a tiny size:
chunks of windows evaluated by worker processes — the language modality with its CLIP term in one piece — must reproduce ONE call of
oracle.training_step on the whole batch (losses and every gradient tensor)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


@pytest.mark.parametrize("case", [dict(seed=5, kind="hulc", max_window=32, B=4, B_lang=4, S=3, use_clip=True, mode=None),
                                  dict(seed=6, kind="mcil", rnn_type="rnn", max_window=32, B=4, S=3, mode=None)], ids=["hulc_pair_clip", "mcil_rnn"])
def test_pooled_chunks_equal_one_training_step(case):
    import hulc_oracle as O
    import oracle_pool
    from hulc_amd import spec
    dims = oracle_pool.case_dims(case)
    P = spec.init_all(dims, seed=case["seed"], ln_jitter=True)
    batch = oracle_pool.case_batch(case)
    G, losses, embs = oracle_pool.oracle_case(case, CH=2, workers=2, P=P if case["kind"] == "mcil" else None, batch=batch if case["kind"] == "mcil" else None)   # both hand-over forms
    L, Gr, caches = O.training_step(P, dims, batch, keep_cache=True)
    tot = sum(losses[s]["kl"] + losses[s]["action"] for s in losses) / len(losses) + 3.0 * sum(losses[s]["clip"] for s in losses)
    assert abs(tot - float(L["total"])) <= 2e-6 * abs(float(L["total"])), (tot, L["total"])
    for s in losses:
        assert np.allclose(embs[s], caches[s]["emb"], rtol=0, atol=1e-6)
        assert abs(losses[s]["kl"] - float(L[f"kl_{s}"])) <= 1e-5 * abs(float(L[f"kl_{s}"])) + 1e-9
    for n in G:
        ref = np.linalg.norm(Gr[n])
        if ref > 1e-8:
            assert np.linalg.norm(G[n] - Gr[n]) / ref < 2e-4, (n, np.linalg.norm(G[n] - Gr[n]) / ref)
