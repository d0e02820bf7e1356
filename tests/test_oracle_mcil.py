"""CPU: the oracle's mcil variant (BiRNN plan recognition, continuous latent, 7-dim mixture decoder) against
fixtures of the unmodified reference in its conf/model/mcil.yaml configuration (tools/gen_golden_mcil.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import hulc_oracle as O  # noqa: E402
from golden_util import MCIL_CASES, check_grads, load_mcil_case  # noqa: E402


@pytest.mark.parametrize("name", list(MCIL_CASES))
def test_mcil_step_matches_reference(name):
    dims, P, batch, fx = load_mcil_case(name)
    losses, G, caches = O.training_step(P, dims, batch, keep_cache=True)
    assert abs(float(losses["total"]) - float(fx["loss_total"])) <= 2e-5 * abs(float(fx["loss_total"]))
    assert abs(float(losses["kl"]) - float(fx["log/train/kl_loss"])) <= 2e-5 * abs(float(fx["log/train/kl_loss"])) + 1e-8
    for sc in batch:
        c = caches[sc]
        assert np.abs(c["plan"] - fx[f"plan_{sc}"]).max() <= 1e-5
        assert np.abs(c["seq_feat"] - fx[f"seq_feat_{sc}"]).max() <= 2e-5
        assert np.abs(c["emb"] - fx[f"emb_{sc}"]).max() <= 2e-5 * np.abs(fx[f"emb_{sc}"]).max()
    bad = check_grads(G, fx, tol_l2=2e-4, tol_norm=2e-4, label=name)      # fixture gradients = fp64 evaluation of the reference (generator docstring)
    assert not bad, bad[:6]
