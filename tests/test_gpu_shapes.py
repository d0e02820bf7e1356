"""GPU (-m gpu): ragged and extreme window shapes through the C-ABI against the oracle — one window of one frame, odd batch / window
lengths, the longest supported window (S = 64, BASELINE config 5's length), in the bf16 bench mode as well (whose kernels have the
shape-dependent tiling: 32-row skinny blocks, band splits, 64-wide attention)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import hulc_oracle as O  # noqa: E402
from hulc_amd import spec  # noqa: E402
from hulc_amd.utils import synthetic  # noqa: E402
from test_gpu_parity import _engine, grads_np, run_step  # noqa: E402


def _case(kind, Bv, Bl, S, seed, use_clip):
    dims = spec.ModelDims(kind=kind, max_window=64, use_clip=use_clip)
    P = spec.init_all(dims, seed=seed, ln_jitter=True)
    batch = synthetic.make_batch(Bv, Bl, S, seed=seed, edge_frac=0.1, aux_mask="all")
    if kind == "mcil":
        for mb in batch.values():
            mb["plan_eps"] = np.random.default_rng(seed).standard_normal((mb["actions"].shape[0], 256)).astype(np.float32)
    return dims, P, batch


@pytest.mark.parametrize("kind,Bv,Bl,S,use_clip", [("hulc", 1, 0, 1, False), ("hulc", 5, 3, 7, True), ("gcbc", 3, 0, 5, False), ("mcil", 1, 2, 3, False),
                                                   ("hulc", 1, 0, 64, False), ("hulc", 2, 1, 33, True)])
def test_ragged_shapes_match_oracle(kind, Bv, Bl, S, use_clip):
    dims, P, batch = _case(kind, Bv, Bl, S, 100 + S, use_clip)
    losses_o, G = O.training_step(P, dims, batch)
    B = max(Bv, Bl)
    for dtype, tol_loss, tol_cos in (("fp32", 2e-5, 0.99999), ("bf16", 5e-3, 0.99)):
        eng = _engine(dims, B, S, dtype, num_classes=dims.mix_classes)
        eng.load_numpy(P)
        tot, _ = run_step(eng, batch)
        ref = float(losses_o["total"])
        assert abs(tot - ref) <= tol_loss * abs(ref), (dtype, tot, ref)
        Gg = grads_np(eng)
        a = np.concatenate([Gg[n].reshape(-1) for n in G]).astype(np.float64)
        b = np.concatenate([G[n].reshape(-1) for n in G]).astype(np.float64)
        cos = float(a @ b / (np.linalg.norm(a) * np.linalg.norm(b)))
        assert cos > tol_cos, (dtype, cos)
        assert abs(np.linalg.norm(a) / np.linalg.norm(b) - 1) < (1e-4 if dtype == "fp32" else 0.05)
        eng.close()


def test_window_longer_than_64_is_rejected():
    dims, P, batch = _case("hulc", 1, 0, 4, 3, False)
    with pytest.raises(RuntimeError):
        _engine(dims, 1, 65, "bf16")
