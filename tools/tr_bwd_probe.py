"""Debug helper (experiment build): dumps the plan-recognition transformer backward intermediates (tr_dx, tr_dy1, tr_bd*, tr_bb*) of one synthetic
step to an .npz; run once with HULC_TR_ATTN_BWD=0 and once with 1, then tools/tr_bwd_cmp.py a.npz b.npz.   python tools/tr_bwd_probe.py out.npz B S"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, d))
from hulc_amd import spec
from hulc_amd.utils import synthetic
from hulc_amd.engine import StepEngine
from test_gpu_parity import to_dev
B, S = int(sys.argv[2]), int(sys.argv[3])
dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=True)
P = spec.init_all(dims, seed=13, ln_jitter=True)
mb = synthetic.make_batch(B, B, S, seed=31)["lang"]
eng = StepEngine(dims, B, S, dtype="bf16", device="cuda:0", dropout_p=0.0, seed=77)
eng.load_numpy(P); eng.zero_grads()
eng.forward_loss(to_dev(mb), True, 1.0, 3.0, step=5); eng.backward(); torch.cuda.synchronize()
out = {}
for n, k in (("tr_dx", 128), ("tr_dy1", 128), ("tr_bd0", 128), ("tr_bd1", 128), ("tr_bb0", 384), ("tr_bb1", 384)):
    t = eng.get_tensor(n, B * S * k)
    out[n] = np.asarray(t, np.float32).reshape(B, S, k)
np.savez(sys.argv[1], **out)
