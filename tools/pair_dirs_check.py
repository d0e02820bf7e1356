"""One full-size (B=64, S=32, bf16) mcil / mcil_gru step; prints loss and gradient norms and saves the flat gradient.  Run once with
HULC_PAIR_DIRS=0 and once with =1 (the switch is read once per process) and compare the two files: the paired-direction launches must
reproduce the sequential recurrences (measured: rel. difference 1.5e-8 = atomics order).  usage: pair_dirs_check.py mcil|mcil_gru out.npy"""
import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hulc_amd import spec
from hulc_amd.engine import StepEngine
import bench
kind = sys.argv[1]
gru = kind == "mcil_gru"
dims = spec.ModelDims(kind="mcil", max_window=32, use_clip=False, rnn_type="gru" if gru else "rnn")
eng = StepEngine(dims, 64, 32, dtype="bf16", device="cuda:0", dropout_p=0.0, seed=1)
eng.load_numpy(spec.init_all(dims, seed=0))
dev = torch.device("cuda:0")
mb = bench.synth_batch(64, 32, dev, 1, False, "fp32")
eng.zero_grads()
l = eng.forward_loss(mb, False, 0.5, 3.0, step=0, sync_losses=True)
eng.backward()
g = eng.flat_grads.double()
print(kind, "PAIR", os.environ.get("HULC_PAIR_DIRS"), "loss", l["total_mod"], "gnorm", float(g.norm()), "gsum", float(g.sum()))
np.save(sys.argv[2], eng.flat_grads.cpu().numpy())
