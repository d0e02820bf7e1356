"""Which parameter gradients differ between two identical fp32-mode steps at full size (B=64, S=32)?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from hulc_amd import spec
from hulc_amd.engine import StepEngine
B, S = int(os.environ.get("B", 64)), 32
dev = torch.device("cuda:0")
dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
mb = synth_batch(B, S, dev, seed=7)
g = torch.Generator(device=dev); g.manual_seed(11)
mb["plan_idx"] = torch.randint(0, 32, (B, 32), device=dev, generator=g, dtype=torch.int32)
eng = StepEngine(dims, B, S, dtype=os.environ.get("DT", "fp32"), device="cuda:0", dropout_p=0.0, seed=3)
eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
gs = []
for _ in range(2):
    eng.zero_grads(); eng.forward_loss(mb, False, 1.0, 3.0, step=0); eng.backward(); torch.cuda.synchronize(); gs.append(eng.flat_grads.clone())
for n, (off, shape) in eng.layout.items():
    k = int(torch.tensor(shape).prod()) if len(shape) else 1
    a, b = gs[0][off:off + k], gs[1][off:off + k]
    if not torch.equal(a, b):
        print(f"{n:60s} max|diff| {(a - b).abs().max().item():.3e}  |g| {a.abs().max().item():.3e}")
