import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from hulc_amd import spec
from hulc_amd.engine import StepEngine
import bench
dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=True)
eng = StepEngine(dims, 32, 32, dtype="bf16", device="cuda:0", dropout_p=0.1, seed=1)
eng.load_numpy(spec.init_all(dims, seed=0))
dev = torch.device("cuda:0")
mods = [("vis", bench.synth_batch(32, 32, dev, 1, False, "u8")), ("lang", bench.synth_batch(32, 32, dev, 2, True, "u8"))]
t0 = time.time(); hist = []
for i in range(1500):
    eng.zero_grads()
    for name, mb in mods:
        l = eng.forward_loss(mb, name == "lang", 0.5, 3.0, step=i, sync_losses=(i % 100 == 0))
        eng.backward()
    eng.adam_step(lr=2e-4)
    if i % 100 == 0:
        hist.append((i, round(l["total_mod"], 4), round(l["clip"], 4)))
torch.cuda.synchronize()
print("steps/s", 1500 / (time.time() - t0)); print(hist)
assert all(np.isfinite(h[1]) for h in hist) and hist[-1][1] < hist[0][1]
assert torch.isfinite(eng.flat_params).all()
print("SOAK OK")
