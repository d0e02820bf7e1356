import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from bench import synth_batch
from hulc_amd import spec
from hulc_amd.engine import StepEngine
dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
dev = torch.device("cuda:0")
B, S = 8, 8
mb = synth_batch(B, S, dev, 1, False)
g = torch.Generator(device=dev); g.manual_seed(5)
mb["plan_idx"] = torch.randint(0, 32, (B, 32), device=dev, generator=g, dtype=torch.int32)
out = {}
for fuse in (0, 1):
    eng = StepEngine(dims, B, S, dtype="bf16", device="cuda:0", dropout_p=0.1, seed=1, num_classes=dims.mix_classes)
    eng.set_option("adam_fused_transposes", fuse)
    eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
    eng.zero_grads(); eng.forward_loss(mb, False, 1.0, 3.0, step=0); eng.backward()
    g0 = eng.flat_grads.clone()
    eng.adam_step(lr=1e-3)
    torch.cuda.synchronize()
    p1 = eng.flat_params.clone(); m1 = eng.adam_m.clone(); v1 = eng.adam_v.clone()
    eng.zero_grads(); l = eng.forward_loss(mb, False, 1.0, 3.0, step=1); eng.backward()
    torch.cuda.synchronize()
    out[fuse] = dict(g0=g0, p=p1, m=m1, v=v1, g1=eng.flat_grads.clone(), l=l["total_mod"], views=eng.views)
    lay = eng.layout
    eng.close()
a, b = out[0], out[1]
print("loss", a["l"], b["l"])
for k in ("g0", "p", "m", "v", "g1"):
    d = (a[k] != b[k])
    print(k, "mismatching elements", int(d.sum()), "max abs diff", float((a[k] - b[k]).abs().max()))
d = (a["p"] != b["p"])
if d.any():
    idx = torch.nonzero(d).flatten()
    print("first mismatch idx", idx[:10].tolist())
    names = sorted(lay.items(), key=lambda kv: kv[1][0])
    for n, (off, shape) in names:
        sz = int(np.prod(shape)) if len(shape) else 1
        c = int(d[off:off + sz].sum())
        if c: print(n, shape, c, "of", sz, float((a["p"][off:off+sz]-b["p"][off:off+sz]).abs().max()))
