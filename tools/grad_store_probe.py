"""Which gradient tensors does the first backward after hulc_zero_grads STORE (no need to zero them) and which does it ACCUMULATE into?
The buffer is NaN-filled behind zero_grads' back: a tensor that comes out finite was stored, one that is NaN was added to.
    python tools/grad_store_probe.py            (GPU)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from hulc_amd import spec
from hulc_amd.engine import StepEngine

def probe(kind, rnn_type, dtype, lang, B=16, S=32):
    mcil = kind == "mcil"
    dims = spec.ModelDims(kind=kind, max_window=32, use_clip=bool(lang) and not mcil, rnn_type=rnn_type)
    eng = StepEngine(dims, B, S, dtype=dtype, device="cuda:0", dropout_p=0.0 if mcil else 0.1, seed=1, num_classes=dims.mix_classes)
    eng.load_numpy(spec.init_all(dims, seed=0))
    dev = torch.device("cuda:0")
    mods = [synth_batch(B // 2 if lang else B, S, dev, 1, False)]
    if lang: mods.append(synth_batch(B // 2, S, dev, 2, True))
    eng.zero_grads()
    eng.flat_grads.fill_(float("nan"))
    if lang: eng.forward_loss_pair(mods[0], mods[1], 0.5, 3.0, step=0)
    else: eng.forward_loss(mods[0], False, 1.0, 3.0, step=0)
    eng.backward(); torch.cuda.synchronize()
    added, stored = [], []
    for n, t in eng.views(eng.flat_grads).items():
        (added if not bool(torch.isfinite(t).all()) else stored).append((n, t.numel()))
    tot = sum(k for _, k in added) + sum(k for _, k in stored)
    print(f"== {kind}/{rnn_type} {dtype} lang={lang}: accumulated-into {len(added)} tensors, {sum(k for _, k in added)/1e6:.2f} M of {tot/1e6:.1f} M elements; largest accumulated:",
          sorted(added, key=lambda x: -x[1])[:12])
    print("   stored tensors < 4096 elements:", [n for n, k in stored if k < 4096][:40])
    eng.close()

for args in (("hulc", "rnn", "bf16", 0), ("hulc", "rnn", "bf16", 1), ("hulc", "rnn", "fp16", 0), ("gcbc", "rnn", "bf16", 1), ("mcil", "rnn", "bf16", 0), ("mcil", "gru", "bf16", 1), ("hulc", "rnn", "fp32", 0)):
    probe(*args)
