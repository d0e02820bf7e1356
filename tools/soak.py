"""Long-run soak on one GPU: 1500 optimizer steps each of (a) HULC vis + lang + CLIP, uint8 ingest, bf16, one pass per modality, (b) the same as one
paired pass in fp16 with the on-device GradScaler, (c) model=mcil (dual persistent recurrences) in bf16 — on a FIXED batch, so the loss must fall;
parameters finite, no persistent-recurrence fallback, scaler sane.   python tools/soak.py > profiles/rNN_soak.txt"""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from hulc_amd import spec
from hulc_amd.engine import StepEngine
import bench
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
for tag, kind, dtype, paired in (("hulc bf16 u8 two passes", "hulc", "bf16", False), ("hulc fp16 paired", "hulc", "fp16", True), ("mcil bf16", "mcil", "bf16", False)):
    mcil = kind == "mcil"
    dims = spec.ModelDims(kind=kind, max_window=32, use_clip=not mcil)
    eng = StepEngine(dims, 64 if paired else 32, 32, dtype=dtype, device="cuda:0", dropout_p=0.0 if mcil else 0.1, seed=1, num_classes=dims.mix_classes)
    eng.load_numpy(spec.init_all(dims, seed=0))
    ing = "u8" if not paired else "fp32"
    mods = [("vis", bench.synth_batch(32, 32, dev, 1, False, ing))] + ([] if mcil else [("lang", bench.synth_batch(32, 32, dev, 2, True, ing))])
    t0 = time.time(); hist = []
    for i in range(N):
        eng.zero_grads()
        sync = i % 100 == 0 or i == N - 1
        if paired:
            lv, ll = eng.forward_loss_pair(mods[0][1], mods[1][1], 0.5, 3.0, step=i) if sync else (None, None)
            if not sync:
                eng.forward_loss_pair(mods[0][1], mods[1][1], 0.5, 3.0, step=i, sync_losses=False)
            eng.backward()
            l = None if lv is None else dict(total_mod=lv["total_mod"] + ll["total_mod"], clip=ll["clip"])
        else:
            for name, mb in mods:
                l = eng.forward_loss(mb, name == "lang", 1.0 / len(mods), 3.0, step=i, sync_losses=sync)
                eng.backward()
        eng.adam_step(lr=2e-4)
        if sync:
            hist.append((i, round(l["total_mod"], 4), round(l["clip"], 4)))
    torch.cuda.synchronize()
    fb = eng.get_option("persistent_rnn_fallbacks")
    print(f"[{tag}] {N} steps, {N / (time.time() - t0):.0f} steps/s, persistent_rnn {eng.get_option('persistent_rnn')} fallbacks {fb}" + (f", scaler {eng.scaler_state()}" if dtype == "fp16" else ""))
    print("   (step, total loss, clip):", hist)
    assert all(np.isfinite(h[1]) for h in hist) and hist[-1][1] < hist[0][1], hist
    assert torch.isfinite(eng.flat_params).all() and fb == 0 and eng.get_option("persistent_rnn") == 1
    eng.close()
print("SOAK OK")
