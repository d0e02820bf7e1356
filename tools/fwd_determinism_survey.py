"""Which of the engine's named tensors differ between identical bf16 steps at the benchmark size?  (Forward tensors behind atomics-free kernels must not.)
python tools/fwd_determinism_survey.py [ingest] [model: hulc | mcil | mcil_gru] [dtype] [S]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synth_batch
from hulc_amd import spec
from hulc_amd.engine import StepEngine
ingest = sys.argv[1] if len(sys.argv) > 1 else "fp32"
model = sys.argv[2] if len(sys.argv) > 2 else "hulc"
dtype = sys.argv[3] if len(sys.argv) > 3 else "bf16"
S = int(sys.argv[4]) if len(sys.argv) > 4 else 32
B = 64 * 32 // S
dev = torch.device("cuda:0")
dims = spec.ModelDims(kind="mcil" if model.startswith("mcil") else "hulc", max_window=max(32, S), use_clip=False, rnn_type="gru" if model == "mcil_gru" else "rnn")
mb = synth_batch(B, S, dev, 7, False, ingest)
g = torch.Generator(device=dev); g.manual_seed(11)
mb["plan_idx"] = torch.randint(0, 32, (B, 32), device=dev, generator=g, dtype=torch.int32)
if dims.kind == "mcil":
    mb["plan_eps"] = torch.randn(B, 256, device=dev, generator=g)
eng = StepEngine(dims, B, S, dtype=dtype, device="cuda:0", dropout_p=0.0 if dims.kind == "mcil" else 0.1, seed=3, num_classes=dims.mix_classes)
if dtype == "fp16":
    eng.scaler_enable(init_scale=1024.0)
eng.load_numpy(spec.init_all(dims, seed=0, ln_jitter=True))
names = dict(birnn_h0=S * B * 2048, birnn_h0_rev=S * B * 2048, birnn_h1=S * B * 4096, plan=B * 256, emb=B * S * 128, s_a3=2048 * 441 * 64, seq_feat=B * 4096, pr_logits=B * 1024, dec_h0=S * B * 2048, dec_h1=S * B * 2048, heads=S * B * 192, a_tcp=S * B * 7,
             dheads=S * B * 192, dec_dz1=S * B * 2048, dec_dz0=S * B * 2048, demb=B * S * 128, dplan=B * 1024, dact3=2048 * 49 * 64, dact2=2048 * 81 * 64, dact1=2048 * 400 * 32)
runs = []
for _ in range(3):
    eng.zero_grads(); eng.forward_loss(mb, False, 1.0, 3.0, step=2, sync_losses=False); eng.backward(); torch.cuda.synchronize()
    r = {}
    for n, k in names.items():
        try: r[n] = eng.get_tensor(n, k).copy()
        except Exception as e: r[n] = None
    r["grads"] = eng.flat_grads.cpu().numpy().copy()
    runs.append(r)
for n in list(names) + ["grads"]:
    if runs[0][n] is None: print(f"{n:10s} unavailable"); continue
    d = [int((runs[0][n] != runs[i][n]).sum()) for i in (1, 2)]
    rel = [float(np.linalg.norm(runs[0][n].astype(np.float64) - runs[i][n]) / max(np.linalg.norm(runs[0][n].astype(np.float64)), 1e-30)) for i in (1, 2)]
    print(f"{n:10s} differing elements {d}  rel-L2 {rel[0]:.2e} {rel[1]:.2e}  of {runs[0][n].size}")
