"""Run-to-run reproducibility of one training step's gradients on the bf16 engine (hulc_tiny fixture inputs, injected plan sample, no dropout):
which tensors differ between evaluations of the same step, and by how much (the check of tests/test_gpu_fp16.py::test_library_rccl_allreduce_world1_and_bucket_plan).
python tools/det_probe.py [repeats] [allreduce]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ("", "tests", "oracle"):          # test helpers only (a tools/ script, not the product)
    sys.path.insert(0, os.path.join(ROOT, d))
from golden_util import load_case
from test_gpu_parity import to_dev, _engine
dims, P, batch, fx = load_case("hulc_tiny")
eng = _engine(dims, 2, 4, "bf16")
eng.load_numpy(P)
ar = len(sys.argv) > 2
if ar:
    eng.comm_init(eng.comm_unique_id(), 0, 1)
def step(last_ar=False):
    eng.zero_grads()
    scopes = list(batch)
    for i, sc in enumerate(scopes):
        eng.forward_loss(to_dev(batch[sc]), "lang" in sc, 1.0 / len(scopes), 3.0, step=0)
        if last_ar and i == len(scopes) - 1:
            eng.backward_allreduce("fp32")
        else:
            eng.backward()
    torch.cuda.synchronize()
    return {n: t.detach().clone() for n, t in eng.views(eng.flat_grads).items()}
ref = step()
for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    g = step(ar)
    bad = []
    for n in ref:
        d = (g[n] - ref[n]).abs()
        viol = (d > 1e-3 * ref[n].abs() + 1e-6).sum().item()
        if viol:
            i = int((d - 1e-3 * ref[n].abs()).argmax())
            bad.append((viol, f"{d.max().item():.2e}", f"{ref[n].abs().max().item():.2e}", n, i, float(ref[n].reshape(-1)[i]), float(g[n].reshape(-1)[i])))
    print(f"run {r}: allclose violations:", sorted(bad, reverse=True)[:4])
