"""conv1 forward, both cameras, fp32 and uint8 boundary: round 6's conflict-free (c, kh) -> lane-group map of the fragment reads (conv_tile.h conv1_ck)
against round 5's (dbg bit 6), alternating in one process; + a bit-level check that the two maps give the same sums up to association.
python tools/time_conv1_kmap.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
Nf = 2048
def run(img, w, bias, out, IMH, OUTH, dbg, mode=4, shifts=None, n=10):
    args = (mode, img.data_ptr(), w.data_ptr(), bias.data_ptr(), shifts.data_ptr() if shifts is not None else None, out.data_ptr(), Nf, IMH, OUTH, dbg, None)
    for _ in range(3): L.check(lib.hulc_k_conv_tile(*args))
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): lib.hulc_k_conv_tile(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cam, IH in (("static", 200), ("gripper", 84)):
    OH = (IH - 8) // 4 + 1
    w = (torch.randn(32, 192, device="cuda") * 0.05).to(torch.bfloat16); b = torch.randn(32, device="cuda") * 0.1
    pad = 10 if IH >= 100 else 4
    sh = torch.randint(0, 2 * pad + 1, (Nf, 2), device="cuda", dtype=torch.int32)
    for name, mode, x, shv in (("fp32", 4, torch.randn(Nf, 3, IH, IH, device="cuda"), None),
                               ("u8", 5, torch.randint(0, 256, (Nf, IH, IH, 3), device="cuda", dtype=torch.int32).to(torch.uint8), None),
                               ("u8+shift", 6, torch.randint(0, 256, (Nf, IH, IH, 3), device="cuda", dtype=torch.int32).to(torch.uint8), sh)):
        o0 = torch.zeros(Nf, OH, OH, 32, device="cuda", dtype=torch.bfloat16); o1 = torch.zeros_like(o0)
        t = {"new": [], "old": []}
        for rep in range(3):
            t["new"].append(run(x, w, b, o1, IH, OH, 0, mode, shv)); t["old"].append(run(x, w, b, o0, IH, OH, 64, mode, shv))
        d = (o0.float() - o1.float()).abs().max().item(); ne = (o0 != o1).float().mean().item()
        print(f"{cam:8s} {name:9s} new {min(t['new']):7.1f} us  old {min(t['old']):7.1f} us   all: {[round(v, 1) for v in t['new']]} vs {[round(v, 1) for v in t['old']]}   max |diff| {d:.3g}, differing outputs {ne:.2e}")
