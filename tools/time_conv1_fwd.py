"""conv1 forward (8x8 stride 4, 3 -> 32) on 2048 frames at the fp32 NCHW boundary: whole kernel against its ablations
(dbg bit 1: no MFMA / epilogue, bit 2: no band staging) for both cameras.   python tools/time_conv1_fwd.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
Nf = 2048
def run(img, w, bias, out, IMH, OUTH, dbg, mode=4, shifts=None):
    args = (mode, img.data_ptr(), w.data_ptr(), bias.data_ptr(), shifts.data_ptr() if shifts is not None else None, out.data_ptr(), Nf, IMH, OUTH, dbg, None)
    for _ in range(3): L.check(lib.hulc_k_conv_tile(*args))
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): lib.hulc_k_conv_tile(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
for cam, IH in (("static", 200), ("gripper", 84)):
    OH = (IH - 8) // 4 + 1
    x = torch.randn(Nf, 3, IH, IH, device="cuda"); w = (torch.randn(32, 192, device="cuda") * 0.05).to(torch.bfloat16); b = torch.zeros(32, device="cuda")
    o = torch.zeros(Nf, OH, OH, 32, device="cuda", dtype=torch.bfloat16)
    mb = Nf * (3 * IH * IH * 4 + OH * OH * 32 * 2) / 1e6
    r = {k: run(x, w, b, o, IH, OH, d) for k, d in (("full", 0), ("no-compute", 2), ("no-staging", 4), ("nothing", 6))}
    print(cam, {k: round(v, 1) for k, v in r.items()}, f"{mb / r['full']:.2f} TB/s of {mb:.0f} MB algorithmic")

# the uint8 HWC boundary (modes 5 / 6 of hulc_k_conv_tile: without / with RandomShiftsAug shifts)
for cam, IH in (("static", 200), ("gripper", 84)):
    OH = (IH - 8) // 4 + 1
    x = torch.randint(0, 256, (Nf, IH, IH, 3), device="cuda", dtype=torch.int32).to(torch.uint8); w = (torch.randn(32, 192, device="cuda") * 0.05).to(torch.bfloat16); b = torch.zeros(32, device="cuda")
    o = torch.zeros(Nf, OH, OH, 32, device="cuda", dtype=torch.bfloat16)
    pad = 10 if IH >= 100 else 4
    sh = torch.randint(0, 2 * pad + 1, (Nf, 2), device="cuda", dtype=torch.int32)
    mb = Nf * (3 * IH * IH + OH * OH * 32 * 2) / 1e6
    for mode, shv, name in ((5, None, "u8"), (6, sh, "u8+shift")):
        r = {k: run(x, w, b, o, IH, OH, d, mode, shv) for k, d in (("full", 0), ("no-compute", 2), ("no-staging", 4), ("nothing", 6))}
        print(cam, name, {k: round(v, 1) for k, v in r.items()}, f"{mb / r['full']:.2f} TB/s of {mb:.0f} MB algorithmic")
