"""conv1 forward from uint8 frames: the LDS-DMA kernel (raw rows -> LDS -> conversion pass; conv1_fwd_u8dma_kernel) against the register-staged kernel with the conversion
from 16-byte windows (conv1_stage_band regconv, dbg bit 8), alternating in one process; outputs compared bit for bit.   python tools/time_conv1_u8reg.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
Nf = 2048
def run(img, w, bias, out, IMH, OUTH, dbg, mode, shifts, n=10):
    args = (mode, img.data_ptr(), w.data_ptr(), bias.data_ptr(), shifts.data_ptr() if shifts is not None else None, out.data_ptr(), Nf, IMH, OUTH, dbg, None)
    for _ in range(3): L.check(lib.hulc_k_conv_tile(*args))
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): lib.hulc_k_conv_tile(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for cam, IH in (("static", 200), ("gripper", 84)):
    OH = (IH - 8) // 4 + 1
    w = (torch.randn(32, 192, device="cuda") * 0.05).to(torch.bfloat16); b = torch.randn(32, device="cuda") * 0.1
    x = torch.randint(0, 256, (Nf, IH, IH, 3), device="cuda", dtype=torch.int32).to(torch.uint8)
    pad = 10 if IH >= 100 else 4
    sh = torch.randint(0, 2 * pad + 1, (Nf, 2), device="cuda", dtype=torch.int32)
    sh[0] = torch.tensor([0, 2 * pad]); sh[1] = torch.tensor([2 * pad, 0]); sh[Nf - 1] = torch.tensor([2 * pad, 2 * pad])       # extreme shifts, the buffer's last frame included
    for name, mode, shv in (("u8", 5, None), ("u8+shift", 6, sh)):
        o0 = torch.zeros(Nf, OH, OH, 32, device="cuda", dtype=torch.bfloat16); o1 = torch.zeros_like(o0)
        t = {"dma": [], "reg": []}
        for rep in range(4):
            t["dma"].append(run(x, w, b, o0, IH, OH, 0, mode, shv)); t["reg"].append(run(x, w, b, o1, IH, OH, 256, mode, shv))
        print(f"{cam:8s} {name:9s} LDS-DMA + conversion pass {min(t['dma']):7.1f} us   windows from registers {min(t['reg']):7.1f} us   outputs equal: {bool(torch.equal(o0, o1))}")
