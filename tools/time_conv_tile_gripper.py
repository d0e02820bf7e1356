"""Time the four conv_tile kernels on the gripper camera's shapes (2048 frames) in their production form (bitmask modes 7/8/9, mode 0 for conv3
forward).  The frames-per-band choice is read once per process: run as `HULC_CT_FPB=n python tools/time_conv_tile_gripper.py` (1 = one frame per
band, unset = the launch's own cost model)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
Nf = 2048
def run(mode, img, w, bias, mask, out, IMH, OUTH, dbg):
    args = (mode, img.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, mask.data_ptr() if mask is not None else None, out.data_ptr(), Nf, IMH, OUTH, dbg, None)
    for _ in range(3): L.check(lib.hulc_k_conv_tile(*args))
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): lib.hulc_k_conv_tile(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
b64 = torch.zeros(64, device="cuda")
i32 = lambda *s: torch.randint(-2**31, 2**31 - 1, s, device="cuda", dtype=torch.int32)
cases = [("fwd3 9->7", 0, (Nf, 9, 9, 64), (64, 576), b64, None, (Nf, 7, 7, 64), 9, 7, 1),
         ("fwd2 20->9 (+bits)", 7, (Nf, 20, 20, 32), (64, 512), b64, i32(Nf, 9, 9, 2), (Nf, 9, 9, 64), 20, 9, 1),
         ("dgrad3 7->9 (bits)", 8, (Nf, 7, 7, 64), (64, 576), None, i32(Nf, 9, 9, 2), (Nf, 9, 9, 64), 7, 9, 32),
         ("dgrad2 9->20 (bits)", 9, (Nf, 9, 9, 64), (128, 256), None, i32(Nf, 20, 20, 1), (Nf, 20, 20, 32), 9, 20, 32)]
res = []
for name, mode, ishape, wshape, bias, mask, oshape, IMH, OUTH, dbg in cases:
    img = torch.randn(*ishape, device="cuda").to(torch.bfloat16); w = (torch.randn(*wshape, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.zeros(*oshape, device="cuda", dtype=torch.bfloat16)
    res.append(f"{name}: {run(mode, img, w, bias, mask, out, IMH, OUTH, dbg):.1f} us")
print(f"HULC_CT_FPB={os.environ.get('HULC_CT_FPB', 'auto')}:  " + "   ".join(res))
