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
H1, H2, H3 = (49, 23, 21) if os.environ.get("SHAPES") == "static" else (20, 9, 7)      # SHAPES=static: the static camera's maps
cases = [(f"fwd3 {H2}->{H3}", 0, (Nf, H2, H2, 64), (64, 576), b64, None, (Nf, H3, H3, 64), H2, H3, 1),
         (f"fwd2 {H1}->{H2} (+bits)", 7, (Nf, H1, H1, 32), (64, 512), b64, i32(Nf, H2, H2, 2), (Nf, H2, H2, 64), H1, H2, 1),
         (f"dgrad3 {H3}->{H2} (bits)", 8, (Nf, H3, H3, 64), (64, 576), None, i32(Nf, H2, H2, 2), (Nf, H2, H2, 64), H3, H2, 32),
         (f"dgrad2 {H2}->{H1} (bits)", 9, (Nf, H2, H2, 64), (128, 256), None, i32(Nf, H1, H1, 1), (Nf, H1, H1, 32), H2, H1, 32)]
res = []
for name, mode, ishape, wshape, bias, mask, oshape, IMH, OUTH, dbg in cases:
    img = torch.randn(*ishape, device="cuda").to(torch.bfloat16); w = (torch.randn(*wshape, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.zeros(*oshape, device="cuda", dtype=torch.bfloat16)
    res.append(f"{name}: {run(mode, img, w, bias, mask, out, IMH, OUTH, dbg):.1f} us")
print(f"HULC_CT_FPB={os.environ.get('HULC_CT_FPB', 'auto')} HULC_CT_NW={os.environ.get('HULC_CT_NW', 'default (8 fwd / 16 dgrad)')}:  " + "   ".join(res))
