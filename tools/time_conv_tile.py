import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
Nf = 2048
def run(mode, img, w, bias, mask, out, IMH, OUTH, dbg):
    for _ in range(2): L.check(lib.hulc_k_conv_tile(mode, img.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, mask.data_ptr() if mask is not None else None, out.data_ptr(), Nf, IMH, OUTH, dbg, None))
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(5): lib.hulc_k_conv_tile(mode, img.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, mask.data_ptr() if mask is not None else None, out.data_ptr(), Nf, IMH, OUTH, dbg, None)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 * 1e3
b64 = torch.zeros(64, device="cuda")
cases = [("fwd3", 0, (Nf, 23, 23, 64), (64, 576), b64, None, (Nf, 21, 21, 64), 23, 21),
         ("fwd2", 1, (Nf, 49, 49, 32), (64, 512), b64, None, (Nf, 23, 23, 64), 49, 23),
         ("dgrad3", 2, (Nf, 21, 21, 64), (64, 576), None, (Nf, 23, 23, 64), (Nf, 23, 23, 64), 21, 23),
         ("dgrad2", 3, (Nf, 23, 23, 64), (128, 256), None, (Nf, 49, 49, 32), (Nf, 49, 49, 32), 23, 49)]
for name, mode, ishape, wshape, bias, mshape, oshape, IMH, OUTH in cases:
    img = torch.randn(*ishape, device="cuda").to(torch.bfloat16); w = (torch.randn(*wshape, device="cuda") * 0.05).to(torch.bfloat16)
    mask = torch.randn(*mshape, device="cuda").to(torch.bfloat16) if mshape else None
    out = torch.zeros(*oshape, device="cuda", dtype=torch.bfloat16)
    t = {k: run(mode, img, w, bias, mask, out, IMH, OUTH, d) for k, d in (("full", 1), ("no-compute", 3), ("no-loads", 5), ("neither", 7), ("no-epi", 9), ("no-mfma", 17), ("no-epi-no-mfma", 25), ("noload-noepi", 13))}
    print(name, {k: round(v) for k, v in t.items()})
