import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
Nf, IH = 2048, 200; OH = 49
X = torch.randn(Nf, 3, IH, IH, device="cuda"); W = torch.randn(32, 192, device="cuda").to(torch.bfloat16); b = torch.zeros(32, device="cuda")
out = torch.zeros(Nf, OH, OH, 32, device="cuda", dtype=torch.bfloat16)
for name, dbg in (("full", 1), ("no-compute", 3), ("no-staging", 5), ("neither", 7)):
    for _ in range(2): L.check(lib.hulc_k_conv_tile(4, X.data_ptr(), W.data_ptr(), b.data_ptr(), None, out.data_ptr(), Nf, IH, OH, dbg, None))
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): L.check(lib.hulc_k_conv_tile(4, X.data_ptr(), W.data_ptr(), b.data_ptr(), None, out.data_ptr(), Nf, IH, OH, dbg, None))
    e1.record(); torch.cuda.synchronize()
    print(f"conv1 fwd static {name}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us")
Xu = torch.randint(0, 256, (Nf, IH, IH, 3), device="cuda", dtype=torch.int32).to(torch.uint8)
sh = torch.randint(0, 21, (Nf, 2), device="cuda", dtype=torch.int32)
for mode, mname, msk in ((5, "u8", None), (6, "u8+shift", sh)):
    for name, dbg in (("full", 1), ("no-compute", 3), ("no-staging", 5)):
        for _ in range(2): L.check(lib.hulc_k_conv_tile(mode, Xu.data_ptr(), W.data_ptr(), b.data_ptr(), msk.data_ptr() if msk is not None else None, out.data_ptr(), Nf, IH, OH, dbg, None))
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): L.check(lib.hulc_k_conv_tile(mode, Xu.data_ptr(), W.data_ptr(), b.data_ptr(), msk.data_ptr() if msk is not None else None, out.data_ptr(), Nf, IH, OH, dbg, None))
        e1.record(); torch.cuda.synchronize()
        print(f"conv1 fwd static {mname} {name}: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us")
# reference: plain copy bandwidth
y = torch.empty_like(X)
for _ in range(2): y.copy_(X)
torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(5): y.copy_(X)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 5 * 1e-3
print(f"torch copy 983 MB: {t*1e6:.0f} us -> {2*X.numel()*4/t/1e12:.2f} TB/s (r+w)")
