import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
M, N, K = 64, 2048, 2048
W = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
H = [torch.randn(M, K, device="cuda").to(torch.bfloat16) for _ in range(33)]
ref = (H[0].float() @ W.float().T)
for var in (82, 202, 92):
    L.check(lib.hulc_k_skinny(H[0].data_ptr(), W.data_ptr(), H[1].data_ptr(), M, N, K, var, None)); torch.cuda.synchronize()
    err = (H[1].float() - ref).abs().max().item() / ref.abs().max().item()
    # dependent chain like the RNN: H[t+1] = f(H[t])
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    for rep in range(2):
        e0.record()
        for t in range(32):
            lib.hulc_k_skinny(H[t].data_ptr(), W.data_ptr(), H[t + 1].data_ptr(), M, N, K, var, None)
        e1.record(); torch.cuda.synchronize()
    print(f"variant NW={var//10} MT={var%10}: err {err:.1e}  chain of 32: {e0.elapsed_time(e1) / 32 * 1e3:.2f} us/launch")
