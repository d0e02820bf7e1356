import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
for (M, N, K) in [(2048, 2048, 2048), (2048, 2048, 1120), (2048, 1120, 2048), (2048, 512, 2048)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    c = torch.zeros(M, N, device="cuda"); bias = torch.zeros(N, device="cuda")
    for flag, name in ((1, "glds 3-stage"), (5, "register-staged")):
        for _ in range(3): lib.hulc_k_gemm_nt(L.DTYPE["bf16"], a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, K, K, N, bias.data_ptr(), flag, None)
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(10): lib.hulc_k_gemm_nt(L.DTYPE["bf16"], a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, K, K, N, bias.data_ptr(), flag, None)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        print(f"{M}x{N}x{K} {name:16s} {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
