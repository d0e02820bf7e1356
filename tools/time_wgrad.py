"""conv2 / conv3 weight-gradient kernels on 2048 static-camera frames, run under `rocprofv3 --kernel-trace --stats` (the entry point allocates and
synchronises per call, so wall / event times are not the kernel's):   rocprofv3 --kernel-trace --stats -d out -- python tools/time_wgrad.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
Nf = int(os.environ.get("NF", 2048))
for which, IH, CI, KH, S in ((3, 23, 64, 3, 1), (2, 49, 32, 4, 2)):
    OH = (IH - KH) // S + 1
    X = torch.randn(Nf, IH, IH, CI, device="cuda").to(torch.bfloat16); dY = torch.randn(Nf, OH, OH, 64, device="cuda").to(torch.bfloat16)
    out = torch.zeros(64, KH * KH * CI, device="cuda")
    for _ in range(6):
        L.check(lib.hulc_k_conv_wgrad(which, X.data_ptr(), dY.data_ptr(), out.data_ptr(), Nf, IH, None))
