"""Does the 256 MB Infinity Cache (MALL) make the conv2 / conv3 forward kernels faster when their input was written just before they run?

The PMC counters cannot tell (FETCH_SIZE counts the L2's fabric requests, MALL hits included), so this times the kernels themselves: the same launch on a chunk of NC
frames whose input was (a) just re-written by a copy (the producer's stores: MALL-resident if the chunk fits), (b) last touched ~2 GB of other traffic ago (HBM).
If (a) is clearly faster, running the encoder in MALL-sized frame chunks through all layers would pay; if not, the kernels are bound by something else.

    python tools/mall_chunk_probe.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
dev = "cuda"
flush_src = torch.empty(1 << 29, device=dev, dtype=torch.uint8)           # 512 MB each: a copy moves 1 GB through the memory side
flush_dst = torch.empty(1 << 29, device=dev, dtype=torch.uint8)


def timed(fn, prep, n=8):
    ts = []
    for _ in range(n):
        prep()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def conv_call(mode, img, w, b, out, Nf, IMH, OUTH):
    L.check(lib.hulc_k_conv_tile(mode, img.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), Nf, IMH, OUTH, 1, None))


for name, mode, CI, KK, IMH, OUTH in (("conv2 fwd (4x4/s2, 32->64, 49->23)", 11, 32, 16, 49, 23), ("conv3 fwd (3x3/s1, 64->64, 23->21)", 10, 64, 9, 23, 21)):
    w = (torch.randn(64, KK * CI, device=dev) * 0.05).to(torch.bfloat16); b = torch.randn(64, device=dev) * 0.1
    for Nf in (256, 512, 1024, 2048):
        img = (torch.randn(Nf, IMH, IMH, CI, device=dev)).to(torch.bfloat16)
        src = img.clone()
        out = torch.zeros(Nf, OUTH, OUTH, 64, device=dev, dtype=torch.bfloat16)
        mb = img.numel() * 2 / 1e6
        fn = lambda: conv_call(mode, img, w, b, out, Nf, IMH, OUTH)
        for _ in range(3): fn()
        def hot(): flush_dst.copy_(flush_src); img.copy_(src)                   # input freshly written (producer's stores)
        def cold(): img.copy_(src); flush_dst.copy_(flush_src); flush_dst.copy_(flush_src)      # 2 GB of other traffic after it
        th, tc = timed(fn, hot), timed(fn, cold)
        th2, tc2 = timed(fn, hot), timed(fn, cold)
        print(f"{name}  Nf {Nf:5d}  input {mb:6.1f} MB   input just written {min(th, th2):7.1f} us ({min(th, th2) / Nf * 1e3:6.1f} ns/frame)   after 2 GB of other traffic {min(tc, tc2):7.1f} us "
              f"({min(tc, tc2) / Nf * 1e3:6.1f} ns/frame)", flush=True)
