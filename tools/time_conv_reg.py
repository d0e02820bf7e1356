"""conv2 / conv3 FORWARD on 2048 frames: the LDS-resident-weights kernels (conv_tile.h, modes 0 / 7) against the weights-in-registers kernels
(conv_reg.h, modes 10 / 17), both cameras' shapes.   python tools/time_conv_reg.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()
Nf = 2048
def run(mode, img, w, bias, mask, out, IMH, OUTH, dbg=1):
    args = (mode, img.data_ptr(), w.data_ptr(), bias.data_ptr(), mask.data_ptr() if mask is not None else None, out.data_ptr(), Nf, IMH, OUTH, dbg, None)
    for _ in range(3): L.check(lib.hulc_k_conv_tile(*args))
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(10): lib.hulc_k_conv_tile(*args)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
b64 = torch.zeros(64, device="cuda")
i32 = lambda *s: torch.zeros(s, device="cuda", dtype=torch.int32)
for cam, (H1, H2, H3) in (("static", (49, 23, 21)), ("gripper", (20, 9, 7))):
    x2 = torch.randn(Nf, H1, H1, 32, device="cuda").to(torch.bfloat16); w2 = (torch.randn(64, 512, device="cuda") * 0.05).to(torch.bfloat16)
    x3 = torch.randn(Nf, H2, H2, 64, device="cuda").to(torch.bfloat16); w3 = (torch.randn(64, 576, device="cuda") * 0.05).to(torch.bfloat16)
    o2 = torch.zeros(Nf, H2, H2, 64, device="cuda", dtype=torch.bfloat16); o3 = torch.zeros(Nf, H3, H3, 64, device="cuda", dtype=torch.bfloat16)
    bits = i32(Nf, H2, H2, 2); bits1 = i32(Nf, H1, H1, 1); wd2 = (torch.randn(128, 256, device="cuda") * 0.05).to(torch.bfloat16)
    mb2 = (Nf * (H1 * H1 * 32 + H2 * H2 * 64) * 2) / 1e6; mb3 = (Nf * (H2 * H2 * 64 + H3 * H3 * 64) * 2) / 1e6
    t = dict(tile2=run(7, x2, w2, b64, bits, o2, H1, H2), reg2=run(17, x2, w2, b64, bits, o2, H1, H2), tile3=run(0, x3, w3, b64, None, o3, H2, H3), reg3=run(10, x3, w3, b64, None, o3, H2, H3),
             tiled3=run(8, o3, w3, b64, bits, x3, H3, H2, 32), regd3=run(18, o3, w3, b64, bits, x3, H3, H2, 0),
             tiled2=run(9, o2, wd2, b64, bits1, x2, H2, H1, 32), regd2=run(19, o2, wd2, b64, bits1, x2, H2, H1, 0),
             w4_2=run(27, x2, w2, b64, bits, o2, H1, H2), w4_3=run(20, x3, w3, b64, None, o3, H2, H3), w4_d3=run(28, o3, w3, b64, bits, x3, H3, H2, 0), w4_d2=run(29, o2, wd2, b64, bits1, x2, H2, H1, 0))
    print(f"{cam}: two workgroups per CU (NWV = 4): conv2 fwd {t['w4_2']:.1f}  conv3 fwd {t['w4_3']:.1f}  conv3 dgrad {t['w4_d3']:.1f}  conv2 dgrad {t['w4_d2']:.1f} us")
    print(f"{cam}: conv2 fwd (+bits) tile {t['tile2']:.1f} us -> reg {t['reg2']:.1f} us ({mb2 / t['reg2']:.2f} TB/s of {mb2:.0f} MB);  "
          f"conv3 fwd tile {t['tile3']:.1f} us -> reg {t['reg3']:.1f} us ({mb3 / t['reg3']:.2f} TB/s of {mb3:.0f} MB);  "
          f"conv3 dgrad (bits) tile {t['tiled3']:.1f} us -> reg {t['regd3']:.1f} us;  conv2 dgrad (bits) tile {t['tiled2']:.1f} us -> reg {t['regd2']:.1f} us")

if os.environ.get("ABLATE"):
    H1, H2, H3 = 49, 23, 21
    x2 = torch.randn(Nf, H1, H1, 32, device="cuda").to(torch.bfloat16); w2 = (torch.randn(64, 512, device="cuda") * 0.05).to(torch.bfloat16)
    x3 = torch.randn(Nf, H2, H2, 64, device="cuda").to(torch.bfloat16); w3 = (torch.randn(64, 576, device="cuda") * 0.05).to(torch.bfloat16)
    o2 = torch.zeros(Nf, H2, H2, 64, device="cuda", dtype=torch.bfloat16); o3 = torch.zeros(Nf, H3, H3, 64, device="cuda", dtype=torch.bfloat16)
    bits = i32(Nf, H2, H2, 2)
    bits1 = i32(Nf, H1, H1, 1); wd2 = (torch.randn(128, 256, device="cuda") * 0.05).to(torch.bfloat16)
    for name, mode, a in (("conv3 fwd", 10, (x3, w3, b64, None, o3, H2, H3)), ("conv2 fwd", 17, (x2, w2, b64, bits, o2, H1, H2)), ("conv3 dgrad", 18, (o3, w3, b64, bits, x3, H3, H2)),
                          ("conv2 dgrad", 19, (o2, wd2, b64, bits1, x2, H2, H1)),
                          ("conv3 fwd w4", 20, (x3, w3, b64, None, o3, H2, H3)), ("conv2 fwd w4", 27, (x2, w2, b64, bits, o2, H1, H2)), ("conv3 dgrad w4", 28, (o3, w3, b64, bits, x3, H3, H2)),
                          ("conv2 dgrad w4", 29, (o2, wd2, b64, bits1, x2, H2, H1)),
                          ("conv3 fwd w4x2", 30, (x3, w3, b64, None, o3, H2, H3)), ("conv2 fwd w4x2", 37, (x2, w2, b64, bits, o2, H1, H2)), ("conv3 dgrad w4x2", 38, (o3, w3, b64, bits, x3, H3, H2)),
                          ("conv2 dgrad w4x2", 39, (o2, wd2, b64, bits1, x2, H2, H1))):
        if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
            continue
        if mode % 10 >= 8 and os.environ.get("STORE_EXP"):       # the data-gradient forms: epilogue / store experiments (needs the AB build: HULC_BUILD_AB=1)
            r = {k: run(mode, *a, d) for k, d in (("epilogue-only", 20), ("epi-only-compact", 20 | 64), ("epi-only-nt", 20 | 128), ("full-compact", 64), ("full-nt", 128))}
            print(name, {k: round(v, 1) for k, v in r.items()})
        r = {k: run(mode, *a, d | (1 if mode % 10 < 8 else 0)) for k, d in (("full", 0), ("no-dma", 4), ("no-compute", 2), ("no-epilogue", 8), ("no-mfma", 16), ("no-mfma-no-epi", 24), ("no-dma-no-epi", 12), ("epilogue-only", 20), ("dma-only", 2), ("nothing", 6))}
        print(name, {k: round(v, 1) for k, v in r.items()})
