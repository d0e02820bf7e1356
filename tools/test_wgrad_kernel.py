"""GPU check + timing of the tr-read conv wgrad kernel against numpy (bf16-rounded inputs, fp64 reference)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()

def ref(X, dY, KH, S):
    n, ih, iw, ci = X.shape; _, oh, ow, co = dY.shape
    out = np.zeros((co, KH, KH, ci))
    for kh in range(KH):
        for kw in range(KH):
            xs = X[:, kh:kh + S * oh:S, kw:kw + S * ow:S, :]
            out[:, kh, kw, :] = np.einsum("nhwc,nhwd->cd", dY, xs)
    return out.reshape(co, -1)

for which, IH, CI, KH, S in ((3, 23, 64, 3, 1), (3, 9, 64, 3, 1), (2, 49, 32, 4, 2), (2, 20, 32, 4, 2)):
    OH = (IH - KH) // S + 1
    for Nf in (3, 37):
        g = torch.Generator(device="cuda"); g.manual_seed(which * 100 + Nf)
        X = torch.randn(Nf, IH, IH, CI, device="cuda", generator=g).to(torch.bfloat16).contiguous()
        dY = (torch.randn(Nf, OH, OH, 64, device="cuda", generator=g) * (torch.arange(64, device="cuda") % 7 + 1)).to(torch.bfloat16).contiguous()
        out = torch.zeros(64, KH * KH * CI, device="cuda")
        L.check(lib.hulc_k_conv_wgrad(which, X.data_ptr(), dY.data_ptr(), out.data_ptr(), Nf, IH, None))
        r = ref(X.float().cpu().numpy().astype(np.float64), dY.float().cpu().numpy().astype(np.float64), KH, S)
        err = np.abs(out.cpu().numpy() - r).max() / np.abs(r).max()
        print(f"conv{which} IH={IH} Nf={Nf}: rel err {err:.2e}", "OK" if err < 1e-4 else "FAIL")
def ref1(X, dY):
    n, c, ih, iw = X.shape; _, oh, ow, co = dY.shape
    out = np.zeros((co, c, 8, 8))
    for kh in range(8):
        for kw in range(8):
            xs = X[:, :, kh:kh + 4 * oh:4, kw:kw + 4 * ow:4]
            out[:, :, kh, kw] = np.einsum("nhwd,nchw->dc", dY, xs)
    return out.reshape(co, -1)

for IH in (200, 84):
    OH = (IH - 8) // 4 + 1
    for Nf in (2, 19):
        g = torch.Generator(device="cuda"); g.manual_seed(IH + Nf)
        X = torch.randn(Nf, 3, IH, IH, device="cuda", generator=g).contiguous()
        dY = (torch.randn(Nf, OH, OH, 32, device="cuda", generator=g) * (torch.arange(32, device="cuda") % 5 + 1)).to(torch.bfloat16).contiguous()
        out = torch.zeros(32, 192, device="cuda")
        L.check(lib.hulc_k_conv_wgrad(1, X.data_ptr(), dY.data_ptr(), out.data_ptr(), Nf, IH, None))
        r = ref1(X.to(torch.bfloat16).float().cpu().numpy().astype(np.float64), dY.float().cpu().numpy().astype(np.float64))
        err = np.abs(out.cpu().numpy() - r).max() / np.abs(r).max()
        print(f"conv1 IH={IH} Nf={Nf}: rel err {err:.2e}", "OK" if err < 1e-4 else "FAIL")

for which, IH, CI, KH, S in ((3, 23, 64, 3, 1), (2, 49, 32, 4, 2)):
    OH = (IH - KH) // S + 1; Nf = 2048
    X = torch.randn(Nf, IH, IH, CI, device="cuda").to(torch.bfloat16); dY = torch.randn(Nf, OH, OH, 64, device="cuda").to(torch.bfloat16)
    out = torch.zeros(64, KH * KH * CI, device="cuda")
    for _ in range(2):
        L.check(lib.hulc_k_conv_wgrad(which, X.data_ptr(), dY.data_ptr(), out.data_ptr(), Nf, IH, None))
    t0 = time.time()
    for _ in range(5):
        L.check(lib.hulc_k_conv_wgrad(which, X.data_ptr(), dY.data_ptr(), out.data_ptr(), Nf, IH, None))
    dt = (time.time() - t0) / 5
    fl = 2.0 * Nf * OH * OH * 64 * KH * KH * CI
    print(f"conv{which} wgrad Nf=2048: {dt * 1e6:.0f} us incl. malloc/sync  ({fl / dt / 1e12:.0f} TFLOP/s)")
