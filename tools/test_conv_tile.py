"""GPU check of the raw-tile conv kernels (fwd + dgrad) against numpy on bf16-rounded data."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hulc_amd import lib as L
lib = L.load()

def bf(x): return torch.from_numpy(x).cuda().to(torch.bfloat16).contiguous()
def f64(t): return t.float().cpu().numpy().astype(np.float64)

def conv_fwd(X, W, b, S):          # X (n,h,w,ci) ; W (co,ci,kh,kw)
    n, ih, iw, ci = X.shape; co, _, kh, kw = W.shape
    oh, ow = (ih - kh) // S + 1, (iw - kw) // S + 1
    out = np.zeros((n, oh, ow, co))
    for a in range(kh):
        for c in range(kw):
            out += np.einsum("nhwc,dc->nhwd", X[:, a:a + S * oh:S, c:c + S * ow:S, :], W[:, :, a, c])
    return np.maximum(out + b, 0)

def conv_dgrad(dY, W, S, IH):      # dY (n,oh,ow,co) -> dX (n,IH,IH,ci)
    n, oh, ow, co = dY.shape; _, ci, kh, kw = W.shape
    dX = np.zeros((n, IH, IH, ci))
    for a in range(kh):
        for c in range(kw):
            dX[:, a:a + S * oh:S, c:c + S * ow:S, :] += np.einsum("nhwd,dc->nhwc", dY, W[:, :, a, c])
    return dX

rng = np.random.default_rng(0)
ok_all = True
for (IH, CI, KH, S, fmode, dmode) in ((23, 64, 3, 1, 0, 2), (9, 64, 3, 1, 0, 2), (49, 32, 4, 2, 1, 3), (20, 32, 4, 2, 1, 3)):
    OH = (IH - KH) // S + 1
    for Nf in (2, 11):
        W = (rng.standard_normal((64, CI, KH, KH)) * 0.1).astype(np.float32)
        Wb = f64(bf(W))
        X = bf(rng.standard_normal((Nf, IH, IH, CI)).astype(np.float32)); b = rng.standard_normal(64).astype(np.float32)
        # packed weights
        wf = bf(np.ascontiguousarray(Wb.transpose(0, 2, 3, 1)).reshape(64, -1).astype(np.float32))
        out = torch.zeros(Nf, OH, OH, 64, device="cuda", dtype=torch.bfloat16)
        bd = torch.from_numpy(b).cuda()
        L.check(lib.hulc_k_conv_tile(fmode, X.data_ptr(), wf.data_ptr(), bd.data_ptr(), None, out.data_ptr(), Nf, IH, OH, 1, None))
        torch.cuda.synchronize()
        ref = conv_fwd(f64(X), Wb, b, S)
        err = np.abs(f64(out) - ref).max() / np.abs(ref).max()
        good = err < 6e-3; ok_all &= good
        print(f"fwd  IH={IH} S={S} Nf={Nf}: rel err {err:.2e}", "OK" if good else "FAIL")
        # dgrad
        dY = bf((rng.standard_normal((Nf, OH, OH, 64)) * (np.arange(64) % 5 + 1)).astype(np.float32))
        TA = KH // S
        wd = np.zeros((S * S, CI, TA, TA, 64))
        for kh in range(KH):
            for kw in range(KH):
                wd[(kh % S) * S + kw % S, :, kh // S, kw // S, :] = Wb[:, :, kh, kw].T
        wdp = bf(wd.reshape(S * S * CI, -1).astype(np.float32))
        mask = bf(rng.standard_normal((Nf, IH, IH, CI)).astype(np.float32))
        dx = torch.full((Nf, IH, IH, CI), 7.0, device="cuda", dtype=torch.bfloat16)
        L.check(lib.hulc_k_conv_tile(dmode, dY.data_ptr(), wdp.data_ptr(), None, mask.data_ptr(), dx.data_ptr(), Nf, OH, IH, 0, None))
        torch.cuda.synchronize()
        ref = conv_dgrad(f64(dY), Wb, S, IH) * (f64(mask) > 0)
        err = np.abs(f64(dx) - ref).max() / np.abs(ref).max()
        good = err < 6e-3; ok_all &= good
        print(f"dgrad IH={IH} S={S} Nf={Nf}: rel err {err:.2e}", "OK" if good else "FAIL")
for IH in (200, 84):
    OH = (IH - 8) // 4 + 1
    for Nf in (2, 7):
        W = (rng.standard_normal((32, 3, 8, 8)) * 0.1).astype(np.float32); Wb = f64(bf(W))
        X = torch.from_numpy(rng.standard_normal((Nf, 3, IH, IH)).astype(np.float32)).cuda(); b = rng.standard_normal(32).astype(np.float32)
        wf = bf(Wb.reshape(32, -1).astype(np.float32)); bd = torch.from_numpy(b).cuda()
        out = torch.zeros(Nf, OH, OH, 32, device="cuda", dtype=torch.bfloat16)
        L.check(lib.hulc_k_conv_tile(4, X.data_ptr(), wf.data_ptr(), bd.data_ptr(), None, out.data_ptr(), Nf, IH, OH, 1, None)); torch.cuda.synchronize()
        Xb = f64(X.to(torch.bfloat16)).transpose(0, 2, 3, 1)
        ref = conv_fwd(Xb, Wb, b, 4)
        err = np.abs(f64(out) - ref).max() / np.abs(ref).max()
        good = err < 6e-3; ok_all &= good
        print(f"conv1 fwd IH={IH} Nf={Nf}: rel err {err:.2e}", "OK" if good else "FAIL")
print("ALL OK" if ok_all else "SOME FAILED")
