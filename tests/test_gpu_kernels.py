"""GPU (-m gpu): the individual bench-mode (bf16) kernels through their C-ABI test entry points, against numpy on
bf16-rounded operands (fp64 reference): raw-tile conv forward / dgrad, tr-read conv wgrad (incl. conv1 from fp32 NCHW),
skinny GEMM, and the ds_read_b64_tr_b16 lane mapping the wgrad kernels are built on."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _lib():
    from hulc_amd import lib as L
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    return L, L.load()


def bf(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda().to(torch.bfloat16).contiguous()


def f64(t):
    return t.float().cpu().numpy().astype(np.float64)


def conv_fwd_ref(X, W, b, S):          # X (n,h,w,ci) ; W (co,ci,kh,kw)
    n, ih, iw, ci = X.shape
    co, _, kh, kw = W.shape
    oh, ow = (ih - kh) // S + 1, (iw - kw) // S + 1
    out = np.zeros((n, oh, ow, co))
    for a in range(kh):
        for c in range(kw):
            out += np.einsum("nhwc,dc->nhwd", X[:, a:a + S * oh:S, c:c + S * ow:S, :], W[:, :, a, c])
    return np.maximum(out + b, 0)


def conv_dgrad_ref(dY, W, S, IH):
    n, oh, ow, co = dY.shape
    _, ci, kh, kw = W.shape
    dX = np.zeros((n, IH, IH, ci))
    for a in range(kh):
        for c in range(kw):
            dX[:, a:a + S * oh:S, c:c + S * ow:S, :] += np.einsum("nhwd,dc->nhwc", dY, W[:, :, a, c])
    return dX


def conv_wgrad_ref(X, dY, KH, S):
    n, ih, iw, ci = X.shape
    _, oh, ow, co = dY.shape
    out = np.zeros((co, KH, KH, ci))
    for kh in range(KH):
        for kw in range(KH):
            out[:, kh, kw, :] = np.einsum("nhwc,nhwd->cd", dY, X[:, kh:kh + S * oh:S, kw:kw + S * ow:S, :])
    return out.reshape(co, -1)


# Nf = 7 / 1501 on the gripper shapes (9x9, 20x20): stacked bands (ConvTileP::FPB frames per band) with a short last band
@pytest.mark.parametrize("IH,CI,KH,S,fmode,dmode,Nf", [(23, 64, 3, 1, 0, 2, 1), (23, 64, 3, 1, 0, 2, 7), (9, 64, 3, 1, 0, 2, 1), (9, 64, 3, 1, 0, 2, 7),
                                                       (9, 64, 3, 1, 0, 2, 1501), (49, 32, 4, 2, 1, 3, 1), (49, 32, 4, 2, 1, 3, 7), (20, 32, 4, 2, 1, 3, 1),
                                                       (20, 32, 4, 2, 1, 3, 7), (20, 32, 4, 2, 1, 3, 1027)])
def test_conv_tile_fwd_and_dgrad(IH, CI, KH, S, fmode, dmode, Nf):
    L, lib = _lib()
    rng = np.random.default_rng(IH * 10 + Nf)
    OH = (IH - KH) // S + 1
    Wb = f64(bf(rng.standard_normal((64, CI, KH, KH)) * 0.1))
    X = bf(rng.standard_normal((Nf, IH, IH, CI)))
    b = rng.standard_normal(64).astype(np.float32)
    wf = bf(Wb.transpose(0, 2, 3, 1).reshape(64, -1))
    out = torch.zeros(Nf, OH, OH, 64, device="cuda", dtype=torch.bfloat16)
    L.check(lib.hulc_k_conv_tile(fmode, X.data_ptr(), wf.data_ptr(), torch.from_numpy(b).cuda().data_ptr(), None, out.data_ptr(), Nf, IH, OH, 1, None))
    torch.cuda.synchronize()
    ref = conv_fwd_ref(f64(X), Wb, b, S)
    assert np.abs(f64(out) - ref).max() / np.abs(ref).max() < 6e-3          # bf16 output rounding
    # dgrad with ReLU mask; parity-class weight pack [class][ci][(a,b,co)]
    dY = bf(rng.standard_normal((Nf, OH, OH, 64)) * (np.arange(64) % 5 + 1))
    TA = KH // S
    wd = np.zeros((S * S, CI, TA, TA, 64))
    for kh in range(KH):
        for kw in range(KH):
            wd[(kh % S) * S + kw % S, :, kh // S, kw // S, :] = Wb[:, :, kh, kw].T
    mask = bf(rng.standard_normal((Nf, IH, IH, CI)))
    dx = torch.full((Nf, IH, IH, CI), 7.0, device="cuda", dtype=torch.bfloat16)      # every pixel must be overwritten
    L.check(lib.hulc_k_conv_tile(dmode, dY.data_ptr(), bf(wd.reshape(S * S * CI, -1)).data_ptr(), None, mask.data_ptr(), dx.data_ptr(), Nf, OH, IH, 0, None))
    torch.cuda.synchronize()
    ref = conv_dgrad_ref(f64(dY), Wb, S, IH) * (f64(mask) > 0)
    assert np.abs(f64(dx) - ref).max() / np.abs(ref).max() < 6e-3


@pytest.mark.parametrize("IH", [200, 84])
def test_conv1_fwd_and_wgrad_from_fp32_nchw(IH):
    L, lib = _lib()
    rng = np.random.default_rng(IH)
    OH = (IH - 8) // 4 + 1
    Nf = 3
    Wb = f64(bf(rng.standard_normal((32, 3, 8, 8)) * 0.1))
    X = torch.from_numpy(rng.standard_normal((Nf, 3, IH, IH)).astype(np.float32)).cuda()
    b = rng.standard_normal(32).astype(np.float32)
    out = torch.zeros(Nf, OH, OH, 32, device="cuda", dtype=torch.bfloat16)
    L.check(lib.hulc_k_conv_tile(4, X.data_ptr(), bf(Wb.reshape(32, -1)).data_ptr(), torch.from_numpy(b).cuda().data_ptr(), None, out.data_ptr(), Nf, IH, OH, 1, None))
    torch.cuda.synchronize()
    Xb = f64(X.to(torch.bfloat16)).transpose(0, 2, 3, 1)
    ref = conv_fwd_ref(Xb, Wb, b, 4)
    assert np.abs(f64(out) - ref).max() / np.abs(ref).max() < 6e-3
    dY = bf(rng.standard_normal((Nf, OH, OH, 32)) * (np.arange(32) % 5 + 1))
    g = torch.zeros(32, 192, device="cuda")
    L.check(lib.hulc_k_conv_wgrad(1, X.data_ptr(), dY.data_ptr(), g.data_ptr(), Nf, IH, None))
    refw = np.zeros((32, 3, 8, 8))
    Xn = f64(X.to(torch.bfloat16))
    for kh in range(8):
        for kw in range(8):
            refw[:, :, kh, kw] = np.einsum("nhwd,nchw->dc", f64(dY), Xn[:, :, kh:kh + 4 * OH:4, kw:kw + 4 * OH:4])
    assert np.abs(g.cpu().numpy() - refw.reshape(32, -1)).max() / np.abs(refw).max() < 1e-4      # fp32 accumulate: tight


@pytest.mark.parametrize("which,IH,CI,KH,S", [(3, 23, 64, 3, 1), (3, 9, 64, 3, 1), (2, 49, 32, 4, 2), (2, 20, 32, 4, 2), (3, 31, 64, 3, 1), (2, 57, 32, 4, 2), (3, 18, 64, 3, 1)])      # 23 / 49 / 31 / 57 / 18: the LDS-DMA kernel (2 - 5 bands per frame, OWp = 24 / 32 / 16)
def test_conv_wgrad_tr(which, IH, CI, KH, S):
    L, lib = _lib()
    rng = np.random.default_rng(which * 100 + IH)
    OH = (IH - KH) // S + 1
    for Nf in (3, 37):
        X = bf(rng.standard_normal((Nf, IH, IH, CI)))
        dY = bf(rng.standard_normal((Nf, OH, OH, 64)) * (np.arange(64) % 7 + 1))
        out = torch.zeros(64, KH * KH * CI, device="cuda")
        L.check(lib.hulc_k_conv_wgrad(which, X.data_ptr(), dY.data_ptr(), out.data_ptr(), Nf, IH, None))
        ref = conv_wgrad_ref(f64(X), f64(dY), KH, S)
        assert np.abs(out.cpu().numpy() - ref).max() / np.abs(ref).max() < 1e-4


# variants 20x: the LDS-DMA kernel with MT = x row tiles per workgroup (201: the 16-row blocks of the M <= 32 recurrent step; 202 with a
# ragged last block and many rows: the K = 2048 many-row routing)
@pytest.mark.parametrize("M,N,K,variant", [(64, 2048, 2048, 82), (64, 2048, 2048, 84), (37, 256, 512, 82), (5, 32, 128, 41), (16, 1024, 4096, 81),
                                           (64, 2048, 2048, 202), (32, 2048, 2048, 201), (27, 2048, 2048, 201), (500, 128, 2048, 202), (64, 6144, 2048, 202),
                                           (300, 128, 3136, 44), (64, 2048, 160, 42), (64, 64, 96, 41),
                                           (64, 2048, 6144, 300), (37, 1024, 4096, 300), (64, 512, 2048, 300)])      # 300: the production router; K = n x 2048 -> K-chunked LDS kernel     # K % 128 != 0: uneven k-step split over the waves
def test_skinny_gemm(M, N, K, variant):
    L, lib = _lib()
    rng = np.random.default_rng(M + N)
    A = bf(rng.standard_normal((M, K)))
    W = bf(rng.standard_normal((N, K)) * 0.05 + np.arange(N)[:, None] * 1e-4)      # asymmetric: a transposed result would show
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    L.check(lib.hulc_k_skinny(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, variant, None))
    torch.cuda.synchronize()
    ref = f64(A) @ f64(W).T
    assert np.abs(f64(out) - ref).max() / np.abs(ref).max() < 6e-3


@pytest.mark.parametrize("N,K", [(384, 128), (128, 384), (2048, 128), (128, 2048), (96, 6144)])
def test_fragment_ordered_weight_copy_layout(N, K):
    """frag_pack_kernel: block (row tile nb, k-step ks) = 1 KB, lane (g, i) at byte 16 (16 g + i) holds W[16 nb + i][32 ks + 8 g .. + 7] — what a
    wave's MFMA fragment load of the GRU step / fused transformer kernels expects as one contiguous instruction (both tile widths of the kernel)."""
    L, lib = _lib()
    rng = np.random.default_rng(N + K)
    W = bf(rng.standard_normal((N, K)))
    out = torch.zeros(N * K, device="cuda", dtype=torch.bfloat16)
    L.check(lib.hulc_k_skinny(None, W.data_ptr(), out.data_ptr(), 1, N, K, 510, None))
    torch.cuda.synchronize()
    w = W.view(torch.int16).cpu().numpy()
    # [nb][i][ks][g][8] -> [nb][ks][g][i][8]
    want = w.reshape(N // 16, 16, K // 32, 4, 8).transpose(0, 2, 3, 1, 4).reshape(-1)
    assert np.array_equal(out.view(torch.int16).cpu().numpy(), want)


@pytest.mark.parametrize("M,N,K", [(64, 2048, 6144), (37, 1024, 4096)])
def test_skinny_kchunk_reads_fragment_ordered_weights(M, N, K):
    """The GRU's BPTT-step kernel on a fragment-ordered copy of W (ldw == 0) against fp64."""
    L, lib = _lib()
    rng = np.random.default_rng(M + N)
    A = bf(rng.standard_normal((M, K)))
    W = bf(rng.standard_normal((N, K)) * 0.05 + np.arange(N)[:, None] * 1e-4)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    L.check(lib.hulc_k_skinny(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, 500, None))
    torch.cuda.synchronize()
    ref = f64(A) @ f64(W).T
    assert np.abs(f64(out) - ref).max() / np.abs(ref).max() < 6e-3


@pytest.mark.parametrize("M", [40, 64])
def test_skinny_dual_launch(M):
    """Two independent recurrent-step GEMMs in one launch (grid.z = 2: the paired directions of the bidirectional plan encoders)."""
    L, lib = _lib()
    N = K = 2048
    rng = np.random.default_rng(M)
    A = bf(rng.standard_normal((2, M, K)))
    W = bf(rng.standard_normal((2, N, K)) * 0.05 + np.arange(N)[None, :, None] * 1e-4)
    out = torch.zeros(2, M, N, device="cuda", dtype=torch.bfloat16)
    L.check(lib.hulc_k_skinny(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, 400, None))
    torch.cuda.synchronize()
    for z in range(2):
        ref = f64(A[z]) @ f64(W[z]).T
        assert np.abs(f64(out[z]) - ref).max() / np.abs(ref).max() < 6e-3, z


def test_tr_read_lane_mapping():
    """ds_read_b64_tr_b16: lane i of a 16-lane group receives element (i & 3) of the chunks addressed by lanes j*4 + (i >> 2)."""
    L, lib = _lib()
    lane = np.arange(64)
    addr = torch.from_numpy(((lane >> 4) * 2048 + (lane & 15) * 64).astype(np.int32)).cuda()
    out = torch.zeros(256, dtype=torch.int16, device="cuda")
    L.check(lib.hulc_k_trread_probe(addr.data_ptr(), out.data_ptr(), None))
    torch.cuda.synchronize()
    o = out.cpu().numpy().astype(np.int64).reshape(64, 4) & 0xFFFF
    for l in range(64):
        for j in range(4):
            assert o[l, j] == (l >> 4) * 2048 + (j * 4 + (l & 15) // 4) * 64 + (l & 15) % 4


@pytest.mark.parametrize("M,N,K", [(2048, 2048, 2048), (2048, 1120, 2048), (2048, 2048, 1120), (1000, 200, 160), (520, 136, 128), (2048, 2048, 64), (640, 256, 96)])
def test_gemm_glds_matches_fp64(M, N, K):
    """3-stage LDS-DMA GEMM (gemm_glds_kernel, the RNN input/weight-gradient GEMMs): ragged tiles, K % 64 == 32 tail, bias + ReLU
    epilogue, against an fp64 product of the same bf16-rounded operands; also equal to the register-staged kernel."""
    from hulc_amd import lib as L
    lib = L.load()
    g = torch.Generator(device="cuda"); g.manual_seed(M + N + K)
    a = (torch.randn(M, K, device="cuda", generator=g) + torch.arange(M, device="cuda")[:, None] * 1e-3).to(torch.bfloat16).contiguous()
    b = (torch.randn(N, K, device="cuda", generator=g) * 0.5 + torch.arange(N, device="cuda")[:, None] * 1e-3).to(torch.bfloat16).contiguous()
    bias = torch.randn(N, device="cuda", generator=g)
    ref = torch.relu(a.double() @ b.double().T + bias.double())
    c = torch.zeros(M, N, device="cuda"); c_old = torch.zeros(M, N, device="cuda")
    L.check(lib.hulc_k_gemm_nt(L.DTYPE["bf16"], a.data_ptr(), b.data_ptr(), c.data_ptr(), M, N, K, K, K, N, bias.data_ptr(), 1, None))
    L.check(lib.hulc_k_gemm_nt(L.DTYPE["bf16"], a.data_ptr(), b.data_ptr(), c_old.data_ptr(), M, N, K, K, K, N, bias.data_ptr(), 5, None))
    torch.cuda.synchronize()
    err = ((c.double() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err                       # fp32 accumulation of exact bf16 products
    assert (c - c_old).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("variant,S", [(0, 5), (0, 32), (0, 47), (1, 1), (1, 5), (1, 8), (1, 16), (1, 17), (1, 20), (1, 32), (2, 33), (2, 47), (2, 64), (2, 9)])
@pytest.mark.parametrize("drop_p", [0.0, 0.1])
def test_attention_kernels(variant, S, drop_p):
    """Forward and backward attention kernels (plan_recognition_net.py:94-117: nn.TransformerEncoderLayer, 8 heads of 16) against float64;
    with dropout the two kernel variants must draw the same mask (the backward re-derives it from the element index)."""
    L, lib = _lib()
    B, D, NH, HD = 3, 128, 8, 16
    g = torch.Generator(device="cuda"); g.manual_seed(S * 10 + variant)
    qkv = torch.randn(B * S, 3 * D, device="cuda", generator=g)
    dao = torch.randn(B * S, D, device="cuda", generator=g)
    P = torch.zeros(B, NH, S, S, device="cuda"); ao = torch.zeros(B * S, D, device="cuda"); dqkv = torch.zeros(B * S, 3 * D, device="cuda")
    seed = 1234567
    L.check(lib.hulc_k_attention(variant, qkv.data_ptr(), P.data_ptr(), ao.data_ptr(), None, None, B, S, drop_p, seed, None))
    L.check(lib.hulc_k_attention(variant, qkv.data_ptr(), P.data_ptr(), None, dao.data_ptr(), dqkv.data_ptr(), B, S, drop_p, seed, None))
    torch.cuda.synchronize()
    x = qkv.double().reshape(B, S, 3, NH, HD).permute(2, 0, 3, 1, 4)          # (3, B, NH, S, HD)
    q, k, v = (t.clone().requires_grad_(True) for t in x)
    Pref = torch.softmax(q @ k.transpose(-1, -2) / 4.0, dim=-1)
    assert (P.double() - Pref).abs().max().item() < 2e-6
    if drop_p > 0:      # the mask the kernel drew: recovered from a second, mask-revealing forward (v = identity-like is not available) -> use variant 0 as the mask oracle
        P0 = torch.zeros_like(P); ao0 = torch.zeros_like(ao); dq0 = torch.zeros_like(dqkv)
        L.check(lib.hulc_k_attention(0, qkv.data_ptr(), P0.data_ptr(), ao0.data_ptr(), None, None, B, S, drop_p, seed, None))
        L.check(lib.hulc_k_attention(0, qkv.data_ptr(), P0.data_ptr(), None, dao.data_ptr(), dq0.data_ptr(), B, S, drop_p, seed, None))
        torch.cuda.synchronize()
        assert (ao - ao0).abs().max().item() < 1e-5 * ao0.abs().max().item() + 1e-6
        assert (dqkv - dq0).abs().max().item() < 1e-5 * dq0.abs().max().item() + 1e-6
        return
    out = (Pref @ v).permute(0, 2, 1, 3).reshape(B * S, D)
    assert (ao.double() - out).abs().max().item() < 1e-5
    out.backward(dao.double())
    ref = torch.stack([q.grad, k.grad, v.grad]).permute(1, 3, 0, 2, 4).reshape(B * S, 3 * D)
    assert (dqkv.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item()


# the weights-in-registers forward kernels (conv_reg.h): modes 10 / 11 = conv3 / conv2 forward, 17 = conv2 forward that also emits its ReLU bitmask.
# Shapes: both cameras (static 23 / 49, gripper 9 / 20 with stacked frames), Nf not a multiple of the stack, > 256 frames (several items per workgroup)
@pytest.mark.parametrize("IH,CI,KH,S,mode,Nf", [(23, 64, 3, 1, 10, 1), (23, 64, 3, 1, 10, 7), (23, 64, 3, 1, 10, 530), (9, 64, 3, 1, 10, 1), (9, 64, 3, 1, 10, 7), (9, 64, 3, 1, 10, 1501),
                                                (49, 32, 4, 2, 11, 1), (49, 32, 4, 2, 11, 7), (49, 32, 4, 2, 11, 300), (20, 32, 4, 2, 11, 1), (20, 32, 4, 2, 11, 7), (20, 32, 4, 2, 11, 1027),
                                                (49, 32, 4, 2, 17, 5), (20, 32, 4, 2, 17, 1027), (31, 64, 3, 1, 10, 3), (30, 32, 4, 2, 11, 3),
                                                # modes 20 / 21 / 27: the same kernels as two 256-thread workgroups per CU (round 4; > 512 frames = several items per workgroup)
                                                (23, 64, 3, 1, 20, 1), (23, 64, 3, 1, 20, 7), (23, 64, 3, 1, 20, 1100), (9, 64, 3, 1, 20, 1), (9, 64, 3, 1, 20, 7), (9, 64, 3, 1, 20, 2051),
                                                (49, 32, 4, 2, 21, 1), (49, 32, 4, 2, 21, 7), (49, 32, 4, 2, 21, 600), (20, 32, 4, 2, 21, 1), (20, 32, 4, 2, 21, 7), (20, 32, 4, 2, 21, 1027),
                                                (49, 32, 4, 2, 27, 5), (20, 32, 4, 2, 27, 2051), (31, 64, 3, 1, 20, 3), (30, 32, 4, 2, 21, 3)])
def test_conv_reg_fwd(IH, CI, KH, S, mode, Nf):
    L, lib = _lib()
    rng = np.random.default_rng(IH * 10 + Nf)
    OH = (IH - KH) // S + 1
    Wb = f64(bf((rng.standard_normal((64, CI, KH, KH)) + np.arange(64)[:, None, None, None] * 0.01) * 0.1))     # asymmetric in the channel index
    X = bf(rng.standard_normal((Nf, IH, IH, CI)))
    b = rng.standard_normal(64).astype(np.float32)
    wf = bf(Wb.transpose(0, 2, 3, 1).reshape(64, -1))
    out = torch.full((Nf, OH, OH, 64), 7.0, device="cuda", dtype=torch.bfloat16)       # every output must be overwritten
    bits = torch.full((Nf, OH, OH, 2), -1, device="cuda", dtype=torch.int32) if mode in (17, 27) else None
    L.check(lib.hulc_k_conv_tile(mode, X.data_ptr(), wf.data_ptr(), torch.from_numpy(b).cuda().data_ptr(), bits.data_ptr() if bits is not None else None,
                                 out.data_ptr(), Nf, IH, OH, 1, None))
    torch.cuda.synchronize()
    ref = conv_fwd_ref(f64(X), Wb, b, S)
    err = np.abs(f64(out) - ref).reshape(Nf, -1).max(1) / np.abs(ref).max()
    assert err.max() < 6e-3, (err.max(), int(err.argmax()))          # bf16 output rounding; worst FRAME
    if bits is not None:
        w = bits.to(torch.int64) & 0xFFFFFFFF
        sh = torch.arange(32, device="cuda")
        got = ((w[..., None] >> sh) & 1).reshape(Nf, OH, OH, 64).bool()
        assert torch.equal(got, out > 0)


# conv3's data gradient on the weights-in-registers kernel (conv_reg.h, REV form): mode 12 = 16-bit mask values, 18 = the production ReLU bitmask.
# IH = the layer INPUT size (output of the gradient); 23 / 9 = the two cameras, 31 = two bands per frame; stacked frames with a short last stack
@pytest.mark.parametrize("IH,Nf,mode", [(23, 1, 12), (23, 7, 12), (23, 530, 18), (9, 1, 12), (9, 7, 18), (9, 1501, 18), (31, 3, 12), (31, 5, 18),
                                        (23, 1, 22), (23, 7, 22), (23, 1100, 28), (9, 1, 22), (9, 7, 28), (9, 2051, 28), (31, 3, 22), (31, 5, 28)])    # 22 / 28: two workgroups per CU
def test_conv_reg_dgrad3(IH, Nf, mode):
    L, lib = _lib()
    rng = np.random.default_rng(IH * 7 + Nf)
    OH = IH - 2
    Wb = f64(bf((rng.standard_normal((64, 64, 3, 3)) + np.arange(64)[None, :, None, None] * 0.01) * 0.1))
    dY = bf(rng.standard_normal((Nf, OH, OH, 64)) * (np.arange(64) % 5 + 1))
    wd = np.zeros((1, 64, 3, 3, 64))
    for kh in range(3):
        for kw in range(3):
            wd[0, :, kh, kw, :] = Wb[:, :, kh, kw].T
    maskv = rng.standard_normal((Nf, IH, IH, 64))
    dx = torch.full((Nf, IH, IH, 64), 7.0, device="cuda", dtype=torch.bfloat16)      # every pixel must be overwritten
    if mode in (12, 22):
        m = bf(maskv)
    else:                                                                             # bit c%32 of word c/32 = (channel c > 0)
        bits = (maskv > 0).reshape(Nf, IH, IH, 2, 32).astype(np.int64)
        words = (bits << np.arange(32)).sum(-1).astype(np.uint32).view(np.int32)
        m = torch.from_numpy(np.ascontiguousarray(words)).cuda()
    L.check(lib.hulc_k_conv_tile(mode, dY.data_ptr(), bf(wd.reshape(64, -1)).data_ptr(), None, m.data_ptr(), dx.data_ptr(), Nf, OH, IH, 0, None))
    torch.cuda.synchronize()
    ref = conv_dgrad_ref(f64(dY), Wb, 1, IH) * (maskv > 0 if mode in (18, 28) else f64(m) > 0)
    err = np.abs(f64(dx) - ref).reshape(Nf, -1).max(1) / np.abs(ref).max()
    assert err.max() < 6e-3, (err.max(), int(err.argmax()))


# conv2's data gradient (4x4 stride 2 -> four parity classes of 2x2-tap correlations) on conv_reg.h: mode 13 = 16-bit mask, 19 = ReLU bitmask (1 word / pixel)
@pytest.mark.parametrize("IH,Nf,mode", [(49, 1, 13), (49, 5, 19), (49, 300, 19), (20, 1, 13), (20, 7, 19), (20, 1027, 19), (30, 3, 13), (33, 2, 19),
                                        (49, 1, 23), (49, 5, 29), (49, 600, 29), (20, 1, 23), (20, 7, 29), (20, 2051, 29), (30, 3, 23), (33, 2, 29)])    # 23 / 29: two workgroups per CU
def test_conv_reg_dgrad2(IH, Nf, mode):
    L, lib = _lib()
    rng = np.random.default_rng(IH * 3 + Nf)
    OH = (IH - 4) // 2 + 1
    Wb = f64(bf((rng.standard_normal((64, 32, 4, 4)) + np.arange(32)[None, :, None, None] * 0.02) * 0.1))
    dY = bf(rng.standard_normal((Nf, OH, OH, 64)) * (np.arange(64) % 5 + 1))
    wd = np.zeros((4, 32, 2, 2, 64))
    for kh in range(4):
        for kw in range(4):
            wd[(kh % 2) * 2 + kw % 2, :, kh // 2, kw // 2, :] = Wb[:, :, kh, kw].T
    maskv = rng.standard_normal((Nf, IH, IH, 32))
    dx = torch.full((Nf, IH, IH, 32), 7.0, device="cuda", dtype=torch.bfloat16)
    if mode in (13, 23):
        m = bf(maskv)
    else:
        bits = (maskv > 0).astype(np.int64)
        words = (bits << np.arange(32)).sum(-1).astype(np.uint32).view(np.int32)
        m = torch.from_numpy(np.ascontiguousarray(words)).cuda()
    L.check(lib.hulc_k_conv_tile(mode, dY.data_ptr(), bf(wd.reshape(4 * 32, -1)).data_ptr(), None, m.data_ptr(), dx.data_ptr(), Nf, OH, IH, 0, None))
    torch.cuda.synchronize()
    ref = conv_dgrad_ref(f64(dY), Wb, 2, IH) * (maskv > 0 if mode in (19, 29) else f64(m) > 0)
    err = np.abs(f64(dx) - ref).reshape(Nf, -1).max(1) / np.abs(ref).max()
    assert err.max() < 6e-3, (err.max(), int(err.argmax()))


# the fused transformer encoder layer (tr_fused.h: one launch per layer in the 16-bit engines, S <= 32) with dropout ON against (a) the SAME
# 16-bit engine running the unfused kernels (hulc_set_option fused_transformer = 0) and (b) the unfused fp32 engine.  All of them derive every
# dropout mask from the same (seed, step, site, element) hashes, so they apply identical masks: (a) may differ by summation order and by the
# 16-bit roundings that move (tight), (b) by 16-bit rounding altogether (loose) — a wrong mask index, a dropped bias / residual or a mis-saved
# tensor is an O(1) error in both.  The backward (unfused kernels everywhere) consumes what the fused forward saved: gradients must agree too.
# S > 32 (BASELINE config 5, max_window 64): the forward and the FFN half of the backward walk a window as two 32-row halves (tr_fused.h WIDE), the attention
# half of the backward stays on the launch-per-op kernels: (2, 33) = a one-row second half, (3, 48) = a ragged one, (4, 64) = config 5's window.
@pytest.mark.parametrize("B,S", [(1, 4), (3, 7), (2, 16), (3, 17), (8, 32), (2, 33), (3, 48), (4, 64)])
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_fused_transformer_layer_matches_unfused_under_dropout(B, S, dtype):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from hulc_amd import spec
    from hulc_amd.engine import StepEngine
    from hulc_amd.utils import synthetic
    from test_gpu_parity import to_dev
    dims = spec.ModelDims(kind="gcbc", max_window=32 if S <= 32 else 64, use_clip=True)       # GCBC + CLIP: the transformer is the only path into the loss's CLIP term
    P = spec.init_all(dims, seed=13, ln_jitter=True)
    batch = synthetic.make_batch(0, B, S, seed=31)["lang"]
    res = {}
    for tag, dt, fused in (("fp32", "fp32", 0), ("unfused", dtype, 0), ("fused", dtype, 1)):
        eng = StepEngine(dims, B, S, dtype=dt, device="cuda:0", dropout_p=0.1, seed=77)
        eng.set_option("fused_transformer", fused)
        gs = 1.0
        if dt == "fp16":
            gs = 128.0
            eng.scaler_enable(init_scale=gs)
        eng.load_numpy(P)
        eng.zero_grads()
        l = eng.forward_loss(to_dev(batch), True, 1.0, 3.0, step=5)
        eng.backward()
        torch.cuda.synchronize()
        G = {n: t.detach().cpu().numpy() / gs for n, t in eng.views(eng.flat_grads).items() if n.startswith("plan_recognition.")}
        assert all(np.isfinite(g).all() for g in G.values()), tag      # (an fp16 overflow would otherwise hide in NaN comparisons)
        res[tag] = dict(loss=l, x=eng.get_tensor("pr_x_final", B * S * 128), x1=eng.get_tensor("pr_x1", B * S * 128), sf=eng.get_tensor("seq_feat", B * 4096),
                        p0=eng.get_tensor("attn_p0", B * 8 * S * S), dx=np.asarray(eng.get_tensor("tr_dx", B * S * 128), np.float64) / gs, G=G)
        eng.close()
    rel = lambda u, v: float(np.linalg.norm(u.astype(np.float64) - v.astype(np.float64)) / max(np.linalg.norm(v.astype(np.float64)), 1e-30))
    b = res["fused"]
    for ref, tol, tolw in ((res["unfused"], 1e-2 if dtype == "bf16" else 2e-3, 6e-2 if dtype == "bf16" else 1.5e-2), (res["fp32"], 3e-2 if dtype == "bf16" else 6e-3, 0.3 if dtype == "bf16" else 0.08)):
        a = ref
        assert rel(b["p0"], a["p0"]) < tol, rel(b["p0"], a["p0"])
        assert rel(b["x1"], a["x1"]) < tol and rel(b["x"], a["x"]) < tol and rel(b["sf"], a["sf"]) < tol, (rel(b["x1"], a["x1"]), rel(b["x"], a["x"]), rel(b["sf"], a["sf"]))
        assert abs(b["loss"]["clip"] - a["loss"]["clip"]) <= 2e-2 * abs(a["loss"]["clip"]) + 1e-4
        # the transformer's input gradient (every data-gradient stage of both layers behind it: the fused attention-half backward with its second
        # 16-row tile left out produced 50 - 100 % errors here for S <= 16 while the parameter gradients below stayed inside their gates)
        assert rel(b["dx"], a["dx"]) < 4 * tol, rel(b["dx"], a["dx"])
        live = [(rel(b["G"][n], a["G"][n]), n) for n in a["G"] if np.linalg.norm(a["G"][n]) > 1e-8]
        assert (len(live) >= 20) == (B > 1), len(live)      # one window: the CLIP softmax over a single row has zero gradient
        # weight matrices: tolw; bias and LayerNorm vectors (sums of 10^3..10^5 rounded terms that largely cancel — the key bias's gradient is
        # exactly zero in exact arithmetic) twice that
        bad = [(e, n) for e, n in live if e >= (tolw if P[n].ndim == 2 else 2 * tolw)]
        assert not bad, sorted(bad)[-3:]


# ---- a whole 2048-wide recurrence as ONE persistent launch (csrc/rnn_persist.h) against float64, every step checked from the device's own
# previous state (so rounding differences do not compound): forward ReLU / tanh with the Zx residual, backward with the ReLU / tanh derivative
# mask and with / without the dH residual, both time directions; B from one window (one group active, 1 of 16 tile columns) to 128 (16 per
# XCD), B = 13 / 100 (a ragged last group), S = 3 .. 33.  The flag words are reused across the launches of one test (launch_index 1, 2, ...),
# as the engine reuses them across steps.
@pytest.mark.parametrize("B,S", [(1, 3), (7, 5), (13, 9), (64, 32), (100, 4), (128, 6), (32, 33)])
def test_rnn_persist_kernel(B, S):
    import ctypes as C
    L, lib = _lib()
    H = 2048
    rng = np.random.default_rng(B * 100 + S)
    W = bf(rng.standard_normal((H, H)) * 0.03)
    flags = torch.zeros(lib.hulc_k_rnn_persist_flag_words(), dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    Wd = f64(W)
    launch = 0
    for mode, act, rev, use_res in [("fwd", 1, False, True), ("fwd", 2, True, True), ("bwd", 1, False, True), ("bwd", 2, True, False), ("bwd", 1, True, False)]:
        res = bf(rng.standard_normal((S, B, H)) * (0.5 if act == 2 else 1.0))
        mask = bf(rng.standard_normal((S, B, H)) * (0.6 if act == 2 else 1.0))
        q0, dq = (S - 1, -1) if rev else (0, 1)
        X = torch.zeros((S, B, H), dtype=torch.bfloat16, device="cuda")
        X[q0] = bf(np.abs(rng.standard_normal((B, H))) if mode == "fwd" else rng.standard_normal((B, H)))
        launch += 1
        L.check(lib.hulc_k_rnn_persist(X.data_ptr(), W.data_ptr(), res.data_ptr() if use_res else None, mask.data_ptr() if mode == "bwd" else None, B, S, q0, dq, act,
                                       flags.data_ptr(), err.data_ptr(), launch, None))
        torch.cuda.synchronize()
        assert int(err.item()) == 0, "persistent launch reported a timeout / census failure"
        Xd, Rd, Md = f64(X), f64(res), f64(mask)
        for s in range(1, S):
            qp, qc = q0 + (s - 1) * dq, q0 + s * dq
            z = Xd[qp] @ Wd.T + (Rd[qc] if use_res else 0.0)
            if mode == "fwd":
                ref = np.maximum(z, 0) if act == 1 else np.tanh(z)
            else:
                ref = z * (Md[qc] > 0) if act == 1 else z * (1 - Md[qc] ** 2)
            e = np.abs(Xd[qc] - ref).max(1) / (np.abs(ref).max(1) + 1e-6)        # per window
            assert e.max() < 6e-3, (mode, act, rev, s, float(e.max()), int(e.argmax()))


def test_rnn_persist_relaunch_never_consumes_an_earlier_launch_states():
    """From the third step on the state is its own ready flag (rnn_persist.h, RP_TAG): a slice counts as written once the pre-filled pattern is gone.
    The SAME state buffer is therefore driven through many launches with alternating inputs — it always holds the previous launch's valid-looking
    states when a launch starts; every step of every launch must follow from that launch's own previous step, never from what was left there."""
    L, lib = _lib()
    H, B, S = 2048, 64, 32
    rng = np.random.default_rng(7)
    W = bf(rng.standard_normal((H, H)) * 0.03)
    flags = torch.zeros(lib.hulc_k_rnn_persist_flag_words(), dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = [bf(rng.standard_normal((S, B, H))) for _ in range(2)]
    x0 = [bf(np.abs(rng.standard_normal((B, H)))) for _ in range(2)]
    X = torch.zeros((S, B, H), dtype=torch.bfloat16, device="cuda")
    Wf = W.float()
    for launch in range(1, 25):
        k = launch & 1
        X[0] = x0[k]
        L.check(lib.hulc_k_rnn_persist(X.data_ptr(), W.data_ptr(), res[k].data_ptr(), None, B, S, 0, 1, 1, flags.data_ptr(), err.data_ptr(), launch, None))
        torch.cuda.synchronize()
        assert int(err.item()) == 0
        ref = torch.relu(X[:-1].float() @ Wf.T + res[k][1:].float())              # every step from the kernel's own previous state
        e = (X[1:].float() - ref).abs().amax(dim=(1, 2)) / ref.abs().amax(dim=(1, 2))
        assert float(e.max()) < 6e-3, (launch, int(e.argmax()) + 1, float(e.max()))
        assert not bool(torch.isnan(X.float()).any())


# the engine's persistent recurrences (action decoder: 2 layers forward + 2 backward) against the same engine with one launch per time step
# (hulc_set_option persistent_rnn = 0): the two paths sum the 2048 products of a state element in different orders, so states and gradients
# agree to 16-bit rounding noise, not bit for bit
@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("B,S", [(5, 7), (16, 32)])
def test_engine_persistent_recurrence_matches_launch_per_step(B, S, dtype):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from hulc_amd import spec
    from hulc_amd.engine import StepEngine
    from hulc_amd.utils import synthetic
    from test_gpu_parity import to_dev
    dims = spec.ModelDims(kind="hulc", max_window=32, use_clip=False)
    P = spec.init_all(dims, seed=5, ln_jitter=True)
    batch = synthetic.make_batch(B, 0, S, seed=17)["vis"]
    res = {}
    for persist in (1, 0):
        eng = StepEngine(dims, B, S, dtype=dtype, device="cuda:0", dropout_p=0.0, seed=9)
        eng.set_option("persistent_rnn", persist)
        gs = 1.0
        if dtype == "fp16":
            gs = 256.0
            eng.scaler_enable(init_scale=gs)
        eng.load_numpy(P)
        eng.zero_grads()
        l = eng.forward_loss(to_dev(batch), False, 1.0, 0.0, step=3)
        eng.backward()
        torch.cuda.synchronize()
        G = {n: t.detach().cpu().numpy() / gs for n, t in eng.views(eng.flat_grads).items()}
        res[persist] = dict(loss=l, G=G)
        eng.close()
    a, b = res[0], res[1]
    assert abs(a["loss"]["action"] - b["loss"]["action"]) <= 2e-3 * abs(a["loss"]["action"]) + 1e-5
    rel = lambda u, v: float(np.linalg.norm(u.astype(np.float64) - v.astype(np.float64)) / max(np.linalg.norm(v.astype(np.float64)), 1e-30))
    worst = max((rel(b["G"][n], a["G"][n]), n) for n in a["G"] if np.linalg.norm(a["G"][n]) > 1e-8)
    assert worst[0] < (3e-2 if dtype == "bf16" else 1.5e-2), worst
