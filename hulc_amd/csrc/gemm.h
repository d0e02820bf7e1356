// hulc_amd/csrc/gemm.h — the one MFMA GEMM core of the HULC step (gfx950 / CDNA4).
//
//   C[M][N] (+)= epilogue( A[M][K] * B[N][K]^T )       ("NT": both operands reduction-contiguous in LDS)
//
// * 256 threads = 4 waves (2x2), tile BM x BN x 32, one LDS buffer + register prefetch of the next K slab
//   (global latency hides behind the MFMAs of the current slab; two barriers per slab).
// * bf16 path: v_mfma_f32_16x16x32_bf16, fragments by ds_read_b128 from an [row][32+8] LDS image;
//   fp32 path (parity mode): v_mfma_f32_16x16x4_f32 (exact f32 FMA chain), [row][32+2] image.
// * Operands come through *loader* functors, which is how the convolutions become implicit GEMMs without an
//   im2col buffer: a loader maps (row, k) -> global address (NCHW fp32 frames for conv1, NHWC activations
//   for conv2/3, zero-padded dY gathers for dgrad).  "Transposed" loaders fetch reduction-major data
//   (dY[pix][co], patch[pix][k]) with coalesced vector loads and transpose on the LDS write — the wgrad path.
// * Output goes through an *output map* (dense with 2-level row map, conv-dgrad parity scatter) and a fused
//   runtime epilogue: alpha, bias(es), residual (optionally row-broadcast), ReLU, ReLU-mask, dropout,
//   accumulate, fp32 or T store, split-K partial slabs.
#pragma once
#include "common.h"

namespace HULC_NS {

struct EpiP {
    void* out = nullptr;
    int out_f32 = 0;
    int accumulate = 0;
    int atomic = 0;              // fp32 out += via atomicAdd (split-K over workgroups; order-nondeterministic at the ulp level)
    long long z_stride = 0;      // element offset between split-K partial slabs
    const float* bias = nullptr;
    const float* bias2 = nullptr;
    const void* res = nullptr;   // residual, dense [rows][res_ld]
    int res_f32 = 0;
    long long res_ld = 0;
    int res_rowmod = 0;          // >0: residual row = r % res_rowmod (broadcast over time)
    int res_late = 0;            // 0: residual added before relu/mask (RNN); 1: after dropout (transformer x + drop(f(x)))
    const void* mask = nullptr;  // T*, same element offsets as out: v *= (mask > 0)   [mask_tanh: v *= 1 - mask^2]
    int mask_tanh = 0;
    int relu = 0;                // 1: ReLU, 2: tanh
    float alpha = 1.f;
    float drop_p = 0.f;          // inverted dropout applied after relu/mask (seeded by element offset)
    unsigned long long drop_seed = 0;
    // second store of the elements whose offset lies in [out2_lo, out2_hi) (both multiples of 4): out2[o - out2_lo] = T(f(v)), f = ReLU
    // (out2_relu) or the ReLU mask of out2_mask[o - out2_lo].  The batched GEMM in front of a recurrence writes the step that needs no
    // multiplication with it: H[0] = relu(Zx[0]) (h_{-1} = 0) going forward, dZ[S-1] = dH[S-1] * (H[S-1] > 0) going backward — each was a
    // 5 us elementwise launch at the head of a chain of dependent launches
    void* out2 = nullptr;
    long long out2_lo = 0, out2_hi = 0;
    const void* out2_mask = nullptr;
    int out2_relu = 0;
    int generic_only = 0;        // A/B: never take the compact epilogue paths (hulc_set_option "epilogue_fast" 0)
};

// ---------------------------------------------------------------------------------------------------------
// loaders
// ---------------------------------------------------------------------------------------------------------
template <typename T>
struct DenseLoader {   // rows x K, K contiguous; row r lives at (r / R1) * s0 + (r % R1) * s1
    static constexpr bool TRANSPOSED = false;
    const T* p;
    int rows;
    int R1;
    long long s0, s1;
    struct Row { const T* base; bool ok; };
    DEVI int num_rows(int) const { return rows; }
    DEVI Row row(int r, int) const {
        Row c;
        c.ok = r < rows;
        long long off = c.ok ? ((long long)(r / R1) * s0 + (long long)(r % R1) * s1) : 0;
        c.base = p + off;
        return c;
    }
    DEVI void fetch(const Row& c, int k0, int kend, T (&v)[8]) const {
        if (!c.ok || k0 >= kend) { zero8<T>(v); return; }
        load8_guard<T>(c.base + k0, kend - k0, v);
    }
};
template <typename T>
static inline DenseLoader<T> dense(const T* p, int rows, long long ld) {
    DenseLoader<T> l; l.p = p; l.rows = rows; l.R1 = 0x7fffffff; l.s0 = 0; l.s1 = ld; return l;
}
template <typename T>
static inline DenseLoader<T> dense_map(const T* p, int rows, int R1, long long s0, long long s1) {
    DenseLoader<T> l; l.p = p; l.rows = rows; l.R1 = R1; l.s0 = s0; l.s1 = s1; return l;
}

// conv geometry shared by the conv loaders
struct ConvGeom {
    int Nf;          // frames
    int IH, IW, C;   // input
    int OH, OW;      // output
    int KH, KW, S;   // kernel, stride
};

// conv1: fp32 NCHW frames (the boundary layout, hulc.py:395-414), K order = (c, kh, kw) = torch weight order
template <typename T>
struct Conv1Loader {
    static constexpr bool TRANSPOSED = false;
    const float* x;
    ConvGeom g;
    struct Row { const float* base; bool ok; };
    DEVI int num_rows(int) const { return g.Nf * g.OH * g.OW; }
    DEVI Row row(int r, int) const {
        Row c;
        c.ok = r < g.Nf * g.OH * g.OW;
        int n = r / (g.OH * g.OW), rem = r % (g.OH * g.OW);
        int oh = rem / g.OW, ow = rem % g.OW;
        c.base = c.ok ? x + ((long long)n * g.C * g.IH + oh * g.S) * g.IW + ow * g.S : x;
        return c;
    }
    DEVI void fetch(const Row& c, int k0, int kend, T (&v)[8]) const {   // KW == 8: one (c,kh) row per piece
        if (!c.ok || k0 >= kend) { zero8<T>(v); return; }
        int ck = k0 >> 3;
        int ch = ck / g.KH, kh = ck % g.KH;
        const float* p = c.base + ((long long)ch * g.IH + kh) * g.IW;
        float f[8];
        load8<float>(p, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = from_f<T>(f[i]);
    }
};

// conv2/3: NHWC activations of T, K order = (kh, kw, ci); KW*C % 8 == 0
template <typename T>
struct ConvNHWCLoader {
    static constexpr bool TRANSPOSED = false;
    const T* x;
    ConvGeom g;
    struct Row { const T* base; bool ok; };
    DEVI int num_rows(int) const { return g.Nf * g.OH * g.OW; }
    DEVI Row row(int r, int) const {
        Row c;
        c.ok = r < g.Nf * g.OH * g.OW;
        int n = r / (g.OH * g.OW), rem = r % (g.OH * g.OW);
        int oh = rem / g.OW, ow = rem % g.OW;
        c.base = c.ok ? x + (((long long)n * g.IH + oh * g.S) * g.IW + ow * g.S) * g.C : x;
        return c;
    }
    DEVI void fetch(const Row& c, int k0, int kend, T (&v)[8]) const {
        if (!c.ok || k0 >= kend) { zero8<T>(v); return; }
        int rowlen = g.KW * g.C;
        int kh = k0 / rowlen, off = k0 % rowlen;
        load8<T>(c.base + (long long)kh * g.IW * g.C + off, v);
    }
};

// conv dgrad: rows = input pixels of parity class zc=(ph,pw); K order = (a, b, co), taps kh = ph + S*a, kw = pw + S*b
template <typename T>
struct ConvDgradLoader {
    static constexpr bool TRANSPOSED = false;
    const T* dy;     // [Nf][OH][OW][CO]
    ConvGeom g;      // geometry of the forward conv (C = its input channels)
    int CO;
    struct Row { int n, i, j; bool ok; };
    DEVI void cls(int zc, int& ph, int& pw, int& Ic, int& Jc) const {
        ph = zc / g.S; pw = zc % g.S;
        Ic = (g.IH - ph + g.S - 1) / g.S; Jc = (g.IW - pw + g.S - 1) / g.S;
    }
    DEVI int num_rows(int zc) const { int ph, pw, Ic, Jc; cls(zc, ph, pw, Ic, Jc); return g.Nf * Ic * Jc; }
    DEVI Row row(int r, int zc) const {
        int ph, pw, Ic, Jc; cls(zc, ph, pw, Ic, Jc);
        Row c;
        c.ok = r < g.Nf * Ic * Jc;
        c.n = r / (Ic * Jc);
        int rem = r % (Ic * Jc);
        c.i = rem / Jc; c.j = rem % Jc;
        return c;
    }
    DEVI void fetch(const Row& c, int k0, int kend, T (&v)[8]) const {
        if (!c.ok || k0 >= kend) { zero8<T>(v); return; }
        int tap = k0 / CO, co = k0 % CO;
        int TB = g.KW / g.S;
        int a = tap / TB, b = tap % TB;
        int oh = c.i - a, ow = c.j - b;
        if (oh < 0 || oh >= g.OH || ow < 0 || ow >= g.OW) { zero8<T>(v); return; }
        load8<T>(dy + (((long long)c.n * g.OH + oh) * g.OW + ow) * CO + co, v);
    }
};

// ---- transposed loaders (wgrad): a piece is 8 consecutive ROWS at one reduction index (= pixel) ----------
template <typename T>
struct PixMajorLoaderT {   // operand[r][k] = src[k][r]   (dY^T: rows = co, k = pixel); row-contiguous source
    static constexpr bool TRANSPOSED = true;
    struct Row {};
    DEVI Row row(int, int) const { return Row{}; }
    const T* p;
    int rows;        // number of rows (Cout)
    long long ld;    // pixel stride
    DEVI int num_rows(int) const { return rows; }
    DEVI void fetchT(int r0, int k, int kend, T (&v)[8]) const {
        if (r0 >= rows || k >= kend) { zero8<T>(v); return; }
        load8_guard<T>(p + (long long)k * ld + r0, rows - r0, v);
    }
};
template <typename T>
struct Conv1LoaderT {      // operand[r=(c,kh,kw)][k=pixel] from fp32 NCHW frames
    static constexpr bool TRANSPOSED = true;
    struct Row {};
    DEVI Row row(int, int) const { return Row{}; }
    const float* x;
    ConvGeom g;
    DEVI int num_rows(int) const { return g.C * g.KH * g.KW; }
    DEVI void fetchT(int r0, int k, int kend, T (&v)[8]) const {
        if (r0 >= g.C * g.KH * g.KW || k >= kend) { zero8<T>(v); return; }
        int n = k / (g.OH * g.OW), rem = k % (g.OH * g.OW);
        int oh = rem / g.OW, ow = rem % g.OW;
        int ck = r0 >> 3, ch = ck / g.KH, kh = ck % g.KH;
        const float* p = x + (((long long)n * g.C + ch) * g.IH + oh * g.S + kh) * g.IW + ow * g.S;
        float f[8];
        load8<float>(p, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = from_f<T>(f[i]);
    }
};
template <typename T>
struct ConvNHWCLoaderT {   // operand[r=(kh,kw,ci)][k=pixel] from NHWC activations
    static constexpr bool TRANSPOSED = true;
    struct Row {};
    DEVI Row row(int, int) const { return Row{}; }
    const T* x;
    ConvGeom g;
    DEVI int num_rows(int) const { return g.KH * g.KW * g.C; }
    DEVI void fetchT(int r0, int k, int kend, T (&v)[8]) const {
        if (r0 >= g.KH * g.KW * g.C || k >= kend) { zero8<T>(v); return; }
        int n = k / (g.OH * g.OW), rem = k % (g.OH * g.OW);
        int oh = rem / g.OW, ow = rem % g.OW;
        int rowlen = g.KW * g.C;
        int kh = r0 / rowlen, off = r0 % rowlen;
        load8<T>(x + (((long long)n * g.IH + oh * g.S + kh) * g.IW + ow * g.S) * g.C + off, v);
    }
};

// ---------------------------------------------------------------------------------------------------------
// output maps
// ---------------------------------------------------------------------------------------------------------
struct DenseOut {
    int R1;
    long long s0, s1;
    DEVI long long offset(int r, int) const { const unsigned q = (unsigned)r / (unsigned)R1; return (long long)q * s0 + (long long)((unsigned)r - q * (unsigned)R1) * s1; }
};
static inline DenseOut dense_out(long long ld) { DenseOut o; o.R1 = 0x7fffffff; o.s0 = 0; o.s1 = ld; return o; }
static inline DenseOut dense_out_map(int R1, long long s0, long long s1) { DenseOut o; o.R1 = R1; o.s0 = s0; o.s1 = s1; return o; }

struct DgradOut {   // scatter rows of parity class zc back to NHWC input pixels
    ConvGeom g;
    DEVI long long offset(int r, int zc) const {
        int ph = zc / g.S, pw = zc % g.S;
        int Ic = (g.IH - ph + g.S - 1) / g.S, Jc = (g.IW - pw + g.S - 1) / g.S;
        int n = r / (Ic * Jc), rem = r % (Ic * Jc);
        int i = rem / Jc, j = rem % Jc;
        return (((long long)n * g.IH + g.S * i + ph) * g.IW + g.S * j + pw) * g.C;
    }
};

// fused epilogue for one output element: alpha, bias(es), early residual, ReLU, ReLU-mask, dropout, late residual,
// accumulate, fp32 / T store.  `o` = element offset in out (and mask); rrow = residual row.
template <typename T>
DEVI void epi_store(const EpiP& ep, float accv, int rrow, int col, long long o) {
    float v = accv * ep.alpha;
    if (ep.bias) v += ep.bias[col];
    if (ep.bias2) v += ep.bias2[col];
    float resv = 0.f;
    if (ep.res) {
        const long long ro = (long long)rrow * ep.res_ld + col;
        resv = ep.res_f32 ? reinterpret_cast<const float*>(ep.res)[ro] : to_f<T>(reinterpret_cast<const T*>(ep.res)[ro]);
    }
    if (!ep.res_late) v += resv;
    if (ep.relu) v = ep.relu == 2 ? tanhf(v) : fmaxf(v, 0.f);
    if (ep.mask) {
        const float m = to_f<T>(reinterpret_cast<const T*>(ep.mask)[o]);
        v = ep.mask_tanh ? v * (1.f - m * m) : (m > 0.f ? v : 0.f);
    }
    if (ep.drop_p > 0.f) {
        const float u = hash_uniform(ep.drop_seed, (unsigned long long)o);
        v = (u < ep.drop_p) ? 0.f : v * (1.f / (1.f - ep.drop_p));
    }
    if (ep.res_late) v += resv;
    if (ep.out_f32) {
        float* op = reinterpret_cast<float*>(ep.out) + o;
        if (ep.atomic) unsafeAtomicAdd(op, v);
        else *op = ep.accumulate ? (*op + v) : v;
    } else {
        T* op = reinterpret_cast<T*>(ep.out) + o;
        *op = from_f<T>(ep.accumulate ? (to_f<T>(*op) + v) : v);
    }
    if (ep.out2 && o >= ep.out2_lo && o < ep.out2_hi) {
        const long long o2 = o - ep.out2_lo;
        float w = ep.out2_relu ? fmaxf(v, 0.f) : v;
        if (ep.out2_mask) w = to_f<T>(reinterpret_cast<const T*>(ep.out2_mask)[o2]) > 0.f ? w : 0.f;
        reinterpret_cast<T*>(ep.out2)[o2] = from_f<T>(w);
    }
}

// 4 consecutive output columns of one row (the MFMA is issued as D^T = B A^T, so a lane owns C[row][col..col+3]):
// vector bias / residual / mask loads and one 16-byte (fp32) or 8-byte (T = bf16) store when aligned, else the scalar path.
// The vector path is split in two so that a kernel can issue the operand loads early (skinny_lds_kernel: before it waits for
// its A/W stream) and apply them after the reduction.
struct EpiPre4 {
    bool vec;
    float b[4], rv[4], mk[4];
};
template <typename T>
DEVI EpiPre4 epi_prefetch4(const EpiP& ep, int rrow, int col, int N, long long o) {
    EpiPre4 p;
    p.vec = (col + 3 < N) && ((o & 3) == 0) && !ep.atomic &&
            (!ep.res || (((long long)rrow * ep.res_ld + col) & 3) == 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) { p.b[r] = 0.f; p.rv[r] = 0.f; p.mk[r] = 1.f; }
    if (!p.vec) return p;
    if (ep.bias) { const float4 b = *reinterpret_cast<const float4*>(ep.bias + col); p.b[0] += b.x; p.b[1] += b.y; p.b[2] += b.z; p.b[3] += b.w; }
    if (ep.bias2) { const float4 b = *reinterpret_cast<const float4*>(ep.bias2 + col); p.b[0] += b.x; p.b[1] += b.y; p.b[2] += b.z; p.b[3] += b.w; }
    if (ep.res) {
        const long long ro = (long long)rrow * ep.res_ld + col;
        if (ep.res_f32) { const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ep.res) + ro); p.rv[0] = x.x; p.rv[1] = x.y; p.rv[2] = x.z; p.rv[3] = x.w; }
        else {
            T t[4];
            if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(t) = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(ep.res) + ro);
            else *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(reinterpret_cast<const T*>(ep.res) + ro);
#pragma unroll
            for (int r = 0; r < 4; ++r) p.rv[r] = to_f<T>(t[r]);
        }
    }
    if (ep.mask) {
        T t[4];
        if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(t) = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(ep.mask) + o);
        else *reinterpret_cast<float4*>(t) = *reinterpret_cast<const float4*>(reinterpret_cast<const T*>(ep.mask) + o);
#pragma unroll
        for (int r = 0; r < 4; ++r) p.mk[r] = to_f<T>(t[r]);
    }
    return p;
}
template <typename T>
DEVI void epi_apply4(const EpiP& ep, const EpiPre4& p, const float (&accv)[4], int rrow, int col, int N, long long o) {
    if (!p.vec) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (col + r < N) epi_store<T>(ep, accv[r], rrow, col + r, o + r);
        return;
    }
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = accv[r] * ep.alpha + p.b[r];
    if (!ep.res_late) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p.rv[r];
    }
    if (ep.relu == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = tanhf(v[r]);
    } else if (ep.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    if (ep.mask) {
        if (ep.mask_tanh) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= 1.f - p.mk[r] * p.mk[r];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = p.mk[r] > 0.f ? v[r] : 0.f;
        }
    }
    if (ep.drop_p > 0.f) {           // same element index and scale as epi_store (the backward re-derives the mask from them)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = hash_uniform(ep.drop_seed, (unsigned long long)(o + r)) < ep.drop_p ? 0.f : v[r] * (1.f / (1.f - ep.drop_p));
    }
    if (ep.res_late) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += p.rv[r];
    }
    if (ep.out_f32) {
        float4* op = reinterpret_cast<float4*>(reinterpret_cast<float*>(ep.out) + o);
        float4 w = make_float4(v[0], v[1], v[2], v[3]);
        if (ep.accumulate) { const float4 old = *op; w.x += old.x; w.y += old.y; w.z += old.z; w.w += old.w; }
        *op = w;
    } else {
        T* op = reinterpret_cast<T*>(ep.out) + o;
        if (ep.accumulate) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += to_f<T>(op[r]);
        }
        if constexpr (sizeof(T) == 2) {
            uint2 w;
            w.x = pack2h(v[0], v[1]);
            w.y = pack2h(v[2], v[3]);
            *reinterpret_cast<uint2*>(op) = w;
        } else {
            *reinterpret_cast<float4*>(op) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
    if (ep.out2 && o >= ep.out2_lo && o < ep.out2_hi) {          // (accumulate is never combined with out2)
        const long long o2 = o - ep.out2_lo;
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = ep.out2_relu ? fmaxf(v[r], 0.f) : v[r];
        T* op2 = reinterpret_cast<T*>(ep.out2) + o2;
        if (ep.out2_mask) {
            const T* mp = reinterpret_cast<const T*>(ep.out2_mask) + o2;
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = to_f<T>(mp[r]) > 0.f ? w[r] : 0.f;
        }
        if constexpr (sizeof(T) == 2) {
            uint2 u;
            u.x = pack2h(w[0], w[1]);
            u.y = pack2h(w[2], w[3]);
            *reinterpret_cast<uint2*>(op2) = u;
        } else {
            *reinterpret_cast<float4*>(op2) = make_float4(w[0], w[1], w[2], w[3]);
        }
    }
}
// The 16-bit-output forms of the decoder's batched GEMMs (input projections with bias / bias2 / a broadcast residual, data gradients; each with the optional SECOND store
// of the rows the following recurrence needs unmultiplied: relu(v) or v under a ReLU mask) as one compact function.  Same arithmetic, in the same order, as the vector
// path of epi_prefetch4 / epi_apply4 with alpha = 1 and no mask / activation / dropout.  Why it exists: the generic path, unrolled over a wave's 8 accumulator quads, is ~12 K
// instructions that every wave executes exactly once — tools/gemm_probe.hip: a feature-less 16-bit store costs 31.4 us through it against 24.6 us with a plain store.
DEVI bool epi_is_fast16(const EpiP& ep) {
    return !ep.generic_only && !ep.out_f32 && !ep.accumulate && !ep.atomic && ep.z_stride == 0 && !(ep.mask && ep.mask_tanh) && ep.relu != 2 && ep.drop_p == 0.f && ep.alpha == 1.f &&
           (!ep.res || ((ep.res_ld & 3) == 0 && !ep.res_late)) && ((uintptr_t)ep.out & 7) == 0 && (!ep.mask || ((uintptr_t)ep.mask & 7) == 0) &&
           (!ep.out2 || (((uintptr_t)ep.out2 & 7) == 0 && (ep.out2_lo & 3) == 0));
}
// operands first (kernels that fetch them under their operand stream), arithmetic + stores later
struct EpiF16Pre { float b[4], rv[4]; uint2 mk; };
DEVI EpiF16Pre epi_fast16_pre(const EpiP& ep, int rrow, int col, long long o) {
    EpiF16Pre p;
#pragma unroll
    for (int r = 0; r < 4; ++r) { p.b[r] = 0.f; p.rv[r] = 0.f; }
    p.mk = uint2{0u, 0u};
    if (ep.bias) { const float4 t = *reinterpret_cast<const float4*>(ep.bias + col); p.b[0] += t.x; p.b[1] += t.y; p.b[2] += t.z; p.b[3] += t.w; }
    if (ep.bias2) { const float4 t = *reinterpret_cast<const float4*>(ep.bias2 + col); p.b[0] += t.x; p.b[1] += t.y; p.b[2] += t.z; p.b[3] += t.w; }
    if (ep.res) {
        const long long ro = (long long)rrow * ep.res_ld + col;
        if (ep.res_f32) { const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ep.res) + ro); p.rv[0] = x.x; p.rv[1] = x.y; p.rv[2] = x.z; p.rv[3] = x.w; }
        else {
            const uint2 x = *reinterpret_cast<const uint2*>(reinterpret_cast<const h16_t*>(ep.res) + ro);
            p.rv[0] = h2f_lo(x.x); p.rv[1] = h2f_hi(x.x); p.rv[2] = h2f_lo(x.y); p.rv[3] = h2f_hi(x.y);
        }
    }
    if (ep.mask) p.mk = *reinterpret_cast<const uint2*>(reinterpret_cast<const h16_t*>(ep.mask) + o);
    return p;
}
DEVI void epi_fast16_apply(const EpiP& ep, const EpiF16Pre& p, const f32x4& acc, long long o) {
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = acc[r] * ep.alpha + p.b[r];
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] += p.rv[r];
    if (ep.relu) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    if (ep.mask) {
        v[0] = h2f_lo(p.mk.x) > 0.f ? v[0] : 0.f; v[1] = h2f_hi(p.mk.x) > 0.f ? v[1] : 0.f; v[2] = h2f_lo(p.mk.y) > 0.f ? v[2] : 0.f; v[3] = h2f_hi(p.mk.y) > 0.f ? v[3] : 0.f;
    }
    *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(ep.out) + o) = uint2{pack2h(v[0], v[1]), pack2h(v[2], v[3])};
    if (ep.out2 && o >= ep.out2_lo && o < ep.out2_hi) {
        const long long o2 = o - ep.out2_lo;
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) w[r] = ep.out2_relu ? fmaxf(v[r], 0.f) : v[r];
        if (ep.out2_mask) {
            const uint2 m = *reinterpret_cast<const uint2*>(reinterpret_cast<const h16_t*>(ep.out2_mask) + o2);
            w[0] = h2f_lo(m.x) > 0.f ? w[0] : 0.f; w[1] = h2f_hi(m.x) > 0.f ? w[1] : 0.f; w[2] = h2f_lo(m.y) > 0.f ? w[2] : 0.f; w[3] = h2f_hi(m.y) > 0.f ? w[3] : 0.f;
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(ep.out2) + o2) = uint2{pack2h(w[0], w[1]), pack2h(w[2], w[3])};
    }
}
DEVI void epi_fast16_4(const EpiP& ep, const f32x4& acc, int rrow, int col, long long o) {
    const EpiF16Pre p = epi_fast16_pre(ep, rrow, col, o);
    epi_fast16_apply(ep, p, acc, o);
}
// the weight-gradient form of an epilogue: fp32 store or accumulate of the bare product (alpha 1, no bias / residual / mask / activation / dropout / second store).
// Wave-uniform; the generic epi_store4 path costs gemm_glds_kernel 2.2 us of its 28 at 2048^3 (tools/gemm_probe.hip: flag tests, prefetch structure, 64-bit offsets)
DEVI bool epi_is_plain_f32(const EpiP& ep) {
    return !ep.generic_only && ep.out_f32 && !ep.atomic && ep.z_stride == 0 && !ep.bias && !ep.bias2 && !ep.res && !ep.mask && !ep.relu && ep.drop_p == 0.f && !ep.out2 && ep.alpha == 1.f;
}
DEVI void epi_plain4(const EpiP& ep, const f32x4& acc, long long o) {
    f32x4* op = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + o);
    *op = ep.accumulate ? *op + acc : acc;
}
template <typename T>
DEVI void epi_store4(const EpiP& ep, const float (&accv)[4], int rrow, int col, int N, long long o) {
    const EpiPre4 p = epi_prefetch4<T>(ep, rrow, col, N, o);
    epi_apply4<T>(ep, p, accv, rrow, col, N, o);
}

// ---------------------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------------------
template <typename T, int BK> struct LdsLd;
template <int BK> struct LdsLd<h16_t, BK> { static constexpr int v = BK + 16; };  // 96 / 160 B rows: pitch = 2 (mod 4) 16-B slots -> conflict-free ds_read_b128 (MI355X_MICROARCH.md §LDS)
template <int BK> struct LdsLd<float, BK> { static constexpr int v = BK + 2; };    // 2*row + g distinct banks for b32 reads

// piece numbering inside a [BR rows][32 k] operand tile:
//   normal loader:      q -> row = q >> 2, k-piece = q & 3          (8 contiguous k per piece)
//   transposed loader:  q -> row-group = q % (BR/8), kk = q / (BR/8) (8 consecutive rows at one k)
template <typename T, int BR, typename L, int BK>
DEVI void tile_fetch(const L& l, const typename L::Row& rc, int r0blk, int q, int k, int kend, T (&v)[8]) {
    if constexpr (L::TRANSPOSED) {
        l.fetchT(r0blk + (q % (BR / 8)) * 8, k + q / (BR / 8), kend, v);
    } else {
        l.fetch(rc, k + (q % (BK / 8)) * 8, kend, v);
    }
}
template <typename T, int BR, bool TRANSPOSED, int BK>
DEVI void tile_commit(T* S, const T (&v)[8], int q) {
    constexpr int LD = LdsLd<T, BK>::v;
    if constexpr (TRANSPOSED) {
        const int rg = q % (BR / 8), kk = q / (BR / 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) S[(rg * 8 + e) * LD + kk] = v[e];
    } else {
        T* d = S + (q / (BK / 8)) * LD + (q % (BK / 8)) * 8;
        if constexpr (sizeof(T) == 2) {
            *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(v);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) reinterpret_cast<float2*>(d)[e] = reinterpret_cast<const float2*>(v)[e];
        }
    }
}

template <typename T, int BM, int BN, typename AL, typename BL, typename OM, int BK = 32>
__global__ void __launch_bounds__(256) gemm_kernel(AL al, BL bl, OM om, EpiP ep, int N, int K, int nsplit, int ksplit) {
    constexpr int LD = LdsLd<T, BK>::v;
    constexpr int PPR = BK / 8;                                  // 8-element pieces per tile row
    constexpr int APT = (BM * PPR + 255) / 256;
    constexpr int BPT = (BN * PPR + 255) / 256;
    constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    __shared__ __attribute__((aligned(16))) T smem[(BM + BN) * LD];
    __shared__ long long rowoff[BM];      // output row offsets: the (division-heavy) output map is evaluated once per row
    T* As = smem;
    T* Bs = smem + BM * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int zs = blockIdx.z % nsplit, zc = blockIdx.z / nsplit;
    const int M = al.num_rows(zc);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= M) return;
    const int kbeg = zs * ksplit;
    const int kend = min(K, kbeg + ksplit);

    typename AL::Row arow[APT];
    typename BL::Row brow[BPT];
#pragma unroll
    for (int i = 0; i < APT; ++i) arow[i] = al.row(m0 + (tid + i * 256) / PPR, zc);
#pragma unroll
    for (int i = 0; i < BPT; ++i) brow[i] = bl.row(n0 + (tid + i * 256) / PPR, zc);

    T ra[APT][8], rb[BPT][8];
    auto fetch_all = [&](int k) {
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int q = tid + i * 256;
            if (q < BM * PPR) tile_fetch<T, BM, AL, BK>(al, arow[i], m0, q, k, kend, ra[i]);
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int q = tid + i * 256;
            if (q < BN * PPR) tile_fetch<T, BN, BL, BK>(bl, brow[i], n0, q, k, kend, rb[i]);
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int wm = wave >> 1, wn = wave & 1;
    if (kbeg < kend) fetch_all(kbeg);
    for (int k = kbeg; k < kend; k += BK) {
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int q = tid + i * 256;
            if (q < BM * PPR) tile_commit<T, BM, AL::TRANSPOSED, BK>(As, ra[i], q);
        }
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int q = tid + i * 256;
            if (q < BN * PPR) tile_commit<T, BN, BL::TRANSPOSED, BK>(Bs, rb[i], q);
        }
        __syncthreads();
        if (k + BK < kend) fetch_all(k + BK);
        if constexpr (sizeof(T) == 2) {
            const T* ap = As + (wm * WM + (lane & 15)) * LD + (lane >> 4) * 8;
            const T* bp = Bs + (wn * WN + (lane & 15)) * LD + (lane >> 4) * 8;
#pragma unroll
            for (int kk = 0; kk < BK / 32; ++kk) {
                h16x8_t a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const h16x8_t*>(ap + i * 16 * LD + kk * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const h16x8_t*>(bp + j * 16 * LD + kk * 32);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = MFMA_16x16x32_H(b[j], a[i], acc[i][j], 0, 0, 0);   // D^T: lane owns 4 consecutive columns
            }
        } else {
            const T* ap = As + (wm * WM + (lane & 15)) * LD + (lane >> 4);
            const T* bp = Bs + (wn * WN + (lane & 15)) * LD + (lane >> 4);
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = ap[i * 16 * LD + kk * 4];
#pragma unroll
                for (int j = 0; j < TN; ++j) b[j] = bp[j * 16 * LD + kk * 4];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: (operands swapped) lane holds C[row = lane&15][col = (lane>>4)*4 .. +3] of each 16x16 tile
    if (tid < BM) rowoff[tid] = (m0 + tid < M) ? om.offset(m0 + tid, zc) : 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * WM + i * 16 + (lane & 15);
        if (row < M) {
            const long long obase = rowoff[row - m0] + (long long)zs * ep.z_stride;
            const int rrow = ep.res_rowmod > 0 ? row % ep.res_rowmod : row;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * WN + j * 16 + (lane >> 4) * 4;
                if (col < N) {
                    const float v4[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    epi_store4<T>(ep, v4, rrow, col, N, obase + col);
                }
            }
        }
    }
}

// host launcher: grid.x over M tiles, grid.y over N tiles, grid.z = classes * nsplit
template <typename T, int BM, int BN, typename AL, typename BL, typename OM, int BK = 32>
static inline void launch_gemm(hipStream_t st, const AL& al, const BL& bl, const OM& om, const EpiP& ep, int Mmax, int N,
                               int K, int nclass = 1, int nsplit = 1) {
    if (Mmax <= 0 || N <= 0) return;
    int ksplit = K;
    if (nsplit > 1) ksplit = ((K + nsplit - 1) / nsplit + BK - 1) / BK * BK;   // empty slabs still write zeros
    dim3 grid((Mmax + BM - 1) / BM, (N + BN - 1) / BN, nclass * nsplit);
    hipLaunchKernelGGL((gemm_kernel<T, BM, BN, AL, BL, OM, BK>), grid, dim3(256), 0, st, al, bl, om, ep, N, K, nsplit, ksplit);
}

// ---------------------------------------------------------------------------------------------------------
// Dense bf16 NT GEMM, 128x128 tile, operands by LDS-DMA (global_load_lds_dwordx4) into a 3-stage LDS ring.
//   The 2048-square RNN GEMMs give exactly one 128x128 tile per CU (4 waves): with a single LDS buffer + register prefetch the
//   L2 latency of every k-step is exposed (64 us at K = 2048 against a 20 us per-CU load floor).  Here two k-steps (BK = 64) are
//   in flight while a third is multiplied; one raw s_barrier per k-step, counted vmcnt (8 DMA instructions per wave per stage).
//   A stage = 16 + 16 pieces of 8 rows x 128 B; piece-local swizzle on the SOURCE address (chunk = slot ^ (row & 6)) makes the
//   ds_read_b128 fragment reads conflict-free (see skinny_lds_kernel).  K % 32 == 0 (a trailing half step multiplies only 32 k).
//   Workgroup -> tile order is XCD-aware: block b runs on XCD b % 8, which gets a contiguous run of tiles in 4-row groups, so
//   each XCD's L2 sees 4 A panels x 8 B panels instead of the whole of B.
// ---------------------------------------------------------------------------------------------------------
template <int NST, int NW>      // NW = 4 (wave tile 64x64) or 8 waves (wave tile 32x64: two waves per SIMD hide each other's DMA issue / LDS latency)
DEVI void gemm_glds_body(const DenseLoader<h16_t>& al, const DenseLoader<h16_t>& bl, const DenseOut& om, const EpiP& ep, int M, int N, int K, int tiles_m, int tiles_n, int bid) {
    constexpr int STAGE = 32 * 1024;
    constexpr int PW = 16 / NW;                  // 8-row pieces of each operand a wave DMAs per stage
    constexpr int TM = 8 / NW * 2 * 2 / 2;       // m-tiles per wave: 4 (NW = 4) or 2 (NW = 8)
    constexpr int WROWS = TM * 16;               // rows of the tile a wave multiplies
    extern __shared__ __attribute__((aligned(16))) char gg_smem[];
    typedef __attribute__((address_space(3))) char lchar;
    lchar* lds = (lchar*)gg_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int tm, tn;
    {
        const int nt = tiles_m * tiles_n, per = nt / 8, rem = nt % 8;
        const int x = bid % 8, q = bid / 8;
        const int tile = x * per + min(x, rem) + q;
        constexpr int GM = 4;
        const int gsz = GM * tiles_n, grp = tile / gsz, first_m = grp * GM, gm = min(GM, tiles_m - first_m);
        tm = first_m + (tile % gsz) % gm;
        tn = (tile % gsz) / gm;
    }
    const int m0 = tm * 128, n0 = tn * 128;
    const int r = lane >> 3, cs = (lane & 7) ^ (r & 6);
    const h16_t* asrc[PW];
    const h16_t* bsrc[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        asrc[j] = al.row(min(m0 + (wave * PW + j) * 8 + r, M - 1), 0).base + cs * 8;
        bsrc[j] = bl.row(min(n0 + (wave * PW + j) * 8 + r, N - 1), 0).base + cs * 8;
    }
    const int nk = (K + 63) >> 6;
    const bool khalf = (K & 63) != 0;
    auto issue = [&](int kt, int buf) {
        lchar* st = lds + buf * STAGE + wave * PW * 1024;
        long long ko = (long long)kt * 64;
        if (khalf && kt == nk - 1 && cs >= 4) ko -= 32;         // chunk beyond K: fetch a valid one instead (never multiplied)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            lds_dma16(asrc[j] + ko, (unsigned)(size_t)(st + j * 1024));
            lds_dma16(bsrc[j] + ko, (unsigned)(size_t)(st + 16384 + j * 1024));
        }
    };
    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 1, wn = wave & 1;
    const int foff = (li >> 3) * 1024 + (li & 7) * 128;
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
        if (j < nk) issue(j, j);
    int buf = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt of THIS wave has landed (later stages may still be in flight: 2*PW DMA instructions each); the barrier makes that
        // true for every wave and also says every wave has finished multiplying stage kt-1, whose buffer the next DMA overwrites
        const int ahead = min(NST - 2, nk - 1 - kt);
        if constexpr (PW == 4) {
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (kt + NST - 1 < nk) issue(kt + NST - 1, buf == 0 ? NST - 1 : buf - 1);
        lchar* sa = lds + buf * STAGE + wm * (WROWS / 8) * 1024 + foff;
        lchar* sb = lds + buf * STAGE + 16384 + wn * 8192 + foff;
        const int nkk = (khalf && kt == nk - 1) ? 1 : 2;
#pragma unroll 1
        for (int kk = 0; kk < nkk; ++kk) {
            const int chunk = (((kk << 2) + g) ^ (li & 6)) << 4;
            h16x8_t a[TM], b[4];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(__attribute__((address_space(3))) h16x8_t*)(sa + i * 2048 + chunk);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(__attribute__((address_space(3))) h16x8_t*)(sb + j * 2048 + chunk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MFMA_16x16x32_H(b[j], a[i], acc[i][j], 0, 0, 0);   // D^T: lane owns 4 consecutive columns
        }
        buf = buf == NST - 1 ? 0 : buf + 1;
    }
    const bool plain = epi_is_plain_f32(ep) && (om.s1 & 3) == 0 && (om.s0 & 3) == 0 && ((uintptr_t)ep.out & 15) == 0;
    const bool fast16 = epi_is_fast16(ep) && (om.s1 & 3) == 0 && (om.s0 & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * WROWS + i * 16 + li;
        if (row < M) {
            const long long obase = om.offset(row, 0);
            const int rrow = ep.res_rowmod > 0 ? row % ep.res_rowmod : row;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + g * 4;
                if (col < N) {
                    if (plain && col + 3 < N) { epi_plain4(ep, acc[i][j], obase + col); continue; }
                    if (fast16 && col + 3 < N) { epi_fast16_4(ep, acc[i][j], rrow, col, obase + col); continue; }
                    const float v4[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    epi_store4<h16_t>(ep, v4, rrow, col, N, obase + col);
                }
            }
        }
    }
}
template <int NST, int NW>
__global__ void __launch_bounds__(NW * 64) gemm_glds_kernel(DenseLoader<h16_t> al, DenseLoader<h16_t> bl, DenseOut om, EpiP ep, int M, int N, int K,
                                                           int tiles_m, int tiles_n) {
    gemm_glds_body<NST, NW>(al, bl, om, ep, M, N, K, tiles_m, tiles_n, (int)blockIdx.x);
}
// ---------------------------------------------------------------------------------------------------------
// Up to three INDEPENDENT gemm_glds problems as one launch (round 6): the decoder backward issues dW_hh1, dW_ih1 and dH0 = dZ1 W_ih1 back to
// back — three 2048^3 products of 256 tiles each, none reading another's output.  As three launches every one of them pays the stream's kernel boundary (all 256
// workgroups of launch i drain their epilogue stores before the first DMA of launch i + 1 is issued) and its own pipeline fill; as ONE grid of 768 workgroups a CU
// starts its next tile the moment its previous workgroup retires.  A workgroup finds its problem from blockIdx.x (uniform: the descriptors stay in scalar
// registers); problem p's tile order is the single launch's (first tile a multiple of 8 -> block b still runs on XCD b % 8).
// ---------------------------------------------------------------------------------------------------------
struct GemmGroupP { DenseLoader<h16_t> a[3], b[3]; DenseOut om[3]; EpiP ep[3]; int M[3], N[3], K[3], tm[3], tn[3], t0[3]; int n; };
template <int NST, int NW>
__global__ void __launch_bounds__(NW * 64) gemm_glds_group_kernel(GemmGroupP g) {
    const int b = (int)blockIdx.x;
    const int p = (g.n > 2 && b >= g.t0[2]) ? 2 : ((g.n > 1 && b >= g.t0[1]) ? 1 : 0);      // uniform
    const int bid = b - g.t0[p];
    if (bid >= g.tm[p] * g.tn[p]) return;                 // a problem's block range is padded to a multiple of 8 (XCD alignment of the next problem)
    gemm_glds_body<NST, NW>(g.a[p], g.b[p], g.om[p], g.ep[p], g.M[p], g.N[p], g.K[p], g.tm[p], g.tn[p], bid);
}
// ---------------------------------------------------------------------------------------------------------
// TWO NT GEMMs that share their A operand up to a shift along K, as one launch (round 5):
//     C1[m][n] = sum_{k = sh}^{K-1} A[m][k] B1[n][k - sh]          C2[m][n] = sum_{k = 0}^{K-1} A[m][k] B2[n][k]
// — the two weight gradients of the decoder's layer 1, dW_hh1 = dZ1[1:]^T H1[:-1] and dW_ih1 = dZ1^T H0 (A = dZ1^T, k = (t, b) token, sh = B tokens = one
// time step).  gemm_glds_kernel is bound by the bytes a CU pulls through its load path (1 MB per 128 x 128 tile at K = 2048 -> ~28 us); here a workgroup owns
// the 128 x 128 tile of BOTH outputs and streams A once: 1.5 MB for what took 2 MB.  Same ring, swizzle and counted waits as above with three operands per
// stage (48 KB x 3 stages = 144 KB of LDS); sh and K are multiples of the 64-wide k-step, so B1's stage kt is simply B1's own step kt - sh / 64 (the first
// sh / 64 steps fetch a valid dummy stage and skip B1's MFMAs).
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) gemm_glds_pair_kernel(DenseLoader<h16_t> al, DenseLoader<h16_t> bl1, DenseLoader<h16_t> bl2, DenseOut om, EpiP ep1, EpiP ep2,
                                                           int M, int N, int K, int shs, int tiles_m, int tiles_n) {
    constexpr int NST = 3, NW = 8, STAGE = 48 * 1024, PW = 2, TM = 2, WROWS = 32;
    extern __shared__ __attribute__((aligned(16))) char gg_smem[];
    typedef __attribute__((address_space(3))) char lchar;
    lchar* lds = (lchar*)gg_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int tm, tn;
    {
        const int nt = tiles_m * tiles_n, per = nt / 8, rem = nt % 8;
        const int x = blockIdx.x % 8, q = blockIdx.x / 8;
        const int tile = x * per + min(x, rem) + q;
        constexpr int GM = 4;
        const int gsz = GM * tiles_n, grp = tile / gsz, first_m = grp * GM, gm = min(GM, tiles_m - first_m);
        tm = first_m + (tile % gsz) % gm;
        tn = (tile % gsz) / gm;
    }
    const int m0 = tm * 128, n0 = tn * 128;
    const int r = lane >> 3, cs = (lane & 7) ^ (r & 6);
    const h16_t* asrc[PW];
    const h16_t* b1src[PW];
    const h16_t* b2src[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        asrc[j] = al.row(min(m0 + (wave * PW + j) * 8 + r, M - 1), 0).base + cs * 8;
        b1src[j] = bl1.row(min(n0 + (wave * PW + j) * 8 + r, N - 1), 0).base + cs * 8;
        b2src[j] = bl2.row(min(n0 + (wave * PW + j) * 8 + r, N - 1), 0).base + cs * 8;
    }
    const int nk = K >> 6;
    auto issue = [&](int kt, int buf) {
        lchar* st = lds + buf * STAGE + wave * PW * 1024;
        const long long ko = (long long)kt * 64, ko1 = (long long)max(kt - shs, 0) * 64;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            lds_dma16(asrc[j] + ko, (unsigned)(size_t)(st + j * 1024));
            lds_dma16(b1src[j] + ko1, (unsigned)(size_t)(st + 16384 + j * 1024));
            lds_dma16(b2src[j] + ko, (unsigned)(size_t)(st + 32768 + j * 1024));
        }
    };
    f32x4 acc1[TM][4], acc2[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc1[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const int wm = wave >> 1, wn = wave & 1;
    const int foff = (li >> 3) * 1024 + (li & 7) * 128;
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
        if (j < nk) issue(j, j);
    int buf = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = min(NST - 2, nk - 1 - kt);             // stages in flight behind stage kt: 3 * PW = 6 DMA instructions per wave each
        if (ahead >= 1) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + NST - 1 < nk) issue(kt + NST - 1, buf == 0 ? NST - 1 : buf - 1);
        lchar* sa = lds + buf * STAGE + wm * (WROWS / 8) * 1024 + foff;
        lchar* sb1 = lds + buf * STAGE + 16384 + wn * 8192 + foff;
        lchar* sb2 = lds + buf * STAGE + 32768 + wn * 8192 + foff;
        const bool with1 = kt >= shs;
#pragma unroll 1
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = (((kk << 2) + g) ^ (li & 6)) << 4;
            // all ten fragments are requested before the first MFMA (a second read phase behind B2's MFMAs exposed the LDS latency once more per k-half)
            h16x8_t a[TM], b[4], c[4];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(__attribute__((address_space(3))) h16x8_t*)(sa + i * 2048 + chunk);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(__attribute__((address_space(3))) h16x8_t*)(sb2 + j * 2048 + chunk);
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = *(__attribute__((address_space(3))) h16x8_t*)(sb1 + j * 2048 + chunk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc2[i][j] = MFMA_16x16x32_H(b[j], a[i], acc2[i][j], 0, 0, 0);
            if (with1) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc1[i][j] = MFMA_16x16x32_H(c[j], a[i], acc1[i][j], 0, 0, 0);
            }
        }
        buf = buf == NST - 1 ? 0 : buf + 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * WROWS + i * 16 + li;
        if (row < M) {
            const long long obase = om.offset(row, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + g * 4;
                if (col < N) {
                    if (((om.s1 | om.s0) & 3) == 0 && col + 3 < N) { epi_plain4(ep1, acc1[i][j], obase + col); epi_plain4(ep2, acc2[i][j], obase + col); continue; }      // plain by gemm_glds_pair_ok
                    const float v1[4] = {acc1[i][j][0], acc1[i][j][1], acc1[i][j][2], acc1[i][j][3]};
                    epi_store4<h16_t>(ep1, v1, row, col, N, obase + col);
                    const float v2[4] = {acc2[i][j][0], acc2[i][j][1], acc2[i][j][2], acc2[i][j][3]};
                    epi_store4<h16_t>(ep2, v2, row, col, N, obase + col);
                }
            }
        }
    }
}
// both outputs [M][N] through the same DenseOut; plain epilogues only (fp32 store / accumulate: the weight-gradient form)
static inline bool gemm_glds_pair_ok(const DenseLoader<h16_t>& a, const DenseLoader<h16_t>& b1, const DenseLoader<h16_t>& b2, const EpiP& ep1, const EpiP& ep2, int M, int N, int K, int sh) {
    auto row_ok = [](const DenseLoader<h16_t>& l) { return (l.s0 % 8) == 0 && (l.s1 % 8) == 0 && ((uintptr_t)l.p % 16) == 0 && l.R1 == 0x7fffffff; };
    // epi_plain4 is a 16-byte f32x4 load / store of the bare product: same conditions as the single-GEMM plain path (alpha 1, 16-byte aligned output)
    auto plain = [](const EpiP& e) {
        return !e.generic_only && e.out_f32 && !e.atomic && e.z_stride == 0 && !e.bias && !e.bias2 && !e.res && !e.mask && !e.relu && e.drop_p == 0.f && !e.out2 && e.alpha == 1.f && ((uintptr_t)e.out & 15) == 0;
    };
    return K >= 192 && (K % 64) == 0 && sh > 0 && (sh % 64) == 0 && sh < K && (M % 128) == 0 && (N % 128) == 0 && row_ok(a) && row_ok(b1) && row_ok(b2) && plain(ep1) && plain(ep2);
}
static inline void launch_gemm_glds_pair(hipStream_t st, const DenseLoader<h16_t>& a, const DenseLoader<h16_t>& b1, const DenseLoader<h16_t>& b2, const DenseOut& om, const EpiP& ep1,
                                         const EpiP& ep2, int M, int N, int K, int sh) {
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute((const void*)gemm_glds_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024); attr_set = true; }
    const int tiles_m = M / 128, tiles_n = N / 128;
    hipLaunchKernelGGL(gemm_glds_pair_kernel, dim3(tiles_m * tiles_n), dim3(512), 144 * 1024, st, a, b1, b2, om, ep1, ep2, M, N, K, sh / 64, tiles_m, tiles_n);
}
static inline bool gemm_glds_ok(const DenseLoader<h16_t>& a, const DenseLoader<h16_t>& b, const EpiP& ep, int M, int N, int K) {
    auto row_ok = [](const DenseLoader<h16_t>& l) { return (l.s0 % 8) == 0 && (l.s1 % 8) == 0 && ((uintptr_t)l.p % 16) == 0; };
    return K >= 64 && (K % 32) == 0 && ep.z_stride == 0 && row_ok(a) && row_ok(b);
}
static inline void launch_gemm_glds(hipStream_t st, const DenseLoader<h16_t>& a, const DenseLoader<h16_t>& b, const DenseOut& om, const EpiP& ep, int M, int N, int K) {
    static const int nw = HULC_SWITCH("HULC_GLDS_NW", 8);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)gemm_glds_kernel<3, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipFuncSetAttribute((const void*)gemm_glds_kernel<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_set = true;
    }
    const int tiles_m = (M + 127) / 128, tiles_n = (N + 127) / 128;
    // round 5 experiment: a FOUR-stage ring (128 KB of LDS, three k-steps of 64 in flight) — same-box A/B in the step: 3.073 / 3.080 / 3.080 ms (3 stages)
    // against 3.082 / 3.095 / 3.091 (4), mcil_gru 7.27 / 7.33 against 7.39 / 7.37: the operand stream of a tile is NOT latency-bound; stays at 3
    static const int nst = HULC_SWITCH("HULC_GLDS_NST", 3);
    if (nst == 4 && nw == 8) {
        static bool a4 = false;
        if (!a4) { hipFuncSetAttribute((const void*)gemm_glds_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); a4 = true; }
        hipLaunchKernelGGL((gemm_glds_kernel<4, 8>), dim3(tiles_m * tiles_n), dim3(512), 128 * 1024, st, a, b, om, ep, M, N, K, tiles_m, tiles_n);
        return;
    }
    if (nw == 4) hipLaunchKernelGGL((gemm_glds_kernel<3, 4>), dim3(tiles_m * tiles_n), dim3(256), 96 * 1024, st, a, b, om, ep, M, N, K, tiles_m, tiles_n);
    else hipLaunchKernelGGL((gemm_glds_kernel<3, 8>), dim3(tiles_m * tiles_n), dim3(512), 96 * 1024, st, a, b, om, ep, M, N, K, tiles_m, tiles_n);
}

static inline void launch_gemm_glds_group(hipStream_t st, GemmGroupP& g) {
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute((const void*)gemm_glds_group_kernel<3, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); attr_set = true; }
    int tot = 0;
    for (int i = 0; i < g.n; ++i) { g.tm[i] = (g.M[i] + 127) / 128; g.tn[i] = (g.N[i] + 127) / 128; g.t0[i] = tot; tot += (g.tm[i] * g.tn[i] + 7) / 8 * 8; }
    hipLaunchKernelGGL((gemm_glds_group_kernel<3, 8>), dim3(tot), dim3(512), 96 * 1024, st, g);
}
// ---------------------------------------------------------------------------------------------------------
// skinny GEMM (bf16): M <= 64 rows (the per-timestep recurrent GEMMs and every M = B MLP layer).
//   out[M][N] = epi(A[M][K] W[N][K]^T), weight-bandwidth bound: one workgroup per 16 output columns (N/16 WGs fill
//   the chip at N = 2048), its 4 waves split K and stream W / A fragments straight from L2 into MFMA operands
//   (no LDS staging: nothing is reused across waves), then a 16 KB LDS tree combines the 4 K-partials.
// Requirements: K % 128 == 0, N % 16 == 0, 16-B aligned rows.
// ---------------------------------------------------------------------------------------------------------
template <int NW, int MT, int KS = 0>   // KS > 0: K == NW*32*KS exactly -> every load of the wave is issued up front (one latency exposure)
__global__ void __launch_bounds__(NW * 64) skinny_gemm_kernel(const h16_t* __restrict__ A, long long lda, const h16_t* __restrict__ W,
                                                             long long ldw, int M, int N, int K, DenseOut om, EpiP ep) {
    __shared__ float red[NW][MT * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const int m0 = blockIdx.y * (MT * 16);          // row block of MT*16 rows
    // K % 32 == 0.  The waves split the K/32 k-steps as evenly as they divide: wave w takes steps/NW (+1 for the first steps % NW waves),
    // so K need not be a multiple of NW*32 (the gripper encoder's fc of K = 3136 = 98 steps ran on the generic 32x32-tile kernel before:
    // 25 serial k-steps per tile, 48 us for 1.6 GFLOP)
    const int ksteps = K >> 5, kbase = ksteps / NW, krem = ksteps % NW;
    const int kq = KS > 0 ? K / NW : (kbase + (wave < krem ? 1 : 0)) * 32;
    const int kb = KS > 0 ? wave * kq : (wave * kbase + min(wave, krem)) * 32;
    const int g = lane >> 4, i = lane & 15;
    const h16_t* wp = W + (long long)min(n0 + i, N - 1) * ldw + kb + g * 8;
    const h16_t* ap[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) ap[mt] = A + (long long)min(m0 + mt * 16 + i, M - 1) * lda + kb + g * 8;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (KS > 0) {
        h16x8_t b[KS], a[KS][MT];
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            b[u] = *reinterpret_cast<const h16x8_t*>(wp + u * 32);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[u][mt] = *reinterpret_cast<const h16x8_t*>(ap[mt] + u * 32);
        }
#pragma unroll
        for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA_16x16x32_H(a[u][mt], b[u], acc[mt], 0, 0, 0);
    } else {
    // batches of 4 k-steps: issue all (1+MT)*4 16-byte loads, then the MFMAs (memory-level parallelism per wave)
    int k = 0;
    for (; k + 128 <= kq; k += 128) {
        h16x8_t b[4], a[4][MT];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            b[u] = *reinterpret_cast<const h16x8_t*>(wp + k + u * 32);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[u][mt] = *reinterpret_cast<const h16x8_t*>(ap[mt] + k + u * 32);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA_16x16x32_H(a[u][mt], b[u], acc[mt], 0, 0, 0);
    }
    for (; k < kq; k += 32) {
        const h16x8_t b = *reinterpret_cast<const h16x8_t*>(wp + k);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const h16x8_t a = *reinterpret_cast<const h16x8_t*>(ap[mt] + k);
            acc[mt] = MFMA_16x16x32_H(a, b, acc[mt], 0, 0, 0);
        }
    }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(mt * 4 + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    for (int idx = tid; idx < MT * 256; idx += NW * 64) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][idx];
        const int mt = idx >> 8, r = (idx >> 6) & 3, l = idx & 63;
        const int row = m0 + mt * 16 + (l >> 4) * 4 + r;
        const int col = n0 + (l & 15);
        if (row < M && col < N) {
            const int rrow = ep.res_rowmod > 0 ? row % ep.res_rowmod : row;
            epi_store<h16_t>(ep, v, rrow, col, om.offset(row, 0) + col);
        }
    }
}
#ifdef HULC_KERNEL_STAMPS   // tools/skinny_stamps.hip: cycle stamps of the phases of one launch (never defined in the library build)
__device__ unsigned long long g_stamps[512 * 64];
#define KSTAMP(n) do { if (lane == 0) g_stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 64 + wave * 8 + (n)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define KSTAMP(n)
#endif
// Fragment-ordered weight copy (round 4).  A 16-bit W[N][K] read as MFMA fragments puts 16 different rows (K x 2 bytes apart) on 16 adjacent lanes:
// every lane's 16 bytes are a line lookup of their own in the texture path (tools/loadbench.hip: 192 KB per CU in 7.4 us fragment-shaped, 4.3 us as
// whole lines).  The recurrent GRU steps pull 192 KB of W per workgroup that way, 60 % of their bytes.  The copy stores each (16-row tile, 32-k step)
// as ONE 1 KB block in lane order — lane (g, i) at byte 16 (16 g + i) holds W[16 nb + i][32 ks + 8 g .. + 7], blocks ordered [nb][ks] — so that a
// fragment load is a contiguous 1 KB wave instruction.  Kernels take it with ldw == 0 (skinny_wfrag_ptr).  N % 16 == 0, K % 256 == 0.
constexpr int FRAG_PACK_MAX = 24;
struct FragPackBatch { const h16_t* src[FRAG_PACK_MAX]; h16_t* dst[FRAG_PACK_MAX]; int N[FRAG_PACK_MAX], K[FRAG_PACK_MAX]; int blk0[FRAG_PACK_MAX + 1]; int n; };
static inline int frag_pack_blocks(int N, int K) { return (N / 16) * (K / ((K % 256) == 0 ? 256 : 128)); }       // N % 16 == 0, K % 128 == 0
__global__ void __launch_bounds__(256) frag_pack_kernel(FragPackBatch b) {
    __shared__ uint4 tile[16][33];
    int j = 0;
    while (j + 1 < b.n && (int)blockIdx.x >= b.blk0[j + 1]) ++j;
    const int K = b.K[j], KT = (K % 256) == 0 ? 256 : 128, CPR = KT >> 3;          // a block repacks 16 rows x KT columns: CPR 16-byte chunks per row
    const int kt_n = K / KT, blk = (int)blockIdx.x - b.blk0[j], nb = blk / kt_n, kt = blk % kt_n, t = threadIdx.x;
    const h16_t* __restrict__ src = b.src[j];
    h16_t* __restrict__ dst = b.dst[j];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int idx = t + 256 * r, row = idx / CPR, ch = idx % CPR;
        if (idx < 16 * CPR) tile[row][ch] = *reinterpret_cast<const uint4*>(src + (long long)(nb * 16 + row) * K + kt * KT + ch * 8);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int idx = t + 256 * r, ksl = idx >> 6, lane = idx & 63, g = lane >> 4, i = lane & 15;
        if (idx < 16 * CPR) *reinterpret_cast<uint4*>(dst + (((long long)nb * (K >> 5) + kt * (KT >> 5) + ksl) * 64 + lane) * 8) = tile[i][ksl * 4 + g];
    }
}
// fragment-ordered W[N][K]: the block of row tile row0 / 16 and k-step k0 / 32, this lane's 16 bytes; consecutive k-steps are 512 elements apart
DEVI const h16_t* wfrag_ptr(const h16_t* Wf, int row0, int K, int k0, int lane) { return Wf + (((long long)(row0 >> 4) * (K >> 5) + (k0 >> 5)) * 64 + lane) * 8; }
// first fragment of lane (g, i) for output tile n0 at k offset kb; consecutive k-steps are `*wstep` elements apart
DEVI const h16_t* skinny_wfrag_ptr(const h16_t* W, long long ldw, int n0, int Nclamp, int K, int kb, int lane, int* wstep) {
    if (ldw == 0) { *wstep = 512; return W + ((long long)(n0 >> 4) * (K >> 5) + (kb >> 5)) * 512 + lane * 8; }
    *wstep = 32;
    return W + (long long)min(n0 + (lane & 15), Nclamp) * ldw + kb + (lane >> 4) * 8;
}
static bool skinny_use_lds = true;
static bool gemm_use_glds = true;   // tests / tools can force the register-fragment kernel
// skinny GEMM, A through LDS (K % 512 == 0, MT*16 rows x K bf16 <= 128 KB): the A rows — 2/3 of the bytes a workgroup pulls, and
// re-read by every column workgroup — arrive as full 128-byte lines by LDS-DMA (global_load_lds_dwordx4: one wave instruction =
// 8 rows x 128 B, XOR-swizzled on the SOURCE address so the later ds_read_b128 fragments are conflict-free) instead of
// fragment-shaped 16 x 64 B loads, which the texture-address path serves at ~2/3 of the full-line rate (tools/loadbench.hip:
// 192 KB/CU in 4.3 us vs 7.4 us).  Each wave DMAs exactly the k-range it multiplies, so no barrier sits between load and MFMA.
// A second, INDEPENDENT problem of the same shape can ride in the same launch (blockIdx.z == 1: A2 / W2 / ep2) — the two directions of a
// bidirectional recurrence advance one time step each per launch, and the chain pays one launch boundary instead of two.
struct Skinny2 { const h16_t* A; const h16_t* W; EpiP ep; };
template <int MT, int KQ32, int NW = 8, bool DUAL = false>   // KQ32 = k-steps (of 32) per wave = K / (NW * 32); DUAL: a separate instantiation, the single-problem code is untouched
__global__ void __launch_bounds__(NW * 64) skinny_lds_kernel(const h16_t* __restrict__ A_, long long lda, const h16_t* __restrict__ W_,
                                                         long long ldw, int M, int N, int K, DenseOut om, EpiP ep_, Skinny2 p2 = Skinny2{}) {
    const bool second = DUAL && blockIdx.z != 0;
    const h16_t* __restrict__ A = second ? p2.A : A_;
    const h16_t* __restrict__ W = second ? p2.W : W_;
    const EpiP& ep = second ? p2.ep : ep_;
    constexpr int PCW = KQ32 / 2;               // 128-byte pieces (64 k) per row per wave
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    typedef __attribute__((address_space(3))) char lchar;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    const int m0 = blockIdx.y * (MT * 16);
    const int kq = KQ32 * 32, kb = wave * kq;
    const int g = lane >> 4, i = lane & 15;
    lchar* wbase = (lchar*)sk_smem + wave * (MT * 2 * PCW * 1024);
    KSTAMP(0);
    // ---- A: MT*2 row groups (8 rows) x PCW pieces, lane = (row l>>3, LDS slot l&7), source chunk = slot ^ (row & 6)
    {
        const int r = lane >> 3, c = (lane & 7) ^ (r & 6);
#pragma unroll
        for (int rg = 0; rg < MT * 2; ++rg) {
            const h16_t* src = A + (long long)min(m0 + rg * 8 + r, M - 1) * lda + kb + c * 8;
#pragma unroll
            for (int pc = 0; pc < PCW; ++pc)
                lds_dma16(src + pc * 64, (unsigned)(size_t)(wbase + (rg * PCW + pc) * 1024));
        }
    }
    // ---- W: fragments straight to registers (one third of the bytes)
    const h16_t* wp = W + (long long)min(n0 + i, N - 1) * ldw + kb + g * 8;
    h16x8_t b[KQ32];
#pragma unroll
    for (int u = 0; u < KQ32; ++u) b[u] = *reinterpret_cast<const h16x8_t*>(wp + u * 32);
    // ---- epilogue operands of this thread's 4 outputs (threads < MT*64), fetched under the operand stream
    const int erow = m0 + (tid >> 6) * 16 + i, ecol = n0 + g * 4;
    const bool ethread = tid < MT * 64 && erow < M;
    const int errow = ep.res_rowmod > 0 ? erow % ep.res_rowmod : erow;
    const long long eo = ethread ? om.offset(erow, 0) + ecol : 0;
    // compact 16-bit epilogue (bias / residual / ReLU / ReLU mask / second store: what the MLP layers and recurrent steps use) beside the generic one
    const bool fast16 = epi_is_fast16(ep) && ecol + 3 < N && (eo & 3) == 0;
    EpiPre4 pre;
    EpiF16Pre fpre;
    if (ethread) { if (fast16) fpre = epi_fast16_pre(ep, errow, ecol, eo); else pre = epi_prefetch4<h16_t>(ep, errow, ecol, N, eo); }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    KSTAMP(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's own DMA has landed (no other wave reads it)
    KSTAMP(2);
    {
        const int r = i & 7;
        lchar* fb = wbase + (i >> 3) * (PCW * 1024) + r * 128;
#pragma unroll
        for (int u = 0; u < KQ32; ++u) {
            const int chunk = ((u & 1) * 4 + g) ^ (r & 6);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h16x8_t a = *(__attribute__((address_space(3))) h16x8_t*)(fb + (mt * 2 * PCW + (u >> 1)) * 1024 + chunk * 16);
                acc[mt] = MFMA_16x16x32_H(b[u], a, acc[mt], 0, 0, 0);   // D^T: lane (row i, g) owns columns g*4..g*4+3
            }
        }
    }
    KSTAMP(3);
    __syncthreads();                                     // every wave is done with its A region: reuse LDS for the K-partials
    KSTAMP(4);
    f32x4* red = reinterpret_cast<f32x4*>(sk_smem);      // [NW][MT][64 lanes]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[(wave * MT + mt) * 64 + lane] = acc[mt];
    __syncthreads();
    if (ethread) {
        f32x4 v = red[tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w * MT * 64 + tid];
        if (fast16) epi_fast16_apply(ep, fpre, v, eo);
        else {
        const float v4[4] = {v[0], v[1], v[2], v[3]};
        epi_apply4<h16_t>(ep, pre, v4, errow, ecol, N, eo);
        }
    }
    KSTAMP(5);
}
// ---------------------------------------------------------------------------------------------------------------------
// One time step of torch.nn.GRU (mcil variant, plan_recognition.rnn_type = nn.GRU; plan_recognition_net.py:12-42) in ONE launch:
//   g = h_prev W_hh^T + b_hh (N = 3 x H gate blocks r | z | n), then r = sig(zx_r + g_r), z = sig(zx_z + g_z), n = tanh(zx_n + r g_n),
//   h' = (1 - z) n + z h_prev;  r, z, n and g_n are kept for the backward.
// Same structure as skinny_lds_kernel (A rows by LDS-DMA, W fragments straight to registers, K split over 16 waves), but a workgroup owns
// the SAME 16 hidden units of all three gate blocks: the A rows it stages (2/3 of the bytes a workgroup pulls) feed three MFMAs instead of
// one, the grid is one workgroup per CU (128 column tiles x 2 row blocks) instead of three waves of them, and the gate arithmetic runs in
// the epilogue on values that never leave the chip (before: a 24 MB fp32 `g` round trip and a second launch per step).
// K = 2048, rows of A 128-byte aligned, H % 16 == 0.
// ---------------------------------------------------------------------------------------------------------------------
struct GruStepP { const h16_t* A; const h16_t* W; const h16_t* zx; const float* bhh; h16_t *h_out, *r_out, *z_out, *n_out, *gn_out; };
template <int MT, int KQ32, int NW>
__global__ void __launch_bounds__(NW * 64) gru_step_lds_kernel(GruStepP q0, GruStepP q1, long long lda, long long ldw, int M, int H) {     // blockIdx.z selects the problem (see Skinny2)
    const GruStepP& q = blockIdx.z ? q1 : q0;
    const h16_t* __restrict__ A = q.A; const h16_t* __restrict__ W = q.W; const h16_t* __restrict__ zx = q.zx; const float* __restrict__ bhh = q.bhh;
    h16_t* __restrict__ h_out = q.h_out; h16_t* __restrict__ r_out = q.r_out; h16_t* __restrict__ z_out = q.z_out; h16_t* __restrict__ n_out = q.n_out; h16_t* __restrict__ gn_out = q.gn_out;
    constexpr int PCW = KQ32 / 2;
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    typedef __attribute__((address_space(3))) char lchar;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;                              // hidden units n0 .. n0+15 of every gate block
    // gridDim.y == 1 (round 5): ONE workgroup walks all 32-row blocks of its 16 hidden units.  The W_hh fragments — 60 % of the bytes a workgroup pulls
    // (192 KB of 320) — are loaded ONCE and stay in registers across the blocks, and the launch is one round of 256 workgroups (x 2 directions) instead
    // of two (512 workgroups of 1024 threads / 128 KB LDS: one per CU at a time).  gridDim.y > 1: one row block per workgroup (M > 64, tests).
    const int nrb = gridDim.y == 1 ? (M + MT * 16 - 1) / (MT * 16) : 1;
    const int kq = KQ32 * 32, kb = wave * kq;
    const int g = lane >> 4, i = lane & 15;
    lchar* wbase = (lchar*)sk_smem + wave * (MT * 2 * PCW * 1024);
    auto dma_rows = [&](int m0) {
        const int r = lane >> 3, c = (lane & 7) ^ (r & 6);
#pragma unroll
        for (int rg = 0; rg < MT * 2; ++rg) {
            const h16_t* src = A + (long long)min(m0 + rg * 8 + r, M - 1) * lda + kb + c * 8;
#pragma unroll
            for (int pc = 0; pc < PCW; ++pc)
                lds_dma16(src + pc * 64, (unsigned)(size_t)(wbase + (rg * PCW + pc) * 1024));
        }
    };
    dma_rows(gridDim.y == 1 ? 0 : blockIdx.y * (MT * 16));
    h16x8_t b[3][KQ32];
#pragma unroll
    for (int gate = 0; gate < 3; ++gate) {
        int wstep;
        const h16_t* wp = skinny_wfrag_ptr(W, ldw, gate * H + n0, 3 * H - 1, H, kb, lane, &wstep);      // ldw == 0: fragment-ordered copy (frag_pack_kernel)
#pragma unroll
        for (int u = 0; u < KQ32; ++u) b[gate][u] = *reinterpret_cast<const h16x8_t*>(wp + u * wstep);
    }
    const int ecol = n0 + g * 4;
    float4 bv[3];
    if (tid < MT * 64) {
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) bv[gate] = *reinterpret_cast<const float4*>(bhh + gate * H + ecol);
    }
#pragma unroll 1
    for (int rb = 0; rb < nrb; ++rb) {
    const int m0 = (gridDim.y == 1 ? rb : blockIdx.y) * (MT * 16);
    // epilogue operands of this thread's 4 hidden units (threads < MT*64), fetched under the operand stream
    const int erow = m0 + (tid >> 6) * 16 + i;
    const bool ethread = tid < MT * 64 && erow < M;
    uint2 zxv[3] = {uint2{0u, 0u}, uint2{0u, 0u}, uint2{0u, 0u}}, hpv = uint2{0u, 0u};
    if (ethread) {
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) zxv[gate] = *reinterpret_cast<const uint2*>(zx + (long long)erow * 3 * H + gate * H + ecol);
        hpv = *reinterpret_cast<const uint2*>(A + (long long)erow * lda + ecol);
    }
    f32x4 acc[MT][3];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) acc[mt][gate] = f32x4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's own DMA has landed (no other wave reads it)
    {
        const int r = i & 7;
        lchar* fb = wbase + (i >> 3) * (PCW * 1024) + r * 128;
#pragma unroll
        for (int u = 0; u < KQ32; ++u) {
            const int chunk = ((u & 1) * 4 + g) ^ (r & 6);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const h16x8_t a = *(__attribute__((address_space(3))) h16x8_t*)(fb + (mt * 2 * PCW + (u >> 1)) * 1024 + chunk * 16);
#pragma unroll
                for (int gate = 0; gate < 3; ++gate) acc[mt][gate] = MFMA_16x16x32_H(b[gate][u], a, acc[mt][gate], 0, 0, 0);   // D^T: lane (row i, g) owns units g*4..g*4+3
            }
        }
    }
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(sk_smem);      // [NW][MT][3][64 lanes]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) red[((wave * MT + mt) * 3 + gate) * 64 + lane] = acc[mt][gate];
    __syncthreads();
    f32x4 gs[3];
    if (ethread) {
        const int mt = tid >> 6, l = tid & 63;
#pragma unroll
        for (int gate = 0; gate < 3; ++gate) {
            f32x4 v = red[((0 * MT + mt) * 3 + gate) * 64 + l];
#pragma unroll
            for (int w = 1; w < NW; ++w) v += red[((w * MT + mt) * 3 + gate) * 64 + l];
            gs[gate] = v;
        }
    }
    if (rb + 1 < nrb) {                                   // the partials are read: the next row block's A rows stream in under the gate arithmetic
        __syncthreads();
        dma_rows(m0 + MT * 16);
    }
    if (ethread) {
        const float* bb[3] = {&bv[0].x, &bv[1].x, &bv[2].x};
        float hn[4], rr[4], zz[4], nn[4], gn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned wr_ = e < 2 ? zxv[0].x : zxv[0].y, wz_ = e < 2 ? zxv[1].x : zxv[1].y, wn_ = e < 2 ? zxv[2].x : zxv[2].y, wh_ = e < 2 ? hpv.x : hpv.y;
            const float xr = (e & 1) ? h2f_hi(wr_) : h2f_lo(wr_), xz = (e & 1) ? h2f_hi(wz_) : h2f_lo(wz_), xn = (e & 1) ? h2f_hi(wn_) : h2f_lo(wn_);
            const float hp = (e & 1) ? h2f_hi(wh_) : h2f_lo(wh_);
            const float gr = gs[0][e] + bb[0][e], gz = gs[1][e] + bb[1][e];
            gn[e] = gs[2][e] + bb[2][e];
            rr[e] = 1.f / (1.f + __expf(-(xr + gr)));
            zz[e] = 1.f / (1.f + __expf(-(xz + gz)));
            nn[e] = tanhf(xn + rr[e] * gn[e]);
            hn[e] = (1.f - zz[e]) * nn[e] + zz[e] * hp;
        }
        const long long o = (long long)erow * H + ecol;
        *reinterpret_cast<uint2*>(h_out + o) = uint2{pack2h(hn[0], hn[1]), pack2h(hn[2], hn[3])};
        *reinterpret_cast<uint2*>(r_out + o) = uint2{pack2h(rr[0], rr[1]), pack2h(rr[2], rr[3])};
        *reinterpret_cast<uint2*>(z_out + o) = uint2{pack2h(zz[0], zz[1]), pack2h(zz[2], zz[3])};
        *reinterpret_cast<uint2*>(n_out + o) = uint2{pack2h(nn[0], nn[1]), pack2h(nn[2], nn[3])};
        *reinterpret_cast<uint2*>(gn_out + o) = uint2{pack2h(gn[0], gn[1]), pack2h(gn[2], gn[3])};
    }
    }
}
// returns false when the shape is not covered (the caller then runs the GEMM + gate kernel pair)
// nprob = 2: q[1] is a second independent recurrence (the other direction of the BiGRU) advanced by the same launch
static inline bool launch_gru_step(hipStream_t st, const GruStepP* q, int nprob, int M, int H, bool wfrag = false) {
    for (int k = 0; k < nprob; ++k)
        if (H != 2048 || M < 1 || ((uintptr_t)q[k].A % 128) != 0 || ((uintptr_t)q[k].W % 16) != 0) return false;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)gru_step_lds_kernel<2, 4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    static const int walk = HULC_SWITCH("HULC_GRU_WALK", 0);      // 1: one workgroup walks the row blocks with W_hh loaded once — measured SLOWER in the step (mcil_gru 7.30 -> 7.47, 7.37 -> 7.63 ms: the row blocks of a workgroup serialise, two rounds of workgroups overlap their loads), kept for the record
    dim3 grid(H / 16, (walk && M <= 128) ? 1 : (M + 31) / 32, nprob);
    // LDS: the A region (16 waves x 2 x 2 x 2 KB = 128 KB) is reused for the 16 x 2 x 3 K-partials (96 KB)
    hipLaunchKernelGGL((gru_step_lds_kernel<2, 4, 16>), grid, dim3(1024), (size_t)16 * 2 * 2 * 2 * 1024, st, q[0], q[nprob - 1], (long long)H, wfrag ? 0ll : (long long)H, M, H);
    return true;
}

// skinny_lds_kernel for K = NCH x 2048 (the GRU's BPTT step: dh = dG W_hh with K = 3 x 2048): the same 16-wave structure walks the K chunks
// one after the other through the SAME LDS regions — a wave only ever reads the region it DMAs into, so there is no barrier between chunks,
// just the wave's own lgkmcnt(0) (fragment reads done) before the next chunk's DMA and vmcnt(0) before its MFMAs.  The W fragments of the
// next chunk are requested before the current chunk's MFMAs.  Replaces the register-fragment kernel for these shapes (18 -> 14 us per step).
// Optional GRU epilogue (BPTT of torch.nn.GRU, plan_recognition_net.py:12-42): the GEMM computes the carry dG[t] W_hh into step t-1; with `gb.dzx` set
// the gate backward of step t-1 runs right here on the fp32 carry — dh = dH[t-1] + carry + direct[t] -> dzx = (dr, dz, dn), dg = (dr, dz, dn r),
// direct[t-1] = dh z — instead of in a launch of its own between every two GEMMs of the chain (97 launches of 4.2 us per step at S = 32).
struct GruBwdP {
    const h16_t *dH, *R, *Z, *Nn, *GN, *Hprev;      // step t-1: incoming state gradient (or null), saved gates, previous state (or null = 0)
    h16_t *dzx, *dg, *direct;                        // outputs for step t-1: [M][3H], [M][3H], [M][H]; dzx == nullptr: plain GEMM epilogue
    const h16_t* direct_in;                          // direct[t] (dh_t z_t), added to the carry
};
struct KChunk2 { const h16_t* A; const h16_t* W; EpiP ep; GruBwdP gb; };
template <int MT, int KQ32, int NW>
__global__ void __launch_bounds__(NW * 64) skinny_lds_kchunk_kernel(const h16_t* __restrict__ A_, long long lda, const h16_t* __restrict__ W_, long long ldw, int M, int N,
                                                                  int K, DenseOut om, EpiP ep_, GruBwdP gb_, KChunk2 p2 = KChunk2{}) {      // blockIdx.z == 1: the second problem (see Skinny2)
    const h16_t* __restrict__ A = blockIdx.z ? p2.A : A_;
    const h16_t* __restrict__ W = blockIdx.z ? p2.W : W_;
    const EpiP& ep = blockIdx.z ? p2.ep : ep_;
    const GruBwdP& gb = blockIdx.z ? p2.gb : gb_;
    constexpr int PCW = KQ32 / 2, KCH = NW * KQ32 * 32;
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    typedef __attribute__((address_space(3))) char lchar;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (MT * 16);
    const int kq = KQ32 * 32, kb = wave * kq;
    const int g = lane >> 4, i = lane & 15;
    const int nch = K / KCH;
    lchar* wbase = (lchar*)sk_smem + wave * (MT * 2 * PCW * 1024);
    auto dma_chunk = [&](int ch) {
        const int r = lane >> 3, c = (lane & 7) ^ (r & 6);
#pragma unroll
        for (int rg = 0; rg < MT * 2; ++rg) {
            const h16_t* src = A + (long long)min(m0 + rg * 8 + r, M - 1) * lda + (long long)ch * KCH + kb + c * 8;
#pragma unroll
            for (int pc = 0; pc < PCW; ++pc)
                lds_dma16(src + pc * 64, (unsigned)(size_t)(wbase + (rg * PCW + pc) * 1024));
        }
    };
    int wstep;
    const h16_t* wp = skinny_wfrag_ptr(W, ldw, n0, N - 1, K, kb, lane, &wstep);      // ldw == 0: fragment-ordered copy (frag_pack_kernel)
    const long long wchunk = (long long)(KCH / 32) * wstep;                                // elements between two K chunks
    dma_chunk(0);
    h16x8_t b[KQ32];
#pragma unroll
    for (int u = 0; u < KQ32; ++u) b[u] = *reinterpret_cast<const h16x8_t*>(wp + u * wstep);
    const int erow = m0 + (tid >> 6) * 16 + i, ecol = n0 + g * 4;
    const bool ethread = tid < MT * 64 && erow < M;
    const int errow = ep.res_rowmod > 0 ? erow % ep.res_rowmod : erow;
    const long long eo = ethread ? om.offset(erow, 0) + ecol : 0;
    EpiPre4 pre;
    uint2 gq[7];                                         // GRU epilogue operands of this thread's 4 outputs: dH, R, Z, N, GN, Hprev, direct_in (4 x 16 bit each)
    const long long gidx = (long long)erow * N + ecol;
    if (ethread) {
        if (gb.dzx) {
            const h16_t* src[7] = {gb.dH, gb.R, gb.Z, gb.Nn, gb.GN, gb.Hprev, gb.direct_in};
#pragma unroll
            for (int q = 0; q < 7; ++q) gq[q] = src[q] ? *reinterpret_cast<const uint2*>(src[q] + gidx) : uint2{0u, 0u};
        } else pre = epi_prefetch4<h16_t>(ep, errow, ecol, N, eo);
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int r7 = i & 7;
    lchar* fb = wbase + (i >> 3) * (PCW * 1024) + r7 * 128;
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this chunk's DMA and W fragments have landed
        h16x8_t a[KQ32][MT];
#pragma unroll
        for (int u = 0; u < KQ32; ++u) {
            const int chunk = ((u & 1) * 4 + g) ^ (r7 & 6);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a[u][mt] = *(__attribute__((address_space(3))) h16x8_t*)(fb + (mt * 2 * PCW + (u >> 1)) * 1024 + chunk * 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // fragments are in registers: the region may be overwritten
        h16x8_t bc[KQ32];
#pragma unroll
        for (int u = 0; u < KQ32; ++u) bc[u] = b[u];
        if (ch + 1 < nch) {                                         // next chunk in flight under this chunk's MFMAs
            dma_chunk(ch + 1);
#pragma unroll
            for (int u = 0; u < KQ32; ++u) b[u] = *reinterpret_cast<const h16x8_t*>(wp + (ch + 1) * wchunk + u * wstep);
        }
#pragma unroll
        for (int u = 0; u < KQ32; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA_16x16x32_H(bc[u], a[u][mt], acc[mt], 0, 0, 0);
    }
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(sk_smem);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) red[(wave * MT + mt) * 64 + lane] = acc[mt];
    __syncthreads();
    if (ethread) {
        f32x4 v = red[tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w * MT * 64 + tid];
        const float v4[4] = {v[0], v[1], v[2], v[3]};
        if (gb.dzx) {
            auto un = [](const uint2& u, int r) { const unsigned w = r < 2 ? u.x : u.y; return (r & 1) ? h2f_hi(w) : h2f_lo(w); };
            float dr[4], dz[4], dn[4], dnr[4], dir[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dh = un(gq[0], r) + v4[r] + un(gq[6], r);
                const float rr = un(gq[1], r), z = un(gq[2], r), n = un(gq[3], r), gn = un(gq[4], r), hp = un(gq[5], r);
                dn[r] = dh * (1.f - z) * (1.f - n * n);
                dz[r] = dh * (hp - n) * z * (1.f - z);
                dr[r] = dn[r] * gn * rr * (1.f - rr);
                dnr[r] = dn[r] * rr;
                dir[r] = dh * z;
            }
            auto st4 = [](h16_t* p, const float (&x)[4]) { uint2 w; w.x = pack2h(x[0], x[1]); w.y = pack2h(x[2], x[3]); *reinterpret_cast<uint2*>(p) = w; };
            const long long o3 = (long long)erow * 3 * N + ecol;
            st4(gb.dzx + o3, dr); st4(gb.dzx + o3 + N, dz); st4(gb.dzx + o3 + 2 * N, dn);
            st4(gb.dg + o3, dr); st4(gb.dg + o3 + N, dz); st4(gb.dg + o3 + 2 * N, dnr);
            st4(gb.direct + gidx, dir);
        } else epi_apply4<h16_t>(ep, pre, v4, errow, ecol, N, eo);
    }
}
static inline bool launch_skinny_lds_kchunk(hipStream_t st, const h16_t* A, long long lda, const h16_t* W, long long ldw, int M, int N, int K, const DenseOut& om,
                                            const EpiP& ep, const GruBwdP& gb = GruBwdP{}, const KChunk2* p2 = nullptr) {
    if (K % 2048 != 0 || K <= 2048 || M > 64 || (lda % 64) != 0 || ((uintptr_t)A % 128) != 0 || (N % 16) != 0) return false;
    if (p2 && ((uintptr_t)p2->A % 128) != 0) return false;
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)skinny_lds_kchunk_kernel<2, 4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL((skinny_lds_kchunk_kernel<2, 4, 16>), dim3(N / 16, (M + 31) / 32, p2 ? 2 : 1), dim3(1024), (size_t)16 * 2 * 2 * 2 * 1024, st, A, lda, W, ldw, M, N, K, om, ep, gb,
                       p2 ? *p2 : KChunk2{});
    return true;
}

template <int MT, int KQ32>
static inline void launch_skinny_lds_t(hipStream_t st, dim3 grid, const h16_t* A, long long lda, const h16_t* W, long long ldw, int M, int N, int K,
                                       const DenseOut& om, const EpiP& ep) {
    const size_t lds = (size_t)8 * MT * 2 * (KQ32 / 2) * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)skinny_lds_kernel<MT, KQ32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((skinny_lds_kernel<MT, KQ32>), grid, dim3(512), lds, st, A, lda, W, ldw, M, N, K, om, ep);
}
// returns false when the shape is not covered (caller uses the register-fragment kernel)
static inline bool launch_skinny_lds(hipStream_t st, const h16_t* A, long long lda, const h16_t* W, long long ldw, int M, int N, int K, int MT,
                                     const DenseOut& om, const EpiP& ep) {
    if (K % 512 != 0 || K > 2048 || (lda % 64) != 0 || ((uintptr_t)A % 128) != 0) return false;
    if ((size_t)MT * 16 * K * 2 > 128 * 1024 || (MT != 2 && MT != 4 && !(MT == 1 && K == 2048))) return false;
    dim3 grid(N / 16, (M + MT * 16 - 1) / (MT * 16));
    const int kq32 = K / 256;
#define SKL(mt, kq) launch_skinny_lds_t<mt, kq>(st, grid, A, lda, W, ldw, M, N, K, om, ep)
    static const int nw16 = HULC_SWITCH("HULC_SKINNY_NW16", 1);   // A/B: -0.6 % of the step
    if (MT == 1) {                                  // M <= 32 recurrent step (32 + 32 windows per GPU): 16-row blocks so that 2 x 128 workgroups fill the chip
        static bool attr1 = false;
        if (!attr1) { hipFuncSetAttribute((const void*)skinny_lds_kernel<1, 4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
        hipLaunchKernelGGL((skinny_lds_kernel<1, 4, 16>), grid, dim3(1024), (size_t)16 * 1 * 2 * 2 * 1024, st, A, lda, W, ldw, M, N, K, om, ep);
        return true;
    }
    if (nw16 && MT == 2 && kq32 == 8) {            // K = 2048 (the recurrent step): 16 waves x 4 k-steps, 4 waves per SIMD overlap DMA issue and MFMAs
        static bool attr16 = false;
        if (!attr16) { hipFuncSetAttribute((const void*)skinny_lds_kernel<2, 4, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr16 = true; }
        hipLaunchKernelGGL((skinny_lds_kernel<2, 4, 16>), grid, dim3(1024), (size_t)16 * 2 * 2 * 2 * 1024, st, A, lda, W, ldw, M, N, K, om, ep);
        return true;
    }
    if (MT == 2) { if (kq32 == 8) SKL(2, 8); else if (kq32 == 4) SKL(2, 4); else if (kq32 == 2) SKL(2, 2); else return false; }
    else { if (kq32 == 4) SKL(4, 4); else if (kq32 == 2) SKL(4, 2); else return false; }
#undef SKL
    return true;
}
// two independent [M <= 64] x [N] x [K = 2048] recurrent steps in ONE launch of skinny_lds_kernel<2, 4, 16> (grid.z = 2); false: shape not covered
static inline bool launch_skinny_lds_dual(hipStream_t st, const h16_t* A0, const h16_t* W0, const EpiP& ep0, const h16_t* A1, const h16_t* W1, const EpiP& ep1, long long lda,
                                          long long ldw, int M, int N, int K, const DenseOut& om) {
    if (!skinny_use_lds || K != 2048 || M <= 32 || M > 64 || (N % 16) != 0 || (lda % 64) != 0 || ((uintptr_t)A0 % 128) != 0 || ((uintptr_t)A1 % 128) != 0 ||
        ((uintptr_t)W0 % 16) != 0 || ((uintptr_t)W1 % 16) != 0 || (ldw % 8) != 0) return false;
    static bool attr16 = false;
    if (!attr16) { hipFuncSetAttribute((const void*)skinny_lds_kernel<2, 4, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr16 = true; }
    Skinny2 p2; p2.A = A1; p2.W = W1; p2.ep = ep1;
    hipLaunchKernelGGL((skinny_lds_kernel<2, 4, 16, true>), dim3(N / 16, (M + 31) / 32, 2), dim3(1024), (size_t)16 * 2 * 2 * 2 * 1024, st, A0, lda, W0, ldw, M, N, K, om, ep0, p2);
    return true;
}
template <int NW>
static inline void launch_skinny_nw(hipStream_t st, const h16_t* A, long long lda, const h16_t* W, long long ldw, int M, int N, int K,
                                    const DenseOut& om, const EpiP& ep) {
    // The kernel is bound by the per-CU load path (~10 B/clk/CU): every workgroup streams its A rows (M x K) and a 16-row W slice.
    // With only N/16 workgroups (128 at N = 2048) half the CUs idle, so split the rows in two 32-row blocks when that fills the chip.
    int MT = M >= 64 ? 4 : (M + 15) / 16;
    if (M > 32 && (N / 16) * ((M + 63) / 64) <= 160) MT = 2;
    static const bool mt2k = HULC_SWITCH("HULC_SKINNY_MT2K", 1) != 0;
    // K = 2048 (GRU recurrent step with N = 3 x 2048; the many-row weight-gradient / small-N GEMMs over 2048 tokens): 32-row blocks keep the
    // LDS-DMA kernel eligible (64 rows x 2048 would not fit LDS) and double the workgroup count of the small-M cases
    if (mt2k && M > 32 && K == 2048 && NW == 8) MT = 2;
    static const bool mt1 = HULC_SWITCH("HULC_SKINNY_MT1", 1) != 0;
    if (mt1 && M > 16 && M <= 32 && K == 2048 && NW == 8 && (N / 16) * 2 <= 320) MT = 1;     // 16-row blocks: twice the workgroups, 128 KB instead of 192 KB each
    if (NW == 8 && skinny_use_lds && launch_skinny_lds(st, A, lda, W, ldw, M, N, K, MT, om, ep)) return;
    dim3 grid(N / 16, (M + MT * 16 - 1) / (MT * 16)), block(NW * 64);
    switch (MT) {
        case 1: hipLaunchKernelGGL((skinny_gemm_kernel<NW, 1>), grid, block, 0, st, A, lda, W, ldw, M, N, K, om, ep); break;
        case 2: hipLaunchKernelGGL((skinny_gemm_kernel<NW, 2>), grid, block, 0, st, A, lda, W, ldw, M, N, K, om, ep); break;
        case 3: hipLaunchKernelGGL((skinny_gemm_kernel<NW, 3>), grid, block, 0, st, A, lda, W, ldw, M, N, K, om, ep); break;
        default: hipLaunchKernelGGL((skinny_gemm_kernel<NW, 4>), grid, block, 0, st, A, lda, W, ldw, M, N, K, om, ep); break;
    }
}
static inline void launch_skinny(hipStream_t st, const h16_t* A, long long lda, const h16_t* W, long long ldw, int M, int N, int K,
                                 const DenseOut& om, const EpiP& ep) {
    static const bool kchunk = HULC_SWITCH("HULC_SKINNY_KCHUNK", 1) != 0;
    if (kchunk && skinny_use_lds && M > 16 && launch_skinny_lds_kchunk(st, A, lda, W, ldw, M, N, K, om, ep)) return;     // K = n x 2048, M <= 64 (GRU BPTT step)
    if (K % 512 == 0) launch_skinny_nw<8>(st, A, lda, W, ldw, M, N, K, om, ep);     // 8 waves x >=2 k-steps
    else launch_skinny_nw<4>(st, A, lda, W, ldw, M, N, K, om, ep);
}
static inline bool skinny_ok(int M, int N, int K, long long lda, long long ldw, const void* A, const void* W) {
    // many-row use: every 64-row block of A is re-read by each of the N/16 column workgroups -> only when M*N is small
    // (short K: the skinny kernel is shorter per launch at M = 2048, N = 128, K = 128..512 (5 vs 12 us in rocprof) but the step got
    //  0.03 ms SLOWER in an A/B on one box — kept off)
    static const bool shortk = HULC_SWITCH("HULC_SKINNY_SHORTK", 0) != 0;
    const bool shape = M <= 64 || ((shortk || K >= 512) && (long long)M * N <= 524288 && (long long)((M + 63) / 64) * (N / 16) >= 16);
    return shape && (K % 32) == 0 && K >= 128 && (N % 16) == 0 && (lda % 8) == 0 && (ldw % 8) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0;
}

}  // namespace HULC_NS
