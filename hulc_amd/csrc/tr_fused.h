// hulc_amd/csrc/tr_fused.h — one plan-recognition transformer encoder layer, FORWARD, as ONE launch (16-bit engines; S <= 32, and 32 < S <= 64 as two 32-row halves per window: WIDE below).
//
// Reference: nn.TransformerEncoderLayer(d_model 128, 8 heads, dim_feedforward 2048, dropout p, post-LN, relu) as built by
// hulc/models/plan_encoders/plan_recognition_net.py:78-92 and run at :112 —
//   x1 = LN1(x + drop(out_proj(attention(x W_qkv^T + b))))
//   x2 = LN2(x1 + drop(W2 drop(relu(x1 W1^T + b1)) + b2))
//
// The unfused path is 7 launches per layer (QKV GEMM, attention, out_proj GEMM, LN, FFN1 GEMM, FFN2 GEMM, LN): 2048 tokens x 128 features are
// far too little work per launch — every one of them is launch / latency bound (4-17 us each, ~65 us per layer) with the CUs mostly idle, and
// co-scheduling independent work next to them does not pay on this part.  Here a workgroup owns ONE WINDOW (S <= 32
// tokens = one 32-row MFMA tile pair) and one quarter of the FFN's hidden units:
//
//   * phase A (computed by all four workgroups of a window; 128 KB of weights, a few hundred MFMAs): [LN of the previous layer's output] ->
//     QKV GEMM -> attention of the 8 heads (one wave per head, two lanes per query row like attention_fwd32_kernel) -> out_proj + dropout +
//     residual -> LN1.  Everything stays in LDS; the tensors the BACKWARD needs (qkv, attention probabilities, attention output, y1 = the
//     LN1 input, LN1 statistics, x1 in fp32 and 16 bit — exactly what the unfused kernels save) are written by hidden-quarter 0 only;
//   * phase B (this workgroup's 512 hidden units): h = drop(relu(x1 W1[q]^T + b1[q])) -> saved (16 bit) and kept in LDS -> partial
//     y2 = h W2[:, q]^T, dropped with the output mask (the mask is linear in the partial sums), + b2 and the residual x1 from quarter 0,
//     added to y2 with fp32 atomics (y2 is zeroed by the position-add kernel).
//   LN2 needs all four partial sums: it is the FIRST step of the next layer's launch (ln_in), or the ordinary LayerNorm kernel after the last layer.
//
// A workgroup pulls 96 + 32 + 128 + 128 KB of weights through its CU's load path (the L2s serve each slice to 64 workgroups); the weight
// fragments of every phase are requested BEFORE the phase in front of it starts (they depend on nothing computed here), so the only exposed
// L2 latency is the first one.  MFMA orientation as in gemm.h: A = weight rows (16 output features), B = 16 tokens from LDS, so a lane owns 4
// consecutive output features of one token.  Dropout masks use the same (seed, element index) hashes as the unfused kernels — the backward
// (engine.h) re-derives them and is unchanged.
#pragma once
#include "common.h"
#include "conv_wgrad.h"   // lds_char

namespace HULC_NS {

struct TrLayerP {
    const float* xin;                 // [N][128] fp32: the layer input x (ln_in = 0), or the previous layer's pre-norm2 sum y2 (ln_in = 1)
    int ln_in;
    const float *ln_g, *ln_b;         // ln_in: norm2 of the previous layer
    float* xf_out; h16_t* xt_out; float* st_out;      // ln_in: the normalised input is saved (fp32, 16 bit, mean / rstd) — quarter 0 writes
    const h16_t *Wqkv, *Wo, *W1, *W2; // [384][128], [128][128], [2048][128], [128][2048], FRAGMENT-ORDERED copies (gemm.h: frag_pack_kernel / wfrag_ptr)
    const float *bqkv, *bo, *b1, *b2, *n1g, *n1b;
    h16_t* qkv; float* Pat; h16_t* ao; float* y1; float* st1; h16_t* x1t; float* x1f; h16_t* hff; float* y2;
    int B, S;
    float dp;
    unsigned long long seed_att, seed_o, seed_h, seed_y;
    long long* stamps;                // TRF_STAMPS builds (tools/tr_fused_bench.hip): shader-clock stamps of workgroup 5's phases
};
#ifdef TRF_STAMPS
#define TRF_STAMP(i) do { if (p.stamps && blockIdx.x == 5 && tid == 0) p.stamps[i] = clock64(); } while (0)
#else
#define TRF_STAMP(i) do { } while (0)
#endif

constexpr int TRF_D = 128, TRF_FF = 2048, TRF_NH = 8, TRF_HD = 16, TRF_NQ = 4, TRF_HQ = TRF_FF / TRF_NQ;       // hidden quarter = 512 units
constexpr int TRF_XP = TRF_D * 2 + 32;          // LDS row pitch (bytes) of a [32][128] 16-bit operand: 18 slots = 2 (mod 4) -> conflict-free ds_read_b128
constexpr int TRF_HP = TRF_HQ * 2 + 32;         // ... of the [32][512] hidden tile: 66 slots
constexpr int TRF_QP = 3 * TRF_D * 2 + 16;      // qkv rows [32][384] 16 bit (read element-wise by the attention)
constexpr size_t TRF_LDS = 32 * TRF_D * 4 * 2 + 32 * TRF_XP * 3 + 32 * TRF_QP + 32 * TRF_HP;
constexpr size_t TRF_LDS_WIDE = TRF_LDS + 32 * TRF_QP;      // q k v rows of the window's OTHER 32-row half

// WIDE (32 < S <= 64, BASELINE config 5): a window is two 32-row HALVES and a workgroup owns (window, half, hidden quarter).  Every phase but the attention is
// row-wise and runs on the workgroup's own half exactly as for S <= 32; the attention needs k and v of all S rows, so phase 0 / 1 also normalise and project the
// other half's rows (16-bit copy in the attention-output tile, which is free until phase 2 ends) into a [64][384] q k v tile, and a lane of phase 2 walks 32 keys
// instead of 16.  The redundant work (one more QKV projection per workgroup: 24 MFMAs per wave) is small next to what a second launch chain costs.
template <bool WIDE>
__global__ void __launch_bounds__(512) tr_layer_fwd_kernel(TrLayerP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef __attribute__((address_space(3))) h16_t lds_h16;
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    lds_char* const xf = (lds_char*)smem;                      // [32][128] fp32: x, then y1 in place
    lds_char* const x1f = xf + 32 * TRF_D * 4;                 // [32][128] fp32: x1
    lds_char* const xb = x1f + 32 * TRF_D * 4;                 // [32][XP] 16 bit: x
    lds_char* const aob = xb + 32 * TRF_XP;                    // attention output
    lds_char* const x1b = aob + 32 * TRF_XP;                   // x1
    lds_char* const qb = x1b + 32 * TRF_XP;                    // qkv
    lds_char* const hb = qb + (WIDE ? 64 : 32) * TRF_QP;       // hidden quarter
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int hq = blockIdx.x % TRF_NQ;
    const int w = WIDE ? blockIdx.x / (2 * TRF_NQ) : blockIdx.x / TRF_NQ, half = WIDE ? (blockIdx.x / TRF_NQ) & 1 : 0;
    const int SW = p.S;                                        // rows of the window (keys of the attention)
    const int r0 = half * 32;                                  // this workgroup's first row in its window
    const int S = WIDE ? min(32, SW - r0) : SW;                // rows this workgroup owns
    const long long row0 = (long long)w * SW + r0;
    const bool lead = hq == 0;
    const bool mt1 = S > 16;                                   // second 16-token tile in use
    const float keep = 1.f / (1.f - p.dp);

    TRF_STAMP(0);
    // ---- weight fragments of the QKV GEMM: requested first (rows n = (3 wave + nt) * 16 + li, 8 k at g * 8 + ks * 32)
    h16x8_t wq[3][4];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) wq[nt][ks] = *reinterpret_cast<const h16x8_t*>(wfrag_ptr(p.Wqkv, (3 * wave + nt) * 16, TRF_D, 0, lane) + ks * 512);

    // every small parameter the later phases need (biases, LayerNorm affine): requested NOW — a dependent global load in front of each
    // phase's epilogue cost an exposed L2 round trip (~1 us) per phase (tools/tr_fused_bench.hip stamps: 3 - 8 us per phase for a few
    // hundred MFMA cycles of work)
    f32x4 pbq[3], pb1[4];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) pbq[nt] = *reinterpret_cast<const f32x4*>(p.bqkv + (3 * wave + nt) * 16 + g * 4);
    const f32x4 pbo = *reinterpret_cast<const f32x4*>(p.bo + wave * 16 + g * 4);
    f32x4 pb2 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (lead) pb2 = *reinterpret_cast<const f32x4*>(p.b2 + wave * 16 + g * 4);
    const float pn1g0 = p.n1g[lane], pn1g1 = p.n1g[lane + 64], pn1b0 = p.n1b[lane], pn1b1 = p.n1b[lane + 64];
    float plg0 = 0.f, plg1 = 0.f, plb0 = 0.f, plb1 = 0.f;
    if (p.ln_in) { plg0 = p.ln_g[lane]; plg1 = p.ln_g[lane + 64]; plb0 = p.ln_b[lane]; plb1 = p.ln_b[lane + 64]; }

    // ---- phase 0: x (optionally LN of the previous layer's sum) -> xf (fp32), xb (16 bit); rows >= S are zero
    {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = wave * 4 + rr;
            float v0 = 0.f, v1 = 0.f;
            if (r < S) {
                const float* xr = p.xin + (row0 + r) * TRF_D;
                v0 = xr[lane]; v1 = xr[lane + 64];
                if (p.ln_in) {
                    const float mean = wave_sum(v0 + v1) * (1.f / TRF_D);
                    const float d0 = v0 - mean, d1 = v1 - mean;
                    const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.f / TRF_D) + 1e-5f);
                    v0 = d0 * rstd * plg0 + plb0;
                    v1 = d1 * rstd * plg1 + plb1;
                    if (lead) {
                        p.xf_out[(row0 + r) * TRF_D + lane] = v0; p.xf_out[(row0 + r) * TRF_D + lane + 64] = v1;
                        p.xt_out[(row0 + r) * TRF_D + lane] = f2h(v0); p.xt_out[(row0 + r) * TRF_D + lane + 64] = f2h(v1);
                        if (lane == 0) { p.st_out[2 * (row0 + r)] = mean; p.st_out[2 * (row0 + r) + 1] = rstd; }
                    }
                }
            }
            *(lds_f32*)(xf + (r * TRF_D + lane) * 4) = v0; *(lds_f32*)(xf + (r * TRF_D + lane + 64) * 4) = v1;
            *(lds_h16*)(xb + r * TRF_XP + lane * 2) = f2h(v0); *(lds_h16*)(xb + r * TRF_XP + (lane + 64) * 2) = f2h(v1);
        }
        if (WIDE) {       // the other half's rows, 16 bit only (operand of its k / v projection): same arithmetic as the workgroup that owns them
            const int o0 = 32 - r0, So = min(32, SW - o0);
            const long long orow0 = (long long)w * SW + o0;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int r = wave * 4 + rr;
                float v0 = 0.f, v1 = 0.f;
                if (r < So) {
                    const float* xr = p.xin + (orow0 + r) * TRF_D;
                    v0 = xr[lane]; v1 = xr[lane + 64];
                    if (p.ln_in) {
                        const float mean = wave_sum(v0 + v1) * (1.f / TRF_D);
                        const float d0 = v0 - mean, d1 = v1 - mean;
                        const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.f / TRF_D) + 1e-5f);
                        v0 = d0 * rstd * plg0 + plb0;
                        v1 = d1 * rstd * plg1 + plb1;
                    }
                }
                *(lds_h16*)(aob + r * TRF_XP + lane * 2) = f2h(v0); *(lds_h16*)(aob + r * TRF_XP + (lane + 64) * 2) = f2h(v1);
            }
        }
    }
    __syncthreads();
    // weight fragments of the later phases: they depend on nothing computed here, so they travel while QKV / attention / LN1 run
    h16x8_t wo[4], w1[4][4], w2[16];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wo[ks] = *reinterpret_cast<const h16x8_t*>(wfrag_ptr(p.Wo, wave * 16, TRF_D, 0, lane) + ks * 512);

    TRF_STAMP(1);
    // ---- phase 1: qkv = x Wqkv^T + b  -> qb (LDS) [+ global]
    {
        f32x4 acc[3][2];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) { acc[nt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[nt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(xb + li * TRF_XP + ks * 64 + g * 16);
            const h16x8_t b1 = *(__attribute__((address_space(3))) h16x8_t*)(xb + (16 + li) * TRF_XP + ks * 64 + g * 16);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                acc[nt][0] = MFMA_16x16x32_H(wq[nt][ks], b0, acc[nt][0], 0, 0, 0);
                if (mt1) acc[nt][1] = MFMA_16x16x32_H(wq[nt][ks], b1, acc[nt][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const int n = (3 * wave + nt) * 16 + g * 4;
            const f32x4 bb = pbq[nt];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m = mt * 16 + li;
                u32x2_t o;
                o[0] = pack2h(acc[nt][mt][0] + bb[0], acc[nt][mt][1] + bb[1]);
                o[1] = pack2h(acc[nt][mt][2] + bb[2], acc[nt][mt][3] + bb[3]);
                *(__attribute__((address_space(3))) u32x2_t*)(qb + (r0 + m) * TRF_QP + n * 2) = o;
                if (lead && m < S) *reinterpret_cast<u32x2_t*>(p.qkv + (row0 + m) * (3 * TRF_D) + n) = o;
            }
        }
        if (WIDE) {       // the other half's rows (their q columns are computed too and never read)
            const int o0 = 32 - r0;
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) { acc[nt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[nt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(aob + li * TRF_XP + ks * 64 + g * 16);
                const h16x8_t b1 = *(__attribute__((address_space(3))) h16x8_t*)(aob + (16 + li) * TRF_XP + ks * 64 + g * 16);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    acc[nt][0] = MFMA_16x16x32_H(wq[nt][ks], b0, acc[nt][0], 0, 0, 0);
                    acc[nt][1] = MFMA_16x16x32_H(wq[nt][ks], b1, acc[nt][1], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int n = (3 * wave + nt) * 16 + g * 4;
                const f32x4 bb = pbq[nt];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    u32x2_t o;
                    o[0] = pack2h(acc[nt][mt][0] + bb[0], acc[nt][mt][1] + bb[1]);
                    o[1] = pack2h(acc[nt][mt][2] + bb[2], acc[nt][mt][3] + bb[3]);
                    *(__attribute__((address_space(3))) u32x2_t*)(qb + (o0 + mt * 16 + li) * TRF_QP + n * 2) = o;
                }
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) pb1[nt] = *reinterpret_cast<const f32x4*>(p.b1 + hq * TRF_HQ + wave * 64 + nt * 16 + g * 4);
    // FFN weight fragments (this workgroup's hidden quarter): W1 rows hq*512 + wave*64 + nt*16 + li; W2 rows wave*16 + li, k in the quarter
    // (WIDE: requested after the attention — with 32 scores per lane next to them the wave spilled 26 registers)
    auto load_w1 = [&]() {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                w1[nt][ks] = *reinterpret_cast<const h16x8_t*>(wfrag_ptr(p.W1, hq * TRF_HQ + wave * 64 + nt * 16, TRF_D, 0, lane) + ks * 512);
    };
    if (!WIDE) load_w1();
    __syncthreads();

    TRF_STAMP(2);
    // ---- phase 2: attention, one wave per head; lane (i = query row, hf = key half) — attention_fwd32_kernel on the LDS copy of qkv.
    // A head's 16 values of a row are 32 contiguous bytes: q / k / v rows are read as two 16-byte LDS loads (the first version read them with
    // 2-byte loads — ~530 LDS instructions per lane; this phase was most of the launch's 35 us attention half)
    {
        const int h = wave, i = lane & 31, hf = lane >> 5;
        auto row16 = [&](const lds_char* r, float (&o)[TRF_HD]) {
            const u32x4_t a = *(const lds_u32x4*)r, b = *(const lds_u32x4*)(r + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[2 * e] = h2f_lo(a[e]); o[2 * e + 1] = h2f_hi(a[e]); o[8 + 2 * e] = h2f_lo(b[e]); o[9 + 2 * e] = h2f_hi(b[e]); }
        };
        constexpr int HJ = WIDE ? 32 : 16;                          // keys per lane
        if (i < S) {
            float q[TRF_HD];
            row16(qb + (r0 + i) * TRF_QP + h * TRF_HD * 2, q);
#pragma unroll
            for (int d = 0; d < TRF_HD; ++d) q[d] *= 0.25f;
            float sc[HJ];
            float m = -INFINITY;
#pragma unroll
            for (int jj = 0; jj < HJ; ++jj) {
                const int j = hf * HJ + jj;
                float kk[TRF_HD];
                row16(qb + (j < SW ? j : 0) * TRF_QP + (TRF_D + h * TRF_HD) * 2, kk);
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < TRF_HD; ++d) s += q[d] * kk[d];
                sc[jj] = j < SW ? s : -INFINITY;
                m = fmaxf(m, sc[jj]);
            }
            m = fmaxf(m, __shfl_xor(m, 32));
            float den = 0.f;
#pragma unroll
            for (int jj = 0; jj < HJ; ++jj) { sc[jj] = (hf * HJ + jj) < SW ? __expf(sc[jj] - m) : 0.f; den += sc[jj]; }
            den += __shfl_xor(den, 32);
            const float inv = 1.f / den;
            float o[TRF_HD];
#pragma unroll
            for (int d = 0; d < TRF_HD; ++d) o[d] = 0.f;
            const long long pbase = (((long long)w * TRF_NH + h) * SW + r0 + i) * SW;
#pragma unroll
            for (int jj = 0; jj < HJ; ++jj) {
                const int j = hf * HJ + jj;
                if (j < SW) {
                    float pr = sc[jj] * inv;
                    if (lead) p.Pat[pbase + j] = pr;              // saved BEFORE its dropout (the backward re-derives the mask)
                    if (p.dp > 0.f) pr = hash_uniform(p.seed_att, (unsigned long long)(pbase + j)) < p.dp ? 0.f : pr * keep;
                    float vv[TRF_HD];
                    row16(qb + j * TRF_QP + (2 * TRF_D + h * TRF_HD) * 2, vv);
#pragma unroll
                    for (int d = 0; d < TRF_HD; ++d) o[d] += pr * vv[d];
                }
            }
#pragma unroll
            for (int d = 0; d < TRF_HD; ++d) o[d] += __shfl_xor(o[d], 32);
            u32x4_t ov;
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[e] = hf ? pack2h(o[8 + 2 * e], o[9 + 2 * e]) : pack2h(o[2 * e], o[2 * e + 1]);
            *(lds_u32x4*)(aob + i * TRF_XP + (h * TRF_HD + hf * 8) * 2) = ov;
            if (lead) *reinterpret_cast<u32x4_t*>(p.ao + (row0 + i) * TRF_D + h * TRF_HD + hf * 8) = ov;
        } else {
            *(lds_u32x4*)(aob + i * TRF_XP + (h * TRF_HD + hf * 8) * 2) = u32x4_t{0u, 0u, 0u, 0u};
        }
    }
    __syncthreads();

    if (WIDE) load_w1();
    // W2 fragments of phase 6: requested only now — together with the attention's working set they exceeded the 256 registers of a wave
    // (28 spilled: a spilled prefetch register turns the asynchronous load into load-wait-store at its issue point)
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w2[ks] = *reinterpret_cast<const h16x8_t*>(wfrag_ptr(p.W2, wave * 16, TRF_FF, hq * TRF_HQ, lane) + ks * 512);
    TRF_STAMP(3);
    // ---- phase 3: y1 = x + drop(ao Wo^T + bo)  (in place in xf) [+ global]
    {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(aob + li * TRF_XP + ks * 64 + g * 16);
            acc[0] = MFMA_16x16x32_H(wo[ks], b0, acc[0], 0, 0, 0);
            if (mt1) {
                const h16x8_t b1 = *(__attribute__((address_space(3))) h16x8_t*)(aob + (16 + li) * TRF_XP + ks * 64 + g * 16);
                acc[1] = MFMA_16x16x32_H(wo[ks], b1, acc[1], 0, 0, 0);
            }
        }
        const int n = wave * 16 + g * 4;
        const f32x4 bb = pbo;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = mt * 16 + li;
            lds_char* xr = xf + (m * TRF_D + n) * 4;
            const f32x4 res = *(__attribute__((address_space(3))) f32x4*)xr;
            f32x4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[mt][r] + bb[r];
                if (p.dp > 0.f) v = hash_uniform(p.seed_o, (unsigned long long)((row0 + m) * TRF_D + n + r)) < p.dp ? 0.f : v * keep;
                y[r] = m < S ? v + res[r] : 0.f;
            }
            *(__attribute__((address_space(3))) f32x4*)xr = y;
            if (lead && m < S) *reinterpret_cast<f32x4*>(p.y1 + (row0 + m) * TRF_D + n) = y;
        }
    }
    __syncthreads();

    TRF_STAMP(4);
    // ---- phase 4: x1 = LN1(y1) -> x1f (fp32), x1b (16 bit) [+ global, statistics]
    {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = wave * 4 + rr;
            float v0 = *(lds_f32*)(xf + (r * TRF_D + lane) * 4), v1 = *(lds_f32*)(xf + (r * TRF_D + lane + 64) * 4);
            float o0 = 0.f, o1 = 0.f;
            if (r < S) {
                const float mean = wave_sum(v0 + v1) * (1.f / TRF_D);
                const float d0 = v0 - mean, d1 = v1 - mean;
                const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) * (1.f / TRF_D) + 1e-5f);
                o0 = d0 * rstd * pn1g0 + pn1b0;
                o1 = d1 * rstd * pn1g1 + pn1b1;
                if (lead) {
                    p.x1f[(row0 + r) * TRF_D + lane] = o0; p.x1f[(row0 + r) * TRF_D + lane + 64] = o1;
                    p.x1t[(row0 + r) * TRF_D + lane] = f2h(o0); p.x1t[(row0 + r) * TRF_D + lane + 64] = f2h(o1);
                    if (lane == 0) { p.st1[2 * (row0 + r)] = mean; p.st1[2 * (row0 + r) + 1] = rstd; }
                }
            }
            *(lds_f32*)(x1f + (r * TRF_D + lane) * 4) = o0; *(lds_f32*)(x1f + (r * TRF_D + lane + 64) * 4) = o1;
            *(lds_h16*)(x1b + r * TRF_XP + lane * 2) = f2h(o0); *(lds_h16*)(x1b + r * TRF_XP + (lane + 64) * 2) = f2h(o1);
        }
    }
    __syncthreads();

    TRF_STAMP(5);
    // ---- phase 5: h = drop(relu(x1 W1[q]^T + b1[q]))  -> hb (LDS) + global hff
    {
        f32x4 acc[4][2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { acc[nt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[nt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(x1b + li * TRF_XP + ks * 64 + g * 16);
            const h16x8_t b1 = *(__attribute__((address_space(3))) h16x8_t*)(x1b + (16 + li) * TRF_XP + ks * 64 + g * 16);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[nt][0] = MFMA_16x16x32_H(w1[nt][ks], b0, acc[nt][0], 0, 0, 0);
                if (mt1) acc[nt][1] = MFMA_16x16x32_H(w1[nt][ks], b1, acc[nt][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int nl = wave * 64 + nt * 16 + g * 4, n = hq * TRF_HQ + nl;
            const f32x4 bb = pb1[nt];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m = mt * 16 + li;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = fmaxf(acc[nt][mt][r] + bb[r], 0.f);
                    if (p.dp > 0.f) v[r] = hash_uniform(p.seed_h, (unsigned long long)((row0 + m) * TRF_FF + n + r)) < p.dp ? 0.f : v[r] * keep;
                    if (m >= S) v[r] = 0.f;
                }
                u32x2_t o;
                o[0] = pack2h(v[0], v[1]); o[1] = pack2h(v[2], v[3]);
                *(__attribute__((address_space(3))) u32x2_t*)(hb + m * TRF_HP + nl * 2) = o;
                if (m < S) *reinterpret_cast<u32x2_t*>(p.hff + (row0 + m) * TRF_FF + n) = o;
            }
        }
    }
    __syncthreads();

    TRF_STAMP(6);
    // ---- phase 6: y2 += drop(h W2[:, q]^T [+ b2]) [+ x1]   (fp32 atomics: four quarters per element)
    {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(hb + li * TRF_HP + ks * 64 + g * 16);
            acc[0] = MFMA_16x16x32_H(w2[ks], b0, acc[0], 0, 0, 0);
            if (mt1) {
                const h16x8_t b1 = *(__attribute__((address_space(3))) h16x8_t*)(hb + (16 + li) * TRF_HP + ks * 64 + g * 16);
                acc[1] = MFMA_16x16x32_H(w2[ks], b1, acc[1], 0, 0, 0);
            }
        }
        const int n = wave * 16 + g * 4;
        const f32x4 bb = pb2;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = mt * 16 + li;
            if (m >= S) continue;
            const f32x4 res = *(__attribute__((address_space(3))) f32x4*)(x1f + (m * TRF_D + n) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[mt][r] + bb[r];
                if (p.dp > 0.f) v = hash_uniform(p.seed_y, (unsigned long long)((row0 + m) * TRF_D + n + r)) < p.dp ? 0.f : v * keep;
                if (lead) v += res[r];
                unsafeAtomicAdd(p.y2 + (row0 + m) * TRF_D + n + r, v);
            }
        }
    }
    TRF_STAMP(7);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The FFN half of a layer's BACKWARD as one launch (was LayerNorm backward + the two data-gradient GEMMs: 11 + 18 + 15 us per layer, each
// latency bound): workgroup = (window, quarter of the hidden units), like the forward.
//   P0  dy2 = LN2_bwd(dx; y2, stats2, g2)           (all four quarters; quarter 0 adds the norm2 parameter gradients and writes dt_c)
//       dt_c = T(dropout_y(dy2))                      the 16-bit operand of linear2's weight gradient (and of P1)
//   P1  dh[:, q] = (dt_c W2[:, q]) / (1 - p) * (h > 0)   -> dt_a (16 bit, global: operand of linear1's weight gradient) and LDS
//   P2  part[q] = dh[:, q] W1[q, :]  (+ dy2 for quarter 0: the residual path)      -> fp32 partial [4][N][128]; the LayerNorm backward that
//       follows (norm1) sums the four partials while it loads its incoming gradient (no atomics: deterministic, nothing to zero)
// ---------------------------------------------------------------------------------------------------------------------------------
struct TrFfnBwdP {
    const float* dx; int bcast; float bdiv;      // incoming gradient [N][128] fp32, or (bcast) one row per window divided by bdiv
    const float *y2, *st2, *n2g;                 // LN2 input, (mean, rstd) per row, gamma
    float *dg2, *db2;                            // norm2 parameter gradients (atomics, quarter 0)
    const h16_t *W2t, *W1t;                      // [2048][128] = linear2.weight^T, [128][2048] = linear1.weight^T, fragment-ordered copies
    const h16_t* hff;                            // [N][2048] saved hidden (post ReLU + dropout)
    h16_t *dt_c, *dt_a;                          // [N][128], [N][2048] 16-bit gradient operands (written)
    float* part;                                 // [4][N][128]
    int B, S; long long N;
    float dp; unsigned long long seed_y;
};
constexpr size_t TRB_LDS = 32 * TRF_D * 4 + 32 * TRF_XP + 32 * TRF_HP + 2 * 8 * 256 * 4;

__global__ void __launch_bounds__(512) tr_ffn_bwd_kernel(TrFfnBwdP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef __attribute__((address_space(3))) h16_t lds_h16;
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    lds_char* const dyf = (lds_char*)smem;                     // [32][128] fp32: dy2
    lds_char* const dcb = dyf + 32 * TRF_D * 4;                // [32][XP] 16 bit: dt_c
    lds_char* const dhb = dcb + 32 * TRF_XP;                   // [32][HP] 16 bit: dh quarter
    lds_char* const red = dhb + 32 * TRF_HP;                   // [2][8 waves][256] fp32: parameter-gradient partials
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    // every phase is row-wise: a window of 32 < S <= 64 rows is walked as two 32-row halves by two workgroups per hidden quarter (w = window, S = this half's rows)
    const int hq = blockIdx.x % TRF_NQ, nh = (p.S + 31) >> 5, wh = blockIdx.x / TRF_NQ;
    const int w = wh / nh, r0 = (wh % nh) * 32;
    const int S = min(32, p.S - r0);
    const long long row0 = (long long)w * p.S + r0;
    const bool lead = hq == 0;
    const bool mt1 = S > 16;

    // weight fragments of P1 (W2t rows hq*512 + wave*64 + nt*16 + li, 4 k-steps over the 128 features): requested first
    h16x8_t w2[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            w2[nt][ks] = *reinterpret_cast<const h16x8_t*>(wfrag_ptr(p.W2t, hq * TRF_HQ + wave * 64 + nt * 16, TRF_D, 0, lane) + ks * 512);

    // ---- P0: LayerNorm backward of the rows of this window (wave = 4 rows, lane = columns lane and lane + 64)
    {
        const float g0 = p.n2g[lane], g1 = p.n2g[lane + 64];
        float sg0 = 0.f, sg1 = 0.f, sb0 = 0.f, sb1 = 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = wave * 4 + rr;
            float o0 = 0.f, o1 = 0.f, t0 = 0.f, t1 = 0.f;
            if (r < S) {
                const long long row = row0 + r;
                const float mean = p.st2[2 * row], rstd = p.st2[2 * row + 1];
                const float* xr = p.y2 + row * TRF_D;
                const float* dr = p.dx + (p.bcast ? (long long)w : row) * TRF_D;
                const float xh0 = (xr[lane] - mean) * rstd, xh1 = (xr[lane + 64] - mean) * rstd;
                float d0 = dr[lane], d1 = dr[lane + 64];
                if (p.bcast) { d0 /= p.bdiv; d1 /= p.bdiv; }
                sg0 += d0 * xh0; sb0 += d0; sg1 += d1 * xh1; sb1 += d1;
                const float q0 = d0 * g0, q1 = d1 * g1;
                const float m1 = wave_sum(q0 + q1) / TRF_D;
                const float m2 = wave_sum(q0 * xh0 + q1 * xh1) / TRF_D;
                o0 = rstd * (q0 - m1 - xh0 * m2); o1 = rstd * (q1 - m1 - xh1 * m2);
                t0 = o0; t1 = o1;
                if (p.dp > 0.f) {
                    t0 = hash_uniform(p.seed_y, (unsigned long long)(row * TRF_D + lane)) < p.dp ? 0.f : t0 / (1.f - p.dp);
                    t1 = hash_uniform(p.seed_y, (unsigned long long)(row * TRF_D + lane + 64)) < p.dp ? 0.f : t1 / (1.f - p.dp);
                }
                if (lead) { p.dt_c[row * TRF_D + lane] = f2h(t0); p.dt_c[row * TRF_D + lane + 64] = f2h(t1); }
            }
            *(lds_f32*)(dyf + (r * TRF_D + lane) * 4) = o0; *(lds_f32*)(dyf + (r * TRF_D + lane + 64) * 4) = o1;
            *(lds_h16*)(dcb + r * TRF_XP + lane * 2) = f2h(t0); *(lds_h16*)(dcb + r * TRF_XP + (lane + 64) * 2) = f2h(t1);
        }
        if (lead) {
            *(lds_f32*)(red + (wave * 256 + lane) * 4) = sg0; *(lds_f32*)(red + (wave * 256 + 64 + lane) * 4) = sg1;
            *(lds_f32*)(red + (wave * 256 + 128 + lane) * 4) = sb0; *(lds_f32*)(red + (wave * 256 + 192 + lane) * 4) = sb1;
        }
    }
    // mask operand of P1 (the saved hidden of this quarter) and the weight fragments of P2: independent of everything above
    u32x2_t hm[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = mt * 16 + li;
            hm[nt][mt] = u32x2_t{0u, 0u};
            if (m < S) hm[nt][mt] = *reinterpret_cast<const u32x2_t*>(p.hff + (row0 + m) * TRF_FF + hq * TRF_HQ + wave * 64 + nt * 16 + g * 4);
        }
    h16x8_t w1[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w1[ks] = *reinterpret_cast<const h16x8_t*>(wfrag_ptr(p.W1t, wave * 16, TRF_FF, hq * TRF_HQ, lane) + ks * 512);
    __syncthreads();
    if (lead && tid < 256) {          // norm2 parameter gradients: 8 wave partials per column -> one atomic each (columns 0..127 dgamma, 128..255 dbeta)
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) sum += *(lds_f32*)(red + (ww * 256 + tid) * 4);
        unsafeAtomicAdd((tid < 128 ? p.dg2 : p.db2) + (tid & 127), sum);
    }

    // ---- P1: dh = (dt_c W2[:, q]) * alpha * (h > 0)  -> dhb (LDS) + dt_a (global)
    {
        const float alpha = p.dp > 0.f ? 1.f / (1.f - p.dp) : 1.f;
        f32x4 acc[4][2];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { acc[nt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[nt][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(dcb + li * TRF_XP + ks * 64 + g * 16);
            const h16x8_t b1 = *(__attribute__((address_space(3))) h16x8_t*)(dcb + (16 + li) * TRF_XP + ks * 64 + g * 16);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                acc[nt][0] = MFMA_16x16x32_H(w2[nt][ks], b0, acc[nt][0], 0, 0, 0);
                if (mt1) acc[nt][1] = MFMA_16x16x32_H(w2[nt][ks], b1, acc[nt][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int nl = wave * 64 + nt * 16 + g * 4, n = hq * TRF_HQ + nl;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int m = mt * 16 + li;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = acc[nt][mt][r] * alpha;
                v[0] = h2f_lo(hm[nt][mt][0]) > 0.f ? v[0] : 0.f; v[1] = h2f_hi(hm[nt][mt][0]) > 0.f ? v[1] : 0.f;
                v[2] = h2f_lo(hm[nt][mt][1]) > 0.f ? v[2] : 0.f; v[3] = h2f_hi(hm[nt][mt][1]) > 0.f ? v[3] : 0.f;
                u32x2_t o;
                o[0] = pack2h(v[0], v[1]); o[1] = pack2h(v[2], v[3]);
                if (m >= S) o = u32x2_t{0u, 0u};
                *(__attribute__((address_space(3))) u32x2_t*)(dhb + m * TRF_HP + nl * 2) = o;
                if (m < S) *reinterpret_cast<u32x2_t*>(p.dt_a + (row0 + m) * TRF_FF + n) = o;
            }
        }
    }
    __syncthreads();

    // ---- P2: part[q] = dh[:, q] W1[q, :] (+ dy2 for quarter 0)
    {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(dhb + li * TRF_HP + ks * 64 + g * 16);
            acc[0] = MFMA_16x16x32_H(w1[ks], b0, acc[0], 0, 0, 0);
            if (mt1) {
                const h16x8_t b1 = *(__attribute__((address_space(3))) h16x8_t*)(dhb + (16 + li) * TRF_HP + ks * 64 + g * 16);
                acc[1] = MFMA_16x16x32_H(w1[ks], b1, acc[1], 0, 0, 0);
            }
        }
        const int n = wave * 16 + g * 4;
        float* const out = p.part + (long long)hq * p.N * TRF_D;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = mt * 16 + li;
            if (m >= S) continue;
            f32x4 v = acc[mt];
            if (lead) { const f32x4 res = *(__attribute__((address_space(3))) f32x4*)(dyf + (m * TRF_D + n) * 4); v += res; }
            *reinterpret_cast<f32x4*>(out + (row0 + m) * TRF_D + n) = v;
        }
    }
}
// ---------------------------------------------------------------------------------------------------------------------------------
// The attention half of a layer's BACKWARD as one launch (was LayerNorm backward 9 us + out_proj data gradient 6.6 + attention backward 15 +
// in_proj data gradient 8 per layer, each a latency chain over 2048 x 128 values): one workgroup per WINDOW, wave = head.
//   P0  dy1 = LN1_bwd(sum of the incoming partials; y1, stats1, g1)  -> dy_f (fp32, global: re-read as the residual in P3), norm1 parameter
//       gradients (atomics);  b_d = T(dropout_o(dy1))  -> LDS + global (operand of out_proj's weight gradient)
//   P1  dao = b_d W_o: wave w computes the 16 features of head w -> that head's dO rows in LDS (16 bit, as the unfused path rounds them)
//   P2  attention backward of head w (attention_bwd32_kernel's arithmetic: two lanes per query row)  -> b_b = d qkv (16 bit) in LDS + global
//       (operand of in_proj's weight gradient)
//   P3  dx = b_b W_in + dy1  -> fp32, global: the layer's input gradient
// Measured (rocprofv3, B = 64 x S = 32): 33 us per layer against 9.1 + 6.6 + 15.2 + 8.2 = 39 us and three launch gaps; phases switched off (experiment
// build, HULC_TRA_DBG): without P0 28 us, without P3 31.8, everything off (weight / qkv loads, barriers) 8.6 — the attention chain of P2 is the
// long pole.  Tried: 16 waves (two rows per wave in P0, one MFMA tile per wave, four lanes per query row): 35 - 38 us, slower; loads of P0 hoisted
// over its four rows and the saved probabilities requested before P1: no change.
// S <= 32, 8 heads of 16.  LDS: b_d 9 KB + d qkv 25 KB + 8 x (q, k, v, dO 16 bit 4.5 KB + dS, dropped P fp32 8.3 KB) + 8 KB of reduction space.
// ---------------------------------------------------------------------------------------------------------------------------------
struct TrAttnBwdP {
    const float* parts; long long part_stride; int nparts;    // incoming gradient of norm1's output: the sum of nparts arrays [N][128]
    const float *y1, *st1, *n1g;                 // LN1 input, (mean, rstd) per row, gamma
    float *dg1, *db1;                            // norm1 parameter gradients (atomics)
    const h16_t *Wot, *Wint;                     // [128][128] = out_proj.weight^T, [128][384] = in_proj_weight^T, fragment-ordered copies
    const h16_t* qkv; const float* Pat;          // saved [N][384] and attention probabilities [B * 8][S][S]
    h16_t *b_d, *b_b;                            // [N][128], [N][384] 16-bit gradient operands (written)
    float *dy_f, *dx;                            // [N][128] fp32
    int B, S;
    float dp; unsigned long long seed_o, seed_att;
    int dbg;                                     // experiment builds: phases switched off (1: P0, 2: P2, 4: P1, 8: P3)
};
constexpr int TRA_HP = 48;                                          // bytes per row of a head's [32][16] 16-bit operand: two 16-byte reads per row
constexpr int TRA_HEADB = 4 * 32 * TRA_HP + 2 * 32 * 33 * 4;        // q, k, v, dO + dS, dropped P
constexpr size_t TRA_LDS = 32 * TRF_XP + 32 * TRF_QP + 8 * 256 * 4 + 8 * TRA_HEADB;

__global__ void __launch_bounds__(512) tr_attn_bwd_kernel(TrAttnBwdP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef __attribute__((address_space(3))) h16_t lds_h16;
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    lds_char* const bdb = (lds_char*)smem;                     // [32][XP] 16 bit: b_d
    lds_char* const dqb = bdb + 32 * TRF_XP;                   // [32][QP] 16 bit: d qkv
    lds_char* const red = dqb + 32 * TRF_QP;                   // [8 waves][256] fp32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    lds_char* const hb = red + 8 * 256 * 4 + wave * TRA_HEADB; // this wave's head
    lds_char* const qh = hb, * const kh = hb + 32 * TRA_HP, * const vh = hb + 2 * 32 * TRA_HP, * const doh = hb + 3 * 32 * TRA_HP;
    lds_char* const dSb = hb + 4 * 32 * TRA_HP;                // [32][33] fp32
    lds_char* const Pdb = dSb + 32 * 33 * 4;
    const int w = blockIdx.x, S = p.S;
    const long long row0 = (long long)w * S;
    const bool mt1 = true;        // both 16-row tiles always (rows past S are zero).  With the second tile's MFMAs under `S > 16`, as in the kernels above,
    // S <= 16 produced wrong d k / d v (transposed dS / P reads) while S >= 17 was exact — not understood; tools/tr_bwd_probe.py + tr_bwd_cmp.py (experiment build, HULC_TR_ATTN_BWD=0 / 1) reproduce it

    // out_proj^T fragments (rows = the 16 features of head `wave`, 4 k-steps over the 128 columns of b_d): requested first
    h16x8_t wo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wo[ks] = *reinterpret_cast<const h16x8_t*>(wfrag_ptr(p.Wot, wave * 16, TRF_D, 0, lane) + ks * 512);

    // ---- P0: LayerNorm backward (wave = 4 rows, lane = columns lane and lane + 64), layernorm_bwd_fused_kernel's arithmetic
    {
        const float g0 = p.n1g[lane], g1 = p.n1g[lane + 64];
        float sg0 = 0.f, sg1 = 0.f, sb0 = 0.f, sb1 = 0.f;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = wave * 4 + rr;
            float t0 = 0.f, t1 = 0.f;
            if (r < S && !(p.dbg & 1)) {
                const long long row = row0 + r;
                const float mean = p.st1[2 * row], rstd = p.st1[2 * row + 1];
                const float* xr = p.y1 + row * TRF_D;
                const float* dr = p.parts + row * TRF_D;
                const float xh0 = (xr[lane] - mean) * rstd, xh1 = (xr[lane + 64] - mean) * rstd;
                float d0 = dr[lane], d1 = dr[lane + 64];
                for (int pp = 1; pp < p.nparts; ++pp) { d0 += dr[pp * p.part_stride + lane]; d1 += dr[pp * p.part_stride + lane + 64]; }
                sg0 += d0 * xh0; sb0 += d0; sg1 += d1 * xh1; sb1 += d1;
                const float q0 = d0 * g0, q1 = d1 * g1;
                const float m1 = wave_sum(q0 + q1) / TRF_D;
                const float m2 = wave_sum(q0 * xh0 + q1 * xh1) / TRF_D;
                const float o0 = rstd * (q0 - m1 - xh0 * m2), o1 = rstd * (q1 - m1 - xh1 * m2);
                p.dy_f[row * TRF_D + lane] = o0; p.dy_f[row * TRF_D + lane + 64] = o1;
                t0 = o0; t1 = o1;
                if (p.dp > 0.f) {
                    t0 = hash_uniform(p.seed_o, (unsigned long long)(row * TRF_D + lane)) < p.dp ? 0.f : t0 / (1.f - p.dp);
                    t1 = hash_uniform(p.seed_o, (unsigned long long)(row * TRF_D + lane + 64)) < p.dp ? 0.f : t1 / (1.f - p.dp);
                }
                p.b_d[row * TRF_D + lane] = f2h(t0); p.b_d[row * TRF_D + lane + 64] = f2h(t1);
            }
            *(lds_h16*)(bdb + r * TRF_XP + lane * 2) = f2h(t0); *(lds_h16*)(bdb + r * TRF_XP + (lane + 64) * 2) = f2h(t1);
        }
        *(lds_f32*)(red + (wave * 256 + lane) * 4) = sg0; *(lds_f32*)(red + (wave * 256 + 64 + lane) * 4) = sg1;
        *(lds_f32*)(red + (wave * 256 + 128 + lane) * 4) = sb0; *(lds_f32*)(red + (wave * 256 + 192 + lane) * 4) = sb1;
    }
    // this head's q (pre-scaled by 1/4: exact), k, v rows and the in_proj^T fragments of P3: independent of everything above
    const int ai = lane & 31, hf = lane >> 5;
    if (ai < S) {
        const h16_t* r = p.qkv + (row0 + ai) * 3 * TRF_D + wave * TRF_HD;
        h16_t a0[16], a1[16];
        if (hf == 0) {
            load8<h16_t>(r, *reinterpret_cast<h16_t(*)[8]>(a0)); load8<h16_t>(r + 8, *reinterpret_cast<h16_t(*)[8]>(a0 + 8));
            load8<h16_t>(r + TRF_D, *reinterpret_cast<h16_t(*)[8]>(a1)); load8<h16_t>(r + TRF_D + 8, *reinterpret_cast<h16_t(*)[8]>(a1 + 8));
#pragma unroll
            for (int d = 0; d < 16; ++d) { *(lds_h16*)(qh + ai * TRA_HP + d * 2) = f2h(h2f(a0[d]) * 0.25f); *(lds_h16*)(kh + ai * TRA_HP + d * 2) = a1[d]; }
        } else {
            load8<h16_t>(r + 2 * TRF_D, *reinterpret_cast<h16_t(*)[8]>(a0)); load8<h16_t>(r + 2 * TRF_D + 8, *reinterpret_cast<h16_t(*)[8]>(a0 + 8));
#pragma unroll
            for (int d = 0; d < 16; ++d) *(lds_h16*)(vh + ai * TRA_HP + d * 2) = a0[d];
        }
    }
    h16x8_t wi[12];
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) wi[ks] = *reinterpret_cast<const h16x8_t*>(wfrag_ptr(p.Wint, wave * 16, 3 * TRF_D, 0, lane) + ks * 512);
    __syncthreads();
    if (tid < 256) {          // norm1 parameter gradients: 8 wave partials per column -> one atomic each
        float sum = 0.f;
#pragma unroll
        for (int ww = 0; ww < 8; ++ww) sum += *(lds_f32*)(red + (ww * 256 + tid) * 4);
        unsafeAtomicAdd((tid < 128 ? p.dg1 : p.db1) + (tid & 127), sum);
    }

    // ---- P1: dao[:, head] = b_d W_o[:, head]  -> dO of this head (16 bit)
    if (!(p.dbg & 4)) {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(bdb + li * TRF_XP + ks * 64 + g * 16);
            acc[0] = MFMA_16x16x32_H(wo[ks], b0, acc[0], 0, 0, 0);
            if (mt1) {
                const h16x8_t b1 = *(__attribute__((address_space(3))) h16x8_t*)(bdb + (16 + li) * TRF_XP + ks * 64 + g * 16);
                acc[1] = MFMA_16x16x32_H(wo[ks], b1, acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = mt * 16 + li;
#pragma unroll
            for (int r = 0; r < 4; ++r) *(lds_h16*)(doh + m * TRA_HP + (g * 4 + r) * 2) = f2h(acc[mt][r]);
        }
    }
    __syncthreads();

    // ---- P2: attention backward of head `wave` (lane = query row ai, half hf of the keys)
    auto row16 = [](lds_char* base, int i, float (&x)[16]) {      // a head's 16 values of row i: two 16-byte LDS reads, unpacked
        const u32x4_t a = *(lds_u32x4*)(base + i * TRA_HP), b = *(lds_u32x4*)(base + i * TRA_HP + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[2 * e] = h2f_lo(a[e]); x[2 * e + 1] = h2f_hi(a[e]); x[8 + 2 * e] = h2f_lo(b[e]); x[8 + 2 * e + 1] = h2f_hi(b[e]); }
    };
    constexpr int HJ = 16;
    if (ai < S && !(p.dbg & 2)) {
        const float* Pr = p.Pat + (((long long)w * TRF_NH + wave) * S + ai) * S;
        float dot = 0.f;
        float dpv[HJ], pv[HJ];
        float dOi[16];
        row16(doh, ai, dOi);
#pragma unroll
        for (int jj = 0; jj < HJ; ++jj) {
            const int j = hf * HJ + jj;
            dpv[jj] = 0.f; pv[jj] = 0.f;
            if (j < S) {
                float dpj = 0.f, vj[16];
                row16(vh, j, vj);
#pragma unroll
                for (int d = 0; d < 16; ++d) dpj += dOi[d] * vj[d];
                const float pr = Pr[j];
                float keep = 1.f;
                if (p.dp > 0.f) keep = hash_uniform(p.seed_att, (((long long)w * TRF_NH + wave) * S + ai) * S + j) < p.dp ? 0.f : 1.f / (1.f - p.dp);
                *(lds_f32*)(Pdb + (ai * 33 + j) * 4) = pr * keep;
                dpj *= keep;
                dpv[jj] = dpj; pv[jj] = pr;
                dot += dpj * pr;
            }
        }
        dot += __shfl_xor(dot, 32);
#pragma unroll
        for (int jj = 0; jj < HJ; ++jj) if (hf * HJ + jj < S) *(lds_f32*)(dSb + (ai * 33 + hf * HJ + jj) * 4) = pv[jj] * (dpv[jj] - dot);
    }
    __syncthreads();
    if (ai < S && !(p.dbg & 2)) {
        float dq[16], dk[16], dv[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) { dq[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
        for (int jj = 0; jj < HJ; ++jj) {
            const int j = hf * HJ + jj;
            if (j >= S) break;
            const float s_ij = *(lds_f32*)(dSb + (ai * 33 + j) * 4), s_ji = *(lds_f32*)(dSb + (j * 33 + ai) * 4), p_ji = *(lds_f32*)(Pdb + (j * 33 + ai) * 4);
            float kj[16], qj[16], oj[16];
            row16(kh, j, kj); row16(qh, j, qj); row16(doh, j, oj);
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                dq[d] += s_ij * kj[d];
                dk[d] += s_ji * qj[d];          // q already carries the 1/sqrt(hd) scale
                dv[d] += p_ji * oj[d];
            }
        }
#pragma unroll
        for (int d = 0; d < 16; ++d) { dq[d] += __shfl_xor(dq[d], 32); dk[d] += __shfl_xor(dk[d], 32); dv[d] += __shfl_xor(dv[d], 32); }
        Vec8<h16_t> oq, ok, ov;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            oq.v[d] = f2h((hf ? dq[8 + d] : dq[d]) * 0.25f); ok.v[d] = f2h(hf ? dk[8 + d] : dk[d]); ov.v[d] = f2h(hf ? dv[8 + d] : dv[d]);
        }
        h16_t* o = p.b_b + (row0 + ai) * 3 * TRF_D + wave * TRF_HD + hf * 8;
        *reinterpret_cast<Vec8<h16_t>*>(o) = oq; *reinterpret_cast<Vec8<h16_t>*>(o + TRF_D) = ok; *reinterpret_cast<Vec8<h16_t>*>(o + 2 * TRF_D) = ov;
        lds_char* lo = dqb + ai * TRF_QP + (wave * TRF_HD + hf * 8) * 2;
        *(lds_u32x4*)(lo) = __builtin_bit_cast(u32x4_t, oq);
        *(lds_u32x4*)(lo + TRF_D * 2) = __builtin_bit_cast(u32x4_t, ok);
        *(lds_u32x4*)(lo + 2 * TRF_D * 2) = __builtin_bit_cast(u32x4_t, ov);
    }
    __syncthreads();

    // ---- P3: dx = d qkv W_in + dy1 (features wave * 16 .. + 15)
    if (!(p.dbg & 8)) {
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(dqb + li * TRF_QP + ks * 64 + g * 16);
            acc[0] = MFMA_16x16x32_H(wi[ks], b0, acc[0], 0, 0, 0);
            if (mt1) {
                const h16x8_t b1 = *(__attribute__((address_space(3))) h16x8_t*)(dqb + (16 + li) * TRF_QP + ks * 64 + g * 16);
                acc[1] = MFMA_16x16x32_H(wi[ks], b1, acc[1], 0, 0, 0);
            }
        }
        const int n = wave * 16 + g * 4;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int m = mt * 16 + li;
            if (m >= S) continue;
            const f32x4 res = *reinterpret_cast<const f32x4*>(p.dy_f + (row0 + m) * TRF_D + n);      // written in P0 by this workgroup (two barriers ago)
            *reinterpret_cast<f32x4*>(p.dx + (row0 + m) * TRF_D + n) = acc[mt] + res;
        }
    }
}
static inline void launch_tr_attn_bwd(hipStream_t st, const TrAttnBwdP& p) {
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute((const void*)tr_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRA_LDS); attr_set = true; }
    hipLaunchKernelGGL(tr_attn_bwd_kernel, dim3(p.B), dim3(512), TRA_LDS, st, p);
}

static inline void launch_tr_ffn_bwd(hipStream_t st, const TrFfnBwdP& p) {
    static bool attr_set = false;
    if (!attr_set) { hipFuncSetAttribute((const void*)tr_ffn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRB_LDS); attr_set = true; }
    hipLaunchKernelGGL(tr_ffn_bwd_kernel, dim3(p.B * ((p.S + 31) / 32) * TRF_NQ), dim3(512), TRB_LDS, st, p);
}

static inline void launch_tr_layer_fwd(hipStream_t st, const TrLayerP& p) {
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)tr_layer_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRF_LDS);
        hipFuncSetAttribute((const void*)tr_layer_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TRF_LDS_WIDE);
        attr_set = true;
    }
    if (p.S > 32) hipLaunchKernelGGL(tr_layer_fwd_kernel<true>, dim3(p.B * 2 * TRF_NQ), dim3(512), TRF_LDS_WIDE, st, p);      // 32 < S <= 64: (window, half, quarter)
    else hipLaunchKernelGGL(tr_layer_fwd_kernel<false>, dim3(p.B * TRF_NQ), dim3(512), TRF_LDS, st, p);
}

}  // namespace HULC_NS
