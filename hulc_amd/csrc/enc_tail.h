// hulc_amd/csrc/enc_tail.h — the dense tail of BOTH perceptual encoders' forward as ONE launch (16-bit engines).
//
// Reference: VisionNetwork.forward after the spatial softmax / VisionNetworkGripper.forward after its conv stack's first Linear
// (hulc/models/perceptual_encoders/vision_network.py:38-61, vision_network_gripper.py:12-38):
//     f1 = relu(x W1^T + b1)  (128 -> 512),   f2 = f1 W2^T + b2  (512 -> 64),   emb[:, col0 : col0 + 64] = LayerNorm(f2)
// per frame, x = the 128 spatial-softmax coordinates (static camera) or the gripper camera's fc7 output.
// Unfused these are three latency-bound launches per camera (12.8 + 5.5 + 5.7 us / 10.2 + 4.9 + 4.9 us for 2048 frames x 0.2 MFLOP); here a
// workgroup owns 16 frames of one camera (blockIdx.y): the 16 x 128 input tile goes to LDS, W1 / W2 fragments to registers (8 waves: wave w
// owns 64 hidden units of fc1; for fc2 wave (n-tile w & 3, K-half w >> 2)), the hidden tile stays in LDS (and is written out: the backward
// needs f1, f2 and the LayerNorm statistics), the two K-halves of fc2 meet in LDS, wave w normalises rows 2w, 2w + 1 (lane = feature).
#pragma once
#include "common.h"
#include "conv_wgrad.h"   // lds_char

namespace HULC_NS {

struct EncTailCam {
    const h16_t* x;                  // [Nf][128]
    const h16_t *W1, *W2;            // [512][128], [64][512]
    const float *b1, *b2, *lng, *lnb;
    h16_t* f1; float* f2; float* lnst;      // saved: [Nf][512], [Nf][64], [Nf][2]
    int col0;                        // column of this camera's 64 features in the embedding
};
// pos != null: the plan-recognition transformer's input is produced here too (posadd_kernel's job, plan_recognition_net.py:96-103):
// x0 = dropout(emb + pos[t]) as fp32 `xf` and 16-bit `xt`, and the two FFN accumulators z0 / z1 of the fused layers start at zero
struct EncTailP { EncTailCam cam[2]; h16_t* emb; int Nf, ldemb; const float* pos; float *xf, *z0, *z1; h16_t* xt; int S; float drop_p; unsigned long long seed; };

constexpr int ET_XP = 128 * 2 + 16, ET_HP = 512 * 2 + 32;
constexpr size_t ET_LDS = 16 * ET_XP + 16 * ET_HP + 2 * 16 * 65 * 4;

__global__ void __launch_bounds__(512) enc_tail_fwd_kernel(EncTailP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    lds_char* const xb = (lds_char*)smem;                   // [16][XP] 16 bit
    lds_char* const hb = xb + 16 * ET_XP;                   // [16][HP] 16 bit
    lds_char* const yb = hb + 16 * ET_HP;                   // [2][16][65] fp32
    const EncTailCam c = p.cam[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const long long row0 = (long long)blockIdx.x * 16;
    const int rows = (int)min((long long)16, (long long)p.Nf - row0);

    // fc1 fragments (hidden units wave*64 + nt*16 + li) and every small parameter: requested first
    h16x8_t w1[4][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w1[nt][ks] = *reinterpret_cast<const h16x8_t*>(c.W1 + (long long)(wave * 64 + nt * 16 + li) * 128 + ks * 32 + g * 8);
    f32x4 pb1[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) pb1[nt] = *reinterpret_cast<const f32x4*>(c.b1 + wave * 64 + nt * 16 + g * 4);
    const int nt2 = wave & 3, kh = wave >> 2;
    const f32x4 pb2 = kh == 0 ? *reinterpret_cast<const f32x4*>(c.b2 + nt2 * 16 + g * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float lg = c.lng[lane], lb = c.lnb[lane];
    // the 16 x 128 input tile: 256 threads x 16 bytes (rows past the end: zeros)
    if (tid < 256) {
        const int r = tid >> 4, ch = tid & 15;
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (r < rows) v = *reinterpret_cast<const u32x4_t*>(c.x + (row0 + r) * 128 + ch * 8);
        *(lds_u32x4*)(xb + r * ET_XP + ch * 16) = v;
    }
    h16x8_t w2[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w2[ks] = *reinterpret_cast<const h16x8_t*>(c.W2 + (long long)(nt2 * 16 + li) * 512 + kh * 256 + ks * 32 + g * 8);
    __syncthreads();
    // ---- fc1 + ReLU -> hb (LDS) + f1 (global)
    {
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(xb + li * ET_XP + ks * 64 + g * 16);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA_16x16x32_H(w1[nt][ks], b0, acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = wave * 64 + nt * 16 + g * 4;
            u32x2_t o;
            o[0] = pack2h(fmaxf(acc[nt][0] + pb1[nt][0], 0.f), fmaxf(acc[nt][1] + pb1[nt][1], 0.f));
            o[1] = pack2h(fmaxf(acc[nt][2] + pb1[nt][2], 0.f), fmaxf(acc[nt][3] + pb1[nt][3], 0.f));
            *(__attribute__((address_space(3))) u32x2_t*)(hb + li * ET_HP + n * 2) = o;
            if (li < rows) *reinterpret_cast<u32x2_t*>(c.f1 + (row0 + li) * 512 + n) = o;
        }
    }
    __syncthreads();
    // ---- fc2: wave (n-tile nt2, K half kh)
    {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(hb + li * ET_HP + (kh * 256 + ks * 32) * 2 + g * 16);
            acc = MFMA_16x16x32_H(w2[ks], b0, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) *(lds_f32*)(yb + ((kh * 16 + li) * 65 + nt2 * 16 + g * 4 + r) * 4) = acc[r] + pb2[r];
    }
    __syncthreads();
    // ---- LayerNorm of rows 2 wave, 2 wave + 1 (lane = feature), layernorm_fwd_kernel's arithmetic
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = wave * 2 + rr;
        if (r >= rows) break;
        const float y = *(lds_f32*)(yb + (r * 65 + lane) * 4) + *(lds_f32*)(yb + ((16 + r) * 65 + lane) * 4);
        const long long row = row0 + r;
        c.f2[row * 64 + lane] = y;
        const float mean = wave_sum(y) / 64;
        const float d = y - mean;
        const float var = wave_sum(d * d) / 64;
        const float rstd = rsqrtf(var + 1e-5f);
        const h16_t e16 = f2h(d * rstd * lg + lb);
        p.emb[row * p.ldemb + c.col0 + lane] = e16;
        if (lane == 0) { c.lnst[2 * row] = mean; c.lnst[2 * row + 1] = rstd; }
        if (p.pos) {
            const long long idx = row * p.ldemb + c.col0 + lane;
            const int t = (int)(row % p.S);
            float v = to_f<h16_t>(e16) + p.pos[t * p.ldemb + c.col0 + lane];
            if (p.drop_p > 0.f) v = hash_uniform(p.seed, idx) < p.drop_p ? 0.f : v / (1.f - p.drop_p);
            p.xf[idx] = v; p.xt[idx] = from_f<h16_t>(v);
            p.z0[idx] = 0.f; p.z1[idx] = 0.f;
        }
    }
}
static inline void launch_enc_tail_fwd(hipStream_t st, const EncTailP& p) {
    hipLaunchKernelGGL(enc_tail_fwd_kernel, dim3((p.Nf + 15) / 16, 2), dim3(512), ET_LDS, st, p);
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward data path of the same tail, both cameras in one launch: LayerNorm backward (+ its gamma / beta gradients), fc2 data gradient through
// the ReLU mask of f1, fc1 data gradient (static camera: fp32, feeds the spatial-softmax backward; gripper camera: through the ReLU mask of the
// fc7 output, 16 bit).  d_f2 and d_f1 are also written out: the weight gradients of fc2 / fc1 are separate GEMMs over all frames.
// Unfused: layernorm_bwd_fused 8.5 us + 64x64-tile GEMM 11.4 us + skinny GEMM 5.7 us per camera.
// ---------------------------------------------------------------------------------------------------------------------
struct EncTailBwdCam {
    const float *f2, *lnst, *lng;
    const h16_t *f1, *W2t, *W1t;     // f1 [Nf][512]; W2t [512][64]; W1t [128][512]
    const h16_t* xmask;              // gripper: the fc7 output [Nf][128] (ReLU mask of dx); static: null
    float *dlng, *dlnb;
    h16_t *d_f2, *d_f1;              // [Nf][64], [Nf][512]
    float* dx_f32; h16_t* dx_t;      // [Nf][128]: one of the two
    int col0;
};
struct EncTailBwdP { EncTailBwdCam cam[2]; const float* demb; int Nf, ldemb; };
constexpr int ET_DP = 64 * 2 + 16;
constexpr size_t ET_BWD_LDS = 16 * ET_DP + 16 * ET_HP + 8 * 128 * 4;

__global__ void __launch_bounds__(512) enc_tail_bwd_kernel(EncTailBwdP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) float lds_f32;
    typedef __attribute__((address_space(3))) h16_t lds_h16;
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    lds_char* const db_ = (lds_char*)smem;                  // [16][DP] d_f2, 16 bit
    lds_char* const hb = db_ + 16 * ET_DP;                  // [16][HP] d_f1, 16 bit
    lds_char* const red = hb + 16 * ET_HP;                  // [8][128] fp32: per-wave gamma / beta partials
    const EncTailBwdCam c = p.cam[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const long long row0 = (long long)blockIdx.x * 16;
    const int rows = (int)min((long long)16, (long long)p.Nf - row0);

    // LayerNorm-backward inputs of rows 2 wave, 2 wave + 1 first, then the weight fragments of both GEMMs
    float dy[2], y[2], mean[2], rstd[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = wave * 2 + rr;
        const long long row = row0 + (r < rows ? r : 0);
        dy[rr] = r < rows ? p.demb[row * p.ldemb + c.col0 + lane] : 0.f;
        y[rr] = c.f2[row * 64 + lane]; mean[rr] = c.lnst[2 * row]; rstd[rr] = c.lnst[2 * row + 1];
    }
    const float lg = c.lng[lane];
    h16x8_t w2[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) w2[nt][ks] = *reinterpret_cast<const h16x8_t*>(c.W2t + (long long)(wave * 64 + nt * 16 + li) * 64 + ks * 32 + g * 8);
    u32x2_t fm[4];                                           // f1 of (row li, hidden wave*64 + nt*16 + g*4 ..+3): the ReLU mask
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) fm[nt] = li < rows ? *reinterpret_cast<const u32x2_t*>(c.f1 + (row0 + li) * 512 + wave * 64 + nt * 16 + g * 4) : u32x2_t{0u, 0u};
    h16x8_t w1[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) w1[ks] = *reinterpret_cast<const h16x8_t*>(c.W1t + (long long)(wave * 16 + li) * 512 + ks * 32 + g * 8);
    u32x2_t xm = u32x2_t{0x3c003c00u, 0x3c003c00u};          // any positive pattern: no mask
    if (c.xmask && li < rows) xm = *reinterpret_cast<const u32x2_t*>(c.xmask + (row0 + li) * 128 + wave * 16 + g * 4);

    // ---- LayerNorm backward (layernorm_bwd_fused_kernel's arithmetic, n = 64)
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = wave * 2 + rr;
        const float xh = (y[rr] - mean[rr]) * rstd[rr];
        sg += dy[rr] * xh; sb += dy[rr];
        const float q = dy[rr] * lg;
        const float m1 = wave_sum(q) / 64;
        const float m2 = wave_sum(q * xh) / 64;
        const float d = r < rows ? rstd[rr] * (q - m1 - xh * m2) : 0.f;
        const h16_t dh = f2h(d);
        *(lds_h16*)(db_ + r * ET_DP + lane * 2) = dh;
        if (r < rows) c.d_f2[(row0 + r) * 64 + lane] = dh;
    }
    *(lds_f32*)(red + (wave * 128 + lane) * 4) = sg;
    *(lds_f32*)(red + (wave * 128 + 64 + lane) * 4) = sb;
    __syncthreads();
    if (tid < 128) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) s += *(lds_f32*)(red + (w * 128 + tid) * 4);
        unsafeAtomicAdd((tid < 64 ? c.dlng : c.dlnb) + (tid & 63), s);
    }
    // ---- d_f1 = (d_f2 W2) masked by f1 > 0
    {
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(db_ + li * ET_DP + ks * 64 + g * 16);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA_16x16x32_H(w2[nt][ks], b0, acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = wave * 64 + nt * 16 + g * 4;
            // f1 is a ReLU output (>= 0 or -0): "f1 > 0" is "magnitude bits != 0"
            const float v0 = (fm[nt][0] & 0x7fffu) ? acc[nt][0] : 0.f, v1 = (fm[nt][0] & 0x7fff0000u) ? acc[nt][1] : 0.f;
            const float v2 = (fm[nt][1] & 0x7fffu) ? acc[nt][2] : 0.f, v3 = (fm[nt][1] & 0x7fff0000u) ? acc[nt][3] : 0.f;
            u32x2_t o; o[0] = pack2h(v0, v1); o[1] = pack2h(v2, v3);
            *(__attribute__((address_space(3))) u32x2_t*)(hb + li * ET_HP + n * 2) = o;
            if (li < rows) *reinterpret_cast<u32x2_t*>(c.d_f1 + (row0 + li) * 512 + n) = o;
        }
    }
    __syncthreads();
    // ---- dx = d_f1 W1: wave = 16 input features, all of K = 512
    {
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const h16x8_t b0 = *(__attribute__((address_space(3))) h16x8_t*)(hb + li * ET_HP + ks * 64 + g * 16);
            acc = MFMA_16x16x32_H(w1[ks], b0, acc, 0, 0, 0);
        }
        if (li < rows) {
            const long long o = (row0 + li) * 128 + wave * 16 + g * 4;
            if (c.dx_f32) *reinterpret_cast<f32x4*>(c.dx_f32 + o) = acc;
            else {
                const float v0 = (xm[0] & 0x7fffu) ? acc[0] : 0.f, v1 = (xm[0] & 0x7fff0000u) ? acc[1] : 0.f;
                const float v2 = (xm[1] & 0x7fffu) ? acc[2] : 0.f, v3 = (xm[1] & 0x7fff0000u) ? acc[3] : 0.f;
                u32x2_t ov; ov[0] = pack2h(v0, v1); ov[1] = pack2h(v2, v3);
                *reinterpret_cast<u32x2_t*>(c.dx_t + o) = ov;
            }
        }
    }
}
static inline void launch_enc_tail_bwd(hipStream_t st, const EncTailBwdP& p) {
    hipLaunchKernelGGL(enc_tail_bwd_kernel, dim3((p.Nf + 15) / 16, 2), dim3(512), ET_BWD_LDS, st, p);
}

}  // namespace HULC_NS
