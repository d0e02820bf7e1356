// hulc_amd/csrc/conv_wgrad.h — convolution weight gradients for NHWC bf16 activations on gfx950.
//
//   dW[co][(kh,kw,ci)] = sum_{n,oh,ow} dY[n][oh][ow][co] * X[n][oh*S+kh][ow*S+kw][ci]
//
// Both MFMA operands are reduction-strided in memory (the reduction runs over pixels, the layouts are channel-fastest),
// which is exactly what ds_read_b64_tr_b16 exists for: the RAW tiles — a band of dY rows and the band of X rows under it —
// are copied once into LDS in their natural [pixel][channel] order, and every operand fragment is produced by transposing
// LDS reads.  No im2col: the (kh,kw) taps are address offsets into the X image, so each input element is fetched from HBM
// once per band instead of KH*KW times.  The reduction unit is an 8-pixel run inside one output row (rows are padded to a
// multiple of 8 with zero dY pixels), so each 16-lane group of an MFMA (k = 32 = 4 groups x 8) can address its own run.
//
// tr-read semantics (probed on MI355X, tools/probe_trread.py): within a 16-lane group, lane i receives element (i & 3) of the
// 8-byte chunks addressed by lanes (j*4 + (i >> 2)), j = 0..3.  So lane a addresses chunk [k-row = a >> 2][cols (a & 3)*4 ..+3]
// and lane i ends up with column i of the 4 x 16 block, rows 0..3 — the K-contiguous fragment the bf16 MFMA wants.
//
// Work split: persistent workgroups (4 waves) loop over frames and row bands, keep the whole CO x (KH*KW*CI) gradient in
// accumulator registers (wave w owns a quarter of the n-tiles), and write one fp32 partial slab each at the end; the existing
// deterministic slab reduction (unpack_conv_wgrad_kernel) finishes the job.
#pragma once
#include "common.h"

namespace HULC_NS {

typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4;

typedef __attribute__((address_space(3))) char lds_char;      // 32-bit LDS pointers: half the address registers of generic ones
DEVI h16x8_t tr_read8(lds_char* p0, lds_char* p1) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    typedef short s16x8_t __attribute__((ext_vector_type(8)));
    const s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(h16x8_t, v);
}

template <int CI, int CO, int KH, int KW, int S>
struct WgradCfg {
    // LDS bytes per pixel.  A ds_read_b64_tr_b16 is served in two 32-lane halves; with the k <-> pixel assignment of the v2 kernel
    // a half reads 8 pixels (stride S) x 32 B, which tiles the 64 banks exactly iff pitch * S = 32 * odd bytes
    // (SQ_LDS_BANK_CONFLICT: 45 % of the LDS cycles with the old +16 pitch at S = 1, 0 with this one)
    static constexpr int XS = (S == 1) ? ((CI * 2 / 32) | 1) * 32 : CI * 2 + 16;
    static constexpr int DYS = ((CO * 2 / 32) | 1) * 32;
    static constexpr int CGN = CI / 16;             // channel groups per tap
    static constexpr int NT = KH * KW * CGN;        // n-tiles (16 columns of the packed K dimension each)
    static constexpr int NTW = NT / 4;              // n-tiles per wave
    static constexpr int CT = CO / 16;              // co-tiles
    static_assert(NT % 4 == 0, "n-tiles must split over 4 waves");
    static size_t lds_bytes(int R, int IW, int OW) {
        const int OWp = (OW + 7) / 8 * 8;
        const int XR = (R - 1) * S + KH;
        return (size_t)(XR * IW + 8 * S + KW) * XS + (size_t)(R * OWp + 8) * DYS;
    }
};

template <int CI, int CO, int KH, int KW, int S>
__global__ void __launch_bounds__(256, 2) conv_wgrad_tr_kernel(const h16_t* __restrict__ X, const h16_t* __restrict__ dY, float* __restrict__ part,
                                                            float* __restrict__ bias_part, int Nf, int IH, int IW, int OH, int OW, int R) {
    using C = WgradCfg<CI, CO, KH, KW, S>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OWp = (OW + 7) & ~7, U = OWp >> 3;
    const int XR = (R - 1) * S + KH;
    const int xpix = XR * IW + 8 * S + KW;          // incl. zero tail read by the padded pixels of the last row
    const int dypix = R * OWp + 8;                  // last 8 pixels: permanent zeros (target of idle lane groups)
    lds_char* ximg = (lds_char*)smem;
    lds_char* dyimg = ximg + xpix * C::XS;
    // zero everything once: pads must be finite (0 * NaN would poison the sum) and dY pads must be 0
    for (int i = tid * 16; i < xpix * C::XS + dypix * C::DYS; i += 256 * 16) *(lds_u32x4*)((lds_char*)smem + i) = u32x4_t{0u, 0u, 0u, 0u};
    __syncthreads();

    f32x4 acc[C::NTW][C::CT];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
        for (int c = 0; c < C::CT; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, a = lane & 15;
    const int prow = a >> 2;                         // pixel (k-row) this lane addresses inside an 8-pixel run: prow, prow + 4
    const int ccol = (a & 3) * 8;                    // byte offset of this lane's 4-channel chunk inside a 16-channel group
    const int nt0 = wave * C::NTW;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // bias gradient: this lane's 8 channels (chunk lane % CH) of every dY pixel it stages

    for (int f = blockIdx.x; f < Nf; f += gridDim.x) {
        for (int oh0 = 0; oh0 < OH; oh0 += R) {
            __syncthreads();                         // previous band fully consumed
            // ---- stage dY band [R][OWp][CO] (rows outside the frame -> zeros): one band row per wave pass, no per-chunk division
            {
                constexpr int CH = CO / 8;           // 16-byte chunks per pixel
                for (int r = wave; r < R; r += 4) {
                    const bool in = oh0 + r < OH;
                    const h16_t* src = dY + ((long long)f * OH + min(oh0 + r, OH - 1)) * OW * CO;
                    for (int i = lane; i < OW * CH; i += 64) {
                        u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + i * 8);
                        if (!in) v = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { bsum[2 * e] += h2f_lo(v[e]); bsum[2 * e + 1] += h2f_hi(v[e]); }
                        *(lds_u32x4*)(dyimg + (r * OWp + i / CH) * C::DYS + (i % CH) * 16) = v;
                    }
                }
            }
            // ---- stage X band [XR][IW][CI] (rows below the frame keep stale finite data: they only meet zero dY rows)
            {
                constexpr int CH = CI / 8;
                const int ih0 = oh0 * S;
                const int rows = min(XR, IH - ih0);
                const int total = rows * IW * CH;
                const h16_t* src = X + ((long long)f * IH + ih0) * IW * CI;
                for (int i0 = tid; i0 < total; i0 += 256 * 8) {          // 8 independent 16-byte loads in flight per thread
                    u32x4_t v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {                           // unconditional (clamped) loads: no branch, no per-load wait
                        const int i = min(i0 + u * 256, total - 1);
                        v[u] = *reinterpret_cast<const u32x4_t*>(src + (long long)(i / CH) * CI + (i % CH) * 8);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + u * 256;
                        if (i < total) *(lds_u32x4*)(ximg + (i / CH) * C::XS + (i % CH) * 16) = v[u];
                    }
                }
            }
            __syncthreads();
            // ---- MFMA over the band: 4 eight-pixel runs (one per lane group) per step
            const int units = R * U;
            for (int u0 = 0; u0 < units; u0 += 4) {
                const int u = u0 + g;
                const bool valid = u < units;
                const int r = valid ? u / U : 0, ow0 = valid ? (u % U) * 8 : 0;
                const int pixA = valid ? r * OWp + ow0 : R * OWp;      // idle groups read the permanent zero pixels
                lds_char* abase = dyimg + (pixA + prow) * C::DYS + ccol;
                h16x8_t af[C::CT];
#pragma unroll
                for (int c = 0; c < C::CT; ++c) af[c] = tr_read8(abase + c * 32, abase + 4 * C::DYS + c * 32);
                const int pixB0 = (r * S) * IW + ow0 * S;
#pragma unroll
                for (int j = 0; j < C::NTW; ++j) {
                    const int nt = nt0 + j;
                    const int tap = nt / C::CGN, cg = nt % C::CGN;
                    const int kh = tap / KW, kw = tap % KW;
                    lds_char* bbase = ximg + (pixB0 + kh * IW + kw + prow * S) * C::XS + cg * 32 + ccol;
                    const h16x8_t bf = tr_read8(bbase, bbase + 4 * S * C::XS);
#pragma unroll
                    for (int c = 0; c < C::CT; ++c) acc[j][c] = MFMA_16x16x32_H(af[c], bf, acc[j][c], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial slab: part[block][co][nt*16 + n], D layout: row (co) = (lane>>4)*4 + r, col (n) = lane & 15
    constexpr int KC = KH * KW * CI;
    float* out = part + (long long)blockIdx.x * CO * KC;
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
        for (int c = 0; c < C::CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long long)(c * 16 + g * 4 + r) * KC + (nt0 + j) * 16 + a] = acc[j][c][r];
    // ---- bias-gradient slab: channel c = 8*(tid % CH) + e is spread over the threads with the same tid % CH
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < CO) {
        constexpr int CH = CO / 8;
        const int cgrp = tid >> 3, e = tid & 7;
        float s = 0.f;
        for (int t = cgrp; t < 256; t += CH) s += red[t * 8 + e];
        unsafeAtomicAdd(bias_part + tid, s);          // bias_part = the bias gradient itself (<= gridDim.x adds per channel)
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v2 of the kernel above: ONE 8-wave workgroup per CU instead of two 4-wave ones.
//   * wave = (n-quarter q, co-half h): 2 co-tiles x NT/4 n-tiles of accumulators (72 VGPRs at conv3 instead of 144), which
//     leaves room for
//   * a register prefetch of the NEXT band (dY rows + X rows, 16 x 16 B per thread) that is in flight during the MFMAs of the
//     current one — the two-workgroup version had nothing in flight while it multiplied and nothing multiplying while it staged;
//   * whole-frame bands where the frame fits (conv3: 152 KB of LDS), balanced bands otherwise (no 1-row tail band).
// Chunk k of a thread is (dY or X, LDS offset, global offset) packed in one register; loads are unconditional (clamped).
// ---------------------------------------------------------------------------------------------------------------------
template <int CI, int CO, int KH, int KW, int S, int NWV, int D = 3, int IWC = 0>      // IWC: compile-time image width (0: run time), see conv_wgrad_dma_kernel
__global__ void __launch_bounds__(NWV * 64) conv_wgrad_tr8_kernel(const h16_t* __restrict__ X, const h16_t* __restrict__ dY, float* __restrict__ part,
                                                            float* __restrict__ bias_part, int Nf, int IH, int IW, int OH, int OW, int R, int nbands, int dbg,
                                                            int* __restrict__ work_ctr, int FPB) {
    // FPB > 1 (small frames, nbands == 1): FPB frames are stacked to one band.  X rows of consecutive frames are contiguous in memory; dY is staged
    // with a pitch of VP = IH / S rows per frame, its OH real rows followed by rows that stay zero from the initial LDS fill, so that dY row v still
    // sits over X row v * S and the pixel runs that cross into the next frame multiply zeros.  R = (FPB - 1) * VP + OH virtual rows.
    using C = WgradCfg<CI, CO, KH, KW, S>;
    constexpr int NTH = NWV * 64, PF = 8192 / NTH, CTH = C::CT / (NWV / 4), CHY = CO / 8, CHX = CI / 8;   // NWV = 8 or 16 waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OWp = (OW + 7) & ~7, U = OWp >> 3;
    const int XR = (R - 1) * S + KH;
    const int xpix = XR * IW + 8 * S + KW;
    const int dypix = R * OWp + 8;
    lds_char* ximg = (lds_char*)smem;
    lds_char* dyimg = ximg + xpix * C::XS;
    for (int i = tid * 16; i < xpix * C::XS + dypix * C::DYS; i += NTH * 16) *(lds_u32x4*)((lds_char*)smem + i) = u32x4_t{0u, 0u, 0u, 0u};

    // ---- this thread's chunks: index ci = tid + k*512 over [dY band chunks | X band chunks]
    const bool stacked = FPB > 1;
    const int VP = IH / S;
    const int ndy = (stacked ? FPB * OH : R) * OW * CHY, nx = XR * IW * CHX, nch = ndy + nx;
    unsigned pk[PF];                               // bit 31: X chunk; bits 17..30: LDS offset / 16; bits 0..16: global element offset / 8
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int ci = min(tid + k * NTH, nch - 1);
        if (ci < ndy) {
            const int pix = ci / CHY, c = ci % CHY;
            int r = pix / OW;
            const int ow = pix % OW;
            if (stacked) r = (r / OH) * VP + r % OH;      // memory row (frame, row) -> virtual row of the stacked band
            pk[k] = ((unsigned)((xpix * C::XS + (r * OWp + ow) * C::DYS + c * 16) >> 4) << 17) | (unsigned)ci;
        } else {
            const int cx = ci - ndy;
            const int pix = cx / CHX, c = cx % CHX;
            pk[k] = 0x80000000u | ((unsigned)((pix * C::XS + c * 16) >> 4) << 17) | (unsigned)cx;
        }
    }
    const int q = wave & 3, h = wave >> 2;
    f32x4 acc[C::NTW][CTH];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
        for (int c = 0; c < CTH; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, a = lane & 15;
    const int prow = a >> 2;
    const int ccol = (a & 3) * 8;
    // n-tile <-> wave as in conv_wgrad_dma_kernel: quarter q = (tap group tg, channel group cgq); tile j of a wave is tap tg * TPG + j, its LDS offset splits into a
    // wave part (added to the image pointer) and a j part that is the same for every wave (an immediate when IW is a compile-time constant)
    constexpr int NTG = 4 / C::CGN, TPG = KH * KW / NTG;
    static_assert(C::CGN <= 4 && 4 % C::CGN == 0 && (KH * KW) % NTG == 0 && TPG == C::NTW && (NTG == 1 || TPG % KW == 0), "n-tile assignment");
    const int IWk = IWC ? IWC : IW;
    const int cgq = q % C::CGN, tg = q / C::CGN;
    const int qoff = ((tg * TPG / KW) * IWk) * C::XS + cgq * 32;
    int toff[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) toff[j] = ((j / KW) * IWk + j % KW) * C::XS;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    u32x4_t pf[PF];
    const int nitems = stacked ? (Nf + FPB - 1) / FPB : Nf * nbands;
    unsigned zmask = 0, zmask_cur = 0;
    auto prefetch = [&](int item) {
        zmask = 0;
        if (dbg & 2) return;
        const int f = stacked ? item * FPB : item / nbands, oh0 = stacked ? 0 : (item % nbands) * R;
        const int nfr = stacked ? min(FPB, Nf - f) : 1;         // the last stacked band may be short: its missing frames stage as zero dY
        const h16_t* ybase = dY + ((long long)f * OH + oh0) * OW * CO;
        const h16_t* xbase = X + ((long long)f * IH + oh0 * S) * IW * CI;
        const int yrows = stacked ? nfr * OH : min(R, OH - oh0), xrows = stacked ? min(XR, nfr * IH) : min(XR, IH - oh0 * S);
        const int ylim = yrows * OW * CHY - 1, xlim = xrows * IW * CHX - 1;   // last in-frame chunk
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const bool isx = (pk[k] >> 31) != 0;
            const int go = (int)(pk[k] & 0x1ffffu);
            const h16_t* src = isx ? xbase + (long long)min(go, xlim) * 8 : ybase + (long long)min(go, ylim) * 8;
            pf[k] = *reinterpret_cast<const u32x4_t*>(src);           // value untouched here: no wait at the issue point
            if (!isx && go > ylim) zmask |= 1u << k;                  // dY rows below the frame become zeros at the LDS write
        }
    };
    __shared__ int s_next[2];
    int item = blockIdx.x, iter = 0;
    if (item < nitems) prefetch(item);
    while (item < nitems) {
        if (work_ctr && tid == 0) s_next[iter & 1] = (int)gridDim.x + atomicAdd(work_ctr, 1);   // dynamic claim (see ConvTileP::work_ctr)
        __syncthreads();                               // previous band fully consumed (first pass: zero fill visible)
        zmask_cur = zmask;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            if (tid + k * NTH < nch) {
                const u32x4_t v = ((zmask_cur >> k) & 1u) ? u32x4_t{0u, 0u, 0u, 0u} : pf[k];
                if (!(pk[k] >> 31)) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bsum[2 * e] += h2f_lo(v[e]); bsum[2 * e + 1] += h2f_hi(v[e]); }
                }
                *(lds_u32x4*)((lds_char*)smem + (((pk[k] >> 17) & 0x3fffu) << 4)) = v;
            }
        }
        __syncthreads();
        item = work_ctr ? s_next[iter & 1] : item + (int)gridDim.x;
        ++iter;
        if (item < nitems) prefetch(item);             // in flight during the MFMAs below
        const int units = (dbg & 1) ? 0 : R * U;
        // k <-> pixel assignment of a 32-pixel step (4 runs of 8): k = g*8 + e; e < 4 (first tr-read): run g>>1, pixel (g&1)*4 + e;
        // e >= 4 (second tr-read): run 2 + (g>>1), pixel (g&1)*4 + e - 4.  A 32-lane half of one read then covers 8 consecutive
        // pixels of ONE run -> conflict-free with the pitches of WgradCfg.
        const int px = (g & 1) * 4 + prow;
        // run u = u0 + hh * 2 + (g >> 1) of the band = (row r, 8-pixel run uo of the row).  Its dY pixel is 8 u (OWp = 8 U); its X pixel offset follows
        // (r, uo) by adds: + step4 per step, + wrapd whenever uo passes U.  (A runtime u / U and u % U per run and step were ~100 VALU instructions
        // next to the step's 18 MFMAs: conv2 / conv3 weight gradients 372 -> 340 us per step.)
        const int q4 = 4 / U, r4 = 4 - q4 * U;
        const int step4 = (q4 * S * IW + r4 * 8 * S) * C::XS, wrapd = (S * IW - U * 8 * S) * C::XS;
        int uo[2], pbv[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int u = hh * 2 + (g >> 1), r = u / U;
            uo[hh] = u - r * U;
            pbv[hh] = ((r * S) * IW + (uo[hh] * 8 + px) * S) * C::XS + ccol;
        }
        const int abase = px * C::DYS + ccol + h * CTH * 32, pb_idle = (px * S) * C::XS + ccol;
#pragma unroll 1
        for (int u0 = 0; u0 < units; u0 += 4) {
            lds_char* ab[2];
            int pb[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int u = u0 + hh * 2 + (g >> 1);
                const bool valid = u < units;
                ab[hh] = dyimg + (valid ? u * 8 : R * OWp) * C::DYS + abase;      // idle runs read the permanent zero pixels
                pb[hh] = valid ? pbv[hh] : pb_idle;
                pbv[hh] += step4; uo[hh] += r4;
                if (uo[hh] >= U) { uo[hh] -= U; pbv[hh] += wrapd; }
            }
            auto bfrag = [&](int j) { return tr_read8(ximg + qoff + pb[0] + toff[j], ximg + qoff + pb[1] + toff[j]); };
            // B fragments run D n-tiles ahead of their MFMAs; A fragments first
            h16x8_t ring[D];
            h16x8_t af[CTH];
#pragma unroll
            for (int c = 0; c < CTH; ++c) af[c] = tr_read8(ab[0] + c * 32, ab[1] + c * 32);
#pragma unroll
            for (int j = 0; j < D && j < C::NTW; ++j) ring[j] = bfrag(j);
            __builtin_amdgcn_sched_barrier(0);          // keep the reads this far ahead: the scheduler otherwise sinks them next to their use
#pragma unroll
            for (int j = 0; j < C::NTW; ++j) {
                const h16x8_t bf = ring[j % D];
#pragma unroll
                for (int c = 0; c < CTH; ++c) acc[j][c] = MFMA_16x16x32_H(af[c], bf, acc[j][c], 0, 0, 0);
                if (j + D < C::NTW) ring[j % D] = bfrag(j + D);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    constexpr int KC = KH * KW * CI;
    float* out = part + (long long)blockIdx.x * CO * KC;
    if (!(dbg & 4) || acc[0][0][0] == 12345.678f)
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
        for (int c = 0; c < CTH; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long long)((h * CTH + c) * 16 + g * 4 + r) * KC + ((tg * TPG + j) * C::CGN + cgq) * 16 + a] = acc[j][c][r];
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < CO) {
        const int cgrp = tid >> 3, e = tid & 7;
        float s = 0.f;
        for (int t = cgrp; t < NTH; t += CHY) s += red[t * 8 + e];
        unsafeAtomicAdd(bias_part + tid, s);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// v3 (round 4): the v2 multiply loop fed by LDS-DMA into TWO band buffers.
//   v2 stages a band through 16 prefetch registers per thread and a commit phase (ds_write_b128 + the bias sums), behind two barriers per band;
//   measured on 2048 static-camera frames (experiment build, rocprofv3): whole kernel 129 / 137 us (conv3 / conv2), without the MFMA loop 71 / 96,
//   without the global loads 101 / 100, with neither 37 / 42 — per band: 2.5 us of commit + barriers, ~3.5 us of load latency that the MFMA loop of
//   ONE resident band cannot cover (the next band's registers are only free after the commit), ~8 us of multiply loop.
//   Here a band = X rows + dY rows of R output rows as ONE contiguous LDS image (same pitches, same tr-read fragments as v2), filled by
//   global_load_lds_dwordx4 in 1 KB pieces (8 waves x 64 lanes x 16 B; a lane decodes its 16-byte slot -> (pixel, chunk) -> global address; pad
//   slots, dY columns >= OW, rows beyond the frame and the trailing zero pixels come from a zero page), the NEXT band's DMA is issued right after the
//   band barrier into the other buffer and lands under the multiply loop; no staging registers (64 VGPRs back), no commit, ONE barrier per band.  The
//   bias gradient (column sums of dY) is read back from LDS (3 x 16 B per thread and band).  Bands are smaller (two must fit: R = 7 rows for
//   conv3, 6 for conv2), which costs 6 - 11 % more multiply steps (partially filled last step of a band).
// Static-camera shapes only; small frames keep v2's stacked bands.
// ---------------------------------------------------------------------------------------------------------------------
DEVI int wg_fdiv(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
// One LDS-DMA piece (64 lanes x 16 B -> 1 KB at the wave-uniform LDS address `dst`), issued as inline assembly.  Reason: the tr-read builtin
// (ds_read_b64_tr_b16) carries no memory operand, so LLVM's waitcnt pass must assume it may read what a pending
// __builtin_amdgcn_global_load_lds is still writing and puts `s_waitcnt vmcnt(0)` in front of the first tr-read after the DMA — the multiply
// loop then waits for the whole next band (measured: whole = DMA + MFMA exactly).  Hidden from the pass, the DMA is ordered by this kernel's
// own `s_waitcnt vmcnt(0)` + barrier only.  (Compiler-inserted vmcnt waits stay safe: unknown extra operations can only make them wait longer.)
DEVI void wg_lds_dma16(const void* src, lds_char* dst) {
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
template <int CI, int CO, int KH, int KW, int S, int IWC = 0>      // IWC: the image width at compile time (0: run time) — the tap offsets of the B fragments become ds_read immediates
__global__ void __launch_bounds__(512) conv_wgrad_dma_kernel(const h16_t* __restrict__ X, const h16_t* __restrict__ dY, float* __restrict__ part,
                                                             float* __restrict__ bias_part, const h16_t* __restrict__ zeros, int Nf, int IH, int IW, int OH, int OW,
                                                             int R, int nbands, int dbg, int* __restrict__ work_ctr) {
    using C = WgradCfg<CI, CO, KH, KW, S>;
    constexpr int NWV = 8, NTH = NWV * 64, CTH = C::CT / (NWV / 4), CHY = CO / 8, CHX = CI / 8, XSS = C::XS / 16, YSS = C::DYS / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OWp = (OW + 7) & ~7, U = OWp >> 3;
    const int XR = (R - 1) * S + KH;
    const int xpix = XR * IW + 8 * S + KW;
    const int dypix = R * OWp + 8;
    const int xslots = xpix * XSS, yslots = dypix * YSS;
    const int bbytes = ((xslots + yslots) * 16 + 1023) & ~1023;      // one band buffer: [X image | dY image], whole 1 KB DMA pieces
    const int npieces = bbytes >> 10;
    lds_char* const lbase = (lds_char*)smem;
    const int nitems = Nf * nbands;
    const float invXSS = 1.f / (float)XSS, invYSS = 1.f / (float)YSS, invIW = 1.f / (float)IW, invOWp = 1.f / (float)OWp, invOW = 1.f / (float)OW;
    // a lane's slot of piece k is the same (pixel, chunk) in every band: decoded ONCE into pk[k] — bit 31: X image, bit 30: a real chunk of a
    // pixel that can lie inside the frame, bits 24..29: its band row, bits 0..23: element offset / 8 from the band's first row — so that issuing a
    // band costs a handful of VALU operations per piece (decoding per band made the issue phase 1.6 us of a 4.3 us band)
    constexpr int PFM = 10;
    unsigned pk[PFM];
#pragma unroll
    for (int k = 0; k < PFM; ++k) {
        int q = (k * NWV + wave) * 64 + lane;
        unsigned v = 0;
        if (q < xslots) {
            const int px = wg_fdiv(q, invXSS), ch = q - px * XSS;
            const int r = wg_fdiv(px, invIW), c = px - r * IW;
            if (ch < CHX && r < XR) v = 0xC0000000u | ((unsigned)r << 24) | (unsigned)(((r * IW + c) * CI + ch * 8) >> 3);
        } else {
            q -= xslots;
            const int px = wg_fdiv(q, invYSS), ch = q - px * YSS;
            const int r = wg_fdiv(px, invOWp), c = px - r * OWp;
            if (ch < CHY && r < R && c < OW) v = 0x40000000u | ((unsigned)r << 24) | (unsigned)(((r * OW + c) * CO + ch * 8) >> 3);
        }
        pk[k] = v;
    }
    auto dma = [&](int item, int buf) {
        if (dbg & 2) return;
        const int f = item / nbands, oh0 = (item - f * nbands) * R, ih0 = oh0 * S;
        const int yrows = min(R, OH - oh0), xrows = min(XR, IH - ih0);
        const h16_t* xsrc = X + ((long long)f * IH + ih0) * IW * CI;
        const h16_t* ysrc = dY + ((long long)f * OH + oh0) * OW * CO;
        lds_char* dst = lbase + buf * bbytes + wave * 1024;
#pragma unroll
        for (int k = 0; k < PFM; ++k) {
            if (k * NWV + wave >= npieces) break;                     // wave-uniform
            const unsigned v = pk[k];
            const bool isx = (v >> 31) != 0;
            const int r = (int)((v >> 24) & 63u);
            const bool ok = ((v >> 30) & 1u) && r < (isx ? xrows : yrows);
            const h16_t* src = ok ? (isx ? xsrc : ysrc) + (long long)(v & 0xffffffu) * 8 : zeros;
            wg_lds_dma16(src, dst + k * NWV * 1024);
        }
    };
    const int q = wave & 3, h = wave >> 2;
    f32x4 acc[C::NTW][CTH];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
        for (int c = 0; c < CTH; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, a = lane & 15;
    const int prow = a >> 2;
    const int ccol = (a & 3) * 8;
    // n-tile <-> wave (round 5): wave quarter q owns channel group q % CGN of the taps [tg * TPG, (tg + 1) * TPG), tg = q / CGN — its n-tile j is tap tg * TPG + j.
    // The LDS offset of tile j is then (wave part) + (j part): ((tg * TPG / KW) * IW) * XS + cgq * 32 goes into the band's base pointer once, and
    // ((j / KW) * IW + j % KW) * XS is the same for every wave — an immediate of the ds_read when IW is a compile-time constant (IWC): 18 VALU adds per multiply step
    // and wave gone.  (Before: tiles q * NTW .. + NTW - 1 in (tap, channel group) order: a different offset list per wave, added with VALU.)
    constexpr int NTG = 4 / C::CGN, TPG = KH * KW / NTG;
    static_assert(C::CGN <= 4 && 4 % C::CGN == 0 && (KH * KW) % NTG == 0 && TPG == C::NTW && (NTG == 1 || TPG % KW == 0), "n-tile assignment");
    const int IWk = IWC ? IWC : IW;
    const int cgq = q % C::CGN, tg = q / C::CGN;
    const int qoff = ((tg * TPG / KW) * IWk) * C::XS + cgq * 32;          // this wave's part of every B-fragment offset
    int toff[C::NTW];
#pragma unroll
    for (int j = 0; j < C::NTW; ++j) toff[j] = ((j / KW) * IWk + j % KW) * C::XS;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    __shared__ int s_next[2];
    int item = blockIdx.x, iter = 0, nb = 0;
    if (item < nitems) dma(item, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int px = (g & 1) * 4 + prow;
    const int q4 = 4 / U, r4 = 4 - q4 * U;
    const int step4 = (q4 * S * IW + r4 * 8 * S) * C::XS, wrapd = (S * IW - U * 8 * S) * C::XS;
    while (item < nitems) {
        if (work_ctr && tid == 0) s_next[iter & 1] = (int)gridDim.x + atomicAdd(work_ctr, 1);   // dynamic claim (see ConvTileP::work_ctr)
        __syncthreads();                               // this band has landed (every wave waited for its own pieces) and nobody reads the other buffer any more
        const int cur = item;
        item = work_ctr ? s_next[iter & 1] : item + (int)gridDim.x;
        ++iter;
        lds_char* const ximg0 = lbase + nb * bbytes;
        lds_char* const dyimg = ximg0 + xslots * 16;
        lds_char* const ximg = ximg0 + qoff;              // B fragments only
        nb ^= 1;
        // the next band streams in under the bias sums and the MFMAs below.  Waves 0-3 issue their pieces now, waves 4-7 (the second wave of each
        // SIMD) after their first multiply step, so that one wave per SIMD multiplies while the other spends its issue time
        bool pend = item < nitems;
        if (pend && wave < 4) { dma(item, nb); pend = false; }
        {                                              // bias gradient: column sums of the band's real dY pixels, from LDS
            const int f = cur / nbands, oh0 = (cur - f * nbands) * R;
            const int nchunk = min(R, OH - oh0) * OW * CHY;
            for (int i = tid; i < nchunk; i += NTH) {  // i % CHY == tid % CHY: a thread keeps one channel group
                const int pix = i / CHY, r = wg_fdiv(pix, invOW), c = pix - r * OW;
                const u32x4_t v = *(lds_u32x4*)(dyimg + (r * OWp + c) * C::DYS + (tid % CHY) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bsum[2 * e] += h2f_lo(v[e]); bsum[2 * e + 1] += h2f_hi(v[e]); }
            }
        }
        const int units = (dbg & 1) ? 0 : R * U;
        int uo[2], pbv[2];
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int u = hh * 2 + (g >> 1), r = u / U;
            uo[hh] = u - r * U;
            pbv[hh] = ((r * S) * IW + (uo[hh] * 8 + px) * S) * C::XS + ccol;
        }
        const int abase = px * C::DYS + ccol + h * CTH * 32, pb_idle = (px * S) * C::XS + ccol;
        // (measured and not kept: all fragments of step s + 1 requested before the MFMAs of step s, two register sets — 94 vs 97 us for the multiply
        //  phase alone, 239 VGPRs; and with ONE B-fragment read per step instead of nine the phase still takes 94 us: the loop is not waiting on LDS)
        auto mstep = [&](const int u0) {
            lds_char* ab[2];
            int pb[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int u = u0 + hh * 2 + (g >> 1);
                const bool valid = u < units;
                ab[hh] = dyimg + (valid ? u * 8 : R * OWp) * C::DYS + abase;      // idle runs read the zero pixels behind the band
                pb[hh] = valid ? pbv[hh] : pb_idle;
                pbv[hh] += step4; uo[hh] += r4;
                if (uo[hh] >= U) { uo[hh] -= U; pbv[hh] += wrapd; }
            }
            auto bfrag = [&](int j) { return tr_read8(ximg + pb[0] + toff[j], ximg + pb[1] + toff[j]); };
            constexpr int D = 3;
            h16x8_t ring[D];
            h16x8_t af[CTH];
#pragma unroll
            for (int c = 0; c < CTH; ++c) af[c] = tr_read8(ab[0] + c * 32, ab[1] + c * 32);
#pragma unroll
            for (int j = 0; j < D && j < C::NTW; ++j) ring[j] = bfrag(j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < C::NTW; ++j) {
                const h16x8_t bf = ring[j % D];
#pragma unroll
                for (int c = 0; c < CTH; ++c) acc[j][c] = MFMA_16x16x32_H(af[c], bf, acc[j][c], 0, 0, 0);
                if (j + D < C::NTW) ring[j % D] = bfrag(j + D);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (units > 0) mstep(0);                       // first step peeled: the second wave of a SIMD issues its DMA pieces behind it, and the loop body stays free of that code
        if (pend) { dma(item, nb); pend = false; }
#pragma unroll 1
        for (int u0 = 4; u0 < units; u0 += 4) mstep(u0);
        if (pend) dma(item, nb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next band have landed
    }
    constexpr int KC = KH * KW * CI;
    float* out = part + (long long)blockIdx.x * CO * KC;
    if (!(dbg & 4) || acc[0][0][0] == 12345.678f)
#pragma unroll
    for (int j = 0; j < C::NTW; ++j)
#pragma unroll
        for (int c = 0; c < CTH; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long long)((h * CTH + c) * 16 + g * 4 + r) * KC + ((tg * TPG + j) * C::CGN + cgq) * 16 + a] = acc[j][c][r];
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < CO) {
        const int cgrp = tid >> 3, e = tid & 7;
        float s = 0.f;
        for (int t = cgrp; t < NTH; t += CHY) s += red[t * 8 + e];
        unsafeAtomicAdd(bias_part + tid, s);
    }
}
// band height for v3: among the heights of which two bands fit the LDS, the one with the fewest multiply steps per frame — a band of r rows is
// ceil(r U / 4) steps of four 8-pixel runs (U = OWp / 8 runs per row; a partially filled last step multiplies zeros) plus about one step's worth of
// barrier / issue overhead.  conv3 (21 rows, U = 3): 8 + 8 + 5 rows = 6 + 6 + 4 steps; conv2 (23 rows): 6 + 6 + 6 + 5.  0 = shape not covered
template <int CI, int CO, int KH, int KW, int S>
static inline int conv_wgrad_dma_rows(int IH, int IW, int OH, int OW, int* nbands_out, size_t* lds_out) {
    using C = WgradCfg<CI, CO, KH, KW, S>;
    const int OWp = (OW + 7) / 8 * 8, U = OWp / 8;
    int bestR = 0, bestc = 1 << 30;
    for (int R = OH; R >= 1; --R) {
        const int XR = (R - 1) * S + KH;
        const size_t band = (((size_t)(XR * IW + 8 * S + KW) * C::XS + (size_t)(R * OWp + 8) * C::DYS) + 1023) / 1024 * 1024;
        const size_t lds = std::max<size_t>(2 * band, 512 * 8 * sizeof(float));
        if (lds > 160 * 1024 - 128 || band > 80 * 1024 || R > 63) continue;      // <= 10 pieces per wave (pk[] in registers), 6 bits of band row in pk
        const int nb = (OH + R - 1) / R, last = OH - (nb - 1) * R;
        const int cost = (nb - 1) * ((R * U + 3) / 4 + 1) + (last * U + 3) / 4 + 1;
        if (cost < bestc) { bestc = cost; bestR = R; *nbands_out = nb; *lds_out = lds; }
    }
    return bestR;
}

template <int CI, int CO, int KH, int KW, int S>
static inline int launch_conv_wgrad_tr(hipStream_t st, const h16_t* X, const h16_t* dY, float* part, float* bias_part, int Nf, int IH, int IW, int OH,
                                       int OW, int max_blocks, int* work_ctr = nullptr, const h16_t* zeros = nullptr) {
    using C = WgradCfg<CI, CO, KH, KW, S>;
    static const bool v1 = HULC_SWITCH("HULC_WGRAD_V1", 0) != 0;
    static const bool v3 = HULC_SWITCH("HULC_WGRAD_DMA", 1) != 0;
    // v3 (LDS-DMA, two band buffers): frames large enough that v2 would not stack them (>= 2 bands of multiply work per frame)
    if (!v1 && v3 && zeros && OH * OW >= 256 && Nf >= 2) {
        int nb = 0; size_t lds = 0;
        const int R = conv_wgrad_dma_rows<CI, CO, KH, KW, S>(IH, IW, OH, OW, &nb, &lds);
        if (R > 0 && (long long)Nf * IH * IW * CI < (1ll << 31)) {
            constexpr int IWS = (CI == 64 && KH == 3) ? 23 : ((CI == 32 && KH == 4) ? 49 : 0);      // the static camera's maps (200 x 200 frames): conv3 reads 23 x 23, conv2 49 x 49
            static bool attr3 = false;
            if (!attr3) {
                hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<CI, CO, KH, KW, S, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
                hipFuncSetAttribute((const void*)conv_wgrad_dma_kernel<CI, CO, KH, KW, S, IWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
                attr3 = true;
            }
            const int grid = std::min(std::min(Nf * nb, 256), max_blocks);
            static const int dbg3 = HULC_SWITCH("HULC_WGRAD_DBG", 0);
            if (IWS && IW == IWS) hipLaunchKernelGGL((conv_wgrad_dma_kernel<CI, CO, KH, KW, S, IWS>), dim3(grid), dim3(512), lds, st, X, dY, part, bias_part, zeros, Nf, IH, IW, OH, OW, R, nb, dbg3, work_ctr);
            else hipLaunchKernelGGL((conv_wgrad_dma_kernel<CI, CO, KH, KW, S, 0>), dim3(grid), dim3(512), lds, st, X, dY, part, bias_part, zeros, Nf, IH, IW, OH, OW, R, nb, dbg3, work_ctr);
            return grid;
        }
    }
    constexpr int NWV = 8;                                   // 16 waves (one co-tile each, 4 waves per SIMD) measured slower: 0.43 vs 0.40 ms/step
    if (!v1) {
        // fewest balanced bands whose chunks fit the 16 x 512 prefetch slots and whose images fit in 160 KB (16 KB kept for the bias reduction)
        for (int nb = 1; nb <= OH; ++nb) {
            int R = (OH + nb - 1) / nb, XR = (R - 1) * S + KH;
            long long chunks = (long long)R * OW * (CO / 8) + (long long)XR * IW * (CI / 8);
            size_t lds = std::max<size_t>(C::lds_bytes(R, IW, OW), NWV * 64 * 8 * sizeof(float));
            if (chunks > 16 * 512 || lds > 160 * 1024 - 64 || (long long)XR * IW * CI >= (1 << 20)) continue;
            // small frames (whole frame per band): stack FPB frames to a band — the two barriers, the staging and the exposed load latency of a band
            // (~5 us, against ~0.5 us of MFMAs for a 7x7 gripper map) are then paid once per FPB frames.  Largest FPB that fits LDS and the prefetch
            // slots and still leaves every workgroup two bands (the second one's loads fly under the first one's MFMAs).
            int fpb = 1;
            static const int fpb_env = HULC_SWITCH("HULC_WG_FPB", -1);     // A/B: 1 = off, n = at most n
            if (nb == 1 && IH % S == 0 && fpb_env != 1) {
                for (int f = 2; f <= 16; ++f) {
                    const int Rf = (f - 1) * (IH / S) + OH, XRf = (Rf - 1) * S + KH;
                    const long long ch = (long long)f * OH * OW * (CO / 8) + (long long)XRf * IW * (CI / 8);
                    const size_t l = std::max<size_t>(C::lds_bytes(Rf, IW, OW), NWV * 64 * 8 * sizeof(float));
                    if (ch > 16 * 512 || l > 160 * 1024 - 64 || (long long)XRf * IW * CI >= (1 << 20)) break;
                    if (fpb_env > 1 ? f > fpb_env : (Nf + f - 1) / f < 2 * std::min(256, max_blocks)) break;
                    fpb = f; R = Rf; XR = XRf; chunks = ch; lds = l;
                }
            }
            static bool attr8 = false;
            if (!attr8) {
                hipFuncSetAttribute((const void*)conv_wgrad_tr8_kernel<CI, CO, KH, KW, S, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
                attr8 = true;
            }
            const int items = fpb > 1 ? (Nf + fpb - 1) / fpb : Nf * nb;
            const int grid = std::min(std::min(items, 256), max_blocks);
            static const int dbg = HULC_SWITCH("HULC_WGRAD_DBG", 0);   // bench ablation only
#ifdef HULC_AB_SWITCHES
            static const int dsel = HULC_SWITCH("HULC_WGRAD_D", 3);      // experiment: depth of the B-fragment ring
            if (dsel == 6 || dsel == 8) {
                static bool a2 = false;
                if (!a2) { hipFuncSetAttribute((const void*)conv_wgrad_tr8_kernel<CI, CO, KH, KW, S, NWV, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
                           hipFuncSetAttribute((const void*)conv_wgrad_tr8_kernel<CI, CO, KH, KW, S, NWV, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); a2 = true; }
                if (dsel == 6) hipLaunchKernelGGL((conv_wgrad_tr8_kernel<CI, CO, KH, KW, S, NWV, 6>), dim3(grid), dim3(NWV * 64), lds, st, X, dY, part, bias_part, Nf, IH, IW, OH, OW, R, nb, dbg, work_ctr, fpb);
                else hipLaunchKernelGGL((conv_wgrad_tr8_kernel<CI, CO, KH, KW, S, NWV, 8>), dim3(grid), dim3(NWV * 64), lds, st, X, dY, part, bias_part, Nf, IH, IW, OH, OW, R, nb, dbg, work_ctr, fpb);
                return grid;
            }
#endif
            {      // the gripper camera's maps (84 x 84 frames): conv3 reads 9 x 9, conv2 20 x 20 — the compile-time-width instance (tap offsets as ds_read immediates)
                constexpr int IWG = (CI == 64 && KH == 3) ? 9 : ((CI == 32 && KH == 4) ? 20 : 0);
                if (IWG && IW == IWG) {
                    static bool ag = false;
                    if (!ag) { hipFuncSetAttribute((const void*)conv_wgrad_tr8_kernel<CI, CO, KH, KW, S, NWV, 3, IWG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64); ag = true; }
                    hipLaunchKernelGGL((conv_wgrad_tr8_kernel<CI, CO, KH, KW, S, NWV, 3, IWG>), dim3(grid), dim3(NWV * 64), lds, st, X, dY, part, bias_part, Nf, IH, IW, OH, OW, R, nb, dbg, work_ctr, fpb);
                    return grid;
                }
            }
            hipLaunchKernelGGL((conv_wgrad_tr8_kernel<CI, CO, KH, KW, S, NWV>), dim3(grid), dim3(NWV * 64), lds, st, X, dY, part, bias_part, Nf, IH, IW, OH, OW, R, nb, dbg, work_ctr, fpb);
            return grid;
        }
    }
    int R = OH;                                              // largest band that keeps two workgroups per CU (<= 78 KB)
    while (R > 1 && C::lds_bytes(R, IW, OW) > 78 * 1024) --R;
    const int nbands = (OH + R - 1) / R;
    R = (OH + nbands - 1) / nbands;                          // balanced (no short tail band)
    const size_t lds = C::lds_bytes(R, IW, OW);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_wgrad_tr_kernel<CI, CO, KH, KW, S>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = Nf < max_blocks ? Nf : max_blocks;
    hipLaunchKernelGGL((conv_wgrad_tr_kernel<CI, CO, KH, KW, S>), dim3(grid), dim3(256), lds, st, X, dY, part, bias_part, Nf, IH, IW, OH, OW, R);
    return grid;                                             // = number of partial slabs written
}

// ---------------------------------------------------------------------------------------------------------------------
// conv1 input band staging, shared by conv1_fwd_kernel and conv1_wgrad_tr_kernel: rows [ih0, ih0+rows) of frame f as a bf16
// [c][row][iw] LDS image.  Two boundary formats:
//   * fp32 NCHW frames already transformed by the reference's dataloader (hulc.py:395-414) — the reference boundary;
//   * uint8 (.., H, W, C) frames straight from the dataset: ScaleImageTensor (x/255), Normalize(0.5, 0.5)
//     and RandomShiftsAug (hulc/utils/transforms.py:8-29: replicate-pad by `pad`, one integer shift per frame; the bilinear
//     grid_sample lands on pixel centres, i.e. out[y][x] = in[clamp(y + sy - pad)][clamp(x + sx - pad)]) are fused into this load —
//     4x fewer input bytes and no fp32 copy of the frames ever exists.
// ---------------------------------------------------------------------------------------------------------------------
#define CONV1_RAW_MARGIN 16          // replicated edge pixels each side of a raw uint8 row (>= the largest RandomShiftsAug pad; 48 bytes)
static inline __host__ __device__ int conv1_raw_pitch16(int IW) { return (((IW + 2 * CONV1_RAW_MARGIN) * 3 + 8 + 7) & ~7) + 15 & ~15; }   // the same rounded to the LDS-DMA path's 16-byte slots
static inline __host__ __device__ int conv1_raw_pitch(int IW) { return ((IW + 2 * CONV1_RAW_MARGIN) * 3 + 8 + 7) & ~7; }   // bytes per raw row (+8: the 16-byte read window)
struct Conv1Src {
    const void* X;         // fp32 NCHW (u8 == 0) or uint8 NHWC (u8 == 1)
    const int* shift;      // u8 only: [Nf][2] = (sx, sy) in [0, 2*pad], or null (no augmentation)
    int u8, pad;
    // fold (round 5; u8 only, 16-bit engines): the affine x = u (2/255) - 1 of ScaleImageTensor + Normalize is taken OUT of the data path.
    // conv1 has no zero padding (RandomShiftsAug pads by replication), so  conv(W, x) + b = (2/255) conv(W, u) + (b - sum_k W[.,k])  and
    // dW = dY * x = (2/255) (dY * u) - db (x) 1:  the MFMA operand is the EXACT 16-bit value of the byte (0..255 are representable: one v_cvt_f32_ubyteN
    // per value instead of extract + convert + fma, and no rounding of the input at all), the scale and the folded bias move into the fp32 epilogue
    // (bias = bias_fold, computed by conv1_bias_fold_kernel from the 16-bit weights), the weight-gradient kernels scale their slab and subtract their
    // own bias partial before they write it.  fold == 0: the value x itself is staged (round 2 - 4).
    int fold;
    // frame store (round 6; u8 only): X is not this batch's (Nf,H,W,C) frames but a device-resident STORE of nstore frames (whole episodes), and the
    // batch's frame f = window f / S, step f % S lives at store index wstart[f / S] + f % S — conv1's forward and weight gradient GATHER their bands by index,
    // no (B,S,H,W,C) tensor is materialised and nothing crosses PCIe per step (include/hulc_hip.h hulc_batch::window_start).  A start outside [0, nstore - S]
    // is clamped (never an out-of-bounds read); shift[] stays indexed by the batch frame f.
    const long long* wstart = nullptr;
    int S = 1;
    long long nstore = 0;
    DEVI long long frame(int f) const {
        if (!wstart) return f;
        const int b = f / S;
        const long long s0 = min(max(wstart[b], 0ll), nstore - (long long)S);
        return s0 + (f - b * S);
    }
    DEVI long long frames_total(int Nf) const { return wstart ? nstore : (long long)Nf; }
    // frame f's RandomShiftsAug offsets (dx, dy) = shift - pad; a column shift outside the contract's [0, 2 pad] is clamped (every uint8 kernel, forward and backward alike: the
    // staged replicate margins / the interior groups of the register kernels assume |dx| <= pad); rows are clamped per row, any dy is safe
    DEVI void offsets(int f, int& dx, int& dy) const {
        dx = dy = 0;
        if (shift) { dx = min(max(shift[2 * f] - pad, -pad), pad); dy = shift[2 * f + 1] - pad; }
    }
};
#define CONV1_FOLD_SCALE (2.f / 255.f)
// bias_fold[o] = b[o] - sum_k W16[o][k] over the packed 16-bit conv1 weights [32][192] (what the MFMAs multiply); one wave per output channel
__global__ void __launch_bounds__(64) conv1_bias_fold_kernel(const h16_t* __restrict__ W, const float* __restrict__ b, float* __restrict__ out) {
    const int o = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int k = lane; k < 192; k += 64) s += h2f(W[o * 192 + k]);
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) out[o] = b[o] - s;
}
DEVI float u8_to_unit(unsigned char b) { return ((float)b / 255.f - 0.5f) / 0.5f; }    // ScaleImageTensor then Normalize(mean .5, std .5)
// element e = tid + m*256 of a [rows][n] grid, m = 0, 1, ...: (row, col) advanced incrementally — a runtime integer division per
// element (~35 VALU instructions) made the staging VALU-bound (one division per load AND per store, 3 channels, every band)
struct RowCol { int r, c; };
struct Step256 {
    int n, dq, dr;
    DEVI explicit Step256(int n_) : n(n_), dq(256 / n_), dr(256 % n_) {}
    DEVI void adv(RowCol& p) const { p.c += dr; p.r += dq; if (p.c >= n) { p.c -= n; ++p.r; } }
};
// ---- uint8 frames converted FROM REGISTERS (round 6): a 4-pixel group (12 bytes of an interleaved RGB row at a shift-dependent byte offset) is fetched as the aligned
// 16-byte WINDOW around it — conv1_window_off gives the window's byte offset in the source row, clamped so that it stays inside the row (the frame buffer's last row
// included) — and conv1_window_group turns the window into the group's three 4-value planes: window -> the 12 bytes of source pixels qp .. qp + 3 -> for output pixel k
// the source pixel clamp(c 4 + k + dx, 0, IW - 1) - qp (= k everywhere but at the row ends, where RandomShiftsAug's replicate pad repeats the edge pixel).
DEVI int conv1_window_off(int c, int dx, int IW, int RB) {
    const int qp = min(max(c * 4 + dx, 0), IW - 4);
    return min((qp * 3) & ~3, RB - 16);
}
template <bool FOLD>
DEVI void conv1_window_group(const u32x4_t& w, int c, int dx, int IW, int RB, unsigned (&lo)[3], unsigned (&hi)[3]) {
    const float sc = 2.f / 255.f, of = -1.f;
    const int p0 = c * 4 + dx, qp = min(max(p0, 0), IW - 4);
    const int a0 = min((qp * 3) & ~3, RB - 16), sh = qp * 3 - a0;      // 0 .. 4
    const bool s4 = sh == 4;
    const unsigned w0 = s4 ? w[1] : w[0], w1 = s4 ? w[2] : w[1], w2 = s4 ? w[3] : w[2], w3 = w[3];
    const unsigned d0 = __builtin_amdgcn_alignbyte(w1, w0, sh & 3), d1 = __builtin_amdgcn_alignbyte(w2, w1, sh & 3), d2 = __builtin_amdgcn_alignbyte(w3, w2, sh & 3);
    const unsigned P[4] = {d0, __builtin_amdgcn_alignbyte(d1, d0, 3), __builtin_amdgcn_alignbyte(d2, d1, 2), d2 >> 8};      // the four source pixels as 24-bit words
    float v[12];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int si = min(max(p0 + kk, 0), IW - 1) - qp;       // 0 .. 3
        const unsigned pw = si == 0 ? P[0] : (si == 1 ? P[1] : (si == 2 ? P[2] : P[3]));
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) { const float x = (float)((pw >> (8 * ch)) & 0xffu); v[kk * 3 + ch] = FOLD ? x : fmaf(x, sc, of); }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) { lo[ch] = pack2h(v[ch], v[3 + ch]); hi[ch] = pack2h(v[6 + ch], v[9 + ch]); }
}
// The same for a group that no shift can push against a row end (conv1_interior_groups): source pixel k of the group is pixel qp + k, qp = 4 c + dx, the window starts at
// (3 qp) & ~3 and the group's 12 bytes at offset sh = (3 qp) & 3 = (3 dx) & 3 of it — ONE value per frame.  3 alignbyte + 12 byte conversions + 6 packs against the ~ 80
// instructions of the clamps and per-pixel selects above (the conversion was VALU-bound: tools/conv1_wgrad_probe.hip).
template <bool FOLD>
DEVI void conv1_window_group_interior(const u32x4_t& w, int sh, unsigned (&lo)[3], unsigned (&hi)[3]) {
    const float sc = 2.f / 255.f, of = -1.f;
    const unsigned d[3] = {__builtin_amdgcn_alignbyte(w[1], w[0], sh), __builtin_amdgcn_alignbyte(w[2], w[1], sh), __builtin_amdgcn_alignbyte(w[3], w[2], sh)};
    float v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { const float x = (float)((d[i >> 2] >> (8 * (i & 3))) & 0xffu); v[i] = FOLD ? x : fmaf(x, sc, of); }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) { lo[ch] = pack2h(v[ch], v[3 + ch]); hi[ch] = pack2h(v[6 + ch], v[9 + ch]); }
}
// groups [cl, cl + wi) of a row's IW / 4 are interior for every shift |dx| <= pad: 4 c - pad >= 0 and 4 c + pad <= IW - 6 (all four pixels inside the row, the window
// not clamped at the row's last 16 bytes); at most `cap` of them, centred.  wi <= 0: no interior (tiny frames).
static inline __host__ __device__ void conv1_interior_groups(int IW, int pad, int cap, int& cl, int& wi) {
    const int lo = (pad + 3) >> 2, hi = (IW - 6 - pad) >= 0 ? (IW - 6 - pad) / 4 + 1 : 0;      // [lo, hi)
    wi = hi - lo < cap ? hi - lo : cap;
    cl = wi > 0 ? lo + (hi - lo - wi) / 2 : 0;
    if (wi < 0) wi = 0;
}
template <bool REGCONV = false>      // compile-time: the window path must not sit (as dead code with live registers) inside the 128-VGPR fp32 forward kernel — it cost that kernel 16 more spills
DEVI void conv1_stage_band(const Conv1Src& s, int f, int ih0, int rows, int IH, int IW, lds_char* ximg, int XR, int XRS, int tid, lds_char* raw) {
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const int W4 = IW >> 2;                                           // 4-pixel groups per row
    const Step256 sq(W4);
    const RowCol q0{tid / W4, tid % W4};
    if (!s.u8) {
        // all three channel planes of the band in ONE pass: element e of the [3][rows][W4] grid of 16-byte quads, NU quads in flight per thread.
        // (A pass per channel had ~3 useful loads in flight per thread for a 16-row band — its other unrolled slots re-read a clamped element —
        // and exposed the load latency three times per band.)
        const float* X = reinterpret_cast<const float*>(s.X);
        const float* src0 = X + ((long long)f * 3 * IH + ih0) * IW;
        const int per = rows * W4, tot = 3 * per;
        const float inv_per = 1.f / (float)per, inv_w4 = 1.f / (float)W4;
        constexpr int NU = 6;       // conv1_fwd_kernel<4> runs at 128 VGPRs next to 48 registers of weight fragments: 10 in flight spilled 38
        for (int e0 = tid; e0 < tot; e0 += 256 * NU) {
            float4 v[NU];
            int pk[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int e = min(e0 + u * 256, tot - 1);
                const int c = (int)(((float)e + 0.5f) * inv_per), rem = e - c * per;
                const int r = (int)(((float)rem + 0.5f) * inv_w4), q4 = rem - r * W4;
                pk[u] = (c << 20) | (r << 8) | q4;
                v[u] = *reinterpret_cast<const float4*>(src0 + ((long long)c * IH + r) * IW + q4 * 4);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (e0 + u * 256 < tot) {
                    const int c = pk[u] >> 20, r = (pk[u] >> 8) & 0xfff, q4 = pk[u] & 0xff;
                    u32x2_t o;
                    o[0] = pack2h(v[u].x, v[u].y);
                    o[1] = pack2h(v[u].z, v[u].w);
                    *(__attribute__((address_space(3))) u32x2_t*)(ximg + (c * XR + r) * XRS + q4 * 8) = o;
                }
            }
        }
        return;
    }
    // uint8: (A) the band's source rows (row clamp of the replicate pad applied) are copied with 8- or 4-byte loads into `raw`, each
    // row behind a margin of CONV1_RAW_MARGIN replicated edge pixels — byte-wide global loads cost a full wave instruction per
    // 64 bytes and made a first version as slow as the 4x larger fp32 path; (B) a 4-pixel group is 12 contiguous bytes of a raw row
    // at a (shift-dependent) unaligned offset: four aligned LDS dwords, v_alignbyte, v_cvt_f32_ubyteN, one FMA per value
    // (b * 2/255 - 1: within one fp32 ulp of the reference's (b/255 - .5)/.5, which the fp32-mode ingest_u8_kernel keeps exactly).
    const unsigned char* base = reinterpret_cast<const unsigned char*>(s.X) + s.frame(f) * IH * IW * 3;
    int dx = 0, dy = 0;
    s.offsets(f, dx, dy);
    const int RB = IW * 3;                                            // bytes per source row (multiple of 4: IW % 4 == 0)
    if constexpr (REGCONV) {
        // one pass: NU windows in flight per thread, converted from the registers into the [c][row][iw] image (no raw rows in LDS, no margins, no barrier in between)
        constexpr int NU = 6;
        // interior groups first (conv1_interior_groups: no clamp, no per-pixel select, one alignbyte shift per frame — see conv1_wgrad_tr2r_kernel), then the row ends
        int cl, WI;
        conv1_interior_groups(IW, s.pad, W4, cl, WI);
        if (WI > 0 && dx >= -s.pad && dx <= s.pad) {
            const Step256 si(WI);
            const int shu = (3 * dx) & 3, off0 = 12 * cl + 3 * dx;
            RowCol p{tid / WI, tid % WI};
            while (p.r < rows) {
                RowCol e[NU];
                u32x4_t w[NU];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    e[u] = p; si.adv(p);
                    const bool in = e[u].r < rows;
                    w[u] = *reinterpret_cast<const u32x4_t*>(base + (long long)min(max(ih0 + (in ? e[u].r : rows - 1) + dy, 0), IH - 1) * RB + ((off0 + 12 * (in ? e[u].c : 0)) & ~3));
                }
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    if (e[u].r < rows) {
                        unsigned lo[3], hi[3];
                        if (s.fold) conv1_window_group_interior<true>(w[u], shu, lo, hi); else conv1_window_group_interior<false>(w[u], shu, lo, hi);
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) *(__attribute__((address_space(3))) u32x2_t*)(ximg + (ch * XR + e[u].r) * XRS + (cl + e[u].c) * 8) = u32x2_t{lo[ch], hi[ch]};
                    }
            }
            const int WE = W4 - WI;
            for (int i = tid; i < rows * WE; i += 256) {
                const int r = i / WE, ce = i - r * WE, c = ce < cl ? ce : ce + WI;
                const u32x4_t w = *reinterpret_cast<const u32x4_t*>(base + (long long)min(max(ih0 + r + dy, 0), IH - 1) * RB + conv1_window_off(c, dx, IW, RB));
                unsigned lo[3], hi[3];
                if (s.fold) conv1_window_group<true>(w, c, dx, IW, RB, lo, hi); else conv1_window_group<false>(w, c, dx, IW, RB, lo, hi);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) *(__attribute__((address_space(3))) u32x2_t*)(ximg + (ch * XR + r) * XRS + c * 8) = u32x2_t{lo[ch], hi[ch]};
            }
            return;
        }
        RowCol p = q0;
        while (p.r < rows) {
            RowCol e[NU];
            u32x4_t w[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                e[u] = p; sq.adv(p);
                const bool in = e[u].r < rows;
                w[u] = *reinterpret_cast<const u32x4_t*>(base + (long long)min(max(ih0 + (in ? e[u].r : rows - 1) + dy, 0), IH - 1) * RB + conv1_window_off(in ? e[u].c : 0, dx, IW, RB));
            }
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (e[u].r < rows) {
                    unsigned lo[3], hi[3];
                    if (s.fold) conv1_window_group<true>(w[u], e[u].c, dx, IW, RB, lo, hi); else conv1_window_group<false>(w[u], e[u].c, dx, IW, RB, lo, hi);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) *(__attribute__((address_space(3))) u32x2_t*)(ximg + (ch * XR + e[u].r) * XRS + e[u].c * 8) = u32x2_t{lo[ch], hi[ch]};
                }
        }
        return;
    }
    const int RP = conv1_raw_pitch(IW);
    constexpr int LM = CONV1_RAW_MARGIN * 3;                          // byte offset of pixel 0 in a raw row (multiple of 8)
    if ((RB & 7) == 0 && ((uintptr_t)base & 7) == 0) {
        typedef unsigned int u32x2v __attribute__((ext_vector_type(2)));
        constexpr int NU = 10;                                        // the whole band in one round of loads per thread
        const int n8 = RB >> 3;
        const Step256 s8(n8);
        RowCol p{tid / n8, tid % n8};
        while (p.r < rows) {
            RowCol e[NU];
            u32x2v v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                e[u] = p; s8.adv(p);
                const bool in = e[u].r < rows;
                v[u] = *reinterpret_cast<const u32x2v*>(base + (long long)min(max(ih0 + (in ? e[u].r : rows - 1) + dy, 0), IH - 1) * RB + (in ? e[u].c : 0) * 8);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (e[u].r < rows) *(__attribute__((address_space(3))) u32x2v*)(raw + e[u].r * RP + LM + e[u].c * 8) = v[u];
        }
    } else {
        constexpr int NU = 8;
        const int n4 = RB >> 2;
        const Step256 s4(n4);
        RowCol p{tid / n4, tid % n4};
        while (p.r < rows) {
            RowCol e[NU];
            unsigned v[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                e[u] = p; s4.adv(p);
                const bool in = e[u].r < rows;
                v[u] = *reinterpret_cast<const unsigned*>(base + (long long)min(max(ih0 + (in ? e[u].r : rows - 1) + dy, 0), IH - 1) * RB + (in ? e[u].c : 0) * 4);
            }
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (e[u].r < rows) *(__attribute__((address_space(3))) unsigned*)(raw + e[u].r * RP + LM + e[u].c * 4) = v[u];
        }
    }
    if (dx != 0) {                                                    // replicate margins (F.pad(..., "replicate")): one thread per (row, side)
        for (int i = tid; i < rows * 2; i += 256) {
            const int rr = i >> 1, side = i & 1;
            const unsigned char* rp = base + (long long)min(max(ih0 + rr + dy, 0), IH - 1) * RB;
            const unsigned w = *reinterpret_cast<const unsigned*>(rp + (side ? RB - 4 : 0));
            const unsigned px = side ? (w >> 8) : (w & 0xffffffu);    // the edge pixel's 3 bytes
            lds_char* dst = raw + rr * RP + (side ? LM + RB : 0);
            for (int k = 0; k < CONV1_RAW_MARGIN; ++k) {
                dst[k * 3 + 0] = (char)(px & 0xff); dst[k * 3 + 1] = (char)((px >> 8) & 0xff); dst[k * 3 + 2] = (char)((px >> 16) & 0xff);
            }
        }
    }
    __syncthreads();
    const float sc = s.fold ? 1.f : 2.f / 255.f, of = s.fold ? 0.f : -1.f;      // fold: the exact value of the byte (Conv1Src::fold)
    for (RowCol p = q0; p.r < rows; sq.adv(p)) {
        const int o = p.r * RP + LM + (p.c * 4 + dx) * 3;             // |dx| <= pad <= CONV1_RAW_MARGIN: stays inside the margins
        const int sh = o & 3;
        const __attribute__((address_space(3))) unsigned* wp = (const __attribute__((address_space(3))) unsigned*)(raw + (o & ~3));
        const unsigned w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
        const unsigned d[3] = {__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh), __builtin_amdgcn_alignbyte(w3, w2, sh)};
        float v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) v[k] = fmaf((float)((d[k >> 2] >> (8 * (k & 3))) & 0xffu), sc, of);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            u32x2_t ov;
            ov[0] = pack2h(v[c], v[3 + c]);
            ov[1] = pack2h(v[6 + c], v[9 + c]);
            *(__attribute__((address_space(3))) u32x2_t*)(ximg + (c * XR + p.r) * XRS + p.c * 8) = ov;
        }
    }
}
// the same transform materialised as fp32 NCHW frames (fp32 parity mode, tests): out[f][c][y][x]
__global__ void ingest_u8_kernel(const unsigned char* __restrict__ in, const int* __restrict__ shift, int pad, int Nf, int IH, int IW, float* __restrict__ out, Conv1Src src = Conv1Src{}) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Nf * 3 * IH * IW) return;
    const int x = (int)(idx % IW), y = (int)((idx / IW) % IH), c = (int)((idx / ((long long)IW * IH)) % 3), f = (int)(idx / ((long long)3 * IW * IH));
    int dx = 0, dy = 0;
    if (shift) { dx = min(max(shift[2 * f] - pad, -pad), pad); dy = shift[2 * f + 1] - pad; }      // as Conv1Src::offsets
    const int sy = min(max(y + dy, 0), IH - 1), sx = min(max(x + dx, 0), IW - 1);
    out[idx] = u8_to_unit(in[((src.frame(f) * IH + sy) * IW + sx) * 3 + c]);       // src: only its frame-store fields are used (frame(f) = f without a store)
}

// ---------------------------------------------------------------------------------------------------------------------
// conv1 (8x8 stride 4, 3 -> 32 channels) weight gradient straight from the fp32 NCHW boundary frames.
//   dW[co][(c,kh,kw)] = sum dY[n][oh][ow][co] * X[n][c][oh*4+kh][ow*4+kw]
// The X band is converted to bf16 while it is staged ([c][row][iw] image), dY is staged as [pix][32]; an n-tile of 16 packed
// columns = (c, two kernel rows kh, 8 kw): lane chunk q covers kh = kh_lo + q/2, kw = (q&1)*4..+3 -> 4 contiguous bf16 in a row.
// ---------------------------------------------------------------------------------------------------------------------
struct Wgrad1Cfg {
    static constexpr int CO = 32, KH = 8, KW = 8, S = 4, C = 3;
    static constexpr int DYS = CO * 2 + 32;     // 96 B = 32 x odd: the 4 pixel rows x 4 chunks of a tr-read group fall on distinct bank pairs (80 B pitched pixel 3 onto pixel 0: 44 % conflict cycles, tools/pmc_sq.sh)
    static size_t lds_bytes(int R, int IW, int OW, bool u8 = false) {
        const int OWp = (OW + 7) / 8 * 8;
        const int XR = (R - 1) * S + KH;
        return (size_t)C * XR * (IW * 2 + 16) + 512 + (size_t)(R * OWp + 8) * DYS + (u8 ? (size_t)XR * conv1_raw_pitch(IW) + 16 : 0);   // + raw uint8 rows
    }
};

__global__ void __launch_bounds__(256, 2) conv1_wgrad_tr_kernel(Conv1Src X, int* __restrict__ work_ctr, const h16_t* __restrict__ dY, float* __restrict__ part,
                                                                float* __restrict__ bias_part, int Nf, int IH, int IW, int OH, int OW, int R) {
    using C = Wgrad1Cfg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OWp = (OW + 7) & ~7, U = OWp >> 3;
    const int XR = (R - 1) * C::S + C::KH;
    const int XRS = IW * 2 + 16;                                  // bytes per staged input row
    const int xbytes = C::C * XR * XRS + 512;
    const int dypix = R * OWp + 8;
    lds_char* ximg = (lds_char*)smem;
    lds_char* dyimg = ximg + xbytes;
    for (int i = tid * 16; i < xbytes + dypix * C::DYS; i += 256 * 16) *(lds_u32x4*)((lds_char*)smem + i) = u32x4_t{0u, 0u, 0u, 0u};
    __syncthreads();

    f32x4 acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, a = lane & 15;
    const int prow = a >> 2, q = a & 3;
    const int ccolA = q * 8;
    const int nt0 = wave * 3;
    const int W4 = IW >> 2;                                       // float4 per input row (IW % 4 == 0)
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    __shared__ int s_nextf[2];
    int fiter = 0;
    for (int f = blockIdx.x; f < Nf;) {
        if (work_ctr && tid == 0) s_nextf[fiter & 1] = (int)gridDim.x + atomicAdd(work_ctr, 1);   // next frame claimed dynamically (see ConvTileP::work_ctr)
        for (int oh0 = 0; oh0 < OH; oh0 += R) {
            __syncthreads();
            {   // dY band, one row per wave pass
                for (int r = wave; r < R; r += 4) {
                    const bool in = oh0 + r < OH;
                    const h16_t* src = dY + ((long long)f * OH + min(oh0 + r, OH - 1)) * OW * C::CO;
                    for (int i = lane; i < OW * 4; i += 64) {
                        u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + i * 8);
                        if (!in) v = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
                        for (int e = 0; e < 4; ++e) { bsum[2 * e] += h2f_lo(v[e]); bsum[2 * e + 1] += h2f_hi(v[e]); }
                        *(lds_u32x4*)(dyimg + (r * OWp + (i >> 2)) * C::DYS + (i & 3) * 16) = v;
                    }
                }
            }
            {   // X band -> bf16 [c][row][iw]
                const int ih0 = oh0 * C::S;
                conv1_stage_band(X, f, ih0, min(XR, IH - ih0), IH, IW, ximg, XR, XRS, tid, dyimg + dypix * C::DYS);
            }
            __syncthreads();
            const int units = R * U;
            for (int u0 = 0; u0 < units; u0 += 4) {
                const int u = u0 + g;
                const bool valid = u < units;
                const int r = valid ? u / U : 0, ow0 = valid ? (u % U) * 8 : 0;
                const int pixA = valid ? r * OWp + ow0 : R * OWp;
                lds_char* abase = dyimg + (pixA + prow) * C::DYS + ccolA;
                h16x8_t af[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) af[c] = tr_read8(abase + c * 32, abase + 4 * C::DYS + c * 32);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int nt = nt0 + j;
                    const int ch = nt >> 2, kh = (nt & 3) * 2 + (q >> 1);
                    lds_char* bbase = ximg + (ch * XR + r * C::S + kh) * XRS + ((ow0 + prow) * C::S + (q & 1) * 4) * 2;
                    const h16x8_t bf = tr_read8(bbase, bbase + 4 * C::S * 2);
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[j][c] = MFMA_16x16x32_H(af[c], bf, acc[j][c], 0, 0, 0);
                }
            }
        }
        f = work_ctr ? s_nextf[fiter & 1] : f + (int)gridDim.x;   // written >= 2 barriers ago; the other slot is the one tid 0 writes next
        ++fiter;
    }
    float* out = part + (long long)blockIdx.x * C::CO * 192;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long long)(c * 16 + g * 4 + r) * 192 + (nt0 + j) * 16 + a] = acc[j][c][r];
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < C::CO) {
        const int cgrp = tid >> 3, e = tid & 7;
        float s = 0.f;
        for (int t = cgrp; t < 256; t += 4) s += red[t * 8 + e];
        unsafeAtomicAdd(bias_part + tid, s);
    }
}

// phase ablation of the two v2 kernels below, compiled in only by tools/conv1_wgrad_probe.hip (-DHULC_W1_PROBE): bit 0 = no multiply loop, bit 1 = no prefetch of the next band,
// bit 2 = no margin fill, bit 3 = no raw -> 16-bit conversion (the last two: uint8 kernel)
#ifdef HULC_W1_PROBE
__device__ int g_w1_probe = 0;
#define W1_PROBE_SKIP(b) ((g_w1_probe & (b)) != 0)
#else
#define W1_PROBE_SKIP(b) false
#endif
// ---------------------------------------------------------------------------------------------------------------------
// v2 of the kernel above for the fp32 NCHW boundary (the headline configuration): 8 waves, 2 workgroups per CU, and the NEXT band's
// frames + dY rows prefetched into registers while the current band multiplies.
//   v1 runs 4 workgroups per CU that each stage a band (loads -> wait -> convert -> LDS) and then multiply it; the HBM stream stalls
//   whenever the resident workgroups are all converting / multiplying, the bands are 3 output rows tall (39 KB of LDS each), so every
//   band re-stages 4 of its 16 input rows (33 % over-fetch) and pays its barriers for 12 new rows.  It moved 1.53 GB in 370 us (3.6 of
//   the ~6.3 TB/s a streaming copy reaches).
//   v2: bands of 7 output rows (32 input rows: 14 % over-fetch, 7 bands per 49-row frame), 13 x 16 B per thread in flight for the whole
//   multiply phase of the previous band (106 KB per workgroup), zero rows applied at the commit, the two wave halves split the pixel
//   units and are summed through LDS at the end.
// ---------------------------------------------------------------------------------------------------------------------
template <int PFX, int PFY>
__global__ void __launch_bounds__(512, 4) conv1_wgrad_tr2_kernel(const float* __restrict__ X, int* __restrict__ work_ctr, const h16_t* __restrict__ dY, float* __restrict__ part,
                                                                 float* __restrict__ bias_part, int Nf, int IH, int IW, int OH, int OW, int R, int nbands) {
    using C = Wgrad1Cfg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OWp = (OW + 7) & ~7, U = OWp >> 3;
    const int XR = (R - 1) * C::S + C::KH;
    const int XRS = IW * 2 + 16;
    const int xbytes = C::C * XR * XRS + 512;
    const int dypix = R * OWp + 8;
    lds_char* ximg = (lds_char*)smem;
    lds_char* dyimg = ximg + xbytes;
    for (int i = tid * 16; i < xbytes + dypix * C::DYS; i += 512 * 16) *(lds_u32x4*)((lds_char*)smem + i) = u32x4_t{0u, 0u, 0u, 0u};
    const int W4 = IW >> 2;
    const int nx = C::C * XR * W4, ny = R * OW * 4;               // 16-byte chunks of a band: fp32 frame quads, dY channel quarters
    // band-invariant slot descriptors, one register each: frame slots (channel << 16 | row << 8 | float4 column), dY slots (row << 16 | chunk)
    int xd[PFX], yd[PFY];
#pragma unroll
    for (int k = 0; k < PFX; ++k) {
        const int e = min(tid + k * 512, nx - 1);
        const int c = e / (XR * W4), rem = e - c * (XR * W4), r = rem / W4, q4 = rem - r * W4;
        xd[k] = (c << 16) | (r << 8) | q4;
    }
#pragma unroll
    for (int k = 0; k < PFY; ++k) {
        const int e = min(tid + k * 512, ny - 1);
        const int r = e / (OW * 4), i = e - r * (OW * 4);
        yd[k] = (r << 16) | i;
    }
    f32x4 acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, a = lane & 15;
    const int prow = a >> 2, q = a & 3;
    const int ccolA = q * 8;
    const int nt0 = (wave & 3) * 3, uh = wave >> 2;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float4 px[PFX];
    u32x4_t py[PFY];
    int xrows = 0, yrows = 0;                                     // rows of the prefetched band inside the frame (others are zeros)
    const int nitems = Nf * nbands;
    auto prefetch = [&](int item) {
        const int f = item / nbands, oh0 = (item % nbands) * R;
        const int ih0 = oh0 * C::S;
        xrows = min(XR, IH - ih0); yrows = min(R, OH - oh0);
        const float* xb = X + ((long long)f * C::C * IH + ih0) * IW;
        const h16_t* yb = dY + ((long long)f * OH + oh0) * OW * C::CO;
#pragma unroll
        for (int k = 0; k < PFX; ++k) {
            const int c = xd[k] >> 16, r = (xd[k] >> 8) & 0xff, q4 = xd[k] & 0xff;
            px[k] = *reinterpret_cast<const float4*>(xb + (c * IH + min(r, xrows - 1)) * IW + q4 * 4);       // rows below the frame: re-read a valid row (zeroed at the commit)
        }
#pragma unroll
        for (int k = 0; k < PFY; ++k) {
            const int r = yd[k] >> 16, i = yd[k] & 0xffff;
            py[k] = *reinterpret_cast<const u32x4_t*>(yb + (min(r, yrows - 1) * OW) * C::CO + i * 8);
        }
    };
    // A workgroup walks the bands of ONE frame back to back (work unit = frame, claimed dynamically): the (KH - S) input rows two
    // consecutive bands share are then re-read by the same CU a few microseconds later and come from L2 instead of HBM (with the bands of
    // a frame dealt to different workgroups the PMC counters showed 27 % more HBM bytes than the frames hold).
    __shared__ int s_next[2];
    int frame = blockIdx.x, fiter = 0, band = 0;
    int item = frame * nbands;
    if (frame < Nf) prefetch(item);
    while (frame < Nf) {
        if (band == 0 && work_ctr && tid == 0) s_next[fiter & 1] = (int)gridDim.x + atomicAdd(work_ctr, 1);
        __syncthreads();                                          // previous band consumed (first pass: zero fill visible)
        const int cxr = xrows, cyr = yrows;
#pragma unroll
        for (int k = 0; k < PFX; ++k)
            if (tid + k * 512 < nx) {
                const int c = xd[k] >> 16, r = (xd[k] >> 8) & 0xff, q4 = xd[k] & 0xff;
                const bool in = r < cxr;
                u32x2_t o;
                o[0] = in ? pack2h(px[k].x, px[k].y) : 0u;
                o[1] = in ? pack2h(px[k].z, px[k].w) : 0u;
                *(__attribute__((address_space(3))) u32x2_t*)(ximg + (c * XR + r) * XRS + q4 * 8) = o;
            }
#pragma unroll
        for (int k = 0; k < PFY; ++k)
            if (tid + k * 512 < ny) {
                const int r = yd[k] >> 16, i = yd[k] & 0xffff;
                const u32x4_t v = r < cyr ? py[k] : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; ++e) { bsum[2 * e] += h2f_lo(v[e]); bsum[2 * e + 1] += h2f_hi(v[e]); }
                *(lds_u32x4*)(dyimg + (r * OWp + (i >> 2)) * C::DYS + (i & 3) * 16) = v;
            }
        __syncthreads();
        if (++band == nbands) {                                   // next frame (its claim was written at the frame's first band: >= 1 barrier ago)
            band = 0;
            frame = work_ctr ? s_next[fiter & 1] : frame + (int)gridDim.x;
            ++fiter;
        }
        item = frame * nbands + band;
        if (frame < Nf && !W1_PROBE_SKIP(2)) prefetch(item);      // in flight during the MFMAs below
        const int units = R * U;
        // run u = u0 + g -> (row ur, run uo of the row) advanced by adds (8 runs per step): a runtime division per step stood next to 6 MFMAs
        int joff[3];                                   // per n-tile: (channel, kernel row) of this lane's B rows + its column half
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int nt = nt0 + j; joff[j] = ((nt >> 2) * XR + (nt & 3) * 2 + (q >> 1)) * XRS + (q & 1) * 8; }
        const int q8 = 8 / U, r8 = 8 - q8 * U;
        int ur, uo;
        { const int u = uh * 4 + g; ur = u / U; uo = u - ur * U; }
#pragma unroll 1
        for (int u0 = W1_PROBE_SKIP(1) ? units : uh * 4; u0 < units; u0 += 8) {
            const int u = u0 + g;
            const bool valid = u < units;
            const int r = valid ? ur : 0, ow0 = valid ? uo * 8 : 0;
            ur += q8; uo += r8;
            if (uo >= U) { uo -= U; ++ur; }
            const int pixA = valid ? r * OWp + ow0 : R * OWp;
            lds_char* abase = dyimg + (pixA + prow) * C::DYS + ccolA;
            h16x8_t af[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) af[c] = tr_read8(abase + c * 32, abase + 4 * C::DYS + c * 32);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                lds_char* bbase = ximg + joff[j] + (r * C::S) * XRS + (ow0 + prow) * C::S * 2;
                const h16x8_t bf = tr_read8(bbase, bbase + 4 * C::S * 2);
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[j][c] = MFMA_16x16x32_H(af[c], bf, acc[j][c], 0, 0, 0);
            }
        }
    }
    // ---- the two unit halves (waves w and w + 4) are summed through LDS, then one slab per workgroup
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (uh == 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) red[((wave & 3) * 6 + j * 2 + c) * 64 + lane] = acc[j][c];
    }
    __syncthreads();
    if (uh == 0) {
        float* out = part + (long long)blockIdx.x * C::CO * 192;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 v = acc[j][c] + red[((wave & 3) * 6 + j * 2 + c) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(long long)(c * 16 + g * 4 + r) * 192 + (nt0 + j) * 16 + a] = v[r];
            }
    }
    __syncthreads();
    float* redf = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) redf[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < C::CO) {
        const int cgrp = tid >> 3, e = tid & 7;
        float sacc = 0.f;
        for (int t = cgrp; t < 512; t += 4) sacc += redf[t * 8 + e];
        unsafeAtomicAdd(bias_part + tid, sacc);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same v2 structure for the uint8 (.., H, W, C) boundary (the uint8 path had kept v1).  The next
// band is prefetched as RAW bytes — 4-byte chunks of the interleaved RGB rows (rows are multiples of 4 bytes for both cameras; 8 for the
// static one only), the row clamp of RandomShiftsAug's replicate pad applied to the source row, plus the two edge pixels of every row —
// 12 + 1 registers per thread for a quarter of the fp32 path's bytes.  Commit = raw rows + replicated margins into LDS, barrier, then the
// ScaleImageTensor / Normalize / column-shift conversion of conv1_stage_band's second stage into the bf16 [c][row][iw] image.
// ---------------------------------------------------------------------------------------------------------------------
template <int PFX, int PFY>
__global__ void __launch_bounds__(512, 4) conv1_wgrad_tr2u_kernel(Conv1Src S, int* __restrict__ work_ctr, const h16_t* __restrict__ dY, float* __restrict__ part,
                                                                  float* __restrict__ bias_part, int Nf, int IH, int IW, int OH, int OW, int R, int nbands) {
    using C = Wgrad1Cfg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OWp = (OW + 7) & ~7, U = OWp >> 3;
    const int XR = (R - 1) * C::S + C::KH;
    const int XRS = IW * 2 + 16;
    const int xbytes = C::C * XR * XRS + 512;
    const int dypix = R * OWp + 8;
    lds_char* ximg = (lds_char*)smem;
    lds_char* dyimg = ximg + xbytes;
    lds_char* raw = dyimg + dypix * C::DYS;
    for (int i = tid * 16; i < xbytes + dypix * C::DYS; i += 512 * 16) *(lds_u32x4*)((lds_char*)smem + i) = u32x4_t{0u, 0u, 0u, 0u};
    const int W4 = IW >> 2;
    const int RB = IW * 3, n4 = RB >> 2;                          // bytes / 4-byte chunks per source row
    const int RP = conv1_raw_pitch(IW);
    constexpr int LM = CONV1_RAW_MARGIN * 3;
    const int nx = XR * n4, ny = R * OW * 4;
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(S.X);
    // raw slot k of this thread = chunk tid + 512 k of the [XR][n4] grid: (row, chunk) advanced incrementally from slot 0's — twelve
    // descriptor registers next to twelve slots of in-flight data spilled 17 registers at the 128-VGPR budget and serialised the prefetch
    // ... and the compiler hoists whatever is band-invariant out of the band loop and spills it just the same (a scratch reload in front of
    // every prefetch load waits for the loads issued before it): slot 0's (row, chunk) is recomputed per band behind an opaque copy of tid
    const int xdq = 512 / n4, xdr = 512 - xdq * n4;
    int yd[PFY];
#pragma unroll
    for (int k = 0; k < PFY; ++k) {
        const int e = min(tid + k * 512, ny - 1);
        const int r = e / (OW * 4), i = e - r * (OW * 4);
        yd[k] = (r << 16) | i;
    }
    f32x4 acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, a = lane & 15;
    const int prow = a >> 2, q = a & 3;
    const int ccolA = q * 8;
    const int nt0 = (wave & 3) * 3, uh = wave >> 2;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    unsigned px[PFX], pe = 0;
    u32x4_t py[PFY];
    int xrows = 0, yrows = 0, pdx = 0;
    auto prefetch = [&](int item) {
        const int f = item / nbands, oh0 = (item % nbands) * R;
        const int ih0 = oh0 * C::S;
        xrows = min(XR, IH - ih0); yrows = min(R, OH - oh0);
        int dy = 0;
        pdx = 0;
        S.offsets(f, pdx, dy);
        const unsigned char* fb = Xb + S.frame(f) * IH * RB;
        const h16_t* yb = dY + ((long long)f * OH + oh0) * OW * C::CO;
        {
            int t = tid;
            asm volatile("" : "+v"(t));
            int r = t / n4, c4 = t - r * n4;
#pragma unroll
            for (int k = 0; k < PFX; ++k) {
                const int rr = min(r, min(XR, xrows) - 1);            // slots past the band / rows below the frame: a valid row (dropped / zeroed at the commit)
                px[k] = *reinterpret_cast<const unsigned*>(fb + min(max(ih0 + rr + dy, 0), IH - 1) * RB + c4 * 4);
                c4 += xdr; r += xdq; if (c4 >= n4) { c4 -= n4; ++r; }
            }
        }
        if (tid < 2 * XR) {                                       // the edge pixels of every row (replicated into the margins at the commit)
            const int rr = tid >> 1, side = tid & 1;
            pe = *reinterpret_cast<const unsigned*>(fb + (long long)min(max(ih0 + min(rr, xrows - 1) + dy, 0), IH - 1) * RB + (side ? RB - 4 : 0));
        }
#pragma unroll
        for (int k = 0; k < PFY; ++k) {
            const int r = yd[k] >> 16, i = yd[k] & 0xffff;
            py[k] = *reinterpret_cast<const u32x4_t*>(yb + (min(r, yrows - 1) * OW) * C::CO + i * 8);
        }
    };
    __shared__ int s_next[2];
    int frame = blockIdx.x, fiter = 0, band = 0;
    int item = frame * nbands;
    if (frame < Nf) prefetch(item);
    while (frame < Nf) {
        if (band == 0 && work_ctr && tid == 0) s_next[fiter & 1] = (int)gridDim.x + atomicAdd(work_ctr, 1);
        __syncthreads();                                          // previous band consumed (first pass: zero fill visible)
        const int cxr = xrows, cyr = yrows, dx = pdx;
        {
            int t = tid;
            asm volatile("" : "+v"(t));
            int r = t / n4, c4 = t - r * n4;
#pragma unroll
            for (int k = 0; k < PFX; ++k) {
                if (t + k * 512 < nx) *(__attribute__((address_space(3))) unsigned*)(raw + r * RP + LM + c4 * 4) = px[k];
                c4 += xdr; r += xdq; if (c4 >= n4) { c4 -= n4; ++r; }
            }
        }
        if (tid < 2 * XR && !W1_PROBE_SKIP(4)) {
            const int rr = tid >> 1, side = tid & 1;
            const unsigned pxl = side ? (pe >> 8) : (pe & 0xffffffu);
            lds_char* dst = raw + rr * RP + (side ? LM + RB : 0);
            for (int k = 0; k < CONV1_RAW_MARGIN; ++k) {
                dst[k * 3 + 0] = (char)(pxl & 0xff); dst[k * 3 + 1] = (char)((pxl >> 8) & 0xff); dst[k * 3 + 2] = (char)((pxl >> 16) & 0xff);
            }
        }
#pragma unroll
        for (int k = 0; k < PFY; ++k)
            if (tid + k * 512 < ny) {
                const int r = yd[k] >> 16, i = yd[k] & 0xffff;
                const u32x4_t v = r < cyr ? py[k] : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; ++e) { bsum[2 * e] += h2f_lo(v[e]); bsum[2 * e + 1] += h2f_hi(v[e]); }
                *(lds_u32x4*)(dyimg + (r * OWp + (i >> 2)) * C::DYS + (i & 3) * 16) = v;
            }
        __syncthreads();                                          // raw rows complete
        {
            const float sc = S.fold ? 1.f : 2.f / 255.f, of = S.fold ? 0.f : -1.f;
            for (int e = W1_PROBE_SKIP(8) ? XR * W4 : tid; e < XR * W4; e += 512) {
                const int r = e / W4, c = e - r * W4;
                u32x2_t ov[3] = {u32x2_t{0u, 0u}, u32x2_t{0u, 0u}, u32x2_t{0u, 0u}};
                if (r < cxr) {                                    // rows below the frame stay zero
                    const int o = r * RP + LM + (c * 4 + dx) * 3;
                    const int sh = o & 3;
                    const __attribute__((address_space(3))) unsigned* wp = (const __attribute__((address_space(3))) unsigned*)(raw + (o & ~3));
                    const unsigned w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3];
                    const unsigned d[3] = {__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh), __builtin_amdgcn_alignbyte(w3, w2, sh)};
                    float v[12];
#pragma unroll
                    for (int k = 0; k < 12; ++k) v[k] = fmaf((float)((d[k >> 2] >> (8 * (k & 3))) & 0xffu), sc, of);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) { ov[ch][0] = pack2h(v[ch], v[3 + ch]); ov[ch][1] = pack2h(v[6 + ch], v[9 + ch]); }
                }
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) *(__attribute__((address_space(3))) u32x2_t*)(ximg + (ch * XR + r) * XRS + c * 8) = ov[ch];
            }
        }
        __syncthreads();
        if (++band == nbands) {
            band = 0;
            frame = work_ctr ? s_next[fiter & 1] : frame + (int)gridDim.x;
            ++fiter;
        }
        item = frame * nbands + band;
        if (frame < Nf && !W1_PROBE_SKIP(2)) prefetch(item);      // in flight during the MFMAs below
        const int units = R * U;
        // run u = u0 + g -> (row ur, run uo of the row) advanced by adds (8 runs per step): a runtime division per step stood next to 6 MFMAs
        int joff[3];                                   // per n-tile: (channel, kernel row) of this lane's B rows + its column half
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int nt = nt0 + j; joff[j] = ((nt >> 2) * XR + (nt & 3) * 2 + (q >> 1)) * XRS + (q & 1) * 8; }
        const int q8 = 8 / U, r8 = 8 - q8 * U;
        int ur, uo;
        { const int u = uh * 4 + g; ur = u / U; uo = u - ur * U; }
#pragma unroll 1
        for (int u0 = W1_PROBE_SKIP(1) ? units : uh * 4; u0 < units; u0 += 8) {
            const int u = u0 + g;
            const bool valid = u < units;
            const int r = valid ? ur : 0, ow0 = valid ? uo * 8 : 0;
            ur += q8; uo += r8;
            if (uo >= U) { uo -= U; ++ur; }
            const int pixA = valid ? r * OWp + ow0 : R * OWp;
            lds_char* abase = dyimg + (pixA + prow) * C::DYS + ccolA;
            h16x8_t af[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) af[c] = tr_read8(abase + c * 32, abase + 4 * C::DYS + c * 32);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                lds_char* bbase = ximg + joff[j] + (r * C::S) * XRS + (ow0 + prow) * C::S * 2;
                const h16x8_t bf = tr_read8(bbase, bbase + 4 * C::S * 2);
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[j][c] = MFMA_16x16x32_H(af[c], bf, acc[j][c], 0, 0, 0);
            }
        }
    }
    // ---- the bias partial of this workgroup first (Conv1Src::fold needs it for the slab), then the two unit halves (waves w and w + 4) are summed
    // through LDS and one slab per workgroup is written
    __syncthreads();
    float* redf = reinterpret_cast<float*>(smem);
    float* dbw = redf + 6144;                                    // [CO]: this workgroup's bias-gradient partial (24 KB in: behind redf's 16 KB and red's 24 KB)
#pragma unroll
    for (int e = 0; e < 8; ++e) redf[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < C::CO) {
        const int cgrp = tid >> 3, e = tid & 7;
        float sacc = 0.f;
        for (int t = cgrp; t < 512; t += 4) sacc += redf[t * 8 + e];
        unsafeAtomicAdd(bias_part + tid, sacc);
        dbw[tid] = sacc;
    }
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (uh == 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) red[((wave & 3) * 6 + j * 2 + c) * 64 + lane] = acc[j][c];
    }
    __syncthreads();
    if (uh == 0) {
        float* out = part + (long long)blockIdx.x * C::CO * 192;
        const float fsc = S.fold ? CONV1_FOLD_SCALE : 1.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 v = acc[j][c] + red[((wave & 3) * 6 + j * 2 + c) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = c * 16 + g * 4 + r;
                    out[(long long)co * 192 + (nt0 + j) * 16 + a] = S.fold ? fmaf(v[r], fsc, -dbw[co]) : v[r];      // dW = (2/255) (dY * u) - db (x) 1
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Round 6: the uint8 kernel above with the conversion done FROM THE PREFETCH REGISTERS (tools/conv1_wgrad_probe.hip: commit + conversion were 119 of its 251 us — raw
// rows into LDS, margins, a barrier, a second pass that reads them back — and the 4 x smaller frames bought nothing over the fp32 boundary's 261 us).  A prefetch slot is
// the aligned 16-byte window around a 4-pixel group's 12 bytes (one dwordx4 load; 33 % more bytes requested, the overlaps hit in L2); the column shift and the replicate
// pad are resolved per pixel from the window at the commit.  One barrier per band less, no raw rows in LDS.
// ---------------------------------------------------------------------------------------------------------------------
// SPLIT (round 6, second pass): slots 0 .. PFX - 2 hold INTERIOR groups only (conv1_interior_groups: no clamp, no per-pixel select, one alignbyte shift per frame),
// slot PFX - 1 the few groups next to the row ends (threads [0, XR * edge groups per row)) with the general conversion — the slot index is an unrolled compile-time
// constant, so no wave pays the slow path for its interior lanes.  Same LDS image, same slabs.
template <int PFX, int PFY, bool SPLIT = false>
__global__ void __launch_bounds__(512, 4) conv1_wgrad_tr2r_kernel(Conv1Src S, int* __restrict__ work_ctr, const h16_t* __restrict__ dY, float* __restrict__ part,
                                                                  float* __restrict__ bias_part, int Nf, int IH, int IW, int OH, int OW, int R, int nbands) {
    using C = Wgrad1Cfg;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int OWp = (OW + 7) & ~7, U = OWp >> 3;
    const int XR = (R - 1) * C::S + C::KH;
    const int XRS = IW * 2 + 16;
    const int xbytes = C::C * XR * XRS + 512;
    const int dypix = R * OWp + 8;
    lds_char* ximg = (lds_char*)smem;
    lds_char* dyimg = ximg + xbytes;
    for (int i = tid * 16; i < xbytes + dypix * C::DYS; i += 512 * 16) *(lds_u32x4*)((lds_char*)smem + i) = u32x4_t{0u, 0u, 0u, 0u};
    const int W4 = IW >> 2;
    const int RB = IW * 3, n4 = RB >> 2;                          // bytes / 4-byte chunks per source row
    const int nx = XR * W4, ny = R * OW * 4;                       // 4-pixel groups of a band (one 16-byte slot each), dY channel quarters
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(S.X);
    // raw slot k of this thread = chunk tid + 512 k of the [XR][n4] grid: (row, chunk) advanced incrementally from slot 0's — twelve
    // descriptor registers next to twelve slots of in-flight data spilled 17 registers at the 128-VGPR budget and serialised the prefetch
    // ... and the compiler hoists whatever is band-invariant out of the band loop and spills it just the same (a scratch reload in front of
    // every prefetch load waits for the loads issued before it): slot 0's (row, chunk) is recomputed per band behind an opaque copy of tid
    const int xdq = 512 / W4, xdr = 512 - xdq * W4;                // slot k of this thread = group tid + 512 k of the [XR][W4] grid, advanced by adds
    // SPLIT: interior groups [cl, cl + WI) of every row in slots 0 .. PFX - 2 ([XR][WI] grid), the WE = W4 - WI others in slot PFX - 1 ([XR][WE] grid)
    int cl = 0, WI = 0;
    if (SPLIT) conv1_interior_groups(IW, S.pad, ((PFX - 1) * 512) / XR, cl, WI);
    const int WE = W4 - WI, ni = XR * WI, ne = XR * WE;
    const int idq = SPLIT ? 512 / max(WI, 1) : 0, idr = 512 - idq * WI;
    const float invWI = 1.f / (float)max(WI, 1), invWE = 1.f / (float)max(WE, 1);
    int yd[PFY];
#pragma unroll
    for (int k = 0; k < PFY; ++k) {
        const int e = min(tid + k * 512, ny - 1);
        const int r = e / (OW * 4), i = e - r * (OW * 4);
        yd[k] = (r << 16) | i;
    }
    f32x4 acc[3][2];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, a = lane & 15;
    const int prow = a >> 2, q = a & 3;
    const int ccolA = q * 8;
    const int nt0 = (wave & 3) * 3, uh = wave >> 2;
    float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    u32x4_t px[PFX];                                              // slot = the aligned 16-byte window that holds a group's 12 bytes
    u32x4_t py[PFY];
    int xrows = 0, yrows = 0, pdx = 0;
    auto prefetch = [&](int item) {
        const int f = item / nbands, oh0 = (item % nbands) * R;
        const int ih0 = oh0 * C::S;
        xrows = min(XR, IH - ih0); yrows = min(R, OH - oh0);
        int dy = 0;
        pdx = 0;
        S.offsets(f, pdx, dy);
        const unsigned char* fb = Xb + S.frame(f) * IH * RB;
        const h16_t* yb = dY + ((long long)f * OH + oh0) * OW * C::CO;
        if constexpr (SPLIT) {
            int t = tid;
            asm volatile("" : "+v"(t));
            const int rmax = min(XR, xrows) - 1;
            int r = wg_fdiv(t, invWI), ci = t - r * WI;
            const int off0 = 12 * cl + 3 * pdx;                      // 3 (4 c + dx) = the group's first byte in its row
#pragma unroll
            for (int k = 0; k < PFX - 1; ++k) {
                px[k] = *reinterpret_cast<const u32x4_t*>(fb + (long long)min(max(ih0 + min(r, rmax) + dy, 0), IH - 1) * RB + ((off0 + 12 * ci) & ~3));
                ci += idr; r += idq; if (ci >= WI) { ci -= WI; ++r; }
            }
            const int re = wg_fdiv(t, invWE), ce = t - re * WE, c = ce < cl ? ce : ce + WI;
            px[PFX - 1] = *reinterpret_cast<const u32x4_t*>(fb + (long long)min(max(ih0 + min(re, rmax) + dy, 0), IH - 1) * RB + conv1_window_off(c, pdx, IW, RB));
        } else {
            int t = tid;
            asm volatile("" : "+v"(t));
            int r = t / W4, c = t - r * W4;
#pragma unroll
            for (int k = 0; k < PFX; ++k) {
                const int rr = min(r, min(XR, xrows) - 1);            // slots past the band / rows below the frame: a valid row (dropped / zeroed at the commit)
                px[k] = *reinterpret_cast<const u32x4_t*>(fb + (long long)min(max(ih0 + rr + dy, 0), IH - 1) * RB + conv1_window_off(c, pdx, IW, RB));
                c += xdr; r += xdq; if (c >= W4) { c -= W4; ++r; }
            }
        }
#pragma unroll
        for (int k = 0; k < PFY; ++k) {
            const int r = yd[k] >> 16, i = yd[k] & 0xffff;
            py[k] = *reinterpret_cast<const u32x4_t*>(yb + (min(r, yrows - 1) * OW) * C::CO + i * 8);
        }
    };
    __shared__ int s_next[2];
    int frame = blockIdx.x, fiter = 0, band = 0;
    int item = frame * nbands;
    if (frame < Nf) prefetch(item);
    while (frame < Nf) {
        if (band == 0 && work_ctr && tid == 0) s_next[fiter & 1] = (int)gridDim.x + atomicAdd(work_ctr, 1);
        __syncthreads();                                          // previous band consumed (first pass: zero fill visible)
        const int cxr = xrows, cyr = yrows, dx = pdx;
        if (!W1_PROBE_SKIP(8)) {
            // the band's 4-pixel groups straight from the prefetch registers into the 16-bit [c][row][iw] image (no raw rows in LDS, no margin fill, no barrier
            // between a raw commit and a conversion pass): window -> the 12 bytes of pixels qp .. qp + 3 -> for output pixel k of the group the source pixel
            // clamp(c 4 + k + dx) - qp (= k everywhere but at the row ends, where RandomShiftsAug's replicate pad repeats the edge pixel)
            auto commit_x = [&](auto FOLD_) __attribute__((always_inline)) {      // the folded path (production) converts the byte as it is: no multiply-add
            constexpr bool FOLD = decltype(FOLD_)::value;
            int t = tid;
            asm volatile("" : "+v"(t));
            if constexpr (SPLIT) {
                const int shu = (3 * dx) & 3;
                int r = wg_fdiv(t, invWI), ci = t - r * WI;
#pragma unroll
                for (int k = 0; k < PFX - 1; ++k) {
                    if (t + k * 512 < ni) {
                        u32x2_t ov[3] = {u32x2_t{0u, 0u}, u32x2_t{0u, 0u}, u32x2_t{0u, 0u}};
                        if (r < cxr) {
                            unsigned lo[3], hi[3];
                            conv1_window_group_interior<FOLD>(px[k], shu, lo, hi);
#pragma unroll
                            for (int ch = 0; ch < 3; ++ch) { ov[ch][0] = lo[ch]; ov[ch][1] = hi[ch]; }
                        }
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) *(__attribute__((address_space(3))) u32x2_t*)(ximg + (ch * XR + r) * XRS + (cl + ci) * 8) = ov[ch];
                    }
                    ci += idr; r += idq; if (ci >= WI) { ci -= WI; ++r; }
                }
                if (t < ne) {
                    const int re = wg_fdiv(t, invWE), ce = t - re * WE, c = ce < cl ? ce : ce + WI;
                    u32x2_t ov[3] = {u32x2_t{0u, 0u}, u32x2_t{0u, 0u}, u32x2_t{0u, 0u}};
                    if (re < cxr) {
                        unsigned lo[3], hi[3];
                        conv1_window_group<FOLD>(px[PFX - 1], c, dx, IW, RB, lo, hi);
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) { ov[ch][0] = lo[ch]; ov[ch][1] = hi[ch]; }
                    }
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) *(__attribute__((address_space(3))) u32x2_t*)(ximg + (ch * XR + re) * XRS + c * 8) = ov[ch];
                }
                return;
            }
            int r = t / W4, c = t - r * W4;
#pragma unroll
            for (int k = 0; k < PFX; ++k) {
                if (t + k * 512 < nx) {
                    u32x2_t ov[3] = {u32x2_t{0u, 0u}, u32x2_t{0u, 0u}, u32x2_t{0u, 0u}};
                    if (r < cxr) {                                // rows below the frame stay zero
                        unsigned lo[3], hi[3];
                        conv1_window_group<FOLD>(px[k], c, dx, IW, RB, lo, hi);
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch) { ov[ch][0] = lo[ch]; ov[ch][1] = hi[ch]; }
                    }
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) *(__attribute__((address_space(3))) u32x2_t*)(ximg + (ch * XR + r) * XRS + c * 8) = ov[ch];
                }
                c += xdr; r += xdq; if (c >= W4) { c -= W4; ++r; }
            }
            };
            if (S.fold) commit_x(std::true_type{}); else commit_x(std::false_type{});
        }
#pragma unroll
        for (int k = 0; k < PFY; ++k)
            if (tid + k * 512 < ny && !W1_PROBE_SKIP(16)) {
                const int r = yd[k] >> 16, i = yd[k] & 0xffff;
                const u32x4_t v = r < cyr ? py[k] : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; ++e) { bsum[2 * e] += h2f_lo(v[e]); bsum[2 * e + 1] += h2f_hi(v[e]); }
                *(lds_u32x4*)(dyimg + (r * OWp + (i >> 2)) * C::DYS + (i & 3) * 16) = v;
            }
        __syncthreads();
        if (++band == nbands) {
            band = 0;
            frame = work_ctr ? s_next[fiter & 1] : frame + (int)gridDim.x;
            ++fiter;
        }
        item = frame * nbands + band;
        if (frame < Nf && !W1_PROBE_SKIP(2)) prefetch(item);      // in flight during the MFMAs below
        const int units = R * U;
        // run u = u0 + g -> (row ur, run uo of the row) advanced by adds (8 runs per step): a runtime division per step stood next to 6 MFMAs
        int joff[3];                                   // per n-tile: (channel, kernel row) of this lane's B rows + its column half
#pragma unroll
        for (int j = 0; j < 3; ++j) { const int nt = nt0 + j; joff[j] = ((nt >> 2) * XR + (nt & 3) * 2 + (q >> 1)) * XRS + (q & 1) * 8; }
        const int q8 = 8 / U, r8 = 8 - q8 * U;
        int ur, uo;
        { const int u = uh * 4 + g; ur = u / U; uo = u - ur * U; }
#pragma unroll 1
        for (int u0 = W1_PROBE_SKIP(1) ? units : uh * 4; u0 < units; u0 += 8) {
            const int u = u0 + g;
            const bool valid = u < units;
            const int r = valid ? ur : 0, ow0 = valid ? uo * 8 : 0;
            ur += q8; uo += r8;
            if (uo >= U) { uo -= U; ++ur; }
            const int pixA = valid ? r * OWp + ow0 : R * OWp;
            lds_char* abase = dyimg + (pixA + prow) * C::DYS + ccolA;
            h16x8_t af[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) af[c] = tr_read8(abase + c * 32, abase + 4 * C::DYS + c * 32);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                lds_char* bbase = ximg + joff[j] + (r * C::S) * XRS + (ow0 + prow) * C::S * 2;
                const h16x8_t bf = tr_read8(bbase, bbase + 4 * C::S * 2);
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[j][c] = MFMA_16x16x32_H(af[c], bf, acc[j][c], 0, 0, 0);
            }
        }
    }
    // ---- the bias partial of this workgroup first (Conv1Src::fold needs it for the slab), then the two unit halves (waves w and w + 4) are summed
    // through LDS and one slab per workgroup is written
    __syncthreads();
    float* redf = reinterpret_cast<float*>(smem);
    float* dbw = redf + 6144;                                    // [CO]: this workgroup's bias-gradient partial (24 KB in: behind redf's 16 KB and red's 24 KB)
#pragma unroll
    for (int e = 0; e < 8; ++e) redf[tid * 8 + e] = bsum[e];
    __syncthreads();
    if (tid < C::CO) {
        const int cgrp = tid >> 3, e = tid & 7;
        float sacc = 0.f;
        for (int t = cgrp; t < 512; t += 4) sacc += redf[t * 8 + e];
        unsafeAtomicAdd(bias_part + tid, sacc);
        dbw[tid] = sacc;
    }
    __syncthreads();
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    if (uh == 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) red[((wave & 3) * 6 + j * 2 + c) * 64 + lane] = acc[j][c];
    }
    __syncthreads();
    if (uh == 0) {
        float* out = part + (long long)blockIdx.x * C::CO * 192;
        const float fsc = S.fold ? CONV1_FOLD_SCALE : 1.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const f32x4 v = acc[j][c] + red[((wave & 3) * 6 + j * 2 + c) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = c * 16 + g * 4 + r;
                    out[(long long)co * 192 + (nt0 + j) * 16 + a] = S.fold ? fmaf(v[r], fsc, -dbw[co]) : v[r];      // dW = (2/255) (dY * u) - db (x) 1
                }
            }
    }
}

inline int g_conv1_wgrad_u8reg = -1;      // tools/conv1_wgrad_probe.hip only: 0 / 1 force the uint8 kernel form (-1: the build's own)
static inline int launch_conv1_wgrad_tr(hipStream_t st, const Conv1Src& X, const h16_t* dY, float* part, float* bias_part, int Nf, int IH, int IW, int OH,
                                        int OW, int max_blocks, int* work_ctr = nullptr) {
    static const int v2 = HULC_SWITCH("HULC_W1_V2", 1);
    static const int u8reg_sw = HULC_SWITCH("HULC_W1_U8REG", 2);      // uint8: 2 = conversion from the prefetch registers with interior / row-end slots (conv1_wgrad_tr2r_kernel<3, 2, true>), 1 = one general slot kind, 0 = round 5's raw rows through LDS
    if (v2 && !X.u8 && (IW % 4) == 0) {
        // tallest band whose images fit 2 workgroups per CU and whose chunks fit the prefetch slots (8 frame + 2 dY registers of 16 B per thread:
        // 10 + 3 slots spilled 46 registers of in-flight data at the 128-VGPR budget of 2 x 8 waves per CU, which serialised the prefetch)
        int R = OH;
        auto fits = [&](int r) {
            const int XR = (r - 1) * Wgrad1Cfg::S + Wgrad1Cfg::KH;
            return Wgrad1Cfg::lds_bytes(r, IW, OW) <= (size_t)79 * 1024 && (long long)3 * XR * (IW / 4) <= 6 * 512 && (long long)r * OW * 4 <= 2 * 512 && XR < 128 && r < 128;
        };
        while (R > 1 && !fits(R)) --R;
        if (fits(R)) {
            const int nb = (OH + R - 1) / R;
            R = (OH + nb - 1) / nb;
            const size_t lds = std::max<size_t>(Wgrad1Cfg::lds_bytes(R, IW, OW), 512 * 8 * sizeof(float));
            static bool attr2 = false;
            if (!attr2) { hipFuncSetAttribute((const void*)conv1_wgrad_tr2_kernel<6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr2 = true; }
            const int grid = std::min(std::min(Nf, 512), max_blocks);
            hipLaunchKernelGGL((conv1_wgrad_tr2_kernel<6, 2>), dim3(grid), dim3(512), lds, st, reinterpret_cast<const float*>(X.X), work_ctr, dY, part, bias_part, Nf, IH, IW, OH, OW, R, nb);
            return grid;
        }
    }
    const int u8form = g_conv1_wgrad_u8reg < 0 ? u8reg_sw : g_conv1_wgrad_u8reg;
    if (v2 && X.u8 && (IW % 4) == 0 && IW >= 8 && u8form >= 2) {
        // round 6, second pass: 2 slots of interior groups (fast conversion) + 1 slot of row-end groups + 2 dY slots per thread (conv1_wgrad_tr2r_kernel<3, 2, true>)
        int R = OH;
        auto fits = [&](int r) {
            const int XR = (r - 1) * Wgrad1Cfg::S + Wgrad1Cfg::KH;
            int cl, wi;
            conv1_interior_groups(IW, X.pad, (2 * 512) / XR, cl, wi);
            return Wgrad1Cfg::lds_bytes(r, IW, OW, false) <= (size_t)79 * 1024 && wi >= 1 && (long long)XR * wi <= 2 * 512 && (long long)XR * (IW / 4 - wi) <= 512 &&
                   (long long)r * OW * 4 <= 2 * 512 && r < 128;
        };
        while (R > 1 && !fits(R)) --R;
        if (fits(R)) {
            const int nb = (OH + R - 1) / R;
            R = (OH + nb - 1) / nb;
            const size_t lds = std::max<size_t>(Wgrad1Cfg::lds_bytes(R, IW, OW, false), 32 * 1024);
            static bool attr2s = false;
            if (!attr2s) { hipFuncSetAttribute((const void*)conv1_wgrad_tr2r_kernel<3, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr2s = true; }
            const int grid = std::min(std::min(Nf, 512), max_blocks);
            hipLaunchKernelGGL((conv1_wgrad_tr2r_kernel<3, 2, true>), dim3(grid), dim3(512), lds, st, X, work_ctr, dY, part, bias_part, Nf, IH, IW, OH, OW, R, nb);
            return grid;
        }
    }
    if (v2 && X.u8 && (IW % 4) == 0 && IW >= 8 && u8form) {
        // uint8 boundary, round 6: 4 window slots of 16 bytes + 2 dY slots per thread, converted from the registers (conv1_wgrad_tr2r_kernel); no raw rows in LDS
        int R = OH;
        auto fits = [&](int r) {
            const int XR = (r - 1) * Wgrad1Cfg::S + Wgrad1Cfg::KH;
            return Wgrad1Cfg::lds_bytes(r, IW, OW, false) <= (size_t)79 * 1024 && (long long)XR * (IW / 4) <= 4 * 512 && (long long)r * OW * 4 <= 2 * 512 && r < 128;
        };
        while (R > 1 && !fits(R)) --R;
        if (fits(R)) {
            const int nb = (OH + R - 1) / R;
            R = (OH + nb - 1) / nb;
            const size_t lds = std::max<size_t>(Wgrad1Cfg::lds_bytes(R, IW, OW, false), 32 * 1024);      // (>= the epilogue's reduction areas: 24.6 KB)
            static bool attr2r = false;
            if (!attr2r) { hipFuncSetAttribute((const void*)conv1_wgrad_tr2r_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr2r = true; }
            const int grid = std::min(std::min(Nf, 512), max_blocks);
            hipLaunchKernelGGL((conv1_wgrad_tr2r_kernel<4, 2>), dim3(grid), dim3(512), lds, st, X, work_ctr, dY, part, bias_part, Nf, IH, IW, OH, OW, R, nb);
            return grid;
        }
    }
    if (v2 && X.u8 && (IW % 4) == 0) {
        // uint8 boundary: 12 raw 4-byte slots + 2 dY slots per thread; the raw rows add their LDS area to the two images
        int R = OH;
        auto fits = [&](int r) {
            const int XR = (r - 1) * Wgrad1Cfg::S + Wgrad1Cfg::KH;
            return Wgrad1Cfg::lds_bytes(r, IW, OW, true) <= (size_t)79 * 1024 && (long long)XR * (IW * 3 / 4) <= 12 * 512 && (long long)r * OW * 4 <= 2 * 512 && 2 * XR <= 512 && r < 128;
        };
        while (R > 1 && !fits(R)) --R;
        if (fits(R)) {
            const int nb = (OH + R - 1) / R;
            R = (OH + nb - 1) / nb;
            const size_t lds = std::max<size_t>(Wgrad1Cfg::lds_bytes(R, IW, OW, true), 512 * 8 * sizeof(float));
            static bool attr2u = false;
            if (!attr2u) { hipFuncSetAttribute((const void*)conv1_wgrad_tr2u_kernel<12, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr2u = true; }
            const int grid = std::min(std::min(Nf, 512), max_blocks);
            hipLaunchKernelGGL((conv1_wgrad_tr2u_kernel<12, 2>), dim3(grid), dim3(512), lds, st, X, work_ctr, dY, part, bias_part, Nf, IH, IW, OH, OW, R, nb);
            return grid;
        }
    }
    static const int lds_kb = HULC_SWITCH("HULC_W1_LDS", 39);   // 4 workgroups per CU (0.44 vs 0.50 ms/step at 2 per CU with 78 KB bands)
    static const int env_wg = HULC_SWITCH("HULC_W1_WG", 0);
    if (env_wg > 0) max_blocks = env_wg;
    int R = OH;
    while (R > 1 && Wgrad1Cfg::lds_bytes(R, IW, OW, X.u8) > (size_t)lds_kb * 1024) --R;
    const size_t lds = Wgrad1Cfg::lds_bytes(R, IW, OW, X.u8);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv1_wgrad_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    const int grid = Nf < max_blocks ? Nf : max_blocks;
    Conv1Src X1 = X; X1.fold = 0;      // this (fallback) kernel writes the plain slab: it stages x itself
    hipLaunchKernelGGL(conv1_wgrad_tr_kernel, dim3(grid), dim3(256), lds, st, X1, work_ctr, dY, part, bias_part, Nf, IH, IW, OH, OW, R);
    return grid;
}

// ---------------------------------------------------------------------------------------------------------------------
// Linear-layer backward for M <= 64 rows (every M = B MLP layer): dW[N][K] += dY^T X and db[N] += colsum(dY) in ONE launch.
// Both operands are reduction(M)-major in memory; their tiles are staged in LDS as they lie ([m][n] and [m][k]) and the MFMA
// fragments come from transposing reads, so no transposed activation copies (and no separate column-sum launches) are needed.
// Orientation: A = X fragment (rows k), B = dY fragment (cols n)  ->  a lane owns dW[n][k..k+3]: float4 read-modify-write.
// ---------------------------------------------------------------------------------------------------------------------
DEVI void lin_bwd_smallm_body(const h16_t* __restrict__ dY, long long ldy, const h16_t* __restrict__ X, long long ldx,
                              int M, int N, int K, float* __restrict__ dW, long long lddw, float* __restrict__ db,
                              float* __restrict__ db2, int mchunk, int store, const int bx, const int by, const int bz, const int gz, char* smem,
                              const bool slabs = false) {
    // slabs: dW is this row chunk's OWN zero-assumed slab (plain stores, reduced later by the batched unpack launch); only the bias uses atomics
    // store != 0: dW / db are known to be all zeros (first backward after hulc_zero_grads): the tile is stored instead of read, added and written
    // Large M (token-major transformer / encoder layers): block z (bz of gz) owns rows [z*mchunk, (z+1)*mchunk), loops over them 64 at a
    // time and adds its partial with fp32 atomics — replaces "transpose dY, transpose X, split-K NT GEMM, column-sum" (4 launches).
    constexpr int TN = 64, TK = 128, YS = TN * 2 + 16, XS = TK * 2 + 16;
    lds_char* yimg = (lds_char*)smem;
    lds_char* ximg = yimg + 64 * YS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = bx * TN, k0 = by * TK;
    const int mbeg = bz * mchunk, mend = min(M, mbeg + mchunk);
    const int g = lane >> 4, a = lane & 15;
    const int prow = a >> 2, ccol = (a & 3) * 8;
    // wave w: n-tile w (16 columns of dY) x 8 k-tiles
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    // the 8 dW quads this lane accumulates into are fetched NOW (single launch over M: plain read-modify-write), so their latency hides
    // under the staging / MFMA phase instead of serialising as 8 dependent load-add-store rounds at the end
    const bool atomic_b = gz > 1;
    const bool atomic = atomic_b && !slabs;
    const int n = n0 + wave * 16 + a;
    // Whole-tile output path (wave-uniform condition): the accumulators leave in ROW shape, not in fragment shape.  A fragment-shaped store puts 16
    // different dW rows on 16 adjacent lanes — 64 separate 16-byte accesses per wave instruction, 64 contiguous bytes per row; the launches that
    // write the M <= 64 Linear layers' fp32 gradients ran at ~3 TB/s that way.  Each wave turns its 16 x 128 tile through a private LDS image (the
    // operand images are dead by then) so that an instruction writes 4 rows x 256 contiguous bytes; the old values are read in the same shape.
    const bool vec_ok = !atomic && (lddw & 3) == 0 && ((reinterpret_cast<uintptr_t>(dW) & 15) == 0) && k0 + TK <= K;
    const int orow = lane >> 4, ocol = (lane & 15) * 4;          // output shape: row 4 it + orow of the wave's 16, columns 64 h + ocol .. + 3
    float4 oldw[8];
    if (vec_ok) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int nr = n0 + wave * 16 + (j & 3) * 4 + orow;
            oldw[j] = (store || nr >= N) ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(dW + (long long)nr * lddw + k0 + (j >> 2) * 64 + ocol);
        }
    }
    // the next 64-row chunk's tiles are requested into registers before the MFMAs of the current one (2 + 4 x 16 bytes per thread)
    u32x4_t ry[2], rx[4];
    auto fetch = [&](int m0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256, m = i / (TN / 8), c = i % (TN / 8);
            h16_t v[8];
            if (m0 + m < mend && n0 + c * 8 < N) load8_guard<h16_t>(dY + (long long)(m0 + m) * ldy + n0 + c * 8, N - (n0 + c * 8), v);
            else zero8<h16_t>(v);
            ry[u] = *reinterpret_cast<const u32x4_t*>(v);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = tid + u * 256, m = i / (TK / 8), c = i % (TK / 8);
            h16_t v[8];
            if (m0 + m < mend && k0 + c * 8 < K) load8_guard<h16_t>(X + (long long)(m0 + m) * ldx + k0 + c * 8, K - (k0 + c * 8), v);
            else zero8<h16_t>(v);
            rx[u] = *reinterpret_cast<const u32x4_t*>(v);
        }
    };
    fetch(mbeg);
    for (int m0 = mbeg; m0 < mend; m0 += 64) {
        if (m0 > mbeg) __syncthreads();
        // stage dY[m0:m0+64][n0:n0+64] and X[m0:m0+64][k0:k0+128] (rows >= mend and columns past the edge -> zeros)
#pragma unroll
        for (int u = 0; u < 2; ++u) { const int i = tid + u * 256; *(lds_u32x4*)(yimg + (i / (TN / 8)) * YS + (i % (TN / 8)) * 16) = ry[u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = tid + u * 256; *(lds_u32x4*)(ximg + (i / (TK / 8)) * XS + (i % (TK / 8)) * 16) = rx[u]; }
        __syncthreads();
        if (m0 + 64 < mend) fetch(m0 + 64);
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {                        // reduction over m: 2 x 32
            const int mrow = ms * 32 + g * 8 + prow;
            lds_char* yb = yimg + mrow * YS + wave * 32 + ccol;
            const h16x8_t yf = tr_read8(yb, yb + 4 * YS);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                lds_char* xb = ximg + mrow * XS + j * 32 + ccol;
                const h16x8_t xf = tr_read8(xb, xb + 4 * XS);
                acc[j] = MFMA_16x16x32_H(xf, yf, acc[j], 0, 0, 0);   // D[row = k][col = n]
            }
        }
        if (db && by == 0) {                                    // column sums of dY: wave w adds rows 16 w .. 16 w + 15 of column `lane`
#pragma unroll 16
            for (int m = 0; m < 16; ++m) bsum += h2f(*(__attribute__((address_space(3))) h16_t*)(yimg + (wave * 16 + m) * YS + lane * 2));
        }
    }
    if (db && by == 0) {                                        // the four waves' partial column sums meet in LDS (the tiles are dead)
        __syncthreads();
        *(__attribute__((address_space(3))) float*)(yimg + (wave * 64 + lane) * 4) = bsum;
        __syncthreads();
        if (tid < TN) {
            bsum = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) bsum += *(__attribute__((address_space(3))) float*)(yimg + (w * 64 + tid) * 4);
        }
    }
    if (vec_ok) {
        constexpr int OP = 64 * 4 + 16;                          // pitch of one row's 64 fp32 columns in the wave's image
        __syncthreads();                                         // every wave is past its last read of the operand images (and of the bias sums)
        lds_char* const tb = (lds_char*)smem + wave * (16 * OP);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) *(__attribute__((address_space(3))) f32x4*)(tb + a * OP + jj * 64 + g * 16) = acc[h * 4 + jj];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const f32x4 v = *(__attribute__((address_space(3))) f32x4*)(tb + (it * 4 + orow) * OP + ocol * 4);
                const int nr = n0 + wave * 16 + it * 4 + orow;
                float4 o = oldw[h * 4 + it];
                o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
                if (nr < N) *reinterpret_cast<float4*>(dW + (long long)nr * lddw + k0 + h * 64 + ocol) = o;
            }
        }
    } else if (n < N) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j * 16 + g * 4;
            float* p = dW + (long long)n * lddw + k;
            if (atomic) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k + r < K) unsafeAtomicAdd(p + r, acc[j][r]);
            } else if (k + 3 < K && ((((uintptr_t)p) & 15) == 0)) {
                // store mode (slabs are never zeroed; first backward after zero_grads): the ragged last k-tile must STORE too, not add to what lies there
                float4 o = store ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<float4*>(p);
                o.x += acc[j][0]; o.y += acc[j][1]; o.z += acc[j][2]; o.w += acc[j][3];
                *reinterpret_cast<float4*>(p) = o;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k + r < K) p[r] = store ? acc[j][r] : p[r] + acc[j][r];
            }
        }
    }
    if (db && by == 0 && tid < TN && n0 + tid < N) {
        if (atomic_b) { unsafeAtomicAdd(db + n0 + tid, bsum); if (db2) unsafeAtomicAdd(db2 + n0 + tid, bsum); }
        else { db[n0 + tid] += bsum; if (db2) db2[n0 + tid] += bsum; }
    }
}

constexpr int LBS_LDS = 64 * (64 * 2 + 16) + 64 * (128 * 2 + 16);
__global__ void __launch_bounds__(256) lin_bwd_smallm_kernel(const h16_t* __restrict__ dY, long long ldy, const h16_t* __restrict__ X, long long ldx,
                                                             int M, int N, int K, float* __restrict__ dW, long long lddw, float* __restrict__ db,
                                                             float* __restrict__ db2, int mchunk, int store = 0) {
    __shared__ __attribute__((aligned(16))) char smem[LBS_LDS];
    lin_bwd_smallm_body(dY, ldy, X, ldx, M, N, K, dW, lddw, db, db2, mchunk, store, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.z, smem);
}
// the weight / bias gradients of up to eight M <= 64 Linear layers in ONE launch (the layers of an MLP's backward: each was a ~9 us launch
// behind its own data-gradient GEMM); problem = the one whose block range holds blockIdx.x
struct LinBwdJob { const h16_t* dY; const h16_t* X; float* dW; float* db; long long ldx, lddw; int N, K, nx, blk0; float* part; };   // part: slabs [gridDim.y][N][K] (mchunk mode) or null
struct LinBwdBatch { LinBwdJob j[8]; int n, M, store, mchunk; };      // mchunk > 0: rows split over blockIdx.y in chunks of mchunk (fp32 atomics)
__global__ void __launch_bounds__(256) lin_bwd_smallm_batched_kernel(LinBwdBatch bt) {
    __shared__ __attribute__((aligned(16))) char smem[LBS_LDS];
    int k = 0;
    while (k + 1 < bt.n && (int)blockIdx.x >= bt.j[k + 1].blk0) ++k;
    const LinBwdJob J = bt.j[k];
    const int b = blockIdx.x - J.blk0;
    const bool slabs = bt.mchunk > 0 && J.part;
    lin_bwd_smallm_body(J.dY, (long long)J.N, J.X, J.ldx, bt.M, J.N, J.K, slabs ? J.part + (long long)blockIdx.y * J.N * J.K : J.dW, slabs ? (long long)J.K : J.lddw, J.db, nullptr,
                        bt.mchunk > 0 ? bt.mchunk : 64, slabs ? 1 : bt.store, b % J.nx, b / J.nx, blockIdx.y, gridDim.y, smem, slabs);
}

}  // namespace HULC_NS
