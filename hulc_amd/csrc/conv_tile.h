// hulc_amd/csrc/conv_tile.h — raw-tile convolution kernels for NHWC bf16 activations on gfx950 (forward and data-gradient).
//
// Instead of an implicit-GEMM gather (every input element re-fetched KH*KW/S^2 times through the loader), a persistent
// workgroup (8 waves) keeps the whole packed weight matrix in LDS and streams *raw* row bands of the image through a second
// LDS region: the (kh,kw) taps are plain address offsets into that band, so each activation byte is read from HBM once.
// The next band is prefetched into registers while the current one is being multiplied.
//
//   forward  (REV=0): out[i][j][cn]            = act( sum_{ta,tb,ck} img[i*SI+ta][j*SI+tb][ck] * W[cn][(ta,tb,ck)] + bias[cn] )
//   dgrad    (REV=1): out[i*OS+ph][j*OS+pw][cn] = mask * sum_{ta,tb,ck} img[i-ta][j-tb][ck] * W[(ph,pw)][cn][(ta,tb,ck)]
//                     (img = dY with implicit zero border; (ph,pw) = stride-parity class of the input pixel, OS = conv stride)
//
// MFMA orientation: A = weight fragment (rows = cn), B = image fragment (cols = 16 flattened pixels of the band), so each lane
// ends up with 4 consecutive output channels of one pixel -> 8-byte stores, 4-channel bias / mask loads.
#pragma once
#include "common.h"
#include "conv_wgrad.h"   // lds_char, u32x4_t

namespace HULC_NS {

struct ConvTileP {
    const h16_t* img; int IMH, IMW;
    const h16_t* w;
    h16_t* out; int OUTH, OUTW;
    const float* bias;
    const h16_t* mask;
    const unsigned* maskbits; // CN/32 words per output pixel, bit c%32 of word c/32 = (layer input channel c > 0) — 16x fewer mask bytes than `mask`;
                             // staged through LDS with the band, so the multiply loop issues no global loads (the next band's prefetch stays in flight)
    int relu;
    int dbg;                 // bench ablation: bit 1 = skip the MFMA/epilogue phase, bit 2 = skip the global prefetch loads
    int Nf, RB, nbands, LW, LR;
    int LP;                  // LDS row pitch of the staged band in pixels (>= LW; a multiple of the input stride)
    int* work_ctr;           // optional zero-initialised device counter: items beyond the first round are CLAIMED (atomicAdd) instead of strided,
                             // so a workgroup that starts late (CUs held by an overlapped RCCL collective) takes less work instead of
                             // doubling the kernel's time
    unsigned* bits_out;      // forward, CN == 64, relu: also emit the ReLU bitmask of the output, 2 words per pixel (bit c of word c/32 = out channel c > 0)
                             // = what the dgrad of this layer's consumer reads as `maskbits` instead of the 16-bit activations (64x fewer bytes)
    int FPB;                 // frames per band (set by launch_conv_tile; > 1 only with nbands == 1): small frames (the gripper camera's 9x9 / 20x20
                             // maps) are stacked FPB to a band so that a band feeds all 8 waves and the per-band barriers / staging are paid once per
                             // FPB frames.  Forward: the frames are contiguous in memory, the band is simply FPB*IMH rows and the output rows that
                             // straddle two frames are computed and dropped.  Dgrad: the band is a VIRTUAL stack with VPI = IMH + TA - 1 rows per
                             // frame (IMH real rows + the zero rows the full correlation needs between frames), so no computed row is wasted.
    int VPI, VPO;            // rows per frame of the stacked band in window-row space (VPI) and in class-output-row space (VPO)
    int MB;                  // conv_reg.h (data-gradient form): bytes of one LDS region holding a band's ReLU bit words
    const h16_t* zeros;      // conv_reg.h data-gradient form: >= 16 zero bytes in device memory (source of the staged zero border)
    h16_t* dump;             // conv_reg.h pipelined-epilogue form (EPI): >= 4 KB of device scratch that receives the stores of pixels which must not be
                             // written (the code between the k-steps of the multiply loop is branch-free); never read
};

// ds_read_b128 is serviced in four fixed 16-lane groups, each mixing lanes of two k-chunk groups g (MI355X_MICROARCH.md §LDS):
// a fragment read (lane = (row li, chunk g), row pitch P 16-byte slots) is conflict-free iff P = 2 (mod 4); any other pitch,
// or rows that are not equidistant, costs 2x.  Hence (a) the slot pitches below and (b) m-tiles index the band by its LDS
// pitch (pixels pi = ri*Q + j, j < Q = LP/SI, columns j >= NJ are computed and discarded) so the 16 pixels of a fragment are
// always equidistant in LDS — no row-wrap jumps.
template <int CK, int CN, int TA, int TB, int SI, int OS, bool REV>
struct ConvTileCfg {
    static constexpr int pad_slots(int base, int mul) {
        for (int e = 1; e < 8; ++e)
            if (((base + e) * mul) % 4 == 2) return e;
        return 1;
    }
    static constexpr int CH = CK / 8;                          // 16-byte chunks per pixel
    static constexpr int XS = (CH + pad_slots(CH, SI)) * 16;   // LDS bytes per staged pixel
    static constexpr int WROW = TA * TB * CK * 2;
    static constexpr int WS = WROW + pad_slots(WROW / 16, 1) * 16;   // LDS bytes per weight row
    static constexpr int NCLS = OS * OS;
    static constexpr int NT = CN / 16;
    static constexpr int PF = 10;                              // prefetch registers (uint4) per thread: bands of up to 5 120 16-byte chunks (only the slots a band shape uses issue loads)
    static constexpr int MT = (CN / 16 <= 2) ? 4 : 2;          // m-tiles (16 pixels) per wave pass: LDS reads per MFMA = (MT+NT)/(MT*NT)
    static constexpr size_t w_bytes() { return (size_t)NCLS * CN * WS; }
    // the band is stored as SI row planes (window row wr -> plane wr % SI, row wr / SI) so that the pixels of consecutive OUTPUT
    // rows are LP apart for every tap: pixel(pi, tap) = pi*SI + plane/row/col offset of the tap
    static constexpr int WPP = (CN + 31) / 32;                 // mask words per output pixel
    static constexpr int MAXMW = 2048;                         // mask words staged per band (one 16-byte register per thread)
    static constexpr size_t band_bytes(int LR, int LP) { return ((size_t)(SI * ((LR + SI - 1) / SI) * LP) * XS + 15) / 16 * 16; }
    static constexpr size_t lds_bytes(int LR, int LP, bool with_mask = true) { return w_bytes() + band_bytes(LR, LP) + CN * 4 + (with_mask ? MAXMW * 4 : 0) + 16; }
};

#ifdef HULC_CT_STAMPS     // tools/ct_stamps.hip only: shader-clock stamps of the phases of every band (never defined in the library build)
__device__ unsigned long long g_ct_stamps[256 * 8 * 64 * 8];     // [workgroup][wave][band iteration][phase]
#define CTSTAMP(n) do { if (lane == 0 && iter < 64) g_ct_stamps[((blockIdx.x * 8 + wave) * 64 + iter) * 8 + (n)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CTSTAMP(n)
#endif

// x / d for 0 <= x < 2^16, 1 <= d <= 64 with inv = 1.f / d: (x + 0.5) / d is at least 0.5 / d away from an integer, far beyond the fp32
// rounding error of the product — 3 VALU instructions instead of the ~40 of a runtime integer division (the band staging and m-tile
// index maths below ran ~2 800 VALU cycles per band per SIMD on divisions alone: 1.2 us of a 7 us band)
DEVI int fast_div(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// NWV = waves per workgroup: 8, or 16 (four waves per SIMD, <= 128 VGPRs: while one wave sits in its epilogue / group setup three others can feed
// the matrix pipe; the prefetch registers halve with the thread count doubling)
template <int CK, int CN, int TA, int TB, int SI, int OS, bool REV, int NWV = 8>
__global__ void __launch_bounds__(NWV * 64) conv_tile_kernel(ConvTileP p) {
    using C = ConvTileCfg<CK, CN, TA, TB, SI, OS, REV>;
    constexpr int NTH = NWV * 64, PF = C::PF * 8 / NWV;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* wl = (lds_char*)smem;
    lds_char* xl = wl + C::NCLS * CN * C::WS;
    lds_char* bl = xl + C::band_bytes(p.LR, p.LP);              // bias[CN] fp32
    lds_char* ml = bl + CN * 4;                                 // mask words of the current band
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    // ---- weights -> LDS (once).  LDS row nn*16 + t of a class holds channel (t>>2)*4*NT + nn*4 + (t&3): after the MFMA lane
    // (pixel li, group g) then owns the 4*NT CONSECUTIVE channels g*4*NT.. of its pixel (16/32-byte epilogue accesses)
    {
        constexpr int ROWCH = C::WROW / 16;                     // 16-byte chunks per weight row
        for (int i = tid; i < C::NCLS * CN * ROWCH; i += NTH) {
            const int r = i / ROWCH, c = i % ROWCH;
            const int cls = r / CN, rr = r % CN, nn = rr >> 4, t = rr & 15;
            const int ch = (t >> 2) * 4 * C::NT + nn * 4 + (t & 3);
            *(lds_u32x4*)(wl + r * C::WS + c * 16) = *reinterpret_cast<const u32x4_t*>(p.w + (long long)(cls * CN + ch) * (TA * TB * CK) + c * 8);
        }
        if (tid < CN) *(__attribute__((address_space(3))) float*)(bl + tid * 4) = p.bias ? p.bias[tid] : 0.f;
    }
    const bool multi = p.FPB > 1;
    const int nitems = multi ? (p.Nf + p.FPB - 1) / p.FPB : p.Nf * p.nbands;
    const int wchunks = p.LR * p.LW * C::CH;
    const int Q = p.LP / SI;                                    // m-index pitch (band pixels per output row)
    const int PLR = (p.LR + SI - 1) / SI;                       // rows per LDS row plane
    const float invLW = 1.f / (float)p.LW, invQ = 1.f / (float)Q;
    const int nslots = (wchunks + NTH - 1) / NTH;                    // prefetch registers this band shape uses (uniform): the others issue NO load —
                                                                // every wave-level load costs the CU's address unit 16 cycles whether its data is used or not
    // ---- prefetch of a band into registers: UNCONDITIONAL loads from clamped addresses; the zero border is applied when the band is
    // committed to LDS (pin bit k), and the multiply loop below issues no global load at all — so nothing forces a wait on these loads
    // before the MFMAs and the next band's HBM latency hides under the current band's multiply phase.  Only the slots the band shape
    // uses issue a load (a wave-level 16-byte load occupies the CU's address unit for 16 cycles whether its data is used or not), and
    // the band-invariant part of every slot's address — window row, clamped column offset, column-in-range — is computed once
    // (pk[k]); per band a slot costs a row clamp and one multiply-add (tools/ct_stamps.hip: this phase was 2 600 of a band's 13 800
    // cycles with the matrix pipes idle; spreading the loads over the MFMA loop instead made the loop slower by as much).
    u32x4_t pf[PF], pfm;
    unsigned pin = 0;
    int mwords = 0;                                             // mask words of the prefetched band
    typedef u32x4_t __attribute__((aligned(4))) u32x4_a4;
    unsigned pk[PF];                                         // bit 31: column inside the image; bit 30: (stacked band) a zero row between frames;
                                                                // bits 16..23: window row (stacked band: row of the stacked frames in memory); bits 0..15: (clamped col * CK + chunk * 8) / 8
    {
        const int clo = REV ? -(TB - 1) : 0;
        const float invVPI = 1.f / (float)max(p.VPI, 1);
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int qc = min(tid + k * NTH, wchunks - 1);
            const int pix = qc / C::CH, c = qc % C::CH;
            const int wr = fast_div(pix, invLW), wc = pix - wr * p.LW;
            const int ic = clo + wc;
            const int icc = min(max(ic, 0), p.IMW - 1);
            unsigned row = (unsigned)wr, gap = 0u;
            if (multi && REV) {                                 // window row -> (frame of the band, row of that frame); rows >= IMH of a frame's VPI are zeros
                const int v = wr - (TA - 1);
                const int ff = v < 0 ? 0 : fast_div(v, invVPI), r = v - ff * p.VPI;
                gap = (v < 0 || r >= p.IMH) ? 0x40000000u : 0u;
                row = (unsigned)(ff * p.IMH + min(max(r, 0), p.IMH - 1));
            }
            pk[k] = ((ic >= 0 && ic < p.IMW) ? 0x80000000u : 0u) | gap | (row << 16) | (unsigned)(icc * C::CH + c);
        }
    }
    const int rowel = p.IMW * CK;                               // elements per image row
    auto prefetch = [&](int item) {
        if (p.dbg & 4) return;
        const int f = multi ? item * p.FPB : item / p.nbands, b = multi ? 0 : item % p.nbands;
        const int nfr = multi ? min(p.FPB, p.Nf - f) : 1;       // frames of this band (the last stacked band may be short: its missing frames stage as zeros)
        const int i0 = b * p.RB;
        const int rlo = multi ? 0 : (REV ? i0 - (TA - 1) : i0 * SI);
        const int imh = nfr * p.IMH;
        const h16_t* base = p.img + (long long)f * p.IMH * rowel;
        pin = 0;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            if (k >= nslots) break;
            const int ir = rlo + (int)((pk[k] >> 16) & 0xffu);
            const int irc = min(max(ir, 0), imh - 1);
            pf[k] = *reinterpret_cast<const u32x4_t*>(base + irc * rowel + (int)(pk[k] & 0xffffu) * 8);   // always-valid address
            pin |= ((ir == irc && (pk[k] >> 30) == 2u) ? 1u : 0u) << k;
        }
        if (p.maskbits) {                                       // ReLU bitmask rows of the band's output rows [i0*OS, (i0+RB)*OS)  (stacked band: of its nfr whole frames)
            const int r0 = i0 * OS, r1 = multi ? nfr * p.OUTH : min((i0 + p.RB) * OS, p.OUTH);
            mwords = max(0, r1 - r0) * p.OUTW * C::WPP;
            const long long mb = ((long long)f * p.OUTH + r0) * p.OUTW * C::WPP;
            const int mo = min(tid * 4, max(mwords - 4, 0));    // clamped: the last thread(s) re-read valid words (their LDS slot is unused)
            pfm = *reinterpret_cast<const u32x4_a4*>(p.maskbits + mb + mo);
        }
    };
    // LDS byte offset of this thread's k-th staged chunk (band-invariant)
    int soff[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int q = min(tid + k * NTH, wchunks - 1);
        const int pix = q / C::CH, c = q % C::CH;
        const int wr = fast_div(pix, invLW), wc = pix - wr * p.LW;
        soff[k] = (((wr % SI) * PLR + wr / SI) * p.LP + wc) * C::XS + c * 16;
    }
    __shared__ int s_next[2];
    int item = blockIdx.x, iter = 0;
    if (item < nitems) prefetch(item);
    while (item < nitems) {
        if (p.work_ctr && tid == 0) s_next[iter & 1] = (int)gridDim.x + atomicAdd(p.work_ctr, 1);   // claimed early: its latency hides under the LDS write
        CTSTAMP(0);
        __syncthreads();                                        // previous band fully consumed (and weights visible)
        CTSTAMP(1);
#pragma unroll
        for (int k = 0; k < PF; ++k)
            if (k < nslots && tid + k * NTH < wchunks) *(lds_u32x4*)(xl + soff[k]) = ((pin >> k) & 1u) ? pf[k] : u32x4_t{0u, 0u, 0u, 0u};
        if (p.maskbits && tid * 4 < mwords) {
            if (tid * 4 + 4 <= mwords) *(lds_u32x4*)(ml + tid * 16) = pfm;
            else {                                              // ragged tail: the clamped load holds words [mwords-4, mwords)
                const int sh = tid * 4 - max(mwords - 4, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (e + sh < 4) *(__attribute__((address_space(3))) unsigned*)(ml + (tid * 4 + e) * 4) = pfm[e + sh];
            }
        }
        CTSTAMP(2);
        __syncthreads();
        CTSTAMP(3);
        const int cur = item;
        item = p.work_ctr ? s_next[iter & 1] : item + (int)gridDim.x;
        // Phase skew between the two waves of a SIMD (waves w and w + 4): the first half issues the next band's prefetch (address
        // arithmetic + loads: VALU / address-unit work) BEFORE its multiply phase, the second half after its first group's MFMAs.  Released
        // by the same barrier, the two would otherwise run every phase in lockstep — both computing addresses while the matrix pipe idles,
        // then both contending for it; skewed, one half's prefetch and epilogues run beside the other half's MFMAs.
        bool pref_pending = item < nitems;
        if (pref_pending && (wave < NWV / 2 || (p.dbg & 64))) { prefetch(item); pref_pending = false; }       // dbg bit 6: no skew (A/B)
        CTSTAMP(4);
        ++iter;
        const int f = multi ? cur * p.FPB : cur / p.nbands, b = multi ? 0 : cur % p.nbands;
        const int i0 = b * p.RB;
        // class rows of the band: one frame's (OUTH - ph) / OS, or — stacked band — VPO per frame but the last
        auto class_rows = [&](int ph) {
            const int NIf = (p.OUTH - ph + OS - 1) / OS;
            return multi ? (REV ? (p.FPB - 1) * p.VPO + NIf : p.RB) : NIf;
        };
        // work items of this band = (parity class, group of MT m-tiles), dealt round-robin to the 8 waves
        auto groups_of = [&](int cls) {
            const int RBe = max(0, min(p.RB, class_rows(cls / OS) - i0));
            return (((RBe * Q + 15) >> 4) + C::MT - 1) / C::MT;
        };
        int total_groups = 0;
#pragma unroll
        for (int c = 0; c < C::NCLS; ++c) total_groups += groups_of(c);
        if (p.dbg & 2) total_groups = 0;
#pragma unroll 1
        for (int wi = wave; wi < total_groups; wi += NWV) {
            int cls = 0, mybase = 0, base = 0;
#pragma unroll
            for (int c = 0; c < C::NCLS; ++c) {
                if (wi >= base) { cls = c; mybase = base; }
                base += groups_of(c);
            }
            const int ph = cls / OS, pw = cls % OS;
            const int NIf = (p.OUTH - ph + OS - 1) / OS;     // class rows of ONE frame
            const int NI = class_rows(ph), NJ = (p.OUTW - pw + OS - 1) / OS;
            const int RBe = min(p.RB, NI - i0);
            const float invVPO = 1.f / (float)max(p.VPO, 1);
            const int npi = RBe * Q;
            const int last = npi - Q + NJ - 1;                  // last valid pixel index: lanes beyond it are clamped (reads stay in the band)
            const int mt0 = (wi - mybase) * C::MT;
            const int pbase = REV ? (TA - 1) * p.LP + (TB - 1) : 0;
            int xoff[C::MT];
            int opix[C::MT];                                    // output pixel offset (elements / CN) or -1
            int moff[C::MT];                                    // LDS byte offset of the pixel's mask word for this lane's channels
#pragma unroll
            for (int mm = 0; mm < C::MT; ++mm) {
                const int pi = (mt0 + mm) * 16 + li;
                const int ri = fast_div(pi, invQ), j = pi - ri * Q;
                bool ok = pi < npi && j < NJ;
                int orow = (i0 + ri) * OS + ph, mrow = ri * OS + ph;      // output row in the frame / in the band's mask rows
                if (multi) {                                    // stacked band: class row ri = frame ff of the band, row rr; rows beyond the frame's own are dropped
                    const int ff = fast_div(ri, invVPO), rr = ri - ff * p.VPO;
                    ok = ok && rr < NIf && f + ff < p.Nf;
                    orow = mrow = ff * p.OUTH + rr * OS + ph;
                }
                xoff[mm] = (min(pi, last) * SI + pbase) * C::XS + g * 16;
                opix[mm] = ok ? (orow * p.OUTW + j * OS + pw) : -1;
                moff[mm] = ok ? ((mrow * p.OUTW + j * OS + pw) * C::WPP + (C::WPP == 2 ? (g >> 1) : 0)) * 4 : 0;
            }
            f32x4 acc[C::MT][C::NT];
#pragma unroll
            for (int mm = 0; mm < C::MT; ++mm)
#pragma unroll
                for (int nn = 0; nn < C::NT; ++nn) acc[mm][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
            constexpr int EW = C::NT / 2;                       // 16-byte words per pixel per lane (lane owns 4*NT consecutive channels)
            lds_char* wrow = wl + (cls * CN + li) * C::WS + g * 16;
            constexpr int KS = CK / 32, NS = TA * TB * KS;      // k-steps of 32; NS is even for every instantiation
            static_assert(NS % 2 == 0, "pipelined loop handles k-steps in pairs");
            h16x8_t xf0[C::MT], wf0[C::NT], xf1[C::MT], wf1[C::NT];
            auto frag_load = [&](h16x8_t (&xf)[C::MT], h16x8_t (&wf)[C::NT], int s) {
                const int tap = s / KS, ks = s % KS;
                const int ta = tap / TB, tb = tap % TB;
                const int toff = (REV ? -(ta * p.LP + tb) : (((ta % SI) * PLR + ta / SI) * p.LP + tb)) * C::XS + ks * 64;
#pragma unroll
                for (int mm = 0; mm < C::MT; ++mm) xf[mm] = *(__attribute__((address_space(3))) h16x8_t*)(xl + xoff[mm] + toff);
#pragma unroll
                for (int nn = 0; nn < C::NT; ++nn) wf[nn] = *(__attribute__((address_space(3))) h16x8_t*)(wrow + nn * 16 * C::WS + s * 64);
            };
            auto frag_mma = [&](h16x8_t (&xf)[C::MT], h16x8_t (&wf)[C::NT]) {
#pragma unroll
                for (int mm = 0; mm < C::MT; ++mm)
#pragma unroll
                    for (int nn = 0; nn < C::NT; ++nn) acc[mm][nn] = MFMA_16x16x32_H(wf[nn], xf[mm], acc[mm][nn], 0, 0, 0);
            };
            // software pipeline: the LDS reads of k-step s+1 are in flight while the MFMAs of k-step s issue
#ifdef HULC_CT_STAMPS
            --iter; CTSTAMP(5); ++iter;
#endif
            if constexpr (NWV > 8) {             // four waves per SIMD hide each other's LDS latency: one fragment set (24 VGPRs less)
#pragma unroll 1
                for (int s1 = (p.dbg & 16) ? NS : 0; s1 < NS; ++s1) { frag_load(xf0, wf0, s1); frag_mma(xf0, wf0); }
            } else {
            frag_load(xf0, wf0, 0);
#pragma unroll 1
            for (int s2 = (p.dbg & 16) ? NS : 0; s2 < NS; s2 += 2) {
                frag_load(xf1, wf1, s2 + 1);
                frag_mma(xf0, wf0);
                if (s2 + 2 < NS) frag_load(xf0, wf0, s2 + 2);
                frag_mma(xf1, wf1);
            }
            }
            if (pref_pending) { prefetch(item); pref_pending = false; }
#ifdef HULC_CT_STAMPS
            asm volatile("s_nop 0" :: "v"(acc[0][0][0]), "v"(acc[C::MT - 1][C::NT - 1][3]));      // the stamp below must follow the last MFMA's result
            --iter; CTSTAMP(6); ++iter;
#endif
            if ((p.dbg & 8) && acc[0][0][0] != 12345.678f) continue;
            // ---- epilogue: lane holds channels g*4*NT .. +4*NT-1 of pixel li of each m-tile
            float bb[NWV > 8 ? 1 : 4 * C::NT];      // 16 waves: the bias quads are read from LDS where they are added (16 VGPRs less)
            if constexpr (NWV <= 8) {
#pragma unroll
            for (int e = 0; e < C::NT; ++e) {
                const f32x4 t = *(__attribute__((address_space(3))) f32x4*)(bl + (g * 4 * C::NT + e * 4) * 4);
                bb[e * 4 + 0] = t[0]; bb[e * 4 + 1] = t[1]; bb[e * 4 + 2] = t[2]; bb[e * 4 + 3] = t[3];
            }
            }
#pragma unroll
            for (int mm = 0; mm < C::MT; ++mm) {
                unsigned obits = 0;                             // forward: ReLU bitmask of this lane's 4*NT channels
                unsigned mkw = 0;
                if (p.maskbits) mkw = *(__attribute__((address_space(3))) unsigned*)(ml + moff[mm]);
                const long long opx = (long long)f * p.OUTH * p.OUTW + max(opix[mm], 0);
                const long long obase = opx * CN + g * 4 * C::NT;
#pragma unroll
                for (int e = 0; e < EW; ++e) {
                    float v[8];
                    if constexpr (NWV > 8) {
                        const f32x4 t0 = *(__attribute__((address_space(3))) f32x4*)(bl + (g * 4 * C::NT + e * 8) * 4), t1 = *(__attribute__((address_space(3))) f32x4*)(bl + (g * 4 * C::NT + e * 8 + 4) * 4);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = acc[mm][e * 2 + (r >> 2)][r & 3] + (r < 4 ? t0[r & 3] : t1[r & 3]);
                    } else {
#pragma unroll
                    for (int r = 0; r < 8; ++r) v[r] = acc[mm][e * 2 + (r >> 2)][r & 3] + bb[e * 8 + r];
                    }
                    if (p.relu) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = fmaxf(v[r], 0.f);
                    }
                    if (p.maskbits) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = ((mkw >> ((g * 4 * C::NT + e * 8 + r) & 31)) & 1u) ? v[r] : 0.f;
                    } else if (p.mask) {                        // 16-bit mask values (per-kernel tests): loaded here, drains the prefetch
                        const u32x4_t mk = *reinterpret_cast<const u32x4_t*>(p.mask + obase + e * 8);
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = ((r & 1) ? h2f_hi(mk[r >> 1]) : h2f_lo(mk[r >> 1])) > 0.f ? v[r] : 0.f;
                    }
                    u32x4_t o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = pack2h(v[2 * r], v[2 * r + 1]);
                    if (opix[mm] >= 0) *reinterpret_cast<u32x4_t*>(p.out + obase + e * 8) = o;
                    if constexpr (!REV && CN == 64) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) obits |= (((o[r] & 0xffffu) ? 1u : 0u) | ((o[r] >> 16) ? 2u : 0u)) << (e * 8 + 2 * r);
                    }
                }
                if constexpr (!REV && CN == 64) {
                    if (p.bits_out) {                           // lanes (li, g = 0..3) of a pixel: g pairs form one 32-bit word
                        unsigned w = obits << ((g & 1) * 16);
                        w |= __shfl_xor(w, 16);
                        if (opix[mm] >= 0 && (g & 1) == 0) p.bits_out[opx * 2 + (g >> 1)] = w;
                    }
                }
            }
#ifdef HULC_CT_STAMPS
            --iter; CTSTAMP(7); ++iter;
#endif
        }
        if (pref_pending) prefetch(item);                       // a wave without a group in this band
    }
}

// host side: pick the band height (fewest 8-wave rounds), launch persistent workgroups. Returns false if the shape does not fit.
template <int CK, int CN, int TA, int TB, int SI, int OS, bool REV, int NWV>
static inline bool launch_conv_tile_nw(hipStream_t st, ConvTileP p) {
    using C = ConvTileCfg<CK, CN, TA, TB, SI, OS, REV>;
    const int NI = REV ? (p.OUTH + OS - 1) / OS : p.OUTH;          // class rows (class 0 is the largest)
    const int NJ = REV ? (p.OUTW + OS - 1) / OS : p.OUTW;
    p.LW = REV ? NJ + TB - 1 : p.IMW;
    p.LP = (p.LW + SI - 1) / SI * SI;
    const int Q = p.LP / SI;
    double best = 1e30;
    int best_nb = 0;
    for (int nb = 1; nb <= NI; ++nb) {
        const int RB = (NI + nb - 1) / nb;
        const int LR = REV ? RB + TA - 1 : (RB - 1) * SI + TA;
        if ((long long)LR * p.LW * C::CH > 512ll * C::PF || C::lds_bytes(LR, p.LP, p.maskbits != nullptr) > 160 * 1024 - 64) continue;   // 64 B left for the kernel's static __shared__ (work-claim slots)
        if (p.maskbits && (long long)RB * OS * p.OUTW * C::WPP > C::MAXMW) continue;                                   // the band's mask rows travel in one register per thread
        double cost = 0.25 * nb;                                   // per-band barrier / staging overhead, in units of one wave round
        for (int b = 0; b < nb; ++b) {
            int groups = 0;
            for (int cls = 0; cls < C::NCLS; ++cls) {
                const int NIc = REV ? (p.OUTH - cls / OS + OS - 1) / OS : NI;
                const int RBe = std::max(0, std::min(RB, NIc - b * RB));
                groups += (((RBe * Q + 15) >> 4) + C::MT - 1) / C::MT;
            }
            cost += (groups + NWV - 1) / NWV;
        }
        if (cost < best) { best = cost; best_nb = nb; }
        if (nb >= 8 && best_nb) break;
    }
    if (!best_nb) return false;
    p.nbands = best_nb;
    p.RB = (NI + best_nb - 1) / best_nb;
    p.LR = REV ? p.RB + TA - 1 : (p.RB - 1) * SI + TA;
    p.FPB = 1; p.VPI = p.VPO = 0;
    // ---- small frames (whole frame per band and still only a fraction of an 8-wave round): stack FPB frames to a band.  Cost of a launch =
    // bands per workgroup x (per-band overhead + 8-wave rounds of the band's groups); one frame per band leaves most waves without a group
    // (7x7 outputs = 2 groups) and pays the two barriers + staging per frame (tools/ct_stamps.hip: ~2.5 us against a ~5.5 us full round).
    static const int fpb_env = HULC_SWITCH("HULC_CT_FPB", -1);      // A/B: 1 = off, n = force n frames where it fits
    if (best_nb == 1 && fpb_env != 1 && p.Nf > 1 && (REV || p.IMH % SI == 0)) {
        const int vpi = REV ? p.IMH + TA - 1 : p.IMH, vpo = REV ? vpi : p.IMH / SI;
        auto band_cost = [&](int fpb, int& RBo, int& LRo) -> double {
            RBo = REV ? fpb * vpo : (fpb * p.IMH - TA) / SI + 1;
            LRo = REV ? RBo + TA - 1 : fpb * p.IMH;
            if (fpb * p.IMH > 255 || LRo > 255) return -1.0;
            if ((long long)LRo * p.LW * C::CH > 512ll * C::PF || C::lds_bytes(LRo, p.LP, p.maskbits != nullptr) > 160 * 1024 - 64) return -1.0;
            if (p.maskbits && (long long)fpb * p.OUTH * p.OUTW * C::WPP > C::MAXMW) return -1.0;
            int groups = 0;
            for (int cls = 0; cls < C::NCLS; ++cls) {
                const int NIf = REV ? (p.OUTH - cls / OS + OS - 1) / OS : NI;
                const int rows = fpb == 1 ? NIf : (REV ? (fpb - 1) * vpo + NIf : RBo);
                groups += (((rows * Q + 15) >> 4) + C::MT - 1) / C::MT;
            }
            const int items = (p.Nf + fpb - 1) / fpb, wgs = std::min(items, 256);
            return (double)((items + wgs - 1) / wgs) * (0.45 + (groups + NWV - 1) / NWV);
        };
        int RB1, LR1, bestf = 1;
        double bc = band_cost(1, RB1, LR1);
        for (int fpb = 2; fpb <= 32; ++fpb) {
            int RBf, LRf;
            const double c = band_cost(fpb, RBf, LRf);
            if (c < 0) break;
            if ((fpb_env > 1 && fpb <= fpb_env) || (fpb_env < 0 && c < bc - 1e-9)) { bc = c; bestf = fpb; p.RB = RBf; p.LR = LRf; }
        }
        if (bestf > 1) { p.FPB = bestf; p.VPI = vpi; p.VPO = vpo; }
    }
    const size_t lds = C::lds_bytes(p.LR, p.LP, p.maskbits != nullptr);      // without a bitmask the 8 KB mask region is not allocated: conv3's forward then holds a WHOLE frame per band
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_tile_kernel<CK, CN, TA, TB, SI, OS, REV, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        attr_set = true;
    }
    const int items = p.FPB > 1 ? (p.Nf + p.FPB - 1) / p.FPB : p.Nf * p.nbands;
    const int grid = items < 256 ? items : 256;
    hipLaunchKernelGGL((conv_tile_kernel<CK, CN, TA, TB, SI, OS, REV, NWV>), dim3(grid), dim3(NWV * 64), lds, st, p);
    return true;
}
template <int CK, int CN, int TA, int TB, int SI, int OS, bool REV>
static inline bool launch_conv_tile(hipStream_t st, ConvTileP p) {
    // 16 waves (four per SIMD, 128 VGPRs, one fragment set instead of the software-pipelined pair): the dgrad kernels fit (0 / 31 spilled registers,
    // none inside the multiply loop) and gain 5-11 % — while one wave sits in its epilogue or group setup three others feed the matrix pipe.  The
    // forward kernels do not fit: the compiler spills the band prefetch registers, i.e. waits for the loads right where they are issued
    // (load phase alone 28 -> 76 us for conv3); they stay at 8 waves.  HULC_CT_NW=8 / 16 forces one width for every kernel (A/B).
    static const int nw = HULC_SWITCH("HULC_CT_NW", 0);
    if constexpr (REV) { if (nw != 8) return launch_conv_tile_nw<CK, CN, TA, TB, SI, OS, REV, 16>(st, p); }
    else { if (nw == 16) return launch_conv_tile_nw<CK, CN, TA, TB, SI, OS, REV, 16>(st, p); }
    return launch_conv_tile_nw<CK, CN, TA, TB, SI, OS, REV, 8>(st, p);
}

// the multiply + epilogue of one staged band (shared by the fp32 / register-staged kernel and the uint8 LDS-DMA kernel below)
DEVI void conv1_fwd_band_tiles(lds_char* ximg, const h16x8_t (&wf)[6][2], const int (&rowsel)[6], const float4 (&bb)[2], h16_t* __restrict__ out,
                               unsigned* __restrict__ maskbits, int f, int oh0, int R, int OH, int OW, int XRS, int dbg, int wave, int g, int li, float osc = 1.f) {
        // osc: Conv1Src::fold — the staged operand is the raw byte value, out = relu(osc * acc + bias_fold)
        const int RBe = min(R, OH - oh0);
        const int npix = RBe * OW;
        const int ntm = (dbg & 2) ? 0 : (npix + 15) >> 4;
        for (int mt = wave * 2; mt < ntm; mt += 8) {                      // two m-tiles per pass: independent MFMA chains interleave
            bool ok[2]; int rr[2], oww[2];
            lds_char* xb[2];
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int pi = (mt + mm) * 16 + li;
                ok[mm] = pi < npix;
                const int pc = ok[mm] ? pi : npix - 1;
                rr[mm] = pc / OW; oww[mm] = pc - rr[mm] * OW;
                xb[mm] = ximg + rr[mm] * 4 * XRS + oww[mm] * 8;
            }
            f32x4 acc[2][2];
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[mm][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 6; ++ks) {
                typedef short s16x4v __attribute__((ext_vector_type(4)));
                typedef short s16x8v __attribute__((ext_vector_type(8)));
                h16x8_t xf[2];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    const s16x4v lo = *(__attribute__((address_space(3))) s16x4v*)(xb[mm] + rowsel[ks]);
                    const s16x4v hi = *(__attribute__((address_space(3))) s16x4v*)(xb[mm] + rowsel[ks] + 8);
                    xf[mm] = __builtin_bit_cast(h16x8_t, (s16x8v)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
                }
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc[mm][ct] = MFMA_16x16x32_H(wf[ks][ct], xf[mm], acc[mm][ct], 0, 0, 0);
            }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                if (!ok[mm]) continue;
                const long long opix = ((long long)f * OH + oh0 + rr[mm]) * OW + oww[mm];
                const long long obase = opix * 32;
                unsigned bits = 0;                                       // ReLU mask of this lane's 8 channels (what conv2's dgrad needs of a1)
                unsigned ow[4];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const int cn0 = g * 8 + ct * 4;
                    const float v0 = fmaxf(fmaf(acc[mm][ct][0], osc, bb[ct].x), 0.f), v1 = fmaxf(fmaf(acc[mm][ct][1], osc, bb[ct].y), 0.f);
                    const float v2 = fmaxf(fmaf(acc[mm][ct][2], osc, bb[ct].z), 0.f), v3 = fmaxf(fmaf(acc[mm][ct][3], osc, bb[ct].w), 0.f);
                    const unsigned ox = pack2h(v0, v1), oy = pack2h(v2, v3);
                    ow[ct * 2] = ox; ow[ct * 2 + 1] = oy;
                    const unsigned nz = ((ox & 0xffffu) ? 1u : 0u) | ((ox >> 16) ? 2u : 0u) | ((oy & 0xffffu) ? 4u : 0u) | ((oy >> 16) ? 8u : 0u);
                    bits |= nz << cn0;
                }
                *reinterpret_cast<uint4*>(out + obase + g * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
                if (maskbits) {                                          // the four lanes of a pixel (g = 0..3) are all active here: OR their bytes
                    bits |= __shfl_xor(bits, 16);
                    bits |= __shfl_xor(bits, 32);
                    if (g == 0) maskbits[opix] = bits;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// conv1 forward (8x8 stride 4, 3 -> 32, + bias + ReLU) straight from the fp32 NCHW boundary frames.
// The band of input rows is converted to bf16 while staged ([c][row][iw] image, each input byte read from HBM once instead
// of 4x); the 32x192 weight matrix lives in registers as MFMA A fragments (12 per lane); an image fragment (16 output pixels
// x 8 kw of one (c,kh) row) is two 8-byte LDS reads.  K order = (c, kh, kw) = torch's weight order.
// ---------------------------------------------------------------------------------------------------------------------
template <int MINW, bool REGCONV = false>
__global__ void __launch_bounds__(256, MINW) conv1_fwd_kernel(Conv1Src X, const h16_t* __restrict__ W, const float* __restrict__ bias,
                                                           h16_t* __restrict__ out, int Nf, int IH, int IW, int OH, int OW, int R, int nbands, int dbg,
                                                           unsigned* __restrict__ maskbits, float* __restrict__ zero8a = nullptr, float* __restrict__ zero8b = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    lds_char* ximg = (lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the step's loss accumulators (8 floats each) are cleared by the forward's FIRST kernel instead of by two 32-byte memsets (~5 us of launch each)
    if (blockIdx.x == 0 && tid < 8) { if (zero8a) zero8a[tid] = 0.f; if (zero8b) zero8b[tid] = 0.f; }
    const int g = lane >> 4, li = lane & 15;
    const int XR = (R - 1) * 4 + 8;
    const int XRS = IW * 2 + 16;
    const int W4 = IW >> 2;
    h16x8_t wf[6][2];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) wf[ks][ct] = *reinterpret_cast<const h16x8_t*>(W + ((li >> 2) * 8 + ct * 4 + (li & 3)) * 192 + ks * 32 + g * 8);
    // ^ A-operand row m of tile ct carries output channel (m >> 2) * 8 + ct * 4 + (m & 3): with the 16x16 C/D map (row = 4 g + r) lane group g then owns
    //   the 8 CONSECUTIVE channels 8 g .. 8 g + 7 of its pixel — one 16-byte store per lane and 1 KB contiguous per wave store (two 8-byte stores to
    //   interleaved 32-byte halves of every pixel before)
    // (c, kh) row of this lane group for each k-step
    int rowsel[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) { const int ck = ks * 4 + g; rowsel[ks] = ((ck >> 3) * XR + (ck & 7)) * XRS; }
    const float4 bb[2] = {*reinterpret_cast<const float4*>(bias + g * 8), *reinterpret_cast<const float4*>(bias + g * 8 + 4)};
    const int nitems = Nf * nbands;
    // frame-wise (fw, experiment knob HULC_C1_FW=1): a workgroup walks the bands of one frame back to back so that the rows two bands share come
    // from L2 — what cut conv1's weight gradient by 25 % measured 0.4 % SLOWER here (4 resident workgroups per CU already re-read those rows
    // from the XCD's L2 within microseconds; the frame-wise order only removes the interleaving of their phases): default item-wise
    const bool fw = Nf >= (int)gridDim.x && !(dbg & 32);
    for (int it = blockIdx.x; fw ? (it < Nf) : (it < nitems); it += gridDim.x)
    for (int bnd = 0; bnd < (fw ? nbands : 1); ++bnd) {
        const int item = fw ? it * nbands + bnd : it;
        const int f = item / nbands, b = item % nbands;
        const int oh0 = b * R;
        const int ih0 = oh0 * 4;
        const int rows = min(XR, IH - ih0);
        __syncthreads();
        if (!(dbg & 4)) conv1_stage_band<REGCONV>(X, f, ih0, rows, IH, IW, ximg, XR, XRS, tid, ximg + 3 * XR * XRS + 64);      // REGCONV: uint8 converted from registers (no raw rows)
        __syncthreads();
        conv1_fwd_band_tiles(ximg, wf, rowsel, bb, out, maskbits, f, oh0, R, OH, OW, XRS, dbg, wave, g, li, (X.u8 && X.fold) ? CONV1_FOLD_SCALE : 1.f);
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// conv1 forward for the uint8 (.., H, W, C) boundary with the NEXT band's raw rows copied global -> LDS by the DMA path (no registers, nothing
// waits for them) while the current band is multiplied.  conv1_fwd_kernel's uint8 branch loads, waits, converts, multiplies per band and relies on
// the other three resident workgroups to cover the load latency: 216 us for the static camera against ~105 us of HBM time (561 MB).
//   * raw layout = conv1_stage_band's: row r at raw + r * RP behind a margin of LM bytes, i.e. 16-byte slots [3 margin | 38 data | 3 margin];
//     slot 3 + c of a row receives source bytes [16 c, 16 c + 16) of the (row-clamped, RandomShiftsAug replicate pad) source row — the last
//     data slot carries 8 bytes of the following row, which nothing reads; at the very end of the frame buffer that lane is switched off and
//     the 8 bytes are patched by one thread.  global_load_lds_dwordx4 accepts 4-byte-aligned sources and writes lane i at M0 + 16 i
//     (tools/dma96_probe.hip; dwordx3 strides 16 too), inactive lanes write nothing.
//   * the replicated column margins (needed only when the frame's column shift is not zero) come from the two edge pixels of every row, loaded
//     into one register at prefetch time and written before the barrier that publishes the raw rows.
// Measured on 2048 static-camera frames (tools/time_conv1_fwd.py): 205 -> 177 us (212 -> 182 with shifts); DMA alone 48 us, + conversion 87, + multiply
// 130.  The same structure with 8 waves, two workgroups per CU and 9-row bands: 177 us — no better; for the fp32 boundary (raw fp32 band through
// the DMA path, LDS -> LDS conversion): 283 against conv1_fwd_kernel's 270 us, both at the mixed read/write streaming floor of ~245 us (tools/mixbench.hip).
// ---------------------------------------------------------------------------------------------------------------------
DEVI void c1_lds_dma16(const void* src, lds_char* dst) {
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
__global__ void __launch_bounds__(256, 4) conv1_fwd_u8dma_kernel(Conv1Src X, const h16_t* __restrict__ W, const float* __restrict__ bias, h16_t* __restrict__ out,
                                                                 int Nf, int IH, int IW, int OH, int OW, int R, int nbands, int dbg, unsigned* __restrict__ maskbits,
                                                                 float* __restrict__ zero8a, float* __restrict__ zero8b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    lds_char* ximg = (lds_char*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x == 0 && tid < 8) { if (zero8a) zero8a[tid] = 0.f; if (zero8b) zero8b[tid] = 0.f; }
    const int g = lane >> 4, li = lane & 15;
    const int XR = (R - 1) * 4 + 8;
    const int XRS = IW * 2 + 16;
    const int W4 = IW >> 2;
    lds_char* raw = ximg + 3 * XR * XRS + 64;
    const int RB = IW * 3, RP = conv1_raw_pitch16(IW);
    constexpr int LM = CONV1_RAW_MARGIN * 3;
    const int SPR = RP >> 4, DS = (RB + 15) >> 4;                 // 16-byte slots per raw row, data slots per row (the last one may be partial)
    const int tailb = RB & 15;                                    // bytes of the last data slot that belong to the row (0: all 16)
    h16x8_t wf[6][2];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) wf[ks][ct] = *reinterpret_cast<const h16x8_t*>(W + ((li >> 2) * 8 + ct * 4 + (li & 3)) * 192 + ks * 32 + g * 8);
    int rowsel[6];
#pragma unroll
    for (int ks = 0; ks < 6; ++ks) { const int ck = ks * 4 + g; rowsel[ks] = ((ck >> 3) * XR + (ck & 7)) * XRS; }
    const float4 bb[2] = {*reinterpret_cast<const float4*>(bias + g * 8), *reinterpret_cast<const float4*>(bias + g * 8 + 4)};
    const unsigned char* Xb = reinterpret_cast<const unsigned char*>(X.X);
    const long long total = X.frames_total(Nf) * IH * RB;         // bytes of the frame buffer (the whole store when the windows are gathered by index)
    const int nitems = Nf * nbands;
    const float inv_spr = 1.f / (float)SPR;
    unsigned pe = 0;                                              // edge pixel of (row tid >> 1, side tid & 1) of the band in flight
    int pdx = 0, prows = 0;
    auto prefetch = [&](int item) {
        const int f = item / nbands, ih0 = (item % nbands) * R * 4;
        const int rows = min(XR, IH - ih0);
        int dy = 0;
        pdx = 0; prows = rows;
        X.offsets(f, pdx, dy);
        const long long fb = X.frame(f) * IH * RB;
        const int nslot = rows * SPR;
        for (int n0 = wave * 64; n0 < nslot; n0 += 256) {
            const int n = n0 + lane;
            const int r = (int)(((float)n + 0.5f) * inv_spr), sl = n - r * SPR;       // exact for n < 2^22 / SPR
            const int c = sl - (LM >> 4);
            const long long so = fb + (long long)min(max(ih0 + r + dy, 0), IH - 1) * RB + c * 16;
            if (n < nslot && c >= 0 && c < DS && so + 16 <= total) c1_lds_dma16(Xb + so, raw + n0 * 16);
        }
        if (tailb && fb + (long long)IH * RB == total && tid == 255) {                // the frame buffer's last row may be part of this band: its partial last slot
            for (int r = 0; r < rows; ++r)
                if (min(max(ih0 + r + dy, 0), IH - 1) == IH - 1) {
                    const unsigned char* sp = Xb + total - tailb;
                    for (int k = 0; k < tailb; k += 4) *(__attribute__((address_space(3))) unsigned*)(raw + r * RP + LM + (DS - 1) * 16 + k) = *reinterpret_cast<const unsigned*>(sp + k);
                }
        }
        if (pdx != 0 && tid < 2 * rows) {
            const int rr = tid >> 1, side = tid & 1;
            pe = *reinterpret_cast<const unsigned*>(Xb + fb + (long long)min(max(ih0 + rr + dy, 0), IH - 1) * RB + (side ? RB - 4 : 0));
        }
    };
    int item = blockIdx.x;
    if (item < nitems) prefetch(item);
    const Step256 sq(W4);
    const RowCol q0{tid / W4, tid % W4};
    for (; item < nitems; item += gridDim.x) {
        const int f = item / nbands, b = item % nbands;
        const int oh0 = b * R;
        const int rows = prows, dx = pdx;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this band's raw rows (and edge pixels) have landed — THIS wave's share of them
        // Round 6 fix: a row's last data slot carries 8 bytes of the FOLLOWING row, and those land on the first 8 bytes of the row's RIGHT margin.  The margins were
        // filled right here, behind this wave's own vmcnt(0) only: a slower wave's DMA piece could land AFTER the fill and put the neighbour row's bytes back into the
        // margin — with a positive column shift ~2.5e-5 of a launch's outputs (the three rightmost output columns of some rows) came out wrong by O(0.1), differently
        // from run to run (tools/time_conv1_u8reg.py found it: two runs of this kernel disagreed with each other; torch agrees with neither before the fix).
        // Every wave's pieces must have landed before a margin is written: one more barrier, only for frames that read the margins.
        if (dx != 0) __syncthreads();
        if (dx != 0 && tid < 2 * rows) {                          // replicate margins (F.pad(..., "replicate"))
            const int rr = tid >> 1, side = tid & 1;
            const unsigned px = side ? (pe >> 8) : (pe & 0xffffffu);
            lds_char* dst = raw + rr * RP + (side ? LM + RB : 0);
            for (int k = 0; k < CONV1_RAW_MARGIN; ++k) {
                dst[k * 3 + 0] = (char)(px & 0xff); dst[k * 3 + 1] = (char)((px >> 8) & 0xff); dst[k * 3 + 2] = (char)((px >> 16) & 0xff);
            }
        }
        __syncthreads();                                          // raw complete for every wave; the previous band's multiply is over (ximg free)
        if (!(dbg & 4)) {
            const float sc = X.fold ? 1.f : 2.f / 255.f, of = X.fold ? 0.f : -1.f;      // fold: the exact value of the byte (Conv1Src::fold)
            for (RowCol p = q0; p.r < rows; sq.adv(p)) {
                const int o = p.r * RP + LM + (p.c * 4 + dx) * 3;
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
        __syncthreads();                                          // ximg complete, raw consumed
        if (item + (int)gridDim.x < nitems) prefetch(item + (int)gridDim.x);         // lands during the multiply below
        conv1_fwd_band_tiles(ximg, wf, rowsel, bb, out, maskbits, f, oh0, R, OH, OW, XRS, dbg, wave, g, li, X.fold ? CONV1_FOLD_SCALE : 1.f);
    }
}
static inline void launch_conv1_fwd(hipStream_t st, const Conv1Src& X, const h16_t* W, const float* bias, h16_t* out, int Nf, int IH, int IW, int OH, int OW, int dbg = 0,
                                    unsigned* maskbits = nullptr, float* zero8a = nullptr, float* zero8b = nullptr) {
    // (tried: an 8-wave, 2-workgroups-per-CU version with the next band prefetched in registers like conv1_wgrad_tr2_kernel — 4.355 vs 4.341
    //  ms/step on one box: with 4 resident workgroups per CU the staging of one already overlaps the MFMAs of the others; not kept)
    // uint8, round 6: HULC_C1_U8REG (or dbg bit 8) = the register-staged kernel with the conversion from 16-byte windows (conv1_stage_band regconv: no raw rows in LDS)
    static const int u8reg = HULC_SWITCH("HULC_C1_U8REG", 0);
    const bool regconv = X.u8 && (u8reg || (dbg & 256)) && (IW % 4) == 0 && IW >= 8 && ((uintptr_t)X.X & 3) == 0;
    if (regconv) dbg |= 256; else dbg &= ~256;
    auto lds_of = [&](int R) { const int XR = (R - 1) * 4 + 8; return (size_t)3 * XR * (IW * 2 + 16) + 64 + ((X.u8 && !regconv) ? (size_t)XR * conv1_raw_pitch16(IW) + 16 : 0); };   // + raw uint8 rows
    static const int lds_kb = HULC_SWITCH("HULC_C1_LDS", 39);   // 4 workgroups per CU: one stages while others multiply (255 vs 299 us at 2 per CU)
    static const int max_wg = HULC_SWITCH("HULC_C1_WG", 1024);
    int R = OH;
    while (R > 1 && lds_of(R) > (size_t)lds_kb * 1024) --R;
    const int nbands = (OH + R - 1) / R;
    R = (OH + nbands - 1) / nbands;
    static const int fw_env = HULC_SWITCH("HULC_C1_FW", 0);    // same-box A/B: frame-wise 4.458 vs item-wise 4.440 ms/step -> off
    if (!fw_env) dbg |= 32;
    static const int occ = HULC_SWITCH("HULC_C1_OCC", 4);      // min waves per SIMD the register allocation targets: 128 VGPRs (5 spilled) lets all 4 workgroups of a CU be resident (133 -> only 3); A/B on one box: -0.8 % of the step
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv1_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)conv1_fwd_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int items = Nf * nbands;
    static const int u8dma = HULC_SWITCH("HULC_C1_U8DMA", 1);
    if (X.u8 && !regconv && u8dma && (IW * 3) % 4 == 0 && ((uintptr_t)X.X & 3) == 0) {
        static bool a2 = false;
        if (!a2) { hipFuncSetAttribute((const void*)conv1_fwd_u8dma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a2 = true; }
        hipLaunchKernelGGL(conv1_fwd_u8dma_kernel, dim3(items < max_wg ? items : max_wg), dim3(256), lds_of(R), st, X, W, bias, out, Nf, IH, IW, OH, OW, R, nbands, dbg, maskbits, zero8a, zero8b);
        return;
    }
    if (regconv) {       // uint8 cross-check path (tests, tools/time_conv1_u8reg.py): its own instance, so that the window code does not weigh on the fp32 kernel's registers
        static bool a3 = false;
        if (!a3) { hipFuncSetAttribute((const void*)conv1_fwd_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); a3 = true; }
        hipLaunchKernelGGL((conv1_fwd_kernel<4, true>), dim3(items < max_wg ? items : max_wg), dim3(256), lds_of(R), st, X, W, bias, out, Nf, IH, IW, OH, OW, R, nbands, dbg, maskbits, zero8a, zero8b);
        return;
    }
    if (occ >= 4) hipLaunchKernelGGL(conv1_fwd_kernel<4>, dim3(items < max_wg ? items : max_wg), dim3(256), lds_of(R), st, X, W, bias, out, Nf, IH, IW, OH, OW, R, nbands, dbg, maskbits, zero8a, zero8b);
    else hipLaunchKernelGGL(conv1_fwd_kernel<2>, dim3(items < max_wg ? items : max_wg), dim3(256), lds_of(R), st, X, W, bias, out, Nf, IH, IW, OH, OW, R, nbands, dbg, maskbits, zero8a, zero8b);
}

}  // namespace HULC_NS
