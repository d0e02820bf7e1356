// hulc_amd/csrc/conv_reg.h — forward conv2 / conv3 (NHWC 16-bit activations, 64 output channels) with the WEIGHTS IN REGISTERS.
//
// conv_tile.h keeps the packed weight matrix in LDS (64-74 KB of the CU's 160 KB) next to one register-staged band: every MFMA of its
// loop reads a weight fragment AND an image fragment from LDS (0.75 reads per MFMA at 2 x 4 fragments = 75 % of the matrix-pipe time),
// the band is staged through 40 prefetch registers and a commit phase, and two barriers per band put all eight waves in lockstep
// (profiles/r02_conv_tile_phase_stamps.txt: multiply loop 49 % of a band).  Round 3 turns the kernel around:
//
//   * v_mfma_f32_32x32x16: A = 32 output channels x 16 k held in REGISTERS for the whole launch (a wave owns one channel half: 32 x K
//     weights = 128 VGPRs for conv2's K = 512, 144 for conv3's K = 576), B = 32 pixels x 16 k: ONE ds_read_b128 per MFMA and per 16 k
//     (0.0625 LDS bytes per MAC against 0.094), and no LDS byte spent on weights;
//   * the freed LDS holds TWO bands (a whole 23 x 23 x 64 frame each for conv3; a third of a 49 x 49 x 32 frame for conv2): the next band
//     arrives by LDS-DMA (global_load_lds_dwordx4, no staging registers, no commit phase) while the current one is multiplied — one
//     barrier per band;
//   * stride-2 input is stored as 2 x 2 row / column parity planes, so the 32 pixels of a fragment are adjacent staged pixels for every
//     tap; the staged pixel pitch is an ODD number of 16-byte slots (5 / 9), which makes every ds_read_b128 lane group hit 16 distinct
//     slot residues (conflict-free: MI355X_MICROARCH.md §LDS — the groups {0-3,12-15,20-27}, ... cover all residues mod 16);
//   * the weight rows are permuted so that a lane ends up with 16 CONSECUTIVE output channels of its pixel (two 16-byte stores), and the
//     ReLU bitmask the next layer's data-gradient kernel reads is emitted with one cross-half shuffle.
//
// Reference arithmetic: nn.Conv2d(32, 64, 4, stride 2) + ReLU, nn.Conv2d(64, 64, 3, stride 1) + ReLU
// (hulc/models/perceptual_encoders/vision_network.py:38-45, vision_network_gripper.py:12-17); fp32 accumulation, bias added in fp32.
#pragma once
#include <type_traits>

#include "conv_tile.h"

namespace HULC_NS {

#ifdef HULC_HALF_F16
#define MFMA_32x32x16_H __builtin_amdgcn_mfma_f32_32x32x16_f16
#else
#define MFMA_32x32x16_H __builtin_amdgcn_mfma_f32_32x32x16_bf16
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifdef HULC_CR_STAMPS     // tools/cr_stamps.hip only: shader-clock stamps of a band's phases, every wave of workgroups 0..15 (never defined in the library build)
__device__ unsigned long long g_cr_stamps[16 * 8 * 32 * 8];      // [workgroup][wave][band iteration][stamp]
#define CRSTAMP(n) do { if (lane == 0 && blockIdx.x < 16 && crit < 32) g_cr_stamps[((blockIdx.x * 8 + wave) * 32 + crit) * 8 + (n)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CRSTAMP(n)
#endif

template <int CK, int TA, int TB, int SI>
struct ConvRegCfg {
    static constexpr int CH = CK / 8;                          // 16-byte chunks per pixel
    static constexpr int XSS = (CH & 1) ? CH : CH + 1;         // staged slots per pixel: odd
    static constexpr int XS = XSS * 16;
    static constexpr int K = TA * TB * CK, NS = K / 16, KS = CK / 16;
    static constexpr int NPL = SI * SI;                        // parity planes
    static constexpr int PIECES = 80;                          // 1 KB DMA pieces per band at most (80 / waves rounds per wave)
    static constexpr size_t band_bytes(int PLR, int PLC) { return ((size_t)NPL * PLR * PLC * XS + 1023) / 1024 * 1024; }   // whole 1 KB DMA pieces
    // nbuf band buffers + bias[64] + (data-gradient form) nbuf regions of `maskb` bytes: the ReLU bit words of a band's output pixels
    static constexpr size_t lds_bytes(int PLR, int PLC, int nbuf = 2, size_t maskb = 0) { return nbuf * (band_bytes(PLR, PLC) + maskb) + 256; }
};

// p.LR = staged rows of a band, p.RB = output rows of a band, p.LW = staged width (REV: IMW + 2 (TB - 1)), p.LP = plane columns (PLC),
// p.VPI = plane rows (PLR), p.FPB frames per band, p.VPO = output-row slots per frame of a stacked band.
//
// REV = false: forward.   out[i][j][cn] = relu( sum_{ta,tb,ck} img[i*SI+ta][j*SI+tb][ck] W[cn][(ta,tb,ck)] + bias[cn] )
// REV = true (SI = 1): data gradient of a stride-1 convolution = the same correlation over the ZERO-PADDED gradient image
//   out[i][j][cn] = mask * sum_{ta,tb,ck} P[i + TA-1-ta][j + TB-1-tb][ck] W[cn][(ta,tb,ck)],  P[r][c] = img[r-(TA-1)][c-(TB-1)] or 0;
//   the border cells are staged from a zero page (p.zeros) by the same DMA; stacked small frames share their TA-1 border rows (a frame
//   takes IMH + TA - 1 staged rows = exactly its OUTH output rows: no computed row is dropped).
//   OS = 2 (conv2, 4x4 stride 2): one such stride-1 correlation per output parity class (ph, pw) with TA = TB = 2 taps, 32 output channels and
//   its own weight slab W[(ph,pw)][cn][(ta,tb,ck)]; out[i*2+ph][j*2+pw][cn].  The eight waves are 4 classes x 2 pixel halves (a wave holds its
//   class's 32 x 256 weights: 64 VGPRs) instead of 2 channel halves x 4 pixel quarters.
//
// NWV = 8 (round 3): ONE 512-thread workgroup per CU, two band buffers, the next band's DMA under the current band's MFMAs.  All eight waves pass
//   through barrier -> multiply -> epilogue together, so the phases ADD (profiles/r03_conv_reg_ablation.txt: full = DMA + MFMA + epilogue).
// NWV = 4 (round 4): TWO co-resident 256-thread workgroups per CU (<= 256 VGPRs per wave, <= 80 KB of LDS each), each with ONE
//   band buffer: load band -> barrier -> multiply + epilogue -> barrier.  The two workgroups of a CU are independent, so they fall out of
//   phase: while one waits for its DMA or sits in its epilogue stores, the other one's waves own the matrix pipes (a SIMD hosts one wave of
//   each) — the overlap the single lockstep workgroup could not produce.  A wave takes twice the tiles per band (half the barriers per tile).
//
// PKR (round 5): the slot -> (staged row, column, chunk) decode of a wave's DMA rounds is the same for every band.  tools/cr_stamps.hip stamps the
//   phases of a band: a wave that issues a band's ten DMA rounds spends 3 000 - 7 000 cycles there — not waiting for the memory system but in the
//   ~50 VALU instructions of that decode per round (three reciprocal divisions, clamps, the 64-bit address), with the matrix pipe idle for that wave.
//   Where the weights leave registers (conv2: 128 / 64 weight registers) or the allocator finds them (conv3 forward at 8 waves), the decode is done
//   ONCE per launch (pk[]: 10 registers) and a round costs a row clamp and one multiply-add: conv2 forward 143 -> 135 us, conv3 forward 81 -> 77 us,
//   conv2 data gradient (2 x 4 waves) 147 -> 140 us on 2048 static frames (profiles/r05_conv_reg_forms.txt).  conv3's data gradient spills with it.
//
// EPI (round 5): the epilogue of a tile rides in the multiply loop of the NEXT tile.  Until now a wave alternated [multiply loop of a tile pair]
//   [epilogue of the pair: ~110 VALU + 4 - 6 stores] and, with the band barrier putting the waves in lockstep, the epilogues of all eight waves fell
//   together — the matrix pipes idle, 10 - 17 % of a band (tools/cr_stamps.hip).  Now a wave walks its tiles ONE at a time with two accumulators in
//   hand: while the 16 / 32 / 36 MFMAs of tile n accumulate (even / odd k-steps into two accumulators), the 16 values of tile n-1 (a third one) are
//   biased / masked / packed two at a time between the k-steps and stored when a 16-byte word group is complete — no epilogue phase, 48 accumulator
//   registers instead of 32 but one fragment ring instead of two, the stores spread over the loop instead of a burst behind it.  The code
//   between the k-steps is branch-free (a wave-uniform branch would split the scheduling region): a pixel that must not be stored (pitch padding,
//   rows beyond the band, "no previous tile") goes to a 4 KB dump page (p.dump).  Production forms only (forward + ReLU, data gradient with ReLU bit
//   words); p.dump == nullptr or the test-only epilogues select the round-4 pair form below.
//
// LDR = 2 (round 5): two of the eight waves are LOADERS.  tools/cr_stamps.hip: a wave that issues its ten LDS-DMA rounds of the next band sits in the
//   issue of those instructions for 2 700 - 4 200 cycles even with the slot decode in registers — the CU's vector-memory queue is shallow, and a band
//   (72 - 77 KB) takes ~7 000 cycles to arrive at the ~10 B/clk a CU gets of the HBM stream, so whoever issues the pieces is blocked for about as long as
//   they take to arrive.  With every wave issuing its share at the start of a band, that blocked time preceded the multiply phase on all of them (the band's
//   phases ADD).  Now waves 6 and 7 do nothing but request the next band (all its pieces, blocked in issue most of the time, no weights, no tiles) and the
//   six compute waves (3 pixel parts x 2 channel halves) never touch the load path: no DMA issue, no vmcnt wait — their output stores are never waited
//   for — one barrier per band.  A band then costs max(arrival of the next band, 4/3 of the old multiply + epilogue time) instead of their sum.
//   Forms with one output class only (OS == 1); the loaders keep their 40 rounds' slot decode in registers (PKR) — recomputing it per round made THEM
//   the bottleneck (conv2 forward 185 us).  MEASURED (2048 static frames, profiles/r05_conv_reg_forms.txt): conv2 forward 132 us against 134 - 137 for
//   the eight-wave form, conv3 forward 93 against 77 (compute-bound: six waves multiply), conv3 data gradient with EPI 107 against 98 for the
//   two-workgroup EPI form — all forms of conv2's forward converge at ~1.4 x the mixed read / write streaming time of its 500 MB, so the blocked
//   issue was not the whole story.  Kept selectable, not the production form.
template <int CK, int TA, int TB, int SI, bool REV, int OS = 1, int NWV = 8, int NBUF_ = 0, bool PKR = false, int EPI = 0, int LDR = 0>
__global__ void __launch_bounds__(NWV * 64, NWV == 4 ? 2 : 1) conv_reg_kernel(ConvTileP p) {
    using C = ConvRegCfg<CK, TA, TB, SI>;
    static_assert(!REV || SI == 1, "the data-gradient forms are stride-1 correlations (per parity class for OS = 2)");
    static_assert(OS == 1 || REV, "output parity classes only exist in the data-gradient form");
    static_assert(NWV == 8 || NWV == 4, "8 waves (one workgroup per CU, two band buffers) or 4 (two workgroups per CU, one buffer each)");
    constexpr int NBUF = NBUF_ ? NBUF_ : (NWV == 8 ? 2 : 1), PF = C::PIECES / NWV;      // band buffers of a workgroup (NWV = 4: 1, or 2 with smaller bands)
    static_assert(LDR == 0 || (LDR == 2 && NWV == 8 && OS == 1), "loader waves: 6 compute + 2 loader waves of a 512-thread workgroup, one output class");
    constexpr int NCW = NWV - LDR;                              // compute waves
    constexpr int NLD = LDR ? LDR : NWV;                        // waves that issue a band's DMA pieces, and the rounds each of them needs
    constexpr int PFL = (C::PIECES + NLD - 1) / NLD;
    constexpr int NCLS = OS * OS, CN = NCLS == 1 ? 64 : 32, PPARTS = NCLS == 1 ? NCW / 2 : NCW / 4, WPP = CN / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, lj = lane & 31;
    const int cls = NCLS == 1 ? 0 : (wave & 3), ph = cls / OS, pw = cls % OS;
    const int chw = NCLS == 1 ? (wave & 1) : 0, pq = NCLS == 1 ? (wave >> 1) : (wave >> 2);      // channel half / parity class, pixel part of this wave
    const int PLR = p.VPI, PLC = p.LP, plane_px = PLR * PLC;
    const size_t bbytes = C::band_bytes(PLR, PLC);
    lds_char* const lbase = (lds_char*)smem;
    lds_char* const bl = lbase + NBUF * bbytes;                // bias[64] fp32
    lds_char* const ml = bl + 256;                             // REV + maskbits: NBUF regions of p.MB bytes — the band's ReLU bit words (round 4)
    if (tid < 64) *(__attribute__((address_space(3))) float*)(bl + tid * 4) = p.bias ? p.bias[tid] : 0.f;
    // ---- weights -> registers (once).  MFMA row i of this wave's A operand carries channel chw*32 + 16*((i>>2)&1) + 4*(i>>3) + (i&3):
    // with the 32x32 C/D map (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) lane half h then owns channels chw*32 + 16h + reg, reg = 0..15
    h16x8_t wf[C::NS];
    if (!LDR || wave < NCW) {
        const int chn = chw * 32 + 16 * ((lj >> 2) & 1) + 4 * (lj >> 3) + (lj & 3);
        const h16_t* wr = p.w + ((long long)cls * CN + chn) * C::K + h * 8;
#pragma unroll
        for (int s = 0; s < C::NS; ++s) wf[s] = *reinterpret_cast<const h16x8_t*>(wr + s * 16);
    }
    // ---- DMA plan: thread (wave, lane) fills LDS slot q = (k*8 + wave)*64 + lane in round k (one wave instruction = 1 KB of consecutive
    // slots).  Slot -> (plane, plane row, plane col, chunk) -> staged (row, col, chunk); the pad slot of a pixel and the cells beyond the
    // band re-load a valid chunk (never read).  pk: bit 31 = column inside the image, bits 20..30 staged row of the band, bits 0..19
    // (image col * CH + chunk)
    const int multi = p.FPB > 1;
    const int nitems = multi ? (p.Nf + p.FPB - 1) / p.FPB : p.Nf * p.nbands;
    const int lw = LDR ? wave - NCW : wave;                     // index among the issuing waves
    const int npieces = (int)(bbytes / 1024), nrounds = (npieces + NLD - 1) / NLD;      // 1 KB pieces of a band; piece k*NLD + lw is this wave's in round k
    const float invX = 1.f / (float)C::XSS, invPP = 1.f / (float)plane_px, invPLCd = 1.f / (float)PLC;
    // slot q of a band -> bit 31 = column inside the image, bits 20..30 staged row of the band, bits 0..19 (image col * CH + chunk).  Recomputed per
    // band (a dozen VALU operations per 16-byte piece) rather than kept in registers: the weights need them
    auto slot_src = [&](int q) -> unsigned {
        const int ps = fast_div(q, invX);
        int chunk = q - ps * C::XSS;
        if (chunk >= C::CH) chunk = 0;
        int pl = fast_div(ps, invPP);
        int rem = ps - pl * plane_px;
        if (pl >= C::NPL) { pl = 0; rem = 0; }
        const int prow = fast_div(rem, invPLCd), pcol = rem - prow * PLC;
        const int r = min(prow * SI + pl / SI, p.LR - 1);
        int c = pcol * SI + pl % SI;
        unsigned cin = 0x80000000u;
        if (REV) { c -= TB - 1; if (c < 0 || c >= p.IMW) cin = 0u; }
        c = min(max(c, 0), p.IMW - 1);
        return cin | ((unsigned)r << 20) | (unsigned)(c * C::CH + chunk);
    };
    const int rowel = p.IMW * CK;
    const float invVPI = 1.f / (float)(p.IMH + TA - 1);
    unsigned pk[PKR ? PFL : 1];
    if (PKR && (!LDR || wave >= NCW)) {                         // LDR: the loader waves hold their 40 rounds' decode in the registers the compute waves give to weights
#pragma unroll
        for (int k = 0; k < PFL; ++k) pk[k] = slot_src(min(k * NLD + lw, npieces - 1) * 64 + lane);
    }
    auto dma = [&](int item, int bi) {
        if (p.dbg & 4) return;                                  // timing ablation (tools/time_conv_reg.py): no loads
        const int f = multi ? item * p.FPB : item / p.nbands, b = multi ? 0 : item % p.nbands;
        const int nfr = multi ? min(p.FPB, p.Nf - f) : 1;
        const int r0 = b * p.RB * SI, rmax = nfr * p.IMH - 1;
        const h16_t* src0 = p.img + (long long)f * p.IMH * rowel;
        lds_char* dst = lbase + bi * bbytes + lw * 1024;
#pragma unroll
        for (int k = 0; k < PFL; ++k) {
            if (k >= nrounds || k * NLD + lw >= npieces) break;     // wave-uniform
            unsigned pkk;
            if (PKR) pkk = pk[k];
            else {
                int q = (k * NLD + lw) * 64 + lane;
                asm volatile("" : "+v"(q));                         // opaque: keeps the (band-invariant) decode from being hoisted back into ten live registers
                pkk = slot_src(q);
            }
            const int sr = r0 + (int)((pkk >> 20) & 0x7ffu);        // staged row of the item
            const h16_t* s;
            if (REV) {        // staged row -> (frame of the stack, image row): rows [0, TA-1) of a frame's IMH + TA - 1 are its zero border
                const int ff = fast_div(sr, invVPI), r = sr - ff * (p.IMH + TA - 1) - (TA - 1);
                const bool ok = (pkk >> 31) && r >= 0 && r < p.IMH && ff < nfr;
                s = ok ? src0 + (long long)(ff * p.IMH + r) * rowel + (int)(pkk & 0xfffffu) * 8 : p.zeros;
            } else s = src0 + (long long)min(sr, rmax) * rowel + (int)(pkk & 0xfffffu) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s, (__attribute__((address_space(3))) void*)(dst + k * NLD * 1024), 16, 0, 0);
        }
        if (REV && p.maskbits) {
            // the ReLU bit words of the item's output pixels (whole rows of one frame, or whole stacked frames: contiguous in memory) ride along as
            // 256-byte pieces.  The epilogue then reads its words from LDS: a VECTOR load inside the tile loop would share vmcnt with the output
            // stores, and the compiler waits vmcnt(0) — i.e. for every store issued so far — at each use of a loaded word (measured: conv2's data
            // gradient spent 120 us in an epilogue whose 315 MB of stores take 50 us when they stream)
            const int orow0 = multi ? 0 : b * p.RB * OS, nrow = multi ? nfr * p.OUTH : min(p.RB * OS, p.OUTH - orow0);
            const int nwords = nrow * p.OUTW * WPP;
            const unsigned* src = p.maskbits + ((long long)f * p.OUTH + orow0) * p.OUTW * WPP;
            lds_char* mdst = ml + bi * p.MB + lw * 256;
            for (int k = 0; (k * NLD + lw) * 64 < nwords; ++k) {      // wave-uniform
                const int w = min((k * NLD + lw) * 64 + lane, nwords - 1);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + w), (__attribute__((address_space(3))) void*)(mdst + k * NLD * 256), 4, 0, 0);
            }
        }
    };
    // per-tap LDS offsets (uniform): plane (ta % SI, tb % SI), shifted by (ta / SI) plane rows and tb / SI columns; REV: the flipped tap
    int toff[TA * TB];
#pragma unroll
    for (int t = 0; t < TA * TB; ++t) {
        const int ta = REV ? TA - 1 - t / TB : t / TB, tb = REV ? TB - 1 - t % TB : t % TB;
        toff[t] = ((((ta % SI) * SI + tb % SI) * PLR + ta / SI) * PLC + tb / SI) * C::XS;
    }
    const float invPLC = 1.f / (float)PLC, invVPO = 1.f / (float)max(p.VPO, 1);
    int item = blockIdx.x, nb = 0;
    if (NBUF == 2 && (!LDR || wave >= NCW)) {
        if (item < nitems) dma(item, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (LDR && wave >= NCW) {                                   // ---- a loader wave: request band i+1 while the compute waves multiply band i
        static_assert(!(LDR && EPI && !REV), "(the forward EPI form has an extra barrier in front of the band loop)");
        int it = item, b = 0;
        while (it < nitems) {
            __syncthreads();                                    // band `it` has landed (this wave waited for its pieces); the compute waves are done with the other buffer
            it += (int)gridDim.x; b ^= 1;
            if (it < nitems) { dma(it, b); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        }
        return;
    }
    const int NI0 = (p.OUTH + OS - 1) / OS, NJ0 = (p.OUTW + OS - 1) / OS;      // output rows / columns of the largest class
    const bool fastmask = REV && p.maskbits && !p.relu;        // the production data-gradient form: 1-bit ReLU mask words, no activation
    // EPI: the tile whose epilogue is pending (accP), its output pixel (-1: none -> dump page) and ReLU bit word
    f32x16 accP;
    int popx = -1; unsigned pmw = 0xffffffffu;
    if (EPI) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accP[r] = 0.f;
    }
    if (EPI && !REV) __syncthreads();                           // bias[] in LDS is complete (the drains read it)
    auto biasE = [&](int c) -> float { return *(const __attribute__((address_space(3))) float*)(bl + (chw * 32 + 16 * h + c) * 4); };
    h16_t* const dumpp = EPI ? p.dump + lane * 16 : nullptr;     // 32 bytes per lane
#ifdef HULC_CR_STAMPS
    int crit = -1;
#endif
    while (item < nitems) {
#ifdef HULC_CR_STAMPS
        ++crit;
#endif
        CRSTAMP(0);
        __syncthreads();                                        // NBUF 2: every wave's share of this band has landed (each waited for its own DMAs
                                                                // before arriving) and every wave is done reading the other buffer
        CRSTAMP(1);
        const int cur = item;
        if (NBUF == 1) dma(cur, 0);                             // one buffer: every wave is done with the previous band -> load this one (the CU's
                                                                // OTHER workgroup multiplies meanwhile)
        item += (int)gridDim.x;
        lds_char* const xb = lbase + nb * bbytes;
        if (NBUF == 2) nb ^= 1;
        const int f = multi ? cur * p.FPB : cur / p.nbands, b = multi ? 0 : cur % p.nbands;
        const int i0 = b * p.RB;
        const int rows_total = multi ? p.RB : NI0;              // output-row slots of the whole item
        const int RBe = min(p.RB, rows_total - i0);
        const int npi = RBe * PLC, ntiles = (p.dbg & 2) ? 0 : (npi + 31) >> 5, last = npi - PLC + NJ0 - 1;      // dbg bit 1: no compute
        // this wave's tiles: a contiguous, balanced share of the band's tiles (10 tiles over 4 parts = 2 + 3 + 2 + 3, walked as pairs + a single)
        const int tbeg = pq * ntiles / PPARTS, tend = (pq + 1) * ntiles / PPARTS;
        // output pixels of a tile pair (-1: nothing to store)
        const int mbase = multi ? f * p.OUTH * p.OUTW : (f * p.OUTH + i0 * OS) * p.OUTW;      // first output pixel whose mask word the band staged
        const lds_char* const mlb = ml + (NBUF == 2 ? (nb ^ 1) : 0) * p.MB;
        auto pix = [&](int t0, int (&opx)[2]) {
            const bool two = t0 + 1 < tend;
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const int pi = t0 * 32 + mm * 32 + lj;
                const int ri = fast_div(pi, invPLC), j = pi - ri * PLC;
                const int ocol = j * OS + pw;
                bool ok = pi < npi && ocol < p.OUTW && (mm == 0 || two);
                int o;
                if (multi) {
                    const int ff = fast_div(ri, invVPO), rr = ri - ff * p.VPO, orow = rr * OS + ph;
                    ok = ok && orow < p.OUTH && f + ff < p.Nf;
                    o = ((f + ff) * p.OUTH + orow) * p.OUTW + ocol;
                } else {
                    const int orow = (i0 + ri) * OS + ph;
                    ok = ok && orow < p.OUTH;
                    o = (f * p.OUTH + orow) * p.OUTW + ocol;
                }
                opx[mm] = ok ? o : -1;
            }
        };
        if (NBUF == 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        // NBUF 2: the next band streams in under this band's MFMAs.  Waves 0-3 issue their DMA pieces now; waves 4-7 (the second wave of each
        // SIMD) after their first tile pair when they have two, so that one wave of a SIMD starts multiplying at once while the other
        // spends its ~0.3 us of DMA issue
        bool pend = NBUF == 2 && !LDR && item < nitems;
        if (pend && (wave < NWV / 2 || tend - tbeg <= 2)) { dma(item, nb); pend = false; }
        bool waited = false;
        // output pixel of ONE tile (EPI form)
        auto pix1 = [&](int t, int (&opx)[2]) __attribute__((always_inline)) {
            const int pi = t * 32 + lj;
            const int ri = fast_div(pi, invPLC), j = pi - ri * PLC;
            const int ocol = j * OS + pw;
            bool ok = pi < npi && ocol < p.OUTW;
            int o;
            if (multi) {
                const int ff = fast_div(ri, invVPO), rr = ri - ff * p.VPO, orow = rr * OS + ph;
                ok = ok && orow < p.OUTH && f + ff < p.Nf;
                o = ((f + ff) * p.OUTH + orow) * p.OUTW + ocol;
            } else {
                const int orow = (i0 + ri) * OS + ph;
                ok = ok && orow < p.OUTH;
                o = (f * p.OUTH + orow) * p.OUTW + ocol;
            }
            opx[0] = ok ? o : -1; opx[1] = -1;
        };

        if (EPI) {
            // one step = tile t multiplied while the pending tile (accP) drains.  EPI == 2: the k-steps alternate between TWO accumulators (even / odd
            // steps, summed at the end; consecutive MFMAs on one accumulator with instructions between them are listed at ~43 cycles each in
            // MI355X_MICROARCH.md) — measured no better than one accumulator here (the second wave of the SIMD fills the gaps) and 16 registers dearer
            auto do_tile = [&](int t) __attribute__((always_inline)) {
                const int pi = t * 32 + lj;
                int opx1[2]; pix1(t, opx1);
                unsigned mw = 0xffffffffu;
                if (REV) mw = *(const __attribute__((address_space(3))) unsigned*)(mlb + ((max(opx1[0], mbase) - mbase) * WPP + chw) * 4);      // (a pixel without output reads word 0: never used)
                lds_char* const x0 = xb + min(pi, last) * C::XS + h * 16;
                // the pending tile: output pointer (dump page if nothing is to be stored) and the mask half of this lane
                h16_t* const optr = popx >= 0 ? p.out + (long long)popx * CN + chw * 32 + 16 * h : dumpp;
                unsigned* const bptr = (!REV && p.bits_out && popx >= 0 && h == 0) ? p.bits_out + (long long)popx * 2 + chw : reinterpret_cast<unsigned*>(dumpp);
                const int mwh = (int)(pmw >> (16 * h));
                u32x4_t o = u32x4_t{0u, 0u, 0u, 0u};
                unsigned obits = 0;
                f32x16 c0, c1;
#pragma unroll
                for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
                h16x8_t xa[3];
                auto ld = [&](int s_, int slot) { xa[slot] = *(__attribute__((address_space(3))) h16x8_t*)(x0 + toff[s_ / C::KS] + (s_ % C::KS) * 32); };
                ld(0, 0);
                if (C::NS > 1) ld(1, 1);
#pragma unroll
                for (int s_ = 0; s_ < C::NS; ++s_) {
                    if (s_ + 2 < C::NS) ld(s_ + 2, (s_ + 2) % 3);
                    if (EPI == 2 && (s_ & 1)) c1 = MFMA_32x32x16_H(wf[s_], xa[s_ % 3], c1, 0, 0, 0);
                    else c0 = MFMA_32x32x16_H(wf[s_], xa[s_ % 3], c0, 0, 0, 0);
                    const int e = (s_ * 8 + C::NS - 1) / C::NS;   // slice e of the pending tile (values 2e, 2e+1) rides behind step floor(e NS / 8)
                    if (e < 8 && (e * C::NS) / 8 == s_) {
                        float v0 = accP[2 * e], v1 = accP[2 * e + 1];
                        if (REV) {
                            v0 = __int_as_float(__float_as_int(v0) & __builtin_amdgcn_sbfe(mwh, 2 * e, 1));
                            v1 = __int_as_float(__float_as_int(v1) & __builtin_amdgcn_sbfe(mwh, 2 * e + 1, 1));
                        } else {                                    // the two bias values come from LDS (16 registers of bias next to 144 of weights spilled)
                            typedef float f32x2_ __attribute__((ext_vector_type(2)));
                            const f32x2_ b2 = *(const __attribute__((address_space(3))) f32x2_*)(bl + (chw * 32 + 16 * h + 2 * e) * 4);
                            v0 = fmaxf(v0 + b2[0], 0.f); v1 = fmaxf(v1 + b2[1], 0.f);
                        }
                        const unsigned w = pack2h(v0, v1);
                        o[e & 3] = w;
                        if (!REV) obits |= (((w & 0xffffu) ? 1u : 0u) | ((w >> 16) ? 2u : 0u)) << (2 * e);
                        if ((e & 3) == 3) reinterpret_cast<u32x4_t*>(optr)[e >> 2] = o;
                        if (!REV && e == 7) {                       // (no bits_out: bptr is the dump page)
                            unsigned wb = obits << (16 * h);
                            wb |= __shfl_xor(wb, 32);
                            *bptr = wb;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) accP[r] = EPI == 2 ? c0[r] + c1[r] : c0[r];
                popx = opx1[0]; pmw = mw;
            };
            CRSTAMP(2);
#pragma unroll 1
            for (int t = tbeg; t < tend; ++t) do_tile(t);
            CRSTAMP(3);
            if (pend) { dma(item, nb); pend = false; }
            if (NBUF == 2 && !LDR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next band's pieces (issued at the band's start) and the stores of the drains
            waited = true;
            CRSTAMP(4);
        }

        if (!EPI) CRSTAMP(2);
#pragma unroll 1
        for (int t0 = tbeg; !EPI && t0 < tend; t0 += 2) {
            const bool two = t0 + 1 < tend;                     // uniform
            const int pi0 = t0 * 32 + lj, pi1 = pi0 + 32;
            int opx[2]; unsigned mw[2] = {0xffffffffu, 0xffffffffu};
            pix(t0, opx);
            if (REV && p.maskbits) {                            // LDS reads, issued in front of the multiply loop's own fragment reads
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
                    if (opx[mm] >= 0) mw[mm] = *(const __attribute__((address_space(3))) unsigned*)(mlb + ((opx[mm] - mbase) * WPP + chw) * 4);
            }
            lds_char* const x0 = xb + min(pi0, last) * C::XS + h * 16;
            lds_char* const x1 = xb + min(pi1, last) * C::XS + h * 16;
            f32x16 acc0, acc1;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
            // multiply loop, fully unrolled, image fragments read TWO k-steps ahead through a ring of three register sets: left to itself the
            // compiler reuses one set and waits for every ds_read right before its MFMA (lgkmcnt(0) between each pair: the matrix pipe sat
            // idle for the LDS latency at every step — 64 us for conv3's forward against 31 us of MFMA issue)
            auto mloop = [&](auto TWO) {
                constexpr bool two_ = decltype(TWO)::value;
                h16x8_t xa[3], xb[3];
                auto ld = [&](int s_, int slot) {
                    const int o = toff[s_ / C::KS] + (s_ % C::KS) * 32;
                    xa[slot] = *(__attribute__((address_space(3))) h16x8_t*)(x0 + o);
                    if (two_) xb[slot] = *(__attribute__((address_space(3))) h16x8_t*)(x1 + o);
                };
                ld(0, 0);
                if (C::NS > 1) ld(1, 1);
#pragma unroll
                for (int s_ = 0; s_ < C::NS; ++s_) {
                    if (s_ + 2 < C::NS) ld(s_ + 2, (s_ + 2) % 3);
                    acc0 = MFMA_32x32x16_H(wf[s_], xa[s_ % 3], acc0, 0, 0, 0);
                    if (two_) acc1 = MFMA_32x32x16_H(wf[s_], xb[s_ % 3], acc1, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);          // the scheduler may not sink the reads of step s+2 below this step's MFMAs
                }
            };
            if (p.dbg & 16) {                                   // ablation: no multiply loop
            } else if (two) mloop(std::true_type{});
            else mloop(std::false_type{});
            if (t0 + 2 >= tend) CRSTAMP(3);
            if (pend) { dma(item, nb); pend = false; }
            if (NBUF == 2 && !LDR && t0 + 2 >= tend) {          // last pair of this wave in the band: its DMA pieces of the NEXT band (issued a
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // multiply loop ago) and the earlier stores are waited for HERE, so that the
                waited = true;                                  // stores below stay in flight across the barrier
            }
            if (t0 + 2 >= tend) CRSTAMP(4);
            if ((p.dbg & 8) && acc0[0] != 12345.678f) continue;   // ablation: no epilogue
            // ---- epilogue: lane = (pixel lj, half h) holds channels chw*32 + 16h + [0, 16)
            if (REV && fastmask) {
                // data gradient, production form: out = bit ? acc : 0 -> a sign-extended 1-bit field ANDed onto the fp32 pattern (2 VALU per value)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    if (mm == 1 && !two) break;
                    const f32x16& a = mm ? acc1 : acc0;
                    const int mwh = (int)(mw[mm] >> (16 * h));
                    u32x4_t o[2];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int m0 = __builtin_amdgcn_sbfe(mwh, 2 * e, 1), m1 = __builtin_amdgcn_sbfe(mwh, 2 * e + 1, 1);
                        const float v0 = __int_as_float(__float_as_int(a[2 * e]) & m0), v1 = __int_as_float(__float_as_int(a[2 * e + 1]) & m1);
                        o[e >> 2][e & 3] = pack2h(v0, v1);
                    }
                    if (opx[mm] >= 0) {
                        long long ob = opx[mm];
#ifdef HULC_AB_SWITCHES
                        if (p.dbg & 64) ob = min((((long long)cur * NCLS + cls) * ntiles + t0 + mm) * 32 + lj, (long long)p.Nf * p.OUTH * p.OUTW - 1);   // experiment: tile-contiguous output
#endif
                        u32x4_t* op = reinterpret_cast<u32x4_t*>(p.out + ob * CN + chw * 32 + 16 * h);
#ifdef HULC_AB_SWITCHES
                        if (p.dbg & 128) { __builtin_nontemporal_store(o[0], op); __builtin_nontemporal_store(o[1], op + 1); } else
#endif
                        { op[0] = o[0]; op[1] = o[1]; }
                    }
                }
                continue;
            }
            f32x4 bb[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) bb[e] = REV ? f32x4{0.f, 0.f, 0.f, 0.f} : *(__attribute__((address_space(3))) f32x4*)(bl + (chw * 32 + 16 * h + 4 * e) * 4);
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                if (mm == 1 && !two) break;
                const f32x16& a = mm ? acc1 : acc0;
                const bool okm = opx[mm] >= 0;
                const long long ob = okm ? (long long)opx[mm] : 0ll;
                h16_t* const optr = p.out + ob * CN + chw * 32 + 16 * h;
                u32x4_t mk[2] = {u32x4_t{0u, 0u, 0u, 0u}, u32x4_t{0u, 0u, 0u, 0u}};
                if (REV && p.mask && okm) {                     // 16-bit mask values (per-kernel tests)
                    const u32x4_t* mp = reinterpret_cast<const u32x4_t*>(p.mask + ob * CN + chw * 32 + 16 * h);
                    mk[0] = mp[0]; mk[1] = mp[1];
                }
                u32x4_t o[2];
                unsigned obits = 0;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v0 = a[2 * e], v1 = a[2 * e + 1];
                    if (!REV) { v0 = fmaxf(v0 + bb[e >> 1][(2 * e) & 3], 0.f); v1 = fmaxf(v1 + bb[e >> 1][(2 * e + 1) & 3], 0.f); }
                    else {
                        if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                        if (p.maskbits) { v0 = ((mw[mm] >> (16 * h + 2 * e)) & 1u) ? v0 : 0.f; v1 = ((mw[mm] >> (16 * h + 2 * e + 1)) & 1u) ? v1 : 0.f; }
                        else if (p.mask) { const unsigned m = mk[e >> 2][e & 3]; v0 = h2f_lo(m) > 0.f ? v0 : 0.f; v1 = h2f_hi(m) > 0.f ? v1 : 0.f; }
                    }
                    const unsigned w = pack2h(v0, v1);
                    o[e >> 2][e & 3] = w;
                    if (!REV) obits |= (((w & 0xffffu) ? 1u : 0u) | ((w >> 16) ? 2u : 0u)) << (2 * e);
                }
                if (okm) {
                    u32x4_t* op = reinterpret_cast<u32x4_t*>(optr);
                    op[0] = o[0]; op[1] = o[1];
                }
                if (!REV && p.bits_out) {                       // word chw of the pixel = channels chw*32 .. +31: halves h = 0 / 1 give bits 0..15 / 16..31
                    unsigned w = obits << (16 * h);
                    w |= __shfl_xor(w, 32);
                    if (okm && h == 0) p.bits_out[ob * 2 + chw] = w;
                }
            }
        }
        CRSTAMP(5);
        if (pend) dma(item, nb);
        if (NBUF == 2 && !LDR && !waited) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (EPI && popx >= 0) {                                     // the last tile of this wave
        const f32x16& a = accP;
        const int mwh = (int)(pmw >> (16 * h));
        u32x4_t o[2];
        unsigned obits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v0 = a[2 * e], v1 = a[2 * e + 1];
            if (REV) {
                v0 = __int_as_float(__float_as_int(v0) & __builtin_amdgcn_sbfe(mwh, 2 * e, 1));
                v1 = __int_as_float(__float_as_int(v1) & __builtin_amdgcn_sbfe(mwh, 2 * e + 1, 1));
            } else { v0 = fmaxf(v0 + biasE(2 * e), 0.f); v1 = fmaxf(v1 + biasE(2 * e + 1), 0.f); }
            const unsigned w = pack2h(v0, v1);
            o[e >> 2][e & 3] = w;
            if (!REV) obits |= (((w & 0xffffu) ? 1u : 0u) | ((w >> 16) ? 2u : 0u)) << (2 * e);
        }
        u32x4_t* op = reinterpret_cast<u32x4_t*>(p.out + (long long)popx * CN + chw * 32 + 16 * h);
        op[0] = o[0]; op[1] = o[1];
    }
    if (EPI && !REV && p.bits_out) {                            // the mask word of that tile (both halves of the wave take part in the shuffle)
        unsigned obits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v0 = fmaxf(accP[2 * e] + biasE(2 * e), 0.f), v1 = fmaxf(accP[2 * e + 1] + biasE(2 * e + 1), 0.f);
            const unsigned w = pack2h(v0, v1);
            obits |= (((w & 0xffffu) ? 1u : 0u) | ((w >> 16) ? 2u : 0u)) << (2 * e);
        }
        unsigned wb = obits << (16 * h);
        wb |= __shfl_xor(wb, 32);
        if (popx >= 0 && h == 0) p.bits_out[(long long)popx * 2 + chw] = wb;
    }
}

// host side: band height for two resident bands (fewest bands), stacked frames for the gripper camera's small maps
inline int g_conv_reg_wgpc = 0;      // tools/cr_bench.hip only: workgroups per CU the band geometry is sized for (0 = the form's own: 1 for 8 waves, 2 for 4)
template <int CK, int TA, int TB, int SI, bool REV, int OS = 1, int NWV = 8, int NBUF_ = 0, bool PKR = false, int EPI = 0, int LDR = 0>
static inline bool launch_conv_reg(hipStream_t st, ConvTileP p) {
    if (EPI && (!p.dump || (REV && (!p.maskbits || p.relu || p.mask))))       // the pipelined epilogue covers the production forms only
        return launch_conv_reg<CK, TA, TB, SI, REV, OS, NWV, NBUF_, PKR, 0, LDR>(st, p);
    using C = ConvRegCfg<CK, TA, TB, SI>;
    constexpr int NBUF = NBUF_ ? NBUF_ : (NWV == 8 ? 2 : 1);                                  // band buffers per workgroup
    const int WGPC = g_conv_reg_wgpc ? g_conv_reg_wgpc : (NWV == 8 ? 1 : 2);                  // workgroups per CU
    if (p.IMW != p.IMH || p.OUTW != p.OUTH) return false;
    if (!REV && (p.mask || p.maskbits || !p.relu)) return false;
    const int NI = REV ? (p.OUTH + OS - 1) / OS : p.OUTH;      // output rows to cover (REV: class rows of the largest class)
    if (REV && (!p.zeros || NI > p.IMH + TA)) return false;
    const size_t cap = 160 * 1024 / WGPC - 64;
    p.LW = REV ? NI + TB - 1 : p.IMW;
    p.LP = (p.LW + SI - 1) / SI;                                // plane columns = m-index pitch
    p.FPB = 1; p.VPO = 0;
    // REV: bytes of the ReLU bit words of a band's output rows (whole 256-byte DMA pieces, one extra piece of slack)
    auto maskb = [&](int out_rows) -> size_t { return (REV && p.maskbits) ? ((size_t)out_rows * p.OUTW * (OS == 1 ? 2 : 1) * 4 + 255) / 256 * 256 + 256 : 0; };
    auto fits = [&](int LR, int PLR, int out_rows) { return C::lds_bytes(PLR, p.LP, NBUF, maskb(out_rows)) <= cap && C::band_bytes(PLR, p.LP) <= (size_t)C::PIECES * 1024 && LR < 2048; };
    int best_nb = 0;
    for (int nb = 1; nb <= NI; ++nb) {
        const int RB = (NI + nb - 1) / nb, LR = REV ? RB + TA - 1 : (RB - 1) * SI + TA, PLR = (LR + SI - 1) / SI;
        if (fits(LR, PLR, std::min(RB * OS, p.OUTH))) { best_nb = nb; break; }
    }
    if (!best_nb) return false;
    p.RB = (NI + best_nb - 1) / best_nb;
    p.nbands = (NI + p.RB - 1) / p.RB;
    p.LR = REV ? p.RB + TA - 1 : (p.RB - 1) * SI + TA;
    p.VPI = (p.LR + SI - 1) / SI;
    if (p.nbands == 1 && p.Nf > 1 && (REV ? p.OUTH == OS * (p.IMH + TA - 1) : p.IMH % SI == 0)) {   // stack FPB frames to a band: a band should feed the 8 waves' 16 tile slots
        const int vpo = REV ? p.IMH + TA - 1 : p.IMH / SI;
        int bestf = 1; double bc = 1e30;
        for (int fpb = 1; fpb <= 32; ++fpb) {
            const int LR = REV ? fpb * vpo + TA - 1 : fpb * p.IMH, PLR = (LR + SI - 1) / SI, RB = REV ? fpb * vpo : (LR - TA) / SI + 1;
            if (!fits(LR, PLR, fpb * p.OUTH)) break;
            constexpr int per = 2 * (OS == 1 ? (NWV - LDR) / 2 : NWV / 4);                                                // a wave pass = 2 tiles x the pixel parts
            const int tiles = (RB * p.LP + 31) / 32, rounds = (tiles + per - 1) / per;
            const int items = (p.Nf + fpb - 1) / fpb, wgs = std::min(items, 256 * WGPC);
            const double c = (double)((items + wgs - 1) / wgs) * (0.35 + rounds);
            if (c < bc - 1e-9) { bc = c; bestf = fpb; }
        }
        if (bestf > 1) {
            p.FPB = bestf; p.VPO = vpo;
            p.LR = REV ? bestf * vpo + TA - 1 : bestf * p.IMH; p.VPI = (p.LR + SI - 1) / SI; p.RB = REV ? bestf * vpo : (p.LR - TA) / SI + 1;
        }
    }
    p.MB = (int)maskb(p.FPB > 1 ? p.FPB * p.OUTH : std::min(p.RB * OS, p.OUTH));
    const size_t lds = C::lds_bytes(p.VPI, p.LP, NBUF, (size_t)p.MB);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)conv_reg_kernel<CK, TA, TB, SI, REV, OS, NWV, NBUF_, PKR, EPI, LDR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / (NWV == 8 ? 1 : 2) - 64);
        attr_set = true;
    }
    const int items = p.FPB > 1 ? (p.Nf + p.FPB - 1) / p.FPB : p.Nf * p.nbands;
    hipLaunchKernelGGL((conv_reg_kernel<CK, TA, TB, SI, REV, OS, NWV, NBUF_, PKR, EPI, LDR>), dim3(items < 256 * WGPC ? items : 256 * WGPC), dim3(NWV * 64), lds, st, p);
    return true;
}
template <int CK, int TA, int TB, int SI>
static inline bool launch_conv_reg_fwd(hipStream_t st, const ConvTileP& p) { return launch_conv_reg<CK, TA, TB, SI, false, 1, 8, 0, true>(st, p); }

}  // namespace HULC_NS
