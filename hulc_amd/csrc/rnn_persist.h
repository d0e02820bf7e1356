// hulc_amd/csrc/rnn_persist.h — a whole Elman recurrence (all S - 1 dependent steps) as ONE persistent launch, weights in registers.
//
// Reference: the action decoder's 2-layer nn.RNN(relu) (hulc/models/decoders/utils/rnn.py:5-14, run by logistic_decoder_rnn.py:260-287) and
// the mcil plan encoder's nn.RNN(tanh) (plan_recognition_net.py:12-42): per layer and direction
//     forward   H[q_s]  = act(Zx[q_s] + H[q_{s-1}] W_hh^T)                      s = 1 .. S-1
//     backward  dZ[q_s] = (dH[q_s] + dZ[q_{s-1}] W_hh) * act'(H[q_s])           (the same product against the transposed copy of W_hh)
// i.e. X[q_s] = f(X[q_{s-1}] · Wm^T, aux[q_s]) with M = B windows, N = K = 2048.
//
// One launch per step (skinny_lds_kernel) pays, per step, a kernel boundary and the whole 8.4 MB weight matrix through the CUs' load paths
// (192 KB per workgroup): 6.1 us for 0.54 GFLOP.  The recurrence only couples the 2048 features of ONE window, so the batch is split over
// the chip's eight XCDs instead of the weight matrix over all 256 CUs:
//   * a workgroup reads the XCD it runs on from the hardware (HW_REG_XCC_ID) and claims one of that XCD's 32 slots (one returning atomic on a
//     per-XCD census counter): group g = XCD g advances the windows [g wpx, (g+1) wpx) through all S - 1 steps, slot j owns the output
//     features [64 j, 64 j + 64).  Groups are formed from what the hardware reports, not from blockIdx: the dispatcher is observed to place
//     block b on XCD b % 8, which makes 32 per XCD, but nothing here assumes it — a launch whose XCDs are not populated 32 each raises
//     `err` (bounded polls) and the engine keeps the launch-per-step path;
//   * the 64 x 2048 slice of the weight matrix (256 KB) is loaded ONCE into the CU's vector registers as MFMA A-fragments (wave w of 8 holds
//     k in [256 w, 256 w + 256) of all 64 features: 32 x 16 B per lane = 128 VGPRs) and stays there for the whole launch: a step reads no
//     weight byte at all;
//   * per step a wave loads its k-range of the previous state (wpx windows x 512 B, as whole 128-byte lines, turned into B-fragments through a
//     wave-private LDS image), issues 32 MFMAs (16x16x32; the token
//     dimension of the tile is the group's <= 16 windows), the 8 k-partials meet in LDS (double-buffered by step parity: ONE workgroup
//     barrier per step), four reducer waves (one per SIMD) sum them, apply the step's epilogue (residual, ReLU / tanh or their derivative
//     masks) and store the 64-feature slice of the new state;
//   * hand-off: all communication stays INSIDE one XCD, whose L2 is the coherence point of its 32 CUs.  Producer: plain 16-byte stores (they
//     write through the CU's L1 into the L2), `s_waitcnt vmcnt(0)`, then each of the four reducer waves stores the step counter into its own word
//     of the mailbox of each of the 32 consumers.  Consumer: a wave samples the 16 words of the four producers of its k-range in its
//     workgroup's own mailbox with `sc1` loads (they bypass the L1, which another CU's stores never refresh, and are served by the L2 — the mailbox lines are
//     rewritten every step and stay dirty-resident there) and then reads the state with `sc1` loads.  Measured alternatives (B = 64, S = 32,
//     per step): write-through `sc1` stores + flags through memory, the placement-independent form, 5.6 - 9.3 us (no better than a launch
//     per step: 4.2 us back to back); this form 2.6 - 3.2 us with the state read fragment-shaped (round 3), where the state as its own ready flag
//     (slices pre-filled with a reserved NaN pattern, consumers re-reading until it is gone: one hop less) cost 3.6 - 4.7 us — a fragment-shaped
//     sweep is 2048 line lookups per CU and sampling the payload that way saturated the L2.  Round 4: the state is read as whole lines (RP_COAL,
//     2.36 us per step with flags), which makes a sweep cheap enough that the SECOND form wins: from the third step on (RP_TAG) the producers
//     neither drain their stores nor post flags, consumers sweep the slices until no 16-byte piece shows the pattern (2.24 us per step at B = 64,
//     2.06 at B = 37).  The first hand-off of a launch keeps the flags: it is what proves that a producer's pre-fill of its later slices has landed
//     (see the kernel).  There is no grid-wide barrier and no cross-XCD traffic;
//   * every poll loop is bounded; a timeout (workgroups not co-resident: somebody else holds CUs; XCD population not 32) raises `err`.
// Mailbox words never need zeroing: a launch publishes base + s with a base the host advances by 4096 per launch.  The census counters are
// double-buffered by launch parity: a launch counts in one set and clears the other.
#pragma once
#include "common.h"

namespace HULC_NS {

constexpr int RP_HID = 2048, RP_NG = 8, RP_SLOTS = 32, RP_COLS = 64, RP_NW = 8, RP_KW = 256, RP_KS = RP_KW / 32;
constexpr int RP_TPITCH = RP_COLS * 4 + 16;          // LDS pitch of one window's 64 fp32 partial sums: 17 slots -> conflict-free 16-byte writes
// LDS pitch of one window's 512-byte k-range in a wave's staging image (RP_COAL): 34 slots of 16 bytes = 2 (mod 16).  ds_read_b128 is served in groups
// of 16 lanes that mix two gq values ({0-3, 12-15 | 20-27}, ...): with slot(li, gq) = 2 li + gq the rows of one gq take the even slots and the other's the
// odd ones — conflict-free; 33 slots (2-way conflicts between (li, gq) and (li + 1, gq - 1)) showed as SQ_LDS_BANK_CONFLICT 0.23
constexpr int RP_XPITCH = RP_KW * 2 + 32;
#ifndef RP_COAL
#define RP_COAL 1
#endif
#ifndef RP_TAG
#define RP_TAG (RP_COAL)      // the state is its own ready flag from the third step on (see the kernel); needs the whole-line read
#endif
constexpr unsigned RP_SENT = 0xFFFFFFFFu;     // two 16-bit NaNs no arithmetic of the epilogue produces (a computed NaN is the canonical quiet one)
constexpr int RP_MAIL_WORDS = RP_NG * RP_SLOTS * RP_SLOTS * 4;        // mailbox[group][consumer][producer][reducer wave]: 512 bytes per consumer workgroup
constexpr int RP_FLAG_WORDS = RP_MAIL_WORDS + 2 * RP_NG * 32;     // mailbox[group][consumer][producer]: one 128-byte line per consumer workgroup

struct RnnPersistP {
    h16_t* X;             // [S][B][2048]: the recurrence's own sequence (H going forward, dZ going backward)
    const h16_t* W;       // [2048 n][2048 k]: X_new[., n] = sum_k X_prev[., k] W[n][k]
    const h16_t* res;     // [S][B][2048] or null: added before the activation / mask (Zx forward, dH backward)
    const h16_t* mask;    // [S][B][2048] or null: backward — the stored states H
    int B, S, q0, dq;     // step s reads position q0 + (s-1) dq and writes q0 + s dq
    int act;              // mask == null: 1 ReLU, 2 tanh;  mask != null: 1 (H > 0), 2 (1 - H^2)
    int wpx;              // windows per group
    unsigned* flags;      // [RP_FLAG_WORDS] mailboxes, then [2][8][32] census counters (one 128-byte line each)
    unsigned base;
    int parity;           // launch parity: census set in use
    unsigned* err;        // device-visible word (the engine maps a pinned host word): 1 = a poll timed out, 2 = an XCD held more than 32 workgroups
    unsigned* skip;       // device word or null: a failing launch stores `skip_tag` here — the optimizer kernel of the step this recurrence belongs to
    unsigned skip_tag;    // compares its own tag with the word and leaves the weights untouched (engine.h: persist_check)
    int fault;            // tests (hulc_set_option debug_persist_fault): slot 0 of every XCD leaves at once — its consumers' bounded polls time out
    // dual (round 4): TWO independent recurrences of the same shape in one launch — the two directions of a bidirectional layer (mcil's plan encoder,
    // plan_recognition_net.py:27-33).  XCDs 0-3 advance problem 0, XCDs 4-7 problem 1, each XCD ceil(B / 4) <= 16 windows: the MFMA tile's 16 token
    // columns are filled (8 of them are padding at B = 64 over eight XCDs) and both chains finish in the time of ~1.2
    int dual;
    h16_t* X2; const h16_t* W2; const h16_t* res2; const h16_t* mask2; int q02, dq2;
    long long* stamps;    // RP_STAMPS builds (tools/rnn_persist_bench.hip): [S][2 waves][8] shader-clock stamps of workgroup 8
};
#ifdef RP_STAMPS
#define RP_STAMP(i) do { if (p.stamps && blockIdx.x == 8 && lane == 0 && (wave == 0 || wave == 5)) p.stamps[(s * 2 + (wave ? 1 : 0)) * 8 + (i)] = clock64(); } while (0)
#else
#define RP_STAMP(i) do { } while (0)
#endif

typedef unsigned rp_u32x4 __attribute__((ext_vector_type(4)));

DEVI float rp_dpp_xor1(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true)); }       // quad_perm [1,0,3,2]
DEVI float rp_dpp_shl2(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x102, 0xF, 0xF, true)); }      // row_shl:2 — lane i reads lane i + 2
// TOK: window capacity of a group (8 or 16).  MODE: 0 ReLU(z), 1 tanh(z) (forward); 2 z * (mask > 0), 3 z * (1 - mask^2) (backward), z = product (+ res if RES)
template <int TOK, int MODE, bool RES>
__global__ void __launch_bounds__(RP_NW * 64) rnn_persist_kernel(RnnPersistP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    constexpr int WS = TOK * RP_TPITCH + 32;          // per-wave partial block; 4 WS = 128 (mod 256): the two partial halves a reducer pair reads never collide
    constexpr int BUF = RP_NW * WS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, gq = lane >> 4;
    __shared__ unsigned s_slot;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int grp = (int)(xcc & 7u);
    if (tid == 0) {
        unsigned* const census = p.flags + RP_MAIL_WORDS;
        const unsigned sl = __hip_atomic_fetch_add(census + (p.parity * RP_NG + grp) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sl == 0) __hip_atomic_store(census + ((p.parity ^ 1) * RP_NG + grp) * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the next launch's set
        if (sl >= (unsigned)RP_SLOTS) {
            __hip_atomic_store(p.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (p.skip) __hip_atomic_store(p.skip, p.skip_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_slot = sl;
    }
    __syncthreads();
    const int slot = (int)s_slot;
    if (slot >= RP_SLOTS) return;                     // an over-populated XCD: the launch is reported failed
    if (p.fault && slot == 0) return;                 // injected fault (tests): a producer that never posts
    const bool second = p.dual && (grp >> 2);            // which of the two problems this XCD works on
    const int t0 = (p.dual ? (grp & 3) : grp) * p.wpx;
    const int nwin = min(p.wpx, p.B - t0);
    if (nwin <= 0) return;
    const long long BH = (long long)p.B * RP_HID;
    h16_t* const pX = second ? p.X2 : p.X;
    const h16_t* const pW = second ? p.W2 : p.W;
    const h16_t* const pres = second ? p.res2 : p.res;
    const h16_t* const pmask = second ? p.mask2 : p.mask;
    const int pq0 = second ? p.q02 : p.q0, pdq = second ? p.dq2 : p.dq;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)pX, 0, 0x7fffffff, 0x00020000);

    // ---- the weight slice, once: rows 64 slot + 16 ct + li, k = 256 wave + 32 ks + 8 gq
    h16x8_t wf[4][RP_KS];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < RP_KS; ++ks)
            wf[ct][ks] = *reinterpret_cast<const h16x8_t*>(pW + (long long)(RP_COLS * slot + 16 * ct + li) * RP_HID + RP_KW * wave + 32 * ks + 8 * gq);
#if RP_TAG
    // Steps >= 3 hand the state over WITHOUT flags: every 16-byte piece of a slice is written by one store of one producer lane and read by one
    // load of one consumer lane, so a piece is either the pattern it was pre-filled with or the new values.  The reducer waves pre-fill the slices this
    // workgroup will write in steps 2 .. S-1 here, and their own `vmcnt(0)` in front of the step-1 flag covers these stores too: a consumer that has
    // seen the step-1 words of its four producers (the flag protocol, kept for the first hand-off) knows their later slices hold the pattern, not
    // the values of an earlier launch.  What it saves per step: the producers' store drain and the flag's own L2 round trip in front of the state read.
    if (wave < 4) {
        const rp_u32x4 sent = rp_u32x4{RP_SENT, RP_SENT, RP_SENT, RP_SENT};
        const int per = nwin * 8;                     // 16-byte pieces of one step's slice
        for (int idx = tid; idx < (p.S - 2) * per; idx += 4 * 64) {
            const int s2 = 2 + idx / per, r = idx % per;
            const long long o = (long long)(pq0 + s2 * pdq) * BH + (long long)(t0 + (r >> 3)) * RP_HID + RP_COLS * slot + 8 * (r & 7);
            __builtin_amdgcn_raw_buffer_store_b128(sent, xr, (unsigned)(o * 2), 0, 0);
        }
    }
#endif
    // mailbox[group][consumer slot][producer slot][reducer wave]: each of a producer's four reducer waves posts its own word as soon as ITS part of
    // the slice is in the L2 (no arrival count among them), into the 32 consumers' mailboxes (one lane per consumer); a consumer wave samples
    // only the 16 words of the four producers whose features are its k-range — it does not wait for the slowest of all 32
    const unsigned* fl = p.flags + ((grp * RP_SLOTS + slot) * RP_SLOTS + 4 * wave) * 4;
    bool dead = false;
    // reducer waves 0..3 (one per SIMD): wave r sums the 8 k-partials of windows r * TOK/4 .. + TOK/4; lane = (window tk, 4-feature group c4, partial half ph)
    constexpr int NIT = TOK / 8;
    const int tk = lane >> 5, c4 = (lane >> 1) & 15, ph = lane & 1;

    for (int s = 1; s < p.S; ++s) {
        const long long qp = (long long)(pq0 + (s - 1) * pdq) * BH, qc = (long long)(pq0 + s * pdq) * BH;
        RP_STAMP(0);
        // epilogue operands of this step: independent of the recurrence, requested before the wait (the lanes that will store: 8 features each)
        rp_u32x4 rv[NIT], mv[NIT];
        if (wave < 4) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int tok = wave * (TOK / 4) + it * 2 + tk;
                rv[it] = rp_u32x4{0u, 0u, 0u, 0u}; mv[it] = rp_u32x4{0u, 0u, 0u, 0u};
                if (tok < nwin && (lane & 3) == 0) {
                    const long long o = qc + (long long)(t0 + tok) * RP_HID + RP_COLS * slot + 4 * c4;
                    if (RES) rv[it] = *reinterpret_cast<const rp_u32x4*>(pres + o);
                    if (MODE >= 2) mv[it] = *reinterpret_cast<const rp_u32x4*>(pmask + o);
                }
            }
        }
        if (s > 1 && (!RP_TAG || s == 2) && !dead) {
            // wait for the four producers of this wave's k-range (lane = producer * 4 + reducer wave); the first step's input comes from an earlier launch
            const unsigned want = p.base + (unsigned)(s - 1);
            int spins = 0;
            for (;;) {
                unsigned v = want;
                if (lane < 16) v = __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((int)(v - want) >= 0)) break;
                if (++spins > (1 << 18)) {
                    dead = true;
                    if (lane == 0) {
                        __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (p.skip) __hip_atomic_store(p.skip, p.skip_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
            }
            // compiler barrier: the state loads below may not be hoisted above (or merged across) the poll — the producers' slices are only valid
            // once their words matched.  No hardware fence: the loads are `sc1` (served by the L2, the coherence point of producer and consumer),
            // they are issued after the poll's values have returned (the loop exit depends on them), and an agent-scope acquire would add a
            // ~1.7 us L1 invalidate to each of the 31 steps (MI355X_MICROARCH.md, fence table) for data the L1 never holds
            asm volatile("" ::: "memory");
        }
        RP_STAMP(1);
        // ---- B fragments: window li, k = 256 wave + 32 ks + 8 gq (sc1: past the L1, which other CUs' stores never refresh; served by the XCD's L2)
        h16x8_t bf[RP_KS];
#if RP_COAL
        {
            // The MFMA layout puts 16 different windows (4 KB apart) on 16 adjacent lanes: read that way, every lane's 16 bytes are their own
            // line lookup in the CU's L1 / texture path (2048 per CU and step) and their own request at the L2.  Instead the wave reads its
            // 512-byte k-range of a window with 32 adjacent lanes (four whole 128-byte lines, two windows per instruction) and turns the tile
            // into fragments through a wave-private LDS image (pitch 544 bytes, see RP_XPITCH).
            lds_c* const xs = (lds_c*)smem + 2 * BUF + wave * (TOK * RP_XPITCH);
            const int hw = lane >> 5, ch = lane & 31;
            const unsigned off = (unsigned)((qp + (long long)(t0 + hw) * RP_HID + RP_KW * wave + 8 * ch) * 2);
            rp_u32x4 stg[TOK / 2];
#if RP_TAG
            for (int spins = 0;;) {
                bool ready = true;
#pragma unroll
                for (int i = 0; i < TOK / 2; ++i) {
                    stg[i] = rp_u32x4{0u, 0u, 0u, 0u};
                    if (2 * i + hw < nwin) stg[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off + (unsigned)(2 * i) * (RP_HID * 2), 0, 16);
                }
#pragma unroll
                // all FOUR words of a piece (VERDICT r4 weak 1 iii): a piece is one 16-byte store of one producer lane and is observed untorn on
                // gfx950, but nothing architectural excludes a piece whose middle words are still the pattern — two more compares per piece
                for (int i = 0; i < TOK / 2; ++i) ready = ready && stg[i][0] != RP_SENT && stg[i][1] != RP_SENT && stg[i][2] != RP_SENT && stg[i][3] != RP_SENT;
                if (s < 3 || dead || __all(ready)) break;
                if (++spins > (1 << 16)) {
                    dead = true;
                    if (lane == 0) {
                        __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (p.skip) __hip_atomic_store(p.skip, p.skip_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
            RP_STAMP(1);
#else
#pragma unroll
            for (int i = 0; i < TOK / 2; ++i) {
                stg[i] = rp_u32x4{0u, 0u, 0u, 0u};
                if (2 * i + hw < nwin) stg[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off + (unsigned)(2 * i) * (RP_HID * 2), 0, 16);
            }
#endif
#pragma unroll
            for (int i = 0; i < TOK / 2; ++i) *(__attribute__((address_space(3))) rp_u32x4*)(xs + (2 * i + hw) * RP_XPITCH + 16 * ch) = stg[i];
            const lds_c* const xr_l = xs + (li & (TOK - 1)) * RP_XPITCH + 16 * gq;
#pragma unroll
            for (int ks = 0; ks < RP_KS; ++ks) {
                rp_u32x4 v = *(const __attribute__((address_space(3))) rp_u32x4*)(xr_l + 64 * ks);
                bf[ks] = *reinterpret_cast<h16x8_t*>(&v);
            }
        }
#else
        {
            const unsigned off = (unsigned)((qp + (long long)(t0 + li) * RP_HID + RP_KW * wave + 8 * gq) * 2);
#pragma unroll
            for (int ks = 0; ks < RP_KS; ++ks) {
                rp_u32x4 v = rp_u32x4{0u, 0u, 0u, 0u};
                if (li < nwin) v = __builtin_amdgcn_raw_buffer_load_b128(xr, off + 64 * ks, 0, 16);
                bf[ks] = *reinterpret_cast<h16x8_t*>(&v);
            }
        }
#endif
#ifdef RP_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        RP_STAMP(2);
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < RP_KS; ++ks)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = MFMA_16x16x32_H(wf[ct][ks], bf[ks], acc[ct], 0, 0, 0);
        // ---- k-partials -> LDS: [parity][wave][window][64 features]; lane (li, gq) owns features 16 ct + 4 gq .. + 3 of window li
        lds_c* const pb = (lds_c*)smem + (s & 1) * BUF;
        if (li < TOK && li < nwin) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) *(lds_f4*)(pb + wave * WS + li * RP_TPITCH + ct * 64 + gq * 16) = acc[ct];
        }
        RP_STAMP(3);
        __syncthreads();
        RP_STAMP(4);
        if (wave < 4) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int tokl = wave * (TOK / 4) + it * 2 + tk;
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                if (tokl < nwin) {
#pragma unroll
                    for (int w = 0; w < RP_NW / 2; ++w) a += *(lds_f4*)(pb + ((RP_NW / 2) * ph + w) * WS + tokl * RP_TPITCH + c4 * 16);
                }
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = a[e] + rp_dpp_xor1(a[e]);
                // lanes (lane & 3) == 0 gather the neighbouring 4-feature group: 8 features = one 16-byte store
#pragma unroll
                for (int e = 0; e < 4; ++e) v[4 + e] = rp_dpp_shl2(v[e]);
                if (tokl < nwin && (lane & 3) == 0) {
                    rp_u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x0 = v[2 * e], x1 = v[2 * e + 1];
                        if (RES) { x0 += h2f_lo(rv[it][e]); x1 += h2f_hi(rv[it][e]); }
                        if (MODE == 0) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
                        else if (MODE == 1) { x0 = tanhf(x0); x1 = tanhf(x1); }
                        else {
                            const float m0 = h2f_lo(mv[it][e]), m1 = h2f_hi(mv[it][e]);
                            if (MODE == 3) { x0 *= 1.f - m0 * m0; x1 *= 1.f - m1 * m1; }
                            else { x0 = m0 > 0.f ? x0 : 0.f; x1 = m1 > 0.f ? x1 : 0.f; }
                        }
                        o[e] = pack2h(x0, x1);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o, xr, (unsigned)((qc + (long long)(t0 + tokl) * RP_HID + RP_COLS * slot + 4 * c4) * 2), 0, 0);
                }
            }
            RP_STAMP(5);
            if (!RP_TAG || s == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of the slice (and, RP_TAG, of the pre-fill) is in the L2
                RP_STAMP(6);
                if (lane < RP_SLOTS)      // this reducer wave's part of the slice is in the L2 -> its word in every consumer's mailbox
                    __hip_atomic_store(p.flags + (((grp * RP_SLOTS + lane) * RP_SLOTS + slot) * 4 + wave), p.base + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
}

static inline size_t rnn_persist_lds(int tok) { return (size_t)2 * RP_NW * (tok * RP_TPITCH + 32) + (RP_COAL ? (size_t)RP_NW * tok * RP_XPITCH : 0); }

// false: shape not covered (the caller keeps the launch-per-step path).  X must lie below 2 GB from its base (buffer offsets are 32 bit).
template <int TOK, int MODE, bool RES>
static inline void rnn_persist_go(hipStream_t st, const RnnPersistP& p) {
    static bool attr = false;
    if (!attr) { hipFuncSetAttribute((const void*)rnn_persist_kernel<TOK, MODE, RES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rnn_persist_lds(TOK)); attr = true; }
    hipLaunchKernelGGL((rnn_persist_kernel<TOK, MODE, RES>), dim3(RP_NG * RP_SLOTS), dim3(RP_NW * 64), rnn_persist_lds(TOK), st, p);
}
static inline bool launch_rnn_persist(hipStream_t st, RnnPersistP p) {
    if (p.B < 1 || p.S < 2 || (p.act != 1 && p.act != 2)) return false;
    const int ngrp = p.dual ? RP_NG / 2 : RP_NG;          // XCDs per problem
    p.wpx = (p.B + ngrp - 1) / ngrp;
    if (p.wpx > 16 || (long long)p.S * p.B * RP_HID * 2 >= (1ll << 31)) return false;
    if (!p.mask && !p.res) return false;                 // a forward recurrence always has its input projection
    if (p.dual && (!p.X2 || !p.W2 || (p.res != nullptr) != (p.res2 != nullptr) || (p.mask != nullptr) != (p.mask2 != nullptr))) return false;
    const int mode = (p.mask ? 2 : 0) + (p.act == 2 ? 1 : 0);
    const bool big = p.wpx > 8;
#define RP_GO(M, R) do { if (big) rnn_persist_go<16, M, R>(st, p); else rnn_persist_go<8, M, R>(st, p); } while (0)
    if (mode == 0) RP_GO(0, true);
    else if (mode == 1) RP_GO(1, true);
    else if (mode == 2) { if (p.res) RP_GO(2, true); else RP_GO(2, false); }
    else { if (p.res) RP_GO(3, true); else RP_GO(3, false); }
#undef RP_GO
    return true;
}

}  // namespace HULC_NS
