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
//   * workgroup (g = blockIdx % 8, j = blockIdx / 8) belongs to group g — observed to be XCD g (MI355X_MICROARCH.md, workgroup dispatch) —
//     which advances the windows [g wpx, (g+1) wpx) through all S - 1 steps; within the group it owns output features [64 j, 64 j + 64);
//   * its 64 x 2048 slice of the weight matrix (256 KB) is loaded ONCE into the CU's vector registers as MFMA A-fragments (wave w holds
//     k in [128 w, 128 w + 128) of all 64 features: 16 x 16 B per lane = 64 VGPRs) and stays there for the whole launch: a step reads no
//     weight byte at all;
//   * per step a wave loads the B-fragments of its k-range of the previous state (wpx windows x 256 B) straight from memory, issues 16 MFMAs
//     (16x16x32; the token dimension of the tile is the group's <= 16 windows), the 16 k-partials meet in LDS (double-buffered by step
//     parity: ONE workgroup barrier per step), wave 0 sums them, applies the step's epilogue (residual, ReLU / tanh or their derivative
//     masks) and publishes the 64-feature slice of the new state;
//   * hand-off inside a group: the slice is stored with 16-byte write-through (`sc1`) stores, the storing wave drains its vmcnt and then
//     stores a step counter to its own flag word (`sc1`); a consuming wave polls exactly the TWO flags of the workgroups that produce its
//     k-range (relaxed agent-scope loads + s_sleep) and reads the state with `sc1` loads — the {sc1 stores, sc1 loads} form of
//     MI355X_MICROARCH.md "inter-workgroup visibility", valid for any placement: if the dispatcher ever put a group's workgroups on several
//     XCDs the launch is slower, not wrong.  There is no grid-wide barrier and no cross-group traffic at all;
//   * every poll loop is bounded; a timeout (workgroups not co-resident: somebody else holds CUs) raises `err` and lets the launch drain.
// Flag words never need zeroing: a launch publishes base + s with a base the host advances by 4096 per launch.
#pragma once
#include "common.h"

namespace HULC_NS {

constexpr int RP_HID = 2048, RP_NG = 8, RP_SLOTS = 32, RP_COLS = 64, RP_NW = 16, RP_KW = 128;
constexpr int RP_TPITCH = RP_COLS * 4 + 16;          // LDS pitch of one window's 64 fp32 partial sums: 17 slots -> conflict-free 16-byte writes
constexpr int RP_FLAG_WORDS = RP_NG * 64;            // one 256-byte line pair per group

struct RnnPersistP {
    h16_t* X;             // [S][B][2048]: the recurrence's own sequence (H going forward, dZ going backward)
    const h16_t* W;       // [2048 n][2048 k]: X_new[., n] = sum_k X_prev[., k] W[n][k]
    const h16_t* res;     // [S][B][2048] or null: added before the activation / mask (Zx forward, dH backward)
    const h16_t* mask;    // [S][B][2048] or null: backward — the stored states H
    int B, S, q0, dq;     // step s reads position q0 + (s-1) dq and writes q0 + s dq
    int act;              // mask == null: 1 ReLU, 2 tanh;  mask != null: 1 (H > 0), 2 (1 - H^2)
    int wpx;              // windows per group
    unsigned* flags;      // [8][64]
    unsigned base;
    unsigned* err;
};

typedef unsigned rp_u32x4 __attribute__((ext_vector_type(4)));

template <int TOK>      // window capacity of a group: 8 or 16
__global__ void __launch_bounds__(1024) rnn_persist_kernel(RnnPersistP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char lds_c;
    typedef __attribute__((address_space(3))) f32x4 lds_f4;
    constexpr int WS = TOK * RP_TPITCH + 32;          // per-wave partial block; + 32 B: the 16 blocks start in different bank groups
    constexpr int BUF = RP_NW * WS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, gq = lane >> 4;
    const int grp = blockIdx.x & (RP_NG - 1), slot = blockIdx.x >> 3;
    const int t0 = grp * p.wpx;
    const int nwin = min(p.wpx, p.B - t0);
    if (nwin <= 0) return;
    const long long BH = (long long)p.B * RP_HID;

    // ---- the weight slice, once: rows 64 slot + 16 ct + li, k = 128 wave + 32 ks + 8 gq
    h16x8_t wf[4][4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            wf[ct][ks] = *reinterpret_cast<const h16x8_t*>(p.W + (long long)(RP_COLS * slot + 16 * ct + li) * RP_HID + RP_KW * wave + 32 * ks + 8 * gq);

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, 0x7fffffff, 0x00020000);
    const unsigned long long* fl = reinterpret_cast<const unsigned long long*>(p.flags + grp * 64 + 2 * wave);
    unsigned* const myflag = p.flags + grp * 64 + slot;
    bool dead = false;
    // wave 0's epilogue items: (window, 8-feature group) = lane (+ 64 for the second half of a 16-window group)
    constexpr int NIT = TOK / 8;

    for (int s = 1; s < p.S; ++s) {
        const long long qp = (long long)(p.q0 + (s - 1) * p.dq) * BH, qc = (long long)(p.q0 + s * p.dq) * BH;
        // epilogue operands of this step: independent of the recurrence, requested before the wait
        rp_u32x4 rv[NIT], mv[NIT];
        if (wave == 0) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int tok = (lane >> 3) + 8 * it;
                rv[it] = rp_u32x4{0u, 0u, 0u, 0u}; mv[it] = rp_u32x4{0u, 0u, 0u, 0u};
                if (tok < nwin) {
                    const long long o = qc + (long long)(t0 + tok) * RP_HID + RP_COLS * slot + 8 * (lane & 7);
                    if (p.res) rv[it] = *reinterpret_cast<const rp_u32x4*>(p.res + o);
                    if (p.mask) mv[it] = *reinterpret_cast<const rp_u32x4*>(p.mask + o);
                }
            }
        }
        // ---- wait for the two producers of this wave's k-range (the first step's input comes from an earlier launch)
        if (s > 1 && !dead) {
            const unsigned want = p.base + (unsigned)(s - 1);
            int spins = 0;
            for (;;) {
                const unsigned long long v = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)((unsigned)v - want) >= 0 && (int)((unsigned)(v >> 32) - want) >= 0) break;
                if (++spins > (1 << 19)) { dead = true; if (lane == 0) __hip_atomic_store(p.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        // ---- B fragments: window li, k = 128 wave + 32 ks + 8 gq (sc1: past the L1, which other CUs' stores never refresh)
        h16x8_t bf[4];
        {
            const unsigned off = (unsigned)((qp + (long long)(t0 + li) * RP_HID + RP_KW * wave + 8 * gq) * 2);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                rp_u32x4 v = rp_u32x4{0u, 0u, 0u, 0u};
                if (li < nwin) v = __builtin_amdgcn_raw_buffer_load_b128(xr, off + 64 * ks, 0, 16);
                bf[ks] = *reinterpret_cast<h16x8_t*>(&v);
            }
        }
        f32x4 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = MFMA_16x16x32_H(wf[ct][ks], bf[ks], acc[ct], 0, 0, 0);
        // ---- k-partials -> LDS: [parity][wave][window][64 features]; lane (li, gq) owns features 16 ct + 4 gq .. + 3 of window li
        lds_c* const pb = (lds_c*)smem + (s & 1) * BUF;
        if (li < TOK && li < nwin) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) *(lds_f4*)(pb + wave * WS + li * RP_TPITCH + ct * 64 + gq * 16) = acc[ct];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int tok = (lane >> 3) + 8 * it, c8 = lane & 7;
                if (tok < nwin) {
                    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = 0; w < RP_NW; ++w) {
                        a0 += *(lds_f4*)(pb + w * WS + tok * RP_TPITCH + c8 * 32);
                        a1 += *(lds_f4*)(pb + w * WS + tok * RP_TPITCH + c8 * 32 + 16);
                    }
                    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    rp_u32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x0 = v[2 * e], x1 = v[2 * e + 1];
                        if (p.res) { x0 += h2f_lo(rv[it][e]); x1 += h2f_hi(rv[it][e]); }
                        if (p.mask) {
                            const float m0 = h2f_lo(mv[it][e]), m1 = h2f_hi(mv[it][e]);
                            if (p.act == 2) { x0 *= 1.f - m0 * m0; x1 *= 1.f - m1 * m1; }
                            else { x0 = m0 > 0.f ? x0 : 0.f; x1 = m1 > 0.f ? x1 : 0.f; }
                        } else if (p.act == 1) { x0 = fmaxf(x0, 0.f); x1 = fmaxf(x1, 0.f); }
                        else if (p.act == 2) { x0 = tanhf(x0); x1 = tanhf(x1); }
                        o[e] = pack2h(x0, x1);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(o, xr, (unsigned)((qc + (long long)(t0 + tok) * RP_HID + RP_COLS * slot + 8 * c8) * 2), 0, 16);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the slice has left this CU before its flag does
            if (lane == 0) __hip_atomic_store(myflag, p.base + (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static inline size_t rnn_persist_lds(int tok) { return (size_t)2 * RP_NW * (tok * RP_TPITCH + 32); }

// false: shape not covered (the caller keeps the launch-per-step path).  X must lie below 2 GB from its base (buffer offsets are 32 bit).
static inline bool launch_rnn_persist(hipStream_t st, RnnPersistP p) {
    if (p.B < 1 || p.S < 2) return false;
    p.wpx = (p.B + RP_NG - 1) / RP_NG;
    if (p.wpx > 16 || (long long)p.S * p.B * RP_HID * 2 >= (1ll << 31)) return false;
    static bool attr8 = false, attr16 = false;
    if (p.wpx <= 8) {
        if (!attr8) { hipFuncSetAttribute((const void*)rnn_persist_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rnn_persist_lds(8)); attr8 = true; }
        hipLaunchKernelGGL(rnn_persist_kernel<8>, dim3(RP_NG * RP_SLOTS), dim3(1024), rnn_persist_lds(8), st, p);
    } else {
        if (!attr16) { hipFuncSetAttribute((const void*)rnn_persist_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rnn_persist_lds(16)); attr16 = true; }
        hipLaunchKernelGGL(rnn_persist_kernel<16>, dim3(RP_NG * RP_SLOTS), dim3(1024), rnn_persist_lds(16), st, p);
    }
    return true;
}

}  // namespace HULC_NS
