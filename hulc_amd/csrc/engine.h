// hulc_amd/csrc/engine.h — host-side orchestration of one HULC / GCBC training step on one MI355X.
// Engine<T> owns the workspace (saved activations, packed/transposed weight copies) and enqueues the kernels of
// forward+loss, backward and Adam on one HIP stream.  T = float (parity mode) or h16_t (bench mode).
// Reference call stack restated here: SURVEY.md §3.2 (hulc/models/hulc.py:390-537).
#pragma once
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "iengine.h"
#include "gemm.h"
#include "kernels.h"
#include "conv_wgrad.h"
#include "conv_tile.h"
#include "conv_reg.h"
#include "tr_fused.h"
#include "rnn_persist.h"
#include "enc_tail.h"

namespace HULC_NS {

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
// HULC_DEBUG_SYNC=1: synchronise after every stage and trace its name to stderr (bring-up / fault localisation)
static inline bool hulc_dbg() { static const bool v = HULC_SWITCH("HULC_DEBUG_SYNC", 0) != 0; return v; }
#define STAGE(name) do { if (hulc_dbg()) { hipError_t e_ = hipStreamSynchronize(st); fprintf(stderr, "[hulc] stage %s -> %s\n", name, hipGetErrorString(e_)); fflush(stderr); } } while (0)
#define HIP_CHECK_VOID(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { hulc_set_error("%s:%d %s", __FILE__, __LINE__, hipGetErrorString(e_)); } } while (0)

template <typename T>
struct Engine : IEngine {
    hulc_config cfg;
    // ---- model dims
    static constexpr int EMB = 128, VF = 64, GOAL = 32, LANG = 384, HID = 2048, NCAT = 32, NCLS = 32, NH = 8, FF = 2048,
                         FCH = 4096, NMIX = 10;
    // per model kind: PLAN = width of the two fc_state outputs (32x32 logits; mcil: mean|var of a 256-d Normal), NDIM = mixture
    // dimensions (6 + discrete gripper head; mcil: 7, no gripper head), NHEAD = packed head columns (16-aligned), DE = decoder's slice
    // of the perceptual embedding (perceptual_emb_slice [64,128]; mcil: all 128)
    int PLAN, NDIM, NHEAD, NO, DE;
    bool mcil, gru = false;       // gru: mcil with plan_recognition.rnn_type = nn.GRU (3 gate blocks per BiRNN weight)
    int dec_plan, KIN;
    int maxB, maxS, maxN;
    // ---- bound flat buffers
    float *P = nullptr, *G = nullptr, *AM = nullptr, *AV = nullptr;
    int64_t numel = 0;
    struct Ref { int64_t off, n; };
    std::map<std::string, Ref> tab;
    // ---- arena
    std::vector<void*> allocs;
    int64_t ws_bytes = 0;
    struct Named { void* p; int64_t n; int kind; };   // kind 0: f32, 1: T, 2: int32
    std::map<std::string, Named> named;

    template <typename U> U* alloc(int64_t n, const char* name = nullptr, int kind = -1) {
        void* p = nullptr;
        int64_t bytes = ((n * (int64_t)sizeof(U) + 255) / 256) * 256 + 256;
        if (hipMalloc(&p, bytes) != hipSuccess) { alloc_failed = true; return nullptr; }
        hipMemsetAsync(p, 0, bytes, st);
        allocs.push_back(p);
        ws_bytes += bytes;
        if (name) named[name] = Named{p, n, kind >= 0 ? kind : (std::is_same<U, float>::value ? 0 : (std::is_same<U, int>::value ? 2 : 1))};
        return (U*)p;
    }
    bool alloc_failed = false;

    // ---- weights
    struct LinW {
        const float* W32 = nullptr; const float* b32 = nullptr; float* dW = nullptr; float* db = nullptr;
        T* W = nullptr; T* Wt = nullptr; int N = 0, K = 0; bool own_w = false;
        T* Wfr = nullptr; T* Wtfr = nullptr;      // fragment-ordered copies of W / Wt (add_frag; gemm.h: frag_pack_kernel)
    };
    struct ConvW {
        const float* W32 = nullptr; const float* b32 = nullptr; float* dW = nullptr; float* db = nullptr;
        T* Wf = nullptr; T* Wd = nullptr; int O = 0, I = 0, KH = 0, KW = 0, S = 0; int nhwc = 0;   // nhwc: packed (kh,kw,ci) order (conv2/3)
    };
    struct EncW { ConvW c1, c2, c3; LinW fc7, fc1, fc2; const float *lng = nullptr, *lnb = nullptr; float *dlng = nullptr, *dlnb = nullptr; bool gripper = false; int IH = 0; int H1 = 0, H2 = 0, H3 = 0; };
    EncW encS, encG;
    LinW pp[5], vg[3], lg[3], tr_in[2], tr_out[2], tr_l1[2], tr_l2[2], pr_fc, pr_fs, whh0, wih1, whh1, cl_im0, cl_im2, cl_la0, cl_la2;
    // mcil plan recognition (plan_recognition_net.py:14-42): 2-layer bidirectional tanh RNN; [layer][direction]
    LinW bw_ih[2][2], bw_hh[2][2];
    const float *bb_ih[2][2], *bb_hh[2][2]; float *dbb_ih[2][2], *dbb_hh[2][2];
    T* gcarB2 = nullptr;      // second direct-path buffer of the paired BiGRU BPTT
    T *bZ0[2] = {nullptr, nullptr}, *bH0[2], *bZ1, *bH1, *bh1b, *bxcat, *bdx, *bdH0[2], *bdZ0[2], *bdZ1, *bdz1b, *plan_t;
    float *plan_f, *plan_eps, *plan_eps_in, *klel;
    // GRU variant: per recurrence c (0: layer 0 fwd, 1: layer 0 reverse, 2: layer 1 fwd; 3: the single evaluated step of layer 1 reverse)
    struct GruBuf { T *Zx, *H, *R, *Z, *N, *GN, *dZx, *dG; } gb[4];
    float* gGf = nullptr; T *gcarA = nullptr, *gcarB = nullptr;
    const float *ln_vg_g, *ln_vg_b, *ln_lg_g, *ln_lg_b, *tr_n1g[2], *tr_n1b[2], *tr_n2g[2], *tr_n2b[2], *pos32, *logit_scale;
    float *d_ln_vg_g, *d_ln_vg_b, *d_ln_lg_g, *d_ln_lg_b, *d_tr_n1g[2], *d_tr_n1b[2], *d_tr_n2g[2], *d_tr_n2b[2], *dpos, *dlogit_scale;
    // decoder input weights W_ih0 [HID][KIN] (sub-blocked) + packed heads
    const float *wih0_32, *bih0, *bhh0, *bih1, *bhh1;
    float *dwih0, *dbih0, *dbhh0, *dbih1, *dbhh1;
    T *wih0 = nullptr, *wih0T = nullptr;
    T *wheads = nullptr, *wheadsT = nullptr; float* bheads = nullptr; float *dwheads_tmp = nullptr, *dbheads_tmp = nullptr;
    const float *head_w32[4], *head_b32[4]; float *head_dw[4], *head_db[4]; int head_rows[4] = {60, 60, 60, 2};
    float* dw7_tmp = nullptr;
    bool bound = false;
    T* wshadow = nullptr;                 // bf16 mode: flat compute copy of all parameters (written by the Adam kernel)
    std::vector<TrDesc> trdesc; TrDesc* trdesc_dev = nullptr; int tr_blocks = 0; unsigned short* blk2desc_dev = nullptr;
    // round 5: Adam writes the transposed copies of the Linear weights itself (kernels.h adam_tiled_kernel).  Two sub-tables of `trdesc`: the matrices that are plain
    // views of the 16-bit shadow (tiled by the optimizer) and the rest (packed sources: the permuted fc7, the decoder heads — still transposed by prepare_weights)
    struct TrTable { TrDesc* desc = nullptr; unsigned short* b2d = nullptr; int blocks = 0, n = 0; };
    TrTable tr_adam, tr_rest;
    long long* adam_chunk_start = nullptr; int* adam_chunk_n = nullptr; int adam_chunks = 0;
    bool adam_fuse_tr = true;               // hulc_set_option "adam_fused_transposes"
    void set_adam_fuse(bool on) override { adam_fuse_tr = on; }
    bool frag_by_adam = false;              // every fragment-ordered weight copy is written by adam_tiled_kernel (TrDesc::fr / frT)
    bool rest_by_pack = false;              // weight_pack_kernel also writes the transposed copies of every matrix in tr_rest
    bool tr_fresh = false;                  // set by optim(): the shadow-sourced transposed copies are those of the current parameters
    void tr_table_free(TrTable& t) { if (t.desc) hipFree(t.desc); if (t.b2d) hipFree(t.b2d); t = TrTable{}; }
    bool tr_table_build(TrTable& t, std::vector<TrDesc> v) {
        tr_table_free(t);
        int blk = 0;
        for (TrDesc& d : v) { d.blk0 = blk; blk += d.tiles_x * cdiv(d.R, TRT); }
        t.blocks = blk; t.n = (int)v.size();
        if (v.empty()) return true;
        std::vector<unsigned short> b2d((size_t)blk);
        for (size_t i = 0; i < v.size(); ++i) { const int end = i + 1 < v.size() ? v[i + 1].blk0 : blk; for (int b = v[i].blk0; b < end; ++b) b2d[b] = (unsigned short)i; }
        if (hipMalloc((void**)&t.desc, sizeof(TrDesc) * v.size()) != hipSuccess || hipMalloc((void**)&t.b2d, sizeof(unsigned short) * b2d.size()) != hipSuccess) return false;
        hipMemcpy(t.desc, v.data(), sizeof(TrDesc) * v.size(), hipMemcpyHostToDevice);
        hipMemcpy(t.b2d, b2d.data(), sizeof(unsigned short) * b2d.size(), hipMemcpyHostToDevice);
        return true;
    }
    // the optimizer's tile / chunk tables: matrices of `trdesc` whose source is a contiguous [R][C] view of the shadow with 4-element alignment are tiled; the chunk
    // list covers the rest of [0, numel)
    void adam_tables_build() {
        tr_table_free(tr_adam); tr_table_free(tr_rest); rest_by_pack = false; frag_by_adam = false;
        if (adam_chunk_start) { hipFree(adam_chunk_start); adam_chunk_start = nullptr; } if (adam_chunk_n) { hipFree(adam_chunk_n); adam_chunk_n = nullptr; }
        adam_chunks = 0;
        if (std::is_same<T, float>::value || !wshadow || (numel & 3)) return;
        std::vector<TrDesc> fused, rest;
        std::vector<std::pair<long long, long long>> rng;
        for (const TrDesc& d : trdesc) {
            const T* src = (const T*)d.src;
            const long long off = src - wshadow;
            const bool in = src >= wshadow && off + (long long)d.R * d.C <= numel && d.lds == d.C && (d.C & 3) == 0 && (off & 3) == 0 && !d.cs;
            if (in) { fused.push_back(d); rng.emplace_back(off, off + (long long)d.R * d.C); } else rest.push_back(d);
        }
        std::sort(rng.begin(), rng.end());
        for (size_t i = 0; i + 1 < rng.size(); ++i) if (rng[i].second > rng[i + 1].first) return;       // overlapping views: keep the flat kernel
        if (fused.empty() || fused.size() > 65535) return;
        std::vector<long long> cs; std::vector<int> cn;
        long long pos = 0;
        auto cover = [&](long long lo, long long hi) { for (long long x = lo; x < hi; x += ADAM_CHUNK) { cs.push_back(x); cn.push_back((int)std::min<long long>(ADAM_CHUNK, hi - x)); } };
        for (auto& r : rng) { cover(pos, r.first); pos = r.second; }
        cover(pos, numel);
        if (!tr_table_build(tr_adam, fused) || !tr_table_build(tr_rest, rest)) { tr_table_free(tr_adam); tr_table_free(tr_rest); return; }
        // every transpose the optimizer does not write comes from a matrix weight_pack_kernel packs (the permuted fc7, the decoder heads): it writes their transposed copies as well
        { int lf = 0; for (const TrDesc& d : fused) if (d.fr && d.frT) ++lf; frag_by_adam = fragbatch.n > 0 && 2 * lf == fragbatch.n && lf == frag_linked; }
        rest_by_pack = true;
        for (const TrDesc& d : rest) {
            const bool f7 = d.src == (const void*)encG.fc7.W && d.dst == (void*)encG.fc7.Wt && d.R == 128 && d.C == 3136 && d.ldt == 128;
            const bool hd = d.src == (const void*)wheads && d.dst == (void*)wheadsT && d.R == NHEAD && d.C == HID && d.ldt == NHEAD;
            if (!f7 && !hd) rest_by_pack = false;
        }
        adam_chunks = (int)cs.size();
        if (adam_chunks) {
            if (hipMalloc((void**)&adam_chunk_start, sizeof(long long) * cs.size()) != hipSuccess || hipMalloc((void**)&adam_chunk_n, sizeof(int) * cn.size()) != hipSuccess) { tr_table_free(tr_adam); tr_table_free(tr_rest); adam_chunks = 0; return; }
            hipMemcpy(adam_chunk_start, cs.data(), sizeof(long long) * cs.size(), hipMemcpyHostToDevice);
            hipMemcpy(adam_chunk_n, cn.data(), sizeof(int) * cn.size(), hipMemcpyHostToDevice);
        }
    }

    // ---- workspace (per modality pass)
    struct EncA { T *a1, *a2, *a3, *ss, *g0, *f1; float *ssstats, *f2, *lnst; unsigned* m1bits = nullptr; unsigned* m2bits = nullptr; } aS, aG;
    T *dact1, *dact2, *dact3, *d_g0, *d_f1, *d_f2t; float* d_ss;
    T *d_f1g = nullptr, *d_f2tg = nullptr;      // 16-bit engines: the gripper encoder's own copies (both tails' data gradients run as one launch)
    T *emb, *lang_t, *gl1, *gl2, *goal_t, *ppx, *ppa[4], *xm, *seqf_t, *embg, *Cb, *Zx0, *Zx1, *H0, *H1, *dheads, *dH1, *dZ1, *dH0, *dZ0, *dC;
    float *gl3, *goal_st, *pp_logits, *seqf, *pr_logits, *probs, *klcat, *dpp_kl, *dpr_kl, *Cplan, *heads, *rowloss, *a_tcp;
    int* pidx; int* pidx_in;
    T *xt[3], *qkv[2], *ao[2], *x1t[2], *hff[2];
    float *xf[3], *Pat[2], *y1[2], *st1[2], *x1f[2], *y2[2], *st2[2];
    float* zero_arena = nullptr; int64_t zero_n = 0; bool arena_clean = false;   // arena_clean: the forward's last launch has cleared it (only a backward writes it)
    int* work_ctrs = nullptr; int work_ctr_next = 0;
    int* next_ctr() { return work_ctrs ? work_ctrs + (work_ctr_next++ & 63) : nullptr; }
    float *demb, *dgoal, *dseqf, *dplan, *dprl, *dppx, *dxa, *dxb, *dy_f, *dxm;
    T *dprl_t, *dppl_t, *dseq_t, *dt_a, *dt_b, *dt_c, *dgl3_t;
    T *tA, *tB; int64_t tcap;
    bool h0t_valid = false;                 // tB2 holds H0^T of the current backward (16-bit engines)
    T* tB2 = nullptr;                       // second transposed-operand buffer (the paired layer-1 weight-gradient GEMM reads H1^T and H0^T at once)
    float *part; int64_t partcap; float* cspart;
    // clip
    int* auxrows; T *sf_m, *im1, *g_m, *la1, *img_t, *txt_t; float *img, *txt, *dimg, *dtxt, *dsf_m, *dg_m; T *dimg_t, *dtxt_t, *dim1, *dla1;
    float* losses;   // [8]: 0 action, 1 kl(sum klcat), 2 clip
    // ---- state of the last forward
    hulc_batch cur; float cur_lw = 0, cur_cw = 0; bool have_fwd = false;

    // =====================================================================================================
    Engine(const hulc_config& c) : cfg(c) {
        memset(&cur2, 0, sizeof(cur2));
        mcil = cfg.kind == HULC_KIND_MCIL || cfg.kind == HULC_KIND_MCIL_GRU;
        gru = cfg.kind == HULC_KIND_MCIL_GRU;
        PLAN = mcil ? 512 : 1024; NDIM = mcil ? 7 : 6; NO = NMIX * NDIM; NHEAD = mcil ? 224 : 192; DE = mcil ? EMB : 64;
        if (mcil) { head_rows[0] = head_rows[1] = head_rows[2] = NO; head_rows[3] = 0; }
        dec_plan = cfg.kind == HULC_KIND_GCBC ? 0 : (mcil ? PLAN / 2 : PLAN);
        KIN = dec_plan + DE + GOAL;
        maxB = cfg.max_batch; maxS = cfg.max_seq; maxN = maxB * maxS;
    }
    ~Engine() override { for (void* p : allocs) hipFree(p); if (rp_err_host) hipHostFree((void*)rp_err_host); if (blk2desc_dev) hipFree(blk2desc_dev); if (trdesc_dev) hipFree(trdesc_dev); tr_table_free(tr_adam); tr_table_free(tr_rest); if (adam_chunk_start) hipFree(adam_chunk_start); if (adam_chunk_n) hipFree(adam_chunk_n); }
    int64_t workspace_bytes() const override { return ws_bytes; }
    void set_kl_beta(float b) override { cfg.kl_beta = b; }
    void set_dropout(float p) override { cfg.dropout_p = p; }

    uint64_t site_seed(int site) const {
        uint64_t z = cfg.seed + 0x9E3779B97F4A7C15ull * (cur.step * 64 + (cur.is_lang ? 32 : 0) + site + 1);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
    }

    // ---------------------------------------------------------------- allocation
    void alloc_enc(EncA& a, int IH, bool gripper, const char* pre) {
        const int H1 = (IH - 8) / 4 + 1, H2 = (H1 - 4) / 2 + 1, H3 = H2 - 2;
        std::string s(pre);
        a.a1 = alloc<T>((int64_t)maxN * H1 * H1 * 32, (s + "a1").c_str());
        static const bool use_bits = HULC_SWITCH("HULC_MASKBITS", 1) != 0;
        a.m1bits = (use_bits && std::is_same<T, h16_t>::value) ? alloc<unsigned>((int64_t)maxN * H1 * H1) : nullptr;   // ReLU bitmask of a1 (conv2 dgrad)
        a.a2 = alloc<T>((int64_t)maxN * H2 * H2 * 64, (s + "a2").c_str());
        a.m2bits = (use_bits && std::is_same<T, h16_t>::value) ? alloc<unsigned>((int64_t)maxN * H2 * H2 * 2) : nullptr;   // ReLU bitmask of a2 (conv3 dgrad), emitted by conv2's forward
        a.a3 = alloc<T>((int64_t)maxN * H3 * H3 * 64, (s + "a3").c_str());
        a.ss = gripper ? nullptr : alloc<T>((int64_t)maxN * 128, (s + "ss").c_str());
        a.ssstats = gripper ? nullptr : alloc<float>((int64_t)maxN * 64 * 4);
        a.g0 = gripper ? alloc<T>((int64_t)maxN * 128, (s + "g0").c_str()) : nullptr;
        a.f1 = alloc<T>((int64_t)maxN * 512, (s + "f1").c_str());
        a.f2 = alloc<float>((int64_t)maxN * 64, (s + "f2").c_str());
        a.lnst = alloc<float>((int64_t)maxN * 2);
    }
    int alloc_all() {
        const int64_t N = maxN, B = maxB, S = maxS, SB = (int64_t)maxS * maxB;
        alloc_enc(aS, 200, false, "s_");
        alloc_enc(aG, 84, true, "g_");
        dact1 = alloc<T>(N * 49 * 49 * 32, "dact1"); dact2 = alloc<T>(N * 23 * 23 * 64, "dact2"); dact3 = alloc<T>(N * 21 * 21 * 64, "dact3");
        d_g0 = alloc<T>(N * 128); d_f1 = alloc<T>(N * 512); d_f2t = alloc<T>(N * 64); d_ss = alloc<float>(N * 128, "d_ss");
        if constexpr (std::is_same<T, h16_t>::value) { d_f1g = alloc<T>(N * 512); d_f2tg = alloc<T>(N * 64); }
        emb = alloc<T>(N * EMB, "emb"); lang_t = alloc<T>(B * LANG); gl1 = alloc<T>(B * HID); gl2 = alloc<T>(B * HID);
        gl3 = alloc<float>(B * GOAL, "goal_pre"); goal_t = alloc<T>(B * GOAL, "goal"); goal_st = alloc<float>(B * 2);
        ppx = alloc<T>(B * (EMB + GOAL)); for (int i = 0; i < 4; ++i) ppa[i] = alloc<T>(B * HID);
        pp_logits = alloc<float>(B * PLAN, "pp_logits");
        for (int l = 0; l < 3; ++l) { xt[l] = alloc<T>(N * EMB); xf[l] = alloc<float>(N * EMB, l == 2 ? "pr_x_final" : (l == 0 ? "pr_x0" : "pr_x1")); }
        for (int l = 0; l < 2; ++l) {
            qkv[l] = alloc<T>(N * 3 * EMB); Pat[l] = alloc<float>(B * NH * S * S, l ? "attn_p1" : "attn_p0"); ao[l] = alloc<T>(N * EMB);
            y1[l] = alloc<float>(N * EMB); st1[l] = alloc<float>(N * 2); x1t[l] = alloc<T>(N * EMB); x1f[l] = alloc<float>(N * EMB);
            hff[l] = alloc<T>(N * FF); y2[l] = alloc<float>(N * EMB); st2[l] = alloc<float>(N * 2);
        }
        xm = alloc<T>(B * EMB); seqf = alloc<float>(B * FCH, "seq_feat"); seqf_t = alloc<T>(B * FCH); pr_logits = alloc<float>(B * PLAN, "pr_logits");
        probs = alloc<float>(B * PLAN, "pr_probs"); klcat = alloc<float>(B * NCAT); dpp_kl = alloc<float>(B * PLAN); dpr_kl = alloc<float>(B * PLAN);
        pidx = alloc<int>(B * NCAT, "plan_idx"); pidx_in = alloc<int>(B * NCAT);
        embg = alloc<T>(SB * DE); Cplan = alloc<float>(B * HID); Cb = alloc<T>(B * HID, "dec_cb");
        Zx0 = alloc<T>(SB * HID); Zx1 = alloc<T>(SB * HID); H0 = alloc<T>(SB * HID, "dec_h0"); H1 = alloc<T>(SB * HID, "dec_h1");
        heads = alloc<float>(SB * NHEAD, "heads"); dheads = alloc<T>(SB * NHEAD, "dheads"); rowloss = alloc<float>(SB * 8); a_tcp = alloc<float>(SB * 7, "a_tcp");
        dH1 = alloc<T>(SB * HID); dZ1 = alloc<T>(SB * HID, "dec_dz1"); dH0 = alloc<T>(SB * HID); dZ0 = alloc<T>(SB * HID, "dec_dz0"); dC = alloc<T>(B * HID);
        // backward scratch that must start at zero lives in ONE arena -> a single memset per backward
        {
            auto r64 = [](int64_t n) { return (n + 63) / 64 * 64; };
            zero_n = r64(N * EMB) + r64(B * GOAL) + r64(B * FCH) + r64((int64_t)NHEAD * HID) + r64(NHEAD) + r64(128 * 3136) + 64;
            zero_arena = alloc<float>(zero_n);
            float* q = zero_arena;
            demb = q; q += r64(N * EMB); dgoal = q; q += r64(B * GOAL); dseqf = q; q += r64(B * FCH);
            dwheads_tmp = q; q += r64((int64_t)NHEAD * HID); dbheads_tmp = q; q += r64(NHEAD); dw7_tmp = q; q += r64(128 * 3136);
            work_ctrs = reinterpret_cast<int*>(q);      // 64 zeroed counters per backward: dynamic work claiming of the persistent encoder-backward kernels
            named["demb"] = Named{demb, N * EMB, 0}; named["dgoal"] = Named{dgoal, B * GOAL, 0}; named["dseq_feat"] = Named{dseqf, B * FCH, 0};
        }
        dplan = alloc<float>(B * PLAN, "dplan"); dprl = alloc<float>(B * PLAN, "dpr_logits"); dppx = alloc<float>(B * (EMB + GOAL));
        dxa = alloc<float>(N * EMB, "tr_dx"); dxb = alloc<float>(N * EMB); dy_f = alloc<float>(N * EMB, "tr_dy1"); dxm = alloc<float>(B * EMB);
        dprl_t = alloc<T>(B * PLAN); dppl_t = alloc<T>(B * PLAN); dseq_t = alloc<T>(B * FCH);
        dt_a = alloc<T>(std::max<int64_t>(N * FF, 2 * B * HID)); dt_b = alloc<T>(N * 3 * EMB); dt_c = alloc<T>(N * EMB); dgl3_t = alloc<T>(B * GOAL);
        tcap = std::max<int64_t>(3136 * ((N + 7) / 8 * 8), std::max<int64_t>((gru ? 3 : 1) * HID * ((SB + 7) / 8 * 8), FCH * ((B + 7) / 8 * 8))) + 4096;
        tA = alloc<T>(tcap); tB = alloc<T>(tcap);
        // conv weight-gradient slabs: the 16-bit engines keep the slabs of ALL convolutions of a backward (one batched unpack launch at its end)
        partcap = std::is_same<T, h16_t>::value ? 96ll * 1024 * 1024 : 1024ll * 64 * 576; part = alloc<float>(partcap); cspart = alloc<float>(1024 * 2048);
        auxrows = alloc<int>(B); sf_m = alloc<T>(B * FCH); im1 = alloc<T>(B * 128); g_m = alloc<T>(B * GOAL); la1 = alloc<T>(B * 128);
        img = alloc<float>(B * GOAL, "clip_img"); txt = alloc<float>(B * GOAL, "clip_txt"); img_t = alloc<T>(B * GOAL); txt_t = alloc<T>(B * GOAL);
        dimg = alloc<float>(B * GOAL); dtxt = alloc<float>(B * GOAL); dimg_t = alloc<T>(B * GOAL); dtxt_t = alloc<T>(B * GOAL);
        dim1 = alloc<T>(B * 128); dla1 = alloc<T>(B * 128); dsf_m = alloc<float>(B * FCH); dg_m = alloc<float>(B * GOAL);
        losses = alloc<float>(8);
        if (gru) {
            for (int c = 0; c < 4; ++c) {
                const int64_t rows = c < 3 ? SB : B;
                gb[c].Zx = alloc<T>(rows * 3 * HID); gb[c].R = alloc<T>(rows * HID); gb[c].Z = alloc<T>(rows * HID); gb[c].N = alloc<T>(rows * HID);
                gb[c].GN = alloc<T>(rows * HID); gb[c].dZx = alloc<T>(rows * 3 * HID); gb[c].dG = alloc<T>(rows * 3 * HID);
            }
            gGf = alloc<float>(B * 3 * HID); gcarA = alloc<T>(B * HID); gcarB = alloc<T>(B * HID);
        }
        if (mcil) {
            for (int d = 0; d < 2; ++d) { bZ0[d] = alloc<T>(SB * HID); bH0[d] = alloc<T>(SB * HID, d ? "birnn_h0_rev" : "birnn_h0"); bdH0[d] = alloc<T>(SB * HID); bdZ0[d] = alloc<T>(SB * HID); }
            bZ1 = alloc<T>(SB * HID); bH1 = alloc<T>(SB * HID, "birnn_h1"); bh1b = alloc<T>(B * HID); bxcat = alloc<T>(B * 2 * HID, "birnn_x"); bdx = alloc<T>(B * 2 * HID);
            bdZ1 = alloc<T>(SB * HID); bdz1b = alloc<T>(B * HID);
            plan_t = alloc<T>(B * PLAN / 2); plan_f = alloc<float>(B * PLAN / 2, "plan"); plan_eps = alloc<float>(B * PLAN / 2); plan_eps_in = alloc<float>(B * PLAN / 2);
            klel = alloc<float>(B * PLAN / 2);
        }
        bheads = alloc<float>(NHEAD); wheads = alloc<T>((int64_t)NHEAD * HID); wheadsT = alloc<T>((int64_t)HID * NHEAD);
        if (alloc_failed) { hulc_set_error("hipMalloc failed while sizing the workspace (B=%d S=%d)", maxB, maxS); return 1; }
#ifdef HULC_HALF_F16
        if (!std::is_same<T, float>::value) return scaler_enable(65536.f, 2.f, 0.5f, 2000);      // torch.cuda.amp.GradScaler() defaults
#endif
        return 0;
    }

    // ---------------------------------------------------------------- binding
    bool has(const std::string& n) const { return tab.count(n) > 0; }
    const float* pw(const std::string& n) { return P + tab.at(n).off; }
    float* gw(const std::string& n) { return G + tab.at(n).off; }
    void bind_lin(LinW& L, const std::string& name, int N, int K, bool bias_suffix = true) {
        L.W32 = pw(name + (bias_suffix ? ".weight" : "")); L.dW = gw(name + (bias_suffix ? ".weight" : ""));
        if (bias_suffix) { L.b32 = pw(name + ".bias"); L.db = gw(name + ".bias"); }
        L.N = N; L.K = K;
        if (std::is_same<T, float>::value) L.W = (T*)L.W32;
        else L.W = wshadow + (L.W32 - P);
        L.own_w = false;
        if (!L.Wt) L.Wt = alloc<T>((int64_t)N * K);
        add_tr(L.W32, L.W, L.Wt, N, K);
    }
    // fragment-ordered weight copies (gemm.h: frag_pack_kernel): the recurrent weights of the GRU plan encoder and the plan-recognition transformer's
    // weights, whose kernels read MFMA fragments straight from global memory.  Jobs are collected at bind time, one batched launch in prepare_weights
    FragPackBatch fragbatch{};
    int frag_blocks = 0;
    void add_frag(LinW& L) {
        if constexpr (std::is_same<T, h16_t>::value) {
            if (!L.Wfr) L.Wfr = alloc<T>((int64_t)L.N * L.K);
            if (!L.Wtfr) L.Wtfr = alloc<T>((int64_t)L.N * L.K);
            if (fragbatch.n + 2 > FRAG_PACK_MAX) return;
            FragPackBatch& fb = fragbatch;
            fb.src[fb.n] = L.W; fb.dst[fb.n] = L.Wfr; fb.N[fb.n] = L.N; fb.K[fb.n] = L.K; fb.blk0[fb.n] = frag_blocks; frag_blocks += frag_pack_blocks(L.N, L.K); ++fb.n;
            fb.src[fb.n] = L.Wt; fb.dst[fb.n] = L.Wtfr; fb.N[fb.n] = L.K; fb.K[fb.n] = L.N; fb.blk0[fb.n] = frag_blocks; frag_blocks += frag_pack_blocks(L.K, L.N); ++fb.n;
            fb.blk0[fb.n] = frag_blocks;
            // the optimizer's tile pass can write both copies (kernels.h adam_tiled_kernel): remember them on the weight's transpose descriptor
            for (TrDesc& d : trdesc)
                if (d.dst == (void*)L.Wt && d.R == L.N && d.C == L.K && (L.N % 32) == 0 && (L.K % 32) == 0) { d.fr = L.Wfr; d.frT = L.Wtfr; ++frag_linked; }
        }
    }
    int frag_linked = 0;                    // weights whose fragment-ordered copies ride on their transpose descriptor (2 frag jobs each)
    static bool frag_weights() { static const bool on = HULC_SWITCH("HULC_FRAG_W", 1) != 0; return on; }
    static constexpr int TRT = std::is_same<T, float>::value ? 32 : 64;     // transpose tile (bf16: 64x64, 16-byte accesses)
    void add_tr(const float* w32, const T* w, T* wt, int R, int C) {
        TrDesc d; d.src = std::is_same<T, float>::value ? (const void*)w32 : (const void*)w; d.dst = wt; d.lds = C; d.ldt = R; d.R = R; d.C = C;
        d.tiles_x = cdiv(C, TRT); d.blk0 = tr_blocks; tr_blocks += d.tiles_x * cdiv(R, TRT);
        trdesc.push_back(d);
    }
    // a transpose whose source is a packed compute-type buffer of this engine (not a parameter): T -> T in both engine kinds
    void add_tr_packed(const T* src, T* wt, int R, int C) {
        TrDesc d; d.src = src; d.dst = wt; d.lds = C; d.ldt = R; d.R = R; d.C = C;
        d.tiles_x = cdiv(C, TRT); d.blk0 = tr_blocks; tr_blocks += d.tiles_x * cdiv(R, TRT);
        trdesc.push_back(d);
    }
    void bind_conv(ConvW& c, const std::string& name, int O, int I, int K, int S, int nhwc) {
        c.W32 = pw(name + ".weight"); c.b32 = pw(name + ".bias"); c.dW = gw(name + ".weight"); c.db = gw(name + ".bias");
        c.O = O; c.I = I; c.KH = c.KW = K; c.S = S; c.nhwc = nhwc;
        if (!c.Wf) c.Wf = alloc<T>((int64_t)O * I * K * K);
        if (nhwc && !c.Wd) c.Wd = alloc<T>((int64_t)O * I * K * K);
    }
    void bind_enc(EncW& e, const std::string& pre, bool gripper, int IH) {
        e.gripper = gripper; e.IH = IH; e.H1 = (IH - 8) / 4 + 1; e.H2 = (e.H1 - 4) / 2 + 1; e.H3 = e.H2 - 2;
        bind_conv(e.c1, pre + "conv_model.0", 32, 3, 8, 4, 0);
        bind_conv(e.c2, pre + "conv_model.2", 64, 32, 4, 2, 1);
        bind_conv(e.c3, pre + "conv_model.4", 64, 64, 3, 1, 1);
        if (gripper) {
            // fc7 consumes the NHWC flatten: keep packed copies (W, Wt own storage even in fp32 mode)
            e.fc7.W32 = pw(pre + "conv_model.7.weight"); e.fc7.b32 = pw(pre + "conv_model.7.bias");
            e.fc7.dW = gw(pre + "conv_model.7.weight"); e.fc7.db = gw(pre + "conv_model.7.bias");
            e.fc7.N = 128; e.fc7.K = 3136;
            if (!e.fc7.W) { e.fc7.W = alloc<T>(128 * 3136); e.fc7.Wt = alloc<T>(128 * 3136); e.fc7.own_w = true; }
            add_tr_packed(e.fc7.W, e.fc7.Wt, 128, 3136);      // transposed copy of the PERMUTED weight (weight_pack_kernel writes it just before the batched transpose)
        }
        bind_lin(e.fc1, pre + "fc1.0", 512, 128);
        bind_lin(e.fc2, pre + "fc2", 64, 512);
        e.lng = pw(pre + "ln.weight"); e.lnb = pw(pre + "ln.bias"); e.dlng = gw(pre + "ln.weight"); e.dlnb = gw(pre + "ln.bias");
    }
    int bind(float* p, float* g, float* m, float* v, int64_t n_, int n, const char* const* names, const int64_t* offs,
             const int64_t* numels) override {
        P = p; G = g; AM = m; AV = v; numel = n_;
        tab.clear(); trdesc.clear(); tr_blocks = 0; fragbatch = FragPackBatch{}; frag_blocks = 0; frag_linked = 0;
        if (!std::is_same<T, float>::value && !wshadow) wshadow = alloc<T>(numel);
        for (int i = 0; i < n; ++i) tab[names[i]] = Ref{offs[i], numels[i]};
        tab_order.assign(tab.begin(), tab.end());
        std::sort(tab_order.begin(), tab_order.end(), [](const std::pair<std::string, Ref>& a, const std::pair<std::string, Ref>& b) { return a.second.off < b.second.off; });
        // data parallelism: the "this step's gradients are garbage" vote rides in an alignment-padding element of the LAST bucket (see skip_vote_put)
        lazy.clear();
        skip_pad = -1;
        // (only a REAL padding element qualifies — hulc_bind_params accepts any 4-aligned layout, so the element behind a tensor is
        //  padding only if the next tensor of the table starts later; a tightly packed layout has no vote word and the vote falls back to skip_pad = -1)
        for (size_t i = 0; i < tab_order.size(); ++i) {
            const auto& kv = tab_order[i];
            const int64_t end = kv.second.off + kv.second.n, nxt = i + 1 < tab_order.size() ? tab_order[i + 1].second.off : numel;
            if (kv.first.compare(0, 19, "perceptual_encoder.") == 0 && end < nxt && end < numel) { skip_pad = end; break; }
        }
        try {
            bind_enc(encS, "perceptual_encoder.rgb_static_encoder.", false, 200);
            bind_enc(encG, "perceptual_encoder.rgb_gripper_encoder.", true, 84);
            const char* ppn[5] = {"plan_proposal.fc_model.0", "plan_proposal.fc_model.2", "plan_proposal.fc_model.4", "plan_proposal.fc_model.6",
                                  "plan_proposal.fc_state.0"};
            const int ppN[5] = {HID, HID, HID, HID, PLAN}, ppK[5] = {EMB + GOAL, HID, HID, HID, HID};
            for (int i = 0; i < 5; ++i) bind_lin(pp[i], ppn[i], ppN[i], ppK[i]);
            const char* vgn[3] = {"visual_goal.mlp.0", "visual_goal.mlp.2", "visual_goal.mlp.4"};
            const char* lgn[3] = {"language_goal.mlp.1", "language_goal.mlp.3", "language_goal.mlp.5"};
            const int gN[3] = {HID, HID, GOAL};
            const int vK[3] = {EMB, HID, HID}, lK[3] = {LANG, HID, HID};
            for (int i = 0; i < 3; ++i) { bind_lin(vg[i], vgn[i], gN[i], vK[i]); bind_lin(lg[i], lgn[i], gN[i], lK[i]); }
            ln_vg_g = pw("visual_goal.ln.weight"); ln_vg_b = pw("visual_goal.ln.bias"); d_ln_vg_g = gw("visual_goal.ln.weight"); d_ln_vg_b = gw("visual_goal.ln.bias");
            ln_lg_g = pw("language_goal.ln.weight"); ln_lg_b = pw("language_goal.ln.bias"); d_ln_lg_g = gw("language_goal.ln.weight"); d_ln_lg_b = gw("language_goal.ln.bias");
            const std::string pr = "plan_recognition.";
            if (mcil) {
                for (int l = 0; l < 2; ++l)
                    for (int d = 0; d < 2; ++d) {
                        const std::string sfx = "_l" + std::to_string(l) + (d ? "_reverse" : "");
                        const std::string bp = pr + "birnn_model.";
                        bind_lin(bw_ih[l][d], bp + "weight_ih" + sfx, (gru ? 3 : 1) * HID, l ? 2 * HID : EMB, false);
                        bind_lin(bw_hh[l][d], bp + "weight_hh" + sfx, (gru ? 3 : 1) * HID, HID, false);
                        if (gru && frag_weights()) add_frag(bw_hh[l][d]);
                        bb_ih[l][d] = pw(bp + "bias_ih" + sfx); bb_hh[l][d] = pw(bp + "bias_hh" + sfx);
                        dbb_ih[l][d] = gw(bp + "bias_ih" + sfx); dbb_hh[l][d] = gw(bp + "bias_hh" + sfx);
                    }
            } else {
            pos32 = pw(pr + "position_embeddings.weight"); dpos = gw(pr + "position_embeddings.weight");
            for (int l = 0; l < 2; ++l) {
                const std::string L = pr + "transformer_encoder.layers." + std::to_string(l) + ".";
                tr_in[l].W32 = pw(L + "self_attn.in_proj_weight"); tr_in[l].dW = gw(L + "self_attn.in_proj_weight");
                tr_in[l].b32 = pw(L + "self_attn.in_proj_bias"); tr_in[l].db = gw(L + "self_attn.in_proj_bias");
                tr_in[l].N = 3 * EMB; tr_in[l].K = EMB;
                tr_in[l].W = std::is_same<T, float>::value ? (T*)tr_in[l].W32 : wshadow + (tr_in[l].W32 - P);
                if (!tr_in[l].Wt) tr_in[l].Wt = alloc<T>(3 * EMB * EMB);
                add_tr(tr_in[l].W32, tr_in[l].W, tr_in[l].Wt, 3 * EMB, EMB);
                bind_lin(tr_out[l], L + "self_attn.out_proj", EMB, EMB);
                bind_lin(tr_l1[l], L + "linear1", FF, EMB);
                bind_lin(tr_l2[l], L + "linear2", EMB, FF);
                for (LinW* w : {&tr_in[l], &tr_out[l], &tr_l1[l], &tr_l2[l]}) add_frag(*w);      // the fused layer kernels (tr_fused.h) read these
                tr_n1g[l] = pw(L + "norm1.weight"); tr_n1b[l] = pw(L + "norm1.bias"); d_tr_n1g[l] = gw(L + "norm1.weight"); d_tr_n1b[l] = gw(L + "norm1.bias");
                tr_n2g[l] = pw(L + "norm2.weight"); tr_n2b[l] = pw(L + "norm2.bias"); d_tr_n2g[l] = gw(L + "norm2.weight"); d_tr_n2b[l] = gw(L + "norm2.bias");
            }
            bind_lin(pr_fc, pr + "fc", FCH, EMB);
            }
            bind_lin(pr_fs, pr + "fc_state.0", PLAN, FCH);
            const std::string ad = "action_decoder.";
            wih0_32 = pw(ad + "rnn.weight_ih_l0"); dwih0 = gw(ad + "rnn.weight_ih_l0");
            bih0 = pw(ad + "rnn.bias_ih_l0"); bhh0 = pw(ad + "rnn.bias_hh_l0"); bih1 = pw(ad + "rnn.bias_ih_l1"); bhh1 = pw(ad + "rnn.bias_hh_l1");
            dbih0 = gw(ad + "rnn.bias_ih_l0"); dbhh0 = gw(ad + "rnn.bias_hh_l0"); dbih1 = gw(ad + "rnn.bias_ih_l1"); dbhh1 = gw(ad + "rnn.bias_hh_l1");
            wih0 = std::is_same<T, float>::value ? (T*)wih0_32 : wshadow + (wih0_32 - P);
            if (!wih0T) wih0T = alloc<T>((int64_t)HID * KIN);
            add_tr(wih0_32, wih0, wih0T, HID, KIN);
            bind_lin(whh0, ad + "rnn.weight_hh_l0", HID, HID, false);
            bind_lin(wih1, ad + "rnn.weight_ih_l1", HID, HID, false);
            bind_lin(whh1, ad + "rnn.weight_hh_l1", HID, HID, false);
            add_tr_packed(wheads, wheadsT, NHEAD, HID);      // the packed heads' transposed copy rides on the batched transpose as well
            const char* hn[4] = {"prob_fc", "mean_fc", "log_scale_fc", "gripper_fc"};
            for (int i = 0; i < 4; ++i) { head_w32[i] = head_b32[i] = nullptr; head_dw[i] = head_db[i] = nullptr; }
            for (int i = 0; i < (mcil ? 3 : 4); ++i) {
                head_w32[i] = pw(ad + hn[i] + ".weight"); head_b32[i] = pw(ad + hn[i] + ".bias");
                head_dw[i] = gw(ad + hn[i] + ".weight"); head_db[i] = gw(ad + hn[i] + ".bias");
            }
            if (cfg.use_clip) {
                bind_lin(cl_im0, "proj_vis_lang.mlp_im.0", 128, FCH); bind_lin(cl_im2, "proj_vis_lang.mlp_im.2", GOAL, 128);
                bind_lin(cl_la0, "proj_vis_lang.mlp_lang.0", 128, GOAL); bind_lin(cl_la2, "proj_vis_lang.mlp_lang.2", GOAL, 128);
                logit_scale = pw("logit_scale"); dlogit_scale = gw("logit_scale");
            }
        } catch (const std::out_of_range&) {
            hulc_set_error("hulc_bind_params: a required parameter name is missing from the table");
            return 1;
        }
        // the weight gradients every writer of which can STORE (see LazyG): the M = B MLPs and the decoder's 2048^2 recurrent / layer-1 input weights
        for (int i = 0; i < 5; ++i) lazy_register(pp[i]);
        for (int i = 0; i < 3; ++i) { lazy_register(vg[i]); lazy_register(lg[i]); }
        lazy_register(pr_fs); lazy_register(whh0); lazy_register(whh1); lazy_register(wih1);
        std::sort(lazy.begin(), lazy.end(), [](const LazyG& a, const LazyG& b) { return a.off < b.off; });
        for (size_t i = 0; i + 1 < lazy.size(); ++i) if (lazy[i].off + lazy[i].n > lazy[i + 1].off) { lazy.clear(); break; }      // overlapping views: no lazy set
        if (blk2desc_dev) { hipFree(blk2desc_dev); blk2desc_dev = nullptr; }
        {
            std::vector<unsigned short> b2d((size_t)std::max(tr_blocks, 1));
            for (size_t i = 0; i < trdesc.size(); ++i) {
                const int end = i + 1 < trdesc.size() ? trdesc[i + 1].blk0 : tr_blocks;
                for (int b = trdesc[i].blk0; b < end; ++b) b2d[b] = (unsigned short)i;
            }
            if (hipMalloc((void**)&blk2desc_dev, b2d.size() * sizeof(unsigned short)) != hipSuccess) alloc_failed = true;
            else hipMemcpy(blk2desc_dev, b2d.data(), b2d.size() * sizeof(unsigned short), hipMemcpyHostToDevice);
        }
        if (trdesc_dev) { hipFree(trdesc_dev); trdesc_dev = nullptr; }
        if (hipMalloc((void**)&trdesc_dev, sizeof(TrDesc) * trdesc.size()) != hipSuccess) alloc_failed = true;
        else hipMemcpy(trdesc_dev, trdesc.data(), sizeof(TrDesc) * trdesc.size(), hipMemcpyHostToDevice);
        adam_tables_build();
        if (alloc_failed) { hulc_set_error("hipMalloc failed while allocating weight copies"); return 1; }
        bound = true;
        return prepare_weights();
    }

    HeadPack head_pack() const {
        HeadPack hp;
        for (int i = 0; i < 4; ++i) { hp.w[i] = head_w32[i]; hp.b[i] = head_b32[i]; hp.dw[i] = head_dw[i]; hp.db[i] = head_db[i]; hp.rows[i] = head_rows[i]; }
        return hp;
    }
    // ---------------------------------------------------------------- small launch helpers
    template <typename TS, typename TD>
    void cast_tr(const TS* src, long long lds_, TD* dst, long long ldd, TD* dstT, long long ldt, int R, int C) {
        dim3 grid(cdiv(C, 32), cdiv(R, 32));
        hipLaunchKernelGGL((cast_transpose_kernel<TS, TD>), grid, dim3(256), 0, st, src, lds_, dst, ldd, dstT, ldt, R, C);
    }
    static int ldpad(int m) { return (m + 7) / 8 * 8; }
    // dstA[c][r] = srcA[r][c] and dstB likewise, one launch
    void transpose_pair(const T* a, long long lda, T* at, int Ra, int Ca, const T* b, long long ldb, T* bt, int Rb, int Cb, long long ldt,
                        float* cs = nullptr, float* cs2 = nullptr) {
        TrPair p;
        p.d[0].cs = cs; p.d[0].cs2 = cs2;
        p.d[0].src = a; p.d[0].dst = at; p.d[0].lds = lda; p.d[0].ldt = ldt; p.d[0].R = Ra; p.d[0].C = Ca; p.d[0].tiles_x = cdiv(Ca, TRT); p.d[0].blk0 = 0;
        const int n0 = p.d[0].tiles_x * cdiv(Ra, TRT);
        p.d[1].src = b; p.d[1].dst = bt; p.d[1].lds = ldb; p.d[1].ldt = ldt; p.d[1].R = Rb; p.d[1].C = Cb; p.d[1].tiles_x = cdiv(Cb, TRT); p.d[1].blk0 = n0;
        const dim3 grid(n0 + p.d[1].tiles_x * cdiv(Rb, TRT));
        if constexpr (std::is_same<T, float>::value) hipLaunchKernelGGL((pair_transpose_kernel<T>), grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(pair_transpose64_kernel, grid, dim3(256), 0, st, p);
    }
    // 16-bit engines: three transposes, one launch (dst leading dimension ldt for all; cs / cs2 = fused column sums of the FIRST source)
    void transpose_triple(const T* a, long long lda, T* at, int Ra, int Ca, const T* b, long long ldb, T* bt, int Rb, int Cb, const T* c, long long ldc, T* ct, int Rc, int Cc,
                          long long ldt, float* cs = nullptr, float* cs2 = nullptr) {
        if constexpr (!std::is_same<T, float>::value) {
            TrMulti m{}; m.n = 3;
            const T* src[3] = {a, b, c}; T* dst[3] = {at, bt, ct}; const long long ld[3] = {lda, ldb, ldc}; const int R[3] = {Ra, Rb, Rc}, C[3] = {Ca, Cb, Cc};
            int blk = 0;
            for (int i = 0; i < 3; ++i) {
                m.d[i].src = src[i]; m.d[i].dst = dst[i]; m.d[i].lds = ld[i]; m.d[i].ldt = ldt; m.d[i].R = R[i]; m.d[i].C = C[i]; m.d[i].tiles_x = cdiv(C[i], TRT); m.d[i].blk0 = blk;
                blk += m.d[i].tiles_x * cdiv(R[i], TRT);
            }
            m.d[0].cs = cs; m.d[0].cs2 = cs2;
            hipLaunchKernelGGL(multi_transpose64_kernel, dim3(blk), dim3(256), 0, st, m);
        }
    }
    template <typename TS, typename TD>
    void copy2d(const TS* src, long long lds_, TD* dst, long long ldd, int R, int C, int acc, float scale = 1.f) {
        hipLaunchKernelGGL((copy2d_kernel<TS, TD>), dim3(cdiv((long long)R * C, 256)), dim3(256), 0, st, src, lds_, dst, ldd, R, C, acc, scale);
    }
    void colsum(const T* x, long long ld, int M, int N, float* out, float* out2 = nullptr, float scale = 1.f) {
        if (ld == N && (N == 32 || N == 64) && M >= 4096) {          // conv bias grads: flat 16-byte streaming
            const int nblk = 256;
            hipLaunchKernelGGL((colsum_flat_kernel<T>), dim3(nblk), dim3(256), 0, st, x, (long long)M * N, N, cspart);
            hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 64)), dim3(256), 0, st, cspart, nblk, N, out, out2, scale);
            return;
        }
        const int nsplit = std::max(1, std::min(256, cdiv(M, 256)));
        const int rps = cdiv(M, nsplit);
        if constexpr (std::is_same<T, float>::value) {
            // fp32 (parity) mode stays bit-reproducible: two-stage, deterministic
            hipLaunchKernelGGL((colsum_kernel<T>), dim3(cdiv(N, 64), nsplit), dim3(256), 0, st, x, ld, M, N, cspart, rps, 0, 1.f, (float*)nullptr);
            hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(N, 64)), dim3(256), 0, st, cspart, nsplit, N, out, out2, scale);
        } else {
            // bf16 (bench) mode: one launch, each row chunk adds its column sums with fp32 atomics
            hipLaunchKernelGGL((colsum_kernel<T>), dim3(cdiv(N, 64), nsplit), dim3(256), 0, st, x, ld, M, N, out, rps, 2, scale, out2);
        }
    }
    // dense NT GEMM with tile selection
    // ---- grouped launches (gemm.h gemm_glds_group_kernel): between gemm_group_begin() and gemm_group_end() up to three INDEPENDENT 128 x 128-tile products are
    // collected and issued as one grid; anything else that arrives in between flushes the queue first, so program order is kept.  The caller vouches for the
    // independence of what it brackets (no queued product reads another's output).
    GemmGroupP gq{}; bool gq_open = false; double gq_fl = 0, gq_by = 0;
    void gemm_group_begin() { static const int sw = HULC_SWITCH("HULC_GEMM_GROUP", 1); gq_open = sw != 0 && gemm_group_mode; gq.n = 0; gq_fl = gq_by = 0; }
    void gemm_group_flush() {
        if constexpr (std::is_same<T, h16_t>::value) {
            if (gq.n == 1) { TimerScope ts(this, "gemm_128x128", "mfma", gq_fl, gq_by); launch_gemm_glds(st, gq.a[0], gq.b[0], gq.om[0], gq.ep[0], gq.M[0], gq.N[0], gq.K[0]); }
            else if (gq.n > 1) { TimerScope ts(this, "gemm_128x128", "mfma", gq_fl, gq_by, 1); launch_gemm_glds_group(st, gq); }
        }
        gq.n = 0; gq_fl = gq_by = 0;
    }
    void gemm_group_end() { gemm_group_flush(); gq_open = false; }
    void gemm(const DenseLoader<T>& a, const DenseLoader<T>& b, const DenseOut& om, const EpiP& ep, int M, int N, int K) {
        static const bool trace = HULC_SWITCH("HULC_TRACE_GEMM", 0) != 0;
        if constexpr (std::is_same<T, h16_t>::value) {
            if (gq_open) {
                const long long w128g = (long long)cdiv(M, 128) * cdiv(N, 128);
                const bool skinny = a.R1 == 0x7fffffff && b.R1 == 0x7fffffff && ep.z_stride == 0 && skinny_ok(M, N, K, a.s1, b.s1, a.p, b.p);
                if (!skinny && M >= 512 && N >= 128 && w128g >= 128 && gemm_use_glds && gemm_glds_ok(a, b, ep, M, N, K)) {
                    if (gq.n == 3) gemm_group_flush();
                    const int i = gq.n++;
                    gq.a[i] = a; gq.b[i] = b; gq.om[i] = om; gq.ep[i] = ep; gq.M[i] = M; gq.N[i] = N; gq.K[i] = K;
                    gq_fl += 2.0 * M * N * K; gq_by += ((double)M * K + (double)N * K + (double)M * N) * sizeof(T);
                    return;
                }
                gemm_group_flush();
            }
        }
        if (trace) fprintf(stderr, "[gemm] M=%d N=%d K=%d lda=%lld ldb=%lld f32out=%d acc=%d atomic=%d\n", M, N, K, a.s1, b.s1, ep.out_f32, ep.accumulate, ep.atomic);
        const double fl = 2.0 * M * N * K, by = ((double)M * K + (double)N * K + (double)M * N) * sizeof(T);
        if constexpr (std::is_same<T, h16_t>::value) {
            if (a.R1 == 0x7fffffff && b.R1 == 0x7fffffff && ep.z_stride == 0 && skinny_ok(M, N, K, a.s1, b.s1, a.p, b.p)) {
                // two roles share the skinny kernels: M <= 64 layers stream a whole weight matrix per launch (weight-bound), many-row GEMMs
                // (small-N heads / weight gradients with K = 2048) stream activations
                TimerScope ts(this, M <= 64 ? "skinny_gemm_m64" : "skinny_gemm_rows", "hbm", fl, by);
                launch_skinny(st, a.p, a.s1, b.p, b.s1, M, N, K, om, ep);
                return;
            }
        }
        // largest tile that still yields >= 128 workgroups (small-N transformer / encoder GEMMs are latency-bound otherwise)
        const long long w128 = (long long)cdiv(M, 128) * cdiv(N, 128), w64 = (long long)cdiv(M, 64) * cdiv(N, 64);
        if (M >= 512 && N >= 128 && w128 >= 128) {
            TimerScope ts(this, "gemm_128x128", "mfma", fl, by);
            if constexpr (std::is_same<T, h16_t>::value) {
                if (gemm_use_glds && gemm_glds_ok(a, b, ep, M, N, K)) { launch_gemm_glds(st, a, b, om, ep, M, N, K); return; }
                if (K >= 128) { launch_gemm<T, 128, 128, DenseLoader<T>, DenseLoader<T>, DenseOut, 64>(st, a, b, om, ep, M, N, K); return; }   // BK = 64: half the barriers per flop
            }
            launch_gemm<T, 128, 128>(st, a, b, om, ep, M, N, K);
        }
        else {
            // small-N / short-K GEMMs (transformer, encoder heads) are bound by the exposed L2 latency of each k-step: a deeper BK means fewer of them
            static const int small_bk = HULC_SWITCH("HULC_SMALL_BK", 128);     // A/B on one box: 4.764 (32) / 4.739 (64) / 4.728 ms per step (128)
            const bool t64 = w64 >= 128 || (M <= 64 && N <= 64);
            if constexpr (std::is_same<T, h16_t>::value) {
                if (small_bk == 128 && K >= 128) {
                    if (t64) launch_gemm<T, 64, 64, DenseLoader<T>, DenseLoader<T>, DenseOut, 128>(st, a, b, om, ep, M, N, K);
                    else launch_gemm<T, 32, 32, DenseLoader<T>, DenseLoader<T>, DenseOut, 128>(st, a, b, om, ep, M, N, K);
                    return;
                }
                if (small_bk == 64 && K >= 64) {
                    if (t64) launch_gemm<T, 64, 64, DenseLoader<T>, DenseLoader<T>, DenseOut, 64>(st, a, b, om, ep, M, N, K);
                    else launch_gemm<T, 32, 32, DenseLoader<T>, DenseLoader<T>, DenseOut, 64>(st, a, b, om, ep, M, N, K);
                    return;
                }
            }
            if (t64) launch_gemm<T, 64, 64>(st, a, b, om, ep, M, N, K);
            else launch_gemm<T, 32, 32>(st, a, b, om, ep, M, N, K);
        }
    }
    // dW[M][N] += A[M][K] B[N][K]^T with fp32 accumulate; few output tiles + long K -> split K across workgroups (atomics)
    void gemm_wgrad(const DenseLoader<T>& a, const DenseLoader<T>& b, float* dW, long long lddw, int M, int N, int K) {
        EpiP ep = epi(dW, true); ep.accumulate = 1;
        if constexpr (std::is_same<T, h16_t>::value) {
            if (a.R1 == 0x7fffffff && b.R1 == 0x7fffffff && skinny_ok(M, N, K, a.s1, b.s1, a.p, b.p)) { gemm(a, b, dense_out(lddw), ep, M, N, K); return; }
        }
        const long long t128 = (long long)cdiv(M, 128) * cdiv(N, 128), t64 = (long long)cdiv(M, 64) * cdiv(N, 64);
        const bool big = M >= 128 && N >= 128;
        const long long tiles = big ? t128 : t64;
        int nsplit = 1;
        if (!std::is_same<T, float>::value && tiles < 128 && K >= 512) nsplit = (int)std::min<long long>(std::min<long long>(256 / tiles, K / 256), 32);
        if (nsplit <= 1) { gemm(a, b, dense_out(lddw), ep, M, N, K); return; }
        ep.atomic = 1;
        TimerScope ts(this, big ? "gemm_128x128" : "gemm_64x64_splitk", "mfma", 2.0 * M * N * K, ((double)M * K + (double)N * K) * sizeof(T) + 4.0 * M * N);
        if (big) launch_gemm<T, 128, 128>(st, a, b, dense_out(lddw), ep, M, N, K, 1, nsplit);
        else launch_gemm<T, 64, 64>(st, a, b, dense_out(lddw), ep, M, N, K, 1, nsplit);
    }
    EpiP epi(void* out, bool f32) const { EpiP e; e.out = out; e.out_f32 = f32 ? 1 : 0; e.generic_only = epilogue_fast ? 0 : 1; return e; }

    // Y[M][N] = X[M][K] W^T (+bias) ...
    void lin_fwd(const T* X, long long ldx, int M, const LinW& L, EpiP ep, long long ldo) {
        if (!ep.bias) ep.bias = L.b32;
        gemm(dense<T>(X, M, ldx), dense<T>(L.W, L.N, L.K), dense_out(ldo), ep, M, L.N, L.K);
    }
    // weight + bias grads of Y = X W^T: dW[N][K] += dY^T X ; db += colsum(dY).  dY [M][N] dense, X [M][K] (ldx)
    void lin_wgrad(const T* dY, const T* X, long long ldx, int M, int N, int K, float* dW, long long lddw, float* db, float* db2 = nullptr) {
        if constexpr (std::is_same<T, h16_t>::value) {
            if (M <= 64) {          // one fused launch: tr-read wgrad + bias grad, no transposed copies
                hipLaunchKernelGGL(lin_bwd_smallm_kernel, dim3(cdiv(N, 64), cdiv(K, 128)), dim3(256), 0, st, dY, (long long)N, X, ldx, M, N, K, dW, lddw, db, db2, 64, grad_first(dW) ? 1 : 0);
                return;
            }
            grad_ensure_zero(dW);      // the paths below accumulate
            static const bool fused_largem = HULC_SWITCH("HULC_LINBWD_LARGEM", 0) != 0;   // measured 0.25 ms/step SLOWER than transposes + NT GEMM (A/B, same box): off
            if (fused_largem && (long long)N * K <= 2048ll * 512) {
                // token-major layers (M = B*S): the same kernel, rows split over blockIdx.z (~256 workgroups), partials by atomics
                const int tiles = cdiv(N, 64) * cdiv(K, 128);
                const int z = std::max(1, std::min(cdiv(M, 64), cdiv(256, tiles)));
                const int mchunk = cdiv(cdiv(M, z), 64) * 64;
                hipLaunchKernelGGL(lin_bwd_smallm_kernel, dim3(cdiv(N, 64), cdiv(K, 128), cdiv(M, mchunk)), dim3(256), 0, st, dY, (long long)N, X, ldx, M, N, K, dW, lddw,
                                   db, db2, mchunk);
                return;
            }
        }
        const int mp = ldpad(M);
        constexpr bool fuse_cs = std::is_same<T, h16_t>::value;     // bf16 (bench) mode: the dY transpose also adds its column sums into db (atomics)
        transpose_pair(dY, N, tA, M, N, X, ldx, tB, M, K, mp, fuse_cs ? db : nullptr, fuse_cs ? db2 : nullptr);
        gemm_wgrad(dense<T>(tA, N, mp), dense<T>(tB, K, mp), dW, lddw, N, K, M);
        if (db && !fuse_cs) colsum(dY, N, M, N, db, db2);
    }
    // dX[M][K] = dY[M][N] W   (via the transposed copy Wt [K][N])
    void lin_dgrad(const T* dY, int M, const LinW& L, EpiP ep, const DenseOut& om) {
        gemm(dense<T>(dY, M, L.N), dense<T>(L.Wt, L.K, L.N), om, ep, M, L.K, L.N);
    }
    void ln_fwd(const float* x, long long ldx, int rows, int n, const float* g, const float* b, T* out, long long ldo, float* outf, long long ldf,
                float* stats) {
        hipLaunchKernelGGL((layernorm_fwd_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), 0, st, x, ldx, rows, n, g, b, out, ldo, outf, ldf, stats);
    }
    // bcast_rows > 0 (fused kernel only, see ln_bwd_can_bcast): dy holds one row per WINDOW, broadcast over its bcast_rows rows and divided by bcast_div
    static constexpr bool ln_bwd_can_bcast = std::is_same<T, h16_t>::value;
    void ln_bwd(const float* dy, long long lddy, const float* x, long long ldx, const float* stats, const float* g, int rows, int n, float* dxf,
                long long ldd, int acc, T* dxt, long long ldt, float* dg, float* db, float drop_p = 0.f, unsigned long long drop_seed = 0,
                int bcast_rows = 0, float bcast_div = 1.f, int dy_parts = 1, long long dy_part_stride = 0) {
        if constexpr (std::is_same<T, h16_t>::value) {
            static const bool fused = HULC_SWITCH("HULC_LN_FUSED", 1) != 0;
            if (fused || bcast_rows > 0 || dy_parts > 1) {
                const int rpb = rows >= 1024 ? 16 : 4;
                hipLaunchKernelGGL((layernorm_bwd_fused_kernel<T>), dim3(cdiv(rows, rpb)), dim3(rpb == 16 ? 1024 : 256), 0, st, dy, lddy, x, ldx, stats, g, rows, n, dxf, ldd, acc, dxt, ldt,
                                   drop_p, drop_seed, rpb, dg, db, bcast_rows, bcast_div, dy_parts, dy_part_stride);
                return;
            }
        }
        if ((drop_p > 0.f || drop_seed != 0) && dxt && dxf) {       // unfused: dx first, its 16-bit copy through the dropout mask in a second launch
            hipLaunchKernelGGL((layernorm_bwd_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), 0, st, dy, lddy, x, ldx, stats, g, rows, n, dxf, ldd, acc, (T*)nullptr, 0);
            hipLaunchKernelGGL((dropout_apply_kernel<T>), dim3(cdiv((long long)rows * n, 256)), dim3(256), 0, st, dxf, (float*)nullptr, dxt, (long long)rows * n, drop_p, drop_seed);
        } else
        hipLaunchKernelGGL((layernorm_bwd_kernel<T>), dim3(cdiv(rows, 4)), dim3(256), 0, st, dy, lddy, x, ldx, stats, g, rows, n, dxf, ldd, acc, dxt, ldt);
        const int nsplit = std::max(1, std::min(64, cdiv(rows, 64)));
        if constexpr (std::is_same<T, h16_t>::value) {
            hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3(cdiv(n, 64), nsplit), dim3(256), 0, st, dy, lddy, x, ldx, stats, rows, n, cdiv(rows, nsplit), cspart, dg, db);
            return;
        }
        hipLaunchKernelGGL(layernorm_param_grad_kernel, dim3(cdiv(n, 64), nsplit), dim3(256), 0, st, dy, lddy, x, ldx, stats, rows, n, cdiv(rows, nsplit), cspart, (float*)nullptr, (float*)nullptr);
        hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(n, 64)), dim3(256), 0, st, cspart, nsplit, n, dg, (float*)nullptr, 1.f);
        hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(n, 64)), dim3(256), 0, st, cspart + (long long)nsplit * n, nsplit, n, db, (float*)nullptr, 1.f);
    }

    // ---------------------------------------------------------------- weight preparation
    // compute-precision copies: (bf16) flat shadow cast, ONE batched launch for every transposed Linear weight, conv packs, the
    // NHWC-permuted gripper fc and the packed decoder heads
    int prepare_weights(bool shadow_fresh = false) override {
        if (!bound) { hulc_set_error("hulc_prepare_weights before hulc_bind_params"); return 1; }
        c1_bias_fold_valid = false;
        if constexpr (!std::is_same<T, float>::value) {
            if (!shadow_fresh) hipLaunchKernelGGL((cast_kernel<float, T>), dim3(2048), dim3(256), 0, st, P, wshadow, (long long)numel);
        }
        {      // the six conv weight packs, the gripper fc7's NHWC column permutation and the packed decoder heads: one launch (weight_pack_kernel) ...
            ConvPackBatch d;
            int k = 0, blk = 0;
            for (EncW* e : {&encS, &encG})
                for (ConvW* c : {&e->c1, &e->c2, &e->c3}) {
                    d.w[k] = c->W32; d.wf[k] = c->Wf; d.wd[k] = c->nhwc ? c->Wd : nullptr; d.O[k] = c->O; d.I[k] = c->I; d.K[k] = c->KH; d.S[k] = c->S; d.nhwc[k] = c->nhwc;
                    d.blk0[k] = blk; blk += cdiv(c->O * c->I * c->KH * c->KW, 256); ++k;
                }
            d.blk0[6] = blk;
            const int blkA = blk, blkB = blkA + cdiv(128 * 3136, 256);
            // packed heads [192][2048]: prob | mean | log_scale | gripper | zero pad
            const HeadPack hp = head_pack();
            const int rows = head_rows[0] + head_rows[1] + head_rows[2] + head_rows[3];
            // 16-bit engines: the transposed copies of the two packed matrices are written by this launch too (rest_by_pack: then no transpose launch is left behind Adam)
            T* const f7t = rest_by_pack ? encG.fc7.Wt : (T*)nullptr;
            T* const wht = rest_by_pack ? wheadsT : (T*)nullptr;
            hipLaunchKernelGGL((weight_pack_kernel<T>), dim3(blkB + cdiv((long long)rows * HID, 256)), dim3(256), 0, st, d, blkA, encG.fc7.W32, encG.fc7.W, 128, 64, 49, blkB, hp,
                               wheads, bheads, HID, f7t, wht, NHEAD);
        }
        // ... then every transposed copy — the Linear weights, the permuted fc7 and the packed heads — in ONE batched launch
        if (std::is_same<T, float>::value) hipLaunchKernelGGL((batched_transpose_kernel<float, T>), dim3(tr_blocks), dim3(256), 0, st, trdesc_dev, (int)trdesc.size());
        else if (tr_fresh && tr_adam.n) {      // the optimizer wrote the shadow-sourced transposed copies (adam_tiled_kernel): only the packed sources are left
            if (tr_rest.blocks && !rest_by_pack) hipLaunchKernelGGL(batched_transpose64_kernel, dim3(tr_rest.blocks), dim3(256), 0, st, tr_rest.desc, tr_rest.n, (const unsigned short*)tr_rest.b2d);
        }
        else hipLaunchKernelGGL(batched_transpose64_kernel, dim3(tr_blocks), dim3(256), 0, st, trdesc_dev, (int)trdesc.size(), (const unsigned short*)blk2desc_dev);
        const bool was_fresh = tr_fresh && tr_adam.n;
        tr_fresh = false;
        if constexpr (std::is_same<T, h16_t>::value) {
            // (the optimizer's tile pass wrote them when every frag job is linked to a tiled descriptor: frag_by_adam)
            if (fragbatch.n && !(was_fresh && frag_by_adam)) hipLaunchKernelGGL(frag_pack_kernel, dim3(frag_blocks), dim3(256), 0, st, fragbatch);
        }
        STAGE("prepare_weights");
        if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in prepare_weights"); return 1; }
        return 0;
    }
    // grads_fresh: the gradient buffer is all zeros (hulc_zero_grads was the last thing that touched it).  The first backward after it may STORE
    // the weight gradients that have a single whole-tensor contribution instead of reading the zeros back and adding (200 MB of reads per step):
    // 0 + x == x exactly, so the result is bit-identical.  A second backward before the next zero_grads (one pass per modality) accumulates.
    // deferred weight gradients of the plan-recognition transformer (16-bit fused path): jobs collected by tr_wgrad_add, one launch in tr_wgrads_flush
    T *trb_c[2] = {nullptr, nullptr}, *trb_a[2] = {nullptr, nullptr}, *trb_d[2] = {nullptr, nullptr}, *trb_b[2] = {nullptr, nullptr};
    LinBwdBatch tr_wjobs{};
    int tr_wblocks = 0;
    void tr_wgrad_add(const T* dY, const T* X, const LinW& L) {
        if constexpr (std::is_same<T, h16_t>::value) {
            LinBwdJob& J = tr_wjobs.j[tr_wjobs.n++];
            J.dY = dY; J.X = X; J.dW = L.dW; J.db = L.db; J.ldx = L.K; J.lddw = L.K; J.N = L.N; J.K = L.K; J.nx = cdiv(L.N, 64); J.blk0 = tr_wblocks; J.part = nullptr;
            tr_wblocks += J.nx * cdiv(L.K, 128);
        }
    }
    void tr_wgrads_flush(int M) {
        if constexpr (std::is_same<T, h16_t>::value) {
            if (tr_wjobs.n == 0) return;
            const int chunk = 256, nz = cdiv(M, chunk);
            tr_wjobs.M = M; tr_wjobs.store = 0; tr_wjobs.mchunk = chunk;
            // slabs [nz][N][K] per job out of the convolution slab arena (free here: the encoders' backward has not started), summed by an unpack launch
            UnpackBatch ub{};
            int64_t cur = 0; int ublocks = 0;
            for (int i = 0; i < tr_wjobs.n; ++i) {
                LinBwdJob& J = tr_wjobs.j[i];
                const int64_t need = (int64_t)nz * J.N * J.K;
                if (part_cur + cur + need > this->partcap) { J.part = nullptr; continue; }      // no room: this job adds with atomics
                J.part = this->part + part_cur + cur; cur += need;
                UnpackJob& U = ub.j[ub.n++];
                U.part = J.part; U.grad = J.dW; U.slab = (long long)J.N * J.K; U.nsplit = nz; U.O = J.N; U.I = J.K; U.KH = U.KW = 1; U.nhwc = 0; U.blk0 = ublocks; U.ysplit = 1;
                ublocks += cdiv(J.N * J.K, 1024);
            }
            hipLaunchKernelGGL(lin_bwd_smallm_batched_kernel, dim3(tr_wblocks, nz), dim3(256), 0, st, tr_wjobs);
            if (ub.n > 0) hipLaunchKernelGGL(unpack_conv_wgrad_batched_kernel, dim3(ublocks, 1), dim3(256), 0, st, ub);
            tr_wjobs.n = 0; tr_wblocks = 0;
        }
    }
    float* dparts = nullptr;                // fused FFN backward: the four hidden-quarter partials of the gradient entering norm1
    h16_t* dump_page = nullptr;             // conv_reg.h EPI form: scratch that receives the stores of pixels which must not be written
    bool grads_fresh = false;
    int wacc() const { return grads_fresh ? 0 : 1; }
    // ---- lazily zeroed weight gradients (round 5, VERDICT r4 #8 ii; 16-bit engines).  hulc_zero_grads used to memset all 188 MB (24.5 us) although the
    // first backward STORES the large Linear weight gradients (0 + x == x).  Now the tensors whose every writer can store (`lazy`: the M = B MLPs'
    // weights through lin_wgrad / mlp_bwd, the decoder's recurrent / layer-1 input weights through their 2048^3 GEMMs) are only MARKED stale by
    // zero_grads and everything else is zeroed by one multi-range launch.  A stale tensor is made valid by whoever touches it first:
    //   * a store-capable site asks grad_first(dW): stale -> it stores (and the mark is cleared); not stale -> it accumulates as before;
    //   * an accumulate-only path calls grad_ensure_zero(dW) first;
    //   * what is still stale when its all-reduce bucket is issued, at the end of the backward, or when the optimizer / a whole-buffer all-reduce
    //     runs (a tensor the step never touched: language_goal.* in a vision-only step) is zeroed then (lazy_sweep).
    // So the buffer the caller sees after hulc_backward is exactly what the full memset produced; between hulc_zero_grads and the next backward the
    // lazy tensors hold the previous step's values (hulc_set_option "lazy_zero_grads" 0 restores the plain memset).
    struct LazyG { int64_t off, n; bool stale; };
    std::vector<LazyG> lazy;
    void lazy_register(const LinW& L) {
        if (!std::is_same<T, h16_t>::value || !L.dW || (int64_t)L.N * L.K < 65536) return;
        const int64_t off = L.dW - G, exact = (int64_t)L.N * L.K;
        int64_t n = (exact + 63) / 64 * 64;
        // the rounded range may cover alignment padding only: never the first elements of the next tensor of a tightly packed layout
        for (const auto& kv : tab_order) if (kv.second.off > off && kv.second.off < off + n) n = exact;
        if (off + n > numel) n = exact;
        if (n % 4 || off % 4) return;      // multi_zero_kernel clears 16-byte pieces
        for (const LazyG& z : lazy) if (z.off == off) return;
        if (lazy.size() < 30 && off >= 0 && off + n <= numel) lazy.push_back(LazyG{off, n, false});
    }
    int lazy_find(const float* dW) const { const int64_t off = dW - G; for (size_t i = 0; i < lazy.size(); ++i) if (lazy[i].off == off) return (int)i; return -1; }
    // store-capable writer of dW: true = STORE
    bool grad_first(const float* dW) {
        const int i = lazy_find(dW);
        if (i < 0) return grads_fresh;
        const bool s = lazy[i].stale;
        lazy[i].stale = false;
        return s;
    }
    void grad_ensure_zero(const float* dW) {
        const int i = lazy_find(dW);
        if (i >= 0 && lazy[i].stale) { hipMemsetAsync(G + lazy[i].off, 0, sizeof(float) * lazy[i].n, st); lazy[i].stale = false; }
    }
    void lazy_sweep(int64_t lo, int64_t hi, hipStream_t s) {
        MultiZero mz{}; int k = 0; long long mx = 0;
        for (LazyG& z : lazy)
            if (z.stale && z.off >= lo && z.off + z.n <= hi) { mz.p[k] = G + z.off; mz.n[k] = z.n; mx = std::max<long long>(mx, z.n); ++k; z.stale = false; }
        if (k) hipLaunchKernelGGL(multi_zero_kernel, dim3((unsigned)std::min<long long>(256, cdiv(mx, 4 * 256 * 8)), k), dim3(256), 0, s, mz);
    }
    int flush_grads() override {
        if (!bound) { hulc_set_error("hulc_flush_grads before hulc_bind_params"); return 1; }
        lazy_sweep(0, numel, st);
        return 0;
    }
    int zero_grads() override {
        if (!bound) { hulc_set_error("hulc_zero_grads before hulc_bind_params"); return 1; }
        if (lazy.empty() || !lazy_zero_mode) {
            HIP_CHECK(hipMemsetAsync(G, 0, numel * sizeof(float), st));
            for (LazyG& z : lazy) z.stale = false;
        } else {
            // the complement of the lazy tensors, as <= 31 ranges in one launch
            MultiZero mz{}; int k = 0; long long mx = 0; int64_t cur = 0;
            auto add = [&](int64_t lo, int64_t hi) { if (hi > lo) { mz.p[k] = G + lo; mz.n[k] = hi - lo; mx = std::max<long long>(mx, hi - lo); ++k; } };
            for (LazyG& z : lazy) { add(cur, z.off); cur = z.off + z.n; z.stale = true; }
            add(cur, numel / 4 * 4);
            if (numel % 4) HIP_CHECK(hipMemsetAsync(G + numel / 4 * 4, 0, sizeof(float) * (numel % 4), st));
            if (k) hipLaunchKernelGGL(multi_zero_kernel, dim3((unsigned)std::min<long long>(256, cdiv(mx, 4 * 256 * 8)), k), dim3(256), 0, st, mz);
        }
        grads_fresh = true; bwd_since_opt = false;
        return 0;
    }

#include "engine_encoders.inc"      // the perceptual encoders: conv1 sources (fp32 / uint8 / frame store), conv2 / conv3 forward, data and weight gradients, spatial softmax, the fc tails
#include "engine_forward.inc"      // MLP helper, the forward pieces shared by training / validation / rollout, hulc_forward_loss and hulc_forward_loss_pair
#include "engine_inference.inc"      // validation forward (a20), the CLIP ground-truth metric, the stateful rollout
#include "engine_recurrent.inc"      // the 2048-wide recurrences: persistent launches (rnn_persist.h) with their probe / fallback protocol, the mcil BiRNN and BiGRU plan encoders (a12, a19)
#include "engine_backward.inc"      // gradient all-reduce buckets + the job-wide skip vote (comm.h), hulc_backward / hulc_backward_part / hulc_backward_allreduce (a16-a17)
    // ---------------------------------------------------------------- dynamic loss scaling (kernels.h: ScalerState)
    ScalerState* scaler = nullptr;          // device; null = off (fp32 / bf16 default)
    const float* lscale() const { return scaler ? &scaler->scale : nullptr; }
    int scaler_enable(float init_scale, float growth, float backoff, int interval) override {
        if (!(init_scale > 0.f)) { scaler = nullptr; return 0; }          // <= 0: off
        if (!(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f) || interval < 1) { hulc_set_error("hulc_scaler_enable: need growth >= 1, 0 < backoff <= 1, interval >= 1"); return 1; }
        if (!scaler_mem) { scaler_mem = alloc<ScalerState>(1); if (alloc_failed) { hulc_set_error("hulc_scaler_enable: allocation failed"); return 1; } }
        ScalerState h; memset(&h, 0, sizeof(h));
        h.scale = init_scale; h.growth = growth; h.backoff = backoff; h.interval = interval;
        HIP_CHECK(hipMemcpyAsync(scaler_mem, &h, sizeof(h), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        scaler = scaler_mem;
        return 0;
    }
    ScalerState* scaler_mem = nullptr;
    int scaler_get(float* scale, int32_t* tracker, int64_t* skipped, int32_t* last_inf, int64_t* taken) override {
        if (!scaler) { if (scale) *scale = 1.f; if (tracker) *tracker = 0; if (skipped) *skipped = 0; if (last_inf) *last_inf = 0; if (taken) *taken = 0; return 0; }
        ScalerState h;
        HIP_CHECK(hipMemcpyAsync(&h, scaler, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (scale) *scale = h.scale; if (tracker) *tracker = h.growth_tracker; if (skipped) *skipped = h.skipped; if (last_inf) *last_inf = h.last_found_inf; if (taken) *taken = h.steps;
        return 0;
    }
    // taken >= 0 also restores the count of optimizer steps actually taken — Adam's bias corrections run on it in fp16 mode (adam_kernel), so a
    // resumed run must get it back (torch keeps it as the optimizer state's `step`, which GradScaler.step never advances on a skipped step)
    int scaler_set(float scale, int32_t tracker, int64_t taken) override {
        if (!scaler) { hulc_set_error("hulc_scaler_set: the loss scaler is off (hulc_scaler_enable first)"); return 1; }
        if (!(scale > 0.f) || tracker < 0) { hulc_set_error("hulc_scaler_set: need scale > 0, growth_tracker >= 0"); return 1; }
        ScalerState h;
        HIP_CHECK(hipMemcpyAsync(&h, scaler, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        h.scale = scale; h.growth_tracker = tracker;
        if (taken >= 0) h.steps = (int)taken;
        HIP_CHECK(hipMemcpyAsync(scaler, &h, sizeof(h), hipMemcpyHostToDevice, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    }

    int optim(const hulc_optim& o) override {
        if (!bound) { hulc_set_error("hulc_optimizer_step before hulc_bind_params"); return 1; }
        persist_check("hulc_optimizer_step", false);
        lazy_sweep(0, numel, st);            // an optimizer step without a backward behind hulc_zero_grads: the lazily zeroed tensors become zeros now
        bwd_since_opt = false;
        const unsigned tag = ++opt_seq;
        if (o.kind != HULC_OPT_ADAM && o.kind != HULC_OPT_ADAMW && o.kind != HULC_OPT_SGD) { hulc_set_error("hulc_optimizer_step: unknown optimizer kind %d", (int)o.kind); return 1; }
        if (o.step < 1) { hulc_set_error("hulc_optimizer_step: step counts from 1 (got %lld)", (long long)o.step); return 1; }
        const float lr = o.lr, b1 = o.beta1, b2 = o.beta2, eps = o.eps, gscale = o.grad_scale;
        const int64_t step = o.step;
        const double bc1d = 1.0 - pow((double)b1, (double)step), bc2d = 1.0 - pow((double)b2, (double)step);
        h16_t* const shadow = std::is_same<T, float>::value ? (h16_t*)nullptr : (h16_t*)wshadow;
        if (scaler) hipLaunchKernelGGL(nonfinite_check_kernel, dim3(2048), dim3(256), 0, st, G, (long long)numel, scaler);      // after the (host-side) all-reduce: every rank sees the same flag
        if (o.kind == HULC_OPT_SGD)
            hipLaunchKernelGGL(sgd_kernel, dim3(2048), dim3(256), 0, st, P, G, AM, (long long)numel, lr, o.momentum, o.dampening, o.weight_decay, (int)(o.nesterov != 0),
                               (int)(step == 1), gscale, shadow, (const ScalerState*)scaler, (const unsigned*)rp_skip, tag);
        else if (adam_fuse_tr && tr_adam.n && shadow) {
            AdamArgs a{P, G, AM, AV, lr, b1, b2, eps, (float)bc1d, (float)sqrt(bc2d), gscale, o.weight_decay, (int)(o.kind == HULC_OPT_ADAMW), shadow, (const ScalerState*)scaler, (const unsigned*)rp_skip, tag};
            hipLaunchKernelGGL(adam_tiled_kernel, dim3(tr_adam.blocks + adam_chunks), dim3(256), 0, st, a, (const TrDesc*)tr_adam.desc, (const unsigned short*)tr_adam.b2d, tr_adam.blocks,
                               (const long long*)adam_chunk_start, (const int*)adam_chunk_n);
            tr_fresh = true;
        } else
            hipLaunchKernelGGL(adam_kernel, dim3(2048), dim3(256), 0, st, P, G, AM, AV, (long long)numel, lr, b1, b2, eps, (float)bc1d, (float)sqrt(bc2d), gscale,
                               shadow, (const ScalerState*)scaler, o.weight_decay, (int)(o.kind == HULC_OPT_ADAMW), (const unsigned*)rp_skip, tag);
        if (scaler) hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, st, scaler, (const unsigned*)rp_skip, tag);
        if (hipGetLastError() != hipSuccess) { hulc_set_error("adam launch failed"); return 1; }
        return prepare_weights(true);
    }

    int get_tensor(const char* name, float* out, int64_t cap, int64_t* n) override {
        auto it = named.find(name);
        if (it == named.end()) { hulc_set_error("hulc_get_tensor: unknown tensor '%s'", name); return 1; }
        const Named& t = it->second;
        int64_t cnt = std::min<int64_t>(cap, t.n);
        *n = cnt;
        HIP_CHECK(hipStreamSynchronize(st));
        if (t.kind == 0) { HIP_CHECK(hipMemcpy(out, t.p, cnt * sizeof(float), hipMemcpyDeviceToHost)); }
        else if (t.kind == 1) {
            std::vector<T> tmp(cnt);
            HIP_CHECK(hipMemcpy(tmp.data(), t.p, cnt * sizeof(T), hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < cnt; ++i) out[i] = host_to_f(tmp[i]);
        } else {
            std::vector<int> tmp(cnt);
            HIP_CHECK(hipMemcpy(tmp.data(), t.p, cnt * sizeof(int), hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < cnt; ++i) out[i] = (float)tmp[i];
        }
        return 0;
    }
    static float host_to_f(float x) { return x; }
#ifdef HULC_HALF_F16
    static float host_to_f(h16_t x) {      // IEEE binary16 -> float on the host
        const uint32_t sgn = (uint32_t)(x >> 15) << 31, ex = (x >> 10) & 31, man = x & 1023;
        uint32_t u;
        if (ex == 0) { float f = ldexpf((float)man, -24); memcpy(&u, &f, 4); u |= sgn; }
        else if (ex == 31) u = sgn | 0x7f800000u | (man << 13);
        else u = sgn | ((ex + 112) << 23) | (man << 13);
        float f; memcpy(&f, &u, 4); return f;
    }
#else
    static float host_to_f(h16_t x) { uint32_t u = ((uint32_t)x) << 16; float f; memcpy(&f, &u, 4); return f; }
#endif
    int get_plan_idx(int32_t* out, int64_t cap) override {
        HIP_CHECK(hipStreamSynchronize(st));
        int64_t cnt = std::min<int64_t>(cap, (int64_t)cur.B * NCAT);
        HIP_CHECK(hipMemcpy(out, pidx, cnt * sizeof(int), hipMemcpyDeviceToHost));
        return 0;
    }
};

// per-kernel test entry (hulc_k_gemm_nt): C (fp32) = relu?(A B^T + bias) through the production GEMM kernels of this translation unit's
// 16-bit type (or fp32 when is_f32; bf16 unit only).  relu bit 2 (value 4): force the register-staged kernel instead of the LDS-DMA one.
int k_gemm_nt(int is_f32, const void* A, const void* B, float* C, int M, int N, int K, long long lda, long long ldb, long long ldc, const float* bias, int relu,
              void* stream) {
    hipStream_t st = (hipStream_t)stream;
    EpiP ep; ep.out = C; ep.out_f32 = 1; ep.bias = bias; ep.relu = relu & 1;
    if (!is_f32 && !(relu & 4) && M >= 512 && N >= 128) {
        const DenseLoader<h16_t> a = dense<h16_t>((const h16_t*)A, M, lda), b = dense<h16_t>((const h16_t*)B, N, ldb);
        if (gemm_glds_ok(a, b, ep, M, N, K)) {
            launch_gemm_glds(st, a, b, dense_out(ldc), ep, M, N, K);
            if (hipGetLastError() != hipSuccess) { hulc_set_error("hulc_k_gemm_nt: launch failed"); return 1; }
            return 0;
        }
    }
    if (is_f32) {
#ifdef HULC_HALF_F16
        hulc_set_error("hulc_k_gemm_nt: the fp32 kernels live in the bf16 translation unit"); return 1;
#else
        if (M >= 512 && N >= 128) launch_gemm<float, 128, 128>(st, dense<float>((const float*)A, M, lda), dense<float>((const float*)B, N, ldb), dense_out(ldc), ep, M, N, K);
        else launch_gemm<float, 64, 64>(st, dense<float>((const float*)A, M, lda), dense<float>((const float*)B, N, ldb), dense_out(ldc), ep, M, N, K);
#endif
    } else {
        if (M >= 512 && N >= 128) launch_gemm<h16_t, 128, 128>(st, dense<h16_t>((const h16_t*)A, M, lda), dense<h16_t>((const h16_t*)B, N, ldb), dense_out(ldc), ep, M, N, K);
        else launch_gemm<h16_t, 64, 64>(st, dense<h16_t>((const h16_t*)A, M, lda), dense<h16_t>((const h16_t*)B, N, ldb), dense_out(ldc), ep, M, N, K);
    }
    if (hipGetLastError() != hipSuccess) { hulc_set_error("hulc_k_gemm_nt: launch failed"); return 1; }
    return 0;
}

// factory of this translation unit (iengine.h): fp32 + bf16 engines in capi.hip, the fp16 engine in engine_f16.hip
IEngine* make_engine(const hulc_config& cfg, int* rc) {
#ifdef HULC_HALF_F16
    auto* e = new Engine<h16_t>(cfg); *rc = e->alloc_all(); return e;
#else
    if (cfg.dtype == HULC_DTYPE_F32) { auto* e = new Engine<float>(cfg); *rc = e->alloc_all(); return e; }
    auto* e = new Engine<h16_t>(cfg); *rc = e->alloc_all(); return e;
#endif
}

}  // namespace HULC_NS
