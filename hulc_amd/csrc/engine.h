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

    // ---------------------------------------------------------------- encoders
    ConvGeom geom(int Nf, int IH, int C, int K, int S) const {
        ConvGeom g; g.Nf = Nf; g.IH = g.IW = IH; g.C = C; g.KH = g.KW = K; g.S = S; g.OH = g.OW = (IH - K) / S + 1; return g;
    }
    // conv1 input of the current batch: the reference's fp32 NCHW frames, or uint8 NHWC frames + the fused dataloader transforms
    Conv1Src conv1_src(const hulc_batch& b, bool gripper) const {
        Conv1Src s;
        s.X = gripper ? b.rgb_gripper : b.rgb_static;
        s.u8 = b.frames_u8 != 0;
        s.shift = s.u8 ? (gripper ? b.shift_gripper : b.shift_static) : nullptr;
        s.pad = gripper ? b.pad_gripper : b.pad_static;
        // 16-bit engines: the dataloader's affine is folded out of the uint8 data path (conv_wgrad.h Conv1Src::fold); the fp32 (parity) engine converts
        // exactly like the reference (ingest_u8_kernel)
        s.fold = (s.u8 && std::is_same<T, h16_t>::value && u8_fold_mode) ? 1 : 0;
        if (s.u8 && b.window_start) { s.wstart = reinterpret_cast<const long long*>(b.window_start); s.S = b.S; s.nstore = b.store_frames; }      // windows gathered from the frame store
        return s;
    }
    // b - sum_k W16 of the two conv1 layers (Conv1Src::fold), recomputed after every weight refresh, only when a uint8 batch asks for it
    float* c1_bias_fold[2] = {nullptr, nullptr};
    bool c1_bias_fold_valid = false;
    const float* conv1_bias(const EncW& e, const Conv1Src& src) {
        if (!src.fold) return e.c1.b32;
        if constexpr (std::is_same<T, h16_t>::value) {
            if (!c1_bias_fold[0]) { c1_bias_fold[0] = alloc<float>(64); c1_bias_fold[1] = alloc<float>(64); }
            if (!c1_bias_fold_valid) {
                hipLaunchKernelGGL(conv1_bias_fold_kernel, dim3(32), dim3(64), 0, st, encS.c1.Wf, encS.c1.b32, c1_bias_fold[0]);
                hipLaunchKernelGGL(conv1_bias_fold_kernel, dim3(32), dim3(64), 0, st, encG.c1.Wf, encG.c1.b32, c1_bias_fold[1]);
                c1_bias_fold_valid = true;
            }
            return c1_bias_fold[e.gripper ? 1 : 0];
        }
        return e.c1.b32;
    }
    // actions of the current batch: the reference's relative actions, or absolute targets + RelativeActions applied here
    float* act_rel = nullptr;
    const float* actions_of(const hulc_batch& b) {
        if (!b.actions_absolute) return b.actions;
        if (!act_rel) act_rel = alloc<float>((int64_t)maxN * 7);
        hipLaunchKernelGGL(relative_actions_kernel, dim3(cdiv(b.B * b.S, 256)), dim3(256), 0, st, b.actions, b.robot_obs, b.B * b.S, b.max_rel_pos, b.max_rel_orn, act_rel);
        return act_rel;
    }
    float* x32[2] = {nullptr, nullptr};       // fp32 (parity) mode + uint8 ingest: the transformed frames are materialised once per step
    const float* conv1_f32(const Conv1Src& src, bool gripper, int Nf, int IH, long long frame_off = 0) {
        if (!src.u8) return reinterpret_cast<const float*>(src.X);
        float*& buf = x32[gripper ? 1 : 0];
        if (!buf) buf = alloc<float>((int64_t)maxN * 3 * IH * IH);
        const long long n = (long long)Nf * 3 * IH * IH;
        float* dst = buf + frame_off * 3 * IH * IH;
        hipLaunchKernelGGL(ingest_u8_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(src.X), src.shift, src.pad, Nf, IH, IH, dst, src);
        return dst;
    }
    // ---- paired pass (vis + lang windows of one step as ONE 2B-window pass, hulc_forward_loss_pair): rows [0, pairBv) are the vis
    // windows, the rest the lang windows; frames (and their shifts) stay in the two batches' own buffers, everything else is joint
    bool pair = false; int pairBv = 0; hulc_batch cur2;
    float *act_j = nullptr, *ro_j = nullptr, *eps_j = nullptr, *losses2 = nullptr; int* aux_j = nullptr;
    // src2 (paired pass): frames [0, Nf/2) come from src, [Nf/2, Nf) from src2 — conv1 runs once per source, the rest on all Nf frames
    void enc_fwd(const EncW& e, EncA& a, const Conv1Src& src, int Nf, int col0, const Conv1Src* src2 = nullptr, bool defer_tail = false) {
        ConvGeom g1 = geom(Nf, e.IH, 3, 8, 4), g2 = geom(Nf, e.H1, 32, 4, 2), g3 = geom(Nf, e.H2, 64, 3, 1);
        for (int h = 0; h < (src2 ? 2 : 1); ++h) {
            const Conv1Src& sh = h ? *src2 : src;
            const int nf = src2 ? Nf / 2 : Nf;
            const long long foff = h ? Nf / 2 : 0, poff = foff * g1.OH * g1.OW;
            ConvGeom gh = geom(nf, e.IH, 3, 8, 4);
            if constexpr (std::is_same<T, h16_t>::value) {
                const double px = (double)nf * g1.OH * g1.OW;
                TimerScope ts(this, "conv1_fwd", "hbm", 2.0 * px * 32 * 192, (double)nf * 3 * e.IH * e.IH * (sh.u8 ? 1 : 4) + px * 32 * 2);
                launch_conv1_fwd(st, sh, e.c1.Wf, conv1_bias(e, sh), a.a1 + poff * 32, nf, e.IH, e.IH, g1.OH, g1.OW, 0, a.m1bits ? a.m1bits + poff : nullptr, pend_zero[0], pend_zero[1]);
                pend_zero[0] = pend_zero[1] = nullptr;
            } else {
                const float* x = conv1_f32(sh, e.gripper, nf, e.IH, foff);
                Conv1Loader<T> l{x, gh};
                EpiP ep = epi(a.a1 + poff * 32, false); ep.bias = e.c1.b32; ep.relu = 1;
                launch_gemm<T, 128, 32>(st, l, dense<T>(e.c1.Wf, 32, 192), dense_out(32), ep, nf * g1.OH * g1.OW, 32, 192);
            }
        }
        bool tiled = false;
        if constexpr (std::is_same<T, h16_t>::value) {   // raw-tile kernels (conv_tile.h): weights resident in LDS, bands streamed once
            ConvTileP p2{}; p2.img = a.a1; p2.IMH = p2.IMW = e.H1; p2.w = e.c2.Wf; p2.out = a.a2; p2.OUTH = p2.OUTW = e.H2; p2.bias = e.c2.b32; p2.relu = 1; p2.Nf = Nf; p2.bits_out = a.m2bits;
            ConvTileP p3{}; p3.img = a.a2; p3.IMH = p3.IMW = e.H2; p3.w = e.c3.Wf; p3.out = a.a3; p3.OUTH = p3.OUTW = e.H3; p3.bias = e.c3.b32; p3.relu = 1; p3.Nf = Nf;
            const double px2 = (double)Nf * e.H2 * e.H2, px3 = (double)Nf * e.H3 * e.H3, px1 = (double)Nf * e.H1 * e.H1;
            TimerScope ts(this, "conv_tile_fwd", "mfma", 2.0 * px2 * 64 * 512 + 2.0 * px3 * 64 * 576, (px1 * 32 + 2 * px2 * 64 + px3 * 64) * 2);
            static const int conv_reg = HULC_SWITCH("HULC_CONV_REG", 7);      // A/B: bit 0 = conv2 forward, bit 1 = conv3 forward, bit 2 = conv3 data gradient on the weights-in-registers kernel (conv_reg.h)
            // HULC_CONV_REG_W4 (same bits): the form with two co-resident 256-thread workgroups per CU (conv_reg.h, NWV = 4)
            static const int w4 = HULC_SWITCH("HULC_CONV_REG_W4", 11);          // in the step: conv2 fwd 84.0 vs 85.4 us, conv3 fwd 54.7 vs 56.5, conv2 dgrad 79.8 vs 87.9 (launch pairs' average, two-workgroup form first)
            // HULC_CONV_REG_PK (round 5, same bits): large maps (the static camera) take the one-workgroup form with the band-invariant DMA slot decode
            // held in registers (conv_reg.h: PKR) — standalone on 2048 static frames conv2 forward 135 us against 142 (two workgroups) / 143, conv3
            // forward 77 against 84 / 81; the gripper camera's stacked small maps stay on the two-workgroup form (30.5 against 34.8 us)
            static const int pkr = HULC_SWITCH("HULC_CONV_REG_PK", 11);
            const bool big = e.H2 >= 16;
            const bool t2 = ((conv_reg & 1) && (((w4 & 1) && !(big && (pkr & 1))) ? launch_conv_reg<32, 4, 4, 2, false, 1, 4, 0, true>(st, p2) : launch_conv_reg_fwd<32, 4, 4, 2>(st, p2))) || launch_conv_tile<32, 64, 4, 4, 2, 1, false>(st, p2);
            const bool t3 = ((conv_reg & 2) && (((w4 & 2) && !(big && (pkr & 2))) ? launch_conv_reg<64, 3, 3, 1, false, 1, 4>(st, p3) : launch_conv_reg_fwd<64, 3, 3, 1>(st, p3))) || launch_conv_tile<64, 64, 3, 3, 1, 1, false>(st, p3);
            tiled = t2 && t3;
        }
        if (!tiled) {
            {
                ConvNHWCLoader<T> l{a.a1, g2};
                EpiP ep = epi(a.a2, false); ep.bias = e.c2.b32; ep.relu = 1;
                launch_gemm<T, 128, 64>(st, l, dense<T>(e.c2.Wf, 64, 512), dense_out(64), ep, Nf * g2.OH * g2.OW, 64, 512);
            }
            {
                ConvNHWCLoader<T> l{a.a2, g3};
                EpiP ep = epi(a.a3, false); ep.bias = e.c3.b32; ep.relu = 1;
                launch_gemm<T, 128, 64>(st, l, dense<T>(e.c3.Wf, 64, 576), dense_out(64), ep, Nf * g3.OH * g3.OW, 64, 576);
            }
        }
        const T* fin; int fk;
        if (!e.gripper) {
            if constexpr (std::is_same<T, h16_t>::value) hipLaunchKernelGGL(spatial_softmax_fwd64_kernel, dim3(Nf), dim3(256), 0, st, a.a3, e.H3, e.H3, a.ss, a.ssstats);
            else hipLaunchKernelGGL((spatial_softmax_fwd_kernel<T>), dim3(Nf), dim3(256), 0, st, a.a3, e.H3, e.H3, 64, a.ss, (float*)nullptr, a.ssstats);
            fin = a.ss; fk = 128;
        } else {
            EpiP ep = epi(a.g0, false); ep.relu = 1;
            lin_fwd(a.a3, 3136, Nf, e.fc7, ep, 128);
            fin = a.g0; fk = 128;
        }
        if (defer_tail) return;             // 16-bit engines: the dense tails of both cameras run as one launch (enc_tail_fwd_both)
        { EpiP ep = epi(a.f1, false); ep.relu = 1; lin_fwd(fin, fk, Nf, e.fc1, ep, 512); }
        { EpiP ep = epi(a.f2, true); lin_fwd(a.f1, 512, Nf, e.fc2, ep, 64); }
        ln_fwd(a.f2, 64, Nf, 64, e.lng, e.lnb, emb + col0, EMB, nullptr, 0, a.lnst);
    }
    // enc_tail.h is written for the reference's tail widths (128 -> 512 -> 64, vision_network.py:46-52 / vision_network_gripper.py:18-27) and EMB = 2 x 64
    bool enc_tail_fusable() const {
        if constexpr (!std::is_same<T, h16_t>::value) return false;
        auto ok = [](const EncW& e) { return e.fc1.K == 128 && e.fc1.N == 512 && e.fc2.K == 512 && e.fc2.N == 64; };
        return ok(encS) && ok(encG) && EMB == 128;
    }
    // fc1 + ReLU, fc2 and the LayerNorm of both encoders in one launch (enc_tail.h)
    // with_x0: the same launch also writes the plan-recognition transformer's input (enc_tail.h) — pr_fwd then skips its posadd launch
    bool x0_done = false;
    void enc_tail_fwd_both(int Nf, bool with_x0 = false, int S = 1, float dp = 0.f) {
        if constexpr (std::is_same<T, h16_t>::value) {
            EncTailP q{};
            if (with_x0) { q.pos = pos32; q.xf = xf[0]; q.xt = xt[0]; q.z0 = y2[0]; q.z1 = y2[1]; q.S = S; q.drop_p = dp; q.seed = site_seed(0); x0_done = true; }
            const EncW* ew[2] = {&encS, &encG};
            EncA* ea[2] = {&aS, &aG};
            for (int k = 0; k < 2; ++k) {
                EncTailCam& c = q.cam[k];
                c.x = ew[k]->gripper ? ea[k]->g0 : ea[k]->ss; c.W1 = ew[k]->fc1.W; c.W2 = ew[k]->fc2.W; c.b1 = ew[k]->fc1.b32; c.b2 = ew[k]->fc2.b32;
                c.lng = ew[k]->lng; c.lnb = ew[k]->lnb; c.f1 = ea[k]->f1; c.f2 = ea[k]->f2; c.lnst = ea[k]->lnst; c.col0 = k * 64;
            }
            q.emb = emb; q.Nf = Nf; q.ldemb = EMB;
            launch_enc_tail_fwd(st, q);
        }
    }
    // the data-gradient chain of both tails (LayerNorm, fc2, fc1) in one launch; the weight gradients follow in enc_bwd
    void enc_tail_bwd_both(int Nf) {
        if constexpr (std::is_same<T, h16_t>::value) {
            EncTailBwdP q{};
            const EncW* ew[2] = {&encS, &encG};
            EncA* ea[2] = {&aS, &aG};
            for (int k = 0; k < 2; ++k) {
                EncTailBwdCam& c = q.cam[k];
                c.f2 = ea[k]->f2; c.lnst = ea[k]->lnst; c.lng = ew[k]->lng; c.f1 = ea[k]->f1; c.W2t = ew[k]->fc2.Wt; c.W1t = ew[k]->fc1.Wt;
                c.xmask = ew[k]->gripper ? ea[k]->g0 : nullptr; c.dlng = ew[k]->dlng; c.dlnb = ew[k]->dlnb;
                c.d_f2 = ew[k]->gripper ? d_f2tg : d_f2t; c.d_f1 = ew[k]->gripper ? d_f1g : d_f1;
                c.dx_f32 = ew[k]->gripper ? nullptr : d_ss; c.dx_t = ew[k]->gripper ? d_g0 : nullptr; c.col0 = k * 64;
            }
            q.demb = demb; q.Nf = Nf; q.ldemb = EMB;
            launch_enc_tail_bwd(st, q);
            // weight / bias gradients of the four Linear layers: one launch, frames split over blockIdx.y (fp32 atomics), operands read as they lie
            static const int chunk = HULC_SWITCH("HULC_ENC_WGRAD_CHUNK", 256);
            tail_wgrad_done = chunk > 0;
            if (tail_wgrad_done) {
                LinBwdBatch bt{}; bt.M = Nf; bt.store = 0; bt.mchunk = chunk;
                int blk = 0;
                const int nz = cdiv(Nf, chunk);
                static const bool slab_ok = HULC_SWITCH("HULC_ENC_WGRAD_SLABS", 1) != 0;
                for (int k = 0; k < 2; ++k)
                    for (int l = 0; l < 2; ++l) {          // l = 0: fc2 (dY = d_f2, X = f1);  1: fc1 (dY = d_f1, X = the tail's input)
                        const LinW& L = l ? ew[k]->fc1 : ew[k]->fc2;
                        LinBwdJob& J = bt.j[bt.n++];
                        J.dY = l ? q.cam[k].d_f1 : q.cam[k].d_f2; J.X = l ? (ew[k]->gripper ? ea[k]->g0 : ea[k]->ss) : ea[k]->f1;
                        J.dW = L.dW; J.db = L.db; J.ldx = L.K; J.lddw = L.K; J.N = L.N; J.K = L.K; J.nx = cdiv(L.N, 64); J.blk0 = blk;
                        blk += J.nx * cdiv(L.K, 128);
                        // every row chunk writes its own slab; the slabs are summed into the gradient by the encoders' one unpack launch (no
                        // per-element atomics here: 1.5 M of them made this launch 39 us)
                        const int64_t need = (int64_t)nz * L.N * L.K;
                        if (slab_ok && unpack_jobs.n < 12 && part_cur + need <= this->partcap) {
                            J.part = this->part + part_cur;
                            UnpackJob& U = unpack_jobs.j[unpack_jobs.n++];
                            U.part = J.part; U.grad = L.dW; U.slab = (long long)L.N * L.K; U.nsplit = nz; U.O = L.N; U.I = L.K; U.KH = U.KW = 1; U.nhwc = 0; U.blk0 = unpack_blocks; U.ysplit = 1;
                            unpack_blocks += cdiv(L.N * L.K, 1024); part_cur += need;
                        }
                    }
                // + the gripper camera's first Linear (3136 -> 128, dY = d_g0 which the launch above just wrote): its slabs are in the packed (NHWC)
                // column order of a3 and the unpack launch lands them in the torch layout (the conv weights' permutation with a 7 x 7 "kernel")
                tail_fc7_done = false;
                {
                    const LinW& L = encG.fc7;
                    const int64_t need = (int64_t)nz * L.N * L.K;
                    if (slab_ok && L.N == 128 && L.K == 3136 && unpack_jobs.n < 12 && part_cur + need <= this->partcap) {
                        LinBwdJob& J = bt.j[bt.n++];
                        J.dY = d_g0; J.X = aG.a3; J.dW = nullptr; J.db = L.db; J.ldx = L.K; J.lddw = L.K; J.N = L.N; J.K = L.K; J.nx = cdiv(L.N, 64); J.blk0 = blk;
                        blk += J.nx * cdiv(L.K, 128);
                        J.part = this->part + part_cur;
                        UnpackJob& U = unpack_jobs.j[unpack_jobs.n++];
                        U.part = J.part; U.grad = L.dW; U.slab = (long long)L.N * L.K; U.nsplit = nz; U.O = L.N; U.I = 64; U.KH = U.KW = 7; U.nhwc = 1; U.blk0 = unpack_blocks; U.ysplit = 1;
                        unpack_blocks += cdiv(L.N * L.K, 1024); part_cur += need;
                        tail_fc7_done = true;
                    }
                }
                hipLaunchKernelGGL(lin_bwd_smallm_batched_kernel, dim3(blk, nz), dim3(256), 0, st, bt);
            }
        }
    }
    bool tail_wgrad_done = false, tail_fc7_done = false;
    float* pend_zero[2] = {nullptr, nullptr};   // loss accumulators the next conv1 forward launch clears (16-bit engines)
    Conv1Src wgrad_src;                       // conv1 only: set by enc_bwd before conv_wgrad(e.c1, ...)
    // 16-bit engines: the slab -> gradient reductions of the encoders' convolutions are collected and run as ONE launch (flush_unpacks) after
    // both encoders' backward instead of one ~6-20 us launch behind each of the six weight-gradient kernels
    UnpackBatch unpack_jobs{};
    int64_t part_cur = 0;
    int unpack_blocks = 0;
    void flush_unpacks() {
        static const int yparts = HULC_SWITCH("HULC_UNPACK_YB", 8);      // same-box: 4 / 8 / 16 / 32 parts = 60 / 55 / 59 / 86 us (with 8 slab quads in flight per thread)
        if (unpack_jobs.n > 0) hipLaunchKernelGGL(unpack_conv_wgrad_batched_kernel, dim3(unpack_blocks, yparts), dim3(256), 0, st, unpack_jobs);
        unpack_jobs.n = 0; part_cur = 0; unpack_blocks = 0;
    }
    void conv_wgrad(const ConvW& c, const T* dy, const void* xin, const ConvGeom& g, bool conv1) {
        const int Kc = c.I * c.KH * c.KW;
        const long long npix = (long long)g.Nf * g.OH * g.OW;
        int nsplit = 0;
        float* const part = this->part + part_cur;
        const int64_t partcap = this->partcap - part_cur;
        if constexpr (std::is_same<T, h16_t>::value) {
            // raw-tile + transposing-LDS-read kernel (conv_wgrad.h); slabs = persistent workgroups
            TimerScope ts(this, conv1 ? "conv1_wgrad" : "conv_wgrad_tr", conv1 ? "hbm" : "mfma", 2.0 * npix * c.O * Kc,
                          conv1 ? ((double)g.Nf * 3 * g.IH * g.IW * (wgrad_src.u8 ? 1 : 4) + npix * c.O * 2) : ((double)g.Nf * g.IH * g.IW * c.I * 2 + npix * c.O * 2));
            if (conv1)
                nsplit = launch_conv1_wgrad_tr(st, wgrad_src, dy, part, c.db, g.Nf, g.IH, g.IW, g.OH, g.OW, 1024, next_ctr());
            else if (!conv1 && c.I == 64 && c.KH == 3)
                nsplit = launch_conv_wgrad_tr<64, 64, 3, 3, 1>(st, (const h16_t*)xin, dy, part, c.db, g.Nf, g.IH, g.IW, g.OH, g.OW, 512, next_ctr(), wgrad_zero_page());
            else if (!conv1 && c.I == 32 && c.KH == 4)
                nsplit = launch_conv_wgrad_tr<32, 64, 4, 4, 2>(st, (const h16_t*)xin, dy, part, c.db, g.Nf, g.IH, g.IW, g.OH, g.OW, 512, next_ctr(), wgrad_zero_page());
        }
        bool bias_done = false;
        if (nsplit > 0) bias_done = true;     // the tr kernels added the bias gradient themselves (atomics)
        if (nsplit == 0) {
            nsplit = (int)std::min<long long>(std::max<long long>(1, npix / 2048), partcap / ((long long)c.O * Kc));
            nsplit = std::min(nsplit, 256);
            EpiP ep = epi(part, true); ep.z_stride = (long long)c.O * Kc;
            PixMajorLoaderT<T> la{}; la.p = dy; la.rows = c.O; la.ld = c.O;
            if (conv1) {
                Conv1LoaderT<T> lb{(const float*)xin, g};
                launch_gemm<T, 32, 64>(st, la, lb, dense_out(Kc), ep, c.O, Kc, (int)npix, 1, nsplit);
            } else {
                ConvNHWCLoaderT<T> lb{(const T*)xin, g};
                launch_gemm<T, 64, 64>(st, la, lb, dense_out(Kc), ep, c.O, Kc, (int)npix, 1, nsplit);
            }
        }
        // slab parts over grid.y, each landing with one fp32 atomic per element.  16 parts: more (21 / 24 / 64 for conv3 / conv2 / conv1) made
        // every launch slower (23 / 11.3 / 10.9 us against 18.6 / 9.5 / 9.6: the scattered atomics, not the slab stream, are the cost)
        const int ybl = cdiv(c.O * Kc, 1024);
        if constexpr (std::is_same<T, h16_t>::value) {
            if (nsplit >= 64 && unpack_jobs.n < 12 && part_cur + (int64_t)nsplit * c.O * Kc <= this->partcap) {
                UnpackJob& J = unpack_jobs.j[unpack_jobs.n++];
                J.part = part; J.grad = c.dW; J.slab = (long long)c.O * Kc; J.nsplit = nsplit; J.O = c.O; J.I = c.I; J.KH = c.KH; J.KW = c.KW; J.nhwc = c.nhwc; J.blk0 = unpack_blocks; J.ysplit = 0;
                unpack_blocks += ybl; part_cur += (int64_t)nsplit * c.O * Kc;
                if (!bias_done) colsum(dy, c.O, (int)npix, c.O, c.db);
                return;
            }
        }
        static const int ypart_env = HULC_SWITCH("HULC_UNPACK_Y", 16);
        const int yparts = (!std::is_same<T, float>::value && nsplit >= 64) ? ypart_env : 1;
        hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3(ybl, yparts), dim3(256), 0, st, part, nsplit, (long long)c.O * Kc,   // fp32 (parity) mode: one deterministic pass, no atomics
                           c.dW, c.O, c.I, c.KH, c.KW, c.nhwc);
        if (!bias_done) colsum(dy, c.O, (int)npix, c.O, c.db);
    }
    h16_t* zero_page = nullptr;
    const h16_t* wgrad_zero_page() {
        if constexpr (std::is_same<T, h16_t>::value) { if (!zero_page) zero_page = alloc<h16_t>(128); }     // zero-initialised by alloc()
        return zero_page;
    }
    void conv_dgrad(const ConvW& c, const T* dy, const ConvGeom& g, T* dx, const T* mask, const unsigned* maskbits = nullptr) {
        if constexpr (std::is_same<T, h16_t>::value) {
            ConvTileP p{}; p.img = dy; p.IMH = g.OH; p.IMW = g.OW; p.w = c.Wd; p.out = dx; p.OUTH = g.IH; p.OUTW = g.IW; p.mask = maskbits ? nullptr : mask; p.maskbits = maskbits; p.Nf = g.Nf; p.work_ctr = next_ctr();
            bool ok = false;
            const double pin = (double)g.Nf * g.IH * g.IW, pout = (double)g.Nf * g.OH * g.OW;
            // algorithmic bytes as SURVEY 8(d) counts them: dY read once, dX written once, 2 B each; the ReLU mask of the layer below is read as BIT words
            // (pin * I / 8 bytes) when the forward left them, as a second 16-bit activation read otherwise (the bit form was priced as a full read)
            TimerScope ts(this, "conv_tile_dgrad", "mfma", 2.0 * pout * c.O * c.I * c.KH * c.KW, (pout * c.O + pin * c.I) * 2 + (maskbits ? pin * c.I / 8 : (mask ? pin * c.I * 2 : 0)));
            static const int conv_reg = HULC_SWITCH("HULC_CONV_REG", 15);      // bit 2: conv3, bit 3: conv2 data gradient on conv_reg.h
            static const int w4 = HULC_SWITCH("HULC_CONV_REG_W4", 11);         // same bits: two 256-thread workgroups per CU (NWV = 4).  conv3's data gradient (bit 2) stays on the one-workgroup form:
                                                                               // 144 weight registers + the zero-border decode spill 16 registers at 256 VGPRs (90.3 vs 83.5 us in the step, 3.470 vs 3.448 ms/step)
            if (c.KH == 3 && c.S == 1 && c.I == 64 && c.O == 64) {
                if (!zero_page) zero_page = alloc<h16_t>(128);      // zero-initialised by alloc(): the staged zero border of the data-gradient form
                p.zeros = zero_page;
                // round 5: the pipelined-epilogue form (conv_reg.h EPI: a tile's epilogue rides in the next tile's multiply loop; one tile at a time also ends
                // the spills of the pair form at 144 weight registers) on two workgroups per CU: 129 -> 97 us standalone on 2048 static frames, 30.6 -> 28.2 gripper
                static const int epi = HULC_SWITCH("HULC_CONV_REG_EPI", 12);      // bit 2: conv3, bit 3: conv2 data gradient
                if (!dump_page) dump_page = alloc<h16_t>(4096);
                p.dump = dump_page;
                ok = ((conv_reg & 4) && maskbits && ((epi & 4) ? launch_conv_reg<64, 3, 3, 1, true, 1, 4, 0, true, 1>(st, p)
                                                       : ((w4 & 4) ? launch_conv_reg<64, 3, 3, 1, true, 1, 4>(st, p) : launch_conv_reg<64, 3, 3, 1, true>(st, p)))) || launch_conv_tile<64, 64, 3, 3, 1, 1, true>(st, p);
            }
            else if (c.KH == 4 && c.S == 2 && c.I == 32 && c.O == 64) {
                if (!zero_page) zero_page = alloc<h16_t>(128);
                p.zeros = zero_page;
                static const int pkr = HULC_SWITCH("HULC_CONV_REG_PK", 11);       // bit 3: slot decode in registers (no spill at 64 weight registers): 147 -> 140 us standalone
                static const int epi = HULC_SWITCH("HULC_CONV_REG_EPI", 12);      // bit 3: pipelined epilogue (142 -> 137 us standalone static, 29.7 -> 28.7 gripper)
                if (!dump_page) dump_page = alloc<h16_t>(4096);
                p.dump = dump_page;
                ok = ((conv_reg & 8) && maskbits && ((w4 & 8) ? ((epi & 8) ? launch_conv_reg<64, 2, 2, 1, true, 2, 4, 0, true, 1>(st, p) : (pkr & 8) ? launch_conv_reg<64, 2, 2, 1, true, 2, 4, 0, true>(st, p) : launch_conv_reg<64, 2, 2, 1, true, 2, 4>(st, p))
                                                              : launch_conv_reg<64, 2, 2, 1, true, 2>(st, p))) || launch_conv_tile<64, 32, 2, 2, 1, 2, true>(st, p);
            }
            if (ok) return;
        }
        ConvDgradLoader<T> l{dy, g, c.O};
        DgradOut om{g};
        EpiP ep = epi(dx, false); ep.mask = mask;
        const int ncls = g.S * g.S;
        const int Kd = (g.KH / g.S) * (g.KW / g.S) * c.O;
        const int Ic = (g.IH + g.S - 1) / g.S;
        const int Mmax = g.Nf * Ic * Ic;
        for (int zc = 0; zc < ncls; ++zc)   // one launch per parity class: the packed weight slab differs per class
            launch_dgrad(l, dense<T>(c.Wd + (long long)zc * c.I * Kd, c.I, Kd), om, ep, Mmax, c.I, Kd, zc);
    }
    void launch_dgrad(const ConvDgradLoader<T>& l, const DenseLoader<T>& wb, const DgradOut& om, const EpiP& ep, int Mmax, int N, int K, int zc) {
        // one parity class per launch: wrap the loader/out-map so that blockIdx.z == 0 maps to class zc
        ClassShift<ConvDgradLoader<T>> ls{l, zc};
        ClassShiftOut<DgradOut> os{om, zc};
        if (N <= 32) launch_gemm<T, 128, 32>(st, ls, wb, os, ep, Mmax, N, K);
        else launch_gemm<T, 128, 64>(st, ls, wb, os, ep, Mmax, N, K);
    }
    template <typename L> struct ClassShift {
        static constexpr bool TRANSPOSED = false;
        L l; int zc;
        using Row = typename L::Row;
        DEVI int num_rows(int) const { return l.num_rows(zc); }
        DEVI Row row(int r, int) const { return l.row(r, zc); }
        DEVI void fetch(const Row& c, int k0, int kend, T (&v)[8]) const { l.fetch(c, k0, kend, v); }
    };
    template <typename O> struct ClassShiftOut {
        O o; int zc;
        DEVI long long offset(int r, int) const { return o.offset(r, zc); }
    };
    void enc_bwd(const EncW& e, EncA& a, const Conv1Src& src, int Nf, int col0, const Conv1Src* src2 = nullptr, bool tail_done = false) {
        wgrad_src = src;
        const float* x = nullptr;
        if constexpr (std::is_same<T, float>::value) x = src.u8 ? x32[e.gripper ? 1 : 0] : reinterpret_cast<const float*>(src.X);   // materialised by the forward
        ConvGeom g1 = geom(Nf, e.IH, 3, 8, 4), g2 = geom(Nf, e.H1, 32, 4, 2), g3 = geom(Nf, e.H2, 64, 3, 1);
        T* const d_f1 = tail_done && e.gripper ? d_f1g : this->d_f1;
        T* const d_f2t = tail_done && e.gripper ? d_f2tg : this->d_f2t;
        // LN bwd on demb[:, col0:col0+64]
        if (!tail_done) ln_bwd(demb + col0, EMB, a.f2, 64, a.lnst, e.lng, Nf, 64, nullptr, 0, 0, d_f2t, 64, e.dlng, e.dlnb);
        // fc2
        if (!tail_done) { EpiP ep = epi(d_f1, false); ep.mask = a.f1; lin_dgrad(d_f2t, Nf, e.fc2, ep, dense_out(512)); }
        const bool wdone = tail_done && tail_wgrad_done;
        if (!wdone) lin_wgrad(d_f2t, a.f1, 512, Nf, 64, 512, e.fc2.dW, 512, e.fc2.db);
        const int H3 = e.H3;
        if (!e.gripper) {
            if (!tail_done) { EpiP ep = epi(d_ss, true); lin_dgrad(d_f1, Nf, e.fc1, ep, dense_out(128)); }
            if (!wdone) lin_wgrad(d_f1, a.ss, 128, Nf, 512, 128, e.fc1.dW, 128, e.fc1.db);
            if constexpr (std::is_same<T, h16_t>::value) hipLaunchKernelGGL(spatial_softmax_bwd64_kernel, dim3(Nf), dim3(256), 0, st, a.a3, a.ssstats, d_ss, H3, H3, dact3);
            else hipLaunchKernelGGL((spatial_softmax_bwd_kernel<T>), dim3(Nf), dim3(256), 0, st, a.a3, a.ssstats, d_ss, H3, H3, 64, dact3);
        } else {
            if (!tail_done) { EpiP ep = epi(d_g0, false); ep.mask = a.g0; lin_dgrad(d_f1, Nf, e.fc1, ep, dense_out(128)); }
            if (!wdone) lin_wgrad(d_f1, a.g0, 128, Nf, 512, 128, e.fc1.dW, 128, e.fc1.db);
            { EpiP ep = epi(dact3, false); ep.mask = a.a3; lin_dgrad(d_g0, Nf, e.fc7, ep, dense_out(3136)); }
            // dW7 in packed (NHWC) column order -> temp, then permute-accumulate into the torch-layout grad
            if (!(wdone && tail_fc7_done)) {
                lin_wgrad(d_g0, a.a3, 3136, Nf, 128, 3136, dw7_tmp, 3136, e.fc7.db);
                hipLaunchKernelGGL((permute_cols_kernel<float, float>), dim3(cdiv(128 * 3136, 256)), dim3(256), 0, st, dw7_tmp, e.fc7.dW, 128, 64, 49, 1, 1);
            }
        }
        conv_wgrad(e.c3, dact3, a.a2, g3, false);
        conv_dgrad(e.c3, dact3, g3, dact2, a.a2, a.m2bits);
        conv_wgrad(e.c2, dact2, a.a1, g2, false);
        conv_dgrad(e.c2, dact2, g2, dact1, a.a1, a.m1bits);
        if (!src2) { conv_wgrad(e.c1, dact1, x, g1, true); return; }
        for (int h = 0; h < 2; ++h) {          // paired pass: the weight gradient of conv1 once per frame source
            const Conv1Src& sh = h ? *src2 : src;
            const long long foff = h ? Nf / 2 : 0;
            wgrad_src = sh;
            const float* xh = nullptr;
            if constexpr (std::is_same<T, float>::value) xh = sh.u8 ? x32[e.gripper ? 1 : 0] + foff * 3 * e.IH * e.IH : reinterpret_cast<const float*>(sh.X);
            conv_wgrad(e.c1, dact1 + foff * g1.OH * g1.OW * 32, xh, geom(Nf / 2, e.IH, 3, 8, 4), true);
        }
    }

    // ---------------------------------------------------------------- MLP helper (ReLU between layers, none after last)
    // acts[i] = output of layer i (T) ; last layer output fp32 (outf)
    void mlp_fwd(const T* x, long long ldx, int M, LinW* L, int n, T** acts, float* outf, T* outt) {
        const T* in = x; long long ld = ldx;
        for (int i = 0; i < n; ++i) {
            if (i < n - 1) { EpiP ep = epi(acts[i], false); ep.relu = 1; lin_fwd(in, ld, M, L[i], ep, L[i].N); in = acts[i]; ld = L[i].N; }
            else {
                if (outf) { EpiP ep = epi(outf, true); lin_fwd(in, ld, M, L[i], ep, L[i].N); }
                if (outt) { EpiP ep = epi(outt, false); lin_fwd(in, ld, M, L[i], ep, L[i].N); }
            }
        }
    }
    // dy: T [M][N_last]; x: first-layer input (ldx). dxf: optional fp32 output (accumulating) with map
    T* mlp_dy[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    void mlp_bwd(const T* dy, const T* x, long long ldx, int M, LinW* L, int n, T** acts, T* s0, T* s1, float* dxf, const DenseOut* om, int dx_acc) {
        const T* d = dy;
        if constexpr (std::is_same<T, h16_t>::value) {
            // M <= 64 (every M = B MLP): the data-gradient chain first, every layer's incoming gradient kept in its own buffer, then the weight /
            // bias gradients of ALL layers in one launch (lin_bwd_smallm_batched_kernel) instead of one ~9 us launch per layer
            if (M <= 64 && n <= 8) {
                // one store flag for the whole batch: every layer's weight gradient is in the same state (fresh zeros / stale = first write of the step, or
                // already written = a second backward before the optimizer); a mixed batch (never seen) zeroes its stale tensors and accumulates
                bool all_first = true;
                for (int i = 0; i < n; ++i) { const int z = lazy_find(L[i].dW); all_first = all_first && (z >= 0 ? lazy[z].stale : grads_fresh); }
                for (int i = 0; i < n; ++i) { if (all_first) grad_first(L[i].dW); else grad_ensure_zero(L[i].dW); }
                LinBwdBatch bt{}; bt.M = M; bt.store = all_first ? 1 : 0;
                int blk = 0;
                for (int i = n - 1; i >= 0; --i) {
                    const T* in = i > 0 ? acts[i - 1] : x;
                    const long long ld = i > 0 ? L[i - 1].N : ldx;
                    LinBwdJob& J = bt.j[bt.n++];
                    J.dY = d; J.X = in; J.dW = L[i].dW; J.db = L[i].db; J.ldx = ld; J.lddw = L[i].K; J.N = L[i].N; J.K = L[i].K; J.nx = cdiv(L[i].N, 64); J.blk0 = blk;
                    blk += J.nx * cdiv(L[i].K, 128);
                    if (i > 0) {
                        if (!mlp_dy[i]) mlp_dy[i] = alloc<T>((int64_t)64 * 2048);
                        T* o = L[i].K <= 2048 ? mlp_dy[i] : (T*)nullptr;
                        if (!o) { hulc_set_error("mlp_bwd: hidden width %d exceeds the per-layer gradient buffers", L[i].K); return; }
                        EpiP ep = epi(o, false); ep.mask = acts[i - 1];
                        lin_dgrad(d, M, L[i], ep, dense_out(L[i].K));
                        d = o;
                    } else if (dxf) {
                        EpiP ep = epi(dxf, true); ep.accumulate = dx_acc;
                        lin_dgrad(d, M, L[i], ep, *om);
                    }
                }
                hipLaunchKernelGGL(lin_bwd_smallm_batched_kernel, dim3(blk), dim3(256), 0, st, bt);
                return;
            }
        }
        T* scratch[2] = {s0, s1};
        for (int i = n - 1; i >= 0; --i) {
            const T* in = i > 0 ? acts[i - 1] : x;
            const long long ld = i > 0 ? L[i - 1].N : ldx;
            lin_wgrad(d, in, ld, M, L[i].N, L[i].K, L[i].dW, L[i].K, L[i].db);
            if (i > 0) {
                T* o = scratch[i & 1];
                EpiP ep = epi(o, false); ep.mask = acts[i - 1];
                lin_dgrad(d, M, L[i], ep, dense_out(L[i].K));
                d = o;
            } else if (dxf) {
                EpiP ep = epi(dxf, true); ep.accumulate = dx_acc;
                lin_dgrad(d, M, L[i], ep, *om);
            }
        }
    }

    // ---------------------------------------------------------------- forward pieces (shared by training, validation and rollout)
    // perceptual encoders + goal encoder + plan proposal MLP
    void trunk_fwd(const hulc_batch* b, float dp) {
        const int B = b->B, S = b->S, N = B * S;
        const bool hulc = cfg.kind == HULC_KIND_HULC;
        bool ppx_packed = false;
        // ---- perceptual encoders (concat_encoders.py:59-109): static -> emb[..., 0:64], gripper -> emb[..., 64:128]
        {
            const Conv1Src s2s = conv1_src(cur2, false), s2g = conv1_src(cur2, true);
            const bool tail_fused = enc_tail_fusable();
            enc_fwd(encS, aS, conv1_src(*b, false), N, 0, pair ? &s2s : nullptr, tail_fused);
            STAGE("enc_static_fwd");
            enc_fwd(encG, aG, conv1_src(*b, true), N, 64, pair ? &s2g : nullptr, tail_fused);
            x0_done = false;
            if (tail_fused) enc_tail_fwd_both(N, !mcil && tr_fused_mode && S <= 64 && EMB == 128, S, dp);
            STAGE("enc_gripper_fwd");
        }
        // ---- goal encoder (goal_encoders.py:31-36 / 64-69)
        if (pair) {       // rows [0, Bv): visual goal = emb[:, -1]; rows [Bv, B): language goal
            const int Bv = pairBv, Bl = B - pairBv;
            T* av[2] = {gl1, gl2};
            T* al[2] = {gl1 + (long long)Bv * HID, gl2 + (long long)Bv * HID};
            mlp_fwd(emb + (long long)(S - 1) * EMB, (long long)S * EMB, Bv, vg, 3, av, gl3, nullptr);
            ln_fwd(gl3, GOAL, Bv, GOAL, ln_vg_g, ln_vg_b, goal_t, GOAL, nullptr, 0, goal_st);
            hipLaunchKernelGGL((cast_kernel<float, T>), dim3(cdiv(Bl * LANG, 256)), dim3(256), 0, st, b->lang, lang_t, (long long)Bl * LANG);
            mlp_fwd(lang_t, LANG, Bl, lg, 3, al, gl3 + Bv * GOAL, nullptr);
            ln_fwd(gl3 + Bv * GOAL, GOAL, Bl, GOAL, ln_lg_g, ln_lg_b, goal_t + Bv * GOAL, GOAL, nullptr, 0, goal_st + 2 * Bv);
        } else {
            T* acts[2] = {gl1, gl2};
            // hulc / mcil: the LayerNorm launch also packs the plan proposal's input rows [emb[:,0,:] | goal] (goal_ln_concat_kernel)
            auto goal_ln = [&](const float* g_, const float* b_) {
                if (hulc || mcil) {
                    hipLaunchKernelGGL((goal_ln_concat_kernel<T>), dim3(cdiv(B, 4) + cdiv(B * EMB, 256)), dim3(256), 0, st, gl3, B, GOAL, g_, b_, goal_t, goal_st, emb, (long long)S * EMB, EMB, ppx);
                    ppx_packed = true;
                } else ln_fwd(gl3, GOAL, B, GOAL, g_, b_, goal_t, GOAL, nullptr, 0, goal_st);
            };
            if (b->is_lang) {
                hipLaunchKernelGGL((cast_kernel<float, T>), dim3(cdiv(B * LANG, 256)), dim3(256), 0, st, b->lang, lang_t, (long long)B * LANG);
                mlp_fwd(lang_t, LANG, B, lg, 3, acts, gl3, nullptr);
                goal_ln(ln_lg_g, ln_lg_b);
            } else {
                mlp_fwd(emb + (long long)(S - 1) * EMB, (long long)S * EMB, B, vg, 3, acts, gl3, nullptr);
                goal_ln(ln_vg_g, ln_vg_b);
            }
        }
        // ---- plan proposal (plan_proposal_net.py:42-47)
        if (hulc || mcil) {
            if (!ppx_packed) hipLaunchKernelGGL((concat_pp_kernel<T>), dim3(cdiv(B * (EMB + GOAL), 256)), dim3(256), 0, st, emb, (long long)S * EMB, EMB, goal_t, GOAL, B, ppx);
            mlp_fwd(ppx, EMB + GOAL, B, pp, 5, ppa, pp_logits, nullptr);
        }
        STAGE("goal+pp_fwd");
    }
    // plan recognition transformer -> seq_feat, pr_logits
    void pr_fwd(int B, int S, float dp) {
        const int N = B * S;
        // ---- plan recognition transformer (plan_recognition_net.py:94-117)
        bool fused = false;
        if constexpr (std::is_same<T, h16_t>::value) {
            fused = tr_fused_mode && S <= 64;          // 32 < S <= 64 (config 5): two 32-row halves per window, tr_fused.h WIDE
        }
        if (!(fused && x0_done))
        hipLaunchKernelGGL((posadd_kernel<T>), dim3(cdiv((long long)N * EMB, 256)), dim3(256), 0, st, emb, pos32, B, S, EMB, xf[0], xt[0], dp, site_seed(0),
                           fused ? y2[0] : (float*)nullptr, fused ? y2[1] : (float*)nullptr);
        x0_done = false;
        if constexpr (std::is_same<T, h16_t>::value) {
            if (fused) {        // one launch per encoder layer (tr_fused.h); norm2 of layer 0 is the first step of layer 1's launch, the last norm2 a LayerNorm launch
                for (int l = 0; l < 2; ++l) {
                    TrLayerP q{};
                    q.xin = l ? y2[0] : xf[0]; q.ln_in = l; q.ln_g = tr_n2g[0]; q.ln_b = tr_n2b[0]; q.xf_out = xf[1]; q.xt_out = xt[1]; q.st_out = st2[0];
                    q.Wqkv = tr_in[l].Wfr; q.Wo = tr_out[l].Wfr; q.W1 = tr_l1[l].Wfr; q.W2 = tr_l2[l].Wfr;
                    q.bqkv = tr_in[l].b32; q.bo = tr_out[l].b32; q.b1 = tr_l1[l].b32; q.b2 = tr_l2[l].b32; q.n1g = tr_n1g[l]; q.n1b = tr_n1b[l];
                    q.qkv = qkv[l]; q.Pat = Pat[l]; q.ao = ao[l]; q.y1 = y1[l]; q.st1 = st1[l]; q.x1t = x1t[l]; q.x1f = x1f[l]; q.hff = hff[l]; q.y2 = y2[l];
                    q.B = B; q.S = S; q.dp = dp;
                    q.seed_att = site_seed(1 + 4 * l); q.seed_o = site_seed(2 + 4 * l); q.seed_h = site_seed(3 + 4 * l); q.seed_y = site_seed(4 + 4 * l);
                    // per token: QKV 3 x 128 x 128, out projection 128 x 128, FFN 2 x 128 x 2048, attention 2 x S x 128 MACs
                    TimerScope ts(this, "transformer_fused", "mfma", 2.0 * N * (4.0 * EMB * EMB + 2.0 * EMB * FF + 2.0 * S * EMB), (double)N * (4 * EMB + FF) * sizeof(T));
                    launch_tr_layer_fwd(st, q);
                }
                // the last norm2 and the mean over the window in one launch (the normalised rows are kept for tests only)
                hipLaunchKernelGGL((layernorm_mean_kernel<T>), dim3(B), dim3(1024), 0, st, y2[1], S, EMB, tr_n2g[1], tr_n2b[1], st2[1], xm, xf[2]);
            }
        }
        for (int l = 0; l < 2 && !fused; ++l) {
            { EpiP ep = epi(qkv[l], false); lin_fwd(xt[l], EMB, N, tr_in[l], ep, 3 * EMB); }
            // two lanes per query row in the 16-bit engines; the fp32 (parity) engine keeps the one-lane kernel's summation order: the hulc_visonly
            // fixture has an FFN pre-activation within fp32 epsilon of zero, and an epsilon-level change upstream flips its ReLU (1e-3 gradient gate)
            static const bool att32 = (HULC_SWITCH("HULC_ATT32", 1) != 0) && !std::is_same<T, float>::value;
            if (S <= 32 && att32) hipLaunchKernelGGL((attention_fwd32_kernel<T>), dim3(B * NH), dim3(64), 0, st, qkv[l], B, S, EMB, NH, Pat[l], ao[l], dp, site_seed(1 + 4 * l));
            else if (S <= 32) hipLaunchKernelGGL((attention_fwd_kernel<T, 32>), dim3(B * NH), dim3(64), 0, st, qkv[l], B, S, EMB, NH, Pat[l], ao[l], dp, site_seed(1 + 4 * l));
            else if (att32) hipLaunchKernelGGL((attention_fwd64_kernel<T>), dim3(B * NH), dim3(256), 0, st, qkv[l], B, S, EMB, NH, Pat[l], ao[l], dp, site_seed(1 + 4 * l));
            else hipLaunchKernelGGL((attention_fwd_kernel<T, 64>), dim3(B * NH), dim3(64), 0, st, qkv[l], B, S, EMB, NH, Pat[l], ao[l], dp, site_seed(1 + 4 * l));
            { EpiP ep = epi(y1[l], true); ep.res = xf[l]; ep.res_f32 = 1; ep.res_ld = EMB; ep.res_late = 1; ep.drop_p = dp; ep.drop_seed = site_seed(2 + 4 * l);
              lin_fwd(ao[l], EMB, N, tr_out[l], ep, EMB); }
            ln_fwd(y1[l], EMB, N, EMB, tr_n1g[l], tr_n1b[l], x1t[l], EMB, x1f[l], EMB, st1[l]);
            { EpiP ep = epi(hff[l], false); ep.relu = 1; ep.drop_p = dp; ep.drop_seed = site_seed(3 + 4 * l); lin_fwd(x1t[l], EMB, N, tr_l1[l], ep, FF); }
            { EpiP ep = epi(y2[l], true); ep.res = x1f[l]; ep.res_f32 = 1; ep.res_ld = EMB; ep.res_late = 1; ep.drop_p = dp; ep.drop_seed = site_seed(4 + 4 * l);
              lin_fwd(hff[l], FF, N, tr_l2[l], ep, EMB); }
            ln_fwd(y2[l], EMB, N, EMB, tr_n2g[l], tr_n2b[l], xt[l + 1], EMB, xf[l + 1], EMB, st2[l]);
        }
        // mean over S commutes with the affine fc (:113-114): seq_feat = fc(mean_t x)
        if (!fused) hipLaunchKernelGGL((mean_over_s_kernel<T>), dim3(cdiv(B * EMB, 256)), dim3(256), 0, st, xf[2], B, S, EMB, xm);
        { EpiP ep = epi(seqf, true); ep.out2 = seqf_t; ep.out2_lo = 0; ep.out2_hi = (long long)B * FCH;      // the 16-bit copy the next GEMM reads: a second store of the epilogue
          lin_fwd(xm, EMB, B, pr_fc, ep, FCH); }
        { EpiP ep = epi(pr_logits, true); lin_fwd(seqf_t, FCH, B, pr_fs, ep, PLAN); }
        STAGE("plan_recognition_fwd");
    }
    // action decoder up to the packed heads [S*B][NHEAD] (logistic_decoder_rnn.py:260-287); plan/goal terms hoisted out of the time loop.
    // h0_0 / h0_1: previous hidden states [B][HID] of the two layers (stateful rollout, :107-111) or null (h0 = 0).
    void dec_fwd(const int* plan_idx, int B, int S, const T* h0_0, const T* h0_1) {
        const int SB = S * B;
        const bool hulc = cfg.kind == HULC_KIND_HULC;
        // 16-bit engines (hulc / gcbc): the goal term of the time-invariant decoder input is added inside the plan-gather launch; the fp32 (parity)
        // engine and mcil (whose plan term is a GEMM already) keep the K = 32 GEMM and its summation order
        const bool cb_fused = !mcil && !std::is_same<T, float>::value;
            // time-major copy of the gripper half of emb: embg[t*B+b][0:64] = emb[b][t][64:128]
            if (mcil) {      // continuous plan (B,256): a GEMM against the plan columns of W_ih0 instead of the one-hot column gather
                hipLaunchKernelGGL((gather_embg_kernel<T>), dim3(cdiv(SB * DE, 256)), dim3(256), 0, st, emb, embg, B, S, DE);
                EpiP ep = epi(Cplan, true); ep.bias = bih0; ep.bias2 = bhh0;
                gemm(dense<T>(plan_t, B, dec_plan), dense<T>(wih0, HID, KIN), dense_out(HID), ep, B, HID, dec_plan);
            } else      // the one-hot plan gather and the time-major embedding copy are independent: one launch
            hipLaunchKernelGGL((plan_gather_t_kernel<T>), dim3(cdiv(B * HID, 256) + cdiv(SB * DE, 256)), dim3(256), 0, st, wih0T, plan_idx, B, hulc ? NCAT : 0, NCLS, HID, bih0,
                               bhh0, Cplan, emb, embg, S, DE, goal_t, GOAL, dec_plan + DE, cb_fused ? Cb : (T*)nullptr);
            if (!cb_fused) { EpiP ep = epi(Cb, false); ep.res = Cplan; ep.res_f32 = 1; ep.res_ld = HID;
              gemm(dense<T>(goal_t, B, GOAL), dense<T>(wih0 + dec_plan + DE, HID, KIN), dense_out(HID), ep, B, HID, GOAL); }
            const long long BH = (long long)B * HID;
            { EpiP ep = epi(Zx0, false); ep.res = Cb; ep.res_ld = HID; ep.res_rowmod = B;
              if (!h0_0) { ep.out2 = H0; ep.out2_lo = 0; ep.out2_hi = BH; ep.out2_relu = 1; }      // h_{-1} = 0: H0[0] = relu(Zx0[0]) written here
              gemm(dense<T>(embg, SB, DE), dense<T>(wih0 + dec_plan, HID, KIN), dense_out(HID), ep, SB, HID, DE); }
            rnn_fwd(Zx0, H0, whh0, B, S, h0_0, 1, false, !h0_0);
            { EpiP ep = epi(Zx1, false); ep.bias = bih1; ep.bias2 = bhh1;
              if (!h0_1) { ep.out2 = H1; ep.out2_lo = 0; ep.out2_hi = BH; ep.out2_relu = 1; }
              gemm(dense<T>(H0, SB, HID), dense<T>(wih1.W, HID, HID), dense_out(HID), ep, SB, HID, HID); }
            rnn_fwd(Zx1, H1, whh1, B, S, h0_1, 1, false, !h0_1);
            { EpiP ep = epi(heads, true); ep.bias = bheads;
              gemm(dense<T>(H1, SB, HID), dense<T>(wheads, NHEAD, HID), dense_out(NHEAD), ep, SB, NHEAD, HID); }
    }

    // ---------------------------------------------------------------- forward
    int forward(const hulc_batch* b, float lw, float cw, float* out, int on_host) override {
        pair = false;
        int rc = forward_impl(b, lw, cw, out, on_host);
        // a persistent recurrence of THIS forward timed out and the stream is drained (losses read back): run it again, one launch per step
        if (!rc && out && on_host && persist_check("hulc_forward_loss", true)) rc = forward_impl(b, lw, cw, out, on_host);
        return rc;
    }
    // vis + lang windows of one step as ONE pass over Bv + Bl windows (hulc.py:433-469 runs them one after the other): the encoders, plan
    // networks and the decoder are shared, only the goal encoder (rows [0,Bv): visual, [Bv,B): language) and the CLIP rows differ.  The
    // latency-bound part of the step (recurrent steps, M <= 64 GEMMs, transformer) then runs once at 2B rows instead of twice at B.
    // Needs Bv == Bl (the per-modality means then share one gradient scale).  out: [total, kl, action, clip] of vis, then of lang.
    std::vector<int> aux_host;
    int* pidx_j = nullptr;
    int forward_pair(const hulc_batch* vb, const hulc_batch* lb, float lw, float cw, float* out, int on_host) override {
        if (!bound) { hulc_set_error("hulc_forward_loss_pair before hulc_bind_params"); return 1; }
        if (vb->is_lang || !lb->is_lang || !lb->lang) { hulc_set_error("hulc_forward_loss_pair: first batch must be the vis modality, second the lang modality with embeddings"); return 1; }
        if (vb->B != lb->B || vb->S != lb->S) { hulc_set_error("hulc_forward_loss_pair: both modalities need the same B and S (got %dx%d and %dx%d)", vb->B, vb->S, lb->B, lb->S); return 1; }
        for (const hulc_batch* b : {vb, lb})
            if (b->window_start && (!b->frames_u8 || b->store_frames < b->S)) { hulc_set_error("window_start (frame store) needs frames_u8 and store_frames >= S (got frames_u8=%d, store_frames=%lld, S=%d)", b->frames_u8, (long long)b->store_frames, b->S); return 1; }
        if (vb->frames_u8 != lb->frames_u8 || vb->actions_absolute != lb->actions_absolute || vb->max_rel_pos != lb->max_rel_pos || vb->max_rel_orn != lb->max_rel_orn) {
            hulc_set_error("hulc_forward_loss_pair: both modalities must use the same ingest options"); return 1; }
        if ((vb->plan_idx != nullptr) != (lb->plan_idx != nullptr) || (vb->plan_eps != nullptr) != (lb->plan_eps != nullptr)) {
            hulc_set_error("hulc_forward_loss_pair: inject the plan draw for both modalities or for neither"); return 1; }
        const int Bv = vb->B, B = 2 * Bv, S = vb->S;
        if (B > maxB) { hulc_set_error("hulc_forward_loss_pair: %d + %d windows exceed max_batch=%d", Bv, Bv, maxB); return 1; }
        if (!act_j) {
            act_j = alloc<float>((int64_t)maxN * 7); ro_j = alloc<float>((int64_t)maxN * 15); eps_j = alloc<float>((int64_t)maxB * 256);
            pidx_j = alloc<int>((int64_t)maxB * NCAT); losses2 = alloc<float>(8);
            if (alloc_failed) { hulc_set_error("hulc_forward_loss_pair: workspace allocation failed"); return 1; }
        }
        const size_t na = sizeof(float) * Bv * S * 7, nr = sizeof(float) * Bv * S * 15;
        MultiCopy mc{}; int nseg = 0;          // the joins of the two modalities' actions / robot_obs (device memory by the hulc_batch contract): one launch
        auto seg = [&](void* dst, const void* src, size_t bytes) { mc.dst[nseg] = (unsigned*)dst; mc.src[nseg] = (const unsigned*)src; mc.words[nseg] = (int)(bytes / 4); ++nseg; };
        seg(act_j, vb->actions, na); seg((char*)act_j + na, lb->actions, na);
        seg(ro_j, vb->robot_obs, nr); seg((char*)ro_j + nr, lb->robot_obs, nr);
        hulc_batch jb = *vb;
        jb.B = B; jb.actions = act_j; jb.robot_obs = ro_j; jb.lang = lb->lang; jb.is_lang = 0;
        hipLaunchKernelGGL(multi_copy_kernel, dim3(16, nseg), dim3(256), 0, st, mc);
        if (vb->plan_idx) {                    // injected draws (parity tests) may live in host memory: plain copies
            HIP_CHECK(hipMemcpyAsync(pidx_j, vb->plan_idx, sizeof(int) * Bv * NCAT, hipMemcpyDefault, st));
            HIP_CHECK(hipMemcpyAsync(pidx_j + Bv * NCAT, lb->plan_idx, sizeof(int) * Bv * NCAT, hipMemcpyDefault, st));
            jb.plan_idx = pidx_j;
        }
        if (vb->plan_eps) {
            const size_t ne = sizeof(float) * Bv * (PLAN / 2);
            HIP_CHECK(hipMemcpyAsync(eps_j, vb->plan_eps, ne, hipMemcpyDefault, st)); HIP_CHECK(hipMemcpyAsync((char*)eps_j + ne, lb->plan_eps, ne, hipMemcpyDefault, st));
            jb.plan_eps = eps_j;
        }
        aux_host.assign(lb->aux_rows ? lb->aux_rows : nullptr, lb->aux_rows ? lb->aux_rows + lb->n_aux : nullptr);
        for (int& r : aux_host) r += Bv;
        jb.aux_rows = aux_host.data(); jb.n_aux = lb->aux_rows ? lb->n_aux : 0;
        pair = true; pairBv = Bv; cur2 = *lb;
        int rc = forward_impl(&jb, lw, cw, out, on_host);
        if (!rc && out && on_host && persist_check("hulc_forward_loss_pair", true)) rc = forward_impl(&jb, lw, cw, out, on_host);
        return rc;
    }
    int forward_impl(const hulc_batch* b, float lw, float cw, float* out, int on_host) {
        if (!bound) { hulc_set_error("hulc_forward_loss before hulc_bind_params"); return 1; }
        persist_check("hulc_forward_loss", false);
        if (b->B < 1 || b->S < 1 || b->B > maxB || b->S > maxS || b->S > cfg.max_window || b->S > 64) {
            hulc_set_error("batch (B=%d,S=%d) exceeds workspace (max_batch=%d,max_seq=%d,max_window=%d)", b->B, b->S, maxB, maxS, cfg.max_window);
            return 1;
        }
        if (b->is_lang && !b->lang) { hulc_set_error("lang modality batch without language embeddings (hulc.py:440 KeyError 'lang')"); return 1; }
        if (b->actions_absolute && !(b->max_rel_pos > 0.f && b->max_rel_orn > 0.f)) { hulc_set_error("actions_absolute needs max_rel_pos > 0 and max_rel_orn > 0 (RelativeActions, transforms.py:35-37)"); return 1; }
        if (b->window_start && (!b->frames_u8 || b->store_frames < b->S)) { hulc_set_error("window_start (frame store) needs frames_u8 and store_frames >= S (got frames_u8=%d, store_frames=%lld, S=%d)", b->frames_u8, (long long)b->store_frames, b->S); return 1; }
        cur = *b; cur_lw = lw; cur_cw = cw; have_fwd = false; val_clip_n = 0;
        const int B = b->B, S = b->S, N = B * S, SB = S * B;
        const bool hulc = cfg.kind == HULC_KIND_HULC;
        const float dp = cfg.dropout_p;
        const int Bm = pair ? pairBv : B;                    // windows per modality: the reference's means (and so the gradient scales) are per modality
        const float* kl_src = nullptr; int kl_n = 0;
        if constexpr (std::is_same<T, h16_t>::value) {      // cleared by the first conv1 launch of trunk_fwd (nothing accumulates a loss before the encoders are done)
            pend_zero[0] = losses; pend_zero[1] = pair ? losses2 : nullptr;
        } else {
            HIP_CHECK(hipMemsetAsync(losses, 0, 8 * sizeof(float), st));
            if (pair) HIP_CHECK(hipMemsetAsync(losses2, 0, 8 * sizeof(float), st));
        }
        trunk_fwd(b, dp);
        if (pend_zero[0]) { hulc_set_error("forward: the loss accumulators were not cleared (no conv1 launch took them)"); return 1; }
        if (mcil) {
            if (gru) bigru_fwd(B, S); else birnn_fwd(B, S);
            const int n = PLAN / 2;
            const float* eps = nullptr;
            if (b->plan_eps) { HIP_CHECK(hipMemcpyAsync(plan_eps_in, b->plan_eps, sizeof(float) * B * n, hipMemcpyDefault, st)); eps = plan_eps_in; }
            const float wpp = lw * cfg.kl_beta * cfg.kl_balancing_mix / Bm, wpr = lw * cfg.kl_beta * (1.f - cfg.kl_balancing_mix) / Bm;
            hipLaunchKernelGGL((normal_kl_sample_kernel<T>), dim3(cdiv(B * n, 256)), dim3(256), 0, st, pr_logits, pp_logits, B, n, eps, plan_eps, plan_f, plan_t, klel,
                               dpp_kl, dpr_kl, wpp, wpr, site_seed(20), lscale());
            if (pair) {
                hipLaunchKernelGGL(sum_reduce_kernel, dim3(1), dim3(256), 0, st, klel, Bm * n, cfg.kl_beta / Bm, losses + 1);
                hipLaunchKernelGGL(sum_reduce_kernel, dim3(1), dim3(256), 0, st, klel + (long long)Bm * n, Bm * n, cfg.kl_beta / Bm, losses2 + 1);
            } else { kl_src = klel; kl_n = Bm * n; }      // summed by finish_losses_kernel at the end of the forward
        } else pr_fwd(B, S, dp);
        // ---- sample + KL (hulc.py:289-296, 539-561)
        if (hulc) {
            const int* idx_in = nullptr;
            if (b->plan_idx) { HIP_CHECK(hipMemcpyAsync(pidx_in, b->plan_idx, sizeof(int) * B * NCAT, hipMemcpyDefault, st)); idx_in = pidx_in; }
            const float wpp = lw * cfg.kl_beta * cfg.kl_balancing_mix / Bm, wpr = lw * cfg.kl_beta * (1.f - cfg.kl_balancing_mix) / Bm;
            hipLaunchKernelGGL(plan_kl_sample_kernel, dim3(B * NCAT), dim3(64), 0, st, pr_logits, pp_logits, B, NCAT, NCLS, idx_in, pidx, probs, klcat, dpp_kl,
                               dpr_kl, wpp, wpr, site_seed(20), lscale());
            if (pair) {
                hipLaunchKernelGGL(sum_reduce_kernel, dim3(1), dim3(256), 0, st, klcat, Bm * NCAT, cfg.kl_beta / Bm, losses + 1);
                hipLaunchKernelGGL(sum_reduce_kernel, dim3(1), dim3(256), 0, st, klcat + Bm * NCAT, Bm * NCAT, cfg.kl_beta / Bm, losses2 + 1);
            } else { kl_src = klcat; kl_n = Bm * NCAT; }
        }
        // ---- action decoder (logistic_decoder_rnn.py:260-287): plan/goal terms hoisted out of the time loop
        {
            dec_fwd(pidx, B, S, nullptr, nullptr);
            // mcil_default.yaml: gripper_control false (no tcp-frame transform), discrete_gripper false (7th mixture dimension instead of the CE head)
            static const int ll_block = HULC_SWITCH("HULC_LL_BLOCK", 64);     // one wave per workgroup: 256 CUs x 1 wave instead of 64 CUs x 4 (the kernel is one long serial chain per thread)
            // 16-bit engines: one lane per mixture component (kernels.h logistic_loss_wide_kernel: 17.9 -> ~6 us); the fp32 parity engine keeps the serial kernel's summation order
            static const int ll_wide = HULC_SWITCH("HULC_LL_WIDE", 1);
            if (ll_wide && !std::is_same<T, float>::value && NMIX <= 16 && NDIM <= 7)
                hipLaunchKernelGGL((logistic_loss_wide_kernel<T>), dim3(SB), dim3(128), 0, st, heads, NHEAD, actions_of(*b), b->robot_obs, B, S, NMIX, NDIM,
                                   cfg.num_classes, cfg.log_scale_min, cfg.gripper_alpha, mcil ? 0 : 1, lw / (float)(S * Bm), rowloss, a_tcp, dheads, mcil ? 0 : 1, lscale());
            else
            hipLaunchKernelGGL((logistic_loss_kernel<T, NMIX>), dim3(cdiv(SB * 8, ll_block)), dim3(ll_block), 0, st, heads, NHEAD, actions_of(*b), b->robot_obs, B, S, NMIX, NDIM,
                               cfg.num_classes, cfg.log_scale_min, cfg.gripper_alpha, mcil ? 0 : 1, lw / (float)(S * Bm), rowloss, a_tcp, dheads, mcil ? 0 : 1, lscale());
            if (pair) hipLaunchKernelGGL(sum_rows_pair_kernel, dim3(1), dim3(256), 0, st, rowloss, SB, B, pairBv, 1.f / (S * Bm), losses + 0, losses2 + 0);
            // one modality: the action-loss sum, the KL sum, the packing and the copy to a device `out` are ONE launch at the end (finish_losses_kernel)
        }
        STAGE("decoder_fwd");
        // ---- CLIP auxiliary loss (hulc.py:650-695), lang modality, masked rows
        clip_n = 0;
        if ((b->is_lang || pair) && cfg.use_clip && b->n_aux > 0) {
            const int n = b->n_aux;
            if (n > 64 || n > B) { hulc_set_error("clip aux rows n=%d unsupported (max 64, <= B)", n); return 1; }
            clip_n = n;
            HIP_CHECK(hipMemcpyAsync(auxrows, b->aux_rows, sizeof(int) * n, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((gather_rows_kernel<T, T>), dim3(cdiv(n * FCH, 256)), dim3(256), 0, st, seqf_t, (long long)FCH, auxrows, n, FCH, sf_m);
            hipLaunchKernelGGL((gather_rows_kernel<T, T>), dim3(cdiv(n * GOAL, 256)), dim3(256), 0, st, goal_t, (long long)GOAL, auxrows, n, GOAL, g_m);
            { EpiP ep = epi(im1, false); ep.relu = 1; lin_fwd(sf_m, FCH, n, cl_im0, ep, 128); }
            { EpiP ep = epi(img, true); lin_fwd(im1, 128, n, cl_im2, ep, GOAL); }
            { EpiP ep = epi(la1, false); ep.relu = 1; lin_fwd(g_m, GOAL, n, cl_la0, ep, 128); }
            { EpiP ep = epi(txt, true); lin_fwd(la1, 128, n, cl_la2, ep, GOAL); }
            static const bool clip_wide = HULC_SWITCH("HULC_CLIP_WIDE", 1) != 0;
            if (clip_wide && GOAL <= 32 && !std::is_same<T, float>::value)
                hipLaunchKernelGGL(clip_loss_wide_kernel, dim3(1), dim3(1024), 0, st, img, txt, n, GOAL, logit_scale, cw, (pair ? losses2 : losses) + 2, dimg, dtxt, dlogit_scale, lscale());
            else
            hipLaunchKernelGGL(clip_loss_kernel, dim3(1), dim3(64), 0, st, img, txt, n, GOAL, logit_scale, cw, (pair ? losses2 : losses) + 2, dimg, dtxt, dlogit_scale, lscale());
        }
        STAGE("clip_fwd");
        if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in forward"); return 1; }
        have_fwd = true;
        // [total_mod, kl, action, clip]; a device `out` is written by the kernels themselves
        float* const dev_out = (out && !on_host) ? out : nullptr;
        if (!pair) {     // + the backward's zero arena (4 MB), cleared by 255 more blocks of the same launch instead of a memset at the head of the backward
            hipLaunchKernelGGL(finish_losses_kernel, dim3(256), dim3(256), 0, st, rowloss, SB * 8, 1.f / SB, kl_src, kl_n, cfg.kl_beta / Bm, losses, dev_out,
                               reinterpret_cast<float4*>(zero_arena), (long long)(zero_n / 4));
            arena_clean = true;
        }
        else if (out) {
            hipLaunchKernelGGL(pack_losses_kernel, dim3(1), dim3(1), 0, st, losses, dev_out);
            hipLaunchKernelGGL(pack_losses_kernel, dim3(1), dim3(1), 0, st, losses2, dev_out ? dev_out + 4 : (float*)nullptr);
        }
        if (out && on_host) {
            HIP_CHECK(hipMemcpyAsync(out, losses + 4, 4 * sizeof(float), hipMemcpyDeviceToHost, st));
            if (pair) HIP_CHECK(hipMemcpyAsync(out + 4, losses2 + 4, 4 * sizeof(float), hipMemcpyDeviceToHost, st));
            HIP_CHECK(hipStreamSynchronize(st));
        }
        return 0;
    }
    int clip_n = 0;

    // ---------------------------------------------------------------- validation forward (hulc.py:301-388, 770-797)
    float* valm = nullptr;            // [32]: 0 loss_pp, 1 loss_pr, 2 kl; 8..14 pp (mae[6], sr); 16..22 pr
    float *pred_pp = nullptr, *pred_pr = nullptr, *nz_mix = nullptr, *nz_act = nullptr;
    int* pidx_pp = nullptr;
    void val_alloc() {
        if (valm) return;
        valm = alloc<float>(32); pred_pp = alloc<float>((int64_t)maxB * maxS * 7); pred_pr = alloc<float>((int64_t)maxB * maxS * 7);
        nz_mix = alloc<float>((int64_t)maxB * maxS * NDIM * NMIX); nz_act = alloc<float>((int64_t)maxB * maxS * NDIM);
        pidx_pp = alloc<int>((int64_t)maxB * NCAT);
    }
    // mcil: plan ~ N(mean, std) of `state` (B, PLAN) into plan_f / plan_t, or the injected (B, PLAN/2) draw; with kl_with the per-element
    // KL(state || kl_with) lands in klel (no gradient weights)
    void sample_cont(const float* state, const float* kl_with, const float* inject, int B, uint64_t seed) {
        const int n = PLAN / 2;
        hipLaunchKernelGGL((normal_kl_sample_kernel<T>), dim3(cdiv(B * n, 256)), dim3(256), 0, st, state, kl_with, B, n, (const float*)nullptr, plan_eps, plan_f, plan_t, klel,
                           dpp_kl, dpr_kl, 0.f, 0.f, seed);
        if (inject) {
            hipMemcpyAsync(plan_f, inject, sizeof(float) * B * n, hipMemcpyDefault, st);
            hipLaunchKernelGGL((cast_kernel<float, T>), dim3(cdiv(B * n, 256)), dim3(256), 0, st, plan_f, plan_t, (long long)B * n);
        }
    }
    int validate(const hulc_batch* b, const hulc_val_noise* nz, float* out17, int32_t* plan_pp_out, int32_t* plan_pr_out, float* pred_pp_out,
                 float* pred_pr_out) override {
        persist_check("hulc_validate", false);
        int rc = validate_impl(b, nz, out17, plan_pp_out, plan_pr_out, pred_pp_out, pred_pr_out);
        // validate ends with a stream synchronisation: a timed-out persistent recurrence is visible here -> the metrics are recomputed on the per-step path
        if (!rc && persist_check("hulc_validate", true)) rc = validate_impl(b, nz, out17, plan_pp_out, plan_pr_out, pred_pp_out, pred_pr_out);
        return rc;
    }
    int validate_impl(const hulc_batch* b, const hulc_val_noise* nz, float* out17, int32_t* plan_pp_out, int32_t* plan_pr_out, float* pred_pp_out,
                      float* pred_pr_out) {
        if (!bound) { hulc_set_error("hulc_validate before hulc_bind_params"); return 1; }
        const bool hulc = cfg.kind == HULC_KIND_HULC;     // GCBC (gcbc.py:214-246): one decoder pass without a plan, reported in the "pp" slots
        if (b->B < 1 || b->S < 1 || b->B > maxB || b->S > maxS || b->S > cfg.max_window || b->S > 64) {
            hulc_set_error("batch (B=%d,S=%d) exceeds workspace (max_batch=%d,max_seq=%d,max_window=%d)", b->B, b->S, maxB, maxS, cfg.max_window);
            return 1;
        }
        if (b->is_lang && !b->lang) { hulc_set_error("lang modality batch without language embeddings (hulc.py:440 KeyError 'lang')"); return 1; }
        if (b->actions_absolute && !(b->max_rel_pos > 0.f && b->max_rel_orn > 0.f)) { hulc_set_error("actions_absolute needs max_rel_pos > 0 and max_rel_orn > 0 (RelativeActions, transforms.py:35-37)"); return 1; }
        val_alloc();
        if (alloc_failed) { hulc_set_error("hulc_validate: workspace allocation failed"); return 1; }
        if (b->window_start && (!b->frames_u8 || b->store_frames < b->S)) { hulc_set_error("window_start (frame store) needs frames_u8 and store_frames >= S"); return 1; }
        static const hulc_val_noise none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (!nz) nz = &none;
        cur = *b; have_fwd = false; pair = false; val_clip_n = 0;
        const int B = b->B, S = b->S, SB = S * B;
        HIP_CHECK(hipMemsetAsync(valm, 0, 32 * sizeof(float), st));
        trunk_fwd(b, 0.f);
        if (gru) bigru_fwd(B, S); else if (mcil) birnn_fwd(B, S); else pr_fwd(B, S, 0.f);
        // KL (beta-scaled) + recognition sample, then the proposal sample (no KL terms: second logits pointer null)
        const int* in_pr = nullptr;
        if (hulc) {
        if (nz->plan_idx_pr) { HIP_CHECK(hipMemcpyAsync(pidx_in, nz->plan_idx_pr, sizeof(int) * B * NCAT, hipMemcpyDefault, st)); in_pr = pidx_in; }
        hipLaunchKernelGGL(plan_kl_sample_kernel, dim3(B * NCAT), dim3(64), 0, st, pr_logits, pp_logits, B, NCAT, NCLS, in_pr, pidx, probs, klcat, dpp_kl, dpr_kl, 0.f,
                           0.f, site_seed(40));
        hipLaunchKernelGGL(sum_reduce_kernel, dim3(1), dim3(256), 0, st, klcat, B * NCAT, cfg.kl_beta / B, valm + 2);
        const int* in_pp = nullptr;
        if (nz->plan_idx_pp) { HIP_CHECK(hipMemcpyAsync(pidx_in, nz->plan_idx_pp, sizeof(int) * B * NCAT, hipMemcpyDefault, st)); in_pp = pidx_in; }
        hipLaunchKernelGGL(plan_kl_sample_kernel, dim3(B * NCAT), dim3(64), 0, st, pp_logits, (const float*)nullptr, B, NCAT, NCLS, in_pp, pidx_pp, probs, klcat,
                           dpp_kl, dpr_kl, 0.f, 0.f, site_seed(41));
        }
        const float* acts = actions_of(*b);
        for (int pass = 0; pass < ((hulc || mcil) ? 2 : 1); ++pass) {          // 0: plan proposal, 1: plan recognition (loss_and_act, logistic_decoder_rnn.py:85-100)
            if (mcil) {      // continuous plan: Independent(Normal).sample() (distributions.py:37-38) or the injected draw (B,256) fp32
                sample_cont(pass == 0 ? pp_logits : pr_logits, pass == 0 ? nullptr : pp_logits, reinterpret_cast<const float*>(pass == 0 ? nz->plan_idx_pp : nz->plan_idx_pr),
                            B, site_seed(41 - pass));
                if (pass == 1) hipLaunchKernelGGL(sum_reduce_kernel, dim3(1), dim3(256), 0, st, klel, B * (PLAN / 2), cfg.kl_beta / B, valm + 2);
                int32_t* po = pass == 0 ? plan_pp_out : plan_pr_out;
                if (po) HIP_CHECK(hipMemcpyAsync(po, plan_f, sizeof(float) * B * (PLAN / 2), hipMemcpyDefault, st));
            }
            dec_fwd(pass == 0 ? pidx_pp : pidx, B, S, nullptr, nullptr);
            hipLaunchKernelGGL((logistic_loss_kernel<T, NMIX>), dim3(cdiv(SB * 8, 256)), dim3(256), 0, st, heads, NHEAD, acts, b->robot_obs, B, S, NMIX, NDIM,
                               cfg.num_classes, cfg.log_scale_min, cfg.gripper_alpha, mcil ? 0 : 1, 0.f, rowloss, a_tcp, dheads, mcil ? 0 : 1);
            hipLaunchKernelGGL(sum_reduce_kernel, dim3(1), dim3(256), 0, st, rowloss, SB * 8, 1.f / SB, valm + pass);
            const float* um = pass == 0 ? nz->u_mix_pp : nz->u_mix_pr;
            const float* ua = pass == 0 ? nz->u_act_pp : nz->u_act_pr;
            if (um) { HIP_CHECK(hipMemcpyAsync(nz_mix, um, sizeof(float) * SB * NDIM * NMIX, hipMemcpyDefault, st)); um = nz_mix; }
            if (ua) { HIP_CHECK(hipMemcpyAsync(nz_act, ua, sizeof(float) * SB * NDIM, hipMemcpyDefault, st)); ua = nz_act; }
            hipLaunchKernelGGL(logistic_sample_kernel, dim3(cdiv(SB, 64)), dim3(64), 0, st, heads, NHEAD, b->robot_obs, acts, um, ua, B, S, NMIX, NDIM,
                               cfg.log_scale_min, mcil ? 0 : 1, site_seed(42 + pass), pass == 0 ? pred_pp : pred_pr, valm + 8 + 8 * pass, mcil ? 0 : 1);
        }
        // val/val_pred_clip_loss (hulc.py:804-808): the CLIP auxiliary loss of the lang modality on the masked rows, forward only
        if (b->is_lang && cfg.use_clip && b->n_aux > 0) {
            const int n = b->n_aux;
            if (n > 64 || n > B) { hulc_set_error("clip aux rows n=%d unsupported (max 64, <= B)", n); return 1; }
            HIP_CHECK(hipMemcpyAsync(auxrows, b->aux_rows, sizeof(int) * n, hipMemcpyHostToDevice, st));
            hipLaunchKernelGGL((gather_rows_kernel<T, T>), dim3(cdiv(n * FCH, 256)), dim3(256), 0, st, seqf_t, (long long)FCH, auxrows, n, FCH, sf_m);
            hipLaunchKernelGGL((gather_rows_kernel<T, T>), dim3(cdiv(n * GOAL, 256)), dim3(256), 0, st, goal_t, (long long)GOAL, auxrows, n, GOAL, g_m);
            { EpiP ep = epi(im1, false); ep.relu = 1; lin_fwd(sf_m, FCH, n, cl_im0, ep, 128); }
            { EpiP ep = epi(img, true); lin_fwd(im1, 128, n, cl_im2, ep, GOAL); }
            { EpiP ep = epi(la1, false); ep.relu = 1; lin_fwd(g_m, GOAL, n, cl_la0, ep, 128); }
            { EpiP ep = epi(txt, true); lin_fwd(la1, 128, n, cl_la2, ep, GOAL); }
            hipLaunchKernelGGL(clip_loss_kernel, dim3(1), dim3(64), 0, st, img, txt, n, GOAL, logit_scale, 0.f, valm + 3, dimg, dtxt, valm + 31);
            val_clip_n = n;
        }
        STAGE("validate");
        if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in validate"); return 1; }
        if (plan_pp_out && hulc) HIP_CHECK(hipMemcpyAsync(plan_pp_out, pidx_pp, sizeof(int) * B * NCAT, hipMemcpyDefault, st));
        if (plan_pr_out && hulc) HIP_CHECK(hipMemcpyAsync(plan_pr_out, pidx, sizeof(int) * B * NCAT, hipMemcpyDefault, st));
        if (pred_pp_out) HIP_CHECK(hipMemcpyAsync(pred_pp_out, pred_pp, sizeof(float) * SB * 7, hipMemcpyDefault, st));
        if (pred_pr_out && (hulc || mcil)) HIP_CHECK(hipMemcpyAsync(pred_pr_out, pred_pr, sizeof(float) * SB * 7, hipMemcpyDefault, st));
        float h[32];
        HIP_CHECK(hipMemcpyAsync(h, valm, sizeof(h), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (out17) {
            out17[0] = h[0]; out17[1] = h[1]; out17[2] = h[2]; out17[3] = h[14]; out17[4] = h[22];
            for (int i = 0; i < 6; ++i) { out17[5 + i] = h[8 + i]; out17[11 + i] = h[16 + i]; }
            out17[17] = h[3];
        }
        return 0;
    }

    // ---------------------------------------------------------------- CLIP ground-truth metric (hulc.py:967-974, 980-1043)
    // encode: language_goal(lang_emb) then proj_vis_lang.mlp_lang, kept as fp32 (m, GOAL) in `slot` (0: training instructions, 1: validation
    // instructions) until the next encode of that slot.  scores: exp(logit_scale) * normalised image projections of the masked rows of the LAST
    // hulc_validate (lang modality) times the normalised slot rows -> (n, m) on the host.  Both overwrite goal-encoder activations: not between a
    // forward and its backward.
    int val_clip_n = 0;
    float* gt_txt[2] = {nullptr, nullptr}; int gt_m[2] = {0, 0}, gt_cap[2] = {0, 0};
    float *gt_in = nullptr, *gt_out = nullptr; int64_t gt_in_cap = 0, gt_out_cap = 0;
    template <typename U> bool gt_grow(U*& p, int64_t& cap, int64_t n) {
        if (n <= cap) return true;
        void* q = nullptr;
        if (hipMalloc(&q, n * sizeof(U) + 256) != hipSuccess) return false;
        allocs.push_back(q);       // the old block stays until the engine goes: kernels in flight may still read it
        p = (U*)q; cap = n;
        return true;
    }
    int clip_gt_encode(const float* lang_emb, int m, int slot) override {
        if (!bound) { hulc_set_error("hulc_clip_gt_encode before hulc_bind_params"); return 1; }
        if (!cfg.use_clip || cfg.kind != HULC_KIND_HULC) { hulc_set_error("hulc_clip_gt_encode: the context has no CLIP head (use_clip_auxiliary_loss, hulc.py:703)"); return 1; }
        if (have_fwd) { hulc_set_error("hulc_clip_gt_encode between a forward and its backward"); return 1; }
        if (slot < 0 || slot > 1 || m < 1 || !lang_emb) { hulc_set_error("hulc_clip_gt_encode: slot %d, m %d", slot, m); return 1; }
        int64_t capi = gt_cap[slot];
        if (!gt_grow(gt_txt[slot], capi, (int64_t)m * GOAL) || !gt_grow(gt_in, gt_in_cap, (int64_t)m * LANG)) { hulc_set_error("hulc_clip_gt_encode: allocation failed"); return 1; }
        gt_cap[slot] = (int)capi;
        HIP_CHECK(hipMemcpyAsync(gt_in, lang_emb, sizeof(float) * (int64_t)m * LANG, hipMemcpyDefault, st));
        T* acts[2] = {gl1, gl2};
        for (int r0 = 0; r0 < m; r0 += maxB) {
            const int rows = std::min(maxB, m - r0);
            hipLaunchKernelGGL((cast_kernel<float, T>), dim3(cdiv(rows * LANG, 256)), dim3(256), 0, st, gt_in + (int64_t)r0 * LANG, lang_t, (long long)rows * LANG);
            mlp_fwd(lang_t, LANG, rows, lg, 3, acts, gl3, nullptr);
            ln_fwd(gl3, GOAL, rows, GOAL, ln_lg_g, ln_lg_b, goal_t, GOAL, nullptr, 0, goal_st);
            { EpiP ep = epi(la1, false); ep.relu = 1; lin_fwd(goal_t, GOAL, rows, cl_la0, ep, 128); }
            { EpiP ep = epi(gt_txt[slot] + (int64_t)r0 * GOAL, true); lin_fwd(la1, 128, rows, cl_la2, ep, GOAL); }
        }
        gt_m[slot] = m;
        if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in hulc_clip_gt_encode"); return 1; }
        HIP_CHECK(hipStreamSynchronize(st));          // lang_emb may be a host buffer the caller frees
        return 0;
    }
    int clip_gt_scores(int slot, float* out_host, int64_t cap, int32_t* n_out, int32_t* m_out) override {
        if (slot < 0 || slot > 1 || gt_m[slot] < 1) { hulc_set_error("hulc_clip_gt_scores: slot %d holds no encoded instructions", slot); return 1; }
        if (val_clip_n < 1) { hulc_set_error("hulc_clip_gt_scores: the last hulc_validate had no masked lang rows (hulc.py:988-989 returns early)"); return 1; }
        const int n = val_clip_n, m = gt_m[slot];
        if (n_out) *n_out = n;
        if (m_out) *m_out = m;
        if (!out_host && cap == 0) return 0;                                   // shape query
        if (!out_host || cap < (int64_t)n * m) { hulc_set_error("hulc_clip_gt_scores: buffer of %lld floats, need %lld", (long long)cap, (long long)n * m); return 1; }
        if (!gt_grow(gt_out, gt_out_cap, (int64_t)n * m)) { hulc_set_error("hulc_clip_gt_scores: allocation failed"); return 1; }
        hipLaunchKernelGGL(clip_gt_scores_kernel, dim3(cdiv(m, 64), n), dim3(64), 0, st, img, gt_txt[slot], n, m, GOAL, logit_scale, gt_out);
        if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in hulc_clip_gt_scores"); return 1; }
        HIP_CHECK(hipMemcpyAsync(out_host, gt_out, sizeof(float) * (int64_t)n * m, hipMemcpyDefault, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    }

    // ---------------------------------------------------------------- rollout (hulc.py:843-957), B = 1
    int* roll_plan = nullptr; T* roll_plan_c = nullptr; T *roll_goal = nullptr, *roll_h0 = nullptr, *roll_h1 = nullptr;
    float *roll_fs = nullptr, *roll_fg = nullptr, *roll_ro = nullptr, *roll_pred = nullptr, *roll_pred_goal = nullptr;
    bool roll_has_h = false, roll_has_plan = false;
    uint64_t roll_counter = 0;
    void roll_alloc() {
        if (roll_plan) return;
        val_alloc();
        roll_plan = alloc<int>(NCAT); roll_plan_c = alloc<T>(PLAN); roll_goal = alloc<T>(GOAL); roll_h0 = alloc<T>(HID); roll_h1 = alloc<T>(HID);
        roll_fs = alloc<float>(2ll * 3 * encS.IH * encS.IH); roll_fg = alloc<float>(2ll * 3 * encG.IH * encG.IH);
        roll_ro = alloc<float>(16); roll_pred = alloc<float>(8); roll_pred_goal = alloc<float>(GOAL);
    }
    int rollout_reset() override { roll_has_h = false; roll_has_plan = false; roll_counter = 0; return 0; }
    int rollout_plan(const hulc_rollout_obs* obs, const float* goal_static, const float* goal_gripper, const float* goal_lang, const int32_t* plan_inject,
                     int32_t* plan_out) override {
        if (!bound) { hulc_set_error("hulc_rollout_plan before hulc_bind_params"); return 1; }
        const bool gcbc = cfg.kind == HULC_KIND_GCBC;      // gcbc.py:286-320: only the latent goal is encoded (once per rollout), the decoder acts without a plan
        if ((goal_lang != nullptr) == (goal_static != nullptr && goal_gripper != nullptr)) {
            hulc_set_error("hulc_rollout_plan: give either the two goal images or the language embedding");
            return 1;
        }
        if (maxS < 2 && !goal_lang) { hulc_set_error("hulc_rollout_plan: a visual goal needs max_seq >= 2 (obs + goal frame form one window, hulc.py:917-919)"); return 1; }
        roll_alloc();
        if (alloc_failed) { hulc_set_error("hulc_rollout_plan: workspace allocation failed"); return 1; }
        have_fwd = false; pair = false;
        hulc_batch bb; memset(&bb, 0, sizeof(bb));
        bb.B = 1; bb.step = roll_counter;
        if (goal_lang) {
            bb.S = 1; bb.is_lang = 1; bb.rgb_static = obs->rgb_static; bb.rgb_gripper = obs->rgb_gripper; bb.lang = goal_lang;
        } else {
            const size_t ns = sizeof(float) * 3 * encS.IH * encS.IH, ng = sizeof(float) * 3 * encG.IH * encG.IH;
            HIP_CHECK(hipMemcpyAsync(roll_fs, obs->rgb_static, ns, hipMemcpyDefault, st));
            HIP_CHECK(hipMemcpyAsync((char*)roll_fs + ns, goal_static, ns, hipMemcpyDefault, st));
            HIP_CHECK(hipMemcpyAsync(roll_fg, obs->rgb_gripper, ng, hipMemcpyDefault, st));
            HIP_CHECK(hipMemcpyAsync((char*)roll_fg + ng, goal_gripper, ng, hipMemcpyDefault, st));
            bb.S = 2; bb.is_lang = 0; bb.rgb_static = roll_fs; bb.rgb_gripper = roll_fg;
        }
        cur = bb;
        trunk_fwd(&bb, 0.f);
        const int* inj = nullptr;
        if (mcil) {     // continuous plan (256) fp32: sampled from the proposal Normal or injected; kept in roll_plan_c for the following act() calls
            sample_cont(pp_logits, nullptr, reinterpret_cast<const float*>(plan_inject), 1, site_seed(50));
            HIP_CHECK(hipMemcpyAsync(roll_plan_c, plan_t, sizeof(T) * (PLAN / 2), hipMemcpyDeviceToDevice, st));
            if (plan_out) HIP_CHECK(hipMemcpyAsync(plan_out, plan_f, sizeof(float) * (PLAN / 2), hipMemcpyDefault, st));
        } else if (!gcbc) {
        if (plan_inject) { HIP_CHECK(hipMemcpyAsync(pidx_in, plan_inject, sizeof(int) * NCAT, hipMemcpyDefault, st)); inj = pidx_in; }
        hipLaunchKernelGGL(plan_kl_sample_kernel, dim3(NCAT), dim3(64), 0, st, pp_logits, (const float*)nullptr, 1, NCAT, NCLS, inj, roll_plan, probs, klcat, dpp_kl,
                           dpr_kl, 0.f, 0.f, site_seed(50));
        }
        HIP_CHECK(hipMemcpyAsync(roll_goal, goal_t, sizeof(T) * GOAL, hipMemcpyDeviceToDevice, st));
        if (!gcbc) roll_has_h = false;                 // action_decoder.clear_hidden_state() (hulc.py:925 / :946); GCBC.step never clears it (gcbc.py:286-320)
        roll_has_plan = true;
        if (plan_out && !mcil && !gcbc) HIP_CHECK(hipMemcpyAsync(plan_out, roll_plan, sizeof(int) * NCAT, hipMemcpyDefault, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in rollout_plan"); return 1; }
        return 0;
    }
    // the latent goal / plan of a rollout as VALUES (hulc.py:881-948: predict_with_plan / get_pp_plan_* take and return them)
    int rollout_get_goal(float* latent_goal_out) override {
        if (!roll_has_plan) { hulc_set_error("hulc_rollout_get_goal before hulc_rollout_plan"); return 1; }
        hipLaunchKernelGGL((cast_kernel<T, float>), dim3(1), dim3(64), 0, st, roll_goal, roll_pred_goal, (long long)GOAL);
        HIP_CHECK(hipMemcpyAsync(latent_goal_out, roll_pred_goal, sizeof(float) * GOAL, hipMemcpyDefault, st));
        HIP_CHECK(hipStreamSynchronize(st));
        return 0;
    }
    int rollout_set_state(const void* plan, const float* latent_goal) override {
        if (!bound) { hulc_set_error("hulc_rollout_set_state before hulc_bind_params"); return 1; }
        const bool gcbc = cfg.kind == HULC_KIND_GCBC;
        if (!gcbc && !plan) { hulc_set_error("hulc_rollout_set_state: null plan"); return 1; }
        roll_alloc();
        if (alloc_failed) { hulc_set_error("hulc_rollout_set_state: workspace allocation failed"); return 1; }
        HIP_CHECK(hipMemcpyAsync(roll_pred_goal, latent_goal, sizeof(float) * GOAL, hipMemcpyDefault, st));
        hipLaunchKernelGGL((cast_kernel<float, T>), dim3(1), dim3(64), 0, st, roll_pred_goal, roll_goal, (long long)GOAL);
        if (mcil) {
            HIP_CHECK(hipMemcpyAsync(plan_f, plan, sizeof(float) * (PLAN / 2), hipMemcpyDefault, st));
            hipLaunchKernelGGL((cast_kernel<float, T>), dim3(1), dim3(256), 0, st, plan_f, roll_plan_c, (long long)(PLAN / 2));
        } else if (!gcbc) {
            HIP_CHECK(hipMemcpyAsync(roll_plan, plan, sizeof(int) * NCAT, hipMemcpyDefault, st));
            // a caller-supplied class index is a column of the plan embedding gather in dec_fwd: keep it inside [0, NCLS)
            hipLaunchKernelGGL(clamp_index_kernel, dim3(1), dim3(64), 0, st, roll_plan, NCAT, NCLS);
        }
        HIP_CHECK(hipStreamSynchronize(st));
        roll_has_plan = true;
        return 0;
    }
    int rollout_act(const hulc_rollout_obs* obs, const float* u_mix, const float* u_act, float* action_out) override {
        if (!roll_has_plan) { hulc_set_error("hulc_rollout_act before hulc_rollout_plan (Hulc.step replans at rollout_step_counter %% replan_freq == 0)"); return 1; }
        have_fwd = false;
        hulc_batch bb; memset(&bb, 0, sizeof(bb));
        bb.B = 1; bb.S = 1; bb.step = roll_counter++;
        cur = bb;
        enc_fwd(encS, aS, Conv1Src{obs->rgb_static, nullptr, 0, 0}, 1, 0);
        enc_fwd(encG, aG, Conv1Src{obs->rgb_gripper, nullptr, 0, 0}, 1, 64);
        HIP_CHECK(hipMemcpyAsync(goal_t, roll_goal, sizeof(T) * GOAL, hipMemcpyDeviceToDevice, st));
        if (mcil) HIP_CHECK(hipMemcpyAsync(plan_t, roll_plan_c, sizeof(T) * (PLAN / 2), hipMemcpyDeviceToDevice, st));
        dec_fwd(roll_plan, 1, 1, roll_has_h ? roll_h0 : nullptr, roll_has_h ? roll_h1 : nullptr);
        HIP_CHECK(hipMemcpyAsync(roll_h0, H0, sizeof(T) * HID, hipMemcpyDeviceToDevice, st));
        HIP_CHECK(hipMemcpyAsync(roll_h1, H1, sizeof(T) * HID, hipMemcpyDeviceToDevice, st));
        roll_has_h = true;
        HIP_CHECK(hipMemcpyAsync(roll_ro, obs->robot_obs_raw, sizeof(float) * 15, hipMemcpyDefault, st));
        if (u_mix) { HIP_CHECK(hipMemcpyAsync(nz_mix, u_mix, sizeof(float) * NDIM * NMIX, hipMemcpyDefault, st)); u_mix = nz_mix; }
        if (u_act) { HIP_CHECK(hipMemcpyAsync(nz_act, u_act, sizeof(float) * NDIM, hipMemcpyDefault, st)); u_act = nz_act; }
        hipLaunchKernelGGL(logistic_sample_kernel, dim3(1), dim3(64), 0, st, heads, NHEAD, roll_ro, (const float*)nullptr, u_mix, u_act, 1, 1, NMIX, NDIM,
                           cfg.log_scale_min, mcil ? 0 : 1, site_seed(51), roll_pred, (float*)nullptr, mcil ? 0 : 1);
        HIP_CHECK(hipMemcpyAsync(action_out, roll_pred, sizeof(float) * 7, hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in rollout_act"); return 1; }
        return 0;
    }

    // ---- a whole recurrence X[q_s] = f(X[q_{s-1}] Wm^T, aux[q_s]), s = 1..S-1, as ONE persistent launch (rnn_persist.h; 16-bit engines, 2048-wide
    // state).  false = not taken (fp32 engine, option off, shape not covered, a gradient collective in flight, or this device failed an earlier
    // launch): the caller runs one launch per step.  The FIRST launch of a context is followed by a stream synchronisation and a look at the
    // error word.  A timeout in a later launch (the kernel's polls are bounded) never reaches the weights and never fails a step (round 4):
    //   * the failing kernel also stores the tag of the optimizer step it belongs to into a DEVICE word; adam_kernel / sgd_kernel compare it
    //     with their own tag and return without touching p / m / v (the step is dropped, exactly like a GradScaler-skipped step);
    //   * at the API calls that end in a stream synchronisation (forward with losses read back, validate) the host sees the error word right
    //     there, switches the context to one launch per step and RUNS THE CALL AGAIN — its results are valid;
    //   * elsewhere (backward) the host notices at the next API call: persistent mode goes off, a warning is printed, `persistent_rnn_fallbacks`
    //     counts it (hulc_get_option).
    unsigned* rp_flags = nullptr;
    volatile unsigned* rp_err_host = nullptr;
    unsigned* rp_err_dev = nullptr;
    unsigned* rp_skip = nullptr;          // device word: tag of the optimizer step whose recurrence failed
    unsigned opt_seq = 0;                 // optimizer calls so far; recurrences launched now belong to step opt_seq + 1
    bool bwd_since_opt = false;           // a backward has accumulated into G since the last optimizer step / zero_grads
    long long rp_fallbacks = 0;
    unsigned rp_launches = 1;
    int rp_B = -1;
    bool rp_probed = false, rp_ok = false;
    // persistent launches need all 256 workgroups co-resident.  While a bucket of hulc_backward_allreduce is in flight RCCL's kernels hold CUs on
    // the high-priority collectives' stream, so a recurrence that follows an issued bucket (mcil: the plan encoder's BiRNN backward runs after the
    // decoder bucket has left) takes the launch-per-step path — `persist_under_comm` = 1 lifts that (measure on the target box first)
    bool comm_in_flight() const { return ar_dtype >= 0 && ar_sent != 0 && !persist_under_comm; }
    bool persist_usable(int B, int S) const {
        return std::is_same<T, h16_t>::value && persist_mode && HID == RP_HID && S >= 3 && B <= 16 * RP_NG && !(rp_probed && !rp_ok) && !comm_in_flight();
    }
    // X2 != null: a second, independent recurrence of the same shape in the same launch (rnn_persist.h: dual) — 4 XCDs each, B <= 64
    bool rnn_persist(T* X, const T* Wm, const T* res, const T* mask, int B, int S, int q0, int dq, int act,
                     T* X2 = nullptr, const T* Wm2 = nullptr, const T* res2 = nullptr, const T* mask2 = nullptr, int q02 = 0, int dq2 = 0) {
        if constexpr (!std::is_same<T, h16_t>::value) return false;
        else {
            if (!persist_usable(B, S)) return false;
            if (X2 && B > 16 * (RP_NG / 2)) return false;
            if (!rp_flags) {
                rp_flags = alloc<unsigned>(RP_FLAG_WORDS);
                if (!rp_skip) rp_skip = alloc<unsigned>(64);
                void* h = nullptr;
                if (alloc_failed || hipHostMalloc(&h, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&rp_err_dev, h, 0) != hipSuccess) { rp_probed = true; rp_ok = false; return false; }
                rp_err_host = (volatile unsigned*)h; *rp_err_host = 0;
            }
            const int bkey = X2 ? -B : B;                      // a dual launch uses another window -> XCD assignment
            if (bkey != rp_B || rp_launches >= (1u << 19)) {   // another set of active groups, or the step counters near their wrap: restart the counters from a clean slate
                hipMemsetAsync(rp_flags, 0, sizeof(unsigned) * RP_FLAG_WORDS, st);
                rp_launches = 1; rp_B = bkey;
            }
            RnnPersistP p{};
            p.X = X; p.W = Wm; p.res = res; p.mask = mask; p.B = B; p.S = S; p.q0 = q0; p.dq = dq; p.act = act;
            if (X2) { p.dual = 1; p.X2 = X2; p.W2 = Wm2; p.res2 = res2; p.mask2 = mask2; p.q02 = q02; p.dq2 = dq2; }
            p.flags = rp_flags; p.base = rp_launches << 12; p.parity = (int)(rp_launches & 1u); p.err = rp_err_dev; p.stamps = nullptr;
            p.skip = rp_skip; p.skip_tag = opt_seq + 1;
            if (rp_probed && persist_fault > 0) { p.fault = 1; --persist_fault; }
            ++rp_launches;
            TimerScope ts(this, "rnn_persist", "mfma", (X2 ? 2.0 : 1.0) * 2.0 * B * HID * HID * (S - 1), (X2 ? 2.0 : 1.0) * ((double)HID * HID * sizeof(T) + 3.0 * S * B * HID * sizeof(T)), 1);
            if (!launch_rnn_persist(st, p)) return false;
            if (!rp_probed) {
                hipStreamSynchronize(st);
                rp_probed = true; rp_ok = *rp_err_host == 0;
                if (!rp_ok) {
                    fprintf(stderr, "hulc: persistent recurrence unavailable on this device (census / co-residency check failed, code %u): one launch per time step\n", *rp_err_host);
                    *rp_err_host = 0;
                    return false;
                }
            }
            return true;
        }
    }
    // true if a persistent recurrence timed out since the last check.  The context then runs one launch per step from here on.
    // synced: the stream is drained and the caller is about to redo its own work (the skip tag is cleared: the redo makes the step whole again);
    // otherwise the failed launch belonged to an earlier, asynchronous call — its optimizer step skips itself on the device.
    bool persist_check(const char* where, bool synced) {
        if (!rp_err_host || *rp_err_host == 0) return false;
        const unsigned code = *rp_err_host;
        *rp_err_host = 0; rp_ok = false; rp_probed = true; ++rp_fallbacks;
        fprintf(stderr, "hulc: %s: a persistent recurrence launch timed out (code %u: the GPU's CUs were not all available — shared with another process or a "
                        "collective?).  %s; persistent_rnn is now off for this context (one launch per time step).\n", where, code,
                synced ? (bwd_since_opt ? "The call is run again on the launch-per-step path; a backward of this optimizer step ran before it, so the step is skipped on the device"
                                        : "The call is run again on the launch-per-step path")
                       : "The optimizer step it belonged to is skipped on the device (weights untouched)");
        // synced = the caller is about to run its own forward again.  That makes the step whole only if the failed launch was the caller's: the
        // timeout may also belong to an earlier, still asynchronous BACKWARD of the same optimizer step (fwd(vis), bwd(vis), fwd(lang): the check
        // at the entry of the second forward cannot see a backward that is still running) — its garbage is already accumulated in G, so the tag
        // stays and the step is dropped.  Only a step that has no backward behind it yet is cleared.
        if (synced && rp_skip && !bwd_since_opt) hipMemset(rp_skip, 0, sizeof(unsigned));
        return true;
    }
    int get_option(const char* name, long long* value) override {
        if (name && !strcmp(name, "persistent_rnn")) { *value = persist_mode && !(rp_probed && !rp_ok); return 0; }
        if (name && !strcmp(name, "persistent_rnn_fallbacks")) { *value = rp_fallbacks; return 0; }
        if (name && !strcmp(name, "fused_transformer")) { *value = tr_fused_mode; return 0; }
        if (name && !strcmp(name, "persist_under_comm")) { *value = persist_under_comm; return 0; }
        if (name && !strcmp(name, "comm_timing")) { *value = comm_timing; return 0; }
        hulc_set_error("hulc_get_option: unknown option '%s'", name ? name : "(null)");
        return 1;
    }

    // H[t] = act(Zx[t] + H[t-1] Whh^T), time-major [S][B][HID].  act 1: ReLU (action decoder), 2: tanh (mcil BiRNN); rev: the
    // recurrence runs from t = S-1 down to 0 (nn.RNN's reverse direction, outputs stay at their own positions)
    // first_done: H[0] = act(Zx[0]) was already written by the GEMM that produced Zx (EpiP::out2)
    void rnn_fwd(const T* Zx, T* H, const LinW& whh, int B, int S, const T* h0 = nullptr, int act = 1, bool rev = false, bool first_done = false) {
        const long long BH = (long long)B * HID;
        auto at = [&](int i) { return (long long)(rev ? S - 1 - i : i) * BH; };
        if (h0) {
            EpiP ep = epi(H + at(0), false); ep.res = Zx + at(0); ep.res_ld = HID; ep.relu = act;
            gemm(dense<T>(h0, B, HID), dense<T>(whh.W, HID, HID), dense_out(HID), ep, B, HID, HID);
        } else if (!first_done) hipLaunchKernelGGL((relu_copy_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, Zx + at(0), H + at(0), BH, act);
        if (rnn_persist(H, whh.W, Zx, nullptr, B, S, rev ? S - 1 : 0, rev ? -1 : 1, act)) return;
        TimerScope ts(this, "rnn_step_gemm", "hbm", 2.0 * B * HID * HID * (S - 1), ((double)HID * HID + 3.0 * B * HID) * sizeof(T) * (S - 1), S - 1);
        for (int i = 1; i < S; ++i) {
            EpiP ep = epi(H + at(i), false); ep.res = Zx + at(i); ep.res_ld = HID; ep.relu = act;
            gemm(dense<T>(H + at(i - 1), B, HID), dense<T>(whh.W, HID, HID), dense_out(HID), ep, B, HID, HID);
        }
    }
    // dZ[t] = (dH[t] + dZ[t+1] Whh) * act'(H[t]).  dH_last_only: dH is [B][HID], the gradient of the LAST processed state alone.
    // last_done: dZ[S-1] = dH[S-1] * act'(H[S-1]) was already written by the GEMM that produced dH (EpiP::out2)
    void rnn_bwd(const T* dH, const T* H, T* dZ, const LinW& whh, int B, int S, int act = 1, bool rev = false, bool dH_last_only = false, bool last_done = false) {
        const long long BH = (long long)B * HID;
        auto at = [&](int i) { return (long long)(rev ? S - 1 - i : i) * BH; };
        if (!last_done) hipLaunchKernelGGL((mask_mul_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, dH_last_only ? dH : dH + at(S - 1), H + at(S - 1), dZ + at(S - 1), BH, act);
        if (rnn_persist(dZ, whh.Wt, dH_last_only ? nullptr : dH, H, B, S, rev ? 0 : S - 1, rev ? 1 : -1, act)) return;
        TimerScope ts(this, "rnn_step_gemm", "hbm", 2.0 * B * HID * HID * (S - 1), ((double)HID * HID + 4.0 * B * HID) * sizeof(T) * (S - 1), S - 1);
        for (int i = S - 2; i >= 0; --i) {
            EpiP ep = epi(dZ + at(i), false); ep.mask = H + at(i); ep.mask_tanh = act == 2;
            if (!dH_last_only) { ep.res = dH + at(i); ep.res_ld = HID; }
            gemm(dense<T>(dZ + at(i + 1), B, HID), dense<T>(whh.Wt, HID, HID), dense_out(HID), ep, B, HID, HID);
        }
    }

    // The two directions of a bidirectional layer are independent chains of S dependent launches each: advanced in lockstep, one launch (grid.z = 2)
    // carries step i of both — the same work per launch boundary paid once instead of twice (gemm.h: Skinny2).  Direction 0 runs t = 0..S-1,
    // direction 1 (reverse) t = S-1..0.  Shapes the dual launch does not cover fall back to the two sequential recurrences.
    static bool pair_dirs() { static const bool on = HULC_SWITCH("HULC_PAIR_DIRS", 1) != 0; return on; }
    void rnn_fwd2(T* const Zx[2], T* const H[2], const LinW* const whh[2], int B, int S, int act) {
        const long long BH = (long long)B * HID;
        bool dual = false;
        if constexpr (std::is_same<T, h16_t>::value) dual = pair_dirs() && S > 1 && !persist_usable(B, S);     // two persistent launches (2 x ~90 us at S = 32) beat 31 paired ones
        if (dual) {
            auto at = [&](int d, int i) { return (long long)(d ? S - 1 - i : i) * BH; };
            for (int d = 0; d < 2; ++d) hipLaunchKernelGGL((relu_copy_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, Zx[d] + at(d, 0), H[d] + at(d, 0), BH, act);
            TimerScope ts(this, "rnn_step_gemm", "hbm", 4.0 * B * HID * HID * (S - 1), 2 * ((double)HID * HID + 3.0 * B * HID) * sizeof(T) * (S - 1), S - 1);
            for (int i = 1; i < S; ++i) {
                EpiP ep[2];
                for (int d = 0; d < 2; ++d) { ep[d] = epi(H[d] + at(d, i), false); ep[d].bias = nullptr; ep[d].res = Zx[d] + at(d, i); ep[d].res_ld = HID; ep[d].relu = act; }
                if constexpr (std::is_same<T, h16_t>::value) {
                    if (launch_skinny_lds_dual(st, H[0] + at(0, i - 1), whh[0]->W, ep[0], H[1] + at(1, i - 1), whh[1]->W, ep[1], HID, HID, B, HID, HID, dense_out(HID))) continue;
                }
                for (int d = 0; d < 2; ++d) gemm(dense<T>(H[d] + at(d, i - 1), B, HID), dense<T>(whh[d]->W, HID, HID), dense_out(HID), ep[d], B, HID, HID);
            }
            return;
        }
        if constexpr (std::is_same<T, h16_t>::value) {
            // both directions as ONE persistent launch, four XCDs each (rnn_persist.h: dual)
            static const bool dualp = HULC_SWITCH("HULC_PERSIST_DUAL", 1) != 0;
            if (dualp && persist_usable(B, S) && B <= 16 * (RP_NG / 2)) {
                auto at = [&](int d, int i) { return (long long)(d ? S - 1 - i : i) * BH; };
                for (int d = 0; d < 2; ++d) hipLaunchKernelGGL((relu_copy_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, Zx[d] + at(d, 0), H[d] + at(d, 0), BH, act);
                if (rnn_persist(H[0], whh[0]->W, Zx[0], nullptr, B, S, 0, 1, act, H[1], whh[1]->W, Zx[1], nullptr, S - 1, -1)) return;
                for (int d = 0; d < 2; ++d) rnn_fwd(Zx[d], H[d], *whh[d], B, S, nullptr, act, d == 1, true);      // not taken: sequential chains (first step done)
                return;
            }
        }
        for (int d = 0; d < 2; ++d) rnn_fwd(Zx[d], H[d], *whh[d], B, S, nullptr, act, d == 1);
    }
    void rnn_bwd2(T* const dH[2], T* const H[2], T* const dZ[2], const LinW* const whh[2], int B, int S, int act) {
        const long long BH = (long long)B * HID;
        bool dual = false;
        if constexpr (std::is_same<T, h16_t>::value) dual = pair_dirs() && S > 1 && !persist_usable(B, S);
        if (dual) {
            auto at = [&](int d, int i) { return (long long)(d ? S - 1 - i : i) * BH; };
            for (int d = 0; d < 2; ++d)
                hipLaunchKernelGGL((mask_mul_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, dH[d] + at(d, S - 1), H[d] + at(d, S - 1), dZ[d] + at(d, S - 1), BH, act);
            TimerScope ts(this, "rnn_step_gemm", "hbm", 4.0 * B * HID * HID * (S - 1), 2 * ((double)HID * HID + 4.0 * B * HID) * sizeof(T) * (S - 1), S - 1);
            for (int i = S - 2; i >= 0; --i) {
                EpiP ep[2];
                for (int d = 0; d < 2; ++d) {
                    ep[d] = epi(dZ[d] + at(d, i), false); ep[d].mask = H[d] + at(d, i); ep[d].mask_tanh = act == 2; ep[d].res = dH[d] + at(d, i); ep[d].res_ld = HID;
                }
                if constexpr (std::is_same<T, h16_t>::value) {
                    if (launch_skinny_lds_dual(st, dZ[0] + at(0, i + 1), whh[0]->Wt, ep[0], dZ[1] + at(1, i + 1), whh[1]->Wt, ep[1], HID, HID, B, HID, HID, dense_out(HID))) continue;
                }
                for (int d = 0; d < 2; ++d) gemm(dense<T>(dZ[d] + at(d, i + 1), B, HID), dense<T>(whh[d]->Wt, HID, HID), dense_out(HID), ep[d], B, HID, HID);
            }
            return;
        }
        if constexpr (std::is_same<T, h16_t>::value) {
            static const bool dualp = HULC_SWITCH("HULC_PERSIST_DUAL", 1) != 0;
            if (dualp && persist_usable(B, S) && B <= 16 * (RP_NG / 2)) {
                auto at = [&](int d, int i) { return (long long)(d ? S - 1 - i : i) * BH; };
                for (int d = 0; d < 2; ++d)
                    hipLaunchKernelGGL((mask_mul_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, dH[d] + at(d, S - 1), H[d] + at(d, S - 1), dZ[d] + at(d, S - 1), BH, act);
                if (rnn_persist(dZ[0], whh[0]->Wt, dH[0], H[0], B, S, S - 1, -1, act, dZ[1], whh[1]->Wt, dH[1], H[1], 0, 1)) return;
                for (int d = 0; d < 2; ++d) rnn_bwd(dH[d], H[d], dZ[d], *whh[d], B, S, act, d == 1, false, true);
                return;
            }
        }
        for (int d = 0; d < 2; ++d) rnn_bwd(dH[d], H[d], dZ[d], *whh[d], B, S, act, d == 1, false);
    }

    // ---------------------------------------------------------------- mcil plan recognition (plan_recognition_net.py:14-42)
    // nn.RNN(tanh, 2 layers, bidirectional) over the time-major embedding; x = output[:, -1] = [fwd state after the last step |
    // reverse state at the last position (its FIRST step, so the layer-1 reverse recurrence never runs)]; pr_state = fc_state(x).
    void birnn_fwd(int B, int S) {
        const int SB = S * B;
        const long long BH = (long long)B * HID;
        hipLaunchKernelGGL((gather_embg_kernel<T>), dim3(cdiv(SB * EMB, 256)), dim3(256), 0, st, emb, embg, B, S, EMB);
        for (int d = 0; d < 2; ++d) {
            EpiP ep = epi(bZ0[d], false); ep.bias = bb_ih[0][d]; ep.bias2 = bb_hh[0][d];
            gemm(dense<T>(embg, SB, EMB), dense<T>(bw_ih[0][d].W, HID, EMB), dense_out(HID), ep, SB, HID, EMB);
        }
        { const LinW* w2[2] = {&bw_hh[0][0], &bw_hh[0][1]}; rnn_fwd2(bZ0, bH0, w2, B, S, 2); }
        // layer 1 forward direction: input [H0f | H0b] -> two K = 2048 GEMMs against the column halves of W_ih_l1
        { EpiP ep = epi(bZ1, false); ep.bias = bb_ih[1][0]; ep.bias2 = bb_hh[1][0];
          gemm(dense<T>(bH0[0], SB, HID), dense<T>(bw_ih[1][0].W, HID, 2 * HID), dense_out(HID), ep, SB, HID, HID); }
        { EpiP ep = epi(bZ1, false); ep.res = bZ1; ep.res_ld = HID;
          gemm(dense<T>(bH0[1], SB, HID), dense<T>(bw_ih[1][0].W + HID, HID, 2 * HID), dense_out(HID), ep, SB, HID, HID); }
        rnn_fwd(bZ1, bH1, bw_hh[1][0], B, S, nullptr, 2, false);
        // layer 1 reverse direction at position S-1 only (h_prev = 0): tanh(W_ih [H0f|H0b][S-1] + b)
        { EpiP ep = epi(bh1b, false); ep.bias = bb_ih[1][1]; ep.bias2 = bb_hh[1][1];
          gemm(dense<T>(bH0[0] + (S - 1) * BH, B, HID), dense<T>(bw_ih[1][1].W, HID, 2 * HID), dense_out(HID), ep, B, HID, HID); }
        { EpiP ep = epi(bh1b, false); ep.res = bh1b; ep.res_ld = HID; ep.relu = 2;
          gemm(dense<T>(bH0[1] + (S - 1) * BH, B, HID), dense<T>(bw_ih[1][1].W + HID, HID, 2 * HID), dense_out(HID), ep, B, HID, HID); }
        copy2d<T, T>(bH1 + (S - 1) * BH, HID, bxcat, 2 * HID, B, HID, 0);
        copy2d<T, T>(bh1b, HID, bxcat + HID, 2 * HID, B, HID, 0);
        { EpiP ep = epi(pr_logits, true); lin_fwd(bxcat, 2 * HID, B, pr_fs, ep, PLAN); }
        STAGE("birnn_fwd");
    }
    // ---------------------------------------------------------------- the same with rnn_type = nn.GRU (BASELINE config 4)
    // one direction of one layer: Zx (incl. b_ih) [S][B][3H] -> H [S][B][H]; per step one M = B GEMM against W_hh (N = 3H) + the gate kernel
    void gru_recur_fwd(GruBuf& g, const LinW& whh, const float* bhh, int B, int S, bool rev) {
        const long long BH = (long long)B * HID;
        auto at = [&](int i) { return (long long)(rev ? S - 1 - i : i); };
        for (int i = 0; i < S; ++i) {
            const long long t = at(i);
            const T* hp = i ? g.H + at(i - 1) * BH : nullptr;
            if constexpr (std::is_same<T, h16_t>::value) {      // GEMM + gate arithmetic of the step in one launch (gemm.h: gru_step_lds_kernel)
                static const bool fused = HULC_SWITCH("HULC_GRU_FUSED", 1) != 0;
                if (i && fused) {
                    TimerScope ts(this, "gru_step", "hbm", 2.0 * B * 3 * HID * HID, ((double)3 * HID * HID + 9.0 * B * HID) * sizeof(T));
                    const GruStepP q{hp, whh.Wfr ? whh.Wfr : whh.W, g.Zx + t * 3 * BH, bhh, g.H + t * BH, g.R + t * BH, g.Z + t * BH, g.N + t * BH, g.GN + t * BH};
                    if (launch_gru_step(st, &q, 1, B, HID, whh.Wfr != nullptr)) continue;
                }
            }
            if (i) { EpiP ep = epi(gGf, true); ep.bias = bhh; gemm(dense<T>(hp, B, HID), dense<T>(whh.W, 3 * HID, HID), dense_out(3 * HID), ep, B, 3 * HID, HID); }
            hipLaunchKernelGGL((gru_gate_fwd_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, g.Zx + t * 3 * BH, i ? gGf : (const float*)nullptr, bhh, hp, B, HID,
                               g.H + t * BH, g.R + t * BH, g.Z + t * BH, g.N + t * BH, g.GN + t * BH);
        }
    }
    // BPTT of it: dH [S][B][H] (or, dH_last_only, the gradient [B][H] of the last processed state) -> g.dZx, g.dG
    void gru_recur_bwd(GruBuf& g, const T* dH, const LinW& whh, int B, int S, bool rev, bool dH_last_only) {
        const long long BH = (long long)B * HID;
        auto at = [&](int i) { return (long long)(rev ? S - 1 - i : i); };
        bool fused_prev = false;                 // the gate backward of this step already ran in the previous GEMM's epilogue
        for (int i = S - 1; i >= 0; --i) {
            const long long t = at(i);
            const T* dh = dH_last_only ? (i == S - 1 ? dH : nullptr) : dH + t * BH;
            if (!fused_prev)
            hipLaunchKernelGGL((gru_gate_bwd_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, dh, i == S - 1 ? (const T*)nullptr : gcarA, g.R + t * BH, g.Z + t * BH,
                               g.N + t * BH, g.GN + t * BH, i ? g.H + at(i - 1) * BH : (const T*)nullptr, B, HID, g.dZx + t * 3 * BH, g.dG + t * 3 * BH, gcarB);
            fused_prev = false;
            if (!i) break;
            if constexpr (std::is_same<T, h16_t>::value) {      // carry GEMM + the gate backward of step i-1 in one launch (gemm.h: GruBwdP)
                static const bool fused = HULC_SWITCH("HULC_GRU_FUSED_BWD", 1) != 0;
                if (fused && skinny_use_lds) {
                    const long long tp = at(i - 1);
                    GruBwdP gbp{};
                    gbp.dH = dH_last_only ? nullptr : dH + tp * BH;
                    gbp.R = g.R + tp * BH; gbp.Z = g.Z + tp * BH; gbp.Nn = g.N + tp * BH; gbp.GN = g.GN + tp * BH;
                    gbp.Hprev = i - 1 ? g.H + at(i - 2) * BH : nullptr;
                    gbp.dzx = g.dZx + tp * 3 * BH; gbp.dg = g.dG + tp * 3 * BH; gbp.direct = gcarB; gbp.direct_in = gcarB;     // each thread reads its 4 direct[t] values before it writes direct[t-1] over them
                    EpiP ep = epi(gcarA, false);
                    if (launch_skinny_lds_kchunk(st, g.dG + t * 3 * BH, 3 * HID, whh.Wtfr ? whh.Wtfr : whh.Wt, whh.Wtfr ? 0 : 3 * HID, B, HID, 3 * HID, dense_out(HID), ep, gbp)) { fused_prev = true; continue; }
                }
            }
            { EpiP ep = epi(gcarA, false); ep.res = gcarB; ep.res_ld = HID;
              gemm(dense<T>(g.dG + t * 3 * BH, B, 3 * HID), dense<T>(whh.Wt, HID, 3 * HID), dense_out(HID), ep, B, HID, 3 * HID); }
        }
    }
    // both directions of a BiGRU layer in lockstep (see rnn_fwd2): g[0] runs t = 0..S-1, g[1] t = S-1..0
    void gru_recur_fwd2(GruBuf* const g[2], const LinW* const whh[2], const float* const bhh[2], int B, int S) {
        const long long BH = (long long)B * HID;
        bool dual = false;
        if constexpr (std::is_same<T, h16_t>::value) dual = pair_dirs() && S > 1 && skinny_use_lds;
        if (dual) {
            auto at = [&](int d, int i) { return (long long)(d ? S - 1 - i : i); };
            for (int i = 0; i < S; ++i) {
                if constexpr (std::is_same<T, h16_t>::value) {
                    if (i) {
                        GruStepP q[2];
                        for (int d = 0; d < 2; ++d) {
                            const long long t = at(d, i);
                            q[d] = GruStepP{g[d]->H + at(d, i - 1) * BH, whh[d]->Wfr ? whh[d]->Wfr : whh[d]->W, g[d]->Zx + t * 3 * BH, bhh[d], g[d]->H + t * BH, g[d]->R + t * BH, g[d]->Z + t * BH,
                                            g[d]->N + t * BH, g[d]->GN + t * BH};
                        }
                        TimerScope ts(this, "gru_step", "hbm", 4.0 * B * 3 * HID * HID, 2 * ((double)3 * HID * HID + 9.0 * B * HID) * sizeof(T));
                        if (launch_gru_step(st, q, 2, B, HID, whh[0]->Wfr != nullptr)) continue;
                    }
                }
                for (int d = 0; d < 2; ++d) {
                    const long long t = at(d, i);
                    const T* hp = i ? g[d]->H + at(d, i - 1) * BH : nullptr;
                    if (i) { EpiP ep = epi(gGf, true); ep.bias = bhh[d]; gemm(dense<T>(hp, B, HID), dense<T>(whh[d]->W, 3 * HID, HID), dense_out(3 * HID), ep, B, 3 * HID, HID); }
                    hipLaunchKernelGGL((gru_gate_fwd_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, g[d]->Zx + t * 3 * BH, i ? gGf : (const float*)nullptr, bhh[d], hp, B, HID,
                                       g[d]->H + t * BH, g[d]->R + t * BH, g[d]->Z + t * BH, g[d]->N + t * BH, g[d]->GN + t * BH);
                }
            }
            return;
        }
        for (int d = 0; d < 2; ++d) gru_recur_fwd(*g[d], *whh[d], bhh[d], B, S, d == 1);
    }
    void gru_recur_bwd2(GruBuf* const g[2], T* const dH[2], const LinW* const whh[2], int B, int S) {
        const long long BH = (long long)B * HID;
        bool dual = false;
        if constexpr (std::is_same<T, h16_t>::value) dual = pair_dirs() && S > 1 && skinny_use_lds && B <= 64 && (3 * HID) % 2048 == 0 && 3 * HID > 2048;   // = what launch_skinny_lds_kchunk covers
        if constexpr (std::is_same<T, h16_t>::value) {
            if (dual) {
                if (!gcarB2) gcarB2 = alloc<T>((int64_t)maxB * HID);
                auto at = [&](int d, int i) { return (long long)(d ? S - 1 - i : i); };
                T* carry[2] = {gcarB, gcarB2};
                for (int d = 0; d < 2; ++d) {      // last processed step of each direction: no carry yet
                    const long long t = at(d, S - 1);
                    hipLaunchKernelGGL((gru_gate_bwd_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, dH[d] + t * BH, (const T*)nullptr, g[d]->R + t * BH, g[d]->Z + t * BH,
                                       g[d]->N + t * BH, g[d]->GN + t * BH, g[d]->H + at(d, S - 2) * BH, B, HID, g[d]->dZx + t * 3 * BH, g[d]->dG + t * 3 * BH, carry[d]);
                }
                bool ok = true;
                for (int i = S - 1; i >= 1 && ok; --i) {
                    GruBwdP gbp[2];
                    for (int d = 0; d < 2; ++d) {
                        const long long tp = at(d, i - 1);
                        gbp[d] = GruBwdP{};
                        gbp[d].dH = dH[d] + tp * BH;
                        gbp[d].R = g[d]->R + tp * BH; gbp[d].Z = g[d]->Z + tp * BH; gbp[d].Nn = g[d]->N + tp * BH; gbp[d].GN = g[d]->GN + tp * BH;
                        gbp[d].Hprev = i - 1 ? g[d]->H + at(d, i - 2) * BH : nullptr;
                        gbp[d].dzx = g[d]->dZx + tp * 3 * BH; gbp[d].dg = g[d]->dG + tp * 3 * BH; gbp[d].direct = carry[d]; gbp[d].direct_in = carry[d];
                    }
                    EpiP ep = epi(gcarA, false);
                    const bool fr = whh[0]->Wtfr != nullptr;
                    KChunk2 p2; p2.A = g[1]->dG + at(1, i) * 3 * BH; p2.W = fr ? whh[1]->Wtfr : whh[1]->Wt; p2.ep = ep; p2.gb = gbp[1];
                    ok = launch_skinny_lds_kchunk(st, g[0]->dG + at(0, i) * 3 * BH, 3 * HID, fr ? whh[0]->Wtfr : whh[0]->Wt, fr ? 0 : 3 * HID, B, HID, 3 * HID, dense_out(HID), ep, gbp[0], &p2);
                }
                if (ok) return;
                hulc_set_error("gru_recur_bwd2: dual launch rejected mid-chain");      // shapes are checked identically every step: cannot happen after the first
                return;
            }
        }
        for (int d = 0; d < 2; ++d) gru_recur_bwd(*g[d], dH[d], *whh[d], B, S, d == 1, false);
    }
    // weight / bias gradients of one recurrence from dZx, dG, its states H and its input X [S][B][K] (ldx), into dW_ih (+ column offset, lddw)
    void gru_param_grads(GruBuf& g, const LinW& wih, const LinW& whh, float* dbih, float* dbhh, int B, int S, bool rev) {
        const int SB = S * B, mp = ldpad(SB), H3 = 3 * HID;
        transpose_pair(g.dG, H3, tA, SB, H3, g.H, HID, tB, SB, HID, mp);
        if (S > 1) { EpiP ep = epi(whh.dW, true); ep.accumulate = 1;      // forward: dG[t] x H[t-1]; reverse: dG[t] x H[t+1]
          gemm(dense<T>(tA + (rev ? 0 : B), H3, mp), dense<T>(tB + (rev ? B : 0), HID, mp), dense_out(HID), ep, H3, HID, (S - 1) * B); }
        colsum(g.dG, H3, SB, H3, dbhh);
        colsum(g.dZx, H3, SB, H3, dbih);
        (void)wih;
    }
    void bigru_fwd(int B, int S) {
        const int SB = S * B, H3 = 3 * HID;
        const long long BH = (long long)B * HID;
        gb[0].H = bH0[0]; gb[1].H = bH0[1]; gb[2].H = bH1; gb[3].H = bh1b;
        hipLaunchKernelGGL((gather_embg_kernel<T>), dim3(cdiv(SB * EMB, 256)), dim3(256), 0, st, emb, embg, B, S, EMB);
        for (int d = 0; d < 2; ++d) {
            EpiP ep = epi(gb[d].Zx, false); ep.bias = bb_ih[0][d];
            gemm(dense<T>(embg, SB, EMB), dense<T>(bw_ih[0][d].W, H3, EMB), dense_out(H3), ep, SB, H3, EMB);
        }
        { GruBuf* g2[2] = {&gb[0], &gb[1]}; const LinW* w2[2] = {&bw_hh[0][0], &bw_hh[0][1]}; const float* b2[2] = {bb_hh[0][0], bb_hh[0][1]};
          gru_recur_fwd2(g2, w2, b2, B, S); }
        for (int d = 0; d < 2; ++d) {      // layer 1 input [H0f | H0b]: two K = 2048 GEMMs against the column halves of W_ih_l1; d = 1: reverse direction, t = S-1 only
            const int M = d ? B : SB;
            const long long off = d ? (S - 1) * BH : 0;
            GruBuf& g = gb[2 + d];
            { EpiP ep = epi(g.Zx, false); ep.bias = bb_ih[1][d];
              gemm(dense<T>(bH0[0] + off, M, HID), dense<T>(bw_ih[1][d].W, H3, 2 * HID), dense_out(H3), ep, M, H3, HID); }
            { EpiP ep = epi(g.Zx, false); ep.res = g.Zx; ep.res_ld = H3;
              gemm(dense<T>(bH0[1] + off, M, HID), dense<T>(bw_ih[1][d].W + HID, H3, 2 * HID), dense_out(H3), ep, M, H3, HID); }
            gru_recur_fwd(g, bw_hh[1][d], bb_hh[1][d], B, d ? 1 : S, false);
        }
        copy2d<T, T>(bH1 + (S - 1) * BH, HID, bxcat, 2 * HID, B, HID, 0);
        copy2d<T, T>(bh1b, HID, bxcat + HID, 2 * HID, B, HID, 0);
        { EpiP ep = epi(pr_logits, true); lin_fwd(bxcat, 2 * HID, B, pr_fs, ep, PLAN); }
        STAGE("bigru_fwd");
    }
    void bigru_bwd(const T* dpr, int B, int S) {
        const int SB = S * B, mp = ldpad(SB), H3 = 3 * HID;
        const long long BH = (long long)B * HID;
        lin_wgrad(dpr, bxcat, 2 * HID, B, PLAN, 2 * HID, pr_fs.dW, 2 * HID, pr_fs.db);
        { EpiP ep = epi(bdx, false); lin_dgrad(dpr, B, pr_fs, ep, dense_out(2 * HID)); }
        // ---- layer 1: reverse direction (its one evaluated step, h_prev = 0: weight_hh_l1_reverse gets no gradient, its bias does) and forward BPTT
        copy2d<T, T>(bdx + HID, 2 * HID, dt_a + BH, HID, B, HID, 0);
        gru_recur_bwd(gb[3], dt_a + BH, bw_hh[1][1], B, 1, false, true);
        lin_wgrad(gb[3].dZx, bH0[0] + (S - 1) * BH, HID, B, H3, HID, bw_ih[1][1].dW, 2 * HID, dbb_ih[1][1]);
        lin_wgrad(gb[3].dZx, bH0[1] + (S - 1) * BH, HID, B, H3, HID, bw_ih[1][1].dW + HID, 2 * HID, nullptr);
        colsum(gb[3].dG, H3, B, H3, dbb_hh[1][1]);
        copy2d<T, T>(bdx, 2 * HID, dt_a, HID, B, HID, 0);
        gru_recur_bwd(gb[2], dt_a, bw_hh[1][0], B, S, false, true);
        for (int d = 0; d < 2; ++d) {      // d (layer-0 outputs) = dZx1 W_ih_l1 (+ the reverse direction's step at t = S-1)
            { EpiP ep = epi(bdH0[d], false);
              gemm(dense<T>(gb[2].dZx, SB, H3), dense<T>(bw_ih[1][0].Wt + (long long)d * HID * H3, HID, H3), dense_out(HID), ep, SB, HID, H3); }
            { EpiP ep = epi(bdH0[d] + (S - 1) * BH, false); ep.res = bdH0[d] + (S - 1) * BH; ep.res_ld = HID;
              gemm(dense<T>(gb[3].dZx, B, H3), dense<T>(bw_ih[1][1].Wt + (long long)d * HID * H3, HID, H3), dense_out(HID), ep, B, HID, H3); }
        }
        gru_param_grads(gb[2], bw_ih[1][0], bw_hh[1][0], dbb_ih[1][0], dbb_hh[1][0], B, S, false);
        cast_tr<T, T>(gb[2].dZx, H3, nullptr, 0, tA, mp, SB, H3);
        for (int d = 0; d < 2; ++d) {
            cast_tr<T, T>(bH0[d], HID, nullptr, 0, tB, mp, SB, HID);
            EpiP ep = epi(bw_ih[1][0].dW + d * HID, true); ep.accumulate = 1;
            gemm(dense<T>(tA, H3, mp), dense<T>(tB, HID, mp), dense_out(2 * HID), ep, H3, HID, SB);
        }
        // ---- layer 0, both directions
        { GruBuf* g2[2] = {&gb[0], &gb[1]}; const LinW* w2[2] = {&bw_hh[0][0], &bw_hh[0][1]}; gru_recur_bwd2(g2, bdH0, w2, B, S); }
        for (int d = 0; d < 2; ++d) {
            gru_param_grads(gb[d], bw_ih[0][d], bw_hh[0][d], dbb_ih[0][d], dbb_hh[0][d], B, S, d == 1);
            transpose_pair(gb[d].dZx, H3, tA, SB, H3, embg, EMB, tB, SB, EMB, mp);
            { EpiP ep = epi(bw_ih[0][d].dW, true); ep.accumulate = 1; gemm(dense<T>(tA, H3, mp), dense<T>(tB, EMB, mp), dense_out(EMB), ep, H3, EMB, SB); }
            { EpiP ep = epi(demb, true); ep.accumulate = 1;
              gemm(dense<T>(gb[d].dZx, SB, H3), dense<T>(bw_ih[0][d].Wt, EMB, H3), dense_out_map(B, EMB, (long long)S * EMB), ep, SB, EMB, H3); }
        }
        STAGE("bigru_bwd");
    }
    // dpr: d pr_state (T, [B][PLAN]).  Accumulates the BiRNN / fc_state parameter gradients and adds d emb into demb (B,S,128).
    void birnn_bwd(const T* dpr, int B, int S) {
        const int SB = S * B, mp = ldpad(SB);
        const long long BH = (long long)B * HID;
        lin_wgrad(dpr, bxcat, 2 * HID, B, PLAN, 2 * HID, pr_fs.dW, 2 * HID, pr_fs.db);
        { EpiP ep = epi(bdx, false); lin_dgrad(dpr, B, pr_fs, ep, dense_out(2 * HID)); }
        // ---- layer 1, reverse direction: one step, no recurrence (weight_hh_l1_reverse receives no gradient)
        copy2d<T, T>(bdx + HID, 2 * HID, dt_a + BH, HID, B, HID, 0);
        hipLaunchKernelGGL((mask_mul_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, dt_a + BH, bh1b, bdz1b, BH, 2);
        lin_wgrad(bdz1b, bH0[0] + (S - 1) * BH, HID, B, HID, HID, bw_ih[1][1].dW, 2 * HID, dbb_ih[1][1], dbb_hh[1][1]);
        lin_wgrad(bdz1b, bH0[1] + (S - 1) * BH, HID, B, HID, HID, bw_ih[1][1].dW + HID, 2 * HID, nullptr);
        // ---- layer 1, forward direction: BPTT from the last state
        copy2d<T, T>(bdx, 2 * HID, dt_a, HID, B, HID, 0);       // its only consumer is x = output[:, -1]: dH = bdx[:, 0:HID] at t = S-1, zero elsewhere
        rnn_bwd(dt_a, bH1, bdZ1, bw_hh[1][0], B, S, 2, false, true);
        // d (layer-0 outputs) = dZ1 W_ih_l1 (+ the reverse direction's single step at t = S-1)
        for (int d = 0; d < 2; ++d) {
            { EpiP ep = epi(bdH0[d], false);
              gemm(dense<T>(bdZ1, SB, HID), dense<T>(bw_ih[1][0].Wt + (long long)d * HID * HID, HID, HID), dense_out(HID), ep, SB, HID, HID); }
            { EpiP ep = epi(bdH0[d] + (S - 1) * BH, false); ep.res = bdH0[d] + (S - 1) * BH; ep.res_ld = HID;
              gemm(dense<T>(bdz1b, B, HID), dense<T>(bw_ih[1][1].Wt + (long long)d * HID * HID, HID, HID), dense_out(HID), ep, B, HID, HID); }
        }
        // layer-1 forward weights
        transpose_pair(bdZ1, HID, tA, SB, HID, bH1, HID, tB, SB, HID, mp);
        if (S > 1) { EpiP ep = epi(bw_hh[1][0].dW, true); ep.accumulate = 1;
          gemm(dense<T>(tA + B, HID, mp), dense<T>(tB, HID, mp), dense_out(HID), ep, HID, HID, (S - 1) * B); }
        for (int d = 0; d < 2; ++d) {
            cast_tr<T, T>(bH0[d], HID, nullptr, 0, tB, mp, SB, HID);
            EpiP ep = epi(bw_ih[1][0].dW + d * HID, true); ep.accumulate = 1;
            gemm(dense<T>(tA, HID, mp), dense<T>(tB, HID, mp), dense_out(2 * HID), ep, HID, HID, SB);
        }
        colsum(bdZ1, HID, SB, HID, dbb_ih[1][0], dbb_hh[1][0]);
        // ---- layer 0, both directions (BPTT of the two in lockstep)
        { const LinW* w2[2] = {&bw_hh[0][0], &bw_hh[0][1]}; rnn_bwd2(bdH0, bH0, bdZ0, w2, B, S, 2); }
        for (int d = 0; d < 2; ++d) {
            transpose_pair(bdZ0[d], HID, tA, SB, HID, bH0[d], HID, tB, SB, HID, mp);
            if (S > 1) { EpiP ep = epi(bw_hh[0][d].dW, true); ep.accumulate = 1;     // forward: dZ[t] x H[t-1]; reverse: dZ[t] x H[t+1]
              gemm(dense<T>(tA + (d ? 0 : B), HID, mp), dense<T>(tB + (d ? B : 0), HID, mp), dense_out(HID), ep, HID, HID, (S - 1) * B); }
            cast_tr<T, T>(embg, EMB, nullptr, 0, tB, mp, SB, EMB);
            { EpiP ep = epi(bw_ih[0][d].dW, true); ep.accumulate = 1; gemm(dense<T>(tA, HID, mp), dense<T>(tB, EMB, mp), dense_out(EMB), ep, HID, EMB, SB); }
            colsum(bdZ0[d], HID, SB, HID, dbb_ih[0][d], dbb_hh[0][d]);
            { EpiP ep = epi(demb, true); ep.accumulate = 1;
              gemm(dense<T>(bdZ0[d], SB, HID), dense<T>(bw_ih[0][d].Wt, EMB, HID), dense_out_map(B, EMB, (long long)S * EMB), ep, SB, EMB, HID); }
        }
        STAGE("birnn_bwd");
    }

    // ---------------------------------------------------------------- backward
    // ---------------------------------------------------------------- gradient all-reduce buckets (comm.h)
    // Module groups of the flat buffer (hulc_amd/spec.py::layout keeps each group contiguous), in the order the backward finalises them.
    struct Bucket { int64_t lo, hi; };
    Bucket group_range(const char* prefix) const {          // [first element, start of the tensor behind the last one) of the tensors named prefix*: padding
        int64_t lo = numel, hi = 0;                          // between tensors (hulc_amd.spec: 64 elements) rides with the group, a packed layout has none
        const std::string a(prefix);
        for (size_t i = 0; i < tab_order.size(); ++i) {
            const auto& kv = tab_order[i];
            if (kv.first.compare(0, a.size(), a) == 0) { lo = std::min(lo, kv.second.off); hi = std::max(hi, i + 1 < tab_order.size() ? tab_order[i + 1].second.off : numel); }
        }
        if (hi <= lo) return Bucket{0, 0};
        return Bucket{lo, std::min<int64_t>(hi, numel)};
    }
    std::vector<std::pair<std::string, Ref>> tab_order;       // (name, ref) sorted by offset
    // issue order: [action_decoder .. end of buffer] (decoder, CLIP head, logit_scale) | plan_proposal | plan_recognition | goal encoders | perceptual encoders
    std::vector<Bucket> bucket_plan() const {
        std::vector<Bucket> v;
        const Bucket dec = group_range("action_decoder.");
        v.push_back(Bucket{dec.lo, numel});
        v.push_back(group_range("plan_proposal."));
        v.push_back(group_range("plan_recognition."));
        const Bucket vg = group_range("visual_goal."), lg = group_range("language_goal.");
        v.push_back(Bucket{std::min(vg.lo, lg.lo), std::max(vg.hi, lg.hi)});
        v.push_back(group_range("perceptual_encoder."));
        return v;
    }
    int comm_buckets(int64_t* lo, int64_t* hi, int cap) override {
        if (!bound) { hulc_set_error("hulc_comm_buckets before hulc_bind_params"); return -1; }
        const std::vector<Bucket> v = bucket_plan();
        for (int i = 0; i < (int)v.size() && i < cap; ++i) { lo[i] = v[i].lo; hi[i] = v[i].hi; }
        return (int)v.size();
    }
    // ---- a timed-out persistent recurrence under data parallelism: the failing rank's garbage gradients are SUMMED into every rank's
    // buffer, so skipping the optimizer step must be a decision of the whole job — not of the one rank whose launch failed (the others would
    // apply the garbage and the ranks' weights would diverge).  The vote costs no collective of its own: before the last bucket (the perceptual
    // encoders', whose tensors leave 64-element alignment padding) is reduced, each rank writes 1.0 into ONE padding element of its gradient
    // buffer if its skip word carries this step's tag, else 0.0; after the SUM a non-zero element means "some rank failed" -> every rank
    // sets its own skip word (adam / sgd / scaler_update then return without touching p / m / v) and clears the element.
    // A layout without such a padding element (hulc_bind_params accepts any 4-aligned, tightly packed table) votes through a word of the engine's own
    // (vote_word): one extra 4-byte all-reduce behind the range that starts the buffer (the last bucket issued) — never through an element that belongs to a tensor.
    int64_t skip_pad = -1;
    float* vote_word = nullptr;
    int64_t vote_pad() const { return force_vote_word ? -1 : skip_pad; }
    float* vote_ptr() { if (vote_pad() >= 0) return G + skip_pad; if (!vote_word) vote_word = alloc<float>(64); return vote_word; }
    void skip_vote_put(hipStream_t s) {
        if (!rp_skip) rp_skip = alloc<unsigned>(64);
        float* w = vote_ptr();
        if (rp_skip && w) hipLaunchKernelGGL(dp_skip_put_kernel, dim3(1), dim3(1), 0, s, (const unsigned*)rp_skip, opt_seq + 1, w);
    }
    void skip_vote_get(hipStream_t s) {
        float* w = vote_ptr();
        if (!rp_skip || !w) return;
        hipLaunchKernelGGL(dp_skip_get_kernel, dim3(1), dim3(1), 0, s, w, rp_skip, opt_seq + 1);
    }
    void dp_skip_vote(int phase) override { if (phase == 1) skip_vote_put(st); else if (phase == 2) skip_vote_get(st); }
    int ar_dtype = -1;          // >= 0 while a backward with overlapped all-reduce is running: bucket dtype
    unsigned ar_sent = 0;       // bit i: bucket i already issued in this backward
    // SUM all-reduce of G[lo, hi) on the collectives' stream, ordered after everything enqueued on `st` so far
    int reduce_range(int64_t lo, int64_t hi, int dtype, int span = -1) {
        if (hi <= lo) return 0;
        GradComm& c = *comm;
        const bool vote = vote_pad() >= 0 ? (skip_pad >= lo && skip_pad < hi) : lo == 0;      // the range that carries the job-wide skip vote: the bucket issued LAST (perceptual encoders, offset 0) / the whole buffer
        if (vote) skip_vote_put(st);
        c.gate_from(st);
        const size_t n = (size_t)(hi - lo);
        if (span >= 0) c.span_begin(span, (dtype == HULC_DTYPE_F32 ? 4.0 : 2.0) * n);
        int rc;
        if (dtype == HULC_DTYPE_BF16 || dtype == HULC_DTYPE_F16) {
            // 16-bit wire format: G -> staging (this unit's 16-bit type), all-reduce, widen back.  bf16 keeps fp32's range (no scaling needed);
            // fp16 is offered for the fp16 engine, whose gradients are already loss-scaled into fp16's range.
            if (c.stage_elems < numel) {
                if (c.stage) hipFree(c.stage);
                if (hipMalloc(&c.stage, (size_t)numel * 2 + 256) != hipSuccess) { hulc_set_error("hulc_allreduce_grads: staging buffer allocation failed"); return 1; }
                c.stage_elems = numel;
            }
            h16_t* sg = reinterpret_cast<h16_t*>(c.stage) + lo;
            hipLaunchKernelGGL((cast_kernel<float, h16_t>), dim3(std::min<long long>(2048, cdiv((long long)n, 1024))), dim3(256), 0, c.cs, G + lo, sg, (long long)n);
#ifdef HULC_HALF_F16
            const int wire = GradComm::F16;
#else
            const int wire = GradComm::BF16;
#endif
            rc = GradComm::api().allreduce(sg, sg, n, wire, GradComm::SUM, c.comm, c.cs);
            hipLaunchKernelGGL((cast_kernel<h16_t, float>), dim3(std::min<long long>(2048, cdiv((long long)n, 1024))), dim3(256), 0, c.cs, sg, G + lo, (long long)n);
            c.bytes_reduced += 2.0 * n;
        } else {
            rc = GradComm::api().allreduce(G + lo, G + lo, n, GradComm::F32, GradComm::SUM, c.comm, c.cs);
            c.bytes_reduced += 4.0 * n;
        }
        c.n_collectives++;
        if (vote && vote_pad() < 0 && rc == 0 && vote_ptr()) { rc = GradComm::api().allreduce(vote_word, vote_word, 1, GradComm::F32, GradComm::SUM, c.comm, c.cs); c.n_collectives++; c.bytes_reduced += 4.0; }
        if (vote) skip_vote_get(c.cs);
        if (span >= 0) c.span_end(span);
        if (rc != 0) { hulc_set_error("ncclAllReduce failed: %s", GradComm::err(rc)); return 1; }
        return 0;
    }
    // called by backward() after the stage that finalises bucket i has been enqueued
    int bucket_ready(int i) {
        if (ar_dtype < 0 || (ar_sent >> i) & 1u) return 0;
        const std::vector<Bucket> v = bucket_plan();
        ar_sent |= 1u << i;
        lazy_sweep(v[i].lo, v[i].hi, st);    // the bucket is final: a lazily zeroed tensor in it that no writer touched becomes zeros before it goes on the wire
        return reduce_range(v[i].lo, v[i].hi, ar_dtype, comm_timing ? i : -1);
    }
    int check_ar_dtype(int dtype, const char* who) {
        if (!comm) { hulc_set_error("%s: no communicator (hulc_comm_init first)", who); return 1; }
        if (dtype != HULC_DTYPE_F32 && dtype != HULC_DTYPE_BF16 && dtype != HULC_DTYPE_F16) { hulc_set_error("%s: bucket dtype must be HULC_DTYPE_F32 or a 16-bit type", who); return 1; }
        if (dtype != HULC_DTYPE_F32) {
            if (std::is_same<T, float>::value) { hulc_set_error("%s: 16-bit buckets need a bf16 / fp16 engine (the fp32 engine has no 16-bit kernels in its unit)", who); return 1; }
#ifdef HULC_HALF_F16
            if (dtype != HULC_DTYPE_F16) { hulc_set_error("%s: the fp16 engine offers fp16 buckets (loss-scaled gradients), not bf16", who); return 1; }
#else
            if (dtype != HULC_DTYPE_BF16) { hulc_set_error("%s: the bf16 engine offers bf16 buckets, not fp16", who); return 1; }
#endif
        }
        return 0;
    }
    int allreduce_grads(int dtype) override {
        if (check_ar_dtype(dtype, "hulc_allreduce_grads")) return 1;
        if (!bound) { hulc_set_error("hulc_allreduce_grads before hulc_bind_params"); return 1; }
        if (bwd_stage != 0) { hulc_set_error("hulc_allreduce_grads: encoder part of the backward still pending"); return 1; }
        lazy_sweep(0, numel, st);
        if (reduce_range(0, numel, dtype)) return 1;
        comm->gate_to(st);
        return 0;
    }
    // the buckets must partition [0, numel): every gradient element reduced exactly once
    bool bucket_plan_ok() const {
        std::vector<Bucket> v = bucket_plan();
        v.erase(std::remove_if(v.begin(), v.end(), [](const Bucket& b) { return b.hi <= b.lo; }), v.end());
        std::sort(v.begin(), v.end(), [](const Bucket& a, const Bucket& b) { return a.lo < b.lo; });
        if (v.empty() || v.front().lo != 0 || v.back().hi != numel) return false;
        for (size_t i = 0; i + 1 < v.size(); ++i) if (v[i].hi != v[i + 1].lo) return false;
        return true;
    }
    int backward_allreduce(int dtype) override {
        if (check_ar_dtype(dtype, "hulc_backward_allreduce")) return 1;
        if (!bucket_plan_ok()) { hulc_set_error("hulc_backward_allreduce: the module-group buckets do not partition the gradient buffer (layout changed?)"); return 1; }
        ar_dtype = dtype; ar_sent = 0;
        if (comm_timing) comm->bwd_mark(true, st);
        int rc = backward(-1);
        if (!rc) for (int i = 0; i < 5 && !rc; ++i) rc = bucket_ready(i);     // whatever no stage hook covered (model kinds without that stage)
        ar_dtype = -1;
        if (rc) return rc;
        if (comm_timing) comm->bwd_mark(false, st);
        comm->gate_to(st);                 // Adam (or anything enqueued next on the engine stream) runs after the last collective
        return 0;
    }

    int bwd_stage = 0;    // 0: nothing pending; 1: part 0 done, encoders pending
    int backward(int part = -1) override {
        if (!have_fwd) { hulc_set_error("hulc_backward without a preceding hulc_forward_loss"); return 1; }
        persist_check("hulc_backward", false);
        bwd_since_opt = true;
        if (part == 1 && bwd_stage != 1) { hulc_set_error("hulc_backward_part(1) must follow hulc_backward_part(0)"); return 1; }
        if (part != 1 && bwd_stage != 0) { hulc_set_error("hulc_backward: encoder part of the previous backward still pending"); return 1; }
        const hulc_batch* b = &cur;
        const int B = b->B, S = b->S, N = B * S, SB = S * B;
        const bool hulc = cfg.kind == HULC_KIND_HULC;
        const float dp = cfg.dropout_p;
        const long long BH = (long long)B * HID;
        if (part == 1) goto encoders;
        if (!arena_clean) HIP_CHECK(hipMemsetAsync(zero_arena, 0, sizeof(float) * zero_n, st));   // demb, dgoal, dseqf, heads / fc7 gradient temporaries, work counters
        arena_clean = false;
        work_ctr_next = 0;
        // tests (hulc_set_option debug_poison_partials): the weight-gradient slab arena is never zeroed — every slab element must be WRITTEN before the
        // unpack launches sum it.  NaN-filling it makes a slab cell that is read-modify-written (the ragged last k-tile of fc7) visible
        if (poison_partials) HIP_CHECK(hipMemsetAsync(this->part, 0xFF, sizeof(float) * (size_t)this->partcap, st));
        {
        bool have_dseq = false, dseq_cast_done = false;
        // ---- CLIP backward
        if (clip_n > 0) {
            const int n = clip_n;
            hipLaunchKernelGGL((cast_kernel<float, T>), dim3(1), dim3(256), 0, st, dimg, dimg_t, (long long)n * GOAL);
            hipLaunchKernelGGL((cast_kernel<float, T>), dim3(1), dim3(256), 0, st, dtxt, dtxt_t, (long long)n * GOAL);
            // image branch: img = im2(relu(im0(sf)))
            lin_wgrad(dimg_t, im1, 128, n, GOAL, 128, cl_im2.dW, 128, cl_im2.db);
            { EpiP ep = epi(dim1, false); ep.mask = im1; lin_dgrad(dimg_t, n, cl_im2, ep, dense_out(128)); }
            lin_wgrad(dim1, sf_m, FCH, n, 128, FCH, cl_im0.dW, FCH, cl_im0.db);
            { EpiP ep = epi(dsf_m, true); lin_dgrad(dim1, n, cl_im0, ep, dense_out(FCH)); }
            hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(cdiv(n * FCH, 256)), dim3(256), 0, st, dsf_m, auxrows, n, FCH, dseqf, (long long)FCH);
            // text branch
            lin_wgrad(dtxt_t, la1, 128, n, GOAL, 128, cl_la2.dW, 128, cl_la2.db);
            { EpiP ep = epi(dla1, false); ep.mask = la1; lin_dgrad(dtxt_t, n, cl_la2, ep, dense_out(128)); }
            lin_wgrad(dla1, g_m, GOAL, n, 128, GOAL, cl_la0.dW, GOAL, cl_la0.db);
            { EpiP ep = epi(dg_m, true); lin_dgrad(dla1, n, cl_la0, ep, dense_out(GOAL)); }
            hipLaunchKernelGGL(scatter_rows_add_kernel, dim3(cdiv(n * GOAL, 256)), dim3(256), 0, st, dg_m, auxrows, n, GOAL, dgoal, (long long)GOAL);
            have_dseq = true;
        }
        // ---- decoder backward
        STAGE("clip_bwd");
        {
            // heads
            const long long lastBH = (long long)(S - 1) * BH;          // the BPTT's first step (t = S-1) needs no multiplication: written by the GEMM that produces dH
            { EpiP ep = epi(dH1, false); ep.out2 = dZ1 + lastBH; ep.out2_lo = lastBH; ep.out2_hi = lastBH + BH; ep.out2_mask = H1 + lastBH;
              gemm(dense<T>(dheads, SB, NHEAD), dense<T>(wheadsT, HID, NHEAD), dense_out(HID), ep, SB, HID, NHEAD); }
            {
                const HeadPack hp = head_pack();
                const int rows = head_rows[0] + head_rows[1] + head_rows[2] + head_rows[3];
                bool slabs = false;
                if constexpr (std::is_same<T, h16_t>::value) {
                    // row-split weight gradient (see tr_wgrads_flush): one slab per 256 rows in the (idle) convolution slab arena, summed by the unpack launch
                    const int nz = cdiv(SB, 256);
                    if (SB > 64 && part_cur + (int64_t)nz * NHEAD * HID <= this->partcap) {
                        LinBwdBatch bt{}; bt.M = SB; bt.store = 0; bt.mchunk = 256; bt.n = 1;
                        LinBwdJob& J = bt.j[0];
                        J.dY = dheads; J.X = H1; J.dW = nullptr; J.db = dbheads_tmp; J.ldx = HID; J.lddw = HID; J.N = NHEAD; J.K = HID; J.nx = cdiv(NHEAD, 64); J.blk0 = 0;
                        J.part = this->part + part_cur;
                        hipLaunchKernelGGL(lin_bwd_smallm_batched_kernel, dim3(J.nx * cdiv(HID, 128), nz), dim3(256), 0, st, bt);
                        hipLaunchKernelGGL(unpack_heads_grad_kernel, dim3(cdiv((long long)rows * HID, 256)), dim3(256), 0, st, hp, J.part, dbheads_tmp, HID, nz, (long long)NHEAD * HID);
                        slabs = true;
                    }
                }
                if (!slabs) {
                    lin_wgrad(dheads, H1, HID, SB, NHEAD, HID, dwheads_tmp, HID, dbheads_tmp);
                    hipLaunchKernelGGL(unpack_heads_grad_kernel, dim3(cdiv((long long)rows * HID, 256)), dim3(256), 0, st, hp, dwheads_tmp, dbheads_tmp, HID, 1, 0ll);
                }
            }
            // layer 1 BPTT
            rnn_bwd(dH1, H1, dZ1, whh1, B, S, 1, false, false, true);
            {
                const int mp = ldpad(SB);
                constexpr bool fuse_cs = std::is_same<T, h16_t>::value;     // 16-bit engines: the dZ transpose adds its column sums (= both bias gradients) on the way
                // 16-bit engines: H0^T is needed twice (dW_ih1 here, dW_hh0 below): transposed ONCE, by the same launch as dZ1^T and H1^T, into its own buffer
                // (was: a cast_transpose launch here and a second transpose of H0 next to dZ0 below: 4 launches -> 2)
                bool h0t_kept = false;
                if constexpr (std::is_same<T, h16_t>::value) {
                    if (!tB2) tB2 = alloc<T>(tcap);
                    if (tB2) { transpose_triple(dZ1, HID, tA, SB, HID, H1, HID, tB, SB, HID, H0, HID, tB2, SB, HID, mp, dbih1, dbhh1); h0t_kept = true; }
                }
                if (!h0t_kept)
                transpose_pair(dZ1, HID, tA, SB, HID, H1, HID, tB, SB, HID, mp, fuse_cs ? dbih1 : nullptr, fuse_cs ? dbhh1 : nullptr);
                h0t_valid = h0t_kept;
                bool paired = false;
                if constexpr (std::is_same<T, h16_t>::value) {
                    // dW_hh1 = dZ1[1:]^T H1[:-1] and dW_ih1 = dZ1^T H0 share dZ1^T up to a shift of one time step (B tokens): ONE launch that streams it once
                    // (gemm.h gemm_glds_pair_kernel: 1.5 MB per CU instead of 2 x 1 MB); needs H0^T next to H1^T (tB2) and B % 64 == 0
                    const bool pair_sw = gemm_pair_mode;          // hulc_set_option "gemm_pair" (default 0: measured slower than the two launches)
                    EpiP e1 = epi(whh1.dW, true), e2 = epi(wih1.dW, true);
                    if (pair_sw && gemm_use_glds && S > 1 && gemm_glds_pair_ok(dense<T>(tA, HID, mp), dense<T>(tB, HID, mp), dense<T>(tB, HID, mp), e1, e2, HID, HID, SB, B)) {
                        if (!tB2) tB2 = alloc<T>(tcap);
                        if (tB2) {
                            if (!h0t_kept) cast_tr<T, T>(H0, HID, nullptr, 0, tB2, mp, SB, HID);
                            e1.accumulate = grad_first(whh1.dW) ? 0 : 1; e2.accumulate = grad_first(wih1.dW) ? 0 : 1;
                            TimerScope ts(this, "gemm_128x128", "mfma", 2.0 * HID * HID * ((double)SB + (double)(S - 1) * B), (3.0 * HID * SB) * sizeof(T) + 8.0 * HID * HID, 1);
                            launch_gemm_glds_pair(st, dense<T>(tA, HID, mp), dense<T>(tB, HID, mp), dense<T>(tB2, HID, mp), dense_out(HID), e1, e2, HID, HID, SB, B);
                            paired = true;
                        }
                    }
                }
                // dW_hh1, dW_ih1 and (below) dH0 = dZ1 W_ih1 read dZ1 / dZ1^T, H1^T, H0^T, W_ih1^T and write three different buffers: one grouped launch
                if (!paired && h0t_kept && fuse_cs) gemm_group_begin();
                if (!paired) {
                if (S > 1) { EpiP ep = epi(whh1.dW, true); ep.accumulate = grad_first(whh1.dW) ? 0 : 1;
                  gemm(dense<T>(tA + B, HID, mp), dense<T>(tB, HID, mp), dense_out(HID), ep, HID, HID, (S - 1) * B); }
                if (!h0t_kept) cast_tr<T, T>(H0, HID, nullptr, 0, tB, mp, SB, HID);
                { EpiP ep = epi(wih1.dW, true); ep.accumulate = grad_first(wih1.dW) ? 0 : 1; gemm(dense<T>(tA, HID, mp), dense<T>(h0t_kept ? tB2 : tB, HID, mp), dense_out(HID), ep, HID, HID, SB); }
                }
                if (!fuse_cs) colsum(dZ1, HID, SB, HID, dbih1, dbhh1);
            }
            { EpiP ep = epi(dH0, false); ep.out2 = dZ0 + lastBH; ep.out2_lo = lastBH; ep.out2_hi = lastBH + BH; ep.out2_mask = H0 + lastBH;
              gemm(dense<T>(dZ1, SB, HID), dense<T>(wih1.Wt, HID, HID), dense_out(HID), ep, SB, HID, HID); }
            gemm_group_end();
            // layer 0 BPTT
            rnn_bwd(dH0, H0, dZ0, whh0, B, S, 1, false, false, true);
            {
                const int mp = ldpad(SB);
                // the column sums of dZ0 over all (t, b) rows = those of dC = sum_t dZ0: both bias gradients of layer 0 ride on this transpose too
                if (h0t_valid) {      // H0^T is still in tB2 (layer 1's launch above): dZ0^T and embg^T in one launch, no second transpose of H0
                    transpose_pair(dZ0, HID, tA, SB, HID, embg, DE, tB, SB, DE, mp, dbih0, dbhh0);
                    if (S > 1) { EpiP ep = epi(whh0.dW, true); ep.accumulate = grad_first(whh0.dW) ? 0 : 1;
                      gemm(dense<T>(tA + B, HID, mp), dense<T>(tB2, HID, mp), dense_out(HID), ep, HID, HID, (S - 1) * B); }
                } else {
                transpose_pair(dZ0, HID, tA, SB, HID, H0, HID, tB, SB, HID, mp, std::is_same<T, h16_t>::value ? dbih0 : nullptr, std::is_same<T, h16_t>::value ? dbhh0 : nullptr);
                if (S > 1) { EpiP ep = epi(whh0.dW, true); ep.accumulate = grad_first(whh0.dW) ? 0 : 1;
                  gemm(dense<T>(tA + B, HID, mp), dense<T>(tB, HID, mp), dense_out(HID), ep, HID, HID, (S - 1) * B); }
                cast_tr<T, T>(embg, DE, nullptr, 0, tB, mp, SB, DE);
                }
                { EpiP ep = epi(dwih0 + dec_plan, true); ep.accumulate = 1; gemm(dense<T>(tA, HID, mp), dense<T>(tB, DE, mp), dense_out(KIN), ep, HID, DE, SB); }
            }
            // d emb (gripper half), scattered back to (B,S,128)[..., 64:128]
            { EpiP ep = epi(demb + (EMB - DE), true); ep.accumulate = 1;
              gemm(dense<T>(dZ0, SB, HID), dense<T>(wih0T + (long long)dec_plan * HID, DE, HID), dense_out_map(B, EMB, (long long)S * EMB), ep, SB, DE, HID); }
            hipLaunchKernelGGL((sum_over_t_kernel<T>), dim3(cdiv(BH, 256)), dim3(256), 0, st, dZ0, S, BH, dC);
            if constexpr (!std::is_same<T, h16_t>::value) colsum(dC, HID, B, HID, dbih0, dbhh0);
            { EpiP ep = epi(dgoal, true); ep.accumulate = 1;
              gemm(dense<T>(dC, B, HID), dense<T>(wih0T + (long long)(dec_plan + DE) * HID, GOAL, HID), dense_out(GOAL), ep, B, GOAL, HID); }
            {
                const int mp = ldpad(B);
                transpose_pair(dC, HID, tA, B, HID, goal_t, GOAL, tB, B, GOAL, mp);
                EpiP ep = epi(dwih0 + dec_plan + DE, true); ep.accumulate = 1;
                gemm(dense<T>(tA, HID, mp), dense<T>(tB, GOAL, mp), dense_out(KIN), ep, HID, GOAL, B);
            }
            if (hulc) {
                { EpiP ep = epi(dplan, true); gemm(dense<T>(dC, B, HID), dense<T>(wih0T, PLAN, HID), dense_out(PLAN), ep, B, PLAN, HID); }
                // 16-bit engines: four waves per tile + stores into a fresh buffer (the plan columns of dW_ih0 have no other writer); the fp32 engine keeps the serial sum order
                if (!std::is_same<T, float>::value && NCLS <= 32) hipLaunchKernelGGL((plan_scatter_grad_w4_kernel<T>), dim3(NCAT, cdiv(HID, 64)), dim3(256), 0, st, dC, pidx, B, NCAT, NCLS, HID, KIN, dwih0, grads_fresh ? 1 : 0);
                else
                hipLaunchKernelGGL((plan_scatter_grad_lds_kernel<T>), dim3(NCAT, cdiv(HID, 64)), dim3(64), 0, st, dC, pidx, B, NCAT, NCLS, HID, KIN, dwih0);
            }
            if (mcil) {
                { EpiP ep = epi(dplan, true); gemm(dense<T>(dC, B, HID), dense<T>(wih0T, dec_plan, HID), dense_out(dec_plan), ep, B, dec_plan, HID); }
                lin_wgrad(dC, plan_t, dec_plan, B, HID, dec_plan, dwih0, KIN, nullptr);
            }
        }
        STAGE("decoder_bwd");
        // mcil with the tanh-RNN plan encoder: its BiRNN backward — four persistent recurrences, which need all 256 CUs resident — comes BEHIND the decoder.  With
        // the decoder's and the plan proposal's buckets already on the wire RCCL's kernels hold CUs and those recurrences fell to one launch per step
        // (comm_in_flight; VERDICT r5 weak #11: the N = 8 line of config 4 slower per GPU than N = 1 by construction).  Round 6: the two buckets are HELD until the
        // BiRNN backward has been enqueued — same bucket order on every rank (0, 1, 2, ...), 121 MB leave ~0.4 ms later and still have the plan-recognition tail,
        // the goal encoders and the whole encoder backward (~1.9 ms) to hide under; the recurrences stay persistent.
        const bool hold_buckets = hold_buckets_mode && mcil && !gru && ar_dtype >= 0 && persist_mode && !persist_under_comm && !(rp_probed && !rp_ok);
        if (!hold_buckets && bucket_ready(0)) return 1;       // action_decoder.*, proj_vis_lang.*, logit_scale are final: their all-reduce starts under the rest of the backward
        // ---- straight-through + KL -> logits grads; plan proposal backward
        if (hulc) {
            hipLaunchKernelGGL((st_softmax_bwd_kernel<T>), dim3(B * NCAT), dim3(64), 0, st, probs, dplan, dpr_kl, NCLS, dprl, dprl_t, dpp_kl, dppl_t);
            DenseOut om = dense_out(EMB + GOAL);
            mlp_bwd(dppl_t, ppx, EMB + GOAL, B, pp, 5, ppa, dt_a, dt_a + (long long)B * HID, dppx, &om, 0);
            hipLaunchKernelGGL(pp_input_bwd_kernel, dim3(cdiv(B * (EMB + GOAL), 256)), dim3(256), 0, st, dppx, B, EMB, GOAL, demb, (long long)S * EMB, dgoal);
            if (bucket_ready(1)) return 1;   // plan_proposal.* final
            // fc_state of plan recognition
            lin_wgrad(dprl_t, seqf_t, FCH, B, PLAN, FCH, pr_fs.dW, FCH, pr_fs.db);
            { EpiP ep = epi(dseqf, true);
              if (have_dseq) ep.accumulate = 1;       // the CLIP branch already wrote its share
              else { ep.out2 = dseq_t; ep.out2_lo = 0; ep.out2_hi = (long long)B * FCH; dseq_cast_done = true; }     // sole contribution: plain store + the 16-bit copy in the same epilogue
              lin_dgrad(dprl_t, B, pr_fs, ep, dense_out(FCH)); }
            have_dseq = true;
        }
        // ---- mcil: reparametrised sample + KL -> fc_state grads; plan proposal and BiRNN backward
        if (mcil) {
            const int n = PLAN / 2;
            hipLaunchKernelGGL((normal_rsample_bwd_kernel<T>), dim3(cdiv(B * n, 256)), dim3(256), 0, st, dplan, plan_eps, pr_logits, dpr_kl, B, n, dprl_t);
            hipLaunchKernelGGL((cast_kernel<float, T>), dim3(cdiv(B * PLAN, 256)), dim3(256), 0, st, dpp_kl, dppl_t, (long long)B * PLAN);
            DenseOut om = dense_out(EMB + GOAL);
            mlp_bwd(dppl_t, ppx, EMB + GOAL, B, pp, 5, ppa, dt_a, dt_a + (long long)B * HID, dppx, &om, 0);
            hipLaunchKernelGGL(pp_input_bwd_kernel, dim3(cdiv(B * (EMB + GOAL), 256)), dim3(256), 0, st, dppx, B, EMB, GOAL, demb, (long long)S * EMB, dgoal);
            if (!hold_buckets && bucket_ready(1)) return 1;   // plan_proposal.* final
            if (gru) bigru_bwd(dprl_t, B, S); else birnn_bwd(dprl_t, B, S);
            if (hold_buckets && (bucket_ready(0) || bucket_ready(1))) return 1;      // the held buckets, in order, behind the persistent BiRNN backward
        }
        // ---- plan recognition backward
        if (have_dseq) {
            if (!dseq_cast_done) hipLaunchKernelGGL((cast_kernel<float, T>), dim3(cdiv(B * FCH, 256)), dim3(256), 0, st, dseqf, dseq_t, (long long)B * FCH);
            lin_wgrad(dseq_t, xm, EMB, B, FCH, EMB, pr_fc.dW, EMB, pr_fc.db);
            { EpiP ep = epi(dxm, true); lin_dgrad(dseq_t, B, pr_fc, ep, dense_out(EMB)); }
            float* dx = dxa; float* dnext = dxb;
            if (!ln_bwd_can_bcast) hipLaunchKernelGGL(bcast_over_s_kernel, dim3(cdiv((long long)N * EMB, 256)), dim3(256), 0, st, dxm, B, S, EMB, dx);
            for (int l = 1; l >= 0; --l) {
                // LN2 (16-bit engines: the last layer's incoming gradient dxm / S is broadcast over the window inside the kernel)
                const bool bc = ln_bwd_can_bcast && l == 1;
                bool ffn_fused = false;
                if constexpr (std::is_same<T, h16_t>::value) ffn_fused = tr_fused_mode && S <= 64;       // row-wise: half windows for S > 32
                // 16-bit fused path: every incoming gradient of the layer's four Linear layers stays in its own buffer, and the eight weight / bias
                // gradients of both layers run as ONE row-split launch after the loop (tr_wgrads_flush) instead of 16 transposes + GEMMs
                T *b_c = dt_c, *b_a = dt_a, *b_d = dt_c, *b_b = dt_b;
                bool defer_w = false;
                if constexpr (std::is_same<T, h16_t>::value) {
                    static const int defer_sw = HULC_SWITCH("HULC_TR_WGRAD_BATCH", 1);
                    defer_w = ffn_fused && defer_sw;
                    if (defer_w) {
                        if (!trb_c[l]) { trb_c[l] = alloc<T>((int64_t)maxN * EMB); trb_a[l] = alloc<T>((int64_t)maxN * FF); trb_d[l] = alloc<T>((int64_t)maxN * EMB, l ? "tr_bd1" : "tr_bd0"); trb_b[l] = alloc<T>((int64_t)maxN * 3 * EMB, l ? "tr_bb1" : "tr_bb0"); }
                        b_c = trb_c[l]; b_a = trb_a[l]; b_d = trb_d[l]; b_b = trb_b[l];
                    }
                }
                if constexpr (std::is_same<T, h16_t>::value) {
                    if (ffn_fused) {      // LN2 backward + both data-gradient GEMMs of the FFN: one launch (tr_fused.h); the weight gradients read what it wrote
                        if (!dparts) dparts = alloc<float>(4ll * maxN * EMB);
                        TrFfnBwdP q{};
                        q.dx = bc ? dxm : dx; q.bcast = bc ? 1 : 0; q.bdiv = (float)S; q.y2 = y2[l]; q.st2 = st2[l]; q.n2g = tr_n2g[l]; q.dg2 = d_tr_n2g[l]; q.db2 = d_tr_n2b[l];
                        q.W2t = tr_l2[l].Wtfr; q.W1t = tr_l1[l].Wtfr; q.hff = hff[l]; q.dt_c = b_c; q.dt_a = b_a; q.part = dparts; q.B = B; q.S = S; q.N = N; q.dp = dp;
                        q.seed_y = site_seed(4 + 4 * l);
                        {
                            TimerScope ts(this, "transformer_fused", "mfma", 2.0 * N * (2.0 * EMB * FF), (double)N * (2 * EMB + FF) * sizeof(T));
                            launch_tr_ffn_bwd(st, q);
                        }
                        if (defer_w) {
                            tr_wgrad_add(b_c, hff[l], tr_l2[l]); tr_wgrad_add(b_a, x1t[l], tr_l1[l]);
                        } else {
                            lin_wgrad(dt_c, hff[l], FF, N, EMB, FF, tr_l2[l].dW, FF, tr_l2[l].db);
                            lin_wgrad(dt_a, x1t[l], EMB, N, FF, EMB, tr_l1[l].dW, EMB, tr_l1[l].db);
                        }
                    }
                }
                if (!ffn_fused) {
                ln_bwd(bc ? dxm : dx, EMB, y2[l], EMB, st2[l], tr_n2g[l], N, EMB, dy_f, EMB, 0, dt_c, EMB, d_tr_n2g[l], d_tr_n2b[l], dp, site_seed(4 + 4 * l),
                       bc ? S : 0, (float)S);   // dt_c = dropout mask of the FFN branch applied to dy_f
                lin_wgrad(dt_c, hff[l], FF, N, EMB, FF, tr_l2[l].dW, FF, tr_l2[l].db);
                { EpiP ep = epi(dt_a, false); ep.mask = hff[l]; ep.alpha = dp > 0.f ? 1.f / (1.f - dp) : 1.f; lin_dgrad(dt_c, N, tr_l2[l], ep, dense_out(FF)); }
                lin_wgrad(dt_a, x1t[l], EMB, N, FF, EMB, tr_l1[l].dW, EMB, tr_l1[l].db);
                { EpiP ep = epi(dnext, true); ep.res = dy_f; ep.res_f32 = 1; ep.res_ld = EMB; lin_dgrad(dt_a, N, tr_l1[l], ep, dense_out(EMB)); }
                }
                // 16-bit fused path: LN1 backward, out_proj data gradient, attention backward and in_proj data gradient of a window in one launch (tr_fused.h)
                bool attn_fused = false;
                if constexpr (std::is_same<T, h16_t>::value) {
                    static const int sw = HULC_SWITCH("HULC_TR_ATTN_BWD", 1);
                    attn_fused = ffn_fused && defer_w && sw && EMB == 128 && NH == 8 && S <= 32;       // a window's dS / P tiles of 8 heads: 115 KB of LDS at S = 32, 266 KB at 64 -> launch per op
                    if (attn_fused) {
                        TrAttnBwdP q{};
                        q.parts = dparts; q.part_stride = (long long)N * EMB; q.nparts = 4; q.y1 = y1[l]; q.st1 = st1[l]; q.n1g = tr_n1g[l]; q.dg1 = d_tr_n1g[l]; q.db1 = d_tr_n1b[l];
                        q.Wot = tr_out[l].Wtfr; q.Wint = tr_in[l].Wtfr; q.qkv = qkv[l]; q.Pat = Pat[l]; q.b_d = b_d; q.b_b = b_b; q.dy_f = dy_f; q.dx = dx;
                        q.B = B; q.S = S; q.dp = dp; q.seed_o = site_seed(2 + 4 * l); q.seed_att = site_seed(1 + 4 * l);
                        static const int tra_dbg = HULC_SWITCH("HULC_TRA_DBG", 0);
                        q.dbg = tra_dbg;
                        launch_tr_attn_bwd(st, q);
                        tr_wgrad_add(b_d, ao[l], tr_out[l]);
                        tr_wgrad_add(b_b, xt[l], tr_in[l]);
                        continue;
                    }
                }
                // LN1 (after the fused FFN backward its incoming gradient is the sum of the four hidden-quarter partials)
                ln_bwd(ffn_fused ? dparts : dnext, EMB, y1[l], EMB, st1[l], tr_n1g[l], N, EMB, dy_f, EMB, 0, b_d, EMB, d_tr_n1g[l], d_tr_n1b[l], dp, site_seed(2 + 4 * l), 0, 1.f,
                       ffn_fused ? 4 : 1, (long long)N * EMB);
                if (defer_w) tr_wgrad_add(b_d, ao[l], tr_out[l]);
                else lin_wgrad(b_d, ao[l], EMB, N, EMB, EMB, tr_out[l].dW, EMB, tr_out[l].db);
                { EpiP ep = epi(dt_a, false); lin_dgrad(b_d, N, tr_out[l], ep, dense_out(EMB)); }
                static const bool att32 = (HULC_SWITCH("HULC_ATT32", 1) != 0) && !std::is_same<T, float>::value;
                if (S <= 32 && att32) hipLaunchKernelGGL((attention_bwd32_kernel<T>), dim3(B * NH), dim3(64), 0, st, qkv[l], Pat[l], dt_a, B, S, EMB, NH, b_b, dp, site_seed(1 + 4 * l));
                else if (S <= 32) hipLaunchKernelGGL((attention_bwd_kernel<T, 32>), dim3(B * NH), dim3(64), 0, st, qkv[l], Pat[l], dt_a, B, S, EMB, NH, b_b, dp, site_seed(1 + 4 * l));
                else if (att32) {
                    static bool attr = false;
                    if (!attr) { hipFuncSetAttribute((const void*)attention_bwd64_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_BWD64_LDS); attr = true; }
                    hipLaunchKernelGGL((attention_bwd64_kernel<T>), dim3(B * NH), dim3(256), ATT_BWD64_LDS, st, qkv[l], Pat[l], dt_a, B, S, EMB, NH, b_b, dp, site_seed(1 + 4 * l));
                }
                else hipLaunchKernelGGL((attention_bwd_kernel<T, 64>), dim3(B * NH), dim3(64), 0, st, qkv[l], Pat[l], dt_a, B, S, EMB, NH, b_b, dp, site_seed(1 + 4 * l));
                if (defer_w) tr_wgrad_add(b_b, xt[l], tr_in[l]);
                else lin_wgrad(b_b, xt[l], EMB, N, 3 * EMB, EMB, tr_in[l].dW, EMB, tr_in[l].db);
                { EpiP ep = epi(dx, true); ep.res = dy_f; ep.res_f32 = 1; ep.res_ld = EMB; lin_dgrad(b_b, N, tr_in[l], ep, dense_out(EMB)); }
            }
            tr_wgrads_flush(N);
            // x0 = dropout(emb + pos): d(emb) += mask*dx ; dpos += sum_b
            {
                const int bchunk = std::is_same<T, float>::value ? B : 8;       // fp32 (parity) engine: one deterministic pass over the windows
                hipLaunchKernelGGL(pr_input_bwd_kernel, dim3(cdiv(S * EMB, 256), cdiv(B, bchunk)), dim3(256), 0, st, dx, B, S, EMB, dp, site_seed(0), demb, dpos, bchunk);
            }
        }
        STAGE("plan_recognition_bwd");
        if (bucket_ready(1) || bucket_ready(2)) return 1;   // plan_recognition.* final (plan_proposal too for the kinds that never touch it)
        // ---- goal encoder backward
        if (pair) {
            const int Bv = pairBv, Bl = B - pairBv;
            T* av[2] = {gl1, gl2};
            T* al[2] = {gl1 + (long long)Bv * HID, gl2 + (long long)Bv * HID};
            ln_bwd(dgoal, GOAL, gl3, GOAL, goal_st, ln_vg_g, Bv, GOAL, nullptr, 0, 0, dgl3_t, GOAL, d_ln_vg_g, d_ln_vg_b);
            DenseOut om = dense_out((long long)S * EMB);
            mlp_bwd(dgl3_t, emb + (long long)(S - 1) * EMB, (long long)S * EMB, Bv, vg, 3, av, dt_a, dt_a + (long long)B * HID, demb + (long long)(S - 1) * EMB, &om, 1);
            ln_bwd(dgoal + Bv * GOAL, GOAL, gl3 + Bv * GOAL, GOAL, goal_st + 2 * Bv, ln_lg_g, Bl, GOAL, nullptr, 0, 0, dgl3_t + Bv * GOAL, GOAL, d_ln_lg_g, d_ln_lg_b);
            mlp_bwd(dgl3_t + Bv * GOAL, lang_t, LANG, Bl, lg, 3, al, dt_a, dt_a + (long long)B * HID, nullptr, nullptr, 0);
        } else {
            T* acts[2] = {gl1, gl2};
            if (b->is_lang) {
                ln_bwd(dgoal, GOAL, gl3, GOAL, goal_st, ln_lg_g, B, GOAL, nullptr, 0, 0, dgl3_t, GOAL, d_ln_lg_g, d_ln_lg_b);
                mlp_bwd(dgl3_t, lang_t, LANG, B, lg, 3, acts, dt_a, dt_a + (long long)B * HID, nullptr, nullptr, 0);
            } else {
                ln_bwd(dgoal, GOAL, gl3, GOAL, goal_st, ln_vg_g, B, GOAL, nullptr, 0, 0, dgl3_t, GOAL, d_ln_vg_g, d_ln_vg_b);
                DenseOut om = dense_out((long long)S * EMB);
                mlp_bwd(dgl3_t, emb + (long long)(S - 1) * EMB, (long long)S * EMB, B, vg, 3, acts, dt_a, dt_a + (long long)B * HID,
                        demb + (long long)(S - 1) * EMB, &om, 1);
            }
        }
        }
        if (bucket_ready(3)) return 1;       // visual_goal.*, language_goal.* final
        if (part == 0) {
            lazy_sweep(0, numel, st);        // every lazy tensor is final here (none belongs to the encoders): what no writer touched is zeroed now
            if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in backward"); return 1; }
            bwd_stage = 1;
            return 0;
        }
    encoders:
        // ---- encoders backward
        {
            const Conv1Src s2s = conv1_src(cur2, false), s2g = conv1_src(cur2, true);
            const bool tail_fused = enc_tail_fusable();
            if (tail_fused) enc_tail_bwd_both(N);
            enc_bwd(encS, aS, conv1_src(*b, false), N, 0, pair ? &s2s : nullptr, tail_fused);
            STAGE("enc_static_bwd");
            enc_bwd(encG, aG, conv1_src(*b, true), N, 64, pair ? &s2g : nullptr, tail_fused);
            flush_unpacks();
            STAGE("enc_gripper_bwd");
        }
        if (bucket_ready(4)) return 1;       // perceptual_encoder.* final
        lazy_sweep(0, numel, st);
        if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in backward"); return 1; }
        have_fwd = false;
        bwd_stage = 0;
        grads_fresh = false;
        return 0;
    }

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
