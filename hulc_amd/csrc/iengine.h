// hulc_amd/csrc/iengine.h — the interface the C-ABI (capi.hip) drives; implemented by Engine<T> (engine.h) once per compute type:
// fp32 (parity) and bf16 in capi.hip's translation unit, fp16 in engine_f16.hip (common.h explains the per-TU half format).
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/hulc_hip.h"
#include "comm.h"

void hulc_set_error(const char* fmt, ...);

struct IEngine {
    virtual ~IEngine() { delete comm; delete comm_pending; }
    virtual int bind(float* p, float* g, float* m, float* v, int64_t numel, int n, const char* const* names, const int64_t* offs,
                     const int64_t* numels) = 0;
    virtual int prepare_weights(bool shadow_fresh = false) = 0;
    virtual int zero_grads() = 0;
    virtual int flush_grads() = 0;
    virtual int forward(const hulc_batch* b, float lw, float cw, float* out, int on_host) = 0;
    virtual int forward_pair(const hulc_batch* vis, const hulc_batch* lang, float lw, float cw, float* out8, int on_host) = 0;
    virtual int backward(int part = -1) = 0;   // -1: everything; 0: all but the perceptual encoders; 1: encoders (after part 0)
    virtual int validate(const hulc_batch* b, const hulc_val_noise* nz, float* out17, int32_t* plan_pp_out, int32_t* plan_pr_out, float* pred_pp_out,
                         float* pred_pr_out) = 0;
    virtual int clip_gt_encode(const float* lang_emb, int m, int slot) = 0;
    virtual int clip_gt_scores(int slot, float* out_host, int64_t cap, int32_t* n_out, int32_t* m_out) = 0;
    virtual int rollout_reset() = 0;
    virtual int rollout_plan(const hulc_rollout_obs* obs, const float* goal_static, const float* goal_gripper, const float* goal_lang,
                             const int32_t* plan_inject, int32_t* plan_out) = 0;
    virtual int rollout_act(const hulc_rollout_obs* obs, const float* u_mix, const float* u_act, float* action_out) = 0;
    virtual int rollout_get_goal(float* latent_goal_out) = 0;
    virtual int rollout_set_state(const void* plan, const float* latent_goal) = 0;
    virtual int optim(const hulc_optim& o) = 0;
    int adam(float lr, float b1, float b2, float eps, int64_t step, float gscale) {
        hulc_optim o{}; o.kind = HULC_OPT_ADAM; o.lr = lr; o.beta1 = b1; o.beta2 = b2; o.eps = eps; o.step = step; o.grad_scale = gscale;
        return optim(o);
    }
    virtual int scaler_enable(float init_scale, float growth, float backoff, int interval) = 0;
    virtual int scaler_get(float* scale, int32_t* tracker, int64_t* skipped, int32_t* last_inf, int64_t* taken) = 0;
    virtual int scaler_set(float scale, int32_t tracker, int64_t taken) = 0;
    // ---- data-parallel gradient all-reduce (comm.h): whole buffer after a finished backward, or bucketed inside the backward
    virtual int allreduce_grads(int bucket_dtype) = 0;
    virtual int backward_allreduce(int bucket_dtype) = 0;
    virtual int comm_buckets(int64_t* lo, int64_t* hi, int cap) = 0;        // the bucket ranges in issue order; returns their number
    GradComm* comm = nullptr;
    GradComm* comm_pending = nullptr;       // prepared (RCCL resolved, private stream) but not yet initialised
    int comm_prepare() {
        if (comm || comm_pending) return 0;
        GradComm* c = new GradComm();
        if (c->prepare()) { delete c; return 1; }
        comm_pending = c;
        return 0;
    }
    int comm_init(const void* unique_id, int rank, int world) {
        if (comm) { hulc_set_error("hulc_comm_init: this context already has a communicator"); return 1; }
        if (comm_prepare()) return 1;
        GradComm* c = comm_pending; comm_pending = nullptr;
        if (c->init(unique_id, rank, world)) { delete c; return 1; }
        comm = c;
        return 0;
    }
    int comm_destroy() { delete comm; comm = nullptr; delete comm_pending; comm_pending = nullptr; return 0; }
    virtual int get_tensor(const char* name, float* out, int64_t cap, int64_t* n) = 0;
    virtual int get_plan_idx(int32_t* out, int64_t cap) = 0;
    virtual int64_t workspace_bytes() const = 0;
    virtual void set_kl_beta(float b) = 0;
    virtual void set_dropout(float p) = 0;
    void set_timing(bool on, const char* filter) { timing = on; timing_filter = (filter && *filter) ? std::string(",") + filter + "," : std::string(); }
    // runtime options (hulc_set_option).  "persistent_rnn": 1 (default) = the 2048-wide recurrences of the 16-bit engines run as one
    // persistent launch each (rnn_persist.h) after a first launch has verified the XCD census on this device; 0 = one launch per time step
    // (what a process that SHARES the GPU's CUs with another process must choose: the persistent launch needs all 256 CUs resident)
    // "fused_transformer": 1 (default) = one launch per plan-recognition encoder layer in the forward of the 16-bit engines (tr_fused.h, S <= 32);
    // 0 = the seven unfused launches per layer (what the fp32 engine and S > 32 run) — tests compare the two
    // "persist_under_comm": 0 (default) = recurrences that follow an issued all-reduce bucket of hulc_backward_allreduce run one launch per step (RCCL's
    // kernels hold CUs; a persistent launch needs all of them); 1 = keep them persistent.  "comm_timing": 1 = hulc_backward_allreduce records events around
    // every bucket's collective and at the end of the backward (hulc_comm_timeline).  "debug_poison_partials": tests only — fills the weight-gradient
    // partial arena with NaN before the next backward (a slab that is read before it is written then shows up in the gradients).
    // "lazy_zero_grads": 1 (default; 16-bit engines) = hulc_zero_grads only marks the large store-first weight gradients stale instead of zeroing them
    // (engine.h: LazyG); 0 = the plain memset of the whole buffer.
    int persist_mode = 1, tr_fused_mode = 1, persist_under_comm = 0, comm_timing = 0, poison_partials = 0, persist_fault = 0, lazy_zero_mode = 1, u8_fold_mode = 1, force_vote_word = 0, gemm_group_mode = 0, hold_buckets_mode = 1;
    virtual int get_option(const char* name, long long* value) = 0;
    virtual void dp_skip_vote(int phase) = 0;
    virtual void set_adam_fuse(bool on) = 0;
    int set_option(const char* name, long long value) {
        if (name && !strcmp(name, "persistent_rnn")) { persist_mode = value != 0; return 0; }
        if (name && !strcmp(name, "fused_transformer")) { tr_fused_mode = value != 0; return 0; }
        if (name && !strcmp(name, "persist_under_comm")) { persist_under_comm = value != 0; return 0; }
        // "dp_hold_buckets" (default 1): mcil with the tanh-RNN plan encoder under hulc_backward_allreduce — the decoder's and the plan proposal's buckets are issued BEHIND
        // the BiRNN backward so that its four recurrences stay persistent (no RCCL kernel holds CUs yet); 0 = issue them as early as possible (round 5), the BiRNN
        // backward then runs one launch per step unless persist_under_comm is set
        if (name && !strcmp(name, "dp_hold_buckets")) { hold_buckets_mode = value != 0; return 0; }
        if (name && !strcmp(name, "comm_timing")) { comm_timing = value != 0; return 0; }
        if (name && !strcmp(name, "debug_poison_partials")) { poison_partials = value != 0; return 0; }
        if (name && !strcmp(name, "lazy_zero_grads")) { lazy_zero_mode = value != 0; return 0; }
        // "gemm_pair" (default 0; 16-bit engines): the decoder's two layer-1 weight gradients, which share dZ1^T up to a shift of one time step, run as ONE launch that
        // streams the shared operand once (gemm.h gemm_glds_pair_kernel; needs the batch to be a multiple of 64 windows); 0 = two launches
        if (name && !strcmp(name, "gemm_pair")) { gemm_pair_mode = value != 0; return 0; }
        // "gemm_group" (default 0: measured equal to the three launches, profiles/r06_ab_gemm_group.txt; 16-bit engines): the decoder backward's three independent 2048^3 products (dW_hh1, dW_ih1, dH0) as ONE grouped launch (gemm.h gemm_glds_group_kernel)
        if (name && !strcmp(name, "gemm_group")) { gemm_group_mode = value != 0; return 0; }
        if (name && !strcmp(name, "epilogue_fast")) { epilogue_fast = value != 0; return 0; }
        // "adam_fused_transposes" (default 1; 16-bit engines): the Adam / AdamW step writes the transposed 16-bit weight copies itself (kernels.h adam_tiled_kernel); 0 = flat pass + batched transpose
        if (name && !strcmp(name, "adam_fused_transposes")) { set_adam_fuse(value != 0); return 0; }
        if (name && !strcmp(name, "timer_event_fence")) { event_flags = value != 0 ? hipEventDefault : hipEventDisableSystemFence; for (auto& kv : timers) { for (auto& ev : kv.second.ev) { hipEventDestroy(ev.first); hipEventDestroy(ev.second); } kv.second.ev.clear(); kv.second.used = 0; } return 0; }
        // "u8_fold": 1 (default; 16-bit engines) = the uint8 ingest path multiplies the exact byte values and applies x = u (2/255) - 1 in conv1's epilogue /
        // weight-gradient slabs (conv_wgrad.h Conv1Src::fold); 0 = the value x itself is staged (16-bit rounded), as in rounds 2 - 4
        if (name && !strcmp(name, "u8_fold")) { u8_fold_mode = value != 0; return 0; }
        // "dp_skip_vote": for a gradient all-reduce done OUTSIDE the library (the torch.distributed fallback): 1 right before the collective that covers the
        // perceptual-encoder gradients, 2 right after it — the job-wide "a recurrence of this step failed on some rank" vote (engine.h skip_vote_put)
        if (name && !strcmp(name, "dp_skip_vote")) { dp_skip_vote((int)value); return 0; }
        if (name && !strcmp(name, "debug_vote_word")) { force_vote_word = value != 0; return 0; }      // tests: vote through the engine's own word even if the layout has a padding element
        if (name && !strcmp(name, "debug_persist_fault")) { persist_fault = (int)value; return 0; }      // tests: the next `value` persistent launches (after the probed first one) lose a producer
        hulc_set_error("hulc_set_option: unknown option '%s'", name ? name : "(null)");
        return 1;
    }
    hipStream_t st = nullptr;
    // ---- per-kernel-class HIP-event timers (bench.py roofline leg): events are recorded on `st` around the launches of a class
    struct KTimer { std::string name, bound; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; size_t used = 0; double flops = 0, bytes = 0; long long launches = 0; };
    std::map<std::string, KTimer> timers;
    bool epilogue_fast = true;              // hulc_set_option "epilogue_fast" 0: every GEMM epilogue through the generic path (A/B)
    bool gemm_pair_mode = false;            // measured SLOWER in the step (3.048 against 3.032 ms, class gemm_128x128 0.225 against 0.217 ms): off
    bool timing = false;
    std::string timing_filter;      // empty = every class; else ",a,b,": only the listed classes (keeps event overhead out of the timed region)
    int timer_depth = 0;            // a group scope (e.g. the S recurrent steps) suppresses the per-launch scopes inside it
    unsigned event_flags = hipEventDisableSystemFence;     // hulc_set_option "timer_event_fence" 1: default events (A/B of the timers' own cost)
    struct TimerScope {
        IEngine* e; IEngine::KTimer* t;
        TimerScope(IEngine* e_, const char* name, const char* bound, double flops, double bytes, int nlaunch = 1) : e(e_), t(nullptr) {
            if (!e->timing || e->timer_depth > 0) return;
            if (!e->timing_filter.empty() && e->timing_filter.find(std::string(",") + name + ",") == std::string::npos) return;
            e->timer_depth++;
            t = &e->timers[name];
            if (t->name.empty()) { t->name = name; t->bound = bound; }
            // timing-only events: no system-scope fence when they complete (hipEventDisableSystemFence) — a default event's cache write-back + invalidate stalled the
            // stream for 5.5 us per record behind the conv kernels (profiles/r05_host_lead.txt: 20 records = 117 us of a 3.1 ms step); timers_read synchronises the stream
            if (t->used == t->ev.size()) { hipEvent_t a, b; hipEventCreateWithFlags(&a, e->event_flags); hipEventCreateWithFlags(&b, e->event_flags); t->ev.emplace_back(a, b); }
            t->flops += flops; t->bytes += bytes; t->launches += nlaunch;
            hipEventRecord(t->ev[t->used].first, e->st);
        }
        ~TimerScope() { if (t) { hipEventRecord(t->ev[t->used].second, e->st); t->used++; e->timer_depth--; } }
    };
    int timers_read(char* out, int64_t cap, bool reset) {
        hipStreamSynchronize(st);
        std::string js = "{";
        bool first = true;
        for (auto& kv : timers) {
            KTimer& t = kv.second;
            double ms = 0;
            for (size_t i = 0; i < t.used; ++i) { float x = 0; hipEventElapsedTime(&x, t.ev[i].first, t.ev[i].second); ms += x; }
            char buf[512];
            snprintf(buf, sizeof(buf), "%s\"%s\": {\"bound\": \"%s\", \"launches\": %lld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}", first ? "" : ", ",
                     t.name.c_str(), t.bound.c_str(), t.launches, ms, t.flops, t.bytes);
            js += buf; first = false;
            if (reset) { t.used = 0; t.flops = t.bytes = 0; t.launches = 0; }
        }
        js += "}";
        if ((int64_t)js.size() + 1 > cap) { hulc_set_error("hulc_timers_read: buffer too small"); return 1; }
        memcpy(out, js.c_str(), js.size() + 1);
        return 0;
    }
};

// factories, one per translation unit
#define HULC_TU_DECLS                                                                                                                         \
    IEngine* make_engine(const hulc_config& cfg, int* rc);                                                                                   \
    int k_gemm_nt(int is_f32, const void* A, const void* B, float* C, int M, int N, int K, long long lda, long long ldb, long long ldc, \
                  const float* bias, int relu, void* stream);
namespace hulc_bf16 { HULC_TU_DECLS }   // capi.hip: HULC_DTYPE_F32 / HULC_DTYPE_BF16
namespace hulc_f16 { HULC_TU_DECLS }    // engine_f16.hip: HULC_DTYPE_F16
