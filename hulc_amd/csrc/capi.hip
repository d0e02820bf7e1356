// hulc_amd/csrc/capi.hip — extern "C" boundary of libhulc_hip.so (see include/hulc_hip.h).
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <algorithm>
#include <stdexcept>

#include "engine.h"
#include "sbert.h"

using namespace hulc_bf16;      // this translation unit: fp32 (parity) + bf16 engines and the per-kernel test entry points; fp16: engine_f16.hip

static thread_local char g_err[1024] = "";
void hulc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct hulc_ctx {
    IEngine* e = nullptr;
};

extern "C" {

const char* hulc_last_error(void) { return g_err; }

int hulc_ctx_create(const hulc_config* cfg, hulc_ctx** out) {
    if (!cfg || !out) { hulc_set_error("hulc_ctx_create: null argument"); return 1; }
    if (cfg->max_seq > 64 || cfg->max_seq < 1 || cfg->max_batch < 1 || cfg->max_window < cfg->max_seq) {
        hulc_set_error("hulc_ctx_create: need 1 <= max_seq <= 64, max_batch >= 1, max_window >= max_seq");
        return 1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { hulc_set_error("hulc_ctx_create: no HIP device visible"); return 1; }
    hulc_ctx* c = new hulc_ctx();
    int rc = 1;
    if (cfg->dtype == HULC_DTYPE_F32 || cfg->dtype == HULC_DTYPE_BF16) c->e = hulc_bf16::make_engine(*cfg, &rc);
    else if (cfg->dtype == HULC_DTYPE_F16) c->e = hulc_f16::make_engine(*cfg, &rc);
    else { hulc_set_error("hulc_ctx_create: unknown dtype %d", cfg->dtype); delete c; return 1; }
    if (rc) { delete c->e; delete c; return rc; }
    *out = c;
    return 0;
}
int hulc_ctx_destroy(hulc_ctx* ctx) {
    if (!ctx) return 0;
    hipDeviceSynchronize();
    delete ctx->e;
    delete ctx;
    return 0;
}
int hulc_set_stream(hulc_ctx* ctx, void* s) { ctx->e->st = (hipStream_t)s; return 0; }
int64_t hulc_workspace_bytes(const hulc_ctx* ctx) { return ctx->e->workspace_bytes(); }
int hulc_bind_params(hulc_ctx* ctx, float* p, float* g, float* m, float* v, int64_t numel, int32_t n, const char* const* names, const int64_t* offs,
                     const int64_t* numels) {
    if (!ctx || !p || !g || !m || !v) { hulc_set_error("hulc_bind_params: null argument"); return 1; }
    return ctx->e->bind(p, g, m, v, numel, n, names, offs, numels);
}
int hulc_prepare_weights(hulc_ctx* ctx) { return ctx->e->prepare_weights(); }
int hulc_zero_grads(hulc_ctx* ctx) { return ctx->e->zero_grads(); }
int hulc_flush_grads(hulc_ctx* ctx) { return ctx->e->flush_grads(); }
int hulc_forward_loss(hulc_ctx* ctx, const hulc_batch* b, float lw, float cw, float* out, int32_t on_host) {
    if (!ctx || !b) { hulc_set_error("hulc_forward_loss: null argument"); return 1; }
    return ctx->e->forward(b, lw, cw, out, on_host);
}
int hulc_forward_loss_pair(hulc_ctx* ctx, const hulc_batch* v, const hulc_batch* l, float lw, float cw, float* out, int32_t on_host) {
    if (!ctx || !v || !l) { hulc_set_error("hulc_forward_loss_pair: null argument"); return 1; }
    return ctx->e->forward_pair(v, l, lw, cw, out, on_host);
}
int hulc_backward(hulc_ctx* ctx) { return ctx->e->backward(-1); }
int hulc_backward_part(hulc_ctx* ctx, int32_t part) {
    if (part != 0 && part != 1) { hulc_set_error("hulc_backward_part: part must be 0 or 1"); return 1; }
    return ctx->e->backward(part);
}
int hulc_validate(hulc_ctx* ctx, const hulc_batch* batch, const hulc_val_noise* noise, float* out_host, int32_t* plan_idx_pp_out, int32_t* plan_idx_pr_out,
                  float* pred_pp_out, float* pred_pr_out) {
    if (!batch) { hulc_set_error("hulc_validate: null batch"); return 1; }
    return ctx->e->validate(batch, noise, out_host, plan_idx_pp_out, plan_idx_pr_out, pred_pp_out, pred_pr_out);
}
int hulc_clip_gt_encode(hulc_ctx* ctx, const float* lang_emb, int32_t m, int32_t slot) {
    if (!ctx) { hulc_set_error("hulc_clip_gt_encode: null context"); return 1; }
    return ctx->e->clip_gt_encode(lang_emb, m, slot);
}
int hulc_clip_gt_scores(hulc_ctx* ctx, int32_t slot, float* scores_host, int64_t cap_floats, int32_t* n_out, int32_t* m_out) {
    if (!ctx) { hulc_set_error("hulc_clip_gt_scores: null context"); return 1; }
    return ctx->e->clip_gt_scores(slot, scores_host, cap_floats, n_out, m_out);
}
int hulc_rollout_reset(hulc_ctx* ctx) { return ctx->e->rollout_reset(); }
int hulc_rollout_plan(hulc_ctx* ctx, const hulc_rollout_obs* obs, const float* goal_rgb_static, const float* goal_rgb_gripper, const float* goal_lang,
                      const int32_t* plan_idx_inject, int32_t* plan_idx_out) {
    if (!obs || !obs->rgb_static || !obs->rgb_gripper) { hulc_set_error("hulc_rollout_plan: null observation"); return 1; }
    return ctx->e->rollout_plan(obs, goal_rgb_static, goal_rgb_gripper, goal_lang, plan_idx_inject, plan_idx_out);
}
int hulc_rollout_get_goal(hulc_ctx* ctx, float* latent_goal_out) {
    if (!latent_goal_out) { hulc_set_error("hulc_rollout_get_goal: null output"); return 1; }
    return ctx->e->rollout_get_goal(latent_goal_out);
}
int hulc_rollout_set_state(hulc_ctx* ctx, const void* plan, const float* latent_goal) {
    if (!latent_goal) { hulc_set_error("hulc_rollout_set_state: null latent goal"); return 1; }
    return ctx->e->rollout_set_state(plan, latent_goal);
}
int hulc_rollout_act(hulc_ctx* ctx, const hulc_rollout_obs* obs, const float* u_mix, const float* u_act, float* action_out_host) {
    if (!obs || !obs->rgb_static || !obs->rgb_gripper || !obs->robot_obs_raw || !action_out_host) { hulc_set_error("hulc_rollout_act: null argument"); return 1; }
    return ctx->e->rollout_act(obs, u_mix, u_act, action_out_host);
}
int hulc_sbert_create(const hulc_sbert_config* cfg, hulc_sbert** out) {
    if (!cfg || !out) { hulc_set_error("hulc_sbert_create: null argument"); return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { hulc_set_error("hulc_sbert_create: no HIP device visible"); return 1; }
    if (cfg->hidden % 64 != 0 || cfg->hidden > 1024 || cfg->hidden % cfg->heads != 0 || cfg->hidden / cfg->heads > 64 || cfg->max_tokens > 128 || cfg->layers < 1) {
        hulc_set_error("hulc_sbert_create: unsupported shape (hidden %% 64 == 0, <= 1024; head dim <= 64; <= 128 tokens)");
        return 1;
    }
    hulc_sbert* c = new hulc_sbert();
    c->cfg = *cfg;
    if (!c->init()) { delete c; hulc_set_error("hulc_sbert_create: workspace allocation failed"); return 1; }
    *out = c;
    return 0;
}
int hulc_sbert_destroy(hulc_sbert* ctx) { delete ctx; return 0; }
int hulc_sbert_set_stream(hulc_sbert* ctx, void* s) { ctx->st = (hipStream_t)s; return 0; }
int hulc_sbert_bind(hulc_sbert* ctx, const float* flat, int64_t numel, int32_t n, const char* const* names, const int64_t* offs, const int64_t* numels) {
    if (!ctx || !flat || !names || !offs || !numels) { hulc_set_error("hulc_sbert_bind: null argument"); return 1; }
    ctx->w.clear();
    for (int i = 0; i < n; ++i) {
        if (offs[i] < 0 || offs[i] + numels[i] > numel) { hulc_set_error("hulc_sbert_bind: tensor %s out of range", names[i]); return 1; }
        ctx->w[names[i]] = flat + offs[i];
    }
    const long long H = ctx->cfg.hidden, I = ctx->cfg.intermediate;
    auto need = [&](const std::string& nm, long long cnt) {
        for (int i = 0; i < n; ++i)
            if (nm == names[i]) return numels[i] == cnt;
        return false;
    };
    bool ok = need("embeddings.word_embeddings.weight", (long long)ctx->cfg.vocab * H) && need("embeddings.position_embeddings.weight", (long long)ctx->cfg.max_position * H) &&
              need("embeddings.LayerNorm.weight", H) && need("embeddings.LayerNorm.bias", H);
    for (int l = 0; l < ctx->cfg.layers && ok; ++l) {
        const std::string p = "encoder.layer." + std::to_string(l) + ".";
        ok = need(p + "attention.self.query.weight", H * H) && need(p + "attention.self.key.weight", H * H) && need(p + "attention.self.value.weight", H * H) &&
             need(p + "attention.self.query.bias", H) && need(p + "attention.self.key.bias", H) && need(p + "attention.self.value.bias", H) &&
             need(p + "attention.output.dense.weight", H * H) && need(p + "attention.output.dense.bias", H) && need(p + "attention.output.LayerNorm.weight", H) &&
             need(p + "attention.output.LayerNorm.bias", H) && need(p + "intermediate.dense.weight", I * H) && need(p + "intermediate.dense.bias", I) &&
             need(p + "output.dense.weight", H * I) && need(p + "output.dense.bias", H) && need(p + "output.LayerNorm.weight", H) && need(p + "output.LayerNorm.bias", H);
    }
    if (!ok || !ctx->get("embeddings.token_type_embeddings.weight")) { hulc_set_error("hulc_sbert_bind: a BertModel tensor is missing or has the wrong size"); return 1; }
    ctx->bound = true;
    return 0;
}
int hulc_sbert_encode(hulc_sbert* ctx, const int32_t* ids, const int32_t* mask, int32_t B, int32_t L, float* out) {
    if (!ctx || !ids || !mask || !out) { hulc_set_error("hulc_sbert_encode: null argument"); return 1; }
    return ctx->encode(ids, mask, B, L, out);
}
int hulc_adam_step(hulc_ctx* ctx, float lr, float b1, float b2, float eps, int64_t step, float gs) { return ctx->e->adam(lr, b1, b2, eps, step, gs); }
int hulc_optimizer_step(hulc_ctx* ctx, const hulc_optim* opt) {
    if (!ctx || !opt) { hulc_set_error("hulc_optimizer_step: null argument"); return 1; }
    return ctx->e->optim(*opt);
}
int hulc_comm_unique_id(void* out, int64_t cap) {
    if (!out || cap < 128) { hulc_set_error("hulc_comm_unique_id: need a 128-byte buffer"); return 1; }
    if (!GradComm::load_api()) return 1;
    GradComm::UniqueId id;
    const int rc = GradComm::api().get_id(&id);
    if (rc != 0) { hulc_set_error("ncclGetUniqueId failed: %s", GradComm::err(rc)); return 1; }
    memcpy(out, &id, 128);
    return 0;
}
int hulc_comm_prepare(hulc_ctx* ctx) {
    if (!ctx) { hulc_set_error("hulc_comm_prepare: null context"); return 1; }
    return ctx->e->comm_prepare();
}
int hulc_comm_init(hulc_ctx* ctx, const void* unique_id, int32_t rank, int32_t world) {
    if (!ctx || !unique_id || world < 1 || rank < 0 || rank >= world) { hulc_set_error("hulc_comm_init: bad argument"); return 1; }
    return ctx->e->comm_init(unique_id, rank, world);
}
int hulc_comm_destroy(hulc_ctx* ctx) { return ctx ? ctx->e->comm_destroy() : 0; }
int hulc_comm_buckets(hulc_ctx* ctx, int64_t* lo, int64_t* hi, int32_t cap) {
    if (!ctx || !lo || !hi) { hulc_set_error("hulc_comm_buckets: null argument"); return -1; }
    return ctx->e->comm_buckets(lo, hi, cap);
}
int hulc_comm_stats(hulc_ctx* ctx, int64_t* n_collectives, double* bytes) {
    if (!ctx || !ctx->e->comm) { hulc_set_error("hulc_comm_stats: no communicator"); return 1; }
    if (n_collectives) *n_collectives = ctx->e->comm->n_collectives;
    if (bytes) *bytes = ctx->e->comm->bytes_reduced;
    return 0;
}
int hulc_comm_size(hulc_ctx* ctx, int32_t* rank, int32_t* world) {
    if (!ctx || !ctx->e->comm) { hulc_set_error("hulc_comm_size: no communicator"); return 1; }
    int r = -1, w = -1;
    if (ctx->e->comm->size(&r, &w)) return 1;
    if (rank) *rank = r;
    if (world) *world = w;
    return 0;
}
int hulc_comm_timeline(hulc_ctx* ctx, double* out, int32_t cap_buckets, double* backward_us) {
    if (!ctx || !ctx->e->comm || !out) { hulc_set_error("hulc_comm_timeline: no communicator / null buffer"); return -1; }
    return ctx->e->comm->timeline(out, cap_buckets, backward_us, ctx->e->st);
}
int hulc_allreduce_grads(hulc_ctx* ctx, int32_t bucket_dtype) {
    if (!ctx) { hulc_set_error("hulc_allreduce_grads: null context"); return 1; }
    return ctx->e->allreduce_grads(bucket_dtype);
}
int hulc_backward_allreduce(hulc_ctx* ctx, int32_t bucket_dtype) {
    if (!ctx) { hulc_set_error("hulc_backward_allreduce: null context"); return 1; }
    return ctx->e->backward_allreduce(bucket_dtype);
}
int hulc_scaler_enable(hulc_ctx* ctx, float init_scale, float growth_factor, float backoff_factor, int32_t growth_interval) {
    if (!ctx) { hulc_set_error("hulc_scaler_enable: null context"); return 1; }
    return ctx->e->scaler_enable(init_scale, growth_factor, backoff_factor, growth_interval);
}
int hulc_scaler_get(hulc_ctx* ctx, float* scale, int32_t* growth_tracker, int64_t* skipped_steps, int32_t* last_found_inf, int64_t* taken_steps) {
    if (!ctx) { hulc_set_error("hulc_scaler_get: null context"); return 1; }
    return ctx->e->scaler_get(scale, growth_tracker, skipped_steps, last_found_inf, taken_steps);
}
int hulc_scaler_set(hulc_ctx* ctx, float scale, int32_t growth_tracker, int64_t taken_steps) {
    if (!ctx) { hulc_set_error("hulc_scaler_set: null context"); return 1; }
    return ctx->e->scaler_set(scale, growth_tracker, taken_steps);
}
int hulc_set_kl_beta(hulc_ctx* ctx, float b) { ctx->e->set_kl_beta(b); return 0; }
int hulc_set_dropout(hulc_ctx* ctx, float p) {
    if (p < 0.f || p >= 1.f) { hulc_set_error("hulc_set_dropout: p must be in [0,1)"); return 1; }
    ctx->e->set_dropout(p);
    return 0;
}
int hulc_set_option(hulc_ctx* ctx, const char* name, int64_t value) {
    if (!ctx) { hulc_set_error("hulc_set_option: null context"); return 1; }
    return ctx->e->set_option(name, (long long)value);
}
int hulc_get_option(hulc_ctx* ctx, const char* name, int64_t* value) {
    if (!ctx || !value) { hulc_set_error("hulc_get_option: null argument"); return 1; }
    long long v = 0;
    const int rc = ctx->e->get_option(name, &v);
    *value = (int64_t)v;
    return rc;
}
int hulc_timers_enable(hulc_ctx* ctx, int32_t on, const char* only_class) { ctx->e->set_timing(on != 0, only_class); return 0; }
int hulc_timers_read(hulc_ctx* ctx, char* json_out, int64_t cap, int32_t reset) { return ctx->e->timers_read(json_out, cap, reset != 0); }
int hulc_get_tensor(hulc_ctx* ctx, const char* name, float* out, int64_t cap, int64_t* n) { return ctx->e->get_tensor(name, out, cap, n); }
int hulc_get_plan_idx(hulc_ctx* ctx, int32_t* out, int64_t cap) { return ctx->e->get_plan_idx(out, cap); }

int hulc_k_gemm_nt(int32_t dtype, const void* A, const void* B, float* C, int32_t M, int32_t N, int32_t K, int64_t lda, int64_t ldb, int64_t ldc,
                   const float* bias, int32_t relu, void* stream) {
    if (dtype == HULC_DTYPE_F16) return hulc_f16::k_gemm_nt(0, A, B, C, M, N, K, lda, ldb, ldc, bias, relu, stream);
    return hulc_bf16::k_gemm_nt(dtype == HULC_DTYPE_F32, A, B, C, M, N, K, lda, ldb, ldc, bias, relu, stream);
}
int hulc_k_cast(int32_t dtype, const float* src, void* dst, int64_t n, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (dtype == HULC_DTYPE_F32) hipLaunchKernelGGL((cast_kernel<float, float>), dim3(1024), dim3(256), 0, st, src, (float*)dst, (long long)n);
    else hipLaunchKernelGGL((cast_kernel<float, h16_t>), dim3(1024), dim3(256), 0, st, src, (h16_t*)dst, (long long)n);
    if (hipGetLastError() != hipSuccess) { hulc_set_error("hulc_k_cast: launch failed"); return 1; }
    return 0;
}

int hulc_k_attention(int32_t variant, const float* qkv, float* P, float* ao, const float* dao, float* dqkv, int32_t B, int32_t S, float drop_p,
                     uint64_t seed, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int D = 128, NH = 8;
    if (S < 1 || S > 64 || (variant == 1 && S > 32)) { hulc_set_error("hulc_k_attention: S=%d not covered by variant %d", S, variant); return 1; }
    if (variant == 2) {                         // four waves per (b, head): the S <= 64 kernels the 16-bit engines run for S > 32
        if (!dao) hipLaunchKernelGGL((attention_fwd64_kernel<float>), dim3(B * NH), dim3(256), 0, st, qkv, B, S, D, NH, P, ao, drop_p, (unsigned long long)seed);
        else {
            hipFuncSetAttribute((const void*)attention_bwd64_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_BWD64_LDS);
            hipLaunchKernelGGL((attention_bwd64_kernel<float>), dim3(B * NH), dim3(256), ATT_BWD64_LDS, st, qkv, P, dao, B, S, D, NH, dqkv, drop_p, (unsigned long long)seed);
        }
        if (hipGetLastError() != hipSuccess) { hulc_set_error("hulc_k_attention: launch failed"); return 1; }
        return 0;
    }
    const dim3 grid(B * NH), block(64);
    if (!dao) {
        if (variant == 1) hipLaunchKernelGGL((attention_fwd32_kernel<float>), grid, block, 0, st, qkv, B, S, D, NH, P, ao, drop_p, (unsigned long long)seed);
        else if (S <= 32) hipLaunchKernelGGL((attention_fwd_kernel<float, 32>), grid, block, 0, st, qkv, B, S, D, NH, P, ao, drop_p, (unsigned long long)seed);
        else hipLaunchKernelGGL((attention_fwd_kernel<float, 64>), grid, block, 0, st, qkv, B, S, D, NH, P, ao, drop_p, (unsigned long long)seed);
    } else {
        if (variant == 1) hipLaunchKernelGGL((attention_bwd32_kernel<float>), grid, block, 0, st, qkv, P, dao, B, S, D, NH, dqkv, drop_p, (unsigned long long)seed);
        else if (S <= 32) hipLaunchKernelGGL((attention_bwd_kernel<float, 32>), grid, block, 0, st, qkv, P, dao, B, S, D, NH, dqkv, drop_p, (unsigned long long)seed);
        else hipLaunchKernelGGL((attention_bwd_kernel<float, 64>), grid, block, 0, st, qkv, P, dao, B, S, D, NH, dqkv, drop_p, (unsigned long long)seed);
    }
    if (hipGetLastError() != hipSuccess) { hulc_set_error("hulc_k_attention: launch failed"); return 1; }
    return 0;
}

// conv wgrad unit-test entry (bf16 NHWC): out[CO][KH*KW*CI] (packed (kh,kw,ci) order, fp32, overwritten)
int hulc_k_conv_wgrad(int32_t which, const void* X, const void* dY, float* out, int32_t Nf, int32_t IH, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    float* part = nullptr;
    static float* bias_tmp = nullptr;
    if (!bias_tmp && hipMalloc(&bias_tmp, sizeof(float) * 512 * 64) != hipSuccess) { hulc_set_error("hulc_k_conv_wgrad: hipMalloc failed"); return 1; }
    const int KC = which == 3 ? 576 : (which == 2 ? 512 : 192);
    const int CO = which == 1 ? 32 : 64;
    if (hipMalloc(&part, sizeof(float) * 512ll * 64 * KC) != hipSuccess) { hulc_set_error("hulc_k_conv_wgrad: hipMalloc failed"); return 1; }
    int ns;
    static h16_t* zp = nullptr;                   // zero page of the LDS-DMA kernel (pad slots, columns / rows beyond the frame)
    if (!zp) { if (hipMalloc(&zp, 256) != hipSuccess) { hulc_set_error("hulc_k_conv_wgrad: hipMalloc failed"); return 1; } hipMemset(zp, 0, 256); }
    if (which == 3) { const int OH = IH - 2; ns = launch_conv_wgrad_tr<64, 64, 3, 3, 1>(st, (const h16_t*)X, (const h16_t*)dY, part, bias_tmp, Nf, IH, IH, OH, OH, 512, nullptr, zp); }
    else if (which == 2) { const int OH = (IH - 4) / 2 + 1; ns = launch_conv_wgrad_tr<32, 64, 4, 4, 2>(st, (const h16_t*)X, (const h16_t*)dY, part, bias_tmp, Nf, IH, IH, OH, OH, 512, nullptr, zp); }
    else if (which == 1) { const int OH = (IH - 8) / 4 + 1; ns = launch_conv1_wgrad_tr(st, Conv1Src{X, nullptr, 0, 0}, (const h16_t*)dY, part, bias_tmp, Nf, IH, IH, OH, OH, 512); }
    else { hipFree(part); hulc_set_error("hulc_k_conv_wgrad: which must be 1, 2 or 3"); return 1; }
    hipMemsetAsync(out, 0, sizeof(float) * CO * KC, st);
    hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3((CO * KC + 1023) / 1024), dim3(256), 0, st, part, ns, (long long)CO * KC, out, CO, KC, 1, 1, 0);
    hipError_t e = hipStreamSynchronize(st);
    hipFree(part);
    if (e != hipSuccess) { hulc_set_error("hulc_k_conv_wgrad: %s", hipGetErrorString(e)); return 1; }
    return 0;
}

// host arithmetic only (no device needed): the interior group range the uint8 conv1 kernels convert without clamps (conv_wgrad.h::conv1_interior_groups)
int hulc_k_conv1_interior_groups(int32_t IW, int32_t pad, int32_t cap, int32_t* first, int32_t* count) {
    int cl = 0, wi = 0;
    conv1_interior_groups(IW, pad, cap, cl, wi);
    *first = cl; *count = wi;
    return 0;
}

// conv1's weight gradient from uint8 (Nf,IH,IH,3) frames alone (tests): dW (32,192) [torch (o, c, kh, kw) order] and db (32) of the frames after ScaleImageTensor / Normalize /
// RandomShiftsAug — form 0: raw rows through LDS (conv1_wgrad_tr2u_kernel), 1: conversion from the prefetch registers (conv1_wgrad_tr2r_kernel), 2: the same with interior / row-end slots; fold = Conv1Src::fold
int hulc_k_conv1_wgrad_u8(const void* X, const int32_t* shifts, int32_t pad, const void* dY, float* dw_out, float* db_out, int32_t Nf, int32_t IH, int32_t form, int32_t fold, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    float *part = nullptr, *bias = nullptr;
    if (hipMalloc(&part, sizeof(float) * 512ll * 32 * 192) != hipSuccess || hipMalloc(&bias, sizeof(float) * 64) != hipSuccess) { hulc_set_error("hulc_k_conv1_wgrad_u8: hipMalloc failed"); return 1; }
    hipMemsetAsync(bias, 0, sizeof(float) * 64, st);
    Conv1Src src{}; src.X = X; src.shift = shifts; src.u8 = 1; src.pad = pad; src.fold = fold != 0;
    const int OH = (IH - 8) / 4 + 1;
    const int keep = g_conv1_wgrad_u8reg;
    g_conv1_wgrad_u8reg = form;
    const int ns = launch_conv1_wgrad_tr(st, src, (const h16_t*)dY, part, bias, Nf, IH, IH, OH, OH, 512);
    g_conv1_wgrad_u8reg = keep;
    hipMemsetAsync(dw_out, 0, sizeof(float) * 32 * 192, st);
    hipLaunchKernelGGL(unpack_conv_wgrad_kernel, dim3((32 * 192 + 1023) / 1024), dim3(256), 0, st, part, ns, (long long)32 * 192, dw_out, 32, 192, 1, 1, 0);
    hipMemcpyAsync(db_out, bias, sizeof(float) * 32, hipMemcpyDeviceToDevice, st);
    hipError_t e = hipStreamSynchronize(st);
    hipFree(part); hipFree(bias);
    if (e != hipSuccess) { hulc_set_error("hulc_k_conv1_wgrad_u8: %s", hipGetErrorString(e)); return 1; }
    return 0;
}

// raw-tile conv kernels alone (bf16 NHWC): mode 0 fwd 3x3/s1 64->64, 1 fwd 4x4/s2 32->64, 2 dgrad of (0), 3 dgrad of (1)
int hulc_k_conv_tile(int32_t mode, const void* img, const void* w, const float* bias, const void* mask, void* out, int32_t Nf, int32_t IMH,
                     int32_t OUTH, int32_t relu, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    ConvTileP p{}; p.img = (const h16_t*)img; p.IMH = p.IMW = IMH; p.w = (const h16_t*)w; p.out = (h16_t*)out; p.OUTH = p.OUTW = OUTH;
    p.bias = bias; p.mask = (const h16_t*)mask; p.relu = relu & 1; p.dbg = relu & (30 | 64 | 128 | 256); p.Nf = Nf;
    // modes 7..9 = the production forms: 7 = mode 1 that also EMITS the ReLU bitmask of its output into `mask` (unsigned[Nf][OUTH][OUTW][2]);
    // 8 / 9 = modes 2 / 3 with `mask` = ReLU bitmask words (2 / 1 per output pixel) staged through LDS.  relu bit 5 (32): dynamic work claiming.
    if (mode == 7) { p.mask = nullptr; p.bits_out = (unsigned*)mask; mode = 1; }
    else if (mode == 8 || mode == 9) { p.mask = nullptr; p.maskbits = (const unsigned*)mask; mode -= 6; }
    if (relu & 32) {
        static int* ctr = nullptr;
        if (!ctr && hipMalloc(&ctr, 256) != hipSuccess) { hulc_set_error("hulc_k_conv_tile: hipMalloc failed"); return 1; }
        hipMemsetAsync(ctr, 0, 256, st);
        p.work_ctr = ctr;
    }
    {   // conv_reg.h pipelined-epilogue forms (modes 28 / 29 and what falls back from them): a scratch page for the stores of pixels that must not be written
        static void* dp = nullptr;
        if (!dp && hipMalloc(&dp, 8192) != hipSuccess) { hulc_set_error("hulc_k_conv_tile: hipMalloc failed"); return 1; }
        p.dump = (h16_t*)dp;
    }
    bool ok = false;
    // modes 10 / 11 / 17: modes 0 / 1 / 7 on the weights-in-registers kernels (conv_reg.h); 12 / 18: modes 2 / 8 (conv3 data gradient, 16-bit mask / bitmask)
    if (mode == 17) { p.mask = nullptr; p.bits_out = (unsigned*)mask; mode = 11; }
    if (mode == 18) { p.mask = nullptr; p.maskbits = (const unsigned*)mask; mode = 12; }
    if (mode == 19) { p.mask = nullptr; p.maskbits = (const unsigned*)mask; mode = 13; }      // 13 / 19: modes 3 / 9 (conv2 data gradient, four parity classes)
    if (mode == 12 || mode == 13) {
        static void* zp = nullptr;
        if (!zp) { if (hipMalloc(&zp, 256) != hipSuccess) { hulc_set_error("hulc_k_conv_tile: hipMalloc failed"); return 1; } hipMemset(zp, 0, 256); }
        p.zeros = (const h16_t*)zp;
    }
    // modes 20 / 21 / 27 / 22 / 28 / 23 / 29: modes 10 .. 19 in the form with two 256-thread workgroups per CU (conv_reg.h, NWV = 4)
    if (mode == 27) { p.mask = nullptr; p.bits_out = (unsigned*)mask; mode = 21; }
    if (mode == 28) { p.mask = nullptr; p.maskbits = (const unsigned*)mask; mode = 22; }
    if (mode == 29) { p.mask = nullptr; p.maskbits = (const unsigned*)mask; mode = 23; }
    if (mode == 22 || mode == 23) {
        static void* zp2 = nullptr;
        if (!zp2) { if (hipMalloc(&zp2, 256) != hipSuccess) { hulc_set_error("hulc_k_conv_tile: hipMalloc failed"); return 1; } hipMemset(zp2, 0, 256); }
        p.zeros = (const h16_t*)zp2;
    }
    // modes 30 .. 39: the two-workgroup form with TWO (smaller) band buffers per workgroup
    if (mode == 37) { p.mask = nullptr; p.bits_out = (unsigned*)mask; mode = 31; }
    if (mode == 38) { p.mask = nullptr; p.maskbits = (const unsigned*)mask; mode = 32; }
    if (mode == 39) { p.mask = nullptr; p.maskbits = (const unsigned*)mask; mode = 33; }
    if (mode == 32 || mode == 33) {
        static void* zp3 = nullptr;
        if (!zp3) { if (hipMalloc(&zp3, 256) != hipSuccess) { hulc_set_error("hulc_k_conv_tile: hipMalloc failed"); return 1; } hipMemset(zp3, 0, 256); }
        p.zeros = (const h16_t*)zp3;
    }
    if (mode == 30) ok = launch_conv_reg<64, 3, 3, 1, false, 1, 4, 2>(st, p);
    else if (mode == 31) ok = launch_conv_reg<32, 4, 4, 2, false, 1, 4, 2>(st, p);
    else if (mode == 32) ok = launch_conv_reg<64, 3, 3, 1, true, 1, 4, 2>(st, p);
    else if (mode == 33) ok = launch_conv_reg<64, 2, 2, 1, true, 2, 4, 2>(st, p);
    else if (mode == 20) ok = launch_conv_reg<64, 3, 3, 1, false, 1, 4>(st, p);
    else if (mode == 21) ok = launch_conv_reg<32, 4, 4, 2, false, 1, 4, 0, true>(st, p);      // the production form for small maps: slot decode in registers (conv_reg.h PKR)
    else if (mode == 22) ok = launch_conv_reg<64, 3, 3, 1, true, 1, 4, 0, true, 1>(st, p);    // the production forms: PKR + pipelined epilogue (EPI; 16-bit mask values
    else if (mode == 23) ok = launch_conv_reg<64, 2, 2, 1, true, 2, 4, 0, true, 1>(st, p);    // = modes 22 / 23 fall back to the pair form inside launch_conv_reg)
    else if (mode == 10) ok = launch_conv_reg_fwd<64, 3, 3, 1>(st, p);
    else if (mode == 11) ok = launch_conv_reg_fwd<32, 4, 4, 2>(st, p);
    else if (mode == 12) ok = launch_conv_reg<64, 3, 3, 1, true>(st, p);
    else if (mode == 13) ok = launch_conv_reg<64, 2, 2, 1, true, 2>(st, p);
    else if (mode == 0) ok = launch_conv_tile<64, 64, 3, 3, 1, 1, false>(st, p);
    else if (mode == 1) ok = launch_conv_tile<32, 64, 4, 4, 2, 1, false>(st, p);
    else if (mode == 2) ok = launch_conv_tile<64, 64, 3, 3, 1, 1, true>(st, p);
    else if (mode == 3) ok = launch_conv_tile<64, 32, 2, 2, 1, 2, true>(st, p);
    else if (mode == 4) { launch_conv1_fwd(st, Conv1Src{img, nullptr, 0, 0}, (const h16_t*)w, bias, (h16_t*)out, Nf, IMH, IMH, OUTH, OUTH, relu & ~1); ok = true; }
    else if (mode == 5 || mode == 6) {      // conv1 forward from uint8 NHWC frames; mode 6: `mask` = (Nf,2) int32 RandomShiftsAug shifts, pad 10 (IMH >= 100) or 4
        launch_conv1_fwd(st, Conv1Src{img, mode == 6 ? (const int*)mask : nullptr, 1, IMH >= 100 ? 10 : 4}, (const h16_t*)w, bias, (h16_t*)out, Nf, IMH, IMH, OUTH,
                         OUTH, relu & ~1);
        ok = true;
    }
    if (!ok) { hulc_set_error("hulc_k_conv_tile: unsupported mode/shape"); return 1; }
    if (hipGetLastError() != hipSuccess) { hulc_set_error("hulc_k_conv_tile: launch failed"); return 1; }
    return 0;
}

// skinny GEMM alone (bf16): out[M][N] (bf16) = A[M][K] W[N][K]^T ; variant = NW*10 + MT (e.g. 82 = 8 waves, 32-row blocks)
int hulc_k_skinny(const void* A, const void* W, void* out, int32_t M, int32_t N, int32_t K, int32_t variant, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    EpiP ep; ep.out = out; ep.out_f32 = 0;
    const int NW = variant / 10, MT = variant % 10;
    dim3 grid(N / 16, MT > 0 ? (M + MT * 16 - 1) / (MT * 16) : 1);
    const h16_t* a = (const h16_t*)A; const h16_t* w = (const h16_t*)W;
#define SK(nw, mt) hipLaunchKernelGGL((skinny_gemm_kernel<nw, mt>), grid, dim3(nw * 64), 0, st, a, (long long)K, w, (long long)K, M, N, K, dense_out(N), ep)
    if (NW == 8 && MT == 2) SK(8, 2); else if (NW == 8 && MT == 4) SK(8, 4); else if (NW == 4 && MT == 2) SK(4, 2); else if (NW == 4 && MT == 4) SK(4, 4);
    else if (NW == 8 && MT == 1) SK(8, 1); else if (NW == 4 && MT == 1) SK(4, 1);
    else if (NW == 16 && MT == 2) SK(16, 2); else if (NW == 16 && MT == 4) SK(16, 4);
    else if (NW == 9 && MT == 2) hipLaunchKernelGGL((skinny_gemm_kernel<8, 2, 8>), grid, dim3(512), 0, st, a, (long long)K, w, (long long)K, M, N, K, dense_out(N), ep);
    else if (NW == 17 && MT == 2) hipLaunchKernelGGL((skinny_gemm_kernel<16, 2, 4>), grid, dim3(1024), 0, st, a, (long long)K, w, (long long)K, M, N, K, dense_out(N), ep);
    else if (NW == 20) { if (!launch_skinny_lds(st, a, (long long)K, w, (long long)K, M, N, K, MT, dense_out(N), ep)) { hulc_set_error("hulc_k_skinny: shape not covered by the LDS kernel"); return 1; } }
    else if (NW == 30) launch_skinny(st, a, (long long)K, w, (long long)K, M, N, K, dense_out(N), ep);       // the production router (incl. the K-chunked LDS kernel)
    else if (NW == 40) {      // two independent problems in one launch (the paired directions of a bidirectional recurrence): the second one lives at A + M*K, W + N*K, out + M*N
        EpiP ep2 = ep; ep2.out = (h16_t*)out + (long long)M * N;
        if (!launch_skinny_lds_dual(st, a, w, ep, a + (long long)M * K, w + (long long)N * K, ep2, (long long)K, (long long)K, M, N, K, dense_out(N))) {
            hulc_set_error("hulc_k_skinny: shape not covered by the dual launch"); return 1; }
    }
    else if (NW == 50 || NW == 51) {      // fragment-ordered weight copies (gemm.h: frag_pack_kernel).  500: W repacked into a scratch copy, then the K-chunked LDS kernel reads
        // it with ldw == 0 (K = n x 2048, M <= 64: the GRU's BPTT step);  510: the repacked W itself lands in `out` (N x K elements; K % 128 == 0, N % 16 == 0)
        if ((N % 16) != 0 || (K % 128) != 0) { hulc_set_error("hulc_k_skinny: fragment order needs N %% 16 == 0 and K %% 128 == 0"); return 1; }
        h16_t* wf = (h16_t*)out;
        if (NW == 50 && hipMalloc((void**)&wf, (size_t)N * K * sizeof(h16_t)) != hipSuccess) { hulc_set_error("hulc_k_skinny: scratch allocation failed"); return 1; }
        FragPackBatch fb{};
        fb.src[0] = w; fb.dst[0] = wf; fb.N[0] = N; fb.K[0] = K; fb.blk0[0] = 0; fb.blk0[1] = frag_pack_blocks(N, K); fb.n = 1;
        hipLaunchKernelGGL(frag_pack_kernel, dim3(fb.blk0[1]), dim3(256), 0, st, fb);
        if (NW == 50) {
            const bool ok = launch_skinny_lds_kchunk(st, a, (long long)K, wf, 0ll, M, N, K, dense_out(N), ep);
            hipStreamSynchronize(st);
            hipFree(wf);
            if (!ok) { hulc_set_error("hulc_k_skinny: shape not covered by the K-chunked kernel"); return 1; }
        }
    }
    else { hulc_set_error("hulc_k_skinny: unsupported variant"); return 1; }
#undef SK
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// one whole recurrence as a persistent launch (rnn_persist.h), bf16: X [S][B][2048] with X[q0] given; flags = RP_FLAG_WORDS zeroed uint32 words
// (reused across calls with increasing launch_index >= 1), err = one zeroed uint32
int hulc_k_rnn_persist(void* X, const void* W, const void* res, const void* mask, int32_t B, int32_t S, int32_t q0, int32_t dq, int32_t act, uint32_t* flags,
                       uint32_t* err, uint32_t launch_index, void* stream) {
    RnnPersistP p{};
    p.X = (h16_t*)X; p.W = (const h16_t*)W; p.res = (const h16_t*)res; p.mask = (const h16_t*)mask; p.B = B; p.S = S; p.q0 = q0; p.dq = dq; p.act = act;
    p.flags = flags; p.base = launch_index << 12; p.parity = (int)(launch_index & 1u); p.err = err; p.stamps = nullptr;
    if (!launch_rnn_persist((hipStream_t)stream, p)) { hulc_set_error("hulc_k_rnn_persist: shape not covered (S >= 2, B <= 128)"); return 1; }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int32_t hulc_k_rnn_persist_flag_words(void) { return RP_FLAG_WORDS; }

// probe of ds_read_b64_tr_b16 semantics (tests/tools only): LDS image lds[i] = i (uint16), lane l reads at element index addr[l]
__global__ void trread_probe_kernel(const int* __restrict__ addr_in, unsigned short* __restrict__ out) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int a = addr_in[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + a));
#pragma unroll
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int hulc_k_trread_probe(const int32_t* addr, uint16_t* out, void* stream) {
    hipLaunchKernelGGL(trread_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const int*)addr, (unsigned short*)out);
    if (hipGetLastError() != hipSuccess) { hulc_set_error("trread probe launch failed"); return 1; }
    return 0;
}

}  // extern "C"
