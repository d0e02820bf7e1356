// hulc_amd/csrc/sbert.h — MiniLM sentence encoder, forward only, fp32.
//
// The reference's SBert (hulc/models/encoders/language_network.py:8-17) wraps sentence_transformers' "all-MiniLM-L6-v2"
// (conf/model/sbert.yaml:2): a BERT encoder (6 layers, hidden 384, 12 heads of 32, FFN 1536, GELU(erf), post-LN eps 1e-12,
// learned position + token-type embeddings) followed by attention-masked mean pooling and L2 normalisation.  It produces the
// 384-d `lang` embeddings the training step consumes (hulc/models/hulc.py:440) and the goal embedding of a language rollout
// (:855-858): B = 1..34 sentences of <= 128 tokens — a latency-sized workload, so everything runs in fp32 on the existing
// MFMA-f32 GEMM core (gemm_kernel<float>) with small wave-per-row kernels around it.  Weights are bound like the step's:
// one flat fp32 device buffer + (Hugging Face state_dict name -> offset) pairs.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "../../include/hulc_hip.h"
#include "gemm.h"
#include "kernels.h"

namespace HULC_NS {

// x[t][:] = LayerNorm(word[ids[t]] + pos[t % L] + type[0])       one wave per token, H % 64 == 0, H <= 1024
__global__ void __launch_bounds__(256) sbert_embed_ln_kernel(const int* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                                             const float* __restrict__ type0, const float* __restrict__ g, const float* __restrict__ b, int T,
                                                             int L, int H, int vocab, float eps, float* __restrict__ out) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    const int id = min(max(ids[t], 0), vocab - 1);
    const float* w = word + (long long)id * H;
    const float* p = pos + (long long)(t % L) * H;
    float v[16];
    float s = 0.f;
    const int n = H >> 6;
    for (int k = 0; k < n; ++k) { v[k] = w[lane + 64 * k] + p[lane + 64 * k] + type0[lane + 64 * k]; s += v[k]; }
    const float mean = wave_sum(s) / H;
    float q = 0.f;
    for (int k = 0; k < n; ++k) { v[k] -= mean; q += v[k] * v[k]; }
    const float rstd = rsqrtf(wave_sum(q) / H + eps);
    for (int k = 0; k < n; ++k) out[(long long)t * H + lane + 64 * k] = v[k] * rstd * g[lane + 64 * k] + b[lane + 64 * k];
}
// out[t][:] = LayerNorm(x[t][:])   (x already holds dense + bias + residual)
__global__ void __launch_bounds__(256) sbert_ln_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b, int T, int H,
                                                       float eps, float* __restrict__ out) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    float v[16];
    float s = 0.f;
    const int n = H >> 6;
    for (int k = 0; k < n; ++k) { v[k] = x[(long long)t * H + lane + 64 * k]; s += v[k]; }
    const float mean = wave_sum(s) / H;
    float q = 0.f;
    for (int k = 0; k < n; ++k) { v[k] -= mean; q += v[k] * v[k]; }
    const float rstd = rsqrtf(wave_sum(q) / H + eps);
    for (int k = 0; k < n; ++k) out[(long long)t * H + lane + 64 * k] = v[k] * rstd * g[lane + 64 * k] + b[lane + 64 * k];
}
// self-attention of one (sentence b, head h, query i) per wave: lanes = keys (two per lane up to L = 128), padded keys masked out
// like transformers' additive -inf bias; qkv [T][3H] (q | k | v), ctx [T][H].  Head dim D <= 64.
__global__ void __launch_bounds__(256) sbert_attention_kernel(const float* __restrict__ qkv, const int* __restrict__ mask, int B, int L, int H, int NH,
                                                              float* __restrict__ ctx) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= B * NH * L) return;
    const int i = w % L, h = (w / L) % NH, b = w / (L * NH);
    const int D = H / NH;
    const float scale = rsqrtf((float)D);
    const float* q = qkv + ((long long)(b * L + i) * 3 * H) + h * D;
    float sc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int j = lane + 64 * u;
        float s = -INFINITY;
        if (j < L && mask[b * L + j] != 0) {
            const float* k = qkv + ((long long)(b * L + j) * 3 * H) + H + h * D;
            s = 0.f;
            for (int d = 0; d < D; ++d) s += q[d] * k[d];
            s *= scale;
        }
        sc[u] = s;
    }
    const float m = wave_max(fmaxf(sc[0], sc[1]));
    const float e0 = sc[0] == -INFINITY ? 0.f : __expf(sc[0] - m), e1 = sc[1] == -INFINITY ? 0.f : __expf(sc[1] - m);
    const float inv = 1.f / wave_sum(e0 + e1);
    float* o = ctx + (long long)(b * L + i) * H + h * D;
    for (int d = 0; d < D; ++d) {
        float a = 0.f;
        if (lane < L) a += e0 * qkv[((long long)(b * L + lane) * 3 * H) + 2 * H + h * D + d];
        if (lane + 64 < L) a += e1 * qkv[((long long)(b * L + lane + 64) * 3 * H) + 2 * H + h * D + d];
        a = wave_sum(a);
        if (lane == 0) o[d] = a * inv;
    }
}
__global__ void sbert_gelu_kernel(float* __restrict__ x, long long n) {      // transformers "gelu": 0.5 x (1 + erf(x / sqrt 2))
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; x[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
}
// sentence embedding: mean over unmasked tokens (sentence_transformers Pooling, mean mode, clamp(min=1e-9)), then L2 normalise
// (Normalize module of all-MiniLM-L6-v2, eps 1e-12); one wave per sentence
__global__ void __launch_bounds__(256) sbert_pool_kernel(const float* __restrict__ x, const int* __restrict__ mask, int B, int L, int H, int normalize,
                                                         float* __restrict__ out) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    float cnt = 0.f;
    for (int j = 0; j < L; ++j) cnt += mask[b * L + j] != 0 ? 1.f : 0.f;
    cnt = fmaxf(cnt, 1e-9f);
    float v[16], q = 0.f;
    const int n = H >> 6;
    for (int k = 0; k < n; ++k) {
        float s = 0.f;
        for (int j = 0; j < L; ++j)
            if (mask[b * L + j] != 0) s += x[(long long)(b * L + j) * H + lane + 64 * k];
        v[k] = s / cnt; q += v[k] * v[k];
    }
    const float nrm = normalize ? fmaxf(sqrtf(wave_sum(q)), 1e-12f) : 1.f;
    for (int k = 0; k < n; ++k) out[(long long)b * H + lane + 64 * k] = v[k] / nrm;
}

}  // namespace HULC_NS

using namespace HULC_NS;   // sbert.h is included by capi.hip only (fp32: lives in the bf16/fp32 translation unit)
struct hulc_sbert {
    hulc_sbert_config cfg;
    hipStream_t st = nullptr;
    const float* P = nullptr;
    std::map<std::string, const float*> w;
    float *x = nullptr, *x1 = nullptr, *qkv = nullptr, *ctx = nullptr, *y = nullptr, *ff = nullptr, *emb_out = nullptr;
    int *ids = nullptr, *mask = nullptr;
    std::vector<void*> allocs;
    bool bound = false;
    template <typename U> U* alloc(long long n) {
        void* p = nullptr;
        if (hipMalloc(&p, n * sizeof(U) + 256) != hipSuccess) return nullptr;
        allocs.push_back(p);
        return (U*)p;
    }
    ~hulc_sbert() { for (void* p : allocs) hipFree(p); }
    bool init() {
        const long long T = (long long)cfg.max_sentences * cfg.max_tokens, H = cfg.hidden;
        x = alloc<float>(T * H); x1 = alloc<float>(T * H); qkv = alloc<float>(T * 3 * H); ctx = alloc<float>(T * H); y = alloc<float>(T * H);
        ff = alloc<float>(T * cfg.intermediate); emb_out = alloc<float>((long long)cfg.max_sentences * H);
        ids = alloc<int>(T); mask = alloc<int>(T);
        return x && x1 && qkv && ctx && y && ff && emb_out && ids && mask;
    }
    const float* get(const std::string& n) const {
        auto it = w.find(n);
        return it == w.end() ? nullptr : it->second;
    }
    // Y[T][N] = X[T][K] W[N][K]^T + bias (+ residual)
    void linear(const float* X, int T, int K, const float* W, const float* bias, int N, float* Y, long long ldy, const float* res) {
        EpiP ep; ep.out = Y; ep.out_f32 = 1; ep.bias = bias;
        if (res) { ep.res = res; ep.res_f32 = 1; ep.res_ld = N; }
        launch_gemm<float, 64, 64>(st, dense<float>(X, T, K), dense<float>(W, N, K), dense_out(ldy), ep, T, N, K);
    }
    int encode(const int32_t* ids_in, const int32_t* mask_in, int B, int L, float* out) {
        if (!bound) { hulc_set_error("hulc_sbert_encode before hulc_sbert_bind"); return 1; }
        if (B < 1 || L < 1 || B > cfg.max_sentences || L > cfg.max_tokens || L > 128 || L > cfg.max_position) {
            hulc_set_error("hulc_sbert_encode: batch (%d sentences x %d tokens) exceeds the context (%d x %d, <= 128 tokens)", B, L, cfg.max_sentences, cfg.max_tokens);
            return 1;
        }
        const int T = B * L, H = cfg.hidden, I = cfg.intermediate;
        HIP_CHECK(hipMemcpyAsync(ids, ids_in, sizeof(int) * T, hipMemcpyDefault, st));
        HIP_CHECK(hipMemcpyAsync(mask, mask_in, sizeof(int) * T, hipMemcpyDefault, st));
        const std::string e = "embeddings.";
        hipLaunchKernelGGL(sbert_embed_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, st, ids, get(e + "word_embeddings.weight"), get(e + "position_embeddings.weight"),
                           get(e + "token_type_embeddings.weight"), get(e + "LayerNorm.weight"), get(e + "LayerNorm.bias"), T, L, H, cfg.vocab, cfg.ln_eps, x);
        for (int l = 0; l < cfg.layers; ++l) {
            const std::string p = "encoder.layer." + std::to_string(l) + ".";
            const char* nm[3] = {"query", "key", "value"};
            for (int k = 0; k < 3; ++k)
                linear(x, T, H, get(p + "attention.self." + nm[k] + ".weight"), get(p + "attention.self." + nm[k] + ".bias"), H, qkv + k * H, 3 * H, nullptr);
            hipLaunchKernelGGL(sbert_attention_kernel, dim3((B * cfg.heads * L + 3) / 4), dim3(256), 0, st, qkv, mask, B, L, H, cfg.heads, ctx);
            linear(ctx, T, H, get(p + "attention.output.dense.weight"), get(p + "attention.output.dense.bias"), H, y, H, x);
            hipLaunchKernelGGL(sbert_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, st, y, get(p + "attention.output.LayerNorm.weight"),
                               get(p + "attention.output.LayerNorm.bias"), T, H, cfg.ln_eps, x1);
            linear(x1, T, H, get(p + "intermediate.dense.weight"), get(p + "intermediate.dense.bias"), I, ff, I, nullptr);
            hipLaunchKernelGGL(sbert_gelu_kernel, dim3((unsigned)(((long long)T * I + 255) / 256)), dim3(256), 0, st, ff, (long long)T * I);
            linear(ff, T, I, get(p + "output.dense.weight"), get(p + "output.dense.bias"), H, y, H, x1);
            hipLaunchKernelGGL(sbert_ln_kernel, dim3((T + 3) / 4), dim3(256), 0, st, y, get(p + "output.LayerNorm.weight"), get(p + "output.LayerNorm.bias"), T, H,
                               cfg.ln_eps, x);
        }
        hipLaunchKernelGGL(sbert_pool_kernel, dim3((B + 3) / 4), dim3(256), 0, st, x, mask, B, L, H, cfg.normalize, emb_out);
        HIP_CHECK(hipMemcpyAsync(out, emb_out, sizeof(float) * B * H, hipMemcpyDefault, st));
        HIP_CHECK(hipStreamSynchronize(st));
        if (hipGetLastError() != hipSuccess) { hulc_set_error("kernel launch failed in hulc_sbert_encode"); return 1; }
        return 0;
    }
};
