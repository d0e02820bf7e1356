// tools/rnn_persist_bench.hip — standalone check + timing of hulc_amd/csrc/rnn_persist.h (whole recurrence in one persistent launch).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rnn_persist_bench.hip -o tools/bin/rnn_persist_bench && tools/bin/rnn_persist_bench [B] [S]
// Forward (ReLU, residual Zx) and backward (mask H, residual dH) forms against a CPU recurrence that rounds every state to bf16 like the
// kernel does; then the time per step from 50 back-to-back launches (HIP events).
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <cstring>
void hulc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#include "../hulc_amd/csrc/rnn_persist.h"
using namespace hulc_bf16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float b2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng = 12345u;
static float urand() { rng = rng * 1664525u + 1013904223u; return ((rng >> 8) + 0.5f) / 16777216.f * 2.f - 1.f; }

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, S = argc > 2 ? atoi(argv[2]) : 32, H = RP_HID;
    const size_t nx = (size_t)S * B * H;
    std::vector<uint16_t> W((size_t)H * H), X(nx), R(nx), M(nx);
    for (auto& w : W) w = f2b(urand() * 0.03f);
    for (auto& r : R) r = f2b(urand());
    for (auto& m : M) m = f2b(urand());
    uint16_t *dW, *dX, *dR, *dM; unsigned *flags, *err;
    CHECK(hipMalloc(&dW, W.size() * 2)); CHECK(hipMalloc(&dX, nx * 2)); CHECK(hipMalloc(&dR, nx * 2)); CHECK(hipMalloc(&dM, nx * 2));
    CHECK(hipMalloc(&flags, RP_FLAG_WORDS * 4)); CHECK(hipMalloc(&err, 4));
    CHECK(hipMemset(flags, 0, RP_FLAG_WORDS * 4)); CHECK(hipMemset(err, 0, 4));
    CHECK(hipMemcpy(dW, W.data(), W.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dR, R.data(), nx * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dM, M.data(), nx * 2, hipMemcpyHostToDevice));
    unsigned base = 4096;
    std::vector<float> Wf(W.size());
    for (size_t i = 0; i < W.size(); ++i) Wf[i] = b2f(W[i]);
    for (int mode = 0; mode < 2; ++mode) {       // 0: forward relu, q ascending;  1: backward mask, q descending
        for (size_t i = 0; i < nx; ++i) X[i] = f2b(0.f);
        const int q0 = mode ? S - 1 : 0, dq = mode ? -1 : 1;
        for (int b = 0; b < B; ++b) for (int k = 0; k < H; ++k) X[((size_t)q0 * B + b) * H + k] = f2b(mode ? urand() : fmaxf(urand(), 0.f));
        CHECK(hipMemcpy(dX, X.data(), nx * 2, hipMemcpyHostToDevice));
        RnnPersistP p{}; p.X = dX; p.W = dW; p.res = dR; p.mask = mode ? dM : nullptr; p.B = B; p.S = S; p.q0 = q0; p.dq = dq; p.act = 1; p.flags = flags; p.base = base; p.err = err; p.parity = (base >> 12) & 1;
        base += 4096;
        if (!launch_rnn_persist(0, p)) { printf("shape not covered\n"); return 1; }
        CHECK(hipDeviceSynchronize());
        std::vector<uint16_t> G(nx);
        CHECK(hipMemcpy(G.data(), dX, nx * 2, hipMemcpyDeviceToHost));
        unsigned herr; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        // CPU: a few windows only (2048^2 per window-step)
        double worst = 0; int bad = 0;
        for (int b : {0, B / 2, B - 1, 7 % B, 8 % B}) {
            std::vector<float> h(H), hn(H);
            for (int k = 0; k < H; ++k) h[k] = b2f(X[((size_t)q0 * B + b) * H + k]);
            for (int s = 1; s < S; ++s) {
                const size_t o = ((size_t)(q0 + s * dq) * B + b) * H;
                double num = 0, den = 0;
                for (int n = 0; n < H; ++n) {
                    double a = 0; const float* wr = &Wf[(size_t)n * H];
                    for (int k = 0; k < H; ++k) a += (double)h[k] * wr[k];
                    float v = (float)a + b2f(R[o + n]);
                    if (mode) v = b2f(M[o + n]) > 0.f ? v : 0.f; else v = fmaxf(v, 0.f);
                    hn[n] = b2f(f2b(v));
                    const double d = b2f(G[o + n]) - hn[n];
                    num += d * d; den += (double)hn[n] * hn[n];
                }
                const double rel = sqrt(num / (den + 1e-30));
                if (rel > worst) worst = rel;
                if (rel > 2e-2) ++bad;
                // continue from the GPU's own state so that rounding differences do not compound through the chain
                for (int n = 0; n < H; ++n) h[n] = b2f(G[o + n]);
            }
        }
        printf("mode %d  B=%d S=%d  worst per-step rel-L2 %.3e  bad %d  err %u\n", mode, B, S, worst, bad, herr);
    }
    // timing (forward form)
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    RnnPersistP p{}; p.X = dX; p.W = dW; p.res = dR; p.B = B; p.S = S; p.q0 = 0; p.dq = 1; p.act = 1; p.flags = flags; p.err = err;
    for (int rep = 0; rep < 3; ++rep) {
        const int L = 50;
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < L; ++i) { p.base = base; p.parity = (base >> 12) & 1; base += 4096; launch_rnn_persist(0, p); }
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%.2f us per launch, %.3f us per step (S-1 = %d steps)\n", ms * 1000.f / L, ms * 1000.f / L / (S - 1), S - 1);
    }
    unsigned herr; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("err %u\n", herr);
#ifdef RP_STAMPS
    {
        long long* st; CHECK(hipMalloc(&st, (size_t)S * 16 * 8)); CHECK(hipMemset(st, 0, (size_t)S * 16 * 8));
        p.stamps = st; p.base = base; p.parity = (base >> 12) & 1; base += 4096; launch_rnn_persist(0, p); CHECK(hipDeviceSynchronize());
        std::vector<long long> h((size_t)S * 16); CHECK(hipMemcpy(h.data(), st, (size_t)S * 16 * 8, hipMemcpyDeviceToHost));
        printf("shader-clock stamps of workgroup 8 (cycles since the step's loop top of wave 0; wave 0 | wave 5): top poll loaded mfma+lds barrier epilogue drained\n");
        for (int s = 2; s < S; s += (S > 12 ? 4 : 1)) {
            const long long z = h[(size_t)(s * 2) * 8];
            printf("step %2d  w0:", s); for (int i = 0; i < 7; ++i) printf(" %6lld", h[(size_t)(s * 2) * 8 + i] - z);
            printf("   w5:"); for (int i = 0; i < 5; ++i) printf(" %6lld", h[(size_t)(s * 2 + 1) * 8 + i] - z);
            printf("   next top %6lld\n", s + 1 < S ? h[(size_t)((s + 1) * 2) * 8] - z : 0);
        }
    }
#endif
    return 0;
}
