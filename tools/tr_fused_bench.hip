// NOTE (round 5): written against the round-4 headers.  It no longer builds / runs against the current ones (the fused kernels read fragment-ordered weights since round 4's
// last day; the conv_tile stamp build faults): kept for the profiles of rounds 2 - 4 it produced (profiles/r0[234]_*), not part of tools/refresh_profiles.sh any more.
// tools/tr_fused_bench.hip — timing + phase stamps of the fused transformer encoder layer forward (hulc_amd/csrc/tr_fused.h) on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTRF_STAMPS tools/tr_fused_bench.hip -o tools/bin/tr_fused_bench && tools/bin/tr_fused_bench
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
void hulc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
#include "../hulc_amd/csrc/tr_fused.h"
using namespace hulc_bf16;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
template <typename U> U* dalloc(size_t n, int fill) { U* p; CHECK(hipMalloc(&p, n * sizeof(U))); CHECK(hipMemset(p, fill, n * sizeof(U))); return p; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, S = argc > 2 ? atoi(argv[2]) : 32; const size_t N = (size_t)B * S;
    TrLayerP q{};
    q.xin = dalloc<float>(N * 128, 0); q.ln_in = 0; q.ln_g = dalloc<float>(128, 0); q.ln_b = dalloc<float>(128, 0);
    q.xf_out = dalloc<float>(N * 128, 0); q.xt_out = dalloc<h16_t>(N * 128, 0); q.st_out = dalloc<float>(N * 2, 0);
    q.Wqkv = dalloc<h16_t>(384 * 128, 0x3c); q.Wo = dalloc<h16_t>(128 * 128, 0x3c); q.W1 = dalloc<h16_t>(2048 * 128, 0x3c); q.W2 = dalloc<h16_t>(128 * 2048, 0x3c);
    q.bqkv = dalloc<float>(384, 0); q.bo = dalloc<float>(128, 0); q.b1 = dalloc<float>(2048, 0); q.b2 = dalloc<float>(128, 0); q.n1g = dalloc<float>(128, 0); q.n1b = dalloc<float>(128, 0);
    q.qkv = dalloc<h16_t>(N * 384, 0); q.Pat = dalloc<float>((size_t)B * 8 * S * S, 0); q.ao = dalloc<h16_t>(N * 128, 0); q.y1 = dalloc<float>(N * 128, 0); q.st1 = dalloc<float>(N * 2, 0);
    q.x1t = dalloc<h16_t>(N * 128, 0); q.x1f = dalloc<float>(N * 128, 0); q.hff = dalloc<h16_t>(N * 2048, 0); q.y2 = dalloc<float>(N * 128, 0);
    q.B = B; q.S = S; q.dp = 0.1f; q.seed_att = 1; q.seed_o = 2; q.seed_h = 3; q.seed_y = 4; q.stamps = nullptr;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 50; ++i) launch_tr_layer_fwd(0, q);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%.2f us per launch\n", ms * 1000.f / 50);
    }
    long long* st = dalloc<long long>(16, 0); q.stamps = st;
    launch_tr_layer_fwd(0, q); CHECK(hipDeviceSynchronize());
    long long h[16]; CHECK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    const char* nm[8] = {"start", "x staged", "qkv", "attention", "out-proj", "LN1", "FFN1", "FFN2+atomics"};
    for (int i = 1; i < 8; ++i) printf("%-14s +%6lld cycles (at %6lld)\n", nm[i], h[i] - h[i - 1], h[i] - h[0]);
    return 0;
}
