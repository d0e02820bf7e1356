// tools only: phase timing (shader-clock stamps) of one skinny_lds_kernel launch inside a dependent chain.
#define HULC_KERNEL_STAMPS 1
#include "../hulc_amd/csrc/gemm.h"
#include <cstdio>
#include <vector>
#include <algorithm>
int main() {
    const int M = 64, N = 2048, K = 2048;
    h16_t *A, *W, *O;
    hipMalloc(&A, (size_t)33 * M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&O, (size_t)M * N * 2);
    hipMemset(A, 0, (size_t)33 * M * K * 2); hipMemset(W, 0, (size_t)N * K * 2);
    EpiP ep; 
    for (int rep = 0; rep < 3; ++rep)
        for (int t = 0; t < 32; ++t) {
            ep.out = A + (size_t)(t + 1) * M * K; ep.out_f32 = 0;
            launch_skinny_lds(0, A + (size_t)t * M * K, K, W, K, M, N, K, 2, dense_out(N), ep);
        }
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(512 * 64);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamps), h.size() * 8);
    // per workgroup: earliest entry, per-phase max over waves
    double sum[6] = {0}; double mx[6] = {0};
    int nwg = 256;
    for (int b = 0; b < nwg; ++b) {
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 8; ++w) t0 = std::min(t0, h[b * 64 + w * 8]);
        for (int n = 0; n < 6; ++n) {
            unsigned long long m = 0;
            for (int w = 0; w < 8; ++w) m = std::max(m, h[b * 64 + w * 8 + n]);
            sum[n] += (double)(m - t0); mx[n] = std::max(mx[n], (double)(m - t0));
        }
    }
    const char* names[6] = {"entry(all waves)", "loads issued", "loads landed", "mfma done", "barrier", "exit"};
    for (int n = 0; n < 6; ++n) printf("%-18s avg %8.0f  max %8.0f  (s_memtime ticks since the workgroup's first wave)\n", names[n], sum[n] / nwg, mx[n]);
    // spread of workgroup entry times across the grid
    unsigned long long lo = ~0ull, hi = 0, ex = 0;
    for (int b = 0; b < nwg; ++b) { lo = std::min(lo, h[b * 64]); hi = std::max(hi, h[b * 64]); for (int w = 0; w < 8; ++w) ex = std::max(ex, h[b * 64 + w * 8 + 5]); }
    printf("grid: first entry -> last entry %llu ticks, first entry -> last exit %llu ticks\n", hi - lo, ex - lo);
    return 0;
}
