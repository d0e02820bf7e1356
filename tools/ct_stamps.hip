// NOTE (round 5): written against the round-4 headers.  It no longer builds / runs against the current ones (the fused kernels read fragment-ordered weights since round 4's
// last day; the conv_tile stamp build faults): kept for the profiles of rounds 2 - 4 it produced (profiles/r0[234]_*), not part of tools/refresh_profiles.sh any more.
// tools only: per-band phase timing (shader-clock stamps, lane 0 of every wave) of the raw-tile conv kernels on random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ct_stamps.hip -o tools/bin/ct_stamps && tools/bin/ct_stamps
#define HULC_CT_STAMPS 1
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../hulc_amd/csrc/conv_tile.h"
void hulc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
using namespace hulc_bf16;

static void fill_random(h16_t* d, size_t n, float scale) {
    std::vector<h16_t> h(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float f = ((int)(s >> 9) - (1 << 22)) * (scale / (1 << 22));
        unsigned u; memcpy(&u, &f, 4);
        h[i] = (h16_t)((u + 0x8000u) >> 16);
    }
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
}

template <int CK, int CN, int TA, int TB, int SI, int OS, bool REV>
static void run(const char* name, int Nf, int IMH, int OUTH, int wrows, int wcols, bool mask, int dbg = 0) {
    h16_t *img, *w, *out; float* bias; unsigned* bits;
    const size_t nimg = (size_t)Nf * IMH * IMH * CK, nout = (size_t)Nf * OUTH * OUTH * CN;
    hipMalloc(&img, nimg * 2 + 256); hipMalloc(&w, (size_t)wrows * wcols * 2 + 256); hipMalloc(&out, nout * 2 + 256); hipMalloc(&bias, 256 * 4);
    hipMalloc(&bits, (size_t)Nf * OUTH * OUTH * 2 * 4 + 256);
    fill_random(img, nimg, 1.f); fill_random(w, (size_t)wrows * wcols, 0.1f);
    hipMemset(bias, 0, 256 * 4); hipMemset(bits, 0x5a, (size_t)Nf * OUTH * OUTH * 2 * 4);
    ConvTileP p{}; p.img = img; p.IMH = p.IMW = IMH; p.w = w; p.out = out; p.OUTH = p.OUTW = OUTH; p.bias = REV ? nullptr : bias; p.relu = REV ? 0 : 1; p.Nf = Nf;
    p.dbg = dbg;
    if (mask) p.maskbits = bits;
    if (!REV && CN == 64 && SI == 2) p.bits_out = bits;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch_conv_tile<CK, CN, TA, TB, SI, OS, REV>(0, p);
    hipDeviceSynchronize();
    {   // stale stamps of an earlier case (more bands per workgroup) must not be read as bands of this one
        std::vector<unsigned long long> z(256 * 8 * 64 * 8, 0ull);
        hipMemcpyToSymbol(HIP_SYMBOL(g_ct_stamps), z.data(), z.size() * 8);
    }
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) launch_conv_tile<CK, CN, TA, TB, SI, OS, REV>(0, p);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(256 * 8 * 64 * 8);
    hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_ct_stamps), h.size() * 8);
    // phases per band (wave-averaged over workgroups / waves / bands 1..n-2): deltas between consecutive stamps
    const char* nm[8] = {"wait barrier1 (others' epilogues)", "commit band -> LDS (incl. vmcnt wait)", "wait barrier2", "prefetch issue + addr math", "earlier groups of the band + last group's setup", "MFMA loop (last group)", "epilogue (last group)", "(to next band start)"};
    double acc[8] = {0}; long cnt = 0; double band_total = 0;
    int nb_seen = 0;
    for (int b = 0; b < 256; ++b)
        for (int wv = 0; wv < 8; ++wv) {
            const unsigned long long* s = &h[((size_t)(b * 8 + wv) * 64) * 8];
            int nb = 0;
            while (nb < 63 && s[(nb + 1) * 8] > s[nb * 8] && s[nb * 8] != 0) ++nb;
            nb_seen = std::max(nb_seen, nb);
            for (int it = 1; it + 1 < nb; ++it) {
                const unsigned long long* q = s + it * 8;
                if (!(q[7] > q[6] && q[6] > q[5] && q[5] >= q[4])) continue;        // wave had no group in this band
                for (int ph = 0; ph < 7; ++ph) acc[ph] += (double)(q[ph + 1] - q[ph]);
                acc[7] += (double)(s[(it + 1) * 8] - q[7]);
                band_total += (double)(s[(it + 1) * 8] - q[0]);
                ++cnt;
            }
        }
    printf("%s: %.1f us per launch, %d bands per workgroup; per band (s_memtime ticks; stamps 5-7 are overwritten per group, so they time the wave's LAST group of the band), %ld samples\n", name, ms / 5 * 1e3, nb_seen, cnt);
    for (int ph = 0; ph < 8; ++ph) printf("   %-52s %8.0f ticks  %5.1f %%\n", nm[ph], acc[ph] / cnt, 100.0 * acc[ph] / band_total);
    printf("   band total %.0f ticks -> %.2f us per band if the launch is all bands\n", band_total / cnt, ms / 5 * 1e3 / std::max(1, nb_seen));
    hipFree(img); hipFree(w); hipFree(out); hipFree(bias); hipFree(bits);
}

int main() {
    const int Nf = 2048;
    run<64, 64, 3, 3, 1, 1, false>("fwd3  (23x23x64 -> 21x21x64)", Nf, 23, 21, 64, 576, false);
    run<32, 64, 4, 4, 2, 1, false>("fwd2  (49x49x32 -> 23x23x64)", Nf, 49, 23, 64, 512, false);
    run<64, 64, 3, 3, 1, 1, true>("dgrad3 (21x21x64 -> 23x23x64)", Nf, 21, 23, 64, 576, true);
    run<64, 32, 2, 2, 1, 2, true>("dgrad2 (23x23x64 -> 49x49x32)", Nf, 23, 49, 128, 256, true);
    run<64, 64, 3, 3, 1, 1, false>("fwd3 gripper (9x9 -> 7x7)", Nf, 9, 7, 64, 576, false);
    printf("---- the same without the phase skew between wave halves (dbg bit 6)\n");
    run<64, 64, 3, 3, 1, 1, false>("fwd3  no-skew", Nf, 23, 21, 64, 576, false, 64);
    run<32, 64, 4, 4, 2, 1, false>("fwd2  no-skew", Nf, 49, 23, 64, 512, false, 64);
    run<64, 64, 3, 3, 1, 1, true>("dgrad3 no-skew", Nf, 21, 23, 64, 576, true, 64);
    run<64, 32, 2, 2, 1, 2, true>("dgrad2 no-skew", Nf, 23, 49, 128, 256, true, 64);
    return 0;
}
