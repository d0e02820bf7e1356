// tools only: where a band's time goes inside conv_reg_kernel (csrc/conv_reg.h) — shader-clock stamps (s_memtime, lane 0 of every wave of workgroups
// 0..15) of the band phases: wait at the band barrier | DMA issue + tile setup | multiply loops (+ the earlier pairs' epilogues) | the wait for the next
// band's DMA (which on gfx950 also drains the output stores) | the last pair's epilogue.   2048 static-camera frames, random data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cr_stamps.hip -o tools/bin/cr_stamps && tools/bin/cr_stamps
#define HULC_CR_STAMPS 1
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../hulc_amd/csrc/conv_reg.h"
void hulc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
using namespace hulc_bf16;

static std::vector<h16_t> rnd16(size_t n, float scale, unsigned s) {
    std::vector<h16_t> h(n);
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; const float f = ((int)(s >> 9) - (1 << 22)) * (scale / (1 << 22)); unsigned u; memcpy(&u, &f, 4); h[i] = (h16_t)((u + 0x8000u) >> 16); }
    return h;
}
template <typename T> static T* dev(const std::vector<T>& h) { T* d; hipMalloc(&d, h.size() * sizeof(T) + 512); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d; }

template <int CK, int TA, int TB, int SI, bool REV, int OS, int NWV, int NBUF, bool ORD, int EPI = 0>
static void run(const char* name, int Nf, int IMH, int OUTH) {
    constexpr int CN = OS == 1 ? 64 : 32, K = TA * TB * CK, NCLS = OS * OS;
    const size_t nimg = (size_t)Nf * IMH * IMH * CK, nout = (size_t)Nf * OUTH * OUTH * CN;
    h16_t *img = dev(rnd16(nimg, 1.f, 1u)), *w = dev(rnd16((size_t)NCLS * CN * K, 0.05f, 2u)), *out; hipMalloc(&out, nout * 2 + 512);
    std::vector<float> hb(64, 0.01f); float* bias = dev(hb);
    std::vector<unsigned> hbits((size_t)Nf * OUTH * OUTH * 2, 0x5a5a5a5au); unsigned* bits = dev(hbits);
    void* zp; hipMalloc(&zp, 256); hipMemset(zp, 0, 256);
    void* dumpb; hipMalloc(&dumpb, 8192);
    ConvTileP p{}; p.dump = (h16_t*)dumpb; p.img = img; p.IMH = p.IMW = IMH; p.w = w; p.out = out; p.OUTH = p.OUTW = OUTH; p.Nf = Nf;
    if (REV) { p.maskbits = bits; p.zeros = (const h16_t*)zp; } else { p.bias = bias; p.relu = 1; if (SI == 2) p.bits_out = bits; }
    for (int i = 0; i < 3; ++i) launch_conv_reg<CK, TA, TB, SI, REV, OS, NWV, NBUF, ORD, EPI>(0, p);
    hipDeviceSynchronize();
    std::vector<unsigned long long> z(16 * 8 * 32 * 8, 0ull);
    hipMemcpyToSymbol(HIP_SYMBOL(g_cr_stamps), z.data(), z.size() * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch_conv_reg<CK, TA, TB, SI, REV, OS, NWV, NBUF, ORD, EPI>(0, p);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(g_cr_stamps), z.size() * 8);
    const char* nm[6] = {"wait at the band barrier", "DMA issue (waves that issue now) + tile setup", "multiply loops + earlier pairs' epilogues", "late DMA issue (second wave of a SIMD) + wait (vmcnt)", "last pair's epilogue", "(to next band)"};
    for (int half = 0; half < (NWV == 8 ? 2 : 1); ++half) {
        double acc[6] = {0}, tot = 0; long cnt = 0; int nb_seen = 0;
        for (int b = 0; b < 16; ++b)
            for (int wv = half * (NWV / 2); wv < (NWV == 8 ? (half + 1) * 4 : NWV); ++wv) {
                const unsigned long long* s = &z[((size_t)(b * 8 + wv) * 32) * 8];
                int nb = 0; while (nb < 31 && s[(nb + 1) * 8] > s[nb * 8] && s[nb * 8]) ++nb;
                nb_seen = std::max(nb_seen, nb);
                for (int it = 1; it + 1 < nb; ++it) {
                    const unsigned long long* q = s + it * 8;
                    if (!(q[5] >= q[4] && q[4] >= q[3] && q[3] >= q[2] && q[2] >= q[1] && q[1] >= q[0])) continue;
                    for (int ph = 0; ph < 5; ++ph) acc[ph] += (double)(q[ph + 1] - q[ph]);
                    acc[5] += (double)(s[(it + 1) * 8] - q[5]);
                    tot += (double)(s[(it + 1) * 8] - q[0]); ++cnt;
                }
            }
        if (!cnt) continue;
        printf("%s, waves %d-%d: %.1f us per launch, %d bands per workgroup; per band in s_memtime ticks (100 MHz -> x 10 ns), %ld samples\n", name, half * (NWV / 2), NWV == 8 ? half * 4 + 3 : NWV - 1, ms * 1e3, nb_seen + 1, cnt);
        for (int ph = 0; ph < 6; ++ph) printf("   %-62s %8.1f ticks  %5.1f %%\n", nm[ph], acc[ph] / cnt, 100.0 * acc[ph] / tot);
        printf("   band total %.1f ticks\n", tot / cnt);
    }
    hipFree(img); hipFree(w); hipFree(out); hipFree(bias); hipFree(bits); hipFree(zp);
}

int main() {
    const int Nf = 2048;
    run<64, 3, 3, 1, false, 1, 8, 0, false>("conv3 fwd, 8 waves (r4 product)", Nf, 23, 21);
    run<64, 3, 3, 1, false, 1, 8, 0, true>("conv3 fwd, 8 waves, slot decode in registers", Nf, 23, 21);
    run<32, 4, 4, 2, false, 1, 8, 0, false>("conv2 fwd, 8 waves", Nf, 49, 23);
    run<32, 4, 4, 2, false, 1, 8, 0, true>("conv2 fwd, 8 waves, slot decode in registers", Nf, 49, 23);
    run<32, 4, 4, 2, false, 1, 4, 0, false>("conv2 fwd, 2 x 4 waves (r4 product)", Nf, 49, 23);
    run<64, 3, 3, 1, true, 1, 8, 0, false>("conv3 dgrad, 8 waves (r4 product)", Nf, 21, 23);
    run<64, 2, 2, 1, true, 2, 8, 0, false>("conv2 dgrad, 8 waves", Nf, 23, 49);
    run<64, 2, 2, 1, true, 2, 4, 0, false>("conv2 dgrad, 2 x 4 waves (r4 product)", Nf, 23, 49);
    // round 5 production forms of the data gradients: pipelined epilogue (stamps 2 -> 3 = the tile loop with the drains inside, 3 -> 4 = the wait, no epilogue phase)
    run<64, 3, 3, 1, true, 1, 4, 0, true, 1>("conv3 dgrad, 2 x 4 waves, pipelined epilogue + slot registers (r5 product)", Nf, 21, 23);
    run<64, 2, 2, 1, true, 2, 4, 0, true, 1>("conv2 dgrad, 2 x 4 waves, pipelined epilogue + slot registers (r5 product)", Nf, 23, 49);
    return 0;
}
