// tools only: the weights-in-registers conv2 / conv3 kernels (csrc/conv_reg.h) in their workgroup / slot-decode forms, standalone:
// every form's output compared with the production form's (bit-identical: a pixel's dot product has one summation order in all of them) and
// with a direct fp32 evaluation on a sample of pixels, time per launch on 2048 static-camera and gripper-camera frames, phase ablations.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/cr_bench.hip -o tools/bin/cr_bench && tools/bin/cr_bench [ablate]
#include <cstdio>
#include <cstdlib>
#include <cstdarg>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../hulc_amd/csrc/conv_reg.h"
void hulc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
using namespace hulc_bf16;

static float b2f(h16_t v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static std::vector<h16_t> rnd16(size_t n, float scale, unsigned s) {
    std::vector<h16_t> h(n);
    for (size_t i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        const float f = ((int)(s >> 9) - (1 << 22)) * (scale / (1 << 22));
        unsigned u; memcpy(&u, &f, 4);
        h[i] = (h16_t)((u + 0x8000u) >> 16);
    }
    return h;
}
template <typename T> static T* dev(const std::vector<T>& h) { T* d; hipMalloc(&d, h.size() * sizeof(T) + 512); hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice); return d; }

struct Case { const char* name; int kind; int IMH, OUTH; };      // kind 0: conv3 fwd, 1: conv2 fwd, 2: conv3 dgrad, 3: conv2 dgrad

template <int CK, int TA, int TB, int SI, bool REV, int OS, int NWV, int NBUF, bool ORD, int EPI = 0, int LDR = 0>
static float run_form(ConvTileP p, int reps, bool* ok) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    *ok = launch_conv_reg<CK, TA, TB, SI, REV, OS, NWV, NBUF, ORD, EPI, LDR>(0, p);
    if (!*ok) return 0.f;
    for (int i = 0; i < 2; ++i) launch_conv_reg<CK, TA, TB, SI, REV, OS, NWV, NBUF, ORD, EPI, LDR>(0, p);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch_conv_reg<CK, TA, TB, SI, REV, OS, NWV, NBUF, ORD, EPI, LDR>(0, p);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) { *ok = false; }
    return ms / reps * 1e3f;
}

template <int CK, int TA, int TB, int SI, bool REV, int OS>
static void bench(const char* name, int Nf, int IMH, int OUTH, bool ablate) {
    constexpr int CN = OS == 1 ? 64 : 32, K = TA * TB * CK, NCLS = OS * OS;
    const size_t nimg = (size_t)Nf * IMH * IMH * CK, nout = (size_t)Nf * OUTH * OUTH * CN, nw = (size_t)NCLS * CN * K;
    std::vector<h16_t> himg = rnd16(nimg, 1.f, 12345u), hw = rnd16(nw, 0.05f, 777u);
    h16_t *img = dev(himg), *w = dev(hw), *out, *ref;
    hipMalloc(&out, nout * 2 + 512); hipMalloc(&ref, nout * 2 + 512);
    std::vector<float> hb(64); for (int i = 0; i < 64; ++i) hb[i] = 0.01f * (i - 30);
    float* bias = dev(hb);
    const int WPP = REV ? (OS == 1 ? 2 : 1) : 2;
    std::vector<unsigned> hbits((size_t)Nf * OUTH * OUTH * WPP);
    { unsigned s = 99u; for (auto& v : hbits) { s = s * 1664525u + 1013904223u; v = s ^ (s >> 7); } }
    unsigned *bits = dev(hbits), *bits_out, *bits_ref;
    hipMalloc(&bits_out, hbits.size() * 4 + 512); hipMalloc(&bits_ref, hbits.size() * 4 + 512);
    void* zp; hipMalloc(&zp, 256); hipMemset(zp, 0, 256);
    void* dumpb; hipMalloc(&dumpb, 8192);
    ConvTileP p{}; p.dump = (h16_t*)dumpb; p.img = img; p.IMH = p.IMW = IMH; p.w = w; p.OUTH = p.OUTW = OUTH; p.Nf = Nf;
    if (REV) { p.maskbits = bits; p.zeros = (const h16_t*)zp; p.relu = 0; } else { p.bias = bias; p.relu = 1; if (SI == 2) p.bits_out = bits_ref; }
    bool ok;
    // reference = the production form of round 4 (8 waves, two buffers, old order; conv3 dgrad ran it too)
    hipMemset(ref, 0xEE, nout * 2);
    p.out = ref;
    const float t_ref = run_form<CK, TA, TB, SI, REV, OS, 8, 0, false>(p, 10, &ok);
    std::vector<h16_t> href(nout); hipMemcpy(href.data(), ref, nout * 2, hipMemcpyDeviceToHost);
    // direct evaluation of a sample of output pixels in double
    double worst = 0;
    {
        unsigned s = 4242u;
        for (int it = 0; it < 400; ++it) {
            s = s * 1664525u + 1013904223u; const int f = (s >> 8) % Nf;
            s = s * 1664525u + 1013904223u; const int oi = (s >> 8) % OUTH;
            s = s * 1664525u + 1013904223u; const int oj = (s >> 8) % OUTH;
            s = s * 1664525u + 1013904223u; const int cn = (s >> 8) % CN;
            double acc = 0;
            if (!REV) {
                for (int ta = 0; ta < TA; ++ta) for (int tb = 0; tb < TB; ++tb) for (int ck = 0; ck < CK; ++ck)
                    acc += (double)b2f(himg[(((size_t)f * IMH + oi * SI + ta) * IMH + oj * SI + tb) * CK + ck]) * b2f(hw[(size_t)cn * K + (ta * TB + tb) * CK + ck]);
                acc = std::max(acc + hb[cn], 0.0);
            } else {
                const int ph = oi % OS, pw = oj % OS, ci = oi / OS, cj = oj / OS, cls = ph * OS + pw;
                for (int ta = 0; ta < TA; ++ta) for (int tb = 0; tb < TB; ++tb) {
                    const int r = ci - ta, c = cj - tb;
                    if (r < 0 || r >= IMH || c < 0 || c >= IMH) continue;
                    for (int ck = 0; ck < CK; ++ck)
                        acc += (double)b2f(himg[(((size_t)f * IMH + r) * IMH + c) * CK + ck]) * b2f(hw[((size_t)cls * CN + cn) * K + (ta * TB + tb) * CK + ck]);
                }
                const unsigned wd = hbits[(((size_t)f * OUTH + oi) * OUTH + oj) * WPP + cn / 32];
                if (!((wd >> (cn % 32)) & 1u)) acc = 0;
            }
            const double got = b2f(href[(((size_t)f * OUTH + oi) * OUTH + oj) * CN + cn]);
            worst = std::max(worst, std::fabs(got - acc) / (std::fabs(acc) + 0.05));
        }
    }
    printf("%-28s Nf=%d  round-4 form (8 waves, 2 buffers) %.1f us   [direct fp64 sample: worst rel %.2e %s]\n", name, Nf, t_ref, worst, worst < 2e-2 ? "ok" : "MISMATCH");
    auto cmp = [&](const char* form, float t) {
        std::vector<h16_t> h(nout); hipMemcpy(h.data(), out, nout * 2, hipMemcpyDeviceToHost);
        size_t bad = 0; double worstd = 0;
        for (size_t i = 0; i < nout; ++i) if (h[i] != href[i]) { ++bad; worstd = std::max(worstd, std::fabs((double)b2f(h[i]) - b2f(href[i])) / (std::fabs((double)b2f(href[i])) + 0.05)); }
        size_t badb = 0;
        if (!REV && SI == 2) {
            std::vector<unsigned> a(hbits.size()), b(hbits.size());
            hipMemcpy(a.data(), bits_out, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), bits_ref, b.size() * 4, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < a.size(); ++i) badb += a[i] != b[i];
        }
        printf("    %-44s %7.1f us   %s", form, t, bad || badb ? (worstd < 1.6e-2 ? "equal to 16-bit rounding" : "DIFFERS") : "bit-identical");
        if (bad || badb) printf(" (%zu values off by <= %.1e rel, %zu mask words)", bad, worstd, badb);
        printf("\n");
    };
    p.out = out; if (!REV && SI == 2) p.bits_out = bits_out;
#define FORM(nwv, nbuf, ord, label) FORME(nwv, nbuf, ord, 0, label)
#define FORML(epi, label) do { if constexpr (OS == 1) { hipMemset(out, 0xEE, nout * 2); if (!REV && SI == 2) hipMemset(bits_out, 0xEE, hbits.size() * 4); const float t = run_form<CK, TA, TB, SI, REV, OS, 8, 0, true, epi, 2>(p, 10, &ok); if (ok) cmp(label, t); else printf("    %-44s not launchable\n", label); } } while (0)
#define FORME(nwv, nbuf, ord, epi, label) do { hipMemset(out, 0xEE, nout * 2); if (!REV && SI == 2) hipMemset(bits_out, 0xEE, hbits.size() * 4); const float t = run_form<CK, TA, TB, SI, REV, OS, nwv, nbuf, ord, epi>(p, 10, &ok); if (ok) cmp(label, t); else printf("    %-44s not launchable\n", label); } while (0)
    FORM(8, 0, true, "8 waves, 2 buffers, slot decode in registers");
    FORM(4, 0, false, "2 x 4 waves, 1 buffer each");
    FORM(4, 0, true, "2 x 4 waves, 1 buffer, slot decode in registers");
    FORME(8, 0, false, 1, "8 waves, pipelined epilogue (1 accumulator)");
    FORME(8, 0, true, 1, "8 waves, pipelined epilogue (1) + slot registers");
    FORME(4, 0, false, 1, "2 x 4 waves, pipelined epilogue (1)");
    FORME(4, 0, true, 1, "2 x 4 waves, pipelined epilogue (1) + slot registers");
    FORML(0, "6 compute + 2 LOADER waves");
    if constexpr (REV) FORML(1, "6 compute + 2 LOADER waves, pipelined epilogue");
    // workgroups per CU: the band geometry sized for 3 / 4 co-resident 4-wave workgroups (53 / 40 KB of LDS each) and for 2 co-resident 8-wave ones
    for (int wg : {3, 4}) {
        g_conv_reg_wgpc = wg;
        char lb[96]; snprintf(lb, sizeof(lb), "%d x 4 waves per CU, slot registers%s", wg, REV ? ", pipelined epilogue" : "");
        if constexpr (REV) FORME(4, 0, true, 1, lb); else FORM(4, 0, true, lb);
    }
    g_conv_reg_wgpc = 2;
    FORM(8, 0, true, "2 x 8 waves per CU (2 buffers each), slot registers");
    g_conv_reg_wgpc = 0;
    if (ablate) {
        const int flags[] = {0, 4, 8, 16, 12, 20, 24, 2};
        const char* fn[] = {"full", "no-dma", "no-epilogue", "no-mfma", "mfma-only", "epilogue-only", "dma-only", "no-compute"};
        for (int v = 0; v < 3; ++v) {
            printf("    ablation %s:", v == 0 ? "8w + registers" : (v == 1 ? "2x4w" : "8w (r4)"));
            for (int k = 0; k < 8; ++k) {
                p.dbg = flags[k];
                float t = v == 0 ? run_form<CK, TA, TB, SI, REV, OS, 8, 0, true>(p, 10, &ok) : (v == 1 ? run_form<CK, TA, TB, SI, REV, OS, 4, 0, false>(p, 10, &ok) : run_form<CK, TA, TB, SI, REV, OS, 8, 0, false>(p, 10, &ok));
                printf("  %s %.1f", fn[k], t);
            }
            printf("\n");
            p.dbg = 0;
        }
    }
    hipFree(img); hipFree(w); hipFree(out); hipFree(ref); hipFree(bias); hipFree(bits); hipFree(bits_out); hipFree(bits_ref); hipFree(zp);
}

int main(int argc, char** argv) {
    const bool ablate = argc > 1 && !strcmp(argv[1], "ablate");
    for (int cam = 0; cam < 2; ++cam) {
        const int H1 = cam ? 20 : 49, H2 = cam ? 9 : 23, H3 = cam ? 7 : 21, Nf = cam ? 2051 : 2048;
        printf("==== %s camera ====\n", cam ? "gripper" : "static");
        bench<64, 3, 3, 1, false, 1>("conv3 fwd", Nf, H2, H3, ablate && !cam);
        bench<32, 4, 4, 2, false, 1>("conv2 fwd (+bits)", Nf, H1, H2, ablate && !cam);
        bench<64, 3, 3, 1, true, 1>("conv3 dgrad (bits)", Nf, H3, H2, ablate && !cam);
        bench<64, 2, 2, 1, true, 2>("conv2 dgrad (bits)", Nf, H2, H1, ablate && !cam);
    }
    return 0;
}
