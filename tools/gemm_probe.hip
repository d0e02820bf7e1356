// tools only: gemm_glds_kernel<3, 8> (csrc/gemm.h) at 2048^3 standalone against tools/ingest_probe.hip's k-loop skeleton (19.9 us + 3 us of epilogue stores).
// Arms: the production kernel with a plain fp32 epilogue; the same with accumulate; bf16 output.  Operands random bf16, hot (repeated on the same buffers).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/gemm_probe.hip -o tools/bin/gemm_probe && tools/bin/gemm_probe
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <vector>
#include "../hulc_amd/csrc/gemm.h"
void hulc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
using namespace hulc_bf16;
__device__ __forceinline__ void pdma16(const void* src, unsigned ldsaddr) {
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)ldsaddr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
namespace hulc_bf16 {
template <int NST, int NW, int VAR>      // NW = 4 (wave tile 64x64) or 8 waves (wave tile 32x64: two waves per SIMD hide each other's DMA issue / LDS latency)
__global__ void __launch_bounds__(NW * 64) glds_copy(DenseLoader<h16_t> al, DenseLoader<h16_t> bl, DenseOut om, EpiP ep, int M, int N, int K,
                                                           int tiles_m, int tiles_n) {
    constexpr int STAGE = 32 * 1024;
    constexpr int PW = 16 / NW;                  // 8-row pieces of each operand a wave DMAs per stage
    constexpr int TM = 8 / NW * 2 * 2 / 2;       // m-tiles per wave: 4 (NW = 4) or 2 (NW = 8)
    constexpr int WROWS = TM * 16;               // rows of the tile a wave multiplies
    extern __shared__ __attribute__((aligned(16))) char gg_smem[];
    typedef __attribute__((address_space(3))) char lchar;
    lchar* lds = (lchar*)gg_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int tm, tn;
    {
        const int nt = tiles_m * tiles_n, per = nt / 8, rem = nt % 8;
        const int x = blockIdx.x % 8, q = blockIdx.x / 8;
        const int tile = x * per + min(x, rem) + q;
        constexpr int GM = 4;
        const int gsz = GM * tiles_n, grp = tile / gsz, first_m = grp * GM, gm = min(GM, tiles_m - first_m);
        tm = first_m + (tile % gsz) % gm;
        tn = (tile % gsz) / gm;
    }
    const int m0 = tm * 128, n0 = tn * 128;
    const int r = lane >> 3, cs = (lane & 7) ^ (r & 6);
    const h16_t* asrc[PW];
    const h16_t* bsrc[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        asrc[j] = al.row(min(m0 + (wave * PW + j) * 8 + r, M - 1), 0).base + cs * 8;
        bsrc[j] = bl.row(min(n0 + (wave * PW + j) * 8 + r, N - 1), 0).base + cs * 8;
    }
    const int nk = (K + 63) >> 6;
    const bool khalf = (K & 63) != 0;
    auto issue = [&](int kt, int buf) {
        lchar* st = lds + buf * STAGE + wave * PW * 1024;
        long long ko = (long long)kt * 64;
        if (khalf && kt == nk - 1 && cs >= 4) ko -= 32;         // chunk beyond K: fetch a valid one instead (never multiplied)
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            if (VAR >= 2) { pdma16(asrc[j] + ko, (unsigned)(size_t)(st + j * 1024)); pdma16(bsrc[j] + ko, (unsigned)(size_t)(st + 16384 + j * 1024)); } else {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[j] + ko), (__attribute__((address_space(3))) void*)(st + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[j] + ko), (__attribute__((address_space(3))) void*)(st + 16384 + j * 1024), 16, 0, 0); }
        }
    };
    f32x4 acc[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 1, wn = wave & 1;
    const int foff = (li >> 3) * 1024 + (li & 7) * 128;
#pragma unroll
    for (int j = 0; j < NST - 1; ++j)
        if (j < nk) issue(j, j);
    int buf = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt of THIS wave has landed (later stages may still be in flight: 2*PW DMA instructions each); the barrier makes that
        // true for every wave and also says every wave has finished multiplying stage kt-1, whose buffer the next DMA overwrites
        const int ahead = min(NST - 2, nk - 1 - kt);
        if constexpr (PW == 4) {
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        } else {
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (kt + NST - 1 < nk) issue(kt + NST - 1, buf == 0 ? NST - 1 : buf - 1);
        lchar* sa = lds + buf * STAGE + wm * (WROWS / 8) * 1024 + foff;
        lchar* sb = lds + buf * STAGE + 16384 + wn * 8192 + foff;
        const int nkk = (khalf && kt == nk - 1) ? 1 : 2;
#pragma unroll 1
        for (int kk = 0; kk < (VAR >= 3 ? 2 : nkk); ++kk) {
            const int chunk = (((kk << 2) + g) ^ (li & 6)) << 4;
            h16x8_t a[TM], b[4];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(__attribute__((address_space(3))) h16x8_t*)(sa + i * 2048 + chunk);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *(__attribute__((address_space(3))) h16x8_t*)(sb + j * 2048 + chunk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = MFMA_16x16x32_H(b[j], a[i], acc[i][j], 0, 0, 0);   // D^T: lane owns 4 consecutive columns
        }
        buf = buf == NST - 1 ? 0 : buf + 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + wm * WROWS + i * 16 + li;
        if (row < M) {
            const long long obase = om.offset(row, 0);
            const int rrow = ep.res_rowmod > 0 ? row % ep.res_rowmod : row;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = n0 + wn * 64 + j * 16 + g * 4;
                if (col < N) {
                    if (VAR == 4) { uint2 w; w.x = pack2h(acc[i][j][0], acc[i][j][1]); w.y = pack2h(acc[i][j][2], acc[i][j][3]); *reinterpret_cast<uint2*>(reinterpret_cast<h16_t*>(ep.out) + obase + col) = w; }
                    else if (VAR >= 1) { *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(ep.out) + obase + col) = acc[i][j]; }
                    else {
                    const float v4[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    epi_store4<h16_t>(ep, v4, rrow, col, N, obase + col);
                    }
                }
            }
        }
    }
}
}
template <int VAR> static float time_copy(const h16_t* a, const h16_t* b, float* c, int M, int N, int K) {
    EpiP ep{}; ep.out = c; ep.out_f32 = 1;
    hipFuncSetAttribute((const void*)glds_copy<3, 8, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&]() { hipLaunchKernelGGL((glds_copy<3, 8, VAR>), dim3(256), dim3(512), 96 * 1024, 0, dense<h16_t>(a, M, K), dense<h16_t>(b, N, K), dense_out(N), ep, M, N, K, 16, 16); };
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(e0);
    for (int i = 0; i < 50; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000 / 50;
}
int main() {
    const int M = 2048, N = 2048, K = 2048;
    std::vector<h16_t> h((size_t)M * K);
    unsigned s = 1u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; const float f = ((int)(s >> 9) - (1 << 22)) * (1.f / (1 << 22)); unsigned u; memcpy(&u, &f, 4); v = (h16_t)((u + 0x8000u) >> 16); }
    h16_t *a, *b; float* c; h16_t* c16;
    hipMalloc(&a, h.size() * 2); hipMalloc(&b, h.size() * 2); hipMalloc(&c, (size_t)M * N * 4); hipMalloc(&c16, (size_t)M * N * 2);
    hipMemcpy(a, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int arm = 0; arm < 3; ++arm) {
        EpiP ep{}; ep.out = arm == 2 ? (void*)c16 : (void*)c; ep.out_f32 = arm == 2 ? 0 : 1; ep.accumulate = arm == 1;
        auto launch = [&]() { launch_gemm_glds(0, dense<h16_t>(a, M, K), dense<h16_t>(b, N, K), dense_out(N), ep, M, N, K); };
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("gemm_glds_kernel<3,8> 2048^3, %-28s %7.2f us\n", arm == 0 ? "fp32 store" : (arm == 1 ? "fp32 accumulate" : "16-bit store"), ms * 1000 / 50);
    }
    for (int rep = 0; rep < 3; ++rep) printf("alternating: builtin DMA %7.2f us   inline-asm DMA %7.2f us   (plain epilogue both)\n", time_copy<1>(a, b, c, M, N, K), time_copy<2>(a, b, c, M, N, K));
    printf("local copy of the kernel:            %7.2f us\n", time_copy<0>(a, b, c, M, N, K));
    printf("... plain float4 epilogue:           %7.2f us\n", time_copy<1>(a, b, c, M, N, K));
    printf("... + DMA as inline asm:             %7.2f us\n", time_copy<2>(a, b, c, M, N, K));
    printf("... + both k-halves unconditionally: %7.2f us\n", time_copy<3>(a, b, c, M, N, K));
    printf("inline-asm DMA + PLAIN 16-bit store (8-byte pieces per lane): %7.2f us\n", time_copy<4>(a, b, c, M, N, K));
    return 0;
}
