// tools only (round 6): would a 256 x 256 tile with the K range split four ways beat gemm_glds_kernel's 128 x 128 tiles at 2048^3?
// gemm_glds_kernel is bound by what a CU can pull through its load path (1 MB of operands per 128 x 128 x 2048 tile at ~23 B/clk: 19 us, profiles/r05_ingest_probe.txt).
// A 256 x 256 x 512 slice is the same 67 MFLOP per workgroup with HALF the operand bytes (512 KB) and half the LDS fragment traffic per MFMA (wave tile 64 x 128);
// the price is a reduction over the four K slices of a tile.  All four slices of a tile run on ONE XCD (blockIdx % 8), so the reduction can go through that XCD's L2.
//   arm 0: production gemm_glds_kernel (fp32 store)
//   arm 1: 256 x 256 x (K / 4) slices, every workgroup stores its own fp32 partial (4 x the output bytes; checks the k-loop's speed and its arithmetic)
//   arm 2: the same + reduce-scatter among the four workgroups of a tile through L2 (each owns 64 rows of the tile: writes the other 192 rows of its partial to scratch,
//          flags, reads the three partials of its own rows, stores the sum) — the form a product kernel would have
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value tools/gemm256_probe.hip -o tools/bin/gemm256_probe && tools/bin/gemm256_probe
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cmath>
#include <vector>
#include "../hulc_amd/csrc/gemm.h"
void hulc_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
using namespace hulc_bf16;
__device__ __forceinline__ void pdma16(const void* src, unsigned ldsaddr) {
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)ldsaddr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
namespace hulc_bf16 {
// C[m][n] = sum_k A[m][k] B[n][k] (both K-major, ld = K), M = N = 2048 here; tile 256 x 256, slice = K / 4; 8 waves, wave tile 64 x 128; stage = 64 k of both operands
// (64 KB), two stages.  EXCH: 0 = partial per slice into out + slice * M * N; 1 = reduce-scatter through `scratch` ([tile][slice][256][256] fp32) and `flags` ([tile][4]).
template <int EXCH>
__global__ void __launch_bounds__(512) gemm256_kernel(const h16_t* __restrict__ A, const h16_t* __restrict__ B, float* __restrict__ out, float* __restrict__ scratch,
                                                      int* __restrict__ flags, int M, int N, int K, int epoch) {
    constexpr int STAGE = 64 * 1024, PW = 4, TM = 4, TN = 8;
    extern __shared__ __attribute__((aligned(16))) char gg_smem[];
    typedef __attribute__((address_space(3))) char lchar;
    lchar* lds = (lchar*)gg_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    // block -> (tile, slice): XCD x = b % 8 owns the 2 x 4 block of tiles (rows 2 (x >> 1) .. + 1, columns 4 (x & 1) .. + 3), four consecutive workgroups of an XCD = the four slices of a tile
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tl = q >> 2, slice = q & 3;
    const int tm = 2 * (x >> 1) + (tl & 1), tn = 4 * (x & 1) + (tl >> 1);
    const int tile = tm * (N / 256) + tn;
    const int m0 = tm * 256, n0 = tn * 256;
    const int KS = K / 4, k0 = slice * KS;
    const int r = lane >> 3, cs = (lane & 7) ^ (r & 6);
    const h16_t* asrc[PW];
    const h16_t* bsrc[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
        asrc[j] = A + (long long)(m0 + (wave * PW + j) * 8 + r) * K + k0 + cs * 8;
        bsrc[j] = B + (long long)(n0 + (wave * PW + j) * 8 + r) * K + k0 + cs * 8;
    }
    const int nk = KS >> 6;
    auto issue = [&](int kt, int buf) {
        lchar* st = lds + buf * STAGE + wave * PW * 1024;
        const long long ko = (long long)kt * 64;
#pragma unroll
        for (int j = 0; j < PW; ++j) {
            pdma16(asrc[j] + ko, (unsigned)(size_t)(st + j * 1024));
            pdma16(bsrc[j] + ko, (unsigned)(size_t)(st + 32768 + j * 1024));
        }
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 1, wn = wave & 1;
    const int foff = (li >> 3) * 1024 + (li & 7) * 128;
    issue(0, 0);
    int buf = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // stage kt has landed for every wave; every wave is done with stage kt - 1
        if (kt + 1 < nk) issue(kt + 1, buf ^ 1);
        lchar* sa = lds + buf * STAGE + wm * 8 * 1024 + foff;
        lchar* sb = lds + buf * STAGE + 32768 + wn * 16 * 1024 + foff;
#pragma unroll 1
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = (((kk << 2) + g) ^ (li & 6)) << 4;
            h16x8_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(__attribute__((address_space(3))) h16x8_t*)(sa + i * 2048 + chunk);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(__attribute__((address_space(3))) h16x8_t*)(sb + j * 2048 + chunk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MFMA_16x16x32_H(b[j], a[i], acc[i][j], 0, 0, 0);   // D^T: lane owns 4 consecutive columns of row li
        }
        buf ^= 1;
    }
    if (EXCH == 0) {
        float* o = out + (long long)slice * M * N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * 64 + i * 16 + li;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * 128 + j * 16 + g * 4;
                *reinterpret_cast<f32x4*>(o + (long long)row * N + col) = acc[i][j];
            }
        }
        return;
    }
    // ---- reduce-scatter: wave row wm (64 rows of the tile) belongs to slice wm.  A wave whose rows belong to another slice writes its partial to that slice's inbox;
    // the owner adds the three inbox partials of its rows to its own and stores.  inbox[tile][owner][from][64][256] fp32, from != owner.
    float* const tbase = scratch + (long long)tile * 4 * 4 * 64 * 256;
    if (wm != slice) {
        float* o = tbase + ((long long)wm * 4 + slice) * 64 * 256;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                *reinterpret_cast<f32x4*>(o + (i * 16 + li) * 256 + wn * 128 + j * 16 + g * 4) = acc[i][j];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's inbox stores have left (they are in the XCD's L2: same-XCD readers see them)
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(flags + tile * 4 + slice, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // "slice has delivered": relaxed — every wave's stores are acknowledged by the (shared) L2 already; a release would write the whole L2 back
    if (wm == slice) {
        // wait for the three other slices of this tile
        if (lane == 0) {
#pragma unroll 1
            for (int s = 0; s < 4; ++s) {
                if (s == slice) continue;
                for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(flags + tile * 4 + s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; ++spin) __builtin_amdgcn_s_sleep(1);      // bounded: a probe must not hang the box
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            if (s == slice) continue;
            const float* in = tbase + ((long long)slice * 4 + s) * 64 * 256;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in + (i * 16 + li) * 256 + wn * 128 + j * 16 + g * 4));
                    acc[i][j] += v;
                }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * 64 + i * 16 + li;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int col = n0 + wn * 128 + j * 16 + g * 4;
                *reinterpret_cast<f32x4*>(out + (long long)row * N + col) = acc[i][j];
            }
        }
    }
}
// 128 x 256 tiles, K split in TWO: 768 KB of operands per workgroup (1 MB today), 64 KB of partials exchanged per workgroup (2 MB per XCD: stays in its L2).
// 8 waves as 2 x 4, wave tile 64 x 64; stage = 64 k: A 16 KB + B 32 KB, three stages (144 KB).
template <int EXCH>
__global__ void __launch_bounds__(512) gemm_s2_kernel(const h16_t* __restrict__ A, const h16_t* __restrict__ B, float* __restrict__ out, float* __restrict__ scratch,
                                                      int* __restrict__ flags, int M, int N, int K, int epoch) {
    constexpr int STAGE = 48 * 1024, NST = 3, TM = 4, TN = 4;
    extern __shared__ __attribute__((aligned(16))) char gg_smem[];
    typedef __attribute__((address_space(3))) char lchar;
    lchar* lds = (lchar*)gg_smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    // XCD x = b % 8 owns a 4 x 4 block of tiles (16 row tiles x 8 column tiles in all: XCD grid 4 x 2); two consecutive workgroups of an XCD = the two slices of a tile
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tl = q >> 1, slice = q & 1;
    const int tm = 4 * (x >> 1) + (tl & 3), tn = 4 * (x & 1) + (tl >> 2);
    const int tile = tm * (N / 256) + tn;
    const int m0 = tm * 128, n0 = tn * 256;
    const int KS = K / 2, k0 = slice * KS;
    const int r = lane >> 3, cs = (lane & 7) ^ (r & 6);
    const h16_t* asrc[2];
    const h16_t* bsrc[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) asrc[j] = A + (long long)(m0 + (wave * 2 + j) * 8 + r) * K + k0 + cs * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) bsrc[j] = B + (long long)(n0 + (wave * 4 + j) * 8 + r) * K + k0 + cs * 8;
    const int nk = KS >> 6;
    auto issue = [&](int kt, int buf) {
        lchar* st = lds + buf * STAGE;
        const long long ko = (long long)kt * 64;
#pragma unroll
        for (int j = 0; j < 2; ++j) pdma16(asrc[j] + ko, (unsigned)(size_t)(st + (wave * 2 + j) * 1024));
#pragma unroll
        for (int j = 0; j < 4; ++j) pdma16(bsrc[j] + ko, (unsigned)(size_t)(st + 16384 + (wave * 4 + j) * 1024));
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 2, wn = wave & 3;
    const int foff = (li >> 3) * 1024 + (li & 7) * 128;
    issue(0, 0);
    if (nk > 1) issue(1, 1);
    int buf = 0;
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");      // 6 DMA instructions per wave and stage: stage kt + 1 may still be in flight
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (kt + 2 < nk) issue(kt + 2, buf == 0 ? 2 : buf - 1);
        lchar* sa = lds + buf * STAGE + wm * 8 * 1024 + foff;
        lchar* sb = lds + buf * STAGE + 16384 + wn * 8 * 1024 + foff;
#pragma unroll 1
        for (int kk = 0; kk < 2; ++kk) {
            const int chunk = (((kk << 2) + g) ^ (li & 6)) << 4;
            h16x8_t a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(__attribute__((address_space(3))) h16x8_t*)(sa + i * 2048 + chunk);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *(__attribute__((address_space(3))) h16x8_t*)(sb + j * 2048 + chunk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = MFMA_16x16x32_H(b[j], a[i], acc[i][j], 0, 0, 0);
        }
        buf = buf == NST - 1 ? 0 : buf + 1;
    }
    if (EXCH == 0) {
        float* o = out + (long long)slice * M * N;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * 64 + i * 16 + li;
#pragma unroll
            for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(o + (long long)row * N + n0 + wn * 64 + j * 16 + g * 4) = acc[i][j];
        }
        return;
    }
    // wave row wm (64 rows) belongs to slice wm: the other slice's waves send theirs to its inbox [tile][owner][64][256]
    float* const tbase = scratch + (long long)tile * 2 * 64 * 256;
    if (wm != slice) {
        float* o = tbase + (long long)wm * 64 * 256;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(o + (i * 16 + li) * 256 + wn * 64 + j * 16 + g * 4) = acc[i][j];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(flags + tile * 4 + slice, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (wm == slice) {
        if (lane == 0)
            for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(flags + tile * 4 + (slice ^ 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch; ++spin) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_wave_barrier();
        const float* in = tbase + (long long)slice * 64 * 256;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(in + (i * 16 + li) * 256 + wn * 64 + j * 16 + g * 4));
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = m0 + wm * 64 + i * 16 + li;
#pragma unroll
            for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(out + (long long)row * N + n0 + wn * 64 + j * 16 + g * 4) = acc[i][j];
        }
    }
}
}  // namespace hulc_bf16

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int M = 2048, N = 2048, K = 2048;
    std::vector<h16_t> ha((size_t)M * K), hb((size_t)N * K);
    unsigned s = 1u;
    auto fill = [&](std::vector<h16_t>& h) { for (auto& v : h) { s = s * 1664525u + 1013904223u; const float f = ((int)(s >> 9) - (1 << 22)) * (1.f / (1 << 22)); unsigned u; memcpy(&u, &f, 4); v = (h16_t)((u + 0x8000u) >> 16); } };
    fill(ha); fill(hb);
    h16_t *a, *b; float *c, *c4, *scratch; int* flags;
    hipMalloc(&a, ha.size() * 2); hipMalloc(&b, hb.size() * 2); hipMalloc(&c, (size_t)M * N * 4); hipMalloc(&c4, (size_t)4 * M * N * 4);
    hipMalloc(&scratch, (size_t)64 * 4 * 4 * 64 * 256 * 4); hipMalloc(&flags, 1024 * sizeof(int)); hipMemset(flags, 0, 1024 * sizeof(int));
    hipMemcpy(a, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)gemm256_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute((const void*)gemm256_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    // reference: the production kernel
    EpiP ep{}; ep.out = c; ep.out_f32 = 1;
    launch_gemm_glds(0, dense<h16_t>(a, M, K), dense<h16_t>(b, N, K), dense_out(N), ep, M, N, K);
    hipDeviceSynchronize();
    std::vector<float> ref((size_t)M * N); hipMemcpy(ref.data(), c, ref.size() * 4, hipMemcpyDeviceToHost);
    // arm 1 correctness
    hipLaunchKernelGGL((gemm256_kernel<0>), dim3(256), dim3(512), 128 * 1024, 0, a, b, c4, scratch, flags, M, N, K, 0);
    hipDeviceSynchronize();
    {   std::vector<float> h((size_t)4 * M * N); hipMemcpy(h.data(), c4, h.size() * 4, hipMemcpyDeviceToHost);
        double dmax = 0, vmax = 0;
        for (size_t i = 0; i < ref.size(); ++i) { const double v = (double)h[i] + h[i + ref.size()] + h[i + 2 * ref.size()] + h[i + 3 * ref.size()]; dmax = fmax(dmax, fabs(v - ref[i])); vmax = fmax(vmax, fabs(ref[i])); }
        printf("arm 1 (partials summed on the host) vs gemm_glds_kernel: max |diff| %.3g of %.3g\n", dmax, vmax); }
    int epoch = 0, epoch2 = 0;
    hipMemset(c, 0, (size_t)M * N * 4);
    hipLaunchKernelGGL((gemm256_kernel<1>), dim3(256), dim3(512), 128 * 1024, 0, a, b, c, scratch, flags, M, N, K, ++epoch);
    if (hipDeviceSynchronize() != hipSuccess) { printf("arm 2 failed\n"); return 1; }
    {   std::vector<float> h((size_t)M * N); hipMemcpy(h.data(), c, h.size() * 4, hipMemcpyDeviceToHost);
        double dmax = 0, vmax = 0;
        for (size_t i = 0; i < ref.size(); ++i) { dmax = fmax(dmax, fabs((double)h[i] - ref[i])); vmax = fmax(vmax, fabs(ref[i])); }
        printf("arm 2 (reduce-scatter through L2) vs gemm_glds_kernel: max |diff| %.3g of %.3g\n", dmax, vmax); }
    hipFuncSetAttribute((const void*)gemm_s2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    hipFuncSetAttribute((const void*)gemm_s2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    hipMemset(c, 0, (size_t)M * N * 4);
    hipLaunchKernelGGL((gemm_s2_kernel<1>), dim3(256), dim3(512), 144 * 1024, 0, a, b, c, scratch, flags + 512, M, N, K, ++epoch2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("arm 4 failed\n"); return 1; }
    {   std::vector<float> h((size_t)M * N); hipMemcpy(h.data(), c, h.size() * 4, hipMemcpyDeviceToHost);
        double dmax = 0, vmax = 0;
        for (size_t i = 0; i < ref.size(); ++i) { dmax = fmax(dmax, fabs((double)h[i] - ref[i])); vmax = fmax(vmax, fabs(ref[i])); }
        printf("arm 4 (128 x 256 tiles, two slices, exchange through L2) vs gemm_glds_kernel: max |diff| %.3g of %.3g\n", dmax, vmax); }
    for (int rep = 0; rep < 3; ++rep) {
        float t[5];
        for (int arm = 0; arm < 5; ++arm) {
            auto launch = [&]() {
                if (arm == 0) launch_gemm_glds(0, dense<h16_t>(a, M, K), dense<h16_t>(b, N, K), dense_out(N), ep, M, N, K);
                else if (arm == 1) hipLaunchKernelGGL((gemm256_kernel<0>), dim3(256), dim3(512), 128 * 1024, 0, a, b, c4, scratch, flags, M, N, K, 0);
                else if (arm == 2) hipLaunchKernelGGL((gemm256_kernel<1>), dim3(256), dim3(512), 128 * 1024, 0, a, b, c, scratch, flags, M, N, K, ++epoch);
                else if (arm == 3) hipLaunchKernelGGL((gemm_s2_kernel<0>), dim3(256), dim3(512), 144 * 1024, 0, a, b, c4, scratch, flags, M, N, K, 0);
                else hipLaunchKernelGGL((gemm_s2_kernel<1>), dim3(256), dim3(512), 144 * 1024, 0, a, b, c, scratch, flags + 512, M, N, K, ++epoch2);
            };
            for (int i = 0; i < 5; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); t[arm] = ms * 1000 / 50;
        }
        printf("2048^3:  gemm_glds_kernel (128 x 128 x 2048 tiles) %6.2f us   256 x 256 x 512 slices, partials stored %6.2f us   ... reduced through L2 %6.2f us   |  128 x 256 x 1024 slices, partials stored %6.2f us   ... exchanged through L2 %6.2f us\n", t[0], t[1], t[2], t[3], t[4]);
    }
    return 0;
}
