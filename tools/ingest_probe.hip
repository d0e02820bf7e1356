// tools only: how fast can ONE workgroup per CU (512 threads) pull L2-resident data, by path?  gemm_glds_kernel ingests 1 MB per 128 x 128 tile at K = 2048 in ~28 us
// (~16 B/clk per CU) and neither a deeper ring nor fewer bytes per output moved it.  Arms, each workgroup streams `kb` KB that 16 workgroups share
// (a 16 MB working set for 256 workgroups: L2 / memory-side-cache resident, like a GEMM's operand panels):
//   dma   global_load_lds_dwordx4 only (1 KB per wave instruction), 3 x 32 KB ring, counted waits like the GEMM
//   vgpr  global_load_dwordx4 into registers only (16 B per lane, 1 KB per wave instruction, fully coalesced)
//   mix   half of the bytes by each path, issued interleaved
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ingest_probe.hip -o tools/bin/ingest_probe && tools/bin/ingest_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16(const void* src, unsigned ldsaddr) {
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)ldsaddr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
template <int MODE>
__global__ void __launch_bounds__(512) ingest(const u32x4* __restrict__ buf, size_t panel_vec, int steps, unsigned* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 16 workgroups share a panel (like the tiles of a GEMM row): panel = blockIdx.x / 16
    const u32x4* p = buf + (size_t)(blockIdx.x / 16) * panel_vec;
    const unsigned ldsbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)smem);
    unsigned acc = 0;
    // a step = 32 KB: 32 wave-instructions of 1 KB; wave w issues pieces w, w + 8, w + 16, w + 24
    for (int s = 0; s < steps; ++s) {
        const u32x4* sp = p + (size_t)s * 2048;           // 2048 vectors = 32 KB
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dma16(sp + (wave + 8 * j) * 64 + lane, ldsbase + ((s % 3) * 32768 + (wave + 8 * j) * 1024));
            if (s >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (MODE == 1) {
            u32x4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = sp[(wave + 8 * j) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc ^= v[j][0] ^ v[j][3];
        } else {
            u32x4 v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                dma16(sp + (wave + 8 * j) * 64 + lane, ldsbase + ((s % 3) * 32768 + (wave + 8 * j) * 1024));
                v[j] = sp[(wave + 8 * (j + 2)) * 64 + lane];
            }
            acc ^= v[0][0] ^ v[1][3];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc ^= *(__attribute__((address_space(3))) unsigned*)(smem + tid * 4);
    if (acc == 0x12345678u) sink[0] = acc;
}
int main() {
    const int steps = 32;                                   // 32 x 32 KB = 1 MB per workgroup
    const size_t panel_vec = (size_t)steps * 2048;          // 1 MB panels, 16 of them
    u32x4* buf; hipMalloc(&buf, 16 * panel_vec * 16); hipMemset(buf, 1, 16 * panel_vec * 16);
    unsigned* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)ingest<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)ingest<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    hipFuncSetAttribute((const void*)ingest<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    const char* names[3] = {"LDS-DMA only", "global -> VGPR only", "half / half interleaved"};
    for (int mode = 0; mode < 3; ++mode) {
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(ingest<0>, dim3(256), dim3(512), 96 * 1024, 0, buf, panel_vec, steps, sink);
            if (mode == 1) hipLaunchKernelGGL(ingest<1>, dim3(256), dim3(512), 96 * 1024, 0, buf, panel_vec, steps, sink);
            if (mode == 2) hipLaunchKernelGGL(ingest<2>, dim3(256), dim3(512), 96 * 1024, 0, buf, panel_vec, steps, sink);
        };
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 50; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1000 / 50;
        printf("%-26s 1 MB per workgroup x 256: %7.2f us  -> %5.1f GB/s per CU, %5.1f B/clk at 2.4 GHz\n", names[mode], us, 1.048576e-3 / (us * 1e-6), 1048576.0 / (us * 1e-6) / 2.4e9);
    }
    return 0;
}
