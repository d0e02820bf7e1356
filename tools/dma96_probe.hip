// gfx950 LDS-DMA probe: where does lane i of global_load_lds_dwordx3 / dwordx4 land in LDS (M0 + i * 12 / 16?), and are 4-byte-aligned (not 16-byte
// aligned) global sources accepted?   hipcc --offload-arch=gfx950 -O2 -o /tmp/dma96 tools/dma96_probe.hip && /tmp/dma96
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) char lds_char;
__device__ __forceinline__ void dma12(const void* src, lds_char* dst) {
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
__device__ __forceinline__ void dma16(const void* src, lds_char* dst) {
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
__global__ void probe(const unsigned char* src, unsigned char* out, int off, int mode, int active) {
    __shared__ __attribute__((aligned(16))) char smem[4096];
    lds_char* l = (lds_char*)smem;
    const int lane = threadIdx.x;
    for (int i = lane; i < 4096; i += 64) l[i] = (char)0xEE;
    __syncthreads();
    if (lane < active) {
        if (mode == 12) dma12(src + off + lane * 12, l + 64);
        else dma16(src + off + lane * 16, l + 64);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = lane; i < 4096; i += 64) out[i] = (unsigned char)l[i];
}
int main() {
    std::vector<unsigned char> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = (unsigned char)((i * 7 + 3) & 0xff);
    unsigned char *d, *o;
    hipMalloc(&d, 8192); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    for (int mode : {12, 16}) for (int off : {0, 4, 8, 12, 20}) for (int active : {64, 37}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, off, mode, active);
        std::vector<unsigned char> r(4096);
        if (hipMemcpy(r.data(), o, 4096, hipMemcpyDeviceToHost) != hipSuccess) { printf("mode %d off %d: FAULT\n", mode, off); return 1; }
        int bad = 0, first = -1;
        for (int i = 0; i < active * mode; ++i) if (r[64 + i] != h[off + i]) { if (first < 0) first = i; ++bad; }
        int spill = 0;
        for (int i = 64 + active * mode; i < 4096; ++i) if (r[i] != 0xEE) ++spill;
        for (int i = 0; i < 64; ++i) if (r[i] != 0xEE) ++spill;
        printf("mode dwordx%d src offset %2d active %2d: %s (mismatches %d first %d, bytes written outside the lane-contiguous range %d)\n", mode / 4, off, active, bad || spill ? "DIFFERENT" : "contiguous lane*size OK", bad, first, spill);
    }
    return 0;
}
