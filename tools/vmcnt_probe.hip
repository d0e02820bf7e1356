// tools only: does gfx950 retire a wave's vector-memory operations IN ORDER with respect to vmcnt — an LDS-DMA load (global_load_lds) issued
// BEFORE a store, waited for with a COUNTED s_waitcnt vmcnt(1)?  If the counter could drop for the (fast, L2-hit) store while the (slow,
// HBM-cold) DMA is still in flight, the wait would pass early and the ds_read below would see the sentinel instead of the data.
// gfx9-family hardware is documented to return loads and stores of a wave in issue order (one counter, no vscnt before gfx10) and LLVM's
// waitcnt pass relies on it for ordinary loads/stores; the LDS-DMA form is what this probe adds.  Control arm: vmcnt(2) (no wait) must FAIL —
// it proves the probe can see an early pass.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/vmcnt_probe.hip -o tools/bin/vmcnt_probe && tools/bin/vmcnt_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one kernel per arm (the wait count is part of the asm text)
#define PROBE_KERNEL(NAME, BODY)                                                                                                                  \
    __global__ void __launch_bounds__(256) NAME(const u32x4* __restrict__ big, size_t nvec, unsigned* __restrict__ hot, unsigned* __restrict__ fails, int iters) { \
        __shared__ __attribute__((aligned(16))) u32x4 lds[256];                                                                                   \
        const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                                                            \
        const unsigned ldsbase = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) u32x4*)lds + wave * 1024);                                  \
        const unsigned myaddr = ldsbase + lane * 16;                                                                                              \
        unsigned bad = 0;                                                                                                                         \
        size_t idx = ((size_t)blockIdx.x * 977 + wave * 131) * 64 % (nvec - 64);                                                                  \
        unsigned* hp = hot + ((size_t)blockIdx.x * 256 + tid) * 4;                                                                                \
        for (int it = 0; it < iters; ++it) {                                                                                                      \
            idx = (idx * 2862933555777941757ull + 3037000493ull) % (nvec - 64);                                                                   \
            idx &= ~(size_t)63;                                                                                                                   \
            const u32x4* src = big + idx + lane;                                                                                                  \
            u32x4 got;                                                                                                                            \
            const u32x4 sent = {0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};                                                              \
            *(__attribute__((address_space(3))) u32x4*)(size_t)myaddr = sent;                                                                     \
            unsigned sval = (unsigned)it;                                                                                                         \
            asm volatile(BODY : "=&v"(got) : "v"(src), "v"(hp), "s"(ldsbase), "v"(sval), "v"(myaddr) : "memory", "m0");                            \
            const unsigned e = (unsigned)((idx + lane) * 4);                                                                                      \
            if (got[0] != e || got[1] != e + 1 || got[2] != e + 2 || got[3] != e + 3) ++bad;                                                      \
        }                                                                                                                                         \
        if (bad) atomicAdd(fails, bad);                                                                                                           \
    }

PROBE_KERNEL(probe_dma_store_vm1,
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_store_dword %2, %4, off\n\t"
    "s_waitcnt vmcnt(1)\n\tds_read_b128 %0, %5\n\ts_waitcnt lgkmcnt(0)\n\t")
PROBE_KERNEL(probe_control_vm2,
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_store_dword %2, %4, off\n\t"
    "s_waitcnt vmcnt(2)\n\tds_read_b128 %0, %5\n\ts_waitcnt lgkmcnt(0)\n\t")
PROBE_KERNEL(probe_dma_4stores_vm4,
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_store_dword %2, %4, off\n\t"
    "global_store_dword %2, %4, off offset:4\n\tglobal_store_dword %2, %4, off offset:8\n\tglobal_store_dword %2, %4, off offset:12\n\t"
    "s_waitcnt vmcnt(4)\n\tds_read_b128 %0, %5\n\ts_waitcnt lgkmcnt(0)\n\t")
PROBE_KERNEL(probe_dma_vm0,
    "s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_store_dword %2, %4, off\n\t"
    "s_waitcnt vmcnt(0)\n\tds_read_b128 %0, %5\n\ts_waitcnt lgkmcnt(0)\n\t")

int main() {
    const size_t nvec = (size_t)1 << 27;            // 2 GB of 16-byte vectors: every random 1 KB run is HBM-cold
    u32x4* big; hipMalloc(&big, nvec * 16);
    {   // big[i] = {4i, 4i+1, 4i+2, 4i+3}
        std::vector<unsigned> h((size_t)1 << 24);
        for (size_t off = 0; off < nvec * 4; off += h.size()) {
            for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(off + i);
            hipMemcpy((unsigned*)big + off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        }
    }
    unsigned *hot, *fails; hipMalloc(&hot, 1024 * 256 * 16); hipMalloc(&fails, 16);
    const int iters = 2000, grid = 1024;
    struct Arm { const char* name; void (*k)(const u32x4*, size_t, unsigned*, unsigned*, int); } arms[] = {
        {"DMA(cold) ; store(hot) ; vmcnt(0)  [sanity: must be 0]", probe_dma_vm0},
        {"DMA(cold) ; store(hot) ; vmcnt(1)  [the question]", probe_dma_store_vm1},
        {"DMA(cold) ; 4 stores(hot) ; vmcnt(4)  [the question]", probe_dma_4stores_vm4},
        {"DMA(cold) ; store(hot) ; vmcnt(2)  [control: no wait, must FAIL]", probe_control_vm2}};
    for (auto& a : arms) {
        hipMemset(fails, 0, 16);
        hipLaunchKernelGGL(a.k, dim3(grid), dim3(256), 0, 0, big, nvec, hot, fails, iters);
        hipDeviceSynchronize();
        unsigned f = 0; hipMemcpy(&f, fails, 4, hipMemcpyDeviceToHost);
        printf("%-70s early passes: %u of %llu lane-checks  (%s)\n", a.name, f, (unsigned long long)grid * 256 * iters, hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
