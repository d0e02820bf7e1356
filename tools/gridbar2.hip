// Micro-benchmark (tools only), follow-up to tools/gridbar.hip with the FAST primitives of MI355X_MICROARCH.md's price list:
//   * barrier-xcd: XCD-hierarchical grid barrier — per-XCD arrival counter, the XCD's last arriver joins a top-level counter, the top-level
//     last arriver bumps a generation word per XCD; everybody polls ONE word with relaxed (sc1) loads + s_sleep and does ONE agent acquire
//     after the match (the round-1 probe polled a single counter with ACQUIRE loads = the 13 us row of the table);
//   * publish with write-through `sc1` stores (8-byte agent-scope relaxed stores: no release fence, every storing wave drains its vmcnt);
//   * what one step of a persistent recurrent kernel then costs: publish the workgroup's 32 x 16 tile of H_t (1 KB bf16), barrier,
//     read the 128 KB of H_t its next step multiplies (32 rows x 2048 k) with 16-byte loads, 8 in flight per thread.
// Compare with the launch-per-step kernel: 6.0 us per step including its 4.1 us operand stream.
//   hipcc --offload-arch=gfx950 -O3 tools/gridbar2.hip -o tools/bin/gridbar2
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

struct Bar {
    unsigned xcd_arrive[8 * 32];     // one 128-byte line per XCD
    unsigned xcd_gen[8 * 32];
    unsigned top[32];
    unsigned xcd_pop[8 * 32];        // census: workgroups resident on each XCD
    unsigned census_done[32];
    unsigned err[32];
};

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 7u; }

// placement-independent: counts come from the census, nothing assumes block -> XCD mapping
__device__ __forceinline__ bool barrier_xcd(Bar* b, unsigned xcd, unsigned round, unsigned nxcd_active) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const unsigned pop = __hip_atomic_load(&b->xcd_pop[xcd * 32], RLX_AGENT);
        const unsigned prev = __hip_atomic_fetch_add(&b->xcd_arrive[xcd * 32], 1u, RLX_AGENT);
        if (prev + 1 == pop * (round + 1)) {                              // this XCD's last arriver: its leader for this round
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tprev = __hip_atomic_fetch_add(&b->top[0], 1u, RLX_AGENT);
            if (tprev + 1 == nxcd_active * (round + 1)) {                 // last XCD: release everybody
                for (unsigned x = 0; x < 8; ++x) __hip_atomic_store(&b->xcd_gen[x * 32], round + 1, RLX_AGENT);
            }
        }
        int spins = 0;
        while (__hip_atomic_load(&b->xcd_gen[xcd * 32], RLX_AGENT) < round + 1) {
            if (++spins > (1 << 22)) { __hip_atomic_store(&b->err[0], 1u, RLX_AGENT); ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

template <int MODE>   // 0: barrier only; 1: publish 1 KB (sc1) + barrier + read rd_kb KB
__global__ void __launch_bounds__(1024) persist(Bar* b, u64* hbuf /* 2 x 256 KB */, int rounds, int rd_kb, unsigned* out) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wg = blockIdx.x;
    __shared__ unsigned s_xcd, s_nx;
    if (tid == 0) {                                                       // census (once): how many workgroups live on my XCD, how many XCDs are populated
        const unsigned x = xcc_id();
        __hip_atomic_fetch_add(&b->xcd_pop[x * 32], 1u, RLX_AGENT);
        __hip_atomic_fetch_add(&b->census_done[0], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(&b->census_done[0], RLX_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned nx = 0;
        for (unsigned i = 0; i < 8; ++i) nx += __hip_atomic_load(&b->xcd_pop[i * 32], RLX_AGENT) > 0;
        s_xcd = x; s_nx = nx;
    }
    __syncthreads();
    const unsigned xcd = s_xcd, nx = s_nx;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 1) {
            u64* dst = hbuf + (size_t)((r + 1) & 1) * 32768 + wg * 128;   // my 1 KB tile of H_{t}: 128 x 8-byte write-through stores (two waves)
            if (tid < 128) __hip_atomic_store(dst + tid, ((u64)(unsigned)r << 32) | (unsigned)(acc.x + tid), RLX_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains before the workgroup arrives
        }
        if (!barrier_xcd(b, xcd, (unsigned)r, nx)) break;
        if (MODE == 1) {
            const u32x4* src = reinterpret_cast<const u32x4*>(hbuf + (size_t)((r + 1) & 1) * 32768);      // 256 KB = 16384 x 16 B
            const int half = (wg & 1) * 8192;                             // the 32 rows (128 KB) this workgroup's tile multiplies
            for (int p = 0; p < rd_kb / 16; p += 8) {
                u32x4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = src[half + ((p + u) * 1024 + tid) % 8192];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc += v[u];
            }
        }
    }
    if (tid == 0) out[wg] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    Bar* bar; u64* hbuf; unsigned* out;
    CHECK(hipMalloc(&bar, sizeof(Bar))); CHECK(hipMalloc(&hbuf, 2 * 32768 * 8)); CHECK(hipMalloc(&out, 4096));
    CHECK(hipMemset(hbuf, 1, 2 * 32768 * 8));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)persist<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void*)persist<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const int rounds = 500;
    for (int mode = 0; mode < 2; ++mode)
        for (int rd : {0, 64, 128}) {
            if ((mode == 0) != (rd == 0)) continue;
            float best = 1e9f; unsigned herr = 0, pop[8 * 32];
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(bar, 0, sizeof(Bar)));
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(persist<0>, dim3(256), dim3(1024), 128 * 1024, 0, bar, hbuf, rounds, rd, out);
                else hipLaunchKernelGGL(persist<1>, dim3(256), dim3(1024), 128 * 1024, 0, bar, hbuf, rounds, rd, out);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CHECK(hipMemcpy(&herr, bar->err, 4, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(pop, bar->xcd_pop, sizeof(pop), hipMemcpyDeviceToHost));
            }
            printf("256 workgroups x 1024 threads, %s%s: %.2f us per round%s   (workgroups per XCD:", mode ? "publish 1 KB (sc1) + barrier-xcd + read " : "barrier-xcd only",
                   mode ? (rd == 64 ? "64 KB" : "128 KB") : "", best * 1e3f / rounds, herr ? "  (SPIN LIMIT HIT)" : "");
            for (int x = 0; x < 8; ++x) printf(" %u", pop[x * 32]);
            printf(")\n");
        }
    return 0;
}
