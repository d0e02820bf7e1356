// Micro-benchmark (tools only): what does one step of a PERSISTENT recurrent kernel cost on MI355X?
// 256 workgroups x 1024 threads (one per CU, 128 KB LDS each so no two share a CU).  Each round every workgroup
//   (a) pulls `rd_kb` KB of the buffer the OTHER workgroups wrote in the previous round (agent-scope visible),
//   (b) writes its own 1 KB slice with agent-scope release,
//   (c) joins a grid barrier (monotonic counter in device memory, agent-scope atomics, bounded spin).
// Prints microseconds per round for: barrier only, barrier + exchange.  Compare with the ~2.7 us dependent-launch floor
// (tools/loadbench.hip) + the 4.1 us operand stream of the launch-per-step kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1 << 22) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
    return ok;
}

template <int MODE>   // 0: barrier only; 1: + exchange
__global__ void __launch_bounds__(1024) persist(unsigned* ctr, unsigned* err, u32x4* buf /* 2 x [256][64] u32x4 = 2 x 256 KB */, int rounds, int rd_kb, unsigned* out) {
    extern __shared__ char smem[];
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        if (MODE == 1) {
            const u32x4* src = buf + (size_t)(r & 1) * 256 * 64;          // previous round's slices
            u32x4* dst = buf + (size_t)((r + 1) & 1) * 256 * 64;
            // read rd_kb KB: 1024 threads x 16 B = 16 KB per pass
            for (int p = 0; p < rd_kb / 16; ++p) {
                const int i = (p * 1024 + tid + wg * 64) % (256 * 64);
                acc += src[i];       // ordinary loads: the barrier's agent-scope acquire invalidated this CU's L1 / the XCD's non-local L2 lines
            }
            if (tid < 64) { u32x4 w = acc; w.x += r; dst[wg * 64 + tid] = w; }
            __threadfence();
        }
        if (!grid_barrier(ctr, (unsigned)(r + 1) * nwg, err)) break;
    }
    if (tid == 0) out[wg] = acc.x + acc.y + acc.z + acc.w;
}

int main() {
    unsigned *ctr, *err, *out; u32x4* buf;
    CHECK(hipMalloc(&ctr, 64)); CHECK(hipMalloc(&err, 64)); CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&buf, 2 * 256 * 64 * 16));
    CHECK(hipMemset(buf, 1, 2 * 256 * 64 * 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)persist<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    CHECK(hipFuncSetAttribute((const void*)persist<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    const int rounds = 500;
    for (int nwg : {256, 64}) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int rd : {16, 64, 128}) {
                if (mode == 0 && rd != 16) continue;
                float best = 1e9f; unsigned herr = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipMemset(ctr, 0, 64)); CHECK(hipMemset(err, 0, 64));
                    CHECK(hipEventRecord(e0));
                    if (mode == 0) hipLaunchKernelGGL(persist<0>, dim3(nwg), dim3(1024), 128 * 1024, 0, ctr, err, buf, rounds, rd, out);
                    else hipLaunchKernelGGL(persist<1>, dim3(nwg), dim3(1024), 128 * 1024, 0, ctr, err, buf, rounds, rd, out);
                    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                    CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
                }
                printf("nwg=%d mode=%s rd=%3d KB/wg : %.2f us per round%s\n", nwg, mode ? "barrier+exchange" : "barrier only", mode ? rd : 0, best * 1e3f / rounds, herr ? "  (SPIN LIMIT HIT)" : "");
            }
        }
    }
    return 0;
}
