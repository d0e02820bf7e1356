// Micro-benchmark (tools only): per-CU load throughput of L2/MALL-resident data for different wave access shapes.
// 256 workgroups x 512 threads, each pulling BYTES of a shared 8 MB + 256 KB working set (the recurrent skinny GEMM's footprint).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// mode 0: coalesced (each wave instruction = 1 KB contiguous)      mode 1: fragment-shaped (16 rows x 64 B, row stride 4 KB)
// mode 2: fragment-shaped, 32 B per row (8 rows... like a 16x16x16)  mode 3: 128 B per row (8 lanes per row, 8 rows)
template <int MODE, int NLOAD>
__global__ void __launch_bounds__(512) loadbench(const char* __restrict__ buf, unsigned* __restrict__ out, int rows_per_wg, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the workgroup owns rows [r0, r0 + rows_per_wg) of a [2112][4096 B] matrix (like W slice + A rows)
    const int r0 = (blockIdx.x * 16) % 2048;
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v[NLOAD];
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) {
            const int idx = (it * NLOAD + u) * 8 + wave;            // wave-instruction index within the workgroup
            long long off;
            if (MODE == 0) {                                          // 1 KB contiguous: row = idx / 4, quarter = idx % 4
                off = (long long)(r0 + (idx >> 2) % rows_per_wg) * 4096 + (idx & 3) * 1024 + lane * 16;
            } else if (MODE == 1) {                                   // 16 rows x 64 B
                const int rb = (idx >> 6) * 16, kc = idx & 63;
                off = (long long)(r0 + (rb + (lane & 15)) % rows_per_wg) * 4096 + kc * 64 + (lane >> 4) * 16;
            } else {                                                  // 8 rows x 128 B
                const int rb = (idx >> 5) * 8, kc = idx & 31;
                off = (long long)(r0 + (rb + (lane >> 3)) % rows_per_wg) * 4096 + kc * 128 + (lane & 7) * 16;
            }
            v[u] = *reinterpret_cast<const u32x4*>(buf + off);
        }
#pragma unroll
        for (int u = 0; u < NLOAD; ++u) acc += v[u];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) out[blockIdx.x] = 1;
}

int main() {
    char* buf; unsigned* out;
    const size_t bytes = 2112ull * 4096 + (1 << 20);
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 1, bytes)); CHECK(hipMalloc(&out, 4096));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int rows, int iters, int nload) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int t = 0; t < 32; ++t) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, buf, out, rows, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double kb = (double)iters * nload * 8 * 1024 / 1024.0;
        printf("%-34s rows %3d  %6.0f KB/WG  %6.2f us/launch  %5.1f B/clk/CU @2.1GHz\n", name, rows, kb, ms / 32 * 1e3, kb * 1024 / (ms / 32 * 1e-3) / 2.1e9);
        return 0;
    };
    // 192 KB per workgroup = 48 rows x 4 KB, as the recurrent skinny GEMM; 24 wave-instructions per wave
    run("coalesced 1KB, 24 in flight", loadbench<0, 24>, 48, 1, 24);
    run("coalesced 1KB, 12 x2", loadbench<0, 12>, 48, 2, 12);
    run("fragment 16x64B, 24 in flight", loadbench<1, 24>, 48, 1, 24);
    run("fragment 16x64B, 12 x2", loadbench<1, 12>, 48, 2, 12);
    run("8 rows x 128B, 24 in flight", loadbench<2, 24>, 48, 1, 24);
    run("coalesced, 64 KB/WG", loadbench<0, 8>, 16, 1, 8);
    run("fragment, 64 KB/WG", loadbench<1, 8>, 16, 1, 8);
    run("coalesced, 384 KB/WG", loadbench<0, 24>, 96, 2, 24);
    run("fragment, 384 KB/WG", loadbench<1, 24>, 96, 2, 24);
    run("coalesced, 1.5 MB/WG", loadbench<0, 24>, 384, 8, 24);
    run("fragment, 1.5 MB/WG", loadbench<1, 24>, 384, 8, 24);
    run("8x128B, 1.5 MB/WG", loadbench<2, 24>, 384, 8, 24);
    run("empty-ish (1 load)", loadbench<0, 1>, 1, 1, 1);
    return 0;
}
