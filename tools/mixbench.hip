// Micro-benchmark (tools only): what the memory system delivers for conv1's traffic shape — 983 MB of fp32 frames read once, 315 MB of bf16
// activations written (3.1 : 1), plain streaming kernels, no LDS, no MFMA.  The conv1 forward kernel's achievable HBM time is THIS, not the
// 8 TB/s headline: if it lands at ~5.2 TB/s the kernel is at its roofline, if at ~6.3 the kernel leaves bandwidth on the table.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// MODE 0: read only (sum).  1: write only.  2: read 4 x float4 (64 B) -> write 16 B... ratio R reads per write chosen by NR.
template <int MODE, int NR>
__global__ void __launch_bounds__(256) mixbench(const f32x4* __restrict__ in, u32x4* __restrict__ out, long long nread /*float4*/, long long nwrite /*u32x4*/, unsigned* sink) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x, gsz = (long long)gridDim.x * blockDim.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0) {
        for (long long i = gid; i < nread; i += gsz * NR) {
            f32x4 v[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) { const long long j = i + u * gsz; v[u] = j < nread ? in[j] : f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
            for (int u = 0; u < NR; ++u) acc += v[u];
        }
    } else if (MODE == 1) {
        for (long long i = gid; i < nwrite; i += gsz) out[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
    } else {
        // one "item" = NR float4 reads + 1 16-byte write; items strided over the grid
        const long long nitem = nwrite;
        for (long long it = gid; it < nitem; it += gsz) {
            f32x4 v[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) { const long long j = it + u * nitem; v[u] = j < nread ? in[j] : f32x4{0.f, 0.f, 0.f, 0.f}; }
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < NR; ++u) s += v[u];
            out[it] = u32x4{__float_as_uint(s[0]), __float_as_uint(s[1]), __float_as_uint(s[2]), __float_as_uint(s[3])};
        }
    }
    if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e30f) sink[0] = 1;
}

int main() {
    const long long rbytes = 2048ll * 3 * 200 * 200 * 4, wbytes = 2048ll * 49 * 49 * 32 * 2;
    const long long nread = rbytes / 16, nwrite = wbytes / 16;
    f32x4* in; u32x4* out; unsigned* sink;
    CHECK(hipMalloc(&in, rbytes)); CHECK(hipMalloc(&out, wbytes)); CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(in, 0, rbytes)); CHECK(hipMemset(out, 0, wbytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int grid, double bytes) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            for (int t = 0; t < 6; ++t) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, nread, nwrite, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-52s grid %5d  %7.1f us  %5.2f TB/s\n", name, grid, best / 6 * 1e3, bytes / (best / 6 * 1e-3) / 1e12);
    };
    for (int grid : {1024, 4096, 16384}) {
        run("read 983 MB (4 float4 in flight per thread)", mixbench<0, 4>, grid, (double)rbytes);
        run("read 983 MB (8 in flight)", mixbench<0, 8>, grid, (double)rbytes);
        run("write 315 MB", mixbench<1, 1>, grid, (double)wbytes);
        run("read 983 MB + write 315 MB interleaved (3 reads per write)", mixbench<2, 3>, grid, (double)(rbytes * 3 / 3.12 + wbytes));
    }
    return 0;
}
