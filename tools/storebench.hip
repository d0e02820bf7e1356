// Micro-benchmark (tools only): HBM WRITE throughput of a 315 MB output (conv2's data gradient: 2048 frames x 49 x 49 pixels x 64 B) for the
// lane -> address shapes a convolution epilogue can produce.  512 workgroups x 256 threads (two per CU), 16-byte stores.
//   0  coalesced: a wave instruction writes 1 KB contiguous
//   1  conv_reg's parity-class epilogue: lane (pixel lj, half h) writes 2 x 16 B at pixel(2 lj + pw) * 64 + 32 h (+0, +16): 32 lines per instruction, 2 pieces each
//   2  tile-contiguous pixels: pixel(lj) * 64 + 32 h (+0, +16): 16 lines per instruction, 4 pieces each
//   3  a lane owns a whole 64-byte pixel (4 stores at +0 .. +48), pixels of a wave contiguous: 32 lines per instruction, ONE 16-byte piece per line and instruction
//   4  like 1 but both parity classes from one wave back to back (the line's two halves 1 instruction apart)
//   5  64 B per lane-quad: lane l writes 16 B at (l / 4) * 128 + (l % 4) * 16 (half lines, the other half one instruction later)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256, 2) storebench(char* __restrict__ out, long long npix, int nt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lj = lane & 31, h = lane >> 5;
    const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    const u32x4 v = {(unsigned)lane, 1u, 2u, 3u};
    const long long ntile = npix / 64;                      // 64 pixels (4 KB) per tile
    for (long long t = gw; t < ntile; t += nw) {
        char* base = out + t * 4096;
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(base + k * 1024 + lane * 16) = v;
        } else if (MODE == 1) {
            // the tile's 64 pixels = 32 pixel pairs; this wave instance plays parity class pw = t & 1 of a DIFFERENT wave's twin: emulate by writing class (t & 1)
            // of tile t / 2 * 2 ... simpler: two passes far apart in time: class 0 of every tile first, class 1 in a second sweep
        } else if (MODE == 2) {
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                char* p = base + mm * 2048 + lj * 64 + 32 * h;
                *reinterpret_cast<u32x4*>(p) = v; *reinterpret_cast<u32x4*>(p + 16) = v;
            }
        } else if (MODE == 3) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(base + lane * 64 + k * 16) = v;
        } else if (MODE == 4) {
#pragma unroll
            for (int pw = 0; pw < 2; ++pw) {
                char* p = base + (2 * lj + pw) * 64 + 32 * h;
                *reinterpret_cast<u32x4*>(p) = v; *reinterpret_cast<u32x4*>(p + 16) = v;
            }
        } else if (MODE == 5) {
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4*>(base + (k >> 1) * 2048 + (lane >> 2) * 128 + (k & 1) * 64 + (lane & 3) * 16) = v;
        }
    }
    if (MODE == 1) {
        for (int pw = 0; pw < 2; ++pw)
            for (long long t = gw; t < ntile; t += nw) {
                char* p = out + t * 4096 + (2 * lj + pw) * 64 + 32 * h;
                *reinterpret_cast<u32x4*>(p) = v; *reinterpret_cast<u32x4*>(p + 16) = v;
            }
    }
}

int main() {
    char* buf;
    const long long npix = 2048ll * 49 * 49 / 64 * 64;
    const size_t bytes = (size_t)npix * 64;
    CHECK(hipMalloc(&buf, bytes + 4096)); CHECK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](const char* name, auto kern, int grid) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            for (int t = 0; t < 8; ++t) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, buf, npix, 0);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-64s grid %4d  %7.1f us  %5.2f TB/s\n", name, grid, best / 8 * 1e3, bytes / (best / 8 * 1e-3) / 1e12);
        return 0;
    };
    for (int grid : {512, 2048}) {
        run("0 coalesced 1 KB per instruction", storebench<0>, grid);
        run("1 parity classes far apart (2 half-line pieces per line, instr)", storebench<1>, grid);
        run("2 tile-contiguous pixels (32 B per lane)", storebench<2>, grid);
        run("3 64 B per lane (one piece per line per instruction)", storebench<3>, grid);
        run("4 both parity classes back to back", storebench<4>, grid);
        run("5 64 B per lane quad", storebench<5>, grid);
    }
    return 0;
}
