// tools only: what does the END of a kernel cost when its output is still dirty in the XCDs' L2s?  The kernel trace of a step shows 5 - 11 us
// between the end of every conv kernel (hundreds of MB of output, the last tens of MB still in L2) and the start of the next launch, and 0 us
// between small launches.  Arms: a writer kernel that stores `mb` MB (plain / non-temporal / sc0 sc1 write-through stores) followed by a
// one-wave kernel; the pair is timed back-to-back 200 times (events), and the writer alone via its own duration = pair - trivial pair.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/l2flush_probe.hip -o tools/bin/l2flush_probe && tools/bin/l2flush_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE> __global__ void __launch_bounds__(256) writer(f4* __restrict__ out, size_t nvec, float v) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        f4 x = {v, v + 1, v + 2, v + 3};
        if (MODE == 0) out[i] = x;
        else if (MODE == 1) __builtin_nontemporal_store(x, out + i);
        else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(out + i), "v"(x) : "memory");
    }
}
__global__ void tiny(float* p) { if (threadIdx.x == 0) p[0] += 1.f; }

int main() {
    const size_t cap = (size_t)1 << 30;
    f4* buf; hipMalloc(&buf, cap); float* t; hipMalloc(&t, 64); hipMemset(t, 0, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 200;
    const char* names[3] = {"plain stores", "non-temporal stores", "sc0 sc1 (write-through) stores"};
    printf("%-34s %8s %12s %12s %12s\n", "writer", "MB", "pair us", "writer-only", "tiny-only");
    float tiny_us = 0;
    { for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, t);
      hipEventRecord(e0); for (int i = 0; i < reps * 2; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, t); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); tiny_us = ms * 1000 / (reps * 2); }
    for (int mode = 0; mode < 3; ++mode)
        for (size_t mb : {4, 16, 32, 64, 256, 1024}) {
            const size_t nvec = mb * 1024 * 1024 / 16;
            auto launch = [&](bool with_tiny) {
                if (mode == 0) hipLaunchKernelGGL(writer<0>, dim3(2048), dim3(256), 0, 0, buf, nvec, 1.f);
                if (mode == 1) hipLaunchKernelGGL(writer<1>, dim3(2048), dim3(256), 0, 0, buf, nvec, 1.f);
                if (mode == 2) hipLaunchKernelGGL(writer<2>, dim3(2048), dim3(256), 0, 0, buf, nvec, 1.f);
                if (with_tiny) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, 0, t);
            };
            float us[2];
            for (int w = 0; w < 2; ++w) {
                for (int i = 0; i < 10; ++i) launch(w);
                hipEventRecord(e0); for (int i = 0; i < reps; ++i) launch(w); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); us[w] = ms * 1000 / reps;
            }
            printf("%-34s %8zu %12.2f %12.2f %12.2f   -> %.0f GB/s writer-only\n", names[mode], mb, us[1], us[0], tiny_us, mb * 1.048576e-3 / (us[0] * 1e-6) * 1e-0 / 1e0);
        }
    return 0;
}
