// tools only: does the STRIDE between the 256 partial slabs of a conv weight gradient matter for the slab-sum launch (unpack_conv_wgrad_batched_kernel: 45 us for
// 167 MB = 3.7 TB/s)?  conv2's slab is 131072 B — a power of two: the 8 loads a thread keeps in flight (8 slabs, same offset) then differ only in address bits >= 17.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/slabsum_probe.hip -o tools/bin/slabsum_probe && tools/bin/slabsum_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void slabsum(const float* __restrict__ part, long long slab, int nsplit, int total, float* __restrict__ grad) {
    const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (idx >= total) return;
    const int per = (nsplit + gridDim.y - 1) / gridDim.y;
    const int z0 = blockIdx.y * per, z1 = min(nsplit, z0 + per);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto add4 = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    int z = z0;
    for (; z + 7 < z1; z += 8) {
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(part + (long long)(z + u) * slab + idx);
        add4(s0, t[0]); add4(s1, t[1]); add4(s2, t[2]); add4(s3, t[3]); add4(s0, t[4]); add4(s1, t[5]); add4(s2, t[6]); add4(s3, t[7]);
    }
    const float v[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
    for (int e = 0; e < 4; ++e) atomicAdd(grad + idx + e, v[e]);
}
int main() {
    const int nsplit = 256;
    float* buf; hipMalloc(&buf, (size_t)1 << 31); hipMemset(buf, 0, (size_t)1 << 31);      // 2 GB: eight rotating slab sets, colder than the 256 MB memory-side cache
    float* grad; hipMalloc(&grad, 1 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char* name; int total; long long slab; } cases[] = {
        {"conv2 32768 el, slab 131072 B (as today)", 32768, 32768}, {"conv2, slab + 256 B", 32768, 32768 + 64}, {"conv2, slab + 4352 B", 32768, 32768 + 1088},
        {"conv3 36864 el, slab 147456 B (as today)", 36864, 36864}, {"conv3, slab + 256 B", 36864, 36864 + 64}, {"conv3, slab + 4352 B", 36864, 36864 + 1088}};
    for (auto& c : cases)
        for (int yparts : {8, 16}) {
            const size_t set = (size_t)nsplit * c.slab + 1024;
            const int nset = (int)(((size_t)1 << 29) / set);
            const dim3 grid((c.total / 4 + 255) / 256, yparts);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(slabsum, grid, dim3(256), 0, 0, buf + (size_t)(i % nset) * set, c.slab, nsplit, c.total, grad);
            hipEventRecord(e0);
            const int reps = 40;
            for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(slabsum, grid, dim3(256), 0, 0, buf + (size_t)(i % nset) * set, c.slab, nsplit, c.total, grad);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1000 / reps, gb = (double)nsplit * c.total * 4 / 1e9;
            printf("%-44s y-parts %2d  %7.2f us  %6.0f GB/s\n", c.name, yparts, us, gb / (us * 1e-6));
        }
    return 0;
}
