// tools only (round 6): what would it cost the conv weight-gradient kernels to ADD their accumulator tile into a per-XCD slab with fp32 atomics instead of writing
// one slab per workgroup?  Today 256 workgroups x 128 KB (conv2) / 147 KB (conv3) per camera = 167 MB of slabs are written and then re-read by
// unpack_conv_wgrad_batched_kernel (55 us per step).  With 8 slabs per convolution (one per XCD: all adders of a slab share an L2) the slab sum would read 5 MB.
// Arms: plain store of the tile to the workgroup's own slab (today) | atomic add into slab[XCC_ID] | atomic add into slab[blockIdx % 8] (dispatch order: block b -> XCD b % 8)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/slab_atomic_probe.hip -o tools/bin/slab_atomic_probe && tools/bin/slab_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v)); return v; }
template <int MODE>
__global__ void __launch_bounds__(512) tile_out(float* __restrict__ part, int elems, int nslab_pad) {
    // a thread owns elems / 512 values of the tile in the accumulator layout of the wgrad kernels (4 consecutive floats per lane and MFMA tile)
    const int tid = threadIdx.x;
    float* dst;
    if (MODE == 0) dst = part + (long long)blockIdx.x * nslab_pad;
    else if (MODE == 1) dst = part + (long long)xcc_id() * nslab_pad;
    else dst = part + (long long)(blockIdx.x & 7) * nslab_pad;
    for (int i = tid * 4; i < elems; i += 512 * 4) {
        const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
        if (MODE == 0) *reinterpret_cast<float4*>(dst + i) = v;
        else { unsafeAtomicAdd(dst + i, v.x); unsafeAtomicAdd(dst + i + 1, v.y); unsafeAtomicAdd(dst + i + 2, v.z); unsafeAtomicAdd(dst + i + 3, v.w); }
    }
}
int main() {
    float* buf; hipMalloc(&buf, (size_t)256 * 40960 * 4 * 4); hipMemset(buf, 0, (size_t)256 * 40960 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int elems : {32768, 36864}) {
        auto run = [&](int mode) {
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(tile_out<0>, dim3(256), dim3(512), 0, 0, buf, elems, 40960);
                else if (mode == 1) hipLaunchKernelGGL(tile_out<1>, dim3(256), dim3(512), 0, 0, buf, elems, 40960);
                else hipLaunchKernelGGL(tile_out<2>, dim3(256), dim3(512), 0, 0, buf, elems, 40960);
            };
            for (int i = 0; i < 3; ++i) launch();
            hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            return ms * 1000 / 50;
        };
        printf("tile of %d fp32 per workgroup, 256 workgroups: plain store to own slab %6.2f us | atomics into slab[XCC_ID] %6.2f us | atomics into slab[block %% 8] %6.2f us\n",
               elems, run(0), run(1), run(2));
    }
    // correctness of the XCC-local form: every slab's sum over its adders
    hipMemset(buf, 0, (size_t)8 * 40960 * 4);
    hipLaunchKernelGGL(tile_out<1>, dim3(256), dim3(512), 0, 0, buf, 32768, 40960);
    hipDeviceSynchronize();
    static float h[8 * 40960];
    hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost);
    double tot = 0; for (int s = 0; s < 8; ++s) tot += h[s * 40960];
    printf("sum over the 8 slabs of element 0: %.0f (expect 256); slab 0 holds %.0f adders\n", tot, h[0]);
    return 0;
}
