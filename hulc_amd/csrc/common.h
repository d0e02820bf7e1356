// hulc_amd/csrc/common.h — shared device helpers for the gfx950 (MI355X) HULC training-step kernels.
// Wave = 64 lanes everywhere.  The 16-bit compute type of the half-precision engines is selected PER TRANSLATION UNIT:
//   capi.hip         (no macro)          -> bf16  (namespace hulc_bf16; also holds the fp32 parity engine)
//   engine_f16.hip   (-DHULC_HALF_F16)   -> IEEE fp16 (namespace hulc_f16; the reference's `precision: 16`,
//                                           conf/trainer/play_trainer.yaml:3, run with dynamic loss scaling)
// Both store the type as raw uint16 (h16_t); only the conversions and the MFMA opcode differ
// (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x32_f16 share one fragment layout), so every kernel is written once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

void hulc_set_error(const char* fmt, ...);

#define HIP_CHECK(x)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (x);                                                                             \
        if (e_ != hipSuccess) {                                                                          \
            hulc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #x, hipGetErrorString(e_));            \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)

#define DEVI __device__ __forceinline__

// One LDS-DMA instruction (global_load_lds_dwordx4: 64 lanes x 16 B from per-lane global addresses -> 1 KB at the wave-uniform LDS address `lds_addr`), issued as
// inline assembly.  Round 5 (tools/gemm_probe.hip): gemm_glds_kernel at 2048^3 takes 28.7 us with __builtin_amdgcn_global_load_lds and 25.6 us with this form — the
// same instructions, but the compiler schedules around the builtin (it drops the wait state behind `s_mov m0` by moving a VALU into it and orders the next piece's
// address arithmetic in between).  The DMA is invisible to LLVM's waitcnt pass this way: every consumer waits with its OWN `s_waitcnt vmcnt(N)` + barrier before it
// reads the LDS data (all kernels here do); compiler-inserted vmcnt waits stay safe, unknown extra operations can only make them wait longer.
// Used by the GEMM family (gemm.h); conv_reg.h keeps the builtin (tools/cr_bench.hip built both ways: no difference there — its DMA phase is bound by the HBM stream).
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
__device__ __forceinline__ void lds_dma16(const void* src, unsigned lds_addr) {
    const unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(a) : "memory", "m0");
}
#endif

// Kernel-routing / debugging switches.  The PRODUCTION build has none: every HULC_SWITCH("NAME", default) is the compile-time constant
// `default`, so the library contains exactly one code path per launch site (an environment switch per A/B experiment doubled
// an untested path each).  An experiment build (`HULC_BUILD_AB=1 python -c "import __graft_entry__ as g; g.build()"`, i.e. -DHULC_AB_SWITCHES)
// reads the same names from the environment once per process; tools/ab_env.sh needs that build.
#ifdef HULC_AB_SWITCHES
#include <cstdlib>
static inline int hulc_switch_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define HULC_SWITCH(name, dflt) hulc_switch_env(name, dflt)
#else
#define HULC_SWITCH(name, dflt) (dflt)
#endif

#ifdef HULC_HALF_F16
#define HULC_NS hulc_f16
#define HULC_HALF_NAME "fp16"
#else
#define HULC_NS hulc_bf16
#define HULC_HALF_NAME "bf16"
#endif

namespace HULC_NS {

typedef uint16_t h16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#ifdef HULC_HALF_F16
typedef _Float16 h16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2_t __attribute__((ext_vector_type(2)));
#define MFMA_16x16x32_H __builtin_amdgcn_mfma_f32_16x16x32_f16
DEVI float h2f(h16_t x) { return (float)*reinterpret_cast<const _Float16*>(&x); }
// fp32 -> fp16, round to nearest even; overflow -> inf (caught by the loss scaler's non-finite check)
DEVI h16_t f2h(float f) {
    _Float16 b = (_Float16)f;
    return *reinterpret_cast<h16_t*>(&b);
}
#else
typedef __bf16 h16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 h16x2_t __attribute__((ext_vector_type(2)));
#define MFMA_16x16x32_H __builtin_amdgcn_mfma_f32_16x16x32_bf16
DEVI float h2f(h16_t x) { return __uint_as_float(((uint32_t)x) << 16); }
// fp32 -> bf16, round to nearest even: gfx950 has it in hardware (v_cvt_pk_bf16_f32, two values per instruction)
DEVI h16_t f2h(float f) {
    __bf16 b = (__bf16)f;
    return *reinterpret_cast<h16_t*>(&b);
}
#endif
DEVI unsigned pack2h(float lo, float hi) {
    h16x2_t b = __builtin_convertvector(f32x2_t{lo, hi}, h16x2_t);
    return *reinterpret_cast<unsigned*>(&b);
}
// the two halves of a packed pair (low = first element)
DEVI float h2f_lo(unsigned w) { return h2f((h16_t)(w & 0xffffu)); }
DEVI float h2f_hi(unsigned w) {
#ifdef HULC_HALF_F16
    return h2f((h16_t)(w >> 16));
#else
    return __uint_as_float(w & 0xffff0000u);
#endif
}
template <typename T> DEVI float to_f(T x);
template <> DEVI float to_f<float>(float x) { return x; }
template <> DEVI float to_f<h16_t>(h16_t x) { return h2f(x); }
template <typename T> DEVI T from_f(float x);
template <> DEVI float from_f<float>(float x) { return x; }
template <> DEVI h16_t from_f<h16_t>(float x) { return f2h(x); }

// 8 contiguous elements of T (16 B for bf16, 32 B for fp32)
template <typename T> struct Vec8 { T v[8]; };

template <typename T> DEVI void load8(const T* p, T (&v)[8]);
template <> DEVI void load8<h16_t>(const h16_t* p, h16_t (&v)[8]) {
    if ((((uintptr_t)p) & 15) == 0) {
        uint4 u = *reinterpret_cast<const uint4*>(p);
        *reinterpret_cast<uint4*>(v) = u;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[i];
    }
}
template <> DEVI void load8<float>(const float* p, float (&v)[8]) {
    if ((((uintptr_t)p) & 15) == 0) {
        float4 a = reinterpret_cast<const float4*>(p)[0];
        float4 b = reinterpret_cast<const float4*>(p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = p[i];
    }
}
template <typename T> DEVI void load8_guard(const T* p, int nvalid, T (&v)[8]) {
    if (nvalid >= 8) { load8<T>(p, v); return; }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (i < nvalid) ? p[i] : from_f<T>(0.f);
}
template <typename T> DEVI void zero8(T (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = from_f<T>(0.f);
}

// wave-level reductions (64 lanes)
DEVI float wave_sum(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
DEVI float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

// counter-based RNG (dropout masks, Gumbel noise): splitmix-style 64->32 hash, uniform in (0,1)
DEVI uint32_t hash_u32(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (uint32_t)(z >> 32);
}
DEVI float hash_uniform(uint64_t seed, uint64_t idx) {
    return ((hash_u32(seed, idx) >> 8) + 0.5f) * (1.0f / 16777216.0f);
}

}  // namespace HULC_NS
