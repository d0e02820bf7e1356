// hulc_amd/csrc/kernels.h — the non-GEMM kernels of the HULC step (gfx950): layout packs/transposes, column sums,
// spatial softmax, LayerNorm, tiny-S attention, plan sample/KL, decoder glue, logistic-mixture loss, CLIP loss, Adam.
// All reductions are wave64 shuffles or LDS trees; everything HBM-facing is coalesced along the fastest dim.
#pragma once
#include "common.h"

namespace HULC_NS {

// =========================================================================================================
// casts, transposes, packs
// =========================================================================================================
// dst[r][c] = src[r][c] (cast) and/or dstT[c][r] = src[r][c]   (32x32 tiles through LDS, 256 threads)
template <typename TS, typename TD>
__global__ void __launch_bounds__(256) cast_transpose_kernel(const TS* __restrict__ src, long long lds_, TD* __restrict__ dst,
                                                             long long ldd, TD* __restrict__ dstT, long long ldt, int R, int C) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            v = to_f<TS>(src[(long long)r * lds_ + c]);
            if (dst) dst[(long long)r * ldd + c] = from_f<TD>(v);
        }
        tile[ty + i * 8][tx] = v;
    }
    if (!dstT) return;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, r = r0 + tx;
        if (r < R && c < C) dstT[(long long)c * ldt + r] = from_f<TD>(tile[tx][ty + i * 8]);
    }
}

// many transposes in one launch: block b belongs to descriptor d with blk0[d] <= b < blk0[d+1]; dstT[c][r] = src[r][c]
struct TrDesc { const void* src; void* dst; long long lds, ldt; int R, C, blk0, tiles_x; float* cs = nullptr; float* cs2 = nullptr;
                void* fr = nullptr; void* frT = nullptr; };      // fr / frT (adam_tiled_kernel only; R % 16 == 0, C % 32 == 0 and vice versa): fragment-ordered copies of src / of its transpose (gemm.h frag_pack_kernel's layout)
   // cs: optional column sums of src (bias gradient), bf16 64x64 path only, atomics
template <typename TS, typename TD>
__global__ void __launch_bounds__(256) batched_transpose_kernel(const TrDesc* __restrict__ desc, int ndesc) {
    __shared__ float tile[32][33];
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (desc[mid].blk0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const int d = lo;
    const TrDesc D = desc[d];
    const int b = blockIdx.x - D.blk0;
    const int c0 = (b % D.tiles_x) * 32, r0 = (b / D.tiles_x) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const TS* src = reinterpret_cast<const TS*>(D.src);
    TD* dst = reinterpret_cast<TD*>(D.dst);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, c = c0 + tx;
        tile[ty + i * 8][tx] = (r < D.R && c < D.C) ? to_f<TS>(src[(long long)r * D.lds + c]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, r = r0 + tx;
        if (r < D.R && c < D.C) dst[(long long)c * D.ldt + r] = from_f<TD>(tile[tx][ty + i * 8]);
    }
}

// 64x64-tile bf16 transpose: 16-byte global accesses on both sides (the 32x32 element-wise tile above moves 64-byte row
// segments: 1.6 TB/s on the 94 MB weight set); falls back to guarded element accesses on ragged / unaligned tiles.
DEVI void transpose_tile64_store(const TrDesc& D, int b, unsigned short (*tile)[66]);
DEVI void transpose_tile64_h16(const TrDesc& D, int b, unsigned short (*tile)[66]) {
    const int c0 = (b % D.tiles_x) * 64, r0 = (b / D.tiles_x) * 64;
    const h16_t* src = reinterpret_cast<const h16_t*>(D.src);
    h16_t* dst = reinterpret_cast<h16_t*>(D.dst);
    const bool vin = (D.lds % 8) == 0 && ((uintptr_t)src % 16) == 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int q = threadIdx.x + i * 256, row = q >> 3, cc = (q & 7) * 8;
        const int r = r0 + row, c = c0 + cc;
        unsigned short v[8];
        if (vin && r < D.R && c + 7 < D.C) {
            *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(src + (long long)r * D.lds + c);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (r < D.R && c + e < D.C) ? src[(long long)r * D.lds + c + e] : (unsigned short)0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) *reinterpret_cast<unsigned*>(&tile[row][cc + 2 * e]) = (unsigned)v[2 * e] | ((unsigned)v[2 * e + 1] << 16);
    }
    __syncthreads();
    transpose_tile64_store(D, b, tile);
}
// second half of a 64 x 64 16-bit tile transpose: the tile is in LDS (behind a barrier), pad rows / columns zero
DEVI void transpose_tile64_store(const TrDesc& D, int b, unsigned short (*tile)[66]) {
    const int c0 = (b % D.tiles_x) * 64, r0 = (b / D.tiles_x) * 64;
    h16_t* dst = reinterpret_cast<h16_t*>(D.dst);
    const bool vout = (D.ldt % 8) == 0 && ((uintptr_t)dst % 16) == 0;
    if (D.cs && threadIdx.x < 64 && c0 + (int)threadIdx.x < D.C) {      // fused bias gradient: column sums of this tile (pad rows are zeros)
        float sum = 0.f;
#pragma unroll 16
        for (int rr = 0; rr < 64; ++rr) sum += h2f(tile[rr][threadIdx.x]);
        unsafeAtomicAdd(D.cs + c0 + threadIdx.x, sum);
        if (D.cs2) unsafeAtomicAdd(D.cs2 + c0 + threadIdx.x, sum);
    }
    // read-out: a thread takes TWO adjacent columns of the tile (one 32-bit LDS read per source row instead of two 16-bit ones: the kernel is bound by its
    // LDS instruction count, 24 -> 16 per thread) and writes their 8-row runs as two 16-byte stores; 8 adjacent lanes still cover 128 contiguous bytes
    {
        const int q = threadIdx.x, cp = (q >> 3) * 2, oc = (q & 7) * 8;
        const int r = r0 + oc;
        unsigned w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = *reinterpret_cast<const unsigned*>(&tile[oc + e][cp]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + cp + h;
            if (c >= D.C) continue;
            unsigned short v[8];
            unsigned* vw = reinterpret_cast<unsigned*>(v);
#pragma unroll
            for (int e = 0; e < 4; ++e) vw[e] = __builtin_amdgcn_perm(w[2 * e + 1], w[2 * e], h ? 0x07060302u : 0x05040100u);
            if (D.frT && r + 7 < D.R)      // Wt[c][r .. r+7] as one 16-byte fragment chunk: row tile c / 16, row c % 16, k-step r / 32, lane group (r % 32) / 8
                *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(D.frT) + ((((long long)(c >> 4) * (D.R >> 5) + (r >> 5)) * 64 + ((r & 31) >> 3) * 16 + (c & 15)) << 3)) = *reinterpret_cast<const uint4*>(v);
            if (vout && r + 7 < D.R) {
                *reinterpret_cast<uint4*>(dst + (long long)c * D.ldt + r) = *reinterpret_cast<const uint4*>(v);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (r + e < D.R) dst[(long long)c * D.ldt + r + e] = v[e];
        }
    }
}
}
// descriptor of block b: the last one whose first block is <= b.  Bisection: the linear walk made a block of a late matrix wait for up to ~60
// dependent descriptor loads before it touched its tile
DEVI int trdesc_find(const TrDesc* __restrict__ desc, int ndesc, int b) {
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].blk0 <= b) lo = mid; else hi = mid - 1;
    }
    return lo;
}
__global__ void __launch_bounds__(256) batched_transpose64_kernel(const TrDesc* __restrict__ desc, int ndesc, const unsigned short* __restrict__ blk2desc = nullptr) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][66];
    // blk2desc: the descriptor index of every block, precomputed on the host (one load instead of a chain of ~6 dependent ones per block)
    const int d = blk2desc ? (int)blk2desc[blockIdx.x] : trdesc_find(desc, ndesc, (int)blockIdx.x);
    const TrDesc D = desc[d];
    transpose_tile64_h16(D, blockIdx.x - D.blk0, tile);
}
// two transposes in one launch (the dY / X pair of a large-M Linear weight gradient); descriptors by value
struct TrPair { TrDesc d[2]; };
__global__ void __launch_bounds__(256) pair_transpose64_kernel(TrPair pr) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][66];
    const TrDesc D = ((int)blockIdx.x >= pr.d[1].blk0) ? pr.d[1] : pr.d[0];
    transpose_tile64_h16(D, blockIdx.x - D.blk0, tile);
}
// up to four transposes in one launch (round 5: the decoder's dZ1 / H1 / H0 of the layer-1 weight gradients)
struct TrMulti { TrDesc d[4]; int n; };
__global__ void __launch_bounds__(256) multi_transpose64_kernel(TrMulti m) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][66];
    int k = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i) if (i < m.n && (int)blockIdx.x >= m.d[i].blk0) k = i;
    const TrDesc D = m.d[k];
    transpose_tile64_h16(D, blockIdx.x - D.blk0, tile);
}
template <typename T>
__global__ void __launch_bounds__(256) pair_transpose_kernel(TrPair pr) {
    __shared__ float tile[32][33];
    const TrDesc D = ((int)blockIdx.x >= pr.d[1].blk0) ? pr.d[1] : pr.d[0];
    const int b = blockIdx.x - D.blk0;
    const int c0 = (b % D.tiles_x) * 32, r0 = (b / D.tiles_x) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const T* src = reinterpret_cast<const T*>(D.src);
    T* dst = reinterpret_cast<T*>(D.dst);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + i * 8, c = c0 + tx;
        tile[ty + i * 8][tx] = (r < D.R && c < D.C) ? to_f<T>(src[(long long)r * D.lds + c]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + i * 8, r = r0 + tx;
        if (r < D.R && c < D.C) dst[(long long)c * D.ldt + r] = from_f<T>(tile[tx][ty + i * 8]);
    }
}

// conv weights: torch (O, I, KH, KW) fp32 ->
//   fwd pack  Wf[o][(kh,kw,ci)]                       (conv2/3; conv1 keeps torch's (c,kh,kw) order = plain cast)
//   dgrad pack Wd[zc][ci][(a,b,co)] = W[co][ci][ph+S*a][pw+S*b],  zc = ph*S+pw
// the six conv layers of the two encoders in ONE launch (after every optimizer step): block b belongs to the conv whose block range holds it
struct ConvPackBatch { const float* w[6]; void* wf[6]; void* wd[6]; int O[6], I[6], K[6], S[6], nhwc[6], blk0[7]; };
template <typename T>
__global__ void pack_conv_w_batched_kernel(ConvPackBatch d) {
    int c = 0;
    while (c < 5 && (int)blockIdx.x >= d.blk0[c + 1]) ++c;
    const int idx = (blockIdx.x - d.blk0[c]) * blockDim.x + threadIdx.x;
    const int O = d.O[c], I = d.I[c], KH = d.K[c], KW = d.K[c], S = d.S[c];
    if (idx >= O * I * KH * KW) return;
    int kw = idx % KW, t = idx / KW;
    int kh = t % KH; t /= KH;
    int ci = t % I, o = t / I;
    const float v = d.w[c][idx];
    T* wf = reinterpret_cast<T*>(d.wf[c]);
    T* wd = reinterpret_cast<T*>(d.wd[c]);
    if (wf) {
        if (d.nhwc[c]) wf[(long long)o * (KH * KW * I) + (kh * KW + kw) * I + ci] = from_f<T>(v);
        else wf[idx] = from_f<T>(v);
    }
    if (wd) {
        const int ph = kh % S, a = kh / S, pw = kw % S, b = kw / S;
        const int TA = KH / S, TB = KW / S;
        const int zc = ph * S + pw;
        wd[((long long)zc * I + ci) * (TA * TB * O) + (a * TB + b) * O + o] = from_f<T>(v);
    }
}

// conv wgrad partial slabs [nsplit][O][Kc] (packed K order) -> summed, un-permuted, accumulated into torch-layout grad.
// grid.y splits the slab range; each part lands with one fp32 atomicAdd (<= gridDim.y adds per element).
__global__ void unpack_conv_wgrad_kernel(const float* __restrict__ part, int nsplit, long long slab, float* __restrict__ grad,
                                         int O, int I, int KH, int KW, int nhwc_fwd) {
    const int idx = (blockIdx.x * blockDim.x + threadIdx.x) * 4;   // 4 consecutive elements of the packed layout (16-byte coalesced slab reads)
    const int total = O * I * KH * KW;                              // a multiple of 4 for every conv of the model
    if (idx >= total) return;
    const int per = (nsplit + gridDim.y - 1) / gridDim.y;
    const int z0 = blockIdx.y * per, z1 = min(nsplit, z0 + per);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto add4 = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    int z = z0;
    for (; z + 7 < z1; z += 8) {                    // eight slab quads in flight per thread (four were one L2-miss latency per 4 slabs: 16 slabs = 4 rounds)
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(part + (long long)(z + u) * slab + idx);
        add4(s0, t[0]); add4(s1, t[1]); add4(s2, t[2]); add4(s3, t[3]);
        add4(s0, t[4]); add4(s1, t[5]); add4(s2, t[6]); add4(s3, t[7]);
    }
    for (; z + 3 < z1; z += 4) {
        const float4 a = *reinterpret_cast<const float4*>(part + (long long)z * slab + idx), b = *reinterpret_cast<const float4*>(part + (long long)(z + 1) * slab + idx);
        const float4 c = *reinterpret_cast<const float4*>(part + (long long)(z + 2) * slab + idx), d = *reinterpret_cast<const float4*>(part + (long long)(z + 3) * slab + idx);
        add4(s0, a); add4(s1, b); add4(s2, c); add4(s3, d);
    }
    for (; z < z1; ++z) add4(s0, *reinterpret_cast<const float4*>(part + (long long)z * slab + idx));
    const float v[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int dst = idx + e;
        if (nhwc_fwd) {   // packed idx = o*(KH*KW*I) + (kh*KW+kw)*I + ci  ->  torch ((o*I+ci)*KH+kh)*KW+kw
            int ci = dst % I, t = dst / I;
            int kw = t % KW; t /= KW;
            int kh = t % KH, o = t / KH;
            dst = ((o * I + ci) * KH + kh) * KW + kw;
        }
        if (gridDim.y > 1) unsafeAtomicAdd(grad + dst, v[e]);
        else grad[dst] += v[e];
    }
}

// the same for up to eight convolutions in ONE launch (the encoders' backward deferred its six unpack launches to its end): job = the
// convolution whose block range holds blockIdx.x; every job owns its own slab region
struct UnpackJob { const float* part; float* grad; long long slab; int nsplit, O, I, KH, KW, nhwc, blk0, ysplit; };   // ysplit > 0: this job's slabs over that many grid.y parts only
struct UnpackBatch { UnpackJob j[12]; int n; };
__global__ void unpack_conv_wgrad_batched_kernel(UnpackBatch ub) {
    int k = 0;
    while (k + 1 < ub.n && (int)blockIdx.x >= ub.j[k + 1].blk0) ++k;
    const UnpackJob J = ub.j[k];
    const int idx = ((blockIdx.x - J.blk0) * blockDim.x + threadIdx.x) * 4;
    const int total = J.O * J.I * J.KH * J.KW;
    if (idx >= total) return;
    const int gy = J.ysplit > 0 ? J.ysplit : (int)gridDim.y;
    if ((int)blockIdx.y >= gy) return;
    const int per = (J.nsplit + gy - 1) / gy;
    const int z0 = blockIdx.y * per, z1 = min(J.nsplit, z0 + per);
    if (z0 >= z1) return;
    const float* part = J.part;
    const long long slab = J.slab;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto add4 = [](float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; };
    int z = z0;
    for (; z + 7 < z1; z += 8) {                    // eight slab quads in flight per thread (four were one L2-miss latency per 4 slabs: 16 slabs = 4 rounds)
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(part + (long long)(z + u) * slab + idx);
        add4(s0, t[0]); add4(s1, t[1]); add4(s2, t[2]); add4(s3, t[3]);
        add4(s0, t[4]); add4(s1, t[5]); add4(s2, t[6]); add4(s3, t[7]);
    }
    for (; z + 3 < z1; z += 4) {
        const float4 a = *reinterpret_cast<const float4*>(part + (long long)z * slab + idx), b = *reinterpret_cast<const float4*>(part + (long long)(z + 1) * slab + idx);
        const float4 c = *reinterpret_cast<const float4*>(part + (long long)(z + 2) * slab + idx), d = *reinterpret_cast<const float4*>(part + (long long)(z + 3) * slab + idx);
        add4(s0, a); add4(s1, b); add4(s2, c); add4(s3, d);
    }
    for (; z < z1; ++z) add4(s0, *reinterpret_cast<const float4*>(part + (long long)z * slab + idx));
    const float v[4] = {(s0.x + s1.x) + (s2.x + s3.x), (s0.y + s1.y) + (s2.y + s3.y), (s0.z + s1.z) + (s2.z + s3.z), (s0.w + s1.w) + (s2.w + s3.w)};
    if (gy == 1 && !J.nhwc) {            // this thread is the only writer of its four elements: plain read-add-write
        float4* g4 = reinterpret_cast<float4*>(J.grad + idx);
        float4 o = *g4;
        o.x += v[0]; o.y += v[1]; o.z += v[2]; o.w += v[3];
        *g4 = o;
        return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int dst = idx + e;
        if (J.nhwc) {
            int ci = dst % J.I, t = dst / J.I;
            int kw = t % J.KW; t /= J.KW;
            int kh = t % J.KH, o = t / J.KH;
            dst = ((o * J.I + ci) * J.KH + kh) * J.KW + kw;
        }
        if (gy == 1) J.grad[dst] += v[e];          // only writer of the element
        else unsafeAtomicAdd(J.grad + dst, v[e]);
    }
}

// dst[r][perm(c)] = src[r][c] with c = ch*P + p  ->  perm(c) = p*CH + ch   (torch Flatten(C,H,W) <-> NHWC flatten)
template <typename TS, typename TD>
__global__ void permute_cols_kernel(const TS* __restrict__ src, TD* __restrict__ dst, int R, int CH, int P, int inverse,
                                    int accumulate) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)R * CH * P;
    if (idx >= total) return;
    const int c = idx % (CH * P);
    const long long r = idx / (CH * P);
    // forward: src col (ch*P+p) -> dst col (p*CH+ch); inverse: src col (p*CH+ch) -> dst col (ch*P+p)
    int d;
    if (!inverse) { int ch = c / P, p = c % P; d = p * CH + ch; }
    else { int p = c / CH, ch = c % CH; d = ch * P + p; }
    const float v = to_f<TS>(src[idx]);
    TD* o = dst + r * (CH * P) + d;
    *o = from_f<TD>(accumulate ? to_f<TD>(*o) + v : v);
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long n) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = from_f<TD>(to_f<TS>(src[i]));
}

// generic strided 2D copy/cast: dst[r*ldd + c] (+)= src[r*lds + c]
template <typename TS, typename TD>
__global__ void copy2d_kernel(const TS* __restrict__ src, long long lds_, TD* __restrict__ dst, long long ldd, int R, int C,
                              int accumulate, float scale) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)R * C) return;
    const int c = idx % C;
    const long long r = idx / C;
    const float v = to_f<TS>(src[r * lds_ + c]) * scale;
    TD* o = dst + r * ldd + c;
    *o = from_f<TD>(accumulate ? to_f<TD>(*o) + v : v);
}

// the plan proposal's input gradient [B][E + G] added onto its two sources in one launch: columns [0, E) -> d emb[:, 0, :] (row stride ld_e), [E, E + G) -> d goal
__global__ void pp_input_bwd_kernel(const float* __restrict__ dppx, int B, int E, int G, float* __restrict__ demb, long long ld_e, float* __restrict__ dgoal) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * (E + G)) return;
    const int r = idx / (E + G), c = idx - r * (E + G);
    float* o = c < E ? demb + (long long)r * ld_e + c : dgoal + (long long)r * G + (c - E);
    *o += dppx[idx];
}

// =========================================================================================================
// column sums (bias grads): out[n] (+)= scale * sum_m X[m][n]; two-stage when rows are split
// =========================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) colsum_kernel(const T* __restrict__ x, long long ld, int M, int N, float* __restrict__ out,
                                                     int rows_per_split, int direct_accumulate, float scale, float* __restrict__ out2 = nullptr) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_split, rend = min(M, rbeg + rows_per_split);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (c < N) {
        const T* p = x + c;
        int r = rbeg + rl;
        for (; r + 12 < rend; r += 16) {
            s0 += to_f<T>(p[(long long)r * ld]); s1 += to_f<T>(p[(long long)(r + 4) * ld]);
            s2 += to_f<T>(p[(long long)(r + 8) * ld]); s3 += to_f<T>(p[(long long)(r + 12) * ld]);
        }
        for (; r < rend; r += 4) s0 += to_f<T>(p[(long long)r * ld]);
    }
    red[rl][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rl == 0 && c < N) {
        const float s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        if (direct_accumulate == 2) { unsafeAtomicAdd(out + c, s * scale); if (out2) unsafeAtomicAdd(out2 + c, s * scale); }
        else if (direct_accumulate) out[c] += s * scale;
        else out[(long long)blockIdx.y * N + c] = s;
    }
}
// column sums of a contiguous [M][N] matrix with N in {32, 64}: every thread streams 8-element (16 B for bf16) chunks whose
// column group is fixed (gridDim.x*256*8 % N == 0), partial[blockIdx.x][N] written deterministically
template <typename T>
__global__ void __launch_bounds__(256) colsum_flat_kernel(const T* __restrict__ x, long long total, int N, float* __restrict__ part) {
    __shared__ float red[256][9];
    const long long nchunk = total >> 3;
    const long long stride = (long long)gridDim.x * 256;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < nchunk; c += stride) {
        T v[8];
        load8<T>(x + c * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += to_f<T>(v[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    // thread t's column group = (t*8) % N ; groups repeat every N/8 threads
    const int G = N >> 3;
    if (threadIdx.x < N) {
        const int grp = threadIdx.x >> 3, e = threadIdx.x & 7;
        float a = 0.f;
        for (int t = grp; t < 256; t += G) a += red[t][e];
        part[(long long)blockIdx.x * N + threadIdx.x] = a;
    }
}
// out[c] (+out2[c]) += scale * sum_z part[z][c] : 64 columns x 4 partial-lanes per block
__global__ void __launch_bounds__(256) colsum_final_kernel(const float* __restrict__ part, int nsplit, int N, float* __restrict__ out,
                                                           float* __restrict__ out2, float scale) {
    __shared__ float red[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), zl = threadIdx.x >> 6;
    float s = 0.f;
    if (c < N)
        for (int z = zl; z < nsplit; z += 4) s += part[(long long)z * N + c];
    red[zl][threadIdx.x & 63] = s;
    __syncthreads();
    if (zl == 0 && c < N) {
        s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
        out[c] += s * scale;
        if (out2) out2[c] += s * scale;
    }
}

// =========================================================================================================
// spatial softmax (vision_network.py:100-108) on NHWC features [N][HW][C]; one block per frame, 4 x C threads
// out[n][2c] = sum p * lin[h], out[n][2c+1] = sum p * lin[w]; also saves (max, 1/sum) for backward
// =========================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) spatial_softmax_fwd_kernel(const T* __restrict__ f, int H, int W, int C, T* __restrict__ out,
                                                                  float* __restrict__ out_f32, float* __restrict__ stats /*[N][C][4]*/) {
    __shared__ float sm[4][64], ss[4][64], sx[4][64], sy[4][64];
    const int n = blockIdx.x, c = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int HW = H * W;
    const T* p = f + (long long)n * HW * C + c;
    float m = -INFINITY, s = 0.f, ax = 0.f, ay = 0.f;
    const float sh = 2.f / (H - 1), sw = 2.f / (W - 1);
    if (c < C) {
        for (int q = part; q < HW; q += 4) {
            const float v = to_f<T>(p[(long long)q * C]);
            const int h = q / W, w = q % W;
            const float lx = -1.f + sh * h, ly = -1.f + sw * w;
            if (v > m) {
                const float sc = __expf(m - v);
                s *= sc; ax *= sc; ay *= sc;
                m = v;
            }
            const float e = __expf(v - m);
            s += e; ax += e * lx; ay += e * ly;
        }
    }
    sm[part][c] = m; ss[part][c] = s; sx[part][c] = ax; sy[part][c] = ay;
    __syncthreads();
    if (part == 0 && c < C) {
        float M = fmaxf(fmaxf(sm[0][c], sm[1][c]), fmaxf(sm[2][c], sm[3][c]));
        float S = 0.f, X = 0.f, Y = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sc = (sm[k][c] == -INFINITY) ? 0.f : __expf(sm[k][c] - M);
            S += ss[k][c] * sc; X += sx[k][c] * sc; Y += sy[k][c] * sc;
        }
        const float inv = 1.f / S;
        const float ex = X * inv, ey = Y * inv;
        const long long o = (long long)n * 2 * C + 2 * c;
        if (out) { out[o] = from_f<T>(ex); out[o + 1] = from_f<T>(ey); }
        if (out_f32) { out_f32[o] = ex; out_f32[o + 1] = ey; }
        float* st = stats + ((long long)n * C + c) * 4;
        st[0] = M; st[1] = inv; st[2] = ex; st[3] = ey;
    }
}
// dF[n][q][c] = p * (dex*(lin_h - ex) + dey*(lin_w - ey)) * (F > 0)      (ReLU of conv3 fused)
template <typename T>
__global__ void __launch_bounds__(256) spatial_softmax_bwd_kernel(const T* __restrict__ f, const float* __restrict__ stats,
                                                                  const float* __restrict__ dout /*[N][2C] fp32*/, int H, int W, int C,
                                                                  T* __restrict__ df) {
    const int n = blockIdx.x, c = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int HW = H * W;
    if (c >= C) return;
    const float* st = stats + ((long long)n * C + c) * 4;
    const float M = st[0], inv = st[1], ex = st[2], ey = st[3];
    const float dex = dout[(long long)n * 2 * C + 2 * c], dey = dout[(long long)n * 2 * C + 2 * c + 1];
    const float sh = 2.f / (H - 1), sw = 2.f / (W - 1);
    const long long base = (long long)n * HW * C + c;
    for (int q = part; q < HW; q += 4) {
        const float v = to_f<T>(f[base + (long long)q * C]);
        const int h = q / W, w = q % W;
        const float lx = -1.f + sh * h, ly = -1.f + sw * w;
        const float p = __expf(v - M) * inv;
        const float g = (v > 0.f) ? p * (dex * (lx - ex) + dey * (ly - ey)) : 0.f;
        df[base + (long long)q * C] = from_f<T>(g);
    }
}

// bf16, C = 64 versions: a thread owns 8 channels (one 16-byte load per pixel) of every 32nd pixel, so a frame is 14 wide
// iterations instead of 110 two-byte ones; the forward keeps 8 online-softmax states per thread and merges the 32 pixel groups in LDS.
__global__ void __launch_bounds__(256) spatial_softmax_fwd64_kernel(const h16_t* __restrict__ f, int H, int W, h16_t* __restrict__ out,
                                                                    float* __restrict__ stats /*[N][64][4]*/) {
    __shared__ float sm[32][65], ss[32][65], sx[32][65], sy[32][65];
    const int n = blockIdx.x, cg = threadIdx.x & 7, pg = threadIdx.x >> 3;
    const int HW = H * W;
    const h16_t* p = f + (long long)n * HW * 64 + cg * 8;
    float m[8], s[8], ax[8], ay[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { m[e] = -INFINITY; s[e] = 0.f; ax[e] = 0.f; ay[e] = 0.f; }
    const float sh = 2.f / (H - 1), sw = 2.f / (W - 1);
    constexpr int SSM_NIT = 7;                    // 64-position rounds held in registers: 448 >= the 21 x 21 map of the static camera
    if (HW <= SSM_NIT * 64) {
        // Two passes over registers instead of the online form below: the whole map of this thread (<= 14 positions x 8 channels, 56 VGPRs) is requested
        // up front, pass 1 takes the thread's maximum, pass 2 one exponential per value — the online form pays a compare, a conditional rescale and
        // two exponentials per value and was VALU-bound (36 us for the 115 MB of the static camera's maps, 21 us of HBM time)
#ifdef HULC_HALF_F16
        constexpr unsigned NEG_INF2 = 0xFC00FC00u;
#else
        constexpr unsigned NEG_INF2 = 0xFF80FF80u;
#endif
        uint4 raw[SSM_NIT][2];
#pragma unroll
        for (int it = 0; it < SSM_NIT; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = pg + it * 64 + u * 32;
                raw[it][u] = uint4{NEG_INF2, NEG_INF2, NEG_INF2, NEG_INF2};
                if (q < HW) raw[it][u] = *reinterpret_cast<const uint4*>(p + (long long)q * 64);
            }
#pragma unroll
        for (int it = 0; it < SSM_NIT; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned wd[4] = {raw[it][u].x, raw[it][u].y, raw[it][u].z, raw[it][u].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], (e & 1) ? h2f_hi(wd[e >> 1]) : h2f_lo(wd[e >> 1]));
            }
        float mu[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) mu[e] = m[e] == -INFINITY ? 0.f : m[e];      // a thread without a position: exp(-inf - 0) = 0, its maximum stays -inf for the merge
        int h = pg / W, w = pg - h * W;
#pragma unroll
        for (int it = 0; it < SSM_NIT; ++it)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float lx = -1.f + sh * h, ly = -1.f + sw * w;
                const unsigned wd[4] = {raw[it][u].x, raw[it][u].y, raw[it][u].z, raw[it][u].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float v = (e & 1) ? h2f_hi(wd[e >> 1]) : h2f_lo(wd[e >> 1]);
                    const float ex = __expf(v - mu[e]);
                    s[e] += ex; ax[e] += ex * lx; ay[e] += ex * ly;
                }
                w += 32;                                                           // next position of this thread: q + 32
                while (w >= W) { w -= W; ++h; }
            }
    } else
    for (int q0 = pg; q0 < HW; q0 += 64) {
        uint4 raw[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) raw[u] = *reinterpret_cast<const uint4*>(p + (long long)min(q0 + u * 32, HW - 1) * 64);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = q0 + u * 32;
            if (q >= HW) break;
            const int h = q / W, w = q - h * W;
            const float lx = -1.f + sh * h, ly = -1.f + sw * w;
            const unsigned wd[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (e & 1) ? h2f_hi(wd[e >> 1]) : h2f_lo(wd[e >> 1]);
                if (v > m[e]) {
                    const float sc = __expf(m[e] - v);
                    s[e] *= sc; ax[e] *= sc; ay[e] *= sc;
                    m[e] = v;
                }
                const float ex = __expf(v - m[e]);
                s[e] += ex; ax[e] += ex * lx; ay[e] += ex * ly;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sm[pg][cg * 8 + e] = m[e]; ss[pg][cg * 8 + e] = s[e]; sx[pg][cg * 8 + e] = ax[e]; sy[pg][cg * 8 + e] = ay[e]; }
    __syncthreads();
    // merge of the 32 position groups in two levels: all four waves merge 8 groups each, one wave the four results (a single wave walking all
    // 32 was a ~1.5 us serial tail per frame with the other three waves idle)
    __shared__ float t2[4][4][64];
    {
        const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
        float M = -INFINITY;
#pragma unroll
        for (int k = 0; k < 8; ++k) M = fmaxf(M, sm[part * 8 + k][c]);
        float S = 0.f, X = 0.f, Y = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float mk = sm[part * 8 + k][c];
            const float sc = (mk == -INFINITY) ? 0.f : __expf(mk - M);
            S += ss[part * 8 + k][c] * sc; X += sx[part * 8 + k][c] * sc; Y += sy[part * 8 + k][c] * sc;
        }
        t2[0][part][c] = M; t2[1][part][c] = S; t2[2][part][c] = X; t2[3][part][c] = Y;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x;
        float M = -INFINITY;
#pragma unroll
        for (int k = 0; k < 4; ++k) M = fmaxf(M, t2[0][k][c]);
        float S = 0.f, X = 0.f, Y = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sc = (t2[0][k][c] == -INFINITY) ? 0.f : __expf(t2[0][k][c] - M);
            S += t2[1][k][c] * sc; X += t2[2][k][c] * sc; Y += t2[3][k][c] * sc;
        }
        const float inv = 1.f / S;
        const float ex = X * inv, ey = Y * inv;
        const long long o = (long long)n * 128 + 2 * c;
        out[o] = f2h(ex); out[o + 1] = f2h(ey);
        float* st = stats + ((long long)n * 64 + c) * 4;
        st[0] = M; st[1] = inv; st[2] = ex; st[3] = ey;
    }
}
__global__ void __launch_bounds__(256) spatial_softmax_bwd64_kernel(const h16_t* __restrict__ f, const float* __restrict__ stats,
                                                                    const float* __restrict__ dout /*[N][128] fp32*/, int H, int W, h16_t* __restrict__ df) {
    const int n = blockIdx.x, cg = threadIdx.x & 7, pg = threadIdx.x >> 3;
    const int HW = H * W;
    float M[8], inv[8], ex[8], ey[8], dex[8], dey[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float4 st = *reinterpret_cast<const float4*>(stats + ((long long)n * 64 + cg * 8 + e) * 4);
        M[e] = st.x; inv[e] = st.y; ex[e] = st.z; ey[e] = st.w;
        const float2 d = *reinterpret_cast<const float2*>(dout + (long long)n * 128 + 2 * (cg * 8 + e));
        dex[e] = d.x; dey[e] = d.y;
    }
    const float sh = 2.f / (H - 1), sw = 2.f / (W - 1);
    const long long base = (long long)n * HW * 64 + cg * 8;
    for (int q0 = pg; q0 < HW; q0 += 64) {
        uint4 raw[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) raw[u] = *reinterpret_cast<const uint4*>(f + base + (long long)min(q0 + u * 32, HW - 1) * 64);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = q0 + u * 32;
            if (q >= HW) break;
            const int h = q / W, w = q - h * W;
            const float lx = -1.f + sh * h, ly = -1.f + sw * w;
            const unsigned wd[4] = {raw[u].x, raw[u].y, raw[u].z, raw[u].w};
            float g[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = (e & 1) ? h2f_hi(wd[e >> 1]) : h2f_lo(wd[e >> 1]);
                const float pr = __expf(v - M[e]) * inv[e];
                g[e] = (v > 0.f) ? pr * (dex[e] * (lx - ex[e]) + dey[e] * (ly - ey[e])) : 0.f;
            }
            uint4 o;
            o.x = pack2h(g[0], g[1]); o.y = pack2h(g[2], g[3]); o.z = pack2h(g[4], g[5]); o.w = pack2h(g[6], g[7]);
            *reinterpret_cast<uint4*>(df + base + (long long)q * 64) = o;
        }
    }
}

// the four decoder heads (prob | mean | log_scale | gripper) as one packed [NHEAD][HID] GEMM operand: ONE launch packs weights
// + biases, ONE launch adds the packed gradient back into the four parameter gradients (was 8 + 8 copy2d launches per step)
struct HeadPack { const float* w[4]; const float* b[4]; float* dw[4]; float* db[4]; int rows[4]; };
template <typename T>
__global__ void pack_heads_kernel(HeadPack hp, T* __restrict__ wheads, float* __restrict__ bheads, int HID) {
    const int total_rows = hp.rows[0] + hp.rows[1] + hp.rows[2] + hp.rows[3];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)total_rows * HID) return;
    int r = (int)(idx / HID);
    const int k = (int)(idx % HID);
    int i = 0;
    while (i < 3 && r >= hp.rows[i]) { r -= hp.rows[i]; ++i; }
    wheads[idx] = from_f<T>(hp.w[i][(long long)r * HID + k]);
    if (k == 0) bheads[idx / HID] = hp.b[i][r];
}
// the three independent repacks that follow an optimizer step in ONE launch (they were three): blocks [0, blkA) the six conv weight packs
// (pack_conv_w_batched_kernel's body), [blkA, blkB) the gripper fc7's NHWC column permutation (permute_cols_kernel, R x (CH * P) fp32 -> T),
// [blkB, ..) the packed decoder heads (pack_heads_kernel).  Their transposed copies ride on the batched transpose launch that follows.
template <typename T>
__global__ void __launch_bounds__(256) weight_pack_kernel(ConvPackBatch d, int blkA, const float* __restrict__ fc7_src, T* __restrict__ fc7_dst, int R7, int CH7,
                                                          int P7, int blkB, HeadPack hp, T* __restrict__ wheads, float* __restrict__ bheads, int HID,
                                                          T* __restrict__ fc7_dstT = nullptr, T* __restrict__ wheadsT = nullptr, int ldth = 0) {
    // fc7_dstT / wheadsT (round 5): the TRANSPOSED copies of the two packed matrices ([CH7*P7][R7] and [HID][ldth]) written here as well — 0.8 M scattered 2-byte
    // stores instead of a transpose launch behind this one
    const int bid = blockIdx.x;
    if (bid < blkA) {
        int c = 0;
        while (c < 5 && bid >= d.blk0[c + 1]) ++c;
        const int idx = (bid - d.blk0[c]) * blockDim.x + threadIdx.x;
        const int O = d.O[c], I = d.I[c], KH = d.K[c], KW = d.K[c], S = d.S[c];
        if (idx >= O * I * KH * KW) return;
        int kw = idx % KW, t = idx / KW;
        int kh = t % KH; t /= KH;
        int ci = t % I, o = t / I;
        const float v = d.w[c][idx];
        T* wf = reinterpret_cast<T*>(d.wf[c]);
        T* wd = reinterpret_cast<T*>(d.wd[c]);
        if (wf) {
            if (d.nhwc[c]) wf[(long long)o * (KH * KW * I) + (kh * KW + kw) * I + ci] = from_f<T>(v);
            else wf[idx] = from_f<T>(v);
        }
        if (wd) {
            const int ph = kh % S, a = kh / S, pw = kw % S, b = kw / S;
            const int TA = KH / S, TB = KW / S;
            const int zc = ph * S + pw;
            wd[((long long)zc * I + ci) * (TA * TB * O) + (a * TB + b) * O + o] = from_f<T>(v);
        }
        return;
    }
    if (bid < blkB) {
        const long long idx = (long long)(bid - blkA) * blockDim.x + threadIdx.x;
        if (idx >= (long long)R7 * CH7 * P7) return;
        const int c = idx % (CH7 * P7);
        const long long r = idx / (CH7 * P7);
        const int ch = c / P7, pp = c % P7;
        const T v7 = from_f<T>(fc7_src[idx]);
        fc7_dst[r * (CH7 * P7) + pp * CH7 + ch] = v7;
        if (fc7_dstT) fc7_dstT[(long long)(pp * CH7 + ch) * R7 + r] = v7;
        return;
    }
    const int total_rows = hp.rows[0] + hp.rows[1] + hp.rows[2] + hp.rows[3];
    const long long idx = (long long)(bid - blkB) * blockDim.x + threadIdx.x;
    if (idx >= (long long)total_rows * HID) return;
    int r = (int)(idx / HID);
    const int k = (int)(idx % HID);
    int i = 0;
    while (i < 3 && r >= hp.rows[i]) { r -= hp.rows[i]; ++i; }
    const T vh = from_f<T>(hp.w[i][(long long)r * HID + k]);
    wheads[idx] = vh;
    if (wheadsT) wheadsT[(long long)k * ldth + idx / HID] = vh;
    if (k == 0) bheads[idx / HID] = hp.b[i][r];
}
__global__ void unpack_heads_grad_kernel(HeadPack hp, const float* __restrict__ dw, const float* __restrict__ db, int HID, int nslab = 1, long long slab = 0) {
    const int total_rows = hp.rows[0] + hp.rows[1] + hp.rows[2] + hp.rows[3];
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)total_rows * HID) return;
    int r = (int)(idx / HID);
    const int k = (int)(idx % HID);
    int i = 0;
    while (i < 3 && r >= hp.rows[i]) { r -= hp.rows[i]; ++i; }
    float a = dw[idx];
    for (int z = 1; z < nslab; ++z) a += dw[(long long)z * slab + idx];      // row-split weight gradient: one slab per row chunk
    hp.dw[i][(long long)r * HID + k] += a;
    if (k == 0) hp.db[i][r] += db[idx / HID];
}

// =========================================================================================================
// LayerNorm over the last dim n (32/64/128), one wave per row; biased variance, eps 1e-5
// =========================================================================================================
template <typename T>
__global__ void __launch_bounds__(256) layernorm_fwd_kernel(const float* __restrict__ x, long long ldx, int rows, int n,
                                                            const float* __restrict__ g, const float* __restrict__ b, T* __restrict__ out,
                                                            long long ldo, float* __restrict__ out_f32, long long ldf,
                                                            float* __restrict__ stats /*[rows][2]*/) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + (long long)row * ldx;
    float v0 = lane < n ? xr[lane] : 0.f, v1 = lane + 64 < n ? xr[lane + 64] : 0.f;
    const float mean = wave_sum(v0 + v1) / n;
    const float d0 = lane < n ? v0 - mean : 0.f, d1 = lane + 64 < n ? v1 - mean : 0.f;
    const float var = wave_sum(d0 * d0 + d1 * d1) / n;
    const float rstd = rsqrtf(var + 1e-5f);
    if (lane < n) {
        const float y = d0 * rstd * g[lane] + b[lane];
        if (out) out[(long long)row * ldo + lane] = from_f<T>(y);
        if (out_f32) out_f32[(long long)row * ldf + lane] = y;
    }
    if (lane + 64 < n) {
        const float y = d1 * rstd * g[lane + 64] + b[lane + 64];
        if (out) out[(long long)row * ldo + lane + 64] = from_f<T>(y);
        if (out_f32) out_f32[(long long)row * ldf + lane + 64] = y;
    }
    if (lane == 0 && stats) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
// dx = rstd * (dxh - mean(dxh) - xh * mean(dxh*xh)),  dxh = dy*g ; writes fp32 dx (optionally accumulating) and/or T dx
template <typename T>
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                            long long ldx, const float* __restrict__ stats, const float* __restrict__ g,
                                                            int rows, int n, float* __restrict__ dx_f32, long long ldd, int accumulate,
                                                            T* __restrict__ dx_t, long long ldt) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float* xr = x + (long long)row * ldx;
    const float* dr = dy + (long long)row * lddy;
    float xh0 = 0.f, xh1 = 0.f, q0 = 0.f, q1 = 0.f;
    if (lane < n) { xh0 = (xr[lane] - mean) * rstd; q0 = dr[lane] * g[lane]; }
    if (lane + 64 < n) { xh1 = (xr[lane + 64] - mean) * rstd; q1 = dr[lane + 64] * g[lane + 64]; }
    const float m1 = wave_sum(q0 + q1) / n;
    const float m2 = wave_sum(q0 * xh0 + q1 * xh1) / n;
    if (lane < n) {
        const float d = rstd * (q0 - m1 - xh0 * m2);
        if (dx_f32) { float* o = dx_f32 + (long long)row * ldd + lane; *o = accumulate ? *o + d : d; }
        if (dx_t) dx_t[(long long)row * ldt + lane] = from_f<T>(d);
    }
    if (lane + 64 < n) {
        const float d = rstd * (q1 - m1 - xh1 * m2);
        if (dx_f32) { float* o = dx_f32 + (long long)row * ldd + lane + 64; *o = accumulate ? *o + d : d; }
        if (dx_t) dx_t[(long long)row * ldt + lane + 64] = from_f<T>(d);
    }
}
// 16-bit engines: the whole LayerNorm backward in ONE launch.  A workgroup walks `rows_per_block` rows (one wave per row at a time): dx as above — the
// 16-bit copy optionally through the dropout mask of the residual branch the gradient flows into next (the forward applied that mask in the
// producing GEMM's epilogue; index = row * n + col as there) — and the parameter gradients from per-lane partial sums over the block's rows,
// reduced over the 4 waves in LDS and added with one fp32 atomic per column per block.  Replaces layernorm_bwd_kernel +
// layernorm_param_grad_kernel (+ dropout_apply_kernel in the transformer): 7 + 7 + 4 launches of ~5 us per step become 7.
template <typename T>
__global__ void __launch_bounds__(1024) layernorm_bwd_fused_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                                  long long ldx, const float* __restrict__ stats, const float* __restrict__ g,
                                                                  int rows, int n, float* __restrict__ dx_f32, long long ldd, int accumulate,
                                                                  T* __restrict__ dx_t, long long ldt, float drop_p, unsigned long long seed,
                                                                  int rows_per_block, float* __restrict__ dg, float* __restrict__ db,
                                                                  int dy_rowdiv = 0, float dy_div = 1.f, int dy_parts = 1, long long dy_part_stride = 0) {
    // dy_parts > 1: the incoming gradient is the sum of dy_parts arrays dy + i * dy_part_stride (the fused FFN backward's four partials)
    // dy_rowdiv > 0: the incoming gradient is a per-WINDOW vector broadcast over the window's dy_rowdiv rows and divided by dy_div
    // (the mean over S in front of plan_recognition.fc: bcast_over_s_kernel's job, read here instead of materialised)
    __shared__ float red[16][256];          // blockDim.x = 256 (4 waves) or 1024 (16 waves: a row per wave of a 16-row block — the rows' load -> reduce -> store chains run side by side)
    const int nwv = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    const float g0 = lane < n ? g[lane] : 0.f, g1 = lane + 64 < n ? g[lane + 64] : 0.f;
    float sg0 = 0.f, sg1 = 0.f, sb0 = 0.f, sb1 = 0.f;
    for (int row = r0 + wave; row < r1; row += nwv) {
        const float mean = stats[2 * row], rstd = stats[2 * row + 1];
        const float* xr = x + (long long)row * ldx;
        const float* dr = dy + (long long)(dy_rowdiv > 0 ? row / dy_rowdiv : row) * lddy;
        float xh0 = 0.f, xh1 = 0.f, d0 = 0.f, d1 = 0.f;
        if (lane < n) { xh0 = (xr[lane] - mean) * rstd; d0 = dr[lane]; for (int pp = 1; pp < dy_parts; ++pp) d0 += dr[pp * dy_part_stride + lane]; }
        if (lane + 64 < n) { xh1 = (xr[lane + 64] - mean) * rstd; d1 = dr[lane + 64]; for (int pp = 1; pp < dy_parts; ++pp) d1 += dr[pp * dy_part_stride + lane + 64]; }
        if (dy_rowdiv > 0) { d0 /= dy_div; d1 /= dy_div; }
        sg0 += d0 * xh0; sb0 += d0; sg1 += d1 * xh1; sb1 += d1;
        const float q0 = d0 * g0, q1 = d1 * g1;
        const float m1 = wave_sum(q0 + q1) / n;
        const float m2 = wave_sum(q0 * xh0 + q1 * xh1) / n;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int col = lane + h * 64;
            if (col < n) {
                const float d = rstd * ((h ? q1 : q0) - m1 - (h ? xh1 : xh0) * m2);
                if (dx_f32) { float* o = dx_f32 + (long long)row * ldd + col; *o = accumulate ? *o + d : d; }
                if (dx_t) {
                    float v = d;
                    if (drop_p > 0.f) v = hash_uniform(seed, (long long)row * n + col) < drop_p ? 0.f : v / (1.f - drop_p);
                    dx_t[(long long)row * ldt + col] = from_f<T>(v);
                }
            }
        }
    }
    if (!dg) return;
    red[wave][lane] = sg0; red[wave][64 + lane] = sg1; red[wave][128 + lane] = sb0; red[wave][192 + lane] = sb1;
    __syncthreads();
    const int t = threadIdx.x, col = t & 127;
    if (t < 256 && col < n) {
        float a = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        for (int w = 4; w < nwv; w += 4) a += (red[w][t] + red[w + 1][t]) + (red[w + 2][t] + red[w + 3][t]);
        unsafeAtomicAdd((t < 128 ? dg : db) + col, a);
    }
}
// partial sums for dgamma[c] = sum_r dy*xhat and dbeta[c] = sum_r dy over a row chunk (grid.y); part = [2][nsplit][n]
__global__ void __launch_bounds__(256) layernorm_param_grad_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ x,
                                                                   long long ldx, const float* __restrict__ stats, int rows, int n,
                                                                   int rows_per_split, float* __restrict__ part,
                                                                   float* __restrict__ dg_atomic = nullptr, float* __restrict__ db_atomic = nullptr) {
    __shared__ float r1[4][64], r2[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int rbeg = blockIdx.y * rows_per_split, rend = min(rows, rbeg + rows_per_split);
    float a = 0.f, b = 0.f;
    if (c < n)
        for (int r = rbeg + rl; r < rend; r += 4) {
            const float d = dy[(long long)r * lddy + c];
            a += d * (x[(long long)r * ldx + c] - stats[2 * r]) * stats[2 * r + 1];
            b += d;
        }
    r1[rl][threadIdx.x & 63] = a; r2[rl][threadIdx.x & 63] = b;
    __syncthreads();
    if (rl == 0 && c < n) {
        const int t = threadIdx.x;
        const float sa = (r1[0][t] + r1[1][t]) + (r1[2][t] + r1[3][t]), sb = (r2[0][t] + r2[1][t]) + (r2[2][t] + r2[3][t]);
        if (dg_atomic) {                                            // bf16 (bench) mode: no second stage
            unsafeAtomicAdd(dg_atomic + c, sa); unsafeAtomicAdd(db_atomic + c, sb);
        } else {
            part[(long long)blockIdx.y * n + c] = sa;
            part[((long long)gridDim.y + blockIdx.y) * n + c] = sb;
        }
    }
}

// =========================================================================================================
// plan-recognition transformer glue (plan_recognition_net.py:94-117)
// =========================================================================================================
// x0[b,t,:] = emb[b,t,:] + pos[t,:] (+ dropout)  -> fp32 residual stream and T GEMM operand
template <typename T>
__global__ void posadd_kernel(const T* __restrict__ emb, const float* __restrict__ pos, int B, int S, int D, float* __restrict__ xf,
                              T* __restrict__ xt, float drop_p, unsigned long long seed, float* __restrict__ z0 = nullptr, float* __restrict__ z1 = nullptr) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * S * D) return;
    if (z0) { z0[idx] = 0.f; z1[idx] = 0.f; }      // the fused transformer layers (tr_fused.h) accumulate their FFN output quarters into these
    const int d = idx % D;
    const int t = (idx / D) % S;
    float v = to_f<T>(emb[idx]) + pos[t * D + d];
    if (drop_p > 0.f) v = hash_uniform(seed, idx) < drop_p ? 0.f : v / (1.f - drop_p);
    xf[idx] = v;
    xt[idx] = from_f<T>(v);
}
// elementwise dropout mask application (backward of a dropout whose forward was fused elsewhere)
template <typename T>
__global__ void dropout_apply_kernel(const float* __restrict__ src, float* __restrict__ dstf, T* __restrict__ dstt, long long n, float drop_p,
                                     unsigned long long seed) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    float v = src[idx];
    if (drop_p > 0.f) v = hash_uniform(seed, idx) < drop_p ? 0.f : v / (1.f - drop_p);
    if (dstf) dstf[idx] = v;
    if (dstt) dstt[idx] = from_f<T>(v);
}

// attention for tiny S (<= SMAX = 32 or 64), head_dim 16: one block (64 threads) per (b, head); the key loops are fully unrolled over SMAX
// (predicated on j < S) so the per-row score arrays stay in registers; qkv [B*S][3D] T, row = b*S+t
// P saved as fp32 [B][H][S][S] (post-softmax, pre-dropout); out ao [B*S][D] T
template <typename T, int SMAX>
__global__ void __launch_bounds__(64) attention_fwd_kernel(const T* __restrict__ qkv, int B, int S, int D, int NH, float* __restrict__ P,
                                                           T* __restrict__ ao, float drop_p, unsigned long long seed) {
    constexpr int HD = 16;
    __shared__ float q[SMAX][HD + 1], k[SMAX][HD + 1], v[SMAX][HD + 1];
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, i = threadIdx.x;
    if (i < S) {
        const T* r = qkv + (long long)(b * S + i) * 3 * D + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) { q[i][d] = to_f<T>(r[d]) * 0.25f; k[i][d] = to_f<T>(r[D + d]); v[i][d] = to_f<T>(r[2 * D + d]); }
    }
    __syncthreads();
    if (i >= S) return;
    float sc[SMAX];
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < SMAX; ++j) {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += q[i][d] * k[j < S ? j : 0][d];
        sc[j] = j < S ? s : -INFINITY;
        m = fmaxf(m, sc[j]);
    }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < SMAX; ++j) { sc[j] = j < S ? __expf(sc[j] - m) : 0.f; den += sc[j]; }
    const float inv = 1.f / den;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    float* Pr = P + (((long long)b * NH + h) * S + i) * S;
#pragma unroll
    for (int j = 0; j < SMAX; ++j) {
        if (j >= S) break;
        float p = sc[j] * inv;
        Pr[j] = p;
        if (drop_p > 0.f) p = hash_uniform(seed, (((long long)b * NH + h) * S + i) * S + j) < drop_p ? 0.f : p / (1.f - drop_p);
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] += p * v[j][d];
    }
    T* orow = ao + (long long)(b * S + i) * D + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) orow[d] = from_f<T>(o[d]);
}
// backward: dao [B*S][D] (T) -> dqkv [B*S][3D] (T)
template <typename T, int SMAX>
__global__ void __launch_bounds__(64) attention_bwd_kernel(const T* __restrict__ qkv, const float* __restrict__ P, const T* __restrict__ dao,
                                                           int B, int S, int D, int NH, T* __restrict__ dqkv, float drop_p,
                                                           unsigned long long seed) {
    constexpr int HD = 16;
    __shared__ float q[SMAX][HD + 1], k[SMAX][HD + 1], v[SMAX][HD + 1], dO[SMAX][HD + 1];
    __shared__ float dS[SMAX][SMAX + 1];    // dS[i][j]
    __shared__ float Pd[SMAX][SMAX + 1];    // dropped P[i][j] (for dV)
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, i = threadIdx.x;
    if (i < S) {
        const T* r = qkv + (long long)(b * S + i) * 3 * D + h * HD;
        const T* g = dao + (long long)(b * S + i) * D + h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            q[i][d] = to_f<T>(r[d]) * 0.25f; k[i][d] = to_f<T>(r[D + d]); v[i][d] = to_f<T>(r[2 * D + d]);
            dO[i][d] = to_f<T>(g[d]);
        }
    }
    __syncthreads();
    if (i < S) {
        const float* Pr = P + (((long long)b * NH + h) * S + i) * S;
        float dot = 0.f;
        float dp[SMAX];
#pragma unroll
        for (int j = 0; j < SMAX; ++j) {
            if (j >= S) break;
            float dpj = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) dpj += dO[i][d] * v[j][d];
            const float p = Pr[j];
            float keep = 1.f;
            if (drop_p > 0.f) keep = hash_uniform(seed, (((long long)b * NH + h) * S + i) * S + j) < drop_p ? 0.f : 1.f / (1.f - drop_p);
            Pd[i][j] = p * keep;
            dpj *= keep;              // grad w.r.t. pre-dropout P
            dp[j] = dpj;
            dot += dpj * p;
        }
#pragma unroll
        for (int j = 0; j < SMAX; ++j) if (j < S) dS[i][j] = Pr[j] * (dp[j] - dot);
    }
    __syncthreads();
    if (i >= S) return;
    float dq[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { dq[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    for (int j = 0; j < S; ++j) {
        const float s_ij = dS[i][j], s_ji = dS[j][i], p_ji = Pd[j][i];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            dq[d] += s_ij * k[j][d];
            dk[d] += s_ji * q[j][d];          // q already carries the 1/sqrt(hd) scale
            dv[d] += p_ji * dO[j][d];
        }
    }
    T* o = dqkv + (long long)(b * S + i) * 3 * D + h * HD;
#pragma unroll
    for (int d = 0; d < HD; ++d) { o[d] = from_f<T>(dq[d] * 0.25f); o[D + d] = from_f<T>(dk[d]); o[2 * D + d] = from_f<T>(dv[d]); }
}

// S <= 32 (the benchmark's windows): the same two kernels with BOTH halves of the wave working — lane (i, hf = lane >> 5) handles the keys
// [16 hf, 16 hf + 16) of query row i and the halves are combined with one cross-half shuffle per value.  The one-lane-per-query kernels above leave
// 32 of 64 lanes idle and run at half a wave per SIMD, i.e. at the latency of one thread's serial chain (14 / 22 us per launch for 2 MB of data).
template <typename T>
__global__ void __launch_bounds__(64) attention_fwd32_kernel(const T* __restrict__ qkv, int B, int S, int D, int NH, float* __restrict__ P,
                                                             T* __restrict__ ao, float drop_p, unsigned long long seed) {
    constexpr int HD = 16, SMAX = 32, HJ = 16;
    __shared__ float q[SMAX][HD + 1], k[SMAX][HD + 1], v[SMAX][HD + 1];
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, i = threadIdx.x & 31, hf = threadIdx.x >> 5;
    if (i < S) {
        const T* r = qkv + (long long)(b * S + i) * 3 * D + h * HD;
        if (hf == 0) {
#pragma unroll
            for (int d = 0; d < HD; ++d) { q[i][d] = to_f<T>(r[d]) * 0.25f; k[i][d] = to_f<T>(r[D + d]); }
        } else {
#pragma unroll
            for (int d = 0; d < HD; ++d) v[i][d] = to_f<T>(r[2 * D + d]);
        }
    }
    __syncthreads();
    if (i >= S) return;                       // both halves of a query leave together: the shuffles below pair live lanes only
    float sc[HJ];
    float m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < HJ; ++jj) {
        const int j = hf * HJ + jj;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += q[i][d] * k[j < S ? j : 0][d];
        sc[jj] = j < S ? s : -INFINITY;
        m = fmaxf(m, sc[jj]);
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float den = 0.f;
#pragma unroll
    for (int jj = 0; jj < HJ; ++jj) { sc[jj] = (hf * HJ + jj) < S ? __expf(sc[jj] - m) : 0.f; den += sc[jj]; }
    den += __shfl_xor(den, 32);
    const float inv = 1.f / den;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    float* Pr = P + (((long long)b * NH + h) * S + i) * S;
#pragma unroll
    for (int jj = 0; jj < HJ; ++jj) {
        const int j = hf * HJ + jj;
        if (j < S) {
            float p = sc[jj] * inv;
            Pr[j] = p;
            if (drop_p > 0.f) p = hash_uniform(seed, (((long long)b * NH + h) * S + i) * S + j) < drop_p ? 0.f : p / (1.f - drop_p);
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] += p * v[j][d];
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] += __shfl_xor(o[d], 32);
    T* orow = ao + (long long)(b * S + i) * D + h * HD + hf * 8;
#pragma unroll
    for (int d = 0; d < 8; ++d) orow[d] = from_f<T>(hf ? o[8 + d] : o[d]);
}
template <typename T>
__global__ void __launch_bounds__(64) attention_bwd32_kernel(const T* __restrict__ qkv, const float* __restrict__ P, const T* __restrict__ dao,
                                                             int B, int S, int D, int NH, T* __restrict__ dqkv, float drop_p,
                                                             unsigned long long seed) {
    constexpr int HD = 16, SMAX = 32, HJ = 16;
    __shared__ float q[SMAX][HD + 1], k[SMAX][HD + 1], v[SMAX][HD + 1], dO[SMAX][HD + 1];
    __shared__ float dS[SMAX][SMAX + 1];    // dS[i][j]
    __shared__ float Pd[SMAX][SMAX + 1];    // dropped P[i][j] (for dV)
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, i = threadIdx.x & 31, hf = threadIdx.x >> 5;
    if (i < S) {
        const T* r = qkv + (long long)(b * S + i) * 3 * D + h * HD;
        const T* g = dao + (long long)(b * S + i) * D + h * HD;
        // a head's 16 values of a row are contiguous: two vector loads per operand instead of sixteen element loads (the kernel is one
        // latency chain per wave; 32 dependent 2-byte loads per lane were most of its 16 us)
        T a0[HD], a1[HD];
        if (hf == 0) { load8<T>(r, *reinterpret_cast<T(*)[8]>(a0)); load8<T>(r + 8, *reinterpret_cast<T(*)[8]>(a0 + 8)); load8<T>(r + D, *reinterpret_cast<T(*)[8]>(a1)); load8<T>(r + D + 8, *reinterpret_cast<T(*)[8]>(a1 + 8)); }
        else { load8<T>(r + 2 * D, *reinterpret_cast<T(*)[8]>(a0)); load8<T>(r + 2 * D + 8, *reinterpret_cast<T(*)[8]>(a0 + 8)); load8<T>(g, *reinterpret_cast<T(*)[8]>(a1)); load8<T>(g + 8, *reinterpret_cast<T(*)[8]>(a1 + 8)); }
        if (hf == 0) {
#pragma unroll
            for (int d = 0; d < HD; ++d) { q[i][d] = to_f<T>(a0[d]) * 0.25f; k[i][d] = to_f<T>(a1[d]); }
        } else {
#pragma unroll
            for (int d = 0; d < HD; ++d) { v[i][d] = to_f<T>(a0[d]); dO[i][d] = to_f<T>(a1[d]); }
        }
    }
    __syncthreads();
    if (i < S) {
        const float* Pr = P + (((long long)b * NH + h) * S + i) * S;
        float dot = 0.f;
        float dp[HJ], pv[HJ];
#pragma unroll
        for (int jj = 0; jj < HJ; ++jj) {
            const int j = hf * HJ + jj;
            dp[jj] = 0.f; pv[jj] = 0.f;
            if (j < S) {
                float dpj = 0.f;
#pragma unroll
                for (int d = 0; d < HD; ++d) dpj += dO[i][d] * v[j][d];
                const float p = Pr[j];
                float keep = 1.f;
                if (drop_p > 0.f) keep = hash_uniform(seed, (((long long)b * NH + h) * S + i) * S + j) < drop_p ? 0.f : 1.f / (1.f - drop_p);
                Pd[i][j] = p * keep;
                dpj *= keep;              // grad w.r.t. pre-dropout P
                dp[jj] = dpj; pv[jj] = p;
                dot += dpj * p;
            }
        }
        dot += __shfl_xor(dot, 32);
#pragma unroll
        for (int jj = 0; jj < HJ; ++jj) if (hf * HJ + jj < S) dS[i][hf * HJ + jj] = pv[jj] * (dp[jj] - dot);
    }
    __syncthreads();
    if (i >= S) return;
    float dq[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { dq[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    for (int jj = 0; jj < HJ; ++jj) {
        const int j = hf * HJ + jj;
        if (j >= S) break;
        const float s_ij = dS[i][j], s_ji = dS[j][i], p_ji = Pd[j][i];
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            dq[d] += s_ij * k[j][d];
            dk[d] += s_ji * q[j][d];          // q already carries the 1/sqrt(hd) scale
            dv[d] += p_ji * dO[j][d];
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) { dq[d] += __shfl_xor(dq[d], 32); dk[d] += __shfl_xor(dk[d], 32); dv[d] += __shfl_xor(dv[d], 32); }
    T* o = dqkv + (long long)(b * S + i) * 3 * D + h * HD + hf * 8;
    Vec8<T> oq, ok, ov;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        const int e = hf * 8 + d;
        oq.v[d] = from_f<T>((hf ? dq[8 + d] : dq[d]) * 0.25f); ok.v[d] = from_f<T>(hf ? dk[8 + d] : dk[d]); ov.v[d] = from_f<T>(hf ? dv[8 + d] : dv[d]);
        (void)e;
    }
    *reinterpret_cast<Vec8<T>*>(o) = oq; *reinterpret_cast<Vec8<T>*>(o + D) = ok; *reinterpret_cast<Vec8<T>*>(o + 2 * D) = ov;
}

// 32 < S <= 64 (BASELINE config 5's windows): four waves per (b, head), wave w handles the keys [16 w, 16 w + 16) of every query row (lane = row);
// the four partials of a row meet in LDS.  The one-lane-per-row kernels take 47 / 52 us per launch at S = 64, B = 32 (64 serial keys per thread).
template <typename T>
__global__ void __launch_bounds__(256) attention_fwd64_kernel(const T* __restrict__ qkv, int B, int S, int D, int NH, float* __restrict__ P,
                                                              T* __restrict__ ao, float drop_p, unsigned long long seed) {
    constexpr int HD = 16, SMAX = 64, HJ = 16;
    __shared__ float q[SMAX][HD + 1], k[SMAX][HD + 1], v[SMAX][HD + 1];
    __shared__ float pm[4][SMAX], pd[4][SMAX], po[4][SMAX][HD + 1];
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, i = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (i < S) {
        const T* r = qkv + (long long)(b * S + i) * 3 * D + h * HD;
#pragma unroll
        for (int d = w * 4; d < w * 4 + 4; ++d) { q[i][d] = to_f<T>(r[d]) * 0.25f; k[i][d] = to_f<T>(r[D + d]); v[i][d] = to_f<T>(r[2 * D + d]); }
    }
    __syncthreads();
    const bool live = i < S;
    float sc[HJ];
    float m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < HJ; ++jj) {
        const int j = w * HJ + jj;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s += q[live ? i : 0][d] * k[j < S ? j : 0][d];
        sc[jj] = (live && j < S) ? s : -INFINITY;
        m = fmaxf(m, sc[jj]);
    }
    pm[w][i] = m;
    __syncthreads();
    m = fmaxf(fmaxf(pm[0][i], pm[1][i]), fmaxf(pm[2][i], pm[3][i]));
    float den = 0.f;
#pragma unroll
    for (int jj = 0; jj < HJ; ++jj) { sc[jj] = (live && w * HJ + jj < S) ? __expf(sc[jj] - m) : 0.f; den += sc[jj]; }
    pd[w][i] = den;
    __syncthreads();
    den = (pd[0][i] + pd[1][i]) + (pd[2][i] + pd[3][i]);
    const float inv = live ? 1.f / den : 0.f;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.f;
    float* Pr = P + (((long long)b * NH + h) * S + (live ? i : 0)) * S;
#pragma unroll
    for (int jj = 0; jj < HJ; ++jj) {
        const int j = w * HJ + jj;
        if (live && j < S) {
            float p = sc[jj] * inv;
            Pr[j] = p;
            if (drop_p > 0.f) p = hash_uniform(seed, (((long long)b * NH + h) * S + i) * S + j) < drop_p ? 0.f : p / (1.f - drop_p);
#pragma unroll
            for (int d = 0; d < HD; ++d) o[d] += p * v[j][d];
        }
    }
#pragma unroll
    for (int d = 0; d < HD; ++d) po[w][i][d] = o[d];
    __syncthreads();
    if (!live) return;
    T* orow = ao + (long long)(b * S + i) * D + h * HD + w * 4;
#pragma unroll
    for (int d = 0; d < 4; ++d) { const int e = w * 4 + d; orow[d] = from_f<T>((po[0][i][e] + po[1][i][e]) + (po[2][i][e] + po[3][i][e])); }
}
template <typename T>
__global__ void __launch_bounds__(256) attention_bwd64_kernel(const T* __restrict__ qkv, const float* __restrict__ P, const T* __restrict__ dao,
                                                              int B, int S, int D, int NH, T* __restrict__ dqkv, float drop_p,
                                                              unsigned long long seed) {
    constexpr int HD = 16, SMAX = 64, HJ = 16;
    extern __shared__ __attribute__((aligned(16))) float att_smem[];
    float (*q)[HD + 1] = reinterpret_cast<float (*)[HD + 1]>(att_smem);
    float (*k)[HD + 1] = q + SMAX;
    float (*v)[HD + 1] = k + SMAX;
    float (*dO)[HD + 1] = v + SMAX;
    float (*dS)[SMAX + 1] = reinterpret_cast<float (*)[SMAX + 1]>(dO + SMAX);     // dS[i][j]
    float (*Pd)[SMAX + 1] = dS + SMAX;                                             // dropped P[i][j] (for dV)
    float (*pdot)[SMAX] = reinterpret_cast<float (*)[SMAX]>(Pd + SMAX);            // [4][SMAX]
    float (*pg)[SMAX][3 * HD + 1] = reinterpret_cast<float (*)[SMAX][3 * HD + 1]>(pdot + 4);   // [4][SMAX][49]: partial dq | dk | dv
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, i = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (i < S) {
        const T* r = qkv + (long long)(b * S + i) * 3 * D + h * HD;
        const T* g = dao + (long long)(b * S + i) * D + h * HD;
#pragma unroll
        for (int d = w * 4; d < w * 4 + 4; ++d) {
            q[i][d] = to_f<T>(r[d]) * 0.25f; k[i][d] = to_f<T>(r[D + d]); v[i][d] = to_f<T>(r[2 * D + d]); dO[i][d] = to_f<T>(g[d]);
        }
    }
    __syncthreads();
    const bool live = i < S;
    const float* Pr = P + (((long long)b * NH + h) * S + (live ? i : 0)) * S;
    float dot = 0.f;
    float dp[HJ], pv[HJ];
#pragma unroll
    for (int jj = 0; jj < HJ; ++jj) {
        const int j = w * HJ + jj;
        dp[jj] = 0.f; pv[jj] = 0.f;
        if (live && j < S) {
            float dpj = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) dpj += dO[i][d] * v[j][d];
            const float p = Pr[j];
            float keep = 1.f;
            if (drop_p > 0.f) keep = hash_uniform(seed, (((long long)b * NH + h) * S + i) * S + j) < drop_p ? 0.f : 1.f / (1.f - drop_p);
            Pd[i][j] = p * keep;
            dpj *= keep;
            dp[jj] = dpj; pv[jj] = p;
            dot += dpj * p;
        }
    }
    pdot[w][i] = dot;
    __syncthreads();
    dot = (pdot[0][i] + pdot[1][i]) + (pdot[2][i] + pdot[3][i]);
#pragma unroll
    for (int jj = 0; jj < HJ; ++jj) if (live && w * HJ + jj < S) dS[i][w * HJ + jj] = pv[jj] * (dp[jj] - dot);
    __syncthreads();
    float dq[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { dq[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    if (live)
        for (int jj = 0; jj < HJ; ++jj) {
            const int j = w * HJ + jj;
            if (j >= S) break;
            const float s_ij = dS[i][j], s_ji = dS[j][i], p_ji = Pd[j][i];
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dq[d] += s_ij * k[j][d];
                dk[d] += s_ji * q[j][d];
                dv[d] += p_ji * dO[j][d];
            }
        }
#pragma unroll
    for (int d = 0; d < HD; ++d) { pg[w][i][d] = dq[d]; pg[w][i][HD + d] = dk[d]; pg[w][i][2 * HD + d] = dv[d]; }
    __syncthreads();
    if (!live) return;
    T* o = dqkv + (long long)(b * S + i) * 3 * D + h * HD + w * 4;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int e = w * 4 + d;
        const float sq = (pg[0][i][e] + pg[1][i][e]) + (pg[2][i][e] + pg[3][i][e]);
        const float sk = (pg[0][i][HD + e] + pg[1][i][HD + e]) + (pg[2][i][HD + e] + pg[3][i][HD + e]);
        const float sv = (pg[0][i][2 * HD + e] + pg[1][i][2 * HD + e]) + (pg[2][i][2 * HD + e] + pg[3][i][2 * HD + e]);
        o[d] = from_f<T>(sq * 0.25f); o[D + d] = from_f<T>(sk); o[2 * D + d] = from_f<T>(sv);
    }
}
static constexpr size_t ATT_BWD64_LDS = sizeof(float) * (4 * 64 * 17 + 2 * 64 * 65 + 4 * 64 + 4 * 64 * 49);

// xm[b][d] = mean_t x[b][t][d]   (fp32 in, T out)
template <typename T>
__global__ void mean_over_s_kernel(const float* __restrict__ x, int B, int S, int D, T* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * D) return;
    const int b = idx / D, d = idx % D;
    float s = 0.f;
#pragma unroll 8
    for (int t = 0; t < S; ++t) s += x[((long long)b * S + t) * D + d];      // eight independent loads in flight (the sum order is unchanged)
    out[idx] = from_f<T>(s / S);
}
// LayerNorm of the S rows of window b followed by their mean over S (the plan-recognition transformer's last norm2 + the mean that feeds fc,
// plan_recognition_net.py:110-114): one block per window, a wave per row (16 rows at a time); stats [rows][2] kept for the backward, n <= 128.
template <typename T>
__global__ void __launch_bounds__(1024) layernorm_mean_kernel(const float* __restrict__ x, int S, int n, const float* __restrict__ g, const float* __restrict__ bta,
                                                             float* __restrict__ stats, T* __restrict__ out, float* __restrict__ yout) {
    __shared__ float part[16][128];
    const int b = blockIdx.x, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float a0 = 0.f, a1 = 0.f;
    for (int t = w; t < S; t += 16) {
        const long long row = (long long)b * S + t;
        const float* xr = x + row * n;
        const float v0 = lane < n ? xr[lane] : 0.f, v1 = lane + 64 < n ? xr[lane + 64] : 0.f;
        const float mean = wave_sum(v0 + v1) / n;
        const float d0 = lane < n ? v0 - mean : 0.f, d1 = lane + 64 < n ? v1 - mean : 0.f;
        const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1) / n + 1e-5f);
        if (lane < n) { const float y = d0 * rstd * g[lane] + bta[lane]; a0 += y; yout[row * n + lane] = y; }
        if (lane + 64 < n) { const float y = d1 * rstd * g[lane + 64] + bta[lane + 64]; a1 += y; yout[row * n + lane + 64] = y; }
        if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    }
    part[w][lane] = a0; part[w][lane + 64] = a1;
    __syncthreads();
    if (threadIdx.x < n) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[k][threadIdx.x];
        out[(long long)b * n + threadIdx.x] = from_f<T>(s / S);
    }
}
// dx[b][t][d] = dxm[b][d] / S
__global__ void bcast_over_s_kernel(const float* __restrict__ dxm, int B, int S, int D, float* __restrict__ dx) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)B * S * D) return;
    const int d = idx % D;
    const int b = idx / ((long long)S * D);
    dx[idx] = dxm[b * D + d] / S;
}
// backward of x0 = dropout(emb + pos) in ONE launch (was dropout_apply + copy2d + pos_grad): g = mask * dx / (1 - p);
// demb[b][t][d] += g ; dpos[t][d] += sum_b g.  One thread per (t, d) and window chunk (grid.y chunks of `bchunk` windows, eight in flight);
// grid.y == 1 (fp32 parity engine) adds b ascending like pos_grad_kernel, grid.y > 1 lands each chunk's partial sum with one fp32 atomic.
__global__ void __launch_bounds__(256) pr_input_bwd_kernel(const float* __restrict__ dx, int B, int S, int D, float drop_p, unsigned long long seed,
                                                           float* __restrict__ demb, float* __restrict__ dpos, int bchunk) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * D) return;
    const long long SD = (long long)S * D;
    const int blo = blockIdx.y * bchunk, bhi = min(B, blo + bchunk);
    float s = 0.f;
    for (int b0 = blo; b0 < bhi; b0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = b0 + u < bhi ? dx[(b0 + u) * SD + idx] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (b0 + u >= bhi) break;
            const long long o = (b0 + u) * SD + idx;
            float g = v[u];
            if (drop_p > 0.f) g = hash_uniform(seed, o) < drop_p ? 0.f : g / (1.f - drop_p);
            demb[o] += g;
            s += g;
        }
    }
    if (gridDim.y > 1) unsafeAtomicAdd(dpos + idx, s);
    else dpos[idx] += s;
}
// dpos[t][d] += sum_b dx[b][t][d]
__global__ void pos_grad_kernel(const float* __restrict__ dx, int B, int S, int D, float* __restrict__ dpos) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * D) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dx[(long long)b * S * D + idx];
    dpos[idx] += s;
}

// =========================================================================================================
// plan distribution: categorical sample, KL balancing (hulc.py:539-561), straight-through (distributions.py:23-27)
// one wave per (b, category); NCLS <= 64
// =========================================================================================================
__global__ void __launch_bounds__(64) plan_kl_sample_kernel(const float* __restrict__ pr_logits, const float* __restrict__ pp_logits, int B,
                                                            int NCAT, int NCLS, const int* __restrict__ idx_in, int* __restrict__ idx_out,
                                                            float* __restrict__ probs /*[B][NCAT][NCLS]*/, float* __restrict__ kl_cat /*[B][NCAT]*/,
                                                            float* __restrict__ dpp, float* __restrict__ dpr, float w_pp, float w_pr,
                                                            unsigned long long seed, const float* __restrict__ lscale = nullptr) {
    if (lscale) { const float ls = lscale[0]; w_pp *= ls; w_pr *= ls; }      // dynamic loss scale (fp16 mode): gradients only, the KL value is unscaled
    const int bc = blockIdx.x, lane = threadIdx.x;
    const long long base = (long long)bc * NCLS;
    const bool ok = lane < NCLS;
    const float y = ok ? pr_logits[base + lane] : -INFINITY;
    const float my = wave_max(y);
    const float ey = ok ? __expf(y - my) : 0.f;
    const float sy = wave_sum(ey);
    const float a = ok ? (y - my) - __logf(sy) : 0.f;
    const float p = ey / sy;
    if (ok) probs[base + lane] = p;
    if (pp_logits) {
        const float z = ok ? pp_logits[base + lane] : -INFINITY;
        const float mz = wave_max(z);
        const float ez = ok ? __expf(z - mz) : 0.f;
        const float sz = wave_sum(ez);
        const float bb = ok ? (z - mz) - __logf(sz) : 0.f;
        const float q = ez / sz;
        const float klc = wave_sum(ok ? p * (a - bb) : 0.f);
        if (lane == 0) kl_cat[bc] = klc;
        if (ok) {
            dpp[base + lane] = w_pp * (q - p);
            dpr[base + lane] = w_pr * p * ((a - bb) - klc);
        }
    }
    // categorical sample by Gumbel-max unless injected
    int sel;
    if (idx_in) sel = idx_in[bc];
    else {
        float gmb = ok ? y - __logf(-__logf(hash_uniform(seed, base + lane))) : -INFINITY;
        const float gm = wave_max(gmb);
        unsigned long long ball = __ballot(ok && gmb == gm);
        sel = __ffsll((long long)ball) - 1;
    }
    if (lane == 0) idx_out[bc] = sel;
}
// mcil: continuous latent plan.  state = [mean | var] (B, 2n), std = softplus(var) + 1e-4 (distributions.py:55-59);
// KL(N(m1,s1) || N(m2,s2)) per element with the balancing weights (hulc.py:539-561), reparametrised sample plan = m1 + s1 * eps
// (hulc.py:289) with an injected or Box-Muller draw.  One thread per (b, j).
DEVI float softplus_k(float x) { return x > 20.f ? x : (x < -20.f ? __expf(x) : log1pf(__expf(x))); }
template <typename T>
__global__ void normal_kl_sample_kernel(const float* __restrict__ pr_state, const float* __restrict__ pp_state, int B, int n,
                                        const float* __restrict__ eps_in, float* __restrict__ eps_out, float* __restrict__ plan_f,
                                        T* __restrict__ plan_t, float* __restrict__ kl_elem, float* __restrict__ dpp, float* __restrict__ dpr,
                                        float w_pp, float w_pr, unsigned long long seed, const float* __restrict__ lscale = nullptr) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n) return;
    if (lscale) { const float ls = lscale[0]; w_pp *= ls; w_pr *= ls; }
    const int b = idx / n, j = idx % n;
    const long long o = (long long)b * 2 * n + j;
    const float m1 = pr_state[o], v1 = pr_state[o + n], s1 = softplus_k(v1) + 1e-4f;
    float e;
    if (eps_in) e = eps_in[idx];
    else {
        const float u1 = fmaxf(hash_uniform(seed, 2ull * idx), 1e-12f), u2 = hash_uniform(seed, 2ull * idx + 1);
        e = sqrtf(-2.f * __logf(u1)) * __cosf(6.28318530717958647692f * u2);
    }
    eps_out[idx] = e;
    const float pl = m1 + s1 * e;
    plan_f[idx] = pl;
    plan_t[idx] = from_f<T>(pl);
    if (pp_state) {
        const float m2 = pp_state[o], v2 = pp_state[o + n], s2 = softplus_k(v2) + 1e-4f;
        const float dm = m1 - m2, i2 = 1.f / (s2 * s2);
        kl_elem[idx] = __logf(s2 / s1) + (s1 * s1 + dm * dm) * 0.5f * i2 - 0.5f;
        const float sg1 = 1.f / (1.f + __expf(-v1)), sg2 = 1.f / (1.f + __expf(-v2));
        dpp[o] = w_pp * (-dm * i2);
        dpp[o + n] = w_pp * (1.f / s2 - (s1 * s1 + dm * dm) * i2 / s2) * sg2;
        dpr[o] = w_pr * (dm * i2);
        dpr[o + n] = w_pr * (-1.f / s1 + s1 * i2) * sg1;
    }
}
// d pr_state = [dplan | dplan * eps * sigmoid(var)] + dpr_kl
template <typename T>
__global__ void normal_rsample_bwd_kernel(const float* __restrict__ dplan, const float* __restrict__ eps, const float* __restrict__ pr_state,
                                          const float* __restrict__ dpr_kl, int B, int n, T* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * n) return;
    const int b = idx / n, j = idx % n;
    const long long o = (long long)b * 2 * n + j;
    const float g = dplan[idx], v1 = pr_state[o + n];
    out[o] = from_f<T>(g + dpr_kl[o]);
    out[o + n] = from_f<T>(g * eps[idx] / (1.f + __expf(-v1)) + dpr_kl[o + n]);
}
// mcil with plan_recognition.rnn_type = nn.GRU (torch.nn.GRU cell): one time step of one direction, elementwise part.
// zx (B,3H) = W_i x + b_i (gate blocks r | z | n), g (B,3H) fp32 = W_h h + b_h or null (h = 0: g = b_h), hprev (B,H) or null.
//   r = sig(zx_r + g_r), z = sig(zx_z + g_z), n = tanh(zx_n + r * g_n), h' = (1 - z) n + z h
// r, z, n and g_n are kept for the backward.
template <typename T>
__global__ void gru_gate_fwd_kernel(const T* __restrict__ zx, const float* __restrict__ g, const float* __restrict__ bhh, const T* __restrict__ hprev,
                                    int B, int H, T* __restrict__ h_out, T* __restrict__ r_out, T* __restrict__ z_out, T* __restrict__ n_out,
                                    T* __restrict__ gn_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, j = idx % H;
    const long long o3 = (long long)b * 3 * H + j;
    const float gr = g ? g[o3] : bhh[j], gz = g ? g[o3 + H] : bhh[H + j], gn = g ? g[o3 + 2 * H] : bhh[2 * H + j];
    const float r = 1.f / (1.f + __expf(-(to_f<T>(zx[o3]) + gr)));
    const float z = 1.f / (1.f + __expf(-(to_f<T>(zx[o3 + H]) + gz)));
    const float n = tanhf(to_f<T>(zx[o3 + 2 * H]) + r * gn);
    const float hp = hprev ? to_f<T>(hprev[idx]) : 0.f;
    h_out[idx] = from_f<T>((1.f - z) * n + z * hp);
    r_out[idx] = from_f<T>(r); z_out[idx] = from_f<T>(z); n_out[idx] = from_f<T>(n); gn_out[idx] = from_f<T>(gn);
}
// backward of the same step: dh = dH (or 0) + carry (or 0) -> gradients of the input-side pre-activations dzx = (dr, dz, dn), of the
// hidden-side ones dg = (dr, dz, dn * r), and the direct path dh * z to the previous state (the caller adds dg W_h)
template <typename T>
__global__ void gru_gate_bwd_kernel(const T* __restrict__ dH, const T* __restrict__ carry, const T* __restrict__ r_, const T* __restrict__ z_,
                                    const T* __restrict__ n_, const T* __restrict__ gn_, const T* __restrict__ hprev, int B, int H,
                                    T* __restrict__ dzx, T* __restrict__ dg, T* __restrict__ direct) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H) return;
    const int b = idx / H, j = idx % H;
    const long long o3 = (long long)b * 3 * H + j;
    const float dh = (dH ? to_f<T>(dH[idx]) : 0.f) + (carry ? to_f<T>(carry[idx]) : 0.f);
    const float r = to_f<T>(r_[idx]), z = to_f<T>(z_[idx]), n = to_f<T>(n_[idx]), gn = to_f<T>(gn_[idx]);
    const float hp = hprev ? to_f<T>(hprev[idx]) : 0.f;
    const float dn = dh * (1.f - z) * (1.f - n * n);
    const float dz = dh * (hp - n) * z * (1.f - z);
    const float dr = dn * gn * r * (1.f - r);
    dzx[o3] = from_f<T>(dr); dzx[o3 + H] = from_f<T>(dz); dzx[o3 + 2 * H] = from_f<T>(dn);
    dg[o3] = from_f<T>(dr); dg[o3 + H] = from_f<T>(dz); dg[o3 + 2 * H] = from_f<T>(dn * r);
    direct[idx] = from_f<T>(dh * z);
}
// dpr_logits[b][cat][:] = probs * (dplan - sum(probs*dplan)) + dpr_kl
// also emits the compute-precision copies the two backward passes start from: out_t (this gradient) and dpp_t (= dpp_kl, the plan-proposal
// logits' KL gradient of the same shape) — two cast launches less
template <typename T>
__global__ void __launch_bounds__(64) st_softmax_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ dplan,
                                                            const float* __restrict__ dpr_kl, int NCLS, float* __restrict__ out, T* __restrict__ out_t,
                                                            const float* __restrict__ dpp_kl, T* __restrict__ dpp_t) {
    const long long base = (long long)blockIdx.x * NCLS;
    const int lane = threadIdx.x;
    const bool ok = lane < NCLS;
    const float p = ok ? probs[base + lane] : 0.f, d = ok ? dplan[base + lane] : 0.f;
    const float dot = wave_sum(p * d);
    if (ok) {
        const float v = p * (d - dot) + dpr_kl[base + lane];
        out[base + lane] = v;
        out_t[base + lane] = from_f<T>(v);
        dpp_t[base + lane] = from_f<T>(dpp_kl[base + lane]);
    }
}

// =========================================================================================================
// decoder glue (logistic_decoder_rnn.py:260-287 with the plan/goal terms hoisted out of the time loop)
// =========================================================================================================
// Cplan[b][i] = b_ih[i] + b_hh[i] + sum_cat WihT[cat*NCLS + idx[b][cat]][i]     (one-hot plan x W_ih[:, plan]^T)
// read from the transposed compute copy WihT [KIN][H]: consecutive threads read consecutive i (coalesced)
template <typename T>
__global__ void plan_gather_t_kernel(const T* __restrict__ w_t /*[KIN][H]*/, const int* __restrict__ idx, int B, int NCAT, int NCLS, int H,
                                     const float* __restrict__ b1, const float* __restrict__ b2, float* __restrict__ out,
                                     const T* __restrict__ emb = nullptr, T* __restrict__ embg = nullptr, int S = 0, int W = 0,
                                     const T* __restrict__ goal = nullptr, int G = 0, int grow0 = 0, T* __restrict__ cb = nullptr) {
    // cb (optional): the decoder's whole time-invariant input term in one launch — Cb[b][i] = Cplan[b][i] + sum_k goal[b][k] WihT[grow0 + k][i]
    // (the K = 32 GEMM that used to follow this launch)
    // blocks past the B * H / blockDim of the plan gather carry the (independent) time-major copy of the embedding's last W columns
    // (gather_embg_kernel's job: embg[(t*B+b)*W + c] = emb[(b*S+t)*128 + (128-W) + c]) — one launch instead of two
    const int nb_plan = (B * H + blockDim.x - 1) / blockDim.x;
    if ((int)blockIdx.x >= nb_plan) {
        const int i2 = (blockIdx.x - nb_plan) * blockDim.x + threadIdx.x;
        if (i2 < S * B * W) {
            const int c = i2 % W, r = i2 / W;
            const int t = r / B, b = r % B;
            embg[i2] = emb[((long long)b * S + t) * 128 + (128 - W) + c];
        }
        return;
    }
    // H % blockDim == 0: a block lies within one b, so its NCAT row indices are fetched once and the NCAT weight loads are independent
    __shared__ int sidx[64];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = (blockIdx.x * blockDim.x) / H, i = gid % H;
    if ((int)threadIdx.x < NCAT) sidx[threadIdx.x] = idx[b * NCAT + threadIdx.x];
    __syncthreads();
    if (gid >= B * H) return;
    float s = b1[i] + b2[i];
    if (NCAT == 32) {
        float w[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) w[c] = to_f<T>(w_t[(long long)(c * NCLS + sidx[c]) * H + i]);
#pragma unroll
        for (int c = 0; c < 32; ++c) s += w[c];
    } else
        for (int c = 0; c < NCAT; ++c) s += to_f<T>(w_t[(long long)(c * NCLS + sidx[c]) * H + i]);
    out[gid] = s;
    if (cb) {
        float a = 0.f;
        if (G == 32) {           // 32 independent loads in flight, like the plan rows above
            float w[32];
#pragma unroll
            for (int k = 0; k < 32; ++k) w[k] = to_f<T>(w_t[(long long)(grow0 + k) * H + i]);
#pragma unroll
            for (int k = 0; k < 32; ++k) a += to_f<T>(goal[b * 32 + k]) * w[k];
        } else
            for (int k = 0; k < G; ++k) a += to_f<T>(goal[b * G + k]) * to_f<T>(w_t[(long long)(grow0 + k) * H + i]);
        cb[gid] = from_f<T>(s + a);
    }
}
// dW_ih[i][cat*NCLS + cls] += sum_{b: idx[b][cat] == cls} dC[b][i], b-ordered.  Block (cat, 64-wide i tile): the [NCLS][64] tile is
// accumulated in LDS with coalesced dC reads, then added to dW with NCLS consecutive floats per row (128-byte segments).
template <typename T>
__global__ void __launch_bounds__(64) plan_scatter_grad_lds_kernel(const T* __restrict__ dC, const int* __restrict__ idx, int B, int NCAT, int NCLS, int H,
                                                                   int KIN, float* __restrict__ dw) {
    __shared__ float tile[32][65];
    __shared__ int sidx[64];
    const int c = blockIdx.x, i0 = blockIdx.y * 64, t = threadIdx.x;
    for (int k = 0; k < 32; ++k) tile[k][t] = 0.f;
    for (int b0 = 0; b0 < B; b0 += 64) {          // 64 windows at a time: indices once per block, the 64 dC loads of a lane issued together
        __syncthreads();
        sidx[t] = b0 + t < B ? idx[(b0 + t) * NCAT + c] : 0;
        float v[64];
#pragma unroll
        for (int b = 0; b < 64; ++b) v[b] = (b0 + b < B && i0 + t < H) ? to_f<T>(dC[(long long)(b0 + b) * H + i0 + t]) : 0.f;
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 64; ++b) tile[sidx[b]][t] += v[b];
    }
    __syncthreads();
    const int cls = t & 31, half = t >> 5;
    float old[32];                                    // the 32 read-modify-writes of a lane: all loads first (one latency, not 32)
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int r = half + 2 * k;
        old[k] = (i0 + r < H && cls < NCLS) ? dw[(long long)(i0 + r) * KIN + c * NCLS + cls] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const int r = half + 2 * k;
        if (i0 + r < H && cls < NCLS) dw[(long long)(i0 + r) * KIN + c * NCLS + cls] = old[k] + tile[cls][r];
    }
}
// the same with four waves per (category, 64-unit tile) — each wave scatters 16 of every 64 windows into its OWN LDS tile (the serial kernel's 64 dependent LDS
// read-add-writes per lane were its long pole: 16.6 us for an 8 MB read-modify-write) — and plain stores when the gradient buffer is known to be zero (`store`)
template <typename T>
__global__ void __launch_bounds__(256) plan_scatter_grad_w4_kernel(const T* __restrict__ dC, const int* __restrict__ idx, int B, int NCAT, int NCLS, int H,
                                                                   int KIN, float* __restrict__ dw, int store) {
    __shared__ float tile[4][32][65];
    __shared__ int sidx[64];
    const int c = blockIdx.x, i0 = blockIdx.y * 64, tid = threadIdx.x, t = tid & 63, w = tid >> 6;
    for (int k = 0; k < 32; ++k) tile[w][k][t] = 0.f;
    for (int b0 = 0; b0 < B; b0 += 64) {
        __syncthreads();
        if (tid < 64) sidx[tid] = b0 + tid < B ? idx[(b0 + tid) * NCAT + c] : 0;
        float v[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) { const int bb = b0 + w * 16 + b; v[b] = (bb < B && i0 + t < H) ? to_f<T>(dC[(long long)bb * H + i0 + t]) : 0.f; }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16; ++b) tile[w][sidx[w * 16 + b]][t] += v[b];      // window order within a wave as in the serial kernel; the four partial sums meet below
    }
    __syncthreads();
    const int cls = tid & 31, rs = tid >> 5;
    float old[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = rs + 8 * k;
        old[k] = (!store && i0 + r < H && cls < NCLS) ? dw[(long long)(i0 + r) * KIN + c * NCLS + cls] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int r = rs + 8 * k;
        if (i0 + r < H && cls < NCLS) dw[(long long)(i0 + r) * KIN + c * NCLS + cls] = old[k] + ((tile[0][cls][r] + tile[1][cls][r]) + (tile[2][cls][r] + tile[3][cls][r]));
    }
}
// out[r][c] = relu(x[r][c])   (act 2: tanh)
template <typename T>
__global__ void relu_copy_kernel(const T* __restrict__ x, T* __restrict__ out, long long n, int act = 1) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = to_f<T>(x[i]); out[i] = from_f<T>(act == 2 ? tanhf(v) : fmaxf(v, 0.f)); }
}
// out = g * (h > 0)   (act 2: g * (1 - h^2))
template <typename T>
__global__ void mask_mul_kernel(const T* __restrict__ g, const T* __restrict__ h, T* __restrict__ out, long long n, int act = 1) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (act == 2) { const float hv = to_f<T>(h[i]); out[i] = from_f<T>(to_f<T>(g[i]) * (1.f - hv * hv)); }
    else out[i] = to_f<T>(h[i]) > 0.f ? g[i] : from_f<T>(0.f);
}
// out[b][c] = sum_t x[t][b][c]   (time-major), fp32 accumulate
template <typename T>
__global__ void sum_over_t_kernel(const T* __restrict__ x, int S, long long BH, T* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BH) return;
    float s = 0.f;
#pragma unroll 8
    for (int t = 0; t < S; ++t) s += to_f<T>(x[(long long)t * BH + i]);
    out[i] = from_f<T>(s);
}
// The goal encoder's LayerNorm (n <= 64 features) and the plan proposal's input rows [emb[:,0,:] | goal] in one launch: blocks [0, ceil(rows / 4))
// normalise (goal rows to `goal` and to the goal columns of `ppx`), the remaining blocks copy the embedding columns.
template <typename T>
__global__ void __launch_bounds__(256) goal_ln_concat_kernel(const float* __restrict__ x, int rows, int n, const float* __restrict__ g, const float* __restrict__ b,
                                                             T* __restrict__ goal, float* __restrict__ stats, const T* __restrict__ emb, long long ld_emb_b, int E,
                                                             T* __restrict__ ppx) {
    const int nln = (rows + 3) >> 2;
    if ((int)blockIdx.x >= nln) {
        const int idx = ((int)blockIdx.x - nln) * 256 + threadIdx.x;
        if (idx < rows * E) { const int r = idx / E, c = idx - r * E; ppx[(long long)r * (E + n) + c] = emb[(long long)r * ld_emb_b + c]; }
        return;
    }
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float v = lane < n ? x[(long long)row * n + lane] : 0.f;
    const float mean = wave_sum(v) / n;
    const float d = lane < n ? v - mean : 0.f;
    const float rstd = rsqrtf(wave_sum(d * d) / n + 1e-5f);
    if (lane < n) {
        const T y = from_f<T>(d * rstd * g[lane] + b[lane]);
        goal[(long long)row * n + lane] = y;
        ppx[(long long)row * (E + n) + E + lane] = y;
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}
// pack [emb[:,0,:] | goal] rows for the plan proposal input
template <typename T>
__global__ void concat_pp_kernel(const T* __restrict__ emb, long long ld_emb_b, int E, const T* __restrict__ goal, int G, int B,
                                 T* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * (E + G)) return;
    const int b = idx / (E + G), c = idx % (E + G);
    out[idx] = c < E ? emb[(long long)b * ld_emb_b + c] : goal[b * G + (c - E)];
}

// =========================================================================================================
// world->tcp action transform (gripper_control.py:16-36) + discretized logistic mixture NLL + gripper CE
// (logistic_decoder_rnn.py:136-152,184-231), forward + gradient w.r.t. the 182 head outputs, one thread per (t,b) row.
// heads row layout (packed): [prob 0..59 | mean 60..119 | log_scale 120..179 | gripper 180..181 | pad], ldh = 192
// =========================================================================================================
DEVI void euler_xyz(float a, float b, float c, float (&R)[9]) {
    float sa, ca, sb, cb, sc, cc;
    sincosf(a, &sa, &ca); sincosf(b, &sb, &cb); sincosf(c, &sc, &cc);
    R[0] = cb * cc;                 R[1] = -cb * sc;                R[2] = sb;
    R[3] = ca * sc + sa * sb * cc;  R[4] = ca * cc - sa * sb * sc;  R[5] = -sa * cb;
    R[6] = sa * sc - ca * sb * cc;  R[7] = sa * cc + ca * sb * sc;  R[8] = ca * cb;
}
DEVI float softplusf(float x) { return x > 20.f ? x : (x < -20.f ? __expf(x) : log1pf(__expf(x))); }
DEVI float sigmoidf(float x) { return 1.f / (1.f + __expf(-x)); }

// =========================================================================================================
// validation / rollout: sample an action from the logistic mixture heads (logistic_decoder_rnn.py:234-258),
// map it from the tcp frame back to the world frame (gripper_control.py:39-63) and, when ground truth is given, accumulate the
// metrics lmp_val reports (hulc.py:347-358): per-dimension mean |error| and the binary gripper success rate.
// One thread per time-major row r = t*B + b.  u_mix [B][S][6][NMIX] / u_act [B][S][6] are the two uniform draws in [0,1)
// (injected for parity, else the counter RNG).  metrics[0..5] += |err_d| / (B*S), metrics[6] += match / (B*S).
// =========================================================================================================
__global__ void logistic_sample_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ robot_obs /*[B][S][15]*/,
                                       const float* __restrict__ actions_gt /*[B][S][7] or null*/, const float* __restrict__ u_mix,
                                       const float* __restrict__ u_act, int B, int S, int NMIX, int NDIM, float log_scale_min, int gripper_control,
                                       unsigned long long seed, float* __restrict__ pred_out /*[B][S][7]*/, float* __restrict__ metrics,
                                       int discrete_gripper = 1) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= B * S) return;
    const int t = r / B, b = r % B;
    const long long bs = (long long)b * S + t;
    const float* hr = heads + (long long)r * ldh;
    const int NO = NMIX * NDIM;
    const float r1 = 1e-5f, r2 = 1.f - 1e-5f;
    float a[7];
    for (int d = 0; d < NDIM; ++d) {
        int ksel = 0;
        float best = -INFINITY;
        for (int k = 0; k < NMIX; ++k) {
            const float u = u_mix ? u_mix[(bs * NDIM + d) * NMIX + k] : hash_uniform(seed, (unsigned long long)((bs * NDIM + d) * NMIX + k));
            const float tt = (r1 - r2) * u + r2;
            const float g = hr[d * NMIX + k] - __logf(-__logf(tt));
            if (g > best) { best = g; ksel = k; }                     // first maximum, like torch.argmax
        }
        const float mu = hr[NO + d * NMIX + ksel];
        const float ls = fmaxf(hr[2 * NO + d * NMIX + ksel], log_scale_min);
        const float uu = u_act ? u_act[bs * NDIM + d] : hash_uniform(seed ^ 0x9e3779b97f4a7c15ull, (unsigned long long)(bs * NDIM + d));
        const float u = (r1 - r2) * uu + r2;
        a[d] = mu + __expf(ls) * (__logf(u) - __logf(1.f - u));
    }
    if (discrete_gripper) a[6] = (hr[3 * NO + 1] > hr[3 * NO]) ? 1.f : -1.f;               // gripper_bounds[argmax]
    float w[7];
    if (gripper_control) {
        const float* ro = robot_obs + bs * 15;
        float R[9], Rr[9];
        euler_xyz(ro[3], ro[4], ro[5], R);
        euler_xyz(a[3] * 0.01f, a[4] * 0.01f, a[5] * 0.01f, Rr);
#pragma unroll
        for (int i = 0; i < 3; ++i) w[i] = R[3 * i] * a[0] + R[3 * i + 1] * a[1] + R[3 * i + 2] * a[2];
        auto Mij = [&](int i, int j) { return R[3 * i] * Rr[3 * j] + R[3 * i + 1] * Rr[3 * j + 1] + R[3 * i + 2] * Rr[3 * j + 2]; };   // R * Rr^T
        float o[3] = {atan2f(-Mij(1, 2), Mij(2, 2)), asinf(fminf(1.f, fmaxf(-1.f, Mij(0, 2)))), atan2f(-Mij(0, 1), Mij(0, 0))};
        const float PI = 3.14159265358979323846f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            o[i] -= ro[3 + i];
            if (o[i] < -PI) o[i] += 2.f * PI;
            if (o[i] > PI) o[i] -= 2.f * PI;
            w[3 + i] = o[i] * 100.f;
        }
        w[6] = a[6];
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) w[i] = a[i];
    }
    if (pred_out)
#pragma unroll
        for (int i = 0; i < 7; ++i) pred_out[bs * 7 + i] = w[i];
    if (metrics && actions_gt) {
        const float inv = 1.f / (float)(B * S);
        const float* gt = actions_gt + bs * 7;
#pragma unroll
        for (int i = 0; i < 6; ++i) atomicAdd(metrics + i, fabsf(w[i] - gt[i]) * inv);
        atomicAdd(metrics + 6, (((w[6] > 0.f) ? 1.f : -1.f) == gt[6]) ? inv : 0.f);
    }
}

// RelativeActions (hulc/utils/transforms.py:32-56): absolute tcp targets -> the relative, clipped and scaled actions the policy is
// trained on.  One thread per (window, step) row; robot_obs is the raw 15-d state (position 0:3, euler orientation 3:6).
__global__ void relative_actions_kernel(const float* __restrict__ actions_abs, const float* __restrict__ robot_obs, int rows, float max_pos,
                                        float max_orn, float* __restrict__ out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* a = actions_abs + (long long)r * 7;
    const float* ro = robot_obs + (long long)r * 15;
    float* o = out + (long long)r * 7;
    const float PI = 3.14159265358979323846f;
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = fminf(fmaxf(a[i] - ro[i], -max_pos), max_pos) / max_pos;
#pragma unroll
    for (int i = 3; i < 6; ++i) {
        float x = (a[i] - ro[i]) + PI;                       // batch_angle_between: (diff + pi) mod 2pi - pi, Python's non-negative modulo
        x -= 2.f * PI * floorf(x / (2.f * PI));
        o[i] = fminf(fmaxf(x - PI, -max_orn), max_orn) / max_orn;
    }
    o[6] = a[6];
}

// one thread per (row, slot): slots 0..NDIM-1 = mixture dimension d, slot NDIM = gripper cross entropy, last slot idle;
// row_loss is [rows][8] (summed deterministically afterwards).  Each thread recomputes the row's tcp-frame action (cheap) so the
// 7 partial losses of a row run in parallel instead of serially in one lane.
template <typename T, int NMIXC>      // NMIXC = n_mixtures at compile time: the per-mixture arrays stay in registers (NMIX must equal it)
// launched with <= 256 threads: without the bound the compiler budgets for 1024 (128 VGPRs) and spilled 46 registers
__global__ void __launch_bounds__(256) logistic_loss_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ actions /*[B][S][7]*/,
                                     const float* __restrict__ robot_obs /*[B][S][15]*/, int B, int S, int NMIX, int NDIM, int num_classes,
                                     float log_scale_min, float gripper_alpha, int gripper_control, float grad_scale,
                                     float* __restrict__ row_loss, float* __restrict__ a_tcp_out, T* __restrict__ dheads,
                                     int discrete_gripper = 1, const float* __restrict__ lscale = nullptr) {
    if (lscale) grad_scale *= lscale[0];       // dynamic loss scale (fp16 mode): d heads only, the loss rows stay unscaled
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = gid >> 3, slot = gid & 7;                  // time-major row: r = t*B + b
    if (r >= B * S) return;
    const int t = r / B, b = r % B;
    const float* act = actions + ((long long)b * S + t) * 7;
    float at[7];
    if (gripper_control) {
        const float* ro = robot_obs + ((long long)b * S + t) * 15;
        float R[9], Rn[9];
        euler_xyz(ro[3], ro[4], ro[5], R);
        euler_xyz(ro[3] + act[3] * 0.01f, ro[4] + act[4] * 0.01f, ro[5] + act[5] * 0.01f, Rn);
#pragma unroll
        for (int i = 0; i < 3; ++i) at[i] = R[0 + i] * act[0] + R[3 + i] * act[1] + R[6 + i] * act[2];
        auto Mij = [&](int i, int j) { return Rn[0 + i] * R[0 + j] + Rn[3 + i] * R[3 + j] + Rn[6 + i] * R[6 + j]; };
        float o[3] = {atan2f(-Mij(1, 2), Mij(2, 2)), asinf(fminf(1.f, fmaxf(-1.f, Mij(0, 2)))), atan2f(-Mij(0, 1), Mij(0, 0))};
        const float PI = 3.14159265358979323846f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (o[i] < -PI) o[i] += 2.f * PI;
            if (o[i] > PI) o[i] -= 2.f * PI;
            at[3 + i] = o[i] * 100.f;
        }
        at[6] = act[6];
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) at[i] = act[i];
    }
    if (a_tcp_out && slot == 7)
#pragma unroll
        for (int i = 0; i < 7; ++i) a_tcp_out[((long long)b * S + t) * 7 + i] = at[i];

    const float* hr = heads + (long long)r * ldh;
    T* dr = dheads + (long long)r * ldh;
    const int NO = NMIX * NDIM;
    float loss = 0.f;
    if (slot < NDIM) {
        const int d = slot;
        const float hb = 1.f / (num_classes - 1);    // (max-min)/2/(nc-1) with bounds +-1 (conf/datamodule/default.yaml)
        const float logc = __logf((num_classes - 1) * 0.5f);
        float a = at[0];
#pragma unroll
        for (int i = 1; i < 7; ++i) a = (d == i) ? at[i] : a;
        float lp[NMIXC], dlogp_dmean[NMIXC], dlogp_dls[NMIXC];
        float mlog = -INFINITY;
        _Pragma("unroll") for (int k = 0; k < NMIXC; ++k) mlog = fmaxf(mlog, hr[d * NMIX + k]);
        float slog = 0.f;
        _Pragma("unroll") for (int k = 0; k < NMIXC; ++k) slog += __expf(hr[d * NMIX + k] - mlog);
        const float lz = mlog + __logf(slog);
        float mx = -INFINITY;
        _Pragma("unroll") for (int k = 0; k < NMIXC; ++k) {
            const float mu = hr[NO + d * NMIX + k];
            const float lsr = hr[2 * NO + d * NMIX + k];
            const float ls = fmaxf(lsr, log_scale_min);
            const float inv = __expf(-ls);
            const float cen = a - mu;
            const float plus = inv * (cen + hb), minus = inv * (cen - hb), mid = inv * cen;
            const float sp = sigmoidf(plus), sm = sigmoidf(minus);
            const float delta = sp - sm;
            float logp, gp = 0.f, gm = 0.f, gmid = 0.f, direct = 0.f;
            if (a < -1.f + 1e-3f) { logp = plus - softplusf(plus); gp = sigmoidf(-plus); }
            else if (a > 1.f - 1e-3f) { logp = -softplusf(minus); gm = -sm; }
            else if (delta > 1e-5f) { logp = __logf(fmaxf(delta, 1e-12f)); gp = sp * (1.f - sp) / delta; gm = -sm * (1.f - sm) / delta; }
            else { logp = mid - ls - 2.f * softplusf(mid) - logc; gmid = 1.f - 2.f * sigmoidf(mid); direct = -1.f; }
            dlogp_dmean[k] = -inv * (gp + gm + gmid);
            dlogp_dls[k] = (lsr >= log_scale_min) ? (-(gp * plus + gm * minus + gmid * mid) + direct) : 0.f;
            lp[k] = logp + (hr[d * NMIX + k] - lz);
            mx = fmaxf(mx, lp[k]);
        }
        float se = 0.f;
        _Pragma("unroll") for (int k = 0; k < NMIXC; ++k) se += __expf(lp[k] - mx);
        const float lse = mx + __logf(se);
        loss = -lse;
        _Pragma("unroll") for (int k = 0; k < NMIXC; ++k) {
            const float w = __expf(lp[k] - lse);
            const float pi = __expf(hr[d * NMIX + k] - lz);
            dr[d * NMIX + k] = from_f<T>(-(w - pi) * grad_scale);
            dr[NO + d * NMIX + k] = from_f<T>(-w * dlogp_dmean[k] * grad_scale);
            dr[2 * NO + d * NMIX + k] = from_f<T>(-w * dlogp_dls[k] * grad_scale);
        }
    } else if (slot == NDIM && discrete_gripper) {
        // gripper cross entropy: label -1 -> 0 else (long)value  (logistic_decoder_rnn.py:144-151)
        const float g0 = hr[3 * NO], g1 = hr[3 * NO + 1];
        const int lab = (at[6] == -1.f) ? 0 : (int)at[6];
        const float m = fmaxf(g0, g1);
        const float lz = m + __logf(__expf(g0 - m) + __expf(g1 - m));
        loss = gripper_alpha * (lz - (lab == 0 ? g0 : g1));
        const float p0 = __expf(g0 - lz), p1 = __expf(g1 - lz);
        dr[3 * NO] = from_f<T>(gripper_alpha * (p0 - (lab == 0 ? 1.f : 0.f)) * grad_scale);
        dr[3 * NO + 1] = from_f<T>(gripper_alpha * (p1 - (lab == 1 ? 1.f : 0.f)) * grad_scale);
    }
    if (slot == 7)
        for (int c = 3 * NO + (discrete_gripper ? 2 : 0); c < ldh; ++c) dr[c] = from_f<T>(0.f);
    row_loss[gid] = loss;
}

// The same loss with one LANE per mixture component (16-bit engines, training step; round 5).  logistic_loss_kernel walks a dimension's 10 components serially in
// one thread: ~2 100 dependent instructions per thread on one wave per SIMD (40 KB of straight-line code that every wave runs once) = 17.9 us for 2 048 tokens.
// Here a token is a 128-thread block: thread (d = tid >> 4, k = tid & 15) owns component k of dimension d, the max / sum-exp reductions are 16-lane butterflies,
// the per-component arithmetic is the serial kernel's line by line (every lane derives the token's tcp-frame action itself: lock-step work costs no time).  Sums over
// the components are associated as a tree instead of left to right: last-bit differences in lz / lse.  Row losses land in the same [token][8] slots.
template <typename T>
__global__ void __launch_bounds__(128) logistic_loss_wide_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ actions, const float* __restrict__ robot_obs,
                                                                 int B, int S, int NMIX, int NDIM, int num_classes, float log_scale_min, float gripper_alpha, int gripper_control,
                                                                 float grad_scale, float* __restrict__ row_loss, float* __restrict__ a_tcp_out, T* __restrict__ dheads,
                                                                 int discrete_gripper, const float* __restrict__ lscale) {
    if (lscale) grad_scale *= lscale[0];
    const int r = blockIdx.x, tid = threadIdx.x, d = tid >> 4, k = tid & 15;      // time-major row r = t*B + b
    const int t = r / B, b = r % B;
    const float* act = actions + ((long long)b * S + t) * 7;
    float at[7];
    if (gripper_control) {
        const float* ro = robot_obs + ((long long)b * S + t) * 15;
        float R[9], Rn[9];
        euler_xyz(ro[3], ro[4], ro[5], R);
        euler_xyz(ro[3] + act[3] * 0.01f, ro[4] + act[4] * 0.01f, ro[5] + act[5] * 0.01f, Rn);
#pragma unroll
        for (int i = 0; i < 3; ++i) at[i] = R[0 + i] * act[0] + R[3 + i] * act[1] + R[6 + i] * act[2];
        auto Mij = [&](int i, int j) { return Rn[0 + i] * R[0 + j] + Rn[3 + i] * R[3 + j] + Rn[6 + i] * R[6 + j]; };
        float o[3] = {atan2f(-Mij(1, 2), Mij(2, 2)), asinf(fminf(1.f, fmaxf(-1.f, Mij(0, 2)))), atan2f(-Mij(0, 1), Mij(0, 0))};
        const float PI = 3.14159265358979323846f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (o[i] < -PI) o[i] += 2.f * PI;
            if (o[i] > PI) o[i] -= 2.f * PI;
            at[3 + i] = o[i] * 100.f;
        }
        at[6] = act[6];
    } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) at[i] = act[i];
    }
    if (a_tcp_out && tid < 7) {
        float v = at[0];
#pragma unroll
        for (int i = 1; i < 7; ++i) v = (tid == i) ? at[i] : v;
        a_tcp_out[((long long)b * S + t) * 7 + tid] = v;
    }
    const float* hr = heads + (long long)r * ldh;
    T* dr = dheads + (long long)r * ldh;
    const int NO = NMIX * NDIM;
    auto gmax = [](float v) { v = fmaxf(v, __shfl_xor(v, 8, 16)); v = fmaxf(v, __shfl_xor(v, 4, 16)); v = fmaxf(v, __shfl_xor(v, 2, 16)); return fmaxf(v, __shfl_xor(v, 1, 16)); };
    auto gsum = [](float v) { v += __shfl_xor(v, 8, 16); v += __shfl_xor(v, 4, 16); v += __shfl_xor(v, 2, 16); return v + __shfl_xor(v, 1, 16); };
    float loss = 0.f;
    if (d < NDIM) {                                          // group-uniform (16 lanes)
        const bool live = k < NMIX;
        const int kk = live ? k : 0;
        const float hb = 1.f / (num_classes - 1);
        const float logc = __logf((num_classes - 1) * 0.5f);
        float a = at[0];
#pragma unroll
        for (int i = 1; i < 7; ++i) a = (d == i) ? at[i] : a;
        const float lg = hr[d * NMIX + kk];
        const float mlog = gmax(live ? lg : -INFINITY);
        const float slog = gsum(live ? __expf(lg - mlog) : 0.f);
        const float lz = mlog + __logf(slog);
        const float mu = hr[NO + d * NMIX + kk];
        const float lsr = hr[2 * NO + d * NMIX + kk];
        const float ls = fmaxf(lsr, log_scale_min);
        const float inv = __expf(-ls);
        const float cen = a - mu;
        const float plus = inv * (cen + hb), minus = inv * (cen - hb), mid = inv * cen;
        const float sp = sigmoidf(plus), sm = sigmoidf(minus);
        const float delta = sp - sm;
        float logp, gp = 0.f, gm = 0.f, gmid = 0.f, direct = 0.f;
        if (a < -1.f + 1e-3f) { logp = plus - softplusf(plus); gp = sigmoidf(-plus); }
        else if (a > 1.f - 1e-3f) { logp = -softplusf(minus); gm = -sm; }
        else if (delta > 1e-5f) { logp = __logf(fmaxf(delta, 1e-12f)); gp = sp * (1.f - sp) / delta; gm = -sm * (1.f - sm) / delta; }
        else { logp = mid - ls - 2.f * softplusf(mid) - logc; gmid = 1.f - 2.f * sigmoidf(mid); direct = -1.f; }
        const float dlogp_dmean = -inv * (gp + gm + gmid);
        const float dlogp_dls = (lsr >= log_scale_min) ? (-(gp * plus + gm * minus + gmid * mid) + direct) : 0.f;
        const float lp = logp + (lg - lz);
        const float mx = gmax(live ? lp : -INFINITY);
        const float se = gsum(live ? __expf(lp - mx) : 0.f);
        const float lse = mx + __logf(se);
        loss = -lse;
        if (live) {
            const float w = __expf(lp - lse);
            const float pi = __expf(lg - lz);
            dr[d * NMIX + k] = from_f<T>(-(w - pi) * grad_scale);
            dr[NO + d * NMIX + k] = from_f<T>(-w * dlogp_dmean * grad_scale);
            dr[2 * NO + d * NMIX + k] = from_f<T>(-w * dlogp_dls * grad_scale);
        }
    } else if (d == NDIM && discrete_gripper) {
        const float g0 = hr[3 * NO], g1 = hr[3 * NO + 1];
        const int lab = (at[6] == -1.f) ? 0 : (int)at[6];
        const float m = fmaxf(g0, g1);
        const float lz = m + __logf(__expf(g0 - m) + __expf(g1 - m));
        loss = gripper_alpha * (lz - (lab == 0 ? g0 : g1));
        if (k == 0) {
            const float p0 = __expf(g0 - lz), p1 = __expf(g1 - lz);
            dr[3 * NO] = from_f<T>(gripper_alpha * (p0 - (lab == 0 ? 1.f : 0.f)) * grad_scale);
            dr[3 * NO + 1] = from_f<T>(gripper_alpha * (p1 - (lab == 1 ? 1.f : 0.f)) * grad_scale);
        }
    }
    if (d == 7)                                              // the zero pad columns of the packed heads row
        for (int c = 3 * NO + (discrete_gripper ? 2 : 0) + k; c < ldh; c += 16) dr[c] = from_f<T>(0.f);
    if (k == 0) row_loss[r * 8 + d] = loss;
}

// scale * sum(x[0..n)) by one block of 256 threads (deterministic tree); the value is returned to thread 0
DEVI float block_sum256(const float* __restrict__ x, int n, float* red) {
    float s = 0.f;
    if ((n & 3) == 0 && (reinterpret_cast<unsigned long long>(x) & 15) == 0) {      // 16-byte loads, 4 independent chains in flight
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const int n4 = n >> 2;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = threadIdx.x;
        for (; i + 768 < n4; i += 1024) {
            const float4 a = x4[i], b = x4[i + 256], c = x4[i + 512], d = x4[i + 768];
            s0 += (a.x + a.y) + (a.z + a.w); s1 += (b.x + b.y) + (b.z + b.w); s2 += (c.x + c.y) + (c.z + c.w); s3 += (d.x + d.y) + (d.z + d.w);
        }
        for (; i < n4; i += 256) { const float4 a = x4[i]; s0 += (a.x + a.y) + (a.z + a.w); }
        s = (s0 + s1) + (s2 + s3);
    } else
        for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    __syncthreads();
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    return red[0];
}
// the end of a one-modality forward in ONE launch (was: two sum_reduce launches, pack_losses, a 16-byte device copy): l[0] = s0 * sum(x0)
// (action loss), l[1] = s1 * sum(x1) (KL; x1 == null keeps l[1]), l[4..7] = [action + kl, kl, action, clip], optionally copied to a DEVICE out[4].
// Same summation order as sum_reduce_kernel.
// Blocks 1 .. gridDim.x-1 (optional) clear the backward's zero arena [zp, zp + nz4) — the memset the backward would otherwise start with.
__global__ void __launch_bounds__(256) finish_losses_kernel(const float* __restrict__ x0, int n0, float s0, const float* __restrict__ x1, int n1, float s1,
                                                            float* __restrict__ l, float* __restrict__ out, float4* __restrict__ zp = nullptr, long long nz4 = 0) {
    if (blockIdx.x > 0) {
        for (long long i = (long long)(blockIdx.x - 1) * 256 + threadIdx.x; i < nz4; i += (long long)(gridDim.x - 1) * 256) zp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    __shared__ float red[256];
    const float a = block_sum256(x0, n0, red) * s0;
    float k = 0.f;
    if (x1) k = block_sum256(x1, n1, red) * s1;
    if (threadIdx.x == 0) {
        if (!x1) k = l[1];
        l[0] = a; l[1] = k;
        const float c = l[2];
        l[4] = a + k; l[5] = k; l[6] = a; l[7] = c;
        if (out) { out[0] = a + k; out[1] = k; out[2] = a; out[3] = c; }
    }
}

// out[0] = scale * sum(x[0..n))   (single block, deterministic tree)
__global__ void __launch_bounds__(256) sum_reduce_kernel(const float* __restrict__ x, int n, float scale, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.f;
    if ((n & 3) == 0 && (reinterpret_cast<unsigned long long>(x) & 15) == 0) {      // 16-byte loads, 4 independent chains in flight
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const int n4 = n >> 2;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int i = threadIdx.x;
        for (; i + 768 < n4; i += 1024) {
            const float4 a = x4[i], b = x4[i + 256], c = x4[i + 512], d = x4[i + 768];
            s0 += (a.x + a.y) + (a.z + a.w); s1 += (b.x + b.y) + (b.z + b.w); s2 += (c.x + c.y) + (c.z + c.w); s3 += (d.x + d.y) + (d.z + d.w);
        }
        for (; i < n4; i += 256) { const float4 a = x4[i]; s0 += (a.x + a.y) + (a.z + a.w); }
        s = (s0 + s1) + (s2 + s3);
    } else
        for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// up to 32 ranges of a float buffer zeroed in ONE launch (engine.h zero_grads: everything but the lazily stored weight gradients); 16-byte stores,
// ranges are 64-element aligned (spec.layout pads every tensor).  blockIdx.y = range
struct MultiZero { float* p[32]; long long n[32]; };
__global__ void __launch_bounds__(256) multi_zero_kernel(MultiZero mz) {
    float4* __restrict__ d = reinterpret_cast<float4*>(mz.p[blockIdx.y]);
    const long long n4 = mz.n[blockIdx.y] >> 2;
    const float4 z = {0.f, 0.f, 0.f, 0.f};
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) d[i] = z;
}

// up to 8 small device-to-device copies in ONE launch (the paired pass joins the two modalities' actions / robot_obs / injected draws: six
// hipMemcpyAsync of a few KB cost ~5 us each on the engine's stream); 4-byte words, blockIdx.y = segment
struct MultiCopy { const unsigned* src[8]; unsigned* dst[8]; int words[8]; };
__global__ void multi_copy_kernel(MultiCopy mc) {
    const int seg = blockIdx.y;
    const unsigned* __restrict__ s = mc.src[seg];
    unsigned* __restrict__ d = mc.dst[seg];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < mc.words[seg]; i += gridDim.x * blockDim.x) d[i] = s[i];
}
// paired pass: the per-row loss partials [S*B][8] (time-major rows t*B + b) summed separately for the windows b < Bv and b >= Bv
// (a thread takes whole rows: one modulo and two 16-byte loads per row — the element-wise version paid a runtime division per float, 18 us)
__global__ void __launch_bounds__(256) sum_rows_pair_kernel(const float* __restrict__ x, int rows, int B, int Bv, float scale, float* __restrict__ out_v,
                                                            float* __restrict__ out_l) {
    __shared__ float red[2][256];
    float sv = 0.f, sl = 0.f;
    for (int r = threadIdx.x; r < rows; r += 256) {
        const float4 a = *reinterpret_cast<const float4*>(x + (long long)r * 8), b = *reinterpret_cast<const float4*>(x + (long long)r * 8 + 4);
        const float v = ((a.x + a.y) + (a.z + a.w)) + ((b.x + b.y) + (b.z + b.w));
        if (r % B < Bv) sv += v; else sl += v;
    }
    red[0][threadIdx.x] = sv; red[1][threadIdx.x] = sl;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { red[0][threadIdx.x] += red[0][threadIdx.x + o]; red[1][threadIdx.x] += red[1][threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out_v[0] = red[0][0] * scale; out_l[0] = red[1][0] * scale; }
}

// =========================================================================================================
// CLIP ground-truth scores (hulc.py:1024-1029): out[i][j] = exp(logit_scale) * <img_i / |img_i|, txt_j / |txt_j|>; grid (ceil(m/64), n)
// =========================================================================================================
__global__ void __launch_bounds__(64) clip_gt_scores_kernel(const float* __restrict__ img, const float* __restrict__ txt, int n, int m, int D,
                                                            const float* __restrict__ logit_scale, float* __restrict__ out) {
    const int i = blockIdx.y, j = blockIdx.x * 64 + threadIdx.x;
    if (i >= n || j >= m) return;
    float a = 0.f, b = 0.f;
    for (int d = 0; d < D; ++d) { a += img[i * D + d] * img[i * D + d]; b += txt[(long long)j * D + d] * txt[(long long)j * D + d]; }
    const float na = sqrtf(a), nb = sqrtf(b), s = __expf(logit_scale[0]);
    float c = 0.f;
    for (int d = 0; d < D; ++d) c += (s * (img[i * D + d] / na)) * (txt[(long long)j * D + d] / nb);
    out[(long long)i * m + j] = c;
}

// =========================================================================================================
// CLIP-style auxiliary loss (hulc.py:679-695) on n <= 64 rows of 32-d projections; single block of 64 threads
// writes loss, d img, d txt (already times `w`), d logit_scale (accumulated)
// =========================================================================================================
__global__ void __launch_bounds__(64) clip_loss_kernel(const float* __restrict__ img, const float* __restrict__ txt, int n, int D,
                                                       const float* __restrict__ logit_scale, float w, float* __restrict__ loss_out,
                                                       float* __restrict__ dimg, float* __restrict__ dtxt, float* __restrict__ dlogit_scale,
                                                       const float* __restrict__ lscale = nullptr) {
    if (lscale) w *= lscale[0];
    __shared__ float in_[64][33], tn_[64][33], ni[64], nt[64], L[64][65], dL[64][65], rowlse[64], collse[64], red[64];
    const int i = threadIdx.x;
    const float s = __expf(logit_scale[0]);
    if (i < n) {
        float a = 0.f, b = 0.f;
        for (int d = 0; d < D; ++d) { a += img[i * D + d] * img[i * D + d]; b += txt[i * D + d] * txt[i * D + d]; }
        ni[i] = sqrtf(a); nt[i] = sqrtf(b);
        for (int d = 0; d < D; ++d) { in_[i][d] = img[i * D + d] / ni[i]; tn_[i][d] = txt[i * D + d] / nt[i]; }
    }
    __syncthreads();
    if (i < n) {
        float m = -INFINITY;
        for (int j = 0; j < n; ++j) {
            float c = 0.f;
            for (int d = 0; d < D; ++d) c += in_[i][d] * tn_[j][d];
            L[i][j] = s * c;
            m = fmaxf(m, L[i][j]);
        }
        float se = 0.f;
        for (int j = 0; j < n; ++j) se += __expf(L[i][j] - m);
        rowlse[i] = m + __logf(se);
    }
    __syncthreads();
    if (i < n) {
        float m = -INFINITY;
        for (int j = 0; j < n; ++j) m = fmaxf(m, L[j][i]);
        float se = 0.f;
        for (int j = 0; j < n; ++j) se += __expf(L[j][i] - m);
        collse[i] = m + __logf(se);
    }
    __syncthreads();
    float part = 0.f, dsp = 0.f;
    if (i < n) {
        part = (rowlse[i] - L[i][i]) + (collse[i] - L[i][i]);
        for (int j = 0; j < n; ++j) {
            const float g = ((__expf(L[i][j] - rowlse[i]) - (i == j)) + (__expf(L[i][j] - collse[j]) - (i == j))) / (2.f * n);
            dL[i][j] = g;
            dsp += g * (L[i][j] / s);
        }
    }
    red[i] = part;
    __syncthreads();
    if (i == 0) {
        float t = 0.f;
        for (int j = 0; j < n; ++j) t += red[j];
        loss_out[0] = t / (2.f * n);
    }
    __syncthreads();
    red[i] = dsp;
    __syncthreads();
    if (i == 0) {
        float t = 0.f;
        for (int j = 0; j < n; ++j) t += red[j];
        dlogit_scale[0] += w * t * s;
    }
    if (i < n) {
        float din[32], dtn[32];
        for (int d = 0; d < D; ++d) { din[d] = 0.f; dtn[d] = 0.f; }
        for (int j = 0; j < n; ++j) {
            const float gij = dL[i][j], gji = dL[j][i];
            for (int d = 0; d < D; ++d) { din[d] += s * gij * tn_[j][d]; dtn[d] += s * gji * in_[j][d]; }
        }
        float di = 0.f, dt = 0.f;
        for (int d = 0; d < D; ++d) { di += in_[i][d] * din[d]; dt += tn_[i][d] * dtn[d]; }
        for (int d = 0; d < D; ++d) {
            dimg[i * D + d] = w * (din[d] - in_[i][d] * di) / ni[i];
            dtxt[i * D + d] = w * (dtn[d] - tn_[i][d] * dt) / nt[i];
        }
    }
}
// The same with 1024 threads (one workgroup): the 64-thread kernel walks n x n x D products per thread in series (83 us at n = 32 rows); here a
// thread owns one (i, j) logit, then one (row, d) gradient element — every loop is n or D long.  Deterministic (fixed-order tree reductions).
__global__ void __launch_bounds__(1024) clip_loss_wide_kernel(const float* __restrict__ img, const float* __restrict__ txt, int n, int D,
                                                              const float* __restrict__ logit_scale, float w, float* __restrict__ loss_out,
                                                              float* __restrict__ dimg, float* __restrict__ dtxt, float* __restrict__ dlogit_scale,
                                                              const float* __restrict__ lscale = nullptr) {
    if (lscale) w *= lscale[0];
    __shared__ float in_[64][33], tn_[64][33], ni[64], nt[64], L[64][65], dL[64][65], rowlse[64], collse[64], red[2][16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const float s = __expf(logit_scale[0]);
    // ---- norms: thread (row r, half h): h = 0 img, 1 txt
    if (t < 2 * n) {
        const int r = t >> 1, h = t & 1;
        const float* src = (h ? txt : img) + (long long)r * D;
        float a = 0.f;
        for (int d = 0; d < D; ++d) a += src[d] * src[d];
        a = sqrtf(a);
        (h ? nt : ni)[r] = a;
        for (int d = 0; d < D; ++d) (h ? tn_ : in_)[r][d] = src[d] / a;
    }
    __syncthreads();
    for (int p = t; p < n * n; p += 1024) {
        const int i = p / n, j = p - i * n;
        float c = 0.f;
        for (int d = 0; d < D; ++d) c += in_[i][d] * tn_[j][d];
        L[i][j] = s * c;
    }
    __syncthreads();
    if (t < 2 * n) {                                  // t < n: row i = t; else column i = t - n
        const bool col = t >= n;
        const int i = col ? t - n : t;
        float m = -INFINITY;
        for (int j = 0; j < n; ++j) m = fmaxf(m, col ? L[j][i] : L[i][j]);
        float se = 0.f;
        for (int j = 0; j < n; ++j) se += __expf((col ? L[j][i] : L[i][j]) - m);
        (col ? collse : rowlse)[i] = m + __logf(se);
    }
    __syncthreads();
    float part = 0.f, dsp = 0.f;
    for (int p = t; p < n * n; p += 1024) {
        const int i = p / n, j = p - i * n;
        const float g = ((__expf(L[i][j] - rowlse[i]) - (i == j)) + (__expf(L[i][j] - collse[j]) - (i == j))) / (2.f * n);
        dL[i][j] = g;
        dsp += g * (L[i][j] / s);
        if (i == j) part += (rowlse[i] - L[i][i]) + (collse[i] - L[i][i]);
    }
    part = wave_sum(part); dsp = wave_sum(dsp);
    if (lane == 0) { red[0][wave] = part; red[1][wave] = dsp; }
    __syncthreads();
    if (t == 0) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < 16; ++k) { a += red[0][k]; b += red[1][k]; }
        loss_out[0] = a / (2.f * n);
        dlogit_scale[0] += w * b * s;
    }
    // ---- gradients: thread (tensor h, row i, column d); D == 32 -> the 32 d's of a row are one half-wave
    for (int e = t; e < 2 * n * 32; e += 1024) {
        const int h = e / (n * 32), r = (e >> 5) % n, d = e & 31;
        float acc = 0.f;
        if (d < D) {
            if (h == 0) { for (int j = 0; j < n; ++j) acc += dL[r][j] * tn_[j][d]; }
            else        { for (int j = 0; j < n; ++j) acc += dL[j][r] * in_[j][d]; }
        }
        acc *= s;
        const float own = d < D ? (h ? tn_[r][d] : in_[r][d]) : 0.f;
        float dotp = own * acc;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) dotp += __shfl_xor(dotp, o, 32);
        if (d < D) (h ? dtxt : dimg)[(long long)r * D + d] = w * (acc - own * dotp) / (h ? nt[r] : ni[r]);
    }
}
// gather rows by index list: dst[i][:] = src[rows[i]][:]  ; scatter-add reverse
template <typename TS, typename TD>
__global__ void gather_rows_kernel(const TS* __restrict__ src, long long lds_, const int* __restrict__ rows, int n, int C, TD* __restrict__ dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * C) return;
    const int i = idx / C, c = idx % C;
    dst[idx] = from_f<TD>(to_f<TS>(src[(long long)rows[i] * lds_ + c]));
}
__global__ void scatter_rows_add_kernel(const float* __restrict__ src, const int* __restrict__ rows, int n, int C, float* __restrict__ dst,
                                        long long ldd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * C) return;
    const int i = idx / C, c = idx % C;
    dst[(long long)rows[i] * ldd + c] += src[idx];
}

// =========================================================================================================
// Adam (torch.optim.Adam defaults; conf/model/optimizer/adam.yaml) over the flat parameter buffer,
// fused with the gradient scale (1/world for the DP mean) — one pass over p, g, m, v.
// =========================================================================================================
// =========================================================================================================
// Dynamic loss scaling (fp16 mode) — torch.cuda.amp.GradScaler semantics, the scaler Lightning's native-AMP plugin drives at
// `precision: 16` (conf/trainer/play_trainer.yaml:3): the loss gradient is multiplied by `scale` at its sources (logistic / KL /
// CLIP loss kernels read scale from here), the optimizer step divides it out again and is SKIPPED when any gradient is non-finite;
// scale *= backoff on a skipped step, *= growth after `interval` consecutive good steps (torch/amp/grad_scaler.py, _amp_update_scale_).
// The state lives on the device so that no step waits for the host.
// =========================================================================================================
struct ScalerState {
    float scale;            // read by the loss kernels as lscale[0] (first member)
    float growth, backoff;
    int interval;
    int growth_tracker;
    int found_inf;          // set by nonfinite_check_kernel of the current step
    int last_found_inf;     // of the most recent finished step
    int skipped;            // optimizer steps skipped so far
    int steps;              // optimizer steps taken (not skipped)
};
// found_inf |= any(!isfinite(g))   (grid-stride, 16-byte loads; one atomic per wave that saw one)
__global__ void __launch_bounds__(256) nonfinite_check_kernel(const float* __restrict__ g, long long n, ScalerState* __restrict__ ss) {
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    unsigned bad = 0;
    for (; i + 3 < n; i += stride) {
        const uint4 u = *reinterpret_cast<const uint4*>(g + i);
        // non-finite <=> exponent field all ones
        bad |= ((u.x & 0x7f800000u) == 0x7f800000u) | ((u.y & 0x7f800000u) == 0x7f800000u) | ((u.z & 0x7f800000u) == 0x7f800000u) | ((u.w & 0x7f800000u) == 0x7f800000u);
    }
    if (i < n)
        for (long long j = i; j < n && j < i + 4; ++j) bad |= ((__float_as_uint(g[j]) & 0x7f800000u) == 0x7f800000u);
    if (__any(bad != 0) && (threadIdx.x & 63) == 0) atomicOr(&ss->found_inf, 1);
}
// data-parallel skip vote (engine.h skip_vote_put / skip_vote_get): one alignment-padding element of the gradient buffer carries "my recurrence
// of this step failed" through the gradients' own SUM all-reduce
__global__ void clamp_index_kernel(int* __restrict__ idx, int n, int ncls) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) idx[i] = min(max(idx[i], 0), ncls - 1);
}
__global__ void dp_skip_put_kernel(const unsigned* __restrict__ skip, unsigned tag, float* __restrict__ pad) { *pad = (*skip == tag) ? 1.f : 0.f; }
__global__ void dp_skip_get_kernel(float* __restrict__ pad, unsigned* __restrict__ skip, unsigned tag) {
    if (!(*pad == 0.f)) *skip = tag;      // any rank voted (NaN included)
    *pad = 0.f;
}
// skip / tag: a persistent recurrence of this step failed (rnn_persist.h) — the optimizer left the weights alone, the scaler state stays as it is
__global__ void scaler_update_kernel(ScalerState* __restrict__ ss, const unsigned* __restrict__ skip = nullptr, unsigned tag = 0) {
    if (skip && *skip == tag) { ss->found_inf = 0; return; }
    if (ss->found_inf) { ss->scale *= ss->backoff; ss->growth_tracker = 0; ss->skipped++; }
    else {
        ss->steps++;
        const int ok = ss->growth_tracker + 1;
        if (ok == ss->interval) {
            const float ns = ss->scale * ss->growth;
            if (isfinite(ns)) ss->scale = ns;
            ss->growth_tracker = 0;
        } else ss->growth_tracker = ok;
    }
    ss->last_found_inf = ss->found_inf;
    ss->found_inf = 0;
}

// wd != 0: torch.optim.Adam's L2 term (g += wd p; decoupled = 0) or torch.optim.AdamW's decoupled decay (p *= 1 - lr wd before the update)
struct AdamArgs { float* p; const float* g; float* m; float* v; float lr, b1, b2, eps, bc1, bc2_sqrt, gscale, wd; int decoupled; h16_t* shadow; const ScalerState* ss; const unsigned* skip; unsigned tag; };
// one element quad of the Adam / AdamW step.  Floating-point contraction is OFF in here: the flat and the tiled kernel must produce the same bits, and whether
// the compiler fuses a multiply into a following add is otherwise its choice per call site (measured: 342 k of 46 M parameters differed by one ulp after 3 steps)
struct AdamQuad { float4 pp, gg, mm, vv; };
DEVI void adam_load4(const AdamArgs& a, long long i, AdamQuad& q) {
    typedef float f32x4nt __attribute__((ext_vector_type(4)));
    q.pp = *reinterpret_cast<const float4*>(a.p + i);
    { const f32x4nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(a.g + i)); q.gg = make_float4(t[0], t[1], t[2], t[3]); }
    { const f32x4nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(a.m + i)); q.mm = make_float4(t[0], t[1], t[2], t[3]); }
    { const f32x4nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4nt*>(a.v + i)); q.vv = make_float4(t[0], t[1], t[2], t[3]); }
}
DEVI void adam_finish4(const AdamArgs& a, float bc1, float bc2_sqrt, float gscale, long long i, AdamQuad& q, unsigned& lo, unsigned& hi) {
#pragma clang fp contract(off)
    typedef float f32x4nt __attribute__((ext_vector_type(4)));
    float4 &pp = q.pp, &gg = q.gg, &mm = q.mm, &vv = q.vv;
    float* P = &pp.x; float* G = &gg.x; float* Mv = &mm.x; float* V = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float gr = G[e] * gscale;
        if (a.wd != 0.f) {
            if (a.decoupled) P[e] *= 1.f - a.lr * a.wd;
            else gr += a.wd * P[e];
        }
        Mv[e] = a.b1 * Mv[e] + (1.f - a.b1) * gr;
        V[e] = a.b2 * V[e] + (1.f - a.b2) * gr * gr;
        P[e] -= (a.lr / bc1) * Mv[e] / (sqrtf(V[e]) / bc2_sqrt + a.eps);
    }
    *reinterpret_cast<float4*>(a.p + i) = pp;
    lo = pack2h(pp.x, pp.y); hi = pack2h(pp.z, pp.w);
    if (a.shadow) *reinterpret_cast<uint2*>(a.shadow + i) = uint2{lo, hi};
    __builtin_nontemporal_store(f32x4nt{mm.x, mm.y, mm.z, mm.w}, reinterpret_cast<f32x4nt*>(a.m + i));
    __builtin_nontemporal_store(f32x4nt{vv.x, vv.y, vv.z, vv.w}, reinterpret_cast<f32x4nt*>(a.v + i));
}
DEVI void adam_update4(const AdamArgs& a, float bc1, float bc2_sqrt, float gscale, long long i, unsigned& lo, unsigned& hi) {
    AdamQuad q;
    adam_load4(a, i, q);
    adam_finish4(a, bc1, bc2_sqrt, gscale, i, q, lo, hi);
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                            float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale, h16_t* __restrict__ shadow,
                            const ScalerState* __restrict__ ss = nullptr, float wd = 0.f, int decoupled = 0, const unsigned* __restrict__ skip = nullptr,
                            unsigned tag = 0) {
    if (skip && *skip == tag) return;   // a persistent recurrence of this step timed out (rnn_persist.h): its gradients are garbage, the step is dropped
    if (ss) {                           // fp16 mode: unscale; a step with non-finite gradients changes nothing (GradScaler.step skips
        if (ss->found_inf) return;      // optimizer.step(), so Adam's own step count — the bias corrections — only counts the steps taken)
        gscale /= ss->scale;
        const float t = (float)(ss->steps + 1);
        bc1 = 1.f - powf(b1, t);
        bc2_sqrt = sqrtf(1.f - powf(b2, t));
    }
    // gradient and moments are touched once per step: non-temporal loads / stores keep them from displacing the parameters and their
    // 16-bit shadow (read next by the weight repacks and the forward) in the L2 / memory-side cache (same-box A/B: -0.014 ms/step)
    const AdamArgs a{p, g, m, v, lr, b1, b2, eps, bc1, bc2_sqrt, gscale, wd, decoupled, shadow, ss, skip, tag};
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) { unsigned lo, hi; adam_update4(a, bc1, bc2_sqrt, gscale, i, lo, hi); }
}

// Adam with the TRANSPOSED 16-bit weight copies written by the same pass (16-bit engines, round 5).  The backward's data-gradient GEMMs read W^T; until round 4 a
// batched transpose re-read the fresh 16-bit shadow (94 MB) after every optimizer step and wrote 94 MB (37 us).  Here the Linear weights that have a transposed copy
// are updated tile-wise: block = one 64 x 64 tile of one matrix (the batched transpose's own block -> tile map: TrDesc, blk2desc): p / g / m / v are read and written
// as 16 rows x 256 contiguous bytes per pass (same per-element arithmetic and the same non-temporal hints as adam_kernel), the 16-bit values go to the shadow AND
// into an LDS tile whose read-out is the transpose's (transpose_tile64_store).  Everything else — biases, LayerNorm and conv weights, matrices without a transposed
// copy — is covered by 4096-element CHUNK blocks behind the tile blocks (chunk table built at bind time: the complement of the tiled matrices in the flat buffer).
constexpr int ADAM_CHUNK = 4096;
__global__ void __launch_bounds__(256) adam_tiled_kernel(AdamArgs a, const TrDesc* __restrict__ desc, const unsigned short* __restrict__ blk2desc, int ntile,
                                                         const long long* __restrict__ chunk_start, const int* __restrict__ chunk_n) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[64][66];
    if (a.skip && *a.skip == a.tag) return;
    float gscale = a.gscale, bc1 = a.bc1, bc2_sqrt = a.bc2_sqrt;
    if (a.ss) {
        if (a.ss->found_inf) return;
        gscale /= a.ss->scale;
        const float t = (float)(a.ss->steps + 1);
        bc1 = 1.f - powf(a.b1, t);
        bc2_sqrt = sqrtf(1.f - powf(a.b2, t));
    }
    const int tid = threadIdx.x;
    if ((int)blockIdx.x < ntile) {
        const TrDesc D = desc[blk2desc[blockIdx.x]];
        const int b = blockIdx.x - D.blk0;
        const int c0 = (b % D.tiles_x) * 64, r0 = (b / D.tiles_x) * 64;
        const long long off = reinterpret_cast<const h16_t*>(D.src) - a.shadow;       // the matrix's offset in the flat parameter buffer
        // all four passes' 16 loads are requested before the first update (64 registers): the tile's 64 KB of p / g / m / v are in flight together
        const int cc = (tid & 15) * 4, c = c0 + cc;
        AdamQuad q[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int r = r0 + ps * 16 + (tid >> 4);
            if (r < D.R && c < D.C) adam_load4(a, off + (long long)r * D.C + c, q[ps]);       // C % 4 == 0 (checked at bind)
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = ps * 16 + (tid >> 4), r = r0 + row;
            unsigned lo = 0u, hi = 0u;
            if (r < D.R && c < D.C) adam_finish4(a, bc1, bc2_sqrt, gscale, off + (long long)r * D.C + c, q[ps], lo, hi);
            *reinterpret_cast<unsigned*>(&tile[row][cc]) = lo;
            *reinterpret_cast<unsigned*>(&tile[row][cc + 2]) = hi;
        }
        __syncthreads();
        if (D.fr) {      // fragment-ordered copy of W: chunk (row, 8 k) -> row tile row / 16, k-step k / 32, lane (k % 32) / 8 * 16 + row % 16 (two chunks per thread)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int q = tid + u * 256, row = q >> 3, c8 = q & 7;
                const int r = r0 + row, k = c0 + c8 * 8;
                if (r < D.R && k + 7 < D.C) {
                    uint4 v;
                    v.x = *reinterpret_cast<const unsigned*>(&tile[row][c8 * 8]); v.y = *reinterpret_cast<const unsigned*>(&tile[row][c8 * 8 + 2]);
                    v.z = *reinterpret_cast<const unsigned*>(&tile[row][c8 * 8 + 4]); v.w = *reinterpret_cast<const unsigned*>(&tile[row][c8 * 8 + 6]);
                    *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(D.fr) + ((((long long)(r >> 4) * (D.C >> 5) + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (r & 15)) << 3)) = v;
                }
            }
        }
        transpose_tile64_store(D, b, tile);
    } else {
        const int j = blockIdx.x - ntile;
        const long long start = chunk_start[j];
        const int n = chunk_n[j];
        for (int i = tid * 4; i + 3 < n; i += 1024) { unsigned lo, hi; adam_update4(a, bc1, bc2_sqrt, gscale, start + i, lo, hi); }
    }
}

// torch.optim.SGD (conf/model/optimizer/sgd.yaml: momentum 0.9) over the flat buffer; buf = the bound first-moment buffer
__global__ void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, long long n, float lr, float momentum,
                           float dampening, float wd, int nesterov, int first, float gscale, h16_t* __restrict__ shadow,
                           const ScalerState* __restrict__ ss = nullptr, const unsigned* __restrict__ skip = nullptr, unsigned tag = 0) {
    if (skip && *skip == tag) return;
    if (ss) {
        if (ss->found_inf) return;
        gscale /= ss->scale;
        first = ss->steps == 0;
    }
    long long i = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long stride = (long long)gridDim.x * blockDim.x * 4;
    for (; i + 3 < n; i += stride) {
        float4 pp = *reinterpret_cast<float4*>(p + i), gg = *reinterpret_cast<const float4*>(g + i), bb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (momentum != 0.f && !first) bb = *reinterpret_cast<const float4*>(buf + i);
        float* P = &pp.x; float* G = &gg.x; float* Bf = &bb.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float gr = G[e] * gscale;
            if (wd != 0.f) gr += wd * P[e];
            if (momentum != 0.f) {
                Bf[e] = first ? gr : momentum * Bf[e] + (1.f - dampening) * gr;
                gr = nesterov ? gr + momentum * Bf[e] : Bf[e];
            }
            P[e] -= lr * gr;
        }
        *reinterpret_cast<float4*>(p + i) = pp;
        if (momentum != 0.f) *reinterpret_cast<float4*>(buf + i) = bb;
        if (shadow) {
            uint2 o;
            o.x = pack2h(pp.x, pp.y);
            o.y = pack2h(pp.z, pp.w);
            *reinterpret_cast<uint2*>(shadow + i) = o;
        }
    }
}

// embg[(t*B+b)*W + c] = emb[(b*S+t)*128 + (128-W) + c]   (time-major copy of the last W columns: W = 64 is the gripper half,
// perceptual_emb_slice [64,128]; W = 128 the whole embedding, mcil)
template <typename T>
__global__ void gather_embg_kernel(const T* __restrict__ emb, T* __restrict__ out, int B, int S, int W = 64) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= S * B * W) return;
    const int c = idx % W, r = idx / W;
    const int t = r / B, b = r % B;
    out[idx] = emb[((long long)b * S + t) * 128 + (128 - W) + c];
}
// losses[4..7] = [action + kl, kl, action, clip] (+ a copy to a DEVICE out[4])
__global__ void pack_losses_kernel(float* __restrict__ l, float* __restrict__ out = nullptr) {
    l[4] = l[0] + l[1]; l[5] = l[1]; l[6] = l[0]; l[7] = l[2];
    if (out) { out[0] = l[4]; out[1] = l[5]; out[2] = l[6]; out[3] = l[7]; }
}

}  // namespace HULC_NS
