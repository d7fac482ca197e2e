// The dense (token-wise) part of the transformer blocks as gfx950 kernels: nn.Linear (+ bias, + GELU, + residual) and nn.LayerNorm.
//
// Reference ops: SABlock's qkv / out_proj and MLPBlock's linear1 -> GELU -> linear2 (monai/networks/blocks/selfattention.py:105-218,
// mlp.py:56-80), TransformerBlock's norm1 / norm2 and the residual sums (transformerblock.py:24-105), PatchEmbeddingBlock's conv
// projection seen as a linear map of the flattened patches (patchembedding.py:32-142), SwinUNETR's WindowAttention.qkv / proj and
// Mlp (nets/swin_unetr.py:426-532, mlp.py).
//
// linear_h2_kernel: Y[M, N] = act(X[M, K] . W[N, K]^T + bias) (+ R), fp32 in and out, evaluated on the fp16 matrix cores in the
// two-piece split precision of conv3d_h2.h (x = hi + lo fp16 pieces, products hi*hi + lo*hi + hi*lo, fp32 accumulate: fp32-equivalent;
// weights pre-split and pre-scaled by a power of two once per layer, activations split while they are staged).  GEMM tile of a
// workgroup (4 waves): 128 rows x 64 columns, K in chunks of 16 (one MFMA k-step); wave w owns rows 32 w .. +31 x both 32-column
// blocks.  A tile [piece][k-group][128 rows][8 ch] (8 KB) and B tile [piece][k-group][64 cols][8 ch] (4 KB, one contiguous 4 KB copy
// of the packed weights) are double-buffered in LDS (24 KB: several workgroups per CU hide each other's staging).  D layout: lane =
// column, registers = rows, so every store instruction writes two 128-byte row segments.
#pragma once
#include "common.h"
#include "conv3d_h2.h"

namespace mh {

constexpr int DN_BM = 128, DN_BN = 64, DN_BK = 16;
constexpr int DN_AV = 2 * DN_BM;                          // uint4 per piece of an A tile: [k-group][row]
constexpr int DN_BV = 2 * DN_BN;                          // uint4 per piece of a B tile: [k-group][column]
constexpr int DN_BT = 2 * DN_BV;                          // uint4 per packed (column tile, chunk) slab: 256 = one per thread

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// ACT: 0 none, 1 GELU (erf form, nn.GELU()).  RES: add r[m][n] after the activation.
// rowmap (or null; round 5): the result of input row m is written to (and its residual read from) row rowmap[m], rows with rowmap[m] < 0 are dropped -- SwinUNETR's
// proj -> window_reverse -> roll back -> crop -> shortcut + x (swin_unetr.py:650-672) inside the projection's epilogue, with the map of layernorm_vec_kernel.
template <int ACT, bool RES>
__global__ void __launch_bounds__(256)
linear_h2_kernel(const float* __restrict__ x, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias,
                 const float* __restrict__ r, float* __restrict__ y, int M, int N, int K, int ntn, const int* __restrict__ rowmap) {
    __shared__ uint4 as[2][2 * DN_AV];
    __shared__ uint4 bs[2][DN_BT];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = (int)(lid % (unsigned)ntn), tm = (int)(lid / (unsigned)ntn);       // column tiles of one row tile are neighbours: X stays in L2
    const int m0 = tm * DN_BM, n0 = tn * DN_BN;
    const int nkc = (K + DN_BK - 1) / DN_BK;

    // A staging: thread (row = tid >> 1, k-group = tid & 1) converts 8 consecutive k of its row
    const int srow = tid >> 1, skg = tid & 1;
    const bool rok = m0 + srow < M;
    const float* xrow = x + (long long)(rok ? m0 + srow : 0) * K + 8 * skg;
    const uint4* wsl = wp + (long long)tn * nkc * DN_BT + tid;
    f32x4 xa, xb;
    uint4 wv;
#define MH_DN_ISSUE(C)                                                                                \
    {                                                                                                 \
        const int k_ = (C) * DN_BK + 8 * skg;                                                         \
        xa = (rok && k_ + 4 <= K) ? *reinterpret_cast<const f32x4*>(xrow + (C) * DN_BK) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};     \
        xb = (rok && k_ + 8 <= K) ? *reinterpret_cast<const f32x4*>(xrow + (C) * DN_BK + 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f}; \
        wv = wsl[(long long)(C) * DN_BT];                                                             \
    }
#define MH_DN_COMMIT(BUF)                                                                             \
    {                                                                                                 \
        _Float16 h_[8], l_[8];                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { h2_split(xa[i], h_[i], l_[i]); h2_split(xb[i], h_[4 + i], l_[4 + i]); } \
        f16x8 hv_, lv_;                                                                               \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) { hv_[i] = h_[i]; lv_[i] = l_[i]; }             \
        as[BUF][skg * DN_BM + srow] = __builtin_bit_cast(uint4, hv_);                                 \
        as[BUF][DN_AV + skg * DN_BM + srow] = __builtin_bit_cast(uint4, lv_);                         \
        bs[BUF][tid] = wv;                                                                            \
    }

    const int r32 = lane & 31, kg = lane >> 5;
    const int abase = kg * DN_BM + 32 * wave + r32;
    const int bbase = kg * DN_BN + r32;
    f32x16 acc[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nb][i] = 0.0f;

    MH_DN_ISSUE(0)
    MH_DN_COMMIT(0)
    __syncthreads();
    for (int c = 0; c < nkc; ++c) {
        const int buf = c & 1;
        if (c + 1 < nkc) MH_DN_ISSUE(c + 1)
        const uint4 ah = as[buf][abase], al = as[buf][DN_AV + abase];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const uint4 bh = bs[buf][bbase + 32 * nb], bl = bs[buf][DN_BV + bbase + 32 * nb];
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bh), acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, bh), acc[nb], 0, 0, 0);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, bl), acc[nb], 0, 0, 0);
        }
        if (c + 1 < nkc) MH_DN_COMMIT(buf ^ 1)
        __syncthreads();
    }
#undef MH_DN_COMMIT
#undef MH_DN_ISSUE

    const float inv_scale = wtail[0];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = n0 + 32 * nb + r32;
        const float bn = (bias && n < N) ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int mi = m0 + 32 * wave + 8 * (i >> 2) + 4 * kg + (i & 3);
            const int m = (rowmap && mi < M) ? rowmap[mi] : mi;          // scatter form: output (and residual) row of input row mi, < 0 = dropped
            if (mi < M && m >= 0 && n < N) {
                float v = fmaf(acc[nb][i], inv_scale, bn);
                if (ACT == 1) v = gelu_erf(v);
                if (RES) v += r[(long long)m * N + n];
                y[(long long)m * N + n] = v;
            }
        }
    }
}

// The same map with a larger workgroup tile for the large token counts of the ViT blocks (round 3).  At 128 x 64 the kernel is bound by what it pulls out of
// L2, not by the matrix pipe: every 64-column tile re-reads its 128 rows of X, every row tile the whole weight matrix -- 12 KB per chunk and 24 matrix
// instructions, ~7.6 TB/s of L2 -> CU traffic for the qkv map of UNETR (M = 13 824 tokens) in 64-byte pieces.  Here a workgroup of 8 waves owns
// (128 MT) x 128: wave w = rows 32 MT (w >> 1) .. x columns 64 (w & 1) .., MT x 2 accumulator tiles; per chunk 16 KB (MT = 1) feed 48 matrix instructions:
// 2/3 of the bytes per flop.  Same packed weights (two neighbouring 64-column slabs), same staging, same arithmetic and summation order per output: bit-identical
// results.  Measured at M = 13 824 (profiles/r03_linear_bench.json): MT = 1 is 1.2-1.5 x the 128 x 64 kernel (qkv 0.288 -> 0.197 ms); MT = 2 (256 x 128, half the
// bytes per flop but 118 registers and 48 KB of LDS: two workgroups per CU) is slower than MT = 1 (0.219 ms) and is not instantiated.
template <int ACT, bool RES, int MT>
__global__ void __launch_bounds__(512)
linear_h2_big_kernel(const float* __restrict__ x, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias,
                     const float* __restrict__ r, float* __restrict__ y, int M, int N, int K, int ntn, int nslab, const int* __restrict__ rowmap) {
    constexpr int BM = DN_BM * MT, BN = 2 * DN_BN;
    constexpr int AV = 2 * BM;                               // uint4 per piece of an A tile: [k-group][row]
    __shared__ uint4 as[2][2 * AV];
    __shared__ uint4 bs[2][2 * DN_BT];                       // two 64-column slabs
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = (int)(lid % (unsigned)ntn), tm = (int)(lid / (unsigned)ntn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int nkc = (K + DN_BK - 1) / DN_BK;

    // A staging: thread (row = tid >> 1, k-group = tid & 1) converts 8 consecutive k of its row (MT = 1: the first 256 threads); B: one uint4 of the two slabs per thread
    const int srow = tid >> 1, skg = tid & 1;
    const bool stage_a = srow < BM;
    const bool rok = stage_a && m0 + srow < M;
    const float* xrow = x + (long long)(rok ? m0 + srow : 0) * K + 8 * skg;
    const int slab = 2 * tn + (tid >> 8);
    const bool sok = slab < nslab;
    const uint4* wsl = wp + (long long)(sok ? slab : 0) * nkc * DN_BT + (tid & 255);
    f32x4 xa, xb;
    uint4 wv;
#define MH_DB_ISSUE(C)                                                                                \
    {                                                                                                 \
        const int k_ = (C) * DN_BK + 8 * skg;                                                         \
        xa = (rok && k_ + 4 <= K) ? *reinterpret_cast<const f32x4*>(xrow + (C) * DN_BK) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};     \
        xb = (rok && k_ + 8 <= K) ? *reinterpret_cast<const f32x4*>(xrow + (C) * DN_BK + 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f}; \
        wv = sok ? wsl[(long long)(C) * DN_BT] : make_uint4(0u, 0u, 0u, 0u);                          \
    }
#define MH_DB_COMMIT(BUF)                                                                             \
    {                                                                                                 \
        if (stage_a) {                                                                                \
            _Float16 h_[8], l_[8];                                                                    \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) { h2_split(xa[i], h_[i], l_[i]); h2_split(xb[i], h_[4 + i], l_[4 + i]); } \
            f16x8 hv_, lv_;                                                                           \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) { hv_[i] = h_[i]; lv_[i] = l_[i]; }         \
            as[BUF][skg * BM + srow] = __builtin_bit_cast(uint4, hv_);                                \
            as[BUF][AV + skg * BM + srow] = __builtin_bit_cast(uint4, lv_);                           \
        }                                                                                             \
        bs[BUF][tid] = wv;                                                                            \
    }

    const int r32 = lane & 31, kg = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int abase = kg * BM + 32 * MT * wr + r32;
    const int bbase = wc * DN_BT + kg * DN_BN + r32;
    f32x16 acc[MT][2];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][nb][i] = 0.0f;

    MH_DB_ISSUE(0)
    MH_DB_COMMIT(0)
    __syncthreads();
    for (int c = 0; c < nkc; ++c) {
        const int buf = c & 1;
        if (c + 1 < nkc) MH_DB_ISSUE(c + 1)
        uint4 ah[MT], al[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) { ah[mt] = as[buf][abase + 32 * mt]; al[mt] = as[buf][AV + abase + 32 * mt]; }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const uint4 bh = bs[buf][bbase + 32 * nb], bl = bs[buf][DN_BV + bbase + 32 * nb];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, bh), acc[mt][nb], 0, 0, 0);
                acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al[mt]), __builtin_bit_cast(f16x8, bh), acc[mt][nb], 0, 0, 0);
                acc[mt][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, bl), acc[mt][nb], 0, 0, 0);
            }
        }
        if (c + 1 < nkc) MH_DB_COMMIT(buf ^ 1)
        __syncthreads();
    }
#undef MH_DB_COMMIT
#undef MH_DB_ISSUE

    const float inv_scale = wtail[0];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = n0 + 64 * wc + 32 * nb + r32;
        const float bn = (bias && n < N) ? bias[n] : 0.0f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int mi = m0 + 32 * MT * wr + 32 * mt + 8 * (i >> 2) + 4 * kg + (i & 3);
                const int m = (rowmap && mi < M) ? rowmap[mi] : mi;
                if (mi < M && m >= 0 && n < N) {
                    float v = fmaf(acc[mt][nb][i], inv_scale, bn);
                    if (ACT == 1) v = gelu_erf(v);
                    if (RES) v += r[(long long)m * N + n];
                    y[(long long)m * N + n] = v;
                }
            }
    }
}

// Measured and not kept (round 6, profiles/r06_linear_variants.txt, tools/experiments/r06_linear_variants.patch): two 16-wide chunks per barrier with whole-cache-line X
// loads (bit-identical, same time), four chunks per barrier (128 KB of LDS, one workgroup per CU: 0.77 x), the A operand from global memory straight into registers
// (conversion repeated by the two waves that share the rows: 0.77 x).  Round 5's four-wave 64 x 64 form (never wired into a launcher) went into the same patch file.
// w [N][K] -> [column tile][chunk][piece][k-group][64 columns][8 k] fp16, zero padded in N and K; tail = {1 / scale, scale} (the scale
// kernel of conv3d_h2.h).  One thread per (n, k) of the padded matrix.
__global__ void __launch_bounds__(256)
linear_h2_pack_kernel(const float* __restrict__ w, int N, int K, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int nkc = (K + DN_BK - 1) / DN_BK, Kp = nkc * DN_BK;
    const int Np = (N + DN_BN - 1) / DN_BN * DN_BN;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)Np * Kp) return;
    const int k = (int)(idx % Kp), n = (int)(idx / Kp);
    const float v = (n < N && k < K) ? w[(long long)n * K + k] * tail[1] : 0.0f;
    _Float16 pc[2];
    h2_split(v, pc[0], pc[1]);
    _Float16* slab = packed + ((long long)(n / DN_BN) * nkc + k / DN_BK) * (DN_BT * 8LL);
#pragma unroll
    for (int p = 0; p < 2; ++p) slab[((p * 2 + (k % DN_BK) / 8) * DN_BN + n % DN_BN) * 8 + k % 8] = pc[p];
}

// nn.LayerNorm over the last dimension: one wave per row, two passes over registers (mean, then the centred sum of squares), biased
// variance, y = (x - mean) * rsqrt(var + eps) * gamma + beta.  NV = values per lane (K <= 64 NV), chosen by the launcher: with the
// 64-iteration form a 48-feature row of SwinUNETR's first stage ran 64x the instructions it needs (12 % of that network's step).
constexpr int LN_MAXV = 64;
template <int NV>
__global__ void __launch_bounds__(256)
layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ y,
                 int M, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long long)row * K;
    float v[NV];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = lane + 64 * i;
        v[i] = k < K ? xr[k] : 0.0f;
        s += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)K;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = lane + 64 * i;
        const float d = k < K ? v[i] - mean : 0.0f;
        q = fmaf(d, d, q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)K + eps);
    float* yr = y + (long long)row * K;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = lane + 64 * i;
        if (k < K) yr[k] = fmaf((v[i] - mean) * rstd, gamma ? gamma[k] : 1.0f, beta ? beta[k] : 0.0f);
    }
}

// Round 5: rows whose length is a multiple of 4 and at most 1024 (SwinUNETR's 48 ... 768-feature stages, ViT-B's 768; 16-byte aligned rows): G = 16 | 32 | 64 lanes per
// row, NV 16-byte vectors per lane, 64 / G rows per wave.  The one-wave-per-row form moved 192 bytes per wave instruction at 48 features (1.56 ms for 7 M rows =
// 1.75 TB/s).  Same two-pass arithmetic; the partial sums are formed per lane and then across the row's lanes.
// src_row (or null): output row r is the normalised INPUT row src_row[r], or zeros where src_row[r] < 0 -- SwinUNETR's norm1 -> pad -> roll -> window_partition
// (swin_unetr.py:624-648) as one pass: the map holds, per (window, token), the voxel row it comes from (-1 = padding, which the reference appends after the norm).
template <int G, int NV>
__global__ void __launch_bounds__(256)
layernorm_vec_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float* __restrict__ y,
                     int M, int K, const int* __restrict__ src_row) {
    constexpr int RPW = 64 / G;                                // rows per wave
    const int lane = threadIdx.x & 63, sub = lane % G;
    const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G;
    const bool rok = row < M;
    const long long srow = rok ? (src_row ? (long long)src_row[row] : row) : -1;
    f32x4 v[NV];
    bool ok[NV];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ok[i] = srow >= 0 && 4 * (sub + G * i) < K;
        v[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        if (ok[i]) v[i] = *reinterpret_cast<const f32x4*>(x + srow * K + 4 * (sub + G * i));
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)K;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = ok[i] ? f32x4{v[i][0] - mean, v[i][1] - mean, v[i][2] - mean, v[i][3] - mean} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        q = fmaf(v[i][3], v[i][3], fmaf(v[i][2], v[i][2], fmaf(v[i][1], v[i][1], fmaf(v[i][0], v[i][0], q))));
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)K + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = 4 * (sub + G * i);
        if (!rok || k >= K) continue;
        f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
        if (srow >= 0) {
            f32x4 g = {1.0f, 1.0f, 1.0f, 1.0f}, b = {0.0f, 0.0f, 0.0f, 0.0f};
            if (gamma) g = *reinterpret_cast<const f32x4*>(gamma + k);
            if (beta) b = *reinterpret_cast<const f32x4*>(beta + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(v[i][e] * rstd, g[e], b[e]);
        }
        *reinterpret_cast<f32x4*>(y + row * K + k) = o;
    }
}

}  // namespace mh
