// Multi-head self-attention core of UNETR's ViT encoder (softmax(Q K^T * scale) V) on the matrix cores.
//
// Reference: SABlock.forward, monai/networks/blocks/selfattention.py:156-218 -- qkv = Linear(x) rearranged
// "b h (qkv l d) -> qkv b l h d", att = einsum("blxd,blyd->blxy") * scale, softmax(-1), einsum("bhxy,bhyd->bhxd"),
// "b l h d -> b h (l d)".  The kernel consumes the qkv projection output directly ([B][S][3*heads*HD]) and writes
// the pre-out_proj tensor ([B][S][heads*HD]): neither the [B,heads,S,S] score tensor nor the rearranged copies exist.
// (Round 1-2 ran this on the fp32 matrix cores with K and V of a head resident in LDS: S <= 224, 36-38 TF = 0.23 of the fp32 peak -- one LDS read
// per 32x32x2 MFMA.  Replaced by the streaming split-precision kernel below.)
#pragma once
#include "common.h"

namespace mh {

// ---------------------------------------------------------------------------------------------------
// Attention for ANY sequence length, on the fp16 matrix cores in two-piece split precision (fp32-equivalent, like linear_h2_kernel and
// conv3d_k3_h2_kernel: every fp32 operand x = hi + lo with hi = fp16(x), lo = fp16(x - hi); a product is hi*hi + lo*hi + hi*lo, each exact in fp32,
// accumulated in fp32 by v_mfma_f32_32x32x16_f16) -- 3/16 of the fp32 matrix-core cycles.  Keys / values stream through LDS in tiles of 32 with an online softmax, so the
// reference's own docstring example (UNETR img_size 128^3 = 512 tokens, monai/networks/nets/unetr.py:75) and anything longer run on it.
//
// A workgroup = 4 waves = 128 queries of one (batch, head); wave w owns queries 32 w .. 32 w + 31 and keeps, per lane (query l & 31, half l >> 5):
//   * Q as the B operand of S^T = K Q^T (hi / lo pieces, HD / 16 k-steps);
//   * the score tile S^T in the accumulator layout of the 32x32 MFMA: register r = key (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the tile, i.e. the two lanes of
//     a query hold its 32 keys between them -- max / sum need one lane-pair shuffle;
//   * O^T = V^T P^T (M = head dim, N = query, K = key): the B operand P^T wants, per k-step s and half, the keys of registers 8 s .. 8 s + 7 of THIS lane, so
//     the probabilities never leave their registers; V^T is staged in LDS with the tile's keys permuted into that order (position 16 s + 8 half + j <->
//     key (r & 3) + 8 (r >> 2) + 4 half, r = 8 s + j).  O^T's accumulator holds, in the lane of a query, that query's outputs: the running rescale
//     by exp(m_old - m_new) is a per-lane scalar.
// Range: q, k, v must stay below 65504 in magnitude (fp16); they are outputs of a Linear over LayerNorm-ed tokens (|x| of order 1-10) -- a value beyond it
// turns the affected outputs into NaN (loud), it does not saturate silently.  exp via v_exp_f32 (exp2 of the log2(e)-scaled argument, <= 1 ulp + the
// argument's rounding: 1e-6 relative at |s - m| ~ 20, where the probability itself is 2e-9).
template <int HD>      // head dimension: a multiple of 32, <= 128
__global__ void __launch_bounds__(256) attention_h2_kernel(const float* __restrict__ qkv, float* __restrict__ out, int S, int heads, float scale) {
    constexpr int KS = HD / 16, NT = HD / 32, PER = HD / 8;       // k-steps of K Q^T, 32-row tiles of O^T, floats a thread stages per key row
    constexpr int KP = HD + 8, VP = 40;                            // LDS pitches in halves (16-byte aligned rows that start in different banks)
    __shared__ __attribute__((aligned(16))) _Float16 ks[2][2][32 * KP];     // [buffer][piece][key][d]
    __shared__ __attribute__((aligned(16))) _Float16 vs[2][2][HD * VP];     // [buffer][piece][d][permuted key]
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.y, b = blockIdx.z;
    const int hd = heads * HD;
    const float* base = qkv + (long long)b * S * 3 * hd + head * HD;
    const int query = blockIdx.x * 128 + wave * 32 + li;
    const int ntiles = (S + 31) / 32;

    // this lane's Q slices: d = 16 s + 8 hi + j
    f16x8 qh[KS], ql[KS];
    {
        const float* qrow = base + (long long)min(query, S - 1) * 3 * hd;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * hi), c = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * hi + 4);
            const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const _Float16 h = (_Float16)v[j];
                qh[s][j] = h;
                ql[s][j] = (_Float16)(v[j] - (float)h);
            }
        }
    }
    // staging: thread -> key row tid >> 3 of the tile, PER consecutive head-dim elements
    const int skey = tid >> 3, sd0 = (tid & 7) * PER;
    const int spos = ((((skey & 3) + 4 * (skey >> 3)) >> 3) << 4) + (((skey >> 2) & 1) << 3) + (((skey & 3) + 4 * (skey >> 3)) & 7);   // permuted position of key skey
    float kreg[PER], vreg[PER];
#define MH_AT_LOAD(T)                                                                                 \
    {                                                                                                 \
        const int key_ = (T) * 32 + skey;                                                             \
        const float* row_ = base + (long long)min(key_, S - 1) * 3 * hd + sd0;                        \
        const float z_ = key_ < S ? 1.0f : 0.0f;           /* rows beyond the sequence stage zeros */ \
        _Pragma("unroll") for (int i = 0; i < PER; i += 4) {                                          \
            const f32x4 a_ = *reinterpret_cast<const f32x4*>(row_ + hd + i), c_ = *reinterpret_cast<const f32x4*>(row_ + 2 * hd + i); \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) { kreg[i + e] = a_[e] * z_; vreg[i + e] = c_[e] * z_; } \
        }                                                                                             \
    }
#define MH_AT_STORE(BUF)                                                                              \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < PER; i += 4) {                                          \
            f16x4 h_, l_;                                                                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                           \
                h_[e] = (_Float16)kreg[i + e];                                                        \
                l_[e] = (_Float16)(kreg[i + e] - (float)h_[e]);                                       \
            }                                                                                         \
            *reinterpret_cast<f16x4*>(&ks[BUF][0][skey * KP + sd0 + i]) = h_;                         \
            *reinterpret_cast<f16x4*>(&ks[BUF][1][skey * KP + sd0 + i]) = l_;                         \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                             \
            const _Float16 h_ = (_Float16)vreg[i];                                                    \
            vs[BUF][0][(sd0 + i) * VP + spos] = h_;                                                   \
            vs[BUF][1][(sd0 + i) * VP + spos] = (_Float16)(vreg[i] - (float)h_);                      \
        }                                                                                             \
    }

    f32x16 o[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[nt][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    const float c2 = scale * 1.44269504088896341f;          // scores are kept in log2 units: p = exp2(s c2 - m)

    MH_AT_LOAD(0)
    MH_AT_STORE(0)
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) MH_AT_LOAD(t + 1)               // in flight during this tile's matrix work
        // scores of this tile: S^T = K Q^T
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(&ks[buf][0][li * KP + 16 * s + 8 * hi]);
            const f16x8 al = *reinterpret_cast<const f16x8*>(&ks[buf][1][li * KP + 16 * s + 8 * hi]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], acc, 0, 0, 0);
        }
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            acc[r] = key < S ? acc[r] * c2 : -INFINITY;
            mt = fmaxf(mt, acc[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float corr = exp2f(m_run - m_new);            // first tile: exp2(-inf) = 0
        float psum = 0.0f;
        f16x8 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = exp2f(acc[r] - m_new);
            psum += p;
            const _Float16 h = (_Float16)p;
            ph[r >> 3][r & 7] = h;
            pl[r >> 3][r & 7] = (_Float16)(p - (float)h);
        }
        l_run = l_run * corr + psum;
        m_run = m_new;
        // O^T = O^T corr + V^T P^T
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[nt][r] *= corr;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 vh = *reinterpret_cast<const f16x8*>(&vs[buf][0][(32 * nt + li) * VP + 16 * s + 8 * hi]);
                const f16x8 vl = *reinterpret_cast<const f16x8*>(&vs[buf][1][(32 * nt + li) * VP + 16 * s + 8 * hi]);
                o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s], o[nt], 0, 0, 0);
                o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s], o[nt], 0, 0, 0);
                o[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s], o[nt], 0, 0, 0);
            }
        }
        if (t + 1 < ntiles) MH_AT_STORE(buf ^ 1)            // the other buffer: its last readers passed the barrier of the previous iteration
        __syncthreads();
    }
#undef MH_AT_STORE
#undef MH_AT_LOAD
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
    if (query < S) {
        float* orow = out + ((long long)b * S + query) * hd + head * HD;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)         // registers 4 g .. 4 g + 3 = four consecutive head-dim elements 32 nt + 8 g + 4 hi ..
                *reinterpret_cast<f32x4*>(orow + 32 * nt + 8 * g + 4 * hi) =
                    f32x4{o[nt][4 * g] * inv, o[nt][4 * g + 1] * inv, o[nt][4 * g + 2] * inv, o[nt][4 * g + 3] * inv};
    }
}

// ---------------------------------------------------------------------------------------------------
// Window attention of the Swin transformer (WindowAttention.forward, monai/networks/nets/swin_unetr.py:519-541):
//   attn = softmax((q * scale) k^T + relative_position_bias[head] + mask[window % nW]) v      per (window, head)
// on the qkv projection's output [BW][S][3][heads][HD]; writes [BW][S][heads * HD] (the pre-`proj` tensor).  Small heads
// (HD = 8 / 16 / 32 at feature sizes 24 / 48 / 96), S = window volume (343 for 7^3; 27 ... 216 when the feature map is
// smaller than the window): the products are too thin for a 32x32 MFMA tile to pay (K = HD), so this is a VALU kernel --
// one workgroup per (window, head), K and V of the head in LDS (broadcast reads), one query row per thread with an online
// softmax; bias and mask rows are read TRANSPOSED ([key][query]) so that the threads of a wave read consecutive addresses.
// `bias_t` [heads][S][S] = bias[head][query][key] stored as [head][key][query]; `mask` [nW][S][S] (symmetric: 0 / -100) or null.
constexpr int WA_MAX_TOKENS = 352;       // 7^3 = 343 tokens per window

template <int HD>
__global__ void __launch_bounds__(256) window_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_t,
                                                               const float* __restrict__ mask, float* __restrict__ out, int S, int heads, int nW,
                                                               float scale) {
    __shared__ __attribute__((aligned(16))) float ks[WA_MAX_TOKENS * HD];      // [S][HD]
    __shared__ __attribute__((aligned(16))) float vs[WA_MAX_TOKENS * HD];
    const int tid = threadIdx.x;
    const int head = blockIdx.x, w = blockIdx.y;
    const int C3 = 3 * heads * HD;
    const float* base = qkv + (long long)w * S * C3 + head * HD;
    for (int i = tid; i < S * (HD / 4); i += 256) {
        const int r = i / (HD / 4), c4 = i - r * (HD / 4);
        reinterpret_cast<f32x4*>(ks)[i] = *reinterpret_cast<const f32x4*>(base + (long long)r * C3 + heads * HD + 4 * c4);
        reinterpret_cast<f32x4*>(vs)[i] = *reinterpret_cast<const f32x4*>(base + (long long)r * C3 + 2 * heads * HD + 4 * c4);
    }
    __syncthreads();
    const float* bt = bias_t ? bias_t + (long long)head * S * S : nullptr;
    const float* mk = mask ? mask + (long long)(w % nW) * S * S : nullptr;
    for (int r = tid; r < S; r += 256) {
        float q[HD];
#pragma unroll
        for (int c4 = 0; c4 < HD / 4; ++c4) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(base + (long long)r * C3 + 4 * c4);
            q[4 * c4] = t[0] * scale; q[4 * c4 + 1] = t[1] * scale; q[4 * c4 + 2] = t[2] * scale; q[4 * c4 + 3] = t[3] * scale;
        }
        float m = -INFINITY, l = 0.0f;
        float acc[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] = 0.0f;
        for (int j = 0; j < S; ++j) {
            float sc = 0.0f;
#pragma unroll
            for (int d = 0; d < HD; ++d) sc = fmaf(q[d], ks[j * HD + d], sc);
            if (bt) sc += bt[(long long)j * S + r];
            if (mk) sc += mk[(long long)j * S + r];
            const float mn = fmaxf(m, sc);
            const float corr = expf(m - mn), pj = expf(sc - mn);
            l = fmaf(l, corr, pj);
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(acc[d], corr, pj * vs[j * HD + d]);
            m = mn;
        }
        const float inv = 1.0f / l;
        float* o = out + ((long long)w * S + r) * (heads * HD) + head * HD;
#pragma unroll
        for (int c4 = 0; c4 < HD / 4; ++c4)
            *reinterpret_cast<f32x4*>(o + 4 * c4) = f32x4{acc[4 * c4] * inv, acc[4 * c4 + 1] * inv, acc[4 * c4 + 2] * inv, acc[4 * c4 + 3] * inv};
    }
}


// ---------------------------------------------------------------------------------------------------
// Round 4: the same window attention on the fp16 matrix cores -- attention_h2_kernel's streaming split-precision core (keys / values in LDS tiles of 32,
// online softmax, probabilities kept in the accumulator layout) with the relative position bias and the shift mask ADDED TO THE SCORE TILE before the softmax
// (one coalesced load per register: bias_t / mask are key-major, the 32 lanes of a tile row are 32 consecutive queries).  Head dims 16 and 32 (SwinUNETR feature
// sizes 48 / 96): the K Q^T product has HD / 16 k-steps, O^T = V^T P^T one 32-row tile whose rows beyond HD multiply zeros (half of its matrix work is padding at
// HD = 16 -- still a fraction of the VALU kernel's time: S^2 (2 HD) multiply-adds per head become S^2 / 1024 x 9 matrix instructions).
// A workgroup = 4 waves = 128 queries of one (window, head).  fp32-equivalent like every split-precision kernel here; q, k, v are LayerNorm-ed projections (|x| ~ 1-10).
//
// Round 5, REL: the bias and the mask are no longer READ as S x S tables (470 KB each per head / mask window at 7^3 tokens: 351 KB per workgroup against 66 KB of q, k, v --
// the kernel ran at the L2's pace, 18.7 % of SwinUNETR's step) but EVALUATED from what they are made of.  The relative position index is linear in the token
// coordinates, index[q][k] = c(q) - c(k) + off (swin_unetr.py:492-519: the per-axis differences, shifted and multiplied up), so the bias of a pair is one gather
// from the head's column of the table (at most WA_REL_ROWS floats in LDS) at coord[q] - coord[k] + off; the shift mask is -100 where the two tokens' region ids
// differ (swin_unetr.py:774-812).  Per key the workgroup keeps coord | region << 16 in LDS; a lane reads the four keys of an accumulator group with one 16-byte read.
// The sums are formed in the order of the table form (0 + bias, + mask): the results are bit-identical to it.
struct WinRel {
    const float* table;      // [rows][heads]: the relative_position_bias_table parameter as it is
    const int* coord;        // [S]: index[t][0] of token t
    const int* region;       // [nW][S] region ids of the shifted volume's windows, or null
    int rows, off;           // off = index[0][0]
};
constexpr int WA_REL_ROWS = 4096, WA_REL_TOKENS = 1024;

template <int HD, bool REL>      // 16 | 32
__global__ void __launch_bounds__(256) window_attention_h2_kernel(const float* __restrict__ qkv, const float* __restrict__ bias_t, const float* __restrict__ mask,
                                                                  float* __restrict__ out, int S, int heads, int nW, float scale, WinRel rel) {
    constexpr int KS = HD / 16, PER = HD / 8;                  // k-steps of K Q^T; floats a thread stages per key row (2 | 4)
    constexpr int KP = HD + 8, VP = 40;                        // LDS pitches in halves
    __shared__ __attribute__((aligned(16))) _Float16 ks[2][2][32 * KP];     // [buffer][piece][key][d]
    __shared__ __attribute__((aligned(16))) _Float16 vs[2][2][32 * VP];     // [buffer][piece][d (rows >= HD stay zero)][permuted key]
    __shared__ float tab[REL ? WA_REL_ROWS : 1];                            // this head's column of the bias table
    __shared__ __attribute__((aligned(16))) unsigned crs[REL ? WA_REL_TOKENS : 4];     // coord | region << 16 per key (zeros beyond the window)
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int head = blockIdx.y, w = blockIdx.z;
    const int hd = heads * HD;
    const float* base = qkv + (long long)w * S * 3 * hd + head * HD;
    const int query = blockIdx.x * 128 + wave * 32 + li;
    const int qc = min(query, S - 1);
    const int ntiles = (S + 31) / 32;
    const float* bt = bias_t ? bias_t + (long long)head * S * S + qc : nullptr;           // + key * S
    const float* mk = mask ? mask + (long long)(w % nW) * S * S + qc : nullptr;

    for (int i = tid; i < 2 * 2 * 32 * VP / 2; i += 256) reinterpret_cast<unsigned*>(&vs[0][0][0])[i] = 0u;      // the padding rows of V^T
    unsigned cq = 0u, rq = 0u;
    if (REL) {
        const int* reg = rel.region ? rel.region + (long long)(w % nW) * S : nullptr;
        for (int i = tid; i < rel.rows; i += 256) tab[i] = rel.table[(long long)i * heads + head];
        for (int i = tid; i < ntiles * 32; i += 256) crs[i] = i < S ? ((unsigned)rel.coord[i] | ((reg ? (unsigned)reg[i] : 0u) << 16)) : 0u;
        cq = (unsigned)(rel.coord[qc] + rel.off);
        rq = reg ? (unsigned)reg[qc] : 0u;
    }
    f16x8 qh[KS], ql[KS];
    {
        const float* qrow = base + (long long)qc * 3 * hd;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * hi), c = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * hi + 4);
            const float v[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const _Float16 h = (_Float16)v[j];
                qh[s][j] = h;
                ql[s][j] = (_Float16)(v[j] - (float)h);
            }
        }
    }
    // staging: thread -> key row tid >> 3 of the tile, PER consecutive head-dim elements
    const int skey = tid >> 3, sd0 = (tid & 7) * PER;
    const int spos = ((((skey & 3) + 4 * (skey >> 3)) >> 3) << 4) + (((skey >> 2) & 1) << 3) + (((skey & 3) + 4 * (skey >> 3)) & 7);   // permuted position of key skey (attention_h2_kernel)
    float kreg[PER], vreg[PER];
#define MH_WA_LOAD(T)                                                                                 \
    {                                                                                                 \
        const int key_ = (T) * 32 + skey;                                                             \
        const float* row_ = base + (long long)min(key_, S - 1) * 3 * hd + sd0;                        \
        const float z_ = key_ < S ? 1.0f : 0.0f;           /* rows beyond the window stage zeros */   \
        _Pragma("unroll") for (int i = 0; i < PER; i += 2) {                                          \
            const f32x2 a_ = *reinterpret_cast<const f32x2*>(row_ + hd + i), c_ = *reinterpret_cast<const f32x2*>(row_ + 2 * hd + i); \
            kreg[i] = a_[0] * z_; kreg[i + 1] = a_[1] * z_; vreg[i] = c_[0] * z_; vreg[i + 1] = c_[1] * z_; \
        }                                                                                             \
    }
#define MH_WA_STORE(BUF)                                                                              \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < PER; i += 2) {                                          \
            f16x2 h_, l_;                                                                             \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                           \
                h_[e] = (_Float16)kreg[i + e];                                                        \
                l_[e] = (_Float16)(kreg[i + e] - (float)h_[e]);                                       \
            }                                                                                         \
            *reinterpret_cast<f16x2*>(&ks[BUF][0][skey * KP + sd0 + i]) = h_;                         \
            *reinterpret_cast<f16x2*>(&ks[BUF][1][skey * KP + sd0 + i]) = l_;                         \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                             \
            const _Float16 h_ = (_Float16)vreg[i];                                                    \
            vs[BUF][0][(sd0 + i) * VP + spos] = h_;                                                   \
            vs[BUF][1][(sd0 + i) * VP + spos] = (_Float16)(vreg[i] - (float)h_);                      \
        }                                                                                             \
    }
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    constexpr float LOG2E = 1.44269504088896341f;
    const float c2 = scale * LOG2E;                         // scores are kept in log2 units: p = exp2((q k scale + bias + mask) log2e - m)

    MH_WA_LOAD(0)
    __syncthreads();                                        // (the zeroed V^T rows)
    MH_WA_STORE(0)
    __syncthreads();
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) MH_WA_LOAD(t + 1)
        // bias + mask of this tile's (key, query) pairs: requested before the matrix work that they are added to
        float add[16];
        if (REL) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32x4 v_ = *reinterpret_cast<const u32x4*>(&crs[t * 32 + 8 * g + 4 * hi]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned idx = min(cq - (v_[e] & 0xffffu), (unsigned)(rel.rows - 1));
                    float a_ = 0.0f + tab[idx];
                    if (rel.region) a_ += (v_[e] >> 16) != rq ? -100.0f : 0.0f;
                    add[4 * g + e] = a_;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = min(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi, S - 1);
                float a_ = 0.0f;
                if (bt) a_ += bt[(long long)key * S];
                if (mk) a_ += mk[(long long)key * S];
                add[r] = a_;
            }
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const f16x8 ah = *reinterpret_cast<const f16x8*>(&ks[buf][0][li * KP + 16 * s + 8 * hi]);
            const f16x8 al = *reinterpret_cast<const f16x8*>(&ks[buf][1][li * KP + 16 * s + 8 * hi]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, qh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, qh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ql[s], acc, 0, 0, 0);
        }
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            acc[r] = key < S ? fmaf(add[r], LOG2E, acc[r] * c2) : -INFINITY;
            mt = fmaxf(mt, acc[r]);
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float m_new = fmaxf(m_run, mt);
        const float corr = exp2f(m_run - m_new);
        float psum = 0.0f;
        f16x8 ph[2], pl[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = exp2f(acc[r] - m_new);
            psum += p;
            const _Float16 h = (_Float16)p;
            ph[r >> 3][r & 7] = h;
            pl[r >> 3][r & 7] = (_Float16)(p - (float)h);
        }
        l_run = l_run * corr + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= corr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f16x8 vh = *reinterpret_cast<const f16x8*>(&vs[buf][0][li * VP + 16 * s + 8 * hi]);
            const f16x8 vl = *reinterpret_cast<const f16x8*>(&vs[buf][1][li * VP + 16 * s + 8 * hi]);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s], o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s], o, 0, 0, 0);
        }
        if (t + 1 < ntiles) MH_WA_STORE(buf ^ 1)
        __syncthreads();
    }
#undef MH_WA_STORE
#undef MH_WA_LOAD
    const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
    if (query < S) {
        float* orow = out + ((long long)w * S + query) * hd + head * HD;
#pragma unroll
        for (int g = 0; g < HD / 8; ++g)        // registers 4 g .. 4 g + 3 = head-dim elements 8 g + 4 hi .. (rows >= HD of the tile are padding)
            *reinterpret_cast<f32x4*>(orow + 8 * g + 4 * hi) = f32x4{o[4 * g] * inv, o[4 * g + 1] * inv, o[4 * g + 2] * inv, o[4 * g + 3] * inv};
    }
}

}  // namespace mh
