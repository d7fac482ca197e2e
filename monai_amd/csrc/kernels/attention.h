// Multi-head self-attention core of UNETR's ViT encoder (softmax(Q K^T * scale) V) on the fp32 matrix cores.
//
// Reference: SABlock.forward, monai/networks/blocks/selfattention.py:156-218 -- qkv = Linear(x) rearranged
// "b h (qkv l d) -> qkv b l h d", att = einsum("blxd,blyd->blxy") * scale, softmax(-1), einsum("bhxy,bhyd->bhxd"),
// "b l h d -> b h (l d)".  This kernel consumes the qkv projection output directly ([B][S][3*heads*64]) and writes
// the pre-out_proj tensor ([B][S][heads*64]): neither the [B,heads,S,S] score tensor nor the rearranged copies exist.
//
// One workgroup = one (batch, head); K (transposed) and V of the head sit in LDS (2 x 57 KB at S = 216, of 160 KB).
// A wave owns 32 queries at a time and computes the TRANSPOSED score tile S^T = K Q^T with v_mfma_f32_32x32x2_f32:
// lane l then holds, for query (l & 31), the keys (r&3)+8(r>>2)+4(l>>5) of every key tile in its accumulator
// registers -- exactly the A-operand layout of the following P V product (A[i=query][k] lives in lane i + 32k), so the
// softmax runs in registers (one lane-pair shuffle for max and sum) and P is never moved: the k-slices of each PV MFMA
// are simply the key pair (key, key+4) that the two half-waves already hold.  fp32 in, fp32 accumulate, exact expf.
#pragma once
#include "common.h"

namespace mh {

template <int KT>   // key tiles of 32 (sequence length <= 32*KT)
__global__ void __launch_bounds__(256) attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int S, int heads, float scale) {
    constexpr int KP = KT * 32, KSTR = KP + 1;       // padded key count; odd row stride: conflict-free transposed writes
    __shared__ float kt_s[64 * KSTR];                 // K^T : [d][key]
    __shared__ float v_s[KP * 64];                    // V   : [key][d]
    __shared__ float lsum[4 * 32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
    const int head = blockIdx.x, b = blockIdx.y;
    const int hd = heads * 64;
    const float* base = qkv + (long long)b * S * 3 * hd;

    for (int i = tid; i < KP * 64; i += 256) {
        const int s = i >> 6, d = i & 63;
        float kv = 0.0f, vv = 0.0f;
        if (s < S) {
            const float* row = base + (long long)s * 3 * hd + head * 64 + d;
            kv = row[hd];
            vv = row[2 * hd];
        }
        kt_s[d * KSTR + s] = kv;
        v_s[i] = vv;
    }
    __syncthreads();

    for (int qt = wave; qt < KT; qt += 4) {
        const int query = qt * 32 + li;
        float qreg[32];
#pragma unroll
        for (int s = 0; s < 32; ++s) qreg[s] = query < S ? base[(long long)query * 3 * hd + head * 64 + 2 * s + hi] : 0.0f;

        f32x16 acc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kt][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 32; ++s)
                acc[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kt_s[(2 * s + hi) * KSTR + kt * 32 + li], qreg[s], acc[kt], 0, 0, 0);
        }
        // softmax over the keys of query (l & 31): this lane and its partner lane ^ 32 hold them all
        float m = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float v = key < S ? acc[kt][r] * scale : -3.0e38f;
                acc[kt][r] = v;
                m = fmaxf(m, v);
            }
        m = fmaxf(m, __shfl_xor(m, 32));
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = expf(acc[kt][r] - m);
                acc[kt][r] = p;
                sum += p;
            }
        sum += __shfl_xor(sum, 32);
        if (hi == 0) lsum[wave * 32 + li] = sum;

        f32x16 o[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[nt][r] = 0.0f;
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    o[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[kt][r], v_s[key * 64 + nt * 32 + li], o[nt], 0, 0, 0);
                }
        }
        // D layout of o: lane = head-dim column (l & 31), registers = query rows (r&3)+8(r>>2)+4(l>>5)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                const int q = qt * 32 + row;
                if (q < S) out[((long long)b * S + q) * hd + head * 64 + nt * 32 + li] = o[nt][r] / lsum[wave * 32 + row];
            }
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

}  // namespace mh
