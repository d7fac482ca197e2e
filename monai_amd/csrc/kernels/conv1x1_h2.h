// 1x1x1 convolution of act(in) with ALL its output channels from ONE read of the input, on the fp16 matrix cores in the two-piece split precision of conv3d_h2.h
// (round 5).  Reference op: UnetResBlock.conv3 -- the shortcut of a residual block whose channel count changes (monai/networks/blocks/dynunet_block.py:72-111), with
// norm3's InstanceNorm statistics out of the same launch.
//
// Why: conv1x1_kernel (nn_simple.h) keeps 16 output channels per thread and re-reads the input once per group of 16 -- SwinUNETR(48)'s decoder1 shortcut (96 -> 48
// channels @ 96^3 x 64 windows) read 3 x 21.7 GB for 10.9 GB of output and was 9.5 % of that network's step.  The op is HBM-bound only if the input is read once, and
// 96 x 48 multiply-adds per voxel do not fit the vector ALU beside that stream; as a GEMM on the matrix cores they are a quarter of the memory time.
//
// GEMM: M = output channels (A = the packed weights, LDS-resident for the whole launch), N = voxels (B = the activated input, loaded from HBM STRAIGHT INTO the matrix
// operand layout: lane (r32, kg) of a wave owns the four consecutive voxels 4 r32 .. + 3 -- one 16-byte load per channel, 512 contiguous bytes per half wave -- of the
// eight channels 16 s + 8 kg .. + 7 of k-step s: exactly the B operand of v_mfma_f32_32x32x16_f16 for four N-tiles, no LDS staging, no barrier in the main loop),
// K = input channels.  The D layout (lane = voxel column, registers = output channels) gives 16-byte stores of the same four voxels per output channel.
// A workgroup = 8 waves = 1024 consecutive voxels of one sample (the statistics tile of conv1x1_kernel: the record count does not change); MT x 32 output channels per
// pass (MT = 2: up to 64), further groups of 64 in further launches.  Input range: as conv3d_h2.h -- the records carry bounds, the sample's largest puts the scale.
// Statistics: {count, mean, M2 about the workgroup's mean} per (n, cout, workgroup), two passes over the accumulators as conv1x1_kernel.
#pragma once
#include "common.h"
#include "conv3d_h2.h"

namespace mh {

constexpr int C1H_KS = 32;                                  // k-steps of 16 input channels whose weights are in LDS at a time (512 channels; more: reloaded per chunk)
constexpr int C1H_CHUNKS = 3;                               // weight chunks per launch: Cin <= 1536
constexpr int C1H_SLAB = 2 * 2 * 64;                        // uint4 per k-step of a pass: [piece][m-tile][kg][32 couts]

// 16 values per lane summed over the 32 lanes of the lane's half wave (the halves hold different output channels): 16 shuffles instead of 80; every lane ends with the
// complete sum of the value index c1h_owner(lane) (lanes differing in bit 0 hold the same one)
__device__ __forceinline__ int c1h_owner(int lane) { return ((lane & 16) ? 8 : 0) + ((lane & 8) ? 4 : 0) + ((lane & 4) ? 2 : 0) + ((lane & 2) ? 1 : 0); }
__device__ __forceinline__ float c1h_half_sum16(float (&s)[16], int lane) {
    int off = 16;
#pragma unroll
    for (int n = 16; n > 1; n >>= 1, off >>= 1) {
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int j = 0; j < n / 2; ++j) {
            const float keep = hi ? s[j + n / 2] : s[j], send = hi ? s[j] : s[j + n / 2];
            s[j] = keep + __shfl_xor(send, off);
        }
    }
    return s[0] + __shfl_xor(s[0], 1);
}

template <int MT, bool STATS>
__global__ void __launch_bounds__(512)
conv1x1_h2_kernel(Tensor in, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias, Tensor out, int co0,
                  float* __restrict__ stats, int tiles) {
    __shared__ uint4 ws[C1H_KS * C1H_SLAB];                 // 128 KB: the pass's weights
    __shared__ float nrm_s[3 * 16 * C1H_KS * C1H_CHUNKS];   // {alpha, beta, slope} per input channel
    __shared__ unsigned bound_s[8];
    __shared__ float red_s[8][64], mean_s[64];
    const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = in.C, n = blockIdx.y;
    const long long DHW = (long long)in.D * in.H * in.W;
    const int nks = (Cin + 15) / 16;
    const long long idx = (long long)blockIdx.x * 1024 + wave * 128 + 4 * r32;
    const bool valid = idx < DHW;                            // DHW % 4 == 0 (launcher): the lane's four voxels are inside or outside together

    for (int i = tid; i < min(nks, C1H_KS) * C1H_SLAB; i += 512) ws[i] = wp[i];
    unsigned mb = 0u;
    for (int c = tid; c < 16 * nks; c += 512) {
        const float4 a = c < Cin ? *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * c) : make_float4(1.0f, 0.0f, 1.0f, 1.0f);
        nrm_s[3 * c] = a.x; nrm_s[3 * c + 1] = a.y; nrm_s[3 * c + 2] = a.z;
        const unsigned bb = abs_bits(a.w);
        mb = max(mb, bb == 0u ? 0x7fc00000u : bb);           // no bound given counts as non-finite (conv3d_h2.h)
    }
    mb = wave_umax(mb);
    if (lane == 0) bound_s[wave] = mb;
    __syncthreads();
    mb = bound_s[0];
#pragma unroll
    for (int w = 1; w < 8; ++w) mb = max(mb, bound_s[w]);
    const bool poisoned = mb >= 0x7f800000u;
    const int e_in = poisoned ? 0 : min(max(15 - ((int)(mb >> 23) - 126), -100), 100);
    const float p_in = __uint_as_float((unsigned)(e_in + 127) << 23);

    f32x16 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][t][i] = 0.0f;

    const float* src = in.data + (long long)n * in.n_stride + (valid ? idx : 0);
    f32x4 x[8], xn[8];
#define MH_C1H_LOAD(S, X)                                                                             \
    {                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                               \
            const int c_ = min(16 * (S) + 8 * kg + j, Cin - 1);     /* channels beyond Cin meet zero weights */ \
            X[j] = *reinterpret_cast<const f32x4*>(src + (long long)c_ * DHW);                        \
        }                                                                                             \
    }
    MH_C1H_LOAD(0, x)
    for (int s = 0; s < nks; ++s) {
        if (s > 0 && s % C1H_KS == 0) {                      // more than 512 input channels: the next chunk of the weights replaces the one in LDS
            __syncthreads();
            for (int i = tid; i < min(nks - s, C1H_KS) * C1H_SLAB; i += 512) ws[i] = wp[(long long)s * C1H_SLAB + i];
            __syncthreads();
        }
        if (s + 1 < nks) MH_C1H_LOAD(s + 1, xn)
        // activate, scale by 2^e_in and split this k-step's 8 channels x 4 voxels
        f16x8 bh[4], bl[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c_ = 16 * s + 8 * kg + j;
            const float al = nrm_s[3 * c_] * p_in, be = nrm_s[3 * c_ + 1] * p_in, sl = nrm_s[3 * c_ + 2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                _Float16 h, l;
                h2_split(act(x[j][t], al, be, sl), h, l);
                bh[t][j] = h;
                bl[t][j] = l;
            }
        }
        const uint4* wk = ws + (s % C1H_KS) * C1H_SLAB + kg * 32 + r32;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f16x8 ah = __builtin_bit_cast(f16x8, wk[mt * 64]), al = __builtin_bit_cast(f16x8, wk[2 * 64 + mt * 64]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[t], acc[mt][t], 0, 0, 0);
                acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[t], acc[mt][t], 0, 0, 0);
                acc[mt][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[t], acc[mt][t], 0, 0, 0);
            }
        }
        if (s + 1 < nks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = xn[j];
        }
    }
#undef MH_C1H_LOAD

    // scale back: 2^-(weight scale exponent) * 2^-e_in as two power-of-two factors (conv3d_h2.h); a poisoned bound turns the sample's output into NaN
    const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - e_in;
    const int ta = t_ / 2, tb = t_ - ta;
    const float inv_a = poisoned ? __uint_as_float(0x7fc00000u) : __uint_as_float((unsigned)(ta + 127) << 23), inv_b = __uint_as_float((unsigned)(tb + 127) << 23);
    const int Cout = out.C;
    float* dst = out.data + (long long)n * out.n_stride + (valid ? idx : 0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = co0 + 32 * mt + 8 * (i >> 2) + 4 * kg + (i & 3);
            const float bv = (bias && co < Cout) ? bias[co] : 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[mt][t][i] = (acc[mt][t][i] * inv_a) * inv_b + bv;
            if (valid && co < Cout)
                *reinterpret_cast<f32x4*>(dst + (long long)co * DHW) = f32x4{acc[mt][0][i], acc[mt][1][i], acc[mt][2][i], acc[mt][3][i]};
        }
    if (STATS) {
        const float tot = (float)min(1024LL, DHW - (long long)blockIdx.x * 1024);
        const int own = c1h_owner(lane);
        const int oc = 8 * (own >> 2) + 4 * kg + (own & 3);         // output channel (inside an m-tile) of the value this lane ends up owning
        float v[16];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = valid ? (acc[mt][0][i] + acc[mt][1][i]) + (acc[mt][2][i] + acc[mt][3][i]) : 0.0f;
            const float r = c1h_half_sum16(v, lane);
            if ((lane & 1) == 0) red_s[wave][32 * mt + oc] = r;
        }
        __syncthreads();
        if (tid < 32 * MT) mean_s[tid] = (((red_s[0][tid] + red_s[1][tid]) + (red_s[2][tid] + red_s[3][tid])) + ((red_s[4][tid] + red_s[5][tid]) + (red_s[6][tid] + red_s[7][tid]))) / tot;
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float m_ = mean_s[32 * mt + 8 * (i >> 2) + 4 * kg + (i & 3)];
                float q = 0.0f;
#pragma unroll
                for (int t = 0; t < 4; ++t) { const float d = acc[mt][t][i] - m_; q = fmaf(d, d, q); }
                v[i] = valid ? q : 0.0f;
            }
            const float r = c1h_half_sum16(v, lane);
            if ((lane & 1) == 0) red_s[wave][32 * mt + oc] = r;          // (the means were consumed before the barrier above; the sums are rewritten after it)
        }
        __syncthreads();
        if (tid < 32 * MT && co0 + tid < Cout) {
            float* rec = stats + (((long long)n * Cout + co0 + tid) * tiles + blockIdx.x) * 3;
            rec[0] = tot;
            rec[1] = mean_s[tid];
            rec[2] = ((red_s[0][tid] + red_s[1][tid]) + (red_s[2][tid] + red_s[3][tid])) + ((red_s[4][tid] + red_s[5][tid]) + (red_s[6][tid] + red_s[7][tid]));
        }
    }
}

// w [Cout][Cin] -> per pass of 64 output channels [k-step][piece][m-tile][kg][32 couts] x 8 input channels fp16, zero padded in both directions, scaled by tail[1]
// (conv3d_k3_h2_scale_kernel).  One thread per (padded cout, padded cin).
__global__ void __launch_bounds__(256)
conv1x1_h2_pack_kernel(const float* __restrict__ w, int Cout, int Cin, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int nks = (Cin + 15) / 16, Kp = 16 * nks;
    const int Cp = (Cout + 63) / 64 * 64;
    const long long id = (long long)blockIdx.x * 256 + threadIdx.x;
    if (id >= (long long)Cp * Kp) return;
    const int k = (int)(id % Kp), co = (int)(id / Kp);
    const float v = (co < Cout && k < Cin) ? w[(long long)co * Cin + k] * tail[1] : 0.0f;
    _Float16 pc[2];
    h2_split(v, pc[0], pc[1]);
    const int pass = co / 64, mt = (co % 64) / 32, r = co % 32, s = k / 16, kgp = (k % 16) / 8, e = k % 8;
#pragma unroll
    for (int p = 0; p < 2; ++p)
        packed[((((long long)pass * nks + s) * C1H_SLAB) + (p * 2 + mt) * 64 + kgp * 32 + r) * 8 + e] = pc[p];
}

}  // namespace mh
