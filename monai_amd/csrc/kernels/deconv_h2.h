// ConvTranspose3d k = 2, s = 2 of act(in) on the fp16 matrix cores in conv3d_h2.h's two-piece split precision (round 5).
//
// Reference op: the up-sampling of UpCat (monai/networks/nets/basic_unet.py:130-178 -> UpSample "deconv"), DynUNet's UnetUpBlock.transp_conv
// (monai/networks/blocks/dynunet_block.py:188-201) and UNETR's transposed convolutions (monai/networks/blocks/unetr_block.py).  Every input voxel owns a disjoint
// 2 x 2 x 2 output block: out[co][2 z + pz][2 y + py][2 x + px] = b[co] + sum_ci W[ci][co][pz][py][px] act(x[ci][z][y][x]) -- ONE GEMM with M = (cout, parity) rows,
// N = input voxels, K = input channels, whose result is stored pixel-shuffled.  deconv_k2s2_kernel (nn_simple.h) evaluates it on the vector ALU with four output
// channels per thread: bound by its 8 Cin Cout multiply-adds per voxel (64 -> 32 channels @ 48^3 x 64 windows: 3.9 ms for 7.2 GB of stores) and re-reading the input
// once per group of four output channels.  On the matrix cores the multiply-adds are a quarter of the store time.
//
// Mapping (conv1x1_h2.h's idea: the activated input goes from HBM STRAIGHT INTO the B operand, no LDS staging).  A wave owns 32 consecutive input voxels = one
// N-tile; lane (r32, kg) loads the 8 channels 16 s + 8 kg .. + 7 of k-step s for voxel r32 (dword loads, 128 contiguous bytes per half wave and channel),
// activates, scales by the sample's power of two and splits them.  M-tile mt = output channels 4 mt .. 4 mt + 3 x 8 parities, row = 8 a + 4 pz + 2 py + px: in the D
// layout lane (voxel r32, kg) then holds, per register quad i >> 2 = a, exactly {pz = kg} x {py} x {px = 0, 1} -- the two fine x of a (cout, fine z, fine y) are one
// 8-byte store, and the 32 lanes of a half wave write 256 contiguous bytes.  A workgroup = 4 waves = 128 voxels x MT M-tiles (MT = 8: 32 output channels, 128
// accumulator registers; 4: 16); the weights of 4 k-steps (64 channels: 64 KB at MT = 8) sit in LDS, two workgroups per CU; more input channels reload the
// buffer per chunk.  The kernel is bound by its stores (4 x the matrix time at 64 -> 32 channels), so there is no software pipelining beyond the next k-step's loads.
// Input range and records: conv3d_h2.h's contract (bounds in the records, power-of-two scale, NaN for a poisoned sample).  Output bound: the workgroup's maximum
// |value| folded into the records of its channels by one atomic instruction (nn_simple.h's rule: the group's maximum bounds each of its channels).
#pragma once
#include "common.h"
#include "conv3d_h2.h"

namespace mh {

constexpr int DH_KS = 4;                                    // k-steps (of 16 input channels) whose weights are in LDS at a time
constexpr int DH_CIN_MAX = 1024;                            // input channels: their {alpha, beta, slope} records sit in LDS

template <int MT>
__global__ void __launch_bounds__(256, 2)
deconv_k2s2_h2_kernel(Tensor in, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias, Tensor out) {
    __shared__ uint4 ws[DH_KS * MT * 128];                  // [k-step][m-tile][piece][kg][32 rows]
    __shared__ float nrm_s[3 * DH_CIN_MAX];
    __shared__ unsigned bound_s[4];
    const int tid = threadIdx.x, lane = tid & 63, r32 = lane & 31, kg = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = in.C, Cout = out.C, Di = in.D, Hi = in.H, Wi = in.W;
    const int cg = blockIdx.y, n = blockIdx.z;
    const long long ivol = (long long)Di * Hi * Wi;
    const int nks = Cin / 16;
    const long long idx0 = (long long)blockIdx.x * 128 + wave * 32 + r32;
    const bool valid = idx0 < ivol;
    const long long idx = valid ? idx0 : ivol - 1;           // lanes past the volume recompute its last voxel and store nothing

    const uint4* wcg = wp + (long long)cg * nks * (MT * 128);
    for (int i = tid; i < min(nks, DH_KS) * MT * 128; i += 256) ws[i] = wcg[i];
    unsigned mb = 0u;
    for (int c = tid; c < Cin; c += 256) {
        const float4 a = *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * c);
        nrm_s[3 * c] = a.x; nrm_s[3 * c + 1] = a.y; nrm_s[3 * c + 2] = a.z;
        const unsigned bb = abs_bits(a.w);
        mb = max(mb, bb == 0u ? 0x7fc00000u : bb);           // no bound given counts as non-finite (conv3d_h2.h)
    }
    mb = wave_umax(mb);
    if (lane == 0) bound_s[wave] = mb;
    __syncthreads();
    mb = max(max(bound_s[0], bound_s[1]), max(bound_s[2], bound_s[3]));
    const bool poisoned = mb >= 0x7f800000u;
    const int e_in = poisoned ? 0 : min(max(15 - ((int)(mb >> 23) - 126), -100), 100);
    const float p_in = __uint_as_float((unsigned)(e_in + 127) << 23);

    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[mt][i] = 0.0f;

    const float* src = in.data + (long long)n * in.n_stride + idx + (long long)(8 * kg) * ivol;
    float x[8], xn[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = src[(long long)j * ivol];
    for (int s = 0; s < nks; ++s) {
        if (s > 0 && s % DH_KS == 0) {                       // more than 64 input channels: the next chunk of the weights replaces the one in LDS
            __syncthreads();
            for (int i = tid; i < min(nks - s, DH_KS) * MT * 128; i += 256) ws[i] = wcg[(long long)s * (MT * 128) + i];
            __syncthreads();
        }
        if (s + 1 < nks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) xn[j] = src[(long long)(16 * (s + 1) + j) * ivol];
        }
        f16x8 bh, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c_ = 16 * s + 8 * kg + j;
            _Float16 h, l;
            h2_split(act(x[j], nrm_s[3 * c_] * p_in, nrm_s[3 * c_ + 1] * p_in, nrm_s[3 * c_ + 2]), h, l);
            bh[j] = h;
            bl[j] = l;
        }
        const uint4* wk = ws + (s % DH_KS) * (MT * 128) + kg * 32 + r32;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f16x8 ah = __builtin_bit_cast(f16x8, wk[mt * 128]), al = __builtin_bit_cast(f16x8, wk[mt * 128 + 64]);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[mt], 0, 0, 0);
        }
        if (s + 1 < nks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = xn[j];
        }
    }

    // scale back: 2^-(weight scale exponent) * 2^-e_in as two power-of-two factors (conv3d_h2.h); a poisoned bound turns the sample's output into NaN
    const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - e_in;
    const int ta = t_ / 2, tb = t_ - ta;
    const float inv_a = poisoned ? __uint_as_float(0x7fc00000u) : __uint_as_float((unsigned)(ta + 127) << 23), inv_b = __uint_as_float((unsigned)(tb + 127) << 23);
    const int xq = (int)(idx % Wi);
    const long long tq = idx / Wi;
    const int yq = (int)(tq % Hi), zq = (int)(tq / Hi);
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const long long ovol = 8 * ivol;
    // register 4 a + 2 py + px of M-tile mt = output channel 4 mt + a at fine (2 z + kg, 2 y + py, 2 x + px)
    float* dst = out.data + (long long)n * out.n_stride + (long long)(cg * 4 * MT) * ovol + ((long long)(2 * zq + kg) * Ho + 2 * yq) * Wo + 2 * xq;
    unsigned m = 0u;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float bv = bias ? bias[cg * 4 * MT + 4 * mt + a] : 0.0f;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
                const f32x2 v = {(acc[mt][4 * a + 2 * py] * inv_a) * inv_b + bv, (acc[mt][4 * a + 2 * py + 1] * inv_a) * inv_b + bv};
                m = max(m, max(abs_bits(v[0]), abs_bits(v[1])));
                if (valid) *reinterpret_cast<f32x2*>(dst + (long long)(4 * mt + a) * ovol + (long long)py * Wo) = v;
            }
        }
    if (out.nrm) {
        m = wave_umax(valid ? m : 0u);
        __syncthreads();                                     // bound_s was read above by every wave
        if (lane == 0) bound_s[wave] = m;
        __syncthreads();                                     // one atomic instruction per WORKGROUP: the waves of a (sample, channel) all target the same records
        m = max(max(bound_s[0], bound_s[1]), max(bound_s[2], bound_s[3]));
        if (tid < 4 * MT && m != 0u) atomicMax(reinterpret_cast<unsigned*>(bound_slot(out, n, cg * 4 * MT + tid)), m);
    }
}

// w [Cin][Cout][2][2][2] -> [cout group of 4 MT][k-step][m-tile][piece][kg][32 rows = 8 a + 4 pz + 2 py + px] x 8 input channels fp16, scaled by tail[1]
// (conv3d_k3_h2_scale_kernel).  One thread per (cin, cout).
__global__ void __launch_bounds__(256)
deconv_k2s2_h2_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int MT, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int co = idx % Cout, ci = idx / Cout;
    const int nks = Cin / 16, G = 4 * MT;
    const int cg = co / G, mt = (co % G) / 4, a = co % 4, ks = ci / 16, kgi = (ci % 16) / 8, j = ci % 8;
    const float s = tail[1];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        _Float16 pc[2];
        h2_split(w[((long long)ci * Cout + co) * 8 + p] * s, pc[0], pc[1]);
        const long long tile = ((long long)cg * nks + ks) * MT + mt;
#pragma unroll
        for (int q = 0; q < 2; ++q) packed[((tile * 2 + q) * 2 + kgi) * 256LL + (8 * a + p) * 8 + j] = pc[q];
    }
}

}  // namespace mh
