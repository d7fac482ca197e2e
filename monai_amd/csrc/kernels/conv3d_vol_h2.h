// Conv3d 3x3x3 / stride 1 / zero padding 1 for SMALL volumes (D H W <= 256 voxels: the 6^3 level of a 96^3 window) on the fp16 matrix cores in conv3d_h2.h's two-piece
// split precision (round 5).
//
// Reference op: the same nn.Conv3d of `Convolution` (monai/networks/blocks/convolutions.py:98-171) that conv3d_h2.h serves, fed by the previous block's deferred
// InstanceNorm + LeakyReLU.  conv3d_h2.h marches 16 x 16 (y, x) regions along z: a 6 x 6 plane fills 14 % of a region, so the selector kept such levels on the exact-fp32
// tiles (conv3d_mfma.h, 55-75 TFLOP/s: 2.9 % of the BasicUNet step, the bottleneck of the nnU-Net shape, SwinUNETR's deep levels).  Here the WHOLE volume of one sample
// is the tile: M = the flattened voxel index z H W + y W + x (216 of 256 rows used at 6^3), N = 32 NCG output channels per workgroup, K = 16 channels per instruction --
// no z-march, no rotating accumulator sets, no halo recomputation, and the output index of a voxel IS its M index (16-byte stores of four consecutive voxels).
// A workgroup (8 waves, a 32-voxel M block each) owns (sample, group of 32 NCG output channels).  Per 16-channel chunk the zero-padded volume (D + 2)(H + 2)(W + 2) <= 512
// cells is staged once -- four dword loads per (cell, channel quad) task, activated under the records (LDS, pre-multiplied by the sample's power of two), split, 8 bytes
// per piece -- into one of two buffers and read by all 27 taps as shifted 16-byte operand reads (tap offset (kz (H + 2) + ky)(W + 2) + kx cells); a STEP = (chunk, kz)
// = 9 tap matrices (4 KB NCG each) copied global -> registers -> LDS a step ahead, one barrier per step.  Epilogue: exact power-of-two scale-back, bias, stores,
// InstanceNorm statistics {count, mean, M2} -- ONE record per (n, cout): the workgroup holds the whole plane set.
// Input range and records: conv3d_h2.h's contract (bounds in the records, power-of-two scale, NaN for a poisoned sample).
#pragma once
#include "common.h"
#include "conv3d_h2.h"

namespace mh {

constexpr int HV_NT = 512;
constexpr int HV_CELLS = 512;                              // cells of the zero-padded volume
constexpr int HV_XB = 4 * HV_CELLS;                        // uint4 per operand buffer: [piece][k group][cell]
constexpr int HV_XSLOTS = 4;                               // (cell, channel quad) tasks per thread and chunk: 4 x 512 = 512 cells x 4 quads
constexpr int HV_CIN_MAX = 768;                            // input channels whose records sit in LDS (9 KB: with the two operand and the two tap buffers 156 of the 160 KB)
constexpr unsigned HV_DROP = 0x80000000u;

template <int NCG, bool STATS>
__global__ void __launch_bounds__(HV_NT, 1)
conv3d_k3_vol_h2_kernel(Tensor in, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias, Tensor out, float* __restrict__ stats) {
    constexpr int CW = 32 * NCG;                            // couts per workgroup
    constexpr int WT = 9 * 4 * CW;                          // uint4 of a step's 9 tap matrices: [tap][piece][k group][cout]
    constexpr int WSLOTS = (WT + HV_NT - 1) / HV_NT;
    constexpr int WB = WSLOTS * HV_NT;                      // uint4 per weight buffer (one cell per copy slot: no bound check)
    __shared__ uint4 xbuf[2 * HV_XB];
    __shared__ uint4 wbuf[2 * WB];
    __shared__ float nrm_s[3 * HV_CIN_MAX];
    __shared__ unsigned bound_s[HV_NT / 64];
    __shared__ float red[(HV_NT / 64) * 32 * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = in.C, Cout = out.C, D = in.D, H = in.H, W = in.W;
    const int HW = H * W, vol = D * HW, PH = H + 2, PW = W + 2, pcells = (D + 2) * PH * PW;
    const int nch = Cin / 16;
    const unsigned ncg = (unsigned)(Cout / CW);
    const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(lid % ncg), n = (int)(lid / ncg);

    // ---- records -> LDS, the sample's input scale 2^e_in from their bounds (conv3d_h2.h)
    unsigned mb = 0u;
    for (int c = tid; c < Cin; c += HV_NT) {
        const float4 a = *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * c);
        nrm_s[3 * c] = a.x; nrm_s[3 * c + 1] = a.y; nrm_s[3 * c + 2] = a.z;
        const unsigned bb = abs_bits(a.w);
        mb = max(mb, bb == 0u ? 0x7fc00000u : bb);           // no bound given counts as non-finite
    }
    mb = wave_umax(mb);
    if (lane == 0) bound_s[wave] = mb;
    __syncthreads();
    mb = bound_s[0];
#pragma unroll
    for (int w = 1; w < HV_NT / 64; ++w) mb = max(mb, bound_s[w]);
    const bool poisoned = mb >= 0x7f800000u;
    const int e_in = poisoned ? 0 : min(max(15 - ((int)(mb >> 23) - 126), -100), 100);
    {
        const float p_ = __uint_as_float((unsigned)(e_in + 127) << 23);
        for (int c = tid; c < Cin; c += HV_NT) { nrm_s[3 * c] *= p_; nrm_s[3 * c + 1] *= p_; }      // each thread rescales the records it wrote
    }

    // ---- staging tasks of this thread: (padded cell, channel quad) -> four dword loads; activate + scale + split; 8 bytes of each piece
    unsigned xoff[HV_XSLOTS];
    int xcell[HV_XSLOTS];
#pragma unroll
    for (int s = 0; s < HV_XSLOTS; ++s) {
        const int t = tid + HV_NT * s;                       // < 2048 = 512 cells x 4 quads
        const int q = t >> 9, cell = t & 511;
        const int pz = cell / (PH * PW), r_ = cell - pz * (PH * PW), py = r_ / PW, px = r_ - py * PW;
        const int gz = pz - 1, gy = py - 1, gx = px - 1;
        const bool ok = cell < pcells && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        xoff[s] = ok ? 4u * (unsigned)((4 * q) * vol + gz * HW + gy * W + gx) : HV_DROP;
        xcell[s] = (((q >> 1) * HV_CELLS + cell) * 2 + (q & 1));
    }
    const float* fsrc = in.data + (long long)n * in.n_stride;
    const long long frest = (long long)(in.N - n) * in.n_stride * 4;
    const uint4* wcg = wp + (long long)cg * nch * 27 * 4 * CW;
    float xraw[HV_XSLOTS][4];
    u32x4 wreg[WSLOTS];
    auto load_x = [&](int ch) {
        const long long fo_ = (long long)(16 * ch) * vol;
        const long long left_ = frest - fo_ * 4;
        const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fsrc) + fo_, 0, (int)(left_ < 0x7fffffffLL ? left_ : 0x7fffffffLL), 0x00020000);
#pragma unroll
        for (int s = 0; s < HV_XSLOTS; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) xraw[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xoff[s], (unsigned)(i * vol * 4), 0));
    };
    auto convert_store = [&](int bufi, int ch) {
        u32x2* xh = reinterpret_cast<u32x2*>(xbuf + bufi * HV_XB);
#pragma unroll
        for (int s = 0; s < HV_XSLOTS; ++s) {
            const int q = (tid + HV_NT * s) >> 9;
            const bool keep = xoff[s] != HV_DROP;
            _Float16 h_[4], l_[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = 16 * ch + 4 * q + i;
                const float ya = act(xraw[s][i], nrm_s[3 * c], nrm_s[3 * c + 1], nrm_s[3 * c + 2]);
                h2_split(keep ? ya : 0.0f, h_[i], l_[i]);   // zero padding is zero AFTER the activation
            }
            const f16x2 h01 = {h_[0], h_[1]}, h23 = {h_[2], h_[3]}, l01 = {l_[0], l_[1]}, l23 = {l_[2], l_[3]};
            xh[xcell[s]] = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
            xh[xcell[s] + 4 * HV_CELLS] = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
        }
    };
    auto load_w = [&](int ch, int kz) {
        const auto wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wcg) + ((long long)ch * 27 + kz * 9) * 4 * CW, 0, WT * 16, 0x00020000);
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) wreg[s] = __builtin_amdgcn_raw_buffer_load_b128(wr, 16u * (unsigned)(tid + HV_NT * s), 0, 0);
    };
    auto store_w = [&](int bufi) {
        u32x4* wd = reinterpret_cast<u32x4*>(wbuf + bufi * WB);
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) wd[tid + HV_NT * s] = wreg[s];
    };

    // ---- operands of this lane: A = voxel m = 32 wave + (lane & 31) at the corner of its 3 x 3 x 3 neighbourhood in padded cells, k group = lane >> 5; B = cout (lane & 31)
    const int r32 = lane & 31, kg = lane >> 5;
    const int m_a = min(wave * 32 + r32, vol - 1);           // M rows beyond the volume compute on the last voxel's cells: dropped later
    const int az = m_a / HW, ar = m_a - az * HW, ay = ar / W, ax = ar - ay * W;
    const int abase = kg * HV_CELLS + (az * PH + ay) * PW + ax;
    const int bbase = kg * CW + r32;

    f32x16 acc[NCG];
#pragma unroll
    for (int g = 0; g < NCG; ++g)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[g][i] = 0.0f;

    load_x(0);
    load_w(0, 0);
    __syncthreads();                                          // the rescaled records are complete
    convert_store(0, 0);
    store_w(0);
    __syncthreads();
    const int T = 3 * nch;
    for (int s = 0; s < T; ++s) {
        const int ch = s / 3, kz = s - 3 * ch;
        const bool has_next = s + 1 < T, stage_x = kz == 0 && ch + 1 < nch;
        if (has_next) load_w((s + 1) / 3, (s + 1) % 3);
        if (stage_x) load_x(ch + 1);
        const uint4* xb = xbuf + (ch & 1) * HV_XB + abase + kz * (PH * PW);
        const uint4* wb = wbuf + (s & 1) * WB + bbase;
        // 9 groups (ky, kx) of 3 NCG matrix instructions; group G + 1's operands are fetched into a second register set while group G multiplies (a wave issues in order:
        // an operand read right in front of its use costs the LDS latency every time, conv3d_h2.h)
        uint4 aq[2][2], bq[2][NCG][2];
        auto fetch = [&](int G) {
            const int ky = G / 3, kx = G - 3 * ky;
            const uint4* ap = xb + ky * PW + kx;
            aq[G & 1][0] = ap[0];
            aq[G & 1][1] = ap[2 * HV_CELLS];
#pragma unroll
            for (int g = 0; g < NCG; ++g) {
                const uint4* bp = wb + (G * 4) * CW + g * 32;
                bq[G & 1][g][0] = bp[0];
                bq[G & 1][g][1] = bp[2 * CW];
            }
        };
        fetch(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int G = 0; G < 9; ++G) {
            if (G + 1 < 9) fetch(G + 1);
            const f16x8 ah = __builtin_bit_cast(f16x8, aq[G & 1][0]), al = __builtin_bit_cast(f16x8, aq[G & 1][1]);
#pragma unroll
            for (int g = 0; g < NCG; ++g) {
                const f16x8 bh = __builtin_bit_cast(f16x8, bq[G & 1][g][0]), bl = __builtin_bit_cast(f16x8, bq[G & 1][g][1]);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < 3 * NCG; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NCG == 1 ? 2 : 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (has_next) store_w((s + 1) & 1);
        if (stage_x) convert_store((ch + 1) & 1, ch + 1);
        __syncthreads();
    }

    // ---- epilogue: register 4 j + i of an accumulator = voxel 32 wave + 8 j + 4 kg + i, lane & 31 = cout
    float inv_a, inv_b;
    {
        const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - e_in;
        const int t1_ = t_ / 2, t2_ = t_ - t1_;
        inv_a = poisoned ? __uint_as_float(0x7fc00000u) : __uint_as_float((unsigned)(t1_ + 127) << 23);
        inv_b = __uint_as_float((unsigned)(t2_ + 127) << 23);
    }
    const bool vec = (vol & 3) == 0;
#pragma unroll
    for (int g = 0; g < NCG; ++g) {
        const int co = cg * CW + g * 32 + r32;
        const float bco = bias ? bias[co] : 0.0f;
        float* dst = out.data + (long long)n * out.n_stride + (long long)co * vol;
        float psum = 0.0f, pcnt = 0.0f;
        f32x4 o_[4];
        int nval[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m0 = wave * 32 + 8 * j + 4 * kg;
            nval[j] = min(max(vol - m0, 0), 4);
            f32x4 v = {acc[g][4 * j], acc[g][4 * j + 1], acc[g][4 * j + 2], acc[g][4 * j + 3]};
            v = v * inv_a * inv_b + bco;
            o_[j] = v;
            if (vec) {
                if (nval[j] == 4) *reinterpret_cast<f32x4*>(dst + m0) = v;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < nval[j]) dst[m0 + i] = v[i];
            }
            if (STATS) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float w_ = i < nval[j] ? 1.0f : 0.0f;
                    pcnt += w_;
                    psum += v[i] * w_;
                }
            }
        }
        if (STATS) {
            const float pmean = pcnt > 0.0f ? psum / (pcnt > 0.0f ? pcnt : 1.0f) : 0.0f;
            float pm2 = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d_ = o_[j][i] - pmean;
                    pm2 += i < nval[j] ? d_ * d_ : 0.0f;
                }
            Stat r_, ot;
            r_.n = pcnt; r_.mean = pmean; r_.m2 = pm2;
            ot.n = __shfl_xor(r_.n, 32);
            ot.mean = __shfl_xor(r_.mean, 32);
            ot.m2 = __shfl_xor(r_.m2, 32);
            r_ = kg == 0 ? stat_merge(r_, ot) : stat_merge(ot, r_);
            if (kg == 0) { red[(wave * 32 + r32) * 3] = r_.n; red[(wave * 32 + r32) * 3 + 1] = r_.mean; red[(wave * 32 + r32) * 3 + 2] = r_.m2; }
            __syncthreads();
            if (tid < 32) {
                Stat st;
                st.n = 0.0f; st.mean = 0.0f; st.m2 = 0.0f;
#pragma unroll
                for (int w = 0; w < HV_NT / 64; ++w) {
                    Stat o2;
                    o2.n = red[(w * 32 + tid) * 3]; o2.mean = red[(w * 32 + tid) * 3 + 1]; o2.m2 = red[(w * 32 + tid) * 3 + 2];
                    st = stat_merge(st, o2);
                }
                float* rec = stats + ((long long)n * Cout + cg * CW + g * 32 + tid) * 3;
                rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
            }
            __syncthreads();
        }
    }
}

// w [Cout][Cin][3][3][3] -> [cout group of 32 NCG][chunk][27 taps kz ky kx][piece][k group][32 NCG couts][8 channels] fp16, scaled by tail[1]
// (conv3d_k3_h2_scale_kernel).  One thread per (cout, cin).
__global__ void __launch_bounds__(256)
conv3d_k3_vol_h2_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int ncgw, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const int CW = 32 * ncgw, nch = Cin / 16;
    const float s = tail[1];
    const int cg = co / CW, col = co % CW, chunk = ci / 16, kgi = (ci % 16) / 8, j = ci % 8;
    for (int tap = 0; tap < 27; ++tap) {
        _Float16 pc[2];
        h2_split(w[((long long)co * Cin + ci) * 27 + tap] * s, pc[0], pc[1]);
        _Float16* mat = packed + (((long long)(cg * nch + chunk) * 27 + tap) * 4 * CW) * 8LL;
#pragma unroll
        for (int q = 0; q < 2; ++q) mat[((q * 2 + kgi) * CW + col) * 8 + j] = pc[q];
    }
}

}  // namespace mh
