// Conv3d 3x3x3 / stride 1 / zero padding 1 on the FP16 matrix cores of gfx950: the split-precision arithmetic of conv3d_h2.h behind a
// Winograd F(2, 3) minimal-filtering transform ALONG Z -- 18 instead of 27 multiply-adds per (voxel, cin, cout), i.e. 2 instead of 3 fp16
// matrix instructions per fp32 multiply-add of the direct convolution (VERDICT r03 "Next round" item 2).
//
// Same reference op and "normalise on load" contract as conv3d_h2.h / conv3d_mfma.h (nn.Conv3d of `Convolution`,
// monai/networks/blocks/convolutions.py:98-171, fed by the previous block's deferred InstanceNorm + LeakyReLU).
//
// Arithmetic.  For a PAIR of output planes z0, z0 + 1 and the four activated input planes a, b, c, d = z0 - 1 .. z0 + 2 (zero outside the volume):
//     V0 = a - c      V1 = b + c      V2 = c - b      V3 = b - d                     (fp32, plane-wise: no tile structure, no halo exchange)
//     U0 = g0         U1 = (g0 + g1 + g2) / 2         U2 = (g0 - g1 + g2) / 2        U3 = g2      (per (ky, kx, cin, cout), g_k = the z-taps; packed once)
//     m_j = sum over (ky, kx, cin) of V_j(y + ky - 1, x + kx - 1, cin) * U_j(ky, kx, cin, cout)     (9 in-plane taps, DIRECT -- the GEMM of conv3d_h2.h per position)
//     Y(z0) = (m0 + m1) + m2      Y(z0 + 1) = (m1 - m2) - m3
// Every product V * U is evaluated in two-piece fp16 split precision exactly as in conv3d_h2.h (x = hi + lo, hi*hi + lo*hi + hi*lo, each piece product exact in
// fp32, fp32 accumulation in v_mfma_f32_32x32x16_f16); the transforms are fp32 sums of two / three values (F(2, 3) has no constants but 1/2: the error of the
// fp32 Winograd kernel conv3d_wino2p.h, 2.3e-6 on the logits, tools/winograd_numerics.py).  Range: the input scale puts the largest record bound below 2^14
// (conv3d_h2.h: 2^15), one bit of head room for V = a sum of two; the weight scale leaves |U| <= 1.5 * 2^13.
//
// Why along z and not in the plane: the in-plane form F(2x2, 3x3) needs 16 transform positions x three live output planes of
// accumulators (z-streaming) -- 192 KB of registers per 32 tiles -- and 96 KB of transformed weights per 16-channel step, i.e. 8x the weight stream
// per matrix cycle of conv3d_h2.h; with z INSIDE the minimal-filtering tile the two output planes are stationary (4 positions x 16 registers),
// the transform is a subtraction of two registers, the GEMM M dimension stays "32 voxels of a row" and the epilogue stays in-lane.  The price: each pair
// reads four input planes (2x the input loads of the direct kernel; L2 hits) and the A operand is no longer shared by three z-taps (4 LDS reads per
// 3 MFMAs instead of 8 per 9: 167 of the LDS's 256 B/clk at a full matrix pipe).
//
// Mapping.  A workgroup = 8 waves owns an 8 x 32 (y, x) region x 32 couts x one z-chunk (an even number of planes) and marches pair by pair; wave w owns row w
// of the region = one 32-voxel M block, N = 32 couts, K = 16 input channels per instruction.  A SUB-STEP = two positions (V0, V1 | V2, V3) x 16 channels: the two
// transformed planes [2 positions][2 pieces][2 k-groups][10 x 34 voxels + 4 dump cells][8 ch] (43 KB) and their weight slabs [2][2 pieces][9 taps][2][32 couts][8 ch]
// (36 KB) sit in one of two LDS buffers (158 KB + the input records); one barrier per sub-step of 54 MFMAs per wave, which alternate between the two
// positions' accumulators (consecutive matrix instructions never depend on each other).  The staging work -- activation, the z-transform, the hi / lo split, LDS
// writes, the weight stream, the loads of the next pair's planes -- is branch-free and dealt out over the MFMA gaps by sched_group_barrier, as in conv3d_h2.h.
// Raw planes live in three register sets (b, c and a|d): S0 converts V2, V3 from (b, c, d) and then loads (a', b', c') of the next (pair, chunk); S1 converts
// V0', V1' and then loads d'.  A plane outside the volume is loaded through a buffer descriptor of zero records (zeros, no traffic) with its records zeroed.
// The completed pair: Y(z0) -> 16 registers, Y(z0 + 1) in place in m3's registers; their scale-back / bias / stores / statistics ride in the first sub-step of
// the next pair (taps 1-4 and 5-8), exactly the four pieces of conv3d_h2.h.
#pragma once
#include "common.h"
#include "conv3d_h2.h"

namespace mh {

constexpr int HZ_BY = 8, HZ_BX = 32;                        // region of a workgroup (rows x columns): a wave = one row of 32 voxels
constexpr int HZ_RY = HZ_BY + 2, HZ_RX = HZ_BX + 2;         // staged rows, columns (= LDS row pitch: consecutive 16-byte cells of a row are bank-conflict free)
constexpr int HZ_NV = HZ_RY * HZ_RX, HZ_HALF = HZ_NV / 2;   // staged voxels 340, per staging half-workgroup 170 <= 64 * 3 slots
constexpr int HZ_PV = HZ_NV + 4;                            // cells of a staged plane: + 4 dump cells (zero padding / idle lanes write there)
constexpr int HZ_XV = 2 * HZ_PV;                            // uint4 per piece of a position: [k-group][cell]
constexpr int HZ_XP = 2 * HZ_XV;                            // uint4 per position (two pieces): 1376
constexpr int HZ_WV = 9 * 2 * H2_CN;                        // uint4 per piece of a position's weight slab: [tap][k-group][cout] = 576
constexpr int HZ_WP = 2 * HZ_WV;                            // uint4 per position (two pieces): 1152
constexpr int HZ_BUF = 2 * HZ_XP + 2 * HZ_WP;               // uint4 per LDS buffer (two positions): 5056 = 80 896 bytes
constexpr int HZ_WSUB = 2 * HZ_WP;                          // uint4 of weights per sub-step: 2304 = 4 x 512 uint4 + 512 uint2
constexpr int HZ_SLOTS = 3;                                 // staging tasks per lane: (voxel, 4 channels)
constexpr int HZ_NRM_MAX = 128;                             // input channels (their records {alpha, beta, slope, K} x 4 per quad sit in LDS: 2 KB -- all 160 KB are taken)
static_assert(HZ_HALF * 2 == HZ_NV && HZ_HALF <= 64 * HZ_SLOTS, "staging slots");
static_assert(2 * HZ_BUF * 16 + 4 * HZ_NRM_MAX * 4 <= 160 * 1024, "LDS budget");

template <bool STATS, bool NRM>
__global__ void __launch_bounds__(512, 1)
conv3d_k3_h2z_kernel(Tensor in, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias, Tensor out,
                     float* __restrict__ stats, int bxn, int byn, int zchunk, unsigned nblk) {
    __shared__ uint4 smem[2 * HZ_BUF];
    __shared__ __attribute__((aligned(16))) float nrm_s[NRM ? 4 * HZ_NRM_MAX : 4];      // per channel quad {alpha x 4, beta x 4, slope x 4, K x 4} (K: the activation's med3 constant)
    unsigned* const bound_s = reinterpret_cast<unsigned*>(smem + 2 * HZ_XP);      // 8 words of the (not yet loaded) weight area of buffer 0
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = in.C, Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long HW = (long long)H * W, DHW = (long long)D * HW;
    const int NCH = Cin / H2_KC;

    // launch geometry of conv3d_h2.h: 1-D over (window, region, cout group), cout group fastest, XCD-aware
    const unsigned ncg = (unsigned)(Cout / H2_CN);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(lid % ncg);
    lid /= ncg;
    const unsigned b = lid % nblk;
    const int n = (int)(lid / nblk);
    const int x0 = (int)(b % bxn) * HZ_BX, y0 = (int)((b / bxn) % byn) * HZ_BY;
    const int zs = (int)(b / (bxn * byn)) * zchunk, ze = min(zs + zchunk, D);
    const int NP = (ze - zs + 1) >> 1;                        // pairs of output planes this workgroup computes (the last may hold one plane)

    // input staging (conv3d_h2.h): wave w converts channels 4q .. 4q+3 (q = w >> 1) of the step for voxels (w & 1) * 170 + lane + 64 j
    const int q = wave >> 1;
    unsigned soff[HZ_SLOTS];          // BYTE offsets into a channel plane
    int loff[HZ_SLOTS];               // destination in units of 8 bytes inside a piece
#pragma unroll
    for (int j = 0; j < HZ_SLOTS; ++j) {
        const int e0 = lane + 64 * j;
        const int e = min((wave & 1) * HZ_HALF + e0, HZ_NV - 1);
        const int ly = e / HZ_RX, lx = e - ly * HZ_RX;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        const bool ok = e0 < HZ_HALF && gy >= 0 && gy < H && gx >= 0 && gx < W;
        soff[j] = ok ? 4u * (unsigned)(gy * W + gx) : 0u;
        loff[j] = ((q >> 1) * HZ_PV + (ok ? e : HZ_NV + (lane & 3))) * 2 + (q & 1);
    }
    for (int i = tid; i < 2 * HZ_BUF; i += 512) smem[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();                  // (bound_s aliases the zeroed area)
    if (NRM) {
        unsigned mb = 0u;
        for (int c = tid; c < Cin; c += 512) {
            const float4 a = *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * c);
            float* r_ = nrm_s + 16 * (c >> 2) + (c & 3);
            r_[0] = a.x; r_[4] = a.y; r_[8] = a.z;
            r_[12] = __uint_as_float(a.z <= 1.0f ? 0x7f800000u : 0xff800000u);
            const unsigned bb = abs_bits(a.w);
            mb = max(mb, bb == 0u ? 0x7fc00000u : bb);        // no bound given counts as non-finite
        }
        mb = wave_umax(mb);
        if (lane == 0) bound_s[wave] = mb;
    }
    __syncthreads();
    // input scale 2^e_in from the largest bound of the sample: bound * 2^e_in < 2^14 (one bit of head room for the transform's sums of two)
    int e_in = 0;
    bool poisoned = false;
    if (NRM) {
        unsigned mb = bound_s[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) mb = max(mb, bound_s[w]);
        poisoned = mb >= 0x7f800000u;
        e_in = poisoned ? 0 : min(max(14 - ((int)(mb >> 23) - 126), -100), 100);
        const float p_ = __uint_as_float((unsigned)(e_in + 127) << 23);
        __syncthreads();              // every thread has read the bounds: the words go back to zero before the weight area is used
        if (tid < 8) bound_s[tid] = 0u;
        for (int c = tid; c < Cin; c += 512) { float* r_ = nrm_s + 16 * (c >> 2) + (c & 3); r_[0] *= p_; r_[4] *= p_; }
        __syncthreads();
    }

    const float* const src = in.data + (long long)n * in.n_stride + (long long)(4 * q) * DHW;
    const u32x4* const wg4 = reinterpret_cast<const u32x4*>(wp) + (long long)cg * NCH * (2 * HZ_WSUB) + tid;
    const u32x2* const wg2 = reinterpret_cast<const u32x2*>(reinterpret_cast<const u32x4*>(wp) + (long long)cg * NCH * (2 * HZ_WSUB) + 2048) + tid;

    const float m1 = h2_minus_one();
    // raw planes of the current (pair, chunk): three register sets -- rb = b, rc = c, rad = a (while V0 is formed) | d (while V3 is formed)
    float rb[HZ_SLOTS][4], rc[HZ_SLOTS][4], rad[HZ_SLOTS][4];
    u32x4 win[4];
    u32x2 win2;
    f32x4 nq_a = {1.0f, 1.0f, 1.0f, 1.0f}, nq_b = {0.0f, 0.0f, 0.0f, 0.0f}, nq_s = nq_a, nq_k = nq_a;      // records of the quad being converted (chunk of the planes in the registers)

    // state of the load stream: `ld_i` = index of the (pair, chunk) whose planes are loaded next; ld_ok = it exists
    int ld_t = 0, ld_c = 0;           // pair, chunk of the planes to load next
    // ---- staging pieces (all branch-free) ----
    // plane z of (pair ld_t, chunk ld_c) into register set R_: a descriptor of zero records for a plane outside the volume / beyond the last body -> zeros, no traffic
#define MH_HZ_LDPLANE(R_, DZ_)                                                                        \
    {                                                                                                 \
        const int z_ = zs + 2 * ld_t + (DZ_);                                                         \
        const bool v_ = ld_t < NP && z_ >= 0 && z_ < D;                                               \
        const float* p_ = src + (long long)ld_c * (H2_KC * DHW) + (long long)min(max(z_, 0), D - 1) * HW; \
        const auto xr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p_), 0, v_ ? 0x7fffffff : 0, 0x00020000); \
        _Pragma("unroll") for (int j = 0; j < HZ_SLOTS; ++j)                                          \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
                R_[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr_, soff[j], (unsigned)(i * DHW * 4), 0)); \
    }
    // validity of plane DZ_ of the pair whose planes are in the registers (`cv_t`), wave-uniform: a plane outside the volume loads zeros and its activation is
    // replaced by zero (a select on a scalar condition: no register)
#define MH_HZ_VALID(DZ_) (cv_t < NP && zs + 2 * cv_t + (DZ_) >= 0 && zs + 2 * cv_t + (DZ_) < D)
    // activation: y = fma(x, alpha, beta); y > 0 ? y : y * slope  ==  med3(y, y * slope, K) with K = +inf for slope <= 1 (the larger of the two), -inf for slope > 1
    // (the smaller) -- two instructions instead of three; the value is one of y, y * slope either way (for y = +-0 and a NEGATIVE slope the zero's sign may differ)
#define MH_HZ_ACT(R_, J, FV_)                                                                         \
    MH_HZ_ACT_IMPL(R_, J, FV_)
#ifdef HZX_OLD_ACT        /* development switch: compare + select */
#define MH_HZ_ACT_IMPL(R_, J, FV_)                                                                    \
    if (NRM) { _Pragma("unroll") for (int i = 0; i < 4; ++i) { const float y_ = act(R_[J][i], nq_a[i], nq_b[i], nq_s[i]); R_[J][i] = (FV_) ? y_ : 0.0f; } }
#else
#define MH_HZ_ACT_IMPL(R_, J, FV_)                                                                    \
    if (NRM) {      /* a plane outside the volume: its loads returned zeros and its records are zeroed -> fma(0, 0, 0) = 0 -> med3(0, 0, K) = 0 */ \
        const float v_ = (FV_) ? 1.0f : 0.0f;                                                         \
        const f32x4 ma_ = nq_a * v_, mb_ = nq_b * v_;                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { const float y_ = fmaf(R_[J][i], ma_[i], mb_[i]); R_[J][i] = __builtin_amdgcn_fmed3f(y_, y_ * nq_s[i], nq_k[i]); } \
    }
#endif
#ifdef HZX_OLD_SPLIT      /* development switch (tools/ubench/h2z_variants.hip): the four-instruction split */
#define MH_HZ_SPLIT2(A_, B_, H_, L_) { _Float16 h0_, h1_, l0_, l1_; h2_split(A_, h0_, l0_); h2_split(B_, h1_, l1_); H_ = f16x2{h0_, h1_}; L_ = f16x2{l0_, l1_}; }
#else
#define MH_HZ_SPLIT2(A_, B_, H_, L_) h2_split_pair(A_, B_, m1, H_, L_);
#endif
    // split 4 channels of a voxel and write 8 bytes of the high plane, 8 of the low one of position P_ in the buffer after bcur
#define MH_HZ_PUT(P_, J, V_)                                                                          \
    {                                                                                                 \
        u32x2* xh_ = reinterpret_cast<u32x2*>(smem + (bcur ^ 1) * HZ_BUF + (P_) * HZ_XP);             \
        f16x2 h01_, h23_, l01_, l23_;                                                                 \
        MH_HZ_SPLIT2(V_[0], V_[1], h01_, l01_) MH_HZ_SPLIT2(V_[2], V_[3], h23_, l23_)                 \
        xh_[loff[J]] = u32x2{__builtin_bit_cast(unsigned, h01_), __builtin_bit_cast(unsigned, h23_)}; \
        xh_[loff[J] + 2 * HZ_XV] = u32x2{__builtin_bit_cast(unsigned, l01_), __builtin_bit_cast(unsigned, l23_)}; \
    }
    // S0's conversion, slot J: d activated, V2 = c - b -> position 0, V3 = b - d -> position 1 of the other buffer
#define MH_HZ_CONV23(J)                                                                               \
    {                                                                                                 \
        MH_HZ_ACT(rad, J, fv_d)                                                                       \
        float v2_[4], v3_[4];                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { v2_[i] = rc[J][i] - rb[J][i]; v3_[i] = rb[J][i] - rad[J][i]; } \
        MH_HZ_PUT(0, J, v2_) MH_HZ_PUT(1, J, v3_)                                                      \
    }
    // S1's conversion, slot J: a', b', c' activated (b', c' stay in their registers for V2, V3), V0 = a - c -> position 0, V1 = b + c -> position 1
#define MH_HZ_CONV01(J)                                                                               \
    {                                                                                                 \
        MH_HZ_ACT(rad, J, fv_a) MH_HZ_ACT(rb, J, fv_b) MH_HZ_ACT(rc, J, fv_c)                          \
        float v0_[4], v1_[4];                                                                         \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) { v0_[i] = rad[J][i] - rc[J][i]; v1_[i] = rb[J][i] + rc[J][i]; } \
        MH_HZ_PUT(0, J, v0_) MH_HZ_PUT(1, J, v1_)                                                      \
    }
    // records of the quad of chunk C_ (alpha, beta pre-multiplied by 2^e_in)
#define MH_HZ_NRMLD(C_)                                                                               \
    if (NRM) {                                                                                        \
        const f32x4* a_ = reinterpret_cast<const f32x4*>(nrm_s + 16 * (4 * (C_) + q));                \
        nq_a = a_[0]; nq_b = a_[1]; nq_s = a_[2]; nq_k = a_[3];                                       \
    }
    // weights of sub-step index W_ (= 2 * chunk + s) of this cout group: 4 x 16 bytes + 8 bytes per thread, registers -> the buffer after bcur
#define MH_HZ_LDW(W_)                                                                                 \
    {       /* the slab index goes through a value barrier: for a pair's first chunk it is a constant, and loop-invariant loads hoisted out of the march would hold 18 registers for its whole length */ \
        int wi_ = (W_);                                                                               \
        MH_OPAQUE_S(wi_);                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) win[j] = wg4[(long long)wi_ * HZ_WSUB + 512 * j]; \
        win2 = wg2[(long long)wi_ * (2 * HZ_WSUB)];                                                   \
    }
#define MH_HZ_WST                                                                                     \
    {                                                                                                 \
        u32x4* w4_ = reinterpret_cast<u32x4*>(smem + (bcur ^ 1) * HZ_BUF + 2 * HZ_XP);                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) w4_[tid + 512 * j] = win[j];                    \
        reinterpret_cast<u32x2*>(w4_ + 2048)[tid] = win2;                                             \
    }

    // operands of this lane: A = voxel (row w, x = lane & 31), B = cout (lane & 31); k-group = lane >> 5
    const int r32 = lane & 31, kg = lane >> 5;
    const int abase = kg * HZ_PV + wave * HZ_RX + r32;
    const int bbase = kg * H2_CN + r32;

    f32x16 acc[4], acce;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[s][i] = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acce[i] = 0.0f;

    // epilogue (conv3d_h2.h, WIDE geometry): lane = cout r32; register 4 j + i = voxel x = 8 j + 4 kg + i of the wave's row
    const int co = cg * H2_CN + r32;
    const float bco = bias ? bias[co] : 0.0f;
    float inv_a, inv_b;
    {
        const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - e_in;      // wtail[1] = the weights' power-of-two scale
        const int t1_ = t_ / 2, t2_ = t_ - t1_;
        inv_a = poisoned ? __uint_as_float(0x7fc00000u) : __uint_as_float((unsigned)(t1_ + 127) << 23);
        inv_b = __uint_as_float((unsigned)(t2_ + 127) << 23);
    }
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out.data + (long long)n * out.n_stride + (long long)(cg * H2_CN) * DHW, 0, (int)(H2_CN * DHW * 4), 0x00020000);
    constexpr unsigned HZ_DROP = 0x80000000u;
    unsigned ooff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xg_ = 8 * j + 4 * kg;
        const bool ok_ = y0 + wave < H && x0 + xg_ < W;
        ooff[j] = ok_ ? 4u * (unsigned)((long long)r32 * DHW + (long long)(y0 + wave) * W + x0 + xg_) : HZ_DROP;
    }
    float esum_ = 0.0f, ecnt_ = 0.0f, em2_ = 0.0f, emean_ = 0.0f;
    Stat run;
    run.n = 0.0f; run.mean = 0.0f; run.m2 = 0.0f;
    int pend = 0, pend_z = 0;         // acce holds a completed plane (pend_z) waiting for its epilogue
    int pend1 = 0, pend1_z = 0;       // acc[3] holds the second plane of the previous pair (moved to acce in tap 5 of the next sub-step 0)

    // operands: A (the transformed voxels) of the next tap is read one tap ahead into the other register set; B (the weights) has ONE set, refilled inside
    // the tap as soon as its last matrix instruction has issued -- high pieces after the fourth, low pieces after the sixth (256 registers per wave)
    uint4 ah[2][2], al[2][2], bh[2], bl[2];                // A: [operand buffer][position]; B: [position]
#define MH_HZ_FETCH_A(OB, T_)                                                                         \
    {                                                                                                 \
        constexpr int aoff_ = ((T_) / 3) * HZ_RX + (T_) % 3;                                          \
        const uint4* xb_ = smem + bcur * HZ_BUF + abase + aoff_;                                      \
        _Pragma("unroll") for (int p = 0; p < 2; ++p) { ah[OB][p] = xb_[p * HZ_XP]; al[OB][p] = xb_[p * HZ_XP + HZ_XV]; } \
    }
#define MH_HZ_FETCH_BH(T_)                                                                            \
    {                                                                                                 \
        const uint4* wb_ = smem + bcur * HZ_BUF + 2 * HZ_XP + (T_) * (2 * H2_CN) + bbase;             \
        bh[0] = wb_[0]; bh[1] = wb_[HZ_WP];                                                           \
    }
#define MH_HZ_FETCH_BL(T_)                                                                            \
    {                                                                                                 \
        const uint4* wb_ = smem + bcur * HZ_BUF + 2 * HZ_XP + (T_) * (2 * H2_CN) + bbase;             \
        bl[0] = wb_[HZ_WV]; bl[1] = wb_[HZ_WP + HZ_WV];                                               \
    }
#define MH_HZ_FETCH(OB, T_) { MH_HZ_FETCH_A(OB, T_) MH_HZ_FETCH_BH(T_) MH_HZ_FETCH_BL(T_) }
#define MH_HZ_MM(S, A, B) acc[S] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), acc[S], 0, 0, 0);
    // one tap: six MFMAs alternating between the two positions' accumulators (P0_ = 0 | 2), the next tap's operand reads, a piece of the staging work; the
    // scheduler deals reads / vector work / memory work over the six gaps
#define MH_HZ_GAP(ND_, NV_)                                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, ND_, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x006, NV_, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x230, 2, 0);
#define MH_HZ_TAPV(T_, P0_, NV_, ...)                                                                 \
    {                                                                                                 \
        constexpr int nt_ = (T_) + 1 < 9 ? (T_) + 1 : 0;                                              \
        if ((T_) + 1 < 9) MH_HZ_FETCH_A(((T_) + 1) & 1, nt_)                                          \
        __VA_ARGS__                                                                                   \
        MH_HZ_MM(P0_, ah[(T_) & 1][0], bh[0]) MH_HZ_MM(P0_ + 1, ah[(T_) & 1][1], bh[1])                \
        MH_HZ_MM(P0_, al[(T_) & 1][0], bh[0]) MH_HZ_MM(P0_ + 1, al[(T_) & 1][1], bh[1])                \
        if ((T_) + 1 < 9) MH_HZ_FETCH_BH(nt_)                                                         \
        MH_HZ_MM(P0_, ah[(T_) & 1][0], bl[0]) MH_HZ_MM(P0_ + 1, ah[(T_) & 1][1], bl[1])                \
        if ((T_) + 1 < 9) MH_HZ_FETCH_BL(nt_)                                                         \
        MH_HZ_GAP(2, NV_) MH_HZ_GAP(2, NV_) MH_HZ_GAP(0, NV_) MH_HZ_GAP(2, NV_) MH_HZ_GAP(0, NV_) MH_HZ_GAP(2, NV_) \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define MH_HZ_TAP(T_, P0_, ...) MH_HZ_TAPV(T_, P0_, 6, __VA_ARGS__)
    // the completed output plane in acce: the four branch-free epilogue pieces of conv3d_h2.h
#define MH_HZ_EMIT_A                                                                                  \
    {                                                                                                 \
        const unsigned so_ = (unsigned)pend_z * (unsigned)(HW * 4);                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {      /* the scaled-back values replace the raw sums in acce: the statistics pieces read them there */ \
            const f32x4 o_ = f32x4{acce[4 * j], acce[4 * j + 1], acce[4 * j + 2], acce[4 * j + 3]} * inv_a * inv_b + bco; \
            acce[4 * j] = o_[0]; acce[4 * j + 1] = o_[1]; acce[4 * j + 2] = o_[2]; acce[4 * j + 3] = o_[3];  \
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o_), orsrc, (pend ? ooff[j] : HZ_DROP) + so_, 0, 0); \
        }                                                                                             \
    }
#define MH_HZ_EMIT_B1                                                                                 \
    if (STATS) {                                                                                      \
        const float pf_ = pend ? 1.0f : 0.0f;                                                         \
        esum_ = 0.0f; ecnt_ = 0.0f;                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            const float w_ = ooff[j] != HZ_DROP ? pf_ : 0.0f;                                         \
            ecnt_ += 4.0f * w_;                                                                       \
            esum_ += ((acce[4 * j] + acce[4 * j + 1]) + (acce[4 * j + 2] + acce[4 * j + 3])) * w_;    \
        }                                                                                             \
        emean_ = ecnt_ > 0.0f ? esum_ / (ecnt_ > 0.0f ? ecnt_ : 1.0f) : 0.0f;                         \
    }
#define MH_HZ_EMIT_B2                                                                                 \
    if (STATS) {                                                                                      \
        em2_ = 0.0f;                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            const f32x4 d_ = f32x4{acce[4 * j], acce[4 * j + 1], acce[4 * j + 2], acce[4 * j + 3]} - emean_; \
            const f32x4 q_ = d_ * d_;                                                                 \
            em2_ += ((q_[0] + q_[1]) + (q_[2] + q_[3])) * (ooff[j] != HZ_DROP && pend ? 1.0f : 0.0f); \
        }                                                                                             \
    }
#define MH_HZ_EMIT_B3                                                                                 \
    {                                                                                                 \
        if (STATS) {                                                                                  \
            Stat loc_;                                                                                \
            loc_.n = ecnt_; loc_.mean = emean_; loc_.m2 = em2_;                                       \
            run = stat_merge_nb(run, loc_);                                                           \
            MH_OPAQUE(run.n); MH_OPAQUE(run.mean); MH_OPAQUE(run.m2);     /* the merge happens HERE: sunk below the chunk loop it would keep the plane's 16 values alive across it */ \
        }                                                                                             \
        pend = 0;                                                                                     \
    }
#define MH_HZ_EMIT { MH_HZ_EMIT_A MH_HZ_EMIT_B1 MH_HZ_EMIT_B2 MH_HZ_EMIT_B3 }
    // the second plane of the previous pair moves from m3's registers into acce (whose plane went out in taps 1-4); m3 restarts from zero
#define MH_HZ_TAKE1                                                                                   \
    {                                                                                                 \
        acce = acc[3];                                                                                \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[3][i] = 0.0f;                              \
        pend = pend1; pend_z = pend1_z; pend1 = 0;                                                    \
    }
#define MH_HZ_NONE

    // ---- prologue: planes a, b, c of body 0 -> V0, V1 in buffer 0 with the weights of sub-step 0; plane d and the weights of sub-step 1 in flight ----
    int bcur = 1;                     // the conversion pieces write the buffer "after" bcur
    int cv_t = 0;                     // pair of the planes in the registers
    bool fv_a, fv_b, fv_c, fv_d;
    MH_HZ_LDPLANE(rad, -1) MH_HZ_LDPLANE(rb, 0) MH_HZ_LDPLANE(rc, 1)
    MH_HZ_LDW(0)
    MH_HZ_NRMLD(0)
    fv_a = MH_HZ_VALID(-1); fv_b = MH_HZ_VALID(0); fv_c = MH_HZ_VALID(1); fv_d = MH_HZ_VALID(2);
    MH_HZ_CONV01(0) MH_HZ_CONV01(1) MH_HZ_CONV01(2)
    MH_HZ_WST
    MH_HZ_LDPLANE(rad, 2)             // d of body 0
    bcur = 0;
    __syncthreads();

    // ---- the march: body (pair t, chunk c) = S0 (positions 0, 1) and S1 (positions 2, 3) ----
    // S0 of a pair's FIRST chunk carries the previous pair's two epilogues (taps 1-4 and 5-8); the other chunks' S0 is the same schedule without them
    // a body = S0 + S1 of (pair t, chunk c); written out twice -- a pair's first chunk with the epilogue pieces, the loop over its other chunks without -- so
    // that no two-sided branch joins the 250 live registers (a diamond around S0 alone cost 340 bytes of scratch per lane)
#define MH_HZ_BODY(C_, E_A, E_B1, E_B2, E_B3, E_TAKE, NV_)                                            \
        {                                                                                             \
            /* next body (its planes are loaded during this one); beyond the last: ld_t = NP -> zero-record descriptors */ \
            const bool wrap = (C_) + 1 == NCH;                                                        \
            ld_t = wrap ? t + 1 : t; ld_c = wrap ? 0 : (C_) + 1;                                      \
            const int wnext = (ld_t < NP) ? 2 * ld_c : 2 * (C_);      /* weights of the next body's S0 (beyond the last: any valid slab) */ \
            /* ---- S0: MFMAs of V0, V1; V2, V3 of this body -> other buffer; then the next body's a', b', c' ---- */ \
            MH_HZ_FETCH(0, 0)                                                                             \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            MH_HZ_TAPV(0, 0, 11, MH_HZ_CONV23(0))                                                         \
            MH_HZ_TAPV(1, 0, 11, MH_HZ_CONV23(1))                                                         \
            MH_HZ_TAPV(2, 0, 11, MH_HZ_CONV23(2))                                                         \
            MH_HZ_TAPV(3, 0, NV_, MH_HZ_LDW(2 * (C_) + 1) E_A)                                            \
            MH_HZ_TAPV(4, 0, NV_, MH_HZ_LDPLANE(rad, -1) E_B1 E_B2)                                       \
            MH_HZ_TAPV(5, 0, NV_, MH_HZ_LDPLANE(rb, 0) E_B3 E_TAKE)                                       \
            MH_HZ_TAPV(6, 0, NV_, MH_HZ_LDPLANE(rc, 1) E_A)                                               \
            MH_HZ_TAPV(7, 0, NV_, E_B1 E_B2)                                                              \
            MH_HZ_TAPV(8, 0, NV_, MH_HZ_WST MH_HZ_NRMLD(ld_t < NP ? ld_c : (C_)) E_B3)                    \
            __syncthreads();                                                                          \
            bcur ^= 1;                                                                                \
            cv_t = ld_t;              /* the planes in the registers now belong to the next body */   \
            fv_a = MH_HZ_VALID(-1); fv_b = MH_HZ_VALID(0); fv_c = MH_HZ_VALID(1); fv_d = MH_HZ_VALID(2); \
            /* ---- S1: MFMAs of V2, V3; V0', V1' of the next body -> other buffer; then its d' ---- */ \
            MH_HZ_FETCH(0, 0)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            MH_HZ_TAPV(0, 2, 11, MH_HZ_CONV01(0))                                                     \
            MH_HZ_TAPV(1, 2, 11, MH_HZ_CONV01(1))                                                     \
            MH_HZ_TAPV(2, 2, 11, MH_HZ_CONV01(2))                                                     \
            MH_HZ_TAP(3, 2, MH_HZ_LDW(wnext))                                                         \
            MH_HZ_TAP(4, 2, MH_HZ_LDPLANE(rad, 2))                                                    \
            MH_HZ_TAP(5, 2, MH_HZ_NONE)                                                               \
            MH_HZ_TAP(6, 2, MH_HZ_NONE)                                                               \
            MH_HZ_TAP(7, 2, MH_HZ_NONE)                                                               \
            MH_HZ_TAP(8, 2, MH_HZ_WST)                                                                \
            __syncthreads();                                                                          \
            bcur ^= 1;                                                                                \
        }
    for (int t = 0; t < NP; ++t) {
        MH_HZ_BODY(0, MH_HZ_EMIT_A, MH_HZ_EMIT_B1, MH_HZ_EMIT_B2, MH_HZ_EMIT_B3, MH_HZ_TAKE1, 8)
        for (int c = 1; c < NCH; ++c) MH_HZ_BODY(c, MH_HZ_NONE, MH_HZ_NONE, MH_HZ_NONE, MH_HZ_NONE, MH_HZ_NONE, 6)
        // the pair is complete: inverse transform (fp32, in registers); its two planes go out during the next pair's first S0
        const int z0 = zs + 2 * t;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float m0 = acc[0][k], m1 = acc[1][k], m2 = acc[2][k], m3 = acc[3][k];
            acce[k] = (m0 + m1) + m2;
            acc[3][k] = (m1 - m2) - m3;
            acc[0][k] = 0.0f; acc[1][k] = 0.0f; acc[2][k] = 0.0f;
        }
        pend = 1; pend_z = z0;
        pend1 = z0 + 1 < ze ? 1 : 0; pend1_z = z0 + 1;
    }
#undef MH_HZ_BODY
    // the last pair's planes
    if (pend) MH_HZ_EMIT
    MH_HZ_TAKE1
    if (pend) MH_HZ_EMIT
#undef MH_HZ_NONE
#undef MH_HZ_TAKE1
#undef MH_HZ_EMIT
#undef MH_HZ_EMIT_B3
#undef MH_HZ_EMIT_B2
#undef MH_HZ_EMIT_B1
#undef MH_HZ_EMIT_A
#undef MH_HZ_TAP
#undef MH_HZ_TAPV
#undef MH_HZ_MM
#undef MH_HZ_FETCH
#undef MH_HZ_FETCH_BL
#undef MH_HZ_FETCH_BH
#undef MH_HZ_FETCH_A
#undef MH_HZ_GAP
#undef MH_HZ_WST
#undef MH_HZ_LDW
#undef MH_HZ_NRMLD
#undef MH_HZ_CONV01
#undef MH_HZ_CONV23
#undef MH_HZ_PUT
#undef MH_HZ_ACT
#undef MH_HZ_ACT_IMPL
#undef MH_HZ_SPLIT2
#undef MH_HZ_VALID
#undef MH_HZ_LDPLANE

    if (STATS) {
        // the two k-group halves of a lane pair hold disjoint voxels of the same cout; then the eight waves merge through LDS (conv3d_h2.h)
        {
            Stat ot;
            ot.n = __shfl_xor(run.n, 32);
            ot.mean = __shfl_xor(run.mean, 32);
            ot.m2 = __shfl_xor(run.m2, 32);
            run = kg == 0 ? stat_merge(run, ot) : stat_merge(ot, run);
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (kg == 0) {
            red[(wave * H2_CN + r32) * 3] = run.n; red[(wave * H2_CN + r32) * 3 + 1] = run.mean; red[(wave * H2_CN + r32) * 3 + 2] = run.m2;
        }
        __syncthreads();
        if (tid < H2_CN) {
            Stat st;
            st.n = 0.0f; st.mean = 0.0f; st.m2 = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                Stat ot;
                ot.n = red[(w * H2_CN + tid) * 3]; ot.mean = red[(w * H2_CN + tid) * 3 + 1]; ot.m2 = red[(w * H2_CN + tid) * 3 + 2];
                st = stat_merge(st, ot);
            }
            float* rec = stats + (((long long)n * Cout + cg * H2_CN + tid) * nblk + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}

// Weight preparation: the scale launch of conv3d_h2.h (tail = {1 / S, S}), then
// w [Cout][Cin][3][3][3] -> [cout group][chunk][position 0..3][piece][tap (ky, kx)][k-group][32 couts][8 channels] fp16 with the z-taps g0, g1, g2 of every
// (cout, cin, ky, kx) transformed in fp32: U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 (the sums are formed from the SCALED weights: the
// scale is a power of two, so they equal the scaled sums; the halving is exact).  One thread per (cout, cin).
__global__ void __launch_bounds__(256)
conv3d_k3_h2z_pack_kernel(const float* __restrict__ w, int Cin, int Cout, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const int nchunk = Cin / H2_KC;
    const float s = tail[1];
    _Float16* slab = packed + ((long long)(co / H2_CN) * nchunk + ci / H2_KC) * (4LL * HZ_WP * 8);
    const float* wc = w + ((long long)co * Cin + ci) * 27;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float g0 = wc[tap] * s, g1 = wc[9 + tap] * s, g2 = wc[18 + tap] * s;
        const float u[4] = {g0, ((g0 + g1) + g2) * 0.5f, ((g0 - g1) + g2) * 0.5f, g2};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            _Float16 pc[2];
            h2_split(u[p], pc[0], pc[1]);
#pragma unroll
            for (int piece = 0; piece < 2; ++piece)
                slab[((((p * 2 + piece) * 9 + tap) * 2 + (ci % H2_KC) / 8) * H2_CN + (co % H2_CN)) * 8 + (ci % 8)] = pc[piece];
        }
    }
}

}  // namespace mh
