// conv3d_h2z.h's convolution -- the split-precision product behind the Winograd F(2, 3) transform along z -- laid out for ONE wave per SIMD: four waves of 512
// registers instead of eight of 256.  Round-4 counters of the eight-wave kernel (profiles/r04_pmc_h2z_vs_h2.txt): the matrix pipe is busy 0.46 of the cycles; per MFMA
// it issues 1.55x the LDS traffic and 2.3x the vector-memory instructions of the direct kernel.  Here a wave owns TWO rows of the 8 x 32 region (w and w + 4):
// every weight operand read feeds two row blocks (12 operand reads per 12 MFMAs instead of 8 per 6), an accumulator returns every fourth matrix instruction,
// 128 accumulator registers sit in the AGPR half of the file, and all operand sets are double buffered.  Arithmetic, LDS image, weight packing and results are
// those of conv3d_h2z.h, bit for bit (the same sums in the same order per output element).
#pragma once
#include "conv3d_h2z.h"

namespace mh {

constexpr int HW_SLOTS = 6;                                 // staging tasks per lane: (voxel, 4 channels); 340 voxels x 4 quads over 256 lanes
constexpr int HW_WSLOTS = HZ_WSUB / 256;                    // uint4 of a sub-step's weights per thread: 9
static_assert(HZ_NV <= 64 * HW_SLOTS && HW_WSLOTS * 256 == HZ_WSUB, "staging slots");

template <bool STATS, bool NRM>
__global__ void __launch_bounds__(256, 1)
conv3d_k3_h2zw_kernel(Tensor in, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias, Tensor out,
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

    // input staging: wave w (of 4) converts channels 4w .. 4w+3 of the step for ALL 340 staged voxels: lane + 64 j, j < 6
    const int q = wave;
    unsigned soff[HW_SLOTS];          // BYTE offsets into a channel plane
    int loff[HW_SLOTS];               // destination in units of 8 bytes inside a piece
#pragma unroll
    for (int j = 0; j < HW_SLOTS; ++j) {
        const int e0 = lane + 64 * j;
        const int e = min(e0, HZ_NV - 1);
        const int ly = e / HZ_RX, lx = e - ly * HZ_RX;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        const bool ok = e0 < HZ_NV && gy >= 0 && gy < H && gx >= 0 && gx < W;
        soff[j] = ok ? 4u * (unsigned)(gy * W + gx) : 0u;
        loff[j] = ((q >> 1) * HZ_PV + (ok ? e : HZ_NV + (lane & 3))) * 2 + (q & 1);
    }
    for (int i = tid; i < 2 * HZ_BUF; i += 256) smem[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();                  // (bound_s aliases the zeroed area)
    if (NRM) {
        unsigned mb = 0u;
        for (int c = tid; c < Cin; c += 256) {
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
        for (int w = 1; w < 4; ++w) mb = max(mb, bound_s[w]);
        poisoned = mb >= 0x7f800000u;
        e_in = poisoned ? 0 : min(max(14 - ((int)(mb >> 23) - 126), -100), 100);
        const float p_ = __uint_as_float((unsigned)(e_in + 127) << 23);
        __syncthreads();              // every thread has read the bounds: the words go back to zero before the weight area is used
        if (tid < 8) bound_s[tid] = 0u;
        for (int c = tid; c < Cin; c += 256) { float* r_ = nrm_s + 16 * (c >> 2) + (c & 3); r_[0] *= p_; r_[4] *= p_; }
        __syncthreads();
    }

    const float* const src = in.data + (long long)n * in.n_stride + (long long)(4 * q) * DHW;
    const u32x4* const wg4 = reinterpret_cast<const u32x4*>(wp) + (long long)cg * NCH * (2 * HZ_WSUB) + tid;

    const float m1 = h2_minus_one();
    // raw planes of the current (pair, chunk): three register sets -- rb = b, rc = c, rad = a (while V0 is formed) | d (while V3 is formed)
    float rb[HW_SLOTS][4], rc[HW_SLOTS][4], rad[HW_SLOTS][4];
    u32x4 win[HW_WSLOTS];
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
        _Pragma("unroll") for (int j = 0; j < HW_SLOTS; ++j)                                          \
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
        _Pragma("unroll") for (int j = 0; j < HW_WSLOTS; ++j) win[j] = wg4[(long long)wi_ * HZ_WSUB + 256 * j]; \
    }
#define MH_HZ_WST                                                                                     \
    {                                                                                                 \
        u32x4* w4_ = reinterpret_cast<u32x4*>(smem + (bcur ^ 1) * HZ_BUF + 2 * HZ_XP);                \
        _Pragma("unroll") for (int j = 0; j < HW_WSLOTS; ++j) w4_[tid + 256 * j] = win[j];            \
    }

    // operands of this lane: A = voxel (rows w and w + 4, x = lane & 31), B = cout (lane & 31); k-group = lane >> 5
    const int r32 = lane & 31, kg = lane >> 5;
    const int abase = kg * HZ_PV + wave * HZ_RX + r32;            // row block 1: + 4 * HZ_RX
    const int bbase = kg * H2_CN + r32;

    f32x16 acc[8], acce[2];           // acc[4 r + p]: row block r, transform position p
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[s][i] = 0.0f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acce[r][i] = 0.0f;

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
    unsigned ooff[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xg_ = 8 * j + 4 * kg, yr_ = y0 + wave + 4 * r;
            const bool ok_ = yr_ < H && x0 + xg_ < W;
            ooff[r][j] = ok_ ? 4u * (unsigned)((long long)r32 * DHW + (long long)yr_ * W + x0 + xg_) : HZ_DROP;
        }
    float esum_ = 0.0f, ecnt_ = 0.0f, em2_ = 0.0f, emean_ = 0.0f;
    Stat run;
    run.n = 0.0f; run.mean = 0.0f; run.m2 = 0.0f;
    int pend = 0, pend_z = 0;         // acce holds a completed plane (pend_z) waiting for its epilogue
    int pend1 = 0, pend1_z = 0;       // acc[3] holds the second plane of the previous pair (moved to acce in tap 5 of the next sub-step 0)

    // operands: A (the transformed voxels of both row blocks) of the next tap is read one tap ahead into the other register set; B (the weights) has ONE set,
    // refilled inside the tap as soon as its last matrix instruction has issued -- high pieces after the eighth, low pieces after the twelfth
    uint4 ah[2][2][2], al[2][2][2], bh[2], bl[2];          // A: [operand buffer][row block][position]; B: [position]
#define MH_HZ_FETCH_A(OB, T_)                                                                         \
    {                                                                                                 \
        constexpr int aoff_ = ((T_) / 3) * HZ_RX + (T_) % 3;                                          \
        const uint4* xb_ = smem + bcur * HZ_BUF + abase + aoff_;                                      \
        _Pragma("unroll") for (int r = 0; r < 2; ++r)                                                 \
            _Pragma("unroll") for (int p = 0; p < 2; ++p) { ah[OB][r][p] = xb_[p * HZ_XP + r * 4 * HZ_RX]; al[OB][r][p] = xb_[p * HZ_XP + HZ_XV + r * 4 * HZ_RX]; } \
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
    // one tap: twelve MFMAs -- two row blocks x two positions (P0_ = 0 | 2) x three piece products; every weight operand feeds both row blocks, an accumulator
    // returns every fourth instruction; the next tap's operand reads and a piece of the staging work are dealt over the twelve gaps
#define MH_HZ_GAP(ND_, NV_)                                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, ND_, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x006, NV_, 0);                                          \
        __builtin_amdgcn_sched_group_barrier(0x230, 2, 0);
#define MH_HZ_MM4(OB, P0_, AX_, BX_)                                                                  \
        MH_HZ_MM(P0_, AX_[OB][0][0], BX_[0]) MH_HZ_MM(4 + P0_, AX_[OB][1][0], BX_[0])                   \
        MH_HZ_MM(P0_ + 1, AX_[OB][0][1], BX_[1]) MH_HZ_MM(4 + P0_ + 1, AX_[OB][1][1], BX_[1])
#define MH_HZ_TAPV(T_, P0_, NV_, ...)                                                                 \
    {                                                                                                 \
        constexpr int nt_ = (T_) + 1 < 9 ? (T_) + 1 : 0;                                              \
        if ((T_) + 1 < 9) MH_HZ_FETCH_A(((T_) + 1) & 1, nt_)                                          \
        __VA_ARGS__                                                                                   \
        MH_HZ_MM4((T_) & 1, P0_, ah, bh)                                                              \
        MH_HZ_MM4((T_) & 1, P0_, al, bh)                                                              \
        if ((T_) + 1 < 9) MH_HZ_FETCH_BH(nt_)                                                         \
        MH_HZ_MM4((T_) & 1, P0_, ah, bl)                                                              \
        if ((T_) + 1 < 9) MH_HZ_FETCH_BL(nt_)                                                         \
        MH_HZ_GAP(2, NV_) MH_HZ_GAP(2, NV_) MH_HZ_GAP(2, NV_) MH_HZ_GAP(2, NV_) MH_HZ_GAP(0, NV_) MH_HZ_GAP(0, NV_) MH_HZ_GAP(0, NV_) MH_HZ_GAP(2, NV_) \
        MH_HZ_GAP(0, NV_) MH_HZ_GAP(0, NV_) MH_HZ_GAP(0, NV_) MH_HZ_GAP(2, NV_)                        \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define MH_HZ_TAP(T_, P0_, ...) MH_HZ_TAPV(T_, P0_, 4, __VA_ARGS__)
    // the completed output plane in acce: the four branch-free epilogue pieces of conv3d_h2.h
#define MH_HZ_EMIT_A(R_)                                                                              \
    {                                                                                                 \
        const unsigned so_ = (unsigned)pend_z * (unsigned)(HW * 4);                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {      /* the scaled-back values replace the raw sums in acce: the statistics pieces read them there */ \
            const f32x4 o_ = f32x4{acce[R_][4 * j], acce[R_][4 * j + 1], acce[R_][4 * j + 2], acce[R_][4 * j + 3]} * inv_a * inv_b + bco; \
            acce[R_][4 * j] = o_[0]; acce[R_][4 * j + 1] = o_[1]; acce[R_][4 * j + 2] = o_[2]; acce[R_][4 * j + 3] = o_[3];  \
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o_), orsrc, (pend ? ooff[R_][j] : HZ_DROP) + so_, 0, 0); \
        }                                                                                             \
    }
#define MH_HZ_EMIT_B1(R_)                                                                             \
    if (STATS) {                                                                                      \
        const float pf_ = pend ? 1.0f : 0.0f;                                                         \
        esum_ = 0.0f; ecnt_ = 0.0f;                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            const float w_ = ooff[R_][j] != HZ_DROP ? pf_ : 0.0f;                                     \
            ecnt_ += 4.0f * w_;                                                                       \
            esum_ += ((acce[R_][4 * j] + acce[R_][4 * j + 1]) + (acce[R_][4 * j + 2] + acce[R_][4 * j + 3])) * w_; \
        }                                                                                             \
        emean_ = ecnt_ > 0.0f ? esum_ / (ecnt_ > 0.0f ? ecnt_ : 1.0f) : 0.0f;                         \
    }
#define MH_HZ_EMIT_B2(R_)                                                                             \
    if (STATS) {                                                                                      \
        em2_ = 0.0f;                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            const f32x4 d_ = f32x4{acce[R_][4 * j], acce[R_][4 * j + 1], acce[R_][4 * j + 2], acce[R_][4 * j + 3]} - emean_; \
            const f32x4 q_ = d_ * d_;                                                                 \
            em2_ += ((q_[0] + q_[1]) + (q_[2] + q_[3])) * (ooff[R_][j] != HZ_DROP && pend ? 1.0f : 0.0f); \
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
    }
#define MH_HZ_EMIT_END pend = 0;
#define MH_HZ_EMIT { MH_HZ_EMIT_A(0) MH_HZ_EMIT_B1(0) MH_HZ_EMIT_B2(0) MH_HZ_EMIT_B3 MH_HZ_EMIT_A(1) MH_HZ_EMIT_B1(1) MH_HZ_EMIT_B2(1) MH_HZ_EMIT_B3 MH_HZ_EMIT_END }
    // the second plane of the previous pair moves from m3's registers into acce (whose plane went out in taps 1-4); m3 restarts from zero
#define MH_HZ_TAKE1                                                                                   \
    {                                                                                                 \
        acce[0] = acc[3]; acce[1] = acc[7];                                                           \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) { acc[3][i] = 0.0f; acc[7][i] = 0.0f; }        \
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
    MH_HZ_CONV01(0) MH_HZ_CONV01(1) MH_HZ_CONV01(2) MH_HZ_CONV01(3) MH_HZ_CONV01(4) MH_HZ_CONV01(5)
    MH_HZ_WST
    MH_HZ_LDPLANE(rad, 2)             // d of body 0
    bcur = 0;
    __syncthreads();

    // ---- the march: body (pair t, chunk c) = S0 (positions 0, 1) and S1 (positions 2, 3) ----
    // S0 of a pair's FIRST chunk carries the previous pair's two epilogues (taps 1-4 and 5-8); the other chunks' S0 is the same schedule without them
    // a body = S0 + S1 of (pair t, chunk c); written out twice -- a pair's first chunk with the epilogue pieces, the loop over its other chunks without -- so
    // that no two-sided branch joins the 250 live registers (a diamond around S0 alone cost 340 bytes of scratch per lane)
#define MH_HZ_BODY(C_, EMIT_, NV_)                                                                    \
        {                                                                                             \
            /* next body (its planes are loaded during this one); beyond the last: ld_t = NP -> zero-record descriptors */ \
            const bool wrap = (C_) + 1 == NCH;                                                        \
            ld_t = wrap ? t + 1 : t; ld_c = wrap ? 0 : (C_) + 1;                                      \
            const int wnext = (ld_t < NP) ? 2 * ld_c : 2 * (C_);      /* weights of the next body's S0 (beyond the last: any valid slab) */ \
            /* ---- S0: MFMAs of V0, V1; V2, V3 of this body -> other buffer; then the next body's a', b', c'.  EMIT_: the previous pair's four plane rows ---- */ \
            MH_HZ_FETCH(0, 0)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            MH_HZ_TAPV(0, 0, NV_, MH_HZ_LDW(2 * (C_) + 1) MH_HZ_CONV23(0) EMIT_(MH_HZ_EMIT_A(0) MH_HZ_EMIT_B1(0))) \
            MH_HZ_TAPV(1, 0, NV_, MH_HZ_CONV23(1) EMIT_(MH_HZ_EMIT_B2(0) MH_HZ_EMIT_B3))              \
            MH_HZ_TAPV(2, 0, NV_, MH_HZ_CONV23(2) EMIT_(MH_HZ_EMIT_A(1) MH_HZ_EMIT_B1(1)))            \
            MH_HZ_TAPV(3, 0, NV_, MH_HZ_CONV23(3) EMIT_(MH_HZ_EMIT_B2(1) MH_HZ_EMIT_B3 MH_HZ_TAKE1))  \
            MH_HZ_TAPV(4, 0, NV_, MH_HZ_CONV23(4) EMIT_(MH_HZ_EMIT_A(0) MH_HZ_EMIT_B1(0)))            \
            MH_HZ_TAPV(5, 0, NV_, MH_HZ_CONV23(5) EMIT_(MH_HZ_EMIT_B2(0) MH_HZ_EMIT_B3))              \
            MH_HZ_TAPV(6, 0, NV_, MH_HZ_LDPLANE(rad, -1) EMIT_(MH_HZ_EMIT_A(1) MH_HZ_EMIT_B1(1)))     \
            MH_HZ_TAPV(7, 0, NV_, MH_HZ_LDPLANE(rb, 0) EMIT_(MH_HZ_EMIT_B2(1) MH_HZ_EMIT_B3 MH_HZ_EMIT_END)) \
            MH_HZ_TAPV(8, 0, 4, MH_HZ_WST MH_HZ_LDPLANE(rc, 1) MH_HZ_NRMLD(ld_t < NP ? ld_c : (C_))) \
            __syncthreads();                                                                          \
            bcur ^= 1;                                                                                \
            cv_t = ld_t;              /* the planes in the registers now belong to the next body */   \
            fv_a = MH_HZ_VALID(-1); fv_b = MH_HZ_VALID(0); fv_c = MH_HZ_VALID(1); fv_d = MH_HZ_VALID(2); \
            /* ---- S1: MFMAs of V2, V3; V0', V1' of the next body -> other buffer; then its d' ---- */ \
            MH_HZ_FETCH(0, 0)                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            MH_HZ_TAPV(0, 2, 7, MH_HZ_LDW(wnext) MH_HZ_CONV01(0))                                     \
            MH_HZ_TAPV(1, 2, 7, MH_HZ_CONV01(1))                                                      \
            MH_HZ_TAPV(2, 2, 7, MH_HZ_CONV01(2))                                                      \
            MH_HZ_TAPV(3, 2, 7, MH_HZ_CONV01(3))                                                      \
            MH_HZ_TAPV(4, 2, 7, MH_HZ_CONV01(4))                                                      \
            MH_HZ_TAPV(5, 2, 7, MH_HZ_CONV01(5))                                                      \
            MH_HZ_TAP(6, 2, MH_HZ_LDPLANE(rad, 2))                                                    \
            MH_HZ_TAP(7, 2, MH_HZ_WST)                                                                \
            MH_HZ_TAP(8, 2, MH_HZ_NONE)                                                               \
            __syncthreads();                                                                          \
            bcur ^= 1;                                                                                \
        }
#define MH_HZ_YES(...) __VA_ARGS__
#define MH_HZ_NO(...)
    for (int t = 0; t < NP; ++t) {
        MH_HZ_BODY(0, MH_HZ_YES, 8)
        for (int c = 1; c < NCH; ++c) MH_HZ_BODY(c, MH_HZ_NO, 4)
        // the pair is complete: inverse transform (fp32, in registers) of both row blocks; the four plane rows go out during the next pair's first S0
        const int z0 = zs + 2 * t;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const float m0 = acc[4 * r][k], m1 = acc[4 * r + 1][k], m2 = acc[4 * r + 2][k], m3 = acc[4 * r + 3][k];
                acce[r][k] = (m0 + m1) + m2;
                acc[4 * r + 3][k] = (m1 - m2) - m3;
                acc[4 * r][k] = 0.0f; acc[4 * r + 1][k] = 0.0f; acc[4 * r + 2][k] = 0.0f;
            }
        pend = 1; pend_z = z0;
        pend1 = z0 + 1 < ze ? 1 : 0; pend1_z = z0 + 1;
    }
#undef MH_HZ_YES
#undef MH_HZ_NO
#undef MH_HZ_BODY
    // the last pair's planes
    if (pend) MH_HZ_EMIT
    MH_HZ_TAKE1
    if (pend) MH_HZ_EMIT
#undef MH_HZ_NONE
#undef MH_HZ_TAKE1
#undef MH_HZ_EMIT
#undef MH_HZ_EMIT_B3
#undef MH_HZ_EMIT_END
#undef MH_HZ_MM4
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
            for (int w = 0; w < 4; ++w) {
                Stat ot;
                ot.n = red[(w * H2_CN + tid) * 3]; ot.mean = red[(w * H2_CN + tid) * 3 + 1]; ot.m2 = red[(w * H2_CN + tid) * 3 + 2];
                st = stat_merge(st, ot);
            }
            float* rec = stats + (((long long)n * Cout + cg * H2_CN + tid) * nblk + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}


}  // namespace mh
