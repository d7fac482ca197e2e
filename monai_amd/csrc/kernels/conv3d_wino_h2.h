// Conv3d 3x3x3 / stride 1 / zero padding 1 for 32 input channels: in-plane Winograd F(2x2, 3x3) IN FRONT OF the two-piece split-precision product of
// conv3d_h2.h -- 2.25x fewer matrix instructions per output voxel (16 transform positions x 3 z-taps per 2 x 2 outputs instead of 27 taps x 4), z-streaming.
// Same reference op, "normalise on load" contract, record / bound contract and tolerance class as conv3d_h2.h (nn.Conv3d of `Convolution`,
// monai/networks/blocks/convolutions.py:98-171, fed by the previous block's deferred InstanceNorm + LeakyReLU).
//
// Why this shape (round 6; DESIGN.md 5.3 sized the corners that keep the transformed weights in LDS or stream them from L2 and found none that fits): the transformed
// weights live in REGISTERS.  A workgroup is 4 waves, one per SIMD, each with the full 512-register budget.  Wave i owns ROW i of the 4 x 4 transform positions
// xi = (i, j): its B operands U[(i, j)][z-tap t][32 cin][32 cout] as hi / lo fp16 pieces are 4 x 3 x 2 cout blocks x 2 pieces = 48 operands of
// v_mfma_f32_16x16x32_f16 = 192 registers, loaded once per workgroup.  M = 16 tiles (a 4 x 16 output region = 2 x 8 tiles of 2 x 2), N = 16 couts per block, K = all
// 32 input channels in ONE instruction: the transform-domain sums of the three live output planes are 4 j x 3 planes x 2 blocks x 4 = 96 registers.
//
// Per input plane p: (1) all waves stage the activated, scaled fp32 plane region [6 x 18 positions][32 channels] into LDS (144-byte position pitch: the 16-byte operand
// reads of a lane group cover all 64 banks once); (2) wave i reads the two rows of every tile's 4 x 4 patch that ROW i of B^T d B needs (i = 0: d0 - d2, 1: d1 + d2,
// 2: d2 - d1, 3: d1 - d3 -- a wave-uniform (row, row, sign)), forms V[i][0..3] in fp32 and splits each into hi / lo fp16 (the split comes AFTER the transform: the
// sums are exact to fp32 rounding); (3) 72 matrix instructions: for every j, z-tap t and cout block  acc[plane p + 1 - t][j] += Vh Uh + Vl Uh + Vh Ul;
// (4) when output plane p - 1 is complete its wave-local half of the inverse transform Z[i][b'] = sum_j M[i][j] A^T[b'][j] goes to LDS, and after the plane's barrier
// every wave finishes a quarter of the outputs Y[a'][b'] = sum_i A^T[a'][i] Z[i][b'] (scale back, bias, statistics, one 16-byte store per output row).
// The accumulator sets rotate by NAME (the plane loop is unrolled three times), not by register moves; a fresh set starts from the instruction's zero C operand.
//
// Weight preparation (conv3d_k3_h2w_pack_kernel): U = G g G^T per z-tap in fp64, scaled by a power of two (a quarter of conv3d_h2.h's: |U| <= 2.25 max |g|), split
// into hi / lo fp16, stored in the waves' register order [cout group][wave i][48 operands][64 lanes][8 halves].  The input scale leaves two more bits of headroom than
// conv3d_h2.h (|V| <= 4 max |d|).
#pragma once
#include "common.h"
#include "conv3d_h2.h"

namespace mh {

constexpr int HWG_BY = 4, HWG_BX = 16;                      // outputs of a workgroup's region: 2 x 8 tiles of 2 x 2
constexpr int HWG_RY = HWG_BY + 2, HWG_RX = HWG_BX + 2;     // staged rows, columns
constexpr int HWG_NP = HWG_RY * HWG_RX;                     // staged positions (108)
constexpr int HWG_PB = 144;                                 // bytes per position: 32 channels fp32 + 16 (bank spread)
constexpr int HWG_DB = (HWG_NP + 1) * HWG_PB;               // bytes per input buffer (one dump position at the end): 15 696
constexpr int HWG_ZB = 16 * 1024;                           // bytes per Z exchange buffer: [4 i][2 blocks][2 register pairs][64 lanes][16 bytes]
constexpr int HWG_OPS = 48;                                 // B operands per wave: [j][t][cout block][piece]
constexpr int HWG_CIN = 32, HWG_CN = 32;

__device__ __forceinline__ void hw_split4(const f32x4 v, u32x2& hi, u32x2& lo) {
    _Float16 h_[4], l_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) h2_split(v[i], h_[i], l_[i]);
    const f16x2 h01 = {h_[0], h_[1]}, h23 = {h_[2], h_[3]}, l01 = {l_[0], l_[1]}, l23 = {l_[2], l_[3]};
    hi = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
    lo = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
}

template <bool STATS, bool ACC = false, bool POOL = false>
__global__ void __launch_bounds__(256)
conv3d_k3_h2w_kernel(Tensor in, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias, Tensor out,
                     float* __restrict__ stats, int bxn, int byn, int zchunk, unsigned nblk, float* __restrict__ pmax, float* __restrict__ pmin,
                     long long pool_n_stride) {
    __shared__ __attribute__((aligned(16))) char smem[2 * HWG_DB + 2 * HWG_ZB];
    __shared__ unsigned bound_s[4];
    char* const ds = smem;
    char* const zs_ = smem + 2 * HWG_DB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long HW = (long long)H * W, DHW = (long long)D * HW;

    const unsigned ncg = (unsigned)(Cout / HWG_CN);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(lid % ncg);
    lid /= ncg;
    const unsigned b = lid % nblk;
    const int n = (int)(lid / nblk);
    const int x0 = (int)(b % bxn) * HWG_BX, y0 = (int)((b / bxn) % byn) * HWG_BY;
    const int zs = (int)(b / (bxn * byn)) * zchunk, ze = min(zs + zchunk, D);
    const int p_first = max(zs - 1, 0), p_last = min(ze, D - 1);

    // ---- staging: wave w converts channel quads 2w, 2w + 1 for positions lane, lane + 64 of the 108 -------------------------------------------------------
    unsigned soff[2];                 // byte offset of the position inside a channel plane
    int loff[2];                      // byte offset of the position's cell in an input buffer (without the quad)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int e = lane + 64 * j;
        const int ec = min(e, HWG_NP - 1);
        const int ly = ec / HWG_RX, lx = ec - ly * HWG_RX;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        const bool ok = e < HWG_NP && gy >= 0 && gy < H && gx >= 0 && gx < W;
        soff[j] = ok ? 4u * (unsigned)(gy * W + gx) : 0u;
        loff[j] = (ok ? ec : HWG_NP) * HWG_PB;       // zero padding / idle lanes: the dump cell; the real cell stays zero
    }
    for (int i = tid; i < 2 * HWG_DB / 16; i += 256) reinterpret_cast<uint4*>(ds)[i] = make_uint4(0u, 0u, 0u, 0u);
    // records of this wave's eight channels {alpha, beta, slope} (constant over the march) and the sample's largest bound
    f32x4 nra[2], nrb[2], nrs[2];
    {
        unsigned mb = 0u;
        if (lane < HWG_CIN) {
            const float4 a = *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * lane);
            const unsigned bb = abs_bits(a.w);
            mb = bb == 0u ? 0x7fc00000u : bb;             // no bound given counts as non-finite
        }
        mb = wave_umax(mb);
        if (tid == 0) bound_s[0] = mb;
    }
    __syncthreads();
    int e_in = 0;
    bool poisoned = false;
    {
        const unsigned mb = bound_s[0];
        poisoned = mb >= 0x7f800000u;
        // bound < 2^eb  ->  bound * 2^(13 - eb) < 2^13: the transform's sums of four stay below 2^15
        e_in = poisoned ? 0 : min(max(13 - ((int)(mb >> 23) - 126), -100), 100);
        const float p_ = __uint_as_float((unsigned)(e_in + 127) << 23);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 a = *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * (8 * wave + 4 * q + i));
                nra[q][i] = a.x * p_; nrb[q][i] = a.y * p_; nrs[q][i] = a.z;
            }
    }
    const float* xptr = in.data + (long long)n * in.n_stride + (long long)p_first * HW + (long long)(8 * wave) * DHW;
    float xin[2][2][4];
#define MH_HW_LDX                                                                                     \
    {                                                                                                 \
        const auto xr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xptr), 0, 0x7fffffff, 0x00020000); \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j)                                             \
                _Pragma("unroll") for (int i = 0; i < 4; ++i)                                         \
                    xin[q][j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr_, soff[j], (unsigned)((4 * q + i) * DHW * 4), 0)); \
    }
#define MH_HW_CONV(BUF)                                                                               \
    {                                                                                                 \
        char* const db_ = ds + (BUF) * HWG_DB + (8 * wave) * 4;                                       \
        _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                 \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                           \
                f32x4 y_;                                                                             \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) y_[i] = act(xin[q][j][i], nra[q][i], nrb[q][i], nrs[q][i]); \
                *reinterpret_cast<f32x4*>(db_ + loff[j] + 16 * q) = y_;                               \
            }                                                                                         \
    }
    int staged = p_first;             // plane whose loads are in the registers
#define MH_HW_ADV { const bool adv_ = staged < p_last; xptr += adv_ ? HW : 0LL; staged += adv_ ? 1 : 0; }

    // ---- B operands: the wave's 48 register sets -----------------------------------------------------------------------------------------------------------
    u32x4 wu[HWG_OPS];
    {
        const u32x4* wg = reinterpret_cast<const u32x4*>(wp) + ((long long)(cg * 4 + wave) * HWG_OPS) * 64 + lane;
#pragma unroll
        for (int r = 0; r < HWG_OPS; ++r) wu[r] = wg[r * 64];
    }
#define MH_HW_U(J, T, CB, PC) __builtin_bit_cast(f16x8, wu[(((J) * 3 + (T)) * 2 + (CB)) * 2 + (PC)])

    // ---- transform operands of this lane: tile row rho = lane & 15 (rows 0-3: ty 0, tx 0-3; 4-11: ty 1, tx 0-7; 12-15: ty 0, tx 4-7 -- the 16-byte reads of every
    // lane group of ds_read_b128 then cover the 64 banks once), channels 4 kg .. +3 and 16 + 4 kg .. +3 with kg = lane >> 4 -------------------------------------
    const int rho = lane & 15, kg = lane >> 4;
    const int tty = (rho >= 4 && rho < 12) ? 1 : 0, ttx = rho < 4 ? rho : rho < 12 ? rho - 4 : rho - 8;
    // row i of B^T d B:  i = 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3
    const int ra = wave == 0 ? 0 : wave == 2 ? 2 : 1, rb = wave == 0 ? 2 : wave == 1 ? 2 : wave == 2 ? 1 : 3;
    const float sgn = wave == 1 ? 1.0f : -1.0f;
    const f32x4 sgn4 = {sgn, sgn, sgn, sgn};
    const int ab0 = ((2 * tty + ra) * HWG_RX + 2 * ttx) * HWG_PB + kg * 16;
    const int ab1 = ((2 * tty + rb) * HWG_RX + 2 * ttx) * HWG_PB + kg * 16;

    f32x4 acc[3][4][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[s][j][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- finishing role of this wave: cout block fcb = wave >> 1, register pair frp = wave & 1 of every lane's four tile rows -------------------------------
    const int fcb = wave >> 1, frp = wave & 1;
    const int g4 = lane >> 4;                                  // D rows 4 g4 + r  ->  g4 0: ty 0, tx 0-3; 1: ty 1, tx 0-3; 2: ty 1, tx 4-7; 3: ty 0, tx 4-7
    const int fty = (g4 == 1 || g4 == 2) ? 1 : 0, ftx = (g4 >= 2 ? 4 : 0) + 2 * frp;
    const int co_l = fcb * 16 + (lane & 15), co = cg * HWG_CN + co_l;
    const float bco = bias ? bias[co] : 0.0f;
    float inv_a, inv_b;
    {
        const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - e_in;      // wtail[1] = the weights' power-of-two scale
        const int t1_ = t_ / 2, t2_ = t_ - t1_;
        inv_a = poisoned ? __uint_as_float(0x7fc00000u) : __uint_as_float((unsigned)(t1_ + 127) << 23);
        inv_b = __uint_as_float((unsigned)(t2_ + 127) << 23);
    }
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out.data + (long long)n * out.n_stride + (long long)(cg * HWG_CN) * DHW, 0, (int)(HWG_CN * DHW * 4), 0x00020000);
    // byte offset of output row a' = 0 of this lane inside plane 0 of its cout (row a' = 1: + 4 W)
    const unsigned ooff = 4u * (unsigned)((long long)co_l * DHW + (long long)(y0 + 2 * fty) * W + x0 + 2 * ftx);
    Stat run;
    run.n = 0.0f; run.mean = 0.0f; run.m2 = 0.0f;
    const long long PHW = (long long)(H / 2) * (W / 2), PDHW = (long long)(D / 2) * PHW;
    const auto pxrs = __builtin_amdgcn_make_buffer_rsrc(POOL ? pmax + (long long)n * pool_n_stride + (long long)(cg * HWG_CN) * PDHW : out.data, 0, POOL ? (int)(HWG_CN * PDHW * 4) : 0, 0x00020000);
    const auto pnrs = __builtin_amdgcn_make_buffer_rsrc(POOL ? pmin + (long long)n * pool_n_stride + (long long)(cg * HWG_CN) * PDHW : out.data, 0, POOL ? (int)(HWG_CN * PDHW * 4) : 0, 0x00020000);
    const unsigned poff = 4u * (unsigned)((long long)co_l * PDHW + (long long)((y0 >> 1) + fty) * (W / 2) + (x0 >> 1) + ftx);
    f32x2 hmx = {0.0f, 0.0f}, hmn = {0.0f, 0.0f};           // POOL: the even plane's in-plane maxima / minima

    // the old values of the plane being finished (ACC): requested before the plane's matrix instructions, consumed after its barrier
    f32x4 pv0 = {0.0f, 0.0f, 0.0f, 0.0f}, pv1 = pv0;
#define MH_HW_PV_LOAD(Z)                                                                              \
    if (ACC) {                                                                                        \
        const unsigned po_ = ooff + (unsigned)(Z) * (unsigned)(HW * 4);                               \
        pv0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(orsrc, po_, 0, 0));     \
        pv1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(orsrc, po_ + 4u * (unsigned)W, 0, 0)); \
    }

    // ---- one input plane: transform, 72 matrix instructions (S0 = the fresh set of output plane p + 1, S1 = plane p, S2 = plane p - 1) -------------------------
#define MH_HW_MM(S, J, CB, A, B, FRESH)                                                               \
    acc[S][J][CB] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A), B, (FRESH) ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc[S][J][CB], 0, 0, 0);
#define MH_HW_PLANE(S0, S1, S2)                                                                       \
    {                                                                                                 \
        const char* const db_ = ds + cur * HWG_DB;                                                    \
        u32x4 ah_[4], al_[4];                                                                         \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                               \
            f32x4 r_[4];                                                                              \
            _Pragma("unroll") for (int bq = 0; bq < 4; ++bq) {                                        \
                const f32x4 u_ = *reinterpret_cast<const f32x4*>(db_ + ab0 + bq * HWG_PB + h * 64);   \
                const f32x4 v_ = *reinterpret_cast<const f32x4*>(db_ + ab1 + bq * HWG_PB + h * 64);   \
                r_[bq] = __builtin_elementwise_fma(v_, sgn4, u_);                                     \
            }                                                                                         \
            const f32x4 v0_ = r_[0] - r_[2], v1_ = r_[1] + r_[2], v2_ = r_[2] - r_[1], v3_ = r_[1] - r_[3]; \
            u32x2 hh_, ll_;                                                                           \
            hw_split4(v0_, hh_, ll_); ah_[0][2 * h] = hh_[0]; ah_[0][2 * h + 1] = hh_[1]; al_[0][2 * h] = ll_[0]; al_[0][2 * h + 1] = ll_[1]; \
            hw_split4(v1_, hh_, ll_); ah_[1][2 * h] = hh_[0]; ah_[1][2 * h + 1] = hh_[1]; al_[1][2 * h] = ll_[0]; al_[1][2 * h + 1] = ll_[1]; \
            hw_split4(v2_, hh_, ll_); ah_[2][2 * h] = hh_[0]; ah_[2][2 * h + 1] = hh_[1]; al_[2][2 * h] = ll_[0]; al_[2][2 * h + 1] = ll_[1]; \
            hw_split4(v3_, hh_, ll_); ah_[3][2 * h] = hh_[0]; ah_[3][2 * h + 1] = hh_[1]; al_[3][2 * h] = ll_[0]; al_[3][2 * h + 1] = ll_[1]; \
        }                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
            _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {                                        \
                MH_HW_MM(S0, j, cb, ah_[j], MH_HW_U(j, 0, cb, 0), true)  MH_HW_MM(S1, j, cb, ah_[j], MH_HW_U(j, 1, cb, 0), false) MH_HW_MM(S2, j, cb, ah_[j], MH_HW_U(j, 2, cb, 0), false) \
                MH_HW_MM(S0, j, cb, al_[j], MH_HW_U(j, 0, cb, 0), false) MH_HW_MM(S1, j, cb, al_[j], MH_HW_U(j, 1, cb, 0), false) MH_HW_MM(S2, j, cb, al_[j], MH_HW_U(j, 2, cb, 0), false) \
                MH_HW_MM(S0, j, cb, ah_[j], MH_HW_U(j, 0, cb, 1), false) MH_HW_MM(S1, j, cb, ah_[j], MH_HW_U(j, 1, cb, 1), false) MH_HW_MM(S2, j, cb, ah_[j], MH_HW_U(j, 2, cb, 1), false) \
            }                                                                                         \
    }
    // a plane that does not exist (p = -1, p = D): nothing is added, the fresh set must still start from zero
#define MH_HW_SKIP(S0)                                                                                \
    { _Pragma("unroll") for (int j = 0; j < 4; ++j) _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) acc[S0][j][cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    // the wave-local half of the inverse transform of the completed set: Z[b' = 0] = M0 + M1 + M2, Z[b' = 1] = M1 - M2 - M3, as 16-byte items
    // {Z0[r], Z1[r], Z0[r + 1], Z1[r + 1]} = four consecutive x of one output row once the rows are combined
#define MH_HW_ZOUT(S2)                                                                                \
    {                                                                                                 \
        char* const zb_ = zs_ + zc * HWG_ZB + lane * 16;                                              \
        _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {                                            \
            const f32x4 z0_ = (acc[S2][0][cb] + acc[S2][1][cb]) + acc[S2][2][cb];                     \
            const f32x4 z1_ = (acc[S2][1][cb] - acc[S2][2][cb]) - acc[S2][3][cb];                     \
            _Pragma("unroll") for (int rp = 0; rp < 2; ++rp)                                          \
                *reinterpret_cast<f32x4*>(zb_ + ((wave * 2 + cb) * 2 + rp) * 1024) = f32x4{z0_[2 * rp], z1_[2 * rp], z0_[2 * rp + 1], z1_[2 * rp + 1]}; \
        }                                                                                             \
    }
    // after the barrier: rows a' = 0: Z_0 + Z_1 + Z_2, a' = 1: Z_1 - Z_2 - Z_3 of this wave's quarter; scale back, bias, (old values,) stores, statistics, pooling
#define MH_HW_FINISH(ZQ)                                                                              \
    {                                                                                                 \
        const char* const zb_ = zs_ + zc * HWG_ZB + lane * 16 + (fcb * 2 + frp) * 1024;               \
        const f32x4 q0_ = *reinterpret_cast<const f32x4*>(zb_), q1_ = *reinterpret_cast<const f32x4*>(zb_ + 4096);       \
        const f32x4 q2_ = *reinterpret_cast<const f32x4*>(zb_ + 8192), q3_ = *reinterpret_cast<const f32x4*>(zb_ + 12288); \
        f32x4 o0_ = ((q0_ + q1_) + q2_) * inv_a * inv_b + bco, o1_ = ((q1_ - q2_) - q3_) * inv_a * inv_b + bco; \
        if (ACC) { o0_ += pv0; o1_ += pv1; }                                                          \
        const unsigned so_ = ooff + (unsigned)(ZQ) * (unsigned)(HW * 4);                              \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0_), orsrc, so_, 0, 0);     \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1_), orsrc, so_ + 4u * (unsigned)W, 0, 0); \
        if (STATS) {                                                                                  \
            const float sum_ = ((o0_[0] + o0_[1]) + (o0_[2] + o0_[3])) + ((o1_[0] + o1_[1]) + (o1_[2] + o1_[3])); \
            const float mean_ = sum_ * 0.125f;                                                        \
            const f32x4 d0_ = o0_ - mean_, d1_ = o1_ - mean_;                                         \
            const f32x4 s0_ = d0_ * d0_, s1_ = d1_ * d1_;                                             \
            Stat loc_;                                                                                \
            loc_.n = 8.0f; loc_.mean = mean_;                                                         \
            loc_.m2 = ((s0_[0] + s0_[1]) + (s0_[2] + s0_[3])) + ((s1_[0] + s1_[1]) + (s1_[2] + s1_[3])); \
            run = stat_merge_nb(run, loc_);                                                           \
        }                                                                                             \
        if (POOL) {                                                                                   \
            const f32x2 ym_ = {fmaxf(fmaxf(o0_[0], o0_[1]), fmaxf(o1_[0], o1_[1])), fmaxf(fmaxf(o0_[2], o0_[3]), fmaxf(o1_[2], o1_[3]))}; \
            const f32x2 yn_ = {fminf(fminf(o0_[0], o0_[1]), fminf(o1_[0], o1_[1])), fminf(fminf(o0_[2], o0_[3]), fminf(o1_[2], o1_[3]))}; \
            if ((ZQ) & 1) {                                                                           \
                const f32x2 fm_ = {fmaxf(ym_[0], hmx[0]), fmaxf(ym_[1], hmx[1])}, fn_ = {fminf(yn_[0], hmn[0]), fminf(yn_[1], hmn[1])}; \
                const unsigned po_ = poff + (unsigned)((ZQ) >> 1) * (unsigned)(PHW * 4);              \
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, fm_), pxrs, po_, 0, 0); \
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, fn_), pnrs, po_, 0, 0); \
            }                                                                                         \
            hmx = ym_; hmn = yn_;                                                                     \
        }                                                                                             \
    }
    // one iteration of the march at input plane P with the set roles (S0, S1, S2)
#define MH_HW_ITER(P, S0, S1, S2)                                                                     \
    {                                                                                                 \
        const int p_ = (P);                                                                           \
        const bool emit_ = p_ - 1 >= zs;                                                              \
        if (emit_) MH_HW_PV_LOAD(p_ - 1)                                                              \
        if (p_ >= p_first && p_ <= p_last) {                                                          \
            MH_HW_PLANE(S0, S1, S2)                                                                   \
            MH_HW_CONV(cur ^ 1)                                                                       \
            MH_HW_LDX MH_HW_ADV                                                                       \
            cur ^= 1;                                                                                 \
        } else MH_HW_SKIP(S0)                                                                         \
        if (emit_) MH_HW_ZOUT(S2)                                                                     \
        __syncthreads();                                                                              \
        if (emit_) { MH_HW_FINISH(p_ - 1) zc ^= 1; }                                                  \
    }

    // prologue: plane p_first into buffer 0, the loads of the next plane in flight
    int cur = 0, zc = 0;
    MH_HW_LDX MH_HW_ADV
    __syncthreads();                  // the zeroed buffers
    MH_HW_CONV(0)
    MH_HW_LDX MH_HW_ADV
    __syncthreads();
    for (int p = zs - 1; p <= ze; p += 3) {
        MH_HW_ITER(p, 0, 1, 2)
        if (p + 1 > ze) break;
        MH_HW_ITER(p + 1, 2, 0, 1)
        if (p + 2 > ze) break;
        MH_HW_ITER(p + 2, 1, 2, 0)
    }
#undef MH_HW_ITER
#undef MH_HW_FINISH
#undef MH_HW_ZOUT
#undef MH_HW_SKIP
#undef MH_HW_PLANE
#undef MH_HW_MM
#undef MH_HW_PV_LOAD
#undef MH_HW_U
#undef MH_HW_ADV
#undef MH_HW_CONV
#undef MH_HW_LDX

    if (STATS) {
        // a cout's voxels sit in the four lane groups (lane >> 4) of the two waves that share its cout block
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
            Stat ot;
            ot.n = __shfl_xor(run.n, o); ot.mean = __shfl_xor(run.mean, o); ot.m2 = __shfl_xor(run.m2, o);
            run = (lane & o) == 0 ? stat_merge(run, ot) : stat_merge(ot, run);
        }
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        if (lane < 16) { red[(wave * 16 + lane) * 3] = run.n; red[(wave * 16 + lane) * 3 + 1] = run.mean; red[(wave * 16 + lane) * 3 + 2] = run.m2; }
        __syncthreads();
        if (tid < HWG_CN) {
            const int cb = tid >> 4, c16 = tid & 15;
            Stat a, bq;
            a.n = red[((2 * cb) * 16 + c16) * 3]; a.mean = red[((2 * cb) * 16 + c16) * 3 + 1]; a.m2 = red[((2 * cb) * 16 + c16) * 3 + 2];
            bq.n = red[((2 * cb + 1) * 16 + c16) * 3]; bq.mean = red[((2 * cb + 1) * 16 + c16) * 3 + 1]; bq.m2 = red[((2 * cb + 1) * 16 + c16) * 3 + 2];
            const Stat st = stat_merge(a, bq);
            float* rec = stats + (((long long)n * Cout + cg * HWG_CN + tid) * nblk + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}

// Weight preparation.  The scale: conv3d_k3_h2_scale_kernel's power of two, a quarter of it (|G g G^T| <= 2.25 max |g|).
__global__ void conv3d_k3_h2w_scale_fix_kernel(float* __restrict__ tail) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { tail[0] *= 4.0f; tail[1] *= 0.25f; }
}
// w [Cout][32][3][3][3] -> [cout group][wave i][operand ((j 3 + t) 2 + cout block) 2 + piece][lane = kg 16 + cout % 16][8 halves]: the lane's K elements are
// channels 4 kg .. +3 (e = 0..3) and 16 + 4 kg .. +3 (e = 4..7).  One thread per (cout, cin); U = G g G^T in fp64.
__global__ void __launch_bounds__(256)
conv3d_k3_h2w_pack_kernel(const float* __restrict__ w, int Cin, int Cout, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const float s = tail[1];
    const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int cg = co / HWG_CN, cb = (co % HWG_CN) / 16, nn = co % 16;
    const int kg = ci < 16 ? ci / 4 : (ci - 16) / 4, e = ci < 16 ? ci % 4 : 4 + (ci - 16) % 4;
    const int ln = kg * 16 + nn;
    for (int t = 0; t < 3; ++t) {
        double g[3][3], tm[4][3];
        for (int ky = 0; ky < 3; ++ky)
            for (int kx = 0; kx < 3; ++kx) g[ky][kx] = (double)w[((long long)co * Cin + ci) * 27 + t * 9 + ky * 3 + kx];
        for (int i = 0; i < 4; ++i)
            for (int kx = 0; kx < 3; ++kx) tm[i][kx] = G[i][0] * g[0][kx] + G[i][1] * g[1][kx] + G[i][2] * g[2][kx];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const double u = tm[i][0] * G[j][0] + tm[i][1] * G[j][1] + tm[i][2] * G[j][2];
                _Float16 pc[2];
                h2_split((float)(u * (double)s), pc[0], pc[1]);
                for (int p = 0; p < 2; ++p) {
                    const long long r = (((long long)(cg * 4 + i) * HWG_OPS) + ((j * 3 + t) * 2 + cb) * 2 + p) * 64 + ln;
                    packed[r * 8 + e] = pc[p];
                }
            }
    }
}

}  // namespace mh
