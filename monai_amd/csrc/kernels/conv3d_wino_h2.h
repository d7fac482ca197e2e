// Conv3d 3x3x3 / stride 1 / zero padding 1 for 32 input channels: in-plane Winograd F(2x2, 3x3) IN FRONT OF the two-piece split-precision product of
// conv3d_h2.h -- 2.25x fewer matrix instructions per output voxel (16 transform positions x 3 z-taps per 2 x 2 outputs instead of 27 taps x 4), z-streaming.
// Same reference op, "normalise on load" contract, record / bound contract and tolerance class as conv3d_h2.h (nn.Conv3d of `Convolution`,
// monai/networks/blocks/convolutions.py:98-171, fed by the previous block's deferred InstanceNorm + LeakyReLU).
//
// Why this shape (round 6): the transformed weights of 32 x 32 channels are 192 KB as hi / lo fp16 pieces -- more than the LDS, and streaming them from L2
// would cost more than the matrix instructions save.  They live in REGISTERS instead: a workgroup is 8 waves (two per SIMD, 256 registers each); wave jp 4 + i owns
// row i and column pair jp of the 4 x 4 transform positions: its B operands U[(i, j)][z-tap t][32 cin][32 cout] as hi / lo pieces are 2 x 3 x 2 cout blocks x 2 pieces
// = 24 operands of v_mfma_f32_16x16x32_f16 = 96 registers, loaded once per workgroup.  M = 16 tiles (a 4 x 16 output region = 2 x 8 tiles of 2 x 2), N = 16 couts per
// block, K = all 32 input channels in ONE instruction; the transform-domain sums of the three live output planes are 2 x 3 x 2 x 4 = 48 registers.
//
// Per input plane p: (1) the activated, scaled fp32 plane region [6 x 18 positions][32 channels] sits in one of three LDS buffers (144-byte position pitch: the 16-byte
// operand reads of a lane group cover all 64 banks once; staged two planes ahead, a channel quad per wave); (2) the wave reads the two rows and three columns of every
// tile's 4 x 4 patch that its positions need (row i of B^T d B: i = 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3 -- wave-uniform (row, row, sign); the columns likewise),
// forms its two V[i][j] in packed fp32 and splits each into hi / lo fp16 (the split comes AFTER the transform: the sums are exact to fp32 rounding); (3) 36 matrix
// instructions: for both positions, every z-tap t and cout block  acc[plane p + 1 - t] += Vh Uh + Vl Uh + Vh Ul;  (4) when output plane p - 1 is complete the wave's two
// terms of the inverse transform's column half go to LDS (no register shuffling: s = M_a + M_b and M_b as plain 16-byte items), and two iterations later every wave
// finishes a quarter of a row pair  Y[a'][b'] = sum_i A^T[a'][i] Z[i][b']  from 8-byte reads (scale back, bias, statistics as shifted sums, one 16-byte store per row).
// An iteration is two barrier-separated phases -- transform | matrix instructions + staging + finishing + Z items -- and the jp = 1 waves run one phase behind the jp = 0
// waves, so the two waves of a SIMD are never in the same kind of phase.  The accumulator sets rotate by NAME (the march is unrolled three times); a fresh set starts from
// the instruction's zero C operand.  ACC: out += ... (old values requested at the start of the iteration's vector phase); POOL: MaxPool3d(2) of the result leaves with it
// (the 2 x 2 window of a tile is the lane pair l, l ^ 32: one v_permlane32_swap_b32).
//
// What bounds it (profiles/r06_h2w_variants.txt, r06_valu_rates.txt): the instruction COUNT.  A wave issues a vector instruction every ~8 cycles whatever its width, and every
// vector / LDS / scalar instruction takes SIMD time away from the matrix pipe (26-28 % busy); hence the widest forms everywhere (packed fp32, three-instruction pair split,
// median-of-three activation, 16-byte LDS items).  Measured forms that did not beat this one: four waves with 512 registers, sixteen waves with 128, a six-fold unrolled
// march with compile-time buffers (tools/experiments/h2w_v5, h2w_v6).
//
// Weight preparation (conv3d_k3_h2w_pack_kernel): U = G g G^T per z-tap in fp64, scaled by a power of two (a quarter of conv3d_h2.h's: |U| <= 2.25 max |g|), split
// into hi / lo fp16, stored in the waves' register order [cout group][wave][24 operands][64 lanes][8 halves].  The input scale leaves two more bits of headroom than
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
constexpr int HWG_ZB = 32 * 1024;                           // bytes per Z exchange buffer: [8 waves = 2 column halves x 4 rows i][2 blocks][2 register pairs][64 lanes][16 bytes]
constexpr int HWG_OPS = 24;                                 // B operands per wave: [slot][t][cout block][piece]
constexpr int HWG_CIN = 32, HWG_CN = 32;
constexpr int HWG_WAVES = 8;                                // waves of a workgroup
constexpr unsigned HWG_DROP = 0x80000000u;                 // a buffer offset beyond every buffer: the hardware drops the store / returns zeros

__device__ __forceinline__ void hw_split4(const f32x4 v, u32x2& hi, u32x2& lo) {
    _Float16 h_[4], l_[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // h2_split with the residual kept scalar (see hw_fma4)
        h_[i] = (_Float16)v[i];
        float r = v[i] - (float)h_[i];
        MH_OPAQUE(r);
        l_[i] = (_Float16)r;
    }
    const f16x2 h01 = {h_[0], h_[1]}, h23 = {h_[2], h_[3]}, l01 = {l_[0], l_[1]}, l23 = {l_[2], l_[3]};
    hi = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
    lo = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
}

// What bounds this kernel's vector phases is the NUMBER of instructions a wave issues (one wave issues a vector instruction every ~9 cycles whatever its width:
// tools/ubench/valu_rates.hip, profiles/r06_valu_rates.txt), so the element-wise work is written in the widest forms the ISA has: packed fp32 (v_pk_fma_f32,
// v_pk_add_f32, v_pk_mul_f32: two lanes' worth per instruction) and the pairwise split (one v_cvt_pk_f16_f32 for two high pieces, one v_fma_mix_f32 per residual
// -- the fp16 half is widened inside the instruction -- and one v_cvt_pk_f16_f32 for the two low pieces: 2 instead of 4 instructions per value, the same bits).
__device__ __forceinline__ f32x4 hw_fma4(const f32x4 a, const float s, const f32x4 c) { return __builtin_elementwise_fma(a, f32x4{s, s, s, s}, c); }
__device__ __forceinline__ void hw_split4p(const f32x4 v, const float m1, u32x2& hi, u32x2& lo) {
#ifdef MH_SIMT_EMULATOR
    f16x2 h01, l01, h23, l23;
    h2_split_pair(v[0], v[1], m1, h01, l01);
    h2_split_pair(v[2], v[3], m1, h23, l23);
    hi = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
    lo = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
#else
    // three instructions per pair: both high pieces, then each residual v - hi rounded to fp16 straight into its half of the low register (the fp16 source half is
    // picked by op_sel; hipcc's own code for h2_split_pair converts every high piece a second time on its own to have it in a low half)
    unsigned h0, h1, l0, l1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h0) : "v"(v[0]), "v"(v[1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h1) : "v"(v[2]), "v"(v[3]));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h0), "v"(m1), "v"(v[0]));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h1), "v"(m1), "v"(v[2]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l0) : "v"(h0), "v"(m1), "v"(v[1]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l1) : "v"(h1), "v"(m1), "v"(v[3]));
    hi = u32x2{h0, h1};
    lo = u32x2{l0, l1};
#endif
}
// the value of lane ^ 32 (hi = this lane is in the upper half): one v_permlane32_swap_b32 of two copies -- {x.lo, x.lo} and {x.hi, x.hi} -- and a select; no LDS round trip
__device__ __forceinline__ float hw_xchg32(const float x, const int hi) {
#ifdef MH_SIMT_EMULATOR
    (void)hi;
    return __shfl_xor(x, 32);
#else
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x), false, false);
    return __builtin_bit_cast(float, hi ? r[0] : r[1]);
#endif
}
// keeps a wave-uniform two-way choice a BRANCH (the optimiser otherwise turns it into one v_cndmask per element)
#ifdef MH_SIMT_EMULATOR
#define MH_HW_KEEP_BRANCH ((void)0)
#else
#define MH_HW_KEEP_BRANCH asm volatile("" ::: "memory")
#endif

template <bool STATS, bool ACC = false, bool POOL = false>
__global__ void __launch_bounds__(512)
conv3d_k3_h2w_kernel(Tensor in, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias, Tensor out,
                     float* __restrict__ stats, int bxn, int byn, int zchunk, unsigned nblk, float* __restrict__ pmax, float* __restrict__ pmin,
                     long long pool_n_stride) {
    __shared__ __attribute__((aligned(16))) char smem[3 * HWG_DB + 2 * HWG_ZB];
    __shared__ unsigned bound_s[4];
    char* const ds = smem;
    char* const zs_ = smem + 3 * HWG_DB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, jp = wave >> 2;                    // row of transform positions, pair of columns (jp 0: j = 0, 1; jp 1: j = 3, 2)
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

    // ---- staging: wave w converts channel quad w (channels 4 w .. 4 w + 3); lane l < 54 owns the position PAIR (row l / 9, columns 2 (l % 9), + 1) of the
    // 6 x 18 region: one 8-byte load per channel (half the vector-memory instructions of one 4-byte load per position: the texture-address units are the busiest
    // unit of this kernel, profiles/r06_pmc_h2w_mem.txt), two 16-byte LDS cells.  A pair whose first column lies left of the volume (x = -1) or whose second lies
    // right of it (x = W) is moved one column inwards: the extra element rewrites a neighbour's cell with the same value, the padding cell stays zero, and no
    // load ever leaves its row.  Rows outside the volume and idle lanes load offset 0 and write the dump cell.
    unsigned soff;                    // byte offset of the pair's first element inside a channel plane
    int loff[2];                      // byte offsets of the two elements' cells in an input buffer (without the quad)
    {
        const int pr = min(lane, 53);
        const int ly = pr / 9;
        int lx = 2 * (pr - 9 * ly);
        const int gy = y0 + ly - 1;
        int gx = x0 + lx - 1;
        if (gx < 0) { gx += 1; lx += 1; }                       // (x0 = 0, first pair)
        if (gx + 1 >= W) { gx -= 1; lx -= 1; }                   // (x0 + 16 = W, last pair)
        const bool ok = lane < 54 && gy >= 0 && gy < H;
        soff = ok ? 4u * (unsigned)(gy * W + gx) : 0u;
        loff[0] = (ok ? ly * HWG_RX + lx : HWG_NP) * HWG_PB;       // zero padding / idle lanes: the dump cell; the real cell stays zero
        loff[1] = (ok ? ly * HWG_RX + lx + 1 : HWG_NP) * HWG_PB;
    }
    for (int i = tid; i < (3 * HWG_DB + 2 * HWG_ZB) / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);      // the Z buffers too: a plane that does not exist is finished from zeros, not from NaN bit patterns
    // records of this wave's four channels {alpha, beta, slope} (constant over the march) and the sample's largest bound
    f32x4 nra, nrb, nrs;
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
        for (int i = 0; i < 4; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * (4 * wave + i));
            nra[i] = a.x * p_; nrb[i] = a.y * p_; nrs[i] = a.z;
        }
    }
    // the activation t > 0 ? t : t s in ONE branch-free instruction: for s <= 1 it is max(t, t s), for s > 1 min(t, t s) -- the median of (t, t s, +inf | -inf).  (A NaN
    // input never gets here as a result: its record's bound is non-finite and the whole sample is poisoned.)
    f32x4 nrk;
#pragma unroll
    for (int i = 0; i < 4; ++i) nrk[i] = nrs[i] <= 1.0f ? __builtin_inff() : -__builtin_inff();
    // ---- staging loads: ONE buffer descriptor over the wave's four channel volumes; the plane advance lives in the lanes' byte offsets (two additions per
    // plane), the channel in the instruction's scalar offset (three constants) -- no descriptor arithmetic in the march ---------------------------------------
    const auto xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in.data + (long long)n * in.n_stride + (long long)(4 * wave) * DHW), 0, 0x7fffffff, 0x00020000);
    const unsigned hw4 = (unsigned)(HW * 4);
    unsigned voff = soff + (unsigned)p_first * hw4;
    const unsigned cso1 = (unsigned)(DHW * 4), cso2 = 2u * cso1, cso3 = 3u * cso1;
    f32x2 xin[4];                     // [channel of the quad][element of the pair]
#define MH_HW_LDX                                                                                     \
    {                                                                                                 \
        xin[0] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, 0u, 0));   \
        xin[1] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, cso1, 0)); \
        xin[2] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, cso2, 0)); \
        xin[3] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, voff, cso3, 0)); \
    }
#define MH_HW_CONV(DBO)                                                                               \
    {                                                                                                 \
        char* const db_ = ds + (DBO) + 16 * wave;                                                     \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
            f32x4 y_;                                                                                 \
            const f32x4 t_ = __builtin_elementwise_fma(f32x4{xin[0][j], xin[1][j], xin[2][j], xin[3][j]}, nra, nrb), u_ = t_ * nrs; \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) y_[i] = __builtin_amdgcn_fmed3f(t_[i], u_[i], nrk[i]);      /* slope <= 1: max(t, t s), slope > 1: min(t, t s)  ==  t > 0 ? t : t s */ \
            *reinterpret_cast<f32x4*>(db_ + loff[j]) = y_;                                            \
        }                                                                                             \
    }
    int staged = p_first;             // plane whose loads are in the registers
#define MH_HW_ADV { voff += staged < p_last ? hw4 : 0u; staged += 1; }

    // ---- B operands: the wave's 24 register sets [slot][z-tap][cout block][piece] -------------------------------------------------------------------------
    u32x4 wu[HWG_OPS];
    {
        const u32x4* wg = reinterpret_cast<const u32x4*>(wp) + ((long long)(cg * 8 + wave) * HWG_OPS) * 64 + lane;
#pragma unroll
        for (int r = 0; r < HWG_OPS; ++r) wu[r] = wg[r * 64];
    }
#define MH_HW_U(SL, T, CB, PC) __builtin_bit_cast(f16x8, wu[(((SL) * 3 + (T)) * 2 + (CB)) * 2 + (PC)])

    // ---- transform operands of this lane: tile row rho = lane & 15 (rows 0-3: ty 0, tx 0-3; 4-11: ty 1, tx 0-7; 12-15: ty 0, tx 4-7 -- the 16-byte reads of every
    // lane group of ds_read_b128 then cover the 64 banks once), channels 4 kg .. +3 and 16 + 4 kg .. +3 with kg = lane >> 4 -------------------------------------
    const int rho = lane & 15, kg = lane >> 4;
    const int tty = (rho >= 4 && rho < 12) ? 1 : 0, ttx = rho < 4 ? rho : rho < 12 ? rho - 4 : rho - 8;
    // row i of B^T d B:  i = 0: d0 - d2,  1: d1 + d2,  2: d2 - d1,  3: d1 - d3;  the wave's columns: jp, jp + 1, jp + 2 of the patch
    const int ra = wi == 0 ? 0 : wi == 2 ? 2 : 1, rb = wi == 0 ? 2 : wi == 1 ? 2 : wi == 2 ? 1 : 3;
    const float sgn = wi == 1 ? 1.0f : -1.0f;
    const float fsg = jp ? -1.0f : 1.0f;
    const float m1 = h2_minus_one();
    const int ab0 = ((2 * tty + ra) * HWG_RX + 2 * ttx + jp) * HWG_PB + kg * 16;
    const int ab1 = ((2 * tty + rb) * HWG_RX + 2 * ttx + jp) * HWG_PB + kg * 16;

    f32x4 acc[3][2][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[s][j][c] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- finishing role: wave = (cout block fcb, register pair frp, half fh of the 64 items); lanes 0-31 finish output row a' = 0 of items 32 fh + (lane & 31),
    // lanes 32-63 row a' = 1 of the same items (the 2 x 2 pooling window of a tile is then the lane pair l, l ^ 32) ---------------------------------------------
    const int fcb = wave >> 2, frp = (wave >> 1) & 1, fh = wave & 1;
    const int il = 32 * fh + (lane & 31), fa = lane >> 5;
    const int g4 = il >> 4;                                    // D rows 4 g4 + r  ->  g4 0: ty 0, tx 0-3; 1: ty 1, tx 0-3; 2: ty 1, tx 4-7; 3: ty 0, tx 4-7
    const int fty = (g4 == 1 || g4 == 2) ? 1 : 0, ftx = (g4 >= 2 ? 4 : 0) + 2 * frp;
    const int co_l = fcb * 16 + (il & 15), co = cg * HWG_CN + co_l;
    const float bco = bias ? bias[co] : 0.0f;
    const float fs2 = fa ? -1.0f : 1.0f;
    // Z exchange: wave w = jp 4 + i leaves, per cout block, s = slot 0 + slot 1 and a1 = slot 1 of its completed set as two 16-byte items per lane (no register
    // shuffling): [parity][wave][cout block][s | a1][lane][4 tile rows].  With (M0, M1) in the jp 0 wave and (M3, M2) in the jp 1 wave of row i:
    // Z[b' = 0] = M0 + M1 + M2 = s(jp 0) + a1(jp 1),  Z[b' = 1] = M1 - M2 - M3 = a1(jp 0) - s(jp 1).  The finishing lane reads the register pair frp of its item.
    const int zwr = wave * 4096 + lane * 16;
    const int zrd = fa * 4096 + fcb * 2048 + il * 16 + frp * 8;         // row i0 = a' of the jp 0 half (jp 1: + 16384; row + 1: + 4096; a1: + 1024)
    float inv_a, inv_b;
    {
        const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - e_in;      // wtail[1] = the weights' power-of-two scale
        const int t1_ = t_ / 2, t2_ = t_ - t1_;
        inv_a = poisoned ? __uint_as_float(0x7fc00000u) : __uint_as_float((unsigned)(t1_ + 127) << 23);
        inv_b = __uint_as_float((unsigned)(t2_ + 127) << 23);
    }
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out.data + (long long)n * out.n_stride + (long long)(cg * HWG_CN) * DHW, 0, (int)(HWG_CN * DHW * 4), 0x00020000);
    const unsigned ooff = 4u * (unsigned)((long long)co_l * DHW + (long long)(y0 + 2 * fty + fa) * W + x0 + 2 * ftx);
    float pivot = 0.0f;                                      // statistics of this lane's outputs as shifted sums (see MH_HW_FINISH)
    bool have_c = false;
    f32x4 s1 = {0.0f, 0.0f, 0.0f, 0.0f}, s2 = s1;
    const long long PHW = (long long)(H / 2) * (W / 2), PDHW = (long long)(D / 2) * PHW;
    // POOL: lanes 0-31 store the maxima, lanes 32-63 the minima of their tile pair
    const auto prs = __builtin_amdgcn_make_buffer_rsrc(POOL ? (fa ? pmin : pmax) + (long long)n * pool_n_stride + (long long)(cg * HWG_CN) * PDHW : out.data, 0, POOL ? (int)(HWG_CN * PDHW * 4) : 0, 0x00020000);
    const unsigned poff = 4u * (unsigned)((long long)co_l * PDHW + (long long)((y0 >> 1) + fty) * (W / 2) + (x0 >> 1) + ftx);
    const float pk_ = fa ? -__builtin_inff() : __builtin_inff();
    f32x2 hm = {0.0f, 0.0f};                                 // POOL: the even plane's in-plane maxima (lanes 0-31) / minima (lanes 32-63)
    f32x4 pv = {0.0f, 0.0f, 0.0f, 0.0f};                     // ACC: the old values of the row being finished

    // ---- the pieces ---------------------------------------------------------------------------------------------------------------------------------------
    // transform of the plane at buffer offset DBO: 12 reads of 16 bytes; E = rA - rC (jp 0: V0, jp 1: V3), F = jp 0: rB + rC (V1), jp 1: rB - rA (V2); hi / lo split
    u32x4 ah[2], al[2];
#define MH_HW_XFORM(DBO)                                                                              \
    {                                                                                                 \
        const char* const db_ = ds + (DBO);                                                           \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                               \
            f32x4 r_[3];                                                                              \
            _Pragma("unroll") for (int bq = 0; bq < 3; ++bq) {                                        \
                const f32x4 u_ = *reinterpret_cast<const f32x4*>(db_ + ab0 + bq * HWG_PB + h * 64);   \
                const f32x4 v_ = *reinterpret_cast<const f32x4*>(db_ + ab1 + bq * HWG_PB + h * 64);   \
                r_[bq] = hw_fma4(v_, sgn, u_);                                                        \
            }                                                                                         \
            const f32x4 e_ = hw_fma4(r_[2], m1, r_[0]);                                               \
            f32x4 f_;                                                                                 \
            if (jp) { f_ = hw_fma4(r_[0], m1, r_[1]); MH_HW_KEEP_BRANCH; } else f_ = r_[1] + r_[2];   \
            u32x2 hh_, ll_;                                                                           \
            hw_split4p(e_, m1, hh_, ll_); ah[0][2 * h] = hh_[0]; ah[0][2 * h + 1] = hh_[1]; al[0][2 * h] = ll_[0]; al[0][2 * h + 1] = ll_[1]; \
            hw_split4p(f_, m1, hh_, ll_); ah[1][2 * h] = hh_[0]; ah[1][2 * h + 1] = hh_[1]; al[1][2 * h] = ll_[0]; al[1][2 * h + 1] = ll_[1]; \
        }                                                                                             \
    }
    // 36 matrix instructions (S0 = the fresh set of output plane p + 1, S1 = plane p, S2 = plane p - 1)
#define MH_HW_MM(S, SL, CB, A, B, FRESH)                                                              \
    acc[S][SL][CB] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, A), B, (FRESH) ? f32x4{0.0f, 0.0f, 0.0f, 0.0f} : acc[S][SL][CB], 0, 0, 0);
#define MH_HW_MMS(S0, S1, S2)                                                                         \
    _Pragma("unroll") for (int sl = 0; sl < 2; ++sl)                                                  \
        _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {                                            \
            MH_HW_MM(S0, sl, cb, ah[sl], MH_HW_U(sl, 0, cb, 0), true)  MH_HW_MM(S1, sl, cb, ah[sl], MH_HW_U(sl, 1, cb, 0), false) MH_HW_MM(S2, sl, cb, ah[sl], MH_HW_U(sl, 2, cb, 0), false) \
            MH_HW_MM(S0, sl, cb, al[sl], MH_HW_U(sl, 0, cb, 0), false) MH_HW_MM(S1, sl, cb, al[sl], MH_HW_U(sl, 1, cb, 0), false) MH_HW_MM(S2, sl, cb, al[sl], MH_HW_U(sl, 2, cb, 0), false) \
            MH_HW_MM(S0, sl, cb, ah[sl], MH_HW_U(sl, 0, cb, 1), false) MH_HW_MM(S1, sl, cb, ah[sl], MH_HW_U(sl, 1, cb, 1), false) MH_HW_MM(S2, sl, cb, ah[sl], MH_HW_U(sl, 2, cb, 1), false) \
        }
    // a plane that does not exist (p = -1, p = D): nothing is added, the fresh set must still start from zero
#define MH_HW_SKIP(S0)                                                                                \
    { _Pragma("unroll") for (int sl = 0; sl < 2; ++sl) _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) acc[S0][sl][cb] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    // the wave's part of the inverse transform's column half of the completed set: Z[b' = 0] = M0 + M1 + M2, Z[b' = 1] = M1 - M2 - M3; this wave holds
    // (slot 0, slot 1) = (M0, M1) [jp 0] or (M3, M2) [jp 1]: (M0 + M1, M1) or (M2, -(M2 + M3)); items {Z0[r], Z1[r], Z0[r + 1], Z1[r + 1]}
#define MH_HW_ZOUT(S2, Q)                                                                             \
    {                                                                                                 \
        char* const zb_ = zs_ + ((Q) & 1) * HWG_ZB + zwr;                                             \
        _Pragma("unroll") for (int cb = 0; cb < 2; ++cb) {                                            \
            *reinterpret_cast<f32x4*>(zb_ + cb * 2048) = acc[S2][0][cb] + acc[S2][1][cb];             \
            *reinterpret_cast<f32x4*>(zb_ + cb * 2048 + 1024) = acc[S2][1][cb];                       \
        }                                                                                             \
    }
    // output plane Q (ON: it exists and belongs to this chunk -- otherwise the stores go beyond the buffer and the statistics get weight 0):
    // row a' = 0: Z_0 + Z_1 + Z_2, a' = 1: Z_1 - Z_2 - Z_3 with Z_i = the two column halves' sum; scale back, bias, (old values,) store, statistics, pooling
#define MH_HW_FINISH(Q, ON)                                                                           \
    {                                                                                                 \
        const bool on_ = (ON);                                                                        \
        const char* const zb_ = zs_ + ((Q) & 1) * HWG_ZB + zrd;                                       \
        f32x2 z0_[3], z1_[3];                                                                         \
        _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                               \
            z0_[i] = *reinterpret_cast<const f32x2*>(zb_ + i * 4096) + *reinterpret_cast<const f32x2*>(zb_ + i * 4096 + 16384 + 1024);        \
            z1_[i] = *reinterpret_cast<const f32x2*>(zb_ + i * 4096 + 1024) - *reinterpret_cast<const f32x2*>(zb_ + i * 4096 + 16384);        \
        }                                                                                             \
        const f32x2 fs_ = {fs2, fs2};                                                                 \
        const f32x2 y0_ = __builtin_elementwise_fma(z0_[2], fs_, __builtin_elementwise_fma(z0_[1], fs_, z0_[0])) * inv_a; \
        const f32x2 y1_ = __builtin_elementwise_fma(z1_[2], fs_, __builtin_elementwise_fma(z1_[1], fs_, z1_[0])) * inv_a; \
        f32x4 o_ = {__builtin_fmaf(y0_[0], inv_b, bco), __builtin_fmaf(y1_[0], inv_b, bco), __builtin_fmaf(y0_[1], inv_b, bco), __builtin_fmaf(y1_[1], inv_b, bco)}; \
        if (ACC) o_ += pv;      /* the old values: requested at the start of this iteration's vector phase (a barrier and the transform ago) */ \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o_), orsrc, on_ ? ooff + (unsigned)(Q) * (unsigned)(HW * 4) : HWG_DROP, 0, 0); \
        if (STATS) {        /* sums of the deviations from a pivot (the lane's first value) and of their squares, branch-free: a plane that does not exist has weight 0 */ \
            pivot = (on_ && !have_c) ? o_[0] : pivot;                                                 \
            have_c = have_c || on_;                                                                   \
            const f32x4 d_ = o_ - pivot, dw_ = d_ * (on_ ? 1.0f : 0.0f);                              \
            s1 += dw_;                                                                                \
            s2 = __builtin_elementwise_fma(dw_, d_, s2);                                              \
        }                                                                                             \
        if (POOL && !(HWX_OFF & 32)) {         /* the tile pair's x pairs in-lane; the other row sits in lane ^ 32: lanes 0-31 (maxima) receive the partner's maxima, lanes 32-63 (minima) its minima */ \
            const f32x2 mx_ = {fmaxf(o_[0], o_[1]), fmaxf(o_[2], o_[3])}, mn_ = {fminf(o_[0], o_[1]), fminf(o_[2], o_[3])}; \
            const f32x2 own_ = fa ? mn_ : mx_, snd_ = fa ? mx_ : mn_;                                 \
            const f32x2 rcv_ = {hw_xchg32(snd_[0], fa), hw_xchg32(snd_[1], fa)};                      \
            const f32x2 y_ = {__builtin_amdgcn_fmed3f(own_[0], rcv_[0], pk_), __builtin_amdgcn_fmed3f(own_[1], rcv_[1], pk_)};      /* pk_ = +inf: max, -inf: min */ \
            const f32x2 f_ = {__builtin_amdgcn_fmed3f(y_[0], hm[0], pk_), __builtin_amdgcn_fmed3f(y_[1], hm[1], pk_)}; \
            const unsigned po_ = (on_ && ((Q) & 1) && !(HWX_OFF & 64)) ? poff + (unsigned)((Q) >> 1) * (unsigned)(PHW * 4) : HWG_DROP; \
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, f_), prs, po_, 0, 0);     \
            hm = on_ ? y_ : hm;                                                                       \
        }                                                                                             \
    }
#ifdef HWX_PROF      // tools/ubench/h2w_variants.hip: cycles of the four segments of an iteration, summed per wave, written through `pmax` by workgroup 0
#define MH_HW_T(K) { const long long t_ = __builtin_readcyclecounter(); tacc[K] += t_ - tlast; tlast = t_; }
    long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#else
#define MH_HW_T(K)
#endif
#ifndef HWX_OFF
#define HWX_OFF 0      // ablation switches (bits): 1 no transform, 2 no staging, 4 no finishing, 8 no matrix instructions, 16 no Z exchange, 32 no pooling arithmetic, 64 pooled stores beyond the buffer
#endif
    // what the scheduler may place per gap between two matrix instructions of the matrix phase.  Measured per form (profiles/r06_h2w_ab.txt): the plain form is best left to
    // the compiler's own order (7.07 against 7.22 ms), the accumulating form needs the dealing (8.5 against 11.2 ms: without it the old-value loads end up in front of the
    // conversion's wait), the pooling form does not care (7.98 / 8.02)
#ifndef HWX_DEAL
#define HWX_DEAL 2
#endif
#define MH_HW_DEAL                                                                                    \
    if constexpr (ACC || POOL) {                                                                      \
        _Pragma("unroll") for (int g_ = 0; g_ < 36; ++g_) {                                           \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x006, HWX_DEAL, 0);                                 \
            __builtin_amdgcn_sched_group_barrier(0x230, 1, 0);                                        \
        }                                                                                             \
    }
    // One iteration = two phases, each closed by a barrier: the VECTOR phase (transform of plane P into the operand registers) and the MATRIX phase (36 matrix
    // instructions; dealt out over their gaps: the staging of plane P + 2, the loads of plane P + 3, the finishing of output plane P - 2; then the Z halves of
    // output plane P - 1).  The waves of column pair jp = 1 -- the other wave of every SIMD -- run one phase behind (one barrier before the loop, the jp = 0 waves
    // one more at the end): the two waves of a SIMD are never in the same kind of phase.  With g the global phase count: jp 0 transforms plane P in phase 2 P and
    // multiplies in 2 P + 1, jp 1 in 2 P + 1 and 2 P + 2.  Plane P + 2 is staged in phases 2 P + 1 / 2 P + 2 into the buffer plane P - 1 was last read from in
    // phase 2 P - 1, and first read in phase 2 P + 4.  The Z halves of output plane q are written in phases 2 q + 3 / 2 q + 4, read in 2 q + 5 / 2 q + 6 (iteration
    // q + 2 of either pair) and their buffer (q & 1) is rewritten by plane q + 2 from phase 2 q + 7 on.
#define MH_HW_ITER(P, R)                                                                              \
    {                                                                                                 \
        constexpr int S0 = (3 - (R)) % 3, S1 = (S0 + 1) % 3, S2 = (S0 + 2) % 3;      /* R 0: (0, 1, 2), 1: (2, 0, 1), 2: (1, 2, 0) */ \
        constexpr int db0 = (R) * HWG_DB, db2 = (((R) + 2) % 3) * HWG_DB;            /* the iteration's own buffer and the one being staged: compile-time LDS offsets */ \
        const int p_ = (P);                                                                           \
        const bool valid_ = p_ >= p_first && p_ <= p_last;                                            \
        const int qf_ = p_ - 2;                                                                       \
        if (ACC) { const bool nx_ = qf_ >= zs && qf_ < ze; pv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(orsrc, nx_ ? ooff + (unsigned)qf_ * hw4 : HWG_DROP, 0, 0)); } \
        if (valid_) { if (!(HWX_OFF & 1)) MH_HW_XFORM(db0) }                                          \
        MH_HW_T(0)                                                                                    \
        __syncthreads();                                                                              \
        MH_HW_T(1)                                                                                    \
        if (valid_) {                                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            if (!(HWX_OFF & 2)) { MH_HW_CONV(db2) MH_HW_LDX MH_HW_ADV }                               \
            if (!(HWX_OFF & 4)) MH_HW_FINISH(qf_, qf_ >= zs && qf_ < ze)                              \
            if (!(HWX_OFF & 8)) MH_HW_MMS(S0, S1, S2)                                                 \
            MH_HW_DEAL                                                                                \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        } else {                                                                                      \
            MH_HW_SKIP(S0)                                                                            \
            if (!(HWX_OFF & 4)) MH_HW_FINISH(qf_, qf_ >= zs && qf_ < ze)                              \
        }                                                                                             \
        if (!(HWX_OFF & 16)) if (p_ - 1 >= zs && p_ - 1 < ze) MH_HW_ZOUT(S2, p_ - 1)                  \
        MH_HW_T(2)                                                                                    \
        __syncthreads();                                                                              \
        MH_HW_T(3)                                                                                    \
    }

    // prologue: iteration k = p - (zs - 1) works on buffer k % 3 (the buffers rotate every iteration, existing plane or not, so their LDS offsets are compile-time constants
    // of the three-fold unrolled march): planes p_first, p_first + 1 into buffers k0 % 3, (k0 + 1) % 3 with k0 = p_first - (zs - 1); the loads of plane p_first + 2 in flight
    {
        const int k0 = p_first - (zs - 1);
        MH_HW_LDX MH_HW_ADV
        __syncthreads();                  // the zeroed buffers
        MH_HW_CONV((k0 % 3) * HWG_DB)
        MH_HW_LDX MH_HW_ADV
        MH_HW_CONV(((k0 + 1) % 3) * HWG_DB)
        MH_HW_LDX MH_HW_ADV
        __syncthreads();
    }
    if (jp) __syncthreads();
    // whole blocks of three iterations up to p = ze + 1 (the iterations behind it find nothing to do but keep the barrier count)
    for (int p = zs - 1; p <= ze + 1; p += 3) {
        MH_HW_ITER(p, 0)
        MH_HW_ITER(p + 1, 1)
        MH_HW_ITER(p + 2, 2)
    }
    if (!jp) __syncthreads();
#ifdef HWX_PROF
    if (blockIdx.x == 0 && lane == 0) { _Pragma("unroll") for (int k = 0; k < 8; ++k) reinterpret_cast<long long*>(pmax)[wave * 8 + k] = tacc[k]; }
#endif
#undef MH_HW_T
#undef MH_HW_DEAL
#undef MH_HW_ITER
#undef MH_HW_FINISH
#undef MH_HW_ZOUT
#undef MH_HW_SKIP
#undef MH_HW_MMS
#undef MH_HW_MM
#undef MH_HW_XFORM
#undef MH_HW_U
#undef MH_HW_ADV
#undef MH_HW_CONV
#undef MH_HW_LDX

    if (STATS) {
        // the lane's {count, mean, M2} from its shifted sums: 4 values per plane of the chunk
        Stat run;
        {
            const float a1 = (s1[0] + s1[1]) + (s1[2] + s1[3]), a2 = (s2[0] + s2[1]) + (s2[2] + s2[3]);
            run.n = 4.0f * (float)(ze - zs);
            const float dm = a1 / run.n;
            run.mean = pivot + dm;
            run.m2 = fmaxf(a2 - a1 * dm, 0.0f);
        }
        // a cout's voxels sit in the four lane groups 16 apart of the lower half, the two rows (lane ^ 32) and the four waves that share its cout block
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
            Stat st;
            st.n = 0.0f; st.mean = 0.0f; st.m2 = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                Stat ot;
                ot.n = red[((4 * cb + w) * 16 + c16) * 3]; ot.mean = red[((4 * cb + w) * 16 + c16) * 3 + 1]; ot.m2 = red[((4 * cb + w) * 16 + c16) * 3 + 2];
                st = stat_merge(st, ot);
            }
            float* rec = stats + (((long long)n * Cout + cg * HWG_CN + tid) * nblk + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}

// Weight preparation.  The scale: conv3d_k3_h2_scale_kernel's power of two, a quarter of it (|G g G^T| <= 2.25 max |g|).
__global__ void conv3d_k3_h2w_scale_fix_kernel(float* __restrict__ tail) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { tail[0] *= 4.0f; tail[1] *= 0.25f; }
}
// w [Cout][32][3][3][3] -> [cout group][wave jp 4 + i][operand ((slot 3 + t) 2 + cout block) 2 + piece][lane = kg 16 + cout % 16][8 halves]: the lane's K elements are
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
                const int jpq = j >> 1, slot = (j == 0 || j == 3) ? 0 : 1;      // wave = jp 4 + i holds (slot 0, slot 1) = (j 0, j 1) or (j 3, j 2)
                for (int p = 0; p < 2; ++p) {
                    const long long r = (((long long)(cg * 8 + jpq * 4 + i) * HWG_OPS) + ((slot * 3 + t) * 2 + cb) * 2 + p) * 64 + ln;
                    packed[r * 8 + e] = pc[p];
                }
            }
    }
}

}  // namespace mh
