// Conv3d 3x3x3 / stride 1 / zero padding 1 as an implicit GEMM on the BF16 matrix cores of gfx950 in split-precision
// arithmetic: fp32-equivalent results at a fraction of the fp32 matrix-core cycles (EXPERIMENTAL, opt-in:
// MONAI_AMD_CONV_ALGO=split).
//
// Same reference op and "normalise on load" contract as conv3d_mfma.h (nn.Conv3d of `Convolution`,
// monai/networks/blocks/convolutions.py:98-171, fed by the previous block's deferred InstanceNorm + LeakyReLU).
//
// Arithmetic.  Every fp32 operand x is split into three bf16 pieces x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi),
// lo = bf16(x - hi - mid): 24 significand bits, the split is exact up to the last piece's rounding); the activated input
// is split while it is staged into LDS, the weights once when they are packed.  A product x * w is evaluated as the six
// piece products hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid -- each EXACT in fp32 (8 x 8 significand bits) -- and
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16; the dropped terms are <= 2^-24 relative.  Measured on the oracle
// network (tools/split_precision_numerics.py): max |logit difference| 3.6e-6 and identical argmax, the noise level of two
// fp32 summation orders.  One bf16 MFMA covers 16 input channels in 32 cycles where the fp32 one covers 2 in 64: six of
// them per tap are 6/16 of the fp32 matrix cycles of a direct convolution, 27 * 6/16 = 10.1 fp32-tap equivalents against
// the 12 of the in-plane Winograd kernel -- and without its transforms, which on gfx950 cannot overlap with fp32 MFMAs.
//
// Mapping.  GEMM M = output voxels, N = output channels, K = 16 input channels per instruction, one instruction group per
// tap.  Operand A lane l holds voxel (l & 31) x channels 8 (l >> 5) .. +7 (one 16-byte LDS read of a [voxel][16 channel]
// tile, k-group major so that a row of 8 voxels is 128 contiguous bytes), operand B cout (l & 31) x the same 8 channels; D is the fp32 32x32 layout (lane = one cout, 16 voxels).
// A workgroup = 4 waves = an output tile of 4 (z) x 8 x 8 voxels x 32 couts; wave w owns the z = w slice as two
// M-blocks (y 0-3, y 4-7).  Per 16-channel chunk the halo tile 6 x 10 x 10 x 16 channels (three pieces, 57.6 KB) and the
// weight slab [piece][27 taps][32 couts][16 channels] (82.9 KB) are staged in LDS; the matrix loop is 27 taps x
// (9 operand reads -> 12 MFMAs), operands of tap t+1 requested before the MFMAs of tap t issue.
#pragma once
#include "common.h"

namespace mh {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int SP_TZ = 4, SP_TY = 8, SP_TX = 8;                       // output tile of a workgroup
constexpr int SP_HY = SP_TY + 2, SP_HX = SP_TX + 2;
constexpr int SP_HV = (SP_TZ + 2) * SP_HY * SP_HX;                   // 600 halo voxels
constexpr int SP_CC = 16, SP_CN = 32, SP_NP = 3;                     // channels per chunk (= MFMA K), couts per workgroup, pieces
constexpr int SP_XV = SP_HV * 2;                                     // uint4 per piece of the input tile: [k-group][voxel]
constexpr int SP_WV = 27 * SP_CN * 2;                                // uint4 per piece of the weight slab: [tap][k-group][cout]
constexpr int SP_NRM_MAX = 512;

struct SplitPieces { bf16x8 p[SP_NP]; };

__device__ __forceinline__ void sp_split(float f, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)f;
    const float r1 = f - (float)hi;
    mid = (__bf16)r1;
    lo = (__bf16)(r1 - (float)mid);
}

// T = z-tiles per workgroup: the chunk's weight slab is staged once for T tiles (a single-tile workgroup spends more time
// on its weight copy, its exposed first loads and its epilogue than in the matrix loop: measured 11.6 of 20.1 ms), each tile
// keeps its 2 x 16 accumulators while the chunks stream by, and the input of the next (tile, chunk) is in flight during the
// matrix loop of the current one.
template <bool STATS, bool NRM, int T>
__global__ void __launch_bounds__(256, 1)
conv3d_k3_split_kernel(Tensor in, const uint4* __restrict__ wp, const float* __restrict__ bias, Tensor out, float* __restrict__ stats,
                       int bxn, int byn, unsigned nblk) {
    __shared__ uint4 xs[SP_NP * SP_XV];
    __shared__ uint4 ws[SP_NP * SP_WV];
    __shared__ float4 nrm_s[NRM ? SP_NRM_MAX : 1];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = in.C, Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long HW = (long long)H * W, DHW = (long long)D * HW;
    const int nchunk = Cin / SP_CC;
    const unsigned ncg = (unsigned)(Cout / SP_CN);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(lid % ncg);
    lid /= ncg;
    const unsigned b = lid % nblk;
    const int n = (int)(lid / nblk);
    const int x0 = (int)(b % bxn) * SP_TX, y0 = (int)((b / bxn) % byn) * SP_TY, z0 = (int)(b / (bxn * byn)) * (SP_TZ * T);
    const float* src = in.data + (long long)n * in.n_stride;

    if (NRM) {
        for (int c = tid; c < Cin; c += 256) nrm_s[c] = *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * c);
    }

    // staging tasks of this thread: (halo voxel v, half h of the chunk's channels), t = tid + 256 r; the (y, x) part of the
    // position is fixed for the whole workgroup, the z part moves with the tile
    constexpr int NTASK = (2 * SP_HV + 255) / 256;      // 5
    long long oxy[NTASK];
    int lidx[NTASK], hzr[NTASK];
    unsigned xym = 0u;          // bit r: (y, x) inside the volume; bit 8 + r: the task exists
#pragma unroll
    for (int r = 0; r < NTASK; ++r) {
        const int t = tid + 256 * r;
        const int h = t >= SP_HV ? 1 : 0, v = t - h * SP_HV;
        const int hz = v / (SP_HY * SP_HX), hy = (v / SP_HX) % SP_HY, hx = v % SP_HX;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool task = t < 2 * SP_HV;
        const bool ok = task && gy >= 0 && gy < H && gx >= 0 && gx < W;
        xym |= (unsigned)ok << r;
        xym |= (unsigned)task << (8 + r);
        oxy[r] = ok ? (long long)gy * W + gx + (long long)(8 * h) * DHW : 0;
        hzr[r] = task ? hz : 1;
        lidx[r] = task ? h * SP_HV + v : 0;        // [k-group][voxel]: the 8 lanes of a row read 128 contiguous bytes (no bank conflicts)
    }

    // operand addresses (uint4 units): A = voxel (lane & 31) of M-block mb at tap (0,0,0), B = cout (lane & 31)
    const int li = lane & 31, kg = lane >> 5;
    const int abase0 = kg * SP_HV + (wave * SP_HY + (li >> 3)) * SP_HX + (li & 7);        // M-block 0 (y 0-3)
    const int abase1 = abase0 + 4 * SP_HX;                                                // M-block 1 (y 4-7)
    const int bbase = kg * SP_CN + li;

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[t][0][i] = 0.0f; acc[t][1][i] = 0.0f; }

    const uint4* wg = wp + (long long)cg * nchunk * (SP_NP * SP_WV);
    float pre[NTASK][8];
    unsigned okn = 0u, okc = 0u;       // validity bits of the tasks in flight (next) / being converted (current)
#define MH_SP_ISSUE_INPUT(CH, TI)                                                                           \
    {                                                                                                       \
        const float* sc_ = src + (long long)((CH) * SP_CC) * DHW;                                           \
        okn = 0u;                                                                                           \
        _Pragma("unroll") for (int r = 0; r < NTASK; ++r) {                                                 \
            const int gz_ = z0 + SP_TZ * (TI) + hzr[r] - 1;                                                 \
            const bool ok_ = ((xym >> r) & 1u) && gz_ >= 0 && gz_ < D;                                      \
            okn |= (unsigned)ok_ << r;                                                                      \
            const long long o_ = ok_ ? (long long)gz_ * HW + oxy[r] : 0;                                    \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) pre[r][j] = sc_[o_ + (long long)j * DHW];         \
        }                                                                                                   \
    }
#define MH_SP_CONVERT(CH)                                                                                   \
    _Pragma("unroll") for (int r = 0; r < NTASK; ++r) {                                                     \
        const bool ok = (okc >> r) & 1u;                                                                    \
        const int h = (tid + 256 * r) >= SP_HV ? 1 : 0;                                                     \
        bf16x8 ph, pm, pl;                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                     \
            float v = pre[r][j];                                                                            \
            if (NRM) {                                                                                      \
                const float4 a = nrm_s[(CH) * SP_CC + 8 * h + j];                                           \
                v = act(v, a.x, a.y, a.z);                                                                  \
            }                                                                                               \
            v = ok ? v : 0.0f;                                                                              \
            __bf16 hi, mid, lo;                                                                             \
            sp_split(v, hi, mid, lo);                                                                       \
            ph[j] = hi; pm[j] = mid; pl[j] = lo;                                                            \
        }                                                                                                   \
        if ((xym >> (8 + r)) & 1u) {                                                                        \
            xs[lidx[r]] = __builtin_bit_cast(uint4, ph);                                                    \
            xs[SP_XV + lidx[r]] = __builtin_bit_cast(uint4, pm);                                            \
            xs[2 * SP_XV + lidx[r]] = __builtin_bit_cast(uint4, pl);                                        \
        }                                                                                                   \
    }
    uint4 a0[2][SP_NP], a1[2][SP_NP], bb[2][SP_NP];
#define MH_SP_FETCH(BUF, TAP)                                                                               \
    {                                                                                                       \
        constexpr int kz_ = (TAP) / 9, ky_ = ((TAP) / 3) % 3, kx_ = (TAP) % 3;                              \
        constexpr int aoff_ = (kz_ * SP_HY + ky_) * SP_HX + kx_;                                            \
        _Pragma("unroll") for (int p = 0; p < SP_NP; ++p) {                                                 \
            a0[BUF][p] = xs[p * SP_XV + abase0 + aoff_];                                                    \
            a1[BUF][p] = xs[p * SP_XV + abase1 + aoff_];                                                    \
            bb[BUF][p] = ws[p * SP_WV + (TAP) * (SP_CN * 2) + bbase];                                       \
        }                                                                                                   \
    }
#define MH_SP_MFMA(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), ACC, 0, 0, 0);
#ifndef SP_PRODUCTS
#define SP_PRODUCTS 6          /* development knob: 3 = hi*mid, mid*hi, hi*hi only (4e-5 logit error on the oracle network) */
#endif
#if SP_PRODUCTS == 6
#define MH_SP_TAP(BUF, TI)                                                                                  \
    {   /* in the order the pieces arrive from LDS (hi, mid, lo): hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi */ \
        MH_SP_MFMA(acc[TI][0], a0[BUF][0], bb[BUF][0]) MH_SP_MFMA(acc[TI][1], a1[BUF][0], bb[BUF][0])       \
        MH_SP_MFMA(acc[TI][0], a0[BUF][0], bb[BUF][1]) MH_SP_MFMA(acc[TI][1], a1[BUF][0], bb[BUF][1])       \
        MH_SP_MFMA(acc[TI][0], a0[BUF][1], bb[BUF][0]) MH_SP_MFMA(acc[TI][1], a1[BUF][1], bb[BUF][0])       \
        MH_SP_MFMA(acc[TI][0], a0[BUF][1], bb[BUF][1]) MH_SP_MFMA(acc[TI][1], a1[BUF][1], bb[BUF][1])       \
        MH_SP_MFMA(acc[TI][0], a0[BUF][0], bb[BUF][2]) MH_SP_MFMA(acc[TI][1], a1[BUF][0], bb[BUF][2])       \
        MH_SP_MFMA(acc[TI][0], a0[BUF][2], bb[BUF][0]) MH_SP_MFMA(acc[TI][1], a1[BUF][2], bb[BUF][0])       \
    }
#else
#define MH_SP_TAP(BUF, TI)                                                                                  \
    {                                                                                                       \
        MH_SP_MFMA(acc[TI][0], a0[BUF][0], bb[BUF][1]) MH_SP_MFMA(acc[TI][1], a1[BUF][0], bb[BUF][1])       \
        MH_SP_MFMA(acc[TI][0], a0[BUF][1], bb[BUF][0]) MH_SP_MFMA(acc[TI][1], a1[BUF][1], bb[BUF][0])       \
        MH_SP_MFMA(acc[TI][0], a0[BUF][0], bb[BUF][0]) MH_SP_MFMA(acc[TI][1], a1[BUF][0], bb[BUF][0])       \
    }
#endif
// The operand reads of tap t+1 are INTERLEAVED with the MFMAs of tap t (one read behind each MFMA): issued as a burst in
// front of them, the 9 reads of the four (barrier-synchronised) waves queue up on the LDS for about as long as the 12 MFMAs
// take, and an in-order wave cannot start its MFMAs before its last read has been accepted (measured: 820 cycles per tap).
#define MH_SP_STEP(TAP, TI)                                                                                 \
    {                                                                                                       \
        if ((TAP) + 1 < 27) MH_SP_FETCH(((TAP) + 1) & 1, ((TAP) + 1 < 27 ? (TAP) + 1 : 0))                  \
        MH_SP_TAP((TAP) & 1, TI)                                                                            \
        if ((TAP) + 1 < 27) {                                                                               \
            _Pragma("unroll") for (int i_ = 0; i_ < 3 * SP_NP; ++i_) {                                      \
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      /* one MFMA */                      \
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      /* one LDS read */                  \
            }                                                                                               \
        }                                                                                                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);             /* the remaining MFMAs */           \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    }
    // development-only timing knobs (results become wrong): where do the cycles outside the matrix loop go?
#ifdef SP_DIAG_NOLOAD
#define MH_SP_DIAG_LOAD_GUARD if (false)
#else
#define MH_SP_DIAG_LOAD_GUARD
#endif
#ifdef SP_DIAG_NOCONVERT
#define MH_SP_DIAG_CONVERT_GUARD(F_) if (ch == 0 && (F_))
#else
#define MH_SP_DIAG_CONVERT_GUARD(F_)
#endif
    // one (tile, chunk): the registers in flight become the LDS tile, the next (tile, chunk) is requested, 27 taps
#define MH_SP_TILE(TI, FIRST)                                                                               \
    if ((TI) < T) {                                                                                         \
        if (!(FIRST)) __syncthreads();       /* the previous tile's matrix loop is done with xs */          \
        okc = okn;                                                                                          \
        MH_SP_DIAG_CONVERT_GUARD(FIRST)                                                                     \
        {                                                                                                   \
            MH_SP_CONVERT(ch)                                                                               \
        }                                                                                                   \
        if (FIRST) {                                                                                        \
            _Pragma("unroll") for (int i = 0; i < NWV; ++i)                                                 \
                if (tid + 256 * i < SP_NP * SP_WV) ws[tid + 256 * i] = wr[i];                               \
        }                                                                                                   \
        __syncthreads();                                                                                    \
        MH_SP_DIAG_LOAD_GUARD                                                                               \
        {                                                                                                   \
            if ((TI) + 1 < T) MH_SP_ISSUE_INPUT(ch, (TI) + 1)                                               \
            else if (ch + 1 < nchunk) MH_SP_ISSUE_INPUT(ch + 1, 0)                                          \
        }                                                                                                   \
        MH_SP_FETCH(0, 0)                                                                                   \
        MH_SP_STEP(0, TI) MH_SP_STEP(1, TI) MH_SP_STEP(2, TI) MH_SP_STEP(3, TI) MH_SP_STEP(4, TI) MH_SP_STEP(5, TI)         \
        MH_SP_STEP(6, TI) MH_SP_STEP(7, TI) MH_SP_STEP(8, TI) MH_SP_STEP(9, TI) MH_SP_STEP(10, TI) MH_SP_STEP(11, TI)       \
        MH_SP_STEP(12, TI) MH_SP_STEP(13, TI) MH_SP_STEP(14, TI) MH_SP_STEP(15, TI) MH_SP_STEP(16, TI) MH_SP_STEP(17, TI)   \
        MH_SP_STEP(18, TI) MH_SP_STEP(19, TI) MH_SP_STEP(20, TI) MH_SP_STEP(21, TI) MH_SP_STEP(22, TI) MH_SP_STEP(23, TI)   \
        MH_SP_STEP(24, TI) MH_SP_STEP(25, TI) MH_SP_STEP(26, TI)                                            \
    }
    constexpr int NWV = (SP_NP * SP_WV + 255) / 256;      // 21 uint4 of the weight slab per thread
    MH_SP_ISSUE_INPUT(0, 0)
    for (int ch = 0; ch < nchunk; ++ch) {
        __syncthreads();        // the previous chunk's last matrix loop is done with xs / ws (and nrm_s is visible)
        const uint4* wc = wg + (long long)ch * (SP_NP * SP_WV);
        uint4 wr[NWV];          // requested now, written to LDS after the first tile's input has been converted
#pragma unroll
        for (int i = 0; i < NWV; ++i) wr[i] = wc[min(tid + 256 * i, SP_NP * SP_WV - 1)];
        MH_SP_TILE(0, true)
        MH_SP_TILE(1, false)
        MH_SP_TILE(2, false)
        MH_SP_TILE(3, false)
    }
#undef MH_SP_TILE
#undef MH_SP_DIAG_LOAD_GUARD
#undef MH_SP_DIAG_CONVERT_GUARD
#undef MH_SP_STEP
#undef MH_SP_TAP
#undef MH_SP_MFMA
#undef MH_SP_FETCH
#undef MH_SP_CONVERT
#undef MH_SP_ISSUE_INPUT

    // ---- epilogue: lane = cout (lane & 31); accumulator r = voxel row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the M-block:
    //      x = (r & 3) + 4 (lane >> 5), y = 4 mb + (r >> 2): four x-contiguous values per (mb, r >> 2) -> one 16-byte store
    const int co = cg * SP_CN + li;
    const float bco = bias ? bias[co] : 0.0f;
    float* const obase = out.data + (long long)n * out.n_stride + (long long)co * DHW + (long long)(z0 + wave) * HW + (long long)y0 * W + x0 + 4 * kg;
    Stat run;
    run.n = 0.0f; run.mean = 0.0f; run.m2 = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (t >= T) break;
        f32x16 c0 = acc[t][0], c1 = acc[t][1];
#pragma unroll
        for (int i = 0; i < 16; ++i) { c0[i] += bco; c1[i] += bco; }
        float* ob = obase + (long long)(SP_TZ * t) * HW;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<f32x4*>(ob + (long long)q * W) = f32x4{c0[4 * q], c0[4 * q + 1], c0[4 * q + 2], c0[4 * q + 3]};
            *reinterpret_cast<f32x4*>(ob + (long long)(4 + q) * W) = f32x4{c1[4 * q], c1[4 * q + 1], c1[4 * q + 2], c1[4 * q + 3]};
        }
        if (STATS) {
            float sum = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += c0[i] + c1[i];
            Stat loc;
            loc.n = 32.0f;
            loc.mean = sum * (1.0f / 32.0f);
            float m2 = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float d0 = c0[i] - loc.mean, d1 = c1[i] - loc.mean;
                m2 += d0 * d0 + d1 * d1;
            }
            loc.m2 = m2;
            run = stat_merge(run, loc);
        }
    }
    if (STATS) {
        {   // the two k-groups of a cout (lanes l and l + 32) hold disjoint voxels
            Stat ot;
            ot.n = __shfl_xor(run.n, 32); ot.mean = __shfl_xor(run.mean, 32); ot.m2 = __shfl_xor(run.m2, 32);
            run = stat_merge(run, ot);
        }
        __syncthreads();            // xs is free
        float* red = reinterpret_cast<float*>(xs);
        if (kg == 0) { red[(wave * SP_CN + li) * 3] = run.n; red[(wave * SP_CN + li) * 3 + 1] = run.mean; red[(wave * SP_CN + li) * 3 + 2] = run.m2; }
        __syncthreads();
        if (tid < SP_CN) {
            Stat st;
            st.n = red[tid * 3]; st.mean = red[tid * 3 + 1]; st.m2 = red[tid * 3 + 2];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                Stat ot;
                ot.n = red[(w * SP_CN + tid) * 3]; ot.mean = red[(w * SP_CN + tid) * 3 + 1]; ot.m2 = red[(w * SP_CN + tid) * 3 + 2];
                st = stat_merge(st, ot);
            }
            float* rec = stats + (((long long)n * Cout + cg * SP_CN + tid) * nblk + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}

// Weight pieces in the order operand B is read: packed[cout group][chunk][piece][tap][k-group][cout & 31][cin & 7] (bf16).
// One thread per (cout, cin).
__global__ void __launch_bounds__(256)
conv3d_k3_split_pack_kernel(const float* __restrict__ w, int Cin, int Cout, __bf16* __restrict__ packed) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const int nchunk = Cin / SP_CC;
    const long long slab = ((long long)(co / SP_CN) * nchunk + ci / SP_CC) * SP_NP;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        __bf16 pc[SP_NP];
        sp_split(w[((long long)co * Cin + ci) * 27 + tap], pc[0], pc[1], pc[2]);
#pragma unroll
        for (int p = 0; p < SP_NP; ++p)
            packed[((((slab + p) * 27 + tap) * 2 + (ci % SP_CC) / 8) * SP_CN + (co % SP_CN)) * 8 + (ci % 8)] = pc[p];
    }
}

}  // namespace mh
