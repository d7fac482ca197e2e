// Conv3d 3x3x3 / stride 1 / zero padding 1: Winograd F(2x2, 3x3) in the (y, x) plane, direct three taps along z,
// streamed along z on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32).
//
// Same reference op and "normalise on load" contract as conv3d_mfma.h (nn.Conv3d of `Convolution`,
// monai/networks/blocks/convolutions.py:98-171, fed by the previous block's deferred InstanceNorm + LeakyReLU).
// Output plane z = sum over kz of the 2-D convolution of input plane z + kz - 1 with the 3x3 slice g[kz]; each 2-D
// convolution in minimal-filtering form:  Y = A^T [ sum_kz sum_cin (G g_kz G^T) .* (B^T d B) ] A,  16 products per 2x2
// outputs instead of 36 -> 12 multiply-adds per (output voxel, cin, cout) instead of 27 (2.25x fewer matrix-core
// cycles than conv3d_mfma.h).  The full 3-D form (conv3d_winograd.h: 8 per voxel) needs 256 accumulation registers and
// ~7 other instructions per MFMA, which one wave per SIMD cannot issue in an MFMA's shadow; this form needs ~2.7.
//
// Mapping.  A tile = 2x2 outputs of one plane (4x4 input patch).  GEMM per transform position xi (16 of them) and z-tap:
//   M_xi[tile][cout] += V_xi(plane p)[tile][cin] * U_xi^kz[cin][cout],  M = 16 tiles (a 4x4 arrangement: 8x8 outputs),
//   N = 16 couts, K = 4 input channels.  A wave owns an 8x8 (y, x) output column of 16 couts and marches along z; lane l
//   transforms the patch of tile l & 15 for input channel l >> 4 (8 LDS reads + 32 adds) once per (plane, 4 channels) and
//   feeds 48 MFMAs with it (3 z-taps x 16 positions): input plane p contributes to output planes p+1, p, p-1, held in
//   three rotating sets of 16 accumulators (192 registers); when plane p is done for all input channels, output plane
//   p-1 is complete: inverse transform in registers, bias, statistics, store (each lane ends up with two rows of 8
//   contiguous x of one cout), clear the set.
// A workgroup = 2x2 waves = a 16x16 (y, x) region x 16 couts x one z-chunk.  Per 4-channel step the region's input
// plane [4][18][18] (norm + activation applied, pitch 20: conflict-free patch reads) and the step's weight slab
// [3][16][4][16] go through a ring of three LDS buffers, one barrier per step.  One wave per SIMD hides nothing by
// multithreading, so the step is software-pipelined by hand: while step g's 48 MFMAs issue, the wave commits step g+1's
// staged registers to LDS (first 16 MFMAs), then -- after the barrier -- issues the global loads of step g+3 and reads
// and transforms the patch of step g+1 (last 32 MFMAs).  Loads stay in flight for ~1.3 steps in two alternating
// register sets.
#pragma once
#include "common.h"

namespace mh {

constexpr int W2_B = 16;                                   // region edge (y and x) of a workgroup
constexpr int W2_R = 18, W2_PX = 20;                       // input region edge, LDS row pitch (= 4 mod 16)
constexpr int W2_CS = W2_R * W2_PX;                        // LDS channel stride (360)
constexpr int W2_KC = 4;                                   // input channels per step (MFMA K)
constexpr int W2_XBUF = W2_KC * W2_CS;                     // staged planes of one step (floats)
constexpr int W2_UPITCH = 52;                              // B operands of one lane: 48 floats [kz][xi] + 4 pad (conflict-free 16-byte reads)
constexpr int W2_UBUF = 4096;                              // weight slab of one step: [lane = (cin & 3) * 16 + (cout & 15)][52], zero padded
constexpr int W2_SLOTS = (W2_R * W2_R + 63) / 64;          // 6 region elements per lane (wave w stages channel w)
constexpr int W2_CN = 16;                                  // couts per workgroup
constexpr int W2_NBUF = 3;                                 // LDS ring
constexpr int W2_ZSLAB = 64;                                // zero weights: z-taps that leave the chunk multiply by these
constexpr int W2_SMEM = W2_NBUF * (W2_XBUF + W2_UBUF) + W2_ZSLAB + 64;   // + a dump row for the unused staging slots

#define MH_W2_BT(o0, o1, o2, o3, d0, d1, d2, d3) \
    { o0 = (d0) - (d2); o1 = (d1) + (d2); o2 = (d2) - (d1); o3 = (d1) - (d3); }

template <bool STATS, bool NRM>     // NRM: the input carries a deferred norm + activation record
__global__ void __launch_bounds__(256, 1)
conv3d_k3_wino2d_kernel(Tensor in, const float* __restrict__ up, const float* __restrict__ bias, Tensor out,
                        float* __restrict__ stats, int bxn, int byn, int zchunk, unsigned nblk) {
    __shared__ __attribute__((aligned(16))) float smem[W2_SMEM];
    float* const xs = smem;
    float* const us = smem + W2_NBUF * W2_XBUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t16 = lane & 15, kq = lane >> 4;
    const int Cin = in.C, Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long HW = (long long)H * W, DHW = (long long)D * HW;
    const int KS = Cin / W2_KC;                               // even (the launcher requires Cin % 8 == 0)

    // 1-D launch over (window, region, cout group), cout group fastest, XCD-aware: each XCD gets a contiguous run of this
    // order, so the cout groups of a region (same input) and neighbouring regions (shared halo) run on the same L2 --
    // with a 3-D grid whose x extent is not a multiple of 8 the two cout groups land on different XCDs and the input is
    // fetched from HBM twice (measured: 2.2x the algorithmic input bytes).
    const unsigned ncg = (unsigned)(Cout / W2_CN);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(lid % ncg);
    lid /= ncg;
    const unsigned b = lid % nblk;
    const int n = (int)(lid / nblk);
    const int x0 = (int)(b % bxn) * W2_B, y0 = (int)((b / bxn) % byn) * W2_B;
    const int zs = (int)(b / (bxn * byn)) * zchunk, ze = min(zs + zchunk, D);
    const int p_last = min(ze, D - 1);                        // last real input plane of the chunk (first: max(zs - 1, 0))

    // staging: wave w stages channel (4 s + w) of the current plane; lane elements e = lane + 64 j of the 18 x 18 region
    int soff[W2_SLOTS], loff[W2_SLOTS];
    unsigned sokm = 0u;
#pragma unroll
    for (int j = 0; j < W2_SLOTS; ++j) {
        const int e = lane + 64 * j;
        const int ly = e / W2_R, lx = e - ly * W2_R;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        const bool inreg = e < W2_R * W2_R;
        const bool ok = inreg && gy >= 0 && gy < H && gx >= 0 && gx < W;
        sokm |= (unsigned)ok << j;
        soff[j] = ok ? gy * W + gx : 0;
        loff[j] = inreg ? wave * W2_CS + ly * W2_PX + lx : -1;      // only the last slot can fall outside the region
    }
    const float* src = in.data + (long long)n * in.n_stride + (long long)wave * DHW;
    const f32x4* ug = reinterpret_cast<const f32x4*>(up + (long long)cg * KS * W2_UBUF) + tid;

    // cursors of the load stream, kept as running pointers (a few scalar adds per step instead of 64-bit multiplies):
    // xptr / uptr = region plane and weight slab of the next ISSUE, nptr = norm record of the next COMMIT
    int ip = max(zs - 1, 0), is = 0, cs = 0;
    const float* xptr = src + (long long)ip * HW;
    const f32x4* uptr = ug;
    const long long xstep = (long long)W2_KC * DHW, xwrap = (long long)KS * W2_KC * DHW;
    const float* nptr = NRM ? in.nrm + (long long)n * in.nrm_n_stride + 4LL * wave : nullptr;
    float xin[2][W2_SLOTS];
    f32x4 uin[2][4];
#define MH_W2_ISSUE(SET)                                                                              \
    {                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < W2_SLOTS; ++j) xin[SET][j] = xptr[soff[j]];             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) uin[SET][j] = uptr[256 * j];                    \
        xptr += xstep; uptr += W2_UBUF / 4;                                                           \
        if (++is == KS) {                                                                             \
            is = 0; uptr = ug; xptr -= xwrap;                                                         \
            if (ip < p_last) { ++ip; xptr += HW; }                                                    \
        }                                                                                             \
    }
    // branch-free (it is scheduled into the MFMA shadow): a slot outside the region writes to the dump row
#define MH_W2_COMMIT(SET, BUF)                                                                        \
    {                                                                                                 \
        float4 a_ = make_float4(1.0f, 0.0f, 1.0f, 0.0f);                                              \
        if (NRM) a_ = *reinterpret_cast<const float4*>(nptr + 16 * cs);                               \
        float* xb_ = xs + (BUF) * W2_XBUF;                                                            \
        _Pragma("unroll") for (int j = 0; j < W2_SLOTS; ++j) {                                        \
            const float val_ = ((sokm >> j) & 1u) ? act(xin[SET][j], a_.x, a_.y, a_.z) : 0.0f;        \
            if (64 * j + 63 < W2_R * W2_R) xb_[loff[j]] = val_;                                       \
            else *(loff[j] >= 0 ? xb_ + loff[j] : dump) = val_;                                       \
        }                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
            reinterpret_cast<f32x4*>(us + (BUF) * W2_UBUF)[tid + 256 * j] = uin[SET][j];              \
        if (++cs == KS) cs = 0;                                                                       \
    }

    // this lane's patch: tile (ty, tx) of the wave's 4 x 4, input channel kq of the step
    const int wby = wave >> 1, wbx = wave & 1;
    const int pbase = kq * W2_CS + (8 * wby + 2 * (t16 >> 2)) * W2_PX + 8 * wbx + 2 * (t16 & 3);

    f32x4 acc[3][16];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // epilogue constants: lane (kq, t16) ends up with rows 2 kq, 2 kq + 1 x 8 columns of cout t16 of the wave's 8 x 8 block
    const int co = cg * W2_CN + t16;
    const float bco = bias ? bias[co] : 0.0f;
    const int gy0 = y0 + 8 * wby + 2 * kq, gx0 = x0 + 8 * wbx;
    const bool rok0 = gy0 < H && gx0 < W, rok1 = gy0 + 1 < H && gx0 < W;      // W % 8 == 0: the 8 columns are in or out together
    float* const obase = out.data + (long long)n * out.n_stride + (long long)co * DHW + (long long)gy0 * W + gx0;
    Stat run;
    run.n = 0.0f; run.mean = 0.0f; run.m2 = 0.0f;

    // this lane's transformed patch, two alternating copies: V[PAR] feeds step g while V[1 - PAR] is built for step g + 1
    float vv[2][16];
    float raw[4][4];
#define MH_W2_READ_PATCH(BUF)                                                                         \
    {                                                                                                 \
        const float* xp_ = xs + (BUF) * W2_XBUF + pbase;                                              \
        _Pragma("unroll") for (int y = 0; y < 4; ++y) {                                               \
            const f32x2 lo_ = *reinterpret_cast<const f32x2*>(xp_ + y * W2_PX);                       \
            const f32x2 hi_ = *reinterpret_cast<const f32x2*>(xp_ + y * W2_PX + 2);                   \
            raw[y][0] = lo_[0]; raw[y][1] = lo_[1]; raw[y][2] = hi_[0]; raw[y][3] = hi_[1];           \
        }                                                                                             \
    }
#define MH_W2_TRANSFORM(DST)                                                                          \
    {                                                                                                 \
        float tv[4][4];                                                                               \
        _Pragma("unroll") for (int x = 0; x < 4; ++x)                                                 \
            MH_W2_BT(tv[0][x], tv[1][x], tv[2][x], tv[3][x], raw[0][x], raw[1][x], raw[2][x], raw[3][x]) \
        _Pragma("unroll") for (int y = 0; y < 4; ++y)                                                 \
            MH_W2_BT(vv[DST][y * 4 + 0], vv[DST][y * 4 + 1], vv[DST][y * 4 + 2], vv[DST][y * 4 + 3], tv[y][0], tv[y][1], tv[y][2], tv[y][3]) \
    }
    // B operands live in two alternating register sets: while 16 MFMAs consume one, the other is fetched from LDS for
    // the next 16 (an LDS read issued next to its MFMA would expose the full LDS latency 48 times per step)
    float ubr[2][16];
#define MH_W2_UBLOAD(UR, UB)                                                                          \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                   \
        const f32x4 t4_ = reinterpret_cast<const f32x4*>(UB)[q];                                      \
        ubr[UR][4 * q] = t4_[0]; ubr[UR][4 * q + 1] = t4_[1]; ubr[UR][4 * q + 2] = t4_[2]; ubr[UR][4 * q + 3] = t4_[3]; \
    }
#define MH_W2_MFMA16(SET, UR, PAR)                                                                    \
    _Pragma("unroll") for (int i = 0; i < 16; ++i)                                                    \
        acc[SET][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[PAR][i], ubr[UR][i], acc[SET][i], 0, 0, 0);

    // one 4-channel step of input plane p: SP1 / S0 / SM1 = accumulator sets of output planes p+1 / p / p-1.  No
    // branches (each phase is one scheduling region): a z-tap that leaves the chunk reads the zero slab instead.
#define MH_W2_STEP(PAR, SP1, S0, SM1)                                                                 \
    {                                                                                                 \
        const int bnext_ = bcur + 1 == W2_NBUF ? 0 : bcur + 1;                                        \
        const float* ub_ = us + bcur * W2_UBUF + lane * W2_UPITCH;                                    \
        const float* un_ = us + bnext_ * W2_UBUF + lane * W2_UPITCH;                                  \
        /* On gfx950 an fp32 MFMA does not overlap with the wave's own VALU / LDS instructions, and every switch   */ \
        /* between the two costs ~17 cycles (tools/ubench/issue.hip): the three z-taps issue as bursts of 16       */ \
        /* back-to-back MFMAs, the other work sits in lumps between them (operands are fetched one burst ahead).   */ \
        MH_W2_UBLOAD(1 - (PAR), k1ok ? ub_ + 16 : zslab)                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        MH_W2_MFMA16(SP1, PAR, PAR)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        MH_W2_COMMIT(1 - (PAR), bnext_)                   /* next step's staged registers -> LDS */   \
        __syncthreads();                                                                              \
        MH_W2_UBLOAD(PAR, k2ok ? ub_ + 32 : zslab)                                                    \
        MH_W2_ISSUE(1 - (PAR))                            /* global loads of step g + 3 */            \
        MH_W2_READ_PATCH(bnext_)                          /* patch of step g + 1 */                   \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        MH_W2_MFMA16(S0, 1 - (PAR), PAR)                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        MH_W2_UBLOAD(1 - (PAR), k0next ? un_ : zslab)                                                 \
        MH_W2_TRANSFORM(1 - (PAR))                        /* transform of step g + 1 */               \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        MH_W2_MFMA16(SM1, PAR, PAR)                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        bcur = bnext_;                                                                                \
    }

    // output plane Z is complete in accumulator set S: inverse transform, bias, statistics, store, clear
#define MH_W2_EMIT(S, Z)                                                                              \
    {                                                                                                 \
        f32x4 pr[4][2];                                                                               \
        _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                               \
            const f32x4 m0 = acc[S][a * 4 + 0], m1 = acc[S][a * 4 + 1], m2 = acc[S][a * 4 + 2], m3 = acc[S][a * 4 + 3]; \
            pr[a][0] = (m0 + m1) + m2;                                                                \
            pr[a][1] = (m1 - m2) - m3;                                                                \
        }                                                                                             \
        f32x4 o[2][2];                                                                                \
        _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                               \
            o[0][e] = ((pr[0][e] + pr[1][e]) + pr[2][e]) + bco;                                       \
            o[1][e] = ((pr[1][e] - pr[2][e]) - pr[3][e]) + bco;                                       \
        }                                                                                             \
        float* op_ = obase + (long long)(Z) * HW;                                                     \
        _Pragma("unroll") for (int f = 0; f < 2; ++f) {                                               \
            if (f == 0 ? rok0 : rok1) {                                                               \
                *reinterpret_cast<f32x4*>(op_ + f * W) = f32x4{o[f][0][0], o[f][1][0], o[f][0][1], o[f][1][1]};     \
                *reinterpret_cast<f32x4*>(op_ + f * W + 4) = f32x4{o[f][0][2], o[f][1][2], o[f][0][3], o[f][1][3]}; \
            }                                                                                         \
        }                                                                                             \
        if (STATS) {      /* short dependency chains: quad-wise partial sums, rows masked by 0 / 1 */ \
            Stat loc_;                                                                                \
            const float w0_ = rok0 ? 1.0f : 0.0f, w1_ = rok1 ? 1.0f : 0.0f;                           \
            loc_.n = 8.0f * (w0_ + w1_);                                                              \
            const f32x4 s4_ = (o[0][0] + o[0][1]) * w0_ + (o[1][0] + o[1][1]) * w1_;                  \
            const float sum_ = (s4_[0] + s4_[1]) + (s4_[2] + s4_[3]);                                 \
            loc_.mean = loc_.n > 0.0f ? sum_ / loc_.n : 0.0f;                                         \
            const f32x4 d00_ = o[0][0] - loc_.mean, d01_ = o[0][1] - loc_.mean, d10_ = o[1][0] - loc_.mean, d11_ = o[1][1] - loc_.mean; \
            const f32x4 q4_ = (d00_ * d00_ + d01_ * d01_) * w0_ + (d10_ * d10_ + d11_ * d11_) * w1_;  \
            loc_.m2 = (q4_[0] + q4_[1]) + (q4_[2] + q4_[3]);                                          \
            run = stat_merge(run, loc_);                                                              \
        }                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[S][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};     \
    }

#define MH_W2_PLANE(P, SP1, S0, SM1)                                                                  \
    if ((P) <= ze) {                                                                                  \
        const int p_ = (P);                                                                           \
        if (p_ >= 0 && p_ <= p_last) {                                                                \
            const bool k0ok = p_ + 1 < ze, k1ok = p_ >= zs && p_ < ze, k2ok = p_ - 1 >= zs;           \
            const bool k0nextplane = p_ + 2 < ze;                                                     \
            (void)k0ok;                                                                               \
            for (int s = 0; s < KS; s += 2) {                                                         \
                { const bool k0next = k0ok; MH_W2_STEP(0, SP1, S0, SM1) }                             \
                { const bool k0next = s + 2 < KS ? k0ok : k0nextplane; MH_W2_STEP(1, SP1, S0, SM1) }  \
            }                                                                                         \
        }                                                                                             \
        if (p_ - 1 >= zs) MH_W2_EMIT(SM1, p_ - 1)                                                     \
    }

    // prologue: the first two steps' loads, commit the first, transform its patch, issue the third step's loads
    int bcur = 0;
    float* const zslab_w = smem + W2_NBUF * (W2_XBUF + W2_UBUF);
    if (tid < W2_ZSLAB) zslab_w[tid] = 0.0f;
    const float* const zslab = zslab_w;             // every lane reads the same 16 zeros
    float* const dump = zslab_w + W2_ZSLAB + lane;
    MH_W2_ISSUE(0)
    MH_W2_ISSUE(1)
    MH_W2_COMMIT(0, 0)
    __syncthreads();
    MH_W2_READ_PATCH(0)
    MH_W2_TRANSFORM(0)
    MH_W2_ISSUE(0)
    {   // B operands of the first step's z-tap 0 (the first real plane is zs - 1 or 0: its z-tap 0 feeds plane zs or 1)
        const int pf = max(zs - 1, 0);
        const float* u0 = pf + 1 < ze ? us + lane * W2_UPITCH : zslab;
        MH_W2_UBLOAD(0, u0)
    }

    // accumulator set of output plane z: (z - zs) mod 3; input plane p = zs - 1 + k feeds sets k, k - 1, k - 2 (mod 3)
    for (int p = zs - 1; p <= ze; p += 3) {
        MH_W2_PLANE(p, 0, 2, 1)
        MH_W2_PLANE(p + 1, 1, 0, 2)
        MH_W2_PLANE(p + 2, 2, 1, 0)
    }
#undef MH_W2_PLANE
#undef MH_W2_EMIT
#undef MH_W2_STEP
#undef MH_W2_MFMA16
#undef MH_W2_UBLOAD
#undef MH_W2_TRANSFORM
#undef MH_W2_READ_PATCH
#undef MH_W2_COMMIT
#undef MH_W2_ISSUE

    if (STATS) {
        // lanes kq = 0..3 hold disjoint rows of the same cout; then the four waves (disjoint blocks) merge through LDS
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            Stat ot;
            ot.n = __shfl_xor(run.n, o);
            ot.mean = __shfl_xor(run.mean, o);
            ot.m2 = __shfl_xor(run.m2, o);
            run = stat_merge(run, ot);
        }
        __syncthreads();     // the staging buffers are free
        if (kq == 0) {
            float* red = smem + (wave * W2_CN + t16) * 3;
            red[0] = run.n; red[1] = run.mean; red[2] = run.m2;
        }
        __syncthreads();
        if (tid < W2_CN) {
            Stat st;
            st.n = smem[tid * 3]; st.mean = smem[tid * 3 + 1]; st.m2 = smem[tid * 3 + 2];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                Stat ot;
                ot.n = smem[(w * W2_CN + tid) * 3]; ot.mean = smem[(w * W2_CN + tid) * 3 + 1]; ot.m2 = smem[(w * W2_CN + tid) * 3 + 2];
                st = stat_merge(st, ot);
            }
            float* rec = stats + (((long long)n * Cout + cg * W2_CN + tid) * nblk + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}
#undef MH_W2_BT

// Weight transform U^kz = G g[kz] G^T (y, x), written in the order operand B is read:
// up[cout group][cin step][lane = (cin & 3) * 16 + (cout & 15)][kz * 16 + xi] with lane pitch 52 inside a 4096-float
// slab (the launcher zeroes the padding).  One thread per (cout, cin).
__global__ void __launch_bounds__(256)
conv3d_k3_wino2d_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ up) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const int KS = Cin / W2_KC;
    float* dst = up + ((long long)(co / W2_CN) * KS + ci / W2_KC) * W2_UBUF + ((ci % W2_KC) * 16 + (co % W2_CN)) * W2_UPITCH;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
        float g[3][3], a[4][3], u[4][4];
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = w[((long long)co * Cin + ci) * 27 + kz * 9 + t];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            a[0][x] = g[0][x];
            a[1][x] = 0.5f * ((g[0][x] + g[1][x]) + g[2][x]);
            a[2][x] = 0.5f * ((g[0][x] - g[1][x]) + g[2][x]);
            a[3][x] = g[2][x];
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            u[y][0] = a[y][0];
            u[y][1] = 0.5f * ((a[y][0] + a[y][1]) + a[y][2]);
            u[y][2] = 0.5f * ((a[y][0] - a[y][1]) + a[y][2]);
            u[y][3] = a[y][2];
        }
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) dst[kz * 16 + xi] = u[xi / 4][xi % 4];
    }
}

}  // namespace mh
