// Conv3d 3x3x3 / stride 1 / zero padding 1: the in-plane Winograd F(2x2, 3x3) + three direct z taps of conv3d_wino2d.h with
// the waves of a workgroup SPECIALISED: one matrix wave and one staging wave per SIMD.
//
// Why.  On gfx950 the fp32 MFMA runs on the SIMD's fp32 ALUs: a wave's own VALU / LDS / VMEM instructions do not hide behind
// its MFMAs, and a single wave issues a non-MFMA instruction only every ~5 cycles.  conv3d_wino2d.h (one 384-register wave per
// SIMD doing everything) therefore keeps the matrix pipe 48 % busy; giving each SIMD two IDENTICAL waves (conv3d_wino2p.h)
// gains only ~6 %: both halves run the same phases between the same barriers, so they want the matrix pipe -- and then the
// staging work -- at the same time, and the first to finish idles at the barrier (profiles/r02_pmc_wino2p_v1.txt).  Here the
// roles are split, so the barrier never holds up the matrix pipe:
//   * waves 0-3, one per SIMD -- CONSUMERS: nothing but 48 MFMAs per K-step (an 8x8 output block x 16 couts x 16 transform
//     positions x 3 z-taps, 192 accumulator registers as in conv3d_wino2d.h), their operands from LDS -- 16 ds_read_b128 per
//     48 MFMAs, requested one group of 12 MFMAs ahead -- and, once per output plane, the inverse transform / bias / statistics /
//     store;
//   * waves 4-7, one per SIMD -- PRODUCERS: global loads, normalise + activate, stage the input region and the weight slab in
//     LDS, transform the 4x4 patches of one block (B^T d B) and write them to LDS in the consumers' operand order.  A producer
//     finishes a K-step's work in well under the 1536 cycles the consumer's MFMAs take and then waits at the barrier, which costs
//     the matrix pipe nothing; it runs two K-steps ahead, so the consumer's first operands of a K-step are already on their way
//     when the previous one ends.
// Same reference op, weight packing, statistics records and launch geometry as conv3d_wino2d.h; the convolution values are
// bit-identical to it (same operations, same order).
//
// MEASURED (profiles/r02_wino2_impls.json, 32 -> 32 ch, 96^3, 64 windows): 18.8 ms -- no faster than conv3d_wino2d.h (18.7) and
// slower than conv3d_wino2p.h (17.8), so this kernel is opt-in (MONAI_AMD_W2_IMPL=s).  The reason is the arbitration of the
// shared fp32 ALU: while the matrix wave streams MFMAs back to back, the staging wave on the same SIMD gets ONE VALU instruction
// into each 32-cycle MFMA boundary (20.1 ms without `s_setprio`, where it got none until the matrix wave parked), so its ~90
// VALU instructions per K-step take ~3000 cycles and the matrix wave waits for it at the barrier.  On gfx950 fp32-MFMA time and
// VALU time on a SIMD add up whoever issues them; what two waves per SIMD can hide is latency and issue gaps, not ALU work.
//
// LDS (139 KB): V ring of 3 (transformed patches of a K-step: [block][lane][16 + 4 pad], 20 KB each), U ring of 4 (weight slabs,
// 16 KB each), raw input ring of 2 ([4 ch][18][20], 5.6 KB each).  Time slot t (one barrier each): consumers compute K-step t;
// producers transform K-step t+2, commit K-step t+3 to LDS and issue the global loads of K-step t+4.
#pragma once
#include "common.h"
#include "conv3d_wino2d.h"

namespace mh {

constexpr int W2S_VP = 20;                                 // floats per lane in a V block: 16 values + 4 pad (conflict-free 16-byte reads)
constexpr int W2S_VBLK = 64 * W2S_VP;                      // one block's transformed patches of one K-step
constexpr int W2S_VBUF = 4 * W2S_VBLK;
constexpr int W2S_NV = 3, W2S_NU = 4, W2S_NX = 2;
constexpr int W2S_SMEM = W2S_NV * W2S_VBUF + W2S_NU * W2_UBUF + W2S_NX * W2_XBUF + W2_ZSLAB + 64;

#define MH_W2S_BT(o0, o1, o2, o3, d0, d1, d2, d3) \
    { o0 = (d0) - (d2); o1 = (d1) + (d2); o2 = (d2) - (d1); o3 = (d1) - (d3); }

template <bool STATS, bool NRM>
__global__ void __launch_bounds__(512, 1)
conv3d_k3_wino2s_kernel(Tensor in, const float* __restrict__ up, const float* __restrict__ bias, Tensor out,
                        float* __restrict__ stats, int bxn, int byn, int zchunk, unsigned nblk) {
    __shared__ __attribute__((aligned(16))) float smem[W2S_SMEM];
    float* const vs = smem;
    float* const us = vs + W2S_NV * W2S_VBUF;
    float* const xs = us + W2S_NU * W2_UBUF;
    float* const zslab_w = xs + W2S_NX * W2_XBUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t16 = lane & 15, kq = lane >> 4;
    const int Cin = in.C, Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long HW = (long long)H * W, DHW = (long long)D * HW;
    const int KS = Cin / W2_KC;

    const unsigned ncg = (unsigned)(Cout / W2_CN);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(lid % ncg);
    lid /= ncg;
    const unsigned b = lid % nblk;
    const int n = (int)(lid / nblk);
    const int x0 = (int)(b % bxn) * W2_B, y0 = (int)((b / bxn) % byn) * W2_B;
    const int zs = (int)(b / (bxn * byn)) * zchunk, ze = min(zs + zchunk, D);
    const int p_first = max(zs - 1, 0), p_last = min(ze, D - 1);
    const int T = (p_last - p_first + 1) * KS;               // K-steps this workgroup runs

    if (tid < W2_ZSLAB) zslab_w[tid] = 0.0f;
    const float* const zslab = zslab_w;
    Stat run;
    run.n = 0.0f; run.mean = 0.0f; run.m2 = 0.0f;

    if (wave >= 4) {
        // ================================================================ producer: wave q stages channel q, transforms block q
        const int q = wave - 4;
        // the matrix wave of this SIMD is older and issues MFMAs back to back: without a higher priority this wave's VALU work is
        // starved until the matrix wave parks at the barrier, and the two roles run one after the other (measured: 0.44 of the
        // matrix peak, slower than the unspecialised kernels)
        __builtin_amdgcn_s_setprio(3);
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
            loff[j] = inreg ? q * W2_CS + ly * W2_PX + lx : -1;
        }
        float* const dump = zslab_w + W2_ZSLAB + lane;
        const float* src = in.data + (long long)n * in.n_stride + (long long)q * DHW;
        const f32x4* ug = reinterpret_cast<const f32x4*>(up + (long long)cg * KS * W2_UBUF) + (tid - 256);
        int ip = p_first, is = 0, cs = 0;
        const float* xptr = src + (long long)ip * HW;
        const f32x4* uptr = ug;
        const long long xstep = (long long)W2_KC * DHW, xwrap = (long long)KS * W2_KC * DHW;
        const float* nptr = NRM ? in.nrm + (long long)n * in.nrm_n_stride + 4LL * q : nullptr;
        float xin[W2_SLOTS];
        f32x4 uin[4];
        const int wby = q >> 1, wbx = q & 1;
        const int pbase = kq * W2_CS + (8 * wby + 2 * (t16 >> 2)) * W2_PX + 8 * wbx + 2 * (t16 & 3);
        const int vbase = q * W2S_VBLK + lane * W2S_VP;
        int xc = 0, uc = 0, xt = 0, vt = 0;                  // ring cursors: commit (raw / weights), transform (raw source / V target)
#define MH_W2S_ISSUE                                                                                  \
        {                                                                                             \
            _Pragma("unroll") for (int j = 0; j < W2_SLOTS; ++j) xin[j] = xptr[soff[j]];              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) uin[j] = uptr[256 * j];                     \
            xptr += xstep; uptr += W2_UBUF / 4;                                                       \
            if (++is == KS) { is = 0; uptr = ug; xptr -= xwrap; ++ip; xptr += HW; }                   \
        }
#define MH_W2S_COMMIT                                                                                 \
        {                                                                                             \
            float4 a_ = make_float4(1.0f, 0.0f, 1.0f, 0.0f);                                          \
            if (NRM) a_ = *reinterpret_cast<const float4*>(nptr + 16 * cs);                           \
            float* xb_ = xs + xc * W2_XBUF;                                                           \
            _Pragma("unroll") for (int j = 0; j < W2_SLOTS; ++j) {                                    \
                const float val_ = ((sokm >> j) & 1u) ? act(xin[j], a_.x, a_.y, a_.z) : 0.0f;         \
                if (64 * j + 63 < W2_R * W2_R) xb_[loff[j]] = val_;                                   \
                else *(loff[j] >= 0 ? xb_ + loff[j] : dump) = val_;                                   \
            }                                                                                         \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                             \
                reinterpret_cast<f32x4*>(us + uc * W2_UBUF)[(tid - 256) + 256 * j] = uin[j];          \
            if (++cs == KS) cs = 0;                                                                   \
            xc ^= 1; uc = (uc + 1) & 3;                                                               \
        }
#define MH_W2S_TRANSFORM                                                                              \
        {                                                                                             \
            const float* xp_ = xs + xt * W2_XBUF + pbase;                                             \
            float raw_[4][4], tv_[4][4], vv_[16];                                                     \
            _Pragma("unroll") for (int y = 0; y < 4; ++y) {                                           \
                const f32x2 lo_ = *reinterpret_cast<const f32x2*>(xp_ + y * W2_PX);                   \
                const f32x2 hi_ = *reinterpret_cast<const f32x2*>(xp_ + y * W2_PX + 2);               \
                raw_[y][0] = lo_[0]; raw_[y][1] = lo_[1]; raw_[y][2] = hi_[0]; raw_[y][3] = hi_[1];   \
            }                                                                                         \
            _Pragma("unroll") for (int x = 0; x < 4; ++x)                                             \
                MH_W2S_BT(tv_[0][x], tv_[1][x], tv_[2][x], tv_[3][x], raw_[0][x], raw_[1][x], raw_[2][x], raw_[3][x]) \
            _Pragma("unroll") for (int y = 0; y < 4; ++y)                                             \
                MH_W2S_BT(vv_[y * 4 + 0], vv_[y * 4 + 1], vv_[y * 4 + 2], vv_[y * 4 + 3], tv_[y][0], tv_[y][1], tv_[y][2], tv_[y][3]) \
            f32x4* vd_ = reinterpret_cast<f32x4*>(vs + vt * W2S_VBUF + vbase);                        \
            _Pragma("unroll") for (int g4 = 0; g4 < 4; ++g4)                                          \
                vd_[g4] = f32x4{vv_[4 * g4], vv_[4 * g4 + 1], vv_[4 * g4 + 2], vv_[4 * g4 + 3]};      \
            xt ^= 1; vt = vt == W2S_NV - 1 ? 0 : vt + 1;                                              \
        }
        MH_W2S_ISSUE                                          // K-step 0
        for (int t = -3; t < T; ++t) {
            if (t + 2 >= 0 && t + 2 < T) MH_W2S_TRANSFORM
            __builtin_amdgcn_sched_barrier(0);
            if (t + 3 < T) MH_W2S_COMMIT
            __builtin_amdgcn_sched_barrier(0);
            if (t + 4 < T) MH_W2S_ISSUE
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
#undef MH_W2S_TRANSFORM
#undef MH_W2S_COMMIT
#undef MH_W2S_ISSUE
    } else {
        // ================================================================ consumer: block `wave`, 16 couts, all 16 positions
        const int wby = wave >> 1, wbx = wave & 1;
        f32x4 acc[3][16];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const int co = cg * W2_CN + t16;
        const float bco = bias ? bias[co] : 0.0f;
        const int gy0 = y0 + 8 * wby + 2 * kq, gx0 = x0 + 8 * wbx;
        const bool rok0 = gy0 < H && gx0 < W, rok1 = gy0 + 1 < H && gx0 < W;
        float* const obase = out.data + (long long)n * out.n_stride + (long long)co * DHW + (long long)gy0 * W + gx0;
        const int vlane = wave * W2S_VBLK + lane * W2S_VP, ulane = lane * W2_UPITCH;
        int vr = 0, ur = 0;                                  // ring cursors of the K-step being computed
        bool have = false;                                   // group 0 of the current K-step already requested
        f32x4 av[2], bv[3];
        // operands of group G (transform positions 4 G .. 4 G + 3) of the K-step in rings (VR, UR): A = 4 transformed patch values of
        // this lane's (tile, channel); B(kz) = the 4 weights of this lane's (channel, cout) for z-tap kz (zero slab when the tap
        // leaves the chunk)
#define MH_W2S_LOADA(SET, VP, G) av[SET] = *reinterpret_cast<const f32x4*>((VP) + 4 * (G));
#define MH_W2S_LOADB(KZ, UP, G) bv[KZ] = *reinterpret_cast<const f32x4*>((UP)[KZ] + 4 * (G));
#define MH_W2S_MFMA4(SET, KZ, G, S)                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
            acc[S][4 * (G) + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[SET][j], bv[KZ][j], acc[S][4 * (G) + j], 0, 0, 0);
        // group G of the current K-step: request the next group's A, then per z-tap 4 MFMAs followed by the request of the next
        // group's B for that tap (needed 8 MFMAs = 256 cycles later).  (NVP, NUP, NG) = operand pointers and index of the next group.
#define MH_W2S_GROUP(G, SP1, S0, SM1, NEXT, NVP, NUP, NG)                                             \
        {                                                                                             \
            if (NEXT) MH_W2S_LOADA(((G) + 1) & 1, NVP, NG)                                            \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            MH_W2S_MFMA4((G) & 1, 0, G, SP1)                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            if (NEXT) MH_W2S_LOADB(0, NUP, NG)                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            MH_W2S_MFMA4((G) & 1, 1, G, S0)                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            if (NEXT) MH_W2S_LOADB(1, NUP, NG)                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            MH_W2S_MFMA4((G) & 1, 2, G, SM1)                                                          \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            if (NEXT) MH_W2S_LOADB(2, NUP, NG)                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }
        // one K-step: four groups of 12 MFMAs; MORE = another K-step of the same plane follows (its first group is requested
        // during the last group: the producers finished it two slots ago).  A z-tap that leaves the chunk reads the zero slab
        // (64 zeros: every group offset stays inside it).
#define MH_W2S_KSTEP(SP1, S0, SM1, MORE)                                                              \
        {                                                                                             \
            const int vn_ = vr == W2S_NV - 1 ? 0 : vr + 1, un_ = (ur + 1) & 3;                        \
            const float* vp_ = vs + vr * W2S_VBUF + vlane;                                            \
            const float* vq_ = vs + vn_ * W2S_VBUF + vlane;                                           \
            const float* ub_ = us + ur * W2_UBUF + ulane;                                             \
            const float* uc_ = us + un_ * W2_UBUF + ulane;                                            \
            const float* up_[3] = {k0ok ? ub_ : zslab, k1ok ? ub_ + 16 : zslab, k2ok ? ub_ + 32 : zslab};   \
            const float* uq_[3] = {k0ok ? uc_ : zslab, k1ok ? uc_ + 16 : zslab, k2ok ? uc_ + 32 : zslab};   \
            if (!have) {                                                                              \
                MH_W2S_LOADA(0, vp_, 0)                                                               \
                MH_W2S_LOADB(0, up_, 0) MH_W2S_LOADB(1, up_, 0) MH_W2S_LOADB(2, up_, 0)               \
            }                                                                                         \
            MH_W2S_GROUP(0, SP1, S0, SM1, true, vp_, up_, 1)                                          \
            MH_W2S_GROUP(1, SP1, S0, SM1, true, vp_, up_, 2)                                          \
            MH_W2S_GROUP(2, SP1, S0, SM1, true, vp_, up_, 3)                                          \
            MH_W2S_GROUP(3, SP1, S0, SM1, MORE, vq_, uq_, 0)                                          \
            have = (MORE);                                                                            \
            vr = vn_; ur = un_;                                                                       \
            __syncthreads();                                                                          \
        }
        // output plane Z is complete in accumulator set S: inverse transform, bias, statistics, store, clear (conv3d_wino2d.h)
#define MH_W2S_EMIT(S, Z)                                                                             \
        {                                                                                             \
            f32x4 o_[2][2];                                                                           \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                           \
                f32x4 pr_[4];                                                                         \
                _Pragma("unroll") for (int a = 0; a < 4; ++a) {                                       \
                    const f32x4 m0 = acc[S][a * 4 + 0], m1 = acc[S][a * 4 + 1], m2 = acc[S][a * 4 + 2], m3 = acc[S][a * 4 + 3]; \
                    pr_[a] = e == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;                                \
                }                                                                                     \
                o_[0][e] = ((pr_[0] + pr_[1]) + pr_[2]) + bco;                                        \
                o_[1][e] = ((pr_[1] - pr_[2]) - pr_[3]) + bco;                                        \
            }                                                                                         \
            float* op_ = obase + (long long)(Z) * HW;                                                 \
            _Pragma("unroll") for (int f = 0; f < 2; ++f) {                                           \
                if (f == 0 ? rok0 : rok1) {                                                           \
                    *reinterpret_cast<f32x4*>(op_ + f * W) = f32x4{o_[f][0][0], o_[f][1][0], o_[f][0][1], o_[f][1][1]};     \
                    *reinterpret_cast<f32x4*>(op_ + f * W + 4) = f32x4{o_[f][0][2], o_[f][1][2], o_[f][0][3], o_[f][1][3]}; \
                }                                                                                     \
            }                                                                                         \
            if (STATS) {                                                                              \
                Stat loc_;                                                                            \
                const float w0_ = rok0 ? 1.0f : 0.0f, w1_ = rok1 ? 1.0f : 0.0f;                       \
                loc_.n = 8.0f * (w0_ + w1_);                                                          \
                const f32x4 s4_ = (o_[0][0] + o_[0][1]) * w0_ + (o_[1][0] + o_[1][1]) * w1_;          \
                const float sum_ = (s4_[0] + s4_[1]) + (s4_[2] + s4_[3]);                             \
                loc_.mean = loc_.n > 0.0f ? sum_ / loc_.n : 0.0f;                                     \
                const f32x4 d00_ = o_[0][0] - loc_.mean, d01_ = o_[0][1] - loc_.mean, d10_ = o_[1][0] - loc_.mean, d11_ = o_[1][1] - loc_.mean; \
                const f32x4 q4_ = (d00_ * d00_ + d01_ * d01_) * w0_ + (d10_ * d10_ + d11_ * d11_) * w1_; \
                loc_.m2 = (q4_[0] + q4_[1]) + (q4_[2] + q4_[3]);                                      \
                run = stat_merge(run, loc_);                                                          \
            }                                                                                         \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[S][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; \
        }
#define MH_W2S_PLANE(P, SP1, S0, SM1)                                                                 \
        if ((P) <= ze) {                                                                              \
            const int p_ = (P);                                                                       \
            if (p_ >= 0 && p_ <= p_last) {                                                            \
                const bool k0ok = p_ + 1 < ze, k1ok = p_ >= zs && p_ < ze, k2ok = p_ - 1 >= zs;       \
                for (int s = 0; s < KS - 1; ++s) MH_W2S_KSTEP(SP1, S0, SM1, true)                     \
                MH_W2S_KSTEP(SP1, S0, SM1, false)                                                     \
            }                                                                                         \
            if (p_ - 1 >= zs) MH_W2S_EMIT(SM1, p_ - 1)                                                \
        }
        __syncthreads();                                      // producer slots -3, -2, -1
        __syncthreads();
        __syncthreads();
        for (int p = zs - 1; p <= ze; p += 3) {
            MH_W2S_PLANE(p, 0, 2, 1)
            MH_W2S_PLANE(p + 1, 1, 0, 2)
            MH_W2S_PLANE(p + 2, 2, 1, 0)
        }
#undef MH_W2S_PLANE
#undef MH_W2S_EMIT
#undef MH_W2S_KSTEP
#undef MH_W2S_GROUP
#undef MH_W2S_MFMA4
#undef MH_W2S_LOADB
#undef MH_W2S_LOADA
    }

    if (STATS) {
        // consumer lanes kq = 0..3 hold disjoint rows of the same cout; then the four consumer waves merge through LDS
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            Stat ot;
            ot.n = __shfl_xor(run.n, o);
            ot.mean = __shfl_xor(run.mean, o);
            ot.m2 = __shfl_xor(run.m2, o);
            run = stat_merge(run, ot);
        }
        __syncthreads();     // the rings are free
        if (wave < 4 && kq == 0) {
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

#undef MH_W2S_BT

}  // namespace mh
