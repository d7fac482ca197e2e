// Conv3d 3x3x3 / stride 1 / zero padding 1: the in-plane Winograd F(2x2, 3x3) + three direct z taps of conv3d_wino2d.h,
// re-organised so that TWO waves share each SIMD.
//
// Why.  On gfx950 the fp32 MFMA occupies the SIMD's fp32 ALUs, so a wave's own VALU / LDS / VMEM instructions never hide
// behind its MFMAs, and one wave issues a non-MFMA instruction only every ~5 cycles: conv3d_wino2d.h (one 384-register wave
// per SIMD) keeps the matrix pipe 48 % busy, which is exactly what a synthetic "16 MFMAs + the step's other instructions"
// loop reaches with one wave per SIMD (0.52) -- and the same loop with two waves per SIMD reaches 0.77
// (tools/ubench/coexec2.hip, profiles/r02_ubench_coexec2.txt): while one wave sits in its issue-bound lump or waits on LDS,
// the other one's MFMAs run.  Two waves per SIMD need <= 256 registers per lane, i.e. half the accumulators per wave.
//
// How.  The 16 transform-domain positions xi = (i, j) of a 4x4 Winograd tile are independent GEMMs.  The two waves of a
// pair own the SAME 8x8 output block and the same 16 output channels but different positions: wave half `ph` owns rows
// i in {2 ph, 2 ph + 1} (8 positions x 3 z-taps x 4 registers = 96 accumulators instead of 192).  Each half needs only
// three of the four patch rows and half of the input transform (8 + 8 adds).  The inverse transform separates the same way:
// its first stage (over j) is local to a half; the second stage (over i) is
//     out row 0 = ((r0 + r1) + r2) + bias      out row 1 = ((r1 - r2) - r3) + bias
// so half 0 sends r1, half 1 sends r2 through LDS (once per output plane, not per step) and each half finishes, reduces
// (InstanceNorm statistics) and stores ONE of the two output rows of every tile -- the same operations in the same order
// as conv3d_wino2d.h: the convolution values are bit-identical to it (the statistics records merge in a different order).
//
// A workgroup = 8 waves = (2 x 2 spatial blocks of 8x8 outputs) x (2 position halves) = a 16x16 (y, x) region x 16 couts x
// one z-chunk, marching along z exactly like conv3d_wino2d.h (same weight packing, same statistics records, same launch
// geometry).  A stage = 8 input channels (two MFMA K-steps): wave w stages channel w of the stage's input plane region
// [8][18][18] (norm + activation applied on the way) and 1/8 of the stage's two weight slabs; two LDS buffers, one barrier per
// stage; the global loads of stage g + 2 are issued right after stage g + 1 has been committed to LDS, so they are in flight
// for a whole stage.  Inside a stage each wave runs 2 x 3 bursts of 8 MFMAs with the next burst's B operands requested
// before the current burst issues.
#pragma once
#include "common.h"
#include "conv3d_wino2d.h"

namespace mh {

constexpr int W2P_KB = 8;                                   // input channels per stage
constexpr int W2P_XST = W2P_KB * W2_CS;                     // staged planes of one stage (floats)
constexpr int W2P_UST = 2 * W2_UBUF;                        // two weight slabs per stage
constexpr int W2P_EXCH = 2 * 8 * 2 * 64 * 4;                // two alternating slots of: per wave 2 x f32x4 per lane for the partner half
constexpr int W2P_SMEM = 2 * (W2P_XST + W2P_UST) + W2P_EXCH + W2_ZSLAB + 64;

template <bool STATS, bool NRM>
__global__ void __launch_bounds__(512, 1)
conv3d_k3_wino2p_kernel(Tensor in, const float* __restrict__ up, const float* __restrict__ bias, Tensor out,
                        float* __restrict__ stats, int bxn, int byn, int zchunk, unsigned nblk) {
    __shared__ __attribute__((aligned(16))) float smem[W2P_SMEM];
    float* const xs = smem;
    float* const us = smem + 2 * W2P_XST;
    float* const ex = us + 2 * W2P_UST;
    float* const zslab_w = ex + W2P_EXCH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sb = wave & 3, ph = wave >> 2;                   // waves w and w + 4 (same SIMD) are the two halves of a block
    const int t16 = lane & 15, kq = lane >> 4;
    const int Cin = in.C, Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long HW = (long long)H * W, DHW = (long long)D * HW;
    const int NSTG = Cin / W2P_KB;                            // stages per input plane (the launcher requires Cin % 8 == 0)

    // launch geometry of conv3d_wino2d.h: 1-D over (window, region, cout group), cout group fastest, XCD-aware
    const unsigned ncg = (unsigned)(Cout / W2_CN);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(lid % ncg);
    lid /= ncg;
    const unsigned b = lid % nblk;
    const int n = (int)(lid / nblk);
    const int x0 = (int)(b % bxn) * W2_B, y0 = (int)((b / bxn) % byn) * W2_B;
    const int zs = (int)(b / (bxn * byn)) * zchunk, ze = min(zs + zchunk, D);
    const int p_first = max(zs - 1, 0), p_last = min(ze, D - 1);
    const int T = (p_last - p_first + 1) * NSTG;              // stages this workgroup runs

    // staging: wave w stages channel w of the stage; lane elements e = lane + 64 j of the 18 x 18 region
    unsigned soff[W2_SLOTS];          // BYTE offsets, unsigned: `global_load_dword v, v_off, s[base]` -- no per-load 64-bit VALU address arithmetic
    int loff[W2_SLOTS];
    unsigned sokm = 0u;
#pragma unroll
    for (int j = 0; j < W2_SLOTS; ++j) {
        const int e = lane + 64 * j;
        const int ly = e / W2_R, lx = e - ly * W2_R;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        const bool inreg = e < W2_R * W2_R;
        const bool ok = inreg && gy >= 0 && gy < H && gx >= 0 && gx < W;
        sokm |= (unsigned)ok << j;
        soff[j] = ok ? 4u * (unsigned)(gy * W + gx) : 0u;
        loff[j] = inreg ? wave * W2_CS + ly * W2_PX + lx : -1;
    }
    float* const dump = zslab_w + W2_ZSLAB + lane;
    if (tid < W2_ZSLAB) zslab_w[tid] = 0.0f;
    const float* const zslab = zslab_w;

    const float* src = in.data + (long long)n * in.n_stride + (long long)wave * DHW;
    const f32x4* ug = reinterpret_cast<const f32x4*>(up + (long long)cg * (Cin / W2_KC) * W2_UBUF) + tid;
    int ip = p_first, is = 0, cs = 0;
    const float* xptr = src + (long long)ip * HW;
    const f32x4* uptr = ug;
    const long long xstep = (long long)W2P_KB * DHW, xwrap = (long long)NSTG * W2P_KB * DHW;
    const float* nptr = NRM ? in.nrm + (long long)n * in.nrm_n_stride + 4LL * wave : nullptr;
    float xin[W2_SLOTS];
    f32x4 uin[4];
#define MH_W2P_ISSUE                                                                                  \
    {                                                                                                 \
        const char* xq_ = reinterpret_cast<const char*>(xptr);            /* wave-uniform base + 32-bit lane offset */ \
        _Pragma("unroll") for (int j = 0; j < W2_SLOTS; ++j) xin[j] = *reinterpret_cast<const float*>(xq_ + soff[j]); \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) uin[j] = uptr[512 * j];                         \
        xptr += xstep; uptr += W2P_UST / 4;                                                           \
        if (++is == NSTG) { is = 0; uptr = ug; xptr -= xwrap; ++ip; xptr += HW; }                     \
    }
    // normalise + activate on the way into LDS.  y = fma(x, alpha, beta); leaky(y) = y > 0 ? y : y * slope equals max(y, y * slope)
    // whenever slope <= 1 (LeakyReLU, ReLU, identity): three packed operations per PAIR of elements instead of four scalar ones
    // per element; a slope > 1 (a trained PReLU could have one) takes the general form.  The records of a stage's channel are
    // wave-uniform, so the choice is a scalar branch.
#define MH_W2P_COMMIT(BUF)                                                                            \
    {                                                                                                 \
        float4 a_ = make_float4(1.0f, 0.0f, 1.0f, 0.0f);                                              \
        if (NRM) a_ = *reinterpret_cast<const float4*>(nptr + 4 * W2P_KB * cs);                       \
        float* xb_ = xs + (BUF) * W2P_XST;                                                            \
        float val_[W2_SLOTS];                                                                         \
        if (__builtin_amdgcn_readfirstlane(__float_as_int(a_.z)) <= 0x3f800000) {   /* slope <= 1 (or negative) */ \
            const f32x2 al_ = {a_.x, a_.x}, be_ = {a_.y, a_.y}, sl_ = {a_.z, a_.z};                   \
            _Pragma("unroll") for (int j = 0; j < W2_SLOTS; j += 2) {                                 \
                f32x2 y_ = __builtin_elementwise_fma(f32x2{xin[j], xin[j + 1]}, al_, be_);            \
                y_ = __builtin_elementwise_max(y_, y_ * sl_);                                         \
                val_[j] = y_[0]; val_[j + 1] = y_[1];                                                 \
            }                                                                                         \
        } else {                                                                                      \
            _Pragma("unroll") for (int j = 0; j < W2_SLOTS; ++j) val_[j] = act(xin[j], a_.x, a_.y, a_.z); \
        }                                                                                             \
        _Pragma("unroll") for (int j = 0; j < W2_SLOTS; ++j) {                                        \
            const float v_ = ((sokm >> j) & 1u) ? val_[j] : 0.0f;                                     \
            if (64 * j + 63 < W2_R * W2_R) xb_[loff[j]] = v_;                                         \
            else *(loff[j] >= 0 ? xb_ + loff[j] : dump) = v_;                                         \
        }                                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
            reinterpret_cast<f32x4*>(us + (BUF) * W2P_UST)[tid + 512 * j] = uin[j];                   \
        if (++cs == NSTG) cs = 0;                                                                     \
    }

    // this lane's patch: tile t16 of the block's 4 x 4, input channel kq of the K-step; rows ph .. ph + 2 of its 4 x 4 patch
    const int wby = sb >> 1, wbx = sb & 1;
    const int pbase = kq * W2_CS + (8 * wby + 2 * (t16 >> 2) + ph) * W2_PX + 8 * wbx + 2 * (t16 & 3);
    const int ubase = lane * W2_UPITCH + 8 * ph;              // this half's 8 positions of a z-tap's 16

    f32x4 acc[3][8];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[q][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // epilogue: lane (kq, t16) of half ph finishes output row 2 kq + ph (8 contiguous x) of cout t16 of the block
    const int co = cg * W2_CN + t16;
    const float bco = bias ? bias[co] : 0.0f;
    const int gy = y0 + 8 * wby + 2 * kq + ph, gx0 = x0 + 8 * wbx;
    const bool rok = gy < H && gx0 < W;                        // W % 8 == 0: the 8 columns are in or out together
    float* const obase = out.data + (long long)n * out.n_stride + (long long)co * DHW + (long long)gy * W + gx0;
    f32x4* ex_mine = reinterpret_cast<f32x4*>(ex) + (wave * 2) * 64 + lane;
    const f32x4* ex_other = reinterpret_cast<const f32x4*>(ex) + ((wave ^ 4) * 2) * 64 + lane;
    int ex_flip = 8 * 2 * 64;             // consecutive output planes use alternating exchange slots (a fast half may not overwrite what its partner still reads)
    Stat run;
    run.n = 0.0f; run.mean = 0.0f; run.m2 = 0.0f;

    float vv[8];
    float ubr[2][8];
#define MH_W2P_UBLOAD(UR, UB)                                                                         \
    {                                                                                                 \
        const f32x4 t0_ = reinterpret_cast<const f32x4*>(UB)[0], t1_ = reinterpret_cast<const f32x4*>(UB)[1]; \
        ubr[UR][0] = t0_[0]; ubr[UR][1] = t0_[1]; ubr[UR][2] = t0_[2]; ubr[UR][3] = t0_[3];           \
        ubr[UR][4] = t1_[0]; ubr[UR][5] = t1_[1]; ubr[UR][6] = t1_[2]; ubr[UR][7] = t1_[3];           \
    }
#define MH_W2P_MFMA8(SET, UR)                                                                         \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                     \
        acc[SET][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[i], ubr[UR][i], acc[SET][i], 0, 0, 0);
    // patch rows ph .. ph + 2 -> this half's two rows of B^T d B.  B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:
    // half 0: (d0 - d2, d1 + d2); half 1: (d2 - d1, d1 - d3) -- with rows (a, b, c) = (d_ph, d_ph+1, d_ph+2):
    // half 0: (a - c, b + c); half 1: (b - a, a - c)
#define MH_W2P_PATCH(BUF, SUB)                                                                        \
    {                                                                                                 \
        const float* xp_ = xs + (BUF) * W2P_XST + (SUB) * (W2_KC * W2_CS) + pbase;                    \
        float r_[3][4];                                                                               \
        _Pragma("unroll") for (int y = 0; y < 3; ++y) {                                               \
            const f32x2 lo_ = *reinterpret_cast<const f32x2*>(xp_ + y * W2_PX);                       \
            const f32x2 hi_ = *reinterpret_cast<const f32x2*>(xp_ + y * W2_PX + 2);                   \
            r_[y][0] = lo_[0]; r_[y][1] = lo_[1]; r_[y][2] = hi_[0]; r_[y][3] = hi_[1];               \
        }                                                                                             \
        float t_[2][4];                                                                               \
        if (ph == 0) {                                                                                \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) { t_[0][x] = r_[0][x] - r_[2][x]; t_[1][x] = r_[1][x] + r_[2][x]; } \
        } else {                                                                                      \
            _Pragma("unroll") for (int x = 0; x < 4; ++x) { t_[0][x] = r_[1][x] - r_[0][x]; t_[1][x] = r_[0][x] - r_[2][x]; } \
        }                                                                                             \
        _Pragma("unroll") for (int y = 0; y < 2; ++y) {                                               \
            vv[4 * y + 0] = t_[y][0] - t_[y][2]; vv[4 * y + 1] = t_[y][1] + t_[y][2];                 \
            vv[4 * y + 2] = t_[y][2] - t_[y][1]; vv[4 * y + 3] = t_[y][1] - t_[y][3];                 \
        }                                                                                             \
    }

    // one K-step (4 channels) of the stage in buffer `bcur`: patch -> half transform -> three bursts of 8 MFMAs (z-taps 0, 1, 2
    // into the sets of output planes p+1, p, p-1); a z-tap that leaves the chunk multiplies by the zero slab (no branches)
#define MH_W2P_SUB(SUB, SP1, S0, SM1)                                                                 \
    {                                                                                                 \
        const float* ub_ = us + bcur * W2P_UST + (SUB) * W2_UBUF + ubase;                             \
        MH_W2P_UBLOAD(0, k0ok ? ub_ : zslab)                                                          \
        MH_W2P_PATCH(bcur, SUB)                                                                       \
        MH_W2P_UBLOAD(1, k1ok ? ub_ + 16 : zslab)                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        MH_W2P_MFMA8(SP1, 0)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        MH_W2P_UBLOAD(0, k2ok ? ub_ + 32 : zslab)                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        MH_W2P_MFMA8(S0, 1)                                                                           \
        MH_W2P_MFMA8(SM1, 0)                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
    // head of a stage: commit the next stage's staged registers to the other LDS buffer, issue the loads of the one after it
#define MH_W2P_HEAD                                                                                   \
    {                                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (pend) MH_W2P_EMIT_B                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (gi + 1 < T) MH_W2P_COMMIT(bcur ^ 1)                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        if (gi + 2 < T) MH_W2P_ISSUE                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
    // one stage (8 channels of input plane p).  The two halves of a pair run the SAME sequence in lockstep between barriers, and
    // the fp32 MFMA pipe serialises them -- so they are de-phased on purpose: half 0 does its staging work first while half 1
    // already runs the MFMAs of the first K-step, then they swap (profiles/r02_pmc_wino2p_v1.txt: in lockstep both halves sat in
    // the staging lump together and the matrix pipe idled half of the time).
#define MH_W2P_STAGE(SP1, S0, SM1)                                                                    \
    {                                                                                                 \
        if (ph == 0) MH_W2P_HEAD                                                                      \
        MH_W2P_SUB(0, SP1, S0, SM1)                                                                   \
        if (ph != 0) MH_W2P_HEAD                                                                      \
        MH_W2P_SUB(1, SP1, S0, SM1)                                                                   \
    }

    // output plane Z is complete in accumulator set S.  Part A (after the plane's last stage, followed by one barrier): first
    // inverse stage locally, this half's row for the partner into LDS, clear the set.  Part B (deferred into the staging slot of
    // the NEXT stage, i.e. beside the partner's MFMAs): second stage + bias, store this half's output row, statistics.
    // (Part A inside the stage loop, in front of the stage's own barrier, makes hipcc spill ~500 registers -- measured.)
    f32x4 prh[2][2];
    int pend = 0, pend_z = 0;
#define MH_W2P_EMIT_A(S, Z)                                                                           \
    {                                                                                                 \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                               \
            const f32x4 m0 = acc[S][a * 4 + 0], m1 = acc[S][a * 4 + 1], m2 = acc[S][a * 4 + 2], m3 = acc[S][a * 4 + 3]; \
            prh[a][0] = (m0 + m1) + m2;                                                               \
            prh[a][1] = (m1 - m2) - m3;                                                               \
        }                                                                                             \
        /* half 0 holds r0, r1 and sends r1; half 1 holds r2, r3 and sends r2 */                      \
        ex_mine[0] = ph == 0 ? prh[1][0] : prh[0][0];                                                 \
        ex_mine[64] = ph == 0 ? prh[1][1] : prh[0][1];                                                \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) acc[S][i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};      \
        pend = 1; pend_z = (Z);                                                                       \
    }
#define MH_W2P_EMIT_B                                                                                 \
    {                                                                                                 \
        const f32x4 g0_ = ex_other[0], g1_ = ex_other[64];                                            \
        ex_mine += ex_flip; ex_other += ex_flip; ex_flip = -ex_flip;                                  \
        f32x4 o_[2];                                                                                  \
        if (ph == 0) {                                                                                \
            o_[0] = ((prh[0][0] + prh[1][0]) + g0_) + bco;                                            \
            o_[1] = ((prh[0][1] + prh[1][1]) + g1_) + bco;                                            \
        } else {                                                                                      \
            o_[0] = ((g0_ - prh[0][0]) - prh[1][0]) + bco;                                            \
            o_[1] = ((g1_ - prh[0][1]) - prh[1][1]) + bco;                                            \
        }                                                                                             \
        if (rok) {                                                                                    \
            float* op_ = obase + (long long)pend_z * HW;                                              \
            *reinterpret_cast<f32x4*>(op_) = f32x4{o_[0][0], o_[1][0], o_[0][1], o_[1][1]};           \
            *reinterpret_cast<f32x4*>(op_ + 4) = f32x4{o_[0][2], o_[1][2], o_[0][3], o_[1][3]};       \
        }                                                                                             \
        if (STATS) {                                                                                  \
            Stat loc_;                                                                                \
            const float w_ = rok ? 1.0f : 0.0f;                                                       \
            loc_.n = 8.0f * w_;                                                                       \
            const f32x4 s4_ = (o_[0] + o_[1]) * w_;                                                   \
            const float sum_ = (s4_[0] + s4_[1]) + (s4_[2] + s4_[3]);                                 \
            loc_.mean = loc_.n > 0.0f ? sum_ / loc_.n : 0.0f;                                         \
            const f32x4 d0_ = o_[0] - loc_.mean, d1_ = o_[1] - loc_.mean;                             \
            const f32x4 q4_ = (d0_ * d0_ + d1_ * d1_) * w_;                                           \
            loc_.m2 = (q4_[0] + q4_[1]) + (q4_[2] + q4_[3]);                                          \
            run = stat_merge(run, loc_);                                                              \
        }                                                                                             \
        pend = 0;                                                                                     \
    }

#define MH_W2P_PLANE(P, SP1, S0, SM1)                                                                 \
    if ((P) <= ze) {                                                                                  \
        const int p_ = (P);                                                                           \
        if (p_ >= 0 && p_ <= p_last) {                                                                \
            const bool k0ok = p_ + 1 < ze, k1ok = p_ >= zs && p_ < ze, k2ok = p_ - 1 >= zs;           \
            for (int s = 0; s < NSTG; ++s) {                                                          \
                MH_W2P_STAGE(SP1, S0, SM1)                                                            \
                __syncthreads();                                                                      \
                bcur ^= 1; ++gi;                                                                      \
            }                                                                                         \
        }                                                                                             \
        if (p_ - 1 >= zs) {                                                                           \
            if (pend) MH_W2P_EMIT_B                                                                   \
            MH_W2P_EMIT_A(SM1, p_ - 1)                                                                \
            __syncthreads();                                                                          \
        }                                                                                             \
    }

    // prologue: stage 0 into buffer 0, the loads of stage 1 in flight
    int bcur = 0, gi = 0;
    MH_W2P_ISSUE
    MH_W2P_COMMIT(0)
    if (T > 1) MH_W2P_ISSUE
    __syncthreads();

    // accumulator set of output plane z: (z - zs) mod 3; input plane p = zs - 1 + k feeds sets k, k - 1, k - 2 (mod 3)
    for (int p = zs - 1; p <= ze; p += 3) {
        MH_W2P_PLANE(p, 0, 2, 1)
        MH_W2P_PLANE(p + 1, 1, 0, 2)
        MH_W2P_PLANE(p + 2, 2, 1, 0)
    }
    if (pend) MH_W2P_EMIT_B
#undef MH_W2P_PLANE
#undef MH_W2P_EMIT_A
#undef MH_W2P_EMIT_B
#undef MH_W2P_STAGE
#undef MH_W2P_HEAD
#undef MH_W2P_SUB
#undef MH_W2P_PATCH
#undef MH_W2P_MFMA8
#undef MH_W2P_UBLOAD
#undef MH_W2P_COMMIT
#undef MH_W2P_ISSUE

    if (STATS) {
        // lanes kq = 0..3 hold disjoint rows of the same cout; then the eight waves (disjoint rows / blocks) merge through LDS
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
            // fixed merge order: per block the rows of half 0, then of half 1; blocks 0..3
            Stat st;
            st.n = 0.0f; st.mean = 0.0f; st.m2 = 0.0f;
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int w = blk + 4 * h;
                    Stat ot;
                    ot.n = smem[(w * W2_CN + tid) * 3]; ot.mean = smem[(w * W2_CN + tid) * 3 + 1]; ot.m2 = smem[(w * W2_CN + tid) * 3 + 2];
                    st = stat_merge(st, ot);
                }
            }
            float* rec = stats + (((long long)n * Cout + cg * W2_CN + tid) * nblk + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}

}  // namespace mh
