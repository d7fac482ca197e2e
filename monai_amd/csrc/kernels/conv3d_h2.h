// Conv3d 3x3x3 / stride 1 / zero padding 1 on the FP16 matrix cores of gfx950 in two-piece split-precision arithmetic:
// fp32-equivalent results at 3/16 of the fp32 matrix-core cycles of a direct convolution, z-streaming like conv3d_wino2p.h.
//
// Same reference op and "normalise on load" contract as conv3d_mfma.h (nn.Conv3d of `Convolution`,
// monai/networks/blocks/convolutions.py:98-171, fed by the previous block's deferred InstanceNorm + LeakyReLU).
//
// Arithmetic.  Every fp32 operand is split into two fp16 pieces x = hi + lo (hi = fp16(x), lo = fp16(x - hi): 11 + 11
// significand bits, |x - hi - lo| <= 2^-23 |x|; the weights of a layer are first scaled by a power of two so that their low
// pieces stay normal, the accumulator is scaled back in the epilogue -- both exact).  A product x * w is evaluated as
// hi*hi + lo*hi + hi*lo, each piece product EXACT in fp32 (11 x 11 bits), accumulated in fp32 by v_mfma_f32_32x32x16_f16;
// the dropped lo*lo term is <= 2^-22 relative.  This is the error-free-splitting scheme of Ootomo & Yokota (fp32 GEMM on fp16
// tensor cores), without their second accumulator: gfx950's MFMA keeps fp16 subnormals (tools/ubench/mfma_f16.hip,
// profiles/r02_ubench_mfma_f16.txt), so unscaled low pieces lose nothing that matters.  Measured on the oracle network
// (tools/split_precision_numerics.py, BasicUNet, 2 x 64^3): max |logit difference| 4.0e-6 and identical argmax -- the noise
// level of two fp32 summation orders (3.6e-6); bf16 pieces need three pieces and six products for the same (conv3d_split.h).
// Range.  fp16 pieces need |activated input| < 65504 and lose their low piece below 2^-24, so the kernel does not take the
// input as it comes: every input record carries a bound on |activated value| over its (n, c) plane (common.h -- sqrt(count) *
// |gamma| + |beta| from the InstanceNorm / GroupNorm finalize, max |value| from the raw producers), the workgroup takes the
// power of two 2^s that puts the largest bound of sample n just below 2^15, multiplies the activated input by it (folded into
// the records' alpha and beta as they are copied to LDS: no extra instruction) and the epilogue scales back by 2^-s together with
// the weight scale -- all exact.
// Any finite magnitude is therefore in range, and small-magnitude tensors keep fp32-equivalent relative precision.  A bound
// that is inf / NaN (the input plane or its statistics hold a non-finite value) or 0 (none given: a caller that selected this
// kernel for an unbounded input) makes the sample's whole output NaN: after a normalisation that is exactly what the reference
// computes (the statistics of such a plane are NaN, so is every normalised value and every sum that contains one), and for
// unbounded callers it is a loud failure instead of a silent overflow.  mh_conv3d_k3_select only returns this configuration
// for inputs that carry bounds.
//
// Mapping.  GEMM M = output voxels, N = 32 output channels, K = 16 input channels per instruction, one instruction group per tap.
// A workgroup = 8 waves owns a 16 x 16 (y, x) region x 32 couts x one z-chunk and marches along z: input plane p is
// multiplied with the three z-taps into the accumulator sets of output planes p+1, p, p-1 (three rotating sets of 16
// registers), so every input plane is staged once.  Wave w owns rows 2w, 2w+1 of the region = one 32-voxel M block: operand
// A lane l = voxel (l & 31) x channels 8 (l >> 5) .. +7 (one 16-byte LDS read of a [k-group][voxel][8 channels] plane),
// operand B = cout (l & 31) x the same 8 channels, D lane = one cout x 16 voxels (statistics reduce in-lane).  Row pitch 20
// voxels and the second row rotated by 4 voxels make the 16-byte operand reads bank-conflict free.
//
// A step = 16 input channels of one input plane: the plane region [2 pieces][2 k-groups][18 x 20 voxels][8 ch] (23 KB) and the
// chunk's weight slab [2 pieces][27 taps][2 k-groups][32 couts][8 ch] (55 KB, padded to 57 KB) sit in one of two LDS buffers (157 KB
// + the {alpha, beta, slope} records, 48 bytes per channel quad); one barrier per step.  With at most two chunks (Cin <= 32, template RES) both slabs are
// loaded once and stay; otherwise a slab is streamed from L2 every step.
//
// Schedule.  A wave issues in order and an fp16 MFMA occupies the matrix pipe for 32 cycles: what is placed BETWEEN two MFMAs
// of a wave is free, what sits in a lump before or after them is serial time of that wave.  The first version kept the staging
// (loads, normalise + split, LDS writes, epilogue) in a lump per step: 6700 of 9700 cycles per step were that lump, the matrix
// pipe 42 % busy, and neither de-phasing the two waves of a SIMD nor moving work between them changed anything (the step was
// bounded by one wave's serial time, not by a shared resource; profiles/r02_pmc_h2_v1.txt, r02_h2_step_timeline_v1.txt).  Now
// every staging piece is branch-free (zeroed padding cells and dump cells instead of masks, clamped pointer advance instead of
// tail branches) and is dealt out over the 81 MFMA gaps of the step by sched_group_barrier: 10.7 -> 9.0-9.4 ms for 32 -> 32
// channels at 96^3 x 64 windows (the fp32 Winograd kernel: 17.5 ms; the matrix pipe alone at the sustained clock: 4.7 ms).
// Round 3 (DESIGN_HISTORY 4.1, profiles/r03_h2_*.txt, r03_pmc_h2.txt): the region is 16 x 16 or 8 x 32 (H2Geo<WIDE>, whichever covers the plane with fewer regions);
// the epilogue is four branch-free pieces in taps 3, 4, 6, 7 of the next plane's first step (raw buffer stores whose out-of-range lanes the hardware drops);
// the input records sit in LDS per channel quad and are read in the previous step's last tap; the input planes come in by raw buffer loads (no vector
// address arithmetic): 138 -> 91 vector instructions per step, 8.2-8.5 ms for the same launch, matrix pipe busy 0.71 of the cycles (0.61) -- and a lower
// clock in return: the bare instruction sustains 0.67 of its peak rate on random operands (profiles/r03_ubench_mfma_sustained.txt), this kernel 0.43-0.46.
// Also measured and not kept: 16-byte window loads of the input (fewer vector-memory instructions, 33 % more bytes: slower),
// loads issued up to 8 groups ahead (same), a second input register set (spills at the 256-register limit of two waves per SIMD).
// Timing experiments with parts switched off mislead on this chip: constant operands raise the clock (the matrix pipe draws
// less power), so "without loads" variants ran up to 40 % faster than the instruction stream explains.
#pragma once
#include "common.h"

namespace mh {


constexpr int H2_B = 16;                                   // region edge (y and x) of a workgroup
constexpr int H2_R = 18, H2_RS = 20;                       // input region edge, LDS row pitch in voxels
constexpr int H2_PV = H2_R * H2_RS;                        // voxels of a staged plane (360)
constexpr int H2_KC = 16, H2_CN = 32;                      // input channels per step (MFMA K), couts per workgroup
constexpr int H2_XV = 2 * H2_PV;                           // uint4 per piece of a step's input plane: [k-group][voxel]
constexpr int H2_XB = 2 * H2_XV;                           // uint4 per input buffer (two pieces): 1440
constexpr int H2_WV = 27 * 2 * H2_CN;                      // uint4 per piece of a chunk's weight slab: [tap][k-group][cout]
constexpr int H2_WSLOTS = 7;                               // uint4 of the weight slab per thread
constexpr int H2_WB = 512 * H2_WSLOTS;                     // uint4 per weight buffer: two pieces (3456) padded to 7 per thread (3584)
constexpr int H2_SLOTS = 3;                                // staging tasks per lane: (voxel, 4 channels); 162 voxels per wave
constexpr int H2_TAIL = 4;                                 // floats behind the packed slabs: {1 / scale, scale, 0, 0}
constexpr int H2_NRM_MAX = 256;                            // input channels: their {alpha, beta, slope} records sit in LDS, 12 bytes each (157 + 3 KB = all 160 KB)

// Region geometry of a workgroup: 16 x 16 outputs (a wave = two rows of 16) or, WIDE, 8 x 32 (a wave = one row of 32) -- the same 256 voxels, the same
// 360-voxel staged plane (18 rows x pitch 20 | 10 rows x pitch 36), so everything but the index arithmetic is shared.  The launcher picks the shape that
// covers the (H, W) plane with fewer regions (24 x 24: three 8 x 32 regions instead of four 16 x 16 ones).
template <bool WIDE> struct H2Geo {
    static constexpr int BY = WIDE ? 8 : H2_B, BX = WIDE ? 32 : H2_B;
    static constexpr int RY = BY + 2, RX = BX + 2, RS = RX + 2;      // staged rows, columns, LDS row pitch in voxels
    static constexpr int NV = RY * RX, HALF = NV / 2;                // staged voxels (324 | 340), per staging half-workgroup (162 | 170 <= 64 H2_SLOTS)
    static_assert(RY * RS == H2_PV && HALF * 2 == NV && HALF <= 64 * H2_SLOTS, "staged plane must stay 360 voxels");
};
__host__ __device__ inline bool h2_wide(int H, int W) { return ((W + 31) / 32) * ((H + 7) / 8) < ((W + 15) / 16) * ((H + 15) / 16); }

__device__ __forceinline__ void h2_split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// two values at once, in the form the ISA has instructions for: hi pair = ONE v_cvt_pk_f16_f32, each residual v - hi = ONE v_fma_mix_f32 (the fp16 half is widened
// inside the instruction: fma(hi, -1, v) rounds once, exactly like v - (float)hi), lo pair = ONE v_cvt_pk_f16_f32 -- 2 instead of 4 instructions per value; the
// same bits as h2_split.  Measured in round 4 (profiles/r04_h2_split_pair_ab.txt): in the z-Winograd experiment of that round (removed in round 5) -1 % time; in THIS file's kernel 0 ... +8 % (48^3 level: the
// mixlo / mixhi pair writes the two halves of one register, a serial dependency in the middle of the conversion piece) -- so conv3d_k3_h2_kernel keeps h2_split
// `m1` = -1.0f behind a value barrier (h2_minus_one): with the literal the optimiser rewrites fma(hi, -1, v) into a subtraction and the widening costs its own instruction
__device__ __forceinline__ float h2_minus_one() {
    float m = -1.0f;
    MH_OPAQUE(m);
    return m;
}
__device__ __forceinline__ void h2_split_pair(float v0, float v1, float m1, f16x2& hi, f16x2& lo) {
    hi = f16x2{(_Float16)v0, (_Float16)v1};
    const float r0 = __builtin_fmaf((float)hi[0], m1, v0), r1 = __builtin_fmaf((float)hi[1], m1, v1);
    lo = f16x2{(_Float16)r0, (_Float16)r1};
}

// EMIT / PLAIN step bodies are the same code; the epilogue of a completed plane rides in the first step of the next plane
// RES: at most two channel chunks (Cin <= 32): both weight slabs stay resident in the two LDS weight buffers (chunk = buffer index)
// and are loaded once -- the per-step weight stream from L2 (55 KB per step and CU: 2.2 of 9.4 ms) disappears
// timing experiment of tools/ubench/h2_variants.hip: -DH2X_XTRA=n adds n unused 16-byte LDS reads per tap, operands and results unchanged.  Measured
// (profiles/r03_h2_lds_sensitivity.txt): +12 % reads -> +0.8 % time, +37 % -> +8 %: the operand traffic (65 % of the LDS's 128 B/clk) is felt but is not the bound.
// (Round 3's first experiment replaced the low pieces' reads by copies of the high ones: same time -- but that also changed the matrix pipe's data.)
#ifdef H2X_XTRA
#define MH_H2X_EXTRA_READS(P) _Pragma("unroll") for (int x_ = 0; x_ < H2X_XTRA; ++x_) { u32x4 d_ = reinterpret_cast<const u32x4*>(P)[H2_WV + (x_ + 1) * 64]; asm volatile("" :: "v"(d_)); }
#else
#define MH_H2X_EXTRA_READS(P)
#endif
// C16 (round 4): output channel groups of 16 -- a layer with 16 couts (UNETR's full-resolution levels) would leave half of a 32-column matrix instruction to zero weights.
// The columns carry two z-taps instead: B = [W(kz 0) | W(kz 1)] (accumulator X) and [W(kz 2) | 0] (accumulator Y) -- 6 instead of 9 matrix instructions per (ky, kx) tap
// and 16 channels.  X of input plane p holds in its low 16 columns what output plane p + 1 gets from it and in its high 16 what plane p gets; Y's low 16 belong to plane
// p - 1: a completed plane is  X(p - 2).low + X(p - 1).high + Y(p).low  -- two register-set additions and one 16-lane exchange per plane.  Packed by conv3d_k3_h2c_pack_kernel.
// ACC (round 5; the 16-cout form since round 6: its Y set starts from the old values): the result is ADDED to what `out` holds -- a completed plane starts from the old values instead of from zero.  They are requested a
// whole plane ahead (four 16-byte buffer loads per lane right after the accumulator sets rotate) and enter the fresh set as old * 2^-(scale-back exponent), an exact
// power-of-two product, so the epilogue, its stores and the statistics are untouched: they see the sum.  Used by the UpCat path (kernels/upconv_h2.h writes the
// decoder's up half first, this kernel adds the skip half and leaves the InstanceNorm statistics of the sum).
// POOL (round 5, 16 x 16 regions of the 32-cout form): MaxPool3d(2) of the block that follows leaves the kernel with its result.  The pooled tensor cannot be the pooled
// ACTIVATED values (the InstanceNorm statistics of this very output are not known yet), but activation after normalisation is monotone in the raw value: increasing for
// alpha >= 0, decreasing for alpha < 0 -- so the kernel writes the 2 x 2 x 2 MAXIMUM and MINIMUM of the raw values (pmax, pmin: [N][Cout][D/2][H/2][W/2]) and the consumer
// reads the maxima under THIS tensor's records (mh_pool_select_f32 copies the minima over them for the channels whose alpha turned out negative).  In the epilogue: the x
// pairs of a lane's four-voxel groups are in-lane, the y pair (rows 2w, 2w + 1 of the wave) sits in the partner lane (lane ^ 32: the second row is rotated by 4
// voxels), the z pair is the previous plane's result held in eight registers.
template <bool STATS, bool NRM, bool RES, bool WIDE = false, bool C16 = false, bool ACC = false, bool POOL = false>
__global__ void __launch_bounds__(512, 1)
conv3d_k3_h2_kernel(Tensor in, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ bias, Tensor out,
                    float* __restrict__ stats, int bxn, int byn, int zchunk, unsigned nblk, float* __restrict__ pmax, float* __restrict__ pmin,
                    long long pool_n_stride) {
    using G = H2Geo<WIDE>;
    __shared__ uint4 smem[2 * (H2_XB + H2_WB)];
    // input records in LDS, per QUAD of channels {alpha x 4, beta x 4, slope x 4} (48 bytes): a wave converts one quad per step and reads its records with
    // three 16-byte broadcast reads, pairs of alphas / betas adjacent for packed arithmetic (12-byte records per channel cost 30 register moves and six
    // ds_read2_b64 with an immediate wait per step)
    __shared__ __attribute__((aligned(16))) float nrm_s[NRM ? 3 * H2_NRM_MAX : 4];
    uint4* const xs = smem;
    uint4* const ws = smem + 2 * H2_XB;
    unsigned* const bound_s = reinterpret_cast<unsigned*>(ws);      // 8 words of the (not yet loaded) weight buffer: all 160 KB of LDS are taken
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = in.C, Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long HW = (long long)H * W, DHW = (long long)D * HW;
    const int NCH = Cin / H2_KC;                              // steps per input plane (the launcher requires Cin % 16 == 0)

    // launch geometry of conv3d_wino2d.h: 1-D over (window, region, cout group), cout group fastest, XCD-aware
    constexpr int CG = C16 ? 16 : H2_CN;                               // output channels per workgroup
    const unsigned ncg = (unsigned)((Cout + CG - 1) / CG);            // round 4: the last group may hold fewer than 32 couts (Cout % 16 == 0: 48, 80, ...): its missing
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);                    // couts have zero weights in the packed slab, their stores and statistics are dropped
    const int cg = (int)(lid % ncg);
    lid /= ncg;
    const unsigned b = lid % nblk;
    const int n = (int)(lid / nblk);
    const int x0 = (int)(b % bxn) * G::BX, y0 = (int)((b / bxn) % byn) * G::BY;
    const int zs = (int)(b / (bxn * byn)) * zchunk, ze = min(zs + zchunk, D);
    const int p_first = max(zs - 1, 0), p_last = min(ze, D - 1);
    const int T = (p_last - p_first + 1) * NCH;               // steps this workgroup runs

    // input staging: wave w converts channels 4q .. 4q+3 (q = w >> 1) of the step for voxels (w & 1) * 162 + lane + 64 j.
    // Voxels outside the volume (zero padding) and idle lanes write into the unused pitch columns 18, 19 of their row: the padded
    // cells of both buffers are zeroed once and never written again, so the step body has no masks and no branches.
    const int q = wave >> 1;
    unsigned soff[H2_SLOTS];          // BYTE offsets into a channel plane
    int loff[H2_SLOTS];               // destination in units of 8 bytes inside a piece
#pragma unroll
    for (int j = 0; j < H2_SLOTS; ++j) {
        const int e0 = lane + 64 * j;
        const int e = min((wave & 1) * G::HALF + e0, G::NV - 1);
        const int ly = e / G::RX, lx = e - ly * G::RX;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        const bool ok = e0 < G::HALF && gy >= 0 && gy < H && gx >= 0 && gx < W;
        soff[j] = ok ? 4u * (unsigned)(gy * W + gx) : 0u;
        loff[j] = ((q >> 1) * H2_PV + ly * G::RS + (ok ? lx : G::RX + (lx & 1))) * 2 + (q & 1);
    }
    for (int i = tid; i < 2 * H2_XB; i += 512) xs[i] = make_uint4(0u, 0u, 0u, 0u);
    if (NRM) {
        unsigned mb = 0u;
        for (int c = tid; c < Cin; c += 512) {
            const float4 a = *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * c);
            float* r_ = nrm_s + 12 * (c >> 2) + (c & 3);
            r_[0] = a.x; r_[4] = a.y; r_[8] = a.z;
            const unsigned bb = abs_bits(a.w);
            mb = max(mb, bb == 0u ? 0x7fc00000u : bb);        // no bound given counts as non-finite
        }
        mb = wave_umax(mb);
        if (lane == 0) bound_s[wave] = mb;
    }
    __syncthreads();
    // input scale 2^e_in from the largest bound of the sample (bound < 2^eb  ->  bound * 2^(15 - eb) < 2^15); poisoned: NaN result
    int e_in = 0;
    bool poisoned = false;
    if (NRM) {
        unsigned mb = bound_s[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) mb = max(mb, bound_s[w]);
        poisoned = mb >= 0x7f800000u;
        e_in = poisoned ? 0 : min(max(15 - ((int)(mb >> 23) - 126), -100), 100);
        const float p_ = __uint_as_float((unsigned)(e_in + 127) << 23);
        // 2^e_in folded into the records (each thread rescales the ones it loaded): fma(x, alpha p, beta p) == p fma(x, alpha, beta) exactly -- a power of two
        // commutes with every rounding -- unless alpha p itself leaves fp32's normal range, which takes |alpha| < 1e-8 together with activations > 4e34
        // (or the mirror image): the step loop stays instruction for instruction the unscaled one
        for (int c = tid; c < Cin; c += 512) { float* r_ = nrm_s + 12 * (c >> 2) + (c & 3); r_[0] *= p_; r_[4] *= p_; }
        __syncthreads();
    }

    const float* src = in.data + (long long)n * in.n_stride + (long long)(4 * q) * DHW;
    const u32x4* const wg = reinterpret_cast<const u32x4*>(wp) + (long long)cg * NCH * H2_WB + tid;
    int is = 0, cs = 0;               // chunk of the next loads / of the next conversion
    const float* xptr = src + (long long)p_first * HW;
    int woff = 0;
    const long long xstep = (long long)H2_KC * DHW, xwrap = HW - (long long)(NCH - 1) * H2_KC * DHW;
    float xin[H2_SLOTS][4];
    u32x4 win[H2_WSLOTS];
    f32x4 nq_a = {1.0f, 1.0f, 1.0f, 1.0f}, nq_b = {0.0f, 0.0f, 0.0f, 0.0f}, nq_s = nq_a;      // records of the quad being converted

    // ---- the pieces of a step's staging work; each is branch-free so that it can be interleaved with the step's MFMAs ----
    // loads of the step after next (the registers were consumed by MH_H2_CONV / MH_H2_WST earlier in this step)
#define MH_H2_LDX                                                                                     \
    {       /* raw buffer loads: descriptor on the step's first channel plane (scalar registers), lane offset soff, channel offset as the scalar offset -- no vector address arithmetic */ \
        const auto xr_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xptr), 0, 0x7fffffff, 0x00020000); \
        _Pragma("unroll") for (int j = 0; j < H2_SLOTS; ++j)          /* slot-major: the first conversion piece waits for the first four loads only */ \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                             \
                xin[j][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr_, soff[j], (unsigned)(i * DHW * 4), 0)); \
    }
    // (PRO_: the prologue, where resident slabs are loaded too -- inside the loop the condition is a compile-time one: no branch in the step)
#define MH_H2_LDWX(PRO_, J0, J1)                                                                      \
    if (!RES || (PRO_)) _Pragma("unroll") for (int j = (J0); j < (J1); ++j) win[j] = wg[woff + 512 * j];
#define MH_H2_LDW(J0, J1) MH_H2_LDWX(false, J0, J1)
    // advance to the next (plane, chunk) -- not beyond the last step (the loads then repeat the last step's addresses)
#define MH_H2_ADV                                                                                     \
    {                                                                                                 \
        const bool adv_ = gi + 3 < T;                                                                 \
        const bool wrap_ = is + 1 == NCH;                                                             \
        xptr += adv_ ? (wrap_ ? xwrap : xstep) : 0LL;                                                 \
        woff = adv_ ? (wrap_ ? 0 : woff + H2_WB) : woff;                                              \
        is = adv_ ? (wrap_ ? 0 : is + 1) : is;                                                        \
    }
    // normalise + activate + split slot J on the way into LDS: 4 channels of a voxel -> 8 bytes of the high plane, 8 of the low one
#define MH_H2_NRMLD                                                                                   \
    if (NRM) {      /* the records of the quad of the NEXT conversion (cs was advanced by MH_H2_WST): alpha, beta in LDS are pre-multiplied by 2^e_in: act(x, alpha p, beta p, slope) == p act(x, alpha, beta, slope) */ \
        const f32x4* a_ = reinterpret_cast<const f32x4*>(nrm_s + 12 * (4 * cs + q));                  \
        nq_a = a_[0]; nq_b = a_[1]; nq_s = a_[2];                                                     \
    }
#define MH_H2_CONV(J)                                                                                 \
    {                                                                                                 \
        u32x2* xh_ = reinterpret_cast<u32x2*>(xs + (bcur ^ 1) * H2_XB);                               \
        _Float16 h_[4], l_[4];                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
            float y_ = xin[J][i];                                                                     \
            if (NRM) y_ = act(y_, nq_a[i], nq_b[i], nq_s[i]);                                         \
            h2_split(y_, h_[i], l_[i]);                                                               \
        }                                                                                             \
        const f16x2 h01_ = {h_[0], h_[1]}, h23_ = {h_[2], h_[3]}, l01_ = {l_[0], l_[1]}, l23_ = {l_[2], l_[3]}; \
        xh_[loff[J]] = u32x2{__builtin_bit_cast(unsigned, h01_), __builtin_bit_cast(unsigned, h23_)}; \
        xh_[loff[J] + 2 * H2_XV] = u32x2{__builtin_bit_cast(unsigned, l01_), __builtin_bit_cast(unsigned, l23_)}; \
    }
#define MH_H2_WSTX(PRO_)                                                                              \
    {                                                                                                 \
        if (!RES || (PRO_)) _Pragma("unroll") for (int j = 0; j < H2_WSLOTS; ++j) reinterpret_cast<u32x4*>(ws)[(bcur ^ 1) * H2_WB + tid + 512 * j] = win[j]; \
        cs = cs + 1 == NCH ? 0 : cs + 1;                                                              \
    }

#define MH_H2_WST MH_H2_WSTX(false)

    // operands of this lane: A = voxel (row 2w + (r >> 4), x) with r = lane & 31 and the second row rotated by 4 voxels
    // (16-byte reads of a lane group then cover all 64 banks once); B = cout r; k-group = lane >> 5
    const int r32 = lane & 31, kg = lane >> 5;
    // (WIDE: the wave's 32 voxels are one row -- consecutive 16-byte cells, conflict-free as they are)
    const int arow = r32 >> 4, ax = arow ? ((r32 + 12) & 15) : (r32 & 15);
    const int abase = WIDE ? kg * H2_PV + wave * G::RS + r32 : kg * H2_PV + (2 * wave + arow) * G::RS + ax;
    const int bbase = kg * H2_CN + r32;

    // acc[0], acc[1], acc[2]: output planes p+1, p, p-1 of the current input plane p (rotated once per plane, 48 moves);
    // acce: the completed plane waiting for its epilogue.  (C16: acc[0] = X, acc[1] = Y of the current plane, acc[2] = X of the plane before, accp = X of the one before that)
    f32x16 acc[3], acce, accp;
#pragma unroll
    for (int i = 0; i < 16; ++i) accp[i] = 0.0f;
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[s][i] = 0.0f;
    int pend = 0, pend_z = 0;

    // epilogue: lane = cout r32; register 4 j + i = voxel row (j >> 1) of the wave's two, x = xg(j) + i (WIDE: the wave's one row, x = 8 j + 4 kg + i)
    const int co = cg * CG + (C16 ? (r32 & 15) : r32);
    const bool cok = co < Cout && (!C16 || r32 < 16);                 // C16: a completed plane sits in the lanes of the low 16 columns
    const float bco = (bias && cok) ? bias[co] : 0.0f;
    // scale back: 2^-(weight scale exponent) * 2^-e_in as two power-of-two factors (their product may leave fp32's exponent range)
    float inv_a, inv_b;
    {
        const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - e_in;      // wtail[1] = the weights' power-of-two scale
        const int t1_ = t_ / 2, t2_ = t_ - t1_;
        inv_a = poisoned ? __uint_as_float(0x7fc00000u) : __uint_as_float((unsigned)(t1_ + 127) << 23);
        inv_b = __uint_as_float((unsigned)(t2_ + 127) << 23);
    }
    const int orow = WIDE ? wave : 2 * wave;                 // first region row of this wave
    // Result stores go through a raw buffer over the 32 cout planes of this (sample, cout group): 32-bit byte offsets, and a lane without a voxel
    // (ragged region, no completed plane yet) stores at an offset beyond the buffer, which the hardware drops -- no exec-mask branch, so the
    // epilogue stays inside the scheduling region of the matrix instructions (the launcher keeps 32 planes below 2 GB)
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out.data + (long long)n * out.n_stride + (long long)(cg * CG) * DHW, 0, (int)(min(CG, Cout - cg * CG) * DHW * 4), 0x00020000);
    constexpr unsigned H2_DROP = 0x80000000u;
    unsigned ooff[4];                                        // byte offset of register group j inside an output plane of cout r32 (or H2_DROP)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int xg_ = WIDE ? 8 * j + 4 * kg : j < 2 ? 8 * j + 4 * kg : ((8 * (j - 2) + 4 * kg + 12) & 15);
        const int yr_ = WIDE ? 0 : (j >> 1);
        const bool ok_ = cok && y0 + orow + yr_ < H && x0 + xg_ < W;
        ooff[j] = ok_ ? 4u * (unsigned)((long long)(C16 ? (r32 & 15) : r32) * DHW + (long long)(y0 + orow + yr_) * W + x0 + xg_) : H2_DROP;
    }
    f32x4 pv[4];                                             // ACC: the old values of the plane whose accumulator set starts next
    float pinv_a = 1.0f, pinv_b = 1.0f;                      // ACC: 2^(scale-back exponent), as two factors like inv_a, inv_b
    if (ACC) {
        const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - e_in;
        const int t1_ = t_ / 2, t2_ = t_ - t1_;
        pinv_a = __uint_as_float((unsigned)(127 - t1_) << 23);
        pinv_b = __uint_as_float((unsigned)(127 - t2_) << 23);
    }
#define MH_H2_PV_LOAD(Z)                                                                              \
    if (ACC) {      /* planes outside this workgroup's z-chunk belong to someone else (their sets are dropped): offset beyond the buffer -> zeros */ \
        const int z_ = (Z);                                                                           \
        const unsigned po_ = (z_ >= zs && z_ < ze) ? (unsigned)z_ * (unsigned)(HW * 4) : H2_DROP;     \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
            pv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(orsrc, (po_ == H2_DROP || ooff[j] == H2_DROP) ? H2_DROP : ooff[j] + po_, 0, 0)); \
    }
#define MH_H2_PV_INTO(S)                                                                              \
    if (ACC) {                                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[S][4 * j + i] = pv[j][i] * pinv_a * pinv_b; \
    }
    static_assert(!(POOL && (C16 || WIDE)), "the pooling epilogue exists for 16 x 16 regions of the 32-cout kernel");
    // POOL: lane kg 0 finishes the column groups x0 + 0..3 and x0 + 8..11 (its row-0 groups), lane kg 1 the groups x0 + 4..7 and x0 + 12..15: two pooled columns each,
    // at pooled column x0 / 2 + 4 j + 2 kg
    const long long PHW = (long long)(H / 2) * (W / 2), PDHW = (long long)(D / 2) * PHW;
    f32x2 hmx[2], hmn[2];                                    // the even plane's in-plane maxima / minima, waiting for the odd plane
    unsigned poff[2];
    if (POOL) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int px_ = x0 / 2 + 4 * j + 2 * kg, py_ = (y0 + orow) / 2;
            const bool ok_ = cok && y0 + orow + 1 < H && x0 + 8 * j + 4 * kg + 3 < W;
            poff[j] = ok_ ? 4u * (unsigned)((long long)r32 * PDHW + (long long)py_ * (W / 2) + px_) : H2_DROP;
            hmx[j] = f32x2{0.0f, 0.0f}; hmn[j] = f32x2{0.0f, 0.0f};
        }
    }
    const auto pxrs = __builtin_amdgcn_make_buffer_rsrc(POOL ? pmax + (long long)n * pool_n_stride + (long long)(cg * CG) * PDHW : out.data, 0, POOL ? (int)(min(CG, Cout - cg * CG) * PDHW * 4) : 0, 0x00020000);
    const auto pnrs = __builtin_amdgcn_make_buffer_rsrc(POOL ? pmin + (long long)n * pool_n_stride + (long long)(cg * CG) * PDHW : out.data, 0, POOL ? (int)(min(CG, Cout - cg * CG) * PDHW * 4) : 0, 0x00020000);
    f32x4 o_[4];                                             // the plane being emitted: its pieces A (scale, bias, store), B1-B3 (statistics) sit in different taps
    float esum_ = 0.0f, ecnt_ = 0.0f, em2_ = 0.0f, emean_ = 0.0f;
    Stat run;
    run.n = 0.0f; run.mean = 0.0f; run.m2 = 0.0f;

    uint4 ah[2], al[2], bh[2][3], bl[2][3];
#define MH_H2_FETCH(OB, T_)                                                                           \
    {                                                                                                 \
        constexpr int aoff_ = ((T_) / 3) * G::RS + (T_) % 3;                                          \
        const uint4* xb_ = xs + bcur * H2_XB + abase + aoff_;                                         \
        const uint4* wb_ = ws + bcur * H2_WB + (T_) * (2 * H2_CN) + bbase;                            \
        ah[OB] = xb_[0]; al[OB] = xb_[H2_XV];                                      \
        MH_H2X_EXTRA_READS(wb_)                                                                       \
        _Pragma("unroll") for (int kz = 0; kz < (C16 ? 2 : 3); ++kz) {                                \
            bh[OB][kz] = wb_[kz * (9 * 2 * H2_CN)]; bl[OB][kz] = wb_[H2_WV + kz * (9 * 2 * H2_CN)]; \
        }                                                                                             \
    }
#define MH_H2_MM(S, A, B) acc[S] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), acc[S], 0, 0, 0);
    // z-taps 0, 1, 2 of input plane p feed output planes p+1, p, p-1
#define MH_H2_MFMA9(OB)                                                                               \
    if (C16) {      /* columns [kz 0 | kz 1] -> X, [kz 2 | 0] -> Y */                                 \
        MH_H2_MM(0, ah[OB], bh[OB][0]) MH_H2_MM(1, ah[OB], bh[OB][1])                                 \
        MH_H2_MM(0, al[OB], bh[OB][0]) MH_H2_MM(1, al[OB], bh[OB][1])                                 \
        MH_H2_MM(0, ah[OB], bl[OB][0]) MH_H2_MM(1, ah[OB], bl[OB][1])                                 \
    } else {                                                                                          \
        MH_H2_MM(0, ah[OB], bh[OB][0]) MH_H2_MM(1, ah[OB], bh[OB][1]) MH_H2_MM(2, ah[OB], bh[OB][2])  \
        MH_H2_MM(0, al[OB], bh[OB][0]) MH_H2_MM(1, al[OB], bh[OB][1]) MH_H2_MM(2, al[OB], bh[OB][2])  \
        MH_H2_MM(0, ah[OB], bl[OB][0]) MH_H2_MM(1, ah[OB], bl[OB][1]) MH_H2_MM(2, ah[OB], bl[OB][2])  \
    }
    // one (ky, kx) group: the next group's operand reads, 9 MFMAs and a piece FILL of the staging work.  A wave issues in order
    // and an MFMA occupies the matrix pipe for 32 cycles, so whatever is placed BETWEEN two MFMAs is free, and whatever sits in a
    // lump before or after them is serial time of this wave (measured with the lump form: 6700 of 9700 cycles per step): the
    // scheduler is told to deal the piece out over the gaps.
#define MH_H2_TAPV(T_, NV_, ...)                                                                      \
    {                                                                                                 \
        if ((T_) + 1 < 9) MH_H2_FETCH(((T_) + 1) & 1, ((T_) + 1 < 9 ? (T_) + 1 : 0))                  \
        __VA_ARGS__                                                                                   \
        MH_H2_MFMA9((T_) & 1)                                                                         \
        _Pragma("unroll") for (int g_ = 0; g_ < (C16 ? 6 : 9); ++g_) {      /* C16: the same staging work in two thirds of the gaps */ \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x100, C16 ? 2 : 1, 0);                              \
            __builtin_amdgcn_sched_group_barrier(0x006, C16 ? ((NV_) * 3 + 1) / 2 : (NV_), 0);        \
            __builtin_amdgcn_sched_group_barrier(0x230, C16 ? 2 : 1, 0);                              \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#ifndef H2X_NV
#define H2X_NV 5      // vector instructions the scheduler may place per matrix-instruction gap of a plain tap (tools/ubench/h2_variants.hip: -DH2X_NV=n)
#endif
#define MH_H2_TAP(T_, ...) MH_H2_TAPV(T_, H2X_NV, __VA_ARGS__)
    // the completed output plane (in acce), in four branch-free pieces that ride in the gaps of different taps of the next plane's first step:
    // A scale back, bias, 4 x 16-byte buffer stores per lane (the plane offset goes into the vector offset, NOT into the instruction's scalar offset: a
    // 16-byte buffer store with a scalar-register offset followed at once by a vector write of its data registers stored the NEW value of the second
    // dword on the MI355X -- the compiler only inserts the wait state the ISA asks for when the scalar offset is not a register; placed BEFORE the loads of the step: stores and loads share the vmcnt counter, and a
    // store issued after the loads would make the next conversion wait for the store acknowledgement as well); B1 count / sum / mean, B2 the
    // squared deviations, B3 the merge into the running statistics.  (As one piece with `if (ok) store` and stat_merge's early return the
    // epilogue was five basic blocks of ~200 vector instructions with no matrix instruction among them: 1.5 of 9.4 ms at 32 -> 32 channels.)
#define MH_H2_EMIT_A                                                                                  \
    {                                                                                                 \
        const unsigned so_ = (unsigned)pend_z * (unsigned)(HW * 4);                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            o_[j] = f32x4{acce[4 * j], acce[4 * j + 1], acce[4 * j + 2], acce[4 * j + 3]} * inv_a * inv_b + bco; \
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o_[j]), orsrc, (pend ? ooff[j] : H2_DROP) + so_, 0, 0); \
        }                                                                                             \
    }
#define MH_H2_EMIT_B1                                                                                 \
    if (STATS) {                                                                                      \
        const float pf_ = pend ? 1.0f : 0.0f;                                                         \
        esum_ = 0.0f; ecnt_ = 0.0f;                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            const float w_ = ooff[j] != H2_DROP ? pf_ : 0.0f;                                         \
            ecnt_ += 4.0f * w_;                                                                       \
            esum_ += ((o_[j][0] + o_[j][1]) + (o_[j][2] + o_[j][3])) * w_;                            \
        }                                                                                             \
        emean_ = ecnt_ > 0.0f ? esum_ / (ecnt_ > 0.0f ? ecnt_ : 1.0f) : 0.0f;                         \
    }
#define MH_H2_EMIT_B2                                                                                 \
    if (STATS) {                                                                                      \
        em2_ = 0.0f;                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            const f32x4 d_ = o_[j] - emean_;                                                          \
            const f32x4 q_ = d_ * d_;                                                                 \
            em2_ += ((q_[0] + q_[1]) + (q_[2] + q_[3])) * (ooff[j] != H2_DROP && pend ? 1.0f : 0.0f); \
        }                                                                                             \
    }
    // POOL piece: in-lane x pairs, the partner lane's row-1 groups (lane ^ 32), the plane pair; stores on the odd plane of a pair only (offset beyond the buffer otherwise)
#define MH_H2_EMIT_P                                                                                  \
    if (POOL) {                                                                                       \
        f32x2 xm_[4], xn_[4];                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            xm_[j] = f32x2{fmaxf(o_[j][0], o_[j][1]), fmaxf(o_[j][2], o_[j][3])};                     \
            xn_[j] = f32x2{fminf(o_[j][0], o_[j][1]), fminf(o_[j][2], o_[j][3])};                     \
        }                                                                                             \
        f32x2 rm_[2], rn_[2];       /* the partner's row-1 groups: its j = 2, 3 */                    \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                           \
                rm_[j][i] = __shfl_xor(xm_[2 + j][i], 32);                                            \
                rn_[j][i] = __shfl_xor(xn_[2 + j][i], 32);                                            \
            }                                                                                         \
        const bool odd_ = (pend_z & 1) != 0;                                                          \
        const unsigned pso_ = (unsigned)(pend_z >> 1) * (unsigned)(PHW * 4);                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                               \
            /* kg 0: own group j pairs with the partner's j-th row-1 group; kg 1: with the other one */ \
            const f32x2 pm_ = kg == 0 ? rm_[j] : rm_[1 - j], pn_ = kg == 0 ? rn_[j] : rn_[1 - j];     \
            f32x2 ym_ = f32x2{fmaxf(xm_[j][0], pm_[0]), fmaxf(xm_[j][1], pm_[1])};                    \
            f32x2 yn_ = f32x2{fminf(xn_[j][0], pn_[0]), fminf(xn_[j][1], pn_[1])};                    \
            const f32x2 fm_ = f32x2{fmaxf(ym_[0], hmx[j][0]), fmaxf(ym_[1], hmx[j][1])};              \
            const f32x2 fn_ = f32x2{fminf(yn_[0], hmn[j][0]), fminf(yn_[1], hmn[j][1])};              \
            const unsigned po_ = (pend && odd_ && poff[j] != H2_DROP) ? poff[j] + pso_ : H2_DROP;     \
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, fm_), pxrs, po_, 0, 0);   \
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, fn_), pnrs, po_, 0, 0);   \
            hmx[j] = ym_; hmn[j] = yn_;     /* an odd plane's values are overwritten by the next even plane before they are used */ \
        }                                                                                             \
    }
#define MH_H2_EMIT_B3                                                                                 \
    {                                                                                                 \
        if (STATS) {                                                                                  \
            Stat loc_;                                                                                \
            loc_.n = ecnt_; loc_.mean = emean_; loc_.m2 = em2_;                                       \
            run = stat_merge_nb(run, loc_);                                                           \
        }                                                                                             \
        pend = 0;                                                                                     \
    }
#define MH_H2_EMIT { MH_H2_EMIT_A MH_H2_EMIT_P MH_H2_EMIT_B1 MH_H2_EMIT_B2 MH_H2_EMIT_B3 }
#define MH_H2_NONE
    // a plain step, and the first step of a plane (the previous plane's epilogue pieces in taps 3, 4, 6, 7 with more vector slots per gap)
#define MH_H2_SCHEDULE_PLAIN                                                                          \
        MH_H2_TAP(0, MH_H2_CONV(0)) MH_H2_TAP(1, MH_H2_CONV(1)) MH_H2_TAP(2, MH_H2_CONV(2))           \
        MH_H2_TAP(3, MH_H2_NONE)                                                                      \
        MH_H2_TAP(4, MH_H2_LDX) MH_H2_TAP(5, MH_H2_WST) MH_H2_TAP(6, MH_H2_LDW(0, 4))                 \
        MH_H2_TAP(7, MH_H2_LDW(4, H2_WSLOTS) MH_H2_ADV) MH_H2_TAP(8, MH_H2_NRMLD)
#define MH_H2_SCHEDULE_EMIT                                                                           \
        MH_H2_TAP(0, MH_H2_CONV(0)) MH_H2_TAP(1, MH_H2_CONV(1)) MH_H2_TAP(2, MH_H2_CONV(2))           \
        MH_H2_TAPV(3, 8, MH_H2_EMIT_A)                                                                \
        MH_H2_TAPV(4, 8, MH_H2_LDX MH_H2_EMIT_B1) MH_H2_TAPV(5, POOL ? 10 : 5, MH_H2_WST MH_H2_EMIT_P) MH_H2_TAPV(6, 8, MH_H2_LDW(0, 4) MH_H2_EMIT_B2) \
        MH_H2_TAPV(7, 8, MH_H2_LDW(4, H2_WSLOTS) MH_H2_ADV MH_H2_EMIT_B3) MH_H2_TAP(8, MH_H2_NRMLD)
    // one step (16 channels of input plane p): conversion of the next step's registers into the other LDS buffer first, then the
    // epilogue stores, then the loads of the step after next
#define MH_H2_STEP(SCHED_)                                                                            \
    {                                                                                                 \
        MH_H2_FETCH(0, 0)                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        SCHED_                                                                                        \
    }

    // prologue: step 0 into buffer 0 (the conversion pieces write the buffer "after" bcur: start from 1), the loads of step 1 in flight
    int bcur = 1, gi = -2;
    MH_H2_LDX MH_H2_LDWX(true, 0, H2_WSLOTS) MH_H2_ADV
    gi = -1;
    MH_H2_NRMLD MH_H2_CONV(0) MH_H2_CONV(1) MH_H2_CONV(2) MH_H2_WSTX(true)
    MH_H2_NRMLD                       // the records of the quad the first step converts (every step reads the next step's in its last tap: no LDS round trip in front of the conversion)
    MH_H2_LDX MH_H2_LDWX(true, 0, H2_WSLOTS) MH_H2_ADV
    if (RES) {        // the second slab (chunk 1, or chunk 0 again when there is one chunk) goes into buffer 1 now and stays
        _Pragma("unroll") for (int j = 0; j < H2_WSLOTS; ++j) reinterpret_cast<u32x4*>(ws)[H2_WB + tid + 512 * j] = win[j];
    }
    bcur = 0; gi = 0;
    __syncthreads();

#ifdef H2X_SETPRIO      // experiment (tools/ubench/h2_variants.hip): static priority for the second-dispatched half of the workgroup (MI355X_MICROARCH.md, two waves per SIMD, item 4)
    if (wave >= 4) __builtin_amdgcn_s_setprio(H2X_SETPRIO);
#endif
    if (C16) {
        MH_H2_PV_LOAD(zs - 1)         // ACC, C16: the Y set of input plane p + 1 belongs to output plane p and starts from its old values (the X sets start from zero); the first
                                      // one that counts is Y(zs + 1): the rotation after iteration p takes the values requested here / there a plane earlier
    } else {
        MH_H2_PV_LOAD(zs)             // ACC: set 0 of the first iteration (input plane zs - 1) belongs to output plane zs
        MH_H2_PV_INTO(0)
        MH_H2_PV_LOAD(zs + 1)
    }
    for (int p = zs - 1; p <= ze; ++p) {
        if (p >= p_first && p <= p_last) {
            MH_H2_STEP(MH_H2_SCHEDULE_EMIT)
            __syncthreads();
            bcur ^= 1; ++gi;
            for (int s = 1; s < NCH; ++s) {
                MH_H2_STEP(MH_H2_SCHEDULE_PLAIN)
                __syncthreads();
                bcur ^= 1; ++gi;
            }
        }
        if (p - 1 >= zs) {              // output plane p-1 is complete in set 2 (planes in front of the chunk are simply dropped)
            if (pend) MH_H2_EMIT        // only when no step ran since the previous plane (p == D)
            if (C16) {                  // X(p - 2).low + X(p - 1).high + Y(p).low, in the lanes of the low 16 columns
#pragma unroll
                for (int i = 0; i < 16; ++i) acce[i] = (accp[i] + __shfl_xor(acc[2][i], 16)) + acc[1][i];
            } else {
                acce = acc[2];
            }
            pend = 1; pend_z = p - 1;
        }
        if (C16) {
            accp = acc[2];
            acc[2] = acc[0];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0][i] = 0.0f;
            if (ACC) {                  // the fresh Y set (input plane p + 1) belongs to output plane p: its old values (lanes of the low 16 columns; zeros elsewhere), then request plane p + 1's
                MH_H2_PV_INTO(1)
                MH_H2_PV_LOAD(p + 1)
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[1][i] = 0.0f;
            }
        } else {
            acc[2] = acc[1];
            acc[1] = acc[0];
            if (ACC) {                  // the fresh set belongs to output plane p + 2: its old values (requested a plane ago); request plane p + 3's
                MH_H2_PV_INTO(0)
                MH_H2_PV_LOAD(p + 3)
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0][i] = 0.0f;
            }
        }
    }
    if (pend) MH_H2_EMIT
#undef MH_H2_STEP
#undef MH_H2_SCHEDULE_EMIT
#undef MH_H2_SCHEDULE_PLAIN
#undef MH_H2_NONE
#undef MH_H2_EMIT
#undef MH_H2_EMIT_B3
#undef MH_H2_EMIT_P
#undef MH_H2_EMIT_B2
#undef MH_H2_EMIT_B1
#undef MH_H2_EMIT_A
#undef MH_H2_TAP
#undef MH_H2_TAPV
#undef MH_H2_MFMA9
#undef MH_H2_MM
#undef MH_H2_FETCH
#undef MH_H2_WST
#undef MH_H2_WSTX
#undef MH_H2_CONV
#undef MH_H2_NRMLD
#undef MH_H2_ADV
#undef MH_H2_LDW
#undef MH_H2_LDWX
#undef MH_H2_LDX
#undef MH_H2_PV_INTO
#undef MH_H2_PV_LOAD

    if (STATS) {
        // the two k-group halves of a lane pair hold disjoint voxels of the same cout; then the eight waves merge through LDS
        {
            Stat ot;
            ot.n = __shfl_xor(run.n, 32);
            ot.mean = __shfl_xor(run.mean, 32);
            ot.m2 = __shfl_xor(run.m2, 32);
            run = kg == 0 ? stat_merge(run, ot) : stat_merge(ot, run);
        }
        __syncthreads();     // the staging buffers are free
        float* red = reinterpret_cast<float*>(smem);
        if (kg == 0) {
            red[(wave * H2_CN + r32) * 3] = run.n; red[(wave * H2_CN + r32) * 3 + 1] = run.mean; red[(wave * H2_CN + r32) * 3 + 2] = run.m2;
        }
        __syncthreads();
        if (tid < CG && cg * CG + tid < Cout) {
            Stat st;
            st.n = 0.0f; st.mean = 0.0f; st.m2 = 0.0f;
#pragma unroll
            for (int w = 0; w < 8; ++w) {
                Stat ot;
                ot.n = red[(w * H2_CN + tid) * 3]; ot.mean = red[(w * H2_CN + tid) * 3 + 1]; ot.m2 = red[(w * H2_CN + tid) * 3 + 2];
                st = stat_merge(st, ot);
            }
            float* rec = stats + (((long long)n * Cout + cg * CG + tid) * nblk + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}

// Weight preparation, two launches.  (1) one workgroup: max |w| -> tail = {1 / S, S}, S the power of two that puts the largest
// weight in [2^12, 2^13) (fp16 high pieces far from overflow, low pieces normal down to 2^-17 of the largest weight).
__global__ void __launch_bounds__(1024)
conv3d_k3_h2_scale_kernel(const float* __restrict__ w, long long count, float* __restrict__ tail) {
    __shared__ float red[1024];
    float m = 0.0f;
    for (long long i = threadIdx.x; i < count; i += 1024) {
        const float a = fabsf(w[i]);
        if (a < 3.0e38f) m = fmaxf(m, a);                   // finite values only
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int e = 0;
        float s = 1.0f;
        if (red[0] > 0.0f) {
            frexpf(red[0], &e);                             // red[0] = f * 2^e, f in [0.5, 1)
            e = 13 - e;
            e = e > 100 ? 100 : e < -100 ? -100 : e;
            s = ldexpf(1.0f, e);
        }
        tail[0] = 1.0f / s; tail[1] = s; tail[2] = 0.0f; tail[3] = 0.0f;
    }
}
// (2) w [Cout][Cin][3][3][3] -> [cout group][chunk][piece][tap][k-group][32 couts][8 channels] fp16, each chunk slab padded to
// H2_WB uint4 (the pad is never used as an operand; the buffer is zeroed first).  One thread per (cout, cin).
__global__ void __launch_bounds__(256)
conv3d_k3_h2_pack_kernel(const float* __restrict__ w, int Cin, int Cout, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const int nchunk = Cin / H2_KC;
    const float s = tail[1];
    _Float16* slab = packed + ((long long)(co / H2_CN) * nchunk + ci / H2_KC) * (H2_WB * 8LL);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        _Float16 pc[2];
        h2_split(w[((long long)co * Cin + ci) * 27 + tap] * s, pc[0], pc[1]);
#pragma unroll
        for (int p = 0; p < 2; ++p)
            slab[(((p * 27 + tap) * 2 + (ci % H2_KC) / 8) * H2_CN + (co % H2_CN)) * 8 + (ci % 8)] = pc[p];
    }
}

// The C16 form: w [Cout][Cin][3][3][3] -> [cout group of 16][chunk][piece][slot v * 9 + (ky, kx)][k-group][32 columns][8 channels]: v = 0 columns = [kz 0 | kz 1] of the
// group's 16 couts, v = 1 columns = [kz 2 | zeros]; the slots of v = 2 stay zero (the buffer is zeroed first) and are never multiplied.
__global__ void __launch_bounds__(256)
conv3d_k3_h2c_pack_kernel(const float* __restrict__ w, int Cin, int Cout, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const int nchunk = Cin / H2_KC;
    const float s = tail[1];
    _Float16* slab = packed + ((long long)(co / 16) * nchunk + ci / H2_KC) * (H2_WB * 8LL);
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int kz = tap / 9, t9 = tap % 9;
        const int slot = (kz == 2 ? 9 : 0) + t9, col = (kz == 1 ? 16 : 0) + (co % 16);
        _Float16 pc[2];
        h2_split(w[((long long)co * Cin + ci) * 27 + tap] * s, pc[0], pc[1]);
#pragma unroll
        for (int p = 0; p < 2; ++p)
            slab[(((p * 27 + slot) * 2 + (ci % H2_KC) / 8) * H2_CN + col) * 8 + (ci % 8)] = pc[p];
    }
}

}  // namespace mh
