// Conv3d 3x3x3 / stride 2 / zero padding 1 on the FP16 matrix cores of gfx950 in conv3d_h2.h's two-piece split precision (round 5).
//
// Reference op: the down-sampling convolutions of DynUNet (monai/networks/blocks/dynunet_block.py:135-186 -> get_conv_layer(stride=2)), SegResNet
// (monai/networks/nets/segresnet.py:111-133) and UNet (monai/networks/nets/unet.py:197-237 -> Convolution(strides=2)), fed by the previous block's deferred
// InstanceNorm + LeakyReLU.  Until round 5 they ran on the vector ALU (conv3d_k3_strided_kernel: 41 TFLOP/s on DynUNet's 32 -> 64 @ 96^3, 30 % of that network's step).
//
// Decomposition.  out[o] = sum_k w[k] x[2 o + k - 1] per axis: tap k = 1 reads EVEN input index o, taps k = 0 / 2 read ODD input indices o - 1 / o.  Split the
// input into its 8 parity phases (pz, py, px) -- each a volume of the OUTPUT's extents -- and the strided convolution is a sum of 8 stride-1 convolutions with
// 1 / 2 / 4 / 8 taps (27 in all) whose operand reads are dense.  No zero taps are multiplied (the "space to depth, then a 2^3 kernel over 8 Cin channels" form
// would multiply 64 taps, 37 of them zero).  Only index -1 of an odd phase is padding: the upper end (2 o + 1 <= extent - 1) is always inside an even extent.
//
// Two kernels.  (1) conv3d_s2_split_kernel, an HBM pass: activate (records), scale by the sample's power of two (conv3d_h2.h's bound contract), split into hi + lo
// fp16 pieces and store phase-major in the matrix instruction's operand layout [chunk of 16 ch][phase][piece][k-group][z'][y'][x'][8 ch]: 4 B read + 4 B
// written per element, once -- not once per output channel group.  (2) conv3d_k3s2_h2_kernel, the GEMM: M = output voxels, N = 32 NCG output channels per
// workgroup, K = 16 channels per instruction.  A workgroup owns a TR x TC tile of output rows x columns (TR TC <= 256, chosen by the launcher per plane
// shape: 16 x 16, 10 x 24, 21 x 12 ...: a flattened M index, so 24- and 12-wide planes do not pad to 32 / 16) and marches along z'.  A STEP = (phase, chunk):
// its operand region (TR + 1) x (TC + 1) cells of 16 bytes x [piece][k-group] (<= 19 KB) and its tap matrices (1 .. 8 x 4 KB NCG) are copied global -> registers ->
// LDS a step ahead into the other of two buffers (pure copies: no vector ALU work in the loop), one barrier per step.  Odd-z phases feed TWO output planes (k = 2 of
// plane z', k = 0 of plane z' + 1) from one read of their operands: two accumulator sets that rotate per plane, so every phase plane is staged once.
// Epilogue as in upconv_h2.h: scale back (exact powers of two), bias, stores, InstanceNorm statistics {count, mean, M2} per (n, cout, workgroup).
#pragma once
#include "common.h"
#include "conv3d_h2.h"

namespace mh {

constexpr int S2_NT = 512;                                 // 8 waves, one 32-voxel M block each
constexpr int S2_RVMAX = 304;                              // cells of a staged phase region: (TR + 1) (TC + 1)
constexpr int S2_XSLOTS = 3;                               // copies per thread and step: 3 x 512 >= 4 x 304
constexpr int S2_XB = S2_NT * S2_XSLOTS;                   // uint4 per staged region: [piece][k group][cell] = 4 (TR + 1)(TC + 1) <= 1216, rounded up to one cell per copy slot (no bound check in the copy)
constexpr unsigned S2_DROP = 0x80000000u;                  // a byte offset beyond every buffer: loads return zero, stores are dropped
constexpr int S2_POISON = 0x7fffffff;                      // exponent slot of a sample whose bound is non-finite / missing: its whole output is NaN

__host__ __device__ inline int s2_ntaps(int ph) { return (1 + (ph >> 2)) * (1 + ((ph >> 1) & 1)) * (1 + (ph & 1)); }
__host__ __device__ inline int s2_tap_offset(int ph) {     // prefix sums of s2_ntaps {0, 1, 3, 5, 9, 11, 15, 19}: phases in the order pz * 4 + py * 2 + px; 5-bit fields of one constant (no branch)
    constexpr unsigned long long tab = 0ull | (1ull << 5) | (3ull << 10) | (5ull << 15) | (9ull << 20) | (11ull << 25) | (15ull << 30) | (19ull << 35);
    return (int)((tab >> (5 * ph)) & 31ull);
}

// output tile of a workgroup for an Ho x Wo plane: TC a multiple of 4 (a lane's four consecutive M indices stay in one row), TR TC <= 256, (TR + 1)(TC + 1) <= S2_RVMAX;
// the candidate with the fewest tiles wins (ties: the smaller staged region, then the wider tile).  A pure function of the extents: the statistics record count depends on it.
struct S2Tile { int tr, tc, tyn, txn; };
__host__ __device__ inline S2Tile s2_tile(int Ho, int Wo) {
    const int cand[7] = {8, 12, 16, 24, 32, 48, 60};
    S2Tile best = {0, 0, 0, 0};
    long long bn = -1;
    int brv = 0;
    for (int i = 0; i < 7; ++i) {
        const int tc = cand[i];
        int tr = 256 / tc;
        while ((tr + 1) * (tc + 1) > S2_RVMAX) --tr;
        if (tr > Ho) tr = Ho;
        const int txn = (Wo + tc - 1) / tc, tyn = (Ho + tr - 1) / tr, rv = (tr + 1) * (tc + 1);
        const long long nt = (long long)txn * tyn;
        if (bn < 0 || nt < bn || (nt == bn && rv <= brv)) { bn = nt; brv = rv; best.tr = tr; best.tc = tc; best.tyn = tyn; best.txn = txn; }
    }
    return best;
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// (1) activate + scale + split + phase-major store.  grid (ceil(H W / 256), D, N Cin / 8); thread = one fine (y, x) of plane z, 8 channels (one k group).
// ws [n][chunk][phase][piece][k group][Do][Ho][Wo] uint4; expo [N]: the sample's input exponent e_in (or S2_POISON), written by the sample's first block.
__global__ void __launch_bounds__(256)
conv3d_s2_split_kernel(Tensor in, uint4* __restrict__ ws, int* __restrict__ expo) {
    __shared__ unsigned bound_s[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int Cin = in.C, D = in.D, H = in.H, W = in.W, Do = D / 2, Ho = H / 2, Wo = W / 2;
    const int nc8 = Cin / 8;
    const int n = (int)blockIdx.z / nc8, c8 = (int)blockIdx.z % nc8;
    const int z = (int)blockIdx.y;
    unsigned mb = 0u;
    for (int c = tid; c < Cin; c += 256) {
        const float4 a = load_nrm(in, n, c);
        const unsigned bb = abs_bits(a.w);
        mb = max(mb, in.nrm == nullptr ? abs_bits(1.0f) : (bb == 0u ? 0x7fc00000u : bb));      // no bound given counts as non-finite (conv3d_h2.h)
    }
    mb = wave_umax(mb);
    if (lane == 0) bound_s[tid >> 6] = mb;
    __syncthreads();
    mb = max(max(bound_s[0], bound_s[1]), max(bound_s[2], bound_s[3]));
    const bool poisoned = mb >= 0x7f800000u;
    const int e_in = (poisoned || in.nrm == nullptr) ? 0 : min(max(15 - ((int)(mb >> 23) - 126), -100), 100);
    if (blockIdx.x == 0 && z == 0 && c8 == 0 && tid == 0) expo[n] = poisoned ? S2_POISON : e_in;
    const long long HW = (long long)H * W, DHW = (long long)D * HW;
    const long long idx = (long long)blockIdx.x * 256 + tid;
    if (idx >= HW) return;
    const int y = (int)(idx / W), x = (int)(idx - (long long)y * W);
    const float p_ = __uint_as_float((unsigned)(e_in + 127) << 23);
    const float* src = in.data + (long long)n * in.n_stride + (long long)(8 * c8) * DHW + (long long)z * HW + idx;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[(long long)i * DHW];
    _Float16 h_[8], l_[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 a = load_nrm(in, n, 8 * c8 + i);
        h2_split(act(v[i], a.x * p_, a.y * p_, a.z), h_[i], l_[i]);      // fma(x, alpha p, beta p) == p fma(x, alpha, beta): a power of two commutes with the rounding
    }
    const f16x2 h01 = {h_[0], h_[1]}, h23 = {h_[2], h_[3]}, h45 = {h_[4], h_[5]}, h67 = {h_[6], h_[7]};
    const f16x2 l01 = {l_[0], l_[1]}, l23 = {l_[2], l_[3]}, l45 = {l_[4], l_[5]}, l67 = {l_[6], l_[7]};
    const long long ovol = (long long)Do * Ho * Wo;
    const int chunk = c8 >> 1, kg = c8 & 1, ph = (z & 1) * 4 + (y & 1) * 2 + (x & 1);
    uint4* dst = ws + ((long long)n * (Cin / 16) * 8 + (long long)chunk * 8 + ph) * 4 * ovol + (long long)kg * ovol +
                 ((long long)(z >> 1) * Ho + (y >> 1)) * Wo + (x >> 1);
    dst[0] = make_uint4(__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23), __builtin_bit_cast(unsigned, h45), __builtin_bit_cast(unsigned, h67));
    dst[2 * ovol] = make_uint4(__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23), __builtin_bit_cast(unsigned, l45), __builtin_bit_cast(unsigned, l67));
}

// ---------------------------------------------------------------------------------------------------------------------------------------------------------
// (2) the matrix instructions of one step: NY x NX in-plane taps of the staged phase region, NZ z-taps each (NZ == 2: slot 0 = k 0 -> the NEXT output plane's
// accumulators, slot 1 = k 2 -> this plane's), NCG groups of 32 output channels.  A's two pieces are read once per in-plane tap and meet 2 NZ NCG weight operands.
struct S2NoSide { __device__ __forceinline__ void operator()(int, int) const {} };
template <int NZ, int NY, int NX, int NCG, bool HEAVY, class SIDE>
__device__ __forceinline__ void s2_step_mm(const uint4* __restrict__ xb, const uint4* __restrict__ wb, int abase, int bbase, int RS, int RV, f32x16 (&acc)[2][NCG], SIDE side) {
    constexpr int CW = 32 * NCG, NG = NZ * NY * NX;           // a GROUP = one (in-plane tap, z tap): 3 NCG matrix instructions
    // a wave issues in order: an operand read right in front of its use costs the LDS latency every time (conv3d_h2.h).  Group G + 1's operands are therefore
    // fetched into a second register set while group G multiplies, and the scheduler deals the reads out over the gaps between the matrix instructions.
    uint4 aq[2][2];                                          // [set][piece]: the in-plane tap's A operand (shared by its NZ groups)
    uint4 bq[2][NCG][2];                                     // [set][cout group][piece]
    auto fetch = [&](int G) {
        const int t2 = G / NZ, sz = G % NZ, sy = t2 / NX, sx = t2 % NX;
        if (sz == 0) {
            const int dy = NY == 1 ? 1 : sy, dx = NX == 1 ? 1 : sx;      // staged origin = tile origin - 1: even phase o -> o + 1, odd phase k 0 -> o, k 2 -> o + 1
            const uint4* ap = xb + abase + dy * RS + dx;
            aq[t2 & 1][0] = ap[0];
            aq[t2 & 1][1] = ap[2 * RV];
        }
        const int tis = (sz * NY + sy) * NX + sx;
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            const uint4* bp = wb + (tis * 4) * CW + bbase + g * 32;
            bq[G & 1][g][0] = bp[0];
            bq[G & 1][g][1] = bp[2 * CW];
        }
    };
    fetch(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int G = 0; G < NG; ++G) {
        if (G + 1 < NG) fetch(G + 1);
        side(G, NG);                                          // the fused form's staging work of the NEXT step (vector ALU, LDS writes, loads): it rides between this group's matrix instructions
        const int t2 = G / NZ, sz = G % NZ;
        const int zset = NZ == 1 ? 0 : (sz == 0 ? 1 : 0);     // NZ == 2: slot 0 = k 0 -> the NEXT output plane's accumulators, slot 1 = k 2 -> this plane's
        const f16x8 ah = __builtin_bit_cast(f16x8, aq[t2 & 1][0]), al = __builtin_bit_cast(f16x8, aq[t2 & 1][1]);
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            const f16x8 bh = __builtin_bit_cast(f16x8, bq[G & 1][g][0]), bl = __builtin_bit_cast(f16x8, bq[G & 1][g][1]);
            acc[zset][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[zset][g], 0, 0, 0);
            acc[zset][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[zset][g], 0, 0, 0);
            acc[zset][g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[zset][g], 0, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 3 * NCG; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NCG == 1 ? 2 : 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, HEAVY ? (NCG == 1 ? 16 : 8) : 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// FUSED == false: `in` gives the extents only; xs = the split kernel's workspace, expo its exponents.  FUSED == true (round 5, layers whose split pass costs as much as
// their GEMM): there is no split pass -- the staging reads the fp32 input itself (a (cell, channel quad) task = four dword loads of the phase's fine voxel), activates
// under the records (LDS, pre-multiplied by the sample's power of two), splits and writes the two pieces of the NEXT step's region while THIS step's matrix
// instructions run (s2_step_mm's side work); the loads of the step after next are issued into the same registers right behind the conversion, so they have more than a
// whole step to arrive.  xs / expo are unused.
// wp [cout group][chunk][27 taps in phase order][piece][k group][32 NCG couts] uint4 + tail (conv3d_k3s2_h2_pack_kernel).
constexpr int S2_CIN_MAX = 512;                            // FUSED: input channels whose records sit in LDS
template <int NCG, bool STATS, bool FUSED>
__global__ void __launch_bounds__(S2_NT, 1)
conv3d_k3s2_h2_kernel(Tensor in, const uint4* __restrict__ xs, const int* __restrict__ expo, const uint4* __restrict__ wp, const float* __restrict__ wtail,
                      const float* __restrict__ bias, Tensor out, float* __restrict__ stats, int TR, int TC, int txn, int tyn, int zchunk, unsigned nblk) {
    constexpr int CW = 32 * NCG;                            // couts per workgroup
    constexpr int WB = 8 * 4 * CW;                          // uint4 per weight buffer: up to 8 tap matrices [piece][k group][cout]
    constexpr int WSLOTS = WB / S2_NT;
    __shared__ uint4 xbuf[2 * S2_XB];
    __shared__ uint4 wbuf[2 * WB];
    __shared__ float red[(S2_NT / 64) * 32 * 3];
    __shared__ float nrm_s[FUSED ? 3 * S2_CIN_MAX : 4];
    __shared__ unsigned bound_s[S2_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = in.C, Cout = out.C, Do = out.D, Ho = out.H, Wo = out.W;
    const long long HWo = (long long)Ho * Wo, ovol = (long long)Do * HWo;
    const int nch = Cin / 16;
    const int RS = TC + 1, RV = (TR + 1) * RS;

    const unsigned ncg = (unsigned)(Cout / CW);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(lid % ncg);
    lid /= ncg;
    const unsigned b = lid % nblk;
    const int n = (int)(lid / nblk);
    const int tile = (int)(b % (unsigned)(txn * tyn)), zc = (int)(b / (unsigned)(txn * tyn));
    const int ox0 = (tile % txn) * TC, oy0 = (tile / txn) * TR;
    const int zs = zc * zchunk, ze = min(zs + zchunk, Do);

    // ---- FUSED: records -> LDS, the sample's input scale 2^e_in from their bounds (conv3d_h2.h)
    int e_fused = 0;
    bool poisoned_fused = false;
    if (FUSED) {
        unsigned mb = 0u;
        for (int c = tid; c < Cin; c += S2_NT) {
            const float4 a = *reinterpret_cast<const float4*>(in.nrm + (long long)n * in.nrm_n_stride + 4LL * c);
            nrm_s[3 * c] = a.x; nrm_s[3 * c + 1] = a.y; nrm_s[3 * c + 2] = a.z;
            const unsigned bb = abs_bits(a.w);
            mb = max(mb, bb == 0u ? 0x7fc00000u : bb);       // no bound given counts as non-finite
        }
        mb = wave_umax(mb);
        if (lane == 0) bound_s[wave] = mb;
        __syncthreads();
        mb = bound_s[0];
#pragma unroll
        for (int w = 1; w < S2_NT / 64; ++w) mb = max(mb, bound_s[w]);
        poisoned_fused = mb >= 0x7f800000u;
        e_fused = poisoned_fused ? 0 : min(max(15 - ((int)(mb >> 23) - 126), -100), 100);
        const float p_ = __uint_as_float((unsigned)(e_fused + 127) << 23);
        for (int c = tid; c < Cin; c += S2_NT) { nrm_s[3 * c] *= p_; nrm_s[3 * c + 1] *= p_; }      // each thread rescales the records it wrote
    }

    // ---- copies of this thread: operand cells (split form: their LDS index is the task index; fused form: (cell, channel quad) tasks) and weight cells
    unsigned xoff[S2_XSLOTS];
    int xcell[S2_XSLOTS], xquad[S2_XSLOTS];          // FUSED: destination in 8-byte units inside the high piece of a staged region; channel quad (or -1: no task)
    const int Wf = in.W;
    const long long HWf = (long long)in.H * in.W, DHWf = (long long)in.D * HWf;
#pragma unroll
    for (int s = 0; s < S2_XSLOTS; ++s) {
        // fused form: a thread without a task repeats the last task (the same loads, the same values into the same cell): the staging work has no branch
        const int t = FUSED ? min(tid + S2_NT * s, 4 * RV - 1) : tid + S2_NT * s;
        const int pk = t / RV, v = t - pk * RV;             // split form: pk = piece * 2 + k group; fused form: pk = channel quad of the step's 16 channels
        const int ry = v / RS, rx = v - ry * RS;
        const int gy = oy0 - 1 + ry, gx = ox0 - 1 + rx;
        const bool ok = t < 4 * RV && gy >= 0 && gy < Ho && gx >= 0 && gx < Wo;
        if (FUSED) {
            xoff[s] = ok ? 4u * (unsigned)((long long)(4 * pk) * DHWf + (long long)(2 * gy) * Wf + 2 * gx) : S2_DROP;
            xcell[s] = (((pk >> 1) * RV + v) * 2 + (pk & 1));
            xquad[s] = pk;
        } else {
            xoff[s] = ok ? 16u * (unsigned)((long long)pk * ovol + (long long)gy * Wo + gx) : S2_DROP;
            xcell[s] = 0; xquad[s] = 0;
        }
    }
    const uint4* xsn = FUSED ? nullptr : xs + (long long)n * nch * 32 * ovol;
    const long long xrest = (long long)(in.N - n) * nch * 32 * ovol * 16;              // bytes from this sample's first cell to the end of the workspace
    const float* fsrc = in.data + (long long)n * in.n_stride;
    const long long frest = (long long)(in.N - n) * in.n_stride * 4;
    const uint4* wcg = wp + (long long)cg * nch * 27 * 4 * CW;
    u32x4 xreg[FUSED ? 1 : S2_XSLOTS], wreg[WSLOTS];
    float xraw[FUSED ? S2_XSLOTS : 1][4];
    auto load_x = [&](int it, int ph, int ch) {
        if (FUSED) {
            const long long fo_ = (long long)(16 * ch) * DHWf + (long long)(2 * it + (ph >> 2)) * HWf + (long long)((ph >> 1) & 1) * Wf + (ph & 1);
            const long long left_ = frest - fo_ * 4;
            const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(fsrc) + fo_, 0, (int)(left_ < 0x7fffffffLL ? left_ : 0x7fffffffLL), 0x00020000);
#pragma unroll
            for (int s = 0; s < S2_XSLOTS; ++s)
#pragma unroll
                for (int i = 0; i < 4; ++i) xraw[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, xoff[s], (unsigned)(i * DHWf * 4), 0));
        } else {
            const long long xo_ = ((long long)(ch * 8 + ph) * 4 * ovol + (long long)it * HWo) * 16;
            const long long left_ = xrest - xo_;
            const auto xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(xsn) + xo_ / 16, 0, (int)(left_ < 0x7fffffffLL ? left_ : 0x7fffffffLL), 0x00020000);
#pragma unroll
            for (int s = 0; s < S2_XSLOTS; ++s) xreg[s] = __builtin_amdgcn_raw_buffer_load_b128(xr, xoff[s], 0, 0);
        }
    };
    auto load_w = [&](int ph, int ch) {
        const int nt_ = s2_ntaps(ph);
        const auto wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(wcg) + ((long long)ch * 27 + s2_tap_offset(ph)) * 4 * CW, 0, nt_ * 4 * CW * 16, 0x00020000);
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) wreg[s] = __builtin_amdgcn_raw_buffer_load_b128(wr, 16u * (unsigned)(tid + S2_NT * s), 0, 0);
    };
    auto store_x = [&](int bufi) {                          // split form: copies
        u32x4* xd = reinterpret_cast<u32x4*>(xbuf + bufi * S2_XB);
#pragma unroll
        for (int s = 0; s < S2_XSLOTS; ++s) xd[tid + S2_NT * s] = xreg[FUSED ? 0 : s];
    };
    auto convert_slot = [&](int s, int bufi, int ch) {      // fused form: activate + scale + split slot s of the raw values (channels 16 ch + 4 quad .. + 3) -> 8 bytes of each piece
        u32x2* xh = reinterpret_cast<u32x2*>(xbuf + bufi * S2_XB);
        _Float16 h_[4], l_[4];
        const bool keep = xoff[s] != S2_DROP;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = 16 * ch + 4 * xquad[s] + i;
            const float ya = act(xraw[FUSED ? s : 0][i], nrm_s[3 * c], nrm_s[3 * c + 1], nrm_s[3 * c + 2]);
            const float y = keep ? ya : 0.0f;                // zero padding is zero AFTER the activation (a select, not a branch)
            h2_split(y, h_[i], l_[i]);
        }
        const f16x2 h01 = {h_[0], h_[1]}, h23 = {h_[2], h_[3]}, l01 = {l_[0], l_[1]}, l23 = {l_[2], l_[3]};
        xh[xcell[s]] = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
        xh[xcell[s] + 4 * RV] = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
    };
    auto store_w = [&](int bufi) {
        u32x4* wd = reinterpret_cast<u32x4*>(wbuf + bufi * WB);
#pragma unroll
        for (int s = 0; s < WSLOTS; ++s) wd[tid + S2_NT * s] = wreg[s];
    };

    // ---- operands of this lane: A = the wave's M index 32 wave + (lane & 31) -> (row, column) of the tile, k group = lane >> 5; B = cout (lane & 31)
    const int r32 = lane & 31, kg = lane >> 5;
    const int m_a = wave * 32 + r32;
    const int arow = min(m_a / TC, TR - 1), acol = m_a % TC;                            // an M index beyond the tile computes on row TR - 1's cells: dropped later
    const int abase = kg * RV + arow * RS + acol;
    const int bbase = kg * CW + r32;

    // ---- epilogue geometry: register 4 j + i of an accumulator = M index 32 wave + 8 j + 4 kg + i -> four consecutive columns of one row
    float inv_a, inv_b;
    {
        const int ev = FUSED ? (poisoned_fused ? S2_POISON : e_fused) : expo[n];
        const bool poisoned = ev == S2_POISON;
        const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - (poisoned ? 0 : ev);
        const int t1_ = t_ / 2, t2_ = t_ - t1_;
        inv_a = poisoned ? __uint_as_float(0x7fc00000u) : __uint_as_float((unsigned)(t1_ + 127) << 23);
        inv_b = __uint_as_float((unsigned)(t2_ + 127) << 23);
    }
    const bool vec = (Wo & 3) == 0;                                                     // 16-byte stores: groups of four columns are inside or outside as a whole
    float* const obase = out.data + (long long)n * out.n_stride + (long long)(cg * CW) * ovol;
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(obase, 0, (int)((long long)CW * ovol * 4), 0x00020000);
    unsigned ooff[4];                  // byte offset of group j inside output plane z' = 0 of cout r32 (group g adds 32 planes), or S2_DROP
    int nval[4];                       // valid columns of the group (0 .. 4)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = wave * 32 + 8 * j + 4 * kg;
        const int row = m / TC, col = m - row * TC;
        const int gy = oy0 + row, gx = ox0 + col;
        const bool ok = row < TR && gy < Ho && gx < Wo;
        nval[j] = ok ? min(4, Wo - gx) : 0;
        ooff[j] = ok ? 4u * (unsigned)((long long)r32 * ovol + (long long)gy * Wo + gx) : S2_DROP;
    }
    float bco[NCG];
#pragma unroll
    for (int g = 0; g < NCG; ++g) bco[g] = bias ? bias[cg * CW + g * 32 + r32] : 0.0f;
    Stat run[NCG];
#pragma unroll
    for (int g = 0; g < NCG; ++g) { run[g].n = 0.0f; run[g].mean = 0.0f; run[g].m2 = 0.0f; }

    f32x16 acc[2][NCG];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int g = 0; g < NCG; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[s][g][i] = 0.0f;

    // ---- the march: iteration `it` = coarse plane index; a chunk that does not start at 0 first runs the odd-z phases of plane zs - 1 (their k 0 taps belong to zs)
    int it = zs > 0 ? zs - 1 : zs, ph = zs > 0 ? 4 : 0, ch = 0, buf = 0;
    // the step after (it, ph, ch); beyond the chunk's last step it stays where it is: the fused form's side work then repeats the last step's loads and converts
    // them into the buffer nobody reads any more -- no branch inside the matrix phase
    auto advance = [&](int& i_, int& p_, int& c_) {
        int i2 = i_, p2 = p_, c2 = c_ + 1;
        if (c2 == nch) { c2 = 0; ++p2; }
        if (p2 == 8) { p2 = 0; ++i2; }
        const bool ok_ = i2 < ze;
        i_ = ok_ ? i2 : i_; p_ = ok_ ? p2 : p_; c_ = ok_ ? c2 : c_;
        return ok_;
    };
    // Pipeline (both forms): at the start of step s the registers hold the operand cells and tap matrices of step s + 1 (requested during step s - 1).  Between
    // step s's matrix instructions they go into the other LDS buffer -- nobody reads it before the barrier at the end of the step -- and the requests of step s + 2
    // follow into the same registers: every load has more than a whole step to arrive, with one register set.
    load_x(it, ph, ch);
    load_w(ph, ch);
    if (FUSED) {
        __syncthreads();                                      // the rescaled records are complete
#pragma unroll
        for (int s = 0; s < S2_XSLOTS; ++s) convert_slot(s, 0, ch);
    } else {
        store_x(0);
    }
    store_w(0);
    {
        int i1 = it, p1 = ph, c1 = ch;
        advance(i1, p1, c1);
        load_x(i1, p1, c1);
        load_w(p1, c1);
    }
    __syncthreads();
    while (true) {
        int nit = it, nph = ph, nchk = ch;
        const bool has_next = advance(nit, nph, nchk);
        int n2it = nit, n2ph = nph, n2ch = nchk;
        advance(n2it, n2ph, n2ch);
        const uint4* xb = xbuf + buf * S2_XB;
        const uint4* wb = wbuf + buf * WB;
        // chunk k of the staging work rides in group min(k, groups - 1) of the step: operand slots 0 .. 2 (fused form: activate + scale + split; split form: copies),
        // the operand requests of step s + 2, then the tap matrices and their requests
        auto side = [&](int G, int NG) {
#pragma unroll
            for (int k = 0; k <= S2_XSLOTS + 1; ++k) {
                if ((k < NG - 1 ? k : NG - 1) != G) continue;
                if (k < S2_XSLOTS) {
                    if (FUSED) convert_slot(k, buf ^ 1, nchk);
                    else reinterpret_cast<u32x4*>(xbuf + (buf ^ 1) * S2_XB)[tid + S2_NT * k] = xreg[FUSED ? 0 : k];
                } else if (k == S2_XSLOTS) {
                    load_x(n2it, n2ph, n2ch);
                } else {
                    store_w(buf ^ 1);
                    load_w(n2ph, n2ch);
                }
            }
        };
        switch (ph) {
            case 0: s2_step_mm<1, 1, 1, NCG, FUSED>(xb, wb, abase, bbase, RS, RV, acc, side); break;
            case 1: s2_step_mm<1, 1, 2, NCG, FUSED>(xb, wb, abase, bbase, RS, RV, acc, side); break;
            case 2: s2_step_mm<1, 2, 1, NCG, FUSED>(xb, wb, abase, bbase, RS, RV, acc, side); break;
            case 3: s2_step_mm<1, 2, 2, NCG, FUSED>(xb, wb, abase, bbase, RS, RV, acc, side); break;
            case 4: s2_step_mm<2, 1, 1, NCG, FUSED>(xb, wb, abase, bbase, RS, RV, acc, side); break;
            case 5: s2_step_mm<2, 1, 2, NCG, FUSED>(xb, wb, abase, bbase, RS, RV, acc, side); break;
            case 6: s2_step_mm<2, 2, 1, NCG, FUSED>(xb, wb, abase, bbase, RS, RV, acc, side); break;
            default: s2_step_mm<2, 2, 2, NCG, FUSED>(xb, wb, abase, bbase, RS, RV, acc, side); break;
        }
        if (ph == 7 && ch == nch - 1) {                       // plane `it` is complete (unless it is the run-in plane zs - 1)
            if (it >= zs) {
                const unsigned so_ = (unsigned)it * (unsigned)(HWo * 4);
#pragma unroll
                for (int g = 0; g < NCG; ++g) {
                    float psum = 0.0f, pcnt = 0.0f;
                    f32x4 o_[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x4 v = {acc[0][g][4 * j], acc[0][g][4 * j + 1], acc[0][g][4 * j + 2], acc[0][g][4 * j + 3]};
                        v = v * inv_a * inv_b + bco[g];
                        o_[j] = v;
                        const unsigned go_ = ooff[j] == S2_DROP ? S2_DROP : ooff[j] + so_ + (unsigned)g * (unsigned)(32 * ovol * 4);
                        if (vec) {
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), orsrc, go_, 0, 0);
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < nval[j]) obase[(go_ >> 2) + i] = v[i];
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
                        Stat loc;
                        loc.n = pcnt; loc.mean = pmean; loc.m2 = pm2;
                        run[g] = stat_merge_nb(run[g], loc);
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < NCG; ++g)
#pragma unroll
                for (int i = 0; i < 16; ++i) { acc[0][g][i] = acc[1][g][i]; acc[1][g][i] = 0.0f; }
        }
        __syncthreads();
        if (!has_next) break;
        it = nit; ph = nph; ch = nchk; buf ^= 1;
    }

    if (STATS) {
#pragma unroll
        for (int g = 0; g < NCG; ++g) {
            Stat r_ = run[g], ot;
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
                for (int w = 0; w < S2_NT / 64; ++w) {
                    Stat o2;
                    o2.n = red[(w * 32 + tid) * 3]; o2.mean = red[(w * 32 + tid) * 3 + 1]; o2.m2 = red[(w * 32 + tid) * 3 + 2];
                    st = stat_merge(st, o2);
                }
                float* rec = stats + (((long long)n * Cout + cg * CW + g * 32 + tid) * nblk + b) * 3;
                rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
            }
            __syncthreads();
        }
    }
}

// w [Cout][Cin][3][3][3] -> [cout group of 32 NCG][chunk][27 taps in phase order][piece][k group][32 NCG couts][8 channels] fp16, scaled by tail[1]
// (conv3d_k3_h2_scale_kernel).  One thread per (cout, cin).
__global__ void __launch_bounds__(256)
conv3d_k3s2_h2_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int ncgw, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const int CW = 32 * ncgw, nch = Cin / 16;
    const float s = tail[1];
    const int cg = co / CW, col = co % CW, chunk = ci / 16, kgi = (ci % 16) / 8, j = ci % 8;
    for (int tap = 0; tap < 27; ++tap) {
        const int k3[3] = {tap / 9, (tap / 3) % 3, tap % 3};
        int p[3], sl[3];
        for (int a = 0; a < 3; ++a) { p[a] = k3[a] == 1 ? 0 : 1; sl[a] = k3[a] == 2 ? 1 : 0; }
        const int ph = p[0] * 4 + p[1] * 2 + p[2];
        const int tis = (sl[0] * (1 + p[1]) + sl[1]) * (1 + p[2]) + sl[2];
        _Float16 pc[2];
        h2_split(w[((long long)co * Cin + ci) * 27 + tap] * s, pc[0], pc[1]);
        _Float16* mat = packed + (((long long)(cg * nch + chunk) * 27 + s2_tap_offset(ph) + tis) * 4 * CW) * 8LL;
#pragma unroll
        for (int q = 0; q < 2; ++q) mat[((q * 2 + kgi) * CW + col) * 8 + j] = pc[q];
    }
}

}  // namespace mh
