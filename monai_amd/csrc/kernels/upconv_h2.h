// The decoder's "up" half of an UpCat block without its intermediate tensor (round 5).
//
// Reference op (monai/networks/nets/basic_unet.py:130-178, UpCat.forward): x_0 = ConvTranspose3d(k2, s2)(x) -- no normalisation, no activation -- then
// Conv3d(k3, p1)(cat([x_e, x_0])).  The convolution is linear in its input channels, so its x_0 half is the composition conv3 o deconv2 applied to x, and that
// composition is ONE transposed convolution with kernel 4, stride 2, padding 1 whose weights are products of the two layers' weights summed over the up channels
// (W4[ci][co][u] = sum_cm sum_{(d,k): d - k + 1 = u} Wd[ci][cm][d] Wc[co][cm][k] per axis; the deconvolution's bias becomes a 27-entry table per output channel: which of
// the convolution's taps fall inside the volume depends on first / interior / last position per axis).  Per FINE output voxel that is 2 x 2 x 2 coarse taps x Cin x Cout
// multiply-adds instead of 27 x Cup x Cout on a full-resolution tensor -- 3.4 x fewer at Cin == Cup -- and the full-resolution x_0 (7.2 GB per 64 windows of 96^3)
// is neither written by a transposed convolution nor read by the convolution.  Host side: monai_amd/networks/nets/basic_unet.py (_upcat_fused), which runs the
// composite term first (this kernel WRITES it together with the bias table) and then the convolution's x_e half on conv3d_k3_h2_kernel in its accumulating form
// (out += conv + bias, with the InstanceNorm statistics of the sum).  Where the x_e half runs on another kernel family the order is the other way round: this kernel
// then ADDS to the convolution's raw result in place and produces the statistics itself (RMW).
//
// Arithmetic: conv3d_h2.h's two-piece split precision (hi + lo fp16 pieces of the activated, power-of-two-scaled input and of the scaled composite weights, products
// hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16, fp32 accumulation), the same record / bound contract for the input.  The composite weights are formed in fp32 on
// the host from the fp32 parameters: one rounding per weight more than the two-layer evaluation, 1e-7 relative.
//
// Mapping.  An output voxel's taps depend on its parity per axis (f = 2c + p: p = 0 reads coarse c - 1 and c, p = 1 reads c and c + 1), so voxels of one parity
// class share their weight matrices and form a GEMM: a workgroup owns ONE (z, y) parity pair, a coarse tile of 16 rows x 16 columns (fine: 16 rows of that parity x 32
// columns), 32 output channels, and marches over its coarse z-chunk.  Wave w owns coarse rows 2w, 2w + 1 = a 32-voxel M block and carries BOTH x parities as two
// accumulator sets (so a lane's results are 8 consecutive fine columns per row and group: 16-byte accesses); per fine plane 16 (tz, ty, tx, px) tap matrices x
// Cin / 16 steps x 3 piece products.  The weight slab of the parity pair (64 KB at Cin = 32) stays in LDS for the whole march; coarse input planes are staged once each
// (activated, scaled, split) into a ring of two.
#pragma once
#include "common.h"
#include "conv3d_h2.h"

namespace mh {

constexpr int UC_TY = 16, UC_TX = 16;                      // coarse tile of a workgroup: 8 waves x 2 rows x 16 columns
constexpr int UC_NT = 64 * (UC_TY / 2);                    // threads
constexpr int UC_RY = UC_TY + 1, UC_RX = UC_TX + 2;        // staged coarse rows (one parity needs one halo row) and columns (both x parities: two halo columns)
constexpr int UC_PV = UC_RY * UC_RX;                       // 306 staged voxels per coarse plane
constexpr int UC_CIN = 32;                                 // input channels (two matrix-instruction K steps): the resident weight slab is sized for it
constexpr int UC_XP = 2 * 2 * UC_PV;                       // uint4 per piece of a staged plane: [k step][k group][voxel]
constexpr int UC_XB = 2 * UC_XP;                           // uint4 per staged plane (two pieces): 39 KB
constexpr int UC_WC = 2 * 2 * 2 * 32;                      // uint4 per tap matrix: [piece][k step][k group][cout]
constexpr int UC_WB = 16 * UC_WC;                          // uint4 per (z parity, y parity) weight slab: 16 tap matrices (tz, ty, px, tx) = 64 KB
constexpr int UC_NTASK = UC_PV * (UC_CIN / 4);             // staging tasks per plane: (staged voxel, channel quad)
constexpr int UC_NSLOT = (UC_NTASK + UC_NT - 1) / UC_NT;   // per thread: 5

// taps: coarse offset of tap t (0 / 1) for output parity p is t - 1 + p; the composite kernel index is u + 1 with u = f - 2 c' = [[2, 0], [1, -1]][p][t]
__host__ __device__ inline int uc_kernel_index(int p, int t) { return p == 0 ? (t == 0 ? 3 : 1) : (t == 0 ? 2 : 0); }

// Schedule of a fine plane (two waves per SIMD, one workgroup per CU):
//   1. matrix instructions over the first staged coarse plane; in their gaps the PREVIOUS fine plane's epilogue (bias table, old values, 8 x 16-byte stores per lane,
//      statistics) and then the requests for the raw values of the NEXT coarse plane (5 x 4 loads per thread, after the stores: they share vmcnt),
//   2. matrix instructions over the second staged coarse plane; in their gaps the conversion of those raw values (activate, scale, split) and, RMW, the requests for this
//      plane's old values,
//   3. the sums are scaled back into registers; barrier; the converted plane is written over the coarse plane that is no longer needed; barrier.
// RMW: the result is added to what `out` holds (and STATS are those of the sum); otherwise it is written (the accumulating convolution adds the skip half afterwards)
template <bool STATS, bool RMW>
__global__ void __launch_bounds__(UC_NT, 1)
upconv_k4s2_h2_kernel(Tensor low, const uint4* __restrict__ wp, const float* __restrict__ wtail, const float* __restrict__ btab, Tensor out,
                      float* __restrict__ stats, int txn, int tyn, int zchunk, unsigned nblk) {
    __shared__ uint4 xs[2 * UC_XB];
    __shared__ uint4 ws[UC_WB];
    __shared__ __attribute__((aligned(16))) float nrm_s[3 * UC_CIN];
    __shared__ unsigned bound_s[UC_NT / 64];
    __shared__ float red[(UC_NT / 64) * 32 * 3];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Dl = low.D, Hl = low.H, Wl = low.W, Cout = out.C;
    const int D = out.D, H = out.H, W = out.W;                 // == 2 Dl, 2 Hl, 2 Wl (the launcher checks)
    const long long HWl = (long long)Hl * Wl, DHWl = (long long)Dl * HWl, HW = (long long)H * W, DHW = (long long)D * HW;

    // 1-D launch over (sample, block, cout group); block = (parity pair, z chunk, tile)
    const unsigned ncg = (unsigned)(Cout / 32);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int cg = (int)(lid % ncg);
    lid /= ncg;
    const unsigned b = lid % nblk;
    const int n = (int)(lid / nblk);
    const int tile = (int)(b % (unsigned)(txn * tyn));
    const int rest_ = (int)(b / (unsigned)(txn * tyn));
    const int par = rest_ & 3, chunk = rest_ >> 2;
    const int pz = par >> 1, py = par & 1;
    const int cx0 = (tile % txn) * UC_TX, cy0 = (tile / txn) * UC_TY;
    const int cz_s = chunk * zchunk, cz_e = min(cz_s + zchunk, Dl);

    // ---- records of the input channels -> LDS, the sample's input scale 2^e_in from their bounds (conv3d_h2.h)
    unsigned mb = 0u;
    if (tid < UC_CIN) {
        const float4 a = load_nrm(low, n, tid);
        float* r_ = nrm_s + 12 * (tid >> 2) + (tid & 3);      // [channel quad][alpha | beta | slope][4]: a staging task reads its quad's records as three 16-byte values
        r_[0] = a.x; r_[4] = a.y; r_[8] = a.z;
        const unsigned bb = abs_bits(a.w);
        mb = low.nrm == nullptr ? abs_bits(1.0f) : (bb == 0u ? 0x7fc00000u : bb);      // no bound given counts as non-finite
    }
    mb = wave_umax(mb);
    if (lane == 0) bound_s[wave] = mb;
    // the parity pair's weight slab: resident for the whole march
    {
        const uint4* wsrc = wp + ((long long)(cg * 4 + par)) * UC_WB;
        for (int i = tid; i < UC_WB; i += UC_NT) ws[i] = wsrc[i];
    }
    for (int i = tid; i < 2 * UC_XB; i += UC_NT) xs[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    int e_in = 0;
    bool poisoned = false;
    {
        unsigned m4 = bound_s[0];
#pragma unroll
        for (int w = 1; w < UC_NT / 64; ++w) m4 = max(m4, bound_s[w]);
        poisoned = m4 >= 0x7f800000u;
        e_in = (poisoned || low.nrm == nullptr) ? 0 : min(max(15 - ((int)(m4 >> 23) - 126), -100), 100);
        const float p_ = __uint_as_float((unsigned)(e_in + 127) << 23);
        if (tid < UC_CIN) { nrm_s[12 * (tid >> 2) + (tid & 3)] *= p_; nrm_s[12 * (tid >> 2) + 4 + (tid & 3)] *= p_; }
    }
    __syncthreads();
    float inv_a, inv_b;
    {
        const int t_ = -((int)((__float_as_uint(wtail[1]) >> 23) & 0xffu) - 127) - e_in;
        const int t1_ = t_ / 2, t2_ = t_ - t1_;
        inv_a = poisoned ? __uint_as_float(0x7fc00000u) : __uint_as_float((unsigned)(t1_ + 127) << 23);
        inv_b = __uint_as_float((unsigned)(t2_ + 127) << 23);
    }

    // ---- staging tasks of this thread: (staged voxel, channel quad) -> 4 raw loads one plane ahead; later activate + scale + split, 8 bytes of the high plane and 8 of the low one
    constexpr unsigned UC_DROP = 0x80000000u;
    unsigned soff[UC_NSLOT];          // byte offset of (channel 4 q, y, x) inside the sample's coarse plane z = 0, or UC_DROP (zero padding / no task)
    int scell[UC_NSLOT];              // destination in 8-byte units inside a piece of a staged plane
    int squad[UC_NSLOT];
#pragma unroll
    for (int s = 0; s < UC_NSLOT; ++s) {
        const int t = min(tid + UC_NT * s, UC_NTASK - 1);
        const int q = t / UC_PV, v = t - q * UC_PV;          // channel quad, staged voxel
        const int ly = v / UC_RX, lx = v - ly * UC_RX;
        const int gy = cy0 - (1 - py) + ly, gx = cx0 - 1 + lx;
        const bool ok = tid + UC_NT * s < UC_NTASK && gy >= 0 && gy < Hl && gx >= 0 && gx < Wl;
        soff[s] = ok ? 4u * (unsigned)((long long)(4 * q) * DHWl + (long long)gy * Wl + gx) : UC_DROP;
        // cell [k step = q >> 2][k group = (q >> 1) & 1][voxel] holds 8 channels = two 8-byte halves (q & 1); a thread without a task rewrites the last task's cell with the same zeros
        scell[s] = (((q >> 2) * 2 + ((q >> 1) & 1)) * UC_PV + v) * 2 + (q & 1);
        squad[s] = tid + UC_NT * s < UC_NTASK ? q : -1;
    }
    const float* src = low.data + (long long)n * low.n_stride;
    const long long lrest = (long long)(low.N - n) * low.n_stride * 4;
    float sreg[UC_NSLOT][4];
    auto load_plane = [&](int z, bool on = true) __attribute__((always_inline)) {          // requests only: the values are used a plane later; !on: every offset beyond the buffer (branch-free inside a scheduling region)
        const unsigned offm = on ? 0u : UC_DROP;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src + (long long)z * HWl), 0, (int)(lrest < 0x7fffffffLL ? lrest : 0x7fffffffLL), 0x00020000);
#pragma unroll
        for (int s = 0; s < UC_NSLOT; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) sreg[s][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, soff[s] | offm, (unsigned)(i * DHWl * 4), 0));
    };
    u32x2 chi[UC_NSLOT], clo[UC_NSLOT];                      // the converted pieces of the next plane, held until the plane they replace is no longer read
    const float m1 = h2_minus_one();
    auto convert_regs = [&]() __attribute__((always_inline)) {          // activate + scale + split, branch-free (vector ALU work that rides between the matrix instructions)
#pragma unroll
        for (int s = 0; s < UC_NSLOT; ++s) {
            const f32x4* rq_ = reinterpret_cast<const f32x4*>(nrm_s) + 3 * max(squad[s], 0);
            const f32x4 al_ = rq_[0], be_ = rq_[1], sl_ = rq_[2];
            const unsigned keep_ = ~(unsigned)((int)soff[s] >> 31);      // zero padding / no task: the load returned 0, the activated value must be 0 as well
            float y_[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) y_[i] = __uint_as_float(__float_as_uint(act(sreg[s][i], al_[i], be_[i], sl_[i])) & keep_);
            f16x2 h01, h23, l01, l23;
            h2_split_pair(y_[0], y_[1], m1, h01, l01);
            h2_split_pair(y_[2], y_[3], m1, h23, l23);
            chi[s] = u32x2{__builtin_bit_cast(unsigned, h01), __builtin_bit_cast(unsigned, h23)};
            clo[s] = u32x2{__builtin_bit_cast(unsigned, l01), __builtin_bit_cast(unsigned, l23)};
        }
    };
    auto write_regs = [&](int bufi) __attribute__((always_inline)) {
        u32x2* xh = reinterpret_cast<u32x2*>(xs + bufi * UC_XB);
#pragma unroll
        for (int s = 0; s < UC_NSLOT; ++s)
            if (squad[s] >= 0) { xh[scell[s]] = chi[s]; xh[scell[s] + 2 * UC_XP] = clo[s]; }
    };
    auto convert_plane = [&](int bufi) __attribute__((always_inline)) { convert_regs(); write_regs(bufi); };

    // ---- operands of this lane: A = coarse voxel (row 2 wave + (r >> 4), column r & 15), B = cout r; k group = lane >> 5
    const int r32 = lane & 31, kg = lane >> 5;
    const int abase = kg * UC_PV + (2 * wave + (r32 >> 4)) * UC_RX + (r32 & 15);
    const int bbase = kg * 32 + r32;

    // ---- epilogue geometry: lane = cout; register i of an accumulator = coarse row i >> 3, columns ((i >> 2) & 1) * 8 + kg * 4 + (i & 3)
    const int co = cg * 32 + r32;
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(out.data + (long long)n * out.n_stride + (long long)(cg * 32) * DHW, 0, (int)(32 * DHW * 4), 0x00020000);
    unsigned ooff[4];                  // (coarse row r_, column group g_): byte offset of its 8 fine columns inside an output plane of this cout, or UC_DROP
    int fyv[2];
    bool okg[4];
#pragma unroll
    for (int r_ = 0; r_ < 2; ++r_) fyv[r_] = 2 * (cy0 + 2 * wave + r_) + py;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r_ = j >> 1, g_ = j & 1;
        const int ccol = cx0 + g_ * 8 + kg * 4;
        okg[j] = cy0 + 2 * wave + r_ < Hl && ccol < Wl;       // Wl % 4 == 0: a group of four coarse columns is inside or outside as a whole
        ooff[j] = okg[j] ? 4u * (unsigned)((long long)r32 * DHW + (long long)fyv[r_] * W + 2 * ccol) : UC_DROP;
    }
    Stat run;
    run.n = 0.0f; run.mean = 0.0f; run.m2 = 0.0f;
    // bias table [27][Cout] (class first / interior / last per axis): the 3 z classes x this wave's two rows x 3 x classes of this lane's cout, once (a table load
    // inside the epilogue would sit, with its latency, between the last matrix instruction and the stores of every plane)
    float btr[3][2][3];
#pragma unroll
    for (int cz_ = 0; cz_ < 3; ++cz_)
#pragma unroll
        for (int r_ = 0; r_ < 2; ++r_) {
            const int cly = fyv[r_] == 0 ? 0 : (fyv[r_] == H - 1 ? 2 : 1);
#pragma unroll
            for (int cx_ = 0; cx_ < 3; ++cx_) btr[cz_][r_][cx_] = btab[(long long)((cz_ * 3 + cly) * 3 + cx_) * Cout + co];
        }

    // prologue: the first plane's two coarse planes
    {
        const int za = cz_s - 1 + pz, zb = cz_s + pz;
        if (za >= 0 && za < Dl) { load_plane(za); convert_plane(za & 1); }
        if (zb >= 0 && zb < Dl) { load_plane(zb); convert_plane(zb & 1); }
    }
    __syncthreads();

    // A fine plane's sums leave the matrix phase scaled back (op_) and wait there: their bias, old values, stores and statistics are branch-free pieces that ride in
    // the matrix-instruction gaps of the NEXT plane's first half (conv3d_h2.h's scheme; round 6 -- before, the whole epilogue sat between the two barriers of a plane
    // with no matrix instruction beside it: matrix pipe 0.34).  pend: op_ holds a plane (0 in the first iteration of a chunk: offsets beyond the buffer, weights 0).
    f32x4 op_[8], prev[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { op_[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; prev[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; }
    int pend = 0, fzp = 0;
    float ecnt_ = 0.0f, esum_ = 0.0f, emean_ = 0.0f, em2_ = 0.0f;
    auto request_old = [&](int fz_) __attribute__((always_inline)) {          // RMW: the old values of fine plane fz_, requested half a plane ahead of their use
        const unsigned so_ = (unsigned)fz_ * (unsigned)(HW * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            prev[2 * j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(orsrc, ooff[j] + so_, 0, 0));
            prev[2 * j + 1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(orsrc, ooff[j] + so_ + 16u, 0, 0));
        }
    };
    // piece A (row groups J0_, J0_ + 1): bias table, + old value, two 16-byte stores per group (the plane offset in the VECTOR offset: conv3d_h2.h on the scalar-offset hazard)
#define UC_EMIT_J(J_)                                                                                 \
    {                                                                                                 \
        constexpr int j = (J_), r_ = j >> 1, g_ = j & 1;      /* compile-time: the table stays in registers */ \
        const float bmid = clz == 0 ? btr[0][r_][1] : (clz == 1 ? btr[1][r_][1] : btr[2][r_][1]);     \
        const float bfst = clz == 0 ? btr[0][r_][0] : (clz == 1 ? btr[1][r_][0] : btr[2][r_][0]);     \
        const float blst = clz == 0 ? btr[0][r_][2] : (clz == 1 ? btr[1][r_][2] : btr[2][r_][2]);     \
        const int fx0 = 2 * (cx0 + g_ * 8 + kg * 4);      /* first of this group's 8 fine columns */  \
        const float bfirst = fx0 == 0 ? bfst : bmid, blast = fx0 + 8 == W ? blst : bmid;              \
        const f32x4 b0 = {bfirst, bmid, bmid, bmid}, b1 = {bmid, bmid, bmid, blast};                  \
        f32x4 v0 = op_[2 * j] + b0, v1 = op_[2 * j + 1] + b1;                                         \
        if (RMW) { v0 = v0 + prev[2 * j]; v1 = v1 + prev[2 * j + 1]; }                                \
        const unsigned oo_ = (pend ? ooff[j] : UC_DROP) + so_;                                        \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v0), orsrc, oo_, 0, 0);      \
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v1), orsrc, oo_ + 16u, 0, 0); \
        op_[2 * j] = v0; op_[2 * j + 1] = v1;                                                         \
    }
#define UC_EMIT_A(J0_)                                                                                \
    {                                                                                                 \
        const unsigned so_ = (unsigned)fzp * (unsigned)(HW * 4);                                      \
        const int clz = fzp == 0 ? 0 : (fzp == D - 1 ? 2 : 1);                                        \
        UC_EMIT_J(J0_) UC_EMIT_J((J0_) + 1)                                                           \
    }
#define UC_EMIT_B1                                                                                    \
    if (STATS) {                                                                                      \
        const float pf_ = pend ? 1.0f : 0.0f;                                                         \
        esum_ = 0.0f; ecnt_ = 0.0f;                                                                   \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            const float w_ = okg[j] ? pf_ : 0.0f;                                                     \
            const f32x4 v0 = op_[2 * j], v1 = op_[2 * j + 1];                                         \
            ecnt_ += 8.0f * w_;                                                                       \
            esum_ += (((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v1[0] + v1[1]) + (v1[2] + v1[3]))) * w_; \
        }                                                                                             \
        emean_ = ecnt_ > 0.0f ? esum_ / (ecnt_ > 0.0f ? ecnt_ : 1.0f) : 0.0f;                         \
    }
#define UC_EMIT_B2                                                                                    \
    if (STATS) {                                                                                      \
        em2_ = 0.0f;                                                                                  \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                               \
            const f32x4 d0 = op_[2 * j] - emean_, d1 = op_[2 * j + 1] - emean_;                       \
            const f32x4 q0 = d0 * d0, q1 = d1 * d1;                                                   \
            em2_ += (((q0[0] + q0[1]) + (q0[2] + q0[3])) + ((q1[0] + q1[1]) + (q1[2] + q1[3]))) * (okg[j] && pend ? 1.0f : 0.0f); \
        }                                                                                             \
    }
#define UC_EMIT_B3                                                                                    \
    {                                                                                                 \
        if (STATS) {                                                                                  \
            Stat loc_;                                                                                \
            loc_.n = ecnt_; loc_.mean = emean_; loc_.m2 = em2_;                                       \
            run = stat_merge_nb(run, loc_);                                                           \
        }                                                                                             \
        pend = 0;                                                                                     \
    }
#define UC_NONE
#ifndef UC_DSG
#define UC_DSG 2      // operand reads the scheduler may place per matrix-instruction gap (the next group's eight reads: 2 = over four gaps, 4 = over the first two)
#endif

    for (int c = cz_s; c < cz_e; ++c) {
        const int za = c - 1 + pz, zb = c + pz;               // the two coarse planes of fine plane 2 c + pz
        const int fz = 2 * c + pz;
        const int zn = zb + 1;
        const bool stage_next = c + 1 < cz_e && zn < Dl;

        // matrix instructions: 16 groups (tz, k step, ty, px) of 8 operand reads -> 6 instructions; a group's operands are fetched while the group before it
        // multiplies (two register sets), the scheduler deals the reads -- in the first half together with the previous plane's epilogue pieces and the requests for
        // the next coarse plane's raw values (AFTER the stores: both share vmcnt), in the second half with the conversion of those values -- out over the gaps
        // (a wave issues in order: an operand read right in front of its use costs the LDS latency every time; conv3d_h2.h)
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc[0][i] = 0.0f; acc[1][i] = 0.0f; }
        uint4 ahq[2][2], alq[2][2], bhq[2][2], blq[2][2];
#define UC_FETCH(B_, XB_, TZ_, G_)                                                                    \
    {                                                                                                 \
        constexpr int ks_ = ((G_) >> 2) & 1, ty_ = ((G_) >> 1) & 1, px_ = (G_) & 1;                   \
        const uint4* ap_ = (XB_) + ks_ * (2 * UC_PV) + abase + ty_ * UC_RX + px_;                     \
        const uint4* bp_ = ws + ((((TZ_) * 2 + ty_) * 2 + px_) * 2) * UC_WC + ks_ * 64 + bbase;       \
        /* in the order the matrix instructions consume them (tx 0: ah bh | al bh | ah bl, then tx 1): the first operands are the first to arrive */ \
        ahq[B_][0] = ap_[0]; bhq[B_][0] = bp_[0]; alq[B_][0] = ap_[UC_XP]; blq[B_][0] = bp_[128];     \
        ahq[B_][1] = ap_[1]; bhq[B_][1] = bp_[UC_WC]; alq[B_][1] = ap_[UC_XP + 1]; blq[B_][1] = bp_[UC_WC + 128]; \
    }
#define UC_MM(B_, PX_)                                                                                \
    _Pragma("unroll") for (int tx_ = 0; tx_ < 2; ++tx_) {                                             \
        acc[PX_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ahq[B_][tx_]), __builtin_bit_cast(f16x8, bhq[B_][tx_]), acc[PX_], 0, 0, 0); \
        acc[PX_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, alq[B_][tx_]), __builtin_bit_cast(f16x8, bhq[B_][tx_]), acc[PX_], 0, 0, 0); \
        acc[PX_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ahq[B_][tx_]), __builtin_bit_cast(f16x8, blq[B_][tx_]), acc[PX_], 0, 0, 0); \
    }
    // NV_ vector / scalar and NM_ vector-memory instructions per matrix-instruction gap; the code argument joins the group's scheduling region
#define UC_GROUP(XB_, TZ_, G_, NV_, NM_, ...)                                                         \
    {                                                                                                 \
        if ((G_) + 1 < 8) UC_FETCH(((G_) + 1) & 1, XB_, TZ_, ((G_) + 1 < 8 ? (G_) + 1 : 0))            \
        __VA_ARGS__                                                                                   \
        UC_MM((G_) & 1, (G_) & 1)                                                                     \
        _Pragma("unroll") for (int g_ = 0; g_ < 6; ++g_) {                                            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                        \
            __builtin_amdgcn_sched_group_barrier(0x100, UC_DSG, 0);                                   \
            __builtin_amdgcn_sched_group_barrier(0x006, NV_, 0);                                      \
            if ((NM_) > 0) __builtin_amdgcn_sched_group_barrier(0x010, NM_, 0);                       \
        }                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    }
#define UC_HALF(XB_, TZ_, NV_, REQ_)                                                                  \
    {                                                                                                 \
        UC_FETCH(0, XB_, TZ_, 0)                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        UC_GROUP(XB_, TZ_, 0, NV_, 0, UC_NONE) UC_GROUP(XB_, TZ_, 1, NV_, 0, UC_NONE) UC_GROUP(XB_, TZ_, 2, NV_, 0, UC_NONE) UC_GROUP(XB_, TZ_, 3, NV_, 0, UC_NONE) \
        UC_GROUP(XB_, TZ_, 4, NV_, 2, REQ_) UC_GROUP(XB_, TZ_, 5, NV_, 0, UC_NONE) UC_GROUP(XB_, TZ_, 6, NV_, 0, UC_NONE) UC_GROUP(XB_, TZ_, 7, NV_, 0, UC_NONE) \
    }
    // the first half with the previous plane's epilogue and the next coarse plane's requests in its gaps
#define UC_HALF_EMIT(XB_, TZ_)                                                                        \
    {                                                                                                 \
        UC_FETCH(0, XB_, TZ_, 0)                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        UC_GROUP(XB_, TZ_, 0, 8, 1, UC_EMIT_A(0)) UC_GROUP(XB_, TZ_, 1, 8, 1, UC_EMIT_A(2))           \
        UC_GROUP(XB_, TZ_, 2, 4, 4, load_plane(zn, stage_next);)                                  \
        UC_GROUP(XB_, TZ_, 3, 8, 0, UC_EMIT_B1) UC_GROUP(XB_, TZ_, 4, 8, 0, UC_EMIT_B2) UC_GROUP(XB_, TZ_, 5, 8, 0, UC_EMIT_B3) \
        UC_GROUP(XB_, TZ_, 6, 2, 0, UC_NONE) UC_GROUP(XB_, TZ_, 7, 2, 0, UC_NONE)                     \
    }
#define UC_REQ_OLD if (RMW) request_old(fz);
        if (za >= 0 && za < Dl) {                             // zero padding: nothing to add (wave-uniform); then this is a chunk's first plane and nothing is pending
            const uint4* xb = xs + (za & 1) * UC_XB;
            UC_HALF_EMIT(xb, 0)
        } else if (stage_next) load_plane(zn);
        if (zb >= 0 && zb < Dl) {
            const uint4* xb = xs + (zb & 1) * UC_XB;
            if (stage_next) {      // the raw values requested in the first half have arrived: their conversion shares this half's matrix time
                convert_regs();
                UC_HALF(xb, 1, 8, UC_REQ_OLD)
            } else {
                UC_HALF(xb, 1, 2, UC_REQ_OLD)
            }
        } else {
            if (stage_next) convert_regs();
            UC_REQ_OLD
        }
        // the sums, scaled back, wait for the next plane's first half
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int bi = (j >> 1) * 8 + (j & 1) * 4;
            const f32x4 v0 = {acc[0][bi], acc[1][bi], acc[0][bi + 1], acc[1][bi + 1]};
            const f32x4 v1 = {acc[0][bi + 2], acc[1][bi + 2], acc[0][bi + 3], acc[1][bi + 3]};
            op_[2 * j] = v0 * inv_a * inv_b;
            op_[2 * j + 1] = v1 * inv_a * inv_b;
        }
        pend = 1;
        fzp = fz;
        __syncthreads();                                      // every wave is done with plane za: its buffer takes the next plane
        if (stage_next) write_regs(zn & 1);
        __syncthreads();                                      // the next plane's staged input is complete
    }
    // the chunk's last plane
    UC_EMIT_A(0) UC_EMIT_A(2) UC_EMIT_B1 UC_EMIT_B2 UC_EMIT_B3
#undef UC_REQ_OLD
#undef UC_HALF_EMIT
#undef UC_HALF
#undef UC_GROUP
#undef UC_MM
#undef UC_FETCH
#undef UC_NONE
#undef UC_EMIT_B3
#undef UC_EMIT_B2
#undef UC_EMIT_B1
#undef UC_EMIT_A
#undef UC_EMIT_J

    if (STATS) {
        {
            Stat ot;
            ot.n = __shfl_xor(run.n, 32);
            ot.mean = __shfl_xor(run.mean, 32);
            ot.m2 = __shfl_xor(run.m2, 32);
            run = kg == 0 ? stat_merge(run, ot) : stat_merge(ot, run);
        }
        if (kg == 0) { red[(wave * 32 + r32) * 3] = run.n; red[(wave * 32 + r32) * 3 + 1] = run.mean; red[(wave * 32 + r32) * 3 + 2] = run.m2; }
        __syncthreads();
        if (tid < 32) {
            Stat st;
            st.n = 0.0f; st.mean = 0.0f; st.m2 = 0.0f;
#pragma unroll
            for (int w = 0; w < UC_NT / 64; ++w) {
                Stat ot;
                ot.n = red[(w * 32 + tid) * 3]; ot.mean = red[(w * 32 + tid) * 3 + 1]; ot.m2 = red[(w * 32 + tid) * 3 + 2];
                st = stat_merge(st, ot);
            }
            float* rec = stats + (((long long)n * Cout + cg * 32 + tid) * nblk + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}

// Composite weights w4 [Cin][Cout][4][4][4] (fp32, formed on the host from the two layers' parameters) -> per (cout group, z parity, y parity) slab of 16 tap matrices
// (tz, ty, px, tx), each [piece][k step][k group][32 couts][8 channels] fp16, scaled by tail[1] (conv3d_k3_h2_scale_kernel).  One thread per (ci, co).
__global__ void __launch_bounds__(256)
upconv_k4s2_pack_kernel(const float* __restrict__ w4, int Cin, int Cout, _Float16* __restrict__ packed, const float* __restrict__ tail) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const float s = tail[1];
    const float* w = w4 + ((long long)ci * Cout + co) * 64;
    const int cg = co / 32, col = co % 32, ks = ci / 16, kgi = (ci % 16) / 8, j = ci % 8;
    for (int par = 0; par < 4; ++par)
        for (int combo = 0; combo < 16; ++combo) {
            const int tz = combo >> 3, ty = (combo >> 2) & 1, px = (combo >> 1) & 1, tx = combo & 1;
            const int uz = uc_kernel_index(par >> 1, tz), uy = uc_kernel_index(par & 1, ty), ux = uc_kernel_index(px, tx);
            _Float16 pc[2];
            h2_split(w[(uz * 4 + uy) * 4 + ux] * s, pc[0], pc[1]);
            _Float16* slab = packed + ((long long)((cg * 4 + par) * 16 + combo)) * (UC_WC * 8LL);
#pragma unroll
            for (int p = 0; p < 2; ++p) slab[(((p * 2 + ks) * 2 + kgi) * 32 + col) * 8 + j] = pc[p];
        }
}

}  // namespace mh
