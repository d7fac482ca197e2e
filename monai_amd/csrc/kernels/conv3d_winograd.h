// Conv3d 3x3x3 / stride 1 / zero padding 1 by Winograd F(2x2x2, 3x3x3) on the fp32 matrix cores of gfx950.
//
// Same reference op and same "normalise on load" contract as conv3d_mfma.h (nn.Conv3d of `Convolution`,
// monai/networks/blocks/convolutions.py:98-171, fed by the previous block's deferred InstanceNorm + LeakyReLU).
// The direct implicit GEMM spends 27 multiply-adds per (output voxel, cin, cout); the minimal-filtering form spends
// 64 per 8 output voxels = 8: 3.375x fewer matrix-core cycles, and the matrix cores are what bounds the direct kernel
// (83 % of the fp32 MFMA peak).  fp32 throughout; measured against an fp64 evaluation the BasicUNet logits are as
// accurate as with the direct form (tools/winograd_numerics.py: 2.3e-6 vs 2.7e-6 max abs).
//
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A   applied along z, y and x;   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],
//   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
//
// Mapping.  A tile is a 2x2x2 block of outputs (4x4x4 input patch).  For each of the 64 transform positions xi the
// products form a GEMM  M_xi[tile][cout] = sum_cin V_xi[tile][cin] * U_xi[cin][cout]  -> v_mfma_f32_16x16x4_f32 with
// M = 16 tiles, N = 16 couts, K = 4 input channels.  One wave owns 16 tiles x 16 couts and ALL 64 xi: 64 accumulators
// of 4 registers (256 accumulation registers, one wave per SIMD), so the inverse transform of the accumulators is
// register-local.  Operand A of xi: lane l holds V_xi[tile l & 15][cin l >> 4] -- each lane transforms ITS OWN patch
// (64 LDS reads, 192 adds per 4-channel step, issued in the shadow of 64 MFMAs = 2048 cycles); operand B: lane l holds
// U_xi[cin l >> 4][cout l & 15], 64 consecutive floats of the pre-transformed weights.
// A workgroup = 4 waves = 64 tiles (2 x 4 x 8: a 4 x 8 x 16 output region) x 16 couts; its input region
// [4 cin][6][10][18] (producer's norm + activation applied) and the step's 64 x 64 weight slab are staged through LDS,
// double buffered, one barrier per 4-channel step; every global load is issued a full step before its LDS commit (the
// vector-memory counter is in-order: a B operand fetched from global behind the staging loads would drain them).
// Epilogue: inverse transform, bias, fused InstanceNorm statistics (same record format as conv3d_mfma.h), transpose
// through LDS, row-contiguous float4 stores.
#pragma once
#include "common.h"

namespace mh {

constexpr int WG_OZ = 4, WG_OY = 8, WG_OX = 16;            // output region of a workgroup
constexpr int WG_RZ = 6, WG_RY = 10, WG_RX = 18;           // input region (halo 1)
constexpr int WG_PX = WG_RX, WG_PLANE = WG_RY * WG_PX;     // LDS row pitch / plane: the region is stored densely, so a
constexpr int WG_CS = WG_RZ * WG_PLANE + 8;                // lane's staging slot j sits at the constant offset 64 j
constexpr int WG_KC = 4;                                   // input channels per MFMA (K)
constexpr int WG_BUF = WG_KC * WG_CS;                      // one staging buffer (floats)
constexpr int WG_REGION = WG_RZ * WG_RY * WG_RX;           // 1080 region elements per channel
constexpr int WG_SLOTS = (WG_REGION + 63) / 64;            // 17 per lane: wave w stages channel w of the step
constexpr int WG_CN = 16;                                  // couts per workgroup
constexpr int WG_OS = WG_OZ * WG_OY * WG_OX + 4;           // epilogue LDS stride between couts (516)
constexpr int WG_UBUF = 64 * 64;                           // weight slab of one step: [xi][cin & 3][cout & 15]
constexpr int WG_SMEM = 2 * WG_BUF + 2 * WG_UBUF;
static_assert(WG_CN * WG_OS + 4 * WG_CN * 3 + 256 <= WG_SMEM, "epilogue transpose + statistics scratch fit the staging area");

// 1-D input transform B^T d
#define MH_WBT(o0, o1, o2, o3, d0, d1, d2, d3) \
    { o0 = (d0) - (d2); o1 = (d1) + (d2); o2 = (d2) - (d1); o3 = (d1) - (d3); }

template <bool STATS>
__global__ void __launch_bounds__(256, 1)
conv3d_k3_winograd_kernel(Tensor in, const float* __restrict__ up, const float* __restrict__ bias, Tensor out,
                          float* __restrict__ stats, int rx, int ry, int rz) {
    __shared__ __attribute__((aligned(16))) float smem[WG_SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t16 = lane & 15, kq = lane >> 4;
    const int Cin = in.C, Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long DHW = (long long)D * H * W;
    const int KS = Cin / WG_KC;

    const unsigned nreg = gridDim.x;
    const unsigned b = xcd_remap(blockIdx.x, nreg);
    const int x0 = (int)(b % rx) * WG_OX, y0 = (int)((b / rx) % ry) * WG_OY, z0 = (int)(b / (rx * ry)) * WG_OZ;
    const int cg = blockIdx.y, n = blockIdx.z;
    (void)rz;

    // staging: wave w stages channel (4 s + w); lane elements e = lane + 64 j of the 6 x 10 x 18 region
    unsigned soff[WG_SLOTS];   // BYTE offsets inside a channel volume: 32-bit lane offset + uniform 64-bit base
    unsigned sokm = 0u;
#pragma unroll
    for (int j = 0; j < WG_SLOTS; ++j) {
        const int e = lane + 64 * j;
        const int lz = e / (WG_RY * WG_RX), r2 = e - lz * (WG_RY * WG_RX);
        const int ly = r2 / WG_RX, lx = r2 - ly * WG_RX;
        const int gz = z0 + lz - 1, gy = y0 + ly - 1, gx = x0 + lx - 1;
        const bool ok = e < WG_REGION && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        sokm |= (unsigned)ok << j;
        soff[j] = ok ? 4u * (unsigned)((gz * H + gy) * W + gx) : 0u;
    }
    const float* src = in.data + (long long)n * in.n_stride + (long long)wave * DHW;
    const f32x4* ug = reinterpret_cast<const f32x4*>(up + (long long)cg * KS * WG_UBUF) + tid;
    float* const us = smem + 2 * WG_BUF;
    float xin[WG_SLOTS];
    f32x4 uin[4];
    // loads of step S into registers, slots [J0, J1) of the region and [Q0, Q1) of the weight slab
#define MH_WG_ISSUE_PART(S, J0, J1, Q0, Q1)                                                       \
    {                                                                                             \
        const char* pl_ = reinterpret_cast<const char*>(src + (long long)(S) * WG_KC * DHW);      \
        _Pragma("unroll") for (int j = (J0); j < (J1); ++j)                                       \
            if (j >= 0 && j < WG_SLOTS) xin[j] = *reinterpret_cast<const float*>(pl_ + soff[j]);  \
        _Pragma("unroll") for (int j = (Q0); j < (Q1); ++j)                                       \
            if (j >= 0 && j < 4) uin[j] = ug[(long long)(S) * (WG_UBUF / 4) + 256 * j];           \
    }
#define MH_WG_ISSUE(S) MH_WG_ISSUE_PART(S, 0, WG_SLOTS, 0, 4)
    // registers -> LDS buffer BUF with the producer's norm + activation (a_ = its {alpha, beta, slope})
#define MH_WG_COMMIT_PART(a_, BUF, J0, J1, Q0, Q1)                                                \
    {                                                                                             \
        float* xs_ = smem + (BUF) * WG_BUF + wave * WG_CS + lane;                                 \
        _Pragma("unroll") for (int j = (J0); j < (J1); ++j)                                       \
            if (j >= 0 && j < WG_SLOTS && (64 * j + 63 < WG_REGION || lane + 64 * j < WG_REGION)) \
                xs_[64 * j] = ((sokm >> j) & 1u) ? act(xin[j], a_.x, a_.y, a_.z) : 0.0f;          \
        _Pragma("unroll") for (int j = (Q0); j < (Q1); ++j)                                       \
            if (j >= 0 && j < 4) reinterpret_cast<f32x4*>(us + (BUF) * WG_UBUF)[tid + 256 * j] = uin[j]; \
    }
#define MH_WG_COMMIT(S, BUF)                                                                      \
    {                                                                                             \
        const float4 a0_ = load_nrm(in, n, (S) * WG_KC + wave);                                   \
        MH_WG_COMMIT_PART(a0_, BUF, 0, WG_SLOTS, 0, 4)                                            \
    }

    // this lane's patch: tile (tz, ty, tx) of the workgroup's 2 x 4 x 8, input channel kq of the step
    const int tz = wave >> 1, ty = 2 * (wave & 1) + (t16 >> 3), tx = t16 & 7;
    const int pbase = kq * WG_CS + (2 * tz) * WG_PLANE + (2 * ty) * WG_PX + 2 * tx;

    f32x4 acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // the patch as (x, x+1) register pairs, exactly as the 8-byte LDS reads deliver them: the z and y passes of the
    // input transform are packed adds on those pairs, only the x pass works inside a pair
    f32x2 raw[4][4][2];
#define MH_WG_READ_PATCH_PART(BUF, Z0, Z1)                                                        \
    {                                                                                             \
        const float* xs_ = smem + (BUF) * WG_BUF + pbase;                                         \
        _Pragma("unroll") for (int z = (Z0); z < (Z1); ++z)                                       \
            _Pragma("unroll") for (int y = 0; y < 4; ++y) {                                       \
                raw[z][y][0] = *reinterpret_cast<const f32x2*>(xs_ + z * WG_PLANE + y * WG_PX);      \
                raw[z][y][1] = *reinterpret_cast<const f32x2*>(xs_ + z * WG_PLANE + y * WG_PX + 2);  \
            }                                                                                     \
    }
#define MH_WG_READ_PATCH(BUF) MH_WG_READ_PATCH_PART(BUF, 0, 4)

    // prologue: stage step 0, read its patches and first B operands, have step 1's loads in flight
    MH_WG_ISSUE(0)
    MH_WG_COMMIT(0, 0)
    __syncthreads();
    MH_WG_READ_PATCH(0)
    float ub[2][16];     // B operands: group g reads its 16 slabs from LDS one group ahead (two register sets, static roles)
#pragma unroll
    for (int i = 0; i < 16; ++i) ub[0][i] = us[i * 64 + lane];
    MH_WG_ISSUE(KS > 1 ? 1 : 0)

    // One 4-channel step = four groups of 16 transform positions (xi_z = g).  Groups 0-2 carry the commit of the next
    // step's staged data in their MFMA shadow; after the step's only barrier, group 3 carries the next patch / B-operand
    // reads and the issue of the loads two steps ahead.  The last step commits and re-reads redundantly (no branch: the
    // whole step is two scheduling regions).
#define MH_WG_GROUP(g, FILL)                                                                                       \
    {                                                                                                              \
        f32x2 tyv[4][2];                                                                                           \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                              \
            MH_WBT(tyv[0][h], tyv[1][h], tyv[2][h], tyv[3][h], tzv[0][h], tzv[1][h], tzv[2][h], tzv[3][h])          \
        _Pragma("unroll") for (int y = 0; y < 4; ++y) {                                                            \
            float v0, v1, v2, v3;                                                                                  \
            MH_WBT(v0, v1, v2, v3, tyv[y][0][0], tyv[y][0][1], tyv[y][1][0], tyv[y][1][1])                         \
            acc[(g) * 16 + y * 4 + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, ub[(g) & 1][y * 4 + 0], acc[(g) * 16 + y * 4 + 0], 0, 0, 0); \
            acc[(g) * 16 + y * 4 + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v1, ub[(g) & 1][y * 4 + 1], acc[(g) * 16 + y * 4 + 1], 0, 0, 0); \
            acc[(g) * 16 + y * 4 + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(v2, ub[(g) & 1][y * 4 + 2], acc[(g) * 16 + y * 4 + 2], 0, 0, 0); \
            acc[(g) * 16 + y * 4 + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v3, ub[(g) & 1][y * 4 + 3], acc[(g) * 16 + y * 4 + 3], 0, 0, 0); \
            FILL((g) * 4 + y)                                                                                      \
        }                                                                                                          \
    }
    // fillers, one per row of four MFMAs (row = 4 g + y): rows 0-11 commit the next step's data (two region slots per row,
    // the weight slab in the last four), rows 12-15 fetch the next patch / B operands and issue the loads of step s + 2
#define MH_WG_FILL_A(row) MH_WG_COMMIT_PART(an, (s + 1) & 1, 2 * (row) < 16 ? 2 * (row) : 16 + ((row) - 8), 2 * (row) < 16 ? 2 * (row) + 2 : ((row) == 8 ? 17 : 0), (row) - 8, (row) - 7)
#define MH_WG_FILL_B(row)                                                                                          \
    {                                                                                                              \
        MH_WG_READ_PATCH_PART((s + 1) & 1, (row) - 12, (row) - 11)                                                 \
        _Pragma("unroll") for (int i = 4 * ((row) - 12); i < 4 * ((row) - 11); ++i) ub[0][i] = un[i * 64];         \
        MH_WG_ISSUE_PART(sn2, 5 * ((row) - 12), 5 * ((row) - 11), (row) - 12, (row) - 11)                          \
    }
    for (int s = 0; s < KS; ++s) {
        const float* uc = us + (s & 1) * WG_UBUF + lane;
        const float* un = us + ((s + 1) & 1) * WG_UBUF + lane;
        const int sn = s + 1 < KS ? s + 1 : KS - 1, sn2 = s + 2 < KS ? s + 2 : KS - 1;
        const float4 an = load_nrm(in, n, sn * WG_KC + wave);
#pragma unroll
        for (int g = 0; g < 3; ++g) {
#pragma unroll
            for (int i = 0; i < 16; ++i) ub[(g + 1) & 1][i] = uc[((g + 1) * 16 + i) * 64];
            f32x2 tzv[4][2];
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const f32x2 d0 = raw[0][y][h], d1 = raw[1][y][h], d2 = raw[2][y][h];
                    tzv[y][h] = g == 0 ? d0 - d2 : g == 1 ? d1 + d2 : d2 - d1;
                }
            MH_WG_GROUP(g, MH_WG_FILL_A)
        }
        __syncthreads();
        {
            f32x2 tzv[4][2];
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int h = 0; h < 2; ++h) tzv[y][h] = raw[1][y][h] - raw[3][y][h];
            // the patch is dead from here on
            MH_WG_GROUP(3, MH_WG_FILL_B)
#pragma unroll
            for (int i = 0; i < 16; ++i) {   // spread the fetches over the 16 MFMA gaps of this region
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // two LDS reads
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   // two global loads
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);   // four VALU
            }
        }
    }
#undef MH_WG_FILL_A
#undef MH_WG_FILL_B
#undef MH_WG_GROUP
#undef MH_WG_ISSUE
#undef MH_WG_ISSUE_PART
#undef MH_WG_COMMIT
#undef MH_WG_COMMIT_PART
#undef MH_WG_READ_PATCH
#undef MH_WG_READ_PATCH_PART
    __syncthreads();   // every wave is done with the staging buffers: they become the output transpose area

    // ---- epilogue: inverse transform A^T (x, y, z), bias, statistics, transpose, store ---------------------------
    // The thread index takes a round trip through LDS so that the compiler cannot hoist the epilogue's index arithmetic
    // (~50 registers of addresses) above the matrix loop, where every register is spoken for.
    volatile int* slot = reinterpret_cast<volatile int*>(smem + WG_CN * WG_OS + 4 * WG_CN * 3) + tid;
    *slot = tid;
    const int etid = *slot;
    const int elane = etid & 63, ewave = etid >> 6, et16 = elane & 15, ekq = elane >> 4;
    const int co = cg * WG_CN + et16;
    const float bco = bias ? bias[co] : 0.0f;
    float* outs = smem + et16 * WG_OS;
    float cnt = 0.0f, sum = 0.0f;
    // the four accumulator registers of a lane (tiles 4 ekq + r, r = 0..3, cout et16) go through the transform together
    // as one register quad: every accumulator is read exactly once, whole
    f32x4 yq[8];                       // outputs (dz, dy, dx) of the four tiles
    {
        f32x4 q[4][2][2];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f32x4 p[4][2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 m0 = acc[a * 16 + c * 4 + 0], m1 = acc[a * 16 + c * 4 + 1], m2 = acc[a * 16 + c * 4 + 2], m3 = acc[a * 16 + c * 4 + 3];
                p[c][0] = (m0 + m1) + m2;
                p[c][1] = (m1 - m2) - m3;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                q[a][0][e] = (p[0][e] + p[1][e]) + p[2][e];
                q[a][1][e] = (p[1][e] - p[2][e]) - p[3][e];
            }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                yq[0 * 4 + f * 2 + e] = ((q[0][f][e] + q[1][f][e]) + q[2][f][e]) + bco;
                yq[1 * 4 + f * 2 + e] = ((q[1][f][e] - q[2][f][e]) - q[3][f][e]) + bco;
            }
    }
    float yv[4][8];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int tile = 4 * ekq + r;
        const int oy = 2 * (2 * (ewave & 1) + (tile >> 3)), ox = 2 * (tile & 7), oz = 2 * (ewave >> 1);
        const bool tok = z0 + oz < D && y0 + oy < H && x0 + ox < W;    // even dims: a tile is entirely inside or outside
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = yq[i][r];
            yv[r][i] = v;
            outs[((oz + (i >> 2)) * WG_OY + oy + ((i >> 1) & 1)) * WG_OX + ox + (i & 1)] = v;
            if (tok) { cnt += 1.0f; sum += v; }
        }
    }
    if (STATS) {
        Stat st;
        st.n = cnt;
        st.mean = cnt > 0.0f ? sum / cnt : 0.0f;
        st.m2 = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tile = 4 * ekq + r;
            const int oy = 2 * (2 * (ewave & 1) + (tile >> 3)), ox = 2 * (tile & 7), oz = 2 * (ewave >> 1);
            const bool tok = z0 + oz < D && y0 + oy < H && x0 + ox < W;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float d = yv[r][i] - st.mean;
                if (tok) st.m2 += d * d;
            }
        }
#pragma unroll
        for (int o = 16; o < 64; o <<= 1) {
            Stat ot;
            ot.n = __shfl_xor(st.n, o);
            ot.mean = __shfl_xor(st.mean, o);
            ot.m2 = __shfl_xor(st.m2, o);
            st = stat_merge(st, ot);
        }
        if (ekq == 0) {
            float* red = smem + WG_CN * WG_OS + (ewave * WG_CN + et16) * 3;
            red[0] = st.n; red[1] = st.mean; red[2] = st.m2;
        }
    }
    __syncthreads();
    if (STATS && etid < WG_CN) {
        const float* red = smem + WG_CN * WG_OS;
        Stat st;
        st.n = red[etid * 3]; st.mean = red[etid * 3 + 1]; st.m2 = red[etid * 3 + 2];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            Stat ot;
            ot.n = red[(w * WG_CN + etid) * 3]; ot.mean = red[(w * WG_CN + etid) * 3 + 1]; ot.m2 = red[(w * WG_CN + etid) * 3 + 2];
            st = stat_merge(st, ot);
        }
        float* rec = stats + (((long long)n * Cout + cg * WG_CN + etid) * nreg + b) * 3;
        rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
    }
    // row-contiguous stores: 16 couts x 4 x 8 rows of 16 floats
    float* dst = out.data + (long long)n * out.n_stride + (long long)cg * WG_CN * DHW;
    const bool vec_ok = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(out.data) & 15) == 0) && (out.n_stride % 4 == 0);
#pragma unroll
    for (int k = 0; k < (WG_CN * WG_OZ * WG_OY * WG_OX / 4) / 256; ++k) {
        const int f = etid + 256 * k;
        const int c = f / (WG_OZ * WG_OY * WG_OX / 4), rem = f - c * (WG_OZ * WG_OY * WG_OX / 4);
        const int oz = rem / (WG_OY * WG_OX / 4), r2 = rem - oz * (WG_OY * WG_OX / 4);
        const int oy = r2 / (WG_OX / 4), x4 = (r2 - oy * (WG_OX / 4)) * 4;
        const int z = z0 + oz, y = y0 + oy, x = x0 + x4;
        if (z >= D || y >= H || x >= W) continue;
        const f32x4 v = *reinterpret_cast<const f32x4*>(smem + c * WG_OS + (oz * WG_OY + oy) * WG_OX + x4);
        float* p = dst + (long long)c * DHW + ((long long)z * H + y) * W + x;
        if (vec_ok) *reinterpret_cast<f32x4*>(p) = v;    // W % 4 == 0: the four columns are inside together
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (x + e < W) p[e] = v[e];
        }
    }
}
#undef MH_WBT

// Weight transform U = G g G^T along z, y, x, written in the order operand B is read:
// up[cout group][cin step][xi][(cin & 3) * 16 + (cout & 15)].  One thread per (cout, cin).
__global__ void __launch_bounds__(256)
conv3d_k3_winograd_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ up) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    float g[3][3][3];
#pragma unroll
    for (int t = 0; t < 27; ++t) g[t / 9][(t / 3) % 3][t % 3] = w[((long long)co * Cin + ci) * 27 + t];
    float a[4][3][3], bq[4][4][3], u[4][4][4];
#pragma unroll
    for (int y = 0; y < 3; ++y)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            a[0][y][x] = g[0][y][x];
            a[1][y][x] = 0.5f * ((g[0][y][x] + g[1][y][x]) + g[2][y][x]);
            a[2][y][x] = 0.5f * ((g[0][y][x] - g[1][y][x]) + g[2][y][x]);
            a[3][y][x] = g[2][y][x];
        }
#pragma unroll
    for (int z = 0; z < 4; ++z)
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            bq[z][0][x] = a[z][0][x];
            bq[z][1][x] = 0.5f * ((a[z][0][x] + a[z][1][x]) + a[z][2][x]);
            bq[z][2][x] = 0.5f * ((a[z][0][x] - a[z][1][x]) + a[z][2][x]);
            bq[z][3][x] = a[z][2][x];
        }
#pragma unroll
    for (int z = 0; z < 4; ++z)
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            u[z][y][0] = bq[z][y][0];
            u[z][y][1] = 0.5f * ((bq[z][y][0] + bq[z][y][1]) + bq[z][y][2]);
            u[z][y][2] = 0.5f * ((bq[z][y][0] - bq[z][y][1]) + bq[z][y][2]);
            u[z][y][3] = bq[z][y][2];
        }
    const int KS = Cin / WG_KC;
    float* dst = up + (((long long)(co / WG_CN) * KS + ci / WG_KC) * 64) * 64 + (ci % WG_KC) * 16 + (co % WG_CN);
#pragma unroll
    for (int xi = 0; xi < 64; ++xi) dst[(long long)xi * 64] = u[xi / 16][(xi / 4) % 4][xi % 4];
}

}  // namespace mh
