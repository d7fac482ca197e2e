// Conv3d 3x3x3 / stride 1 / zero padding 1 as an implicit GEMM on the fp32 matrix cores of gfx950.
//
// Reference op: nn.Conv3d inside monai/networks/blocks/convolutions.py:98-171 (`Convolution`), fed by the
// previous block's InstanceNorm+LeakyReLU (acti_norm.py:69-101), which is applied HERE while the input
// halo tile is staged into LDS ("normalise on load").
//
// Mapping.  GEMM M = output voxels, N = output channels, K = (cin, tap).  v_mfma_f32_32x32x2_f32 is an
// exact-fp32 instruction (bitwise a k-ordered fmaf chain) at the fp32 vector rate, so it keeps the fp32
// numerics the 1e-4 parity bound needs while reaching the 157 TF fp32 peak from one wave per SIMD.
//   * an M-tile is 32 voxels shaped MX x MY x MZ (x fastest); operand A lane l holds voxel i = l & 31 of
//     input channel (2p + (l >> 5)) -- both k-slices of the instruction are two consecutive input channels
//     at the same tap, so lanes 0-31 and 32-63 read two x-contiguous runs of the LDS tile (conflict free);
//   * operand B lane l holds weight[cin 2p + (l >> 5)][tap][cout l & 31]: 64 consecutive floats of the
//     LDS weight slab (conflict free);
//   * a wave owns MT M-tiles stacked along y and NT N-tiles; a workgroup is 4 waves arranged WM (along z)
//     x WN (along cout).  Input channels are streamed in chunks of CC through LDS: halo tile
//     [CC][TZ+2][TY+2][TX+2] + weight slab [CC][27][CN].
//   * accumulator D: lane l holds cout (l & 31) for voxels (r&3) + 8(r>>2) + 4(l>>5), r = 0..15: one
//     output channel per lane, so the InstanceNorm statistics of the written values reduce in-lane
//     (then one lane-pair merge, one LDS merge over the WM waves) and come out of the epilogue for free.
#pragma once
#include "common.h"

namespace mh {

template <int MX_, int MY_, int MZ_, int MT_, int WM_, int WN_, int NT_, int CC_, int OCC_ = 2>
struct ConvCfg {
    static constexpr int MX = MX_, MY = MY_, MZ = MZ_, MT = MT_, WM = WM_, WN = WN_, NT = NT_, CC = CC_, OCC = OCC_;
    static constexpr int TX = MX, TY = MY * MT, TZ = MZ * WM;
    static constexpr int TXH = TX + 2, TYH = TY + 2, TZH = TZ + 2;
    static constexpr int PLANE = TXH * TYH;
    static constexpr int CHS = PLANE * TZH;       // floats per input channel in the LDS halo tile
    static constexpr int CN = 32 * NT * WN;       // output channels per workgroup
    static constexpr int IN_FLOATS = CC * CHS;
    static constexpr int W_FLOATS = CC * 27 * CN;
    static constexpr int SLOTS = (CHS + 255) / 256;          // halo-tile elements per thread per channel
    static constexpr int WV4 = (W_FLOATS / 4 + 255) / 256;   // weight float4s per thread per chunk
    static constexpr int NRM_MAX = 768;           // max input channels ({alpha, beta, slope}: 12 bytes each) kept in LDS
    static constexpr int SMEM_FLOATS = IN_FLOATS + W_FLOATS + 3 * NRM_MAX;
    static_assert(MX * MY * MZ == 32, "an M-tile is 32 voxels");
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(CC % 2 == 0, "two input channels per MFMA");
    static_assert(IN_FLOATS % 4 == 0 && W_FLOATS % 4 == 0, "16-byte carve alignment");
    static_assert(3 * WM * CN <= IN_FLOATS + W_FLOATS, "statistics scratch fits the tile area");
};

// One software-pipelined step of the matrix loop (compile-time step index S): consume the operands fetched by the
// previous step, request the next step's operands, issue the MFMAs, fence the scheduler.
template <class Cfg, int S, int NSTEP>
__device__ __forceinline__ void mfma_steps(f32x16 (&acc)[Cfg::MT][Cfg::NT], float (&an)[Cfg::MT], float (&bn)[Cfg::NT],
                                           const float* xs, const float* ws, int abase, int bbase) {
    if constexpr (S < NSTEP) {
        constexpr int MT = Cfg::MT, NT = Cfg::NT, CC = Cfg::CC, CN = Cfg::CN;
        float av[MT], bv[NT];
#pragma unroll
        for (int m = 0; m < MT; ++m) av[m] = an[m];
#pragma unroll
        for (int q = 0; q < NT; ++q) bv[q] = bn[q];
        if constexpr (S + 1 < NSTEP) {
            constexpr int t = (S + 1) / (CC / 2), p = (S + 1) % (CC / 2);
            constexpr int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
#pragma unroll
            for (int q = 0; q < NT; ++q) bn[q] = ws[bbase + (2 * p) * 27 * CN + t * CN + q * 32];
#pragma unroll
            for (int m = 0; m < MT; ++m)
                an[m] = xs[abase + (2 * p) * Cfg::CHS + dz * Cfg::PLANE + (m * Cfg::MY + dy) * Cfg::TXH + dx];
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < NT; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[q], acc[m][q], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_steps<Cfg, S + 1, NSTEP>(acc, an, bn, xs, ws, abase, bbase);
    }
}

// Staging pipeline (per input-channel chunk):   loads(c+1) -> registers are issued BEFORE the MFMA loop of
// chunk c, so their HBM/L2 latency hides under ~10 us of matrix work; after the loop one barrier, then the
// registers are normalised/activated and written to LDS (a few hundred VALU/DS instructions), one barrier,
// next chunk.  Each thread owns the same SLOTS positions of the halo tile in every channel, so the
// global offsets / bounds of its positions are computed once per workgroup.
template <class Cfg, bool STATS>
__global__ void __launch_bounds__(256, Cfg::OCC)
conv3d_k3_mfma_kernel(Tensor in, const float* __restrict__ wp, const float* __restrict__ bias, Tensor out,
                      float* __restrict__ stats, int tiles_x, int tiles_y, int tiles_z) {
    constexpr int MX = Cfg::MX, MY = Cfg::MY, MZ = Cfg::MZ, MT = Cfg::MT, WM = Cfg::WM, NT = Cfg::NT;
    constexpr int CC = Cfg::CC, TXH = Cfg::TXH, PLANE = Cfg::PLANE, CHS = Cfg::CHS, CN = Cfg::CN;
    constexpr int SLOTS = Cfg::SLOTS, WV4 = Cfg::WV4;

    __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM_FLOATS];
    float* xs = smem;
    float* ws = smem + Cfg::IN_FLOATS;
    float* nrm_s = smem + Cfg::IN_FLOATS + Cfg::W_FLOATS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int kh = lane >> 5, li = lane & 31;
    const int Cin = in.C, Cout = out.C, D = out.D, H = out.H, W = out.W;
    const int CinP = (Cin + CC - 1) / CC * CC;    // the packed weights are zero-padded to a multiple of CC
    const long long DHW = (long long)D * H * W;

    const unsigned ntiles = gridDim.x;
    const unsigned b = xcd_remap(blockIdx.x, ntiles);
    const int tx0 = (int)(b % tiles_x) * Cfg::TX;
    const int ty0 = (int)((b / tiles_x) % tiles_y) * Cfg::TY;
    const int tz0 = (int)(b / (tiles_x * tiles_y)) * Cfg::TZ;
    const int ct = blockIdx.y, n = blockIdx.z;

    for (int c = tid; c < CinP; c += 256) {
        const float4 a = c < Cin ? load_nrm(in, n, c) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        nrm_s[3 * c] = a.x; nrm_s[3 * c + 1] = a.y; nrm_s[3 * c + 2] = a.z;
    }

    // this thread's positions in the halo tile: global offset inside a channel plane, and validity
    int soff[SLOTS];
    bool sok[SLOTS];
#pragma unroll
    for (int j = 0; j < SLOTS; ++j) {
        const int r = tid + 256 * j;
        const int lz = r / PLANE, r2 = r - lz * PLANE;
        const int ly = r2 / TXH, lx = r2 - ly * TXH;
        const int gz = tz0 + lz - 1, gy = ty0 + ly - 1, gx = tx0 + lx - 1;
        sok[j] = r < CHS && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        soff[j] = sok[j] ? (gz * H + gy) * W + gx : 0;
    }

    // per-lane LDS bases (floats)
    const int ix = li % MX, iy = (li / MX) % MY, iz = li / (MX * MY);
    const int abase = kh * CHS + (wm * MZ + iz) * PLANE + iy * TXH + ix;
    const int bbase = kh * 27 * CN + wn * NT * 32 + li;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.0f;

    const float* src = in.data + (long long)n * in.n_stride;
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wp + (long long)ct * CinP * 27 * CN);

    float xin[CC][SLOTS];
    f32x4 win[WV4];
    // unconditional loads from clamped addresses (no control flow: they all issue back to back)
#define MH_ISSUE_LOADS(C0)                                                                                 \
    {                                                                                                      \
        _Pragma("unroll") for (int c = 0; c < CC; ++c) {                                                   \
            const int cg = (C0) + c < Cin ? (C0) + c : Cin - 1;                                            \
            const float* plane = src + (long long)cg * DHW;                                                \
            _Pragma("unroll") for (int j = 0; j < SLOTS; ++j) xin[c][j] = plane[soff[j]];                  \
        }                                                                                                  \
        const f32x4* wq = wsrc + (long long)(C0) * 27 * CN / 4;                                            \
        _Pragma("unroll") for (int k = 0; k < WV4; ++k) {                                                  \
            const int i = tid + 256 * k;                                                                   \
            win[k] = wq[i < Cfg::W_FLOATS / 4 ? i : 0];                                                    \
        }                                                                                                  \
    }

    MH_ISSUE_LOADS(0)
    __syncthreads();  // nrm_s visible

    for (int c0 = 0; c0 < CinP; c0 += CC) {
        // registers -> LDS: halo tile with the producer's InstanceNorm + LeakyReLU applied, zero outside the
        // volume (the conv's zero padding acts on the ACTIVATED tensor) and for padded channels
#pragma unroll
        for (int c = 0; c < CC; ++c) {
            const float a_x = nrm_s[3 * (c0 + c)], a_y = nrm_s[3 * (c0 + c) + 1], a_z = nrm_s[3 * (c0 + c) + 2];
            const bool cok = c0 + c < Cin;
#pragma unroll
            for (int j = 0; j < SLOTS; ++j) {
                const int r = tid + 256 * j;
                if (r < CHS) xs[c * CHS + r] = (cok && sok[j]) ? act(xin[c][j], a_x, a_y, a_z) : 0.0f;
            }
        }
#pragma unroll
        for (int k = 0; k < WV4; ++k) {
            const int i = tid + 256 * k;
            if (i < Cfg::W_FLOATS / 4) reinterpret_cast<f32x4*>(ws)[i] = win[k];
        }
        __syncthreads();
        if (c0 + CC < CinP) MH_ISSUE_LOADS(c0 + CC)   // in flight during the matrix loop below

        // matrix loop, software-pipelined by hand: the LDS operands of step s+1 are requested before the MFMAs of
        // step s issue, and sched_barrier(0) keeps the compiler from hoisting further ahead (it otherwise
        // prefetches dozens of steps and spills).  One step = one tap x one channel pair = MT*NT MFMAs.
        constexpr int NSTEP = 27 * (CC / 2);
        float an[MT], bn[NT];
#define MH_LOAD_STEP(S)                                                                                          \
    {                                                                                                            \
        constexpr int t_ = (S) / (CC / 2), p_ = (S) % (CC / 2);                                                  \
        constexpr int dz_ = t_ / 9, dy_ = (t_ / 3) % 3, dx_ = t_ % 3;                                            \
        _Pragma("unroll") for (int q = 0; q < NT; ++q) bn[q] = ws[bbase + (2 * p_) * 27 * CN + t_ * CN + q * 32]; \
        _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                           \
            an[m] = xs[abase + (2 * p_) * CHS + dz_ * PLANE + (m * MY + dy_) * TXH + dx_];                       \
    }
        MH_LOAD_STEP(0)
        mfma_steps<Cfg, 0, NSTEP>(acc, an, bn, xs, ws, abase, bbase);
#undef MH_LOAD_STEP
        __syncthreads();  // every wave is done reading this chunk before LDS is overwritten
    }
#undef MH_ISSUE_LOADS

    // ---- epilogue: bias, store, fused InstanceNorm statistics --------------------------------------
    float* dst = out.data + (long long)n * out.n_stride;
    const bool vec_ok = (MX % 4 == 0) && (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(out.data) & 15) == 0) &&
                        (out.n_stride % 4 == 0);
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int col = (wn * NT + q) * 32 + li;   // cout within the workgroup's CN
        const int co = ct * CN + col;
        const bool cvalid = co < Cout;             // Cout is zero-padded up to a multiple of CN in the packed weights
        const float bco = (bias && cvalid) ? bias[co] : 0.0f;
        float cnt = 0.0f, sum = 0.0f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v[4];
                bool ok[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = g4 * 4 + j;
                    const int row = j + 8 * g4 + 4 * kh;     // (r&3) + 8*(r>>2) + 4*(lane>>5)
                    const int jx = row % MX, jy = (row / MX) % MY, jz = row / (MX * MY);
                    const int z = tz0 + wm * MZ + jz, y = ty0 + m * MY + jy, x = tx0 + jx;
                    ok[j] = cvalid && z < D && y < H && x < W;
                    v[j] = acc[m][q][r] + bco;
                    acc[m][q][r] = v[j];
                    if (ok[j]) { cnt += 1.0f; sum += v[j]; }
                }
                const int row0 = 8 * g4 + 4 * kh;
                const int x0 = tx0 + row0 % MX, y0 = ty0 + m * MY + (row0 / MX) % MY, z0 = tz0 + wm * MZ + row0 / (MX * MY);
                float* p = dst + (long long)co * DHW + ((long long)z0 * H + y0) * W + x0;
                if (vec_ok) {
                    if (ok[0]) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (ok[j]) {
                            const int row = j + 8 * g4 + 4 * kh;
                            const int jx = row % MX, jy = (row / MX) % MY, jz = row / (MX * MY);
                            dst[(long long)co * DHW + ((long long)(tz0 + wm * MZ + jz) * H + (ty0 + m * MY + jy)) * W + tx0 + jx] = v[j];
                        }
                    }
                }
            }
        }
        if (STATS) {
            Stat s;
            s.n = cnt;
            s.mean = cnt > 0.0f ? sum / cnt : 0.0f;
            s.m2 = 0.0f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                    const int jx = row % MX, jy = (row / MX) % MY, jz = row / (MX * MY);
                    const bool ok = cvalid && (tz0 + wm * MZ + jz) < D && (ty0 + m * MY + jy) < H && (tx0 + jx) < W;
                    const float d = acc[m][q][r] - s.mean;
                    if (ok) s.m2 += d * d;
                }
            Stat o;
            o.n = __shfl_xor(s.n, 32);
            o.mean = __shfl_xor(s.mean, 32);
            o.m2 = __shfl_xor(s.m2, 32);
            s = stat_merge(s, o);
            if (kh == 0) {
                float* red = smem + (wm * CN + col) * 3;
                red[0] = s.n; red[1] = s.mean; red[2] = s.m2;
            }
        }
    }
    if (STATS) {
        __syncthreads();
        if (tid < CN && ct * CN + tid < Cout) {
            Stat s;
            s.n = smem[tid * 3]; s.mean = smem[tid * 3 + 1]; s.m2 = smem[tid * 3 + 2];
#pragma unroll
            for (int w = 1; w < WM; ++w) {
                Stat o;
                o.n = smem[(w * CN + tid) * 3]; o.mean = smem[(w * CN + tid) * 3 + 1]; o.m2 = smem[(w * CN + tid) * 3 + 2];
                s = stat_merge(s, o);
            }
            float* rec = stats + (((long long)n * Cout + ct * CN + tid) * ntiles + b) * 3;
            rec[0] = s.n; rec[1] = s.mean; rec[2] = s.m2;
        }
    }
}

// Repack torch conv weights [Cout][Cin][27] into [CoutP/CN][CinP][27][CN] (the per-chunk LDS slab becomes one
// contiguous run); CinP >= Cin and CoutP >= Cout pad with zeros.  CN = Cout is the direct kernel's layout.
__global__ void __launch_bounds__(256)
conv3d_k3_pack_kernel(const float* __restrict__ w, int Cin, int CinP, int Cout, int CoutP, int CN, float* __restrict__ packed) {
    const long long total = (long long)CoutP * CinP * 27;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int col = (int)(idx % CN);
    long long t = idx / CN;
    const int tap = (int)(t % 27); t /= 27;
    const int ci = (int)(t % CinP);
    const int ct = (int)(t / CinP);
    const int co = ct * CN + col;
    packed[idx] = (ci < Cin && co < Cout) ? w[((long long)co * Cin + ci) * 27 + tap] : 0.0f;
}

}  // namespace mh
