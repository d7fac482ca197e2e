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

template <int MX_, int MY_, int MZ_, int MT_, int WM_, int WN_, int NT_, int CC_>
struct ConvCfg {
    static constexpr int MX = MX_, MY = MY_, MZ = MZ_, MT = MT_, WM = WM_, WN = WN_, NT = NT_, CC = CC_;
    static constexpr int TX = MX, TY = MY * MT, TZ = MZ * WM;
    static constexpr int TXH = TX + 2, TYH = TY + 2, TZH = TZ + 2;
    static constexpr int PLANE = TXH * TYH;
    static constexpr int CHS = PLANE * TZH;       // floats per input channel in the LDS halo tile
    static constexpr int CN = 32 * NT * WN;       // output channels per workgroup
    static constexpr int IN_FLOATS = CC * CHS;
    static constexpr int W_FLOATS = CC * 27 * CN;
    static constexpr int NRM_MAX = 512;           // max input channels (float4 each) kept in LDS
    static constexpr int SMEM_FLOATS = IN_FLOATS + W_FLOATS + 4 * NRM_MAX;
    static_assert(MX * MY * MZ == 32, "an M-tile is 32 voxels");
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(CC % 2 == 0, "two input channels per MFMA");
    static_assert(IN_FLOATS % 4 == 0 && W_FLOATS % 4 == 0, "16-byte carve alignment");
    static_assert(3 * WM * CN <= IN_FLOATS + W_FLOATS, "statistics scratch fits the tile area");
};

template <class Cfg, bool STATS>
__global__ void __launch_bounds__(256, 2)  // 2 waves/SIMD: two workgroups per CU (LDS-limited), <= 256 registers
conv3d_k3_mfma_kernel(Tensor in, const float* __restrict__ wp, const float* __restrict__ bias, Tensor out,
                      float* __restrict__ stats, int tiles_x, int tiles_y, int tiles_z) {
    constexpr int MX = Cfg::MX, MY = Cfg::MY, MZ = Cfg::MZ, MT = Cfg::MT, WM = Cfg::WM, NT = Cfg::NT;
    constexpr int CC = Cfg::CC, TXH = Cfg::TXH, PLANE = Cfg::PLANE, CHS = Cfg::CHS, CN = Cfg::CN;

    __shared__ __attribute__((aligned(16))) float smem[Cfg::SMEM_FLOATS];
    float* xs = smem;
    float* ws = smem + Cfg::IN_FLOATS;
    float4* nrm_s = reinterpret_cast<float4*>(smem + Cfg::IN_FLOATS + Cfg::W_FLOATS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const int kh = lane >> 5, li = lane & 31;
    const int Cin = in.C, Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long DHW = (long long)D * H * W;

    const unsigned ntiles = gridDim.x;
    const unsigned b = xcd_remap(blockIdx.x, ntiles);
    const int tx0 = (int)(b % tiles_x) * Cfg::TX;
    const int ty0 = (int)((b / tiles_x) % tiles_y) * Cfg::TY;
    const int tz0 = (int)(b / (tiles_x * tiles_y)) * Cfg::TZ;
    const int ct = blockIdx.y, n = blockIdx.z;

    for (int c = tid; c < Cin; c += 256) nrm_s[c] = load_nrm(in, n, c);

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

    for (int c0 = 0; c0 < Cin; c0 += CC) {
        __syncthreads();  // previous chunk fully consumed (first pass: nrm_s visible)
        {   // weight slab: CC*27*CN contiguous floats of the packed tensor [ct][cin][27][CN]
            const float4* wsrc = reinterpret_cast<const float4*>(wp + ((long long)ct * Cin + c0) * 27 * CN);
            float4* wdst = reinterpret_cast<float4*>(ws);
            for (int i = tid; i < Cfg::W_FLOATS / 4; i += 256) wdst[i] = wsrc[i];
        }
        for (int i = tid; i < Cfg::IN_FLOATS; i += 256) {   // halo tile, normalise + activate on load
            const int c = i / CHS, r = i - c * CHS;
            const int lz = r / PLANE, r2 = r - lz * PLANE;
            const int ly = r2 / TXH, lx = r2 - ly * TXH;
            const int gz = tz0 + lz - 1, gy = ty0 + ly - 1, gx = tx0 + lx - 1;
            float v = 0.0f;
            if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const float4 a = nrm_s[c0 + c];
                v = act(src[(long long)(c0 + c) * DHW + ((long long)gz * H + gy) * W + gx], a.x, a.y, a.z);
            }
            xs[i] = v;
        }
        __syncthreads();

#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int dz = t / 9, dy = (t / 3) % 3, dx = t % 3;
#pragma unroll
            for (int p = 0; p < CC / 2; ++p) {
                float bv[NT], av[MT];
#pragma unroll
                for (int q = 0; q < NT; ++q) bv[q] = ws[bbase + (2 * p) * 27 * CN + t * CN + q * 32];
#pragma unroll
                for (int m = 0; m < MT; ++m) av[m] = xs[abase + (2 * p) * CHS + dz * PLANE + (m * MY + dy) * TXH + dx];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bv[q], acc[m][q], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias, store, fused InstanceNorm statistics --------------------------------------
    float* dst = out.data + (long long)n * out.n_stride;
    const bool vec_ok = (MX % 4 == 0) && (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(out.data) & 15) == 0) &&
                        (out.n_stride % 4 == 0);
    if (STATS) __syncthreads();  // all waves done with xs/ws before it is reused as statistics scratch
#pragma unroll
    for (int q = 0; q < NT; ++q) {
        const int col = (wn * NT + q) * 32 + li;   // cout within the workgroup's CN
        const int co = ct * CN + col;
        const float bco = bias ? bias[co] : 0.0f;
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
                    ok[j] = z < D && y < H && x < W;
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
                    const bool ok = (tz0 + wm * MZ + jz) < D && (ty0 + m * MY + jy) < H && (tx0 + jx) < W;
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
        if (tid < CN) {
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

// Repack torch conv weights [Cout][Cin][27] into [Cout/CN][Cin][27][CN] (the per-chunk LDS slab becomes one
// contiguous run).  CN = Cout gives the layout of the direct kernel.
__global__ void __launch_bounds__(256)
conv3d_k3_pack_kernel(const float* __restrict__ w, int Cin, int Cout, int CN, float* __restrict__ packed) {
    const long long total = (long long)Cout * Cin * 27;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int col = (int)(idx % CN);
    long long t = idx / CN;
    const int tap = (int)(t % 27); t /= 27;
    const int ci = (int)(t % Cin);
    const int ct = (int)(t / Cin);
    const int co = ct * CN + col;
    packed[idx] = w[((long long)co * Cin + ci) * 27 + tap];
}

}  // namespace mh
