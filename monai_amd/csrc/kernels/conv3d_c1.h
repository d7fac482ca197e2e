// Conv3d 3x3x3 / stride 1 / zero padding 1 for ONE input channel (the first layer of every network on this path:
// BasicUNet conv_0.conv_0, monai/networks/nets/basic_unet.py:206 -> Convolution, blocks/convolutions.py:98-171), packed fp32 VALU.
//
// Why not the matrix cores: with Cin = 1 the GEMM K dimension is the 27 taps.  The fp32 MFMA tile (conv3d_mfma.h, ConvCfg<32,1,1,4,4,1,1,2,4>)
// pads the channel pair of v_mfma_f32_32x32x2_f32 with a zero channel -- half of its cycles multiply zeros -- and ran at 2.8 ms per
// 64 windows of 96^3 with 32 output channels (profiles/r02_bench_kernel_trace_stats_v3.txt), against 1.2 ms for writing the 7.25 GB
// result.  The layer is 97.8 GFLOP per launch: 0.62 ms at the 157 TF of v_pk_fma_f32, so a VALU kernel is write-bound.
//
// Mapping.  One thread owns 4 x-consecutive outputs x COT output channels and marches along z; a wave is 8 (x) x 8 (y) threads =
// a 32 x 8 output tile (every store instruction writes eight complete 128-byte lines per cout), a workgroup four such tiles
// stacked along y.  The three input planes around the current output plane live in registers as x-PAIRS (i0,i1)(i2,i3)(i4,i5) of the
// 6-wide row segment a thread needs; the odd pairs (i1,i2)(i3,i4) of the kx = 1 taps are formed per row while it is used.  The
// inner statement is   acc(x, x+1)[cout] = fma(in(x + kx, x + kx + 1), w[tap][cout], acc)   =   v_pk_fma_f32 v, v, s(broadcast):
// the weights are uniform -> scalar registers, fetched by s_load_dwordx16 from the constant address space one tap ahead (an opaque
// offset keeps the compiler from hoisting all 27 x COT of them out of the plane loop and spilling).  The next input plane is
// prefetched into raw registers during the FMAs of the current one.
//
// Statistics (InstanceNorm of the output, the epilogue contract of conv3d_mfma.h): every lane accumulates sum(d) and sum(d^2) of
// d = value - pivot with ONE pivot per (wave, cout) -- the first value lane 0 computes, so d is a local difference and
// sum(d^2) - sum(d)^2 / n does not cancel (a plain sum / sum of squares would for inputs with mean^2 >> variance) -- the lanes add up by
// shuffles, the four waves merge as {n, mean, M2} records (Chan), one record per (n, cout, workgroup).
//
// Arithmetic: exact fp32, one fused multiply-add per tap in tap order (kz, ky, kx ascending) starting from the bias -- within the
// rounding class of the fp32 tiles (different summation order); tests/kernel_cases.py::case_conv3d bounds it against fp64.
#pragma once
#include "common.h"

namespace mh {

#ifdef MH_SIMT_EMULATOR
typedef const float c1_wfloat;
#define MH_OPAQUE_S(x) ((void)0)
#else
typedef const __attribute__((address_space(4))) float c1_wfloat;      // constant address space: uniform loads are scalar loads
#define MH_OPAQUE_S(x) asm volatile("" : "+s"(x))
#endif

constexpr int C1_TX = 32, C1_TY = 32;      // workgroup tile: 8 lanes x 4 voxels, 4 waves x 8 rows

template <int COT, bool STATS, bool NRM>
__global__ void __launch_bounds__(256, 2)
conv3d_k3_c1_kernel(Tensor in, const float* __restrict__ wp, const float* __restrict__ bias, Tensor out, float* __restrict__ stats,
                    int txn, int tyn, int zchunk) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lx = lane & 7, ly = lane >> 3;
    const int Cout = out.C, D = out.D, H = out.H, W = out.W;
    const long long HW = (long long)H * W, DHW = (long long)D * HW;
    const unsigned b = blockIdx.x;
    const int tx = (int)(b % txn), ty = (int)((b / txn) % tyn), tz = (int)(b / (txn * tyn));
    const int cg = blockIdx.y, n = blockIdx.z;
    const int x0 = tx * C1_TX + 4 * lx, y = ty * C1_TY + 8 * wave + ly;
    const int zs = tz * zchunk, ze = min(zs + zchunk, D);
    const bool valid = x0 < W && y < H;            // W % 4 == 0: a thread's four outputs are inside or outside together

    // thread-constant validity of the 3 rows and of the left / right neighbour; clamped row offsets (always readable)
    bool vrow[3];
    long long roff[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        vrow[r] = valid && yy >= 0 && yy < H;
        roff[r] = vrow[r] ? (long long)yy * W + x0 : 0;
    }
    const bool vl = x0 > 0, vr = x0 + 4 < W;
    const float* const src = in.data + (long long)n * in.n_stride;
    const float4 na = NRM ? load_nrm(in, n, 0) : make_float4(1.0f, 0.0f, 1.0f, 0.0f);

    // raw registers of one input plane: 3 rows x {left, 4 values, right}
    f32x4 rm[3];
    float rl[3], rr[3];
#define MH_C1_LOAD(ZZ)                                                                               \
    {                                                                                                \
        const int zc_ = min(max((ZZ), 0), D - 1);                                                    \
        const float* pl_ = src + (long long)zc_ * HW;                                                \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                              \
            const float* row_ = pl_ + roff[r];                                                       \
            rm[r] = *reinterpret_cast<const f32x4*>(row_);                                           \
            rl[r] = row_[vrow[r] && vl ? -1 : 0];                                                    \
            rr[r] = row_[vrow[r] && vr ? 4 : 0];                                                     \
        }                                                                                            \
    }
#define MH_C1_ACT(V) (NRM ? act((V), na.x, na.y, na.z) : (V))
    // ring[p][r][k]: plane p (0: z-1, 1: z, 2: z+1), row r, even pair k = (i_2k, i_2k+1) of the activated, zero-padded row segment
    f32x2 ring[3][3][3];
#define MH_C1_CONVERT(P, ZZ)                                                                         \
    {                                                                                                \
        const bool vz_ = (ZZ) >= 0 && (ZZ) < D;                                                      \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                              \
            const bool ok_ = vz_ && vrow[r];                                                         \
            const float e0_ = ok_ && vl ? MH_C1_ACT(rl[r]) : 0.0f;                                   \
            const float e5_ = ok_ && vr ? MH_C1_ACT(rr[r]) : 0.0f;                                   \
            float m_[4];                                                                             \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) m_[i] = ok_ ? MH_C1_ACT(rm[r][i]) : 0.0f;  \
            ring[P][r][0] = f32x2{e0_, m_[0]};                                                       \
            ring[P][r][1] = f32x2{m_[1], m_[2]};                                                     \
            ring[P][r][2] = f32x2{m_[3], e5_};                                                       \
        }                                                                                            \
    }

    // prologue: planes zs - 1 and zs converted, plane zs + 1 in flight
    MH_C1_LOAD(zs - 1)
    MH_C1_CONVERT(1, zs - 1)
    MH_C1_LOAD(zs)
    MH_C1_CONVERT(2, zs)
    MH_C1_LOAD(zs + 1)

    const int co0 = cg * COT;
    float bco[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) bco[j] = bias ? bias[co0 + j] : 0.0f;
    float* const obase = out.data + (long long)n * out.n_stride + (long long)co0 * DHW + (long long)y * W + x0;

    float ssum[COT], ssq[COT], piv[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) { ssum[j] = 0.0f; ssq[j] = 0.0f; piv[j] = 0.0f; }

    c1_wfloat* const wbase = (c1_wfloat*)wp + co0;
    for (int z = zs; z < ze; ++z) {
        // rotate the ring, bring in plane z + 1 (loaded during the previous iteration), start the loads of plane z + 2
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) { ring[0][r][k] = ring[1][r][k]; ring[1][r][k] = ring[2][r][k]; }
        MH_C1_CONVERT(2, z + 1)
        MH_C1_LOAD(z + 2)

        f32x2 acc[COT][2];
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[j][0] = acc[j][1] = f32x2{bco[j], bco[j]};

        int woff = 0;
        MH_OPAQUE_S(woff);                     // the weight addresses depend on it: their loads stay inside this iteration
        c1_wfloat* wt = wbase + woff;
        float wn[COT];
#pragma unroll
        for (int j = 0; j < COT; ++j) wn[j] = wt[j];
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const int a = t / 9, r = (t / 3) % 3, kx = t % 3;
            float wc[COT];
#pragma unroll
            for (int j = 0; j < COT; ++j) wc[j] = wn[j];
            if (t + 1 < 27) {
#pragma unroll
                for (int j = 0; j < COT; ++j) wn[j] = wt[(t + 1) * Cout + j];
            }
            const f32x2 e0 = ring[a][r][0], e1 = ring[a][r][1], e2 = ring[a][r][2];
            const f32x2 i0 = kx == 0 ? e0 : kx == 1 ? f32x2{e0[1], e1[0]} : e1;
            const f32x2 i1 = kx == 0 ? e1 : kx == 1 ? f32x2{e1[1], e2[0]} : e2;
#pragma unroll
            for (int j = 0; j < COT; ++j) {
                const f32x2 ww = {wc[j], wc[j]};
                acc[j][0] = __builtin_elementwise_fma(i0, ww, acc[j][0]);
                acc[j][1] = __builtin_elementwise_fma(i1, ww, acc[j][1]);
            }
            __builtin_amdgcn_sched_barrier(0);     // one tap per scheduling region: the next tap's scalar loads stay one tap ahead, no further
        }

        if (valid) {
            float* op = obase + (long long)z * HW;
#pragma unroll
            for (int j = 0; j < COT; ++j)
                *reinterpret_cast<f32x4*>(op + (long long)j * DHW) = f32x4{acc[j][0][0], acc[j][0][1], acc[j][1][0], acc[j][1][1]};
        }
        if (STATS) {
            if (z == zs) {
#pragma unroll
                for (int j = 0; j < COT; ++j) piv[j] = __builtin_amdgcn_readfirstlane(__shfl(acc[j][0][0], 0));
            }
#pragma unroll
            for (int j = 0; j < COT; ++j) {
                const f32x2 pv = {piv[j], piv[j]};
                const f32x2 d0 = acc[j][0] - pv, d1 = acc[j][1] - pv;
                const f32x2 t = d0 + d1, u = __builtin_elementwise_fma(d1, d1, d0 * d0);
                ssum[j] += t[0] + t[1];
                ssq[j] += u[0] + u[1];
            }
        }
    }
#undef MH_C1_CONVERT
#undef MH_C1_ACT
#undef MH_C1_LOAD

    if (STATS) {
        __shared__ float red[4][16][3];
        float cnt = valid ? 4.0f * (float)(ze - zs) : 0.0f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            float s = valid ? ssum[j] : 0.0f;
            float q = valid ? ssq[j] : 0.0f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
            if (lane == 0) {
                const float mu = cnt > 0.0f ? s / cnt : 0.0f;
                red[wave][j][0] = cnt;
                red[wave][j][1] = cnt > 0.0f ? piv[j] + mu : 0.0f;
                red[wave][j][2] = cnt > 0.0f ? fmaxf(q - s * mu, 0.0f) : 0.0f;
            }
        }
        __syncthreads();
        if (tid < COT) {
            Stat st;
            st.n = red[0][tid][0]; st.mean = red[0][tid][1]; st.m2 = red[0][tid][2];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                Stat ot;
                ot.n = red[w][tid][0]; ot.mean = red[w][tid][1]; ot.m2 = red[w][tid][2];
                st = stat_merge(st, ot);
            }
            float* rec = stats + (((long long)n * Cout + co0 + tid) * gridDim.x + b) * 3;
            rec[0] = st.n; rec[1] = st.mean; rec[2] = st.m2;
        }
    }
}

}  // namespace mh
