// HBM-bound network kernels of the BasicUNet path (everything except the 3x3x3 implicit-GEMM conv):
// direct conv (first layer / odd channel counts), max-pool, transposed conv k2s2, 1x1 conv, and the
// InstanceNorm statistics.  All of them apply the producer's deferred InstanceNorm+LeakyReLU on load
// (`act`, common.h) and stream W-contiguous rows, one or four voxels per lane.
#pragma once
#include "common.h"

namespace mh {

// ---------------------------------------------------------------------------------------------------
// Conv3d 3x3x3 pad 1, direct form.  One thread = one output voxel x COT output channels; weights come from
// the packed layout [Cin][27][Cout] (block-uniform addresses -> scalar loads).  Used for Cin = 1 (first
// layer: 27 taps, HBM-write bound) and as the general path for channel counts the MFMA tiles do not take.
template <int COT>
__global__ void __launch_bounds__(256)
conv3d_k3_direct_kernel(Tensor in, const float* __restrict__ wp, const float* __restrict__ bias, Tensor out) {
    const int D = out.D, H = out.H, W = out.W, Cin = in.C, Cout = out.C;
    const long long DHW = (long long)D * H * W;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * COT, n = blockIdx.z;
    if (idx >= DHW) return;
    const int x = (int)(idx % W);
    const long long t = idx / W;
    const int y = (int)(t % H), z = (int)(t / H);

    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.0f;

    const float* src = in.data + (long long)n * in.n_stride;
    for (int ci = 0; ci < Cin; ++ci) {
        const float4 a = load_nrm(in, n, ci);
        const float* plane = src + (long long)ci * DHW;
        const float* wrow = wp + (long long)ci * 27 * Cout + co0;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int zz = z + tap / 9 - 1, yy = y + (tap / 3) % 3 - 1, xx = x + tap % 3 - 1;
            float v = 0.0f;
            if (zz >= 0 && zz < D && yy >= 0 && yy < H && xx >= 0 && xx < W)
                v = act(plane[((long long)zz * H + yy) * W + xx], a.x, a.y, a.z);
#pragma unroll
            for (int j = 0; j < COT; ++j)
                if (co0 + j < Cout) acc[j] = fmaf(v, wrow[tap * Cout + j], acc[j]);
        }
    }
    float* dst = out.data + (long long)n * out.n_stride + idx;
#pragma unroll
    for (int j = 0; j < COT; ++j)
        if (co0 + j < Cout) dst[(long long)(co0 + j) * DHW] = acc[j];
}

// ---------------------------------------------------------------------------------------------------
// Strided Conv3d 3x3x3 pad 1 (UNet's down path: monai/networks/nets/unet.py:197-237 -> Convolution(strides=s)).
// Same direct form as above with  input index = stride * output index + tap - 1  per axis;  out dims = floor((in - 1) / stride) + 1.
template <int COT, bool FULL>     // FULL: every channel group is complete (Cout % COT == 0): no guards in the hot loop
__global__ void __launch_bounds__(256)
conv3d_k3_strided_kernel(Tensor in, const float* __restrict__ wp, const float* __restrict__ bias, Tensor out, int sz, int sy, int sx) {
    const int D = in.D, H = in.H, W = in.W, Do = out.D, Ho = out.H, Wo = out.W, Cin = in.C, Cout = out.C;
    const long long ivol = (long long)D * H * W, ovol = (long long)Do * Ho * Wo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * COT, n = blockIdx.z;
    if (idx >= ovol) return;
    const int x = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int y = (int)(t % Ho), z = (int)(t / Ho);
    // per-axis tap offsets clamped into the volume + a 27-bit validity mask: all 27 loads of a channel are issued
    // unconditionally (they overlap in flight), padding taps are zeroed afterwards
    int zo[3], yo[3], xo[3];
    unsigned zm = 0u, ym = 0u, xm = 0u;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int zz = z * sz + k - 1, yy = y * sy + k - 1, xx = x * sx + k - 1;      // one stride per axis (nnU-Net plans such as (1, 2, 2))
        zm |= (unsigned)(zz >= 0 && zz < D) << k; ym |= (unsigned)(yy >= 0 && yy < H) << k; xm |= (unsigned)(xx >= 0 && xx < W) << k;
        zo[k] = min(max(zz, 0), D - 1) * H * W; yo[k] = min(max(yy, 0), H - 1) * W; xo[k] = min(max(xx, 0), W - 1);
    }
    unsigned okm = 0u;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap)
        okm |= (((zm >> (tap / 9)) & (ym >> ((tap / 3) % 3)) & (xm >> (tap % 3))) & 1u) << tap;
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.0f;
    const float* src = in.data + (long long)n * in.n_stride;
    for (int ci = 0; ci < Cin; ++ci) {
        const float4 a = load_nrm(in, n, ci);
        const float* plane = src + (long long)ci * ivol;
        const float* wrow = wp + (long long)ci * 27 * Cout + co0;
        float v[27];
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) v[tap] = plane[zo[tap / 9] + yo[(tap / 3) % 3] + xo[tap % 3]];
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) v[tap] = ((okm >> tap) & 1u) ? act(v[tap], a.x, a.y, a.z) : 0.0f;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
#pragma unroll
            for (int j = 0; j < COT; ++j)
                if (FULL || co0 + j < Cout) acc[j] = fmaf(v[tap], wrow[tap * Cout + j], acc[j]);
        }
    }
    float* dst = out.data + (long long)n * out.n_stride + idx;
#pragma unroll
    for (int j = 0; j < COT; ++j)
        if (co0 + j < Cout) dst[(long long)(co0 + j) * ovol] = acc[j];
}

// ---------------------------------------------------------------------------------------------------
// ConvTranspose3d with kernel == stride == (fz, fy, fx), each 1 or 2 (DynUNet's upsampling for anisotropic nnU-Net plans, e.g. (1, 2, 2):
// monai/networks/blocks/dynunet_block.py:188-201).  Gather form: one thread = one OUTPUT voxel x COT channels; the voxel's input is
// (z / fz, y / fy, x / fx), its tap ((z % fz) * fy + y % fy) * fx + x % fx of the torch layout [Cin][Cout][fz][fy][fx].
template <int COT>
__global__ void __launch_bounds__(256)
deconv_ks_kernel(Tensor in, const float* __restrict__ w, const float* __restrict__ bias, Tensor out, int fz, int fy, int fx) {
    const int Hi = in.H, Wi = in.W, Cin = in.C, Cout = out.C, Ho = out.H, Wo = out.W;
    const long long ivol = (long long)in.D * Hi * Wi, ovol = (long long)out.D * Ho * Wo;
    const long long idx0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * COT, n = blockIdx.z;
    const bool valid = idx0 < ovol;
    const long long idx = valid ? idx0 : ovol - 1;
    const int x = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int y = (int)(t % Ho), z = (int)(t / Ho);
    const int taps = fz * fy * fx, tap = ((z % fz) * fy + y % fy) * fx + x % fx;
    const float* src = in.data + (long long)n * in.n_stride + ((long long)(z / fz) * Hi + y / fy) * Wi + x / fx;
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.0f;
    for (int ci = 0; ci < Cin; ++ci) {
        const float4 a = load_nrm(in, n, ci);
        const float v = act(src[(long long)ci * ivol], a.x, a.y, a.z);
        const float* wr = w + ((long long)ci * Cout + co0) * taps + tap;
#pragma unroll
        for (int j = 0; j < COT; ++j)
            if (co0 + j < Cout) acc[j] = fmaf(v, wr[(long long)j * taps], acc[j]);
    }
    float* dst = out.data + (long long)n * out.n_stride + idx;
#pragma unroll
    for (int j = 0; j < COT; ++j)
        if (co0 + j < Cout) {
            if (valid) dst[(long long)(co0 + j) * ovol] = acc[j];
            if (out.nrm) bound_commit(valid ? abs_bits(acc[j]) : 0u, bound_slot(out, n, co0 + j));
        }
}

// ---------------------------------------------------------------------------------------------------
// ConvTranspose3d k=3, stride 2, padding 1, output_padding 1 (UNet's up path, unet.py:249-294): out dims = 2 * in.
// Output o = 2 i - 1 + k: an even output takes tap 1 of input o/2; an odd output 2i+1 takes tap 2 of input i and tap 0
// of input i+1.  One thread owns an INPUT position and produces its 2x2x2 output block for COT channels from the 2x2x2
// input neighbourhood: the 27 weights of a (ci, co) pair are each used exactly once (1+2+2+2+4+4+4+8 = 27 MACs per
// 8 outputs) -- no tap is tested and thrown away, and the 8 loads of a channel fly together.
// Weights [Cin][Cout][27] (PyTorch ConvTranspose3d layout), read with uniform indices (scalar loads).
template <int COT, bool FULL>
__global__ void __launch_bounds__(256)
deconv_k3s2_kernel(Tensor in, const float* __restrict__ w, const float* __restrict__ bias, Tensor out) {
    const int D = in.D, H = in.H, W = in.W, Cin = in.C, Cout = out.C;
    const long long ivol = (long long)D * H * W, ovol = 8 * ivol;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * COT, n = blockIdx.z;
    if (idx >= ivol) return;
    const int ix = (int)(idx % W);
    const long long t = idx / W;
    const int iy = (int)(t % H), iz = (int)(t / H);
    const bool okz = iz + 1 < D, oky = iy + 1 < H, okx = ix + 1 < W;
    const int oz = okz ? H * W : 0, oy = oky ? W : 0, ox = okx ? 1 : 0;
    float acc[8][COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const float b = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.0f;
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o][j] = b;
    }
    const float* src = in.data + (long long)n * in.n_stride + idx;
    for (int ci = 0; ci < Cin; ++ci) {
        const float4 a = load_nrm(in, n, ci);
        const float* p = src + (long long)ci * ivol;
        float v[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) v[d] = p[(d >> 2) * oz + ((d >> 1) & 1) * oy + (d & 1) * ox];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const bool ok = (!(d >> 2) || okz) && (!((d >> 1) & 1) || oky) && (!(d & 1) || okx);
            v[d] = ok ? act(v[d], a.x, a.y, a.z) : 0.0f;
        }
        const float* wr = w + ((long long)ci * Cout + co0) * 27;
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            if (!FULL && co0 + j >= Cout) continue;
#pragma unroll
            for (int pz = 0; pz < 2; ++pz)
#pragma unroll
                for (int py = 0; py < 2; ++py)
#pragma unroll
                    for (int px = 0; px < 2; ++px)
#pragma unroll
                        for (int dz = 0; dz <= pz; ++dz)
#pragma unroll
                            for (int dy = 0; dy <= py; ++dy)
#pragma unroll
                                for (int dx = 0; dx <= px; ++dx) {
                                    const int kz = pz ? (dz ? 0 : 2) : 1, ky = py ? (dy ? 0 : 2) : 1, kx = px ? (dx ? 0 : 2) : 1;
                                    acc[pz * 4 + py * 2 + px][j] =
                                        fmaf(v[dz * 4 + dy * 2 + dx], wr[j * 27 + kz * 9 + ky * 3 + kx], acc[pz * 4 + py * 2 + px][j]);
                                }
        }
    }
    const int Ho = 2 * H, Wo = 2 * W;
    float* dst = out.data + (long long)n * out.n_stride + ((long long)(2 * iz) * Ho + 2 * iy) * Wo + 2 * ix;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        if (co0 + j >= Cout) continue;
        float* q = dst + (long long)(co0 + j) * ovol;
#pragma unroll
        for (int pz = 0; pz < 2; ++pz)
#pragma unroll
            for (int py = 0; py < 2; ++py)      // the (x, x+1) pair is 8-byte aligned (launcher checks base and strides)
                *reinterpret_cast<float2*>(q + ((long long)pz * Ho + py) * Wo) = make_float2(acc[pz * 4 + py * 2][j], acc[pz * 4 + py * 2 + 1][j]);
    }
}

// ---------------------------------------------------------------------------------------------------
// ConvTranspose3d k=3, any stride s, padding 1, output_padding s-1 (out dims = s * in): generic fallback of the kernel above.
// Gather form: output o takes input i through tap k whenever  i * s + k - 1 == o.  Weights [Cin][Cout][27].
template <int COT>
__global__ void __launch_bounds__(256)
deconv_k3_kernel(Tensor in, const float* __restrict__ w, const float* __restrict__ bias, Tensor out, int stride) {
    const int D = in.D, H = in.H, W = in.W, Do = out.D, Ho = out.H, Wo = out.W, Cin = in.C, Cout = out.C;
    const long long ivol = (long long)D * H * W, ovol = (long long)Do * Ho * Wo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * COT, n = blockIdx.z;
    if (idx >= ovol) return;
    const int x = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int y = (int)(t % Ho), z = (int)(t / Ho);
    // per-axis: which taps hit an input sample, and which one
    int iz[3], iy[3], ix[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int tz = z + 1 - k, ty = y + 1 - k, tx = x + 1 - k;
        iz[k] = (tz >= 0 && tz % stride == 0 && tz / stride < D) ? tz / stride : -1;
        iy[k] = (ty >= 0 && ty % stride == 0 && ty / stride < H) ? ty / stride : -1;
        ix[k] = (tx >= 0 && tx % stride == 0 && tx / stride < W) ? tx / stride : -1;
    }
    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = (bias && co0 + j < Cout) ? bias[co0 + j] : 0.0f;
    const float* src = in.data + (long long)n * in.n_stride;
    for (int ci = 0; ci < Cin; ++ci) {
        const float4 a = load_nrm(in, n, ci);
        const float* plane = src + (long long)ci * ivol;
        const float* wrow = w + ((long long)ci * Cout + co0) * 27;
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
            const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
            float v = 0.0f;
            if (iz[kz] >= 0 && iy[ky] >= 0 && ix[kx] >= 0) v = act(plane[((long long)iz[kz] * H + iy[ky]) * W + ix[kx]], a.x, a.y, a.z);
#pragma unroll
            for (int j = 0; j < COT; ++j)
                if (co0 + j < Cout) acc[j] = fmaf(v, wrow[j * 27 + tap], acc[j]);
        }
    }
    float* dst = out.data + (long long)n * out.n_stride + idx;
#pragma unroll
    for (int j = 0; j < COT; ++j)
        if (co0 + j < Cout) dst[(long long)(co0 + j) * ovol] = acc[j];
}

// ---------------------------------------------------------------------------------------------------
// MaxPool3d(2) of act(in).  One thread per output voxel; the two x-neighbours come in as one float2.  out.D == in.D: the z axis is not
// pooled -- MaxPool2d(2) of a 2-D network running as one plane (or a stack of planes) of this engine.
template <bool PAIR, int NZ>       // NZ = 2: MaxPool3d(2); NZ = 1: plane-wise (the z axis is kept)
__global__ void __launch_bounds__(256) maxpool2_kernel(Tensor in, Tensor out) {
    const int Do = out.D, Ho = out.H, Wo = out.W, H = in.H, W = in.W;
    const long long ovol = (long long)Do * Ho * Wo;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, n = blockIdx.z;
    const float4 a = load_nrm(in, n, c);
    // the pooled values are activated values of `in`: its bound is theirs (one writer per (n, c))
    if (out.nrm && idx == 0) *reinterpret_cast<float4*>(bound_slot(out, n, c) - 3) = make_float4(1.0f, 0.0f, 1.0f, a.w);
    if (idx >= ovol) return;
    const int xo = (int)(idx % Wo);
    const long long t = idx / Wo;
    const int yo = (int)(t % Ho), zo = (int)(t / Ho);
    const float* src = in.data + (long long)n * in.n_stride + (long long)c * in.D * H * W;
    float m = -3.402823466e+38f;
#pragma unroll
    for (int dz = 0; dz < NZ; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const float* row = src + ((long long)(NZ * zo + dz) * H + (2 * yo + dy)) * W + 2 * xo;
            float v0, v1;
            if (PAIR) {
                const float2 q = *reinterpret_cast<const float2*>(row);
                v0 = q.x; v1 = q.y;
            } else {
                v0 = row[0]; v1 = row[1];
            }
            m = fmaxf(m, fmaxf(act(v0, a.x, a.y, a.z), act(v1, a.x, a.y, a.z)));
        }
    out.data[(long long)n * out.n_stride + (long long)c * ovol + idx] = m;
}

// ---------------------------------------------------------------------------------------------------
// The pooling epilogue of the split-precision convolution (conv3d_h2.h, POOL) leaves the 2 x 2 x 2 maxima AND minima of the RAW convolution output: the consumer reads the
// maxima under the producer's records, which is max(act(x)) exactly when alpha >= 0 (normalise + LeakyReLU is then non-decreasing in x).  For a channel whose alpha came
// out negative the maximum of the activated values sits at the raw MINIMUM: its plane is copied over.  One launch per tensor; channels with alpha >= 0 (all of them for
// freshly initialised norms, most of them for trained ones) cost a record read and an exit.
__global__ void __launch_bounds__(256) pool_select_kernel(float* __restrict__ pmax, const float* __restrict__ pmin, const float* __restrict__ nrm, long long nrm_n_stride,
                                                          long long n_stride, long long vol) {
    const int c = blockIdx.y, n = blockIdx.z;
    const float alpha = nrm[(long long)n * nrm_n_stride + 4LL * c];
    if (!(alpha < 0.0f)) return;
    float* d = pmax + (long long)n * n_stride + (long long)c * vol;
    const float* s = pmin + (long long)n * n_stride + (long long)c * vol;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < vol; i += (long long)gridDim.x * 256) d[i] = s[i];
}

// ---------------------------------------------------------------------------------------------------
// ConvTranspose3d k=2 s=2 of act(in): every input voxel owns a disjoint 2x2x2 output block, so it is eight
// independent 1x1 convs.  One thread = one input voxel x COT output channels x 8 taps; weights
// [Cin][Cout][8] are block-uniform (scalar loads); each (cout, dz, dy) row pair is one coalesced float2 store.
template <int COT, bool FULL>     // FULL: Cout % COT == 0 (no channel guards in the hot loop)
__global__ void __launch_bounds__(256)
deconv_k2s2_kernel(Tensor in, const float* __restrict__ w, const float* __restrict__ bias, Tensor out) {
    const int Di = in.D, Hi = in.H, Wi = in.W, Cin = in.C, Cout = out.C;
    const long long ivol = (long long)Di * Hi * Wi;
    const long long idx0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int co0 = blockIdx.y * COT, n = blockIdx.z;
    const bool valid = idx0 < ivol;                  // lanes past the volume recompute its last voxel and store nothing: the wave stays whole for the bound
    const long long idx = valid ? idx0 : ivol - 1;
    const int x = (int)(idx % Wi);
    const long long t = idx / Wi;
    const int y = (int)(t % Hi), z = (int)(t / Hi);

    // the two x-taps (dx = 0, 1) of a (cout, dz, dy) are one register pair: a multiply-add of both is ONE v_pk_fma_f32 (activated input broadcast, the weight pair a scalar
    // register pair) -- the kernel was bound by its 8 Cin Cout scalar multiply-adds per voxel (1.7 of 2.4 ms at 32 -> 32 channels, 48^3 x 64 windows), not by its stores
    f32x2 acc[COT][4];
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const float bj = (bias && (FULL || co0 + j < Cout)) ? bias[co0 + j] : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[j][k] = f32x2{bj, bj};
    }
    const float* src = in.data + (long long)n * in.n_stride + idx;
    constexpr int CB = 8;      // input channels whose loads fly together (one load per iteration would expose its latency Cin times)
    for (int c0 = 0; c0 < Cin; c0 += CB) {
        float v[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) v[c] = src[(long long)min(c0 + c, Cin - 1) * ivol];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int ci = c0 + c;
            if (ci < Cin) {
                const float4 a = load_nrm(in, n, ci);
                const float va = act(v[c], a.x, a.y, a.z);
                const f32x2 vv = {va, va};
                const float* wr = w + ((long long)ci * Cout + co0) * 8;
#pragma unroll
                for (int j = 0; j < COT; ++j)
                    if (FULL || co0 + j < Cout) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) acc[j][k] = __builtin_elementwise_fma(vv, f32x2{wr[j * 8 + 2 * k], wr[j * 8 + 2 * k + 1]}, acc[j][k]);
                    }
            }
        }
    }
    const int Ho = out.H, Wo = out.W;
    const long long ovol = (long long)out.D * Ho * Wo;
    float* dst = out.data + (long long)n * out.n_stride;
#pragma unroll
    for (int j = 0; j < COT; ++j)
        if (FULL || co0 + j < Cout) {
            if (valid) {
#pragma unroll
                for (int dz = 0; dz < 2; ++dz)
#pragma unroll
                    for (int dy = 0; dy < 2; ++dy) {
                        float* p = dst + (long long)(co0 + j) * ovol + ((long long)(2 * z + dz) * Ho + (2 * y + dy)) * Wo + 2 * x;
                        *reinterpret_cast<f32x2*>(p) = acc[j][dz * 2 + dy];
                    }
            }
        }
    if (out.nrm) {
        // magnitude bound of the raw result (common.h): ONE wave reduction over everything the wave wrote, folded into the records of all the channels of
        // the group by one atomic instruction (lane j -> channel co0 + j).  The group's maximum bounds each of its channels (a little looser than per channel;
        // per-channel reductions cost 8 x the shuffles and atomics and were measured at +55 % on this store-bound kernel, profiles/r03_bound_epilogue_first_form_trace.txt)
        unsigned m = 0u;
#pragma unroll
        for (int j = 0; j < COT; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) m = max(m, max(abs_bits(acc[j][k][0]), abs_bits(acc[j][k][1])));
        __shared__ unsigned wmax[4];
        m = wave_umax(valid ? m : 0u);
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();                 // one atomic instruction per WORKGROUP: the waves of a (sample, channel) all target the same records
        const int jl = threadIdx.x;
        m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (jl < COT && (FULL || co0 + jl < Cout) && m != 0u) atomicMax(reinterpret_cast<unsigned*>(bound_slot(out, n, co0 + jl)), m);
    }
}

#ifdef MH_DEV_KNOBS
// Measurement variant (dev builds only, tools/deconv_bench.py): two x-adjacent input voxels per thread, COT output channels, 16-byte stores
// (four consecutive outputs of a row); SWAP: the cout group is the fastest workgroup index.
template <int COT, bool SWAP>
__global__ void __launch_bounds__(256)
deconv_k2s2_x2_kernel(Tensor in, const float* __restrict__ w, const float* __restrict__ bias, Tensor out) {
    const int Di = in.D, Hi = in.H, Wi = in.W, Cin = in.C, Cout = out.C;
    const long long ivol = (long long)Di * Hi * Wi, pairs = ivol / 2;
    const unsigned bx = SWAP ? blockIdx.y : blockIdx.x, by = SWAP ? blockIdx.x : blockIdx.y;
    const long long pidx = (long long)bx * 256 + threadIdx.x;
    const int co0 = by * COT, n = blockIdx.z;
    if (pidx >= pairs) return;
    const long long idx = 2 * pidx;
    const int x = (int)(idx % Wi);
    const long long t = idx / Wi;
    const int y = (int)(t % Hi), z = (int)(t / Hi);
    float acc[COT][2][8];
#pragma unroll
    for (int j = 0; j < COT; ++j)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[j][v][k] = bias ? bias[co0 + j] : 0.0f;
    const float* src = in.data + (long long)n * in.n_stride + idx;
    for (int ci = 0; ci < Cin; ++ci) {
        const float2 q = *reinterpret_cast<const float2*>(src + (long long)ci * ivol);
        const float4 a = load_nrm(in, n, ci);
        const float va[2] = {act(q.x, a.x, a.y, a.z), act(q.y, a.x, a.y, a.z)};
        const float* wr = w + ((long long)ci * Cout + co0) * 8;
#pragma unroll
        for (int j = 0; j < COT; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[j][0][k] = fmaf(va[0], wr[j * 8 + k], acc[j][0][k]);
                acc[j][1][k] = fmaf(va[1], wr[j * 8 + k], acc[j][1][k]);
            }
    }
    const int Ho = out.H, Wo = out.W;
    const long long ovol = (long long)out.D * Ho * Wo;
    float* dst = out.data + (long long)n * out.n_stride;
#pragma unroll
    for (int j = 0; j < COT; ++j)
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                float* p = dst + (long long)(co0 + j) * ovol + ((long long)(2 * z + dz) * Ho + (2 * y + dy)) * Wo + 2 * x;
                *reinterpret_cast<float4*>(p) = make_float4(acc[j][0][dz * 4 + dy * 2], acc[j][0][dz * 4 + dy * 2 + 1], acc[j][1][dz * 4 + dy * 2], acc[j][1][dz * 4 + dy * 2 + 1]);
            }
}
#endif

// Measured and removed (profiles/r02_deconv_bench_v1.json, r03_deconv_bench.json): the same op as ONE GEMM on the fp32 matrix cores, first with this
// kernel's store pattern, then writing complete 256-byte runs through a wave-private LDS tile -- 2.38 / 2.36 ms against 2.44 ms here (32 -> 32 ch @ 48^3
// x 64 windows), slower on the smaller levels: neither the 8 Cin Cout multiply-adds per voxel nor the store granularity bound this op.

// ---------------------------------------------------------------------------------------------------
// Sums of P values per lane over the 64 lanes of a wave with 2 P - 2 + (6 - log2 P) shuffles instead of 6 P: in step k a lane keeps one half of its values and
// sends the other half to the lane `32 >> k` away, so after log2 P steps it holds ONE value -- the partial sum of value index wave_multi_owner<P>(lane) -- and the
// remaining lane bits are reduced the plain way.  Every lane ends with the complete sum of the value index it owns.
template <int P> __device__ __forceinline__ int wave_multi_owner(int lane) {
    int j = 0, off = 32;
#pragma unroll
    for (int n = P; n > 1; n >>= 1, off >>= 1) j += (lane & off) ? (n >> 1) : 0;
    return j;
}
template <int P> __device__ __forceinline__ float wave_multi_sum(float (&s)[P], int lane) {
    static_assert(P >= 1 && P <= 64 && (P & (P - 1)) == 0, "a power of two");
    int off = 32;
#pragma unroll
    for (int n = P; n > 1; n >>= 1, off >>= 1) {
        const bool hi = (lane & off) != 0;
#pragma unroll
        for (int j = 0; j < n / 2; ++j) {
            const float keep = hi ? s[j + n / 2] : s[j], send = hi ? s[j] : s[j + n / 2];
            s[j] = keep + __shfl_xor(send, off);
        }
    }
    float r = s[0];
    for (; off >= 1; off >>= 1) r += __shfl_xor(r, off);
    return r;
}

// Conv3d k=1 of act(in): CO output channels [co0, co0+CO) per thread, VEC voxels per thread.
// STATS: the workgroup also leaves one {count, mean, M2} record per output channel for the 256 x VEC voxels it wrote (M2 about the workgroup's mean: two passes
// over the registers, as instnorm_stats_kernel) at stats[((n * Cout + c) * tiles + blockIdx.x) * 3] -- the InstanceNorm behind a 1x1x1 shortcut convolution
// (UnetResBlock.conv3 / norm3, dynunet_block.py:72-111) needs no pass of its own over the tensor.
template <int CO, int VEC, bool STATS>
__global__ void __launch_bounds__(256)
conv1x1_kernel(Tensor in, const float* __restrict__ w, const float* __restrict__ bias, Tensor out, int co0, float* __restrict__ stats, int tiles) {
    const int Cin = in.C;
    const long long DHW = (long long)in.D * in.H * in.W;
    const long long idx0 = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
    const int n = blockIdx.y;
    const bool valid = idx0 < DHW;
    if (!STATS && !valid) return;
    const long long idx = valid ? idx0 : 0;          // STATS: lanes beyond the plane stay for the reductions (they recompute voxel 0 and contribute nothing)
    float acc[CO][VEC];
#pragma unroll
    for (int j = 0; j < CO; ++j) {
        const float bj = bias ? bias[co0 + j] : 0.0f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[j][v] = bj;
    }
    const float* src = in.data + (long long)n * in.n_stride + idx;
    constexpr int CB = VEC == 4 ? 4 : 8;   // input channels whose loads fly together (one load per iteration would expose its latency Cin times)
    for (int c0 = 0; c0 < Cin; c0 += CB) {
        float xv[CB][VEC];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const float* p = src + (long long)min(c0 + c, Cin - 1) * DHW;
            if (VEC == 4) {
                const float4 q = *reinterpret_cast<const float4*>(p);
                xv[c][0] = q.x; xv[c][1] = q.y; xv[c][2] = q.z; xv[c][3] = q.w;
            } else {
                xv[c][0] = p[0];
            }
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int ci = c0 + c;
            if (ci < Cin) {
                const float4 a = load_nrm(in, n, ci);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float f = act(xv[c][v], a.x, a.y, a.z);
#pragma unroll
                    for (int j = 0; j < CO; ++j) acc[j][v] = fmaf(f, w[(long long)(co0 + j) * Cin + ci], acc[j][v]);
                }
            }
        }
    }
    if (valid) {
        float* dst = out.data + (long long)n * out.n_stride + idx;
#pragma unroll
        for (int j = 0; j < CO; ++j) {
            float* p = dst + (long long)(co0 + j) * DHW;
            if (VEC == 4) {
                *reinterpret_cast<float4*>(p) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
            } else {
                p[0] = acc[j][0];
            }
        }
    }
    if (STATS) {
        constexpr int P = CO <= 1 ? 1 : CO <= 2 ? 2 : CO <= 4 ? 4 : CO <= 8 ? 8 : 16;
        __shared__ float red_s[4][P], red_q[4][P];
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, own = wave_multi_owner<P>(lane);
        const float tot = (float)min((long long)256 * VEC, DHW - (long long)blockIdx.x * 256 * VEC);      // DHW % VEC == 0 (launcher)
        float t[P];
#pragma unroll
        for (int j = 0; j < P; ++j) {
            t[j] = 0.0f;
            if (j < CO && valid) {
#pragma unroll
                for (int v = 0; v < VEC; ++v) t[j] += acc[j][v];
            }
        }
        float r = wave_multi_sum<P>(t, lane);
        if ((lane & (64 / P - 1)) == 0) red_s[wave][own] = r;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < P; ++j) {
            t[j] = 0.0f;
            if (j < CO && valid) {
                const float mean = ((red_s[0][j] + red_s[1][j]) + (red_s[2][j] + red_s[3][j])) / tot;
#pragma unroll
                for (int v = 0; v < VEC; ++v) { const float d = acc[j][v] - mean; t[j] += d * d; }
            }
        }
        r = wave_multi_sum<P>(t, lane);
        if ((lane & (64 / P - 1)) == 0) red_q[wave][own] = r;
        __syncthreads();
        if (threadIdx.x < CO) {
            const int j = threadIdx.x;
            float* rec = stats + (((long long)n * out.C + co0 + j) * tiles + blockIdx.x) * 3;
            rec[0] = tot;
            rec[1] = ((red_s[0][j] + red_s[1][j]) + (red_s[2][j] + red_s[3][j])) / tot;
            rec[2] = (red_q[0][j] + red_q[1][j]) + (red_q[2][j] + red_q[3][j]);
        }
    }
}

// The same 1x1 convolution writing each batch element (a window of the sliding-window inferer) to its own strided destination -- the
// mosaic logits layout of sliding.h: place[n] = {float offset, channel stride, z stride, y stride} relative to `base`; rows (x) stay contiguous.
constexpr int WIN_PLACE_MAX = 64;
struct WinPlace {
    long long off[WIN_PLACE_MAX], sc[WIN_PLACE_MAX], sd[WIN_PLACE_MAX], sh[WIN_PLACE_MAX];
};
#ifndef MH_C1W_CB
#define MH_C1W_CB 4
#endif
template <int CO>
__global__ void __launch_bounds__(256)
conv1x1_windows_kernel(Tensor in, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ base, WinPlace pl, int n0) {
    constexpr int VEC = 4;
    const int Cin = in.C, H = in.H, W = in.W;
    const long long DHW = (long long)in.D * H * W;
    const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
    const int nl = blockIdx.y, n = n0 + nl;
    if (idx >= DHW) return;
    float acc[CO][VEC];
#pragma unroll
    for (int j = 0; j < CO; ++j) {
        const float bj = bias ? bias[j] : 0.0f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[j][v] = bj;
    }
    const float* src = in.data + (long long)n * in.n_stride + idx;
    constexpr int CB = MH_C1W_CB;      // channels whose 16-byte loads fly together
    for (int c0 = 0; c0 < Cin; c0 += CB) {
        float xv[CB][VEC];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const float4 q = *reinterpret_cast<const float4*>(src + (long long)min(c0 + c, Cin - 1) * DHW);
            xv[c][0] = q.x; xv[c][1] = q.y; xv[c][2] = q.z; xv[c][3] = q.w;
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int ci = c0 + c;
            if (ci < Cin) {
                const float4 a = load_nrm(in, n, ci);
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float f = act(xv[c][v], a.x, a.y, a.z);
#pragma unroll
                    for (int j = 0; j < CO; ++j) acc[j][v] = fmaf(f, w[(long long)j * Cin + ci], acc[j][v]);
                }
            }
        }
    }
    const int x = (int)(idx % W);
    const long long t = idx / W;
    const int y = (int)(t % H), z = (int)(t / H);
    float* dst = base + pl.off[nl] + (long long)z * pl.sd[nl] + (long long)y * pl.sh[nl] + x;
#pragma unroll
    for (int j = 0; j < CO; ++j) *reinterpret_cast<float4*>(dst + (long long)j * pl.sc[nl]) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
}

// ---------------------------------------------------------------------------------------------------
// Residual join of UnetResBlock (monai/networks/blocks/dynunet_block.py:96-111): out = lrelu(norm2(conv2) + residual),
// both operands given as raw tensors + their deferred {alpha, beta, slope} records; one read of each, one write.
constexpr int ADD_ACT_CHUNKS = 4;      // consecutive 256 x VEC element chunks per workgroup
template <int VEC>
__global__ void __launch_bounds__(256) add_act_kernel(Tensor a, Tensor b, float slope, Tensor out) {
    const long long DHW = (long long)a.D * a.H * a.W;
    const int c = blockIdx.y, n = blockIdx.z;
    const bool has_b = b.data != nullptr;          // no second operand: out = lrelu(act(a)) (materialise a deferred tensor)
    const float4 na = load_nrm(a, n, c), nb = load_nrm(b, n, c);
    const float* pa0 = a.data + (long long)n * a.n_stride + (long long)c * DHW;
    const float* pb0 = has_b ? b.data + (long long)n * b.n_stride + (long long)c * DHW : pa0;
    float* po0 = out.data + (long long)n * out.n_stride + (long long)c * DHW;
    unsigned m = 0u;
#pragma unroll
    for (int u = 0; u < ADD_ACT_CHUNKS; ++u) {
        const long long idx = (((long long)blockIdx.x * ADD_ACT_CHUNKS + u) * 256 + threadIdx.x) * VEC;
        if (idx >= DHW) break;                       // VEC == 4: DHW % 4 == 0 (launcher), so a group is inside or outside as a whole
        float va[VEC], vb[VEC];
        if (VEC == 4) {
            const float4 qa = *reinterpret_cast<const float4*>(pa0 + idx), qb = *reinterpret_cast<const float4*>(pb0 + idx);
            va[0] = qa.x; va[1] = qa.y; va[2] = qa.z; va[3] = qa.w;
            vb[0] = qb.x; vb[1] = qb.y; vb[2] = qb.z; vb[3] = qb.w;
        } else {
            va[0] = pa0[idx]; vb[0] = pb0[idx];
        }
        float r[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            const float y = act(va[v], na.x, na.y, na.z) + (has_b ? act(vb[v], nb.x, nb.y, nb.z) : 0.0f);
            r[v] = y > 0.0f ? y : y * slope;
            m = max(m, abs_bits(r[v]));
        }
        if (VEC == 4) *reinterpret_cast<float4*>(po0 + idx) = make_float4(r[0], r[1], r[2], r[3]);
        else po0[idx] = r[0];
    }
    if (out.nrm) {            // magnitude bound of the raw sum (common.h): one atomic per workgroup -- every workgroup of a (sample, channel) targets the same record
        __shared__ unsigned wmax[4];
        m = wave_umax(m);
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            m = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
            if (m != 0u) atomicMax(reinterpret_cast<unsigned*>(bound_slot(out, n, c)), m);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The residual join fused into the 1x1x1 output convolution that reads it (round 5): out[co] = bias[co] + sum_ci w[co][ci] lrelu(act(a[ci]) + act(b[ci]), slope) --
// UnetResBlock's lrelu(norm2(conv2) + residual) (dynunet_block.py:96-111) followed by UnetOutBlock (dynunet_block.py:251-268) without the joined tensor in HBM
// (UNETR / SwinUNETR: the last decoder block feeds only the output convolution; 2 x 16 resp. 2 x 48 channels x 96^3 per window less traffic).  CO <= 8 output
// channels per thread, four voxels per thread, the join evaluated in add_act_kernel's order, the channel sum in conv1x1_kernel's: the same bits as the two launches.
template <int CO>
__global__ void __launch_bounds__(256) conv1x1_sum2_kernel(Tensor a, Tensor b, float slope, const float* __restrict__ w, const float* __restrict__ bias, Tensor out) {
    const int Cin = a.C;
    const long long DHW = (long long)a.D * a.H * a.W;
    const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    const int n = blockIdx.y;
    if (idx >= DHW) return;
    float acc[CO][4];
#pragma unroll
    for (int j = 0; j < CO; ++j) {
        const float bj = bias ? bias[j] : 0.0f;
#pragma unroll
        for (int v = 0; v < 4; ++v) acc[j][v] = bj;
    }
    const float* pa = a.data + (long long)n * a.n_stride + idx;
    const float* pb = b.data + (long long)n * b.n_stride + idx;
    constexpr int CB = 4;          // channels whose loads fly together
    for (int c0 = 0; c0 < Cin; c0 += CB) {
        float4 qa[CB], qb[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const long long o = (long long)min(c0 + c, Cin - 1) * DHW;
            qa[c] = *reinterpret_cast<const float4*>(pa + o);
            qb[c] = *reinterpret_cast<const float4*>(pb + o);
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            const int ci = c0 + c;
            if (ci < Cin) {
                const float4 na = load_nrm(a, n, ci), nb = load_nrm(b, n, ci);
                const float va[4] = {qa[c].x, qa[c].y, qa[c].z, qa[c].w}, vb[4] = {qb[c].x, qb[c].y, qb[c].z, qb[c].w};
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const float y = act(va[v], na.x, na.y, na.z) + act(vb[v], nb.x, nb.y, nb.z);
                    const float f = y > 0.0f ? y : y * slope;
#pragma unroll
                    for (int j = 0; j < CO; ++j) acc[j][v] = fmaf(f, w[(long long)j * Cin + ci], acc[j][v]);
                }
            }
        }
    }
    float* dst = out.data + (long long)n * out.n_stride + idx;
#pragma unroll
    for (int j = 0; j < CO; ++j) *reinterpret_cast<float4*>(dst + (long long)j * DHW) = make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]);
}

// ---------------------------------------------------------------------------------------------------
// Replicate padding at the far end of each axis: out[z, y, x] = in[min(z, Di-1), min(y, Hi-1), min(x, Wi-1)] -- UpCat's
// F.pad(x_0, sp, "replicate") for odd encoder extents (monai/networks/nets/basic_unet.py:163-170).  Raw copy (no norm).
__global__ void __launch_bounds__(256) pad_replicate_kernel(Tensor in, Tensor out) {
    const long long ovol = (long long)out.D * out.H * out.W, ivol = (long long)in.D * in.H * in.W;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, n = blockIdx.z;
    if (out.nrm && idx == 0) *reinterpret_cast<float4*>(bound_slot(out, n, c) - 3) = make_float4(1.0f, 0.0f, 1.0f, load_nrm(in, n, c).w);   // a copy: same bound
    if (idx >= ovol) return;
    const int x = (int)(idx % out.W);
    const long long t = idx / out.W;
    const int y = (int)(t % out.H), z = (int)(t / out.H);
    const long long src = ((long long)min(z, in.D - 1) * in.H + min(y, in.H - 1)) * in.W + min(x, in.W - 1);
    out.data[(long long)n * out.n_stride + (long long)c * ovol + idx] = in.data[(long long)n * in.n_stride + (long long)c * ivol + src];
}

// ---------------------------------------------------------------------------------------------------
// SubpixelUpsample behind its convolution (monai/networks/blocks/upsample.py:186-288; pixelshuffle: monai/networks/utils.py:370-412), scale factor 2: the convolution's
// raw output `in` [N][C * FZ * 4][Dl][Hl][Wl] -> `out` [N][C][FZ * Dl][2 Hl][2 Wl] with  sh[c][FZ z + i][2 y + j][2 x + k] = in[c * FZ * 4 + i * 4 + j * 2 + k][z][y][x]
// (FZ = 2: three spatial dimensions; FZ = 1: the one-plane form of two) and, PADPOOL, the "pad then average" of Aitken et al. that follows it: ConstantPad(1 at the START of
// every spatial axis, 0) + AvgPool(kernel 2, stride 1), i.e. out[Z][Y][X] = mean over the 2^dims voxels sh[Z - dz][Y - dy][X - dx], dz, dy, dx in {0, 1}, zeros in front of
// the volume counted (the pool's count_include_pad default).  One thread per output voxel; the sum runs over (dz, dy, dx) in the pool's order.  The output is a raw tensor:
// max |value written| goes into its identity records (common.h).
template <int FZ, bool PADPOOL>
__global__ void __launch_bounds__(256) pixelshuffle_kernel(Tensor in, Tensor out) {
    const long long ovol = (long long)out.D * out.H * out.W, ivol = (long long)in.D * in.H * in.W;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int c = blockIdx.y, n = blockIdx.z;
    unsigned mb = 0u;
    if (idx < ovol) {
        const int X = (int)(idx % out.W);
        const long long t = idx / out.W;
        const int Y = (int)(t % out.H), Z = (int)(t / out.H);
        const float* src = in.data + (long long)n * in.n_stride + (long long)c * (FZ * 4) * ivol;
        float acc = 0.0f;
#pragma unroll
        for (int dz = (PADPOOL && FZ == 2) ? 1 : 0; dz >= 0; --dz)
#pragma unroll
            for (int dy = PADPOOL ? 1 : 0; dy >= 0; --dy)
#pragma unroll
                for (int dx = PADPOOL ? 1 : 0; dx >= 0; --dx) {      // the pool visits its window from the lowest index up: (Z - 1, Y - 1, X - 1) first
                    const int z = Z - dz, y = Y - dy, x = X - dx;
                    float v = 0.0f;
                    if (z >= 0 && y >= 0 && x >= 0) {
                        const int sub = (FZ == 2 ? (z & 1) * 4 : 0) + (y & 1) * 2 + (x & 1);
                        v = src[(long long)sub * ivol + ((long long)(FZ == 2 ? z >> 1 : z) * in.H + (y >> 1)) * in.W + (x >> 1)];
                    }
                    acc += v;
                }
        const float r = PADPOOL ? acc / (FZ == 2 ? 8.0f : 4.0f) : acc;
        out.data[(long long)n * out.n_stride + (long long)c * ovol + idx] = r;
        mb = abs_bits(r);
    }
    if (out.nrm) bound_commit(mb, bound_slot(out, n, c));
}

// ---------------------------------------------------------------------------------------------------
// Identity records {1, 0, 1, FLT_MIN} for the channels a raw producer is about to write (common.h: magnitude bounds).
__global__ void __launch_bounds__(256) nrm_identity_kernel(float* __restrict__ nrm, int N, int C, long long nrm_n_stride) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C) return;
    *reinterpret_cast<float4*>(nrm + (long long)(i / C) * nrm_n_stride + 4LL * (i % C)) = make_float4(1.0f, 0.0f, 1.0f, MH_BOUND_FLOOR);
}

// ---------------------------------------------------------------------------------------------------
// InstanceNorm statistics, stand-alone pass: one {count, mean, M2} record per 4096-element chunk of each
// (n, c) plane; M2 is taken about the chunk mean (two passes over registers), so no E[x^2]-E[x]^2 cancellation.
constexpr int STAT_CHUNK = 4096;

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();  // protect `red` from the previous use
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) instnorm_stats_kernel(Tensor x, float* __restrict__ stats, int tiles) {
    __shared__ float red[4];
    const long long DHW = (long long)x.D * x.H * x.W;
    const int tile = blockIdx.x, c = blockIdx.y, n = blockIdx.z;
    const long long start = (long long)tile * STAT_CHUNK;
    const float* src = x.data + (long long)n * x.n_stride + (long long)c * DHW;
    float v[16];
    float cnt = 0.0f, sum = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long long i = start + j * 256 + threadIdx.x;
        const bool ok = i < DHW;
        v[j] = ok ? src[i] : 0.0f;
        if (ok) { cnt += 1.0f; sum += v[j]; }
    }
    const float tot = block_sum_256(cnt, red);
    const float mean = block_sum_256(sum, red) / tot;
    float m2 = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const long long i = start + j * 256 + threadIdx.x;
        const float d = v[j] - mean;
        if (i < DHW) m2 += d * d;
    }
    m2 = block_sum_256(m2, red);
    if (threadIdx.x == 0) {
        float* rec = stats + (((long long)n * x.C + c) * tiles + tile) * 3;
        rec[0] = tot; rec[1] = mean; rec[2] = m2;
    }
}

// Merge `tiles` records per (n, c) in fp64 and emit the consumer-side {alpha, beta, slope, bound}.
// fp32 steps mirror ATen's CPU batch-norm: invstd = 1/sqrt(var + eps), alpha = gamma*invstd,
// beta = bias - mean*alpha (the reference normalises as x*alpha + beta).
__global__ void __launch_bounds__(64)
instnorm_finalize_kernel(const float* __restrict__ stats, int tiles, int C, int G, const float* __restrict__ gamma,
                         const float* __restrict__ beta, float eps, float slope, float* __restrict__ nrm,
                         long long nrm_n_stride) {
    // G = channels per normalisation group (1: InstanceNorm; > 1: GroupNorm -- the records of a group's channels are
    // contiguous, so a group is simply G * tiles records and every channel of it gets the group's mean / variance)
    const int c = blockIdx.x * G, n = blockIdx.y, lane = threadIdx.x;
    const float* rec = stats + ((long long)n * C + c) * tiles * 3;
    tiles *= G;
    double cnt = 0.0, ws = 0.0;
    for (int i = lane; i < tiles; i += 64) {
        cnt += (double)rec[i * 3];
        ws += (double)rec[i * 3] * (double)rec[i * 3 + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cnt += __shfl_xor(cnt, o);
        ws += __shfl_xor(ws, o);
    }
    const double mean = ws / cnt;
    double m2 = 0.0;
    for (int i = lane; i < tiles; i += 64) {
        const double d = (double)rec[i * 3 + 1] - mean;
        m2 += (double)rec[i * 3 + 2] + (double)rec[i * 3] * d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
    for (int k = lane; k < G; k += 64) {          // the butterfly left the totals in every lane
        const int ch = c + k;
        const float var = (float)(m2 / cnt);
        const float invstd = __fdiv_rn(1.0f, __fsqrt_rn(var + eps));
        const float g = gamma ? gamma[ch] : 1.0f;
        const float bb = beta ? beta[ch] : 0.0f;
        const float alpha = g * invstd;
        float* o = nrm + (long long)n * nrm_n_stride + 4LL * ch;
        o[0] = alpha;
        o[1] = bb - (float)mean * alpha;
        o[2] = slope;
        // |x - mean| <= sqrt(count - 1) * std for every element of the normalised set, so |gamma * xhat + beta| <= |gamma| sqrt(count) + |beta|
        // (times the activation's largest gain): a rigorous bound on what consumers see; NaN / inf statistics give a NaN / inf bound
        o[3] = (fabsf(g) * (float)sqrt(cnt) + fabsf(bb)) * fmaxf(1.0f, fabsf(slope));
    }
}

}  // namespace mh
