// Trilinear / nearest resampling of an NCDHW fp32 volume at coordinates given by a 3x4 voxel-space affine
// (Spacing / SpatialResample / AffineTransform) or by a dense coordinate grid (Resample, grid_pull).
//
// Reference behaviour: monai/networks/layers/spatial_transforms.py:584-591 (F.affine_grid + F.grid_sample, i.e.
// ATen's grid_sampler_3d: unnormalise -> padding rule on the coordinate -> floor / corner weights (x1-x)(y1-y)(z1-z)...
// -> sum of the in-bounds corners) and monai/transforms/spatial/array.py:2109.  The 8-corner sum is evaluated axis by
// axis (x pairs, then y, then z: 7 two-term combines instead of 12 weight products + 8 multiply-adds), which changes
// fp64 roundings at the 1e-16 level only -- below the float32 output cast.
// The reference materialises the (N, D, H, W, 3) sampling grid (24-32 B per output voxel in fp64); here the
// coordinate of an output voxel is three fp64 dot products with the composed 3x4 matrix, evaluated in registers:
// HBM traffic is the input read once (the 8-corner reuse is served by L1/L2) plus the output written once.
#pragma once
#include "common.h"

namespace mh {

enum { RS_NEAREST = 0, RS_LINEAR = 1 };
enum { RS_ZEROS = 0, RS_BORDER = 1, RS_REFLECTION = 2 };

struct ResampleArgs {
    double m[12];        // source index (z, y, x) = m[row*4 + 0..2] . (oz, oy, ox) + m[row*4 + 3]
    int mode, pad, align_corners;
    double ga[3], gb[3]; // dense-grid variant: source index = ga[axis] * coord + gb[axis]
    int C;               // channels sharing the coordinates (N*C of the tensor)
    int Di, Hi, Wi, Do, Ho, Wo;
};

__device__ __forceinline__ double rs_reflect(double in, double twice_low, double twice_high) {
    if (twice_low == twice_high) return 0.0;
    const double mn = twice_low / 2.0, span = (twice_high - twice_low) / 2.0;
    in = fabs(in - mn);
    const double extra = fmod(in, span);
    const long long flips = (long long)floor(in / span);
    return (flips % 2 == 0) ? extra + mn : span - extra + mn;
}

// ATen grid_sampler_compute_source_index after unnormalisation: the padding rule acts on the coordinate.
__device__ __forceinline__ double rs_pad_coord(double x, int size, int pad, int align_corners) {
    if (pad == RS_BORDER) {
        x = fmin((double)(size - 1), fmax(x, 0.0));
    } else if (pad == RS_REFLECTION) {
        x = align_corners ? rs_reflect(x, 0.0, 2.0 * (size - 1)) : rs_reflect(x, -1.0, 2.0 * size - 1.0);
        x = fmin((double)(size - 1), fmax(x, 0.0));
    }
    return x;
}

// One axis of a sampling position: tap indices (-1: the tap is outside the volume and contributes nothing) and weights.
template <typename T> struct AxisTap {
    int i0, i1;
    T w0, w1;
};

// ATen grid_sampler semantics along one axis: padding rule on the coordinate, then nearest (nearbyint, half to even)
// or the two linear taps floor / floor+1 with weights (x1 - x) and (x - x0); taps outside [0, size) are dropped.
template <typename T>
__device__ __forceinline__ AxisTap<T> rs_axis(double coord, int size, int mode, int pad, int align_corners) {
    AxisTap<T> t;
    const T c = (T)rs_pad_coord(coord, size, pad, align_corners);
    if (mode == RS_NEAREST) {
        const double r = nearbyint((double)c);
        t.i0 = (r >= 0.0 && r < (double)size) ? (int)r : -1;
        t.i1 = -1;
        t.w0 = (T)1; t.w1 = (T)0;
    } else {
        const T f = floor(c);
        const double fd = (double)f;
        t.i0 = (fd >= 0.0 && fd < (double)size) ? (int)fd : -1;
        t.i1 = (fd + 1.0 >= 0.0 && fd + 1.0 < (double)size) ? (int)fd + 1 : -1;
        t.w0 = (f + (T)1) - c;
        t.w1 = c - f;
    }
    return t;
}

// value = sum over the (up to) 8 corners of v * wx * wy * wz, combined axis by axis (x, then y, then z).
template <typename T, typename IDX>
__device__ __forceinline__ void rs_gather(const float* __restrict__ src, float* __restrict__ dst, int C, IDX ivol, IDX ovol, IDX oidx,
                                          int Hi, int Wi, const AxisTap<T>& tz, const AxisTap<T>& ty, const AxisTap<T>& tx, bool nearest) {
    if (nearest) {
        const bool ok = tz.i0 >= 0 && ty.i0 >= 0 && tx.i0 >= 0;
        const IDX off = ok ? ((IDX)tz.i0 * Hi + ty.i0) * Wi + tx.i0 : 0;
        for (int c = 0; c < C; ++c) dst[(IDX)c * ovol + oidx] = ok ? src[(IDX)c * ivol + off] : 0.0f;
        return;
    }
    const bool zok[2] = {tz.i0 >= 0, tz.i1 >= 0}, yok[2] = {ty.i0 >= 0, ty.i1 >= 0}, xok[2] = {tx.i0 >= 0, tx.i1 >= 0};
    const IDX zo[2] = {(IDX)(zok[0] ? tz.i0 : 0) * Hi * Wi, (IDX)(zok[1] ? tz.i1 : 0) * Hi * Wi};
    const IDX yo[2] = {(IDX)(yok[0] ? ty.i0 : 0) * Wi, (IDX)(yok[1] ? ty.i1 : 0) * Wi};
    const IDX xo[2] = {(IDX)(xok[0] ? tx.i0 : 0), (IDX)(xok[1] ? tx.i1 : 0)};
    for (int c = 0; c < C; ++c) {
        const float* p = src + (IDX)c * ivol;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[zo[k >> 2] + yo[(k >> 1) & 1] + xo[k & 1]];
        T row[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool rk = zok[r >> 1] && yok[r & 1];
            const T a0 = (rk && xok[0]) ? (T)v[2 * r] : (T)0, a1 = (rk && xok[1]) ? (T)v[2 * r + 1] : (T)0;
            row[r] = a0 * tx.w0 + a1 * tx.w1;
        }
        const T p0 = row[0] * ty.w0 + row[1] * ty.w1, p1 = row[2] * ty.w0 + row[3] * ty.w1;
        dst[(IDX)c * ovol + oidx] = (float)(p0 * tz.w0 + p1 * tz.w1);
    }
}

template <typename T, typename IDX>
__global__ void __launch_bounds__(256) affine_resample_kernel(const float* __restrict__ src, float* __restrict__ dst, ResampleArgs a) {
    const IDX ovol = (IDX)a.Do * a.Ho * a.Wo, ivol = (IDX)a.Di * a.Hi * a.Wi;
    const IDX oidx = (IDX)blockIdx.x * 256 + threadIdx.x;
    if (oidx >= ovol) return;
    const int ox = (int)(oidx % a.Wo);
    const IDX t = oidx / a.Wo;
    const int oy = (int)(t % a.Ho), oz = (int)(t / a.Ho);
    // same association as a 4-term matrix-vector product row: ((m0*z + m1*y) + m2*x) + m3
    const double cz = ((a.m[0] * oz + a.m[1] * oy) + a.m[2] * ox) + a.m[3];
    const double cy = ((a.m[4] * oz + a.m[5] * oy) + a.m[6] * ox) + a.m[7];
    const double cx = ((a.m[8] * oz + a.m[9] * oy) + a.m[10] * ox) + a.m[11];
    const AxisTap<T> tz = rs_axis<T>(cz, a.Di, a.mode, a.pad, a.align_corners);
    const AxisTap<T> ty = rs_axis<T>(cy, a.Hi, a.mode, a.pad, a.align_corners);
    const AxisTap<T> tx = rs_axis<T>(cx, a.Wi, a.mode, a.pad, a.align_corners);
    rs_gather<T, IDX>(src, dst, a.C, ivol, ovol, oidx, a.Hi, a.Wi, tz, ty, tx, a.mode == RS_NEAREST);
}

// Axis-aligned affines (no rotation / shear: every off-diagonal of the 3x3 block is exactly zero -- the Spacingd case)
// are separable: the taps of an output voxel are the product of three per-axis tables of Do + Ho + Wo entries, built
// once by a one-block kernel with the SAME per-axis routine; the volume kernel is then 8 loads + 7 two-term combines.
template <typename T> __global__ void __launch_bounds__(256) resample_axis_table_kernel(AxisTap<T>* __restrict__ tab, ResampleArgs a) {
    const int total = a.Do + a.Ho + a.Wo;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int axis, o, size;
        if (i < a.Do) { axis = 0; o = i; size = a.Di; }
        else if (i < a.Do + a.Ho) { axis = 1; o = i - a.Do; size = a.Hi; }
        else { axis = 2; o = i - a.Do - a.Ho; size = a.Wi; }
        const int oz = axis == 0 ? o : 0, oy = axis == 1 ? o : 0, ox = axis == 2 ? o : 0;
        // identical expression to the general kernel (the other two terms are exact zeros)
        const double c = ((a.m[axis * 4 + 0] * oz + a.m[axis * 4 + 1] * oy) + a.m[axis * 4 + 2] * ox) + a.m[axis * 4 + 3];
        tab[i] = rs_axis<T>(c, size, a.mode, a.pad, a.align_corners);
    }
}

template <typename T, typename IDX>
__global__ void __launch_bounds__(256)
separable_resample_kernel(const float* __restrict__ src, float* __restrict__ dst, const AxisTap<T>* __restrict__ tab, ResampleArgs a) {
    const IDX ovol = (IDX)a.Do * a.Ho * a.Wo, ivol = (IDX)a.Di * a.Hi * a.Wi;
    const IDX oidx = (IDX)blockIdx.x * 256 + threadIdx.x;
    if (oidx >= ovol) return;
    const int ox = (int)(oidx % a.Wo);
    const IDX t = oidx / a.Wo;
    const int oy = (int)(t % a.Ho), oz = (int)(t / a.Ho);
    const AxisTap<T> tz = tab[oz], ty = tab[a.Do + oy], tx = tab[a.Do + a.Ho + ox];
    rs_gather<T, IDX>(src, dst, a.C, ivol, ovol, oidx, a.Hi, a.Wi, tz, ty, tx, a.mode == RS_NEAREST);
}

// Dense grid: coords [3][Do][Ho][Wo] (planes z, y, x), fp32 or fp64 (GT); a per-axis affine (ga, gb) turns the stored
// coordinate convention (centred voxel units or [-1, 1]) into source voxel indices without rewriting the grid.
template <typename T, typename GT>
__global__ void __launch_bounds__(256)
grid_resample_kernel(const float* __restrict__ src, const GT* __restrict__ coords, float* __restrict__ dst, ResampleArgs a) {
    const long long ovol = (long long)a.Do * a.Ho * a.Wo, ivol = (long long)a.Di * a.Hi * a.Wi;
    const long long oidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (oidx >= ovol) return;
    const AxisTap<T> tz = rs_axis<T>(a.ga[0] * (double)coords[oidx] + a.gb[0], a.Di, a.mode, a.pad, a.align_corners);
    const AxisTap<T> ty = rs_axis<T>(a.ga[1] * (double)coords[ovol + oidx] + a.gb[1], a.Hi, a.mode, a.pad, a.align_corners);
    const AxisTap<T> tx = rs_axis<T>(a.ga[2] * (double)coords[2 * ovol + oidx] + a.gb[2], a.Wi, a.mode, a.pad, a.align_corners);
    rs_gather<T, long long>(src, dst, a.C, ivol, ovol, oidx, a.Hi, a.Wi, tz, ty, tx, a.mode == RS_NEAREST);
}

}  // namespace mh
