// Trilinear / nearest resampling of an NCDHW fp32 volume at coordinates given by a 3x4 voxel-space affine
// (Spacing / SpatialResample / AffineTransform) or by a dense coordinate grid (Resample, grid_pull).
//
// Reference behaviour: monai/networks/layers/spatial_transforms.py:584-591 (F.affine_grid + F.grid_sample, i.e.
// ATen's grid_sampler_3d: unnormalise -> padding rule on the coordinate -> floor / 8 corner weights -> sum of the
// in-bounds corners in the order tnw, tne, tsw, tse, bnw, bne, bsw, bse) and monai/transforms/spatial/array.py:2109.
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

template <typename T>
__device__ __forceinline__ void rs_sample(const float* __restrict__ src, float* __restrict__ dst, const ResampleArgs& a,
                                          double cz, double cy, double cx, long long oidx) {
    const long long ivol = (long long)a.Di * a.Hi * a.Wi, ovol = (long long)a.Do * a.Ho * a.Wo;
    const T iz = (T)rs_pad_coord(cz, a.Di, a.pad, a.align_corners);
    const T iy = (T)rs_pad_coord(cy, a.Hi, a.pad, a.align_corners);
    const T ix = (T)rs_pad_coord(cx, a.Wi, a.pad, a.align_corners);
    if (a.mode == RS_NEAREST) {
        const long long z = (long long)nearbyint((double)iz), y = (long long)nearbyint((double)iy), x = (long long)nearbyint((double)ix);
        const bool ok = z >= 0 && z < a.Di && y >= 0 && y < a.Hi && x >= 0 && x < a.Wi;
        const long long off = ok ? (z * a.Hi + y) * a.Wi + x : 0;
        for (int c = 0; c < a.C; ++c) dst[c * ovol + oidx] = ok ? src[c * ivol + off] : 0.0f;
        return;
    }
    const T z0f = floor(iz), y0f = floor(iy), x0f = floor(ix);
    const long long z0 = (long long)z0f, y0 = (long long)y0f, x0 = (long long)x0f;
    const T z1f = z0f + (T)1, y1f = y0f + (T)1, x1f = x0f + (T)1;
    const T wx0 = x1f - ix, wx1 = ix - x0f, wy0 = y1f - iy, wy1 = iy - y0f, wz0 = z1f - iz, wz1 = iz - z0f;
    // corner order and weight products as in ATen's grid_sampler_3d (x fastest: w/e, then n/s = y, then t/b = z)
    const T w[8] = {wx0 * wy0 * wz0, wx1 * wy0 * wz0, wx0 * wy1 * wz0, wx1 * wy1 * wz0,
                    wx0 * wy0 * wz1, wx1 * wy0 * wz1, wx0 * wy1 * wz1, wx1 * wy1 * wz1};
    long long off[8];
    bool ok[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const long long z = z0 + (k >> 2), y = y0 + ((k >> 1) & 1), x = x0 + (k & 1);
        ok[k] = z >= 0 && z < a.Di && y >= 0 && y < a.Hi && x >= 0 && x < a.Wi;
        off[k] = ok[k] ? (z * a.Hi + y) * a.Wi + x : 0;
    }
    for (int c = 0; c < a.C; ++c) {
        const float* p = src + c * ivol;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[off[k]];
        T acc = (T)0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (ok[k]) acc = acc + (T)v[k] * w[k];
        dst[c * ovol + oidx] = (float)acc;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) affine_resample_kernel(const float* __restrict__ src, float* __restrict__ dst, ResampleArgs a) {
#pragma clang fp contract(off)
    const long long ovol = (long long)a.Do * a.Ho * a.Wo;
    const long long oidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (oidx >= ovol) return;
    const int ox = (int)(oidx % a.Wo);
    const long long t = oidx / a.Wo;
    const int oy = (int)(t % a.Ho), oz = (int)(t / a.Ho);
    // same association as a 4-term matrix-vector product row: ((m0*z + m1*y) + m2*x) + m3
    const double cz = ((a.m[0] * oz + a.m[1] * oy) + a.m[2] * ox) + a.m[3];
    const double cy = ((a.m[4] * oz + a.m[5] * oy) + a.m[6] * ox) + a.m[7];
    const double cx = ((a.m[8] * oz + a.m[9] * oy) + a.m[10] * ox) + a.m[11];
    rs_sample<T>(src, dst, a, cz, cy, cx, oidx);
}

// Dense grid: coords [3][Do][Ho][Wo] (planes z, y, x), fp32 or fp64 (GT); a per-axis affine (ga, gb) turns the stored
// coordinate convention (centred voxel units or [-1, 1]) into source voxel indices without rewriting the grid.
template <typename T, typename GT>
__global__ void __launch_bounds__(256)
grid_resample_kernel(const float* __restrict__ src, const GT* __restrict__ coords, float* __restrict__ dst, ResampleArgs a) {
#pragma clang fp contract(off)
    const long long ovol = (long long)a.Do * a.Ho * a.Wo;
    const long long oidx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (oidx >= ovol) return;
    rs_sample<T>(src, dst, a, a.ga[0] * (double)coords[oidx] + a.gb[0], a.ga[1] * (double)coords[ovol + oidx] + a.gb[1],
                 a.ga[2] * (double)coords[2 * ovol + oidx] + a.gb[2], oidx);
}

}  // namespace mh
