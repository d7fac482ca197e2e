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

// rs_axis<T>(coord, size, RS_LINEAR, RS_BORDER, .) without its range tests in floating point: after the clamp 0 <= c <= size - 1 (a NaN
// coordinate becomes 0 in fmax; rounding to float keeps the interval, its ends are integers below 2^24), so floor(c) is a valid index and
// only the upper tap can leave the volume.  Same taps, same weights.
template <typename T> __device__ __forceinline__ AxisTap<T> rs_axis_border_linear(double coord, int size) {
    AxisTap<T> t;
    const T c = (T)fmin((double)(size - 1), fmax(coord, 0.0));
    const T f = floor(c);
    t.i0 = (int)f;
    t.i1 = t.i0 + 1 < size ? t.i0 + 1 : -1;
    t.w0 = (f + (T)1) - c;
    t.w1 = c - f;
    return t;
}

// Two-term combine of one interpolation axis: a0*w0 + a1*w1 with the second product fused (one rounding less than the
// separate multiply-add; every kernel of this file uses this same form, so they agree bit for bit with each other).
__device__ __forceinline__ double rs_comb(double a0, double w0, double a1, double w1) { return fma(a1, w1, a0 * w0); }
__device__ __forceinline__ float rs_comb(float a0, float w0, float a1, float w1) { return fmaf(a1, w1, a0 * w0); }

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
            row[r] = rs_comb(a0, tx.w0, a1, tx.w1);
        }
        const T p0 = rs_comb(row[0], ty.w0, row[1], ty.w1), p1 = rs_comb(row[2], ty.w0, row[3], ty.w1);
        dst[(IDX)c * ovol + oidx] = (float)rs_comb(p0, tz.w0, p1, tz.w1);
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

// The same kernel for the common cases, laid out so that the per-voxel instruction stream is short -- the kernel above is VALU-issue
// bound (~170 vector instructions per output voxel, a third of them the two runtime integer divisions of the linear index and the
// branches over the runtime mode / padding rule):
//   * a workgroup is 4 waves = 4 output rows x 64 consecutive x; (ox, oy, oz) come from the block / thread indices: no division;
//   * interpolation mode and padding rule are template parameters: straight-line code, no reflection path in the binary;
//   * 32-bit offsets inside one channel volume (the launcher checks the extents).
// Coordinates, taps and the 8-corner combine are the routines of the kernel above, evaluated in the same order: bit-identical results
// (tests/transform_cases.py::case_separable_vs_general runs both against the separable path).
template <typename T, int MODE, int PAD>
__global__ void __launch_bounds__(256) affine_resample_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, ResampleArgs a) {
    const int ox = (int)blockIdx.x * 64 + (int)(threadIdx.x & 63);
    const int oy = (int)blockIdx.y * 4 + (int)(threadIdx.x >> 6);
    const int oz = (int)blockIdx.z;
    if (ox >= a.Wo || oy >= a.Ho) return;
    const int ovol = a.Do * a.Ho * a.Wo, ivol = a.Di * a.Hi * a.Wi;
    const int oidx = (oz * a.Ho + oy) * a.Wo + ox;
    const double cz = ((a.m[0] * oz + a.m[1] * oy) + a.m[2] * ox) + a.m[3];
    const double cy = ((a.m[4] * oz + a.m[5] * oy) + a.m[6] * ox) + a.m[7];
    const double cx = ((a.m[8] * oz + a.m[9] * oy) + a.m[10] * ox) + a.m[11];
    AxisTap<T> tz, ty, tx;
    if (MODE == RS_LINEAR && PAD == RS_BORDER) {
        tz = rs_axis_border_linear<T>(cz, a.Di); ty = rs_axis_border_linear<T>(cy, a.Hi); tx = rs_axis_border_linear<T>(cx, a.Wi);
        __builtin_assume(tz.i0 >= 0); __builtin_assume(ty.i0 >= 0); __builtin_assume(tx.i0 >= 0);
    } else {
        tz = rs_axis<T>(cz, a.Di, MODE, PAD, a.align_corners);
        ty = rs_axis<T>(cy, a.Hi, MODE, PAD, a.align_corners);
        tx = rs_axis<T>(cx, a.Wi, MODE, PAD, a.align_corners);
    }
    rs_gather<T, int>(src, dst, a.C, ivol, ovol, oidx, a.Hi, a.Wi, tz, ty, tx, MODE == RS_NEAREST);
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

// LDS-staged variant of the separable kernel: a workgroup owns a TOZ x TOY x TOX block of outputs, finds the
// bounding box of the source voxels its taps touch (min / max over the block's table entries), pulls that box into
// LDS with row-contiguous (coalesced) loads -- every source element once per workgroup instead of up to 8 gathers per
// output through L1 -- and takes the 8 corners from LDS.  If the box does not fit (strong down-sampling, reflection
// wrap-around) the workgroup falls back to global gathers; results are identical either way.
constexpr int RS_TOZ = 4, RS_TOY = 8, RS_TOX = 64, RS_LDS_FLOATS = 6144;

template <typename T>
__global__ void __launch_bounds__(256)
separable_resample_lds_kernel(const float* __restrict__ src, float* __restrict__ dst, const AxisTap<T>* __restrict__ tab, ResampleArgs a) {
    __shared__ float box[RS_LDS_FLOATS];
    __shared__ int lim[6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbx = (a.Wo + RS_TOX - 1) / RS_TOX, nby = (a.Ho + RS_TOY - 1) / RS_TOY;
    const int bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, bz = blockIdx.x / (nbx * nby);
    const int ox0 = bx * RS_TOX, oy0 = by * RS_TOY, oz0 = bz * RS_TOZ;
    const int nz = min(RS_TOZ, a.Do - oz0), ny = min(RS_TOY, a.Ho - oy0), nx = min(RS_TOX, a.Wo - ox0);
    const bool nearest = a.mode == RS_NEAREST;

    // bounding box of the taps.  Border / zeros padding give monotone tap tables, so the box is spanned by the first and
    // last entries of each axis; reflection can fold back, so wave w < 3 scans the entries of axis w.
    if (a.pad == RS_REFLECTION) {
        if (wave < 3) {
            const int n = wave == 0 ? nz : wave == 1 ? ny : nx;
            const int base = wave == 0 ? oz0 : wave == 1 ? a.Do + oy0 : a.Do + a.Ho + ox0;
            int lo = 0x7fffffff, hi = -1;
            if (lane < n) {
                const AxisTap<T> t = tab[base + lane];
                if (t.i0 >= 0) { lo = min(lo, t.i0); hi = max(hi, t.i0); }
                if (t.i1 >= 0) { lo = min(lo, t.i1); hi = max(hi, t.i1); }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                lo = min(lo, __shfl_xor(lo, o));
                hi = max(hi, __shfl_xor(hi, o));
            }
            if (lane == 0) { lim[2 * wave] = lo; lim[2 * wave + 1] = hi; }
        }
    } else if (tid < 3) {
        const int n = tid == 0 ? nz : tid == 1 ? ny : nx;
        const int base = tid == 0 ? oz0 : tid == 1 ? a.Do + oy0 : a.Do + a.Ho + ox0;
        int lo = 0x7fffffff, hi = -1;
        for (int i = 0; i < n; ++i) {          // first valid tap from the front ...
            const AxisTap<T> t = tab[base + i];
            const int v = t.i0 >= 0 ? t.i0 : t.i1;
            if (v >= 0) { lo = v; break; }
        }
        for (int i = n - 1; i >= 0; --i) {     // ... last valid tap from the back
            const AxisTap<T> t = tab[base + i];
            const int v = t.i1 >= 0 ? t.i1 : t.i0;
            if (v >= 0) { hi = v; break; }
        }
        // a decreasing table (negative scale) swaps the roles
        if (hi >= 0 && lo > hi) { const int s0 = lo; lo = hi; hi = s0; }
        if (hi >= 0) {   // taps of an entry pair are (i0, i0+1): widen by the partner tap on both ends
            const AxisTap<T> f = tab[base], l = tab[base + n - 1];
            const int c0 = f.i0 >= 0 ? f.i0 : lo, c1 = f.i1 >= 0 ? f.i1 : lo, c2 = l.i0 >= 0 ? l.i0 : hi, c3 = l.i1 >= 0 ? l.i1 : hi;
            lo = min(min(lo, c0), min(c1, min(c2, c3)));
            hi = max(max(hi, c0), max(c1, max(c2, c3)));
        }
        lim[2 * tid] = lo; lim[2 * tid + 1] = hi;
    }
    __syncthreads();
    const int lz = lim[0], ly = lim[2], lx = lim[4];
    const int ez = lim[1] - lz + 1, ey = lim[3] - ly + 1, ex = lim[5] - lx + 1;
    const bool any = lim[1] >= 0 && lim[3] >= 0 && lim[5] >= 0;      // otherwise every tap is out of the volume
    const bool staged = any && (long long)ez * ey * ex <= RS_LDS_FLOATS;
    const long long ivol = (long long)a.Di * a.Hi * a.Wi, ovol = (long long)a.Do * a.Ho * a.Wo;

    // this thread's outputs: column jx = lane, rows jy = wave and wave + 4, every jz of the block
    const int jx = lane;
    const AxisTap<T> tx = tab[a.Do + a.Ho + ox0 + min(jx, nx - 1)];
    const int xo0 = tx.i0 >= 0 ? tx.i0 - lx : 0, xo1 = tx.i1 >= 0 ? tx.i1 - lx : 0;

    for (int c = 0; c < a.C; ++c) {
        const float* p = src + (long long)c * ivol;
        if (staged) {
            __syncthreads();   // previous channel's reads are done
            const int rows = ez * ey;
            for (int x0 = 0; x0 < ex; x0 += 64) {                 // rows are contiguous along x: coalesced
                const bool xin = x0 + lane < ex;
                for (int r0 = wave; r0 < rows; r0 += 96) {        // up to 24 rows per wave in flight
                    float v[24];
#pragma unroll
                    for (int u = 0; u < 24; ++u) {
                        const int r = r0 + 4 * u;
                        const int rz = r / ey, ry = r - rz * ey;
                        const bool ok = xin && r < rows;
                        v[u] = ok ? p[((long long)(lz + rz) * a.Hi + (ly + ry)) * a.Wi + lx + x0 + lane] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 24; ++u) {
                        const int r = r0 + 4 * u;
                        if (xin && r < rows) box[r * ex + x0 + lane] = v[u];
                    }
                }
            }
            __syncthreads();
        }
        for (int jz = 0; jz < nz; ++jz) {
            const AxisTap<T> tz = tab[oz0 + jz];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int jy = wave + 4 * h;
                if (jy >= ny || jx >= nx) continue;
                const AxisTap<T> ty = tab[a.Do + oy0 + jy];
                const long long oidx = ((long long)(oz0 + jz) * a.Ho + (oy0 + jy)) * a.Wo + ox0 + jx;
                float res;
                if (nearest) {
                    const bool ok = tz.i0 >= 0 && ty.i0 >= 0 && tx.i0 >= 0;
                    if (!ok) res = 0.0f;
                    else if (staged) res = box[((tz.i0 - lz) * ey + (ty.i0 - ly)) * ex + xo0];
                    else res = p[((long long)tz.i0 * a.Hi + ty.i0) * a.Wi + tx.i0];
                } else {
                    const int zi[2] = {tz.i0, tz.i1}, yi[2] = {ty.i0, ty.i1}, xi[2] = {tx.i0, tx.i1};
                    const int xo[2] = {xo0, xo1};
                    T v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int z = zi[k >> 2], y = yi[(k >> 1) & 1], x = xi[k & 1];
                        const bool ok = z >= 0 && y >= 0 && x >= 0;
                        float f = 0.0f;
                        if (ok) f = staged ? box[((z - lz) * ey + (y - ly)) * ex + xo[k & 1]] : p[((long long)z * a.Hi + y) * a.Wi + x];
                        v[k] = (T)f;
                    }
                    const T r0 = rs_comb(v[0], tx.w0, v[1], tx.w1), r1 = rs_comb(v[2], tx.w0, v[3], tx.w1);
                    const T r2 = rs_comb(v[4], tx.w0, v[5], tx.w1), r3 = rs_comb(v[6], tx.w0, v[7], tx.w1);
                    const T p0 = rs_comb(r0, ty.w0, r1, ty.w1), p1 = rs_comb(r2, ty.w0, r3, ty.w1);
                    res = (float)rs_comb(p0, tz.w0, p1, tz.w1);
                }
                dst[(long long)c * ovol + oidx] = res;
            }
        }
    }
}

// Streaming variant for trilinear, non-reflecting, axis-aligned affines (the Spacingd case): trilinear interpolation
// factors into an in-plane (x, then y) interpolation of a SOURCE plane followed by a two-plane z combine.  A workgroup
// owns a 16 x 128 (oy, ox) output tile and marches along oz over one z-chunk:
//   source plane z: the tile's bounding box of (y, x) taps -> LDS (row-contiguous loads, prefetched into registers one
//   plane ahead, double-buffered: one barrier per plane) -> each thread interpolates its 8 outputs in-plane (4 LDS reads
//   at addresses fixed for the whole march + 3 combines each) into a register plane; the last two planes stay in
//   registers, an output plane is one more combine per voxel.  Every source plane is interpolated once per tile (the old
//   kernels redo the in-plane work and all index arithmetic for every output voxel): ~30 VALU instructions per output
//   instead of ~170, which is what bounded the per-voxel kernels (they were VALU-issue bound, not HBM bound).
// Arithmetic is the same rs_comb chain (x, y, z) on the same values as the other kernels: identical results.
constexpr int RZ_TOY = 16, RZ_TOX = 128;

// A thread's outputs are RJ rows x NH columns.  Scalar form (NH = 2): columns lane, lane + 64 of rows wave + NW j -- a store instruction writes 256 contiguous bytes.
// Vector form (round 6, NH = 4): columns 4 (lane & 31) .. + 3 of rows 2 (wave + NW j) + (lane >> 5) -- ONE 16-byte store per row and thread, a store instruction
// writes two whole 512-byte row segments (a quarter of the store instructions; the same values from the same arithmetic: bit-identical output).
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));      // output rows are 4-byte aligned only (Wo is arbitrary)
template <typename T, int RJ, int NH> struct RzTile {     // per-thread constants of the march
    int ad[RJ][NH];                        // LDS address of the (y0, x0) corner of output (row j, column h)
    int adn[RJ][NH];                       // ... of its (y1, x0) corner in the interior path (= ad + row pitch)
    int dx[NH], dy[RJ];                    // boundary tiles: address steps to the x1 / y1 corner (0 when that tap is dropped or clamped)
    T wx0[NH], wx1[NH], wy0[RJ], wy1[RJ];
    unsigned okm;                          // boundary tiles: bit (j*NH+h)*4 + corner set = corner contributes
};

// In-plane (x, then y) interpolation of the staged source plane at this thread's 8 outputs.  Interior tiles (every tap
// valid, so x1 = x0 + 1 and y1 = y0 + 1): two paired LDS reads per output at addresses fixed for the whole march.
template <typename T, int RJ, int NH, bool MASKED>
__device__ __forceinline__ void rz_interp_plane(const float* __restrict__ box, const RzTile<T, RJ, NH>& t, T (&P)[NH * RJ]) {
#pragma unroll
    for (int j = 0; j < RJ; ++j)
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            float v[4];
            if (MASKED) {
                const int a0 = t.ad[j][h], a1 = a0 + t.dy[j];
                v[0] = box[a0]; v[1] = box[a0 + t.dx[h]]; v[2] = box[a1]; v[3] = box[a1 + t.dx[h]];
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = ((t.okm >> ((j * NH + h) * 4 + k)) & 1u) ? v[k] : 0.0f;
            } else {
                const float* b0 = box + t.ad[j][h];
                const float* b1 = box + t.adn[j][h];
                v[0] = b0[0]; v[1] = b0[1]; v[2] = b1[0]; v[3] = b1[1];
            }
            const T r0 = rs_comb((T)v[0], t.wx0[h], (T)v[1], t.wx1[h]), r1 = rs_comb((T)v[2], t.wx0[h], (T)v[3], t.wx1[h]);
            P[j * NH + h] = rs_comb(r0, t.wy0[j], r1, t.wy1[j]);
        }
}

// NLOAD = staged floats per thread and plane (box capacity NT * NLOAD; chosen by the launcher from the scales); NT = threads
// Registers (round 6, profiles/r06_resample_variants.txt): the fp32 form with 512 threads needs 68 registers = 7 waves per SIMD = three workgroups per CU; asked for 8 waves
// per SIMD it fits 62 without scratch = four workgroups, which pays once the launch is cut into enough z-chunks (capi.hip).  The fp64 form (111 registers, four
// workgroups of 256) spills under any tighter cap and runs 1.7x slower: left alone.  -DMH_RS_MINW=n (tools/ubench/resample_variants.hip) overrides.
#ifndef MH_RS_MINW
#define MH_RS_BOUNDS(T_, NT_) __launch_bounds__(NT_, (sizeof(T_) == 4 && (NT_) == 512) ? 8 : 1)
#else
#define MH_RS_BOUNDS(T_, NT_) __launch_bounds__(NT_, MH_RS_MINW)
#endif
// A tap-table entry at a wave-uniform index through the constant address space: a SCALAR load (lgkmcnt).  As a vector load it sits in the vmcnt queue behind the
// LDS-DMA planes in flight, and waiting for it means waiting for all of them (the counter is in order).
template <typename T> __device__ __forceinline__ AxisTap<T> rs_tap_uniform(const AxisTap<T>* __restrict__ tab, int i) {
#if defined(MH_SIMT_EMULATOR) || !defined(__HIP_DEVICE_COMPILE__)
    return tab[i];
#else
    typedef const __attribute__((address_space(4))) AxisTap<T> ctap;
    return ((ctap*)tab)[__builtin_amdgcn_readfirstlane(i)];
#endif
}

// RING (round 6): 0 = the source plane is prefetched into registers one plane ahead and written to one of two LDS buffers; RING = 3: the planes arrive by LDS-DMA
// (common.h: no registers, no ds_write) into a ring of RING LDS slots, up to RING - 1 planes in flight behind the one being interpolated: a workgroup no longer waits a
// full memory latency per source plane.  Same values from the same arithmetic: bit-identical output.
template <typename T, int NLOAD, int NT, bool VEC = false, int RING = 0>
__global__ void MH_RS_BOUNDS(T, NT)
separable_resample_stream_kernel(const float* __restrict__ src, float* __restrict__ dst, const AxisTap<T>* __restrict__ tab, ResampleArgs a,
                                 int zchunk, int nchunk) {
    constexpr int CAP = NLOAD * NT, NW = NT / 64, NH = VEC ? 4 : 2, RJ = VEC ? RZ_TOY / (2 * NW) : RZ_TOY / NW;
    static_assert(RJ >= 1 && RJ * NH * 4 <= 32, "a thread's outputs: rows x columns, four corner bits each in one mask word");
    static_assert(RING == 0 || (RING == 3 && 2 * NLOAD <= 63), "the ring form: three slots, waits of 0 / NLOAD / 2 NLOAD outstanding loads");
    __shared__ float box[RING ? RING : 2][CAP];      // staged as fp32 also for fp64 interpolation (staging doubles measured slower: twice the LDS traffic)
    __shared__ int lim[5], part[6];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this thread's output (row j, column h) inside the tile
    auto rowy = [&](int j) { return VEC ? 2 * (wave + NW * j) + (lane >> 5) : wave + NW * j; };
    auto colx = [&](int h) { return VEC ? 4 * (lane & 31) + h : lane + 64 * h; };
    const int nbx = (a.Wo + RZ_TOX - 1) / RZ_TOX, nby = (a.Ho + RZ_TOY - 1) / RZ_TOY, tiles = nbx * nby;
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);        // neighbouring tiles share an XCD's L2 (their boxes overlap)
    const int tile = (int)(lid % (unsigned)tiles);
    lid /= (unsigned)tiles;
    const int chunk = (int)(lid % (unsigned)nchunk), c = (int)(lid / (unsigned)nchunk);
    const int ox0 = (tile % nbx) * RZ_TOX, oy0 = (tile / nbx) * RZ_TOY;
    const int ny = min(RZ_TOY, a.Ho - oy0), nx = min(RZ_TOX, a.Wo - ox0);
    const int oz_s = chunk * zchunk, oz_e = min(oz_s + zchunk, a.Do);
    const long long iplane = (long long)a.Hi * a.Wi, oplane = (long long)a.Ho * a.Wo;
    const float* p = src + (long long)c * a.Di * iplane;
    float* q = dst + (long long)c * a.Do * oplane;

    // bounding box of the tile's y and x taps: wave 0 reduces the <= 16 row taps, waves 1 and 2 the <= 128 column taps
    // (one tap per lane; a serial scan by one thread costs ~128 dependent-latency loads per workgroup)
    static_assert(RZ_TOY <= 64 && RZ_TOX <= 128, "tap bounding box: one lane per tap in waves 0 (rows) and 1-2 (columns)");
    if (tid == 0) lim[4] = 0;
    {
        int ti = -1;
        if (wave == 0) { if (lane < ny) ti = a.Do + oy0 + lane; }
        else if (wave <= 2) { if (lane + 64 * (wave - 1) < nx) ti = a.Do + a.Ho + ox0 + lane + 64 * (wave - 1); }
        int lo = 0x7fffffff, hi = -1;
        if (ti >= 0) {
            const AxisTap<T> e = tab[ti];
            if (e.i0 >= 0) { lo = min(lo, e.i0); hi = max(hi, e.i0); }
            if (e.i1 >= 0) { lo = min(lo, e.i1); hi = max(hi, e.i1); }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            lo = min(lo, __shfl_xor(lo, o));
            hi = max(hi, __shfl_xor(hi, o));
        }
        if (lane == 0 && wave <= 2) { part[2 * wave] = lo; part[2 * wave + 1] = hi; }
    }
    __syncthreads();
    if (tid == 0) {
        lim[0] = part[0]; lim[1] = part[1];
        lim[2] = min(part[2], part[4]); lim[3] = max(part[3], part[5]);
    }
    __syncthreads();
    const int ly = lim[0], lx = lim[2];
    const int ey = lim[1] - ly + 1, ex = lim[3] - lx + 1;
    const bool any = lim[1] >= 0 && lim[3] >= 0;      // otherwise a whole axis of the tile samples outside the volume
    const bool staged = any && ey * ex <= CAP;

    if (!staged) {
        // every tap of one axis outside the volume (all zeros), or a box larger than the LDS plane: per-voxel gathers
        for (int oz = oz_s; oz < oz_e; ++oz) {
            const AxisTap<T> tz = tab[oz];
            for (int j = 0; j < RJ; ++j)
                for (int h = 0; h < NH; ++h) {
                    const int jy = rowy(j), jx = colx(h);
                    if (jy >= ny || jx >= nx) continue;
                    const AxisTap<T> ty = tab[a.Do + oy0 + jy], tx = tab[a.Do + a.Ho + ox0 + jx];
                    const int zi[2] = {tz.i0, tz.i1}, yi[2] = {ty.i0, ty.i1}, xi[2] = {tx.i0, tx.i1};
                    T v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int z = zi[k >> 2], y = yi[(k >> 1) & 1], x = xi[k & 1];
                        v[k] = (z >= 0 && y >= 0 && x >= 0) ? (T)p[(long long)z * iplane + (long long)y * a.Wi + x] : (T)0;
                    }
                    const T r0 = rs_comb(v[0], tx.w0, v[1], tx.w1), r1 = rs_comb(v[2], tx.w0, v[3], tx.w1);
                    const T r2 = rs_comb(v[4], tx.w0, v[5], tx.w1), r3 = rs_comb(v[6], tx.w0, v[7], tx.w1);
                    const T p0 = rs_comb(r0, ty.w0, r1, ty.w1), p1 = rs_comb(r2, ty.w0, r3, ty.w1);
                    q[(long long)oz * oplane + (long long)(oy0 + jy) * a.Wo + ox0 + jx] = (float)rs_comb(p0, tz.w0, p1, tz.w1);
                }
        }
        return;
    }

    // per-thread march constants: columns colx(h), rows rowy(j)
    RzTile<T, RJ, NH> t;
    t.okm = 0u;
    bool bad = false;
    {
        int xa[NH], ya[RJ];
        bool xok[NH][2], yok[RJ][2];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const AxisTap<T> e = tab[a.Do + a.Ho + ox0 + min(colx(h), nx - 1)];
            xok[h][0] = e.i0 >= 0; xok[h][1] = e.i1 >= 0;
            xa[h] = (xok[h][0] ? e.i0 : xok[h][1] ? e.i1 : lx) - lx;
            t.dx[h] = (xok[h][0] && xok[h][1]) ? e.i1 - e.i0 : 0;
            t.wx0[h] = e.w0; t.wx1[h] = e.w1;
        }
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            const AxisTap<T> e = tab[a.Do + oy0 + min(rowy(j), ny - 1)];
            yok[j][0] = e.i0 >= 0; yok[j][1] = e.i1 >= 0;
            ya[j] = (yok[j][0] ? e.i0 : yok[j][1] ? e.i1 : ly) - ly;
            t.dy[j] = (yok[j][0] && yok[j][1]) ? (e.i1 - e.i0) * ex : 0;
            t.wy0[j] = e.w0; t.wy1[j] = e.w1;
        }
#pragma unroll
        for (int j = 0; j < RJ; ++j)
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                t.ad[j][h] = ya[j] * ex + xa[h];
                t.adn[j][h] = t.ad[j][h] + ex;
                const unsigned m = (unsigned)(yok[j][0] && xok[h][0]) | (unsigned)(yok[j][0] && xok[h][1]) << 1 |
                                   (unsigned)(yok[j][1] && xok[h][0]) << 2 | (unsigned)(yok[j][1] && xok[h][1]) << 3;
                t.okm |= m << ((j * NH + h) * 4);
                bad = bad || m != 15u;
            }
    }
    if (bad) lim[4] = 1;            // benign race: every writer stores the same value
    __syncthreads();
    const bool masked = lim[4] != 0;

    // loader: element i = tid + NT j of the box, row-major with pitch ex
    int goff[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
        const int i = tid + NT * j;
        const int r = i / ex, col = i - r * ex;
        goff[j] = i < ey * ex ? (ly + r) * a.Wi + lx + col : -1;
    }
    float pre[RING ? 1 : NLOAD];
    int pre_z = -1, buf = 0;
    const int dir = a.m[0] < 0.0 ? -1 : 1;
    T Pa[NH * RJ], Pb[NH * RJ];
    int cur0 = -1, cur1 = -1;
    // RING: the window of planes in flight -- plane r_head + k dir sits in (or is on its way into) slot (r_hs + k) % RING for k < r_cnt; all of it wave-uniform.
    // Lanes beyond the box fetch the box's first element (every lane issues every load: the vmcnt arithmetic below counts instructions) and their cells are never read.
    int r_head = 0, r_hs = 0, r_cnt = 0;
    int z_end = dir > 0 ? a.Di - 1 : 0;          // last source plane (in march direction) any output of this chunk reads: nothing beyond it is requested
    if constexpr (RING != 0) {
        const AxisTap<T> tl = rs_tap_uniform(tab, oz_e - 1);
        const int zl = dir > 0 ? max(tl.i0, tl.i1) : ((tl.i0 >= 0 && tl.i1 >= 0) ? min(tl.i0, tl.i1) : max(tl.i0, tl.i1));
        if (zl >= 0) z_end = zl;
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) goff[j] = 4 * (goff[j] >= 0 ? goff[j] : ly * a.Wi + lx);      // BYTE offsets inside a plane: scalar plane base + 32-bit lane offset, no 64-bit address registers
    }
#define RZ_RING_ISSUE(Z, SLOT)                                                                  \
    {                                                                                           \
        /* the plane index comes out of the tap table (a vector load): tell the compiler it is wave-uniform, or the plane base is not a scalar (buffer descriptor) */ \
        const float* pl_ = p + (long long)__builtin_amdgcn_readfirstlane(Z) * iplane;           \
        float* sl_ = box[__builtin_amdgcn_readfirstlane(SLOT)] + 64 * __builtin_amdgcn_readfirstlane(wave); \
        _Pragma("unroll") for (int j = 0; j < NLOAD; ++j) MH_LDS_DMA_F32(pl_, goff[j], sl_ + NT * j); \
    }
#define RZ_RING_PLANE(Z, P)                                                                     \
    {                                                                                           \
        int k_ = ((Z) - r_head) * dir;                                                          \
        if (r_cnt == 0 || k_ < 0 || k_ >= r_cnt) {      /* not in flight (the first plane; a z step beyond the window): drain, make sure nobody reads a slot, restart */ \
            MH_VMCNT_BARRIER(0);                                                                \
            r_head = (Z); r_hs = 0; r_cnt = 1;                                                  \
            RZ_RING_ISSUE((Z), 0)                                                               \
        } else {                                                                                \
            r_head = (Z); r_hs = (r_hs + k_) % RING; r_cnt -= k_;                               \
        }                                                                                       \
        /* plane Z has landed once at most (r_cnt - 1) NLOAD younger loads are outstanding (result stores issued in between only make the wait more conservative) */ \
        if (r_cnt == 1) MH_VMCNT_BARRIER(0);                                                    \
        else if (r_cnt == 2) MH_VMCNT_BARRIER(NLOAD);                                           \
        else MH_VMCNT_BARRIER(2 * NLOAD);                                                       \
        /* every wave is past its reads of the slot consumed before this one: top the window up */ \
        while (r_cnt < RING) {                                                                  \
            const int zn_ = r_head + r_cnt * dir;                                               \
            if (zn_ < 0 || zn_ >= a.Di || (zn_ - z_end) * dir > 0) break;                       \
            RZ_RING_ISSUE(zn_, (r_hs + r_cnt) % RING)                                           \
            ++r_cnt;                                                                            \
        }                                                                                       \
        if (masked) rz_interp_plane<T, RJ, NH, true>(box[r_hs], t, P);                          \
        else rz_interp_plane<T, RJ, NH, false>(box[r_hs], t, P);                                \
    }

#define RZ_LOAD_PLANE(Z)                                                                        \
    {                                                                                           \
        const float* pl_ = p + (long long)(Z) * iplane;                                         \
        _Pragma("unroll") for (int j = 0; j < NLOAD; ++j) pre[j] = goff[j] >= 0 ? pl_[goff[j]] : 0.0f; \
        pre_z = (Z);                                                                            \
    }
#define RZ_COMPUTE_PLANE(Z, P)                                                                  \
    if constexpr (RING != 0) RZ_RING_PLANE(Z, P) else {                                         \
        if (pre_z != (Z)) RZ_LOAD_PLANE(Z)                                                      \
        _Pragma("unroll") for (int j = 0; j < NLOAD; ++j) if (goff[j] >= 0) box[buf][tid + NT * j] = pre[j]; \
        __syncthreads();                                                                        \
        const int zn_ = (Z) + dir;                                                              \
        if (zn_ >= 0 && zn_ < a.Di) RZ_LOAD_PLANE(zn_)                                          \
        if (masked) rz_interp_plane<T, RJ, NH, true>(box[buf], t, P);                           \
        else rz_interp_plane<T, RJ, NH, false>(box[buf], t, P);                                 \
        buf ^= 1;                                                                               \
    }

    for (int oz = oz_s; oz < oz_e; ++oz) {
        const AxisTap<T> tz = RING ? rs_tap_uniform(tab, oz) : tab[oz];
        const bool z0ok = tz.i0 >= 0, z1ok = tz.i1 >= 0;
        if (z0ok || z1ok) {
            const int z0 = z0ok ? tz.i0 : tz.i1, z1 = z1ok ? tz.i1 : tz.i0;
            if (z1 != cur1 && z1 == cur0) {          // decreasing table: the old lower plane becomes the upper one
#pragma unroll
                for (int k = 0; k < NH * RJ; ++k) Pb[k] = Pa[k];
                cur1 = cur0;
            }
            if (z0 != cur0) {
                if (z0 == cur1) {
#pragma unroll
                    for (int k = 0; k < NH * RJ; ++k) Pa[k] = Pb[k];
                } else RZ_COMPUTE_PLANE(z0, Pa)
                cur0 = z0;
            }
            if (z1 != cur1) {
                if (z1 == cur0) {
#pragma unroll
                    for (int k = 0; k < NH * RJ; ++k) Pb[k] = Pa[k];
                } else RZ_COMPUTE_PLANE(z1, Pb)
                cur1 = z1;
            }
        }
#pragma unroll
        for (int j = 0; j < RJ; ++j) {
            float res[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                res[h] = 0.0f;
                if (z0ok || z1ok) {
                    const T p0 = z0ok ? Pa[j * NH + h] : (T)0, p1 = z1ok ? Pb[j * NH + h] : (T)0;
                    res[h] = (float)rs_comb(p0, tz.w0, p1, tz.w1);
                }
            }
            const int jy = rowy(j);
            if (jy >= ny) continue;
            float* qrow = q + (long long)oz * oplane + (long long)(oy0 + jy) * a.Wo + ox0;
            if (VEC && colx(NH - 1) < nx) {
                *reinterpret_cast<f32x4_a4*>(qrow + colx(0)) = f32x4_a4{res[0], res[1], res[2], res[NH - 1]};
            } else {
#pragma unroll
                for (int h = 0; h < NH; ++h)
                    if (colx(h) < nx) qrow[colx(h)] = res[h];
            }
        }
    }
    if constexpr (RING != 0) MH_VMCNT_WAIT(0);      // no LDS-DMA may be in flight when the workgroup's LDS is handed to the next one
#undef RZ_LOAD_PLANE
#undef RZ_COMPUTE_PLANE
#undef RZ_RING_PLANE
#undef RZ_RING_ISSUE
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
