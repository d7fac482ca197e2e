// grid_pull (interpolation order 0 / 1) with the boundary conditions of the reference's native resampler.
//
// Reference: monai::grid_pull (monai/csrc/resample/pushpull.h:58-110) -> pushpull_cpu.cpp
//   check3d :786-838 (extrapolate / TINY in-bounds test), interpolate3d_trilinear :1476-1675,
//   interpolate3d_nearest :2057-2098, index/sign rules monai/csrc/resample/bounds_common.h:30-243.
// Unlike ATen's grid_sampler the boundary condition acts on the INTEGER tap index (an index remap plus a sign),
// coordinates are voxel indices in tensor-axis order, and the sample is  sum_corners sign * value * weight  in the
// order 000,100,010,110,001,101,011,111 (first tensor axis first).  Input (B,C,X,Y,Z), grid (B,Xo,Yo,Zo,3)
// interleaved, output (B,C,Xo,Yo,Zo); fp32 or fp64 end to end like the reference's dispatch.
// One thread per output voxel, channels looped (the reference's kernel shape, but lanes run along the LAST axis:
// coalesced grid reads (3 x 4 B interleaved) and output writes).
#pragma once
#include "common.h"

namespace mh {

enum { GB_REPLICATE = 0, GB_DCT1 = 1, GB_DCT2 = 2, GB_DST1 = 3, GB_DST2 = 4, GB_DFT = 5, GB_SLIDING = 6, GB_ZERO = 7 };

struct GridPullArgs {
    int B, C, X, Y, Z, Xo, Yo, Zo;
    int bound[3], interp[3];
    int extrapolate;
};

__device__ __forceinline__ long long gp_index(int bound, long long c, long long n) {
    switch (bound) {
        case GB_REPLICATE: return c <= 0 ? 0 : (c >= n ? n - 1 : c);
        case GB_DCT1: {
            if (n == 1) return 0;
            const long long t = (n - 1) * 2;
            c = c < 0 ? -c : c;
            c = c % t;
            return c >= n ? t - c : c;
        }
        case GB_DCT2:
        case GB_DST2: {
            const long long t = n * 2;
            c = c < 0 ? t - ((-c - 1) % t) - 1 : c % t;
            return c >= n ? t - c - 1 : c;
        }
        case GB_DST1: {
            if (n == 1) return 0;
            const long long t = (n + 1) * 2;
            c = c == -1 ? 0 : (c < 0 ? -c - 2 : c);
            c = c % t;
            return c == n ? n - 1 : (c > n ? t - c - 2 : c);
        }
        case GB_DFT: return c < 0 ? (n + c % n) % n : c % n;
        default: return c;  // zero: the sign masks out-of-bound taps
    }
}

__device__ __forceinline__ int gp_sign(int bound, long long c, long long n) {
    switch (bound) {
        case GB_DST1: {
            if (n == 1) return 1;
            const long long t = (n + 1) * 2;
            c = c < 0 ? n - c - 1 : c;
            c = c % t;
            if (c % (n + 1) == n) return 0;
            return ((c / (n + 1)) % 2) ? -1 : 1;
        }
        case GB_DST2: {
            c = c < 0 ? n - c - 1 : c;
            return ((c / n) % 2) ? -1 : 1;
        }
        case GB_ZERO: return (c < 0 || c >= n) ? 0 : 1;
        case GB_REPLICATE:
        case GB_DCT1:
        case GB_DCT2:
        case GB_DFT: return 1;
        default: return (c < 0 || c >= n) ? 0 : 1;
    }
}

template <typename T> __device__ __forceinline__ T gp_get(const T* p, long long off, int sign) {
    return sign == -1 ? -p[off] : (sign ? p[off] : (T)0);
}

template <typename T>
__global__ void __launch_bounds__(256) grid_pull_kernel(const T* __restrict__ src, const T* __restrict__ grid, T* __restrict__ out, GridPullArgs a) {
#pragma clang fp contract(off)
    const long long ovol = (long long)a.Xo * a.Yo * a.Zo, ivol = (long long)a.X * a.Y * a.Z;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ovol * a.B) return;
    const long long n = idx / ovol, o = idx % ovol;
    const T* g = grid + idx * 3;
    const T x = g[0], y = g[1], z = g[2];
    const T* sp = src + n * a.C * ivol;
    T* op = out + n * a.C * ovol + o;
    const T tiny = (T)5e-2;
    const bool inb = x >= -tiny && x < (T)(a.X - 1) + tiny && y >= -tiny && y < (T)(a.Y - 1) + tiny && z >= -tiny && z < (T)(a.Z - 1) + tiny;
    if (!(a.extrapolate || inb)) {
        for (int c = 0; c < a.C; ++c) op[c * ovol] = (T)0;
        return;
    }
    const long long sX = (long long)a.Y * a.Z, sY = a.Z;
    // per-axis taps: order 0 -> one tap at round(coord) with weight 1; order 1 -> floor / floor+1
    long long i0[3], i1[3];
    int s0[3], s1[3];
    T w0[3], w1[3];
    const T cc[3] = {x, y, z};
    const long long nn[3] = {a.X, a.Y, a.Z};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (a.interp[d] == 0) {
            const long long r = (long long)round((double)cc[d]);
            s0[d] = gp_sign(a.bound[d], r, nn[d]);
            i0[d] = gp_index(a.bound[d], r, nn[d]);
            s1[d] = 0; i1[d] = 0;
            w0[d] = (T)1; w1[d] = (T)0;
        } else {
            const long long f = (long long)floor((double)cc[d]);
            const T d1 = cc[d] - (T)f;
            w1[d] = d1; w0[d] = (T)1 - d1;
            s1[d] = gp_sign(a.bound[d], f + 1, nn[d]);
            s0[d] = gp_sign(a.bound[d], f, nn[d]);
            i1[d] = gp_index(a.bound[d], f + 1, nn[d]);
            i0[d] = gp_index(a.bound[d], f, nn[d]);
        }
    }
    const bool all_nearest = a.interp[0] == 0 && a.interp[1] == 0 && a.interp[2] == 0;
    if (all_nearest) {
        const int s = s0[2] * s0[1] * s0[0];
        const long long off = i0[0] * sX + i0[1] * sY + i0[2];
        for (int c = 0; c < a.C; ++c) op[c * ovol] = gp_get(sp + c * ivol, off, s);
        return;
    }
    // corner order of interpolate3d_trilinear: 000,100,010,110,001,101,011,111 (x = first tensor axis)
    long long off[8];
    int sg[8];
    T w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int bx = k & 1, by = (k >> 1) & 1, bz = k >> 2;
        off[k] = (bx ? i1[0] : i0[0]) * sX + (by ? i1[1] : i0[1]) * sY + (bz ? i1[2] : i0[2]);
        sg[k] = (bx ? s1[0] : s0[0]) * (by ? s1[1] : s0[1]) * (bz ? s1[2] : s0[2]);
        w[k] = (bx ? w1[0] : w0[0]) * (by ? w1[1] : w0[1]) * (bz ? w1[2] : w0[2]);
    }
    for (int c = 0; c < a.C; ++c) {
        const T* p = sp + c * ivol;
        T acc = gp_get(p, off[0], sg[0]) * w[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) acc = acc + gp_get(p, off[k], sg[k]) * w[k];
        op[c * ovol] = acc;
    }
}

}  // namespace mh
