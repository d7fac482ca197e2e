// Boundary conditions of the reference's native resampler (index / sign rules, monai/csrc/resample/bounds_common.h:30-243),
// shared by the kernels in pushpull.h.  Unlike ATen's grid_sampler the boundary condition acts on the INTEGER tap index:
// an index remap plus a sign (-1 / 0 / +1) that multiplies the value read or the value scattered.
#pragma once
#include "common.h"

namespace mh {

enum { GB_REPLICATE = 0, GB_DCT1 = 1, GB_DCT2 = 2, GB_DST1 = 3, GB_DST2 = 4, GB_DFT = 5, GB_SLIDING = 6, GB_ZERO = 7 };

struct GridPullArgs {
    int B, C, X, Y, Z, Xo, Yo, Zo;
    int bound[3], interp[3];
    int extrapolate;
};

// Every rule maps an in-range index to itself with sign +1: that case (nearly every tap of a sampling grid that stays
// inside the volume) returns before any of the 64-bit modulo arithmetic below.
__device__ __forceinline__ long long gp_index(int bound, long long c, long long n) {
    if (c >= 0 && c < n) return c;
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
            c = c == n ? n - 1 : (c > n ? t - c - 2 : c);
            return c < 0 ? 0 : c;      // c = 2n + 1 maps to -1 in the reference (a tap whose sign is 0, never read there); keep it loadable
        }
        case GB_DFT: return c < 0 ? (n + c % n) % n : c % n;
        default: return c < 0 ? 0 : (c >= n ? n - 1 : c);  // zero: the sign masks out-of-bound taps; the index stays loadable
    }
}

__device__ __forceinline__ int gp_sign(int bound, long long c, long long n) {
    if (c >= 0 && c < n) return 1;
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

// Branch-free: every remapped index is inside the volume, so the load is unconditional and the gathers of one voxel can
// be in flight together (a load guarded by the sign is waited for before the next one is even issued).
template <typename T> __device__ __forceinline__ T gp_get(const T* p, long long off, int sign) {
    const T v = p[off];
    return sign == -1 ? -v : (sign ? v : (T)0);
}

}  // namespace mh
