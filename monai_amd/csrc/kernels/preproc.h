// Intensity / crop pre-processing in front of the sliding-window path: ScaleIntensityRange
// (monai/transforms/intensity/array.py:958-1012), the foreground bounding box of CropForeground
// (generate_spatial_bounding_box, monai/transforms/utils.py:1069-1129) and its crop + constant pad
// (monai/transforms/croppad/array.py:776-960), the flip + axis permutation of Orientation, NormalizeIntensity, ScaleIntensity.  Channel-first fp32 volumes [C][D][H][W], lanes along W.
// All three are HBM-bound: scale reads 4 B + writes 4 B per voxel, the box reads C x 4 B per voxel and writes 24 B per
// workgroup, crop+pad reads <= 4 B and writes 4 B per output voxel.
#pragma once
#include "common.h"

namespace mh {

// y = (x - a_min) / div [ * b_scale + b_min ] [ clamp ]: the reference's operator sequence with every rounding kept
// (true division, separate multiply and add: the library is built with -ffp-contract=off); NaN passes through the clamp
// like torch.clamp.  VEC: 16-byte loads / stores, four voxels per lane.
struct ScaleRange {
    float a_min, div, b_scale, b_min, lo, hi;
    int rescale, clip_lo, clip_hi;
};
__device__ __forceinline__ float scale_range_one(float x, const ScaleRange& p) {
    float v = (x - p.a_min) / p.div;
    if (p.rescale) { v = v * p.b_scale; v = v + p.b_min; }
    if (p.clip_lo && v < p.lo) v = p.lo;
    if (p.clip_hi && v > p.hi) v = p.hi;
    return v;
}
template <bool VEC>
__global__ void __launch_bounds__(256) scale_range_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n, ScaleRange p) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (VEC && i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(src + i);
        float4 r;
        r.x = scale_range_one(v.x, p); r.y = scale_range_one(v.y, p); r.z = scale_range_one(v.z, p); r.w = scale_range_one(v.w, p);
        *reinterpret_cast<float4*>(dst + i) = r;
        return;
    }
    for (long long j = i; j < n && j < i + 4; ++j) dst[j] = scale_range_one(src[j], p);
}

// Foreground box, stage 1: {zmin, ymin, xmin, zmax, ymax, xmax} of the voxels with ANY channel > 0 (NaN is not foreground,
// as `img > 0`), per workgroup.  A wave owns one (z, y) row at a time, lanes stride along x; every lane keeps private
// extrema, merged once per workgroup through LDS.  No global atomics: stage 2 folds the per-workgroup records.
#define MH_BOX_EMPTY_MIN 0x7fffffff
template <bool VEC>
__global__ void __launch_bounds__(256) bbox_partial_kernel(const float* __restrict__ src, int C, int D, int H, int W, int* __restrict__ partial) {
    __shared__ int box[6];
    if (threadIdx.x < 3) box[threadIdx.x] = MH_BOX_EMPTY_MIN;
    else if (threadIdx.x < 6) box[threadIdx.x] = -1;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long rows = (long long)D * H, cs = rows * W;
    int zmin = MH_BOX_EMPTY_MIN, ymin = MH_BOX_EMPTY_MIN, xmin = MH_BOX_EMPTY_MIN, zmax = -1, ymax = -1, xmax = -1;
    for (long long r = (long long)blockIdx.x * 4 + wave; r < rows; r += (long long)gridDim.x * 4) {
        const int z = (int)(r / H), y = (int)(r % H);
        const float* row = src + r * W;
        int lo = MH_BOX_EMPTY_MIN, hi = -1;
        if (VEC) {
            for (int x = lane * 4; x < W; x += 256) {
                bool f0 = false, f1 = false, f2 = false, f3 = false;
                for (int c = 0; c < C; ++c) {
                    const float4 v = *reinterpret_cast<const float4*>(row + (long long)c * cs + x);
                    f0 |= v.x > 0.0f; f1 |= v.y > 0.0f; f2 |= v.z > 0.0f; f3 |= v.w > 0.0f;
                }
                if (f0 | f1 | f2 | f3) {
                    const int first = f0 ? x : f1 ? x + 1 : f2 ? x + 2 : x + 3;
                    const int last = f3 ? x + 3 : f2 ? x + 2 : f1 ? x + 1 : x;
                    lo = min(lo, first); hi = max(hi, last);
                }
            }
        } else {
            for (int x = lane; x < W; x += 64) {
                bool f = false;
                for (int c = 0; c < C; ++c) f |= row[(long long)c * cs + x] > 0.0f;
                if (f) { lo = min(lo, x); hi = max(hi, x); }
            }
        }
        if (hi >= 0) {
            xmin = min(xmin, lo); xmax = max(xmax, hi);
            zmin = min(zmin, z); zmax = max(zmax, z);
            ymin = min(ymin, y); ymax = max(ymax, y);
        }
    }
    if (zmax >= 0) {
        atomicMin(&box[0], zmin); atomicMin(&box[1], ymin); atomicMin(&box[2], xmin);
        atomicMax(&box[3], zmax); atomicMax(&box[4], ymax); atomicMax(&box[5], xmax);
    }
    __syncthreads();
    if (threadIdx.x < 6) partial[(long long)blockIdx.x * 6 + threadIdx.x] = box[threadIdx.x];
}

// stage 2: one workgroup folds the `nparts` records into out[6] (zmax == -1: no foreground at all)
__global__ void __launch_bounds__(256) bbox_final_kernel(const int* __restrict__ partial, int nparts, int* __restrict__ out) {
    __shared__ int box[6];
    if (threadIdx.x < 3) box[threadIdx.x] = MH_BOX_EMPTY_MIN;
    else if (threadIdx.x < 6) box[threadIdx.x] = -1;
    __syncthreads();
    int mn[3] = {MH_BOX_EMPTY_MIN, MH_BOX_EMPTY_MIN, MH_BOX_EMPTY_MIN}, mx[3] = {-1, -1, -1};
    for (int b = threadIdx.x; b < nparts; b += 256)
        for (int k = 0; k < 3; ++k) { mn[k] = min(mn[k], partial[b * 6 + k]); mx[k] = max(mx[k], partial[b * 6 + 3 + k]); }
    if (mx[0] >= 0)
        for (int k = 0; k < 3; ++k) { atomicMin(&box[k], mn[k]); atomicMax(&box[3 + k], mx[k]); }
    __syncthreads();
    if (threadIdx.x < 6) out[threadIdx.x] = box[threadIdx.x];
}

// dst[c][z][y][x] = src[c][z + sz][y + sy][x + sx] inside the source, `value` outside: SpatialCrop of the clipped box and
// the constant pad of the part that sticks out, in one pass.  blockIdx.x = output row (c, z, y), blockIdx.y = x chunk.
__global__ void __launch_bounds__(256) crop_pad_kernel(const float* __restrict__ src, float* __restrict__ dst, int D, int H, int W, int Do, int Ho,
                                                       int Wo, int sz, int sy, int sx, float value) {
    const int x = blockIdx.y * 256 + threadIdx.x;
    if (x >= Wo) return;
    const long long row = blockIdx.x;
    const int y = (int)(row % Ho), z = (int)((row / Ho) % Do);
    const long long c = row / ((long long)Ho * Do);
    const int iz = z + sz, iy = y + sy, ix = x + sx;
    float v = value;
    if (iz >= 0 && iz < D && iy >= 0 && iy < H && ix >= 0 && ix < W) v = src[((c * D + iz) * H + iy) * (long long)W + ix];
    dst[row * Wo + x] = v;
}

// Orientation (monai/transforms/spatial/functional.py:187-229): torch.flip + permute of the spatial axes as one strided
// gather.  dst is dense [C][Do][Ho][Wo]; output axis k walks the source with the signed element stride s_k from `base`.
// Pure flips keep the innermost axis innermost (|sx| == 1): the wave reads one contiguous row, forwards or backwards.
// blockIdx.x = output row (c, z, y), blockIdx.y = x chunk.
__global__ void __launch_bounds__(256) flip_permute_kernel(const float* __restrict__ src, float* __restrict__ dst, int Do, int Ho, int Wo,
                                                           long long c_stride, long long base, long long sz, long long sy, long long sx) {
    const int x = blockIdx.y * 256 + threadIdx.x;
    if (x >= Wo) return;
    const long long row = blockIdx.x;
    const int y = (int)(row % Ho), z = (int)((row / Ho) % Do);
    const long long c = row / ((long long)Ho * Do);
    dst[row * Wo + x] = src[c * c_stride + base + z * sz + y * sy + x * sx];
}

// NormalizeIntensity (monai/transforms/intensity/array.py:816-907): (x - mean) / std over the whole image or per channel,
// optionally over the non-zero voxels only (which are then the only ones rewritten).  Statistics: fp64 sums {count, sum, sum of
// squares} per lane, wave shuffles, one record per workgroup; a one-wave finalize per channel turns the records into the fp32
// {subtrahend, divisor} pair the reference gets from torch.mean / torch.std(unbiased=False) (divisor 0 -> 1), left in device
// memory for the apply pass: no host round trip between the two.  Both passes are HBM-bound: 4 B read per voxel, then 4 + 4 B.
template <bool NONZERO, bool VEC>
__global__ void __launch_bounds__(256) masked_stats_kernel(const float* __restrict__ src, long long n, double* __restrict__ partial) {
    const float* p = src + (long long)blockIdx.y * n;
    double cnt = 0.0, s = 0.0, ss = 0.0;
    const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    if (VEC) {
        for (long long i = t0; i < n / 4; i += step) {
            const float4 v = *reinterpret_cast<const float4*>(p + 4 * i);
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (!NONZERO || e[k] != 0.0f) { cnt += 1.0; s += (double)e[k]; ss += (double)e[k] * (double)e[k]; }
        }
    } else {
        for (long long i = t0; i < n; i += step) {
            const float v = p[i];
            if (!NONZERO || v != 0.0f) { cnt += 1.0; s += (double)v; ss += (double)v * (double)v; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o); s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    __shared__ double red[4][3];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[wave][0] = cnt; red[wave][1] = s; red[wave][2] = ss; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double* o = partial + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        o[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

// one wave per channel: fold `nparts` records -> subdiv[c] = {mean, std or 1}
__global__ void __launch_bounds__(64) masked_stats_finalize_kernel(const double* __restrict__ partial, int nparts, float* __restrict__ subdiv) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const double* rec = partial + (long long)c * nparts * 3;
    double cnt = 0.0, s = 0.0, ss = 0.0;
    for (int i = lane; i < nparts; i += 64) { cnt += rec[i * 3]; s += rec[i * 3 + 1]; ss += rec[i * 3 + 2]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o); s += __shfl_xor(s, o); ss += __shfl_xor(ss, o); }
    if (lane == 0) {
        const double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float sd = (float)sqrt(var);
        subdiv[2 * c] = (float)mean;
        subdiv[2 * c + 1] = sd == 0.0f ? 1.0f : sd;      // array.py:866-868 (a NaN std stays NaN, as in the reference)
    }
}

template <bool NONZERO, bool VEC>
__global__ void __launch_bounds__(256) masked_normalize_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n,
                                                               const float* __restrict__ subdiv) {
    const long long off = (long long)blockIdx.y * n;
    const float sub = subdiv[2 * blockIdx.y], div = subdiv[2 * blockIdx.y + 1];
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (VEC && i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(src + off + i);
        float4 r;
        r.x = (NONZERO && v.x == 0.0f) ? v.x : (v.x - sub) / div;
        r.y = (NONZERO && v.y == 0.0f) ? v.y : (v.y - sub) / div;
        r.z = (NONZERO && v.z == 0.0f) ? v.z : (v.z - sub) / div;
        r.w = (NONZERO && v.w == 0.0f) ? v.w : (v.w - sub) / div;
        *reinterpret_cast<float4*>(dst + off + i) = r;
        return;
    }
    for (long long j = i; j < n && j < i + 4; ++j) {
        const float v = src[off + j];
        dst[off + j] = (NONZERO && v == 0.0f) ? v : (v - sub) / div;
    }
}

// ScaleIntensity (monai/transforms/intensity/array.py:445-491 -> rescale_array, monai/transforms/utils.py:229-251): min / max of the whole
// image or of each channel, then (x - min) / (max - min) [* (maxv - minv) + minv].  Exact reductions (NaN propagates like torch.min / max),
// one {min, max} record per workgroup, a one-wave fold per channel that leaves {min, max} in device memory; the apply pass forms
// max - min in fp32 itself (the reference's rounding) and takes the reference's `min == max` branch (x * minv, or x) per channel.
template <bool VEC>
__global__ void __launch_bounds__(256) minmax_partial_kernel(const float* __restrict__ src, long long n, float* __restrict__ partial) {
    const float* p = src + (long long)blockIdx.y * n;
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    const long long t0 = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    if (VEC) {
        for (long long i = t0; i < n / 4; i += step) {
            const float4 v = *reinterpret_cast<const float4*>(p + 4 * i);
            nan |= (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
            mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
            mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
        }
    } else {
        for (long long i = t0; i < n; i += step) {
            const float v = p[i];
            nan |= v != v;
            mn = fminf(mn, v); mx = fmaxf(mx, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); nan |= __shfl_xor(nan, o);
    }
    __shared__ float red[4][2];
    __shared__ int red_nan[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[wave][0] = mn; red[wave][1] = mx; red_nan[wave] = nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool any_nan = (red_nan[0] | red_nan[1] | red_nan[2] | red_nan[3]) != 0;
        float* o = partial + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        o[0] = any_nan ? NAN : fminf(fminf(red[0][0], red[1][0]), fminf(red[2][0], red[3][0]));
        o[1] = any_nan ? NAN : fmaxf(fmaxf(red[0][1], red[1][1]), fmaxf(red[2][1], red[3][1]));
    }
}
__global__ void __launch_bounds__(64) minmax_final_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ table) {
    if (threadIdx.x != 0) return;
    const float* rec = partial + (long long)blockIdx.x * nparts * 2;
    float mn = INFINITY, mx = -INFINITY;
    bool any_nan = false;
    for (int i = 0; i < nparts; ++i) {
        any_nan |= rec[2 * i] != rec[2 * i];
        mn = fminf(mn, rec[2 * i]); mx = fmaxf(mx, rec[2 * i + 1]);
    }
    table[2 * blockIdx.x] = any_nan ? NAN : mn;
    table[2 * blockIdx.x + 1] = any_nan ? NAN : mx;
}
struct MinMaxScale {
    float b_scale, b_min, flat_mul;
    int rescale, flat_has_mul;
};
__device__ __forceinline__ float minmax_scale_one(float x, float mn, float div, bool flat, const MinMaxScale& p) {
    if (flat) return p.flat_has_mul ? x * p.flat_mul : x;          // rescale_array: `arr * minv if minv is not None else arr`
    float v = (x - mn) / div;
    if (p.rescale) { v = v * p.b_scale; v = v + p.b_min; }
    return v;
}
template <bool VEC>
__global__ void __launch_bounds__(256) minmax_scale_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n,
                                                           const float* __restrict__ table, MinMaxScale p) {
    const long long off = (long long)blockIdx.y * n;
    const float mn = table[2 * blockIdx.y], mx = table[2 * blockIdx.y + 1];
    const bool flat = mn == mx;
    const float div = mx - mn;
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    if (VEC && i + 4 <= n) {
        const float4 v = *reinterpret_cast<const float4*>(src + off + i);
        float4 r;
        r.x = minmax_scale_one(v.x, mn, div, flat, p); r.y = minmax_scale_one(v.y, mn, div, flat, p);
        r.z = minmax_scale_one(v.z, mn, div, flat, p); r.w = minmax_scale_one(v.w, mn, div, flat, p);
        *reinterpret_cast<float4*>(dst + off + i) = r;
        return;
    }
    for (long long j = i; j < n && j < i + 4; ++j) dst[off + j] = minmax_scale_one(src[off + j], mn, div, flat, p);
}

}  // namespace mh
