// Sliding-window inferer kernels: window gather and the fused importance-weighted blend.
// Reference behaviour: monai/inferers/utils.py:215-298 (window loop, `*= w`, `+=`, count map, `/=`).
#pragma once
#include "common.h"

namespace mh {

constexpr int MAX_AXIS_WINDOWS = 160;

// Per-axis window starts of the dense window grid (monai/data/utils.py:166-206), passed by value in the
// kernel arguments: no device-side tables, nothing to upload.
struct WindowGrid {
    int nz, ny, nx;
    int sz[MAX_AXIS_WINDOWS], sy[MAX_AXIS_WINDOWS], sx[MAX_AXIS_WINDOWS];
};

// Covering range of coordinate p on one axis: windows i with s[i] <= p < s[i] + r.  Starts ascend and
// the window size is constant, so the covering set is a contiguous index range [lo, hi].
__device__ __forceinline__ void cover(const int* s, int n, int r, int p, int& lo, int& hi) {
    lo = 0;
    while (lo < n - 1 && s[lo] + r <= p) ++lo;
    hi = lo;
    while (hi < n - 1 && s[hi + 1] <= p) ++hi;
}

// ---------------------------------------------------------------------------------------------------
// Gather `nwin` consecutive windows (first = w0, row-major over the start lists) of one image
// [C][D][H][W] into a dense batch [nwin][C][rd][rh][rw].  One thread per VEC output elements.
template <int VEC>
__global__ void __launch_bounds__(256)
window_extract_kernel(const float* __restrict__ vol, int C, int D, int H, int W, WindowGrid g, int w0, int nwin,
                      int rd, int rh, int rw, float* __restrict__ out) {
    const int rwv = rw / VEC;
    const long long total = (long long)nwin * C * rd * rh * rwv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int xv = (int)(idx % rwv);
    long long t = idx / rwv;
    const int ly = (int)(t % rh); t /= rh;
    const int lz = (int)(t % rd); t /= rd;
    const int c = (int)(t % C);
    const int wl = (int)(t / C);
    const int w = w0 + wl;
    const int ix = w % g.nx, iy = (w / g.nx) % g.ny, iz = w / (g.nx * g.ny);
    const int z = g.sz[iz] + lz, y = g.sy[iy] + ly, x = g.sx[ix] + xv * VEC;
    const float* src = vol + (((long long)c * D + z) * H + y) * W + x;
    float* dst = out + idx * VEC;
    if (VEC == 4) {
        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    } else {
        dst[0] = src[0];
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused blend: one thread owns VEC consecutive output voxels along W and KT output channels
// [k0, k0+KT).  It walks the covering windows in ascending window index and accumulates in the exact
// floating-point order of the reference: acc = acc + fp32(logit * w), cnt = cnt + w, out = acc / cnt
// (separately rounded multiply / add / IEEE divide: no FMA contraction).
// HBM traffic: logits are read exactly once (every window voxel covers exactly one output voxel), the
// output is written once; the importance map (roi^3 floats) and the start lists stay cache resident.
template <int KT, int VEC>
__global__ void __launch_bounds__(256)
sw_blend_kernel(const float* __restrict__ logits, const float* __restrict__ imp, float* __restrict__ out, int K,
                int k0, int D, int H, int W, int rd, int rh, int rw, WindowGrid g, int premul) {
    const int wv = W / VEC;
    const long long total = (long long)D * H * wv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % wv) * VEC;
    const long long t = idx / wv;
    const int y = (int)(t % H), z = (int)(t / H);

    int zlo, zhi, ylo, yhi, xlo, xhi;
    cover(g.sz, g.nz, rd, z, zlo, zhi);
    cover(g.sy, g.ny, rh, y, ylo, yhi);
    cover(g.sx, g.nx, rw, x, xlo, xhi);  // with VEC == 4 all starts and rw are multiples of 4: same set for x..x+3

    const long long plane = (long long)rh * rw, roi = plane * rd;
    float acc[KT][VEC];
    float cnt[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        cnt[v] = 0.0f;
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k][v] = 0.0f;
    }

    for (int iz = zlo; iz <= zhi; ++iz) {
        const int lz = z - g.sz[iz];
        for (int iy = ylo; iy <= yhi; ++iy) {
            const int ly = y - g.sy[iy];
            for (int ix = xlo; ix <= xhi; ++ix) {
                const int lx = x - g.sx[ix];
                const long long w = ((long long)iz * g.ny + iy) * g.nx + ix;
                const long long off = (long long)lz * plane + (long long)ly * rw + lx;
                const float* lp = logits + (w * K + k0) * roi + off;
                float wt[VEC];
                float lv[KT][VEC];
                if (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(imp + off);
                    wt[0] = q.x; wt[1] = q.y; wt[2] = q.z; wt[3] = q.w;
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const float4 a = *reinterpret_cast<const float4*>(lp + k * roi);
                        lv[k][0] = a.x; lv[k][1] = a.y; lv[k][2] = a.z; lv[k][3] = a.w;
                    }
                } else {
                    wt[0] = imp[off];
#pragma unroll
                    for (int k = 0; k < KT; ++k) lv[k][0] = lp[k * roi];
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
#pragma unroll
                    // premul: the logits were already multiplied by their (per-batch) weight (process_fn); x * 1.0f is exact
                    for (int k = 0; k < KT; ++k) acc[k][v] = __fadd_rn(acc[k][v], __fmul_rn(lv[k][v], premul ? 1.0f : wt[v]));
                    cnt[v] = __fadd_rn(cnt[v], wt[v]);
                }
            }
        }
    }

    const long long vox = (long long)D * H * W;
    float* op = out + (long long)k0 * vox + ((long long)z * H + y) * W + x;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        if (VEC == 4) {
            float4 r;
            r.x = __fdiv_rn(acc[k][0], cnt[0]); r.y = __fdiv_rn(acc[k][1], cnt[1]);
            r.z = __fdiv_rn(acc[k][2], cnt[2]); r.w = __fdiv_rn(acc[k][3], cnt[3]);
            *reinterpret_cast<float4*>(op + k * vox) = r;
        } else {
            op[k * vox] = __fdiv_rn(acc[k][0], cnt[0]);
        }
    }
}

// AvgMerger (monai/inferers/merger.py:103-205): `values[slice] += patch; counts[slice] += 1` for one patch -- the patches
// of a PatchInferer arrive one by one through a user-visible Merger object, so the accumulation order (= patch order) is
// the reference's -- and the final `values /= counts`.  One thread per patch element, lanes along x; HBM-bound read-modify-
// write of the patch footprint (12 B per element + 2 B of counts).
__global__ void __launch_bounds__(256)
patch_accumulate_kernel(float* __restrict__ values, unsigned char* __restrict__ counts, const float* __restrict__ patch, int NC, int D, int H, int W,
                        int pd, int ph, int pw, int z0, int y0, int x0) {
    const long long pvol = (long long)pd * ph * pw;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= pvol * NC) return;
    const long long c = idx / pvol, r = idx - c * pvol;
    const int x = (int)(r % pw), y = (int)((r / pw) % ph), z = (int)(r / ((long long)pw * ph));
    const long long o = ((c * D + (z0 + z)) * H + (y0 + y)) * W + (x0 + x);
    values[o] += patch[idx];
    counts[o] = (unsigned char)(counts[o] + 1);
}

__global__ void __launch_bounds__(256)
avg_finalize_kernel(float* __restrict__ values, const unsigned char* __restrict__ counts, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) values[i] = values[i] / (float)counts[i];
}

}  // namespace mh
