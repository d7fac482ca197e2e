// Separable 3-D Gaussian smoothing (GaussianSmooth / GaussianFilter), all three axes in ONE pass over HBM.
//
// Reference: monai/transforms/intensity/array.py:1590-1622 -> GaussianFilter (monai/networks/layers/simplelayers.py:
// 542-595) -> separable_filtering :207-249: per axis a zero-padded copy (F.pad) and a depthwise F.conv3d with a
// [C,1,k,1,1]-style kernel -- three padded copies + three conv passes, ~12x the volume in HBM traffic.
// Here a workgroup owns a 16 x 64 (y, x) column of one channel volume and streams along z:
//   plane z (with its y/x halo, zeros outside the image; prefetched into registers one plane ahead) -> LDS -> x-pass
//   (4 outputs per task from one register window) -> LDS -> y-pass -> one value per owned position, pushed into a
//   per-thread register ring of the last RK filtered planes; the z-pass is a dot product of the ring with the z-kernel.  Every input element is read once (plus the in-plane halo, served by
//   L2) and every output written once: 8 B per voxel, the algorithmic minimum.
// fp32 throughout, taps accumulated in ascending tap order per axis; the axis order (x, y, then z) differs from the
// reference's (first axis first), which only reorders fp32 roundings (tests: 1e-5 tolerance; the reference's own
// tests use 1e-4).
#pragma once
#include "common.h"

namespace mh {

constexpr int GS_MAX_TAPS = 33;   // per axis (sigma up to 4 at the reference's truncation of 4 sigma)
constexpr int GS_TX = 64, GS_TY = 16, GS_P = (GS_TX * GS_TY) / 256;

struct GaussArgs {
    int NC, D, H, W;
    int zchunk, nchunk;                                        // the z axis is cut into nchunk runs of zchunk output planes
    float kz[GS_MAX_TAPS], ky[GS_MAX_TAPS], kx[GS_MAX_TAPS];   // each zero-padded symmetrically to the kernel's RK taps
};

// RK = taps per axis (compile time: every tap loop is unrolled, weights sit in SGPRs), HR = halo.
// Thread t owns column x = t & 63 and the four rows 4*(t >> 6) .. +3 of the 16 x 64 tile.
template <int RK>
__global__ void __launch_bounds__(256) gauss3d_stream_kernel(const float* __restrict__ src, float* __restrict__ dst, GaussArgs a) {
#pragma clang fp contract(off)
    constexpr int HR = (RK - 1) / 2;
    constexpr int INH = GS_TY + 2 * HR;
    constexpr int XSPAN = (4 + 2 * HR + 3) / 4 * 4;          // floats one x-pass task reads (whole float4s)
    constexpr int INW = GS_TX - 4 + XSPAN;                   // row stride of the staged plane (multiple of 4)
    constexpr int NLOAD = (INH * INW + 255) / 256;
    constexpr int NTASK = INH * (GS_TX / 4);                 // x-pass tasks: (row, quad of 4 outputs)
    __shared__ __attribute__((aligned(16))) float in_s[INH * INW];
    __shared__ __attribute__((aligned(16))) float mid_s[INH * GS_TX];
    const int tid = threadIdx.x;
    // 1-D launch, XCD-aware: each XCD's L2 gets a contiguous run of (tile, chunk, volume) work items, so the in-plane
    // halo shared by neighbouring tiles is an L2 hit instead of a second HBM read.
    const int tiles_x = (a.W + GS_TX - 1) / GS_TX, tiles = tiles_x * ((a.H + GS_TY - 1) / GS_TY);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = (int)(lid % (unsigned)tiles);
    lid /= (unsigned)tiles;
    const int chunk = (int)(lid % (unsigned)a.nchunk), nc = (int)(lid / (unsigned)a.nchunk);
    const int tx0 = (tile % tiles_x) * GS_TX, ty0 = (tile / tiles_x) * GS_TY;
    const int zs = chunk * a.zchunk, ze = min(zs + a.zchunk, a.D);      // output planes of this workgroup
    const int zfirst = max(zs - HR, 0), zend = min(ze + HR, a.D + HR);  // filtered planes it needs (zero outside the volume)
    const long long plane = (long long)a.H * a.W;
    const float* vol = src + (long long)nc * a.D * plane;
    float* ovol = dst + (long long)nc * a.D * plane;
    const int x = tid & 63, y4 = (tid >> 6) * 4;

    // staging positions of this thread inside the halo plane (fixed for the whole march along z)
    int goff[NLOAD];
    bool gok[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
        const int i = tid + 256 * j;
        const int ly = i / INW, lx = i - ly * INW;
        const int gy = ty0 + ly - HR, gx = tx0 + lx - HR;
        gok[j] = i < INH * INW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[j] = gok[j] ? gy * a.W + gx : 0;
    }
    float pre[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) pre[j] = vol[(long long)zfirst * plane + goff[j]];

    float ring[GS_P][RK];
#pragma unroll
    for (int p = 0; p < GS_P; ++p)
#pragma unroll
        for (int k = 0; k < RK; ++k) ring[p][k] = 0.0f;

    for (int z = zfirst; z < zend; ++z) {
        const bool live = z < a.D;          // beyond the volume the filtered plane is exactly zero (zero padding)
        if (live) {
#pragma unroll
            for (int j = 0; j < NLOAD; ++j) {
                const int i = tid + 256 * j;
                if (i < INH * INW) in_s[i] = gok[j] ? pre[j] : 0.0f;
            }
        }
        __syncthreads();
        if (z + 1 < zend && z + 1 < a.D) {                  // next plane's loads fly while this plane is filtered
            const float* pl = vol + (long long)(z + 1) * plane;
#pragma unroll
            for (int j = 0; j < NLOAD; ++j) pre[j] = pl[goff[j]];
        }
        if (live) {                         // x-pass: 4 neighbouring outputs per task from one sliding register window
            for (int task = tid; task < NTASK; task += 256) {
                const int r = task / (GS_TX / 4), q = task - r * (GS_TX / 4);
                float win[XSPAN];
                const f32x4* rp = reinterpret_cast<const f32x4*>(in_s + r * INW + 4 * q);
#pragma unroll
                for (int v = 0; v < XSPAN / 4; ++v) {
                    const f32x4 t4 = rp[v];
                    win[4 * v] = t4[0]; win[4 * v + 1] = t4[1]; win[4 * v + 2] = t4[2]; win[4 * v + 3] = t4[3];
                }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < RK; ++k) acc = acc + a.kx[k] * win[e + k];
                    o[e] = acc;
                }
                *reinterpret_cast<f32x4*>(mid_s + r * GS_TX + 4 * q) = o;
            }
        }
        __syncthreads();
        float col[GS_P + 2 * HR];
        if (live) {
#pragma unroll
            for (int j = 0; j < GS_P + 2 * HR; ++j) col[j] = mid_s[(y4 + j) * GS_TX + x];
        }
        const int zo = z - HR, gx = tx0 + x;
#pragma unroll
        for (int p = 0; p < GS_P; ++p) {
            float v = 0.0f;
            if (live) {
#pragma unroll
                for (int k = 0; k < RK; ++k) v = v + a.ky[k] * col[p + k];
            }
#pragma unroll
            for (int k = 0; k < RK - 1; ++k) ring[p][k] = ring[p][k + 1];
            ring[p][RK - 1] = v;
            const int gy = ty0 + y4 + p;
            if (zo >= zs && gy < a.H && gx < a.W) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < RK; ++k) acc = acc + a.kz[k] * ring[p][k];
                ovol[(long long)zo * plane + (long long)gy * a.W + gx] = acc;
            }
        }
    }
}

}  // namespace mh
