// Separable 3-D Gaussian smoothing (GaussianSmooth / GaussianFilter), all three axes in ONE pass over HBM.
//
// Reference: monai/transforms/intensity/array.py:1590-1622 -> GaussianFilter (monai/networks/layers/simplelayers.py:
// 542-595) -> separable_filtering :207-249: per axis a zero-padded copy (F.pad) and a depthwise F.conv3d with a
// [C,1,k,1,1]-style kernel -- three padded copies + three conv passes, ~12x the volume in HBM traffic.
// Here a workgroup owns a 16 x 64 (y, x) column of one channel volume and streams along z:
//   plane z (with its y/x halo, zeros outside the image) -> LDS -> x-pass -> LDS -> y-pass -> one value per owned
//   position, pushed into a per-thread register ring of the last Kz filtered planes; the z-pass is a dot product of
//   the ring with the (front-padded) z-kernel.  Every input element is read once (plus the in-plane halo, served by
//   L2) and every output written once: 8 B per voxel, the algorithmic minimum.
// fp32 throughout, taps accumulated in ascending tap order per axis; the axis order (x, y, then z) differs from the
// reference's (first axis first), which only reorders fp32 roundings (tests: 1e-5 tolerance; the reference's own
// tests use 1e-4).
#pragma once
#include "common.h"

namespace mh {

constexpr int GS_MAX_TAPS = 33;   // per axis (sigma up to 4 at the reference's truncation of 4 sigma)
constexpr int GS_TX = 64, GS_TY = 16, GS_P = (GS_TX * GS_TY) / 256;
constexpr int GS_HALO = (GS_MAX_TAPS - 1) / 2;
constexpr int GS_INW = GS_TX + 2 * GS_HALO, GS_INH = GS_TY + 2 * GS_HALO;

struct GaussArgs {
    int NC, D, H, W;
    int kz_n, ky_n, kx_n;          // odd tap counts
    float kz[GS_MAX_TAPS], ky[GS_MAX_TAPS], kx[GS_MAX_TAPS];
};

template <int RK>
__global__ void __launch_bounds__(256) gauss3d_stream_kernel(const float* __restrict__ src, float* __restrict__ dst, GaussArgs a) {
#pragma clang fp contract(off)
    __shared__ float in_s[GS_INH * GS_INW];
    __shared__ float mid_s[GS_INH * GS_TX];
    const int tid = threadIdx.x;
    const int tiles_x = (a.W + GS_TX - 1) / GS_TX;
    const int tx0 = (int)(blockIdx.x % tiles_x) * GS_TX, ty0 = (int)(blockIdx.x / tiles_x) * GS_TY;
    const int nc = blockIdx.y;
    const int hx = (a.kx_n - 1) / 2, hy = (a.ky_n - 1) / 2, hz = (a.kz_n - 1) / 2;
    const int inw = GS_TX + 2 * hx, inh = GS_TY + 2 * hy;
    const long long plane = (long long)a.H * a.W;
    const float* vol = src + (long long)nc * a.D * plane;
    float* ovol = dst + (long long)nc * a.D * plane;

    // z-kernel padded at the front to RK taps: ring[RK-1] is the newest plane, the dot product uses fixed indices
    float kzp[RK];
#pragma unroll
    for (int k = 0; k < RK; ++k) {
        const int j = k - (RK - a.kz_n);
        kzp[k] = j >= 0 ? a.kz[j] : 0.0f;
    }
    float ring[GS_P][RK];
#pragma unroll
    for (int p = 0; p < GS_P; ++p)
#pragma unroll
        for (int k = 0; k < RK; ++k) ring[p][k] = 0.0f;

    for (int z = 0; z < a.D + hz; ++z) {
        if (z < a.D) {
            const float* pl = vol + (long long)z * plane;
            for (int i = tid; i < inh * inw; i += 256) {
                const int ly = i / inw, lx = i - ly * inw;
                const int gy = ty0 + ly - hy, gx = tx0 + lx - hx;
                in_s[ly * GS_INW + lx] = (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? pl[(long long)gy * a.W + gx] : 0.0f;
            }
        }
        __syncthreads();
        if (z < a.D) {   // x-pass over the rows the y-pass needs
            for (int i = tid; i < inh * GS_TX; i += 256) {
                const int ly = i / GS_TX, lx = i - ly * GS_TX;
                const float* row = in_s + ly * GS_INW + lx;
                float acc = 0.0f;
                for (int k = 0; k < a.kx_n; ++k) acc = acc + a.kx[k] * row[k];
                mid_s[ly * GS_TX + lx] = acc;
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < GS_P; ++p) {
            const int pos = tid + 256 * p;
            const int ly = pos / GS_TX, lx = pos - ly * GS_TX;
            float v = 0.0f;
            if (z < a.D) {
                const float* col = mid_s + ly * GS_TX + lx;
                for (int k = 0; k < a.ky_n; ++k) v = v + a.ky[k] * col[k * GS_TX];
            }
#pragma unroll
            for (int k = 0; k < RK - 1; ++k) ring[p][k] = ring[p][k + 1];
            ring[p][RK - 1] = v;
            const int zo = z - hz, gy = ty0 + ly, gx = tx0 + lx;
            if (zo >= 0 && gy < a.H && gx < a.W) {
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < RK; ++k) acc = acc + kzp[k] * ring[p][k];
                ovol[(long long)zo * plane + (long long)gy * a.W + gx] = acc;
            }
        }
    }
}

}  // namespace mh
