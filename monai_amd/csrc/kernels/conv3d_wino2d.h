// Shared definitions of the in-plane Winograd convolution (configuration MH_CFG_WINO2D; the kernel is conv3d_wino2p.h):
// Conv3d 3x3x3 / stride 1 / zero padding 1 as Winograd F(2x2, 3x3) in the (y, x) plane + three direct taps along z,
// streamed along z on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32).
//
// Same reference op and "normalise on load" contract as conv3d_mfma.h (nn.Conv3d of `Convolution`,
// monai/networks/blocks/convolutions.py:98-171, fed by the previous block's deferred InstanceNorm + LeakyReLU).
// Output plane z = sum over kz of the 2-D convolution of input plane z + kz - 1 with the 3x3 slice g[kz]; each 2-D
// convolution in minimal-filtering form:  Y = A^T [ sum_kz sum_cin (G g_kz G^T) .* (B^T d B) ] A,  16 products per 2x2
// outputs instead of 36 -> 12 multiply-adds per (output voxel, cin, cout) instead of 27 (2.25x fewer matrix-core
// cycles than conv3d_mfma.h).
//
// Mapping.  A tile = 2x2 outputs of one plane (4x4 input patch).  GEMM per transform position xi (16 of them) and z-tap:
//   M_xi[tile][cout] += V_xi(plane p)[tile][cin] * U_xi^kz[cin][cout],  M = 16 tiles (a 4x4 arrangement: 8x8 outputs),
//   N = 16 couts, K = 4 input channels; input plane p contributes to output planes p+1, p, p-1 (rotating accumulator sets).
// This file holds the geometry constants, the LDS layout and the weight transform / packing; round 1's one-wave-per-SIMD
// kernel and round 2's producer / consumer experiment are gone from the tree (git history: conv3d_wino2d.h, conv3d_wino2s.h).
#pragma once
#include "common.h"

namespace mh {

constexpr int W2_B = 16;                                   // region edge (y and x) of a workgroup
constexpr int W2_R = 18, W2_PX = 20;                       // input region edge, LDS row pitch (= 4 mod 16)
constexpr int W2_CS = W2_R * W2_PX;                        // LDS channel stride (360)
constexpr int W2_KC = 4;                                   // input channels per step (MFMA K)
constexpr int W2_XBUF = W2_KC * W2_CS;                     // staged planes of one step (floats)
constexpr int W2_UPITCH = 52;                              // B operands of one lane: 48 floats [kz][xi] + 4 pad (conflict-free 16-byte reads)
constexpr int W2_UBUF = 4096;                              // weight slab of one step: [lane = (cin & 3) * 16 + (cout & 15)][52], zero padded
constexpr int W2_SLOTS = (W2_R * W2_R + 63) / 64;          // 6 region elements per lane (wave w stages channel w)
constexpr int W2_CN = 16;                                  // couts per workgroup
constexpr int W2_ZSLAB = 64;                                // zero weights: z-taps that leave the chunk multiply by these

// Weight transform U^kz = G g[kz] G^T (y, x), written in the order operand B is read:
// up[cout group][cin step][lane = (cin & 3) * 16 + (cout & 15)][kz * 16 + xi] with lane pitch 52 inside a 4096-float
// slab (the launcher zeroes the padding).  One thread per (cout, cin).
__global__ void __launch_bounds__(256)
conv3d_k3_wino2d_pack_kernel(const float* __restrict__ w, int Cin, int Cout, float* __restrict__ up) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Cin * Cout) return;
    const int ci = idx % Cin, co = idx / Cin;
    const int KS = Cin / W2_KC;
    float* dst = up + ((long long)(co / W2_CN) * KS + ci / W2_KC) * W2_UBUF + ((ci % W2_KC) * 16 + (co % W2_CN)) * W2_UPITCH;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
        float g[3][3], a[4][3], u[4][4];
#pragma unroll
        for (int t = 0; t < 9; ++t) g[t / 3][t % 3] = w[((long long)co * Cin + ci) * 27 + kz * 9 + t];
#pragma unroll
        for (int x = 0; x < 3; ++x) {
            a[0][x] = g[0][x];
            a[1][x] = 0.5f * ((g[0][x] + g[1][x]) + g[2][x]);
            a[2][x] = 0.5f * ((g[0][x] - g[1][x]) + g[2][x]);
            a[3][x] = g[2][x];
        }
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            u[y][0] = a[y][0];
            u[y][1] = 0.5f * ((a[y][0] + a[y][1]) + a[y][2]);
            u[y][2] = 0.5f * ((a[y][0] - a[y][1]) + a[y][2]);
            u[y][3] = a[y][2];
        }
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) dst[kz * 16 + xi] = u[xi / 4][xi % 4];
    }
}

}  // namespace mh
