// Post-processing of the blended logits: Activations (sigmoid / softmax over the channel axis) and AsDiscrete (argmax over
// the channel axis, one-hot, threshold, rounding) -- monai/transforms/post/array.py:61-237.  Channel-first tensors
// [C][n] (n = product of the spatial extents), fp32; one thread per voxel, lanes along the last axis, the channel loop
// strides by n (coalesced).  All HBM-bound: argmax reads C x 4 B and writes 4 B per voxel.
#pragma once
#include "common.h"

namespace mh {

enum { PW_SIGMOID = 0, PW_THRESHOLD = 1, PW_ROUND = 2 };
enum { CR_ARGMAX = 0, CR_SOFTMAX = 1 };

__global__ void __launch_bounds__(256) pointwise_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n, int op, float param) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = src[i];
    float r;
    if (op == PW_SIGMOID) r = 1.0f / (1.0f + expf(-v));
    else if (op == PW_THRESHOLD) r = v >= param ? 1.0f : 0.0f;
    else r = rintf(v);                       // torch.round: half to even
    dst[i] = r;
}

// argmax: index of the FIRST maximal value, NaN counts as maximal (torch.argmax); written as float (AsDiscrete's output dtype)
__global__ void __launch_bounds__(256) channel_argmax_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float best = src[i];
    int idx = 0;
    for (int c = 1; c < C; ++c) {
        const float v = src[(long long)c * n + i];
        if (v > best || (v != v && best == best)) { best = v; idx = c; }
    }
    dst[i] = (float)idx;
}

// softmax over the channel axis: exp(x - max) / sum, two passes over the C values of a voxel (C is small: registers / L1)
__global__ void __launch_bounds__(256) channel_softmax_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float m = src[i];
    for (int c = 1; c < C; ++c) m = fmaxf(m, src[(long long)c * n + i]);
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s += expf(src[(long long)c * n + i] - m);
    for (int c = 0; c < C; ++c) dst[(long long)c * n + i] = expf(src[(long long)c * n + i] - m) / s;
}

// one_hot (monai/networks/utils.py:170-221) of float labels [1][n] -> [K][n]
__global__ void __launch_bounds__(256) onehot_kernel(const float* __restrict__ labels, float* __restrict__ dst, int K, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long l = (long long)labels[i];          // labels.long(): truncation
    for (int k = 0; k < K; ++k) dst[(long long)k * n + i] = k == l ? 1.0f : 0.0f;
}

}  // namespace mh
