// TEST INFRASTRUCTURE: a ten-line extension for the loader test (tests/test_extension_loader.py).
#include <hip/hip_runtime.h>
#ifndef AXPB_SCALE
#define AXPB_SCALE 1
#endif
__global__ void axpb_kernel(const float* x, float* y, int n, float b) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = AXPB_SCALE * x[i] + b;
}
extern "C" int axpb_scale(void) { return AXPB_SCALE; }
extern "C" int axpb_f32(const float* x, float* y, int n, float b, void* stream) {
    hipLaunchKernelGGL(axpb_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, n, b);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
