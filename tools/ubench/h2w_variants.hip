// Development aid: times conv3d_k3_h2w_kernel (32 -> 32 channels, 96^3, 64 windows) standalone, so that variants of the kernel header (-DHWX_OFF=<bits> ablations,
// -DHWX_PROF segment cycle counters, any experimental -D switch of conv3d_wino_h2.h) compile in seconds on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imonai_amd/csrc [-D...] tools/ubench/h2w_variants.hip -o /tmp/h2wv
//   /tmp/h2wv <label> [edge] [windows]
// Random inputs (the matrix pipe's clock depends on its operand data: constant operands would flatter every variant).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "monai_amd.h"
#include "kernels/conv3d_wino_h2.h"
using namespace mh;

int main(int argc, char** argv) {
    const int C = 32, K = 32, E = argc > 2 ? atoi(argv[2]) : 96, N = argc > 3 ? atoi(argv[3]) : 64;
    const size_t vox = (size_t)E * E * E;
    float *x, *y, *nrm, *bias, *stats, *packed, *dbg;
    hipMalloc(&x, sizeof(float) * N * C * vox);
    hipMalloc(&y, sizeof(float) * N * K * vox);
    hipMalloc(&nrm, sizeof(float) * N * C * 4);
    hipMalloc(&bias, sizeof(float) * K);
    hipMalloc(&dbg, 8192);
    hipMemset(dbg, 0, 8192);
    const int bxn = E / HWG_BX, byn = E / HWG_BY;
    int nchunk = (32 + bxn * byn - 1) / (bxn * byn);          // the launcher's z-chunk rule (capi.hip hw_zchunk)
    nchunk = nchunk > E / 12 ? E / 12 : nchunk;
    nchunk = nchunk < 1 ? 1 : nchunk;
    int zc = (E + nchunk - 1) / nchunk;
    zc += zc & 1;
    const unsigned nblk = bxn * byn * ((E + zc - 1) / zc);
    hipMalloc(&stats, sizeof(float) * N * K * nblk * 3);
    #ifndef HWX_WAVES
#define HWX_WAVES 8
#endif
    const size_t pf = (size_t)(K / HWG_CN) * HWX_WAVES * HWG_OPS * 64 * 4 + H2_TAIL;
    hipMalloc(&packed, sizeof(float) * pf);
    std::vector<float> h((size_t)C * vox);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    for (int n = 0; n < N; ++n) hipMemcpy(x + (size_t)n * C * vox, h.data(), sizeof(float) * C * vox, hipMemcpyHostToDevice);
    std::vector<float> hn((size_t)N * C * 4);
    for (size_t i = 0; i < hn.size(); i += 4) { hn[i] = 1.1f; hn[i + 1] = 0.1f; hn[i + 2] = 0.1f; hn[i + 3] = 8.0f; }
    hipMemcpy(nrm, hn.data(), sizeof(float) * hn.size(), hipMemcpyHostToDevice);
    std::vector<float> hw((size_t)K * C * 27), hb(K, 0.0f);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.1f; }
    float* w;
    hipMalloc(&w, sizeof(float) * hw.size());
    hipMemcpy(w, hw.data(), sizeof(float) * hw.size(), hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), sizeof(float) * K, hipMemcpyHostToDevice);
    float* tail = packed + (pf - H2_TAIL);
    hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, 0, w, (long long)hw.size(), tail);
    hipLaunchKernelGGL(conv3d_k3_h2w_scale_fix_kernel, dim3(1), dim3(64), 0, 0, tail);
    hipLaunchKernelGGL(conv3d_k3_h2w_pack_kernel, dim3((C * K + 255) / 256), dim3(256), 0, 0, w, C, K, reinterpret_cast<_Float16*>(packed), tail);
    Tensor in{x, (long long)C * (long long)vox, nrm, (long long)C * 4, N, C, E, E, E};
    Tensor out{y, (long long)K * (long long)vox, nullptr, 0, N, K, E, E, E};
    const dim3 grid(nblk * N * (K / HWG_CN));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((conv3d_k3_h2w_kernel<true, false, false>), grid, dim3(64 * HWX_WAVES), 0, 0, in, reinterpret_cast<const uint4*>(packed), tail, bias, out, stats, bxn, byn, zc, nblk, dbg, (float*)nullptr, 0LL);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    std::vector<float> ho(64);
    hipMemcpy(ho.data(), y + 5 * vox + (E / 2) * E * E + (E / 2) * E + 16, 64 * 4, hipMemcpyDeviceToHost);
    printf("%-34s 32 -> 32, %d^3 x %d: %.3f ms  (%.0f TFLOP/s direct-equivalent)  y[..] = %g %g\n", argc > 1 ? argv[1] : "full", E, N, best,
           2.0 * 27 * C * K * vox * N / best / 1e9, ho[0], ho[1]);
#ifdef HWX_PROF
    long long t[128];
    hipMemcpy(t, dbg, sizeof(t), hipMemcpyDeviceToHost);
    const int iters = zc + 3;
    for (int wv = 0; wv < HWX_WAVES; ++wv)
        printf("   wave %d (row %d, columns %d): cycles per iteration: vector phase %5.0f | barrier %5.0f | matrix phase %5.0f | barrier %5.0f\n", wv, wv & 3, wv >> 2,
               (double)t[wv * 8] / iters, (double)t[wv * 8 + 1] / iters, (double)t[wv * 8 + 2] / iters, (double)t[wv * 8 + 3] / iters);
#endif
    return hipGetLastError() != hipSuccess;
}
