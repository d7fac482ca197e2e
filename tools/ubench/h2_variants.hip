// Development aid: times conv3d_k3_h2_kernel (Cin -> 32 channels, 96^3, 64 windows) standalone, so that variants of the kernel header
// (-DH2V_RES=true: resident weight slabs; any experimental -D switch added to conv3d_h2.h) compile in seconds on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc [-D...] tools/ubench/h2_variants.hip -o /tmp/h2v
//   /tmp/h2v <Cin> <label> [alias] [edge] [windows] [Cout]      alias 1: every window reads window 0's input, 2: every window writes window 0's output
// -DH2V_WIDE=true: the 8 x 32 region shape instead of 16 x 16 (e.g. edge 24: three regions per plane instead of four)
// Lesson kept from the round-2 ablations: builds that replace loads by register moves feed the matrix pipe constant data, the chip
// then clocks 25-40 % higher, and the "saving" is mostly that -- only timings with real data count.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "monai_amd.h"
#include "kernels/conv3d_h2.h"
using namespace mh;
#ifndef H2V_RES
#define H2V_RES false
#endif
#ifndef H2V_WIDE
#define H2V_WIDE false
#endif

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 32, E = argc > 4 ? atoi(argv[4]) : 96, N = argc > 5 ? atoi(argv[5]) : 64, K = argc > 6 ? atoi(argv[6]) : 32;
    using G = H2Geo<H2V_WIDE>;
    const size_t vox = (size_t)E * E * E;
    float *x, *y, *nrm, *bias, *stats, *packed;
    hipMalloc(&x, sizeof(float) * N * C * vox);
    hipMalloc(&y, sizeof(float) * N * K * vox);
    hipMalloc(&nrm, sizeof(float) * N * C * 4);
    hipMalloc(&bias, sizeof(float) * K);
    const int bxn = (E + G::BX - 1) / G::BX, byn = (E + G::BY - 1) / G::BY;
    int nchunk = (16 + bxn * byn - 1) / (bxn * byn);          // the launcher's z-chunk rule (capi.hip h2_zchunk)
    nchunk = nchunk > E / 12 ? E / 12 : nchunk;
    const int zc = (E + (nchunk < 1 ? 1 : nchunk) - 1) / (nchunk < 1 ? 1 : nchunk);
    const unsigned nblk = bxn * byn * ((E + zc - 1) / zc);
    hipMalloc(&stats, sizeof(float) * N * K * nblk * 3);
    const size_t pf = (size_t)(C / H2_KC) * (K / H2_CN) * H2_WB * 4 + H2_TAIL;
    hipMalloc(&packed, sizeof(float) * pf);
    hipMemset(packed, 0, sizeof(float) * pf);
    std::vector<float> h((size_t)C * vox);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    for (int n = 0; n < N; ++n) hipMemcpy(x + (size_t)n * C * vox, h.data(), sizeof(float) * C * vox, hipMemcpyHostToDevice);
    std::vector<float> hn((size_t)N * C * 4);
    for (size_t i = 0; i < hn.size(); i += 4) { hn[i] = 1.1f; hn[i + 1] = 0.1f; hn[i + 2] = 0.1f; hn[i + 3] = 8.0f; }      // [3] = bound of the activated input (the kernel scales by it)
    hipMemcpy(nrm, hn.data(), sizeof(float) * hn.size(), hipMemcpyHostToDevice);
    std::vector<float> hw((size_t)K * C * 27), hb(K, 0.0f);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.0f - 0.5f) * 0.1f; }
    float* w;
    hipMalloc(&w, sizeof(float) * hw.size());
    hipMemcpy(w, hw.data(), sizeof(float) * hw.size(), hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), sizeof(float) * K, hipMemcpyHostToDevice);
    float* tail = packed + (pf - H2_TAIL);
    hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, 0, w, (long long)hw.size(), tail);
    hipLaunchKernelGGL(conv3d_k3_h2_pack_kernel, dim3((C * K + 255) / 256), dim3(256), 0, 0, w, C, K, reinterpret_cast<_Float16*>(packed), tail);
    const int alias = argc > 3 ? atoi(argv[3]) : 0;          // 1: every window reads window 0's input, 2: every window writes window 0's output
    Tensor in{x, (alias & 1) ? 0LL : (long long)C * (long long)vox, nrm, (long long)C * 4, N, C, E, E, E};
    Tensor out{y, (alias & 2) ? 0LL : (long long)K * (long long)vox, nullptr, 0, N, K, E, E, E};
    const dim3 grid(nblk * N * (K / H2_CN));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((conv3d_k3_h2_kernel<true, true, H2V_RES, H2V_WIDE>), grid, dim3(512), 0, 0, in, reinterpret_cast<const uint4*>(packed), tail, bias, out, stats, bxn, byn, zc, nblk, (float*)nullptr, (float*)nullptr, 0LL);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    std::vector<float> ho(64);
    hipMemcpy(ho.data(), y + 5 * vox + (E / 2) * E * E + (E / 2) * E + 16, 64 * 4, hipMemcpyDeviceToHost);
    printf("%-28s Cin %d -> %d, %d^3 x %d: %.3f ms  (%.0f TFLOP/s direct-equivalent)  y[..] = %g %g\n", argc > 2 ? argv[2] : "full", C, K, E, N, best,
           2.0 * 27 * C * K * vox * N / best / 1e9, ho[0], ho[1]);
    return hipGetLastError() != hipSuccess;
}
