// Development aid: the z-Winograd split-precision convolution (kernels/conv3d_h2z.h) standalone against the direct split-precision kernel (conv3d_h2.h) on the
// SAME tensors -- time of both (alternating launches, best of 3 after a warm-up) and the largest difference of their outputs:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude -Imonai_amd/csrc [-D...] tools/ubench/h2z_variants.hip -o /tmp/h2zv
//   /tmp/h2zv <Cin> <label> [edge] [windows] [Cout]
// Real (pseudo-random) data only: constant operands raise the matrix pipe's clock (conv3d_h2.h header).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "monai_amd.h"
#include "kernels/conv3d_h2z.h"
using namespace mh;

int main(int argc, char** argv) {
    const int C = argc > 1 ? atoi(argv[1]) : 32, E = argc > 3 ? atoi(argv[3]) : 96, N = argc > 4 ? atoi(argv[4]) : 64, K = argc > 5 ? atoi(argv[5]) : 32;
    const size_t vox = (size_t)E * E * E;
    float *x, *y, *y2, *nrm, *bias, *stats, *stats2, *packed, *packed2;
    hipMalloc(&x, sizeof(float) * N * C * vox);
    hipMalloc(&y, sizeof(float) * N * K * vox);
    hipMalloc(&y2, sizeof(float) * N * K * vox);
    hipMalloc(&nrm, sizeof(float) * N * C * 4);
    hipMalloc(&bias, sizeof(float) * K);
    // the direct kernel: the launcher's rules (capi.hip h2_zchunk, region shape by h2_wide)
    const bool wide = h2_wide(E, E);
    const int bxn = wide ? (E + 31) / 32 : (E + 15) / 16, byn = wide ? (E + 7) / 8 : (E + 15) / 16;
    int nchunk = (16 + bxn * byn - 1) / (bxn * byn);
    nchunk = nchunk > E / 12 ? E / 12 : nchunk;
    nchunk = nchunk < 1 ? 1 : nchunk;
    const int zc = (E + nchunk - 1) / nchunk;
    const unsigned nblk = bxn * byn * ((E + zc - 1) / zc);
    // the z-Winograd kernel: 8 x 32 regions, even z-chunks (capi.hip h2z_zchunk)
    const int zbxn = (E + HZ_BX - 1) / HZ_BX, zbyn = (E + HZ_BY - 1) / HZ_BY;
    int znchunk = (16 + zbxn * zbyn - 1) / (zbxn * zbyn);
    znchunk = znchunk > E / 12 ? E / 12 : znchunk;
    znchunk = znchunk < 1 ? 1 : znchunk;
#ifdef H2ZV_CHUNKS
    znchunk = H2ZV_CHUNKS;
#endif
    const int zzc = (((E + znchunk - 1) / znchunk) + 1) & ~1;
    const unsigned znblk = zbxn * zbyn * ((E + zzc - 1) / zzc);
    hipMalloc(&stats, sizeof(float) * N * K * nblk * 3);
    hipMalloc(&stats2, sizeof(float) * N * K * znblk * 3);
    const size_t pf = (size_t)(C / H2_KC) * (K / H2_CN) * H2_WB * 4 + H2_TAIL;
    const size_t pf2 = (size_t)(C / H2_KC) * (K / H2_CN) * (4 * HZ_WP) * 4 + H2_TAIL;
    hipMalloc(&packed, sizeof(float) * pf);
    hipMalloc(&packed2, sizeof(float) * pf2);
    hipMemset(packed, 0, sizeof(float) * pf);
    hipMemset(packed2, 0, sizeof(float) * pf2);
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
    float* tail2 = packed2 + (pf2 - H2_TAIL);
    hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, 0, w, (long long)hw.size(), tail);
    hipLaunchKernelGGL(conv3d_k3_h2_pack_kernel, dim3((C * K + 255) / 256), dim3(256), 0, 0, w, C, K, reinterpret_cast<_Float16*>(packed), tail);
    hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, 0, w, (long long)hw.size(), tail2);
    hipLaunchKernelGGL(conv3d_k3_h2z_pack_kernel, dim3((C * K + 255) / 256), dim3(256), 0, 0, w, C, K, reinterpret_cast<_Float16*>(packed2), tail2);
    Tensor in{x, (long long)C * (long long)vox, nrm, (long long)C * 4, N, C, E, E, E};
    Tensor out{y, (long long)K * (long long)vox, nullptr, 0, N, K, E, E, E};
    Tensor out2{y2, (long long)K * (long long)vox, nullptr, 0, N, K, E, E, E};
    const dim3 grid(nblk * N * (K / H2_CN)), grid2(znblk * N * (K / H2_CN));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, best2 = 1e9f;
    for (int it = 0; it < 4; ++it) {
        float ms;
        hipEventRecord(e0);
        if (C <= 2 * H2_KC) {
            if (wide) hipLaunchKernelGGL((conv3d_k3_h2_kernel<true, true, true, true>), grid, dim3(512), 0, 0, in, reinterpret_cast<const uint4*>(packed), tail, bias, out, stats, bxn, byn, zc, nblk);
            else hipLaunchKernelGGL((conv3d_k3_h2_kernel<true, true, true, false>), grid, dim3(512), 0, 0, in, reinterpret_cast<const uint4*>(packed), tail, bias, out, stats, bxn, byn, zc, nblk);
        } else {
            if (wide) hipLaunchKernelGGL((conv3d_k3_h2_kernel<true, true, false, true>), grid, dim3(512), 0, 0, in, reinterpret_cast<const uint4*>(packed), tail, bias, out, stats, bxn, byn, zc, nblk);
            else hipLaunchKernelGGL((conv3d_k3_h2_kernel<true, true, false, false>), grid, dim3(512), 0, 0, in, reinterpret_cast<const uint4*>(packed), tail, bias, out, stats, bxn, byn, zc, nblk);
        }
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL((conv3d_k3_h2z_kernel<true, true>), grid2, dim3(512), 0, 0, in, reinterpret_cast<const uint4*>(packed2), tail2, bias, out2, stats2, zbxn, zbyn, zzc, znblk);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best2) best2 = ms;
    }
    // difference of the two kernels on window 0 and the last window (every cout, every voxel)
    double dmax = 0.0, vmax = 0.0;
    std::vector<float> a((size_t)K * vox), b2((size_t)K * vox);
    for (int n : {0, N - 1}) {
        hipMemcpy(a.data(), y + (size_t)n * K * vox, sizeof(float) * K * vox, hipMemcpyDeviceToHost);
        hipMemcpy(b2.data(), y2 + (size_t)n * K * vox, sizeof(float) * K * vox, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < a.size(); ++i) {
            const double d = std::fabs((double)a[i] - (double)b2[i]);
            if (!(d <= dmax)) dmax = d;
            if (std::fabs((double)a[i]) > vmax) vmax = std::fabs((double)a[i]);
        }
    }
    const double tf = 2.0 * 27 * C * K * vox * N / 1e9;
    printf("%-24s Cin %d -> %d, %d^3 x %d: direct %.3f ms (%.0f TF direct-eq) | z-Winograd %.3f ms (%.0f TF direct-eq) = %.3fx | max |dy| %.3g of max |y| %.3g\n",
           argc > 2 ? argv[2] : "default", C, K, E, N, best, tf / best, best2, tf / best2, best / best2, dmax, vmax);
    return hipGetLastError() != hipSuccess;
}
