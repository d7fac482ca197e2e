// Development aid (round 6): separable_resample_stream_kernel standalone at BASELINE.json config 4 (512^3 -> 410 x 410 x 819, affine diag(.8, .8, 1.6) -> pixdim 1,
// trilinear, border), so that variants of kernels/resample.h compile in seconds on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Imonai_amd/csrc [-DRSV_T=float -DRSV_NL=4 -DRSV_NT=512 -DRSV_VEC=true -DRSV_RING=3 -DMH_RS_MINW=8] tools/ubench/resample_variants.hip -o /tmp/rsv
//   /tmp/rsv <label> [chunks]          chunks: z-chunks per (tile, channel) column; default = the launcher's rule (capi.hip stream_chunks) on the occupancy the API reports
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "monai_amd.h"
#include "kernels/resample.h"
using namespace mh;
#ifndef RSV_T
#define RSV_T double
#endif
#ifndef RSV_NL
#define RSV_NL 8
#endif
#ifndef RSV_NT
#define RSV_NT 256
#endif
#ifndef RSV_VEC
#define RSV_VEC true
#endif
#ifndef RSV_RING
#define RSV_RING 0
#endif

int main(int argc, char** argv) {
    const int Di = 512, Hi = 512, Wi = 512, Do = 410, Ho = 410, Wo = 819;
    ResampleArgs a;
    a.mode = RS_LINEAR; a.pad = RS_BORDER; a.align_corners = 0; a.C = 1;
    a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    for (int i = 0; i < 12; ++i) a.m[i] = 0.0;
    for (int i = 0; i < 3; ++i) { a.ga[i] = 1.0; a.gb[i] = 0.0; }
    const double sc[3] = {1.25, 1.25, 0.625};
    for (int r = 0; r < 3; ++r) { a.m[r * 4 + r] = sc[r]; a.m[r * 4 + 3] = 0.5 * sc[r] - 0.5; }
    float *src, *dst;
    AxisTap<RSV_T>* tab;
    const size_t ni = (size_t)Di * Hi * Wi, no = (size_t)Do * Ho * Wo;
    hipMalloc(&src, ni * 4); hipMalloc(&dst, no * 4); hipMalloc(&tab, (Do + Ho + Wo) * sizeof(AxisTap<RSV_T>));
    std::vector<float> h(ni);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f; }
    hipMemcpy(src, h.data(), ni * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((resample_axis_table_kernel<RSV_T>), dim3((Do + Ho + Wo + 255) / 256), dim3(256), 0, 0, tab, a);
    auto kern = separable_resample_stream_kernel<RSV_T, RSV_NL, RSV_NT, RSV_VEC, RSV_RING>;
    int per_cu = 1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, RSV_NT, 0);
    const int slots = 256 * (per_cu < 1 ? 1 : per_cu);
    const long long tiles = (long long)((Wo + RZ_TOX - 1) / RZ_TOX) * ((Ho + RZ_TOY - 1) / RZ_TOY);
    int nchunk = argc > 2 ? atoi(argv[2]) : 0;
    if (nchunk < 1) {          // capi.hip: stream_chunks(tiles, Do, slots, prime 1, min_chunk 8)
        double best = -1.0;
        for (int n = 1; n <= (Do + 7) / 8; ++n) {
            const int zc = (Do + n - 1) / n;
            if ((Do + zc - 1) / zc != n) continue;
            const long long nwg = tiles * n, rounds = (nwg + slots - 1) / slots;
            const double score = (double)nwg / (double)(rounds * slots) * ((double)zc / (double)(zc + 1));
            if (score > best) { best = score; nchunk = n; }
        }
    }
    const int zchunk = (Do + nchunk - 1) / nchunk;
    nchunk = (Do + zchunk - 1) / zchunk;
    const AxisTap<RSV_T>* ct = tab;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0.f;
    for (int it = 0; it < 8; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * nchunk)), dim3(RSV_NT), 0, 0, (const float*)src, dst, ct, a, zchunk, nchunk);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (it >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    std::vector<float> ho(4);
    hipMemcpy(ho.data(), dst + ((size_t)200 * Ho + 200) * Wo + 400, 16, hipMemcpyDeviceToHost);
    double cs = 0.0;
    {
        std::vector<float> all(no);
        hipMemcpy(all.data(), dst, no * 4, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < no; i += 97) cs += all[i];
    }
    const double bytes = 4.0 * (ni + no);
    printf("%-44s occupancy %d WG/CU, %d chunks: best %.4f ms avg %.4f ms = %.3f of 8 TB/s  (checksum %.6f, %s)\n", argc > 1 ? argv[1] : "default", per_cu, nchunk, best, sum / 6.0,
           bytes / (sum / 6.0 * 1e-3) / 8e12, cs, hipGetErrorString(hipGetLastError()));
    return 0;
}
