// Micro-benchmark (development aid): can TWO waves per SIMD, each running the Winograd kernel's shape of work -- bursts of
// back-to-back fp32 MFMAs with lumps of VALU / LDS work between them -- keep the matrix pipe busier than ONE wave per SIMD?
// (coexec.hip showed that a pure-MFMA wave starves a pure-VALU wave; the question here is the interleaved steady state.)
// Also: issue cost of packed adds, 16-byte LDS reads, global vs buffer loads when they sit in a lump between MFMA bursts.
//   hipcc --offload-arch=gfx950 -O3 coexec2.hip -o /tmp/coexec2 && /tmp/coexec2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

#define MFMA16                                                                                          \
    REP4("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n"              \
         "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n")

// MODE: what sits between two bursts of 16 MFMAs
//  0 nothing            1 32 v_add_f32 (4 chains)     2 16 v_pk_add_f32 (4 chains)      3 8 ds_read_b128 (+ wait)
//  4 6 global_load_dword, VGPR address (+ wait)       5 6 buffer_load_dword, SGPR resource + VGPR offset (+ wait)
//  6 the Winograd step's mix: 32 v_add + 8 ds_read_b128 + 6 ds_write_b32 + 6 global loads
template <int MODE>
__global__ void __launch_bounds__(512, 1) k(long long* out, float* sink, const float* src, int iters) {
    __shared__ float lds[8192];
    const int wave = threadIdx.x >> 6;
    float a = threadIdx.x * 0.5f, b = 1.25f, c = 0.75f, d = 2.0f, e = 3.0f, f = 4.0f;
    f32x2 p0 = {1.0f, 2.0f}, p1 = {3.0f, 4.0f}, p2 = {5.0f, 6.0f}, p3 = {7.0f, 8.0f}, pb = {0.5f, 0.25f};
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    f32x4 r0 = {0, 0, 0, 0}, r1 = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = a;
    const int laddr = (threadIdx.x & 255) * 16;
    const float* gp = src + threadIdx.x;
    // buffer resource over `src` (1 MB), raw dword addressing
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
    const int voff = threadIdx.x * 4;
    float g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        asm volatile(MFMA16 : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
        if (MODE == 1 || MODE == 6) {
            asm volatile(REP8("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
                         : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));
        }
        if (MODE == 2) {
            asm volatile(REP4("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
        }
        if (MODE == 3 || MODE == 6) {
            asm volatile(REP4("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n") "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r0), "=v"(r1) : "v"(laddr));
        }
        if (MODE == 6) {
            asm volatile(REP2("ds_write_b32 %0, %1 offset:16384\n ds_write_b32 %0, %2 offset:20480\n ds_write_b32 %0, %3 offset:24576\n")
                         :: "v"(laddr), "v"(a), "v"(d), "v"(e) : "memory");
        }
        if (MODE == 4 || MODE == 6) {
            asm volatile("global_load_dword %0, %6, off\n global_load_dword %1, %6, off offset:1024\n global_load_dword %2, %6, off offset:2048\n"
                         "global_load_dword %3, %6, off offset:3072\n global_load_dword %4, %6, off offset:4095\n global_load_dword %5, %6, off offset:512\n"
                         "s_waitcnt vmcnt(0)\n"
                         : "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3), "=v"(g4), "=v"(g5) : "v"(gp));
        }
        if (MODE == 5) {
            g0 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0);
            g1 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 4096, 0);
            g2 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 8192, 0);
            g3 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 12288, 0);
            g4 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 16384, 0);
            g5 = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 20480, 0);
            asm volatile("s_waitcnt vmcnt(0)" :: "v"(g0), "v"(g1), "v"(g2), "v"(g3), "v"(g4), "v"(g5));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = a + c + d + e + f + acc0[0] + acc1[1] + acc2[2] + acc3[3] + r0[0] + r1[1] + g0 + g1 + g2 + g3 + g4 + g5 +
                                           p0[0] + p1[1] + p2[0] + p3[1];
}

template <int MODE> static void run(const char* name) {
    long long* out; float *sink, *src;
    hipMalloc(&out, 64); hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    const int iters = 2000;
    double t[2];
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int threads = cfg == 0 ? 256 : 512;      // one / two waves per SIMD
        hipMemset(out, 0, 64);
        k<MODE><<<256, threads>>>(out, sink, src, iters);
        k<MODE><<<256, threads>>>(out, sink, src, iters);
        hipDeviceSynchronize();
        long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        long long m = 0;
        for (int w = 0; w < threads / 64; ++w) m = h[w] > m ? h[w] : m;
        t[cfg] = (double)m / iters;
    }
    // matrix-pipe busy fraction: 16 MFMAs x 32 cycles per wave and iteration
    printf("%-66s 1 wave/SIMD: %7.1f ticks/iter (pipe %.2f)   2 waves/SIMD: %7.1f ticks/iter for 2x the work (pipe %.2f)\n", name, t[0], 512.0 / t[0],
           t[1], 1024.0 / t[1]);
    hipFree(out); hipFree(sink); hipFree(src);
}

int main() {
    run<0>("16 MFMA back to back");
    run<1>("16 MFMA + 32 v_add_f32");
    run<2>("16 MFMA + 16 v_pk_add_f32 (same adds, packed)");
    run<3>("16 MFMA + 8 ds_read_b128, waited");
    run<4>("16 MFMA + 6 global_load_dword (VGPR address), waited");
    run<5>("16 MFMA + 6 buffer_load_dword (SGPR resource, VGPR offset), waited");
    run<6>("16 MFMA + 32 v_add + 8 ds_read_b128 + 6 ds_write_b32 + 6 global loads");
    return 0;
}
