// Micro-benchmark (development aid) for the two-piece fp16 split-precision convolution (kernels/conv3d_h2.h):
//  (1) does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (the low pieces of small activations are subnormal)?
//  (2) cycles per back-to-back v_mfma_f32_32x32x16_f16 with 1 and 2 waves per SIMD
//  (3) does another wave's VALU / LDS work on the same SIMD overlap with f16 MFMAs (it does not with fp32 MFMAs:
//      coexec.hip), and what do VALU / ds_read_b128 cost when they sit between the MFMAs of the SAME wave?
//   hipcc --offload-arch=gfx950 -O3 mfma_f16.hip -o /tmp/mfma_f16 && /tmp/mfma_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP2(x) x x
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

__global__ void denorm_kernel(float* out, float av, float bv) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f32x16 acc = {};
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}

#define MFMA8                                                                                               \
    REP2("v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n v_mfma_f32_32x32x16_f16 %1, %4, %5, %1\n"                 \
         "v_mfma_f32_32x32x16_f16 %2, %4, %5, %2\n v_mfma_f32_32x32x16_f16 %3, %4, %5, %3\n")

// ROLE of waves 4-7 (second wave of each SIMD): 0 absent, 1 the same MFMA stream, 2 v_add_f32 x 32 per iteration, 3 v_pk_add/cvt mix,
// 4 8 ds_read_b128 per iteration, 5 8 ds_write_b64 per iteration, 6 8 global_load_dwordx4 per iteration
// SELF: what waves 0-3 put between two bursts of 8 MFMAs: 0 nothing, 1 8 ds_read_b128 (no wait until the next burst's end), 2 16 v_add_f32, 3 both
template <int ROLE, int SELF>
__global__ void __launch_bounds__(512, 1) k(long long* out, float* sink, const float* src, int iters) {
    __shared__ float lds[16384];
    const int wave = threadIdx.x >> 6;
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * threadIdx.x); b[i] = (_Float16)0.5f; }
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, vb = 0.25f;
    f32x4 r0 = {}, r1 = {}, g0 = {}, g1 = {};
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = v0;
    const int laddr = (threadIdx.x & 255) * 16;
    const float* gp = src + threadIdx.x * 4;
    __syncthreads();
    if (wave >= 4 && ROLE == 0) return;
    long long t0 = __builtin_readcyclecounter();
    if (wave < 4 || ROLE == 1) {
        for (int i = 0; i < iters; ++i) {
            asm volatile(MFMA8 : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b));
            if (SELF == 1 || SELF == 3)
                asm volatile(REP4("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n") : "=v"(r0), "=v"(r1) : "v"(laddr));
            if (SELF == 2 || SELF == 3)
                asm volatile(REP4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(vb));
            if (SELF == 1 || SELF == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1));
        }
    } else {
        for (int i = 0; i < iters; ++i) {
            if (ROLE == 2)
                asm volatile(REP8("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(vb));
            if (ROLE == 3)
                asm volatile(REP8("v_cvt_pk_f16_f32 %0, %0, %1\n v_sub_f32 %1, %1, %4\n v_fma_f32 %2, %2, %4, %3\n v_max_f32 %3, %3, %4\n")
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(vb));
            if (ROLE == 4)
                asm volatile(REP4("ds_read_b128 %0, %2\n ds_read_b128 %1, %2 offset:4096\n") "s_waitcnt lgkmcnt(0)\n" : "=v"(r0), "=v"(r1) : "v"(laddr));
            if (ROLE == 5)
                asm volatile(REP4("ds_write_b64 %0, %1 offset:16384\n ds_write_b64 %0, %2 offset:24576\n") "s_waitcnt lgkmcnt(0)\n"
                             :: "v"(laddr >> 1), "v"(*(double*)&r0), "v"(*(double*)&r1) : "memory");
            if (ROLE == 6)
                asm volatile(REP4("global_load_dwordx4 %0, %2, off\n global_load_dwordx4 %1, %2, off offset:2048\n") "s_waitcnt vmcnt(0)\n"
                             : "=v"(g0), "=v"(g1) : "v"(gp));
        }
    }
    long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3] + v0 + v1 + v2 + v3 + r0[0] + r1[1] + g0[0] + g1[1];
}

template <int ROLE, int SELF> static void run(const char* name, int per_iter_other) {
    long long* out; float *sink, *src;
    hipMalloc(&out, 64); hipMalloc(&sink, 512 * 4 * 256); hipMalloc(&src, 1 << 22);
    hipMemset(src, 0, 1 << 22);
    const int iters = 4000;
    k<ROLE, SELF><<<256, 512>>>(out, sink, src, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<ROLE, SELF><<<256, 512>>>(out, sink, src, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double nmfma = 256.0 * (ROLE == 1 ? 8 : 4) * iters * 8.0;
    long long h[8];
    hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("%-64s mfma wave: %6.1f cyc/MFMA", name, (double)h[0] / (iters * 8.0));
    printf("  | wall %.3f ms = %.0f TFLOP/s f16 over 256 CUs", ms, nmfma * 32768.0 / ms / 1e9);
    if (ROLE >= 2) printf("   other wave: %6.1f cyc per op (%d ops/iter)", (double)h[4] / (iters * (double)per_iter_other), per_iter_other);
    printf("\n");
    hipFree(out); hipFree(sink); hipFree(src);
}

int main() {
    float* d; hipMalloc(&d, 4);
    const float cases[][2] = {{9.5367431640625e-07f, 256.f}, {5.9604644775390625e-08f, 1.f}, {3.0517578125e-05f, 3.0517578125e-05f}, {0.5f, 0.25f}};
    for (auto& c : cases) {
        denorm_kernel<<<1, 64>>>(d, c[0], c[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("denorm: a=%g b=%g -> mfma = %.10g   expected (subnormals kept) %.10g\n", c[0], c[1], h, 16.0 * (double)c[0] * (double)c[1]);
    }
    run<0, 0>("1 wave/SIMD, MFMA only", 0);
    run<1, 0>("2 waves/SIMD, both MFMA", 0);
    run<2, 0>("MFMA wave + v_add_f32 wave", 32);
    run<3, 0>("MFMA wave + cvt/sub/fma/max wave", 32);
    run<4, 0>("MFMA wave + ds_read_b128 wave", 8);
    run<5, 0>("MFMA wave + ds_write_b64 wave", 8);
    run<6, 0>("MFMA wave + global_load_dwordx4 wave", 8);
    run<0, 1>("1 wave/SIMD: 8 MFMA + 8 ds_read_b128 per iteration", 0);
    run<0, 2>("1 wave/SIMD: 8 MFMA + 16 v_add_f32 per iteration", 0);
    run<0, 3>("1 wave/SIMD: 8 MFMA + 8 ds_read_b128 + 16 v_add per iteration", 0);
    run<1, 3>("2 waves/SIMD: each 8 MFMA + 8 ds_read_b128 + 16 v_add per iter", 0);
    run<1, 1>("2 waves/SIMD: each 8 MFMA + 8 ds_read_b128 per iteration", 0);
    return 0;
}
