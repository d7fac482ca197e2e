// Micro-benchmark (development aid): do the VALU instructions of one wave overlap the fp32 MFMAs of ANOTHER wave on the
// same SIMD of gfx950?  512-thread blocks, one per CU: waves 0-3 (one per SIMD) run MFMA bursts, waves 4-7 run v_add
// chains (roles: 1 = MFMA, 2 = VALU, 0 = idle).  Time per role combination from s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)

__global__ void __launch_bounds__(512, 1) k(long long* out, float* sink, int iters, int role_lo, int role_hi) {
    const int wave = threadIdx.x >> 6;
    const int role = wave < 4 ? role_lo : role_hi;
    float a = threadIdx.x * 0.5f, b = 1.25f, c = 0.75f, d = 2.0f, e = 3.0f;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    if (role == 1) {
        for (int i = 0; i < iters; ++i)
            asm volatile(REP4("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                              "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n")
                         : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3) : "v"(b), "v"(c));
    } else if (role == 2) {
        for (int i = 0; i < iters; ++i)
            asm volatile(REP16("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
                         : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
    } else if (role == 3) {     // LDS reads
        __shared__ float lds[4096];
        lds[threadIdx.x] = a;
        for (int i = 0; i < iters; ++i)
            asm volatile(REP16("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:1024\n") "s_waitcnt lgkmcnt(0)\n"
                         : "=v"(d), "=v"(e) : "v"((int)((threadIdx.x & 255) * 4)));
    }
    long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[wave] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = a + c + d + e + acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

static void run(const char* name, int lo, int hi) {
    long long* out; float* sink;
    hipMalloc(&out, 64); hipMalloc(&sink, 256 * 512 * 4);
    hipMemset(out, 0, 64);
    const int iters = 2000;
    k<<<256, 512>>>(out, sink, iters, lo, hi);
    k<<<256, 512>>>(out, sink, iters, lo, hi);
    hipDeviceSynchronize();
    long long h[8]; hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("%-52s  waves0-3: %9.1f   waves4-7: %9.1f  ticks/iter\n", name, (double)h[0] / iters, (double)h[4] / iters);
    hipFree(out); hipFree(sink);
}
int main() {
    run("16 MFMA alone (512 cycles of pipe)", 1, 0);
    run("64 v_add alone", 0, 2);
    run("MFMA waves + VALU waves together", 1, 2);
    run("MFMA + MFMA (two MFMA waves per SIMD)", 1, 1);
    run("VALU + VALU", 2, 2);
    run("32 ds_read alone", 0, 3);
    run("MFMA waves + LDS-read waves", 1, 3);
    return 0;
}
