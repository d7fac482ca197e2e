// Micro-benchmark (development aid): instruction issue behaviour of ONE wave per SIMD on gfx950 -- cycles per iteration of
// hand-written streams (s_memtime).  Build: hipcc --offload-arch=gfx950 -O3 issue.hip -o issue ; run: ./issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)
#define REP16(x) REP8(x) REP8(x)

template <int MODE>
__global__ void __launch_bounds__(256, 1) k(long long* out, float* sink, int iters) {
    __shared__ float lds[4096];
    float a = threadIdx.x * 0.5f, b = 1.25f, c = 0.75f, d = 2.0f, e = 3.0f, f = 4.0f;
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    f32x16 big = {0};
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    lds[threadIdx.x] = a;
    __syncthreads();
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // 16 dependent v_add
            asm volatile(REP16("v_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b));
        } else if (MODE == 1) {   // 16 independent v_add (4 chains)
            asm volatile(REP4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n")
                         : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        } else if (MODE == 2) {   // 4 MFMA back to back (independent accumulators)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %4, %5, %0\n v_mfma_f32_16x16x4_f32 %1, %4, %5, %1\n"
                         "v_mfma_f32_16x16x4_f32 %2, %4, %5, %2\n v_mfma_f32_16x16x4_f32 %3, %4, %5, %3\n"
                         : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3) : "v"(b), "v"(c));
        } else if (MODE == 3) {   // 4 x (MFMA + 4 independent v_add)
            asm volatile(
                "v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                "v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                "v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
        } else if (MODE == 4) {   // 4 x (MFMA + 6 independent v_add)
            asm volatile(
                REP4("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n v_add_f32 %4, %4, %9\n v_add_f32 %5, %5, %9\n")
                : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
        } else if (MODE == 5) {   // 4 x (MFMA whose A operand is produced by the preceding v_add + 4 other adds)
            asm volatile(
                REP4("v_add_f32 %4, %4, %8\n s_nop 1\n v_mfma_f32_16x16x4_f32 %0, %4, %9, %0\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
        } else if (MODE == 6) {   // 4 x (MFMA + 2 ds_read_b32 + 2 v_add), reads consumed next iteration
            asm volatile(
                REP4("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n ds_read_b32 %5, %10\n v_add_f32 %4, %4, %8\n ds_read_b32 %6, %10 offset:256\n v_add_f32 %7, %7, %8\n")
                "s_waitcnt lgkmcnt(0)\n"
                : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c), "v"((int)(threadIdx.x * 4)));
        } else if (MODE == 7) {   // 4 x (MFMA + cmp/cndmask chain: v_fma, v_mul, v_cmp, v_cndmask, v_cndmask) = the staging commit
            asm volatile(
                REP4("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_fma_f32 %4, %5, %8, %9\n v_mul_f32 %6, %4, %8\n v_cmp_lt_f32 vcc, 0, %4\n s_nop 1\n v_cndmask_b32 %4, %6, %4, vcc\n v_cndmask_b32 %7, 0, %4, vcc\n")
                : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c) : "vcc");
        } else if (MODE == 9) {   // 4 MFMA, then 16 v_add in 4 chains (batched instead of interleaved)
            asm volatile("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n"
                         "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n"
                         REP4("v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
        } else if (MODE == 10) {  // 16 v_add in 4 chains FIRST, then 4 MFMA
            asm volatile(REP4("v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         "v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_mfma_f32_16x16x4_f32 %1, %8, %9, %1\n"
                         "v_mfma_f32_16x16x4_f32 %2, %8, %9, %2\n v_mfma_f32_16x16x4_f32 %3, %8, %9, %3\n"
                         : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
        } else if (MODE == 11) {  // 4 x (MFMA + 1 v_add)
            asm volatile(REP4("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n v_add_f32 %4, %4, %8\n")
                         : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));
        } else if (MODE == 12) {  // 4 x (MFMA + 2 ds_read_b32), wait at the end
            asm volatile(REP4("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n ds_read_b32 %5, %10\n ds_read_b32 %6, %10 offset:256\n")
                         "s_waitcnt lgkmcnt(0)\n"
                         : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c), "v"((int)(threadIdx.x * 4)));
        } else if (MODE == 13) {  // 4 x (MFMA + 2 s_add)
            asm volatile(REP4("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n")
                         : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c) : "s40", "s41");
        } else if (MODE == 14) {  // 4 x (MFMA + global_load_dword), wait at the end
            asm volatile(REP4("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n global_load_dword %5, %10, off\n")
                         "s_waitcnt vmcnt(0)\n"
                         : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c), "v"(sink + threadIdx.x));
        } else if (MODE == 15) {  // 4 x (32x32x2 MFMA + 4 v_add)
            asm volatile(REP4("v_mfma_f32_32x32x2_f32 %0, %2, %3, %0\n v_add_f32 %1, %1, %2\n v_add_f32 %4, %4, %2\n v_add_f32 %5, %5, %2\n v_add_f32 %6, %6, %2\n")
                         : "+a"(big), "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
        } else if (MODE == 16) {  // 4 x 32x32x2 MFMA back to back
            asm volatile(REP4("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0\n") : "+a"(big) : "v"(b), "v"(c));
        } else if (MODE == 8) {   // 4 x (MFMA + ds_write_b32 + 3 v_add)
            asm volatile(
                REP4("v_mfma_f32_16x16x4_f32 %0, %8, %9, %0\n ds_write_b32 %10, %5\n v_add_f32 %4, %4, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                : "+a"(acc0), "+a"(acc1), "+a"(acc2), "+a"(acc3), "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c), "v"((int)(threadIdx.x * 4)) : "memory");
        }
    }
    long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n s_nop 15");
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    sink[blockIdx.x * 256 + threadIdx.x] = a + c + d + e + f + acc0[0] + acc1[1] + acc2[2] + acc3[3] + big[5] + lds[(threadIdx.x + 1) & 255];
}

template <int MODE> void run(const char* name, int per_iter_mfma) {
    long long* out; float* sink;
    hipMalloc(&out, 8); hipMalloc(&sink, 256 * 256 * 4);
    const int iters = 2000;
    k<MODE><<<256, 256>>>(out, sink, iters);
    k<MODE><<<256, 256>>>(out, sink, iters);
    hipDeviceSynchronize();
    long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-70s %8.1f ticks/iter", name, (double)h / iters);
    if (per_iter_mfma) printf("  (%6.1f per MFMA)", (double)h / iters / per_iter_mfma);
    printf("\n");
    hipFree(out); hipFree(sink);
}

int main() {
    run<0>("16 dependent v_add_f32", 0);
    run<1>("16 v_add_f32 in 4 independent chains", 0);
    run<2>("4 MFMA 16x16x4 f32 back to back", 4);
    run<3>("4 x (MFMA + 4 independent v_add)", 4);
    run<4>("4 x (MFMA + 6 independent v_add)", 4);
    run<5>("4 x (v_add -> s_nop 1 -> MFMA reading it, + 3 v_add)", 4);
    run<6>("4 x (MFMA + 2 ds_read_b32 + 2 v_add), wait at the end", 4);
    run<7>("4 x (MFMA + fma/mul/cmp/nop/cndmask/cndmask chain)", 4);
    run<8>("4 x (MFMA + ds_write_b32 + 3 v_add)", 4);
    run<9>("4 MFMA, then 16 v_add (batched)", 4);
    run<10>("16 v_add, then 4 MFMA (batched)", 4);
    run<11>("4 x (MFMA + 1 v_add)", 4);
    run<12>("4 x (MFMA + 2 ds_read_b32), wait at the end", 4);
    run<13>("4 x (MFMA + 2 s_add_u32)", 4);
    run<14>("4 x (MFMA + global_load_dword), wait at the end", 4);
    run<15>("4 x (32x32x2 MFMA (dependent chain) + 4 v_add)", 4);
    run<16>("4 x 32x32x2 MFMA back to back (dependent chain)", 4);
    return 0;
}
