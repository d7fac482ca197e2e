// Development aid: issue cost (cycles per wave64 instruction, one wave per SIMD, independent chains of 8) of the vector instructions the split-precision kernels
// spend their staging time on: v_fma_f32, v_cvt_f16_f32, v_cvt_f32_f16, v_cvt_pk_f16_f32 (gfx950), v_fma_mix_f32, v_pk_add_f32, v_and_b32, v_cndmask_b32.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int OP, int WPS> __global__ void __launch_bounds__(256 * WPS, 1) k(long long* cyc, float* sink, int iters) {
    float v[8], w[8];
    unsigned sc[8];
    for (int i = 0; i < 8; ++i) sc[i] = __builtin_amdgcn_readfirstlane(i + iters);
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pk[8], pw[8];
    for (int i = 0; i < 8; ++i) { pk[i] = f2{threadIdx.x * 0.001f + i, 1.0f}; pw[i] = f2{1.0f + i * 0.125f, 0.5f}; }
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x * 0.001f + i; w[i] = 1.0f + i * 0.125f; }
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
#define CVT16(i) asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(v[i]) : "v"(w[i]));
#define CVT32(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[i]) : "v"(w[i]));
#define CVTPK(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(v[i]) : "v"(w[i]));
#define MIX(i) asm volatile("v_fma_mix_f32 %0, %1, %1, %0 op_sel_hi:[1,0,0]" : "+v"(v[i]) : "v"(w[i]));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[i]) : "v"(pw[i]));
#define AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define SUB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
#define SALU(i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc[i]) : : "scc");
#define MIXED(i) asm volatile("v_fma_f32 %0, %0, %2, %2\n\ts_add_u32 %1, %1, 1" : "+v"(v[i]), "+s"(sc[i]) : "v"(w[i]) : "scc");
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[i]) : "v"(pw[i]));
        if (OP == 0) { REP8(FMA) }
        if (OP == 1) { REP8(CVT16) }
        if (OP == 2) { REP8(CVT32) }
        if (OP == 3) { REP8(CVTPK) }
        if (OP == 4) { REP8(MIX) }
        if (OP == 6) { REP8(AND) }
        if (OP == 7) { REP8(SUB) }
        if (OP == 8) { REP8(SALU) }
        if (OP == 9) { REP8(MIXED) }
        if (OP == 10) { REP8(PKFMA) }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += v[i] + (float)sc[i] + pk[i][0] + pk[i][1];
    sink[blockIdx.x * 256 * WPS + threadIdx.x] = s;
}
template <int OP, int WPS> void run(const char* name) {
    long long* cyc; float* sink;
    hipMalloc(&cyc, 8); hipMalloc(&sink, 256 * 256 * 4 * WPS);
    const int iters = 20000;
    k<OP, WPS><<<256, 256 * WPS>>>(cyc, sink, 100);
    k<OP, WPS><<<256, 256 * WPS>>>(cyc, sink, iters);
    hipDeviceSynchronize();
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %d wave%s per SIMD: %.2f cycles per instruction of a wave = %.2f per SIMD (8 independent chains)\n", name, WPS, WPS > 1 ? "s" : " ", (double)c / (iters * 8.0), (double)c / (iters * 8.0) / WPS);
    hipFree(cyc); hipFree(sink);
}
int main() {
    run<0, 1>("v_fma_f32"); run<0, 2>("v_fma_f32"); run<0, 4>("v_fma_f32");
    run<10, 1>("v_pk_fma_f32"); run<10, 2>("v_pk_fma_f32"); run<10, 4>("v_pk_fma_f32");
    run<1, 1>("v_cvt_f16_f32"); run<3, 1>("v_cvt_pk_f16_f32"); run<4, 1>("v_fma_mix_f32"); run<6, 1>("v_and_b32"); run<6, 2>("v_and_b32");
    run<8, 1>("s_add_u32"); run<8, 2>("s_add_u32");
    run<9, 1>("v_fma_f32 + s_add_u32 (pair)"); run<9, 2>("v_fma_f32 + s_add_u32 (pair)");
    return 0;
}
