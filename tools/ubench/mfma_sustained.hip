// Micro-benchmark (development aid): what does the fp16 matrix pipe of an MI355X SUSTAIN?  The 2.5 PFLOP/s dense figure is 256 CUs x 4 SIMDs x
// 1024 FLOP/clk at 2.4 GHz; mfma_f16.hip reaches 2.0 PF in 0.5 ms bursts on smooth operands.  The split-precision convolution (conv3d_h2.h) runs
// for seconds on activation data, and the chip lowers its clock under it (DESIGN_HISTORY 4.1).  This program runs NOTHING BUT back-to-back
// v_mfma_f32_32x32x16_f16 (operands in registers, no memory traffic) for ~0.5 s per configuration and reports the rate and the clock of every
// launch: the rate on random operands is the practical ceiling for any kernel built on this instruction.
//   operand data: zero | smooth (0.001 tid, 0.5) | random normal fp16 | split pieces (a "hi" piece ~N(0,1) and a "lo" piece 2^-11 of it, alternating)
//   hipcc --offload-arch=gfx950 -O3 mfma_sustained.hip -o /tmp/mfma_sustained && /tmp/mfma_sustained
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <algorithm>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ inline float gauss(unsigned s) {      // Box-Muller on two hashed uniforms
    const float u1 = (hash32(s) >> 8) * (1.0f / 16777216.0f) + 1e-7f, u2 = (hash32(s ^ 0x9e3779b9u) >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
}

// 16 MFMAs per iteration over 4 accumulators; consecutive instructions use different A and B registers (the multiplier inputs toggle as they do
// in a real kernel, where every instruction has new operands)
#define MM(ACC, A, B) "v_mfma_f32_32x32x16_f16 %" #ACC ", %" #A ", %" #B ", %" #ACC "\n"
#define MFMA16                                                             \
    MM(0, 4, 8) MM(1, 5, 9) MM(2, 6, 10) MM(3, 7, 11)                      \
    MM(0, 5, 10) MM(1, 6, 11) MM(2, 7, 8) MM(3, 4, 9)                      \
    MM(0, 6, 9) MM(1, 7, 10) MM(2, 4, 11) MM(3, 5, 8)                      \
    MM(0, 7, 11) MM(1, 4, 10) MM(2, 5, 9) MM(3, 6, 8)

template <int WAVES>     // waves per workgroup: 4 = one per SIMD, 8 = two per SIMD
__global__ void __launch_bounds__(64 * WAVES, 1) k(long long* cyc, float* sink, int iters, int mode, unsigned seed) {
    f16x8 a[4], b[4];
    const unsigned id = (blockIdx.x * 512u + threadIdx.x) * 64u + seed;
    for (int r = 0; r < 4; ++r)
        for (int i = 0; i < 8; ++i) {
            float av = 0.0f, bv = 0.0f;
            if (mode == 1) { av = 0.001f * threadIdx.x; bv = 0.5f; }
            if (mode == 2) { av = gauss(id + r * 8 + i); bv = gauss(id + 32 + r * 8 + i); }
            if (mode == 3) {       // registers 0, 2: high pieces; 1, 3: low pieces (11 bits further down), as the three products of conv3d_h2.h see them
                const float sa = (r & 1) ? 4.8828125e-4f : 1.0f;
                av = sa * gauss(id + r * 8 + i); bv = ((r & 2) ? 4.8828125e-4f : 1.0f) * gauss(id + 32 + r * 8 + i);
            }
            a[r][i] = (_Float16)av; b[r][i] = (_Float16)(0.05f * bv);
        }
    f32x16 acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
        asm volatile(MFMA16 : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3)
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
}

template <int WAVES> static void run(const char* name, int mode, int iters, int launches) {
    long long* cyc; float* sink;
    hipMalloc(&cyc, 64); hipMalloc(&sink, 256 * 512 * 4);
    std::vector<hipEvent_t> ev(launches + 1);
    for (auto& e : ev) hipEventCreate(&e);
    k<WAVES><<<256, 64 * WAVES>>>(cyc, sink, 1000, mode, 1u);     // warm-up
    hipDeviceSynchronize();
    hipEventRecord(ev[0]);
    for (int l = 0; l < launches; ++l) {
        k<WAVES><<<256, 64 * WAVES>>>(cyc, sink, iters, mode, 7u + l);
        hipEventRecord(ev[l + 1]);
    }
    hipDeviceSynchronize();
    long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double flop = 256.0 * WAVES * iters * 16.0 * (2.0 * 32 * 32 * 16);
    std::vector<double> tf;
    for (int l = 0; l < launches; ++l) {
        float ms = 0.f;
        hipEventElapsedTime(&ms, ev[l], ev[l + 1]);
        tf.push_back(flop / (ms * 1e-3) / 1e12);
    }
    float ms_last = 0.f;
    hipEventElapsedTime(&ms_last, ev[launches - 1], ev[launches]);
    const double cyc_per_mfma = (double)c / (iters * 16.0);
    // the cycle counter of the last launch against its wall time: the clock the CU actually ran at (counter = shader clock on gfx9)
    const double mhz = (double)c / (ms_last * 1e3);
    std::vector<double> s = tf;
    std::sort(s.begin(), s.end());
    printf("%-46s %d wave%s/SIMD  first %7.1f  median %7.1f  last %7.1f TFLOP/s  = %.3f of 2.5 PF | %5.1f counter ticks / MFMA, counter %.0f MHz, %.1f ms / launch x %d\n",
           name, WAVES / 4, WAVES == 4 ? " " : "s", tf[0], s[s.size() / 2], tf.back(), s[s.size() / 2] / 2500.0, cyc_per_mfma, mhz, ms_last, launches);
    fflush(stdout);
    hipFree(cyc); hipFree(sink);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 60000;        // 60000 x 16 MFMAs x 32 cycles = 12.8 ms at 2.4 GHz
    const int launches = argc > 2 ? atoi(argv[2]) : 30;
    const char* names[4] = {"operands zero", "operands smooth (0.001 tid, 0.5)", "operands random normal fp16", "operands hi / lo split pieces (random)"};
    for (int mode = 0; mode < 4; ++mode) {
        run<4>(names[mode], mode, iters, launches);
        run<8>(names[mode], mode, iters / 2, launches);
    }
    return 0;
}
