// Shared device helpers for the gfx950 kernels of the sliding-window segmentation path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/monai_amd.h"

// value barrier for the optimiser (no instruction): keeps a scalar chain out of SLP vectorisation where packing costs more moves than it
// saves.  The SIMT emulator of tests/emu compiles these sources for the host, where the register constraint does not exist.
#ifdef MH_SIMT_EMULATOR
#define MH_OPAQUE(x) ((void)0)
#define MH_OPAQUE_S(x) ((void)0)
#else
#define MH_OPAQUE(x) asm volatile("" : "+v"(x))
#define MH_OPAQUE_S(x) asm volatile("" : "+s"(x))      // the same for a wave-uniform value (scalar register)
#endif

// LDS-DMA: global memory -> LDS without a register round trip (buffer_load_dword ... lds: lane L of the issuing wave lands at LDS offset M0 + 4 L), completion
// counted by vmcnt like any vector-memory load.  Nothing orders a later ds_read behind it except the issuing wave's own s_waitcnt vmcnt + a barrier for the other waves'
// reads (MI355X_MICROARCH.md, "Two waves per SIMD", item 7).  Two things about hipcc (ROCm 7.2) shape the macros: (i) given the builtin forms of the load it puts an
// s_waitcnt vmcnt(0) of its own in front of the next LDS read that MAY alias the destination -- any read of a ring addressed by a run-time slot -- which empties the
// ring; the load is therefore inline assembly (the compiler then knows nothing of it: its own wait counts for other loads only get more conservative, never wrong);
// (ii) __syncthreads() carries a release fence that waits for every outstanding vector-memory operation: MH_VMCNT_BARRIER is the builtin wait + the bare s_barrier.
// MH_LDS_DMA_F32(base, byte offset of this lane, LDS_WAVE_BASE): `base` and LDS_WAVE_BASE wave-uniform.  The SIMT emulator copies at issue.
typedef int mh_i32x4 __attribute__((ext_vector_type(4)));
#ifdef MH_SIMT_EMULATOR
#define MH_LDS_DMA_F32(BASE, VOFF, LDS_WAVE_BASE) ((void)((LDS_WAVE_BASE)[threadIdx.x & 63] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(BASE) + (VOFF))))
#define MH_VMCNT_BARRIER(N) __syncthreads()
#define MH_VMCNT_WAIT(N) ((void)0)
#else
#define MH_LDS_DMA_F32(BASE, VOFF, LDS_WAVE_BASE)                                                                                             \
    {                                                                                                                                         \
        const unsigned long long b_ = (unsigned long long)(BASE);      /* raw buffer descriptor: base, stride 0, 2 GB of records, dword format */ \
        const mh_i32x4 r_ = {__builtin_amdgcn_readfirstlane((int)(unsigned)b_), __builtin_amdgcn_readfirstlane((int)((unsigned)(b_ >> 32) & 0xffffu)), 0x7fffffff, 0x00020000}; \
        /* the low 32 bits of a generic pointer into LDS are its LDS byte offset (the aperture sits in the high half); an addrspacecast here trips hipcc (ROCm 7.2: illegal V_CMP of src_shared_base) */ \
        const unsigned m_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(LDS_WAVE_BASE)); \
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(m_), "v"(VOFF), "s"(r_) : "memory", "m0"); \
    }
#define MH_VMCNT_IMM(N) (0x0f70 | ((N) & 15) | ((((N) >> 4) & 3) << 14))      /* gfx9 s_waitcnt encoding: vmcnt in bits 3:0 and 15:14, expcnt / lgkmcnt left at "no wait" */
#define MH_VMCNT_BARRIER(N) do { __builtin_amdgcn_s_waitcnt(MH_VMCNT_IMM(N)); __builtin_amdgcn_s_barrier(); } while (0)
#define MH_VMCNT_WAIT(N) __builtin_amdgcn_s_waitcnt(MH_VMCNT_IMM(N))
#endif

// write-once result rows of the streaming transforms (Gaussian): non-temporal stores -- 0.231 vs 0.236 ms per 512^3 volume at 9 taps, 0.418 vs 0.435 at 17
// (measured with a -DMH_DEV_NT_STORES build in round 3; for the blend, whose loads and stores interleave per voxel, non-temporal accesses cost 30 %)
#define MH_STREAM_STORE4(ptr, val) __builtin_nontemporal_store((val), reinterpret_cast<mh::f32x4*>(ptr))

namespace mh {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// Activation tensor view, NCDHW fp32: W contiguous, H stride W, D stride H*W, C stride D*H*W, batch
// stride free (so a view can be a channel range of a wider buffer, e.g. one half of a concat buffer).
// `nrm` (may be null) holds one float4 {alpha, beta, slope, bound} per (n, c): the value a consumer must
// see is  y = fma(x, alpha, beta);  y = y > 0 ? y : y * slope  -- i.e. InstanceNorm(affine) followed
// by LeakyReLU, deferred from the producer to the consumer's load (alpha=1, beta=0, slope=1: identity).
// `bound` >= max |y| over the (n, c) plane when the producer knows one (0 = none given; NaN / inf = the plane holds,
// or its statistics were, non-finite): the split-precision convolution scales its input by a power of two taken from it.
struct Tensor {
    float* data;
    long long n_stride;
    const float* nrm;
    long long nrm_n_stride;
    int N, C, D, H, W;
};

__host__ __device__ inline Tensor from_c(const mh_tensor5& t) {
    Tensor r;
    r.data = t.data; r.n_stride = t.n_stride; r.nrm = t.nrm; r.nrm_n_stride = t.nrm_n_stride;
    r.N = t.N; r.C = t.C; r.D = t.D; r.H = t.H; r.W = t.W;
    return r;
}

__device__ __forceinline__ float act(float x, float alpha, float beta, float slope) {
    float y = fmaf(x, alpha, beta);
    return y > 0.0f ? y : y * slope;
}

__device__ __forceinline__ float4 load_nrm(const Tensor& t, int n, int c) {
    if (t.nrm == nullptr) return make_float4(1.0f, 0.0f, 1.0f, 0.0f);
    return *reinterpret_cast<const float4*>(t.nrm + (long long)n * t.nrm_n_stride + 4LL * c);
}

// ---- magnitude bounds of raw (identity-record) tensors --------------------------------------------------------------
// A producer that writes a raw tensor whose consumer may be the split-precision convolution leaves max |value| per
// (n, c) in the 4th float of the tensor's identity record {1, 0, 1, bound}: nrm_identity_kernel resets the records
// (bound = FLT_MIN: "known, all zero so far"), every wave of the producer folds its values with ONE integer atomicMax
// on the bit pattern of |v| -- the order of non-negative floats is the order of their bits, and inf / NaN sort above
// every finite value, so a non-finite element poisons the bound instead of vanishing in an fmax.
constexpr float MH_BOUND_FLOOR = 1.17549435e-38f;      // FLT_MIN
__device__ __forceinline__ unsigned abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned wave_umax(unsigned m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned t = (unsigned)__shfl_xor((int)m, o);
        m = t > m ? t : m;
    }
    return m;
}
// the whole wave calls this (all lanes active or idle lanes passing 0); slot = &record[3]
__device__ __forceinline__ void bound_commit(unsigned m, float* slot) {
    m = wave_umax(m);
    if ((threadIdx.x & 63) == 0 && m != 0u) atomicMax(reinterpret_cast<unsigned*>(slot), m);
}
__device__ __forceinline__ float* bound_slot(const Tensor& t, int n, int c) {
    return const_cast<float*>(t.nrm) + (long long)n * t.nrm_n_stride + 4LL * c + 3;
}

// Statistics record used for instance norm: element count, mean, and M2 = sum (x - mean)^2.
// Combination is Chan et al.'s pairwise update, safe for empty sides.
struct Stat {
    float n, mean, m2;
};
__device__ __forceinline__ Stat stat_merge(Stat a, Stat b) {
    Stat r;
    r.n = a.n + b.n;
    if (r.n <= 0.0f) { r.mean = 0.0f; r.m2 = 0.0f; return r; }
    const float d = b.mean - a.mean;
    const float f = b.n / r.n;
    r.mean = a.mean + d * f;
    r.m2 = a.m2 + b.m2 + d * d * a.n * f;
    return r;
}
// The float of the lane to the left / right (DPP wave_shr:1 / wave_shl:1, one v_mov_b32_dpp, no LDS); lane 0 / lane 63 keep `edge`.  The value barrier is
// needed: hipcc's SLP pass merges four such calls on the elements of a vector into ONE dpp of element 0 (seen with ROCm 7.2: wrong results).
__device__ __forceinline__ float lane_left(float edge, float v) {
    int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false);
    MH_OPAQUE(r);
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float lane_right(float edge, float v) {
    int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false);
    MH_OPAQUE(r);
    return __builtin_bit_cast(float, r);
}

// the same merge without a branch (selects only): for epilogues that are interleaved with matrix instructions -- a branch would end the scheduling region.
// Bit-identical to stat_merge for finite inputs.
__device__ __forceinline__ Stat stat_merge_nb(Stat a, Stat b) {
    Stat r;
    r.n = a.n + b.n;
    const bool e = r.n <= 0.0f;
    const float d = b.mean - a.mean;
    const float f = b.n / (e ? 1.0f : r.n);
    r.mean = e ? 0.0f : a.mean + d * f;
    r.m2 = e ? 0.0f : a.m2 + b.m2 + d * d * a.n * f;
    return r;
}

// XCD-aware, bijective remap of a 1-D block index: the dispatcher places block b on XCD b % 8
// (observed, used for L2 locality only), so give each XCD a contiguous run of logical tiles.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nb) {
    const unsigned q = nb / 8u, r = nb % 8u, x = b % 8u, i = b / 8u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + i;
}

}  // namespace mh
