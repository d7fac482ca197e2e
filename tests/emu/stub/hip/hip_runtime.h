// TEST INFRASTRUCTURE ONLY -- a tiny SIMT emulator that stands in for <hip/hip_runtime.h> so the
// UNMODIFIED product sources (monai_amd/csrc/*.hip + kernels/*.h) can be compiled for x86 with
// clang++ and their index math / LDS staging / barrier logic / MFMA fragment handling can be checked
// against the oracle on the GPU-less build container.  It is never built into, loaded by, or
// reachable from the product package; `tests/emu/build_emu.py` puts this directory first on the
// include path, which is the only way it is ever seen.
//
// Model: one fiber per HIP thread, fibers of a block run on one OS thread (blocks are spread over OS
// threads).  `__syncthreads()` and wave-collectives (shuffles, MFMA) are rendezvous points.  MFMA
// uses the lane<->element maps of /opt/skills/guides/cdna_hip_programming.md section 3 and the
// k-ordered fmaf chain it documents as bit-exact for v_mfma_f32_32x32x2_f32.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

// ------------------------------------------------------------------ vector types
struct uint3 { unsigned x, y, z; };
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct alignas(16) uint4 { unsigned int x, y, z, w; };
struct alignas(8) uint2 { unsigned int x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

// ------------------------------------------------------------------ fibers
extern "C" void emu_switch(void** save_sp, void* load_sp);
__asm__(
    ".text\n.globl emu_switch\n.type emu_switch,@function\nemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size emu_switch, .-emu_switch\n");

namespace emu {
enum State { RUNNABLE = 0, WAVE_WAIT = 1, BLOCK_WAIT = 2, DONE = 3 };
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    uint3 tid{0, 0, 0};
    int lane = 0, wave = 0;
    State state = DONE;
};
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;

struct Ctx {
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    Fiber* cur = nullptr;
    const std::function<void()>* body = nullptr;
    uint3 blockIdx{0, 0, 0};
    dim3 blockDim, gridDim;
    // wave exchange slots: 16 waves x 64 lanes x 2 x 16 bytes
    alignas(16) unsigned char slots[16][64][2][16];
    Ctx() : fibers(MAX_THREADS) {}
    ~Ctx() {
        for (auto& f : fibers) free(f.stack);
    }
};
inline Ctx& ctx() {
    static thread_local Ctx c;
    return c;
}

inline void yield_to_sched(State s) {
    Ctx& c = ctx();
    Fiber* f = c.cur;
    f->state = s;
    emu_switch(&f->sp, c.sched_sp);
}
static void fiber_entry() {
    Ctx& c = ctx();
    (*c.body)();
    yield_to_sched(DONE);
    abort();  // never resumed
}
inline void prepare(Fiber& f) {
    uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + STACK_BYTES) & ~uintptr_t(15);
    void** p = reinterpret_cast<void**>(top);
    *(--p) = nullptr;                                   // fake return address of fiber_entry
    *(--p) = reinterpret_cast<void*>(&fiber_entry);     // popped by `ret`
    for (int i = 0; i < 6; ++i) *(--p) = nullptr;       // rbp rbx r12..r15
    f.sp = p;
    f.state = RUNNABLE;
}
inline void resume(Fiber& f) {
    Ctx& c = ctx();
    c.cur = &f;
    emu_switch(&c.sched_sp, f.sp);
    c.cur = nullptr;
}

inline void run_block(const std::function<void()>& body, dim3 grid, dim3 block, uint3 bidx) {
    Ctx& c = ctx();
    const int nt = int(block.x * block.y * block.z);
    if (nt > MAX_THREADS) { fprintf(stderr, "emu: block too large\n"); abort(); }
    const int nw = (nt + 63) / 64;
    c.body = &body; c.blockIdx = bidx; c.blockDim = block; c.gridDim = grid;
    for (int t = 0; t < nt; ++t) {
        Fiber& f = c.fibers[t];
        f.tid = uint3{unsigned(t % block.x), unsigned((t / block.x) % block.y), unsigned(t / (block.x * block.y))};
        f.lane = t & 63; f.wave = t >> 6;
        if (!f.stack) f.stack = static_cast<char*>(aligned_alloc(64, STACK_BYTES));
        prepare(f);
    }
    for (;;) {
        for (int w = 0; w < nw; ++w) {
            const int lo = w * 64, hi = std::min(nt, lo + 64);
            for (;;) {  // run this wave until every lane sits at a block barrier or has exited
                for (int t = lo; t < hi; ++t)
                    if (c.fibers[t].state == RUNNABLE) resume(c.fibers[t]);
                int nwave = 0, nother = 0;
                for (int t = lo; t < hi; ++t) {
                    State s = c.fibers[t].state;
                    nwave += s == WAVE_WAIT; nother += s != WAVE_WAIT;
                }
                if (nwave && nother) {
                    fprintf(stderr, "emu: divergent wave collective (wave %d: %d lanes at the collective, %d not)\n", w, nwave, nother);
                    abort();
                }
                if (!nwave) break;
                for (int t = lo; t < hi; ++t) c.fibers[t].state = RUNNABLE;  // rendezvous complete
            }
        }
        int nblock = 0, ndone = 0;
        for (int t = 0; t < nt; ++t) { nblock += c.fibers[t].state == BLOCK_WAIT; ndone += c.fibers[t].state == DONE; }
        if (ndone == nt) break;
        if (nblock + ndone != nt) { fprintf(stderr, "emu: scheduler stuck\n"); abort(); }
        if (ndone) { fprintf(stderr, "emu: __syncthreads() skipped by %d exited threads\n", ndone); abort(); }
        for (int t = 0; t < nt; ++t) c.fibers[t].state = RUNNABLE;
    }
}

inline int& num_os_threads() {
    static int n = std::max(1u, std::thread::hardware_concurrency());
    return n;
}
inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
    const long total = long(grid.x) * grid.y * grid.z;
    const int nth = int(std::min<long>(num_os_threads(), total));
    std::atomic<long> next{0};
    auto worker = [&]() {
        for (;;) {
            long b = next.fetch_add(1);
            if (b >= total) break;
            uint3 bi{unsigned(b % grid.x), unsigned((b / grid.x) % grid.y), unsigned(b / (long(grid.x) * grid.y))};
            run_block(body, grid, block, bi);
        }
    };
    if (nth <= 1) { worker(); return; }
    std::vector<std::thread> pool;
    for (int i = 0; i < nth; ++i) pool.emplace_back(worker);
    for (auto& t : pool) t.join();
}

inline void wave_sync() { yield_to_sched(WAVE_WAIT); }
template <class T> inline T wave_read(T v, int src) {
    static_assert(sizeof(T) <= 16, "wave_read");
    Ctx& c = ctx();
    Fiber* f = c.cur;
    memcpy(c.slots[f->wave][f->lane][0], &v, sizeof(T));
    wave_sync();
    T r;
    memcpy(&r, c.slots[f->wave][src & 63][0], sizeof(T));
    wave_sync();
    return r;
}
typedef float f32x16_t __attribute__((ext_vector_type(16)));
inline f32x16_t mfma_f32_32x32x2f32(float a, float b, f32x16_t cin, int, int, int) {
    Ctx& c = ctx();
    Fiber* f = c.cur;
    memcpy(c.slots[f->wave][f->lane][0], &a, 4);
    memcpy(c.slots[f->wave][f->lane][1], &b, 4);
    wave_sync();
    const int lane = f->lane, col = lane & 31;
    f32x16_t d = cin;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = cin[r];
        for (int k = 0; k < 2; ++k) {
            float av, bv;
            memcpy(&av, c.slots[f->wave][row + 32 * k][0], 4);   // A[i=row][k] lives in lane row+32k
            memcpy(&bv, c.slots[f->wave][col + 32 * k][1], 4);   // B[k][j=col] lives in lane col+32k
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    wave_sync();
    return d;
}
// v_mfma_f32_32x32x16_bf16: A[i = l & 31][k = 8 (l >> 5) + j], B[k = 8 (l >> 5) + j][n = l & 31], j = 0..7; D as 32x32x2.
// Products of two bf16 values are exact in fp32; the accumulation is fp32 (the hardware's internal order is not specified).
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
inline f32x16_t mfma_f32_32x32x16_bf16(bf16x8_t a, bf16x8_t b, f32x16_t cin, int, int, int) {
    Ctx& c = ctx();
    Fiber* f = c.cur;
    memcpy(c.slots[f->wave][f->lane][0], &a, 16);
    memcpy(c.slots[f->wave][f->lane][1], &b, 16);
    wave_sync();
    const int lane = f->lane, col = lane & 31;
    f32x16_t d = cin;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = cin[r];
        for (int g = 0; g < 2; ++g) {
            bf16x8_t av, bv;
            memcpy(&av, c.slots[f->wave][row + 32 * g][0], 16);
            memcpy(&bv, c.slots[f->wave][col + 32 * g][1], 16);
            for (int j = 0; j < 8; ++j) acc += (float)av[j] * (float)bv[j];
        }
        d[r] = acc;
    }
    wave_sync();
    return d;
}
// v_mfma_f32_32x32x16_f16: the same operand layout with fp16 elements (products of two fp16 values are exact in fp32)
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
inline f32x16_t mfma_f32_32x32x16_f16(f16x8_t a, f16x8_t b, f32x16_t cin, int, int, int) {
    Ctx& c = ctx();
    Fiber* f = c.cur;
    memcpy(c.slots[f->wave][f->lane][0], &a, 16);
    memcpy(c.slots[f->wave][f->lane][1], &b, 16);
    wave_sync();
    const int lane = f->lane, col = lane & 31;
    f32x16_t d = cin;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float acc = cin[r];
        for (int g = 0; g < 2; ++g) {
            f16x8_t av, bv;
            memcpy(&av, c.slots[f->wave][row + 32 * g][0], 16);
            memcpy(&bv, c.slots[f->wave][col + 32 * g][1], 16);
            for (int j = 0; j < 8; ++j) acc += (float)av[j] * (float)bv[j];
        }
        d[r] = acc;
    }
    wave_sync();
    return d;
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x32_f16: A[i = l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][n = l & 15], j = 0..7; D: col = l & 15, row = 4 (l >> 4) + r
inline f32x4_t mfma_f32_16x16x32_f16(f16x8_t a, f16x8_t b, f32x4_t cin, int, int, int) {
    Ctx& c = ctx();
    Fiber* f = c.cur;
    memcpy(c.slots[f->wave][f->lane][0], &a, 16);
    memcpy(c.slots[f->wave][f->lane][1], &b, 16);
    wave_sync();
    const int lane = f->lane, col = lane & 15;
    f32x4_t d = cin;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = cin[r];
        for (int g = 0; g < 4; ++g) {
            f16x8_t av, bv;
            memcpy(&av, c.slots[f->wave][row + 16 * g][0], 16);
            memcpy(&bv, c.slots[f->wave][col + 16 * g][1], 16);
            for (int j = 0; j < 8; ++j) acc += (float)av[j] * (float)bv[j];
        }
        d[r] = acc;
    }
    wave_sync();
    return d;
}
// v_mfma_f32_16x16x4_f32: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15], D: col = l & 15, row = 4 (l >> 4) + r
inline f32x4_t mfma_f32_16x16x4f32(float a, float b, f32x4_t cin, int, int, int) {
    Ctx& c = ctx();
    Fiber* f = c.cur;
    memcpy(c.slots[f->wave][f->lane][0], &a, 4);
    memcpy(c.slots[f->wave][f->lane][1], &b, 4);
    wave_sync();
    const int lane = f->lane, col = lane & 15;
    f32x4_t d = cin;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (lane >> 4) + r;
        float acc = cin[r];
        for (int k = 0; k < 4; ++k) {
            float av, bv;
            memcpy(&av, c.slots[f->wave][row + 16 * k][0], 4);
            memcpy(&bv, c.slots[f->wave][col + 16 * k][1], 4);
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    wave_sync();
    return d;
}
// DPP wave shifts by one lane (the only controls the product uses): lane l reads lane l-1 (0x138 wave_shr:1) or l+1 (0x130 wave_shl:1); the lane without
// a source keeps `old` (bound_ctrl = 0)
inline int update_dpp(int old, int src, int ctrl, int, int, bool) {
    const int lane = ctx().cur->lane;
    if (ctrl == 0x138) { const int v = wave_read(src, (lane + 63) & 63); return lane == 0 ? old : v; }
    if (ctrl == 0x130) { const int v = wave_read(src, (lane + 1) & 63); return lane == 63 ? old : v; }
    std::fprintf(stderr, "emu: unsupported dpp control 0x%x\n", ctrl);
    std::abort();
}
// raw buffer stores: a {base, num_records} descriptor; a store whose 32-bit offset (+ size) leaves [0, num_records) is dropped, as the hardware does
struct BufRsrc { char* base; unsigned num_records; };
template <class P> inline BufRsrc make_buffer_rsrc(P* p, short, int num_records, int) { return BufRsrc{reinterpret_cast<char*>(p), (unsigned)num_records}; }
template <class V> inline void raw_buffer_store_b128(V data, BufRsrc r, unsigned voffset, unsigned soffset, int) {
    static_assert(sizeof(V) == 16, "b128");
    const unsigned long long off = (unsigned long long)voffset + soffset;
    if (off + 16ull <= r.num_records) std::memcpy(r.base + off, &data, 16);
}
template <class V> inline void raw_buffer_store_b64(V data, BufRsrc r, unsigned voffset, unsigned soffset, int) {
    static_assert(sizeof(V) == 8, "b64");
    const unsigned long long off = (unsigned long long)voffset + soffset;
    if (off + 8ull <= r.num_records) std::memcpy(r.base + off, &data, 8);
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
inline u32x2_t raw_buffer_load_b64(BufRsrc r, unsigned voffset, unsigned soffset, int) {      // out of range reads return zeros
    u32x2_t v = {0u, 0u};
    const unsigned long long off = (unsigned long long)voffset + soffset;
    if (off + 8ull <= r.num_records) std::memcpy(&v, r.base + off, 8);
    return v;
}
inline u32x4_t raw_buffer_load_b128(BufRsrc r, unsigned voffset, unsigned soffset, int) {      // out of range reads return zeros
    u32x4_t v = {0u, 0u, 0u, 0u};
    const unsigned long long off = (unsigned long long)voffset + soffset;
    if (off + 16ull <= r.num_records) std::memcpy(&v, r.base + off, 16);
    return v;
}
inline unsigned raw_buffer_load_b32(BufRsrc r, unsigned voffset, unsigned soffset, int) {
    unsigned v = 0u;
    const unsigned long long off = (unsigned long long)voffset + soffset;
    if (off + 4ull <= r.num_records) std::memcpy(&v, r.base + off, 4);
    return v;
}
}  // namespace emu

// ------------------------------------------------------------------ HIP surface used by the product sources
#define MH_SIMT_EMULATOR 1
#define __global__ static
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define threadIdx (emu::ctx().cur->tid)
#define blockIdx (emu::ctx().blockIdx)
#define blockDim (emu::ctx().blockDim)
#define gridDim (emu::ctx().gridDim)
#define __syncthreads() emu::yield_to_sched(emu::BLOCK_WAIT)
#define __builtin_amdgcn_mfma_f32_32x32x2f32 emu::mfma_f32_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 emu::mfma_f32_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 emu::mfma_f32_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_32x32x16_f16 emu::mfma_f32_32x32x16_f16
#define __builtin_amdgcn_mfma_f32_16x16x32_f16 emu::mfma_f32_16x16x32_f16
#define __builtin_amdgcn_update_dpp emu::update_dpp
#define __builtin_amdgcn_fmed3f(a, b, c) std::fmax(std::fmin((a), (b)), std::fmin(std::fmax((a), (b)), (c)))      // v_med3_f32 (finite / infinite operands)
#define __builtin_amdgcn_make_buffer_rsrc emu::make_buffer_rsrc
#define __builtin_amdgcn_raw_buffer_store_b128 emu::raw_buffer_store_b128
#define __builtin_amdgcn_raw_buffer_store_b64 emu::raw_buffer_store_b64
#define __builtin_amdgcn_raw_buffer_load_b128 emu::raw_buffer_load_b128
#define __builtin_amdgcn_raw_buffer_load_b64 emu::raw_buffer_load_b64
#define __builtin_amdgcn_raw_buffer_load_b32 emu::raw_buffer_load_b32
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_wave_barrier() emu::wave_sync()
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)

template <class T> static inline T __shfl_xor(T v, int mask, int = 64) { return emu::wave_read(v, emu::ctx().cur->lane ^ mask); }
template <class T> static inline T __shfl_down(T v, int d, int = 64) {
    int l = emu::ctx().cur->lane;
    return emu::wave_read(v, l + d < 64 ? l + d : l);
}
template <class T> static inline T __shfl(T v, int src, int = 64) { return emu::wave_read(v, src); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fsqrt_rn(float a) { return std::sqrt(a); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
// 24-bit multiplies: the hardware uses the low 24 bits of each operand (sign-extended for the signed form) -- emulated as such, so that an operand out of range shows
static inline unsigned __umul24(unsigned a, unsigned b) { return (unsigned)((unsigned long long)(a & 0xffffffu) * (b & 0xffffffu)); }
static inline int __mul24(int a, int b) { return (int)((long long)((a << 8) >> 8) * ((b << 8) >> 8)); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
using std::fmaf;
using std::max;
using std::min;

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
// floating-point atomics: workgroups run on a pool of host threads, so these are real compare-and-swap loops
template <class T, class U> static inline T emu_atomic_add(T* p, T v) {
    static_assert(sizeof(T) == sizeof(U), "size");
    U* q = reinterpret_cast<U*>(p);
    U old = __atomic_load_n(q, __ATOMIC_RELAXED), neu;
    T cur;
    do {
        memcpy(&cur, &old, sizeof(T));
        const T sum = cur + v;
        memcpy(&neu, &sum, sizeof(T));
    } while (!__atomic_compare_exchange_n(q, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return cur;
}
static inline int atomicMin(int* p, int v) { const int o = *p; if (v < o) *p = v; return o; }     // LDS only: fibers of one block share a host thread
static inline int atomicMax(int* p, int v) { const int o = *p; if (v > o) *p = v; return o; }
// global memory (magnitude bounds of raw tensors): workgroups run on several host threads
static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline float unsafeAtomicAdd(float* p, float v) { return emu_atomic_add<float, unsigned int>(p, v); }
static inline double unsafeAtomicAdd(double* p, double v) { return emu_atomic_add<double, unsigned long long>(p, v); }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return 0; }
template <typename K> static inline hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, K, int, size_t) { *n = 4; return 0; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
#define hipLaunchKernelGGL(kern, grid, block, shmem, stream, ...)                          \
    do {                                                                                     \
        auto _k = kern;                                                                      \
        emu::launch(dim3(grid), dim3(block), [=]() { _k(__VA_ARGS__); });                    \
    } while (0)
