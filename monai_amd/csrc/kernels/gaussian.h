// Separable 3-D Gaussian smoothing (GaussianSmooth / GaussianFilter), all three axes in ONE pass over HBM.
//
// Reference: monai/transforms/intensity/array.py:1590-1622 -> GaussianFilter (monai/networks/layers/simplelayers.py:
// 542-595) -> separable_filtering :207-249: per axis a zero-padded copy (F.pad) and a depthwise F.conv3d with a
// [C,1,k,1,1]-style kernel -- three padded copies + three conv passes, ~12x the volume in HBM traffic.
// Here a workgroup owns a 16 x 64 (y, x) column of one channel volume and streams along one z-chunk:
//   plane z (with its y/x halo, zeros outside the image; prefetched into registers one plane ahead) -> LDS -> x-pass
//   -> LDS -> y-pass -> a per-thread register ring of the last RK filtered planes; the z-pass is a dot product of the
//   ring with the z-kernel.  Every input element is read once (plus the halos, served by L2: the launch is XCD-aware)
//   and every output written once: 8 B per voxel, the algorithmic minimum.
// The kernel is VALU-issue bound, not HBM bound, unless the taps are packed: every multiply-add below is a
// v_pk_fma_f32 on an (x, x+1) pair of neighbouring outputs.  The x-pass gets its register-aligned pairs for the odd
// taps from a second copy of the staged plane shifted by one float.
// fp32 throughout, taps accumulated in ascending tap order per axis with fused multiply-adds; the axis order (x, y,
// then z) differs from the reference's (first axis first); both only reorder fp32 roundings (tests: 1e-5 tolerance;
// the reference's own tests use 1e-4).
#pragma once
#include "common.h"

namespace mh {

constexpr int GS_MAX_TAPS = 33;   // per axis (sigma up to 4 at the reference's truncation of 4 sigma)
constexpr int GS_TX = 64, GS_TY = 16;

struct GaussArgs {
    int NC, D, H, W;
    int zchunk, nchunk;                                        // the z axis is cut into nchunk runs of zchunk output planes
    int pair_ok;                                               // W even and dst 8-byte aligned: (x, x+1) outputs go out as one store
    float kz[GS_MAX_TAPS], ky[GS_MAX_TAPS], kx[GS_MAX_TAPS];   // each zero-padded symmetrically to the kernel's RK taps
};

__device__ __forceinline__ f32x2 gs_fma(float k, f32x2 v, f32x2 acc) {
    const f32x2 kk = {k, k};
    return __builtin_elementwise_fma(kk, v, acc);
}

// RK = taps per axis (compile time: every tap loop is unrolled, weights sit in SGPRs), HR = halo.  ISO: the three axis
// kernels are identical (scalar sigma), one set of weights is kept (a third of the SGPRs).
// Thread t owns the column pair x = 2 (t & 31), +1 and the rows 2 (t >> 5), +1 of the 16 x 64 tile.
template <int RK, bool ISO>
__global__ void __launch_bounds__(256) gauss3d_stream_kernel(const float* __restrict__ src, float* __restrict__ dst, GaussArgs a) {
    constexpr int HR = (RK - 1) / 2;
    constexpr int INH = GS_TY + 2 * HR;
    constexpr int XSPAN = (RK + 3 + 3) / 4 * 4;              // floats one x-pass task reads per copy (whole float4s)
    constexpr int INW = GS_TX - 4 + XSPAN;                   // row pitch of the staged plane (multiple of 4)
    constexpr int NLOAD = (INH * INW + 255) / 256;
    constexpr int NTASK = INH * (GS_TX / 4);                 // x-pass tasks: (row, quad of 4 outputs)
    __shared__ __attribute__((aligned(16))) float in_a[INH * INW];       // staged plane
    __shared__ __attribute__((aligned(16))) float in_b[INH * INW + 4];   // the same shifted left by one float
    __shared__ __attribute__((aligned(16))) float mid_s[INH * GS_TX];
    const int tid = threadIdx.x;
    const float* wkx = a.kx;
    const float* wky = ISO ? a.kx : a.ky;
    const float* wkz = ISO ? a.kx : a.kz;
    // 1-D launch, XCD-aware: each XCD's L2 gets a contiguous run of (tile, chunk, volume) work items, so the in-plane
    // halo shared by neighbouring tiles is an L2 hit instead of a second HBM read.
    const int tiles_x = (a.W + GS_TX - 1) / GS_TX, tiles = tiles_x * ((a.H + GS_TY - 1) / GS_TY);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = (int)(lid % (unsigned)tiles);
    lid /= (unsigned)tiles;
    const int chunk = (int)(lid % (unsigned)a.nchunk), nc = (int)(lid / (unsigned)a.nchunk);
    const int tx0 = (tile % tiles_x) * GS_TX, ty0 = (tile / tiles_x) * GS_TY;
    const int zs = chunk * a.zchunk, ze = min(zs + a.zchunk, a.D);      // output planes of this workgroup
    const int zfirst = max(zs - HR, 0), zlast = min(ze + HR, a.D);      // source planes it filters; planes beyond the volume are zero
    const long long plane = (long long)a.H * a.W;
    const float* vol = src + (long long)nc * a.D * plane;
    float* ovol = dst + (long long)nc * a.D * plane;
    const int xp = tid & 31, y2 = (tid >> 5) * 2;

    // staging positions of this thread inside the halo plane (fixed for the whole march along z)
    int goff[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
        const int i = tid + 256 * j;
        const int ly = i / INW, lx = i - ly * INW;
        const int gy = ty0 + ly - HR, gx = tx0 + lx - HR;
        const bool ok = i < INH * INW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[j] = ok ? gy * a.W + gx : -1;
    }
    float pre[NLOAD];
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) pre[j] = goff[j] >= 0 ? vol[(long long)zfirst * plane + goff[j]] : 0.0f;

    f32x2 ring[2][RK];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int k = 0; k < RK; ++k) ring[p][k] = f32x2{0.0f, 0.0f};

    const int gx = tx0 + 2 * xp;
    const bool ok0 = ty0 + y2 < a.H && gx < a.W, ok1 = ty0 + y2 + 1 < a.H && gx < a.W;
    const bool pair_store = a.pair_ok != 0 && gx + 1 < a.W;
    const bool second = gx + 1 < a.W;
    float* obase = ovol + (long long)(ty0 + y2) * a.W + gx;

    // z-pass of the ring + store of output plane zo (both rows' accumulation chains interleaved)
#define GS_EMIT(ZO)                                                                                   \
    {                                                                                                 \
        f32x2 acc0 = {0.0f, 0.0f}, acc1 = {0.0f, 0.0f};                                               \
        _Pragma("unroll") for (int k = 0; k < RK; ++k) {                                              \
            acc0 = gs_fma(wkz[k], ring[0][k], acc0);                                                  \
            acc1 = gs_fma(wkz[k], ring[1][k], acc1);                                                  \
        }                                                                                             \
        float* o = obase + (long long)(ZO) * plane;                                                   \
        if (pair_store) {                                                                             \
            if (ok0) *reinterpret_cast<f32x2*>(o) = acc0;                                             \
            if (ok1) *reinterpret_cast<f32x2*>(o + a.W) = acc1;                                       \
        } else {                                                                                      \
            if (ok0) { o[0] = acc0[0]; if (second) o[1] = acc0[1]; }                                  \
            if (ok1) { o[a.W] = acc1[0]; if (second) o[a.W + 1] = acc1[1]; }                          \
        }                                                                                             \
    }

    for (int z = zfirst; z < zlast; ++z) {
#pragma unroll
        for (int j = 0; j < NLOAD; ++j) {
            const int i = tid + 256 * j;
            if (i < INH * INW) {
                in_a[i] = pre[j];
                if (i > 0) in_b[i - 1] = pre[j];
            }
        }
        __syncthreads();
        if (z + 1 < zlast) {                // next plane's loads fly while this plane is filtered
            const float* pl = vol + (long long)(z + 1) * plane;
#pragma unroll
            for (int j = 0; j < NLOAD; ++j) pre[j] = goff[j] >= 0 ? pl[goff[j]] : 0.0f;
        }
        // x-pass: outputs (4q, 4q+1) and (4q+2, 4q+3) of one row; even taps pair up in in_a, odd taps in in_b
        for (int task = tid; task < NTASK; task += 256) {
            const int r = task / (GS_TX / 4), q = task - r * (GS_TX / 4);
            const f32x4* ra = reinterpret_cast<const f32x4*>(in_a + r * INW + 4 * q);
            const f32x4* rb = reinterpret_cast<const f32x4*>(in_b + r * INW + 4 * q);
            f32x2 wa[XSPAN / 2], wb[XSPAN / 2];          // wa[i] = (in[2i], in[2i+1]), wb[i] = (in[2i+1], in[2i+2])
#pragma unroll
            for (int v = 0; v < XSPAN / 4; ++v) {
                const f32x4 t4 = ra[v], u4 = rb[v];
                wa[2 * v] = f32x2{t4[0], t4[1]}; wa[2 * v + 1] = f32x2{t4[2], t4[3]};
                wb[2 * v] = f32x2{u4[0], u4[1]}; wb[2 * v + 1] = f32x2{u4[2], u4[3]};
            }
            f32x2 o0 = {0.0f, 0.0f}, o1 = {0.0f, 0.0f};
#pragma unroll
            for (int k = 0; k < RK; ++k) {
                if (k % 2 == 0) { o0 = gs_fma(wkx[k], wa[k / 2], o0); o1 = gs_fma(wkx[k], wa[k / 2 + 1], o1); }
                else { o0 = gs_fma(wkx[k], wb[k / 2], o0); o1 = gs_fma(wkx[k], wb[k / 2 + 1], o1); }
            }
            *reinterpret_cast<f32x4*>(mid_s + r * GS_TX + 4 * q) = f32x4{o0[0], o0[1], o1[0], o1[1]};
        }
        __syncthreads();
        // y-pass for this thread's two rows (chains interleaved), pushed into the ring
        f32x2 col[2 + 2 * HR];
#pragma unroll
        for (int j = 0; j < 2 + 2 * HR; ++j) col[j] = *reinterpret_cast<const f32x2*>(mid_s + (y2 + j) * GS_TX + 2 * xp);
        f32x2 v0 = {0.0f, 0.0f}, v1 = {0.0f, 0.0f};
#pragma unroll
        for (int k = 0; k < RK; ++k) {
            v0 = gs_fma(wky[k], col[k], v0);
            v1 = gs_fma(wky[k], col[k + 1], v1);
        }
#pragma unroll
        for (int k = 0; k < RK - 1; ++k) { ring[0][k] = ring[0][k + 1]; ring[1][k] = ring[1][k + 1]; }
        ring[0][RK - 1] = v0; ring[1][RK - 1] = v1;
        if (z - HR >= zs) GS_EMIT(z - HR)
    }
    // planes beyond the end of the volume filter to exactly zero (zero padding): drain the ring
    for (int z = zlast; z < ze + HR; ++z) {
#pragma unroll
        for (int k = 0; k < RK - 1; ++k) { ring[0][k] = ring[0][k + 1]; ring[1][k] = ring[1][k + 1]; }
        ring[0][RK - 1] = f32x2{0.0f, 0.0f}; ring[1][RK - 1] = f32x2{0.0f, 0.0f};
        if (z - HR >= zs) GS_EMIT(z - HR)
    }
#undef GS_EMIT
}

// ---------------------------------------------------------------------------------------------------
// Row-vector variant for float4-aligned volumes (W % 4 == 0, 16-byte aligned pointers) -- the same single HBM pass,
// organised so that the memory system, not LDS staging or VALU issue, is the bound:
//   * a workgroup = 8 waves owns a 16 (y) x 256 (x) output tile of one z-chunk; a lane owns FOUR consecutive x (one 16-byte
//     load / store per row and plane: 1 KB per wave-instruction);
//   * x-pass in registers: the wave's row segment goes through a wave-private LDS row (one 16-byte write, 2 * ceil(HR / 4)
//     16-byte reads of the neighbouring lanes' vectors -- no second, shifted copy of the plane, no workgroup barrier);
//     the 4 * HL-float halo on either side of the segment is loaded by the first / last HL lanes;
//   * y-pass through one LDS plane of x-filtered rows (16-byte reads), double buffered: ONE barrier per plane;
//   * z-pass from a register ring of the last RK y-filtered rows (two output rows per wave); for RK <= 9 the march is unrolled
//     RK planes at a time so the ring never moves.
// Tap order, fused multiply-adds and zero padding as in gauss3d_stream_kernel: the results are bit-identical to it.
// (hipcc's SLP pass packs the x-pass tap chains into v_pk_fma_f32 although their register pairs are unaligned; keeping them scalar
// with a value barrier after every FMA was measured SLOWER: 0.313 vs 0.272 ms per 512^3 volume.)
constexpr int GV_LANES = 64, GV_TX = 4 * GV_LANES, GV_TY = 16;
// waves per workgroup: 16 (one output row each, 86 registers, four waves per SIMD) up to 9 taps -- measured 0.241 ms per 512^3 volume
// against 0.268 with 8 waves x 2 rows (206 registers, two waves per SIMD: issue- and latency-bound) and 0.286 for the tile kernel;
// 17 taps keep 8 x 2 (their register ring does not fit 128 registers: 0.47 vs 0.58 ms)
constexpr int gv_waves(int rk) { return rk <= 9 ? 16 : 8; }

template <int RK, bool ISO>
__global__ void __launch_bounds__(64 * gv_waves(RK)) gauss3d_rowvec_kernel(const float* __restrict__ src, float* __restrict__ dst, GaussArgs a) {
    constexpr int GV_WAVES = gv_waves(RK), GV_ROWS = GV_TY / GV_WAVES;
    constexpr int HR = (RK - 1) / 2;
    constexpr int HL = (HR + 3) / 4;                         // halo vectors on either side of a lane's vector
    constexpr int NR = GV_TY + 2 * HR;                       // rows of the tile's halo plane
    constexpr int XR = (NR + GV_WAVES - 1) / GV_WAVES;       // rows a wave x-filters per plane
    constexpr int RP = GV_LANES + 2 * HL;                    // vectors in a wave-private row
    constexpr int OFF = 4 * HL - HR;                         // first tap of output 0 inside the window of 4 + 8 HL floats
    constexpr int DEPTH = 1;                                 // source planes in flight (two were measured: no gain, +40 registers)
    __shared__ __attribute__((aligned(16))) float mid[2][NR * GV_TX];
    __shared__ __attribute__((aligned(16))) float rowbuf[GV_WAVES][RP * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* wkx = a.kx;
    const float* wky = ISO ? a.kx : a.ky;
    const float* wkz = ISO ? a.kx : a.kz;
    const int tiles_x = (a.W + GV_TX - 1) / GV_TX, tiles = tiles_x * ((a.H + GV_TY - 1) / GV_TY);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = (int)(lid % (unsigned)tiles);
    lid /= (unsigned)tiles;
    const int chunk = (int)(lid % (unsigned)a.nchunk), nc = (int)(lid / (unsigned)a.nchunk);
    const int tx0 = (tile % tiles_x) * GV_TX, ty0 = (tile / tiles_x) * GV_TY;
    const int zs = chunk * a.zchunk, ze = min(zs + a.zchunk, a.D);
    const int zfirst = max(zs - HR, 0), zlast = min(ze + HR, a.D);
    const long long plane = (long long)a.H * a.W;
    const float* vol = src + (long long)nc * a.D * plane;
    float* ovol = dst + (long long)nc * a.D * plane;

    // this lane's global offsets inside a plane: centre vector of each of its x-pass rows, and -- for the first / last HL
    // lanes -- the halo vector left / right of the wave's segment (-1: outside the image, zeros)
    const int gx = tx0 + 4 * lane;
    int coff[XR], eoff[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int r = wave + GV_WAVES * i, gy = ty0 + r - HR;
        const bool rok = r < NR && gy >= 0 && gy < a.H;
        coff[i] = rok && gx < a.W ? gy * a.W + gx : -1;
        int hx = -1;
        if (lane < HL) hx = tx0 - 4 * HL + 4 * lane;
        else if (lane >= GV_LANES - HL) hx = tx0 + GV_TX + 4 * (lane - (GV_LANES - HL));
        eoff[i] = rok && hx >= 0 && hx < a.W && (lane < HL || lane >= GV_LANES - HL) ? gy * a.W + hx : -1;
    }
    f32x4 cur[DEPTH][XR], ext[DEPTH][XR];          // two planes in flight: one alone leaves too few bytes outstanding per CU for HBM's latency
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
#define GV_LOAD(Z, SET)                                                                               \
    {                                                                                                 \
        const float* pl_ = vol + (long long)(Z) * plane;                                              \
        _Pragma("unroll") for (int i = 0; i < XR; ++i) {                                              \
            cur[SET][i] = coff[i] >= 0 ? *reinterpret_cast<const f32x4*>(pl_ + coff[i]) : zero4;      \
            ext[SET][i] = eoff[i] >= 0 ? *reinterpret_cast<const f32x4*>(pl_ + eoff[i]) : zero4;      \
        }                                                                                             \
    }
    f32x4 ring[GV_ROWS][RK];
#pragma unroll
    for (int p = 0; p < GV_ROWS; ++p)
#pragma unroll
        for (int k = 0; k < RK; ++k) ring[p][k] = zero4;
    const int y2 = GV_ROWS * wave;
    bool rowok[GV_ROWS];
#pragma unroll
    for (int p = 0; p < GV_ROWS; ++p) rowok[p] = ty0 + y2 + p < a.H && gx < a.W;
    float* const obase = ovol + (long long)(ty0 + y2) * a.W + gx;
    f32x4* const myrow = reinterpret_cast<f32x4*>(rowbuf[wave]);
    int buf = 0;

    // x- and y-pass of source plane Z -> this wave's GV_ROWS y-filtered row vectors v_[]; prefetches plane Z + 1
#define GV_PLANE(Z, V, SET)                                                                           \
    {                                                                                                 \
        float* mb_ = mid[buf];                                                                        \
        _Pragma("unroll") for (int i = 0; i < XR; ++i) {                                              \
            const int r_ = wave + GV_WAVES * i;                                                       \
            if (r_ < NR) {                                                                            \
                myrow[HL + lane] = cur[SET][i];                                                       \
                if (lane < HL) myrow[lane] = ext[SET][i];                                             \
                else if (lane >= GV_LANES - HL) myrow[2 * HL + lane] = ext[SET][i];                   \
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                \
                __builtin_amdgcn_wave_barrier();                                                      \
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                \
                float w_[4 + 8 * HL];                                                                 \
                _Pragma("unroll") for (int j = 0; j <= 2 * HL; ++j) {                                 \
                    const f32x4 t_ = j == HL ? cur[SET][i] : myrow[lane + j];                         \
                    w_[4 * j] = t_[0]; w_[4 * j + 1] = t_[1]; w_[4 * j + 2] = t_[2]; w_[4 * j + 3] = t_[3]; \
                }                                                                                     \
                float o_[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                               \
                _Pragma("unroll") for (int k = 0; k < RK; ++k)                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) o_[e] = fmaf(wkx[k], w_[OFF + e + k], o_[e]); \
                *reinterpret_cast<f32x4*>(mb_ + r_ * GV_TX + 4 * lane) = f32x4{o_[0], o_[1], o_[2], o_[3]}; \
                __builtin_amdgcn_wave_barrier();        /* the row buffer is reused by the next row */ \
            }                                                                                         \
        }                                                                                             \
        if ((Z) + DEPTH < zlast) GV_LOAD((Z) + DEPTH, SET)                                            \
        __syncthreads();                                                                              \
        _Pragma("unroll") for (int p = 0; p < GV_ROWS; ++p) V[p] = zero4;                             \
        _Pragma("unroll") for (int j = 0; j < RK + GV_ROWS - 1; ++j) {                                \
            const f32x4 m_ = *reinterpret_cast<const f32x4*>(mb_ + (y2 + j) * GV_TX + 4 * lane);      \
            _Pragma("unroll") for (int p = 0; p < GV_ROWS; ++p)                                       \
                if (j - p >= 0 && j - p < RK)       /* two v_pk_fma_f32: the kernel is VALU-issue bound, not HBM bound, without them */ \
                    V[p] = __builtin_elementwise_fma(f32x4{wky[j - p], wky[j - p], wky[j - p], wky[j - p]}, m_, V[p]); \
        }                                                                                             \
        buf ^= 1;                                                                                     \
    }
    // z-pass over the ring in slot order S0, S0+1, ... (oldest first) and store of output plane ZO
#define GV_EMIT(ZO, S0)                                                                               \
    {                                                                                                 \
        _Pragma("unroll") for (int p = 0; p < GV_ROWS; ++p) {                                         \
            f32x4 acc_ = zero4;                                                                       \
            _Pragma("unroll") for (int k = 0; k < RK; ++k) {                                          \
                const f32x4 rv_ = ring[p][((S0) + k) % RK];                                           \
                acc_ = __builtin_elementwise_fma(f32x4{wkz[k], wkz[k], wkz[k], wkz[k]}, rv_, acc_);     \
            }                                                                                         \
            if (rowok[p]) MH_STREAM_STORE4(obase + (long long)(ZO) * plane + (long long)p * a.W, acc_);   \
        }                                                                                             \
    }

    GV_LOAD(zfirst, 0)
    if (DEPTH > 1 && zfirst + 1 < zlast) GV_LOAD(zfirst + 1, DEPTH - 1)
    // the ring is shifted by register moves (2 x (RK - 1) 16-byte moves per plane).  Keeping it still -- unrolling the march RK planes
    // deep, or a switch over the slot -- makes hipcc spill or copy the whole ring at every merge point (measured: 254 registers + 34
    // spilled; 401 v_mov_b64 per kernel).
    for (int z = zfirst; z < ze + HR; ++z) {
        f32x4 v_[GV_ROWS];
        if (z < zlast) {
            if (DEPTH > 1 && ((z - zfirst) & 1)) GV_PLANE(z, v_, DEPTH - 1) else GV_PLANE(z, v_, 0)
        } else {
#pragma unroll
            for (int p = 0; p < GV_ROWS; ++p) v_[p] = zero4;              // beyond the volume: zero padding
        }
#pragma unroll
        for (int p = 0; p < GV_ROWS; ++p) {
#pragma unroll
            for (int k = 0; k < RK - 1; ++k) ring[p][k] = ring[p][k + 1];
            ring[p][RK - 1] = v_[p];
        }
        if (z - HR >= zs) GV_EMIT(z - HR, 0)
    }
#undef GV_EMIT
#undef GV_PLANE
#undef GV_LOAD
}

// ---------------------------------------------------------------------------------------------------
// Up to 9 taps per axis, round 3: the row-vector kernel with the two things removed that its instruction stream showed beside the arithmetic
// (per plane and wave 110 vector instructions, of which 54 are the taps' v_pk_fma_f32):
//   * x-pass neighbours by DPP instead of LDS: a lane needs the float4 of the lanes to its left and right (halo of at most 4 floats); two
//     `v_mov_b32 wave_shr:1 / wave_shl:1` per float take them from the neighbouring lanes' REGISTERS -- no wave-private LDS row, no LDS round trip and
//     no wave barriers in front of the taps.  Lanes 0 and 63 keep the `edge` operand: the halo vector they loaded from the neighbouring tile;
//   * z-pass by forward accumulation instead of a register ring: a y-filtered row of source plane z is added, with tap k, to the accumulator of
//     output plane z + HR - k; the accumulator that received its last tap is stored.  The march is unrolled RK planes deep, so that "which accumulator"
//     is a compile-time index: the 28 register moves per plane that shifted the ring are gone.  An output's taps still arrive in ascending order, each
//     as one fused multiply-add starting from zero: bit-identical to the ring form (and to gauss3d_stream_kernel).
// Measured on the MI355X (profiles/r03_gauss_ab.json, 4 x 512^3, bit-identical outputs): 9 taps 0.223-0.226 ms per volume against 0.225-0.237 for
// gauss3d_rowvec_kernel, 5 taps 0.200 = 0.200.  Also measured and not kept: the forward accumulation with the LDS x-pass (30 % fewer vector instructions
// than this form, 0.250 ms: slower) and 8-row tiles with 8 waves (two to three workgroups per CU, 1.33 x the halo rows: 0.246 ms) -- neither the
// instruction count nor the occupancy is what holds this kernel at 0.57-0.60 of 8 TB/s; a 16-byte copy on the same boxes reaches 0.77.
template <int RK, bool ISO>
__global__ void __launch_bounds__(1024) gauss3d_rowdpp_kernel(const float* __restrict__ src, float* __restrict__ dst, GaussArgs a) {
    constexpr int WAVES = 16, TY = WAVES;
    constexpr int HR = (RK - 1) / 2;
    static_assert(HR >= 1 && HR <= 4, "one halo vector per side");
    constexpr int NR = TY + 2 * HR;                          // rows of the tile's halo plane
    constexpr int XR = (NR + WAVES - 1) / WAVES;             // rows a wave x-filters per plane
    constexpr int OFF = 4 - HR;                              // first tap of output 0 inside the window of 12 floats (left | own | right vector)
    __shared__ __attribute__((aligned(16))) float mid[2][NR * GV_TX];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* wkx = a.kx;
    const float* wky = ISO ? a.kx : a.ky;
    const float* wkz = ISO ? a.kx : a.kz;
    const int tiles_x = (a.W + GV_TX - 1) / GV_TX, tiles = tiles_x * ((a.H + TY - 1) / TY);
    unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
    const int tile = (int)(lid % (unsigned)tiles);
    lid /= (unsigned)tiles;
    const int chunk = (int)(lid % (unsigned)a.nchunk), nc = (int)(lid / (unsigned)a.nchunk);
    const int tx0 = (tile % tiles_x) * GV_TX, ty0 = (tile / tiles_x) * TY;
    const int zs = chunk * a.zchunk, ze = min(zs + a.zchunk, a.D);
    const int zfirst = max(zs - HR, 0), zlast = min(ze + HR, a.D);
    const long long plane = (long long)a.H * a.W;
    const float* vol = src + (long long)nc * a.D * plane;
    float* ovol = dst + (long long)nc * a.D * plane;

    // Loads go through a raw buffer over this channel volume (the launcher keeps it below 2 GB): byte offsets of this lane inside a plane -- its own
    // vector of each of its x-pass rows and, lanes 0 / 63, the halo vector left / right of the wave's 256-float segment -- or an offset beyond the
    // buffer, which reads zeros: zero padding without a branch or a select.  The plane offset rides in the instruction's scalar offset.
    const auto vrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(vol), 0, (int)((long long)a.D * plane * 4), 0x00020000);
    constexpr unsigned GD_ZERO = 0x80000000u;
    const int gx = tx0 + 4 * lane;
    unsigned coff[XR], eoff[XR];
#pragma unroll
    for (int i = 0; i < XR; ++i) {
        const int r = wave + WAVES * i, gy = ty0 + r - HR;
        const bool rok = r < NR && gy >= 0 && gy < a.H;
        coff[i] = rok && gx < a.W ? 4u * (unsigned)(gy * a.W + gx) : GD_ZERO;
        const int hx = lane == 0 ? tx0 - 4 : tx0 + GV_TX;
        eoff[i] = rok && (lane == 0 || lane == GV_LANES - 1) && hx >= 0 && hx < a.W ? 4u * (unsigned)(gy * a.W + hx) : GD_ZERO;
    }
    // Round 4: with 7-9 taps the plane BEHIND the next one is requested too (its loads stay in flight for a whole plane step; the move into the working set is
    // renamed away inside the unrolled group): 0.239 against 0.246-0.247 ms per 512^3 volume, bit-identical; with 5 taps the same change costs 5 % (0.217 against
    // 0.2065: 16 more registers, one resident wave less), and the streaming resample lost 7-30 % to it (a resident workgroup less) -- profiles/r04_hbm_two_planes_ahead_ab.jsonl
    constexpr bool TWO_AHEAD = RK >= 7;
    f32x4 cur[XR], ext[XR], ncur[XR], next_[XR];
    const f32x4 zero4 = {0.0f, 0.0f, 0.0f, 0.0f};
    unsigned loff = (unsigned)((long long)zfirst * plane * 4);      // byte offset of the next source plane to load
    float* optr = ovol + (long long)(ty0 + wave) * a.W + gx + ((long long)zfirst - HR) * plane;      // output plane z - HR of the current step
#define GD_LOAD(C_, E_)                                                                               \
    {                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < XR; ++i) {                                              \
            C_[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, coff[i], loff, 0)); \
            E_[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vrsrc, eoff[i], loff, 0)); \
        }                                                                                             \
        loff += (unsigned)(plane * 4);                                                                \
    }
    f32x4 acc[RK];
#pragma unroll
    for (int k = 0; k < RK; ++k) acc[k] = zero4;
    const bool rowok = ty0 + wave < a.H && gx < a.W;
    int buf = 0;

    // x- and y-pass of source plane Z -> this wave's y-filtered row vector V; plane Z + 1 moves into the working set, plane Z + 2 is requested
#define GD_PLANE(Z, V)                                                                                \
    {                                                                                                 \
        float* mb_ = mid[buf];                                                                        \
        _Pragma("unroll") for (int i = 0; i < XR; ++i) {                                              \
            const int r_ = wave + WAVES * i;                                                          \
            if (r_ < NR) {                                                                            \
                float w_[12];                                                                         \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                       \
                    w_[e] = lane_left(ext[i][e], cur[i][e]);                                          \
                    w_[4 + e] = cur[i][e];                                                            \
                    w_[8 + e] = lane_right(ext[i][e], cur[i][e]);                                     \
                }                                                                                     \
                float o_[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                               \
                _Pragma("unroll") for (int k = 0; k < RK; ++k)                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e) o_[e] = fmaf(wkx[k], w_[OFF + e + k], o_[e]); \
                *reinterpret_cast<f32x4*>(mb_ + r_ * GV_TX + 4 * lane) = f32x4{o_[0], o_[1], o_[2], o_[3]}; \
            }                                                                                         \
        }                                                                                             \
        if (TWO_AHEAD) {                                                                              \
            if ((Z) + 1 < zlast) { _Pragma("unroll") for (int i = 0; i < XR; ++i) { cur[i] = ncur[i]; ext[i] = next_[i]; } }   /* requested one plane step ago */ \
            if ((Z) + 2 < zlast) GD_LOAD(ncur, next_)                                                 \
        } else if ((Z) + 1 < zlast) GD_LOAD(cur, ext)                                                 \
        __syncthreads();                                                                              \
        V = zero4;                                                                                    \
        _Pragma("unroll") for (int j = 0; j < RK; ++j) {                                              \
            const f32x4 m_ = *reinterpret_cast<const f32x4*>(mb_ + (wave + j) * GV_TX + 4 * lane);    \
            V = __builtin_elementwise_fma(f32x4{wky[j], wky[j], wky[j], wky[j]}, m_, V);              \
        }                                                                                             \
        buf ^= 1;                                                                                     \
    }

    GD_LOAD(cur, ext)
    if (TWO_AHEAD && zfirst + 1 < zlast) GD_LOAD(ncur, next_)
    // step s of a group of RK planes: slot of output plane zo = (zo - zfirst) mod RK, a compile-time number inside the unrolled group
    for (int zb = zfirst; zb < ze + HR; zb += RK) {
#pragma unroll
        for (int s = 0; s < RK; ++s) {
            const int z = zb + s;
            if (z < ze + HR) {                                  // (no `break`: the group must unroll completely for the slots to be registers)
                f32x4 v_ = zero4;                               // beyond the volume: zero padding
                if (z < zlast) GD_PLANE(z, v_)
#pragma unroll
                for (int k = 0; k < RK; ++k) {
                    const int slot = (s + HR - k + RK) % RK;    // output plane z + HR - k
                    const f32x4 wk_ = {wkz[k], wkz[k], wkz[k], wkz[k]};
                    acc[slot] = __builtin_elementwise_fma(wk_, v_, k == 0 ? zero4 : acc[slot]);
                }
                const int done = (s - HR + RK) % RK;            // output plane z - HR has all its taps
                if (z - HR >= zs && rowok) MH_STREAM_STORE4(optr, acc[done]);
                optr += plane;
            }
        }
    }
#undef GD_PLANE
#undef GD_LOAD
}

}  // namespace mh
