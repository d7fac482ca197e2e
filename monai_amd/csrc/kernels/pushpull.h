// The full monai._C resampling surface: pull / push / count / spatial gradients and their backward passes, B-spline
// interpolation orders 0-7 per axis, the seven index-remapping boundary conditions.
//
// Reference: the `pushpull` dispatcher every entry point of monai/csrc/resample/pushpull.h:58-509 funnels into
// (flags do_pull / do_push / do_count / do_grad / do_sgrad, :24-50), CPU implementation monai/csrc/resample/pushpull_cpu.cpp:
//   check3d :786-838  -- read the coordinate, zero the outputs of a target voxel whose coordinate is out of bounds
//                        (only when !extrapolate; tolerance TINY = 5e-2 :67), pick the path;
//   interpolate3d :938-1146 -- generic path: per-axis tap ranges (interpolation_common.h bounds0-7), weights /
//                        first / second derivatives of the B-spline (fastweight / fastgrad / fasthess), index + sign of
//                        every tap (bounds_common.h), then taps z-outer, y, x-inner;
//   interpolate3d_trilinear :1476-1760, interpolate3d_nearest :2057-2098 -- the paths taken when all three orders
//                        are equal (`iso` :136) to 1 resp. 0; their arithmetic differs from the generic path's, so
//                        they are restated separately;
//   the 2-D / 1-D variants (:1152-1470, :1766-2050, :2104-2175) are the restrictions of the 3-D code to fewer axes.
//
// One kernel serves 1-D / 2-D / 3-D: tensors are viewed as 3-D with trailing axes of size 1 (`ndim` real axes).  A
// padded axis is sampled at coordinate 0: the linear path skips its far corner (weight 0; the reference's 2-D code
// has no such corner) and its near weight is exactly 1, the generic path gives it order 0 (one tap, weight 1,
// derivative 0), so every product and sum has the value the reference's lower-dimensional code computes.
// Gradient / target component axes have `ndim` entries, not 3.
//
// One thread per target voxel, lanes along the last axis (coalesced grid reads and output writes); `push` / `count`
// scatter with hardware floating-point atomics like the reference's CUDA path (summation order is not defined there
// either: parity for those two is to rounding, everything else is bit-exact against the reference's CPU build).
// Arithmetic notes: the reference writes its polynomial constants as double literals, so with scalar_t = float every
// weight is evaluated in double and rounded once on return (or at each assignment to a scalar_t variable); the
// functions below keep that evaluation order.  -ffp-contract=off (no fused multiply-add) like the CPU build.
#pragma once
#include "common.h"
#include "grid_pull.h"

namespace mh {

enum { PP_NEAREST = 0, PP_LINEAR = 1, PP_GENERIC = 2 };

struct PushPullArgs {
    int B, C, X, Y, Z, Xo, Yo, Zo;
    int bound[3], interp[3];
    int ndim;           // real spatial axes = components of grid / gradient / target-gradient vectors
    int extrapolate, path;
    int do_pull, do_push, do_count, do_grad, do_sgrad;
    int trgt_k;         // 0: target is (B, C, spatial); > 0: target carries ndim gradient components (backward of sgrad)
};

// ---- B-spline basis, orders 0-7 (interpolation_common.h:52-640).  `x` is the signed distance coordinate - node.
// S(...) marks a store into a scalar_t variable of the reference (a rounding to T when T = float).
template <typename T> __device__ __forceinline__ T pp_weight(int order, T x) {
#define S(e) ((T)(e))
    x = fabs(x);
    switch (order) {
        case 0: return (T)1;
        case 2:
            if (x < 0.5) return S(0.75 - x * x);
            x = S(1.5 - x);
            return S(0.5 * x * x);
        case 3:
            if (x < 1.) return S((x * x * (x - 2.) * 3. + 4.) / 6.);
            x = S(2. - x);
            return S((x * x * x) / 6.);
        case 4:
            if (x < 0.5) {
                x = x * x;
                return S(x * (x * 0.25 - 0.625) + 115. / 192.);
            }
            if (x < 1.5) return S(x * (x * (x * (5. - x) / 6. - 1.25) + 5. / 24.) + 55. / 96.);
            x = S(x - 2.5);
            x = x * x;
            return S((x * x) / 24.);
        case 5:
            if (x < 1.) {
                const T f = x * x;
                return S(f * (f * (0.25 - x * (1. / 12.)) - 0.5) + 0.55);
            }
            if (x < 2.) return S(x * (x * (x * (x * (x * (1. / 24.) - 0.375) + 1.25) - 1.75) + 0.625) + 0.425);
            {
                const T f = S(3. - x);
                x = f * f;
                return S(f * x * x * (1. / 120.));
            }
        case 6:
            if (x < 0.5) {
                x = x * x;
                return S(x * (x * (7. / 48. - x * (1. / 36.)) - 77. / 192.) + 5887. / 11520.0);
            }
            if (x < 1.5)
                return S(x * (x * (x * (x * (x * (x * (1. / 48.) - 7. / 48.) + 0.328125) - 35. / 288.) - 91. / 256.) - 7. / 768.) + 7861. / 15360.0);
            if (x < 2.5)
                return S(x * (x * (x * (x * (x * (7. / 60. - x * (1. / 120.)) - 0.65625) + 133. / 72.) - 2.5703125) + 1267. / 960.) + 1379. / 7680.0);
            x = S(x - 3.5);
            x = x * (x * x);
            return S(x * x * (1. / 720.));
        case 7:
            if (x < 1.) {
                const T f = x * x;
                return S(f * (f * (f * (x * (1. / 144.) - 1. / 36.) + 1. / 9.) - 1. / 3.) + 151. / 315.0);
            }
            if (x < 2.)
                return S(x * (x * (x * (x * (x * (x * (0.05 - x * (1. / 240.)) - 7. / 30.) + 0.5) - 7. / 18.) - 0.1) - 7. / 90.) + 103. / 210.0);
            if (x < 3.)
                return S(x * (x * (x * (x * (x * (x * (x * (1. / 720.) - 1. / 36.) + 7. / 30.) - 19. / 18.) + 49. / 18.) - 23. / 6.) + 217. / 90.) - 139. / 630.0);
            {
                const T f = S(4. - x);
                x = f * f * f;
                return S((x * x * f) / 5040.);
            }
        default: return (T)1 - x;      // order 1 (and the reference's `default:` branch)
    }
}

template <typename T> __device__ __forceinline__ T pp_grad(int order, T x) {
    if (order == 0) return (T)0;
    if (order == 1 || order > 7) return x < (T)0 ? (T)1 : (T)-1;
    const bool neg = x < 0;
    if (neg) x = -x;
    switch (order) {
        case 2:
            if (x < 0.5) x = S(-2. * x);
            else x = S(x - 1.5);
            break;
        case 3:
            if (x < 1.) x = S(x * (x * 1.5 - 2.));
            else {
                x = S(2. - x);
                x = S(-(x * x) * 0.5);
            }
            break;
        case 4:
            if (x < 0.5) x = S(x * (x * x - 1.25));
            else if (x < 1.5) x = S(x * (x * (x * (-2. / 3.) + 2.5) - 2.5) + 5. / 24.);
            else {
                x = S(x * 2. - 5.);
                x = S((x * x * x) / 48.);
            }
            break;
        case 5:
            if (x < 1.) x = S(x * (x * (x * (x * (-5. / 12.) + 1.)) - 1.));
            else if (x < 2.) x = S(x * (x * (x * (x * (5. / 24.) - 1.5) + 3.75) - 3.5) + 0.625);
            else {
                x = S(x - 3.);
                x = x * x;
                x = S(-(x * x) / 24.);
            }
            break;
        case 6:
            if (x < .5) {
                const T x2 = x * x;
                x = S(x * (x2 * (7. / 12.) - (x2 * x2) / 6. - 77. / 96.));
            } else if (x < 1.5) x = S(x * (x * (x * (x * (x * 0.125 - 35. / 48.) + 1.3125) - 35. / 96.) - 0.7109375) - 7.0 / 768.0);
            else if (x < 2.5) x = S(x * (x * (x * (x * (x * (-1. / 20.) + 7. / 12.) - 2.625) + 133. / 24.) - 5.140625) + 1267. / 960.);
            else {
                x = S(x * 2.);
                x = S(x - 7.);
                const T x2 = x * x;
                x = S((x2 * x2 * x) / 3840.);
            }
            break;
        default:  // 7
            if (x < 1.) {
                const T x2 = x * x;
                x = S(x * (x2 * (x2 * (x * (7. / 144.) - 1. / 6.) + 4. / 9.) - 2. / 3.));
            } else if (x < 2.) x = S(x * (x * (x * (x * (x * (x * (-7. / 240.) + 3. / 10.) - 7. / 6.) + 2.) - 7. / 6.) - 1. / 5.) - 7. / 90.);
            else if (x < 3.) x = S(x * (x * (x * (x * (x * (x * (7. / 720.) - 1. / 6.) + 7. / 6.) - 38. / 9.) + 49. / 6.) - 23. / 3.) + 217. / 90.);
            else {
                x = x - 4;
                x = x * (x * x);
                x = x * x;
                x = S(-x / 720.);
            }
            break;
    }
    return neg ? -x : x;
}

// second derivative: orders 2-4 only (the reference returns 0 for orders 5-7, interpolation_common.h:797-823)
template <typename T> __device__ __forceinline__ T pp_hess(int order, T x) {
    x = fabs(x);
    switch (order) {
        case 2: return x < 0.5 ? (T)-2. : (T)1.;
        case 3: return x < 1. ? S(x * 3. - 2.) : S(2. - x);
        case 4:
            if (x < 0.5) return S((x * x) * 3. - 1.25);
            if (x < 1.5) return S(x * (x * (-2.) + 5.) - 2.5);
            x = S(x * 2. - 5.);
            return S((x * x) / 8.);
        default: return (T)0;
    }
#undef S
}

// first node of the support (interpolation_common.h bounds0-7); the support has order + 1 nodes
template <typename T> __device__ __forceinline__ long long pp_low(int order, T x) {
    switch (order) {
        case 0: return (long long)round(x);
        case 2: return (long long)floor(x - .5);
        case 3: return (long long)floor(x - 1.);
        case 4: return (long long)floor(x - 1.5);
        case 5: return (long long)floor(x - 2.);
        case 6: return (long long)floor(x - 2.5);
        case 7: return (long long)floor(x - 3.);
        default: return (long long)floor(x);
    }
}

template <typename T> __device__ __forceinline__ void pp_add(T* p, long long off, T v, int sign) {
    if (sign == -1) unsafeAtomicAdd(p + off, -v);
    else if (sign) unsafeAtomicAdd(p + off, v);
}

// one tap of one axis: weight, first / second derivative, remapped index and sign
template <typename T> struct PPTap {
    T w, g, h;
    long long i;
    int s;
    __device__ __forceinline__ PPTap(int order, int bound, long long node, T coord, long long n, bool need_g, bool need_h) {
        const T dist = coord - (T)node;
        w = pp_weight(order, dist);
        g = need_g ? pp_grad(order, dist) : (T)0;
        h = need_h ? pp_hess(order, dist) : (T)0;
        s = gp_sign(bound, node, n);
        i = gp_index(bound, node, n);
    }
};

// The launch's flags as the GATHER kernel sees them: push / count always run in pushpull_scatter_kernel, and PULL = true
// pins the rest to "forward pull only" at compile time, so that instantiation carries no code or registers of the
// other modes (the common forward call of grid_pull / Resample).
template <bool PULL> struct PPFlags {
    int B, C, X, Y, Z, Xo, Yo, Zo, ndim, extrapolate;
    int bound[3], interp[3];
    int do_pull, do_push, do_count, do_grad, do_sgrad, trgt_k;
    __device__ __forceinline__ explicit PPFlags(const PushPullArgs& s)
        : B(s.B), C(s.C), X(s.X), Y(s.Y), Z(s.Z), Xo(s.Xo), Yo(s.Yo), Zo(s.Zo), ndim(s.ndim), extrapolate(s.extrapolate),
          do_pull(PULL ? 1 : s.do_pull), do_push(0), do_count(0), do_grad(PULL ? 0 : s.do_grad),
          do_sgrad(PULL ? 0 : s.do_sgrad), trgt_k(PULL ? 0 : s.trgt_k) {
        for (int d = 0; d < 3; ++d) { bound[d] = s.bound[d]; interp[d] = s.interp[d]; }
    }
};

#define MH_PP_UNROLL _Pragma("unroll NT <= 4 ? NT : 1")
// the z taps are evaluated inside a ROLLED z loop (the same arithmetic as a table look-up): unrolling all three tap loops
// keeps NT^3 offsets and values live and needs more than 256 registers at NT = 4
#define MH_PP_ZLOOP                                                                              \
    _Pragma("unroll 1") for (int k = 0; k < nt[2]; ++k)                                          \
        if (const PPTap<T> zt_ = PPTap<T>(zorder, zbound, zlo + k, cc[2], nn[2], need_g, need_h); true)     \
            if (const T wz_ = zt_.w, gz_ = zt_.g, hz_ = zt_.h; true)                             \
                if (const long long iz_ = zt_.i; true)                                           \
                    if (const int sz_ = zt_.s; true)
// Generic path (pushpull_cpu.cpp interpolate3d :938-1146).  NT = compile-time bound on the taps per axis (order + 1):
// with NT <= 4 the tap tables live in registers and the tap loops unroll; NT = 8 serves orders 4-7.
template <typename T, int NT, bool PULL>
__device__ __forceinline__ void pp_generic(const PushPullArgs& a_, const T* __restrict__ sp, const T* __restrict__ tp, T* __restrict__ out,
                                           T* __restrict__ gp, const T (&cc)[3], long long n, long long o, long long ovol, long long ivol,
                                           long long tsc) {
#pragma clang fp contract(off)
    const PPFlags<PULL> a(a_);
    const int K = a.ndim, C = a.C;
    const long long nn[3] = {a.X, a.Y, a.Z};
    const long long st[3] = {(long long)a.Y * a.Z, a.Z, 1};
    T wt[2][NT], gt[2][NT], ht[2][NT];
    long long it[2][NT];
    int sn[2][NT], nt[3];
    // (the reference evaluates the first derivatives only under do_grad || do_sgrad, :1005, and pushes uninitialised values
    // in the backward pass of grid_grad when only the image requires a gradient; here they are always evaluated when used)
    const bool need_g = a.do_grad || a.do_sgrad || (a.do_push && a.trgt_k > 0), need_h = a.do_grad && a.trgt_k > 0;
    const int zorder = K > 2 ? a.interp[2] : 0, zbound = K > 2 ? a.bound[2] : GB_REPLICATE;
    const long long zlo = pp_low(zorder, cc[2]);
    nt[2] = (zorder >= 0 && zorder <= 7 ? zorder : 1) + 1;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        const int order = d < K ? a.interp[d] : 0, bd = d < K ? a.bound[d] : GB_REPLICATE;
        const long long lo = pp_low(order, cc[d]);
        nt[d] = (order >= 0 && order <= 7 ? order : 1) + 1;
        MH_PP_UNROLL for (int t = 0; t < NT; ++t) {
            if (t >= nt[d]) break;
            const PPTap<T> tap(order, bd, lo + t, cc[d], nn[d], need_g, need_h);
            wt[d][t] = tap.w; gt[d][t] = tap.g; ht[d][t] = tap.h; sn[d][t] = tap.s; it[d][t] = tap.i;
        }
    }
    if (a.do_pull) {
        for (int c = 0; c < C; ++c) {
            const T* p = sp + c * ivol;
            T acc = (T)0;
            if (NT <= 4) {
                // the NT x NT gathers of one z tap are issued together, then consumed in the reference's order
                MH_PP_ZLOOP {
                    T raw[NT <= 4 ? NT : 1][NT <= 4 ? NT : 1];
                    MH_PP_UNROLL for (int j = 0; j < (NT <= 4 ? NT : 1); ++j)
                        MH_PP_UNROLL for (int i = 0; i < (NT <= 4 ? NT : 1); ++i)
                            raw[j][i] = p[(i < nt[0] ? it[0][i] : 0) * st[0] + (j < nt[1] ? it[1][j] : 0) * st[1] + iz_];
                    __builtin_amdgcn_sched_barrier(0);
                    MH_PP_UNROLL for (int j = 0; j < (NT <= 4 ? NT : 1); ++j) if (j < nt[1])
                        MH_PP_UNROLL for (int i = 0; i < (NT <= 4 ? NT : 1); ++i) if (i < nt[0]) {
                            const int s_ = sz_ * sn[1][j] * sn[0][i];
                            const T v_ = s_ == -1 ? -raw[j][i] : (s_ ? raw[j][i] : (T)0);
                            acc = acc + v_ * (wt[0][i] * wt[1][j] * wz_);
                        }
                }
            } else {
                MH_PP_ZLOOP
                    for (int j = 0; j < nt[1]; ++j)
                        for (int i = 0; i < nt[0]; ++i)
                            acc = acc + gp_get(p, it[0][i] * st[0] + it[1][j] * st[1] + iz_, sz_ * sn[1][j] * sn[0][i]) *
                                            (wt[0][i] * wt[1][j] * wz_);
            }
            out[(n * C + c) * ovol + o] = acc;
        }
    } else if (a.do_sgrad) {
        for (int c = 0; c < C; ++c) {
            const T* p = sp + c * ivol;
            T r[3] = {(T)0, (T)0, (T)0};
            MH_PP_ZLOOP
                MH_PP_UNROLL for (int j = 0; j < (NT <= 4 ? NT : nt[1]); ++j) if (NT > 4 || j < nt[1])
                    MH_PP_UNROLL for (int i = 0; i < (NT <= 4 ? NT : nt[0]); ++i) if (NT > 4 || i < nt[0]) {
                        const T v = gp_get(p, it[0][i] * st[0] + it[1][j] * st[1] + iz_, sz_ * sn[1][j] * sn[0][i]);
                        r[0] = r[0] + v * (gt[0][i] * wt[1][j] * wz_);
                        r[1] = r[1] + v * (wt[0][i] * gt[1][j] * wz_);
                        r[2] = r[2] + v * (wt[0][i] * wt[1][j] * gz_);
                    }
            for (int k = 0; k < K; ++k) out[((n * C + c) * ovol + o) * K + k] = r[k];
        }
    } else if (a.do_push) {
        MH_PP_ZLOOP
            MH_PP_UNROLL for (int j = 0; j < (NT <= 4 ? NT : nt[1]); ++j) if (NT > 4 || j < nt[1])
                MH_PP_UNROLL for (int i = 0; i < (NT <= 4 ? NT : nt[0]); ++i) if (NT > 4 || i < nt[0]) {
                    const long long off = it[0][i] * st[0] + it[1][j] * st[1] + iz_;
                    const int s = sz_ * sn[1][j] * sn[0][i];
                    for (int c = 0; c < C; ++c) {
                        T val;
                        if (a.trgt_k == 0) val = (wt[0][i] * wt[1][j] * wz_) * tp[c * tsc];
                        else {
                            val = (gt[0][i] * wt[1][j] * wz_) * tp[c * tsc];
                            if (K > 1) val = val + (wt[0][i] * gt[1][j] * wz_) * tp[c * tsc + 1];
                            if (K > 2) val = val + (wt[0][i] * wt[1][j] * gz_) * tp[c * tsc + 2];
                        }
                        pp_add(out + (n * C + c) * ivol, off, val, s);
                    }
                }
    } else if (a.do_count) {
        MH_PP_ZLOOP
            MH_PP_UNROLL for (int j = 0; j < (NT <= 4 ? NT : nt[1]); ++j) if (NT > 4 || j < nt[1])
                MH_PP_UNROLL for (int i = 0; i < (NT <= 4 ? NT : nt[0]); ++i) if (NT > 4 || i < nt[0])
                    pp_add(out + n * ivol, it[0][i] * st[0] + it[1][j] * st[1] + iz_, wt[0][i] * wt[1][j] * wz_, sz_ * sn[1][j] * sn[0][i]);
    }
    if (a.do_grad) {
        T g[3] = {(T)0, (T)0, (T)0};
        MH_PP_ZLOOP
            MH_PP_UNROLL for (int j = 0; j < (NT <= 4 ? NT : nt[1]); ++j) if (NT > 4 || j < nt[1])
                MH_PP_UNROLL for (int i = 0; i < (NT <= 4 ? NT : nt[0]); ++i) if (NT > 4 || i < nt[0]) {
                    const long long off = it[0][i] * st[0] + it[1][j] * st[1] + iz_;
                    const int s = sz_ * sn[1][j] * sn[0][i];
                    const T wx = wt[0][i], wy = wt[1][j], wz = wz_, gx = gt[0][i], gy = gt[1][j], gz = gz_;
                    if (a.trgt_k == 0) {
                        T dot = (T)0;
                        for (int c = 0; c < C; ++c) {
                            const T v = gp_get(sp + c * ivol, off, s);
                            dot = dot + (tp ? v * tp[c * tsc] : v);
                        }
                        g[0] = g[0] + (gx * wy * wz) * dot;
                        g[1] = g[1] + (wx * gy * wz) * dot;
                        g[2] = g[2] + (wx * wy * gz) * dot;
                    } else {
                        const T hx = ht[0][i], hy = ht[1][j], hz = hz_;
                        T dot0 = (T)0, dot1 = (T)0, dot2 = (T)0;
                        for (int c = 0; c < C; ++c) {
                            const T v = gp_get(sp + c * ivol, off, s);
                            dot0 = dot0 + v * tp[c * tsc];
                            if (K > 1) dot1 = dot1 + v * tp[c * tsc + 1];
                            if (K > 2) dot2 = dot2 + v * tp[c * tsc + 2];
                        }
                        // the mixed terms are the reference's (interpolate3d :1129-1131, interpolate2d :1312-1313), as written there
                        if (K == 1) g[0] = g[0] + hx * dot0;
                        else if (K == 2) {
                            g[0] = g[0] + ((hx * wy) * dot0 + (gx * gy) * dot1);
                            g[1] = g[1] + ((gx * gy) * dot0 + (wx * hy) * dot1);
                        } else {
                            g[0] = g[0] + ((hx * wy * wz) * dot0 + (gx * gy * wz) * dot1 + (gx * wy * gz) * dot2);
                            g[1] = g[1] + ((gx * gy * wz) * dot0 + (wx * hy * wz) * dot1 + (wx * gy * gz) * dot2);
                            g[2] = g[2] + ((gx * wy * gz) * dot0 + (wx * gy * gz) * dot1 + (wx * wy * hz) * dot2);
                        }
                    }
                }
        for (int k = 0; k < K; ++k) gp[k] = g[k];
    }
}
#undef MH_PP_ZLOOP
#undef MH_PP_UNROLL

// PATH / NT are compile-time so that each launch carries only its own path's registers (one kernel with all three
// paths inlined needs 256 VGPRs + scratch and runs at one wave per SIMD).
template <typename T, int PATH, int NT, bool PULL>
__global__ void __launch_bounds__(256)
pushpull_kernel(const T* __restrict__ src, const T* __restrict__ grid, const T* __restrict__ trgt, T* __restrict__ out,
                T* __restrict__ grad, PushPullArgs a_) {
#pragma clang fp contract(off)
    const PPFlags<PULL> a(a_);
    const long long ovol = (long long)a.Xo * a.Yo * a.Zo, ivol = (long long)a.X * a.Y * a.Z;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= ovol * a.B) return;
    const long long n = a.B == 1 ? 0 : idx / ovol, o = idx - n * ovol;
    const int K = a.ndim, C = a.C;
    T cc[3] = {(T)0, (T)0, (T)0};
#pragma unroll
    for (int d = 0; d < 3; ++d) if (d < K) cc[d] = grid[idx * K + d];
    const long long nn[3] = {a.X, a.Y, a.Z};
    const long long st[3] = {(long long)a.Y * a.Z, a.Z, 1};        // strides of the source / push volume
    // pull: (B,C,ovol); sgrad: (B,C,ovol,K); push: (B,C,ivol); count: (B,1,ivol); target: (B,C,ovol[,K]); grad: (B,ovol,K)
    const T* sp = src ? src + n * C * ivol : nullptr;
    const long long tk = a.trgt_k > 0 ? K : 1;
    const T* tp = trgt ? trgt + (n * C * ovol + o) * tk : nullptr;
    const long long tsc = ovol * tk;
    T* gp = grad ? grad + idx * K : nullptr;

    // ---- out of bounds and no extrapolation: zeros (check3d :796-819)
    {
        const T tiny = (T)5e-2;
        bool inb = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) if (d < K) inb = inb && cc[d] >= -tiny && cc[d] < (T)(nn[d] - 1) + tiny;
        if (!(a.extrapolate || inb)) {
            if (a.do_pull) for (int c = 0; c < C; ++c) out[(n * C + c) * ovol + o] = (T)0;
            else if (a.do_sgrad) for (int c = 0; c < C; ++c) for (int k = 0; k < K; ++k) out[((n * C + c) * ovol + o) * K + k] = (T)0;
            if (a.do_grad) for (int k = 0; k < K; ++k) gp[k] = (T)0;
            return;
        }
    }

    // =================================================================================================== nearest
    if (PATH == PP_NEAREST) {
        long long off = 0;
        int s = 1;
#pragma unroll
        for (int d = 0; d < 3; ++d)
            if (d < K) {
                const long long r = (long long)round(cc[d]);
                s *= gp_sign(a.bound[d], r, nn[d]);
                off += gp_index(a.bound[d], r, nn[d]) * st[d];
            }
        if (a.do_pull) for (int c = 0; c < C; ++c) out[(n * C + c) * ovol + o] = gp_get(sp + c * ivol, off, s);
        else if (a.do_sgrad) for (int c = 0; c < C; ++c) for (int k = 0; k < K; ++k) out[((n * C + c) * ovol + o) * K + k] = (T)0;
        else if (a.do_push && a.trgt_k == 0) for (int c = 0; c < C; ++c) pp_add(out + (n * C + c) * ivol, off, tp[c * tsc], s);
        else if (a.do_count) pp_add(out + n * ivol, off, (T)1, s);
        if (a.do_grad) for (int k = 0; k < K; ++k) gp[k] = (T)0;
        return;
    }

    // ==================================================================================================== linear
    if (PATH == PP_LINEAR) {
        long long i0[3], i1[3];
        int s0[3], s1[3];
        T d0[3], d1[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const long long f = (long long)floor(cc[d]);
            d1[d] = cc[d] - (T)f;
            d0[d] = (T)(1. - d1[d]);
            const int bd = d < K ? a.bound[d] : GB_REPLICATE;
            s1[d] = gp_sign(bd, f + 1, nn[d]);
            s0[d] = gp_sign(bd, f, nn[d]);
            i1[d] = gp_index(bd, f + 1, nn[d]);
            i0[d] = gp_index(bd, f, nn[d]);
        }
        const int ncorner = 1 << K;
        // corner k: bit 0 = x, bit 1 = y, bit 2 = z -- the order 000,100,010,110,001,101,011,111 of the reference
        long long off[8];
        int sg[8];
        T w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int bx = k & 1, by = (k >> 1) & 1, bz = k >> 2;
            off[k] = (bx ? i1[0] : i0[0]) * st[0] + (by ? i1[1] : i0[1]) * st[1] + (bz ? i1[2] : i0[2]);
            sg[k] = (bx ? s1[0] : s0[0]) * (by ? s1[1] : s0[1]) * (bz ? s1[2] : s0[2]);
            w[k] = (bx ? d1[0] : d0[0]) * (by ? d1[1] : d0[1]) * (bz ? d1[2] : d0[2]);
        }
        if (a.do_grad) {
            T g[3] = {(T)0, (T)0, (T)0};
            for (int c = 0; c < C; ++c) {
                const T* p = sp + c * ivol;
                if (a.trgt_k == 0) {
                    const T t = tp ? tp[c * tsc] : (T)1;
                    for (int k = 0; k < ncorner; ++k) {
                        const int bx = k & 1, by = (k >> 1) & 1, bz = k >> 2;
                        const T wx = bx ? d1[0] : d0[0], wy = by ? d1[1] : d0[1], wz = bz ? d1[2] : d0[2];
                        T v = gp_get(p, off[k], sg[k]);
                        if (tp) v = v * t;
                        const T cx = wy * wz * v, cy = wx * wz * v, cz = wx * wy * v;
                        g[0] = bx ? g[0] + cx : g[0] - cx;
                        g[1] = by ? g[1] + cy : g[1] - cy;
                        g[2] = bz ? g[2] + cz : g[2] - cz;
                    }
                } else {
                    const T t0 = tp[c * tsc], t1 = K > 1 ? tp[c * tsc + 1] : (T)0, t2 = K > 2 ? tp[c * tsc + 2] : (T)0;
                    for (int k = 0; k < ncorner; ++k) {
                        const int bx = k & 1, by = (k >> 1) & 1, bz = k >> 2;
                        const T wx = bx ? d1[0] : d0[0], wy = by ? d1[1] : d0[1], wz = bz ? d1[2] : d0[2];
                        const bool pxy = bx == by, pxz = bx == bz, pyz = by == bz;      // sign products e_x e_y, ...
                        const T v = gp_get(p, off[k], sg[k]);
                        g[0] = g[0] + ((pxy ? wz : -wz) * t1 + (pxz ? wy : -wy) * t2) * v;
                        g[1] = g[1] + ((pxy ? wz : -wz) * t0 + (pyz ? wx : -wx) * t2) * v;
                        g[2] = g[2] + ((pxz ? wy : -wy) * t0 + (pyz ? wx : -wx) * t1) * v;
                    }
                }
            }
            if (a.trgt_k != 0 && K == 1) g[0] = (T)0;       // interpolate1d_linear :2001-2003 leaves the zero-filled gradient
            for (int k = 0; k < K; ++k) gp[k] = g[k];
        }
        if (a.do_pull) {
            for (int c = 0; c < C; ++c) {
                const T* p = sp + c * ivol;
                T v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = p[off[k]];      // all eight in flight (padded axes: index 0, loadable)
                __builtin_amdgcn_sched_barrier(0);                 // keep the loads together: the scheduler otherwise pairs load + use
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = sg[k] == -1 ? -v[k] : (sg[k] ? v[k] : (T)0);
                T acc = v[0] * w[0];
#pragma unroll
                for (int k = 1; k < 8; ++k) if (k < ncorner) acc = acc + v[k] * w[k];
                out[(n * C + c) * ovol + o] = acc;
            }
        } else if (a.do_sgrad) {
            for (int c = 0; c < C; ++c) {
                const T* p = sp + c * ivol;
                T r[3] = {(T)0, (T)0, (T)0};
                for (int k = 0; k < ncorner; ++k) {
                    const int bx = k & 1, by = (k >> 1) & 1, bz = k >> 2;
                    const T wx = bx ? d1[0] : d0[0], wy = by ? d1[1] : d0[1], wz = bz ? d1[2] : d0[2];
                    const T v = gp_get(p, off[k], sg[k]);
                    const T tx = (bx ? wy : -wy) * wz * v, ty = (by ? wx : -wx) * wz * v, tz = (bz ? wx : -wx) * wy * v;
                    r[0] = k ? r[0] + tx : tx;
                    r[1] = k ? r[1] + ty : ty;
                    r[2] = k ? r[2] + tz : tz;
                }
                for (int k = 0; k < K; ++k) out[((n * C + c) * ovol + o) * K + k] = r[k];
            }
        } else if (a.do_push) {
            for (int c = 0; c < C; ++c) {
                T* q = out + (n * C + c) * ivol;
                if (a.trgt_k == 0) {
                    const T t = tp[c * tsc];
                    for (int k = 0; k < ncorner; ++k) pp_add(q, off[k], w[k] * t, sg[k]);
                } else {
                    const T t0 = tp[c * tsc], t1 = K > 1 ? tp[c * tsc + 1] : (T)0, t2 = K > 2 ? tp[c * tsc + 2] : (T)0;
                    for (int k = 0; k < ncorner; ++k) {
                        const int bx = k & 1, by = (k >> 1) & 1, bz = k >> 2;
                        const T wx = bx ? d1[0] : d0[0], wy = by ? d1[1] : d0[1], wz = bz ? d1[2] : d0[2];
                        T val = (bx ? wy : -wy) * wz * t0;
                        if (K > 1) val = val + (by ? wx : -wx) * wz * t1;
                        if (K > 2) val = val + (bz ? wx : -wx) * wy * t2;
                        pp_add(q, off[k], val, sg[k]);
                    }
                }
            }
        } else if (a.do_count) {
            for (int k = 0; k < ncorner; ++k) pp_add(out + n * ivol, off[k], w[k], sg[k]);
        }
        return;
    }

    // =================================================================================================== generic
    if (PATH == PP_GENERIC) pp_generic<T, NT, PULL>(a_, sp, tp, out, gp, cc, n, o, ovol, ivol, tsc);
}

// ---------------------------------------------------------------------------------------------------------------------
// Scatter modes (push, count) with LDS privatisation.  A workgroup owns a small 3-D tile of TARGET voxels; the nodes their
// taps touch form a compact box of the splatted volume for any smooth deformation, so the contributions are first added
// into an LDS copy of that box (ds_add, no L2 round trip) and the box is flushed with ONE global atomic per non-zero
// element: 8 taps x 256 voxels = 2048 global atomics become <= (4+1)(4+1)(16+1) = 425 for trilinear splatting, 16384
// become <= 1280 for cubic.  Taps that leave the box after the boundary remap (wrap / reflect far away), and whole tiles
// whose box exceeds the LDS budget (wild coordinates), fall back to global atomics -- same values, summation order is
// free in every case.  Per-tap values follow the reference's expressions (linear: weights d0 = 1 - d1, d1 and derivative
// weights -1 / +1, which give the products of interpolate3d_trilinear :1698-1745 term by term).
constexpr int PP_WIN = 6144;      // LDS box capacity (elements of T)

template <typename T, int PATH, int NT>
__global__ void __launch_bounds__(256)
pushpull_scatter_kernel(const T* __restrict__ grid, const T* __restrict__ trgt, T* __restrict__ out, PushPullArgs a, int tx, int ty, int tz) {
#pragma clang fp contract(off)
    __shared__ T win[PP_WIN];
    __shared__ int wlo[3], whi[3];
    const int tid = threadIdx.x;
    const int K = a.ndim, C = a.C;
    const long long ovol = (long long)a.Xo * a.Yo * a.Zo, ivol = (long long)a.X * a.Y * a.Z;
    const long long nn[3] = {a.X, a.Y, a.Z};
    const long long st[3] = {(long long)a.Y * a.Z, a.Z, 1};
    // tile decode: lanes run along the last axis
    const int nbz = (a.Zo + tz - 1) / tz, nby = (a.Yo + ty - 1) / ty, nbx = (a.Xo + tx - 1) / tx;
    long long bid = blockIdx.x;
    const int bz = (int)(bid % nbz); bid /= nbz;
    const int by = (int)(bid % nby); bid /= nby;
    const int bx = (int)(bid % nbx);
    const long long n = bid / nbx;
    const int lz = tid % tz, ly = (tid / tz) % ty, lx = tid / (tz * ty);
    const int vx = bx * tx + lx, vy = by * ty + ly, vz = bz * tz + lz;
    bool active = lx < tx && vx < a.Xo && vy < a.Yo && vz < a.Zo;
    const long long o = active ? ((long long)vx * a.Yo + vy) * a.Zo + vz : 0;
    const long long idx = n * ovol + o;
    T cc[3] = {(T)0, (T)0, (T)0};
    if (active) {
#pragma unroll
        for (int d = 0; d < 3; ++d) if (d < K) cc[d] = grid[idx * K + d];
        const T tiny = (T)5e-2;
        bool inb = true;
#pragma unroll
        for (int d = 0; d < 3; ++d) if (d < K) inb = inb && cc[d] >= -tiny && cc[d] < (T)(nn[d] - 1) + tiny;
        active = a.extrapolate || inb;
    }
    const long long tk = a.trgt_k > 0 ? K : 1;
    const T* tp = trgt ? trgt + (n * C * ovol + o) * tk : nullptr;
    const long long tsc = ovol * tk;

    // per-axis tap tables
    T wt[3][NT], gt[3][NT];
    long long it[3][NT], lo[3];
    int sn[3][NT], nt[3];
    const bool need_g = a.trgt_k > 0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const int bd = d < K ? a.bound[d] : GB_REPLICATE;
        if (PATH == PP_NEAREST) {
            lo[d] = (long long)round(cc[d]);
            nt[d] = 1;
            wt[d][0] = (T)1; gt[d][0] = (T)0;
        } else if (PATH == PP_LINEAR) {
            lo[d] = (long long)floor(cc[d]);
            nt[d] = d < K ? 2 : 1;
            const T d1 = cc[d] - (T)lo[d];
            wt[d][0] = (T)(1. - d1); gt[d][0] = (T)-1;
            if (NT > 1) { wt[d][1] = d1; gt[d][1] = (T)1; }
        } else {
            const int order = d < K ? a.interp[d] : 0;
            lo[d] = pp_low(order, cc[d]);
            nt[d] = (order >= 0 && order <= 7 ? order : 1) + 1;
            for (int t = 0; t < NT; ++t) {
                if (t >= nt[d]) break;
                const T dist = cc[d] - (T)(lo[d] + t);
                wt[d][t] = pp_weight(order, dist);
                gt[d][t] = need_g ? pp_grad(order, dist) : (T)0;
            }
        }
        for (int t = 0; t < NT; ++t) {
            if (t >= nt[d]) break;
            sn[d][t] = gp_sign(bd, lo[d] + t, nn[d]);
            it[d][t] = gp_index(bd, lo[d] + t, nn[d]);
        }
    }

    // the tile's box: union of the (unremapped, clamped) node ranges of its active voxels
    if (tid < 3) { wlo[tid] = 0x7fffffff; whi[tid] = -0x7fffffff; }
    __syncthreads();
    {   // wave-reduce first: 256 threads x 6 LDS atomics on six addresses serialise (measured: 10 us per workgroup)
        int mn[3], mx[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const long long a0 = lo[d] < 0 ? 0 : lo[d], a1 = lo[d] + nt[d] - 1 >= nn[d] ? nn[d] - 1 : lo[d] + nt[d] - 1;
            const bool ok = active && a0 <= a1;
            mn[d] = ok ? (int)a0 : 0x7fffffff;
            mx[d] = ok ? (int)a1 : -0x7fffffff;
#pragma unroll
            for (int sft = 32; sft >= 1; sft >>= 1) {
                mn[d] = min(mn[d], __shfl_xor(mn[d], sft));
                mx[d] = max(mx[d], __shfl_xor(mx[d], sft));
            }
        }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int d = 0; d < 3; ++d) { atomicMin(&wlo[d], mn[d]); atomicMax(&whi[d], mx[d]); }
        }
    }
    __syncthreads();
    const int e0 = whi[0] - wlo[0] + 1, e1 = whi[1] - wlo[1] + 1, e2 = whi[2] - wlo[2] + 1;
    // one tap per voxel (nearest) gains nothing from the detour through LDS: straight to L2
    const bool boxed = PATH != PP_NEAREST && e0 > 0 && e1 > 0 && e2 > 0 && (long long)e0 * e1 * e2 <= PP_WIN;
    const int vol = boxed ? e0 * e1 * e2 : 0;
    const int b0 = wlo[0], b1 = wlo[1], b2 = wlo[2];
    for (int e = tid; e < vol; e += 256) win[e] = (T)0;
    __syncthreads();

    const int nch = a.do_count ? 1 : C;
    for (int c = 0; c < nch; ++c) {
        T* q = out + (n * nch + c) * ivol;
        if (active) {
            T t0 = (T)1, t1 = (T)0, t2 = (T)0;
            if (!a.do_count) {
                t0 = tp[c * tsc];
                if (a.trgt_k > 0) { t1 = K > 1 ? tp[c * tsc + 1] : (T)0; t2 = K > 2 ? tp[c * tsc + 2] : (T)0; }
            }
            for (int k = 0; k < nt[2]; ++k)
                for (int j = 0; j < nt[1]; ++j)
                    for (int i = 0; i < nt[0]; ++i) {
                        const int s = sn[2][k] * sn[1][j] * sn[0][i];
                        if (!s) continue;
                        T val;
                        if (a.do_count) val = wt[0][i] * wt[1][j] * wt[2][k];
                        else if (a.trgt_k == 0) val = PATH == PP_NEAREST ? t0 : (wt[0][i] * wt[1][j] * wt[2][k]) * t0;
                        else {
                            val = (gt[0][i] * wt[1][j] * wt[2][k]) * t0;
                            if (K > 1) val = val + (wt[0][i] * gt[1][j] * wt[2][k]) * t1;
                            if (K > 2) val = val + (wt[0][i] * wt[1][j] * gt[2][k]) * t2;
                        }
                        if (s < 0) val = -val;
                        const long long ix = it[0][i], iy = it[1][j], iz = it[2][k];
                        const long long r0 = ix - b0, r1 = iy - b1, r2 = iz - b2;
                        if (boxed && r0 >= 0 && r0 < e0 && r1 >= 0 && r1 < e1 && r2 >= 0 && r2 < e2)
                            unsafeAtomicAdd(&win[(r0 * e1 + r1) * e2 + r2], val);
                        else
                            unsafeAtomicAdd(q + ix * st[0] + iy * st[1] + iz, val);
                    }
        }
        __syncthreads();
        for (int e = tid; e < vol; e += 256) {
            const T v = win[e];
            if (v != (T)0) {
                const int r2 = e % e2, r1 = (e / e2) % e1, r0 = e / (e2 * e1);
                unsafeAtomicAdd(q + (long long)(b0 + r0) * st[0] + (long long)(b1 + r1) * st[1] + (b2 + r2), v);
                win[e] = (T)0;
            }
        }
        __syncthreads();
    }
}

}  // namespace mh
