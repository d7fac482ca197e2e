// C ABI of libmonai_amd.so (declared in include/monai_amd.h): argument checks, kernel-configuration
// choice and launches.  Everything is stream ordered; nothing allocates or synchronises.
#include <hip/hip_runtime.h>

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kernels/common.h"
#include "kernels/attention.h"
#include "kernels/conv3d_mfma.h"
#include "kernels/conv3d_wino2p.h"
#include "kernels/conv3d_h2.h"
#include "kernels/conv3d_wino_h2.h"
#include "kernels/upconv_h2.h"
#include "kernels/conv3d_s2_h2.h"
#include "kernels/deconv_h2.h"
#include "kernels/conv3d_vol_h2.h"
#include "kernels/conv3d_c1.h"
#include "kernels/dense.h"
#include "kernels/gaussian.h"
#include "kernels/grid_pull.h"
#include "kernels/pushpull.h"
#include "kernels/post.h"
#include "kernels/preproc.h"
#include "kernels/nn_simple.h"
#include "kernels/conv1x1_h2.h"
#include "kernels/resample.h"
#include "kernels/sliding.h"

using namespace mh;

#define MH_BLEND_G 2       /* windows requested per batch by the regular-grid blend (measured: profiles/r02_blend_variants_v1.json) */
#define MH_BLEND_NT 0      /* non-temporal logit loads / output stores: slower here (half-rows of a cache line are shared by two requests) */

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
static int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MH_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return MH_OK;
}
static inline unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }
static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline bool fits_i32(long long n) { return n < 2147483647LL; }
static bool dense_ok(const mh_tensor5* t) {
    return t && t->data && t->N > 0 && t->C > 0 && t->D > 0 && t->H > 0 && t->W > 0 &&
           t->n_stride >= (int64_t)t->C * t->D * t->H * t->W;
}

// All mh_* functions below have C linkage through their declarations in include/monai_amd.h.

int mh_version(void) { return 100; }
const char* mh_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------ inferer
static int fill_grid(WindowGrid& g, const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx, int nx) {
    if (!sz || !sy || !sx || nz < 1 || ny < 1 || nx < 1) return fail(MH_ERR_ARG, "window grid: null/empty start list");
    if (nz > MAX_AXIS_WINDOWS || ny > MAX_AXIS_WINDOWS || nx > MAX_AXIS_WINDOWS)
        return fail(MH_ERR_UNSUPPORTED, "window grid: more than %d windows on one axis", MAX_AXIS_WINDOWS);
    g.nz = nz; g.ny = ny; g.nx = nx;
    memset(g.sz, 0, sizeof(g.sz)); memset(g.sy, 0, sizeof(g.sy)); memset(g.sx, 0, sizeof(g.sx));
    memcpy(g.sz, sz, sizeof(int) * nz); memcpy(g.sy, sy, sizeof(int) * ny); memcpy(g.sx, sx, sizeof(int) * nx);
    for (int i = 1; i < nz; ++i) if (sz[i] < sz[i - 1]) return fail(MH_ERR_ARG, "window starts must ascend (z)");
    for (int i = 1; i < ny; ++i) if (sy[i] < sy[i - 1]) return fail(MH_ERR_ARG, "window starts must ascend (y)");
    for (int i = 1; i < nx; ++i) if (sx[i] < sx[i - 1]) return fail(MH_ERR_ARG, "window starts must ascend (x)");
    return MH_OK;
}
static bool all_mult4(const int32_t* s, int n) {
    for (int i = 0; i < n; ++i) if (s[i] % 4) return false;
    return true;
}
// dense_patch_slices' form: start(i) = i * step (i < n - 1), start(n - 1) = last <= (n - 1) * step; `extent` bounds the coordinates
static bool regular_axis(const int32_t* s, int n, int extent, AxisWin& a) {
    if (n < 1 || s[0] != 0) return false;
    a.n = n; a.step = n > 1 ? s[1] : 1; a.last = s[n - 1]; a.magic = 0;
    if (n == 1) { a.step = 1; a.last = 0; return true; }
    if (n == 2) { a.step = s[1] > 0 ? s[1] : 1; a.last = s[1]; }
    if (a.step < 1) return false;
    for (int i = 1; i < n - 1; ++i) if (s[i] != i * a.step) return false;
    if (a.last > (n - 1) * a.step || (n > 2 && a.last <= (n - 2) * a.step)) return false;
    if (a.step > 1 && (long long)extent * a.step < (1LL << 32)) a.magic = (unsigned)((1ULL << 32) / (unsigned)a.step) + 1u;
    return true;
}
static bool regular_grid(RegGrid& g, const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx, int nx, int D, int H, int W) {
    return sz && sy && sx && regular_axis(sz, nz, D, g.z) && regular_axis(sy, ny, H, g.y) && regular_axis(sx, nx, W, g.x);
}
// Development knobs (A/B measurements under tools/): only a library built with -DMH_DEV_KNOBS (python -m monai_amd.build --dev ->
// libmonai_amd_dev.so, never loaded by the product) reads them from the environment.  The shipped library has no hidden state:
// every knob is its measured default and what a call computes depends on its arguments alone.
#ifdef MH_DEV_KNOBS
static const char* knob_str(const char* name) { return getenv(name); }
#else
static inline const char* knob_str(const char*) { return nullptr; }
#endif
static int knob_int(const char* name, int dflt) {
    const char* e = knob_str(name);
    return e && *e ? atoi(e) : dflt;
}

int mh_window_extract_f32(const float* vol, int C, int D, int H, int W, const int32_t* sz, int nz, const int32_t* sy,
                          int ny, const int32_t* sx, int nx, int w0, int nwin, int rd, int rh, int rw, float* out,
                          void* stream) {
    if (!vol || !out || C < 1 || nwin < 1 || w0 < 0) return fail(MH_ERR_ARG, "window_extract: bad argument");
    if (!sz || !sy || !sx || nz < 1 || ny < 1 || nx < 1) return fail(MH_ERR_ARG, "window grid: null/empty start list");
    if ((long long)w0 + nwin > (long long)nz * ny * nx) return fail(MH_ERR_ARG, "window_extract: window range out of grid");
    if (sz[nz - 1] + rd > D || sy[ny - 1] + rh > H || sx[nx - 1] + rw > W || sz[0] < 0 || sy[0] < 0 || sx[0] < 0)
        return fail(MH_ERR_ARG, "window_extract: window leaves the volume");
    const bool v4 = rw % 4 == 0 && W % 4 == 0 && all_mult4(sx, nx) && aligned(vol, 16) && aligned(out, 16);
    const long long total = (long long)nwin * C * rd * rh * (v4 ? rw / 4 : rw);
    if (total > 0x7fffffffLL * 256) return fail(MH_ERR_UNSUPPORTED, "window_extract: problem too large for one launch");
    RegGrid rg;
    if (regular_grid(rg, sz, nz, sy, ny, sx, nx, D, H, W)) {      // dense_patch_slices: closed-form starts, any number of windows per axis
        if (v4)
            hipLaunchKernelGGL((window_extract_reg_kernel<4>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, vol, C, D, H, W, rg,
                               w0, nwin, rd, rh, rw, out);
        else
            hipLaunchKernelGGL((window_extract_reg_kernel<1>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, vol, C, D, H, W, rg,
                               w0, nwin, rd, rh, rw, out);
        return launched("window_extract");
    }
    WindowGrid g;
    if (int e = fill_grid(g, sz, nz, sy, ny, sx, nx)) return e;
    if (v4)
        hipLaunchKernelGGL((window_extract_kernel<4>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, vol, C, D,
                           H, W, g, w0, nwin, rd, rh, rw, out);
    else
        hipLaunchKernelGGL((window_extract_kernel<1>), dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, vol, C, D,
                           H, W, g, w0, nwin, rd, rh, rw, out);
    return launched("window_extract");
}

template <int VEC>
static void launch_blend(int kt, unsigned nb, hipStream_t s, const float* logits, const float* imp, float* out, int K, int k0,
                         int D, int H, int W, int rd, int rh, int rw, const WindowGrid& g, int premul, long long wstride) {
#define MH_BLEND_CASE(KT)                                                                                              \
    case KT:                                                                                                           \
        hipLaunchKernelGGL((sw_blend_kernel<KT, VEC>), dim3(nb), dim3(256), 0, s, logits, imp, out, K, k0, D, H, W, rd, \
                           rh, rw, g, premul, wstride);                                                                \
        break;
    switch (kt) {
        MH_BLEND_CASE(1) MH_BLEND_CASE(2) MH_BLEND_CASE(3) MH_BLEND_CASE(4)
        MH_BLEND_CASE(5) MH_BLEND_CASE(6) MH_BLEND_CASE(7) MH_BLEND_CASE(8)
    }
#undef MH_BLEND_CASE
}

int mh_patch_accumulate_f32(float* values, uint8_t* counts, const float* patch, int NC, int D, int H, int W, int pd, int ph, int pw,
                            int z0, int y0, int x0, void* stream) {
    if (!values || !counts || !patch) return fail(MH_ERR_ARG, "patch_accumulate: null pointer");
    if (NC < 1 || D < 1 || H < 1 || W < 1 || pd < 1 || ph < 1 || pw < 1) return fail(MH_ERR_ARG, "patch_accumulate: bad shape");
    if (z0 < 0 || y0 < 0 || x0 < 0 || z0 + pd > D || y0 + ph > H || x0 + pw > W)
        return fail(MH_ERR_ARG, "patch_accumulate: the patch (%d,%d,%d)+(%d,%d,%d) leaves the merged volume (%d,%d,%d)", z0, y0, x0, pd, ph, pw, D, H, W);
    const long long total = (long long)NC * pd * ph * pw;
    if (total > 0x7fffffffLL * 256) return fail(MH_ERR_UNSUPPORTED, "patch_accumulate: problem too large for one launch");
    hipLaunchKernelGGL(patch_accumulate_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, values, counts, patch, NC, D, H, W, pd, ph, pw, z0, y0, x0);
    return launched("patch_accumulate");
}

int mh_patch_accumulate_batch_f32(float* values, uint8_t* counts, const float* patches, int npatch, const int32_t* loc, int NC, int D, int H, int W, int pd,
                                  int ph, int pw, void* stream) {
    if (!values || !counts || !patches || !loc) return fail(MH_ERR_ARG, "patch_accumulate_batch: null pointer");
    if (NC < 1 || D < 1 || H < 1 || W < 1 || pd < 1 || ph < 1 || pw < 1 || npatch < 1) return fail(MH_ERR_ARG, "patch_accumulate_batch: bad shape");
    for (int p0 = 0; p0 < npatch; p0 += PATCH_BATCH_MAX) {          // more patches than one launch's argument block holds: consecutive launches keep the order
        PatchBatch b;
        b.n = npatch - p0 < PATCH_BATCH_MAX ? npatch - p0 : PATCH_BATCH_MAX;
        int lo[3] = {D, H, W}, hi[3] = {0, 0, 0};
        for (int i = 0; i < b.n; ++i) {
            const int32_t* l = loc + 3 * (p0 + i);
            if (l[0] < 0 || l[1] < 0 || l[2] < 0 || l[0] + pd > D || l[1] + ph > H || l[2] + pw > W)
                return fail(MH_ERR_ARG, "patch_accumulate: the patch (%d,%d,%d)+(%d,%d,%d) leaves the merged volume (%d,%d,%d)", l[0], l[1], l[2], pd, ph, pw, D, H, W);
            b.z[i] = l[0]; b.y[i] = l[1]; b.x[i] = l[2];
            const int e[3] = {l[0] + pd, l[1] + ph, l[2] + pw};
            for (int a = 0; a < 3; ++a) { lo[a] = l[a] < lo[a] ? l[a] : lo[a]; hi[a] = e[a] > hi[a] ? e[a] : hi[a]; }
        }
        const long long total = (long long)NC * (hi[0] - lo[0]) * (hi[1] - lo[1]) * (hi[2] - lo[2]);
        if (total > 0x7fffffffLL * 256) return fail(MH_ERR_UNSUPPORTED, "patch_accumulate_batch: problem too large for one launch");
        hipLaunchKernelGGL(patch_accumulate_batch_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, values, counts,
                           patches + (long long)p0 * NC * pd * ph * pw, NC, D, H, W, pd, ph, pw, b, lo[0], lo[1], lo[2], hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]);
    }
    return launched("patch_accumulate_batch");
}

int mh_avg_finalize_f32(float* values, const uint8_t* counts, int64_t n, void* stream) {
    if (!values || !counts || n < 1) return fail(MH_ERR_ARG, "avg_finalize: bad argument");
    if (n > 0x7fffffffLL * 256) return fail(MH_ERR_UNSUPPORTED, "avg_finalize: problem too large for one launch");
    hipLaunchKernelGGL(avg_finalize_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, values, counts, (long long)n);
    return launched("avg_finalize");
}

// Blend on a regular grid: KT channels per launch (or all K inside the thread for the argmax epilogue), G windows requested
// per batch.  MONAI_AMD_BLEND_G / MONAI_AMD_BLEND_NT are development knobs (profiles/r02_blend_variants.txt).
template <bool ARGMAX>
static int launch_blend_reg(int kt, bool v4, int G, bool nt, unsigned nb, hipStream_t s, const float* logits, const float* imp, void* out,
                            int K, int k0, int D, int H, int W, int rd, int rh, int rw, const RegGrid& g, int premul, int out_u8, long long wstride) {
#define MH_BR(KT, VEC, GG, NTT)                                                                                              \
    hipLaunchKernelGGL((sw_blend_reg_kernel<KT, VEC, GG, NTT, ARGMAX>), dim3(nb), dim3(256), 0, s, logits, imp, out, K, k0, D, H, \
                       W, rd, rh, rw, g, premul, out_u8, wstride)
#define MH_BR_CASE(KT)                                                                      \
    case KT:                                                                                \
        if (!v4) MH_BR(KT, 1, 1, false);                                                    \
        else if (nt) MH_BR(KT, 4, MH_BLEND_G, true);                                        \
        else MH_BR(KT, 4, MH_BLEND_G, false);                                               \
        break;
    if (v4 && kt == 5 && !ARGMAX && G != MH_BLEND_G) {       // the benchmark shape carries every batch size
        switch (G) {
            case 1: if (nt) MH_BR(5, 4, 1, true); else MH_BR(5, 4, 1, false); return MH_OK;
            case 2: if (nt) MH_BR(5, 4, 2, true); else MH_BR(5, 4, 2, false); return MH_OK;
            case 4: if (nt) MH_BR(5, 4, 4, true); else MH_BR(5, 4, 4, false); return MH_OK;
            case 8: if (nt) MH_BR(5, 4, 8, true); else MH_BR(5, 4, 8, false); return MH_OK;
            default: break;
        }
    }
    switch (kt) {
        MH_BR_CASE(1) MH_BR_CASE(2) MH_BR_CASE(3) MH_BR_CASE(4)
        MH_BR_CASE(5) MH_BR_CASE(6) MH_BR_CASE(7) MH_BR_CASE(8)
        default: return fail(MH_ERR_ARG, "sw_blend: bad channel chunk %d", kt);
    }
#undef MH_BR_CASE
#undef MH_BR
    return MH_OK;
}

static int blend_checks(const char* what, const float* logits, const float* imp, const void* out, int K, int D, int H, int W, int rd, int rh, int rw,
                        const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx, int nx, int64_t& wstride) {
    if (!logits || !imp || !out || K < 1 || D < 1 || H < 1 || W < 1) return fail(MH_ERR_ARG, "%s: bad argument", what);
    const int64_t dense = (int64_t)K * rd * rh * rw;
    if (wstride == 0) wstride = dense;
    if (wstride < dense) return fail(MH_ERR_ARG, "%s: window stride %lld is smaller than one window's logits (%lld floats)", what, (long long)wstride, (long long)dense);
    if (!sz || !sy || !sx || nz < 1 || ny < 1 || nx < 1) return fail(MH_ERR_ARG, "window grid: null/empty start list");
    // full coverage: first window at 0, last ends at the image end, no gaps
    const int32_t* ss[3] = {sz, sy, sx};
    const int nn[3] = {nz, ny, nx}, rr[3] = {rd, rh, rw}, dd[3] = {D, H, W};
    for (int a = 0; a < 3; ++a) {
        if (ss[a][0] != 0 || ss[a][nn[a] - 1] + rr[a] != dd[a]) return fail(MH_ERR_ARG, "%s: windows do not span axis %d", what, a);
        for (int i = 1; i < nn[a]; ++i) {
            if (ss[a][i] < ss[a][i - 1]) return fail(MH_ERR_ARG, "window starts must ascend (axis %d)", a);
            if (ss[a][i] > ss[a][i - 1] + rr[a]) return fail(MH_ERR_ARG, "%s: uncovered gap on axis %d", what, a);
        }
    }
    if ((long long)D * H * W > 0x7fffffffLL * 256) return fail(MH_ERR_UNSUPPORTED, "%s: problem too large for one launch", what);
    return MH_OK;
}

int mh_sw_blend_f32(const float* logits, int64_t window_stride, const float* imp, float* out, int K, int D, int H, int W, int rd, int rh, int rw,
                    const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx, int nx, int premultiplied, void* stream) {
    if (int e = blend_checks("sw_blend", logits, imp, out, K, D, H, W, rd, rh, rw, sz, nz, sy, ny, sx, nx, window_stride)) return e;
    const bool v4 = W % 4 == 0 && rw % 4 == 0 && window_stride % 4 == 0 && all_mult4(sx, nx) && aligned(logits, 16) && aligned(imp, 16) && aligned(out, 16);
    const long long total = (long long)D * H * (v4 ? W / 4 : W);
    RegGrid rg;
    if (regular_grid(rg, sz, nz, sy, ny, sx, nx, D, H, W) && knob_int("MONAI_AMD_BLEND_LEGACY", 0) == 0) {
        const int G = knob_int("MONAI_AMD_BLEND_G", MH_BLEND_G);
        const bool nt = knob_int("MONAI_AMD_BLEND_NT", MH_BLEND_NT) != 0;
        for (int k0 = 0; k0 < K; k0 += 8) {
            const int kt = K - k0 < 8 ? K - k0 : 8;
            if (int e = launch_blend_reg<false>(kt, v4, G, nt, blocks_for(total), (hipStream_t)stream, logits, imp, out, K, k0, D, H, W, rd, rh, rw, rg,
                                                premultiplied ? 1 : 0, 0, window_stride))
                return e;
        }
        return launched("sw_blend");
    }
    WindowGrid g;      // irregular start lists (e.g. a multi-resolution output whose scaled starts round unevenly): tables by value
    if (int e = fill_grid(g, sz, nz, sy, ny, sx, nx)) return e;
    for (int k0 = 0; k0 < K; k0 += 8) {
        const int kt = K - k0 < 8 ? K - k0 : 8;
        if (v4) launch_blend<4>(kt, blocks_for(total), (hipStream_t)stream, logits, imp, out, K, k0, D, H, W, rd, rh, rw, g, premultiplied ? 1 : 0, window_stride);
        else launch_blend<1>(kt, blocks_for(total), (hipStream_t)stream, logits, imp, out, K, k0, D, H, W, rd, rh, rw, g, premultiplied ? 1 : 0, window_stride);
    }
    return launched("sw_blend");
}

// the blend in the summation order of the reference's buffered schedule (kernels/sliding.h: sw_blend_buffered_kernel)
int mh_sw_blend_buffered_f32(const float* logits, int64_t window_stride, const float* imp, float* out, int K, int D, int H, int W, int rd, int rh, int rw,
                             const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx, int nx, int buffer_axis, int buffer_steps, int premultiplied, void* stream) {
    if (int e = blend_checks("sw_blend_buffered", logits, imp, out, K, D, H, W, rd, rh, rw, sz, nz, sy, ny, sx, nx, window_stride)) return e;
    if (buffer_axis < 0 || buffer_axis > 2 || buffer_steps < 1) return fail(MH_ERR_ARG, "sw_blend_buffered: buffer_axis in 0..2 and buffer_steps >= 1 (got %d, %d)", buffer_axis, buffer_steps);
    const bool v4 = W % 4 == 0 && rw % 4 == 0 && all_mult4(sx, nx);
    const long long total = (long long)D * H * (v4 ? W / 4 : W);
    WindowGrid g;
    if (int e = fill_grid(g, sz, nz, sy, ny, sx, nx)) return e;
    hipStream_t s = (hipStream_t)stream;
    for (int k0 = 0; k0 < K; k0 += 4) {
        const int kt = K - k0 < 4 ? K - k0 : 4;
#define MH_BB(KT_, V_)                                                                                                                                          \
    {                                                                                                                                                          \
        if (premultiplied) hipLaunchKernelGGL((sw_blend_buffered_kernel<KT_, V_, true>), dim3(blocks_for(total)), dim3(256), 0, s, logits, imp, out, K, k0, D, H, W, rd, rh, rw, g, buffer_axis, buffer_steps, (long long)window_stride); \
        else hipLaunchKernelGGL((sw_blend_buffered_kernel<KT_, V_, false>), dim3(blocks_for(total)), dim3(256), 0, s, logits, imp, out, K, k0, D, H, W, rd, rh, rw, g, buffer_axis, buffer_steps, (long long)window_stride);           \
    }
        if (v4) { if (kt == 4) MH_BB(4, 4) else if (kt == 3) MH_BB(3, 4) else if (kt == 2) MH_BB(2, 4) else MH_BB(1, 4) }
        else { if (kt == 4) MH_BB(4, 1) else if (kt == 3) MH_BB(3, 1) else if (kt == 2) MH_BB(2, 1) else MH_BB(1, 1) }
#undef MH_BB
    }
    return launched("sw_blend_buffered");
}

#ifndef MH_BLEND_IDX32_LIMIT
#define MH_BLEND_IDX32_LIMIT (1LL << 31)      // floats; the SIMT-emulator build lowers it so that its small cases run both index widths (tests/emu/build_emu.py)
#endif
// Mosaic logits layout (kernels/sliding.h): the residue classes per axis are 2^log2m; `class_base` holds the float offset of each class array,
// indexed (cz * 5 + cy) * 5 + cx.  mh_sw_mosaic_class_counts gives the number of windows per class of an axis (what sizes the arrays).
int mh_sw_mosaic_class_counts(int n, int log2m, int32_t* counts5) {
    if (n < 1 || log2m < 0 || log2m > 2 || !counts5) return fail(MH_ERR_ARG, "sw_mosaic_class_counts: bad argument (1, 2 or 4 residue classes)");
    MosaicAxis a;
    mosaic_axis_fill(a, n, log2m);
    for (int c = 0; c < MOSAIC_MAX_CLASSES; ++c) counts5[c] = a.cnt[c];
    return MH_OK;
}

int mh_sw_blend_mosaic_f32(const float* logits, const int64_t* class_base, int log2m_z, int log2m_y, int log2m_x, const float* imp, int imp_factored, float* out, int K,
                           int D, int H, int W, int rd, int rh, int rw, const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx, int nx,
                           void* stream) {
    int64_t dense = 0;
    if (int e = blend_checks("sw_blend_mosaic", logits, imp, out, K, D, H, W, rd, rh, rw, sz, nz, sy, ny, sx, nx, dense)) return e;
    if (!class_base || log2m_z < 0 || log2m_z > 2 || log2m_y < 0 || log2m_y > 2 || log2m_x < 0 || log2m_x > 2) return fail(MH_ERR_ARG, "sw_blend_mosaic: bad layout");
    if (K < 1 || K > 8) return fail(MH_ERR_UNSUPPORTED, "sw_blend_mosaic: 1 .. 8 classes (got %d): use the window-major layout", K);
    RegGrid rg;
    if (!regular_grid(rg, sz, nz, sy, ny, sx, nx, D, H, W)) return fail(MH_ERR_UNSUPPORTED, "sw_blend_mosaic: irregular window starts (use the window-major layout)");
    if (!(W % 4 == 0 && rw % 4 == 0 && all_mult4(sx, nx) && aligned(logits, 16) && aligned(imp, 16) && aligned(out, 16)) || (imp_factored && (rd + rh) % 4))
        return fail(MH_ERR_UNSUPPORTED, "sw_blend_mosaic: needs W, roi width and window starts divisible by 4 and 16-byte aligned buffers");
    // windows i and i + m of an axis must not overlap
    const AxisWin* ax[3] = {&rg.z, &rg.y, &rg.x};
    const int lm[3] = {log2m_z, log2m_y, log2m_x}, rr[3] = {rd, rh, rw};
    for (int a = 0; a < 3; ++a)
        if (ax[a]->n > 2 && (long long)ax[a]->step << lm[a] < rr[a]) return fail(MH_ERR_ARG, "sw_blend_mosaic: %d residue classes do not separate the windows of axis %d", 1 << lm[a], a);
    Mosaic ms;
    mosaic_axis_fill(ms.z, nz, log2m_z); mosaic_axis_fill(ms.y, ny, log2m_y); mosaic_axis_fill(ms.x, nx, log2m_x);
    for (int i = 0; i < MOSAIC_MAX_CLASSES * MOSAIC_MAX_CLASSES * MOSAIC_MAX_CLASSES; ++i) {
        if (class_base[i] % 4) return fail(MH_ERR_ARG, "sw_blend_mosaic: class offsets must be multiples of 4 floats");
        ms.base[i] = class_base[i];
    }
    if (D > 65535) return fail(MH_ERR_UNSUPPORTED, "sw_blend_mosaic: more than 65535 planes (use the window-major layout)");
    const unsigned nb = blocks_for((long long)H * (W / 4));
    hipStream_t s = (hipStream_t)stream;
    // FAST index arithmetic (kernels/sliding.h): 24-bit factors, 32-bit offsets -- when every class array's K channel planes stay below 2^31 floats (the benchmark:
    // 5 x (5 x 96)^3 = 0.55 G), its (z-extent x y-extent) and every single extent below 2^24; else the generic 64-bit form
    long long max_cs = 0, max_zy = 0, max_ext = std::max((long long)rd * rh, (long long)rw);
    for (int a = 0; a < MOSAIC_MAX_CLASSES; ++a)
        for (int b = 0; b < MOSAIC_MAX_CLASSES; ++b) {
            const long long zy = (long long)ms.z.cnt[a] * rd * ((long long)ms.y.cnt[b] * rh);
            max_zy = std::max(max_zy, zy);
            for (int c = 0; c < MOSAIC_MAX_CLASSES; ++c) {
                max_cs = std::max(max_cs, zy * ((long long)ms.x.cnt[c] * rw));
                max_ext = std::max(max_ext, std::max((long long)ms.z.cnt[a] * rd, std::max((long long)ms.y.cnt[b] * rh, (long long)ms.x.cnt[c] * rw)));
            }
        }
    const bool fast = max_cs * K < MH_BLEND_IDX32_LIMIT && max_zy < (1LL << 24) && max_ext < (1LL << 23) && std::max(D, std::max(H, W)) < (1 << 23) && knob_int("MONAI_AMD_BLEND_IDX64", 0) == 0;
#define MH_BM(KT, G_, NT_, SEP_) { if (fast) hipLaunchKernelGGL((sw_blend_mosaic_kernel<KT, G_, NT_, SEP_, true>), dim3(nb, D), dim3(256), 0, s, logits, imp, out, K, D, H, W, rd, rh, rw, rg, ms); \
                                   else hipLaunchKernelGGL((sw_blend_mosaic_kernel<KT, G_, NT_, SEP_, false>), dim3(nb, D), dim3(256), 0, s, logits, imp, out, K, D, H, W, rd, rh, rw, rg, ms); }
#ifdef MH_DEV_KNOBS
    if (K == 5 && !imp_factored) {        // A/B of the window-batch size and of non-temporal accesses on the benchmark shape (tools/blend_bench.py with the -DMH_DEV_KNOBS library)
        const int g_ = knob_int("MONAI_AMD_BLEND_G", MH_BLEND_G), nt_ = knob_int("MONAI_AMD_BLEND_NT", 0);
        if (g_ == 1 && !nt_) { MH_BM(5, 1, false, false) return launched("sw_blend_mosaic"); }
        if (g_ == 1 && nt_) { MH_BM(5, 1, true, false) return launched("sw_blend_mosaic"); }
        if (g_ == 2 && nt_) { MH_BM(5, 2, true, false) return launched("sw_blend_mosaic"); }
        if (g_ == 4 && !nt_) { MH_BM(5, 4, false, false) return launched("sw_blend_mosaic"); }
        if (g_ == 4 && nt_) { MH_BM(5, 4, true, false) return launched("sw_blend_mosaic"); }
    }
#endif
#define MH_BM_CASE(KT) case KT: if (imp_factored) MH_BM(KT, MH_BLEND_G, false, true) else MH_BM(KT, MH_BLEND_G, false, false) break;
    switch (K) { MH_BM_CASE(1) MH_BM_CASE(2) MH_BM_CASE(3) MH_BM_CASE(4) MH_BM_CASE(5) MH_BM_CASE(6) MH_BM_CASE(7) MH_BM_CASE(8) }
#undef MH_BM_CASE
#undef MH_BM
    return launched("sw_blend_mosaic");
}

int mh_sw_blend_argmax_f32(const float* logits, int64_t window_stride, const float* imp, void* labels, int labels_u8, int K, int D, int H, int W, int rd,
                           int rh, int rw, const int32_t* sz, int nz, const int32_t* sy, int ny, const int32_t* sx, int nx, int premultiplied, void* stream) {
    if (int e = blend_checks("sw_blend_argmax", logits, imp, labels, K, D, H, W, rd, rh, rw, sz, nz, sy, ny, sx, nx, window_stride)) return e;
    if (labels_u8 && K > 256) return fail(MH_ERR_UNSUPPORTED, "sw_blend_argmax: %d classes do not fit a uint8 label", K);
    RegGrid rg;
    if (!regular_grid(rg, sz, nz, sy, ny, sx, nx, D, H, W))
        return fail(MH_ERR_UNSUPPORTED, "sw_blend_argmax: irregular window starts (blend, then argmax)");
    const bool v4 = W % 4 == 0 && rw % 4 == 0 && window_stride % 4 == 0 && all_mult4(sx, nx) && aligned(logits, 16) && aligned(imp, 16) &&
                    aligned(labels, labels_u8 ? 4 : 16);
    const long long total = (long long)D * H * (v4 ? W / 4 : W);
    const bool nt = knob_int("MONAI_AMD_BLEND_NT", MH_BLEND_NT) != 0;
    if (int e = launch_blend_reg<true>(K < 8 ? K : 8, v4, MH_BLEND_G, nt, blocks_for(total), (hipStream_t)stream, logits, imp, labels, K, 0, D, H, W, rd, rh,
                                       rw, rg, premultiplied ? 1 : 0, labels_u8 ? 1 : 0, window_stride))
        return e;
    return launched("sw_blend_argmax");
}

// ------------------------------------------------------------------------------------------ conv 3x3x3
//                MX MY MZ MT WM WN NT CC OCC
typedef ConvCfg<32, 1, 1, 4, 4, 1, 1, 8, 2> Cfg1;   // tile 32x4x4, 32 couts   (96^3 level)
typedef ConvCfg<16, 2, 1, 4, 4, 1, 1, 8, 2> Cfg2;   // tile 16x8x4, 32 couts   (48^3 level)
typedef ConvCfg<8, 4, 1, 2, 2, 2, 1, 8, 2> Cfg3;    // tile 8x8x2,  64 couts   (24^3 level)
typedef ConvCfg<4, 4, 2, 1, 1, 4, 1, 4, 2> Cfg4;    // tile 4x4x2, 128 couts   (12^3 level)
typedef ConvCfg<4, 4, 2, 1, 1, 4, 2, 2, 2> Cfg5;    // tile 4x4x2, 256 couts   (6^3 level)
typedef ConvCfg<8, 4, 1, 2, 4, 1, 1, 8, 2> Cfg6;    // tile 8x8x4,  32 couts   (small volumes)
typedef ConvCfg<32, 1, 1, 4, 4, 1, 1, 2, 4> Cfg7;   // tile 32x4x4, 32 couts, 2-channel chunks (first layer, Cin = 1)
typedef ConvCfg<32, 1, 1, 4, 4, 1, 1, 4, 3> Cfg8;   // cfg1 with 4-channel chunks: 3 workgroups per CU
typedef ConvCfg<16, 2, 1, 4, 4, 1, 1, 4, 3> Cfg9;   // cfg2 with 4-channel chunks
typedef ConvCfg<16, 2, 1, 4, 4, 1, 1, 2, 4> Cfg10;  // cfg2 with 2-channel chunks: 4 workgroups per CU
typedef ConvCfg<8, 4, 1, 2, 2, 2, 1, 4, 3> Cfg11;   // cfg3 with 4-channel chunks
typedef ConvCfg<8, 4, 1, 2, 2, 2, 1, 2, 4> Cfg12;   // cfg3 with 2-channel chunks
typedef ConvCfg<4, 4, 2, 1, 1, 4, 1, 2, 4> Cfg13;   // cfg4 with 2-channel chunks
typedef ConvCfg<8, 4, 1, 2, 4, 1, 1, 2, 4> Cfg14;   // cfg6 with 2-channel chunks
#define MH_NUM_CFG 14

struct CfgInfo { int tx, ty, tz, cn, cc; };
#define MH_CFG_ROW(C) {C::TX, C::TY, C::TZ, C::CN, C::CC}
static const CfgInfo kCfg[MH_NUM_CFG + 1] = {
    {0, 0, 0, 0, 0}, MH_CFG_ROW(Cfg1), MH_CFG_ROW(Cfg2), MH_CFG_ROW(Cfg3), MH_CFG_ROW(Cfg4), MH_CFG_ROW(Cfg5),
    MH_CFG_ROW(Cfg6), MH_CFG_ROW(Cfg7), MH_CFG_ROW(Cfg8), MH_CFG_ROW(Cfg9), MH_CFG_ROW(Cfg10), MH_CFG_ROW(Cfg11),
    MH_CFG_ROW(Cfg12), MH_CFG_ROW(Cfg13), MH_CFG_ROW(Cfg14),
};
// measured preference among configurations of equal tile utilisation (kernel sweep on MI355X, profiles/):
// short channel chunks + more resident workgroups beat long chunks
static const double kPref[MH_NUM_CFG + 1] = {0, 1.00, 1.00, 1.00, 1.00, 0.90, 1.00, 1.06, 1.04, 1.03, 1.05, 1.01, 1.02, 1.01, 1.02};
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int cin_padded(int cfg, int Cin) { return cfg == 0 ? Cin : cdiv(Cin, kCfg[cfg].cc) * kCfg[cfg].cc; }
static inline int cout_padded(int cfg, int Cout) { return cfg == 0 ? Cout : cdiv(Cout, kCfg[cfg].cn) * kCfg[cfg].cn; }

// configuration MH_CFG_WINO2D: Winograd F(2x2, 3x3) in-plane + three direct z taps, z-streaming (kernels/conv3d_wino2p.h)
#define MH_CFG_WINO2D (MH_NUM_CFG + 1)
// configuration MH_CFG_H2: z-streaming direct convolution on the fp16 matrix cores in two-piece split precision, fp32-equivalent
// results (kernels/conv3d_h2.h); not counted by mh_conv3d_k3_num_configs (its tolerance class differs from the exact fp32 tiles)
#define MH_CFG_H2 (MH_NUM_CFG + 2)
// configuration MH_CFG_C1: one input channel (the first layer of every network), packed fp32 VALU, write-bound (kernels/conv3d_c1.h);
// exact fp32 like the tiles of 1 .. MH_NUM_CFG, not counted by mh_conv3d_k3_num_configs
#define MH_CFG_C1 (MH_NUM_CFG + 3)
// z-chunks of the one-channel kernel: a pure function of D (the statistics record count depends on it); 24-plane marches keep
// 64 windows of 96^3 at nine whole rounds of the chip's 2048 resident waves
// (configuration id MH_NUM_CFG + 4 was round 4's z-Winograd split-precision experiment: measured equal in time to the direct kernel, removed in round 5)
// configuration MH_CFG_H2C: MH_CFG_H2's kernel with output channel groups of 16, two z-taps sharing a 32-column matrix instruction (kernels/conv3d_h2.h, C16): layers
// with 16 output channels (UNETR's full-resolution levels) at 6 instead of 9 matrix instructions per tap; same arithmetic and tolerance class as MH_CFG_H2
#define MH_CFG_H2C (MH_NUM_CFG + 5)
// configuration MH_CFG_H2V: the split-precision convolution for SMALL volumes (D H W <= 256 voxels: the 6^3 level of a 96^3 window), one sample's whole volume as the
// workgroup's M tile (kernels/conv3d_vol_h2.h): where MH_CFG_H2's 16 x 16 regions would be mostly empty; same arithmetic and tolerance class as MH_CFG_H2
#define MH_CFG_H2V (MH_NUM_CFG + 6)
// configuration MH_CFG_H2W (round 6): in-plane Winograd F(2x2, 3x3) in front of MH_CFG_H2's split-precision product, transformed weights resident in registers
// (kernels/conv3d_wino_h2.h): 32 input channels, planes that tile into 4 x 16 regions; same record / bound contract and tolerance class as MH_CFG_H2
#define MH_CFG_H2W (MH_NUM_CFG + 7)
#ifndef MH_H2W_DEFAULT
#define MH_H2W_DEFAULT 1      // measured faster than MH_CFG_H2 at 96^3 and 48^3 (profiles/r06_h2w_*.txt); -DMH_H2W_DEFAULT=0 or MONAI_AMD_H2W=0 (dev build) gives the direct kernel back
#endif
#define MH_CFG_LAST MH_CFG_H2W
static inline bool hv_fits(int D, int H, int W) {
    const long long vol = (long long)D * H * W, padded = (long long)(D + 2) * (H + 2) * (W + 2);
    return vol >= 64 && vol <= 256 && padded <= HV_CELLS;
}
static inline int c1_chunks(int D) { return D >= 48 ? D / 24 : 1; }
static inline int c1_zchunk(int D) { return cdiv(D, c1_chunks(D)); }
static inline int c1_blocks(int D, int H, int W) { return cdiv(W, C1_TX) * cdiv(H, C1_TY) * cdiv(D, c1_zchunk(D)); }
// z-chunks of the streaming kernel: a pure function of the extents (the statistics record count depends on it)
static inline int wino2d_chunks(int D, int H, int W) {
    const int blocks = cdiv(W, W2_B) * cdiv(H, W2_B);
    if (const char* e = knob_str("MONAI_AMD_W2_CHUNKS")) {          // tuning knob (development)
        const int v = atoi(e);
        if (v >= 1 && v <= D) return v;
    }
    // measured at 64 windows per launch (profiles/): one chunk at 96^3 (36 regions), two at 48^3 (9 regions); every extra
    // chunk re-reads two halo planes
    int nchunk = cdiv(16, blocks);
    if (nchunk > D / 12) nchunk = D / 12;
    if (nchunk < 1) nchunk = 1;
    return nchunk;
}
static inline int wino2d_zchunk(int D, int H, int W) { return cdiv(D, wino2d_chunks(D, H, W)); }
static inline int wino2d_blocks(int D, int H, int W) { return cdiv(W, W2_B) * cdiv(H, W2_B) * cdiv(D, wino2d_zchunk(D, H, W)); }
// the split-precision kernel: regions of 16 x 16 or 8 x 32 outputs (whichever covers the plane with fewer: a pure function of the extents, like the
// statistics record count that depends on it), the same z-chunk rule on its own region count
// its result stores address the 32 cout planes of a workgroup through one raw buffer with 31-bit byte offsets (kernels/conv3d_h2.h)
static inline bool h2_fits(int D, int H, int W) { return (long long)D * H * W * H2_CN * 4 < 0x80000000LL; }
static inline int h2_regions(int H, int W) { return h2_wide(H, W) ? cdiv(W, 32) * cdiv(H, 8) : cdiv(W, H2_B) * cdiv(H, H2_B); }
static inline int h2_zchunk(int D, int H, int W) {
    int nchunk = cdiv(16, h2_regions(H, W));
    if (const char* e = knob_str("MONAI_AMD_W2_CHUNKS")) {          // tuning knob (development)
        const int v = atoi(e);
        if (v >= 1 && v <= D) return cdiv(D, v);
    }
    if (nchunk > D / 12) nchunk = D / 12;
    if (nchunk < 1) nchunk = 1;
    return cdiv(D, nchunk);
}
static inline int h2_blocks(int D, int H, int W) { return h2_regions(H, W) * cdiv(D, h2_zchunk(D, H, W)); }
// the Winograd split-precision kernel: 4 x 16 regions (whole ones only), z-chunks a pure function of the extents (the statistics record count depends on it)
static inline bool hw_fits(int D, int H, int W) { return H >= HWG_BY && W >= HWG_BX && H % HWG_BY == 0 && W % HWG_BX == 0 && D >= 2 && h2_fits(D, H, W); }
static inline int hw_regions(int H, int W) { return (H / HWG_BY) * (W / HWG_BX); }
static inline int hw_zchunk(int D, int H, int W) {
    int nchunk = cdiv(32, hw_regions(H, W));
    if (const char* e = knob_str("MONAI_AMD_HW_CHUNKS")) {          // tuning knob (development)
        const int v = atoi(e);
        if (v >= 1 && v <= D) nchunk = v;
    }
    if (nchunk > D / 12) nchunk = D / 12;
    if (nchunk < 1) nchunk = 1;
    int zc = cdiv(D, nchunk);
    zc += zc & 1;                                                   // even chunks: the pooling epilogue pairs planes inside a chunk
    return zc;
}
static inline int hw_blocks(int D, int H, int W) { return hw_regions(H, W) * cdiv(D, hw_zchunk(D, H, W)); }

int mh_conv3d_k3_num_configs(void) { return MH_CFG_WINO2D; }
int mh_conv3d_k3_h2_config(void) { return MH_CFG_H2; }
int mh_conv3d_k3_h2c_config(void) { return MH_CFG_H2C; }
int mh_conv3d_k3_c1_config(void) { return MH_CFG_C1; }
int mh_conv3d_k3_h2v_config(void) { return MH_CFG_H2V; }
int mh_conv3d_k3_h2w_config(void) { return MH_CFG_H2W; }
int mh_conv3d_k3_h2w_fits(int D, int H, int W) { return hw_fits(D, H, W) ? 1 : 0; }

int mh_conv3d_k3_pool_accepts(int cfg, int Cin, int Cout, int D, int H, int W) {
    if (cfg == MH_CFG_H2W) return Cin == HWG_CIN && Cout >= HWG_CN && Cout % HWG_CN == 0 && D % 2 == 0 && hw_fits(D, H, W);
    if (cfg != MH_CFG_H2 || D < 2 || H < 2 || W < 2 || D % 2 || H % 2 || W % 4) return 0;
    if (!(Cin >= H2_KC && Cin % H2_KC == 0 && Cin <= H2_NRM_MAX && Cout >= H2_CN && Cout % (H2_CN / 2) == 0) || !h2_fits(D, H, W)) return 0;
    return !h2_wide(H, W) && h2_zchunk(D, H, W) % 2 == 0;
}
int mh_conv3d_k3_accepts(int cfg, int Cin, int Cout) {
    if (cfg == 0) return 1;
    if (cfg == MH_CFG_WINO2D) return Cin >= 8 && Cin % 8 == 0 && Cout >= W2_CN && Cout % W2_CN == 0;
    // round 4: the last cout group of the direct split kernel may be half full (Cout % 16 == 0 beyond 32: 48, 80, ... -- SwinUNETR(feature_size 48)'s full-resolution levels)
    if (cfg == MH_CFG_H2) return Cin >= H2_KC && Cin % H2_KC == 0 && Cin <= H2_NRM_MAX && Cout >= H2_CN && Cout % (H2_CN / 2) == 0;
    if (cfg == MH_CFG_H2C) return Cin >= H2_KC && Cin % H2_KC == 0 && Cin <= H2_NRM_MAX && Cout >= 16 && Cout % 16 == 0;
    if (cfg == MH_CFG_C1) return Cin == 1 && Cout >= 8 && Cout % 8 == 0;
    if (cfg == MH_CFG_H2V) return Cin >= 16 && Cin % 16 == 0 && Cin <= HV_CIN_MAX && Cout >= 32 && Cout % 32 == 0;
    if (cfg == MH_CFG_H2W) return Cin == HWG_CIN && Cout >= HWG_CN && Cout % HWG_CN == 0;
    if (cfg < 0 || cfg > MH_NUM_CFG) return 0;
    return Cout >= 1 && cin_padded(cfg, Cin) <= Cfg1::NRM_MAX;
}

// Heuristic choice: the configuration that wastes the least matrix work -- masked voxels of partial tiles and
// zero-padded input channels both count -- preferring the wider cout tile, then the measured preference.
int mh_conv3d_k3_select(int algo, int input_bounded, int Cin, int Cout, int D, int H, int W) {
    if (Cin < 1 || Cout < 1 || D < 1 || H < 1 || W < 1) return fail(MH_ERR_ARG, "conv3d_k3_select: bad argument");
    if (algo < MH_ALGO_AUTO || algo > MH_ALGO_FP32) return fail(MH_ERR_ARG, "conv3d_k3_select: unknown algorithm family %d", algo);
    int best = 0;
    double best_score = 0.0;
    for (int c = 1; c <= MH_NUM_CFG; ++c) {
        const CfgInfo& k = kCfg[c];
        if (!mh_conv3d_k3_accepts(c, Cin, Cout)) continue;
        const double util = (double)D * H * W / ((double)cdiv(D, k.tz) * k.tz * cdiv(H, k.ty) * k.ty * cdiv(W, k.tx) * k.tx);
        const double cpad = ((double)Cin / cin_padded(c, Cin)) * ((double)Cout / cout_padded(c, Cout));
        const double score = util * cpad * (1.0 + 0.02 * (k.cn / 32)) * kPref[c];
        if (score > best_score) { best_score = score; best = c; }
    }
    // in-plane Winograd, z-streaming: 2.25x fewer matrix-core cycles.  On gfx950 the fp32 MFMA does not co-issue with VALU
    // work of the same wave (tools/ubench/issue.hip), so its transforms are paid for serially: measured 1.11-1.13x over the
    // best direct tile at 96^3, 1.04x at 48^3, slower below (profiles/) -- chosen for full 16 x 16 regions of large planes.
    if (mh_conv3d_k3_accepts(MH_CFG_WINO2D, Cin, Cout) && H % 2 == 0 && W % 8 == 0) {
        const bool big = H % W2_B == 0 && W % W2_B == 0 && D >= 48 && H >= 48 && W >= 48;
        if (algo == MH_ALGO_WINO2D || ((algo == MH_ALGO_AUTO || algo == MH_ALGO_FP32) && big)) best = MH_CFG_WINO2D;
    }
    // fp16 two-piece split precision on the fp16 matrix cores: fp32-equivalent results (the oracle network's logits move by 4e-6,
    // the level of two fp32 summation orders) at 1.8-2.1x the speed of the kernels above on every level it takes (96^3 ... 12^3,
    // profiles/r02_h2_vs_wino2p_*.json).  It scales its input into fp16's range by a power of two taken from the bounds the input
    // records carry (kernels/conv3d_h2.h), so it is chosen -- also under MH_ALGO_H2 -- only for inputs that carry them
    // (`input_bounded`); anything else gets the exact fp32 kernels.
    if ((algo == MH_ALGO_AUTO || algo == MH_ALGO_H2) && input_bounded && mh_conv3d_k3_accepts(MH_CFG_H2, Cin, Cout) && W % 4 == 0 && H >= 8 && W >= 8 && h2_fits(D, H, W))
        best = MH_CFG_H2;
    // 16 output channels: the same kernel with two z-taps per matrix instruction (6 instead of 9 per tap; a 32-column group would be half zero weights)
    if ((algo == MH_ALGO_AUTO || algo == MH_ALGO_H2) && input_bounded && Cout == 16 && mh_conv3d_k3_accepts(MH_CFG_H2C, Cin, Cout) && W % 4 == 0 && H >= 8 && W >= 8 && h2_fits(D, H, W)
        && knob_int("MONAI_AMD_H2C", 1) != 0)
        best = MH_CFG_H2C;
    // small volumes (the 6^3 level of a 96^3 window) that the z-marching kernel's 16 x 16 regions would leave mostly empty: one sample's whole volume as the M tile
    if ((algo == MH_ALGO_AUTO || algo == MH_ALGO_H2) && input_bounded && best != MH_CFG_H2 && best != MH_CFG_H2C && mh_conv3d_k3_accepts(MH_CFG_H2V, Cin, Cout) && hv_fits(D, H, W))
        best = MH_CFG_H2V;
    // round 6: 32 input channels on planes of whole 4 x 16 regions: in-plane Winograd in front of the split product (kernels/conv3d_wino_h2.h)
    if (algo == MH_ALGO_AUTO && best == MH_CFG_H2 && mh_conv3d_k3_accepts(MH_CFG_H2W, Cin, Cout) && hw_fits(D, H, W) && D >= 24 && knob_int("MONAI_AMD_H2W", MH_H2W_DEFAULT) != 0)      // MH_ALGO_H2 by name: the direct kernel
        best = MH_CFG_H2W;
    // one input channel: the packed-VALU kernel is write-bound where the fp32 MFMA tile multiplies a zero-padded channel
    if (algo != MH_ALGO_DIRECT && algo != MH_ALGO_WINO2D && mh_conv3d_k3_accepts(MH_CFG_C1, Cin, Cout) && W % 4 == 0 && knob_int("MONAI_AMD_C1", 1) != 0)
        best = MH_CFG_C1;
    return best;
}

int64_t mh_conv3d_k3_packed_floats(int cfg, int Cin, int Cout) {
    if (cfg == MH_CFG_WINO2D) return (int64_t)(Cin / W2_KC) * (Cout / W2_CN) * W2_UBUF;
    if (cfg == MH_CFG_H2) return (int64_t)(Cin / H2_KC) * cdiv(Cout, H2_CN) * H2_WB * 4 + H2_TAIL;   // padded chunk slabs of two fp16 pieces + {1 / scale, scale}
    if (cfg == MH_CFG_H2C) return (int64_t)(Cin / H2_KC) * (Cout / 16) * H2_WB * 4 + H2_TAIL;         // the same slabs, one per group of 16 couts
    if (cfg == MH_CFG_C1) return (int64_t)27 * Cout;                                             // [27 taps][Cout]
    if (cfg == MH_CFG_H2V) return (int64_t)Cout * Cin * 27 + H2_TAIL;                             // two fp16 pieces per weight + {1 / scale, scale}
    if (cfg == MH_CFG_H2W) return (int64_t)(Cout / HWG_CN) * HWG_WAVES * HWG_OPS * 64 * 4 + H2_TAIL;       // [cout group][wave][its operands][64 lanes][8 halves]
    if (cfg < 0 || cfg > MH_NUM_CFG) return fail(MH_ERR_ARG, "conv3d_k3: unknown configuration %d", cfg);
    return (int64_t)cin_padded(cfg, Cin) * cout_padded(cfg, Cout) * 27;
}

int mh_conv3d_k3_pack_f32(int cfg, const float* w, int Cin, int Cout, float* packed, void* stream) {
    if (cfg == MH_CFG_WINO2D) {
        if (!w || !packed || !mh_conv3d_k3_accepts(cfg, Cin, Cout)) return fail(MH_ERR_ARG, "conv3d_k3_pack: in-plane Winograd needs Cin %% 8 == 0, Cout %% 16 == 0");
        if (hipMemsetAsync(packed, 0, sizeof(float) * (size_t)mh_conv3d_k3_packed_floats(cfg, Cin, Cout), (hipStream_t)stream) != hipSuccess)
            return fail(MH_ERR_LAUNCH, "conv3d_k3_pack: memset failed");
        hipLaunchKernelGGL(conv3d_k3_wino2d_pack_kernel, dim3(blocks_for((long long)Cin * Cout)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, packed);
        return launched("conv3d_k3_wino2d_pack");
    }
    if (cfg == MH_CFG_H2C) {
        if (!w || !packed || !mh_conv3d_k3_accepts(cfg, Cin, Cout)) return fail(MH_ERR_ARG, "conv3d_k3_pack: the 16-couts fp16 split kernel needs Cin %% 16 == 0, Cout %% 16 == 0");
        const int64_t slab_floats = mh_conv3d_k3_packed_floats(cfg, Cin, Cout) - H2_TAIL;
        float* tail = packed + slab_floats;
        if (hipMemsetAsync(packed, 0, sizeof(float) * (size_t)slab_floats, (hipStream_t)stream) != hipSuccess)
            return fail(MH_ERR_LAUNCH, "conv3d_k3_pack: memset failed");
        hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)Cin * Cout * 27, tail);
        hipLaunchKernelGGL(conv3d_k3_h2c_pack_kernel, dim3(blocks_for((long long)Cin * Cout)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout,
                           reinterpret_cast<_Float16*>(packed), tail);
        return launched("conv3d_k3_h2c_pack");
    }
    if (cfg == MH_CFG_H2) {
        if (!w || !packed || !mh_conv3d_k3_accepts(cfg, Cin, Cout)) return fail(MH_ERR_ARG, "conv3d_k3_pack: the fp16 split kernel needs Cin %% 16 == 0, Cout %% 16 == 0, Cout >= 32");
        const int64_t slab_floats = mh_conv3d_k3_packed_floats(cfg, Cin, Cout) - H2_TAIL;
        float* tail = packed + slab_floats;
        if (hipMemsetAsync(packed, 0, sizeof(float) * (size_t)slab_floats, (hipStream_t)stream) != hipSuccess)
            return fail(MH_ERR_LAUNCH, "conv3d_k3_pack: memset failed");
        hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)Cin * Cout * 27, tail);
        hipLaunchKernelGGL(conv3d_k3_h2_pack_kernel, dim3(blocks_for((long long)Cin * Cout)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout,
                           reinterpret_cast<_Float16*>(packed), tail);
        return launched("conv3d_k3_h2_pack");
    }
    if (cfg == MH_CFG_H2W) {
        if (!w || !packed || !mh_conv3d_k3_accepts(cfg, Cin, Cout)) return fail(MH_ERR_ARG, "conv3d_k3_pack: the Winograd split kernel needs Cin == 32, Cout %% 32 == 0");
        float* tail = packed + (mh_conv3d_k3_packed_floats(cfg, Cin, Cout) - H2_TAIL);
        hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)Cin * Cout * 27, tail);
        hipLaunchKernelGGL(conv3d_k3_h2w_scale_fix_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, tail);
        hipLaunchKernelGGL(conv3d_k3_h2w_pack_kernel, dim3(blocks_for((long long)Cin * Cout)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout,
                           reinterpret_cast<_Float16*>(packed), tail);
        return launched("conv3d_k3_h2w_pack");
    }
    if (cfg == MH_CFG_H2V) {
        if (!w || !packed || !mh_conv3d_k3_accepts(cfg, Cin, Cout)) return fail(MH_ERR_ARG, "conv3d_k3_pack: the small-volume split kernel needs Cin %% 16 == 0 (<= 768), Cout %% 32 == 0");
        float* tail = packed + (mh_conv3d_k3_packed_floats(cfg, Cin, Cout) - H2_TAIL);
        hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)Cin * Cout * 27, tail);
        hipLaunchKernelGGL(conv3d_k3_vol_h2_pack_kernel, dim3(blocks_for((long long)Cin * Cout)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, Cout % 64 == 0 ? 2 : 1,
                           reinterpret_cast<_Float16*>(packed), tail);
        return launched("conv3d_k3_vol_h2_pack");
    }
    if (cfg == MH_CFG_C1) {          // [Cin = 1][27][Cout]: the direct kernel's layout
        if (!w || !packed || !mh_conv3d_k3_accepts(cfg, Cin, Cout)) return fail(MH_ERR_ARG, "conv3d_k3_pack: the one-channel kernel needs Cin == 1, Cout %% 8 == 0");
        hipLaunchKernelGGL(conv3d_k3_pack_kernel, dim3(blocks_for(27LL * Cout)), dim3(256), 0, (hipStream_t)stream, w, 1, 1, Cout, Cout, Cout, packed);
        return launched("conv3d_k3_pack");
    }
    if (cfg < 0 || cfg > MH_NUM_CFG || !w || !packed) return fail(MH_ERR_ARG, "conv3d_k3_pack: bad argument");
    const int cn = cfg == 0 ? Cout : kCfg[cfg].cn;
    const int cinp = cin_padded(cfg, Cin), coutp = cout_padded(cfg, Cout);
    hipLaunchKernelGGL(conv3d_k3_pack_kernel, dim3(blocks_for((long long)cinp * coutp * 27)), dim3(256), 0, (hipStream_t)stream, w,
                       Cin, cinp, Cout, coutp, cn, packed);
    return launched("conv3d_k3_pack");
}

int mh_conv3d_k3_stat_tiles(int cfg, int D, int H, int W) {
    if (cfg == MH_CFG_WINO2D) return wino2d_blocks(D, H, W);
    if (cfg == MH_CFG_H2 || cfg == MH_CFG_H2C) return h2_blocks(D, H, W);
    if (cfg == MH_CFG_C1) return c1_blocks(D, H, W);
    if (cfg == MH_CFG_H2V) return 1;          // the workgroup holds the sample's whole volume
    if (cfg == MH_CFG_H2W) return hw_fits(D, H, W) ? hw_blocks(D, H, W) : 0;
    if (cfg < 1 || cfg > MH_NUM_CFG) return 0;
    const CfgInfo& k = kCfg[cfg];
    return cdiv(W, k.tx) * cdiv(H, k.ty) * cdiv(D, k.tz);
}

template <class Cfg>
static void launch_mfma(const Tensor& in, const float* wp, const float* bias, const Tensor& out, float* stats, hipStream_t s) {
    const int tx = cdiv(out.W, Cfg::TX), ty = cdiv(out.H, Cfg::TY), tz = cdiv(out.D, Cfg::TZ);
    const dim3 grid((unsigned)(tx * ty * tz), (unsigned)cdiv(out.C, Cfg::CN), (unsigned)out.N);
    if (stats)
        hipLaunchKernelGGL((conv3d_k3_mfma_kernel<Cfg, true>), grid, dim3(256), 0, s, in, wp, bias, out, stats, tx, ty, tz);
    else
        hipLaunchKernelGGL((conv3d_k3_mfma_kernel<Cfg, false>), grid, dim3(256), 0, s, in, wp, bias, out, stats, tx, ty, tz);
}

struct PoolOut { float* mx; float* mn; long long n_stride; };
static int conv3d_k3_launch(int cfg, const mh_tensor5* in_, const float* packed_w, const float* bias, const mh_tensor5* out_, float* stats, void* stream, bool accumulate,
                            const PoolOut* pool = nullptr);
// can mh_conv3d_k3_pool_f32 serve this layer?  the split-precision configuration's 16 x 16 regions (not the 8 x 32 shape), even extents, even z-chunks
int mh_conv3d_k3_pool_accepts(int cfg, int Cin, int Cout, int D, int H, int W);
int mh_conv3d_k3_pool_f32(int cfg, const mh_tensor5* in_, const float* packed_w, const float* bias, const mh_tensor5* out_, float* stats, float* pool_max, float* pool_min,
                          int64_t pool_n_stride, void* stream) {
    if (!in_ || !out_ || !pool_max || !pool_min || !stats || !in_->nrm) return fail(MH_ERR_ARG, "conv3d_k3_pool: input records, statistics and both pooled tensors are required");
    if (!mh_conv3d_k3_pool_accepts(cfg, in_->C, out_->C, out_->D, out_->H, out_->W))
        return fail(MH_ERR_UNSUPPORTED, "conv3d_k3_pool: the pooling epilogue exists for the split-precision configuration on even extents with 16 x 16 regions");
    if (!aligned(pool_max, 8) || !aligned(pool_min, 8) || pool_n_stride % 2 || (long long)out_->C * (out_->D / 2) * (out_->H / 2) * (out_->W / 2) > pool_n_stride)
        return fail(MH_ERR_ARG, "conv3d_k3_pool: pooled tensors must be 8-byte aligned [N][>= C][D/2][H/2][W/2]");
    const PoolOut po{pool_max, pool_min, (long long)pool_n_stride};
    return conv3d_k3_launch(cfg, in_, packed_w, bias, out_, stats, stream, false, &po);
}
int mh_pool_select_f32(float* pool_max, const float* pool_min, const float* nrm, int64_t nrm_n_stride, int N, int C, int64_t n_stride, int64_t vol, void* stream) {
    if (!pool_max || !pool_min || !nrm || N < 1 || C < 1 || vol < 1) return fail(MH_ERR_ARG, "pool_select: bad argument");
    const unsigned nb = (unsigned)std::min<long long>(blocks_for(vol), 64);
    hipLaunchKernelGGL(pool_select_kernel, dim3(nb, (unsigned)C, (unsigned)N), dim3(256), 0, (hipStream_t)stream, pool_max, pool_min, nrm, (long long)nrm_n_stride, (long long)n_stride, (long long)vol);
    return launched("pool_select");
}
int mh_conv3d_k3_f32(int cfg, const mh_tensor5* in_, const float* packed_w, const float* bias, const mh_tensor5* out_,
                     float* stats, void* stream) {
    return conv3d_k3_launch(cfg, in_, packed_w, bias, out_, stats, stream, false);
}
int mh_conv3d_k3_accumulate_f32(int cfg, const mh_tensor5* in_, const float* packed_w, const float* bias, const mh_tensor5* out_,
                                float* stats, void* stream) {
    if ((cfg != MH_CFG_H2 && cfg != MH_CFG_H2C && cfg != MH_CFG_H2W) || !stats || !in_ || !in_->nrm)
        return fail(MH_ERR_UNSUPPORTED, "conv3d_k3_accumulate: the accumulating form exists for the split-precision configuration with input records and statistics");
    return conv3d_k3_launch(cfg, in_, packed_w, bias, out_, stats, stream, true);
}
static int conv3d_k3_launch(int cfg, const mh_tensor5* in_, const float* packed_w, const float* bias, const mh_tensor5* out_,
                            float* stats, void* stream, bool accumulate, const PoolOut* pool) {
    if (!dense_ok(in_) || !dense_ok(out_) || !packed_w) return fail(MH_ERR_ARG, "conv3d_k3: bad tensor");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || in.D != out.D || in.H != out.H || in.W != out.W) return fail(MH_ERR_ARG, "conv3d_k3: shape mismatch");
    if (cfg < 0 || cfg > MH_CFG_LAST) return fail(MH_ERR_ARG, "conv3d_k3: unknown configuration %d", cfg);
    if (in.nrm && !aligned(in.nrm, 16)) return fail(MH_ERR_ARG, "conv3d_k3: nrm must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (cfg == MH_CFG_C1) {
        if (!mh_conv3d_k3_accepts(cfg, in.C, out.C) || in.W % 4)
            return fail(MH_ERR_ARG, "conv3d_k3: the one-channel kernel needs Cin == 1, Cout %% 8 == 0, W %% 4 == 0 (got %d -> %d, %dx%dx%d)", in.C, out.C,
                        in.D, in.H, in.W);
        if (!aligned(out.data, 16) || out.n_stride % 4 || !aligned(in.data, 16) || in.n_stride % 4 || !aligned(packed_w, 4))
            return fail(MH_ERR_ARG, "conv3d_k3: the one-channel kernel needs 16-byte aligned input and output");
        const int txn = cdiv(out.W, C1_TX), tyn = cdiv(out.H, C1_TY), zc = c1_zchunk(out.D);
        // 8 output channels per thread: 92 registers, no scratch, 1.54 ms per 64 windows of 1 -> 32 ch @ 96^3 (16 per thread halve the input
        // reads but need 128 registers + 36 bytes of scratch: 1.86 ms; gpurun_out/r3first -> profiles/r03_c1_cot_ab.txt)
        constexpr int cot = 8;
        const dim3 grid((unsigned)(txn * tyn * cdiv(out.D, zc)), (unsigned)(out.C / cot), (unsigned)out.N);
        if (stats && in.nrm) hipLaunchKernelGGL((conv3d_k3_c1_kernel<cot, true, true>), grid, dim3(256), 0, s, in, packed_w, bias, out, stats, txn, tyn, zc);
        else if (stats) hipLaunchKernelGGL((conv3d_k3_c1_kernel<cot, true, false>), grid, dim3(256), 0, s, in, packed_w, bias, out, stats, txn, tyn, zc);
        else if (in.nrm) hipLaunchKernelGGL((conv3d_k3_c1_kernel<cot, false, true>), grid, dim3(256), 0, s, in, packed_w, bias, out, stats, txn, tyn, zc);
        else hipLaunchKernelGGL((conv3d_k3_c1_kernel<cot, false, false>), grid, dim3(256), 0, s, in, packed_w, bias, out, stats, txn, tyn, zc);
        return launched("conv3d_k3_c1");
    }
    if (cfg == MH_CFG_H2W) {
        if (!mh_conv3d_k3_accepts(cfg, in.C, out.C) || !hw_fits(in.D, in.H, in.W))
            return fail(MH_ERR_ARG, "conv3d_k3: the Winograd split kernel needs Cin == 32, Cout %% 32 == 0, H %% 4 == 0, W %% 16 == 0, D*H*W < 2^24 (got %d -> %d, %dx%dx%d)",
                        in.C, out.C, in.D, in.H, in.W);
        if (!in.nrm) return fail(MH_ERR_ARG, "conv3d_k3: the Winograd split kernel needs input records with magnitude bounds");
        if (!aligned(out.data, 16) || out.n_stride % 4 || !aligned(packed_w, 16)) return fail(MH_ERR_ARG, "conv3d_k3: the Winograd split kernel needs 16-byte aligned output and weights");
        if ((accumulate || pool) && !stats) return fail(MH_ERR_ARG, "conv3d_k3: the accumulating / pooling forms leave statistics");
        const int bxn = out.W / HWG_BX, byn = out.H / HWG_BY, zc = hw_zchunk(out.D, out.H, out.W);
        const unsigned nblk = (unsigned)(bxn * byn * cdiv(out.D, zc));
        const long long total = (long long)nblk * (out.C / HWG_CN) * out.N;
        if (total > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "conv3d_k3: problem too large for one launch");
        const dim3 grid((unsigned)total);
        const uint4* wq = reinterpret_cast<const uint4*>(packed_w);
        const float* tail = packed_w + (mh_conv3d_k3_packed_floats(cfg, in.C, out.C) - H2_TAIL);
        if (pool) hipLaunchKernelGGL((conv3d_k3_h2w_kernel<true, false, true>), grid, dim3(64 * HWG_WAVES), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, pool->mx, pool->mn, pool->n_stride);
        else if (accumulate) hipLaunchKernelGGL((conv3d_k3_h2w_kernel<true, true, false>), grid, dim3(64 * HWG_WAVES), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, (float*)nullptr, (float*)nullptr, 0LL);
        else if (stats) hipLaunchKernelGGL((conv3d_k3_h2w_kernel<true, false, false>), grid, dim3(64 * HWG_WAVES), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, (float*)nullptr, (float*)nullptr, 0LL);
        else hipLaunchKernelGGL((conv3d_k3_h2w_kernel<false, false, false>), grid, dim3(64 * HWG_WAVES), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, (float*)nullptr, (float*)nullptr, 0LL);
        return launched("conv3d_k3_h2w");
    }
    if (cfg == MH_CFG_H2V) {
        if (accumulate || pool) return fail(MH_ERR_UNSUPPORTED, "conv3d_k3: the small-volume split kernel has no accumulating / pooling form");
        if (!mh_conv3d_k3_accepts(cfg, in.C, out.C) || !hv_fits(in.D, in.H, in.W))
            return fail(MH_ERR_ARG, "conv3d_k3: the small-volume split kernel needs Cin %% 16 == 0 (<= 768), Cout %% 32 == 0, 64 <= D*H*W <= 256, (D+2)(H+2)(W+2) <= 512 (got %d -> %d, %dx%dx%d)",
                        in.C, out.C, in.D, in.H, in.W);
        if (!in.nrm) return fail(MH_ERR_ARG, "conv3d_k3: the small-volume split kernel needs input records with magnitude bounds");
        if (!aligned(out.data, 16) || out.n_stride % 4 || !aligned(packed_w, 16)) return fail(MH_ERR_ARG, "conv3d_k3: the small-volume split kernel needs 16-byte aligned output and weights");
        const int ncgw = out.C % 64 == 0 ? 2 : 1;
        const long long total = (long long)(out.C / (32 * ncgw)) * out.N;
        if (total > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "conv3d_k3: problem too large for one launch");
        const uint4* wq = reinterpret_cast<const uint4*>(packed_w);
        const float* tail = packed_w + (mh_conv3d_k3_packed_floats(cfg, in.C, out.C) - H2_TAIL);
        const dim3 grid((unsigned)total);
        if (ncgw == 2) {
            if (stats) hipLaunchKernelGGL((conv3d_k3_vol_h2_kernel<2, true>), grid, dim3(HV_NT), 0, s, in, wq, tail, bias, out, stats);
            else hipLaunchKernelGGL((conv3d_k3_vol_h2_kernel<2, false>), grid, dim3(HV_NT), 0, s, in, wq, tail, bias, out, stats);
        } else {
            if (stats) hipLaunchKernelGGL((conv3d_k3_vol_h2_kernel<1, true>), grid, dim3(HV_NT), 0, s, in, wq, tail, bias, out, stats);
            else hipLaunchKernelGGL((conv3d_k3_vol_h2_kernel<1, false>), grid, dim3(HV_NT), 0, s, in, wq, tail, bias, out, stats);
        }
        return launched("conv3d_k3_vol_h2");
    }
    if (cfg == MH_CFG_H2 || cfg == MH_CFG_H2C) {
        const bool c16 = cfg == MH_CFG_H2C;
        if (!mh_conv3d_k3_accepts(cfg, in.C, out.C) || in.W % 4)
            return fail(MH_ERR_ARG, "conv3d_k3: the fp16 split kernel needs Cin %% 16 == 0, Cout %% 16 == 0 (>= 32; any multiple of 16 in its 16-couts form), W %% 4 == 0 (got %d -> %d, %dx%dx%d)",
                        in.C, out.C, in.D, in.H, in.W);
        if (!aligned(out.data, 16) || out.n_stride % 4 || !aligned(packed_w, 16))
            return fail(MH_ERR_ARG, "conv3d_k3: the fp16 split kernel needs 16-byte aligned output and weights");
        if (!h2_fits(out.D, out.H, out.W))
            return fail(MH_ERR_UNSUPPORTED, "conv3d_k3: the fp16 split kernel addresses 32 output planes with 32-bit offsets (D*H*W < 2^24 voxels); mh_conv3d_k3_select does not return it beyond");
        const bool wide = h2_wide(out.H, out.W);
        const int bxn = wide ? cdiv(out.W, 32) : cdiv(out.W, H2_B), byn = wide ? cdiv(out.H, 8) : cdiv(out.H, H2_B), zc = h2_zchunk(out.D, out.H, out.W);
        const unsigned nblk = (unsigned)(bxn * byn * cdiv(out.D, zc));
        const long long total = (long long)nblk * (c16 ? out.C / 16 : cdiv(out.C, H2_CN)) * out.N;
        if (total > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "conv3d_k3: problem too large for one launch");
        const dim3 grid((unsigned)total);
        const uint4* wq = reinterpret_cast<const uint4*>(packed_w);
        const float* tail = packed_w + (mh_conv3d_k3_packed_floats(cfg, in.C, out.C) - H2_TAIL);
#define MH_H2_LAUNCH(RES_, WIDE_, C16_)                                                                                                                  \
    {                                                                                                                                                \
        if (stats && in.nrm) hipLaunchKernelGGL((conv3d_k3_h2_kernel<true, true, RES_, WIDE_, C16_>), grid, dim3(512), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, (float*)nullptr, (float*)nullptr, 0LL);   \
        else if (stats) hipLaunchKernelGGL((conv3d_k3_h2_kernel<true, false, RES_, WIDE_, C16_>), grid, dim3(512), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, (float*)nullptr, (float*)nullptr, 0LL);       \
        else if (in.nrm) hipLaunchKernelGGL((conv3d_k3_h2_kernel<false, true, RES_, WIDE_, C16_>), grid, dim3(512), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, (float*)nullptr, (float*)nullptr, 0LL);      \
        else hipLaunchKernelGGL((conv3d_k3_h2_kernel<false, false, RES_, WIDE_, C16_>), grid, dim3(512), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, (float*)nullptr, (float*)nullptr, 0LL);                 \
    }
        if (pool) {             // MaxPool3d(2) of the following block leaves with the result (kernels/conv3d_h2.h, POOL): 16 x 16 regions, statistics and records present
#define MH_H2_POOL(RES_) hipLaunchKernelGGL((conv3d_k3_h2_kernel<true, true, RES_, false, false, false, true>), grid, dim3(512), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, pool->mx, pool->mn, pool->n_stride)
            if (in.C <= 2 * H2_KC) MH_H2_POOL(true); else MH_H2_POOL(false);
#undef MH_H2_POOL
            return launched("conv3d_k3_h2_pool");
        }
        if (accumulate) {       // out += conv (kernels/conv3d_h2.h, ACC): statistics and input records present (checked by the entry point)
#define MH_H2_ACC(RES_, WIDE_, C16_) hipLaunchKernelGGL((conv3d_k3_h2_kernel<true, true, RES_, WIDE_, C16_, true>), grid, dim3(512), 0, s, in, wq, tail, bias, out, stats, bxn, byn, zc, nblk, (float*)nullptr, (float*)nullptr, 0LL)
            if (c16) {          // round 6: SegResNet's residual joins at its 16-channel level (x += conv2(...), statistics of the sum)
                if (in.C <= 2 * H2_KC) { if (wide) MH_H2_ACC(true, true, true); else MH_H2_ACC(true, false, true); }
                else { if (wide) MH_H2_ACC(false, true, true); else MH_H2_ACC(false, false, true); }
            } else {
                if (in.C <= 2 * H2_KC) { if (wide) MH_H2_ACC(true, true, false); else MH_H2_ACC(true, false, false); }
                else { if (wide) MH_H2_ACC(false, true, false); else MH_H2_ACC(false, false, false); }
            }
#undef MH_H2_ACC
            return launched("conv3d_k3_h2_acc");
        }
        if (c16) {
            if (in.C <= 2 * H2_KC) { if (wide) MH_H2_LAUNCH(true, true, true) else MH_H2_LAUNCH(true, false, true) }
            else { if (wide) MH_H2_LAUNCH(false, true, true) else MH_H2_LAUNCH(false, false, true) }
        } else {
            if (in.C <= 2 * H2_KC) { if (wide) MH_H2_LAUNCH(true, true, false) else MH_H2_LAUNCH(true, false, false) }
            else { if (wide) MH_H2_LAUNCH(false, true, false) else MH_H2_LAUNCH(false, false, false) }
        }
#undef MH_H2_LAUNCH
        return launched(c16 ? "conv3d_k3_h2c" : "conv3d_k3_h2");
    }
    if (cfg == MH_CFG_WINO2D) {
        if (!mh_conv3d_k3_accepts(cfg, in.C, out.C) || in.H % 2 || in.W % 8)
            return fail(MH_ERR_ARG, "conv3d_k3: in-plane Winograd needs Cin %% 8 == 0, Cout %% 16 == 0, even H and W %% 8 == 0 (got %d -> %d, %dx%dx%d)",
                        in.C, out.C, in.D, in.H, in.W);
        if (!aligned(out.data, 16) || out.n_stride % 4 || !aligned(packed_w, 16))
            return fail(MH_ERR_ARG, "conv3d_k3: in-plane Winograd needs 16-byte aligned output and weights");
        const int bxn = cdiv(out.W, W2_B), byn = cdiv(out.H, W2_B), zc = wino2d_zchunk(out.D, out.H, out.W);
        const unsigned nblk = (unsigned)(bxn * byn * cdiv(out.D, zc));
        const long long total = (long long)nblk * (out.C / W2_CN) * out.N;
        if (total > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "conv3d_k3: problem too large for one launch");
        const dim3 grid((unsigned)total);
        // conv3d_wino2p.h: two 256-register waves per SIMD, the Winograd positions split over the pair (17.7-17.9 ms for 32 -> 32 ch @ 96^3 x 64 windows;
        // the one-wave form of round 1 and the matrix-wave + staging-wave form measured 18.7-18.9 ms, profiles/r02_wino2_impls.json, and were removed)
        if (stats && in.nrm) hipLaunchKernelGGL((conv3d_k3_wino2p_kernel<true, true>), grid, dim3(512), 0, s, in, packed_w, bias, out, stats, bxn, byn, zc, nblk);
        else if (stats) hipLaunchKernelGGL((conv3d_k3_wino2p_kernel<true, false>), grid, dim3(512), 0, s, in, packed_w, bias, out, stats, bxn, byn, zc, nblk);
        else if (in.nrm) hipLaunchKernelGGL((conv3d_k3_wino2p_kernel<false, true>), grid, dim3(512), 0, s, in, packed_w, bias, out, stats, bxn, byn, zc, nblk);
        else hipLaunchKernelGGL((conv3d_k3_wino2p_kernel<false, false>), grid, dim3(512), 0, s, in, packed_w, bias, out, stats, bxn, byn, zc, nblk);
        return launched("conv3d_k3_wino2p");
    }
    if (cfg == 0) {
        if (stats) return fail(MH_ERR_ARG, "conv3d_k3: the direct kernel emits no statistics");
        constexpr int COT = 16;
        const dim3 grid(blocks_for((long long)out.D * out.H * out.W), (unsigned)cdiv(out.C, COT), (unsigned)out.N);
        hipLaunchKernelGGL((conv3d_k3_direct_kernel<COT>), grid, dim3(256), 0, s, in, packed_w, bias, out);
        return launched("conv3d_k3_direct");
    }
    const CfgInfo& k = kCfg[cfg];
    (void)k;
    if (!mh_conv3d_k3_accepts(cfg, in.C, out.C))
        return fail(MH_ERR_ARG, "conv3d_k3: configuration %d does not take Cin=%d Cout=%d", cfg, in.C, out.C);
    if (!aligned(packed_w, 16)) return fail(MH_ERR_ARG, "conv3d_k3: packed weights must be 16-byte aligned");
    switch (cfg) {
        case 1: launch_mfma<Cfg1>(in, packed_w, bias, out, stats, s); break;
        case 2: launch_mfma<Cfg2>(in, packed_w, bias, out, stats, s); break;
        case 3: launch_mfma<Cfg3>(in, packed_w, bias, out, stats, s); break;
        case 4: launch_mfma<Cfg4>(in, packed_w, bias, out, stats, s); break;
        case 5: launch_mfma<Cfg5>(in, packed_w, bias, out, stats, s); break;
        case 6: launch_mfma<Cfg6>(in, packed_w, bias, out, stats, s); break;
        case 7: launch_mfma<Cfg7>(in, packed_w, bias, out, stats, s); break;
        case 8: launch_mfma<Cfg8>(in, packed_w, bias, out, stats, s); break;
        case 9: launch_mfma<Cfg9>(in, packed_w, bias, out, stats, s); break;
        case 10: launch_mfma<Cfg10>(in, packed_w, bias, out, stats, s); break;
        case 11: launch_mfma<Cfg11>(in, packed_w, bias, out, stats, s); break;
        case 12: launch_mfma<Cfg12>(in, packed_w, bias, out, stats, s); break;
        case 13: launch_mfma<Cfg13>(in, packed_w, bias, out, stats, s); break;
        case 14: launch_mfma<Cfg14>(in, packed_w, bias, out, stats, s); break;
    }
    return launched("conv3d_k3_mfma");
}

// ------------------------------------------------------------------------------------------ instance norm
int mh_instnorm_stat_tiles(int D, int H, int W) {
    return (int)(((long long)D * H * W + STAT_CHUNK - 1) / STAT_CHUNK);
}

int mh_instnorm_stats_f32(const mh_tensor5* x_, float* stats, void* stream) {
    if (!dense_ok(x_) || !stats) return fail(MH_ERR_ARG, "instnorm_stats: bad argument");
    const Tensor x = from_c(*x_);
    const int tiles = mh_instnorm_stat_tiles(x.D, x.H, x.W);
    hipLaunchKernelGGL(instnorm_stats_kernel, dim3((unsigned)tiles, (unsigned)x.C, (unsigned)x.N), dim3(256), 0,
                       (hipStream_t)stream, x, stats, tiles);
    return launched("instnorm_stats");
}

int mh_instnorm_finalize_f32(const float* stats, int tiles, int N, int C, const float* gamma, const float* beta, float eps,
                             float slope, float* nrm, int64_t nrm_n_stride, void* stream) {
    if (!stats || !nrm || tiles < 1 || N < 1 || C < 1 || nrm_n_stride < 4LL * C) return fail(MH_ERR_ARG, "instnorm_finalize: bad argument");
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((unsigned)C, (unsigned)N), dim3(64), 0, (hipStream_t)stream, stats, tiles,
                       C, 1, gamma, beta, eps, slope, nrm, (long long)nrm_n_stride);
    return launched("instnorm_finalize");
}

int mh_groupnorm_finalize_f32(const float* stats, int tiles, int N, int C, int groups, const float* gamma, const float* beta, float eps,
                              float slope, float* nrm, int64_t nrm_n_stride, void* stream) {
    if (!stats || !nrm || tiles < 1 || N < 1 || C < 1 || groups < 1 || C % groups || nrm_n_stride < 4LL * C)
        return fail(MH_ERR_ARG, "groupnorm_finalize: bad argument (channels must be divisible by groups)");
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3((unsigned)groups, (unsigned)N), dim3(64), 0, (hipStream_t)stream, stats, tiles,
                       C, C / groups, gamma, beta, eps, slope, nrm, (long long)nrm_n_stride);
    return launched("groupnorm_finalize");
}

// ------------------------------------------------------------------------------------------ pool / deconv / 1x1
int mh_nrm_identity_f32(float* nrm, int N, int C, int64_t nrm_n_stride, void* stream) {
    if (!nrm || N < 1 || C < 1 || nrm_n_stride < 4LL * C || nrm_n_stride % 4 || !aligned(nrm, 16)) return fail(MH_ERR_ARG, "nrm_identity: bad argument");
    hipLaunchKernelGGL(nrm_identity_kernel, dim3(blocks_for((long long)N * C)), dim3(256), 0, (hipStream_t)stream, nrm, N, C, (long long)nrm_n_stride);
    return launched("nrm_identity");
}
// records an output view names (magnitude bounds, kernels/common.h) are written as float4 / by atomics
static bool out_records_ok(const mh_tensor5* t) { return !t->nrm || (aligned(t->nrm, 16) && t->nrm_n_stride % 4 == 0 && t->nrm_n_stride >= 4LL * t->C); }

int mh_maxpool2_f32(const mh_tensor5* in_, const mh_tensor5* out_, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_)) return fail(MH_ERR_ARG, "maxpool2: bad tensor");
    if (!out_records_ok(out_)) return fail(MH_ERR_ARG, "maxpool2: the output view's records must be 16-byte aligned [N][>= C][4] floats");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    // out.D == in.D: plane-wise pooling (MaxPool2d of a 2-D network on this engine); in.D == 1 can only mean that
    if (in.N != out.N || in.C != out.C || (out.D != in.D / 2 && out.D != in.D) || out.D < 1 || out.H != in.H / 2 || out.W != in.W / 2)
        return fail(MH_ERR_ARG, "maxpool2: output must be floor(input/2) (or keep the first axis: plane-wise pooling)");
    const dim3 grid(blocks_for((long long)out.D * out.H * out.W), (unsigned)out.C, (unsigned)out.N);
    const bool pair = in.W % 2 == 0 && aligned(in.data, 8) && in.n_stride % 2 == 0;
    const bool planar = out.D == in.D && out.D != in.D / 2;      // in.D == 1 (a 2-D network's plane): 1 / 2 == 0, so this is unambiguous
    if (planar) {
        if (pair) hipLaunchKernelGGL((maxpool2_kernel<true, 1>), grid, dim3(256), 0, (hipStream_t)stream, in, out);
        else hipLaunchKernelGGL((maxpool2_kernel<false, 1>), grid, dim3(256), 0, (hipStream_t)stream, in, out);
    } else {
        if (pair) hipLaunchKernelGGL((maxpool2_kernel<true, 2>), grid, dim3(256), 0, (hipStream_t)stream, in, out);
        else hipLaunchKernelGGL((maxpool2_kernel<false, 2>), grid, dim3(256), 0, (hipStream_t)stream, in, out);
    }
    return launched("maxpool2");
}

int mh_deconv_k2s2_f32(const mh_tensor5* in_, const float* w, const float* bias, const mh_tensor5* out_, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_) || !w) return fail(MH_ERR_ARG, "deconv_k2s2: bad tensor");
    if (!out_records_ok(out_)) return fail(MH_ERR_ARG, "deconv_k2s2: the output view's records must be 16-byte aligned [N][>= C][4] floats");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || out.D != 2 * in.D || out.H != 2 * in.H || out.W != 2 * in.W)
        return fail(MH_ERR_ARG, "deconv_k2s2: output must be 2x input");
    if (!aligned(out.data, 8) || out.n_stride % 2) return fail(MH_ERR_ARG, "deconv_k2s2: output must be 8-byte aligned");
    const unsigned nb = blocks_for((long long)in.D * in.H * in.W);
#ifdef MH_DEV_KNOBS
    {       // measurement variants (tools/deconv_bench.py): 1 = 8 couts per thread (the round-2 form); 2 / 3 = two voxels per thread, 16-byte stores (3: cout group fastest); 4 = 8 couts, cout group fastest
        const int var = knob_int("MONAI_AMD_DECONV_VAR", 0);
        const bool x2ok = in.W % 2 == 0 && out.C % 4 == 0 && aligned(out.data, 16) && out.n_stride % 4 == 0 && aligned(in.data, 8) && in.n_stride % 2 == 0;
        const unsigned nb2 = blocks_for((long long)in.D * in.H * in.W / 2);
        if (var == 1 && out.C % 8 == 0) { hipLaunchKernelGGL((deconv_k2s2_kernel<8, true>), dim3(nb, (unsigned)(out.C / 8), (unsigned)out.N), dim3(256), 0, (hipStream_t)stream, in, w, bias, out); return launched("deconv_k2s2"); }
        if (var == 2 && x2ok) { hipLaunchKernelGGL((deconv_k2s2_x2_kernel<4, false>), dim3(nb2, (unsigned)(out.C / 4), (unsigned)out.N), dim3(256), 0, (hipStream_t)stream, in, w, bias, out); return launched("deconv_k2s2"); }
        if (var == 3 && x2ok) { hipLaunchKernelGGL((deconv_k2s2_x2_kernel<4, true>), dim3((unsigned)(out.C / 4), nb2, (unsigned)out.N), dim3(256), 0, (hipStream_t)stream, in, w, bias, out); return launched("deconv_k2s2"); }
    }
#endif
    // 4 output channels per thread: 2.13 / 0.46 / 0.24 ms on the BasicUNet decoder shapes against 2.42 / 0.56 / 0.27 with 8 (fewer channel planes written at once,
    // twice the workgroups; two voxels per thread with 16-byte stores measured slower: profiles/r03_deconv_variants.json)
    if (out.C % 4 == 0) hipLaunchKernelGGL((deconv_k2s2_kernel<4, true>), dim3(nb, (unsigned)(out.C / 4), (unsigned)out.N), dim3(256), 0, (hipStream_t)stream, in, w, bias, out);
    else hipLaunchKernelGGL((deconv_k2s2_kernel<4, false>), dim3(nb, (unsigned)cdiv(out.C, 4), (unsigned)out.N), dim3(256), 0, (hipStream_t)stream, in, w, bias, out);
    return launched("deconv_k2s2");
}

// ---- ConvTranspose3d k2 s2 on the fp16 matrix cores in split precision (kernels/deconv_h2.h): one GEMM with (cout, parity) rows, stored pixel-shuffled
int mh_deconv_k2s2_h2_accepts(int Cin, int Cout, int D, int H, int W) {
    return Cin >= 16 && Cin % 16 == 0 && Cin <= DH_CIN_MAX && Cout >= 16 && Cout % 16 == 0 && D >= 1 && H >= 1 && W >= 1 && (long long)D * H * W < 0x10000000LL;
}
int64_t mh_deconv_k2s2_h2_packed_floats(int Cin, int Cout) {
    if (!(Cin >= 16 && Cin % 16 == 0 && Cout >= 16 && Cout % 16 == 0)) return fail(MH_ERR_ARG, "deconv_k2s2_h2: needs Cin %% 16 == 0, Cout %% 16 == 0");
    return (int64_t)Cin * Cout * 8 + H2_TAIL;              // two fp16 pieces per weight + {1 / scale, scale}
}
int mh_deconv_k2s2_h2_pack_f32(const float* w, int Cin, int Cout, float* packed, void* stream) {
    if (!w || !packed) return fail(MH_ERR_ARG, "deconv_k2s2_h2_pack: null pointer");
    const int64_t total = mh_deconv_k2s2_h2_packed_floats(Cin, Cout);
    if (total < 0) return (int)total;
    if (!aligned(packed, 16)) return fail(MH_ERR_ARG, "deconv_k2s2_h2_pack: 16-byte aligned packed buffer required");
    float* tail = packed + (total - H2_TAIL);
    hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)Cin * Cout * 8, tail);
    hipLaunchKernelGGL(deconv_k2s2_h2_pack_kernel, dim3(blocks_for((long long)Cin * Cout)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, Cout % 32 == 0 ? 8 : 4,
                       reinterpret_cast<_Float16*>(packed), tail);
    return launched("deconv_k2s2_h2_pack");
}
int mh_deconv_k2s2_h2_f32(const mh_tensor5* in_, const float* packed, const float* bias, const mh_tensor5* out_, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_) || !packed) return fail(MH_ERR_ARG, "deconv_k2s2_h2: bad tensor");
    if (!out_records_ok(out_)) return fail(MH_ERR_ARG, "deconv_k2s2_h2: the output view's records must be 16-byte aligned [N][>= C][4] floats");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || out.D != 2 * in.D || out.H != 2 * in.H || out.W != 2 * in.W) return fail(MH_ERR_ARG, "deconv_k2s2_h2: output must be 2x input");
    if (!mh_deconv_k2s2_h2_accepts(in.C, out.C, in.D, in.H, in.W))
        return fail(MH_ERR_UNSUPPORTED, "deconv_k2s2_h2: needs Cin %% 16 == 0 (<= %d), Cout %% 16 == 0 (got %d -> %d)", DH_CIN_MAX, in.C, out.C);
    if (!in.nrm) return fail(MH_ERR_ARG, "deconv_k2s2_h2: the input must carry records with magnitude bounds (the split-precision kernels scale their input by them)");
    if (!aligned(in.nrm, 16) || !aligned(out.data, 8) || out.n_stride % 2 || !aligned(packed, 16)) return fail(MH_ERR_ARG, "deconv_k2s2_h2: 16-byte aligned records and weights, 8-byte aligned output required");
    if (out.N > 65535) return fail(MH_ERR_UNSUPPORTED, "deconv_k2s2_h2: at most 65535 samples per launch");
    const uint4* wq = reinterpret_cast<const uint4*>(packed);
    const float* tail = packed + (mh_deconv_k2s2_h2_packed_floats(in.C, out.C) - H2_TAIL);
    const unsigned nb = (unsigned)(((long long)in.D * in.H * in.W + 127) / 128);
    hipStream_t s = (hipStream_t)stream;
    if (out.C % 32 == 0) hipLaunchKernelGGL((deconv_k2s2_h2_kernel<8>), dim3(nb, (unsigned)(out.C / 32), (unsigned)out.N), dim3(256), 0, s, in, wq, tail, bias, out);
    else hipLaunchKernelGGL((deconv_k2s2_h2_kernel<4>), dim3(nb, (unsigned)(out.C / 16), (unsigned)out.N), dim3(256), 0, s, in, wq, tail, bias, out);
    return launched("deconv_k2s2_h2");
}

// ---- UpCat's "up" half as one composite transposed convolution k4 s2 p1 added to the convolution's skip half (kernels/upconv_h2.h)
static inline int upconv_tiles(int Hl, int Wl) { return cdiv(Wl, UC_TX) * cdiv(Hl, UC_TY); }
static inline int upconv_zchunk(int Dl, int Hl, int Wl) {          // a pure function of the extents (the statistics record count depends on it), conv3d_h2's rule
    int nchunk = cdiv(16, upconv_tiles(Hl, Wl));
    if (nchunk > Dl / 12) nchunk = Dl / 12;
    if (nchunk < 1) nchunk = 1;
    return cdiv(Dl, nchunk);
}
int mh_upconv_k4s2_accepts(int Cin, int Cout, int Dl, int Hl, int Wl) {
    return Cin == UC_CIN && Cout >= 32 && Cout % 32 == 0 && Dl >= 1 && Hl >= 1 && Wl >= 4 && Wl % 4 == 0 && h2_fits(2 * Dl, 2 * Hl, 2 * Wl);
}
int64_t mh_upconv_k4s2_packed_floats(int Cin, int Cout) {
    if (Cin != UC_CIN || Cout < 32 || Cout % 32) return fail(MH_ERR_ARG, "upconv_k4s2: needs Cin == 32, Cout %% 32 == 0");
    return (int64_t)(Cout / 32) * 4 * UC_WB * 4 + H2_TAIL;
}
int mh_upconv_k4s2_stat_tiles(int Dl, int Hl, int Wl) { return upconv_tiles(Hl, Wl) * 4 * cdiv(Dl, upconv_zchunk(Dl, Hl, Wl)); }
int mh_upconv_k4s2_pack_f32(const float* w4, int Cin, int Cout, float* packed, void* stream) {
    if (!w4 || !packed) return fail(MH_ERR_ARG, "upconv_k4s2_pack: null pointer");
    const int64_t total = mh_upconv_k4s2_packed_floats(Cin, Cout);
    if (total < 0) return (int)total;
    float* tail = packed + (total - H2_TAIL);
    hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w4, (long long)Cin * Cout * 64, tail);
    hipLaunchKernelGGL(upconv_k4s2_pack_kernel, dim3(blocks_for((long long)Cin * Cout)), dim3(256), 0, (hipStream_t)stream, w4, Cin, Cout, reinterpret_cast<_Float16*>(packed), tail);
    return launched("upconv_k4s2_pack");
}
int mh_upconv_k4s2_f32(const mh_tensor5* low_, const float* packed, const float* bias_table, const mh_tensor5* out_, int accumulate, float* stats, void* stream) {
    if (!dense_ok(low_) || !dense_ok(out_) || !packed || !bias_table) return fail(MH_ERR_ARG, "upconv_k4s2: bad tensor");
    if (stats && !accumulate) return fail(MH_ERR_ARG, "upconv_k4s2: statistics are those of the sum (accumulate != 0); the writing form leaves them to the accumulating convolution");
    const Tensor low = from_c(*low_), out = from_c(*out_);
    if (low.N != out.N || out.D != 2 * low.D || out.H != 2 * low.H || out.W != 2 * low.W) return fail(MH_ERR_ARG, "upconv_k4s2: the output must be 2x the input");
    if (!mh_upconv_k4s2_accepts(low.C, out.C, low.D, low.H, low.W))
        return fail(MH_ERR_UNSUPPORTED, "upconv_k4s2: needs Cin == 32, Cout %% 32 == 0, input W %% 4 == 0, D*H*W of the output < 2^24 voxels (got %d -> %d, %dx%dx%d)", low.C, out.C, low.D, low.H, low.W);
    if (!aligned(out.data, 16) || out.n_stride % 4 || !aligned(packed, 16)) return fail(MH_ERR_ARG, "upconv_k4s2: 16-byte aligned output and weights required");
    const int txn = cdiv(low.W, UC_TX), tyn = cdiv(low.H, UC_TY), zc = upconv_zchunk(low.D, low.H, low.W);
    const unsigned nblk = (unsigned)(txn * tyn * 4 * cdiv(low.D, zc));
    const long long total = (long long)nblk * (out.C / 32) * out.N;
    if (total > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "upconv_k4s2: problem too large for one launch");
    const uint4* wq = reinterpret_cast<const uint4*>(packed);
    const float* tail = packed + (mh_upconv_k4s2_packed_floats(low.C, out.C) - H2_TAIL);
    hipStream_t s = (hipStream_t)stream;
    const dim3 g((unsigned)total), bl(UC_NT);
    if (!accumulate) hipLaunchKernelGGL((upconv_k4s2_h2_kernel<false, false>), g, bl, 0, s, low, wq, tail, bias_table, out, stats, txn, tyn, zc, nblk);
    else if (stats) hipLaunchKernelGGL((upconv_k4s2_h2_kernel<true, true>), g, bl, 0, s, low, wq, tail, bias_table, out, stats, txn, tyn, zc, nblk);
    else hipLaunchKernelGGL((upconv_k4s2_h2_kernel<false, true>), g, bl, 0, s, low, wq, tail, bias_table, out, stats, txn, tyn, zc, nblk);
    return launched("upconv_k4s2");
}

// ---- Conv3d k3 s2 p1 on the fp16 matrix cores in split precision (kernels/conv3d_s2_h2.h): phase-split pass + GEMM over the 8 parity phases
static inline int s2_zchunk(int Do, int tiles) {          // a pure function of the extents (the statistics record count depends on it)
    int nchunk = cdiv(32, tiles);
    if (nchunk > Do / 8) nchunk = Do / 8;
    if (nchunk < 1) nchunk = 1;
    return cdiv(Do, nchunk);
}
int mh_conv3d_k3s2_accepts(int Cin, int Cout, int D, int H, int W) {
    if (!(Cin >= 16 && Cin % 16 == 0 && Cout >= 32 && Cout % 32 == 0 && D >= 2 && H >= 2 && W >= 2) || (D & 1) || (H & 1) || (W & 1)) return 0;
    const long long ovol = (long long)(D / 2) * (H / 2) * (W / 2);
    // 32-bit byte offsets: 64 output planes of a workgroup, 16 input channel planes of a step (fused form), 4 piece volumes of a phase (split form)
    return ovol * 64 * 4 < 0x80000000LL && (long long)D * H * W * 64 < 0x80000000LL && D <= 65535;
}
int64_t mh_conv3d_k3s2_packed_floats(int Cin, int Cout) {
    if (!(Cin >= 16 && Cin % 16 == 0 && Cout >= 32 && Cout % 32 == 0)) return fail(MH_ERR_ARG, "conv3d_k3s2: needs Cin %% 16 == 0, Cout %% 32 == 0");
    return (int64_t)Cout * Cin * 27 + H2_TAIL;
}
int64_t mh_conv3d_k3s2_workspace_floats(int N, int Cin, int D, int H, int W) {
    if (N < 1 || Cin < 16 || Cin % 16 || D < 2 || H < 2 || W < 2) return fail(MH_ERR_ARG, "conv3d_k3s2_workspace: bad argument");
    return (int64_t)N * Cin * D * H * W + ((N + 3) / 4) * 4;
}
int mh_conv3d_k3s2_stat_tiles(int D, int H, int W) {
    const S2Tile t = s2_tile(H / 2, W / 2);
    const int tiles = t.tyn * t.txn;
    return tiles * cdiv(D / 2, s2_zchunk(D / 2, tiles));
}
int mh_conv3d_k3s2_pack_f32(const float* w, int Cin, int Cout, float* packed, void* stream) {
    if (!w || !packed) return fail(MH_ERR_ARG, "conv3d_k3s2_pack: null pointer");
    const int64_t total = mh_conv3d_k3s2_packed_floats(Cin, Cout);
    if (total < 0) return (int)total;
    if (!aligned(packed, 16)) return fail(MH_ERR_ARG, "conv3d_k3s2_pack: 16-byte aligned packed buffer required");
    float* tail = packed + (total - H2_TAIL);
    hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)Cin * Cout * 27, tail);
    hipLaunchKernelGGL(conv3d_k3s2_h2_pack_kernel, dim3(blocks_for((long long)Cin * Cout)), dim3(256), 0, (hipStream_t)stream, w, Cin, Cout, Cout % 64 == 0 ? 2 : 1,
                       reinterpret_cast<_Float16*>(packed), tail);
    return launched("conv3d_k3s2_pack");
}
int mh_conv3d_k3s2_f32(const mh_tensor5* in_, const float* packed, const float* bias, const mh_tensor5* out_, float* workspace, float* stats, int fused, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_) || !packed || (!workspace && !fused)) return fail(MH_ERR_ARG, "conv3d_k3s2: bad argument");
    if (fused && in_->C > S2_CIN_MAX) return fail(MH_ERR_UNSUPPORTED, "conv3d_k3s2: the fused form keeps the records of at most %d input channels", S2_CIN_MAX);
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || out.D * 2 != in.D || out.H * 2 != in.H || out.W * 2 != in.W) return fail(MH_ERR_ARG, "conv3d_k3s2: the output must be half the (even) input extents");
    if (!mh_conv3d_k3s2_accepts(in.C, out.C, in.D, in.H, in.W))
        return fail(MH_ERR_UNSUPPORTED, "conv3d_k3s2: needs Cin %% 16 == 0, Cout %% 32 == 0, even extents (got %d -> %d, %dx%dx%d)", in.C, out.C, in.D, in.H, in.W);
    if (!in.nrm) return fail(MH_ERR_ARG, "conv3d_k3s2: the input must carry records with magnitude bounds (the split-precision kernels scale their input by them)");
    if (!aligned(in.nrm, 16) || !aligned(out.data, 16) || out.n_stride % 4 || !aligned(packed, 16) || (!fused && !aligned(workspace, 16)))
        return fail(MH_ERR_ARG, "conv3d_k3s2: 16-byte aligned records, output, weights and workspace required");
    hipStream_t s = (hipStream_t)stream;
    uint4* xs = fused ? nullptr : reinterpret_cast<uint4*>(workspace);
    int* expo = fused ? nullptr : reinterpret_cast<int*>(workspace + (long long)in.N * in.C * in.D * in.H * in.W);
    if (!fused) {       // the split pass in sample ranges whose (sample, channel group) count fits the grid's z extent
        const int per = in.C / 8, nmax = 65535 / per;
        if (nmax < 1) return fail(MH_ERR_UNSUPPORTED, "conv3d_k3s2: more than 524280 input channels");
        const long long ws_n = (long long)in.C * in.D * in.H * in.W / 4;          // uint4 per sample
        for (int n0 = 0; n0 < in.N; n0 += nmax) {
            Tensor part = in;
            part.N = in.N - n0 < nmax ? in.N - n0 : nmax;
            part.data = in.data + (long long)n0 * in.n_stride;
            part.nrm = in.nrm + (long long)n0 * in.nrm_n_stride;
            hipLaunchKernelGGL(conv3d_s2_split_kernel, dim3(blocks_for((long long)in.H * in.W), (unsigned)in.D, (unsigned)(part.N * per)), dim3(256), 0, s, part, xs + n0 * ws_n, expo + n0);
        }
    }
    const S2Tile t = s2_tile(out.H, out.W);
    const int tiles = t.tyn * t.txn, zc = s2_zchunk(out.D, tiles);
    const unsigned nblk = (unsigned)(tiles * cdiv(out.D, zc));
    const int ncgw = out.C % 64 == 0 ? 2 : 1;
    const long long total = (long long)nblk * (out.C / (32 * ncgw)) * out.N;
    if (total > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "conv3d_k3s2: problem too large for one launch");
    const uint4* wq = reinterpret_cast<const uint4*>(packed);
    const float* tail = packed + (mh_conv3d_k3s2_packed_floats(in.C, out.C) - H2_TAIL);
    const dim3 g((unsigned)total), bl(S2_NT);
#define MH_S2_LAUNCH(NCG_, F_)                                                                                                                          \
    {                                                                                                                                                   \
        if (stats) hipLaunchKernelGGL((conv3d_k3s2_h2_kernel<NCG_, true, F_>), g, bl, 0, s, in, xs, expo, wq, tail, bias, out, stats, t.tr, t.tc, t.txn, t.tyn, zc, nblk);  \
        else hipLaunchKernelGGL((conv3d_k3s2_h2_kernel<NCG_, false, F_>), g, bl, 0, s, in, xs, expo, wq, tail, bias, out, stats, t.tr, t.tc, t.txn, t.tyn, zc, nblk);        \
    }
    if (fused) { if (ncgw == 2) MH_S2_LAUNCH(2, true) else MH_S2_LAUNCH(1, true) }
    else { if (ncgw == 2) MH_S2_LAUNCH(2, false) else MH_S2_LAUNCH(1, false) }
#undef MH_S2_LAUNCH
    return launched("conv3d_k3s2");
}

template <int VEC, bool STATS>
static void launch_1x1(int co, unsigned nb, unsigned nbatch, hipStream_t s, const Tensor& in, const float* w, const float* bias,
                       const Tensor& out, int co0, float* stats) {
#define MH_1X1_CASE(CO)                                                                                               \
    case CO:                                                                                                          \
        hipLaunchKernelGGL((conv1x1_kernel<CO, VEC, STATS>), dim3(nb, nbatch), dim3(256), 0, s, in, w, bias, out, co0, stats, (int)nb);      \
        break;
    switch (co) {
        MH_1X1_CASE(1) MH_1X1_CASE(2) MH_1X1_CASE(3) MH_1X1_CASE(4)
        MH_1X1_CASE(5) MH_1X1_CASE(6) MH_1X1_CASE(7) MH_1X1_CASE(8) MH_1X1_CASE(16)
    }
#undef MH_1X1_CASE
}

// one {count, mean, M2} record per (n, cout, workgroup): the tile count of a stats-fused 1x1x1 convolution (mh_conv1x1_stats_f32) on a D x H x W plane
int mh_conv1x1_stat_tiles(int D, int H, int W) {
    const long long DHW = (long long)D * H * W;
    return (int)blocks_for(DHW % 4 == 0 ? DHW / 4 : DHW);
}

static int conv1x1_impl(const mh_tensor5* in_, const float* w, const float* bias, const mh_tensor5* out_, float* stats, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_) || !w) return fail(MH_ERR_ARG, "conv1x1: bad tensor");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || in.D != out.D || in.H != out.H || in.W != out.W) return fail(MH_ERR_ARG, "conv1x1: shape mismatch");
    const long long DHW = (long long)in.D * in.H * in.W;
    const bool v4 = DHW % 4 == 0 && aligned(in.data, 16) && aligned(out.data, 16) && in.n_stride % 4 == 0 && out.n_stride % 4 == 0;
    if (stats && !v4 && DHW % 4 == 0) return fail(MH_ERR_UNSUPPORTED, "conv1x1_stats: tensors must be 16-byte aligned when D*H*W is a multiple of 4 (the tile count assumes it)");
    const unsigned nb = blocks_for(v4 ? DHW / 4 : DHW);
    for (int co0 = 0; co0 < out.C;) {      // 16 output channels per pass where they exist (the input is read once per pass)
        const int left = out.C - co0;
        const int co = left >= 16 ? 16 : left < 8 ? left : 8;
        if (stats) {
            if (v4) launch_1x1<4, true>(co, nb, (unsigned)in.N, (hipStream_t)stream, in, w, bias, out, co0, stats);
            else launch_1x1<1, true>(co, nb, (unsigned)in.N, (hipStream_t)stream, in, w, bias, out, co0, stats);
        } else {
            if (v4) launch_1x1<4, false>(co, nb, (unsigned)in.N, (hipStream_t)stream, in, w, bias, out, co0, nullptr);
            else launch_1x1<1, false>(co, nb, (unsigned)in.N, (hipStream_t)stream, in, w, bias, out, co0, nullptr);
        }
        co0 += co;
    }
    return launched("conv1x1");
}

int mh_conv1x1_f32(const mh_tensor5* in_, const float* w, const float* bias, const mh_tensor5* out_, void* stream) { return conv1x1_impl(in_, w, bias, out_, nullptr, stream); }

int mh_conv1x1_stats_f32(const mh_tensor5* in_, const float* w, const float* bias, const mh_tensor5* out_, float* stats, void* stream) {
    if (!stats) return fail(MH_ERR_ARG, "conv1x1_stats: null statistics buffer");
    return conv1x1_impl(in_, w, bias, out_, stats, stream);
}

// ---- the 1x1x1 convolution with all output channels from one read of the input (kernels/conv1x1_h2.h) ----
int mh_conv1x1_h2_accepts(int Cin, int Cout, int D, int H, int W) {
    const long long DHW = (long long)D * H * W;
    return Cin >= 1 && Cin <= 16 * C1H_KS * C1H_CHUNKS && Cout >= 1 && DHW >= 4 && DHW % 4 == 0 && (DHW + 1023) / 1024 <= 0x7fffffffLL;
}
int64_t mh_conv1x1_h2_packed_floats(int Cin, int Cout) {
    if (Cin < 1 || Cin > 16 * C1H_KS * C1H_CHUNKS || Cout < 1) return fail(MH_ERR_ARG, "conv1x1_h2: 1 .. %d input channels (got %d)", 16 * C1H_KS * C1H_CHUNKS, Cin);
    return (int64_t)cdiv(Cout, 64) * cdiv(Cin, 16) * C1H_SLAB * 4 + H2_TAIL;
}
int mh_conv1x1_h2_pack_f32(const float* w, int Cout, int Cin, float* packed, void* stream) {
    if (!w || !packed) return fail(MH_ERR_ARG, "conv1x1_h2_pack: null pointer");
    const int64_t total = mh_conv1x1_h2_packed_floats(Cin, Cout);
    if (total < 0) return (int)total;
    if (!aligned(packed, 16)) return fail(MH_ERR_ARG, "conv1x1_h2_pack: 16-byte aligned packed buffer required");
    float* tail = packed + (total - H2_TAIL);
    hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)Cin * Cout, tail);
    const long long padded = (long long)cdiv(Cout, 64) * 64 * cdiv(Cin, 16) * 16;
    hipLaunchKernelGGL(conv1x1_h2_pack_kernel, dim3(blocks_for(padded)), dim3(256), 0, (hipStream_t)stream, w, Cout, Cin, reinterpret_cast<_Float16*>(packed), tail);
    return launched("conv1x1_h2_pack");
}
int mh_conv1x1_h2_f32(const mh_tensor5* in_, const float* packed, const float* bias, const mh_tensor5* out_, float* stats, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_) || !packed) return fail(MH_ERR_ARG, "conv1x1_h2: bad tensor");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || in.D != out.D || in.H != out.H || in.W != out.W) return fail(MH_ERR_ARG, "conv1x1_h2: shape mismatch");
    if (!in.nrm) return fail(MH_ERR_ARG, "conv1x1_h2: the input must carry records with magnitude bounds (the split-precision kernels scale their input by them)");
    if (!mh_conv1x1_h2_accepts(in.C, out.C, in.D, in.H, in.W))
        return fail(MH_ERR_UNSUPPORTED, "conv1x1_h2: needs Cin <= %d and D*H*W %% 4 == 0 (got %d channels, %dx%dx%d)", 16 * C1H_KS * C1H_CHUNKS, in.C, in.D, in.H, in.W);
    if (!aligned(in.data, 16) || !aligned(out.data, 16) || in.n_stride % 4 || out.n_stride % 4 || !aligned(packed, 16))
        return fail(MH_ERR_ARG, "conv1x1_h2: 16-byte aligned tensors and weights required");
    const long long DHW = (long long)in.D * in.H * in.W;
    const int tiles = (int)((DHW + 1023) / 1024), nks = cdiv(in.C, 16);
    const float* tail = packed + (mh_conv1x1_h2_packed_floats(in.C, out.C) - H2_TAIL);
    const dim3 grid((unsigned)tiles, (unsigned)in.N);
    hipStream_t s = (hipStream_t)stream;
    for (int co0 = 0, pass = 0; co0 < out.C; co0 += 64, ++pass) {
        const uint4* wq = reinterpret_cast<const uint4*>(packed) + (long long)pass * nks * C1H_SLAB;
        const bool two = out.C - co0 > 32;
#define MH_C1H_LAUNCH(MT_, ST_) hipLaunchKernelGGL((conv1x1_h2_kernel<MT_, ST_>), grid, dim3(512), 0, s, in, wq, tail, bias, out, co0, stats, tiles)
        if (two && stats) MH_C1H_LAUNCH(2, true);
        else if (two) MH_C1H_LAUNCH(2, false);
        else if (stats) MH_C1H_LAUNCH(1, true);
        else MH_C1H_LAUNCH(1, false);
#undef MH_C1H_LAUNCH
    }
    return launched("conv1x1_h2");
}

int mh_conv1x1_windows_f32(const mh_tensor5* in_, const float* w, const float* bias, float* base, int Cout, const int64_t* place, void* stream) {
    if (!dense_ok(in_) || !w || !base || !place) return fail(MH_ERR_ARG, "conv1x1_windows: bad argument");
    const Tensor in = from_c(*in_);
    if (Cout < 1 || Cout > 8) return fail(MH_ERR_UNSUPPORTED, "conv1x1_windows: 1 .. 8 output channels (got %d)", Cout);
    const long long DHW = (long long)in.D * in.H * in.W;
    if (in.W % 4 || !aligned(in.data, 16) || in.n_stride % 4 || !aligned(base, 16)) return fail(MH_ERR_UNSUPPORTED, "conv1x1_windows: needs W %% 4 == 0 and 16-byte aligned tensors");
    for (int n = 0; n < in.N; ++n)
        for (int q = 0; q < 4; ++q)
            if (place[4 * n + q] % 4) return fail(MH_ERR_ARG, "conv1x1_windows: offsets and strides must be multiples of 4 floats");
    for (int n0 = 0; n0 < in.N; n0 += WIN_PLACE_MAX) {
        const int nn = in.N - n0 < WIN_PLACE_MAX ? in.N - n0 : WIN_PLACE_MAX;
        WinPlace pl;
        for (int i = 0; i < nn; ++i) {
            pl.off[i] = place[4 * (n0 + i)]; pl.sc[i] = place[4 * (n0 + i) + 1]; pl.sd[i] = place[4 * (n0 + i) + 2]; pl.sh[i] = place[4 * (n0 + i) + 3];
        }
        const dim3 grid(blocks_for(DHW / 4), (unsigned)nn);
#define MH_1W_CASE(CO) case CO: hipLaunchKernelGGL((conv1x1_windows_kernel<CO>), grid, dim3(256), 0, (hipStream_t)stream, in, w, bias, base, pl, n0); break;
        switch (Cout) { MH_1W_CASE(1) MH_1W_CASE(2) MH_1W_CASE(3) MH_1W_CASE(4) MH_1W_CASE(5) MH_1W_CASE(6) MH_1W_CASE(7) MH_1W_CASE(8) }
#undef MH_1W_CASE
    }
    return launched("conv1x1_windows");
}


// ------------------------------------------------------------------------------------------ resampling
static int fill_resample(ResampleArgs& a, int NC, int Di, int Hi, int Wi, int Do, int Ho, int Wo, int mode, int pad, int align_corners) {
    if (NC < 1 || Di < 1 || Hi < 1 || Wi < 1 || Do < 1 || Ho < 1 || Wo < 1) return fail(MH_ERR_ARG, "resample: bad shape");
    if (mode != RS_NEAREST && mode != RS_LINEAR) return fail(MH_ERR_UNSUPPORTED, "resample: interpolation mode %d (0 nearest, 1 linear)", mode);
    if (pad < RS_ZEROS || pad > RS_REFLECTION) return fail(MH_ERR_UNSUPPORTED, "resample: padding mode %d (0 zeros, 1 border, 2 reflection)", pad);
    a.mode = mode; a.pad = pad; a.align_corners = align_corners ? 1 : 0; a.C = NC;
    a.Di = Di; a.Hi = Hi; a.Wi = Wi; a.Do = Do; a.Ho = Ho; a.Wo = Wo;
    for (int i = 0; i < 12; ++i) a.m[i] = 0.0;
    for (int i = 0; i < 3; ++i) { a.ga[i] = 1.0; a.gb[i] = 0.0; }
    return MH_OK;
}

// Workgroups of `kernel` that are resident on the device at once (CUs x occupancy), queried once per kernel.
template <typename K> static int resident_wgs(K kernel, int threads) {
    int cus = 0, per_cu = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    return cus * per_cu;
}

// z-chunks of a streaming launch of `units` (tile, channel) columns: the count whose workgroups fill whole rounds of the
// resident slots -- the last, partly filled round of a 2.1-round launch takes as long as a full one -- discounted by the
// `prime` extra planes every chunk has to read before its first output.  Chunks keep >= min_chunk output planes.
static int stream_chunks(long long units, int Do, int slots, int prime, int min_chunk, const char* env_knob) {
    if (const char* e = knob_str(env_knob)) {                          // tuning knob (development)
        const int v = atoi(e);
        if (v >= 1 && v <= Do) return v;
    }
    const int max_chunks = cdiv(Do, min_chunk) < 1 ? 1 : cdiv(Do, min_chunk);
    int best = 1;
    double best_score = -1.0;
    for (int n = 1; n <= max_chunks; ++n) {
        const int zc = cdiv(Do, n), real = cdiv(Do, zc);
        if (real != n) continue;
        const long long nwg = units * n;
        const long long rounds = (nwg + slots - 1) / slots;
        const double score = (double)nwg / (double)(rounds * slots) * ((double)zc / (double)(zc + prime));
        if (score > best_score) { best_score = score; best = n; }
    }
    return best;
}


int64_t mh_affine_resample_workspace_bytes(int Do, int Ho, int Wo) { return (int64_t)(Do + Ho + Wo) * (int64_t)sizeof(AxisTap<double>); }

int mh_affine_resample_f32(const float* src, int NC, int Di, int Hi, int Wi, float* dst, int Do, int Ho, int Wo, const double* m,
                           int mode, int pad, int align_corners, int compute_f64, void* workspace, void* stream) {
    if (!src || !dst || !m) return fail(MH_ERR_ARG, "affine_resample: null pointer");
    ResampleArgs a;
    if (int e = fill_resample(a, NC, Di, Hi, Wi, Do, Ho, Wo, mode, pad, align_corners)) return e;
    for (int i = 0; i < 12; ++i) a.m[i] = m[i];
    hipStream_t s = (hipStream_t)stream;
    const long long ovol = (long long)Do * Ho * Wo;
    const unsigned nb = blocks_for(ovol);
    const bool i32 = fits_i32((long long)NC * ovol) && fits_i32((long long)NC * Di * Hi * Wi);
    const bool axis_aligned = m[1] == 0.0 && m[2] == 0.0 && m[4] == 0.0 && m[6] == 0.0 && m[8] == 0.0 && m[9] == 0.0;
    if (axis_aligned && workspace) {
        const unsigned tb = blocks_for(Do + Ho + Wo);
        const unsigned nblk = (unsigned)(cdiv(Wo, RS_TOX) * cdiv(Ho, RS_TOY) * cdiv(Do, RS_TOZ));
        // trilinear without reflection and a tap box that fits the LDS plane (conservative bound from the scales): the
        // z-streaming kernel; everything else (nearest, reflection, strong in-plane down-sampling) the per-block kernel
        const double sy = m[5] < 0 ? -m[5] : m[5], sx = m[10] < 0 ? -m[10] : m[10];
        const double box_bound = ((double)RZ_TOY * sy + 3.0) * ((double)RZ_TOX * sx + 3.0);
        const bool stream_ok = mode == RS_LINEAR && pad != RS_REFLECTION && box_bound <= 4096.0;
        if (stream_ok) {
            const long long tiles = (long long)cdiv(Wo, RZ_TOX) * cdiv(Ho, RZ_TOY);
            const bool small = box_bound <= 2048.0;
            // fp32 interpolation: 512 threads per workgroup (two output rows per thread instead of four, twice the waves in flight):
            // 0.253-0.267 vs 0.291-0.305 ms per 512^3 volume; fp64 interpolation (the reference default) measures the same either
            // way (0.277-0.299 ms) and keeps 256.  MONAI_AMD_RS_THREADS=256|512 overrides.
            const bool wide = knob_int("MONAI_AMD_RS_THREADS", compute_f64 ? 256 : 512) != 256;
            static int slots[2][2][2] = {{{0, 0}, {0, 0}}, {{0, 0}, {0, 0}}};     // [wide][f64][small]
            int& sl = slots[wide ? 1 : 0][compute_f64 ? 1 : 0][small ? 1 : 0];
            // round 6 (kernels/resample.h, profiles/r06_resample_variants.txt): four consecutive x per lane and one 16-byte store per row (VEC) in every form; the fp64 form
            // of 256 threads takes its source planes by LDS-DMA into a ring of three slots (RING = 3: 0.281 -> 0.270 ms at config 4; the fp32 form does not gain from it).
            // MONAI_AMD_RS_RING=0 in the development build keeps the register prefetch
            static const bool ring = knob_int("MONAI_AMD_RS_RING", 1) != 0;
            const bool use_ring = ring && compute_f64 && !wide;
#define MH_RS_STREAM(T_, NL_, NT_, TAB_)                                                                                                              \
    if (use_ring && (NT_) == 256) hipLaunchKernelGGL((separable_resample_stream_kernel<T_, NL_, 256, true, sizeof(T_) == 8 ? 3 : 0>), g, dim3(256), 0, s, src, dst, TAB_, a, zchunk, nchunk); \
    else hipLaunchKernelGGL((separable_resample_stream_kernel<T_, NL_, NT_, true, 0>), g, dim3(NT_), 0, s, src, dst, TAB_, a, zchunk, nchunk)
#define MH_RS_SLOTS(T_, NL_, NT_) ((use_ring && (NT_) == 256) ? resident_wgs(separable_resample_stream_kernel<T_, NL_, 256, true, sizeof(T_) == 8 ? 3 : 0>, 256) \
                                                             : resident_wgs(separable_resample_stream_kernel<T_, NL_, NT_, true, 0>, NT_))
            if (sl == 0) {
                if (wide) sl = compute_f64 ? (small ? MH_RS_SLOTS(double, 4, 512) : MH_RS_SLOTS(double, 8, 512)) : (small ? MH_RS_SLOTS(float, 4, 512) : MH_RS_SLOTS(float, 8, 512));
                else sl = compute_f64 ? (small ? MH_RS_SLOTS(double, 8, 256) : MH_RS_SLOTS(double, 16, 256)) : (small ? MH_RS_SLOTS(float, 8, 256) : MH_RS_SLOTS(float, 16, 256));
            }
            int nchunk = stream_chunks(tiles * NC, Do, sl, 1, 8, "MONAI_AMD_RS_CHUNKS");
            if (!compute_f64 && wide && !knob_str("MONAI_AMD_RS_CHUNKS")) {
                // round 6 (profiles/r06_resample_variants.txt): with four workgroups of the fp32 form per CU the launch wants MANY rounds of workgroups, not whole ones --
                // 512^3 -> 410 x 410 x 819: 11 chunks (2 rounds) 0.279 ms, 21: 0.256, 41: 0.237, 52: 0.236; the fp64 form is flat over the same sweep and keeps the rule above.
                // >= 7 rounds of the resident slots, z-chunks of at least 8 planes
                const long long units = tiles * NC;
                long long want = (7LL * sl + units - 1) / units;
                const int maxc = Do / 8 < 1 ? 1 : Do / 8;
                want = want < 1 ? 1 : (want > maxc ? maxc : want);
                if ((int)want > nchunk) nchunk = (int)want;
            }
            if (use_ring && !knob_str("MONAI_AMD_RS_CHUNKS")) {      // the ring form: 11 chunks 0.310 ms, 21: 0.279, 26: 0.270, 41: 0.283 -- about 4.5 rounds of the resident slots
                const long long units = tiles * NC;
                long long want = (9LL * sl + 2 * units - 1) / (2 * units);
                const int maxc = Do / 8 < 1 ? 1 : Do / 8;
                want = want < 1 ? 1 : (want > maxc ? maxc : want);
                if ((int)want > nchunk) nchunk = (int)want;
            }
            const int zchunk = cdiv(Do, nchunk);
            nchunk = cdiv(Do, zchunk);
            const long long nwg = tiles * nchunk * NC;
            if (nwg > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "affine_resample: problem too large for one launch");
            const dim3 g((unsigned)nwg);
            if (compute_f64) {
                AxisTap<double>* tab = static_cast<AxisTap<double>*>(workspace);
                hipLaunchKernelGGL((resample_axis_table_kernel<double>), dim3(tb), dim3(256), 0, s, tab, a);
                const AxisTap<double>* ct = tab;
                if (wide) { if (small) MH_RS_STREAM(double, 4, 512, ct); else MH_RS_STREAM(double, 8, 512, ct); }
                else { if (small) MH_RS_STREAM(double, 8, 256, ct); else MH_RS_STREAM(double, 16, 256, ct); }
            } else {
                AxisTap<float>* tab = static_cast<AxisTap<float>*>(workspace);
                hipLaunchKernelGGL((resample_axis_table_kernel<float>), dim3(tb), dim3(256), 0, s, tab, a);
                const AxisTap<float>* ct = tab;
                if (wide) { if (small) MH_RS_STREAM(float, 4, 512, ct); else MH_RS_STREAM(float, 8, 512, ct); }
                else { if (small) MH_RS_STREAM(float, 8, 256, ct); else MH_RS_STREAM(float, 16, 256, ct); }
            }
#undef MH_RS_SLOTS
#undef MH_RS_STREAM
            return launched("separable_resample_stream");
        }
        if (compute_f64) {
            AxisTap<double>* tab = static_cast<AxisTap<double>*>(workspace);
            hipLaunchKernelGGL((resample_axis_table_kernel<double>), dim3(tb), dim3(256), 0, s, tab, a);
            hipLaunchKernelGGL((separable_resample_lds_kernel<double>), dim3(nblk), dim3(256), 0, s, src, dst, (const AxisTap<double>*)tab, a);
        } else {
            AxisTap<float>* tab = static_cast<AxisTap<float>*>(workspace);
            hipLaunchKernelGGL((resample_axis_table_kernel<float>), dim3(tb), dim3(256), 0, s, tab, a);
            hipLaunchKernelGGL((separable_resample_lds_kernel<float>), dim3(nblk), dim3(256), 0, s, src, dst, (const AxisTap<float>*)tab, a);
        }
        return launched("separable_resample");
    }
    // general (rotated / sheared) matrices without reflection: the row-mapped kernel with compile-time mode and padding rule
    // (bit-identical to affine_resample_kernel; MONAI_AMD_RS_GENERAL=linear keeps the linear-index kernel: A/B measurements)
    const char* rs_general = knob_str("MONAI_AMD_RS_GENERAL");
    const bool rows_on = !(rs_general && !strcmp(rs_general, "linear"));
    if (rows_on && i32 && pad != RS_REFLECTION && Do <= 65535 && cdiv(Ho, 4) <= 65535) {
        const dim3 g((unsigned)cdiv(Wo, 64), (unsigned)cdiv(Ho, 4), (unsigned)Do);
#define MH_RS_ROWS(T_, M_, P_) hipLaunchKernelGGL((affine_resample_rows_kernel<T_, M_, P_>), g, dim3(256), 0, s, src, dst, a)
#define MH_RS_ROWS_T(T_)                                                                              \
        {                                                                                             \
            if (mode == RS_LINEAR) { if (pad == RS_BORDER) MH_RS_ROWS(T_, RS_LINEAR, RS_BORDER); else MH_RS_ROWS(T_, RS_LINEAR, RS_ZEROS); } \
            else { if (pad == RS_BORDER) MH_RS_ROWS(T_, RS_NEAREST, RS_BORDER); else MH_RS_ROWS(T_, RS_NEAREST, RS_ZEROS); }                 \
        }
        if (compute_f64) MH_RS_ROWS_T(double) else MH_RS_ROWS_T(float)
#undef MH_RS_ROWS_T
#undef MH_RS_ROWS
        return launched("affine_resample");
    }
    if (compute_f64) {
        if (i32) hipLaunchKernelGGL((affine_resample_kernel<double, int>), dim3(nb), dim3(256), 0, s, src, dst, a);
        else hipLaunchKernelGGL((affine_resample_kernel<double, long long>), dim3(nb), dim3(256), 0, s, src, dst, a);
    } else {
        if (i32) hipLaunchKernelGGL((affine_resample_kernel<float, int>), dim3(nb), dim3(256), 0, s, src, dst, a);
        else hipLaunchKernelGGL((affine_resample_kernel<float, long long>), dim3(nb), dim3(256), 0, s, src, dst, a);
    }
    return launched("affine_resample");
}

int mh_grid_resample_f32(const float* src, int NC, int Di, int Hi, int Wi, const void* coords, int coords_f64, const double* scale3,
                         const double* offset3, float* dst, int Do, int Ho, int Wo, int mode, int pad, int align_corners,
                         int compute_f64, void* stream) {
    if (!src || !dst || !coords) return fail(MH_ERR_ARG, "grid_resample: null pointer");
    ResampleArgs a;
    if (int e = fill_resample(a, NC, Di, Hi, Wi, Do, Ho, Wo, mode, pad, align_corners)) return e;
    for (int i = 0; i < 3; ++i) {
        if (scale3) a.ga[i] = scale3[i];
        if (offset3) a.gb[i] = offset3[i];
    }
    const unsigned nb = blocks_for((long long)Do * Ho * Wo);
    hipStream_t s = (hipStream_t)stream;
    if (coords_f64) {
        const double* g = static_cast<const double*>(coords);
        if (compute_f64) hipLaunchKernelGGL((grid_resample_kernel<double, double>), dim3(nb), dim3(256), 0, s, src, g, dst, a);
        else hipLaunchKernelGGL((grid_resample_kernel<float, double>), dim3(nb), dim3(256), 0, s, src, g, dst, a);
    } else {
        const float* g = static_cast<const float*>(coords);
        if (compute_f64) hipLaunchKernelGGL((grid_resample_kernel<double, float>), dim3(nb), dim3(256), 0, s, src, g, dst, a);
        else hipLaunchKernelGGL((grid_resample_kernel<float, float>), dim3(nb), dim3(256), 0, s, src, g, dst, a);
    }
    return launched("grid_resample");
}

// ------------------------------------------------------------------------------------------ grid_pull (monai._C)
int mh_pushpull(const void* source, const void* grid, const void* target, void* out, void* grad, int is_f64, int ndim, int B,
                int C, int X, int Y, int Z, int Xo, int Yo, int Zo, const int32_t* bound3, const int32_t* interp3,
                int extrapolate, int do_pull, int do_push, int do_count, int do_grad, int do_sgrad, int target_k, void* stream);

// the 3-D pull-only subset kept for ABI compatibility: grid (B,Xo,Yo,Zo,3)
int mh_grid_pull(const void* src, const void* grid, void* out, int is_f64, int B, int C, int X, int Y, int Z, int Xo, int Yo,
                 int Zo, const int32_t* bound3, const int32_t* interp3, int extrapolate, void* stream) {
    if (!src || !grid || !out || !bound3 || !interp3) return fail(MH_ERR_ARG, "grid_pull: null pointer");
    return mh_pushpull(src, grid, nullptr, out, nullptr, is_f64, 3, B, C, X, Y, Z, Xo, Yo, Zo, bound3, interp3, extrapolate, 1, 0, 0, 0, 0, 0, stream);
}

int mh_pushpull(const void* source, const void* grid, const void* target, void* out, void* grad, int is_f64, int ndim, int B,
                int C, int X, int Y, int Z, int Xo, int Yo, int Zo, const int32_t* bound3, const int32_t* interp3,
                int extrapolate, int do_pull, int do_push, int do_count, int do_grad, int do_sgrad, int target_k, void* stream) {
    if (!grid || !bound3 || !interp3) return fail(MH_ERR_ARG, "pushpull: null pointer");
    if (ndim < 1 || ndim > 3) return fail(MH_ERR_ARG, "pushpull: %d spatial dimensions (1, 2 or 3)", ndim);
    if (B < 1 || C < 1 || X < 1 || Y < 1 || Z < 1 || Xo < 1 || Yo < 1 || Zo < 1) return fail(MH_ERR_ARG, "pushpull: bad shape");
    if ((ndim < 3 && (Z != 1 || Zo != 1)) || (ndim < 2 && (Y != 1 || Yo != 1)))
        return fail(MH_ERR_ARG, "pushpull: axes beyond ndim must have size 1");
    const int nout = (do_pull != 0) + (do_sgrad != 0) + (do_push != 0) + (do_count != 0);
    if (nout > 1) return fail(MH_ERR_ARG, "pushpull: at most one of do_pull / do_sgrad / do_push / do_count");
    if (nout + (do_grad != 0) == 0) return MH_OK;
    if (nout && !out) return fail(MH_ERR_ARG, "pushpull: output pointer is null");
    if (do_grad && !grad) return fail(MH_ERR_ARG, "pushpull: gradient pointer is null");
    if ((do_pull || do_sgrad || do_grad) && !source) return fail(MH_ERR_ARG, "pushpull: source is null");
    if (do_push && !target) return fail(MH_ERR_ARG, "pushpull: push needs a target");
    if (target_k && !target) return fail(MH_ERR_ARG, "pushpull: target_k without a target");
    PushPullArgs a;
    a.B = B; a.C = do_count ? 1 : C; a.X = X; a.Y = Y; a.Z = Z; a.Xo = Xo; a.Yo = Yo; a.Zo = Zo;
    a.ndim = ndim; a.extrapolate = extrapolate ? 1 : 0;
    a.do_pull = do_pull ? 1 : 0; a.do_push = do_push ? 1 : 0; a.do_count = do_count ? 1 : 0; a.do_grad = do_grad ? 1 : 0;
    a.do_sgrad = do_sgrad ? 1 : 0; a.trgt_k = target_k ? ndim : 0;
    for (int d = 0; d < 3; ++d) {
        a.bound[d] = bound3[d]; a.interp[d] = interp3[d];
        if (interp3[d] < 0 || interp3[d] > 7) return fail(MH_ERR_ARG, "pushpull: interpolation order %d (0-7)", interp3[d]);
        if (d < ndim && (bound3[d] < 0 || bound3[d] > 7 || bound3[d] == GB_SLIDING))
            return fail(MH_ERR_UNSUPPORTED, "pushpull: bound type %d is not built", bound3[d]);
    }
    const bool iso = interp3[0] == interp3[1] && interp3[0] == interp3[2];
    a.path = iso && interp3[0] == 0 ? PP_NEAREST : iso && interp3[0] == 1 ? PP_LINEAR : PP_GENERIC;
    // The reference's kernels take the third axis' order from the SECOND entry (`interpolation2(info.interpolation1)`,
    // pushpull_cpu.cpp:501 and pushpull_cuda.cu:498) while `iso` above sees the real third entry: reproduced, so that
    // mixed per-axis orders give the reference's results.
    a.interp[2] = interp3[1];
    hipStream_t s = (hipStream_t)stream;
    const size_t esz = is_f64 ? 8 : 4;
    if (do_push || do_count)
        if (hipMemsetAsync(out, 0, esz * (size_t)B * (size_t)a.C * X * Y * Z, s) != hipSuccess) return fail(MH_ERR_LAUNCH, "pushpull: memset failed");
    const long long total = (long long)B * Xo * Yo * Zo;
    if (total > 0x7fffffffLL * 256) return fail(MH_ERR_UNSUPPORTED, "pushpull: problem too large for one launch");
    const unsigned nb = blocks_for(total);
    int maxo = 0;
    for (int d = 0; d < ndim; ++d) maxo = a.interp[d] > maxo ? a.interp[d] : maxo;
    const bool pull_only = do_pull && !do_grad;
    const bool scatter = do_push || do_count;
    PushPullArgs ag = a;                   // the gathering part of the launch (everything but push / count)
    ag.do_push = 0; ag.do_count = 0;
    const bool gather = ag.do_pull || ag.do_sgrad || ag.do_grad;
#define MH_PP_LAUNCH1(T_, PATH_, NT_, PULL_)                                                                              \
    hipLaunchKernelGGL((pushpull_kernel<T_, PATH_, NT_, PULL_>), dim3(nb), dim3(256), 0, s, static_cast<const T_*>(source), \
                       static_cast<const T_*>(grid), static_cast<const T_*>(target), static_cast<T_*>(out), static_cast<T_*>(grad), ag)
#define MH_PP_SCATTER(T_, PATH_, NT_)                                                                                        \
    hipLaunchKernelGGL((pushpull_scatter_kernel<T_, PATH_, NT_>), dim3(nbs), dim3(256), 0, s, static_cast<const T_*>(grid),   \
                       static_cast<const T_*>(target), static_cast<T_*>(out), a, tx, ty, tz)
#define MH_PP_LAUNCH(T_, PATH_, NT_)                                    \
    {                                                                   \
        if (gather) {                                                   \
            if (pull_only) MH_PP_LAUNCH1(T_, PATH_, NT_, true);         \
            else MH_PP_LAUNCH1(T_, PATH_, NT_, false);                  \
        }                                                               \
        if (scatter) MH_PP_SCATTER(T_, PATH_, NT_);                     \
    }
    // scatter modes: a workgroup owns a tile of target voxels (lanes along the last real axis)
    const int tx = ndim == 3 ? 4 : ndim == 2 ? 16 : 256, ty = ndim == 3 ? 4 : ndim == 2 ? 16 : 1, tz = ndim == 3 ? 16 : 1;
    const long long nbs_ll = (long long)B * cdiv(Xo, tx) * cdiv(Yo, ty) * cdiv(Zo, tz);
    if (nbs_ll > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "pushpull: problem too large for one launch");
    const unsigned nbs = (unsigned)nbs_ll;
#define MH_PP_PATHS(T_)                                                     \
    if (a.path == PP_NEAREST) MH_PP_LAUNCH(T_, PP_NEAREST, 1)               \
    else if (a.path == PP_LINEAR) MH_PP_LAUNCH(T_, PP_LINEAR, 2)            \
    else if (maxo <= 2) MH_PP_LAUNCH(T_, PP_GENERIC, 3)                     \
    else if (maxo == 3) MH_PP_LAUNCH(T_, PP_GENERIC, 4)                     \
    else MH_PP_LAUNCH(T_, PP_GENERIC, 8)
    if (is_f64) { MH_PP_PATHS(double) } else { MH_PP_PATHS(float) }
#undef MH_PP_PATHS
#undef MH_PP_LAUNCH
#undef MH_PP_SCATTER
#undef MH_PP_LAUNCH1
    return launched("pushpull");
}

// ------------------------------------------------------------------------------------------ post-processing (Activations / AsDiscrete)
int mh_pointwise_f32(int op, const float* src, float* dst, int64_t n, float param, void* stream) {
    if (!src || !dst || n < 1 || op < 0 || op > PW_ROUND) return fail(MH_ERR_ARG, "pointwise: bad argument");
    if (n > 0x7fffffffLL * 256) return fail(MH_ERR_UNSUPPORTED, "pointwise: problem too large for one launch");
    hipLaunchKernelGGL(pointwise_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n, op, param);
    return launched("pointwise");
}

int mh_channel_reduce_f32(int op, const float* src, float* dst, int C, int64_t n, void* stream) {
    if (!src || !dst || C < 1 || n < 1 || (op != CR_ARGMAX && op != CR_SOFTMAX)) return fail(MH_ERR_ARG, "channel_reduce: bad argument");
    if (n > 0x7fffffffLL * 256) return fail(MH_ERR_UNSUPPORTED, "channel_reduce: problem too large for one launch");
    if (op == CR_ARGMAX) hipLaunchKernelGGL(channel_argmax_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, C, (long long)n);
    else hipLaunchKernelGGL(channel_softmax_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, C, (long long)n);
    return launched("channel_reduce");
}

int mh_onehot_f32(const float* labels, float* dst, int K, int64_t n, void* stream) {
    if (!labels || !dst || K < 1 || n < 1) return fail(MH_ERR_ARG, "onehot: bad argument");
    if (n > 0x7fffffffLL * 256) return fail(MH_ERR_UNSUPPORTED, "onehot: problem too large for one launch");
    hipLaunchKernelGGL(onehot_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, labels, dst, K, (long long)n);
    return launched("onehot");
}

// ------------------------------------------------------------------------------------------ pre-processing (ScaleIntensityRange / CropForeground)
int mh_scale_intensity_range_f32(const float* src, float* dst, int64_t n, float a_min, float a_div, int rescale, float b_scale, float b_min,
                                 int clip_lo, float lo, int clip_hi, float hi, void* stream) {
    if (!src || !dst || n < 1) return fail(MH_ERR_ARG, "scale_intensity_range: bad argument");
    if (n > 0x7fffffffLL * 1024) return fail(MH_ERR_UNSUPPORTED, "scale_intensity_range: problem too large for one launch");
    ScaleRange p;
    p.a_min = a_min; p.div = a_div; p.b_scale = b_scale; p.b_min = b_min; p.lo = lo; p.hi = hi;
    p.rescale = rescale != 0; p.clip_lo = clip_lo != 0; p.clip_hi = clip_hi != 0;
    const dim3 grid(blocks_for((n + 3) / 4));
    if (aligned(src, 16) && aligned(dst, 16))
        hipLaunchKernelGGL(scale_range_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n, p);
    else
        hipLaunchKernelGGL(scale_range_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n, p);
    return launched("scale_intensity_range");
}

int mh_foreground_bbox_workspace_ints(int D, int H) {
    if (D < 1 || H < 1) return fail(MH_ERR_ARG, "foreground_bbox: bad argument");
    const long long groups = ((long long)D * H + 3) / 4;
    return 6 * (int)(groups < 2048 ? groups : 2048);
}

int mh_foreground_bbox_f32(const float* src, int C, int D, int H, int W, int32_t* workspace, int32_t* box6, void* stream) {
    if (!src || !workspace || !box6 || C < 1 || D < 1 || H < 1 || W < 1) return fail(MH_ERR_ARG, "foreground_bbox: bad argument");
    const int parts = mh_foreground_bbox_workspace_ints(D, H) / 6;
    if (W % 4 == 0 && aligned(src, 16) && ((long long)D * H * W) % 4 == 0)
        hipLaunchKernelGGL(bbox_partial_kernel<true>, dim3(parts), dim3(256), 0, (hipStream_t)stream, src, C, D, H, W, workspace);
    else
        hipLaunchKernelGGL(bbox_partial_kernel<false>, dim3(parts), dim3(256), 0, (hipStream_t)stream, src, C, D, H, W, workspace);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, workspace, parts, box6);
    return launched("foreground_bbox");
}

int mh_crop_pad_f32(const float* src, float* dst, int C, int D, int H, int W, int Do, int Ho, int Wo, int sz, int sy, int sx, float value,
                    void* stream) {
    if (!src || !dst || C < 1 || D < 1 || H < 1 || W < 1 || Do < 1 || Ho < 1 || Wo < 1) return fail(MH_ERR_ARG, "crop_pad: bad argument");
    const long long rows = (long long)C * Do * Ho;
    if (rows > 0x7fffffffLL || (Wo + 255) / 256 > 65535) return fail(MH_ERR_UNSUPPORTED, "crop_pad: problem too large for one launch");
    hipLaunchKernelGGL(crop_pad_kernel, dim3((unsigned)rows, (unsigned)((Wo + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, D, H, W, Do,
                       Ho, Wo, sz, sy, sx, value);
    return launched("crop_pad");
}

static inline int normalize_parts(int64_t n) {
    const long long want = (n + 4095) / 4096;        // >= 16 voxels per lane
    return (int)(want < 1 ? 1 : want > 1024 ? 1024 : want);
}

int64_t mh_normalize_stats_workspace_doubles(int C, int64_t n) {
    if (C < 1 || n < 1) return fail(MH_ERR_ARG, "normalize_stats: bad argument");
    return (int64_t)C * normalize_parts(n) * 3;
}

int mh_normalize_stats_f32(const float* src, int C, int64_t n, int nonzero, double* workspace, float* subdiv, void* stream) {
    if (!src || !workspace || !subdiv || C < 1 || C > 65535 || n < 1) return fail(MH_ERR_ARG, "normalize_stats: bad argument");
    const int parts = normalize_parts(n);
    const bool vec = aligned(src, 16) && n % 4 == 0;
    const dim3 grid((unsigned)parts, (unsigned)C);
#define MH_NS_LAUNCH(NZ_, V_) hipLaunchKernelGGL((masked_stats_kernel<NZ_, V_>), grid, dim3(256), 0, (hipStream_t)stream, src, (long long)n, workspace)
    if (nonzero) { if (vec) MH_NS_LAUNCH(true, true); else MH_NS_LAUNCH(true, false); }
    else { if (vec) MH_NS_LAUNCH(false, true); else MH_NS_LAUNCH(false, false); }
#undef MH_NS_LAUNCH
    hipLaunchKernelGGL(masked_stats_finalize_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, workspace, parts, subdiv);
    return launched("normalize_stats");
}

int mh_normalize_apply_f32(const float* src, float* dst, int C, int64_t n, int nonzero, const float* subdiv, void* stream) {
    if (!src || !dst || !subdiv || C < 1 || C > 65535 || n < 1) return fail(MH_ERR_ARG, "normalize_apply: bad argument");
    if (n > 0x7fffffffLL * 1024) return fail(MH_ERR_UNSUPPORTED, "normalize_apply: problem too large for one launch");
    const bool vec = aligned(src, 16) && aligned(dst, 16) && n % 4 == 0;
    const dim3 grid(blocks_for((n + 3) / 4), (unsigned)C);
#define MH_NA_LAUNCH(NZ_, V_) hipLaunchKernelGGL((masked_normalize_kernel<NZ_, V_>), grid, dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n, subdiv)
    if (nonzero) { if (vec) MH_NA_LAUNCH(true, true); else MH_NA_LAUNCH(true, false); }
    else { if (vec) MH_NA_LAUNCH(false, true); else MH_NA_LAUNCH(false, false); }
#undef MH_NA_LAUNCH
    return launched("normalize_apply");
}

int64_t mh_minmax_workspace_floats(int C, int64_t n) {
    if (C < 1 || n < 1) return fail(MH_ERR_ARG, "minmax: bad argument");
    return (int64_t)C * normalize_parts(n) * 2;
}

int mh_minmax_f32(const float* src, int C, int64_t n, float* workspace, float* table, void* stream) {
    if (!src || !workspace || !table || C < 1 || C > 65535 || n < 1) return fail(MH_ERR_ARG, "minmax: bad argument");
    const int parts = normalize_parts(n);
    if (aligned(src, 16) && n % 4 == 0)
        hipLaunchKernelGGL(minmax_partial_kernel<true>, dim3((unsigned)parts, (unsigned)C), dim3(256), 0, (hipStream_t)stream, src, (long long)n, workspace);
    else
        hipLaunchKernelGGL(minmax_partial_kernel<false>, dim3((unsigned)parts, (unsigned)C), dim3(256), 0, (hipStream_t)stream, src, (long long)n, workspace);
    hipLaunchKernelGGL(minmax_final_kernel, dim3((unsigned)C), dim3(64), 0, (hipStream_t)stream, workspace, parts, table);
    return launched("minmax");
}

int mh_minmax_scale_f32(const float* src, float* dst, int C, int64_t n, const float* table, int rescale, float b_scale, float b_min,
                        int flat_has_mul, float flat_mul, void* stream) {
    if (!src || !dst || !table || C < 1 || C > 65535 || n < 1) return fail(MH_ERR_ARG, "minmax_scale: bad argument");
    if (n > 0x7fffffffLL * 1024) return fail(MH_ERR_UNSUPPORTED, "minmax_scale: problem too large for one launch");
    MinMaxScale p;
    p.b_scale = b_scale; p.b_min = b_min; p.flat_mul = flat_mul; p.rescale = rescale != 0; p.flat_has_mul = flat_has_mul != 0;
    const dim3 grid(blocks_for((n + 3) / 4), (unsigned)C);
    if (aligned(src, 16) && aligned(dst, 16) && n % 4 == 0)
        hipLaunchKernelGGL(minmax_scale_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n, table, p);
    else
        hipLaunchKernelGGL(minmax_scale_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, dst, (long long)n, table, p);
    return launched("minmax_scale");
}

int mh_flip_permute_f32(const float* src, float* dst, int C, const int32_t* in_size3, const int32_t* perm3, const int32_t* flip3, void* stream) {
    if (!src || !dst || !in_size3 || !perm3 || !flip3 || C < 1) return fail(MH_ERR_ARG, "flip_permute: bad argument");
    int seen = 0;
    for (int k = 0; k < 3; ++k) {
        if (in_size3[k] < 1 || perm3[k] < 0 || perm3[k] > 2) return fail(MH_ERR_ARG, "flip_permute: bad size / permutation");
        seen |= 1 << perm3[k];
    }
    if (seen != 7) return fail(MH_ERR_ARG, "flip_permute: perm3 is not a permutation of (0, 1, 2)");
    const long long in_stride[3] = {(long long)in_size3[1] * in_size3[2], in_size3[2], 1};
    long long base = 0, st[3];
    int out_size[3];
    for (int k = 0; k < 3; ++k) {            // output axis k reads input axis a = perm3[k], reversed when flip3[a]
        const int a = perm3[k];
        out_size[k] = in_size3[a];
        st[k] = flip3[a] ? -in_stride[a] : in_stride[a];
        if (flip3[a]) base += (long long)(in_size3[a] - 1) * in_stride[a];
    }
    const long long rows = (long long)C * out_size[0] * out_size[1];
    if (rows > 0x7fffffffLL || (out_size[2] + 255) / 256 > 65535) return fail(MH_ERR_UNSUPPORTED, "flip_permute: problem too large for one launch");
    hipLaunchKernelGGL(flip_permute_kernel, dim3((unsigned)rows, (unsigned)((out_size[2] + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       out_size[0], out_size[1], out_size[2], in_stride[0] * in_size3[0], base, st[0], st[1], st[2]);
    return launched("flip_permute");
}

// ------------------------------------------------------------------------------------------ Gaussian smoothing
int mh_separable_filter3d_f32(const float* src, float* dst, int NC, int D, int H, int W, const float* kz, int kz_n, const float* ky,
                              int ky_n, const float* kx, int kx_n, void* stream) {
    if (!src || !dst || !kz || !ky || !kx || NC < 1 || D < 1 || H < 1 || W < 1) return fail(MH_ERR_ARG, "separable_filter3d: bad argument");
    const int n[3] = {kz_n, ky_n, kx_n};
    for (int i = 0; i < 3; ++i) {
        if (n[i] < 1 || n[i] % 2 == 0) return fail(MH_ERR_ARG, "separable_filter3d: tap counts must be odd and positive");
        if (n[i] > GS_MAX_TAPS) return fail(MH_ERR_UNSUPPORTED, "separable_filter3d: %d taps on one axis (at most %d)", n[i], GS_MAX_TAPS);
    }
    if (src == dst) return fail(MH_ERR_ARG, "separable_filter3d: in-place operation is not supported");
    const int kmax = n[0] > n[1] ? (n[0] > n[2] ? n[0] : n[2]) : (n[1] > n[2] ? n[1] : n[2]);
    const int rk = kmax <= 3 ? 3 : kmax <= 5 ? 5 : kmax <= 9 ? 9 : kmax <= 17 ? 17 : 33;
    GaussArgs a;
    a.NC = NC; a.D = D; a.H = H; a.W = W;
    memset(a.kz, 0, sizeof(a.kz)); memset(a.ky, 0, sizeof(a.ky)); memset(a.kx, 0, sizeof(a.kx));
    // centre every kernel inside the RK taps the kernel template unrolls (zero taps on both sides)
    memcpy(a.kz + (rk - kz_n) / 2, kz, sizeof(float) * kz_n);
    memcpy(a.ky + (rk - ky_n) / 2, ky, sizeof(float) * ky_n);
    memcpy(a.kx + (rk - kx_n) / 2, kx, sizeof(float) * kx_n);
    bool iso = kz_n == ky_n && ky_n == kx_n;
    for (int i = 0; iso && i < kz_n; ++i) iso = kz[i] == ky[i] && ky[i] == kx[i];
    hipStream_t s = (hipStream_t)stream;
    // float4-aligned volumes and up to 17 taps: the row-vector kernel (gaussian.h: 16-byte loads / stores, register x-pass, one barrier
    // per plane); MONAI_AMD_GS_IMPL=tile forces the round-1 tile kernel (bit-identical results)
    const char* impl = knob_str("MONAI_AMD_GS_IMPL");
    if (rk <= 17 && W % 4 == 0 && aligned(src, 16) && aligned(dst, 16) && !(impl && impl[0] == 't')) {
        // up to 9 taps: the DPP / forward-accumulation form (gauss3d_rowdpp_kernel); development builds: MONAI_AMD_GS_IMPL=v keeps the round-2 row-vector kernel
        const bool dpp = rk <= 9 && (long long)D * H * W * 4 < 0x80000000LL && !(impl && impl[0] == 'v');      // (its loads address a channel volume with 31-bit byte offsets)
        const long long vtiles = (long long)cdiv(W, GV_TX) * cdiv(H, GV_TY);
        static int vslots[4][2][2];
        const int vi = rk == 3 ? 0 : rk == 5 ? 1 : rk == 9 ? 2 : 3;
        int& vs = vslots[vi][iso ? 1 : 0][dpp ? 1 : 0];
        if (vs == 0) {
#define MH_GV_SLOTS(RK_) vs = iso ? resident_wgs(gauss3d_rowvec_kernel<RK_, true>, 64 * gv_waves(RK_)) : resident_wgs(gauss3d_rowvec_kernel<RK_, false>, 64 * gv_waves(RK_));
#define MH_GD_SLOTS(RK_) vs = iso ? resident_wgs(gauss3d_rowdpp_kernel<RK_, true>, 1024) : resident_wgs(gauss3d_rowdpp_kernel<RK_, false>, 1024);
            switch (rk) {
                case 3: if (dpp) { MH_GD_SLOTS(3) } else { MH_GV_SLOTS(3) } break;
                case 5: if (dpp) { MH_GD_SLOTS(5) } else { MH_GV_SLOTS(5) } break;
                case 9: if (dpp) { MH_GD_SLOTS(9) } else { MH_GV_SLOTS(9) } break;
                default: MH_GV_SLOTS(17) break;
            }
#undef MH_GD_SLOTS
#undef MH_GV_SLOTS
        }
        const int vmin = 4 * rk > 16 ? 4 * rk : 16;
        const int vchunks = stream_chunks(vtiles * NC, D, vs, rk - 1, vmin, "MONAI_AMD_GS_CHUNKS");
        a.zchunk = cdiv(D, vchunks);
        a.nchunk = cdiv(D, a.zchunk);
        a.pair_ok = 1;
        const long long vwg = vtiles * a.nchunk * NC;
        if (vwg > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "separable_filter3d: problem too large for one launch");
#define MH_GV(RK_)                                                                                                        \
        if (iso) hipLaunchKernelGGL((gauss3d_rowvec_kernel<RK_, true>), dim3((unsigned)vwg), dim3(64 * gv_waves(RK_)), 0, s, src, dst, a);   \
        else hipLaunchKernelGGL((gauss3d_rowvec_kernel<RK_, false>), dim3((unsigned)vwg), dim3(64 * gv_waves(RK_)), 0, s, src, dst, a);
#define MH_GD(RK_)                                                                                                        \
        if (iso) hipLaunchKernelGGL((gauss3d_rowdpp_kernel<RK_, true>), dim3((unsigned)vwg), dim3(1024), 0, s, src, dst, a);   \
        else hipLaunchKernelGGL((gauss3d_rowdpp_kernel<RK_, false>), dim3((unsigned)vwg), dim3(1024), 0, s, src, dst, a);
        switch (rk) {
            case 3: if (dpp) { MH_GD(3) } else { MH_GV(3) } break;
            case 5: if (dpp) { MH_GD(5) } else { MH_GV(5) } break;
            case 9: if (dpp) { MH_GD(9) } else { MH_GV(9) } break;
            default: MH_GV(17) break;
        }
#undef MH_GD
#undef MH_GV
        return launched("separable_filter3d_rowvec");
    }
    // cut z into chunks (each re-filters rk-1 halo planes) that fill whole rounds of the resident workgroup slots
    const long long tiles = (long long)cdiv(W, GS_TX) * cdiv(H, GS_TY);
    const int min_chunk = 4 * rk > 16 ? 4 * rk : 16;
    static int slots[5][2];
    const int ri = rk == 3 ? 0 : rk == 5 ? 1 : rk == 9 ? 2 : rk == 17 ? 3 : 4;
    int& sl = slots[ri][iso ? 1 : 0];
    if (sl == 0) {
#define MH_GAUSS_SLOTS(RK_) sl = iso ? resident_wgs(gauss3d_stream_kernel<RK_, true>, 256) : resident_wgs(gauss3d_stream_kernel<RK_, false>, 256);
        switch (rk) {
            case 3: MH_GAUSS_SLOTS(3) break;
            case 5: MH_GAUSS_SLOTS(5) break;
            case 9: MH_GAUSS_SLOTS(9) break;
            case 17: MH_GAUSS_SLOTS(17) break;
            default: MH_GAUSS_SLOTS(33) break;
        }
#undef MH_GAUSS_SLOTS
    }
    const int nchunk = stream_chunks(tiles * NC, D, sl, rk - 1, min_chunk, "MONAI_AMD_GS_CHUNKS");
    a.zchunk = cdiv(D, nchunk);
    a.pair_ok = (W % 2 == 0 && aligned(dst, 8)) ? 1 : 0;
    a.nchunk = cdiv(D, a.zchunk);
    const long long nwg = tiles * a.nchunk * NC;
    if (nwg > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "separable_filter3d: problem too large for one launch");
    const dim3 grid((unsigned)nwg);
#define MH_GAUSS(RK_)                                                                                             \
    if (iso) hipLaunchKernelGGL((gauss3d_stream_kernel<RK_, true>), grid, dim3(256), 0, s, src, dst, a);          \
    else hipLaunchKernelGGL((gauss3d_stream_kernel<RK_, false>), grid, dim3(256), 0, s, src, dst, a);
    switch (rk) {
        case 3: MH_GAUSS(3) break;
        case 5: MH_GAUSS(5) break;
        case 9: MH_GAUSS(9) break;
        case 17: MH_GAUSS(17) break;
        default: MH_GAUSS(33) break;
    }
#undef MH_GAUSS
    return launched("separable_filter3d");
}

// ------------------------------------------------------------------------------------------ UNETR pieces
int mh_add_act_f32(const mh_tensor5* a_, const mh_tensor5* b_, float slope, const mh_tensor5* out_, void* stream) {
    if (!dense_ok(a_) || (b_ && !dense_ok(b_)) || !dense_ok(out_)) return fail(MH_ERR_ARG, "add_act: bad tensor");
    if (!out_records_ok(out_)) return fail(MH_ERR_ARG, "add_act: the output view's records must be 16-byte aligned [N][>= C][4] floats");
    Tensor bnull = from_c(*a_);
    bnull.data = nullptr; bnull.nrm = nullptr;
    const Tensor a = from_c(*a_), b = b_ ? from_c(*b_) : bnull, out = from_c(*out_);
    if (a.N != b.N || a.C != b.C || a.D != b.D || a.H != b.H || a.W != b.W || a.N != out.N || a.C != out.C || a.D != out.D ||
        a.H != out.H || a.W != out.W)
        return fail(MH_ERR_ARG, "add_act: shape mismatch");
    const long long DHW = (long long)a.D * a.H * a.W;
    const bool v4 = DHW % 4 == 0 && aligned(a.data, 16) && (!b.data || aligned(b.data, 16)) && aligned(out.data, 16) && a.n_stride % 4 == 0 &&
                    b.n_stride % 4 == 0 && out.n_stride % 4 == 0;
    const dim3 grid((blocks_for(v4 ? DHW / 4 : DHW) + ADD_ACT_CHUNKS - 1) / ADD_ACT_CHUNKS, (unsigned)a.C, (unsigned)a.N);
    if (v4) hipLaunchKernelGGL((add_act_kernel<4>), grid, dim3(256), 0, (hipStream_t)stream, a, b, slope, out);
    else hipLaunchKernelGGL((add_act_kernel<1>), grid, dim3(256), 0, (hipStream_t)stream, a, b, slope, out);
    return launched("add_act");
}

int mh_conv1x1_sum2_accepts(int Cout, int D, int H, int W) { return Cout >= 1 && Cout <= 8 && ((long long)D * H * W) % 4 == 0; }

int mh_conv1x1_sum2_f32(const mh_tensor5* a_, const mh_tensor5* b_, float slope, const float* w, const float* bias, const mh_tensor5* out_, void* stream) {
    if (!dense_ok(a_) || !dense_ok(b_) || !dense_ok(out_) || !w) return fail(MH_ERR_ARG, "conv1x1_sum2: bad tensor");
    const Tensor a = from_c(*a_), b = from_c(*b_), out = from_c(*out_);
    if (a.N != b.N || a.C != b.C || a.D != b.D || a.H != b.H || a.W != b.W || a.N != out.N || a.D != out.D || a.H != out.H || a.W != out.W)
        return fail(MH_ERR_ARG, "conv1x1_sum2: shape mismatch");
    const long long DHW = (long long)a.D * a.H * a.W;
    if (!mh_conv1x1_sum2_accepts(out.C, a.D, a.H, a.W) || !aligned(a.data, 16) || !aligned(b.data, 16) || !aligned(out.data, 16) || a.n_stride % 4 || b.n_stride % 4 ||
        out.n_stride % 4)
        return fail(MH_ERR_UNSUPPORTED, "conv1x1_sum2: needs 1 .. 8 output channels, D*H*W %% 4 == 0 and 16-byte aligned tensors (got %d channels)", out.C);
    const dim3 grid(blocks_for(DHW / 4), (unsigned)a.N);
#define MH_1S_CASE(CO) case CO: hipLaunchKernelGGL((conv1x1_sum2_kernel<CO>), grid, dim3(256), 0, (hipStream_t)stream, a, b, slope, w, bias, out); break;
    switch (out.C) { MH_1S_CASE(1) MH_1S_CASE(2) MH_1S_CASE(3) MH_1S_CASE(4) MH_1S_CASE(5) MH_1S_CASE(6) MH_1S_CASE(7) MH_1S_CASE(8) }
#undef MH_1S_CASE
    return launched("conv1x1_sum2");
}

int mh_pad_replicate_f32(const mh_tensor5* in_, const mh_tensor5* out_, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_)) return fail(MH_ERR_ARG, "pad_replicate: bad tensor");
    if (!out_records_ok(out_)) return fail(MH_ERR_ARG, "pad_replicate: the output view's records must be 16-byte aligned [N][>= C][4] floats");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || in.C != out.C || out.D < in.D || out.H < in.H || out.W < in.W || out.D > in.D + 1 || out.H > in.H + 1 || out.W > in.W + 1)
        return fail(MH_ERR_ARG, "pad_replicate: output extents must be the input's plus 0 or 1");
    const dim3 grid(blocks_for((long long)out.D * out.H * out.W), (unsigned)out.C, (unsigned)out.N);
    hipLaunchKernelGGL(pad_replicate_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, out);
    return launched("pad_replicate");
}

int mh_pixelshuffle_f32(const mh_tensor5* in_, const mh_tensor5* out_, int fz, int pad_pool, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_)) return fail(MH_ERR_ARG, "pixelshuffle: bad tensor");
    if (!out_records_ok(out_)) return fail(MH_ERR_ARG, "pixelshuffle: the output view's records must be 16-byte aligned [N][>= C][4] floats");
    if (fz != 1 && fz != 2) return fail(MH_ERR_ARG, "pixelshuffle: fz must be 1 (two spatial dimensions on one plane) or 2");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || in.C != out.C * fz * 4 || out.D != fz * in.D || out.H != 2 * in.H || out.W != 2 * in.W)
        return fail(MH_ERR_ARG, "pixelshuffle: [N][C * %d][D][H][W] -> [N][C][%d D][2 H][2 W] expected (got %d ch %dx%dx%d -> %d ch %dx%dx%d)", fz * 4, fz, in.C, in.D, in.H, in.W,
                    out.C, out.D, out.H, out.W);
    if (out.C > 65535 || out.N > 65535) return fail(MH_ERR_UNSUPPORTED, "pixelshuffle: more than 65535 channels / samples in one launch");
    const dim3 grid(blocks_for((long long)out.D * out.H * out.W), (unsigned)out.C, (unsigned)out.N);
    hipStream_t s = (hipStream_t)stream;
    if (fz == 2) {
        if (pad_pool) hipLaunchKernelGGL((pixelshuffle_kernel<2, true>), grid, dim3(256), 0, s, in, out);
        else hipLaunchKernelGGL((pixelshuffle_kernel<2, false>), grid, dim3(256), 0, s, in, out);
    } else {
        if (pad_pool) hipLaunchKernelGGL((pixelshuffle_kernel<1, true>), grid, dim3(256), 0, s, in, out);
        else hipLaunchKernelGGL((pixelshuffle_kernel<1, false>), grid, dim3(256), 0, s, in, out);
    }
    return launched("pixelshuffle");
}

int mh_attention_f32(const float* qkv, float* out, int B, int S, int heads, int head_dim, float scale, void* stream) {
    if (!qkv || !out || B < 1 || S < 1 || heads < 1) return fail(MH_ERR_ARG, "attention: bad argument");
    if (!aligned(qkv, 16) || !aligned(out, 16)) return fail(MH_ERR_ARG, "attention: 16-byte aligned tensors required");
    if (B > 65535 || heads > 65535) return fail(MH_ERR_UNSUPPORTED, "attention: more than 65535 batches / heads in one launch");
    const dim3 grid((unsigned)cdiv(S, 128), (unsigned)heads, (unsigned)B);
    hipStream_t s = (hipStream_t)stream;
    switch (head_dim) {          // any sequence length: keys / values stream through LDS (kernels/attention.h)
        case 32: hipLaunchKernelGGL((attention_h2_kernel<32>), grid, dim3(256), 0, s, qkv, out, S, heads, scale); break;
        case 64: hipLaunchKernelGGL((attention_h2_kernel<64>), grid, dim3(256), 0, s, qkv, out, S, heads, scale); break;
        case 96: hipLaunchKernelGGL((attention_h2_kernel<96>), grid, dim3(256), 0, s, qkv, out, S, heads, scale); break;
        case 128: hipLaunchKernelGGL((attention_h2_kernel<128>), grid, dim3(256), 0, s, qkv, out, S, heads, scale); break;
        default: return fail(MH_ERR_UNSUPPORTED, "attention: head_dim %d is not built (32, 64, 96, 128 are)", head_dim);
    }
    return launched("attention");
}

int mh_window_attention_f32(const float* qkv, const float* bias_t, const float* mask, float* out, int BW, int nW, int S, int heads, int head_dim,
                            float scale, int exact_fp32, void* stream) {
    if (!qkv || !out || BW < 1 || S < 1 || heads < 1 || nW < 1) return fail(MH_ERR_ARG, "window_attention: bad argument");
    if (mask && BW % nW) return fail(MH_ERR_ARG, "window_attention: %d windows are not a multiple of the %d mask windows", BW, nW);
    if (!aligned(qkv, 16) || !aligned(out, 16)) return fail(MH_ERR_ARG, "window_attention: 16-byte aligned tensors required");
    hipStream_t s0 = (hipStream_t)stream;
    // head dims 16 / 32: the streaming split-precision kernel on the fp16 matrix cores (round 4), any number of tokens; head dim 8 (feature size 24): the VALU kernel
    // exact_fp32 (the caller pinned the exact-fp32 family): the VALU kernel wherever its LDS-resident form fits -- the split-precision kernel takes q / k / v as they come
    // (|x| < 65504, no input scaling: ADVICE r4), which hidden states behind a LayerNorm satisfy and a caller who asked for exact fp32 did not sign up for
    if ((head_dim == 16 || head_dim == 32) && !(exact_fp32 && S <= WA_MAX_TOKENS)) {
        const dim3 g2((unsigned)cdiv(S, 128), (unsigned)heads, (unsigned)BW);
        if (head_dim == 16) hipLaunchKernelGGL((window_attention_h2_kernel<16, false>), g2, dim3(256), 0, s0, qkv, bias_t, mask, out, S, heads, nW, scale, WinRel{});
        else hipLaunchKernelGGL((window_attention_h2_kernel<32, false>), g2, dim3(256), 0, s0, qkv, bias_t, mask, out, S, heads, nW, scale, WinRel{});
        return launched("window_attention_h2");
    }
    if (S > WA_MAX_TOKENS) return fail(MH_ERR_UNSUPPORTED, "window_attention: %d tokens per window exceed the LDS-resident limit of %d", S, WA_MAX_TOKENS);
    const size_t lds = 0;
    const dim3 grid((unsigned)heads, (unsigned)BW);
    hipStream_t s = (hipStream_t)stream;
    switch (head_dim) {
        case 8: hipLaunchKernelGGL((window_attention_kernel<8>), grid, dim3(256), lds, s, qkv, bias_t, mask, out, S, heads, nW, scale); break;
        case 16: hipLaunchKernelGGL((window_attention_kernel<16>), grid, dim3(256), lds, s, qkv, bias_t, mask, out, S, heads, nW, scale); break;
        case 32: hipLaunchKernelGGL((window_attention_kernel<32>), grid, dim3(256), lds, s, qkv, bias_t, mask, out, S, heads, nW, scale); break;
        default: return fail(MH_ERR_UNSUPPORTED, "window_attention: head_dim %d is not built (8, 16, 32 are)", head_dim);
    }
    return launched("window_attention");
}

int mh_window_attention_rel_accepts(int S, int head_dim, int table_rows) {
    return (head_dim == 16 || head_dim == 32) && S >= 1 && S <= WA_REL_TOKENS && table_rows >= 1 && table_rows <= WA_REL_ROWS;
}

int mh_window_attention_rel_f32(const float* qkv, const float* rel_table, int table_rows, const int32_t* coord, int coord_off, const int32_t* region,
                                float* out, int BW, int nW, int S, int heads, int head_dim, float scale, void* stream) {
    if (!qkv || !rel_table || !coord || !out || BW < 1 || S < 1 || heads < 1 || nW < 1) return fail(MH_ERR_ARG, "window_attention_rel: bad argument");
    if (region && BW % nW) return fail(MH_ERR_ARG, "window_attention_rel: %d windows are not a multiple of the %d mask windows", BW, nW);
    if (!aligned(qkv, 16) || !aligned(out, 16)) return fail(MH_ERR_ARG, "window_attention_rel: 16-byte aligned tensors required");
    if (!mh_window_attention_rel_accepts(S, head_dim, table_rows))
        return fail(MH_ERR_UNSUPPORTED, "window_attention_rel: head dim %d, %d tokens, %d table rows (built: head dims 16 / 32, <= %d tokens, <= %d rows)", head_dim, S,
                    table_rows, WA_REL_TOKENS, WA_REL_ROWS);
    const dim3 g2((unsigned)cdiv(S, 128), (unsigned)heads, (unsigned)BW);
    const WinRel rel{rel_table, coord, region, table_rows, coord_off};
    if (head_dim == 16)
        hipLaunchKernelGGL((window_attention_h2_kernel<16, true>), g2, dim3(256), 0, (hipStream_t)stream, qkv, (const float*)nullptr, (const float*)nullptr, out, S, heads, nW, scale, rel);
    else
        hipLaunchKernelGGL((window_attention_h2_kernel<32, true>), g2, dim3(256), 0, (hipStream_t)stream, qkv, (const float*)nullptr, (const float*)nullptr, out, S, heads, nW, scale, rel);
    return launched("window_attention_rel");
}

// ------------------------------------------------------------------------------------------ dense transformer pieces
int64_t mh_linear_packed_floats(int N, int K) {
    if (N < 1 || K < 1) return fail(MH_ERR_ARG, "linear_packed_floats: bad argument");
    return (int64_t)cdiv(N, DN_BN) * cdiv(K, DN_BK) * DN_BT * 4 + H2_TAIL;
}

int mh_linear_pack_f32(const float* w, int N, int K, float* packed, void* stream) {
    if (!w || !packed || N < 1 || K < 1) return fail(MH_ERR_ARG, "linear_pack: bad argument");
    if (!aligned(packed, 16)) return fail(MH_ERR_ARG, "linear_pack: 16-byte aligned packed buffer required");
    const int64_t slab_floats = mh_linear_packed_floats(N, K) - H2_TAIL;
    float* tail = packed + slab_floats;
    hipLaunchKernelGGL(conv3d_k3_h2_scale_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, (long long)N * K, tail);
    const long long padded = (long long)cdiv(N, DN_BN) * DN_BN * cdiv(K, DN_BK) * DN_BK;
    hipLaunchKernelGGL(linear_h2_pack_kernel, dim3(blocks_for(padded)), dim3(256), 0, (hipStream_t)stream, w, N, K, reinterpret_cast<_Float16*>(packed), tail);
    return launched("linear_pack");
}

int mh_linear_f32(const float* x, const float* packed_w, const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                  void* stream) {
    return mh_linear_tile_f32(x, packed_w, bias, residual, y, M, N, K, act, 0, stream);
}

static int linear_impl(const float* x, const float* packed_w, const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                       int tile, const int32_t* rowmap_, void* stream);

int mh_linear_tile_f32(const float* x, const float* packed_w, const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                       int tile, void* stream) {
    return linear_impl(x, packed_w, bias, residual, y, M, N, K, act, tile, nullptr, stream);
}

int mh_linear_scatter_f32(const float* x, const float* packed_w, const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                          const int32_t* dst_row, void* stream) {
    if (!dst_row) return fail(MH_ERR_ARG, "linear_scatter: null row map");
    return linear_impl(x, packed_w, bias, residual, y, M, N, K, act, 0, dst_row, stream);
}

static int linear_impl(const float* x, const float* packed_w, const float* bias, const float* residual, float* y, int64_t M, int N, int K, int act,
                       int tile, const int32_t* rowmap_, void* stream) {
    const int* rowmap = reinterpret_cast<const int*>(rowmap_);
    if (!x || !packed_w || !y || M < 1 || N < 1 || K < 1) return fail(MH_ERR_ARG, "linear: bad argument");
    if (tile != 0 && tile != 64 && tile != 128) return fail(MH_ERR_ARG, "linear: tile must be 0, 64 or 128 (got %d)", tile);
    if (K % 4 || !aligned(x, 16) || !aligned(packed_w, 16)) return fail(MH_ERR_ARG, "linear: K %% 4 == 0 and 16-byte aligned x / packed weights required (K = %d)", K);
    if (act < 0 || act > 1) return fail(MH_ERR_UNSUPPORTED, "linear: activation %d is not built (0 none, 1 GELU)", act);
    const int ntn = cdiv(N, DN_BN);
    const long long total = (long long)cdiv((int)((M + DN_BM - 1) / DN_BM), 1) * ntn;
    if (M > 0x7fffffffLL || total > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "linear: problem too large for one launch");
    const uint4* wq = reinterpret_cast<const uint4*>(packed_w);
    const float* tail = packed_w + (mh_linear_packed_floats(N, K) - H2_TAIL);
    hipStream_t s = (hipStream_t)stream;
    // many tokens (the ViT blocks of UNETR: M = 216 x windows): 8-wave workgroups with a 128 x 128 tile -- 2/3 of the L2 traffic per flop (kernels/dense.h) -- when
    // they still fill the chip twice over (tile == 0); measured 1.2-1.5 x the 128 x 64 kernel at M = 13 824 (profiles/r03_linear_bench.json; 256 x 128: slower, removed)
    const int ntn2 = cdiv(N, 2 * DN_BN);
    const long long big1 = (long long)((M + DN_BM - 1) / DN_BM) * ntn2;
    if (tile == 128 || (tile == 0 && big1 >= 512)) {
        const dim3 bgrid((unsigned)big1);
#define MH_LINEAR_BIG(A_, R_) hipLaunchKernelGGL((linear_h2_big_kernel<A_, R_, 1>), bgrid, dim3(512), 0, s, x, wq, tail, bias, residual, y, (int)M, N, K, ntn2, ntn, rowmap)
        if (act == 1 && residual) MH_LINEAR_BIG(1, true);
        else if (act == 1) MH_LINEAR_BIG(1, false);
        else if (residual) MH_LINEAR_BIG(0, true);
        else MH_LINEAR_BIG(0, false);
#undef MH_LINEAR_BIG
        return launched("linear");
    }
    const dim3 grid((unsigned)total);
#define MH_LINEAR_LAUNCH(A_, R_) hipLaunchKernelGGL((linear_h2_kernel<A_, R_>), grid, dim3(256), 0, s, x, wq, tail, bias, residual, y, (int)M, N, K, ntn, rowmap)
    if (act == 1 && residual) MH_LINEAR_LAUNCH(1, true);
    else if (act == 1) MH_LINEAR_LAUNCH(1, false);
    else if (residual) MH_LINEAR_LAUNCH(0, true);
    else MH_LINEAR_LAUNCH(0, false);
#undef MH_LINEAR_LAUNCH
    return launched("linear");
}

static bool layernorm_vec_ok(const float* x, const float* gamma, const float* beta, const float* y, int K) {
    return K % 4 == 0 && K <= 1024 && aligned(x, 16) && aligned(y, 16) && (!gamma || aligned(gamma, 16)) && (!beta || aligned(beta, 16));
}

// 16 / 32 / 64 lanes per row, 16-byte vectors (kernels/dense.h); M = output rows
static void layernorm_vec_launch(const float* x, const float* gamma, const float* beta, float eps, float* y, int64_t M, int K, const int32_t* src_row, hipStream_t s) {
#define MH_LNV_LAUNCH(G_, NV_)                                                                                                                       \
    hipLaunchKernelGGL((layernorm_vec_kernel<G_, NV_>), dim3((unsigned)((M + 4 * (64 / G_) - 1) / (4 * (64 / G_)))), dim3(256), 0, s, x, gamma, beta, eps, y, (int)M, K, \
                       reinterpret_cast<const int*>(src_row))
    if (K <= 64) MH_LNV_LAUNCH(16, 1);
    else if (K <= 128) MH_LNV_LAUNCH(32, 1);
    else if (K <= 256) MH_LNV_LAUNCH(64, 1);
    else if (K <= 512) MH_LNV_LAUNCH(64, 2);
    else if (K <= 768) MH_LNV_LAUNCH(64, 3);
    else MH_LNV_LAUNCH(64, 4);
#undef MH_LNV_LAUNCH
}

int mh_layernorm_f32(const float* x, const float* gamma, const float* beta, float eps, float* y, int64_t M, int K, void* stream) {
    if (!x || !y || M < 1 || K < 1) return fail(MH_ERR_ARG, "layernorm: bad argument");
    if (K > 64 * LN_MAXV) return fail(MH_ERR_UNSUPPORTED, "layernorm: %d features exceed the register-resident limit of %d", K, 64 * LN_MAXV);
    if ((M + 3) / 4 > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "layernorm: too many rows for one launch");
    hipStream_t s = (hipStream_t)stream;
    if (layernorm_vec_ok(x, gamma, beta, y, K)) {
        layernorm_vec_launch(x, gamma, beta, eps, y, M, K, nullptr, s);
        return launched("layernorm");
    }
    const dim3 grid((unsigned)((M + 3) / 4));
#define MH_LN_LAUNCH(NV_) hipLaunchKernelGGL((layernorm_kernel<NV_>), grid, dim3(256), 0, s, x, gamma, beta, eps, y, (int)M, K)
    if (K <= 64) MH_LN_LAUNCH(1);
    else if (K <= 128) MH_LN_LAUNCH(2);
    else if (K <= 256) MH_LN_LAUNCH(4);
    else if (K <= 512) MH_LN_LAUNCH(8);
    else if (K <= 1024) MH_LN_LAUNCH(16);
    else MH_LN_LAUNCH(64);
#undef MH_LN_LAUNCH
    return launched("layernorm");
}

int mh_layernorm_gather_accepts(int K) { return K >= 4 && K % 4 == 0 && K <= 1024; }

int mh_layernorm_gather_f32(const float* x, const float* gamma, const float* beta, float eps, float* y, int64_t M_out, int K, const int32_t* src_row, void* stream) {
    if (!x || !y || !src_row || M_out < 1 || K < 1) return fail(MH_ERR_ARG, "layernorm_gather: bad argument");
    if (!mh_layernorm_gather_accepts(K) || !layernorm_vec_ok(x, gamma, beta, y, K))
        return fail(MH_ERR_UNSUPPORTED, "layernorm_gather: rows of %d features (built: multiples of 4 up to 1024, 16-byte aligned tensors)", K);
    if (M_out > 0x7fffffffLL) return fail(MH_ERR_UNSUPPORTED, "layernorm_gather: too many rows for one launch");
    layernorm_vec_launch(x, gamma, beta, eps, y, M_out, K, src_row, (hipStream_t)stream);
    return launched("layernorm_gather");
}

// ------------------------------------------------------------------------------------------ UNet pieces
int mh_conv3d_k3_strided_f32(const mh_tensor5* in_, const float* packed_w, const float* bias, const mh_tensor5* out_, int stride,
                             void* stream) {
    return mh_conv3d_k3_strided3_f32(in_, packed_w, bias, out_, stride, stride, stride, stream);
}

int mh_conv3d_k3_strided3_f32(const mh_tensor5* in_, const float* packed_w, const float* bias, const mh_tensor5* out_, int sz, int sy, int sx,
                              void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_) || !packed_w || sz < 1 || sy < 1 || sx < 1) return fail(MH_ERR_ARG, "conv3d_k3_strided: bad argument");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || out.D != (in.D - 1) / sz + 1 || out.H != (in.H - 1) / sy + 1 || out.W != (in.W - 1) / sx + 1)
        return fail(MH_ERR_ARG, "conv3d_k3_strided: output must be floor((in - 1) / stride) + 1");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nbv = blocks_for((long long)out.D * out.H * out.W);
    if (out.C <= 8) {      // few output channels (e.g. UNet's 5-class top level): one exact-width pass, no channel guards
#define MH_STRIDED_CASE(C_)                                                                                                     \
    case C_:                                                                                                                    \
        hipLaunchKernelGGL((conv3d_k3_strided_kernel<C_, true>), dim3(nbv, 1u, (unsigned)out.N), dim3(256), 0, s, in, packed_w, bias, out, sz, sy, sx); \
        break;
        switch (out.C) {
            MH_STRIDED_CASE(1) MH_STRIDED_CASE(2) MH_STRIDED_CASE(3) MH_STRIDED_CASE(4)
            MH_STRIDED_CASE(5) MH_STRIDED_CASE(6) MH_STRIDED_CASE(7) MH_STRIDED_CASE(8)
        }
#undef MH_STRIDED_CASE
        return launched("conv3d_k3_strided");
    }
    constexpr int COT = 16;
    const dim3 grid(nbv, (unsigned)cdiv(out.C, COT), (unsigned)out.N);
    if (out.C % COT == 0) hipLaunchKernelGGL((conv3d_k3_strided_kernel<COT, true>), grid, dim3(256), 0, s, in, packed_w, bias, out, sz, sy, sx);
    else hipLaunchKernelGGL((conv3d_k3_strided_kernel<COT, false>), grid, dim3(256), 0, s, in, packed_w, bias, out, sz, sy, sx);
    return launched("conv3d_k3_strided");
}

int mh_deconv_ks_f32(const mh_tensor5* in_, const float* w, const float* bias, const mh_tensor5* out_, int fz, int fy, int fx, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_) || !w || fz < 1 || fz > 2 || fy < 1 || fy > 2 || fx < 1 || fx > 2)
        return fail(MH_ERR_ARG, "deconv_ks: bad argument (factors are 1 or 2 per axis)");
    if (!out_records_ok(out_)) return fail(MH_ERR_ARG, "deconv_ks: the output view's records must be 16-byte aligned [N][>= C][4] floats");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || out.D != fz * in.D || out.H != fy * in.H || out.W != fx * in.W) return fail(MH_ERR_ARG, "deconv_ks: output must be factor * input");
    const unsigned nb = blocks_for((long long)out.D * out.H * out.W);
    hipLaunchKernelGGL((deconv_ks_kernel<8>), dim3(nb, (unsigned)cdiv(out.C, 8), (unsigned)out.N), dim3(256), 0, (hipStream_t)stream, in, w, bias, out, fz, fy, fx);
    return launched("deconv_ks");
}

int mh_deconv_k3_f32(const mh_tensor5* in_, const float* w, const float* bias, const mh_tensor5* out_, int stride, void* stream) {
    if (!dense_ok(in_) || !dense_ok(out_) || !w || stride < 1) return fail(MH_ERR_ARG, "deconv_k3: bad argument");
    const Tensor in = from_c(*in_), out = from_c(*out_);
    if (in.N != out.N || out.D != in.D * stride || out.H != in.H * stride || out.W != in.W * stride)
        return fail(MH_ERR_ARG, "deconv_k3: output must be stride * input (padding 1, output_padding stride - 1)");
    if (stride == 2 && aligned(out.data, 8) && out.n_stride % 2 == 0) {
        const unsigned nb = blocks_for((long long)in.D * in.H * in.W);
        hipStream_t s = (hipStream_t)stream;
        if (out.C % 8 == 0) hipLaunchKernelGGL((deconv_k3s2_kernel<8, true>), dim3(nb, (unsigned)(out.C / 8), (unsigned)out.N), dim3(256), 0, s, in, w, bias, out);
        else if (out.C == 5) hipLaunchKernelGGL((deconv_k3s2_kernel<5, true>), dim3(nb, 1u, (unsigned)out.N), dim3(256), 0, s, in, w, bias, out);
        else if (out.C <= 4) hipLaunchKernelGGL((deconv_k3s2_kernel<4, false>), dim3(nb, 1u, (unsigned)out.N), dim3(256), 0, s, in, w, bias, out);
        else hipLaunchKernelGGL((deconv_k3s2_kernel<8, false>), dim3(nb, (unsigned)cdiv(out.C, 8), (unsigned)out.N), dim3(256), 0, s, in, w, bias, out);
        return launched("deconv_k3s2");
    }
    constexpr int COT = 16;
    const dim3 grid(blocks_for((long long)out.D * out.H * out.W), (unsigned)cdiv(out.C, COT), (unsigned)out.N);
    hipLaunchKernelGGL((deconv_k3_kernel<COT>), grid, dim3(256), 0, (hipStream_t)stream, in, w, bias, out, stride);
    return launched("deconv_k3");
}
