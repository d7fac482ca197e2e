// Sliding-window inferer kernels: window gather and the fused importance-weighted blend.
// Reference behaviour: monai/inferers/utils.py:215-298 (window loop, `*= w`, `+=`, count map, `/=`).
#pragma once
#include <type_traits>

#include "common.h"

namespace mh {

constexpr int MAX_AXIS_WINDOWS = 160;

// Per-axis window starts of the dense window grid (monai/data/utils.py:166-206), passed by value in the
// kernel arguments: no device-side tables, nothing to upload.
struct WindowGrid {
    int nz, ny, nx;
    int sz[MAX_AXIS_WINDOWS], sy[MAX_AXIS_WINDOWS], sx[MAX_AXIS_WINDOWS];
};

// Covering range of coordinate p on one axis: windows i with s[i] <= p < s[i] + r.  Starts ascend and
// the window size is constant, so the covering set is a contiguous index range [lo, hi].
__device__ __forceinline__ void cover(const int* s, int n, int r, int p, int& lo, int& hi) {
    lo = 0;
    while (lo < n - 1 && s[lo] + r <= p) ++lo;
    hi = lo;
    while (hi < n - 1 && s[hi + 1] <= p) ++hi;
}

// ---------------------------------------------------------------------------------------------------
// Gather `nwin` consecutive windows (first = w0, row-major over the start lists) of one image
// [C][D][H][W] into a dense batch [nwin][C][rd][rh][rw].  One thread per VEC output elements.
template <int VEC>
__global__ void __launch_bounds__(256)
window_extract_kernel(const float* __restrict__ vol, int C, int D, int H, int W, WindowGrid g, int w0, int nwin,
                      int rd, int rh, int rw, float* __restrict__ out) {
    const int rwv = rw / VEC;
    const long long total = (long long)nwin * C * rd * rh * rwv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int xv = (int)(idx % rwv);
    long long t = idx / rwv;
    const int ly = (int)(t % rh); t /= rh;
    const int lz = (int)(t % rd); t /= rd;
    const int c = (int)(t % C);
    const int wl = (int)(t / C);
    const int w = w0 + wl;
    const int ix = w % g.nx, iy = (w / g.nx) % g.ny, iz = w / (g.nx * g.ny);
    const int z = g.sz[iz] + lz, y = g.sy[iy] + ly, x = g.sx[ix] + xv * VEC;
    const float* src = vol + (((long long)c * D + z) * H + y) * W + x;
    float* dst = out + idx * VEC;
    if (VEC == 4) {
        *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    } else {
        dst[0] = src[0];
    }
}

// ---------------------------------------------------------------------------------------------------
// Fused blend: one thread owns VEC consecutive output voxels along W and KT output channels
// [k0, k0+KT).  It walks the covering windows in ascending window index and accumulates in the exact
// floating-point order of the reference: acc = acc + fp32(logit * w), cnt = cnt + w, out = acc / cnt
// (separately rounded multiply / add / IEEE divide: no FMA contraction).
// HBM traffic: logits are read exactly once (every window voxel covers exactly one output voxel), the
// output is written once; the importance map (roi^3 floats) and the start lists stay cache resident.
template <int KT, int VEC>
__global__ void __launch_bounds__(256)
sw_blend_kernel(const float* __restrict__ logits, const float* __restrict__ imp, float* __restrict__ out, int K,
                int k0, int D, int H, int W, int rd, int rh, int rw, WindowGrid g, int premul, long long wstride) {
    const int wv = W / VEC;
    const long long total = (long long)D * H * wv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % wv) * VEC;
    const long long t = idx / wv;
    const int y = (int)(t % H), z = (int)(t / H);

    int zlo, zhi, ylo, yhi, xlo, xhi;
    cover(g.sz, g.nz, rd, z, zlo, zhi);
    cover(g.sy, g.ny, rh, y, ylo, yhi);
    cover(g.sx, g.nx, rw, x, xlo, xhi);  // with VEC == 4 all starts and rw are multiples of 4: same set for x..x+3

    const long long plane = (long long)rh * rw, roi = plane * rd;
    float acc[KT][VEC];
    float cnt[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        cnt[v] = 0.0f;
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k][v] = 0.0f;
    }

    for (int iz = zlo; iz <= zhi; ++iz) {
        const int lz = z - g.sz[iz];
        for (int iy = ylo; iy <= yhi; ++iy) {
            const int ly = y - g.sy[iy];
            for (int ix = xlo; ix <= xhi; ++ix) {
                const int lx = x - g.sx[ix];
                const long long w = ((long long)iz * g.ny + iy) * g.nx + ix;
                const long long off = (long long)lz * plane + (long long)ly * rw + lx;
                const float* lp = logits + w * wstride + k0 * roi + off;
                float wt[VEC];
                float lv[KT][VEC];
                if (VEC == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(imp + off);
                    wt[0] = q.x; wt[1] = q.y; wt[2] = q.z; wt[3] = q.w;
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const float4 a = *reinterpret_cast<const float4*>(lp + k * roi);
                        lv[k][0] = a.x; lv[k][1] = a.y; lv[k][2] = a.z; lv[k][3] = a.w;
                    }
                } else {
                    wt[0] = imp[off];
#pragma unroll
                    for (int k = 0; k < KT; ++k) lv[k][0] = lp[k * roi];
                }
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
#pragma unroll
                    // premul: the logits were already multiplied by their (per-batch) weight (process_fn); x * 1.0f is exact
                    for (int k = 0; k < KT; ++k) acc[k][v] = __fadd_rn(acc[k][v], __fmul_rn(lv[k][v], premul ? 1.0f : wt[v]));
                    cnt[v] = __fadd_rn(cnt[v], wt[v]);
                }
            }
        }
    }

    const long long vox = (long long)D * H * W;
    float* op = out + (long long)k0 * vox + ((long long)z * H + y) * W + x;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        if (VEC == 4) {
            float4 r;
            r.x = __fdiv_rn(acc[k][0], cnt[0]); r.y = __fdiv_rn(acc[k][1], cnt[1]);
            r.z = __fdiv_rn(acc[k][2], cnt[2]); r.w = __fdiv_rn(acc[k][3], cnt[3]);
            *reinterpret_cast<float4*>(op + k * vox) = r;
        } else {
            op[k * vox] = __fdiv_rn(acc[k][0], cnt[0]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The blend in the summation order of the reference's BUFFERED schedule (monai/inferers/utils.py:239-253, 276-284, 324-348; `buffer_steps` > 0):
// windows are stably sorted by their start along `buffer_dim` (_create_buffered_slices: order = (start along that axis, row-major index)), consecutive groups of
// `bsteps` distinct starts share a slab buffer -- `buffer[win] += logit * w` in that order from zeros -- and each finished slab is ADDED to the zero-initialised
// output (`output[slab] += buffer`), so a voxel covered by windows of two slabs sums (0 + part_1) + part_2 instead of one running sum; the count map adds the
// weights of all covering windows in the sorted order.  The schedule itself (a memory-saving device on 16-80 GB cards) is not reproduced -- the all-window logits
// sit in HBM -- but its ARITHMETIC is: the same bits as the reference's buffered run.  `bax` = the buffered axis (0 = z, 1 = y, 2 = x of the 3-D view).
// PREMUL (process_fn, utils.py:232-238): `logits` already hold `p * w_t` of the callback's weight map (a separately rounded product, as the reference forms it); the kernel only
// adds, and the count uses `imp` = the map of the batch that was current at the first flush.
template <int KT, int VEC, bool PREMUL = false>
__global__ void __launch_bounds__(256)
sw_blend_buffered_kernel(const float* __restrict__ logits, const float* __restrict__ imp, float* __restrict__ out, int K,
                         int k0, int D, int H, int W, int rd, int rh, int rw, WindowGrid g, int bax, int bsteps, long long wstride) {
    const int wv = W / VEC;
    const long long total = (long long)D * H * wv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % wv) * VEC;
    const long long t = idx / wv;
    const int y = (int)(t % H), z = (int)(t / H);

    int lo[3], hi[3];
    cover(g.sz, g.nz, rd, z, lo[0], hi[0]);
    cover(g.sy, g.ny, rh, y, lo[1], hi[1]);
    cover(g.sx, g.nx, rw, x, lo[2], hi[2]);
    // loop nest: the buffered axis outermost, the other two in their row-major order
    const int a0 = bax, a1 = bax == 0 ? 1 : 0, a2 = bax == 2 ? 1 : 2;

    const long long plane = (long long)rh * rw, roi = plane * rd;
    float acc[KT][VEC], part[KT][VEC];
    float cnt[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        cnt[v] = 0.0f;
#pragma unroll
        for (int k = 0; k < KT; ++k) { acc[k][v] = 0.0f; part[k][v] = 0.0f; }
    }
    int group = lo[a0] / bsteps;
    int i[3];
    for (i[a0] = lo[a0]; i[a0] <= hi[a0]; ++i[a0]) {
        if (i[a0] / bsteps != group) {      // the slab is complete: output += buffer, a new buffer starts from zeros
            group = i[a0] / bsteps;
#pragma unroll
            for (int v = 0; v < VEC; ++v)
#pragma unroll
                for (int k = 0; k < KT; ++k) { acc[k][v] = __fadd_rn(acc[k][v], part[k][v]); part[k][v] = 0.0f; }
        }
        for (i[a1] = lo[a1]; i[a1] <= hi[a1]; ++i[a1]) {
            for (i[a2] = lo[a2]; i[a2] <= hi[a2]; ++i[a2]) {
                const int lz = z - g.sz[i[0]], ly = y - g.sy[i[1]], lx = x - g.sx[i[2]];
                const long long w = ((long long)i[0] * g.ny + i[1]) * g.nx + i[2];
                const long long off = (long long)lz * plane + (long long)ly * rw + lx;
                const float* lp = logits + w * wstride + k0 * roi + off;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
                    const float wt = imp[off + v];
#pragma unroll
                    for (int k = 0; k < KT; ++k) part[k][v] = __fadd_rn(part[k][v], PREMUL ? lp[k * roi + v] : __fmul_rn(lp[k * roi + v], wt));
                    cnt[v] = __fadd_rn(cnt[v], wt);
                }
            }
        }
    }
    const long long vox = (long long)D * H * W;
    float* op = out + (long long)k0 * vox + ((long long)z * H + y) * W + x;
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int v = 0; v < VEC; ++v) op[k * vox + v] = __fdiv_rn(__fadd_rn(acc[k][v], part[k][v]), cnt[v]);
}

// ---------------------------------------------------------------------------------------------------
// Regular window grids.  dense_patch_slices (monai/data/utils.py:166-206) only ever produces starts of the form
//   start(i) = i * step  for i < n - 1,   start(n - 1) = last <= (n - 1) * step      (the clip to image - roi)
// so the covering range of a coordinate has a closed form: no start tables in the kernel arguments, no per-thread table
// walk (a chain of dependent loads in front of the first logit load), and no limit on the number of windows per axis
// (SliceInferer: one window per slice).  `magic` = floor(2^32 / step) + 1 gives floor(p / step) = umulhi(p, magic) for
// p * step < 2^32 (the launcher checks it; 0 = use a real division, e.g. step 1).
struct AxisWin {
    int n, step, last;
    unsigned magic;
};
struct RegGrid {
    AxisWin z, y, x;
};

__device__ __forceinline__ int axis_start(const AxisWin& a, int i) { return i == a.n - 1 ? a.last : i * a.step; }
__device__ __forceinline__ int axis_div(const AxisWin& a, int p) { return a.magic ? (int)__umulhi((unsigned)p, a.magic) : p / a.step; }
// windows i with start(i) <= p < start(i) + r: a contiguous range [lo, hi] (every p is covered by at least one window)
__device__ __forceinline__ void axis_cover(const AxisWin& a, int r, int p, int& lo, int& hi) {
    lo = p >= r ? axis_div(a, p - r) + 1 : 0;
    if (lo > a.n - 1) lo = a.n - 1;
    hi = axis_div(a, p);
    if (hi > a.n - 2) hi = a.n - 2;
    if (p >= a.last) hi = a.n - 1;
    if (hi < lo) hi = lo;
}

template <int VEC>
__global__ void __launch_bounds__(256)
window_extract_reg_kernel(const float* __restrict__ vol, int C, int D, int H, int W, RegGrid g, int w0, int nwin, int rd, int rh,
                          int rw, float* __restrict__ out) {
    const int rwv = rw / VEC;
    const long long total = (long long)nwin * C * rd * rh * rwv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int xv = (int)(idx % rwv);
    long long t = idx / rwv;
    const int ly = (int)(t % rh); t /= rh;
    const int lz = (int)(t % rd); t /= rd;
    const int c = (int)(t % C);
    const int w = w0 + (int)(t / C);
    const int ix = w % g.x.n, iy = (w / g.x.n) % g.y.n, iz = w / (g.x.n * g.y.n);
    const int z = axis_start(g.z, iz) + lz, y = axis_start(g.y, iy) + ly, x = axis_start(g.x, ix) + xv * VEC;
    const float* src = vol + (((long long)c * D + z) * H + y) * W + x;
    float* dst = out + idx * VEC;
    if (VEC == 4) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(src);
    else dst[0] = src[0];
}

// The blend on a regular grid.  Same arithmetic, in the same order, as sw_blend_kernel (acc = acc + fp32(logit * w),
// cnt = cnt + w, out = acc / cnt, windows in ascending index), organised for the memory system: the covering box is walked
// G windows at a time -- the G x KT logit vectors and G weight vectors of a batch are all requested before the first add
// (the adds keep the window order) -- and the read-once logits / write-once output can bypass the caches (NT).
// `wstride` = floats between consecutive windows' logits (>= K * roi): a stride that is not a multiple of a large power of two
// spreads the 8 x K concurrently read (window, class) streams of a voxel over the HBM channels (profiles/r02_ubench_hbm_stream_v2.txt).
// ARGMAX = false writes channels [k0, k0 + KT); ARGMAX = true walks ALL K channels in chunks of KT inside the thread
// and writes only the label -- index of the first maximal blended value, NaN maximal, i.e. torch.argmax of the blended
// logits (AsDiscrete(argmax=True), monai/transforms/post/array.py:132-237) -- as float or uint8: K x 4 B per voxel of output
// traffic become 4 B or 1 B.
template <int KT, int VEC, int G, bool NT, bool ARGMAX>
__global__ void __launch_bounds__(256)
sw_blend_reg_kernel(const float* __restrict__ logits, const float* __restrict__ imp, void* __restrict__ out_, int K, int k0_, int D,
                    int H, int W, int rd, int rh, int rw, RegGrid g, int premul, int out_u8, long long wstride) {
    const int wv = W / VEC;
    const long long total = (long long)D * H * wv;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int x = (int)(idx % wv) * VEC;
    const long long t = idx / wv;
    const int y = (int)(t % H), z = (int)(t / H);

    int zlo, zhi, ylo, yhi, xlo, xhi;
    axis_cover(g.z, rd, z, zlo, zhi);
    axis_cover(g.y, rh, y, ylo, yhi);
    axis_cover(g.x, rw, x, xlo, xhi);     // VEC == 4: starts and rw are multiples of 4, x .. x+3 share the covering set
    const int nw = (zhi - zlo + 1) * (yhi - ylo + 1) * (xhi - xlo + 1);
    const long long plane = (long long)rh * rw, roi = plane * rd;
    const long long vox = (long long)D * H * W;
    const long long opos = ((long long)z * H + y) * W + x;

    float best[VEC];
    int besti[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { best[v] = 0.0f; besti[v] = 0; }

    const int kbeg = ARGMAX ? 0 : k0_, kend = ARGMAX ? K : k0_ + KT;
    for (int k0 = kbeg; k0 < kend; k0 += KT) {
        float acc[KT][VEC];
        float cnt[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            cnt[v] = 0.0f;
#pragma unroll
            for (int k = 0; k < KT; ++k) acc[k][v] = 0.0f;
        }
        int iz = zlo, iy = ylo, ix = xlo;
        for (int done = 0; done < nw; done += G) {
            float wt[G][VEC];
            float lv[G][KT][VEC];
            long long off0 = 0, base0 = 0;
#pragma unroll
            for (int b = 0; b < G; ++b) {
                const bool ok = done + b < nw;
                const int lz = z - axis_start(g.z, iz), ly = y - axis_start(g.y, iy), lx = x - axis_start(g.x, ix);
                const long long w = ((long long)iz * g.y.n + iy) * g.x.n + ix;
                long long off = (long long)lz * plane + (long long)ly * rw + lx;
                long long base = w * wstride + off;
                if (b == 0) { off0 = off; base0 = base; }
                if (!ok) { off = off0; base = base0; }        // past the end of the box: re-request the batch's first window (values unused)
                if (ok) {                                     // next window of the box, last axis fastest = ascending window index
                    if (++ix > xhi) { ix = xlo; if (++iy > yhi) { iy = ylo; ++iz; } }
                }
                if (VEC == 4) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(imp + off);
                    wt[b][0] = q[0]; wt[b][1] = q[1]; wt[b][2] = q[2]; wt[b][3] = q[3];
                } else {
                    wt[b][0] = imp[off];
                }
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    int kc = k0 + k;
                    if (ARGMAX && kc > K - 1) kc = K - 1;       // tail chunk of the argmax walk: clamped, ignored below
                    const float* lp = logits + base + (long long)kc * roi;
                    if (VEC == 4) {
                        const f32x4 a = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(lp)) : *reinterpret_cast<const f32x4*>(lp);
                        lv[b][k][0] = a[0]; lv[b][k][1] = a[1]; lv[b][k][2] = a[2]; lv[b][k][3] = a[3];
                    } else {
                        lv[b][k][0] = NT ? __builtin_nontemporal_load(lp) : *lp;
                    }
                }
            }
#pragma unroll
            for (int b = 0; b < G; ++b) {
                const bool ok = done + b < nw;
#pragma unroll
                for (int v = 0; v < VEC; ++v) {
#pragma unroll
                    for (int k = 0; k < KT; ++k) {
                        const float s = __fadd_rn(acc[k][v], __fmul_rn(lv[b][k][v], premul ? 1.0f : wt[b][v]));
                        acc[k][v] = ok ? s : acc[k][v];
                    }
                    const float c = __fadd_rn(cnt[v], wt[b][v]);
                    cnt[v] = ok ? c : cnt[v];
                }
            }
        }
        if (!ARGMAX) {
            float* op = reinterpret_cast<float*>(out_) + (long long)k0 * vox + opos;
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (VEC == 4) {
                    const f32x4 r = {__fdiv_rn(acc[k][0], cnt[0]), __fdiv_rn(acc[k][1], cnt[1]), __fdiv_rn(acc[k][2], cnt[2]), __fdiv_rn(acc[k][3], cnt[3])};
                    if (NT) __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(op + k * vox));
                    else *reinterpret_cast<f32x4*>(op + k * vox) = r;
                } else {
                    op[k * vox] = __fdiv_rn(acc[k][0], cnt[0]);
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                if (k0 + k < K) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) {
                        const float r = __fdiv_rn(acc[k][v], cnt[v]);
                        const bool take = (k0 + k == 0) || r > best[v] || (r != r && best[v] == best[v]);
                        best[v] = take ? r : best[v];
                        besti[v] = take ? k0 + k : besti[v];
                    }
                }
            }
        }
    }
    if (ARGMAX) {
        if (out_u8) {
            unsigned char* op = reinterpret_cast<unsigned char*>(out_) + opos;
            if (VEC == 4) *reinterpret_cast<unsigned*>(op) = (unsigned)besti[0] | ((unsigned)besti[1] << 8) | ((unsigned)besti[2] << 16) | ((unsigned)besti[3] << 24);
            else op[0] = (unsigned char)besti[0];
        } else {
            float* op = reinterpret_cast<float*>(out_) + opos;
            if (VEC == 4) *reinterpret_cast<f32x4*>(op) = f32x4{(float)besti[0], (float)besti[1], (float)besti[2], (float)besti[3]};
            else op[0] = (float)besti[0];
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Mosaic logits layout (single-GPU fused path).  In the window-major buffer a voxel's 8 ... 27 covering windows are 8 ... 27 x K
// separate streams of 384-byte row pieces; the blend kernel above tops out at 0.62-0.69 of 8 TB/s while a plain read stream reaches
// 0.82-0.87 on the same chip (tools/ubench/hbm_stream.hip).  Windows i and i + m of an axis never overlap when m * step >= roi, so the
// windows of one residue class (i mod m per axis; m a power of two; the clipped last window of each axis is a class of its own) tile
// space without overlap: the logits of class (cz, cy, cx) are stored as ONE dense array [K][cnt_z * rd][cnt_y * rh][cnt_x * rw] -- window
// (jz, jy, jx) of the class at its mosaic position.  At overlap 0.5 (m = 2) consecutive windows of a class are also consecutive in the
// volume, so a wave's 256 consecutive voxels read 1 KB runs from each of the <= 8 (27) class arrays instead of 384-byte pieces of different
// windows.  The predictor's last kernel writes the layout directly (conv1x1_windows_kernel); the arithmetic and its order (ascending
// window index) are those of sw_blend_reg_kernel: the same bits.
constexpr int MOSAIC_MAX_CLASSES = 5;          // m <= 4 residue classes + the last window, per axis
struct MosaicAxis {
    int log2m;
    int cnt[MOSAIC_MAX_CLASSES];               // windows per class
};
struct Mosaic {
    MosaicAxis z, y, x;
    long long base[MOSAIC_MAX_CLASSES * MOSAIC_MAX_CLASSES * MOSAIC_MAX_CLASSES];      // float offset of class (cz, cy, cx)
};
__host__ __device__ inline void mosaic_axis_fill(MosaicAxis& a, int n, int log2m) {
    a.log2m = log2m;
    const int m = 1 << log2m;
    for (int c = 0; c < MOSAIC_MAX_CLASSES; ++c) a.cnt[c] = 0;
    for (int c = 0; c < m; ++c) a.cnt[c] = n - 1 > c ? (n - 1 - c + m - 1) >> log2m : 0;       // i in [0, n - 1) with i mod m == c
    a.cnt[m] = 1;                                                                               // the last window
}
__device__ __forceinline__ void mosaic_axis_locate(const MosaicAxis& a, int n, int i, int& c, int& j) {
    const bool last = i == n - 1;
    c = last ? (1 << a.log2m) : (i & ((1 << a.log2m) - 1));
    j = last ? 0 : (i >> a.log2m);
}

// SEP: the importance map is given by its factors -- fac = [gz (rd) | gy (rh) | gx (rw) | floor]: map[z][y][x] = max(fl(fl(gz[z] * gy[y]) * gx[x]), floor), the
// way the reference builds the gaussian (and the constant) map (monai/data/utils.py:1084-1134) -- and is re-formed in registers, bit for bit: three
// vectors of roi floats that stay in L1 instead of 8 ... 27 loads per output voxel group from a roi^3 map that competes with the logits stream for L2.
// Round 4 (profiles/r04_pmc_hbm_kernels.txt: the vector ALUs were busy for 0.58 of the launch -- 1 856 vector instructions per wave, the blend's own arithmetic a fifth of
// them; the rest was 64-bit index arithmetic, table lookups and selects).  FAST (the launcher proves the bounds, else the generic 64-bit form runs):
//   * a voxel's offset inside its class array and the array's channel stride are 32-bit, formed by 24-bit multiplies (full rate; v_mul_lo_u32 / v_mad_u64_u32 are
//     quarter rate) -- every factor is an extent or a coordinate below 2^24, (z-extent x y-extent) < 2^24, K x stride < 2^31 floats; the K channel pointers are one
//     64-bit add each (the stride is made opaque: the compiler otherwise re-derives every channel's offset through the whole multiply chain);
//   * the class extents come from the class index by arithmetic (mosaic_axis_extent), not from three per-lane table loads per window in front of the logits loads;
//   * whole batches of B windows and the <= B - 1 windows behind them are separate code (mosaic_blend_batch<B> / <1>): no "is this window real" select on every
//     accumulate, and the accumulates pack (v_pk_mul_f32 / v_pk_add_f32);
//   * (z, y, x) from a 2-D grid (blockIdx.y = z) instead of two 64-bit divisions per thread.
// Same operations on the same values in the same order: the same bits.
// the instructions themselves (operands: the low 24 bits; result: the low 32 bits of the product): __mul24 / __umul24 are shift patterns that the optimiser turned back
// into quarter-rate 32-bit multiplies in this kernel.  The SIMT emulator keeps the 24-bit truncation, so an operand out of range shows in the tests.
#ifdef MH_SIMT_EMULATOR
__device__ __forceinline__ int mul24(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ unsigned mos_mul(unsigned a, unsigned b) { return __umul24(a, b); }
#else
__device__ __forceinline__ int mul24(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ unsigned mos_mul(unsigned a, unsigned b) { unsigned r; asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
#endif
__device__ __forceinline__ long long mos_mul(long long a, long long b) { return a * b; }
template <bool FAST> __device__ __forceinline__ int axis_start_t(const AxisWin& a, int i) { return i == a.n - 1 ? a.last : (FAST ? mul24(i, a.step) : i * a.step); }
// windows per class times the window extent: cnt[c] * r of mosaic_axis_fill without the table (c < m: windows i = c, c + m, ... below n - 1; the last window: one)
__device__ __forceinline__ int mosaic_axis_extent(const MosaicAxis& a, int n, int c, bool last, int r) {
    const int cnt = last ? 1 : (n - 2 - c + (1 << a.log2m)) >> a.log2m;
    return mul24(cnt, r);
}

template <int KT, int B, bool NT, bool SEP, bool FAST>
__device__ __forceinline__ void mosaic_blend_batch(const float* __restrict__ logits, const float* __restrict__ imp, int z, int y, int x, int rd, int rh, int rw,
                                                   const RegGrid& g, const Mosaic& ms, int ylo, int yhi, int xlo, int xhi, int& iz, int& iy, int& ix,
                                                   float (&acc)[KT][4], float (&cnt)[4]) {
    using IT = typename std::conditional<FAST, unsigned, long long>::type;
    float wt[B][4];
    f32x4 lv[B][KT];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const int lz = z - axis_start_t<FAST>(g.z, iz), ly = y - axis_start_t<FAST>(g.y, iy), lx = x - axis_start_t<FAST>(g.x, ix);
        int cz, jz, cy, jy, cx, jx;
        mosaic_axis_locate(ms.z, g.z.n, iz, cz, jz);
        mosaic_axis_locate(ms.y, g.y.n, iy, cy, jy);
        mosaic_axis_locate(ms.x, g.x.n, ix, cx, jx);
        const IT Dz = (IT)mosaic_axis_extent(ms.z, g.z.n, cz, iz == g.z.n - 1, rd), Hc = (IT)mosaic_axis_extent(ms.y, g.y.n, cy, iy == g.y.n - 1, rh),
                 Wc = (IT)mosaic_axis_extent(ms.x, g.x.n, cx, ix == g.x.n - 1, rw);
        const IT pos = mos_mul(mos_mul((IT)(mul24(jz, rd) + lz), Hc) + (IT)(mul24(jy, rh) + ly), Wc) + (IT)(mul24(jx, rw) + lx);
        unsigned long long cs = (unsigned long long)mos_mul(mos_mul(Dz, Hc), Wc);
        MH_OPAQUE(cs);
        const float* lp = logits + ms.base[mul24(mul24(cz, MOSAIC_MAX_CLASSES) + cy, MOSAIC_MAX_CLASSES) + cx] + pos;
        if (++ix > xhi) { ix = xlo; if (++iy > yhi) { iy = ylo; ++iz; } }      // next window of the box, last axis fastest = ascending window index
        if (SEP) {
            const float zy = __fmul_rn(imp[lz], imp[rd + ly]), fl_ = imp[rd + rh + rw];
            const f32x4 q = *reinterpret_cast<const f32x4*>(imp + rd + rh + lx);
#pragma unroll
            for (int v = 0; v < 4; ++v) wt[b][v] = fmaxf(__fmul_rn(zy, q[v]), fl_);
        } else {
            const f32x4 q = *reinterpret_cast<const f32x4*>(imp + (unsigned)(mul24(mul24(lz, rh) + ly, rw) + lx));
            wt[b][0] = q[0]; wt[b][1] = q[1]; wt[b][2] = q[2]; wt[b][3] = q[3];
        }
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const f32x4* lp_ = reinterpret_cast<const f32x4*>(lp);
            lv[b][k] = NT ? __builtin_nontemporal_load(lp_) : *lp_;
            lp += cs;
        }
    }
#pragma unroll
    for (int b = 0; b < B; ++b)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int k = 0; k < KT; ++k) acc[k][v] = __fadd_rn(acc[k][v], __fmul_rn(lv[b][k][v], wt[b][v]));
            cnt[v] = __fadd_rn(cnt[v], wt[b][v]);
        }
}

// grid: x over the (y, x-vector) pairs of a plane, y = z; a thread owns all K = KT classes of its four voxels
template <int KT, int G, bool NT, bool SEP, bool FAST = true>
__global__ void __launch_bounds__(256)
sw_blend_mosaic_kernel(const float* __restrict__ logits, const float* __restrict__ imp, float* __restrict__ out, int K, int D, int H, int W, int rd,
                       int rh, int rw, RegGrid g, Mosaic ms) {
    constexpr int VEC = 4;
    const int z = (int)blockIdx.y;
    const unsigned wv = (unsigned)W / VEC;
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= (unsigned)H * wv) return;
    const int y = (int)(i / wv), x = (int)(i - (unsigned)y * wv) * VEC;
    int zlo, zhi, ylo, yhi, xlo, xhi;
    axis_cover(g.z, rd, z, zlo, zhi);
    axis_cover(g.y, rh, y, ylo, yhi);
    axis_cover(g.x, rw, x, xlo, xhi);
    const int nw = (zhi - zlo + 1) * (yhi - ylo + 1) * (xhi - xlo + 1);
    const long long vox = (long long)D * H * W;
    float acc[KT][VEC];
    float cnt[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
        cnt[v] = 0.0f;
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[k][v] = 0.0f;
    }
    int iz = zlo, iy = ylo, ix = xlo, done = 0;
    for (; done + G <= nw; done += G) mosaic_blend_batch<KT, G, NT, SEP, FAST>(logits, imp, z, y, x, rd, rh, rw, g, ms, ylo, yhi, xlo, xhi, iz, iy, ix, acc, cnt);
    for (; done < nw; ++done) mosaic_blend_batch<KT, 1, NT, SEP, FAST>(logits, imp, z, y, x, rd, rh, rw, g, ms, ylo, yhi, xlo, xhi, iz, iy, ix, acc, cnt);
    float* op = out + ((long long)z * H + y) * W + x;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const f32x4 r = {__fdiv_rn(acc[k][0], cnt[0]), __fdiv_rn(acc[k][1], cnt[1]), __fdiv_rn(acc[k][2], cnt[2]), __fdiv_rn(acc[k][3], cnt[3])};
        if (NT) __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(op + k * vox));
        else *reinterpret_cast<f32x4*>(op + k * vox) = r;
    }
}

// AvgMerger (monai/inferers/merger.py:103-205): `values[slice] += patch; counts[slice] += 1` for one patch -- the patches
// of a PatchInferer arrive one by one through a user-visible Merger object, so the accumulation order (= patch order) is
// the reference's -- and the final `values /= counts`.  One thread per patch element, lanes along x; HBM-bound read-modify-
// write of the patch footprint (12 B per element + 2 B of counts).
__global__ void __launch_bounds__(256)
patch_accumulate_kernel(float* __restrict__ values, unsigned char* __restrict__ counts, const float* __restrict__ patch, int NC, int D, int H, int W,
                        int pd, int ph, int pw, int z0, int y0, int x0) {
    const long long pvol = (long long)pd * ph * pw;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= pvol * NC) return;
    const long long c = idx / pvol, r = idx - c * pvol;
    const int x = (int)(r % pw), y = (int)((r / pw) % ph), z = (int)(r / ((long long)pw * ph));
    const long long o = ((c * D + (z0 + z)) * H + (y0 + y)) * W + (x0 + x);
    values[o] += patch[idx];
    counts[o] = (unsigned char)(counts[o] + 1);
}

// A whole batch of patches in ONE launch (what a PatchInferer hands over per network call): gather form -- one thread per element of the
// batch's bounding box in the merged volume walks the patches in batch order and adds those that cover it, so overlapping patches of a
// batch meet in one thread, in the reference's order (patch by patch): same bits as `np` single-patch launches, no atomics.
constexpr int PATCH_BATCH_MAX = 64;
struct PatchBatch {
    int n;
    int z[PATCH_BATCH_MAX], y[PATCH_BATCH_MAX], x[PATCH_BATCH_MAX];
};
__global__ void __launch_bounds__(256)
patch_accumulate_batch_kernel(float* __restrict__ values, unsigned char* __restrict__ counts, const float* __restrict__ patches, int NC, int D, int H,
                              int W, int pd, int ph, int pw, PatchBatch b, int bz0, int by0, int bx0, int bd, int bh, int bw) {
    const long long bvol = (long long)bd * bh * bw;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= bvol * NC) return;
    const long long c = idx / bvol, r = idx - c * bvol;
    const int x = bx0 + (int)(r % bw), y = by0 + (int)((r / bw) % bh), z = bz0 + (int)(r / ((long long)bw * bh));
    const long long o = ((c * D + z) * H + y) * W + x;
    const long long pvol = (long long)pd * ph * pw;
    float v = values[o];
    unsigned cnt = counts[o];
    bool hit = false;
    for (int p = 0; p < b.n; ++p) {
        const int lz = z - b.z[p], ly = y - b.y[p], lx = x - b.x[p];
        if (lz >= 0 && lz < pd && ly >= 0 && ly < ph && lx >= 0 && lx < pw) {
            v += patches[((long long)p * NC + c) * pvol + ((long long)lz * ph + ly) * pw + lx];
            ++cnt;
            hit = true;
        }
    }
    if (hit) { values[o] = v; counts[o] = (unsigned char)cnt; }
}

__global__ void __launch_bounds__(256)
avg_finalize_kernel(float* __restrict__ values, const unsigned char* __restrict__ counts, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) values[i] = values[i] / (float)counts[i];
}

}  // namespace mh
