// Micro-benchmark (development aid, not product code): what a streaming kernel can reach on this MI355X's HBM, as the
// ceiling the HBM-bound kernels (blend, resample, Gaussian) are judged against -- instead of torch's copy_.
//   hipcc --offload-arch=gfx950 -O3 hbm_stream.hip -o /tmp/hbm_stream && /tmp/hbm_stream
// Modes: float4 copy (1 / 4 / 8 vectors per thread, plain and non-temporal, one-shot and persistent grids), read-only
// reduction, write-only fill, and the blend's shape: many concurrent read streams of 384-byte rows (17.7 GB read once,
// 2.7 GB written) in the order the gather blend touches them.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_k(const f4* __restrict__ src, f4* __restrict__ dst, long long n) {
    // block-contiguous: a block owns U * 256 consecutive vectors
    long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long i = base + 256LL * u;
        if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long i = base + 256LL * u;
        if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
    }
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) copy_persist_k(const f4* __restrict__ src, f4* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * (256 * U);
    for (long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x; base < n; base += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = base + 256LL * u;
            if (i < n) v[u] = NT ? __builtin_nontemporal_load(src + i) : src[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long i = base + 256LL * u;
            if (i < n) { if (NT) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; }
        }
    }
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) read_k(const f4* __restrict__ src, float* __restrict__ sink, long long n) {
    long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long i = base + 256LL * u;
        v[u] = i < n ? (NT ? __builtin_nontemporal_load(src + i) : src[i]) : f4{0, 0, 0, 0};
    }
    f4 s = v[0];
#pragma unroll
    for (int u = 1; u < U; ++u) s += v[u];
    if (s[0] + s[1] + s[2] + s[3] == 12345.678f) sink[threadIdx.x] = 1.0f;   // never true for the fill value
}

template <int U, bool NT>
__global__ void __launch_bounds__(256) write_k(f4* __restrict__ dst, long long n, float val) {
    long long base = (long long)blockIdx.x * (256 * U) + threadIdx.x;
    const f4 v = {val, val, val, val};
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long long i = base + 256LL * u;
        if (i < n) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
    }
}

// The blend's access shape without its arithmetic: output [K][D][H][W], windows of r^3 at regular starts (step s, last
// clamped), logits [nwin][K][r^3].  One thread = 4 x-voxels x K classes; it reads, for every covering window, K float4 and
// writes K float4.  ORDER 0: blocks walk the volume linearly (2 rows per block); ORDER 1: blocks walk window-cell by
// window-cell (s x s x W bricks), so fewer distinct windows are live at once (TLB / DRAM page locality probe).
template <int K, bool NT, int ORDER>
__global__ void __launch_bounds__(256) blend_shape_k(const float* __restrict__ logits, float* __restrict__ out, int D, int H, int W, int r, int s,
                                                     int n, long long kstride = 0, long long wstride = 0, long long ostride = 0) {
    const int wv = W / 4;
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    int x, y, z;
    if (ORDER == 0) {
        x = (int)(idx % wv) * 4;
        const long long t = idx / wv;
        y = (int)(t % H); z = (int)(t / H);
    } else {
        // bricks of s (z) x s (y) x W: idx -> (brick, zz, yy, xv)
        x = (int)(idx % wv) * 4;
        long long t = idx / wv;
        const int yy = (int)(t % s); t /= s;
        const int zz = (int)(t % s); t /= s;
        const int by = (int)(t % ((H + s - 1) / s)), bz = (int)(t / ((H + s - 1) / s));
        y = by * s + yy; z = bz * s + zz;
    }
    if (z >= D || y >= H) return;
    const int last = (n - 1) * s < D - r ? (n - 1) * s : D - r;     // D == H == W in this probe
    int lo[3], hi[3];
    const int p[3] = {z, y, x};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int l = p[a] - r >= 0 ? (p[a] - r) / s + 1 : 0;
        if (l > n - 1) l = n - 1;
        int h = p[a] / s;
        if (h > n - 2) h = n - 2;
        if (p[a] >= last) h = n - 1;
        if (h < l) h = l;
        lo[a] = l; hi[a] = h;
    }
    const long long plane = (long long)r * r, roi = kstride ? kstride : plane * r;
    const long long wst = wstride ? wstride : K * roi;
    f4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = f4{0, 0, 0, 0};
    for (int iz = lo[0]; iz <= hi[0]; ++iz) {
        const int lz = z - (iz == n - 1 ? last : iz * s);
        for (int iy = lo[1]; iy <= hi[1]; ++iy) {
            const int ly = y - (iy == n - 1 ? last : iy * s);
            for (int ix = lo[2]; ix <= hi[2]; ++ix) {
                const int lx = x - (ix == n - 1 ? last : ix * s);
                const long long w = ((long long)iz * n + iy) * n + ix;
                const float* lp = logits + w * wst + lz * plane + (long long)ly * r + lx;
                f4 v[K];
#pragma unroll
                for (int k = 0; k < K; ++k) v[k] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f4*>(lp + k * roi)) : *reinterpret_cast<const f4*>(lp + k * roi);
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] += v[k];
            }
        }
    }
    const long long vox = ostride ? ostride : (long long)D * H * W;
    float* op = out + ((long long)z * H + y) * W + x;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (NT) __builtin_nontemporal_store(acc[k], reinterpret_cast<f4*>(op + k * vox));
        else *reinterpret_cast<f4*>(op + k * vox) = acc[k];
    }
}

// The blend shape with the classes interleaved at ROW granularity: logits [nwin][r][r][K][r] -- the K class values of a window
// row are K consecutive 384-byte rows, so a voxel's K loads per window fall into one 1920-byte span (8 streams per voxel
// instead of 8 K).
template <int K>
__global__ void __launch_bounds__(256) blend_shape_rows_k(const float* __restrict__ logits, float* __restrict__ out, int D, int H, int W, int r, int s,
                                                          int n, long long wstride) {
    const int wv = W / 4;
    long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int x = (int)(idx % wv) * 4;
    const long long t = idx / wv;
    const int y = (int)(t % H), z = (int)(t / H);
    if (z >= D) return;
    const int last = (n - 1) * s < D - r ? (n - 1) * s : D - r;
    int lo[3], hi[3];
    const int p[3] = {z, y, x};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int l = p[a] - r >= 0 ? (p[a] - r) / s + 1 : 0;
        if (l > n - 1) l = n - 1;
        int h = p[a] / s;
        if (h > n - 2) h = n - 2;
        if (p[a] >= last) h = n - 1;
        if (h < l) h = l;
        lo[a] = l; hi[a] = h;
    }
    f4 acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = f4{0, 0, 0, 0};
    for (int iz = lo[0]; iz <= hi[0]; ++iz) {
        const int lz = z - (iz == n - 1 ? last : iz * s);
        for (int iy = lo[1]; iy <= hi[1]; ++iy) {
            const int ly = y - (iy == n - 1 ? last : iy * s);
            for (int ix = lo[2]; ix <= hi[2]; ++ix) {
                const int lx = x - (ix == n - 1 ? last : ix * s);
                const long long w = ((long long)iz * n + iy) * n + ix;
                const float* lp = logits + w * wstride + ((long long)(lz * r + ly) * K) * r + lx;
                f4 v[K];
#pragma unroll
                for (int k = 0; k < K; ++k) v[k] = *reinterpret_cast<const f4*>(lp + k * r);
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] += v[k];
            }
        }
    }
    const long long vox = (long long)D * H * W;
    float* op = out + ((long long)z * H + y) * W + x;
#pragma unroll
    for (int k = 0; k < K; ++k) *reinterpret_cast<f4*>(op + k * vox) = acc[k];
}

// S streams `stride` bytes apart, read in lockstep: thread i reads 16 B at offset 16 i of every stream (the blend reads its 40
// (window, class) streams like this: are strides that are multiples of a large power of two slower -- channel camping?)
template <int S>
__global__ void __launch_bounds__(256) strided_read_k(const char* __restrict__ base, long long stride, long long nvec, float* __restrict__ sink) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    f4 v[S];
#pragma unroll
    for (int q = 0; q < S; ++q) v[q] = *reinterpret_cast<const f4*>(base + q * stride + i * 16);
    f4 a = v[0];
#pragma unroll
    for (int q = 1; q < S; ++q) a += v[q];
    if (a[0] + a[1] + a[2] + a[3] == 12345.678f) sink[threadIdx.x] = 1.0f;
}

static double time_ms(hipEvent_t e0, hipEvent_t e1) { float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms; }

#define RUN(NAME, BYTES, REPS, LAUNCH)                                                              \
    {                                                                                               \
        LAUNCH; CHECK(hipDeviceSynchronize());                                                      \
        CHECK(hipEventRecord(e0)); for (int i_ = 0; i_ < (REPS); ++i_) { LAUNCH; }                  \
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());        \
        const double ms_ = time_ms(e0, e1) / (REPS);                                                \
        printf("%-66s %8.3f ms  %8.1f GB/s  (%.3f of 8 TB/s)\n", NAME, ms_, (BYTES) / ms_ * 1e-6, (BYTES) / ms_ * 1e-6 / 8000.0); \
    }

int main() {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long long nvec = (4LL << 30) / 16;          // 4 GiB per buffer (>> the 256 MB Infinity Cache)
    f4 *a, *b; float* sink;
    CHECK(hipMalloc(&a, nvec * 16)); CHECK(hipMalloc(&b, nvec * 16)); CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMemset(a, 0x3c, nvec * 16)); CHECK(hipMemset(b, 0, nvec * 16));
    const double cb = 2.0 * nvec * 16, rb = 1.0 * nvec * 16;
#define NB(U) (unsigned)((nvec + 256LL * (U) - 1) / (256LL * (U)))
    RUN("copy float4, 1 vector per thread", cb, 5, (copy_k<1, false><<<NB(1), 256>>>(a, b, nvec)))
    RUN("copy float4, 4 vectors per thread", cb, 5, (copy_k<4, false><<<NB(4), 256>>>(a, b, nvec)))
    RUN("copy float4, 8 vectors per thread", cb, 5, (copy_k<8, false><<<NB(8), 256>>>(a, b, nvec)))
    RUN("copy float4, 4 vectors per thread, non-temporal", cb, 5, (copy_k<4, true><<<NB(4), 256>>>(a, b, nvec)))
    RUN("copy float4, 8 vectors per thread, non-temporal", cb, 5, (copy_k<8, true><<<NB(8), 256>>>(a, b, nvec)))
    RUN("copy persistent 2048 blocks x 4 vectors", cb, 5, (copy_persist_k<4, false><<<2048, 256>>>(a, b, nvec)))
    RUN("copy persistent 2048 blocks x 4 vectors, non-temporal", cb, 5, (copy_persist_k<4, true><<<2048, 256>>>(a, b, nvec)))
    RUN("copy persistent 4096 blocks x 4 vectors, non-temporal", cb, 5, (copy_persist_k<4, true><<<4096, 256>>>(a, b, nvec)))
    RUN("read-only float4, 4 vectors per thread", rb, 5, (read_k<4, false><<<NB(4), 256>>>(a, sink, nvec)))
    RUN("read-only float4, 8 vectors per thread", rb, 5, (read_k<8, false><<<NB(8), 256>>>(a, sink, nvec)))
    RUN("read-only float4, 8 vectors per thread, non-temporal", rb, 5, (read_k<8, true><<<NB(8), 256>>>(a, sink, nvec)))
    RUN("write-only float4, 4 vectors per thread", rb, 5, (write_k<4, false><<<NB(4), 256>>>(b, nvec, 1.0f)))
    RUN("write-only float4, 4 vectors per thread, non-temporal", rb, 5, (write_k<4, true><<<NB(4), 256>>>(b, nvec, 1.0f)))
    CHECK(hipFree(a)); CHECK(hipFree(b));

    // the blend's shape at the BASELINE sizes: 512^3, r = 96, step 48, 10 windows per axis, K = 5
    const int D = 512, r = 96, s = 48, n = 10, K = 5;
    const long long roi = (long long)r * r * r, nlog = 1000LL * K * roi, nout = (long long)K * D * D * D;
    float *lg, *out;
    CHECK(hipMalloc(&sink, 4096));
    CHECK(hipMalloc(&lg, nlog * 4)); CHECK(hipMalloc(&out, nout * 4));
    CHECK(hipMemset(lg, 0x3c, nlog * 4));
    const double bb = 4.0 * (nlog + nout);
    const unsigned nb0 = (unsigned)(((long long)D * D * (D / 4) + 255) / 256);
    const int bricks = (D + s - 1) / s;
    const unsigned nb1 = (unsigned)(((long long)bricks * bricks * s * s * (D / 4) + 255) / 256);
    RUN("blend shape (no weights), linear block order", bb, 5, (blend_shape_k<5, false, 0><<<nb0, 256>>>(lg, out, D, D, D, r, s, n)))
    RUN("blend shape (no weights), linear block order, non-temporal", bb, 5, (blend_shape_k<5, true, 0><<<nb0, 256>>>(lg, out, D, D, D, r, s, n)))
    RUN("blend shape (no weights), brick block order", bb, 5, (blend_shape_k<5, false, 1><<<nb1, 256>>>(lg, out, D, D, D, r, s, n)))
    RUN("blend shape (no weights), brick block order, non-temporal", bb, 5, (blend_shape_k<5, true, 1><<<nb1, 256>>>(lg, out, D, D, D, r, s, n)))
    CHECK(hipFree(lg)); CHECK(hipFree(out));

    // stream-stride probes: 40 streams of 96 MB each
    {
        const long long per = 96LL << 20, nv = per / 16;
        const long long strides[] = {per, per + 256, per + 4096 + 256, per + (1 << 16) + 256, per + (1 << 17)};
        char* buf; CHECK(hipMalloc(&buf, 40 * (per + (1 << 17)) + (1 << 20))); CHECK(hipMemset(buf, 0x3c, 40 * (per + (1 << 17))));
        for (long long st : strides) {
            if (st < per) continue;
            char nm[128]; snprintf(nm, sizeof nm, "40 lock-step read streams, stride %lld B (= 2^%d x %lld)", st, __builtin_ctzll(st), st >> __builtin_ctzll(st));
            RUN(nm, 40.0 * per, 3, (strided_read_k<40><<<(unsigned)((nv + 255) / 256), 256>>>(buf, st, nv, sink)))
        }
        CHECK(hipFree(buf));
    }
    // the blend shape with padded strides: class stride roi + pad, window stride K * (roi + pad) + pad2, output channel stride vox + pad
    {
        const long long pads[][3] = {{0, 0, 0}, {0, 1088, 0}, {64, 1088, 0}, {1088, 1088, 0}, {0, 1088, 0}, {64, 1088, 0}, {1088, 1088, 0}, {0, 0, 0}};
        for (auto& pd : pads) {
            const long long ks = roi + pd[0], ws = K * ks + pd[1], os = (long long)D * D * D + pd[2];
            float *lg2, *out2;
            CHECK(hipMalloc(&lg2, 1000LL * ws * 4)); CHECK(hipMalloc(&out2, K * os * 4));
            CHECK(hipMemset(lg2, 0x3c, 1000LL * ws * 4));
            char nm[160]; snprintf(nm, sizeof nm, "blend shape, class stride roi+%lld, window stride +%lld, out channel stride +%lld floats", pd[0], pd[1], pd[2]);
            RUN(nm, bb, 5, (blend_shape_k<5, false, 0><<<nb0, 256>>>(lg2, out2, D, D, D, r, s, n, ks, ws, os)))
            CHECK(hipFree(lg2)); CHECK(hipFree(out2));
        }
    }
    {
        const long long pads2[] = {0, 1088, 0, 1088};
        for (long long pd : pads2) {
            const long long ws = (long long)K * roi + pd;
            float *lg2, *out2;
            CHECK(hipMalloc(&lg2, 1000LL * ws * 4)); CHECK(hipMalloc(&out2, (long long)K * D * D * D * 4));
            CHECK(hipMemset(lg2, 0x3c, 1000LL * ws * 4));
            char nm[160]; snprintf(nm, sizeof nm, "blend shape, classes interleaved per window row, window stride +%lld floats", pd);
            RUN(nm, bb, 5, (blend_shape_rows_k<5><<<nb0, 256>>>(lg2, out2, D, D, D, r, s, n, ws)))
            CHECK(hipFree(lg2)); CHECK(hipFree(out2));
        }
    }
    return 0;
}
