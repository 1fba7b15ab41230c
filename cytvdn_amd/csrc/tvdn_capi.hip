// libtvdn_hip.so: C ABI plumbing, the one-pass ("kernel-level") HIP kernels and the synthetic
// input generator.  The fused iteration lives in tvdn_fused.hip.
//
// Arithmetic contract: every operation in the array dtype, left to right
// as the reference writes it, no fused multiply-add (this file is compiled with
// -ffp-contract=off), f32 denormals kept (hipcc default for gfx950).  Reductions are kept in
// f64 by a fixed tree, so they are deterministic and, unlike the reference's OpenMP sums,
// independent of any thread count.
#include <cstdarg>

#include "tvdn_common.hpp"
#include "tvdn_synth_tables.h"

namespace tvdn {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// ---- second reduction stage ------------------------------------------------------------------
__global__ void __launch_bounds__(1024) finalize_kernel(const double *partials, int nblocks, int nv,
                                                         double *out, int accumulate)
{
    __shared__ double red[kPartialWidth][16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int q = 0; q < nv; ++q) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nblocks; i += 1024) s += partials[(size_t)i * kPartialWidth + q];
        s = wave_sum(s);
        if (lane == 0) red[q][w] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < nv) {
        double s = 0.0;
        for (int i = 0; i < 16; ++i) s += red[threadIdx.x][i];
        out[threadIdx.x] = accumulate ? out[threadIdx.x] + s : s;
    }
}

int launch_finalize(tvdn_ctx *ctx, int nblocks, int nv, double *out, hipStream_t s, bool accumulate)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(1024), 0, s, ctx->partials, nblocks, nv, out, accumulate ? 1 : 0);
    TVDN_HIP(hipGetLastError());
    return TVDN_OK;
}

// ---- one-pass kernels ------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T clipval(T a, T val)
{
    // two ternaries exactly as the reference's generated C (anisotropic.c:2423-2437): NaN passes
    const T lo = -val;
    const T t = (lo > a) ? lo : a;
    return (val < t) ? val : t;
}

constexpr int kBlock = 256;

// accumulator_update_{3D,4D}[_FISTA]: cyTVDN/anisotropic.pyx:17-84, :89-164, :169-237, :243-317.
// The reference's serial boundary hyperslab (:56-82) is folded in as a predicated `prev` offset.
template <typename T, bool FISTA>
__global__ void __launch_bounds__(kBlock) acc_update_kernel(const T *__restrict__ a, T *__restrict__ b,
                                                             T *__restrict__ d, T tk, T clip,
                                                             long long total, long long stride,
                                                             long long n_ax, long long edge_delta,
                                                             double *partials)
{
    double acc[1] = {0.0};
    const long long step = (long long)gridDim.x * kBlock;
    for (long long x = (long long)blockIdx.x * kBlock + threadIdx.x; x < total; x += step) {
        const long long c = (x / stride) % n_ax;
        const long long p = (c > 0) ? x - stride : x + edge_delta;
        const T v = (a[x] - a[p]) + b[x];
        const T dn = clipval(v, clip);
        T bn = dn;
        if (FISTA) {
            bn = dn + tk * (dn - d[x]);
            d[x] = dn;
        }
        b[x] = bn;
        acc[0] += fabs((double)bn);
    }
    block_store_partials<1, kBlock>(acc, partials);
}

template <typename T>
struct ReconArgs {
    const T *orig;
    T *recon;
    const T *b[4];
    T lm[4];
    long long n[4], st[4];
    int nax;
    long long total;
};

// datacube_update_{3D,4D}: cyTVDN/utils.pyx:54-125, :131-199 (periodic-wrap branch, BC 0 and 2);
// association of the sum from the generated C, utils.c:5641.
template <typename T, int NAX>
__global__ void __launch_bounds__(kBlock) recon_update_kernel(ReconArgs<T> p, double *partials)
{
    double acc[2] = {0.0, 0.0};
    const long long step = (long long)gridDim.x * kBlock;
    for (long long x = (long long)blockIdx.x * kBlock + threadIdx.x; x < p.total; x += step) {
        long long idx[4];
        long long rem = x;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            idx[q] = rem / p.st[q];
            rem -= idx[q] * p.st[q];
        }
        T s = (T)0;
#pragma unroll
        for (int q = 0; q < NAX; ++q) {
            const int axq = (NAX == 4) ? q : (q == 0 ? 0 : q + 1);
            const long long nx = (idx[axq] + 1 < p.n[axq]) ? x + p.st[axq] : x - idx[axq] * p.st[axq];
            const T term = p.lm[q] * (p.b[q][x] - p.b[q][nx]);
            s = (q == 0) ? term : (s + term);
        }
        const T old = p.recon[x];
        const T nw = p.orig[x] - s;
        p.recon[x] = nw;
        const T df = nw - old;
        acc[0] += fabs((double)df);
        acc[1] += fabs((double)old);
    }
    block_store_partials<2, kBlock>(acc, partials);
}

// sum_square_error_{3D,4D}: cyTVDN/utils.pyx:14-30, :35-49.
template <typename T>
__global__ void __launch_bounds__(kBlock) sse_kernel(const T *__restrict__ a, const T *__restrict__ b,
                                                      long long total, double *partials)
{
    double acc[1] = {0.0};
    const long long step = (long long)gridDim.x * kBlock;
    for (long long x = (long long)blockIdx.x * kBlock + threadIdx.x; x < total; x += step) {
        const T t = a[x] - b[x];
        const T sq = t * t;
        acc[0] += (double)sq;
    }
    block_store_partials<1, kBlock>(acc, partials);
}

static int grid_for(long long total)
{
    long long g = (total + kBlock - 1) / kBlock;
    const long long cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride beyond that
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

static int check_common(tvdn_ctx *ctx, int dtype, int ndim, const int64_t *shape)
{
    TVDN_REQUIRE(ctx != nullptr, "ctx is NULL");
    TVDN_REQUIRE(dtype == TVDN_F32 || dtype == TVDN_F64, "dtype must be TVDN_F32 or TVDN_F64, got %d", dtype);
    TVDN_REQUIRE(ndim == 3 || ndim == 4, "ndim must be 3 or 4, got %d", ndim);
    TVDN_REQUIRE(shape != nullptr, "shape is NULL");
    for (int i = 0; i < ndim; ++i) TVDN_REQUIRE(shape[i] >= 1, "shape[%d] = %lld must be >= 1", i, (long long)shape[i]);
    return TVDN_OK;
}

template <typename T>
static int acc_update_impl(tvdn_ctx *ctx, const Geom &g, const void *a, void *b, void *d, double tk, int cax,
                           double clip, int bc_mode, double *norm_out, hipStream_t s)
{
    const long long n_ax = g.n[cax], stride = g.st[cax];
    long long edge = 0;
    if (bc_mode == TVDN_BC_PERIODIC) edge = (n_ax - 1) * stride;
    if (bc_mode == TVDN_BC_MIRROR) edge = stride;
    const int grid = grid_for(g.total);
    if (d)
        hipLaunchKernelGGL((acc_update_kernel<T, true>), dim3(grid), dim3(kBlock), 0, s, (const T *)a, (T *)b,
                           (T *)d, (T)tk, (T)clip, g.total, stride, n_ax, edge, ctx->partials);
    else
        hipLaunchKernelGGL((acc_update_kernel<T, false>), dim3(grid), dim3(kBlock), 0, s, (const T *)a, (T *)b,
                           (T *)nullptr, (T)tk, (T)clip, g.total, stride, n_ax, edge, ctx->partials);
    TVDN_HIP(hipGetLastError());
    return launch_finalize(ctx, grid, 1, norm_out, s);
}

template <typename T>
static int recon_update_impl(tvdn_ctx *ctx, const Geom &g, const void *orig, void *recon, const void *const *b,
                             const double *lm, double *sums_out, hipStream_t s)
{
    ReconArgs<T> p;
    p.orig = (const T *)orig;
    p.recon = (T *)recon;
    for (int q = 0; q < 4; ++q) {
        p.b[q] = q < g.nax ? (const T *)b[q] : nullptr;
        p.lm[q] = q < g.nax ? (T)lm[q] : (T)0;
        p.n[q] = g.n[q];
        p.st[q] = g.st[q];
    }
    p.nax = g.nax;
    p.total = g.total;
    const int grid = grid_for(g.total);
    if (g.nax == 4)
        hipLaunchKernelGGL((recon_update_kernel<T, 4>), dim3(grid), dim3(kBlock), 0, s, p, ctx->partials);
    else
        hipLaunchKernelGGL((recon_update_kernel<T, 3>), dim3(grid), dim3(kBlock), 0, s, p, ctx->partials);
    TVDN_HIP(hipGetLastError());
    return launch_finalize(ctx, grid, 2, sums_out, s);
}

// ---- synthetic input (cytvdn_amd/synth.py, same integer arithmetic) ---------------------------
__device__ __forceinline__ unsigned hash24(unsigned long long seed, unsigned long long lin)
{
    unsigned long long z = seed + (lin + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (unsigned)(z >> 40);
}

__device__ __forceinline__ bool near6(long long dx, long long dy, const long long (&sx)[6],
                                      const long long (&sy)[6], long long rb2)
{
    bool m = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) m |= ((dx - sx[i]) * (dx - sx[i]) + (dy - sy[i]) * (dy - sy[i]) <= rb2);
    return m;
}

__device__ int level_4d(long long x, long long y, long long qx, long long qy, long long NX, long long NY,
                        long long NQX, long long NQY)
{
    const long long cx = NQX / 2, cy = NQY / 2;
    long long r0 = NQX / 16; if (r0 < 1) r0 = 1;
    long long R = NQX / 4;   if (R < 1) R = 1;
    long long rb = NQX / 20; if (rb < 1) rb = 1;
    const long long h = R / 2, s = (7 * R) / 8;
    const long long dx = qx - cx, dy = qy - cy;
    const long long ax_[6] = {R, -R, h, h, -h, -h}, ay_[6] = {0, 0, s, -s, s, -s};
    const long long bx_[6] = {0, 0, s, s, -s, -s}, by_[6] = {R, -R, h, -h, h, -h};
    const bool central = dx * dx + dy * dy <= r0 * r0;
    const bool in_a = near6(dx, dy, ax_, ay_, rb * rb), in_b = near6(dx, dy, bx_, by_, rb * rb);
    const bool grain_a = 4 * x * NY < 2 * NX * NY + (2 * y - NY) * NX;
    int lvl = 0;
    if (in_b) lvl = grain_a ? 1 : 2;
    if (in_a) lvl = grain_a ? 2 : 1;
    if (central) lvl = 3;
    return lvl;
}

__device__ int level_3d(long long x, long long y, long long e, long long NX, long long NY, long long NE)
{
    long long q = (8 * e) / NE; if (q > 7) q = 7;
    long long base = 7 - q;
    const bool phase_b = 4 * (x * x + y * y) < NX * NX + NY * NY;
    if ((2 * e >= NE) && phase_b) base += 2;
    if (base > 7) base = 7;
    return (int)base + 4;
}

__constant__ unsigned c_thresholds[TVDN_SYNTH_NLEVELS][TVDN_SYNTH_NT] = TVDN_SYNTH_TABLE_INIT;

template <typename T>
__global__ void __launch_bounds__(kBlock) synth_kernel(T *out, long long count, long long lin0, int ndim,
                                                        long long n0, long long n1, long long n2, long long n3,
                                                        unsigned long long seed)
{
    const long long step = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < count; i += step) {
        const long long lin = lin0 + i;
        int lvl;
        if (ndim == 4) {
            const long long qy = lin % n3, qx = (lin / n3) % n2, y = (lin / (n3 * n2)) % n1, x = lin / (n3 * n2 * n1);
            lvl = level_4d(x, y, qx, qy, n0, n1, n2, n3);
        } else {
            const long long e = lin % n2, y = (lin / n2) % n1, x = lin / (n2 * n1);
            lvl = level_3d(x, y, e, n0, n1, n2);
        }
        const unsigned u = hash24(seed, (unsigned long long)lin);
        int k = 0;
        for (int t = 0; t < TVDN_SYNTH_NT; ++t) k += (u >= c_thresholds[lvl][t]) ? 1 : 0;
        out[i] = (T)k;
    }
}

}  // namespace tvdn

using namespace tvdn;

extern "C" {

int tvdn_abi_version(void) { return TVDN_ABI_VERSION; }

const char *tvdn_last_error(void) { return g_err; }

int tvdn_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return TVDN_ERR_NO_DEVICE;
    }
    return n;
}

int tvdn_ctx_create(tvdn_ctx **out, int device)
{
    TVDN_REQUIRE(out != nullptr, "out is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        set_error("no HIP device visible: the product path needs an MI355X (gfx950); there is no CPU fallback");
        return TVDN_ERR_NO_DEVICE;
    }
    TVDN_REQUIRE(device >= 0 && device < n, "device %d out of range (0..%d)", device, n - 1);
    TVDN_HIP(hipSetDevice(device));
    tvdn_ctx *c = new tvdn_ctx;
    c->device = device;
    c->partials = nullptr;
    c->timing = false;
    hipError_t e = hipMalloc((void **)&c->partials, sizeof(double) * (size_t)kMaxPartialBlocks * kPartialWidth);
    if (e != hipSuccess) {
        delete c;
        set_error("hipMalloc(partials) failed: %s", hipGetErrorString(e));
        return TVDN_ERR_HIP;
    }
    *out = c;
    return TVDN_OK;
}

int tvdn_ctx_timing_enable(tvdn_ctx *ctx, int on)
{
    TVDN_REQUIRE(ctx != nullptr, "ctx is NULL");
    ctx->timing = on != 0;
    return TVDN_OK;
}

int tvdn_ctx_timing_read(tvdn_ctx *ctx, double *total_ms, int64_t *launches)
{
    TVDN_REQUIRE(ctx && total_ms && launches, "NULL argument");
    double tot = 0.0;
    int64_t n = 0;
    for (auto &ev : ctx->events) {
        TVDN_HIP(hipEventSynchronize(ev.second));
        float ms = 0.f;
        TVDN_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        tot += ms;
        ++n;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    ctx->events.clear();
    *total_ms = tot;
    *launches = n;
    return TVDN_OK;
}

int tvdn_ctx_destroy(tvdn_ctx *ctx)
{
    if (!ctx) return TVDN_OK;
    for (auto &ev : ctx->events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    hipError_t e = hipFree(ctx->partials);
    delete ctx;
    if (e != hipSuccess) {
        set_error("hipFree failed: %s", hipGetErrorString(e));
        return TVDN_ERR_HIP;
    }
    return TVDN_OK;
}

int tvdn_accumulator_update(tvdn_ctx *ctx, int dtype, int ndim, const int64_t *shape, const void *a, void *b,
                            void *d, double tk, int ax, double clip, int bc_mode, double *norm_out, void *stream)
{
    int rc = check_common(ctx, dtype, ndim, shape);
    if (rc) return rc;
    TVDN_REQUIRE(a && b && norm_out, "a, b and norm_out must be non-NULL");
    TVDN_REQUIRE(ax >= 0 && ax < ndim, "ax = %d out of range for ndim = %d", ax, ndim);
    TVDN_REQUIRE(bc_mode >= 0 && bc_mode <= 2, "bc_mode must be 0, 1 or 2, got %d", bc_mode);
    TVDN_REQUIRE(!(bc_mode == TVDN_BC_MIRROR && shape[ax] < 2), "mirror BC needs shape[ax] >= 2");
    const Geom g = make_geom(ndim, shape);
    const int cax = canon_axis(ndim, ax);
    return dtype == TVDN_F32
               ? acc_update_impl<float>(ctx, g, a, b, d, tk, cax, clip, bc_mode, norm_out, (hipStream_t)stream)
               : acc_update_impl<double>(ctx, g, a, b, d, tk, cax, clip, bc_mode, norm_out, (hipStream_t)stream);
}

int tvdn_datacube_update(tvdn_ctx *ctx, int dtype, int ndim, const int64_t *shape, const void *orig, void *recon,
                         const void *const *b, const double *lambda_mu, int bc_mode, double *sums_out,
                         void *stream)
{
    int rc = check_common(ctx, dtype, ndim, shape);
    if (rc) return rc;
    TVDN_REQUIRE(orig && recon && b && lambda_mu && sums_out, "NULL argument");
    for (int q = 0; q < ndim; ++q) TVDN_REQUIRE(b[q] != nullptr, "b[%d] is NULL", q);
    if (bc_mode == TVDN_BC_MIRROR) {
        set_error("bc_mode 1 (mirror) reconstruction update reads out of bounds upstream (utils.pyx:117-120): unsupported");
        return TVDN_ERR_UNSUPPORTED;
    }
    TVDN_REQUIRE(bc_mode == 0 || bc_mode == 2, "bc_mode must be 0 or 2, got %d", bc_mode);
    const Geom g = make_geom(ndim, shape);
    return dtype == TVDN_F32
               ? recon_update_impl<float>(ctx, g, orig, recon, b, lambda_mu, sums_out, (hipStream_t)stream)
               : recon_update_impl<double>(ctx, g, orig, recon, b, lambda_mu, sums_out, (hipStream_t)stream);
}

int tvdn_sum_square_error(tvdn_ctx *ctx, int dtype, int ndim, const int64_t *shape, const void *a, const void *b,
                          double *out, void *stream)
{
    int rc = check_common(ctx, dtype, ndim, shape);
    if (rc) return rc;
    TVDN_REQUIRE(a && b && out, "NULL argument");
    const Geom g = make_geom(ndim, shape);
    const int grid = grid_for(g.total);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == TVDN_F32)
        hipLaunchKernelGGL((sse_kernel<float>), dim3(grid), dim3(kBlock), 0, s, (const float *)a, (const float *)b,
                           g.total, ctx->partials);
    else
        hipLaunchKernelGGL((sse_kernel<double>), dim3(grid), dim3(kBlock), 0, s, (const double *)a,
                           (const double *)b, g.total, ctx->partials);
    TVDN_HIP(hipGetLastError());
    return launch_finalize(ctx, grid, 1, out, s);
}

int tvdn_synth_fill(int dtype, int ndim, const int64_t *shape, uint64_t seed, int64_t row0, int64_t rows,
                    void *out, void *stream)
{
    TVDN_REQUIRE(dtype == TVDN_F32 || dtype == TVDN_F64, "bad dtype %d", dtype);
    TVDN_REQUIRE(ndim == 3 || ndim == 4, "ndim must be 3 or 4");
    TVDN_REQUIRE(shape && out, "NULL argument");
    TVDN_REQUIRE(row0 >= 0 && rows >= 0 && row0 + rows <= shape[0], "row range out of bounds");
    long long plane = 1;
    for (int i = 1; i < ndim; ++i) plane *= shape[i];
    const long long count = plane * rows, lin0 = plane * row0;
    if (count == 0) return TVDN_OK;
    const int grid = grid_for(count);
    hipStream_t s = (hipStream_t)stream;
    const long long n3 = ndim == 4 ? shape[3] : 1;
    if (dtype == TVDN_F32)
        hipLaunchKernelGGL((synth_kernel<float>), dim3(grid), dim3(kBlock), 0, s, (float *)out, count, lin0, ndim,
                           (long long)shape[0], (long long)shape[1], (long long)shape[2], n3,
                           (unsigned long long)seed);
    else
        hipLaunchKernelGGL((synth_kernel<double>), dim3(grid), dim3(kBlock), 0, s, (double *)out, count, lin0, ndim,
                           (long long)shape[0], (long long)shape[1], (long long)shape[2], n3,
                           (unsigned long long)seed);
    TVDN_HIP(hipGetLastError());
    return TVDN_OK;
}

}  // extern "C"
