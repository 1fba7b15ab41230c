// libtvdn_hip.so: C ABI plumbing (errors, contexts, the second reduction stage) and the synthetic
// input generator.  The one-pass kernels live in tvdn_passes.hip, the fused iteration in tvdn_fused.hip.
//
// Reductions are kept in f64 by a fixed tree (lanes -> waves -> one partial per workgroup -> the fold below),
// so they are deterministic and, unlike the reference's OpenMP sums, independent of any thread count.
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
// Up to kFoldDirect partial rows: one workgroup folds them in index order.  More: a first stage of
// ceil(n / kFoldSegment) workgroups folds one contiguous segment each, then the single workgroup folds those.
// The tree depends on the number of partial rows only.
__global__ void __launch_bounds__(256) fold_kernel(const double *partials, int nblocks, int nv, double *out)
{
    __shared__ double red[kPartialWidth][4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lo = blockIdx.x * kFoldSegment;
    const int hi = (lo + kFoldSegment < nblocks) ? lo + kFoldSegment : nblocks;
    for (int q = 0; q < nv; ++q) {
        double s = 0.0;
        for (int i = lo + (int)threadIdx.x; i < hi; i += 256) s += partials[(size_t)i * kPartialWidth + q];
        s = wave_sum(s);
        if (lane == 0) red[q][w] = s;
    }
    __syncthreads();
    if ((int)threadIdx.x < nv)
        out[(size_t)blockIdx.x * kPartialWidth + threadIdx.x] =
            ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
}

__global__ void __launch_bounds__(1024) finalize_kernel(const double *partials, int nblocks, int nv,
                                                         double *out, int accumulate, double *mirror)
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
        if (accumulate) s = out[threadIdx.x] + s;
        out[threadIdx.x] = s;
        if (mirror) mirror[threadIdx.x] = s;  // (tvdn_ctx::mirror: host memory the device can write)
    }
}

struct DeferTable {
    int rows[kDeferSlots];
    double *out[kDeferSlots];
};

// finalize_kernel for a batch: workgroup b folds the rows of ring slot b into table.out[b] (same tree, never accumulating)
__global__ void __launch_bounds__(1024) finalize_many_kernel(const double *ring, DeferTable table, int nv)
{
    __shared__ double red[kPartialWidth][16];
    const double *partials = ring + (size_t)blockIdx.x * kFoldDirect * kPartialWidth;
    const int nblocks = table.rows[blockIdx.x];
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
        table.out[blockIdx.x][threadIdx.x] = s;
    }
}

void sums_defer_begin(tvdn_ctx *ctx)
{
    ctx->n_pend = 0;
    ctx->deferring = getenv("TVDN_DEFER_SUMS") ? atoi(getenv("TVDN_DEFER_SUMS")) != 0 : true;
}

double *sums_defer_slot(tvdn_ctx *ctx, long long nblocks, bool accumulate)
{
    if (!ctx->deferring || accumulate || nblocks > kFoldDirect) return nullptr;
    if (!ctx->ring) {
        if (hipMalloc((void **)&ctx->ring, sizeof(double) * (size_t)kDeferSlots * kFoldDirect * kPartialWidth) != hipSuccess) {
            (void)hipGetLastError();
            ctx->ring = nullptr;
            ctx->deferring = false;  // no room for the ring: finalize at once, as outside the loops
            return nullptr;
        }
    }
    return ctx->ring + (size_t)ctx->n_pend * kFoldDirect * kPartialWidth;
}

int sums_defer_flush(tvdn_ctx *ctx, hipStream_t s)
{
    if (ctx->n_pend == 0) return TVDN_OK;
    DeferTable t;
    for (int i = 0; i < kDeferSlots; ++i) {
        t.rows[i] = i < ctx->n_pend ? ctx->pend_rows[i] : 0;
        t.out[i] = i < ctx->n_pend ? ctx->pend_out[i] : nullptr;
    }
    hipLaunchKernelGGL(finalize_many_kernel, dim3(ctx->n_pend), dim3(1024), 0, s, (const double *)ctx->ring, t, 3);
    ctx->n_pend = 0;
    TVDN_HIP(hipGetLastError());
    return TVDN_OK;
}

int sums_defer_push(tvdn_ctx *ctx, int nblocks, double *out, hipStream_t s)
{
    ctx->pend_rows[ctx->n_pend] = nblocks;
    ctx->pend_out[ctx->n_pend] = out;
    if (++ctx->n_pend == kDeferSlots) return sums_defer_flush(ctx, s);
    return TVDN_OK;
}

int sums_defer_end(tvdn_ctx *ctx, hipStream_t s)
{
    const int rc = sums_defer_flush(ctx, s);
    ctx->deferring = false;
    ctx->n_pend = 0;
    return rc;
}

int ensure_partials(tvdn_ctx *ctx, long long nblocks)
{
    TVDN_REQUIRE(nblocks >= 1 && nblocks <= kMaxPartialBlocks, "launch of %lld workgroups exceeds the reduction scratch",
                 nblocks);
    if (nblocks <= ctx->partial_cap) return TVDN_OK;
    long long cap = ctx->partial_cap > 0 ? ctx->partial_cap : kInitPartialBlocks;
    while (cap < nblocks) cap *= 2;
    double *fresh = nullptr;
    TVDN_HIP(hipMalloc((void **)&fresh, sizeof(double) * (size_t)cap * kPartialWidth));
    TVDN_HIP(hipDeviceSynchronize());  // earlier launches may still be writing the old scratch
    if (ctx->partials) TVDN_HIP(hipFree(ctx->partials));
    ctx->partials = fresh;
    ctx->partial_cap = cap;
    return TVDN_OK;
}

int launch_finalize(tvdn_ctx *ctx, int nblocks, int nv, double *out, hipStream_t s, bool accumulate)
{
    const double *src = ctx->partials;
    if (nblocks > kFoldDirect) {
        const int g1 = (nblocks + kFoldSegment - 1) / kFoldSegment;
        hipLaunchKernelGGL(fold_kernel, dim3(g1), dim3(256), 0, s, ctx->partials, nblocks, nv, ctx->partials2);
        TVDN_HIP(hipGetLastError());
        src = ctx->partials2;
        nblocks = g1;
    }
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(1024), 0, s, src, nblocks, nv, out, accumulate ? 1 : 0, ctx->mirror);
    TVDN_HIP(hipGetLastError());
    return TVDN_OK;
}

constexpr int kBlock = 256;

static int grid_for(long long total)
{
    long long g = (total + kBlock - 1) / kBlock;
    const long long cap = 256 * 16;  // 256 CUs x 16 blocks, grid-stride beyond that
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
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
    DeviceRestore restore;
    TVDN_HIP(hipSetDevice(device));
    tvdn_ctx *c = new tvdn_ctx;
    c->device = device;
    c->partials = nullptr;
    c->partials2 = nullptr;
    c->partial_cap = kInitPartialBlocks;
    c->ring = nullptr;
    c->deferring = false;
    c->n_pend = 0;
    c->timing = false;
    c->mirror = nullptr;
    hipError_t e = hipMalloc((void **)&c->partials, sizeof(double) * (size_t)kInitPartialBlocks * kPartialWidth);
    if (e == hipSuccess)
        e = hipMalloc((void **)&c->partials2, sizeof(double) * (size_t)(kMaxPartialBlocks / kFoldSegment) * kPartialWidth);
    if (e != hipSuccess) {
        if (c->partials) (void)hipFree(c->partials);
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

int tvdn_ctx_timing_read_each(tvdn_ctx *ctx, double *ms_out, int64_t cap, int64_t *launches)
{
    TVDN_REQUIRE(ctx && launches && (ms_out || cap == 0) && cap >= 0, "bad argument");
    int64_t n = 0;
    for (auto &ev : ctx->events) {
        TVDN_HIP(hipEventSynchronize(ev.second));
        float ms = 0.f;
        TVDN_HIP(hipEventElapsedTime(&ms, ev.first, ev.second));
        if (n < cap) ms_out[n] = ms;
        ++n;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    ctx->events.clear();
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
    (void)hipFree(ctx->partials2);
    if (ctx->ring) (void)hipFree(ctx->ring);
    delete ctx;
    if (e != hipSuccess) {
        set_error("hipFree failed: %s", hipGetErrorString(e));
        return TVDN_ERR_HIP;
    }
    return TVDN_OK;
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
