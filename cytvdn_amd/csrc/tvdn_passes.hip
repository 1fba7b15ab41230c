// One-pass ("kernel-level") HIP kernels: one reference pass each, in place, any shape.
//   tvdn_accumulator_update   accumulator_update_{3D,4D}[_FISTA]   cyTVDN/anisotropic.pyx:17-317
//   tvdn_datacube_update      datacube_update_{3D,4D}              cyTVDN/utils.pyx:54-199
//   tvdn_sum_square_error     sum_square_error_{3D,4D}             cyTVDN/utils.pyx:14-49
// These are what a caller that composes its own loop (cyTVDN/mpi.py:317-398) binds; the whole-iteration
// sweep lives in tvdn_fused.hip.
//
// Built for HBM bandwidth, same decomposition as the fused sweep: canonical block (M, A, B, C), a thread
// owns 16 bytes of C (VEC = 4 floats / 2 doubles) of one (a, b) row and marches along M over a short chunk of
// rows.  Index arithmetic (three divisions) happens once per thread, not per voxel; the M-axis neighbour is a
// register carried along the march (so an axis-0 accumulator update and the axis-0 term of the
// reconstruction update touch memory once per array); in-plane neighbours are extra loads of lines that
// neighbouring threads stream in at the same time.  Arrays read at the own position only use streaming
// (non-temporal) loads, everything written uses streaming stores.  Workgroup ids are remapped per XCD.
// Shapes whose C extent is not a multiple of VEC, or unaligned pointers, take the same code with VEC = 1.
//
// Arithmetic contract: every operation in the array dtype, left to right as the
// reference writes it, no FMA (-ffp-contract=off), f32 denormals kept; sums in f64 by a fixed tree.
#include <cstdlib>

#include "tvdn_common.hpp"

namespace tvdn {

constexpr int kBlock = 256;
constexpr int kReconTA = 2;  // A-rows per thread of the reconstruction update (see recon_update_kernel)
#ifndef TVDN_PASS_M_NT
#define TVDN_PASS_M_NT 1
#endif
// Arrays whose rows a marching thread loads exactly once (the axis-0 operand carried in registers) are streamed too.
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> ldv_m(const T *p)
{
    return TVDN_PASS_M_NT ? ldv_nt<T, VEC>(p) : ldv<T, VEC>(p);
}

// How the (M, A, B, C) block is cut into workgroups: `tiles` workgroups per cross-section, each marching
// `chunk` rows.
struct March {
    long long M, A, B, C;
    long long units, tiles;
    int chunk;
    long long grid;
};

static March make_march(const Geom &g, int vec)
{
    March h;
    h.M = g.n[0]; h.A = g.n[1]; h.B = g.n[2]; h.C = g.n[3];
    h.units = h.A * h.B * (h.C / vec);
    h.tiles = (h.units + kBlock - 1) / kBlock;
    // rows per march: 16 measured best for the accumulator updates on MI355X (interleaved A/B at 256x256x128x128 f32,
    // FISTA form: 8 rows 4.09-4.50 ms, 16 rows 3.56-4.10 ms, 32 rows 3.52-4.10 ms; the other passes do not care)
    long long chunk = 16;
    const char *e = getenv("TVDN_PASS_CHUNK");  // measurement / test knob: rows per march, taken as given
    if (e && atoll(e) > 0)
        chunk = atoll(e);
    else
        while (chunk > 1 && h.tiles * ((h.M + chunk - 1) / chunk) < 256 * 8) chunk /= 2;
    while (h.tiles * ((h.M + chunk - 1) / chunk) > kMaxPartialBlocks && chunk < h.M) chunk *= 2;
    h.chunk = (int)chunk;
    h.grid = h.tiles * ((h.M + chunk - 1) / chunk);
    return h;
}

// ---- accumulator update -----------------------------------------------------------------------------
// AXK: 0 = marching axis M, 1 = an in-plane axis whose neighbours are whole packs (A or B), 2 = the contiguous axis C
template <typename T>
struct AccParams {
    const T *a;
    T *b;
    T *d;
    T tk, clip;
    long long M, A, B, C;
    long long tiles, units;
    int chunk;
    int cax;  // canonical axis 0..3
    int bc;
    double *partials;
};

template <typename T, int VEC, bool FISTA, int AXK>
__global__ void __launch_bounds__(kBlock) acc_update_kernel(AccParams<T> p)
{
    using P = Pack<T, VEC>;
    double acc[1] = {0.0};
    const long long L = xcd_remap(blockIdx.x, gridDim.x);
    const long long chunk_id = L / p.tiles, tile = L % p.tiles;
    const long long u = tile * kBlock + threadIdx.x;
    const long long m0 = chunk_id * p.chunk;
    const long long m1 = (m0 + p.chunk < p.M) ? m0 + p.chunk : p.M;
    if (u < p.units && m0 < m1) {
        const long long LR = p.C / VEC;
        const long long cv = u % LR, bb = (u / LR) % p.B, aa = u / (LR * p.B);
        const long long c0 = cv * VEC;
        const long long SM = p.A * p.B * p.C, SA = p.B * p.C, SB = p.C;
        const long long xs = aa * SA + bb * SB + c0;
        const T tk = p.tk, cl = p.clip;

        // offset of the "previous" element along the axis for in-plane axes (constant along the march):
        // at index 0 Jia-Zhao points at itself, periodic wraps to N-1, mirror takes index 1 (anisotropic.pyx:65-73)
        long long off_prev = 0;
        if (AXK == 1) {
            const long long idx = (p.cax == 1) ? aa : bb, n = (p.cax == 1) ? p.A : p.B, st = (p.cax == 1) ? SA : SB;
            off_prev = idx > 0 ? -st : (p.bc == TVDN_BC_JIA_ZHAO ? 0 : (p.bc == TVDN_BC_PERIODIC ? (n - 1) * st : st));
        } else if (AXK == 2) {
            off_prev = c0 > 0 ? -1 : (p.bc == TVDN_BC_JIA_ZHAO ? 0 : (p.bc == TVDN_BC_PERIODIC ? p.C - 1 : 1));
        }

        P a_prev;  // AXK == 0: row m-1, carried in registers
        if (AXK == 0) {
            long long mp = m0 - 1;
            if (m0 == 0) mp = (p.bc == TVDN_BC_JIA_ZHAO) ? 0 : (p.bc == TVDN_BC_PERIODIC ? p.M - 1 : 1);
            a_prev = ldv<T, VEC>(p.a + mp * SM + xs);
        }
        for (long long m = m0; m < m1; ++m) {
            const long long x = m * SM + xs;
            const P a_cur = (AXK == 0) ? ldv_m<T, VEC>(p.a + x) : ldv<T, VEC>(p.a + x);
            P pv;
            if (AXK == 0) {
                pv = a_prev;
            } else if (AXK == 1) {
                pv = ldv<T, VEC>(p.a + x + off_prev);
            } else {
                pv.v[0] = p.a[x + off_prev];
#pragma unroll
                for (int j = 1; j < VEC; ++j) pv.v[j] = a_cur.v[j - 1];
            }
            const P b_old = ldv_nt<T, VEC>(p.b + x);
            P d_old, b_new, d_new;
            if (FISTA) d_old = ldv_nt<T, VEC>(p.d + x);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const T dn = clipv((a_cur.v[j] - pv.v[j]) + b_old.v[j], cl);
                T bn = dn;
                if (FISTA) {
                    bn = dn + tk * (dn - d_old.v[j]);
                    d_new.v[j] = dn;
                }
                b_new.v[j] = bn;
                acc[0] += fabs((double)bn);
            }
            stv<T, VEC>(p.b + x, b_new);
            if (FISTA) stv<T, VEC>(p.d + x, d_new);
            if (AXK == 0) a_prev = a_cur;
        }
    }
    block_store_partials<1, kBlock>(acc, p.partials);
}

template <typename T, int VEC, bool FISTA>
static void launch_acc_k(const AccParams<T> &p, int axk, long long grid, hipStream_t s)
{
    if (axk == 0)
        hipLaunchKernelGGL((acc_update_kernel<T, VEC, FISTA, 0>), dim3((unsigned)grid), dim3(kBlock), 0, s, p);
    else if (axk == 1)
        hipLaunchKernelGGL((acc_update_kernel<T, VEC, FISTA, 1>), dim3((unsigned)grid), dim3(kBlock), 0, s, p);
    else
        hipLaunchKernelGGL((acc_update_kernel<T, VEC, FISTA, 2>), dim3((unsigned)grid), dim3(kBlock), 0, s, p);
}

template <typename T>
static int acc_update_impl(tvdn_ctx *ctx, const Geom &g, const void *a, void *b, void *d, double tk, int cax,
                           double clip, int bc_mode, double *norm_out, hipStream_t s)
{
    constexpr int VMAX = 16 / sizeof(T);
    const bool al = aligned16(a) && aligned16(b) && aligned16(d);
    const int vec = (al && g.n[3] % VMAX == 0) ? VMAX : 1;
    const March h = make_march(g, vec);
    TVDN_REQUIRE(h.grid >= 1 && h.grid <= kMaxPartialBlocks, "grid %lld out of range", h.grid);
    int rc = ensure_partials(ctx, h.grid);
    if (rc) return rc;
    AccParams<T> p;
    p.a = (const T *)a; p.b = (T *)b; p.d = (T *)d;
    p.tk = (T)tk; p.clip = (T)clip;
    p.M = h.M; p.A = h.A; p.B = h.B; p.C = h.C;
    p.tiles = h.tiles; p.units = h.units; p.chunk = h.chunk;
    p.cax = cax; p.bc = bc_mode;
    p.partials = ctx->partials;
    const int axk = cax == 0 ? 0 : (cax == 3 ? 2 : 1);
    if (vec == VMAX) {
        if (d) launch_acc_k<T, VMAX, true>(p, axk, h.grid, s);
        else launch_acc_k<T, VMAX, false>(p, axk, h.grid, s);
    } else {
        if (d) launch_acc_k<T, 1, true>(p, axk, h.grid, s);
        else launch_acc_k<T, 1, false>(p, axk, h.grid, s);
    }
    TVDN_HIP(hipGetLastError());
    return launch_finalize(ctx, (int)h.grid, 1, norm_out, s);
}

// ---- reconstruction update --------------------------------------------------------------------------
template <typename T>
struct ReconParams {
    const T *orig;
    T *recon;
    const T *b[4];  // slot per REFERENCE axis (3-D: M, B, C)
    T lm[4];
    long long M, A, B, C;
    long long tiles, units;
    int chunk;
    double *partials;
};

// datacube_update_{3D,4D}: periodic-wrap branch (BC 0 and 2), association of the sum from the generated C,
// utils.c:5641: ((t0 + t1) + t2) + t3.
// TA consecutive A-rows per thread (4-D only): the A-neighbour of all but the last of them is a register the thread
// already holds, so of the re-reads that cross workgroups (the next A-row sits 64 KiB - 256 KiB away, beyond what
// the XCD's L2 keeps) only one in TA is left.
template <typename T, int VEC, int NAX, int TA>
__global__ void __launch_bounds__(kBlock) recon_update_kernel(ReconParams<T> p)
{
    using P = Pack<T, VEC>;
    constexpr int iM = 0, iA = 1, iB = NAX - 2, iC = NAX - 1;
    constexpr bool HAS_A = (NAX == 4);
    static_assert(HAS_A || TA == 1, "A-tiling needs the A axis");
    double acc[2] = {0.0, 0.0};
    const long long L = xcd_remap(blockIdx.x, gridDim.x);
    const long long chunk_id = L / p.tiles, tile = L % p.tiles;
    const long long u = tile * kBlock + threadIdx.x;
    const long long m0 = chunk_id * p.chunk;
    const long long m1 = (m0 + p.chunk < p.M) ? m0 + p.chunk : p.M;
    if (u < p.units && m0 < m1) {
        const long long LR = p.C / VEC;
        const long long cv = u % LR, bb = (u / LR) % p.B, a0 = (u / (LR * p.B)) * TA;
        const long long c0 = cv * VEC;
        const long long SM = p.A * p.B * p.C, SA = p.B * p.C, SB = p.C;
        const long long xs = a0 * SA + bb * SB + c0;
        // "next" along each in-plane axis with the periodic wrap of utils.pyx:98-101
        const long long offA = HAS_A ? ((a0 + TA == p.A) ? -(p.A - 1) * SA : SA) : 0;  // from the thread's LAST A-row
        const long long offB = (bb + 1 == p.B) ? -(p.B - 1) * SB : SB;
        const long long offC = (c0 + VEC == p.C) ? -(p.C - VEC) : VEC;  // element that follows the pack
        const T lmM = p.lm[iM], lmB = p.lm[iB], lmC = p.lm[iC];
        const T *bM = p.b[iM], *bB = p.b[iB], *bC = p.b[iC];

        P bM_cur[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) bM_cur[t] = ldv<T, VEC>(bM + m0 * SM + xs + t * SA);
        for (long long m = m0; m < m1; ++m) {
            const long long x0 = m * SM + xs;
            const long long xn0 = ((m + 1 < p.M) ? m + 1 : 0) * SM + xs;
            P bA[TA + 1];
            if (HAS_A) {
#pragma unroll
                for (int t = 0; t < TA; ++t) bA[t] = ldv<T, VEC>(p.b[iA] + x0 + t * SA);
                bA[TA] = ldv<T, VEC>(p.b[iA] + x0 + (TA - 1) * SA + offA);
            }
            // every load of the row before the first use (a wave then has the whole row's requests in flight instead of one
            // axis at a time: the compiler keeps source order across the arithmetic, csrc/tvdn_fused.hip has the same)
            P bM_next[TA], bBo[TA], bBn[TA], bCo[TA], og[TA], old[TA];
            T bCafter[TA];
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const long long x = x0 + t * SA;
                bM_next[t] = ldv_m<T, VEC>(bM + xn0 + t * SA);
                bBo[t] = ldv<T, VEC>(bB + x);
                bBn[t] = ldv<T, VEC>(bB + x + offB);
                bCo[t] = ldv<T, VEC>(bC + x);
                bCafter[t] = bC[x + offC];
                og[t] = ldv_nt<T, VEC>(p.orig + x);
                old[t] = ldv_nt<T, VEC>(p.recon + x);
            }
#pragma unroll
            for (int t = 0; t < TA; ++t) {
                const long long x = x0 + t * SA;
                P s;
#pragma unroll
                for (int j = 0; j < VEC; ++j) s.v[j] = lmM * (bM_cur[t].v[j] - bM_next[t].v[j]);
                if (HAS_A) {
                    const T lmA = p.lm[iA];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) s.v[j] = s.v[j] + lmA * (bA[t].v[j] - bA[t + 1].v[j]);
                }
#pragma unroll
                for (int j = 0; j < VEC; ++j) s.v[j] = s.v[j] + lmB * (bBo[t].v[j] - bBn[t].v[j]);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const T nx = (j + 1 < VEC) ? bCo[t].v[j + 1 < VEC ? j + 1 : 0] : bCafter[t];
                    s.v[j] = s.v[j] + lmC * (bCo[t].v[j] - nx);
                }
                P nw;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    nw.v[j] = og[t].v[j] - s.v[j];
                    const T df = nw.v[j] - old[t].v[j];
                    acc[0] += fabs((double)df);
                    acc[1] += fabs((double)old[t].v[j]);
                }
                stv<T, VEC>(p.recon + x, nw);
                bM_cur[t] = bM_next[t];
            }
        }
    }
    block_store_partials<2, kBlock>(acc, p.partials);
}

template <typename T, int VEC>
static void launch_recon(const ReconParams<T> &p, int nax, int ta, long long grid, hipStream_t s)
{
    const dim3 g((unsigned)grid), blk(kBlock);
    if (nax == 3)
        hipLaunchKernelGGL((recon_update_kernel<T, VEC, 3, 1>), g, blk, 0, s, p);
    else if (ta == 4)
        hipLaunchKernelGGL((recon_update_kernel<T, VEC, 4, 4>), g, blk, 0, s, p);
    else if (ta == 2)
        hipLaunchKernelGGL((recon_update_kernel<T, VEC, 4, 2>), g, blk, 0, s, p);
    else
        hipLaunchKernelGGL((recon_update_kernel<T, VEC, 4, 1>), g, blk, 0, s, p);
}

template <typename T>
static int recon_update_impl(tvdn_ctx *ctx, const Geom &g, const void *orig, void *recon, const void *const *b,
                             const double *lm, double *sums_out, hipStream_t s)
{
    constexpr int VMAX = 16 / sizeof(T);
    bool al = aligned16(orig) && aligned16(recon);
    for (int q = 0; q < g.nax; ++q) al = al && aligned16(b[q]);
    const int vec = (al && g.n[3] % VMAX == 0) ? VMAX : 1;
    // A-rows per thread: 2 when the extent allows it and there is enough work to fill the chip
    int ta = 1;
    if (g.nax == 4 && vec == VMAX) {
        const char *e = getenv("TVDN_RECON_TA");  // measurement / test knob: taken as given where A allows it
        int want = e ? atoi(e) : kReconTA;
        if (want != 4 && want != 2) want = 1;
        while (want > 1 && (g.n[1] % want != 0 || (!e && g.total / VMAX / want < 256LL * 256 * 8))) want /= 2;
        ta = want;
    }
    Geom gt = g;  // the march sees A/ta "rows" of ta A-rows each
    gt.n[1] = g.n[1] / ta;
    March h = make_march(gt, vec);
    TVDN_REQUIRE(h.grid >= 1 && h.grid <= kMaxPartialBlocks, "grid %lld out of range", h.grid);
    int rc = ensure_partials(ctx, h.grid);
    if (rc) return rc;
    ReconParams<T> p;
    p.orig = (const T *)orig;
    p.recon = (T *)recon;
    for (int q = 0; q < 4; ++q) {
        p.b[q] = q < g.nax ? (const T *)b[q] : nullptr;
        p.lm[q] = q < g.nax ? (T)lm[q] : (T)0;
    }
    p.M = g.n[0]; p.A = g.n[1]; p.B = g.n[2]; p.C = g.n[3];
    p.tiles = h.tiles; p.units = h.units; p.chunk = h.chunk;
    p.partials = ctx->partials;
    if (vec == VMAX)
        launch_recon<T, VMAX>(p, g.nax, ta, h.grid, s);
    else
        launch_recon<T, 1>(p, g.nax, 1, h.grid, s);
    TVDN_HIP(hipGetLastError());
    return launch_finalize(ctx, (int)h.grid, 2, sums_out, s);
}

// ---- sum of squared errors --------------------------------------------------------------------------
constexpr int kSsePacks = 8;  // packs per thread: a workgroup streams 32 KiB of each array

template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock) sse_kernel(const T *__restrict__ a, const T *__restrict__ b,
                                                      long long npacks, double *partials)
{
    double acc[1] = {0.0};
    const long long base = (long long)blockIdx.x * (kBlock * kSsePacks) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kSsePacks; ++k) {
        const long long u = base + (long long)k * kBlock;
        if (u < npacks) {
            const Pack<T, VEC> x = ldv_nt<T, VEC>(a + u * VEC), y = ldv_nt<T, VEC>(b + u * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const T t = x.v[j] - y.v[j];
                const T sq = t * t;
                acc[0] += (double)sq;
            }
        }
    }
    block_store_partials<1, kBlock>(acc, partials);
}

template <typename T>
static int sse_impl(tvdn_ctx *ctx, long long total, const void *a, const void *b, double *out, hipStream_t s)
{
    constexpr int VMAX = 16 / sizeof(T);
    const int vec = (aligned16(a) && aligned16(b) && total % VMAX == 0) ? VMAX : 1;
    const long long npacks = total / vec;
    const long long per_block = (long long)kBlock * kSsePacks;
    const long long grid = (npacks + per_block - 1) / per_block;
    TVDN_REQUIRE(grid >= 1 && grid <= kMaxPartialBlocks, "array of %lld elements is too large for one pass", total);
    int rc = ensure_partials(ctx, grid);
    if (rc) return rc;
    if (vec == VMAX)
        hipLaunchKernelGGL((sse_kernel<T, VMAX>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const T *)a, (const T *)b,
                           npacks, ctx->partials);
    else
        hipLaunchKernelGGL((sse_kernel<T, 1>), dim3((unsigned)grid), dim3(kBlock), 0, s, (const T *)a, (const T *)b,
                           npacks, ctx->partials);
    TVDN_HIP(hipGetLastError());
    return launch_finalize(ctx, (int)grid, 1, out, s);
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

}  // namespace tvdn

using namespace tvdn;

extern "C" {

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
    return dtype == TVDN_F32 ? sse_impl<float>(ctx, g.total, a, b, out, (hipStream_t)stream)
                             : sse_impl<double>(ctx, g.total, a, b, out, (hipStream_t)stream);
}

}  // extern "C"
