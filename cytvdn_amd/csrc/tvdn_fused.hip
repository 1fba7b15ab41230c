// Fused anisotropic-TV iteration for gfx950: all accumulator updates and the reconstruction
// update of one iteration (cyTVDN/cyTVDN.py:153-184, :205-230, :372-392, :405-420) in ONE sweep
// that reads each state array once and writes it once (19 array passes for 4-D FISTA instead of
// the reference's 27).
//
// Why it is legal: recon_new(x) needs b_new_ax(x) and b_new_ax(x+e_ax); b_new_ax(y) needs the OLD
// recon at y and y-e_ax and the OLD b_ax, d_ax at y.  All state is double-buffered (the sweep
// reads *_in, writes *_out), so a thread may recompute its +1 neighbour's b_new from old values
// that nobody overwrites.  The recomputed value is bit-identical to the owner's (same inputs,
// same operations, no FMA contraction: this file is compiled with -ffp-contract=off).
//
// Decomposition: canonical block (M, A, B, C), C contiguous (3-D data: A == 1 and absent).
//   * a thread owns VEC consecutive C-elements (16 bytes) of one (a, b) row and MARCHES along M
//     over a chunk of rows, carrying recon(m) and b_new_M(m) in registers, so the M-axis
//     neighbours never touch memory twice;
//   * the A/B/C neighbours are fetched as extra (cache-served) loads of the same lines that the
//     neighbouring threads of the workgroup stream in at the same time;
//   * workgroup ids are remapped so that each XCD (private L2) sweeps a contiguous range of
//     cross-section tiles.
// Slab decomposition along M (one GPU per slab) is expressed by row_lo/row_hi and the edge modes
// of tvdn.h; the caller exchanges recon halo rows between iterations.
#include <cstdlib>

#include "tvdn_common.hpp"

namespace tvdn {

template <typename T>
__device__ __forceinline__ T clipv(T a, T val)
{
    // two ternaries as in the reference's generated C (anisotropic.c:2423-2437): NaN passes through
    const T lo = -val;
    const T t = (lo > a) ? lo : a;
    return (val < t) ? val : t;
}

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
    T v[VEC];
};

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> ldv(const T *p)
{
    return *reinterpret_cast<const Pack<T, VEC> *>(p);
}

template <typename T, int VEC>
__device__ __forceinline__ void stv(T *p, const Pack<T, VEC> &x)
{
    *reinterpret_cast<Pack<T, VEC> *>(p) = x;
}

template <typename T>
struct FusedParams {
    const T *orig;
    const T *r_in;
    T *r_out;
    const T *b_in[4];
    T *b_out[4];
    const T *d_in[4];
    T *d_out[4];
    T tk;
    T clip[4];
    T lm[4];
    long long M, A, B, C;
    long long row_lo, row_hi;
    long long sweep_lo, sweep_hi;  // rows advanced by this launch, inside [row_lo, row_hi)
    int lo_mode, hi_mode, bc;
    int chunk;        // rows per workgroup march
    long long tiles;  // workgroups per cross-section
    long long units;  // A * B * (C / VEC)
    int wga, wgb;     // rows of A and of B covered by one workgroup (0: linear thread->unit map)
    int sync;         // keep the workgroup's waves in step (one barrier per row)
    int xcd;          // 1: remap workgroup ids so each XCD sweeps a contiguous run of tiles
    int fake;         // MEASUREMENT ONLY (wrong results): bit0 zero the A offsets, bit1 the B offsets, bit2 the C offsets
    double *partials;
};

constexpr int kFusedBlock = 256;

// One accumulator update (anisotropic.pyx:50-54 plain, :127-132 FISTA); returns b_new, sets d_new.
template <typename T, bool FISTA>
__device__ __forceinline__ T acc_new(T r_x, T r_prev, T b, T d, T tk, T clip, T &d_new)
{
    const T v = (r_x - r_prev) + b;
    const T dn = clipv(v, clip);
    d_new = dn;
    if (FISTA) return dn + tk * (dn - d);
    return dn;
}

// Axis whose neighbours are whole packs (A and B): update own b/d at x, recompute b_new at the
// +1 neighbour, add lm * (b_new(x) - b_new(x+e)) to `sum` (left-to-right as utils.c:5641).
template <typename T, int VEC, bool FISTA>
__device__ __forceinline__ void axis_pack(const Pack<T, VEC> &r_cur, const T *__restrict__ r_in,
                                          const T *__restrict__ b_in, const T *__restrict__ d_in,
                                          T *__restrict__ b_out, T *__restrict__ d_out, long long x,
                                          long long off_prev, long long off_next, bool self_next, T tk, T cl,
                                          T lm, Pack<T, VEC> &sum, double &bnorm)
{
    using P = Pack<T, VEC>;
    const P rp = ldv<T, VEC>(r_in + x + off_prev);
    const P rn = ldv<T, VEC>(r_in + x + off_next);
    const P b_own = ldv<T, VEC>(b_in + x);
    const P b_nx = ldv<T, VEC>(b_in + x + off_next);
    P d_own, d_nx;
    if (FISTA) {
        d_own = ldv<T, VEC>(d_in + x);
        d_nx = ldv<T, VEC>(d_in + x + off_next);
    }
    P bn_own, dn_own;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        bn_own.v[j] = acc_new<T, FISTA>(r_cur.v[j], rp.v[j], b_own.v[j], FISTA ? d_own.v[j] : (T)0, tk, cl, dn_own.v[j]);
        T unused;
        const T bn_next = acc_new<T, FISTA>(rn.v[j], self_next ? rn.v[j] : r_cur.v[j], b_nx.v[j],
                                            FISTA ? d_nx.v[j] : (T)0, tk, cl, unused);
        sum.v[j] = sum.v[j] + lm * (bn_own.v[j] - bn_next);
        bnorm += fabs((double)bn_own.v[j]);
    }
    stv<T, VEC>(b_out + x, bn_own);
    if (FISTA) stv<T, VEC>(d_out + x, dn_own);
}

// Contiguous axis C: neighbours inside the pack come from registers; only the element before the
// pack and the one after it are fetched.
template <typename T, int VEC, bool FISTA>
__device__ __forceinline__ void axis_contig(const Pack<T, VEC> &r_cur, const T *__restrict__ r_in,
                                            const T *__restrict__ b_in, const T *__restrict__ d_in,
                                            T *__restrict__ b_out, T *__restrict__ d_out, long long x,
                                            long long off_prev, long long off_next, bool self_next, T tk, T cl,
                                            T lm, Pack<T, VEC> &sum, double &bnorm)
{
    using P = Pack<T, VEC>;
    const T r_before = r_in[x + off_prev];
    const T r_after = r_in[x + off_next];
    const T b_after = b_in[x + off_next];
    const T d_after = FISTA ? d_in[x + off_next] : (T)0;
    const P b_own = ldv<T, VEC>(b_in + x);
    P d_own;
    if (FISTA) d_own = ldv<T, VEC>(d_in + x);
    P bn_own, dn_own;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const T rp = (j == 0) ? r_before : r_cur.v[j > 0 ? j - 1 : 0];
        bn_own.v[j] = acc_new<T, FISTA>(r_cur.v[j], rp, b_own.v[j], FISTA ? d_own.v[j] : (T)0, tk, cl, dn_own.v[j]);
        bnorm += fabs((double)bn_own.v[j]);
    }
    T unused;
    const T bn_after = acc_new<T, FISTA>(r_after, self_next ? r_after : r_cur.v[VEC - 1], b_after, d_after, tk, cl, unused);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const T bn_next = (j + 1 < VEC) ? bn_own.v[j + 1 < VEC ? j + 1 : 0] : bn_after;
        sum.v[j] = sum.v[j] + lm * (bn_own.v[j] - bn_next);
    }
    stv<T, VEC>(b_out + x, bn_own);
    if (FISTA) stv<T, VEC>(d_out + x, dn_own);
}

template <typename T, int VEC, int NAX, bool FISTA>
__global__ void __launch_bounds__(kFusedBlock) fused_iter_kernel(FusedParams<T> p)
{
    using P = Pack<T, VEC>;
    constexpr int iM = 0, iA = 1, iB = NAX - 2, iC = NAX - 1;  // accumulator slot per canonical axis
    constexpr bool HAS_A = (NAX == 4);

    // XCD-aware remap (bijective for any grid size): blocks b and b+8 share an XCD and its L2, so
    // each XCD gets a contiguous run of logical ids.
    const long long G = gridDim.x, bid = blockIdx.x;
    const long long q8 = G / 8, r8 = G % 8, xcd = bid % 8;
    const long long L = p.xcd ? (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8 : bid;
    const long long chunk_id = L / p.tiles, tile = L % p.tiles;

    double acc[3] = {0.0, 0.0, 0.0};  // b_norm, sum|delta|, sum|old|
    const long long LRk = p.C / VEC;
    long long u;
    if (p.wga > 0) {
        // 2-D workgroup tile: wga rows of A x wgb rows of B, each row LRk lanes wide
        const long long tilesB = p.B / p.wgb;
        const long long ta = tile / tilesB, tb = tile % tilesB;
        const long long row = threadIdx.x / LRk, cvk = threadIdx.x % LRk;
        const long long ra = row / p.wgb, rb = row % p.wgb;
        u = ((ta * p.wga + ra) * p.B + (tb * p.wgb + rb)) * LRk + cvk;
    } else {
        u = tile * kFusedBlock + threadIdx.x;
    }

    const long long m0 = p.sweep_lo + chunk_id * p.chunk;
    const long long m1 = (m0 + p.chunk < p.sweep_hi) ? m0 + p.chunk : p.sweep_hi;

    if (u < p.units && m0 < m1) {
        const long long LR = p.C / VEC;
        const long long cv = u % LR, bb = (u / LR) % p.B, aa = u / (LR * p.B);
        const long long c0 = cv * VEC;
        const long long SM = p.A * p.B * p.C, SA = p.B * p.C, SB = p.C;
        const long long xs = aa * SA + bb * SB + c0;  // offset inside a row-plane
        const bool bc2 = (p.bc == TVDN_BC_JIA_ZHAO);

        // Neighbour offsets inside the plane (constant along the march).
        // own "prev": at index 0 periodic BC wraps to N-1, Jia-Zhao points at itself (anisotropic.pyx:65-73)
        const long long offA_prev = HAS_A ? (aa > 0 ? -SA : (bc2 ? 0 : (p.A - 1) * SA)) : 0;
        const long long offB_prev = bb > 0 ? -SB : (bc2 ? 0 : (p.B - 1) * SB);
        const long long offC_prev = c0 > 0 ? -1 : (bc2 ? 0 : p.C - 1);
        // "next": periodic wrap (utils.pyx:98-101)
        const bool wrapA = HAS_A && (aa + 1 == p.A), wrapB = (bb + 1 == p.B), wrapC = (c0 + VEC == p.C);
        const long long offA_next = HAS_A ? (wrapA ? -(p.A - 1) * SA : SA) : 0;
        const long long offB_next = wrapB ? -(p.B - 1) * SB : SB;
        const long long offC_next = wrapC ? -(p.C - VEC) : VEC;
        // (timing experiments only; never set by the product path)
        const long long offA_prev_ = (p.fake & 1) ? 0 : offA_prev, offA_next_ = (p.fake & 1) ? 0 : offA_next;
        const long long offB_prev_ = (p.fake & 2) ? 0 : offB_prev, offB_next_ = (p.fake & 2) ? 0 : offB_next;
        const long long offC_prev_ = (p.fake & 4) ? 0 : offC_prev, offC_next_ = (p.fake & 4) ? 0 : offC_next;
        // a wrapped neighbour sits at index 0, where under Jia-Zhao its own "prev" is itself
        const bool selfA = wrapA && bc2, selfB = wrapB && bc2, selfC = wrapC && bc2;

        const T tk = p.tk;
        const T clM = p.clip[iM], clB = p.clip[iB], clC = p.clip[iC];
        const T lmM = p.lm[iM], lmB = p.lm[iB], lmC = p.lm[iC];

        // ---- prologue: M-axis accumulator of row m0 -----------------------------------------------
        P r_cur = ldv<T, VEC>(p.r_in + m0 * SM + xs);
        P bM_cur;
        {
            long long mp;  // the row that precedes m0
            if (m0 > p.row_lo || p.lo_mode == TVDN_EDGE_HALO)
                mp = m0 - 1;
            else
                mp = bc2 ? m0 : p.row_hi - 1;  // TVDN_EDGE_BC: Jia-Zhao -> itself, periodic -> last row
            const P r_prev = ldv<T, VEC>(p.r_in + mp * SM + xs);
            const P b0 = ldv<T, VEC>(p.b_in[iM] + m0 * SM + xs);
            P d0, dn;
            if (FISTA) d0 = ldv<T, VEC>(p.d_in[iM] + m0 * SM + xs);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                bM_cur.v[j] = acc_new<T, FISTA>(r_cur.v[j], r_prev.v[j], b0.v[j], FISTA ? d0.v[j] : (T)0, tk, clM, dn.v[j]);
                acc[0] += fabs((double)bM_cur.v[j]);
            }
            stv<T, VEC>(p.b_out[iM] + m0 * SM + xs, bM_cur);
            if (FISTA) stv<T, VEC>(p.d_out[iM] + m0 * SM + xs, dn);
        }

        // ---- march -------------------------------------------------------------------------------
        for (long long m = m0; m < m1; ++m) {
            if (p.sync) __syncthreads();
            const long long x = m * SM + xs;
            const bool last = (m + 1 == m1);
            const bool at_end = (m + 1 == p.row_hi);

            // (1) M-axis accumulator of the next row (look-ahead by one row)
            P r_next = r_cur, bM_next;
#pragma unroll
            for (int j = 0; j < VEC; ++j) bM_next.v[j] = (T)0;  // TVDN_EDGE_ZERO
            if (!(at_end && p.hi_mode == TVDN_EDGE_ZERO)) {
                const bool wrap = at_end && p.hi_mode == TVDN_EDGE_BC;
                const long long xn = (wrap ? p.row_lo : m + 1) * SM + xs;
                r_next = ldv<T, VEC>(p.r_in + xn);
                const P bn = ldv<T, VEC>(p.b_in[iM] + xn);
                P dn_in, dn;
                if (FISTA) dn_in = ldv<T, VEC>(p.d_in[iM] + xn);
                const bool self = wrap && bc2;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    bM_next.v[j] = acc_new<T, FISTA>(r_next.v[j], self ? r_next.v[j] : r_cur.v[j], bn.v[j],
                                                     FISTA ? dn_in.v[j] : (T)0, tk, clM, dn.v[j]);
                // rows inside the chunk are owned here, and so is a halo row sitting at row_hi
                if (!last || (at_end && p.hi_mode == TVDN_EDGE_HALO)) {
                    stv<T, VEC>(p.b_out[iM] + xn, bM_next);
                    if (FISTA) stv<T, VEC>(p.d_out[iM] + xn, dn);
                }
                if (!last) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[0] += fabs((double)bM_next.v[j]);
                }
            }

            // (2) divergence, axis by axis in the reference's order
            P sum;
#pragma unroll
            for (int j = 0; j < VEC; ++j) sum.v[j] = lmM * (bM_cur.v[j] - bM_next.v[j]);
            if (HAS_A)
                axis_pack<T, VEC, FISTA>(r_cur, p.r_in, p.b_in[iA], p.d_in[iA], p.b_out[iA], p.d_out[iA], x,
                                         offA_prev_, offA_next_, selfA, tk, p.clip[iA], p.lm[iA], sum, acc[0]);
            axis_pack<T, VEC, FISTA>(r_cur, p.r_in, p.b_in[iB], p.d_in[iB], p.b_out[iB], p.d_out[iB], x, offB_prev_,
                                     offB_next_, selfB, tk, clB, lmB, sum, acc[0]);
            axis_contig<T, VEC, FISTA>(r_cur, p.r_in, p.b_in[iC], p.d_in[iC], p.b_out[iC], p.d_out[iC], x, offC_prev_,
                                       offC_next_, selfC, tk, clC, lmC, sum, acc[0]);

            // (3) reconstruction update at row m (utils.pyx:90-104)
            const P og = ldv<T, VEC>(p.orig + x);
            P r_new;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                r_new.v[j] = og.v[j] - sum.v[j];
                const T df = r_new.v[j] - r_cur.v[j];
                acc[1] += fabs((double)df);
                acc[2] += fabs((double)r_cur.v[j]);
            }
            stv<T, VEC>(p.r_out + x, r_new);

            r_cur = r_next;
            bM_cur = bM_next;
        }
    }
    block_store_partials<3, kFusedBlock>(acc, p.partials);
}

template <typename T, int VEC, int NAX, bool FISTA>
static int launch_fused_t(tvdn_ctx *ctx, const FusedParams<T> &p, int grid, hipStream_t s)
{
    hipLaunchKernelGGL((fused_iter_kernel<T, VEC, NAX, FISTA>), dim3(grid), dim3(kFusedBlock), 0, s, p);
    TVDN_HIP(hipGetLastError());
    return TVDN_OK;
}

static bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

template <typename T>
static int iterate_fused_impl(tvdn_ctx *ctx, const tvdn_iter_args *a, double *sums_out, hipStream_t s)
{
    constexpr int VMAX = 16 / sizeof(T);
    const Geom g = make_geom(a->ndim, a->shape);
    const int nax = a->ndim;
    FusedParams<T> p;
    std::memset(&p, 0, sizeof p);
    p.orig = (const T *)a->orig;
    p.r_in = (const T *)a->recon_in;
    p.r_out = (T *)a->recon_out;
    bool al = aligned16(p.orig) && aligned16(p.r_in) && aligned16(p.r_out);
    for (int q = 0; q < nax; ++q) {
        p.b_in[q] = (const T *)a->b_in[q];
        p.b_out[q] = (T *)a->b_out[q];
        p.d_in[q] = a->fista ? (const T *)a->d_in[q] : nullptr;
        p.d_out[q] = a->fista ? (T *)a->d_out[q] : nullptr;
        p.clip[q] = (T)a->clip[q];
        p.lm[q] = (T)a->lambda_mu[q];
        al = al && aligned16(p.b_in[q]) && aligned16(p.b_out[q]) && aligned16(p.d_in[q]) && aligned16(p.d_out[q]);
    }
    p.tk = (T)a->tk;
    p.M = g.n[0]; p.A = g.n[1]; p.B = g.n[2]; p.C = g.n[3];
    p.row_lo = a->row_lo; p.row_hi = a->row_hi;
    const bool whole = (a->sweep_lo == 0 && a->sweep_hi == 0);
    p.sweep_lo = whole ? a->row_lo : a->sweep_lo;
    p.sweep_hi = whole ? a->row_hi : a->sweep_hi;
    p.lo_mode = a->lo_mode; p.hi_mode = a->hi_mode; p.bc = a->bc_mode;
    p.partials = ctx->partials;

    const int vec = (al && (p.C % VMAX == 0)) ? VMAX : 1;
    p.units = p.A * p.B * (p.C / vec);
    p.tiles = (p.units + kFusedBlock - 1) / kFusedBlock;
    // tuning knobs (measurement only): TVDN_WGA = rows of A per workgroup, TVDN_SYNC, TVDN_CHUNK
    const char *e_wga = getenv("TVDN_WGA"), *e_sync = getenv("TVDN_SYNC"), *e_chunk = getenv("TVDN_CHUNK");
    p.wga = 0; p.wgb = 0; p.sync = 0;
    p.xcd = getenv("TVDN_XCD") ? atoi(getenv("TVDN_XCD")) : 1;
    p.fake = getenv("TVDN_FAKE") ? atoi(getenv("TVDN_FAKE")) : 0;
    {
        const long long lr = p.C / vec;
        const long long rows_wg = (lr <= kFusedBlock && kFusedBlock % lr == 0) ? kFusedBlock / lr : 0;
        int want = e_wga ? atoi(e_wga) : 0;
        if (rows_wg > 0 && want > 0 && rows_wg % want == 0 && p.A % want == 0 && p.B % (rows_wg / want) == 0) {
            p.wga = want;
            p.wgb = (int)(rows_wg / want);
            // a uniform barrier needs every thread of the workgroup in the loop: true for full tiles
            p.sync = e_sync ? atoi(e_sync) : 0;
        }
    }
    const long long rows = p.sweep_hi - p.sweep_lo;
    // rows per march: long enough to amortise the look-ahead row (3 extra pack loads per chunk); short
    // marches measured best on MI355X (2..8 rows: 14.7-14.9 ms, 32 rows: 15.3 ms on 256x256x128x128 f32):
    // many short-lived workgroups keep the set of open DRAM pages compact
    long long chunk = e_chunk ? atoll(e_chunk) : 8;
    if (chunk < 1) chunk = 1;
    while (chunk > 4 && p.tiles * ((rows + chunk - 1) / chunk) < 256 * 8) chunk /= 2;
    while (p.tiles * ((rows + chunk - 1) / chunk) > kMaxPartialBlocks && chunk < rows) chunk *= 2;
    p.chunk = (int)chunk;
    const long long nchunks = (rows + chunk - 1) / chunk;
    const long long grid = p.tiles * nchunks;
    TVDN_REQUIRE(grid >= 1 && grid <= kMaxPartialBlocks, "fused grid %lld out of range (max %d)", grid, kMaxPartialBlocks);

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->timing) {
        TVDN_HIP(hipEventCreate(&ev0));
        TVDN_HIP(hipEventCreate(&ev1));
        TVDN_HIP(hipEventRecord(ev0, s));
    }
    int rc;
#define TVDN_LAUNCH(V, N, F) rc = launch_fused_t<T, V, N, F>(ctx, p, (int)grid, s)
    if (vec == VMAX) {
        if (nax == 4) { if (a->fista) TVDN_LAUNCH(VMAX, 4, true); else TVDN_LAUNCH(VMAX, 4, false); }
        else          { if (a->fista) TVDN_LAUNCH(VMAX, 3, true); else TVDN_LAUNCH(VMAX, 3, false); }
    } else {
        if (nax == 4) { if (a->fista) TVDN_LAUNCH(1, 4, true); else TVDN_LAUNCH(1, 4, false); }
        else          { if (a->fista) TVDN_LAUNCH(1, 3, true); else TVDN_LAUNCH(1, 3, false); }
    }
#undef TVDN_LAUNCH
    if (ctx->timing) {
        TVDN_HIP(hipEventRecord(ev1, s));
        ctx->events.emplace_back(ev0, ev1);
    }
    if (rc) return rc;
    return launch_finalize(ctx, (int)grid, 3, sums_out, s, a->accumulate != 0);
}

}  // namespace tvdn

using namespace tvdn;

extern "C" int tvdn_iterate_fused(tvdn_ctx *ctx, const tvdn_iter_args *a, double *sums_out, void *stream)
{
    TVDN_REQUIRE(ctx && a && sums_out, "NULL argument");
    TVDN_REQUIRE(a->dtype == TVDN_F32 || a->dtype == TVDN_F64, "bad dtype %d", a->dtype);
    TVDN_REQUIRE(a->ndim == 3 || a->ndim == 4, "ndim must be 3 or 4, got %d", a->ndim);
    for (int i = 0; i < a->ndim; ++i) TVDN_REQUIRE(a->shape[i] >= 1, "shape[%d] must be >= 1", i);
    if (a->bc_mode == TVDN_BC_MIRROR) {
        set_error("bc_mode 1 (mirror) reconstruction update reads out of bounds upstream (utils.pyx:117-120): unsupported");
        return TVDN_ERR_UNSUPPORTED;
    }
    TVDN_REQUIRE(a->bc_mode == 0 || a->bc_mode == 2, "bc_mode must be 0 or 2, got %d", a->bc_mode);
    TVDN_REQUIRE(0 <= a->row_lo && a->row_lo < a->row_hi && a->row_hi <= a->shape[0],
                 "own rows [%lld,%lld) not inside 0..%lld", (long long)a->row_lo, (long long)a->row_hi,
                 (long long)a->shape[0]);
    TVDN_REQUIRE((a->sweep_lo == 0 && a->sweep_hi == 0) ||
                     (a->row_lo <= a->sweep_lo && a->sweep_lo < a->sweep_hi && a->sweep_hi <= a->row_hi),
                 "sweep rows [%lld,%lld) not inside the own rows", (long long)a->sweep_lo, (long long)a->sweep_hi);
    TVDN_REQUIRE(a->lo_mode == TVDN_EDGE_BC || a->lo_mode == TVDN_EDGE_HALO, "bad lo_mode %d", a->lo_mode);
    TVDN_REQUIRE(a->hi_mode >= TVDN_EDGE_BC && a->hi_mode <= TVDN_EDGE_ZERO, "bad hi_mode %d", a->hi_mode);
    TVDN_REQUIRE(!(a->lo_mode == TVDN_EDGE_HALO && a->row_lo < 1), "lo_mode HALO needs a row below row_lo");
    TVDN_REQUIRE(!(a->hi_mode == TVDN_EDGE_HALO && a->row_hi >= a->shape[0]), "hi_mode HALO needs a row at row_hi");
    TVDN_REQUIRE(!(a->hi_mode == TVDN_EDGE_ZERO && a->bc_mode != TVDN_BC_JIA_ZHAO), "hi_mode ZERO is a Jia-Zhao property");
    TVDN_REQUIRE(!(a->lo_mode == TVDN_EDGE_BC && a->bc_mode == TVDN_BC_PERIODIC && a->hi_mode != TVDN_EDGE_BC),
                 "periodic BC with lo_mode BC needs the whole ring in this block (hi_mode BC)");
    TVDN_REQUIRE(a->orig && a->recon_in && a->recon_out, "NULL state pointer");
    TVDN_REQUIRE(a->recon_in != a->recon_out, "the fused sweep is not in-place: recon_in == recon_out");
    for (int q = 0; q < a->ndim; ++q) {
        TVDN_REQUIRE(a->b_in[q] && a->b_out[q] && a->b_in[q] != a->b_out[q], "b_in[%d]/b_out[%d] NULL or aliased", q, q);
        if (a->fista) TVDN_REQUIRE(a->d_in[q] && a->d_out[q] && a->d_in[q] != a->d_out[q], "d_in[%d]/d_out[%d] NULL or aliased", q, q);
    }
    return a->dtype == TVDN_F32 ? iterate_fused_impl<float>(ctx, a, sums_out, (hipStream_t)stream)
                                : iterate_fused_impl<double>(ctx, a, sums_out, (hipStream_t)stream);
}
