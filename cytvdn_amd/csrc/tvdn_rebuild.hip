// recon rebuilt from the accumulator state: the reconstruction update (cyTVDN/utils.pyx:90-104) applied to state that is
// already there, for the streamed engine's level 0 (tvdn_stream_chain.hip).
//
// Why.  A streamed run used to carry recon across PCIe both ways every pass, although it is redundant: the recon an iteration
// leaves behind IS  orig - sum_ax lm_ax (b_ax - b_ax[next along ax])  of the accumulators it leaves behind (utils.pyx:98-104),
// and with the compact state b_k = d_k + tk_prev (d_k - d_k-1) -- the very expression (anisotropic.pyx:128) that formed it,
// which the fused sweep itself evaluates to rebuild b (tvdn_fused.hip acc_new).  Same operands, same operations in the same
// order, no contraction: the rebuilt recon is the stored one bit for bit (the sweep's b_next of a neighbour is that neighbour's
// own b: same inputs, same operations).  So a pass uploads the data term and the d pairs (9 arrays instead of 10 for 4-D
// FISTA), rebuilds recon on the ring rows, and only the run's last pass brings recon down (8 arrays instead of 9).
//
// What.  Rows [row0, row1) of ring-addressed arrays (row v at slot v % ring, the data term at v % ring_orig):
//   b_q(x)   = D_FORM ? d_q + tk_prev (d_q - dprev_q) : b_q                        own position and the next one along q
//   recon(x) = orig(x) - (((lm0 (b0 - b0+) + lm1 (b1 - b1+)) + lm2 (b2 - b2+)) + lm3 (b3 - b3+))     (utils.c:5641's association)
// "next" wraps periodically inside a plane (utils.pyx:98-101); along axis 0 it is the next ring row, and past the block's last
// row `top` (a Jia-Zhao cube's top face) the identically-zero accumulator of row 0 (tvdn.h TVDN_EDGE_ZERO: finite first rows;
// a run whose first row is not finite keeps shipping recon).  One launch per run of consecutive rows; a thread owns 16 bytes
// of C of one row.  HBM-bound: 1 + 2 nd reads and one write per voxel, the neighbours from L2.
#include "tvdn_common.hpp"

namespace tvdn {

template <typename T>
struct RebuildParams {
    const T *orig;
    T *recon;
    const T *in1[4];  // b (b-form) or d_k-1 (d-form)
    const T *in2[4];  // d_k (d-form)
    const T *next1, *next2;  // axis 0, the row AFTER row1 - 1 as planes of their own (not in the rings yet), or NULL: the ring row
    T lm[4];
    T tk_prev;
    long long A, B, C;
    long long row0, row1, top;  // rows rebuilt; first row that is beyond the block (its axis-0 accumulator reads as zero)
    unsigned ring, ring_orig;
    long long units;
};

template <typename T, int VEC, int NAX, bool D_FORM>
__global__ void __launch_bounds__(256) recon_rebuild_kernel(RebuildParams<T> p)
{
    using P = Pack<T, VEC>;
    constexpr int iA = 1, iB = NAX - 2, iC = NAX - 1;
    constexpr bool HAS_A = (NAX == 4);
    const long long rows = p.row1 - p.row0;
    const long long tiles = (p.units + 255) / 256;
    const long long L = xcd_remap(blockIdx.x, gridDim.x);
    const long long r = L / tiles, tile = L % tiles;
    const long long u = tile * 256 + threadIdx.x;
    if (u >= p.units || r >= rows) return;
    const long long v = p.row0 + r;
    const long long LR = p.C / VEC;
    const long long cv = u % LR, bb = (u / LR) % p.B, aa = u / (LR * p.B);
    const long long c0 = cv * VEC;
    const long long SM = p.A * p.B * p.C, SA = p.B * p.C, SB = p.C;
    const long long xs = aa * SA + bb * SB + c0;
    const long long row = (long long)((unsigned)v % p.ring) * SM, rown = (long long)((unsigned)(v + 1) % p.ring) * SM;
    const bool at_top = (v + 1 == p.top);
    const T tkp = p.tk_prev;
    // b of axis q at plane offset `off` of ring row `rw`
    auto b_pack = [&](int q, long long rw, long long off) {
        const P v1 = ldv<T, VEC>(p.in1[q] + rw + off);
        if (!D_FORM) return v1;
        const P v2 = ldv<T, VEC>(p.in2[q] + rw + off);
        P o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = v2.v[j] + tkp * (v2.v[j] - v1.v[j]);
        return o;
    };
    auto b_one = [&](int q, long long rw, long long off) {
        const T v1 = p.in1[q][rw + off];
        if (!D_FORM) return v1;
        const T v2 = p.in2[q][rw + off];
        return (T)(v2 + tkp * (v2 - v1));
    };
    P sum;
    {   // axis 0: the next ROW (zero past the top face)
        const P own = b_pack(0, row, xs);
        P nx;
        if (at_top) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) nx.v[j] = (T)0;
        } else if (p.next1 && v + 1 == p.row1) {  // the row after the launch's last one: handed over as planes
            const P v1 = ldv<T, VEC>(p.next1 + xs);
            if (D_FORM) {
                const P v2 = ldv<T, VEC>(p.next2 + xs);
#pragma unroll
                for (int j = 0; j < VEC; ++j) nx.v[j] = v2.v[j] + tkp * (v2.v[j] - v1.v[j]);
            } else {
                nx = v1;
            }
        } else {
            nx = b_pack(0, rown, xs);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) sum.v[j] = p.lm[0] * (own.v[j] - nx.v[j]);
    }
    if (HAS_A) {
        const long long off = (aa + 1 == p.A) ? -(p.A - 1) * SA : SA;
        const P own = b_pack(iA, row, xs), nx = b_pack(iA, row, xs + off);
#pragma unroll
        for (int j = 0; j < VEC; ++j) sum.v[j] = sum.v[j] + p.lm[iA] * (own.v[j] - nx.v[j]);
    }
    {
        const long long off = (bb + 1 == p.B) ? -(p.B - 1) * SB : SB;
        const P own = b_pack(iB, row, xs), nx = b_pack(iB, row, xs + off);
#pragma unroll
        for (int j = 0; j < VEC; ++j) sum.v[j] = sum.v[j] + p.lm[iB] * (own.v[j] - nx.v[j]);
    }
    {   // the contiguous axis: neighbours inside the pack from registers, the one after it fetched
        const P own = b_pack(iC, row, xs);
        const long long off_after = (c0 + VEC == p.C) ? -(p.C - VEC) : VEC;
        const T after = b_one(iC, row, xs + off_after);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const T nx = (j + 1 < VEC) ? own.v[j + 1 < VEC ? j + 1 : 0] : after;
            sum.v[j] = sum.v[j] + p.lm[iC] * (own.v[j] - nx);
        }
    }
    const P og = ldv_nt<T, VEC>(p.orig + (long long)((unsigned)v % p.ring_orig) * SM + xs);
    P out;
#pragma unroll
    for (int j = 0; j < VEC; ++j) out.v[j] = og.v[j] - sum.v[j];
    stv<T, VEC>(p.recon + row + xs, out);
}

template <typename T>
static int rebuild_impl(const RebuildArgs &a, hipStream_t s)
{
    constexpr int VMAX = 16 / sizeof(T);
    RebuildParams<T> p;
    std::memset(&p, 0, sizeof p);
    p.orig = (const T *)a.orig;
    p.recon = (T *)a.recon;
    bool al = aligned16(a.orig) && aligned16(a.recon);
    for (int q = 0; q < a.ndim; ++q) {
        p.in1[q] = (const T *)a.in1[q];
        p.in2[q] = (const T *)a.in2[q];
        p.lm[q] = (T)a.lambda_mu[q];
        al = al && aligned16(a.in1[q]) && aligned16(a.in2[q]);
    }
    p.next1 = (const T *)a.next1;
    p.next2 = (const T *)a.next2;
    al = al && aligned16(a.next1) && aligned16(a.next2);
    p.tk_prev = (T)a.tk_prev;
    p.A = a.ndim == 4 ? a.plane_shape[0] : 1;
    p.B = a.plane_shape[a.ndim - 3];
    p.C = a.plane_shape[a.ndim - 2];
    p.row0 = a.row0; p.row1 = a.row1; p.top = a.top;
    p.ring = (unsigned)a.ring_rows;
    p.ring_orig = (unsigned)a.orig_ring_rows;
    const int vec = (al && p.C % VMAX == 0) ? VMAX : 1;
    p.units = p.A * p.B * (p.C / vec);
    const long long tiles = (p.units + 255) / 256, grid = tiles * (a.row1 - a.row0);
    TVDN_REQUIRE(grid >= 1 && grid < (1LL << 31), "recon rebuild: grid %lld out of range", grid);
    const bool d = a.d_form != 0;
#define TVDN_REBUILD(V, N, D) hipLaunchKernelGGL((recon_rebuild_kernel<T, V, N, D>), dim3((unsigned)grid), dim3(256), 0, s, p)
    if (vec == VMAX) {
        if (a.ndim == 4) { if (d) TVDN_REBUILD(VMAX, 4, true); else TVDN_REBUILD(VMAX, 4, false); }
        else             { if (d) TVDN_REBUILD(VMAX, 3, true); else TVDN_REBUILD(VMAX, 3, false); }
    } else {
        if (a.ndim == 4) { if (d) TVDN_REBUILD(1, 4, true); else TVDN_REBUILD(1, 4, false); }
        else             { if (d) TVDN_REBUILD(1, 3, true); else TVDN_REBUILD(1, 3, false); }
    }
#undef TVDN_REBUILD
    TVDN_HIP(hipGetLastError());
    return TVDN_OK;
}

int recon_rebuild(const RebuildArgs &a, hipStream_t s)
{
    TVDN_REQUIRE(a.ndim == 3 || a.ndim == 4, "recon rebuild: ndim %d", a.ndim);
    TVDN_REQUIRE(a.row0 < a.row1 && a.ring_rows > 0 && a.orig_ring_rows > 0, "recon rebuild: bad rows / rings");
    TVDN_REQUIRE(a.row1 - a.row0 + (a.next1 ? 0 : 1) <= a.ring_rows, "recon rebuild: %lld rows and the one after them do not fit a ring of %lld", (long long)(a.row1 - a.row0),
                 (long long)a.ring_rows);
    return a.dtype == TVDN_F32 ? rebuild_impl<float>(a, s) : rebuild_impl<double>(a, s);
}

}  // namespace tvdn

extern "C" int tvdn_recon_from_state(int dtype, int ndim, const int64_t *shape, const void *orig, void *recon, const void *const *d,
                                     const void *const *dprev, const double *lambda_mu, double tk_prev, int64_t row0, int64_t row1, void *stream)
{
    using namespace tvdn;
    TVDN_REQUIRE(dtype == TVDN_F32 || dtype == TVDN_F64, "bad dtype %d", dtype);
    TVDN_REQUIRE(ndim == 3 || ndim == 4, "ndim must be 3 or 4, got %d", ndim);
    TVDN_REQUIRE(shape && orig && recon && d && lambda_mu, "NULL argument");
    for (int i = 0; i < ndim; ++i) TVDN_REQUIRE(shape[i] >= 1, "shape[%d] must be >= 1", i);
    TVDN_REQUIRE(0 <= row0 && row0 < row1 && row1 <= shape[0], "rows [%lld, %lld) not inside 0..%lld", (long long)row0, (long long)row1, (long long)shape[0]);
    RebuildArgs a;
    std::memset(&a, 0, sizeof a);
    a.dtype = dtype;
    a.ndim = ndim;
    for (int i = 1; i < ndim; ++i) a.plane_shape[i - 1] = shape[i];
    a.orig = orig;
    a.recon = recon;
    a.d_form = dprev != nullptr;
    for (int q = 0; q < ndim; ++q) {
        TVDN_REQUIRE(d[q] != nullptr && (!dprev || dprev[q] != nullptr), "NULL state pointer for axis %d", q);
        a.in1[q] = dprev ? dprev[q] : d[q];
        a.in2[q] = dprev ? d[q] : nullptr;
        a.lambda_mu[q] = lambda_mu[q];
    }
    a.tk_prev = tk_prev;
    a.row0 = row0;
    a.row1 = row1;
    a.top = shape[0];
    a.ring_rows = a.orig_ring_rows = shape[0] + 1;  // (contiguous arrays: row v at slot v; the row after the last is never read)
    return recon_rebuild(a, (hipStream_t)stream);
}

