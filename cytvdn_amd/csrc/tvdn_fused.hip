// Fused anisotropic-TV iteration for gfx950: all accumulator updates and the reconstruction
// update of one iteration (cyTVDN/cyTVDN.py:153-184, :205-230, :372-392, :405-420) in ONE sweep
// that reads each state array once and writes it once.
//
// Why it is legal: recon_new(x) needs b_new_ax(x) and b_new_ax(x+e_ax); b_new_ax(y) needs the OLD
// recon at y and y-e_ax and the OLD accumulator state at y.  All state is multi-buffered (the sweep
// reads *_in, writes *_out), so a thread may recompute its +1 neighbour's b_new from old values
// that nobody overwrites.  The recomputed value is bit-identical to the owner's (same inputs,
// same operations, no FMA contraction: this file is compiled with -ffp-contract=off).
//
// State representations (tvdn.h TVDN_ITER_*):
//   PLAIN        b            -> b'                      2 passes per axis
//   FISTA        (b, d)       -> (b', d')                4 passes per axis: the reference's own state
//   FISTA_D      (d_k, d_k-1) -> d_k+1                   3 passes per axis: b_k is not stored but rebuilt
//                as d_k + tk_prev*(d_k - d_k-1), the very expression (anisotropic.pyx:128) that produced it
//   FISTA_D_TO_PLAIN  (d_k, d_k-1) -> b'                 first unaccelerated iteration of a hybrid run
// so a 4-D FISTA iteration moves 15 arrays (r, orig, 8 d in; r, 4 d out) instead of the 19 of the
// reference's representation and the 27 of its five separate passes.
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

// Per-axis accumulator state as the kernel sees it.  Which arrays exist depends on the mode:
//   in1: b (PLAIN, FISTA) or d_k-1 (FISTA_D*);  in2: d_k (all FISTA modes)
//   out1: b' (PLAIN, FISTA, FISTA_D_TO_PLAIN);  out2: d' (FISTA, FISTA_D)
template <typename T>
struct AxisState {
    const T *in1;
    const T *in2;
    T *out1;
    T *out2;
};

template <int MODE>
struct ModeTraits {
    static constexpr bool kIn2 = (MODE != TVDN_ITER_PLAIN);
    static constexpr bool kOut1 = (MODE != TVDN_ITER_FISTA_D);
    static constexpr bool kOut2 = (MODE == TVDN_ITER_FISTA || MODE == TVDN_ITER_FISTA_D);
};

template <typename T>
struct FusedParams {
    const T *orig;
    const T *r_in;
    T *r_out;
    const T *wrap;  // TVDN_EDGE_WRAP: current recon of global row 0 (one plane)
    AxisState<T> ax[4];
    T tk;       // momentum ratio of this iteration
    T tk_prev;  // momentum ratio of the previous iteration (FISTA_D*: rebuilds b_k)
    T clip[4];
    T lm[4];
    long long M, A, B, C;
    long long row_lo, row_hi;
    long long sweep_lo, sweep_hi;  // rows advanced by this launch, inside [row_lo, row_hi)
    int lo_mode, hi_mode, bc;
    int chunk;        // rows per workgroup march
    long long tiles;  // workgroups per cross-section
    long long units;  // A * B * (C / VEC)
    int block;        // threads per workgroup of this launch: kFusedBlock, or kSmallBlock for small grids
    int xcd;          // 1: remap workgroup ids so each XCD sweeps a contiguous run of tiles
    // patch order of the tiles of a cross-section (0 = plain order): patches of patch_a A-rows x patch_t tiles
    long long patch_a, patch_t, tiles_per_arow;
    // column order (0 = off): every XCD sweeps a COLUMN of tiles through all marches, col_group tiles at a time -- march c + 1 of a tile is
    // dispatched one round of the XCD's workgroups after march c, so the row that march c looks ahead at is read again (as march c + 1's
    // first row) while it may still be in the XCD's L2
    long long col_group, n_march;
    // row rings (RING instantiations only): row m of r_in lives at slot m % ring, of the in1 / in2 arrays at m % ring_in1 / m %
    // ring_in2, of r_out at m % ring_rout, of the out1 / out2 arrays at m % ring_out, of orig at m % ring_orig (one ring size for
    // all of them but orig in a plain streamed pass; a "ring" longer than the cube is an array: rows kept in HBM, swept in place)
    unsigned ring, ring_in1, ring_in2, ring_rout, ring_out, ring_orig;
    int chain_lo, store_ahead;  // RING only (tvdn.h TVDN_SWEEP_*): the axis-0 accumulator handed across the cut between two launches
    double *partials;
};

#ifndef TVDN_FUSED_BLOCK
#define TVDN_FUSED_BLOCK 256
#endif
constexpr int kFusedBlock = TVDN_FUSED_BLOCK;  // threads per workgroup (128 / 512 / 1024 lose on every large shape: profiles/r03_ab_inproc_blocks_*.jsonl)
constexpr int kSmallBlock = 128;               // ... but small cross-sections gain 9 % from twice as many workgroups half the size

// One accumulator update at one voxel.  v1/v2 are the values loaded from in1/in2.
// Returns b_new (what the divergence and b_norm use); o1/o2 are what out1/out2 receive.
//   plain  anisotropic.pyx:50-54     b' = clip((r - r_prev) + b)
//   FISTA  anisotropic.pyx:127-132   d' = clip((r - r_prev) + b); b' = d' + tk*(d' - d)
template <typename T, int MODE>
__device__ __forceinline__ T acc_new(T r_x, T r_prev, T v1, T v2, T tk, T tk_prev, T clip, T &o1, T &o2)
{
    T b;
    if (MODE == TVDN_ITER_FISTA_D || MODE == TVDN_ITER_FISTA_D_TO_PLAIN)
        b = v2 + tk_prev * (v2 - v1);  // b_k rebuilt exactly as the previous iteration formed it
    else
        b = v1;
    const T dn = clipv((r_x - r_prev) + b, clip);
    if (MODE == TVDN_ITER_PLAIN || MODE == TVDN_ITER_FISTA_D_TO_PLAIN) {
        o1 = dn;
        o2 = dn;
        return dn;
    }
    const T bn = dn + tk * (dn - v2);
    o1 = bn;
    o2 = dn;
    return bn;
}

// Axis whose neighbours are whole packs (A and B): update own state at x, recompute b_new at the
// +1 neighbour, add lm * (b_new(x) - b_new(x+e)) to `sum` (left-to-right as utils.c:5641).
// Which loads of the A / B / C accumulator state are streaming (non-temporal) ones -- a compile-time mask, so that
// variants can be built and compared (tools/build_variants.sh, tools/ab_inproc.py):
//   bit 0: B own   bit 1: C own   bit 2: A own   bit 3: B next   bit 4: A next
// Measured with variants alternating on one allocation (tools/ab_inproc.py; profiles/r03_ab_inproc_ntmask_*.jsonl):
// streaming the A-axis state a thread reads at its own position is worth 0.2-1.1 % (config 2 / f64); every other bit
// costs 2-27 % (the B and C neighbours are served by the lines the own loads have just brought in).
#ifndef TVDN_NTMASK
#define TVDN_NTMASK 4
#endif
constexpr int kNtMask = TVDN_NTMASK;

// ---- row-based addressing -----------------------------------------------------------------------------------------
// Every access of the sweep is (start of a row-plane of one array: wave-uniform) + (position inside the plane: per
// lane, constant along the march, < 4 GiB).  Expressed as a raw buffer access -- resource = the row's base address in
// four SGPRs, offset = ONE 32-bit VGPR per neighbour shared by all fifteen arrays -- instead of a 64-bit address per
// load, the address arithmetic leaves the vector registers: that is what lets a whole row of loads be in flight
// at 4 waves per SIMD (below).  num_records is the full 32-bit range: the offsets are in range by construction.
typedef int tvdn_i4 __attribute__((ext_vector_type(4)));
typedef int tvdn_i2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t row_rsrc(const void *row_base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(row_base), 0, -1, 0x00020000);
}

// cache-policy bits of the raw buffer accesses (gfx940+: bit 0 = sc0, bit 1 = nt, bit 4 = sc1).  Measurement builds may override
// them (tools/build_variants.sh aux): streaming loads / other loads / stores.  Round 6, variants alternating on one allocation
// (profiles/r06_ab_inproc_aux.jsonl): on the loads nothing beats nt alone on the streams and no bit on the rest; stores with
// nt + sc1 are 0.9-1.3 % FASTER in the float64 forms (config 3: 17.24 -> 17.08 ms, FISTA f64: 23.32 -> 23.09) and 0.2 % slower in
// float32 FISTA (config 2: 11.206 -> 11.231), the same in the other float32 forms -- so the bit goes with the element size.
#ifndef TVDN_LDNT_AUX
#define TVDN_LDNT_AUX 2
#endif
#ifndef TVDN_LD_AUX
#define TVDN_LD_AUX 0
#endif
#ifndef TVDN_ST_AUX
#define TVDN_ST_AUX (kNtStores ? 2 : 0)
#endif
#ifndef TVDN_ST_AUX64
#define TVDN_ST_AUX64 (kNtStores ? 18 : 0)
#endif

template <typename T, int VEC, bool NT>
__device__ __forceinline__ Pack<T, VEC> ldb(const T *row_base, unsigned e)
{
    constexpr int BYTES = (int)sizeof(T) * VEC;
    constexpr int AUX = NT ? TVDN_LDNT_AUX : TVDN_LD_AUX;
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(row_base);
    const unsigned off = e * (unsigned)sizeof(T);
    Pack<T, VEC> x;
    if (BYTES == 16) {
        const tvdn_i4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, AUX);
        __builtin_memcpy(&x, &v, 16);
    } else if (BYTES == 8) {
        const tvdn_i2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, AUX);
        __builtin_memcpy(&x, &v, 8);
    } else {
        const int v = __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, AUX);
        __builtin_memcpy(&x, &v, 4);
    }
    return x;
}

template <typename T>
__device__ __forceinline__ T ldb1(const T *row_base, unsigned e)
{
    return ldb<T, 1, false>(row_base, e).v[0];
}

template <typename T, int VEC>
__device__ __forceinline__ void stb(T *row_base, unsigned e, const Pack<T, VEC> &x)
{
    constexpr int BYTES = (int)sizeof(T) * VEC;
    constexpr int AUX = sizeof(T) == 8 ? TVDN_ST_AUX64 : TVDN_ST_AUX;
    const __amdgpu_buffer_rsrc_t rs = row_rsrc(row_base);
    const unsigned off = e * (unsigned)sizeof(T);
    if (BYTES == 16) {
        tvdn_i4 v;
        __builtin_memcpy(&v, &x, 16);
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX);
    } else if (BYTES == 8) {
        tvdn_i2 v;
        __builtin_memcpy(&v, &x, 8);
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, off, 0, AUX);
    } else {
        int v;
        __builtin_memcpy(&v, &x, 4);
        __builtin_amdgcn_raw_buffer_store_b32(v, rs, off, 0, AUX);
    }
}

// Loads and arithmetic of an axis are separate steps: a row's loads -- M look-ahead, A, B, C and orig, 18 packs + 4 scalars
// in 4-D FISTA -- are ALL issued before the first value is used (fused_iter_kernel), so that a wave has a whole row of
// requests in flight instead of one axis at a time (the compiler keeps source order across the ~100 instructions of an
// axis, and each axis used to end in s_waitcnt vmcnt(0): five dependent round trips to memory per row).
template <typename T, int VEC>
struct PackLoads {  // axes A and B
    Pack<T, VEC> rp, rn, v1_own, v1_nx, v2_own, v2_nx;
};

template <typename T, int VEC>
struct ContigLoads {  // axis C
    T r_before, r_after, v1_after, v2_after;
    Pack<T, VEC> v1_own, v2_own;
};

template <typename T, int VEC, int MODE, bool nt_own, bool nt_next>
__device__ __forceinline__ void load_pack(PackLoads<T, VEC> &l, const T *r_row, const AxisState<T> &s, long long row1, long long row2,
                                          unsigned e0, unsigned e_prev, unsigned e_next)
{
    using M = ModeTraits<MODE>;
    l.rp = ldb<T, VEC, false>(r_row, e_prev);
    l.rn = ldb<T, VEC, false>(r_row, e_next);
    l.v1_own = ldb<T, VEC, nt_own>(s.in1 + row1, e0);
    l.v1_nx = ldb<T, VEC, nt_next>(s.in1 + row1, e_next);
    if (M::kIn2) {
        l.v2_own = ldb<T, VEC, nt_own>(s.in2 + row2, e0);
        l.v2_nx = ldb<T, VEC, nt_next>(s.in2 + row2, e_next);
    }
}

// Axis whose neighbours are whole packs (A and B): update own state at x, recompute b_new at the
// +1 neighbour, add lm * (b_new(x) - b_new(x+e)) to `sum` (left-to-right as utils.c:5641).
template <typename T, int VEC, int MODE>
__device__ __forceinline__ void axis_pack(const Pack<T, VEC> &r_cur, const PackLoads<T, VEC> &l, const AxisState<T> &s,
                                          long long row_o, unsigned e0, bool self_next, T tk, T tkp, T cl, T lm,
                                          Pack<T, VEC> &sum, double &bnorm)
{
    using P = Pack<T, VEC>;
    using M = ModeTraits<MODE>;
    P o1, o2;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const T bn_own = acc_new<T, MODE>(r_cur.v[j], l.rp.v[j], l.v1_own.v[j], M::kIn2 ? l.v2_own.v[j] : (T)0, tk, tkp, cl,
                                          o1.v[j], o2.v[j]);
        T u1, u2;
        const T bn_next = acc_new<T, MODE>(l.rn.v[j], self_next ? l.rn.v[j] : r_cur.v[j], l.v1_nx.v[j],
                                           M::kIn2 ? l.v2_nx.v[j] : (T)0, tk, tkp, cl, u1, u2);
        sum.v[j] = sum.v[j] + lm * (bn_own - bn_next);
        bnorm += fabs((double)bn_own);
    }
    if (M::kOut1) stb<T, VEC>(s.out1 + row_o, e0, o1);
    if (M::kOut2) stb<T, VEC>(s.out2 + row_o, e0, o2);
}

template <typename T, int VEC, int MODE, bool nt_own>
__device__ __forceinline__ void load_contig(ContigLoads<T, VEC> &l, const T *r_row, const AxisState<T> &s, long long row1, long long row2,
                                            unsigned e0, unsigned e_prev, unsigned e_next)
{
    using M = ModeTraits<MODE>;
    l.r_before = ldb1<T>(r_row, e_prev);
    l.r_after = ldb1<T>(r_row, e_next);
    l.v1_after = ldb1<T>(s.in1 + row1, e_next);
    l.v2_after = M::kIn2 ? ldb1<T>(s.in2 + row2, e_next) : (T)0;
    l.v1_own = ldb<T, VEC, nt_own>(s.in1 + row1, e0);
    if (M::kIn2) l.v2_own = ldb<T, VEC, nt_own>(s.in2 + row2, e0);
}

// Contiguous axis C: neighbours inside the pack come from registers; only the element before the
// pack and the one after it are fetched.
template <typename T, int VEC, int MODE>
__device__ __forceinline__ void axis_contig(const Pack<T, VEC> &r_cur, const ContigLoads<T, VEC> &l, const AxisState<T> &s,
                                            long long row_o, unsigned e0, bool self_next, T tk, T tkp, T cl, T lm,
                                            Pack<T, VEC> &sum, double &bnorm)
{
    using P = Pack<T, VEC>;
    using M = ModeTraits<MODE>;
    P o1, o2, bn_own;
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const T rp = (j == 0) ? l.r_before : r_cur.v[j > 0 ? j - 1 : 0];
        bn_own.v[j] = acc_new<T, MODE>(r_cur.v[j], rp, l.v1_own.v[j], M::kIn2 ? l.v2_own.v[j] : (T)0, tk, tkp, cl, o1.v[j],
                                       o2.v[j]);
        bnorm += fabs((double)bn_own.v[j]);
    }
    T u1, u2;
    const T bn_after = acc_new<T, MODE>(l.r_after, self_next ? l.r_after : r_cur.v[VEC - 1], l.v1_after, l.v2_after, tk, tkp,
                                        cl, u1, u2);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const T bn_next = (j + 1 < VEC) ? bn_own.v[j + 1 < VEC ? j + 1 : 0] : bn_after;
        sum.v[j] = sum.v[j] + lm * (bn_own.v[j] - bn_next);
    }
    if (M::kOut1) stb<T, VEC>(s.out1 + row_o, e0, o1);
    if (M::kOut2) stb<T, VEC>(s.out2 + row_o, e0, o2);
}

// RING: the arrays are rings of row-planes (tvdn.h, ring_rows): the only change is where a row starts.  Row numbers
// are wave-uniform, so the modulo is scalar work per row step; a separate instantiation keeps the resident path as is.
template <bool RING>
__device__ __forceinline__ long long row_slot(long long m, unsigned ring)
{
    return RING ? (long long)((unsigned)m % ring) : m;
}

// TVDN_WAVES_PER_EU (measurement builds only, tools/build_variants.sh waves): pin the occupancy the register allocator aims for
#ifdef TVDN_WAVES_PER_EU
#define TVDN_OCCUPANCY __attribute__((amdgpu_waves_per_eu(TVDN_WAVES_PER_EU, TVDN_WAVES_PER_EU)))
#else
#define TVDN_OCCUPANCY
#endif

template <typename T, int VEC, int NAX, int MODE, bool RING, int BLOCK>
__global__ void __launch_bounds__(BLOCK) TVDN_OCCUPANCY fused_iter_kernel(FusedParams<T> p)
{
    using P = Pack<T, VEC>;
    using MT = ModeTraits<MODE>;
    constexpr int iM = 0, iA = 1, iB = NAX - 2, iC = NAX - 1;  // accumulator slot per canonical axis
    constexpr bool HAS_A = (NAX == 4);

    const long long L = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : (long long)blockIdx.x;
    long long chunk_id = L / p.tiles;
    long long tile = L % p.tiles;
    if (p.col_group > 0) {
        const long long per = (long long)gridDim.x / 8, x = L / per, l = L % per;  // (the XCD remap is on: this XCD's run of ids)
        const long long blk = p.n_march * p.col_group, g = l / blk, r = l % blk;
        chunk_id = r / p.col_group;
        tile = x * (p.tiles / 8) + g * p.col_group + r % p.col_group;
    }
    if (HAS_A && p.patch_a > 1) {
        // The workgroups resident on an XCD at one time cover ~128 consecutive tiles.  In plain order that is a run
        // of 512 KiB of B x C, which for long A-rows (config-4 planes: 256 KiB each) holds only two of them, so the
        // A neighbours -- 64 tiles away -- are rarely in flight together.  Patch order makes the run a 2-D patch of
        // patch_a A-rows x patch_t tiles: every row's A neighbours are patch_t tiles away, whatever the plane.
        const long long per = p.patch_a * p.patch_t, n_gt = p.tiles_per_arow / p.patch_t;
        const long long patch = tile / per, r = tile % per;
        const long long pa = patch / n_gt, pt = patch % n_gt;
        tile = (pa * p.patch_a + r / p.patch_t) * p.tiles_per_arow + pt * p.patch_t + r % p.patch_t;
    }

    const long long u = tile * BLOCK + threadIdx.x;
    double acc[3] = {0.0, 0.0, 0.0};  // b_norm, sum|delta|, sum|old|

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
        // a wrapped neighbour sits at index 0, where under Jia-Zhao its own "prev" is itself
        const bool selfA = wrapA && bc2, selfB = wrapB && bc2, selfC = wrapC && bc2;
        // positions inside a row-plane, in elements: 32-bit, constant along the march, shared by every array
        const unsigned e0 = (unsigned)xs;
        const unsigned eA_prev = (unsigned)(xs + offA_prev), eA_next = (unsigned)(xs + offA_next);
        const unsigned eB_prev = (unsigned)(xs + offB_prev), eB_next = (unsigned)(xs + offB_next);
        const unsigned eC_prev = (unsigned)(xs + offC_prev), eC_next = (unsigned)(xs + offC_next);

        const T tk = p.tk, tkp = p.tk_prev;
        const T clM = p.clip[iM], clB = p.clip[iB], clC = p.clip[iC];
        const T lmM = p.lm[iM], lmB = p.lm[iB], lmC = p.lm[iC];
        const AxisState<T> sM = p.ax[iM];

        // ---- prologue: M-axis accumulator of row m0 -----------------------------------------------
        P r_cur = ldb<T, VEC, false>(p.r_in + row_slot<RING>(m0, p.ring) * SM, e0);
        P bM_cur;
        if (RING && p.chain_lo && m0 == p.sweep_lo) {  // (the launch's first march only: its other marches start inside the launch)
            // The launch before this one (same iteration, rows ending at m0) stored the axis-0 output state of row m0 ahead
            // (store_ahead, below): b_new(m0) is that value itself, or -- where d' is what is stored -- d' + tk * (d' - d_k),
            // the expression that formed it (acc_new): same operands, same roundings, one plane read instead of three, no store.
            const long long row0_o = row_slot<RING>(m0, p.ring_out) * SM;
            if (MODE == TVDN_ITER_FISTA_D) {
                const P dn = ldb<T, VEC, false>(sM.out2 + row0_o, e0);
                const P v2 = ldb<T, VEC, kNtLoads>(sM.in2 + row_slot<RING>(m0, p.ring_in2) * SM, e0);
#pragma unroll
                for (int j = 0; j < VEC; ++j) bM_cur.v[j] = dn.v[j] + tk * (dn.v[j] - v2.v[j]);
            } else {
                bM_cur = ldb<T, VEC, false>(sM.out1 + row0_o, e0);  // PLAIN / D_TO_PLAIN: b' = d'; FISTA: b' is stored as such
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[0] += fabs((double)bM_cur.v[j]);
        } else {
            long long mp;  // the row that precedes m0
            if (m0 > p.row_lo || p.lo_mode == TVDN_EDGE_HALO)
                mp = m0 - 1;
            else
                mp = bc2 ? m0 : p.row_hi - 1;  // TVDN_EDGE_BC: Jia-Zhao -> itself, periodic -> last row
            const long long row0 = row_slot<RING>(m0, p.ring) * SM;
            const long long row0_1 = RING ? row_slot<RING>(m0, p.ring_in1) * SM : row0, row0_2 = RING ? row_slot<RING>(m0, p.ring_in2) * SM : row0;
            const long long row0_o = RING ? row_slot<RING>(m0, p.ring_out) * SM : row0;
            const P r_prev = ldb<T, VEC, false>(p.r_in + row_slot<RING>(mp, p.ring) * SM, e0);
            const P v1 = ldb<T, VEC, kNtLoads>(sM.in1 + row0_1, e0);
            P v2, o1, o2;
            if (MT::kIn2) v2 = ldb<T, VEC, kNtLoads>(sM.in2 + row0_2, e0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                bM_cur.v[j] = acc_new<T, MODE>(r_cur.v[j], r_prev.v[j], v1.v[j], MT::kIn2 ? v2.v[j] : (T)0, tk, tkp, clM,
                                               o1.v[j], o2.v[j]);
                acc[0] += fabs((double)bM_cur.v[j]);
            }
            if (MT::kOut1) stb<T, VEC>(sM.out1 + row0_o, e0, o1);
            if (MT::kOut2) stb<T, VEC>(sM.out2 + row0_o, e0, o2);
        }

        // ---- march -------------------------------------------------------------------------------
        for (long long m = m0; m < m1; ++m) {
            // wave-uniform start of this row: in r_in, in the in1 / in2 arrays, in r_out, in the out1 / out2 arrays
            const long long row = row_slot<RING>(m, p.ring) * SM;
            const long long row_1 = RING ? row_slot<RING>(m, p.ring_in1) * SM : row, row_2 = RING ? row_slot<RING>(m, p.ring_in2) * SM : row;
            const long long row_ro = RING ? row_slot<RING>(m, p.ring_rout) * SM : row, row_o = RING ? row_slot<RING>(m, p.ring_out) * SM : row;
            const T *r_row = p.r_in + row;
            const bool last = (m + 1 == m1);
            const bool at_end = (m + 1 == p.row_hi);

            // (0) every load of this row, issued before anything is used: the M look-ahead row, then A, B, C and orig
            P r_next = r_cur, bM_next, mv1, mv2;
#pragma unroll
            for (int j = 0; j < VEC; ++j) bM_next.v[j] = (T)0;  // TVDN_EDGE_ZERO
            const bool look = !(at_end && p.hi_mode == TVDN_EDGE_ZERO);
            const bool wrap = at_end && p.hi_mode == TVDN_EDGE_BC;
            // TVDN_EDGE_WRAP (Jia-Zhao, last slab of several): the wrapped neighbour is global row 0, whose axis-0
            // accumulator upstream forms as clip((r0 - r0) + b0) -- zero while row 0 is finite, NaN from the moment it
            // is not (anisotropic.pyx:65-73).  p.wrap holds row 0's current recon; its state is that zero (the
            // state loads below then land on the own row and are discarded).
            const bool wrapz = at_end && p.hi_mode == TVDN_EDGE_WRAP;
            const long long mn = wrap ? p.row_lo : (wrapz ? m : m + 1);
            const long long rown = row_slot<RING>(mn, p.ring) * SM;
            const long long rown_1 = RING ? row_slot<RING>(mn, p.ring_in1) * SM : rown, rown_2 = RING ? row_slot<RING>(mn, p.ring_in2) * SM : rown;
            const long long rown_o = RING ? row_slot<RING>(mn, p.ring_out) * SM : rown;
#ifdef TVDN_ORIG_FIRST  // (measurement build: the data term's load issued first instead of last; no difference, profiles/r06_ab_inproc_order.jsonl)
            const P og = ldb<T, VEC, kNtLoads>(p.orig + (RING ? row_slot<RING>(m, p.ring_orig) * SM : row), e0);
#endif
            if (look) {
                r_next = ldb<T, VEC, false>(wrapz ? p.wrap : p.r_in + rown, e0);
                mv1 = ldb<T, VEC, kNtLoads>(sM.in1 + rown_1, e0);
                if (MT::kIn2) mv2 = ldb<T, VEC, kNtLoads>(sM.in2 + rown_2, e0);
            }
            // With 64-bit addresses per load this needed 144-163 VGPRs in the 4-D FISTA forms (3 waves per SIMD); with the
            // row-based buffer addressing above it is 112 (f32) / 113 (f64) there and 68-87 elsewhere.  Against the
            // axis-by-axis build of the same day, variants alternating on ONE allocation (profiles/r03_ab_inproc_buffer_
            // addressing.jsonl): 4-D unaccelerated f32 -4.3 %, 3-D 128x128x512 -6.8 %, 3-D 512^3 -2.6 %, config-4 slab
            // -1.3 %, config 2 and config 3 -0.6 %, 4-D FISTA f64 +0.4 %, 3-D unaccelerated +-0.
            PackLoads<T, VEC> lA, lB;
            ContigLoads<T, VEC> lC;
            if (HAS_A)
                load_pack<T, VEC, MODE, (kNtMask & 4) != 0, (kNtMask & 16) != 0>(lA, r_row, p.ax[iA], row_1, row_2, e0, eA_prev, eA_next);
            load_pack<T, VEC, MODE, (kNtMask & 1) != 0, (kNtMask & 8) != 0>(lB, r_row, p.ax[iB], row_1, row_2, e0, eB_prev, eB_next);
            load_contig<T, VEC, MODE, (kNtMask & 2) != 0>(lC, r_row, p.ax[iC], row_1, row_2, e0, eC_prev, eC_next);
#ifndef TVDN_ORIG_FIRST
            const P og = ldb<T, VEC, kNtLoads>(p.orig + (RING ? row_slot<RING>(m, p.ring_orig) * SM : row), e0);
#endif

            // (1) M-axis accumulator of the next row (look-ahead by one row)
            if (look) {
                P o1, o2;
                if (wrapz) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        mv1.v[j] = (T)0;
                        if (MT::kIn2) mv2.v[j] = (T)0;
                    }
                }
                const bool self = (wrap && bc2) || wrapz;
#pragma unroll
                for (int j = 0; j < VEC; ++j)
                    bM_next.v[j] = acc_new<T, MODE>(r_next.v[j], self ? r_next.v[j] : r_cur.v[j], mv1.v[j],
                                                    MT::kIn2 ? mv2.v[j] : (T)0, tk, tkp, clM, o1.v[j], o2.v[j]);
                // rows inside the chunk are owned here, and so is a halo row sitting at row_hi -- and, between two launches of one
                // iteration that hand the accumulator across their cut (store_ahead), the first row of the next launch
                if (!last || (at_end && p.hi_mode == TVDN_EDGE_HALO) || (RING && p.store_ahead && m1 == p.sweep_hi && !at_end)) {
                    if (MT::kOut1) stb<T, VEC>(sM.out1 + rown_o, e0, o1);
                    if (MT::kOut2) stb<T, VEC>(sM.out2 + rown_o, e0, o2);
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
                axis_pack<T, VEC, MODE>(r_cur, lA, p.ax[iA], row_o, e0, selfA, tk, tkp, p.clip[iA], p.lm[iA], sum, acc[0]);
            axis_pack<T, VEC, MODE>(r_cur, lB, p.ax[iB], row_o, e0, selfB, tk, tkp, clB, lmB, sum, acc[0]);
            axis_contig<T, VEC, MODE>(r_cur, lC, p.ax[iC], row_o, e0, selfC, tk, tkp, clC, lmC, sum, acc[0]);

            // (3) reconstruction update at row m (utils.pyx:90-104)
            P r_new;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                r_new.v[j] = og.v[j] - sum.v[j];
                const T df = r_new.v[j] - r_cur.v[j];
                acc[1] += fabs((double)df);
                acc[2] += fabs((double)r_cur.v[j]);
            }
            stb<T, VEC>(p.r_out + row_ro, e0, r_new);

            r_cur = r_next;
            bM_cur = bM_next;
        }
    }
    block_store_partials<3, BLOCK>(acc, p.partials);
}

template <typename T, int VEC, int NAX, int MODE>
static int launch_fused_t(const FusedParams<T> &p, int grid, hipStream_t s)
{
    if (p.ring)
        hipLaunchKernelGGL((fused_iter_kernel<T, VEC, NAX, MODE, true, kFusedBlock>), dim3(grid), dim3(kFusedBlock), 0, s, p);
    else if (p.block == kSmallBlock)
        hipLaunchKernelGGL((fused_iter_kernel<T, VEC, NAX, MODE, false, kSmallBlock>), dim3(grid), dim3(kSmallBlock), 0, s, p);
    else
        hipLaunchKernelGGL((fused_iter_kernel<T, VEC, NAX, MODE, false, kFusedBlock>), dim3(grid), dim3(kFusedBlock), 0, s, p);
    TVDN_HIP(hipGetLastError());
    return TVDN_OK;
}

template <typename T, int VEC, int NAX>
static int launch_fused_m(const FusedParams<T> &p, int mode, int grid, hipStream_t s)
{
    switch (mode) {
    case TVDN_ITER_PLAIN: return launch_fused_t<T, VEC, NAX, TVDN_ITER_PLAIN>(p, grid, s);
    case TVDN_ITER_FISTA: return launch_fused_t<T, VEC, NAX, TVDN_ITER_FISTA>(p, grid, s);
    case TVDN_ITER_FISTA_D: return launch_fused_t<T, VEC, NAX, TVDN_ITER_FISTA_D>(p, grid, s);
    default: return launch_fused_t<T, VEC, NAX, TVDN_ITER_FISTA_D_TO_PLAIN>(p, grid, s);
    }
}

template <typename T>
static int iterate_fused_impl(tvdn_ctx *ctx, const tvdn_iter_args *a, double *sums_out, hipStream_t s)
{
    constexpr int VMAX = 16 / sizeof(T);
    const Geom g = make_geom(a->ndim, a->shape);
    const int nax = a->ndim;
    const int mode = a->mode;
    FusedParams<T> p;
    std::memset(&p, 0, sizeof p);
    p.orig = (const T *)a->orig;
    p.r_in = (const T *)a->recon_in;
    p.r_out = (T *)a->recon_out;
    bool al = aligned16(p.orig) && aligned16(p.r_in) && aligned16(p.r_out);
    for (int q = 0; q < nax; ++q) {
        AxisState<T> &x = p.ax[q];
        switch (mode) {
        case TVDN_ITER_PLAIN:
            x.in1 = (const T *)a->b_in[q]; x.out1 = (T *)a->b_out[q];
            break;
        case TVDN_ITER_FISTA:
            x.in1 = (const T *)a->b_in[q]; x.in2 = (const T *)a->d_in[q];
            x.out1 = (T *)a->b_out[q]; x.out2 = (T *)a->d_out[q];
            break;
        case TVDN_ITER_FISTA_D:
            x.in1 = (const T *)a->dprev_in[q]; x.in2 = (const T *)a->d_in[q]; x.out2 = (T *)a->d_out[q];
            break;
        default:  // TVDN_ITER_FISTA_D_TO_PLAIN
            x.in1 = (const T *)a->dprev_in[q]; x.in2 = (const T *)a->d_in[q]; x.out1 = (T *)a->b_out[q];
            break;
        }
        p.clip[q] = (T)a->clip[q];
        p.lm[q] = (T)a->lambda_mu[q];
        al = al && aligned16(x.in1) && aligned16(x.in2) && aligned16(x.out1) && aligned16(x.out2);
    }
    p.tk = (T)a->tk;
    p.tk_prev = (T)a->tk_prev;
    p.M = g.n[0]; p.A = g.n[1]; p.B = g.n[2]; p.C = g.n[3];
    p.row_lo = a->row_lo; p.row_hi = a->row_hi;
    const bool whole = (a->sweep_lo == 0 && a->sweep_hi == 0);
    p.sweep_lo = whole ? a->row_lo : a->sweep_lo;
    p.sweep_hi = whole ? a->row_hi : a->sweep_hi;
    p.lo_mode = a->lo_mode; p.hi_mode = a->hi_mode; p.bc = a->bc_mode;
    if (a->hi_mode == TVDN_EDGE_WRAP)  // an explicit plane, or by convention the row that follows the own rows
        p.wrap = a->wrap_recon ? (const T *)a->wrap_recon : p.r_in + a->row_hi * (p.A * p.B * p.C);
    p.ring = (unsigned)(a->recon_in_ring_rows ? a->recon_in_ring_rows : a->ring_rows);
    {   // in1 / in2 by mode (above): b_in and d_in are the state of this level, dprev_in that of the level before
        const unsigned cur = (unsigned)(a->cur_ring_rows ? a->cur_ring_rows : a->ring_rows), prv = (unsigned)(a->prev_ring_rows ? a->prev_ring_rows : a->ring_rows);
        const bool d_modes = mode == TVDN_ITER_FISTA_D || mode == TVDN_ITER_FISTA_D_TO_PLAIN;
        p.ring_in1 = d_modes ? prv : cur;
        p.ring_in2 = cur;
    }
    p.ring_rout = (unsigned)(a->recon_out_ring_rows ? a->recon_out_ring_rows : a->ring_rows);
    p.ring_out = (unsigned)(a->out_ring_rows ? a->out_ring_rows : a->ring_rows);
    p.ring_orig = (unsigned)(a->ring_rows ? (a->orig_ring_rows ? a->orig_ring_rows : a->shape[0]) : 0);
    p.chain_lo = (a->ring_rows > 0 && (a->chain & TVDN_SWEEP_CHAIN_LO)) ? 1 : 0;
    p.store_ahead = (a->ring_rows > 0 && (a->chain & TVDN_SWEEP_STORE_AHEAD)) ? 1 : 0;
    if (!p.ring && getenv("TVDN_FORCE_RING") && a->shape[0] < (1LL << 31))  // measurement knob: the ring instantiation on
        p.ring = p.ring_in1 = p.ring_in2 = p.ring_rout = p.ring_out = p.ring_orig = (unsigned)a->shape[0];  // resident arrays (same rows, same bits)
    p.partials = ctx->partials;

    // packs of 16 bytes where the last axis is a multiple of them, else one element per thread (5-7 % slower: 128 x 128 x 513 106
    // against 114 Gvoxel-iters/s).  Packs of 8 bytes for float32 cubes with an even last axis (126, 510 channels) were built and
    // measured in round 6: no faster than the scalar form (128 x 128 x 510: 102.7 against 106.6; 256 x 256 x 126 x 126: 89.8 / 90.3).
    const int vec = (al && (p.C % VMAX == 0)) ? VMAX : 1;
    TVDN_REQUIRE((unsigned long long)(p.A * p.B * p.C) * sizeof(T) < (1ull << 32),
                 "a row-plane (shape[1:]) of %lld elements is 4 GiB or more: positions inside a plane are 32-bit byte offsets",
                 (long long)(p.A * p.B * p.C));
    p.units = p.A * p.B * (p.C / vec);
    const long long rows = p.sweep_hi - p.sweep_lo;
    // tuning knobs (measurement only): TVDN_CHUNK = rows per march, TVDN_XCD = 0 / 1 forces the XCD remap off / on,
    // TVDN_PATCH = "A-rows,tiles" ("0": plain order), TVDN_BLOCK = threads per workgroup (128 / 256)
    const char *e_chunk = getenv("TVDN_CHUNK"), *e_xcd = getenv("TVDN_XCD"), *e_block = getenv("TVDN_BLOCK");
    long long chunk = 8, grid = 0;
    p.block = kFusedBlock;
    for (int attempt = 0; attempt < 2; ++attempt) {
        p.tiles = (p.units + p.block - 1) / p.block;
        // rows per march: long enough to amortise the look-ahead row (3 extra pack loads per chunk), short enough that the
        // workgroups in flight stay close together; 8 beats 2, 4, 6, 12, 16 and 32 on the large shapes when the variants
        // alternate on one allocation (profiles/r03_ab_inproc_blocks_chunks_xcd.jsonl)
        chunk = e_chunk ? atoll(e_chunk) : 8;
        if (chunk < 1) chunk = 1;
        while (chunk > 4 && p.tiles * ((rows + chunk - 1) / chunk) < 256 * 8) chunk /= 2;
        // the smallest cubes are bound by the serial row steps of a march, not by bytes: under 1024 workgroups the marches get
        // shorter still (32 x 32 x 128: 11.5 -> 8.3 us per sweep at one row per march, 64 x 64 x 256: 15.9 -> 15.0 at two,
        // 16 x 16 x 64 x 64: 19.1 -> 17.9; from 96 x 96 x 384 on four rows and more are the faster: profiles/r06_small_chunk.txt)
        while (!e_chunk && chunk > 1 && p.tiles * ((rows + chunk - 1) / chunk) < 1024) chunk /= 2;
        // the reduction scratch grows with the grid (ensure_partials), so big planes keep their short marches;
        // only beyond kMaxPartialBlocks workgroups do the marches get longer
        while (p.tiles * ((rows + chunk - 1) / chunk) > kMaxPartialBlocks && chunk < rows) chunk *= 2;
        grid = p.tiles * ((rows + chunk - 1) / chunk);
        // Small cross-sections (the 128x128x512 shape of BASELINE configs[0]: 2048 workgroups of 256 threads, two per SIMD
        // slot set): workgroups of 128 threads, twice as many, sweep 9 % faster (0.0795 against 0.0872 ms, same file);
        // from 512^3 on the sizes are even, and on every 4-D cube 256 wins.
        const bool small = e_block ? atoi(e_block) == kSmallBlock : grid < 4096;
        if (attempt == 0 && small && !p.ring && vec == VMAX && kFusedBlock != kSmallBlock && p.units % kSmallBlock == 0)
            p.block = kSmallBlock;
        else
            break;
    }
    p.chunk = (int)chunk;
    {
        // Order of the tiles of a cross-section.  A-rows of fewer than 32 tiles (config 2: 16): plain order, and workgroup
        // ids remapped so that every XCD (private L2) sweeps a contiguous run.  Longer A-rows: patches of 32 A-rows x 8
        // tiles walked by all eight XCDs together (no remap) -- consecutive workgroups go to consecutive XCDs, so the A
        // neighbour, 8 tiles on, lands on the SAME XCD one slot later, and the chip as a whole streams one 1 MiB region
        // per array instead of eight.  Variants alternating on one allocation (profiles/r03_ab_inproc_xcd0_patch.jsonl,
        // r03_ab_inproc_blocks_chunks_xcd_2.jsonl): config-4 slab 24.27 -> 23.17 ms (-4.6 %), config 3 (f64, 32 tiles per
        // A-row) 16.93 -> 16.54 ms (-2.3 %); round 2's 8x16 / 16x8 patches under the remap: 23.3-23.4 / 16.7 ms.
        // Config 2 is indifferent to all of it (11.45-11.52 ms), so it keeps the simpler order.
        const long long row_units = p.B * (p.C / vec);
        const long long tr = (row_units % p.block == 0) ? row_units / p.block : 0;
        long long ga = (p.A % 32 == 0) ? 32 : ((p.A % 16 == 0) ? 16 : 8), gt = 8;
        const char *e_patch = getenv("TVDN_PATCH");
        if (e_patch) {
            ga = atoll(e_patch);
            const char *c = strchr(e_patch, ',');
            gt = c ? atoll(c + 1) : 8;
        }
        p.patch_a = p.patch_t = p.tiles_per_arow = 0;
        // ... in launches of many marches (config 3: 32).  The launches of a slab iteration are short -- 8-row edge blocks
        // and a 48-row interior, 1 and 6 marches -- and there the plain order, still without the remap, is the better one
        // (same file of the A/B as bench.py --slab-of 8 runs it, profiles/r03_ab_inproc_slab_mode.jsonl: 22.99 ms against
        // 23.39 with the patches and 23.16 with round 2's order).
        const long long n_march = (rows + chunk - 1) / chunk;
        const bool long_rows = nax == 4 && tr >= 32;
        if (nax == 4 && ga > 1 && gt >= 1 && tr > 0 && tr % gt == 0 && (e_patch ? tr >= gt : (long_rows && n_march >= 16)) && p.A % ga == 0) {
            p.patch_a = ga; p.patch_t = gt; p.tiles_per_arow = tr;
        }
        p.xcd = e_xcd ? atoi(e_xcd) : (long_rows ? 0 : 1);
        // measurement knob TVDN_COLGROUP=n (round 6; profiles/r06_ab_inproc_colgroup.jsonl): column order, n tiles at a time per XCD
        p.col_group = 0;
        p.n_march = n_march;
        if (const char *e_col = getenv("TVDN_COLGROUP")) {
            const long long cg = atoll(e_col);
            if (cg > 0 && p.xcd && p.patch_a == 0 && grid % 8 == 0 && p.tiles % (8 * cg) == 0 && rows % chunk == 0) p.col_group = cg;
        }
    }
    TVDN_REQUIRE(grid >= 1 && grid <= kMaxPartialBlocks, "fused grid %lld out of range (max %d)", grid, kMaxPartialBlocks);
    {
        const int rce = ensure_partials(ctx, grid);
        if (rce) return rce;
        p.partials = ctx->partials;
    }
    // inside the library's own loops a small launch parks its partial rows for a batched fold (tvdn_common.hpp)
    double *parked = sums_defer_slot(ctx, grid, a->accumulate != 0);
    if (parked) {
        p.partials = parked;
    } else {
        const int rcf = sums_defer_flush(ctx, s);  // what is parked lands before a fold that is not
        if (rcf) return rcf;
    }

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->timing) {
        TVDN_HIP(hipEventCreate(&ev0));
        TVDN_HIP(hipEventCreate(&ev1));
        TVDN_HIP(hipEventRecord(ev0, s));
    }
    int rc;
    if (vec == VMAX)
        rc = (nax == 4) ? launch_fused_m<T, VMAX, 4>(p, mode, (int)grid, s) : launch_fused_m<T, VMAX, 3>(p, mode, (int)grid, s);
    else
        rc = (nax == 4) ? launch_fused_m<T, 1, 4>(p, mode, (int)grid, s) : launch_fused_m<T, 1, 3>(p, mode, (int)grid, s);
    if (ctx->timing) {
        TVDN_HIP(hipEventRecord(ev1, s));
        ctx->events.emplace_back(ev0, ev1);
    }
    if (rc) return rc;
    if (parked) return sums_defer_push(ctx, (int)grid, sums_out, s);
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
    TVDN_REQUIRE(a->mode >= TVDN_ITER_PLAIN && a->mode <= TVDN_ITER_FISTA_D_TO_PLAIN, "bad mode %d", a->mode);
    TVDN_REQUIRE(0 <= a->row_lo && a->row_lo < a->row_hi && a->row_hi <= a->shape[0],
                 "own rows [%lld,%lld) not inside 0..%lld", (long long)a->row_lo, (long long)a->row_hi,
                 (long long)a->shape[0]);
    TVDN_REQUIRE((a->sweep_lo == 0 && a->sweep_hi == 0) ||
                     (a->row_lo <= a->sweep_lo && a->sweep_lo < a->sweep_hi && a->sweep_hi <= a->row_hi),
                 "sweep rows [%lld,%lld) not inside the own rows", (long long)a->sweep_lo, (long long)a->sweep_hi);
    TVDN_REQUIRE(a->lo_mode == TVDN_EDGE_BC || a->lo_mode == TVDN_EDGE_HALO, "bad lo_mode %d", a->lo_mode);
    TVDN_REQUIRE(a->hi_mode >= TVDN_EDGE_BC && a->hi_mode <= TVDN_EDGE_WRAP, "bad hi_mode %d", a->hi_mode);
    TVDN_REQUIRE(!(a->hi_mode == TVDN_EDGE_WRAP && a->bc_mode != TVDN_BC_JIA_ZHAO), "hi_mode WRAP is a Jia-Zhao property");
    TVDN_REQUIRE(!(a->hi_mode == TVDN_EDGE_WRAP && !a->wrap_recon && a->row_hi >= a->shape[0]),
                 "hi_mode WRAP needs wrap_recon or a row at row_hi holding global row 0");
    TVDN_REQUIRE(!(a->lo_mode == TVDN_EDGE_HALO && a->row_lo < 1), "lo_mode HALO needs a row below row_lo");
    TVDN_REQUIRE(!(a->hi_mode == TVDN_EDGE_HALO && a->row_hi >= a->shape[0]), "hi_mode HALO needs a row at row_hi");
    TVDN_REQUIRE(!(a->hi_mode == TVDN_EDGE_ZERO && a->bc_mode != TVDN_BC_JIA_ZHAO), "hi_mode ZERO is a Jia-Zhao property");
    TVDN_REQUIRE(!(a->lo_mode == TVDN_EDGE_BC && a->bc_mode == TVDN_BC_PERIODIC && a->hi_mode != TVDN_EDGE_BC),
                 "periodic BC with lo_mode BC needs the whole ring in this block (hi_mode BC)");
    const int64_t own_rings[5] = {a->recon_in_ring_rows, a->cur_ring_rows, a->prev_ring_rows, a->recon_out_ring_rows, a->out_ring_rows};
    TVDN_REQUIRE(a->ring_rows >= 0 && a->orig_ring_rows >= 0, "negative ring size");
    TVDN_REQUIRE(a->ring_rows > 0 || a->orig_ring_rows == 0, "orig_ring_rows without ring_rows");
    for (int64_t r : own_rings) TVDN_REQUIRE(r >= 0 && (r == 0 || a->ring_rows > 0), "a ring size of its own needs ring_rows, and none is negative");
    if (a->ring_rows > 0) {
        const long long s0 = (a->sweep_lo == 0 && a->sweep_hi == 0) ? a->row_lo : a->sweep_lo;
        const long long s1 = (a->sweep_lo == 0 && a->sweep_hi == 0) ? a->row_hi : a->sweep_hi;
        // rows of the level below that one launch reads: the swept rows, the row before and the row after
        const long long need = (s1 - s0) + ((s0 > a->row_lo || a->lo_mode == TVDN_EDGE_HALO) ? 1 : 0) +
                               ((s1 < a->row_hi || a->hi_mode == TVDN_EDGE_HALO) ? 1 : 0);
        TVDN_REQUIRE(a->shape[0] < (1LL << 31) && a->ring_rows < (1LL << 31) && a->orig_ring_rows < (1LL << 31), "row rings index rows with 32 bits");
        for (int i = 0; i < 5; ++i) {  // (what is READ may be a ring of ONE row: every row is that plane -- a constant, e.g. the zeros a run starts from)
            const int64_t r = own_rings[i];
            TVDN_REQUIRE(r < (1LL << 31) && (r == 0 || r >= need || (r == 1 && i < 3)), "ring of %lld rows cannot hold the %lld rows this sweep touches", (long long)r, need);
        }
        TVDN_REQUIRE(a->ring_rows >= need, "ring of %lld rows cannot hold the %lld rows this sweep reads",
                     (long long)a->ring_rows, need);
        TVDN_REQUIRE(a->orig_ring_rows == 0 || a->orig_ring_rows >= s1 - s0, "orig ring shorter than the sweep");
        TVDN_REQUIRE(!(a->hi_mode == TVDN_EDGE_WRAP && !a->wrap_recon), "hi_mode WRAP on a ring needs wrap_recon");
        TVDN_REQUIRE((a->chain & ~(TVDN_SWEEP_CHAIN_LO | TVDN_SWEEP_STORE_AHEAD)) == 0, "unknown bits in chain: %d", a->chain);
        TVDN_REQUIRE(!(a->chain & TVDN_SWEEP_CHAIN_LO) || s0 > a->row_lo, "TVDN_SWEEP_CHAIN_LO needs a launch of this iteration that ended at sweep_lo");
        TVDN_REQUIRE(!(a->chain & TVDN_SWEEP_STORE_AHEAD) || s1 < a->row_hi, "TVDN_SWEEP_STORE_AHEAD needs a row after the sweep inside the own rows");
    }
    TVDN_REQUIRE(a->orig && a->recon_in && a->recon_out, "NULL state pointer");
    TVDN_REQUIRE(a->recon_in != a->recon_out, "the fused sweep is not in-place: recon_in == recon_out");
    for (int q = 0; q < a->ndim; ++q) {
        const bool need_b_in = (a->mode == TVDN_ITER_PLAIN || a->mode == TVDN_ITER_FISTA);
        const bool need_b_out = (a->mode != TVDN_ITER_FISTA_D);
        const bool need_d_in = (a->mode != TVDN_ITER_PLAIN);
        const bool need_d_out = (a->mode == TVDN_ITER_FISTA || a->mode == TVDN_ITER_FISTA_D);
        const bool need_dprev = (a->mode == TVDN_ITER_FISTA_D || a->mode == TVDN_ITER_FISTA_D_TO_PLAIN);
        if (need_b_in) TVDN_REQUIRE(a->b_in[q] != nullptr, "b_in[%d] is NULL", q);
        if (need_b_out) TVDN_REQUIRE(a->b_out[q] != nullptr && a->b_out[q] != a->b_in[q], "b_out[%d] NULL or aliases b_in", q);
        if (need_d_in) TVDN_REQUIRE(a->d_in[q] != nullptr, "d_in[%d] is NULL", q);
        if (need_d_out) TVDN_REQUIRE(a->d_out[q] != nullptr && a->d_out[q] != a->d_in[q], "d_out[%d] NULL or aliases d_in", q);
        if (need_dprev) {
            TVDN_REQUIRE(a->dprev_in[q] != nullptr, "dprev_in[%d] is NULL", q);
            if (need_d_out) TVDN_REQUIRE(a->d_out[q] != a->dprev_in[q], "d_out[%d] aliases dprev_in", q);
            if (need_b_out) TVDN_REQUIRE(a->b_out[q] != a->dprev_in[q] && a->b_out[q] != a->d_in[q], "b_out[%d] aliases the d state", q);
        }
    }
    return a->dtype == TVDN_F32 ? iterate_fused_impl<float>(ctx, a, sums_out, (hipStream_t)stream)
                                : iterate_fused_impl<double>(ctx, a, sums_out, (hipStream_t)stream);
}

// Many iterations behind one call: the loop of cyTVDN/cyTVDN.py:148-242 for a state that already lives in HBM.  The
// host side of an iteration -- which of the rotating arrays plays which role, tk, the sums slot -- is a few dozen
// instructions here against ~25 us of Python per iteration, which is what bounds small cubes (a 1 M-voxel sweep
// takes ~15 us); the launches are the same tvdn_iterate_fused launches, so the bits are the same.
extern "C" int tvdn_iterate_many(tvdn_ctx *ctx, tvdn_many_args *st, int32_t n_fista, const double *ratios, int32_t n_plain,
                                 double *sums_out, void *stream)
{
    TVDN_REQUIRE(ctx && st && sums_out, "NULL argument");
    TVDN_REQUIRE(n_fista >= 0 && n_plain >= 0, "negative iteration count");
    TVDN_REQUIRE(n_fista == 0 || ratios != nullptr, "ratios is NULL");
    TVDN_REQUIRE(n_fista == 0 || st->d_form, "a FISTA iteration cannot follow an unaccelerated one (nor does it upstream)");
    const int nd = st->base.ndim;
    TVDN_REQUIRE(nd == 3 || nd == 4, "ndim must be 3 or 4, got %d", nd);
    tvdn_iter_args it = st->base;
    tvdn::sums_defer_begin(ctx);  // nobody reads the sums between these iterations: fold them in batches
    for (int i = 0; i < n_fista + n_plain; ++i) {
        const bool use_fista = i < n_fista;
        const double ratio = use_fista ? ratios[i] : 0.0;
        tvdn::roles_bind(*st, use_fista, ratio, it);
        it.sweep_lo = it.sweep_hi = 0;
        it.accumulate = 0;
        const int rc = tvdn_iterate_fused(ctx, &it, sums_out + 3 * (size_t)i, stream);
        if (rc) {
            (void)tvdn::sums_defer_end(ctx, (hipStream_t)stream);
            return rc;
        }
        tvdn::roles_advance(*st, use_fista, ratio);
    }
    return tvdn::sums_defer_end(ctx, (hipStream_t)stream);
}

// The schedule's host side for callers that drive tvdn_iterate_fused themselves (cytvdn_amd/engine.py): the same three
// functions tvdn_iterate_many and tvdn_run use (csrc/tvdn_common.hpp).  Pure host arithmetic: no device needed.
extern "C" int tvdn_fista_ratios(int32_t n, double *out)
{
    TVDN_REQUIRE(n >= 0 && (n == 0 || out != nullptr), "bad argument");
    tvdn::fista_ratios(n, out);
    return TVDN_OK;
}

extern "C" int tvdn_iter_mode(int32_t use_fista, int32_t d_form)
{
    TVDN_REQUIRE(!use_fista || d_form, "a FISTA iteration cannot follow an unaccelerated one (nor does it upstream)");
    return tvdn::iter_mode(use_fista != 0, d_form != 0);
}

extern "C" int tvdn_roles_bind(const tvdn_many_args *st, int32_t use_fista, double ratio, tvdn_iter_args *it)
{
    TVDN_REQUIRE(st && it, "NULL argument");
    TVDN_REQUIRE(it->ndim == 3 || it->ndim == 4, "ndim must be 3 or 4, got %d", it->ndim);
    TVDN_REQUIRE(!use_fista || st->d_form, "a FISTA iteration cannot follow an unaccelerated one (nor does it upstream)");
    tvdn::roles_bind(*st, use_fista != 0, ratio, *it);
    return TVDN_OK;
}

extern "C" int tvdn_roles_advance(tvdn_many_args *st, int32_t use_fista, double ratio)
{
    TVDN_REQUIRE(st != nullptr, "NULL argument");
    TVDN_REQUIRE(!use_fista || st->d_form, "a FISTA iteration cannot follow an unaccelerated one (nor does it upstream)");
    tvdn::roles_advance(*st, use_fista != 0, ratio);
    return TVDN_OK;
}
