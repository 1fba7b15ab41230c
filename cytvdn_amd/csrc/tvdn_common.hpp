// Shared host/device helpers for libtvdn_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/tvdn.h"

namespace tvdn {

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const char *fmt, ...);

#define TVDN_HIP(expr)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ::tvdn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                              __FILE__, __LINE__);                                         \
            return TVDN_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

#define TVDN_REQUIRE(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            ::tvdn::set_error(__VA_ARGS__);    \
            return TVDN_ERR_INVALID;           \
        }                                      \
    } while (0)

// "64G" / "512M" / plain bytes from the environment; 0 = not set (TVDN_HBM_LIMIT, TVDN_HOST_LIMIT, ...)
inline size_t env_bytes(const char *name)
{
    const char *e = getenv(name);
    if (!e) return 0;
    char *end = nullptr;
    double v = strtod(e, &end);
    if (end == e || v <= 0) return 0;
    switch (*end) {
    case 'K': case 'k': v *= 1024.0; break;
    case 'M': case 'm': v *= 1024.0 * 1024.0; break;
    case 'G': case 'g': v *= 1024.0 * 1024.0 * 1024.0; break;
    case 'T': case 't': v *= 1024.0 * 1024.0 * 1024.0 * 1024.0; break;
    default: break;
    }
    return (size_t)v;
}

// ---- canonical 4-D geometry ------------------------------------------------------------------
// A 3-D reference array (N0,N1,N2) is handled as (N0,1,N1,N2): the marching axis M stays the
// reference's axis 0, axis A is absent.  Canonical axes: 0=M, 1=A, 2=B, 3=C (C contiguous).
struct Geom {
    long long n[4];   // extents M, A, B, C
    long long st[4];  // element strides
    int nax;          // 3 or 4 regularised axes
    long long total;
};

inline Geom make_geom(int ndim, const int64_t *shape)
{
    Geom g;
    if (ndim == 4) {
        for (int i = 0; i < 4; ++i) g.n[i] = shape[i];
    } else {
        g.n[0] = shape[0];
        g.n[1] = 1;
        g.n[2] = shape[1];
        g.n[3] = shape[2];
    }
    g.st[3] = 1;
    g.st[2] = g.n[3];
    g.st[1] = g.n[3] * g.n[2];
    g.st[0] = g.n[3] * g.n[2] * g.n[1];
    g.nax = ndim;
    g.total = g.n[0] * g.n[1] * g.n[2] * g.n[3];
    return g;
}

// reference axis -> canonical axis
inline int canon_axis(int ndim, int ax) { return (ndim == 4 || ax == 0) ? ax : ax + 1; }

// ---- the host side of an iteration, defined ONCE ---------------------------------------------------
// Every loop in the tree -- tvdn_iterate_many, tvdn_run (resident and streamed), and through the exported
// tvdn_fista_ratios / tvdn_roles_bind / tvdn_roles_advance the Python loop of engine.SlabRunner -- takes its schedule from here.

// (tk-1)/tk_new of iterations 0..n-1: the float64 recurrence of cyTVDN.py:153-156, tk starting at 1.
inline void fista_ratios(int n, double *out)
{
    double tk = 1.0;
    for (int i = 0; i < n; ++i) {
        const double tk_new = (1.0 + std::sqrt(1.0 + 4.0 * (tk * tk))) / 2.0;
        out[i] = (tk - 1.0) / tk_new;
        tk = tk_new;
    }
}

// Which sweep form an iteration takes: FISTA iterations keep the d-form; the first unaccelerated iteration after
// them converts it to b (hybrid schedule, cyTVDN.py:99-108); later ones ping-pong b.
inline int iter_mode(bool use_fista, bool d_form)
{
    return use_fista ? TVDN_ITER_FISTA_D : (d_form ? TVDN_ITER_FISTA_D_TO_PLAIN : TVDN_ITER_PLAIN);
}

// Point `it` at the arrays of the iteration about to run (compact state: three rotating arrays per axis).
inline void roles_bind(const tvdn_many_args &st, bool use_fista, double ratio, tvdn_iter_args &it)
{
    it.recon_in = st.recon[st.cur];
    it.recon_out = st.recon[st.cur ^ 1];
    it.tk = use_fista ? ratio : 0.0;
    it.tk_prev = st.tk_prev;
    for (int q = 0; q < 4; ++q) {
        it.b_in[q] = it.d_in[q] = it.dprev_in[q] = nullptr;
        it.b_out[q] = it.d_out[q] = nullptr;
        if (q >= it.ndim) continue;
        if (use_fista) {
            it.d_in[q] = st.S[q][st.i_d]; it.dprev_in[q] = st.S[q][st.i_prev]; it.d_out[q] = st.S[q][st.i_out];
        } else if (st.d_form) {
            it.d_in[q] = st.S[q][st.i_d]; it.dprev_in[q] = st.S[q][st.i_prev]; it.b_out[q] = st.S[q][st.i_out];
        } else {
            it.b_in[q] = st.S[q][st.i_b]; it.b_out[q] = st.S[q][st.i_bout];
        }
    }
    it.mode = iter_mode(use_fista, st.d_form != 0);
}

// Make the freshly written arrays current: the role rotation after an iteration of the form roles_bind chose.
inline void roles_advance(tvdn_many_args &st, bool use_fista, double ratio)
{
    st.cur ^= 1;
    if (use_fista) {
        const int t = st.i_prev; st.i_prev = st.i_d; st.i_d = st.i_out; st.i_out = t;
        st.tk_prev = ratio;
    } else if (st.d_form) {
        st.i_b = st.i_out; st.i_bout = st.i_prev; st.d_form = 0;  // b now lives where d_k+1 would have gone
    } else {
        const int t = st.i_b; st.i_b = st.i_bout; st.i_bout = t;
    }
}

inline void roles_reset(tvdn_many_args &st, bool d_form)
{
    st.cur = 0;
    st.i_d = 0; st.i_prev = 1; st.i_out = 2;
    st.i_b = 0; st.i_bout = 1;
    st.d_form = d_form ? 1 : 0;
    st.tk_prev = 0.0;
}

// ---- context ---------------------------------------------------------------------------------
// Per-workgroup partial sums of one launch: `partial_cap` rows of kPartialWidth doubles.  The buffer starts at
// kInitPartialBlocks rows and grows on demand (ensure_partials) up to kMaxPartialBlocks, so that even a
// 1024x256x256 plane keeps its short 8-row marches (2^16 tiles x 17 chunks = 1.1 M workgroups).
constexpr int kInitPartialBlocks = 1 << 18;
constexpr int kMaxPartialBlocks = 1 << 22;  // per launch
constexpr int kPartialWidth = 4;            // doubles per block
constexpr int kFoldSegment = 1024;          // partial rows folded by one workgroup of the first finalize stage
constexpr int kFoldDirect = 4096;           // up to this many rows the single-workgroup stage folds them itself

}  // namespace tvdn

struct tvdn_ctx {
    int device;
    double *partials;   // [partial_cap][kPartialWidth]
    double *partials2;  // [kMaxPartialBlocks / kFoldSegment][kPartialWidth]: output of the first finalize stage
    long long partial_cap;
    // Deferred finalize (small grids inside the library's own loops, tvdn_capi.hip sums_defer_*): the sweeps of up to
    // kDeferSlots iterations park their partial rows in `ring` and ONE launch folds them all
    double *ring;  // [kDeferSlots][kFoldDirect][kPartialWidth], allocated on first use
    bool deferring;
    int n_pend;
    int pend_rows[32];
    double *pend_out[32];
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // pending (start, stop) pairs
    // Where the NEXT finalize also leaves its sums (device-visible host memory, nullptr: nowhere): a resident tvdn_run with a
    // stopping rule looks at every iteration's sums while the next iteration already runs (tvdn_run.hip) -- no copy in the stream
    double *mirror;
};

namespace tvdn {

// ---- device-side reduction helpers -----------------------------------------------------------
// Fixed-shape tree: lane shuffles (64-wide wavefront) -> LDS across the block's waves ->
// one partial per block.  A second single-block kernel folds the partials in index order.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int NV, int BLOCK>
__device__ __forceinline__ void block_store_partials(const double (&v)[NV], double *partials)
{
    constexpr int NW = BLOCK / 64;
    __shared__ double red[NV][NW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double s = wave_sum(v[q]);
        if (lane == 0) red[q][w] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += red[threadIdx.x][i];
        partials[(size_t)blockIdx.x * kPartialWidth + threadIdx.x] = s;
    }
}

// Deferred finalize.  A sweep over a small cube is followed by a finalize launch that costs 3.5-5 us of a 15-90 us iteration
// (64x64x256: 18.0 -> 14.6 us without it) although nothing in the NEXT iteration depends on it.  Inside the library's own
// loops (tvdn_iterate_many, the resident tvdn_run without a stopping rule) the sweeps therefore park their partial rows in a
// ring of kDeferSlots slots and one launch folds a whole batch -- the same tree per iteration, so the same bits.
//   sums_defer_begin   from now on launches of <= kFoldDirect workgroups with accumulate == 0 are parked
//   sums_defer_slot    where the next such launch writes its rows (nullptr: not deferring / not eligible)
//   sums_defer_push    the launch has been queued: remember (rows, out); folds the batch when the ring is full
//   sums_defer_flush   fold what is parked (also called before any launch that is finalized at once, to keep the order)
//   sums_defer_end     flush and leave the mode
constexpr int kDeferSlots = 32;
void sums_defer_begin(tvdn_ctx *ctx);
double *sums_defer_slot(tvdn_ctx *ctx, long long nblocks, bool accumulate);
int sums_defer_push(tvdn_ctx *ctx, int nblocks, double *out, hipStream_t s);
int sums_defer_flush(tvdn_ctx *ctx, hipStream_t s);
int sums_defer_end(tvdn_ctx *ctx, hipStream_t s);

// Makes room for `nblocks` partial rows (grows the scratch buffer; a growth synchronises the device once).
int ensure_partials(tvdn_ctx *ctx, long long nblocks);
// tvdn_stream.hip: the out-of-core branch of tvdn_run
int choose_stream_shape(int nd, int64_t n_rows, size_t row_bytes, size_t free_bytes, bool mse, bool wrap, int n_state, bool may_keep,
                        int64_t k_cap, int64_t *rows_out, int64_t *k_out, int64_t *res_out, int64_t n_iters = 0);
// One slab of a streamed DEVICE-LIST run (tvdn_stream.hip run_streamed_slabs): the cube's state lives in page-locked host
// arrays shared by all slabs of the process -- two sets, a pass reads one and writes the other, so a slab may read its
// neighbours' rows (k of them beyond each interior face: the trapezoid of a temporally blocked pass) while they write theirs.
struct SlabShare;
int run_streamed(const tvdn_run_args *a, int64_t rows, int64_t k, int64_t resident_rows, const SlabShare *slab = nullptr);
int run_streamed_slabs(const tvdn_run_args *a, int64_t rows, int64_t k);
// tvdn_stream.hip: the caller's result array page-locked in place (see there); TVDN_OK, or a status with which the caller keeps
// to the pinned lanes
int host_pin_result(void *user, size_t bytes);
void host_unpin_result(void *user);
int run_streamed_rank(const tvdn_run_args *a, int64_t rows, int64_t k);  // one slab of a multi-process run (tvdn_slab_io)
// tvdn_run.hip: the big device block of a run is KEPT when the run ends (one per device) and handed to the next run it fits
// (releasing and re-allocating tens of GiB in quick succession costs seconds); tvdn_release_cache() returns it
hipError_t state_acquire(void **p, size_t bytes, size_t *got_bytes, int device, bool *reused, bool any_larger, double spread_budget_s = 0.25);
void state_release(void *p, size_t bytes, int device);
size_t state_kept_bytes(int device);  // counts as free: the next run takes it over or releases it
// tvdn_run_state.hip: what a resident run needs beside its state -- reduction context, its two streams, the sums' device buffer -- is
// kept between runs too (one set per device): creating and destroying them is 1.5-2 ms of hipMalloc / hipFree / stream calls per run,
// more than the iterations of a short run on a small cube.  tvdn_release_cache() destroys them.
struct RunKit {
    tvdn_ctx *ctx = nullptr;
    hipStream_t main = nullptr, copy = nullptr;
    int main_level = 0, copy_level = 0;  // make_stream levels the streams were made with
    void *sums = nullptr;
    size_t sums_bytes = 0;
};
bool kit_acquire(int device, int main_level, int copy_level, RunKit *k);  // true: a kept set with streams of these levels, now the caller's
void kit_release(int device, RunKit &k);                                  // kept for the next run (idle, reset) or destroyed
// the state's allocation itself (granules unless told otherwise: tvdn_devmem.hip), and the shape of a resident run's pipelined transfers
hipError_t state_malloc(void **p, size_t bytes, int device, bool granules = true, double spread_budget_s = 0.25, const int *peers = nullptr, int n_peers = 0);
void pipeline_plan(int64_t n0, int64_t n_total, int64_t cube_bytes, int32_t out[3]);
// tvdn_run_state.hip: before a device list runs on granule blocks that neighbours on other devices read -- does a peer copy out of
// every such block arrive intact?  (one entry per slab: its block, device, local rows, own rows, halo rows held, copy stream)
struct PeerSlab {
    void *state;
    int device;
    int64_t rows, row_lo, row_hi;
    bool halo_lo, halo_hi;
    hipStream_t copy;
};
bool slab_peer_copy_check(const PeerSlab *sl, int world, size_t row_bytes, int nd, int per_axis, char *why, size_t why_len);
// tvdn_devmem.hip: device memory composed from physical granules (big blocks) or plain hipMalloc; dev_free takes either
struct DevAllocInfo {  // what a block of granules was made from (all zero for a plain block)
    int64_t granule_bytes;
    int32_t granules, pool;  // granules mapped / created to choose them from
    double seconds;
};
hipError_t dev_alloc(void **p, size_t bytes, int device, int *kind, double spread_budget_s = 0.25, DevAllocInfo *info = nullptr, const int *peers = nullptr,
                     int n_peers = 0);
hipError_t dev_free(void *p);  // the first error of the release, if any (everything is attempted whatever fails; stderr says which call)
// a block on granules, re-dealt at another size (contents undefined); hipErrorNotSupported: free and allocate
hipError_t dev_resize(void **p, size_t bytes, int device, double spread_budget_s = 0.25);
// a kept granule block chosen from a short pool, topped up and re-drawn when a run can afford `spread_budget_s` (contents undefined, new address)
hipError_t dev_upgrade(void **p, int device, double spread_budget_s);
int dev_kind(const void *p);
// Entry points that select a device put the calling thread's current device back before they return: a library that leaves
// hipSetDevice(3) behind changes where the caller's next allocation (torch's, say) lands.
struct DeviceRestore {
    int prev = -1;
    DeviceRestore()
    {
        if (hipGetDevice(&prev) != hipSuccess) {
            (void)hipGetLastError();
            prev = -1;
        }
    }
    ~DeviceRestore()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceRestore(const DeviceRestore &) = delete;
    DeviceRestore &operator=(const DeviceRestore &) = delete;
};
// tvdn_rebuild.hip: recon of ring rows [row0, row1) rebuilt from the accumulator state (the streamed engine's level 0)
struct RebuildArgs {
    int dtype, ndim;
    int64_t plane_shape[3];  // shape[1:], ndim - 1 entries
    const void *orig;        // ring of orig_ring_rows planes
    void *recon;             // ring of ring_rows planes, as in1 / in2
    const void *in1[4];      // b (b-form) or d_k-1 (d-form), per reference axis
    const void *in2[4];      // d_k (d-form)
    const void *next1, *next2;  // optional: axis 0's in1 / in2 of row `row1` as planes of their own (a row that is not in the rings yet)
    double lambda_mu[4], tk_prev;
    int d_form;
    int64_t row0, row1, top; // rows rebuilt; the first row beyond the block (a Jia-Zhao top face: its axis-0 accumulator is zero)
    int64_t ring_rows, orig_ring_rows;
};
int recon_rebuild(const RebuildArgs &a, hipStream_t s);
bool io_is_warm(int device);
void io_warm(int device);  // tvdn_hostio.hip: the pinned staging lanes of `device` set up now instead of inside the first transfer
void io_cap_lanes(int n);  // tvdn_hostio.hip: n > 0 holds a cap of n staging lanes per transfer, 0 drops that hold (counted)
// A non-blocking stream in a hardware-queue class of its own.  The runtime multiplexes streams onto a few hardware queues
// per PRIORITY level; two streams that share one execute in submission order, so a transfer's completion marker can sit
// behind every sweep already queued (tvdn_run's last iterations over the download: the first chunk's copy "took" 90 ms,
// exactly until the 72 queued sweeps had drained, 12 ms for every later one; profiles/r03_e2e_pipelined.txt).  Sweeps go
// on a HIGH-priority stream, downloads of the streamed engine on a LOW-priority one, everything else stays normal:
// different pools, no false ordering.  level +1 / 0 / -1; TVDN_STREAM_PRIO=0 makes every level normal (measurement).
int make_stream(hipStream_t *s, int level);
int launch_finalize(tvdn_ctx *ctx, int nblocks, int nv, double *out, hipStream_t s, bool accumulate = false);

// ---- 16-byte packs ---------------------------------------------------------------------------
// A thread owns VEC consecutive elements of the contiguous axis (16 bytes: 4 floats / 2 doubles), or one
// element when the extent or the alignment forbids it.
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
    T v[VEC];
};

template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> ldv(const T *p)
{
    return *reinterpret_cast<const Pack<T, VEC> *>(p);
}

#ifndef TVDN_NT_STORES
#define TVDN_NT_STORES 1
#endif
#ifndef TVDN_NT_LOADS
#define TVDN_NT_LOADS 1
#endif
constexpr bool kNtStores = TVDN_NT_STORES != 0;
constexpr bool kNtLoads = TVDN_NT_LOADS != 0;

// Outputs are written once and not read again by the same sweep: non-temporal (streaming) stores keep them
// from displacing the input lines that neighbouring threads are about to re-read from L2.  Measured on
// config 2, interleaved runs on one device: plain 11.77 ms, nt stores 11.66 ms, nt stores + nt loads of
// the own-position-only arrays 11.44 ms (a pure 10R/5W float4 stream: 11.03 ms plain, 10.76 ms nt).
template <typename T, int VEC>
__device__ __forceinline__ void stv(T *p, const Pack<T, VEC> &x)
{
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    vec_t v;
#pragma unroll
    for (int j = 0; j < VEC; ++j) v[j] = x.v[j];
    if (kNtStores)
        __builtin_nontemporal_store(v, reinterpret_cast<vec_t *>(p));
    else
        *reinterpret_cast<vec_t *>(p) = v;
}

// Arrays read at the thread's own position only: streaming loads.
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> ldv_nt(const T *p)
{
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    Pack<T, VEC> x;
    const vec_t v = kNtLoads ? __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(p)) : *reinterpret_cast<const vec_t *>(p);
#pragma unroll
    for (int j = 0; j < VEC; ++j) x.v[j] = v[j];
    return x;
}

// two ternaries as in the reference's generated C (anisotropic.c:2423-2437): NaN passes through
template <typename T>
__device__ __forceinline__ T clipv(T a, T val)
{
    const T lo = -val;
    const T t = (lo > a) ? lo : a;
    return (val < t) ? val : t;
}

inline bool aligned16(const void *q) { return (reinterpret_cast<uintptr_t>(q) & 15u) == 0; }

// XCD-aware remap of workgroup ids (bijective for any grid size): blocks b and b+8 share an XCD and its L2,
// so each XCD gets a contiguous run of logical ids.
__device__ __forceinline__ long long xcd_remap(long long bid, long long G)
{
    const long long q8 = G / 8, r8 = G % 8, xcd = bid % 8;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bid / 8;
}

}  // namespace tvdn
