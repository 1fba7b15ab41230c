/*
 * tvdn.h -- C ABI of the MI355X-native anisotropic TV denoising hot path (libtvdn_hip.so).
 *
 * The reference (cyTVDN) has no FFI header: its boundary is "Python callables taking NumPy
 * arrays" (cyTVDN/__init__.py:1).  Each entry point below names the reference callable whose
 * arithmetic it replaces; cytvdn_amd/ binds them with ctypes and presents the reference's
 * Python names and signatures on top (INTEGRATION.md shows the binding).
 *
 * Conventions
 *  - All array arguments are DEVICE pointers to C-contiguous blocks of `dtype`
 *    (TVDN_F32 / TVDN_F64); `shape` is a HOST array of `ndim` (3 or 4) extents in the
 *    reference's axis order.  Scalars that the reference converts to the array dtype
 *    (clip, tk, lambda_mu) are passed as double and rounded to `dtype` inside, which is
 *    what the Cython fused-type call does.
 *  - Reductions (the Python floats the reference kernels return) are written to DEVICE
 *    double slots, accumulated in f64 by a fixed tree (deterministic run to run).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are
 *    asynchronous with respect to the host; nothing here synchronises.
 *  - Every function returns TVDN_OK (0) or a negative tvdn_status; tvdn_last_error()
 *    describes the last failure of the calling thread.  Nothing falls back to the CPU.
 */
#ifndef TVDN_H
#define TVDN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVDN_ABI_VERSION 9

typedef enum tvdn_dtype { TVDN_F32 = 0, TVDN_F64 = 1 } tvdn_dtype;

typedef enum tvdn_status {
    TVDN_OK = 0,
    TVDN_ERR_INVALID = -1,     /* bad argument (ndim, ax, bc_mode, shape, NULL pointer, ...)   */
    TVDN_ERR_UNSUPPORTED = -2, /* defined upstream as undefined behaviour (mirror-BC recon)     */
    TVDN_ERR_HIP = -3,         /* a HIP runtime call failed; see tvdn_last_error()               */
    TVDN_ERR_NO_DEVICE = -4    /* no gfx950 device visible                                       */
} tvdn_status;

/* Boundary-condition modes, as the reference numbers them (anisotropic.pyx:20-23). */
#define TVDN_BC_PERIODIC 0
#define TVDN_BC_MIRROR 1
#define TVDN_BC_JIA_ZHAO 2

int tvdn_abi_version(void);
const char *tvdn_last_error(void);
/* Number of visible HIP devices, or a negative status.  Does not create a context. */
int tvdn_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Context: per-device scratch for the two-stage reductions (a few hundred KiB).
 * One context may be used from one stream at a time.
 * ---------------------------------------------------------------------------------------- */
typedef struct tvdn_ctx tvdn_ctx;
int tvdn_ctx_create(tvdn_ctx **out, int device);
int tvdn_ctx_destroy(tvdn_ctx *ctx);

/* Measurement aid (bench.py): while enabled, every tvdn_iterate_fused call brackets its sweep
 * kernel (not the small reduction epilogue) with a pair of HIP events recorded on the call's
 * stream.  tvdn_ctx_timing_read synchronises those events, adds their elapsed times to
 * *total_ms / *launches, and forgets them. */
int tvdn_ctx_timing_enable(tvdn_ctx *ctx, int on);
int tvdn_ctx_timing_read(tvdn_ctx *ctx, double *total_ms, int64_t *launches);
/* The same, launch by launch in issue order: the first min(cap, *launches) durations go to ms_out (bench.py forms the
 * per-step minimum / median / maximum from them). */
int tvdn_ctx_timing_read_each(tvdn_ctx *ctx, double *ms_out, int64_t cap, int64_t *launches);

/* ------------------------------------------------------------------------------------------
 * Kernel-level entry points: one reference pass each, in place, any shape.
 * ---------------------------------------------------------------------------------------- */

/* accumulator_update_{3D,4D}        (cyTVDN/anisotropic.pyx:169-237, :17-84)     when d == NULL
 * accumulator_update_{3D,4D}_FISTA  (cyTVDN/anisotropic.pyx:243-317, :89-164)    when d != NULL
 *   b <- clip((a - a[prev along ax]) + b)  [FISTA: d_new = that; b <- d_new + tk*(d_new - d); d <- d_new]
 *   *norm_out (device double) <- sum |b_new|   (the reference's return value)
 * bc_mode 0, 1 or 2 (mirror needs shape[ax] >= 2). */
int tvdn_accumulator_update(tvdn_ctx *ctx, int dtype, int ndim, const int64_t *shape,
                            const void *a, void *b, void *d, double tk, int ax, double clip,
                            int bc_mode, double *norm_out, void *stream);

/* datacube_update_{3D,4D}  (cyTVDN/utils.pyx:131-199, :54-125), bc_mode 0 or 2.
 *   recon <- orig - sum_ax lambda_mu[ax] * (b_ax - b_ax[next along ax, periodic wrap])
 *   sums_out[0] <- sum |recon_new - recon_old|, sums_out[1] <- sum |recon_old|  (device doubles;
 *   the reference returns sums_out[0] / sums_out[1]).
 * `b` is a HOST array of ndim device pointers; `lambda_mu` a HOST array of ndim doubles.
 * bc_mode 1 returns TVDN_ERR_UNSUPPORTED: upstream it indexes out of bounds (utils.pyx:117-120). */
int tvdn_datacube_update(tvdn_ctx *ctx, int dtype, int ndim, const int64_t *shape, const void *orig,
                         void *recon, const void *const *b, const double *lambda_mu, int bc_mode,
                         double *sums_out, void *stream);

/* ABI 7.  datacube_update_{3D,4D} (cyTVDN/utils.pyx:54-125, :131-199, Jia-Zhao / periodic wrap inside a plane) on the COMPACT
 * state: recon <- orig - sum_ax lambda_mu[ax] * (b_ax - b_ax[next along ax]) with b_ax = d_ax + tk_prev (d_ax - dprev_ax) when
 * `dprev` is given (the expression that formed b, anisotropic.pyx:128), else b_ax = d_ax as they are.  Bit-identical to the
 * recon the iteration that left this state behind wrote -- which is why a streamed run does not carry recon across PCIe between
 * its passes (csrc/tvdn_rebuild.hip).  Rows [row0, row1) of a block of shape[0] rows; past the last row the axis-0 accumulator
 * reads as zero (the Jia-Zhao wrap onto a finite first row).  `d` / `dprev`: HOST arrays of ndim device pointers. */
int tvdn_recon_from_state(int dtype, int ndim, const int64_t *shape, const void *orig, void *recon, const void *const *d,
                          const void *const *dprev, const double *lambda_mu, double tk_prev, int64_t row0, int64_t row1, void *stream);

/* sum_square_error_{3D,4D}  (cyTVDN/utils.pyx:35-49, :14-30):  *out <- sum (a-b)^2. */
int tvdn_sum_square_error(tvdn_ctx *ctx, int dtype, int ndim, const int64_t *shape, const void *a,
                          const void *b, double *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Loop-level entry point: ONE full iteration (all accumulator updates + the reconstruction
 * update, cyTVDN/cyTVDN.py:153-184 / :205-230 / :372-392 / :405-420) as a single fused sweep
 * that reads every state array once and writes it once.
 *
 * State is multi-buffered: the sweep reads *_in and writes *_out (no array is updated in place, so
 * neighbouring tiles may recompute each other's halo values); the caller rotates the roles after
 * each call.  Four state representations (`mode`):
 *   TVDN_ITER_PLAIN            b_in -> b_out                       unaccelerated iteration
 *   TVDN_ITER_FISTA            (b_in, d_in) -> (b_out, d_out)      the reference's own (b, d) state
 *   TVDN_ITER_FISTA_D          (d_in = d_k, dprev_in = d_k-1) -> d_out = d_k+1
 *                              b_k is not stored: it is rebuilt as d_k + tk_prev*(d_k - d_k-1), the
 *                              expression that formed it (anisotropic.pyx:128) -- same bits, 15 array
 *                              passes per 4-D iteration instead of 19.  The first call passes zeros.
 *   TVDN_ITER_FISTA_D_TO_PLAIN (d_in, dprev_in) -> b_out           first unaccelerated iteration after
 *                              FISTA_D iterations (hybrid schedule, cyTVDN.py:99-108)
 *
 * Slab decomposition along axis 0 (one GPU per slab): the local block has `shape[0]` rows of
 * which rows [row_lo, row_hi) are this slab's own; a row below row_lo / at row_hi is a halo
 * row holding the neighbour's current recon (exchanged by the caller between iterations).
 *   lo_mode  TVDN_EDGE_BC    row_lo is the global first row: apply bc_mode there
 *            TVDN_EDGE_HALO  row row_lo-1 is a halo row
 *   hi_mode  TVDN_EDGE_BC    row_hi-1 is the global last row and row 0 of this block is the
 *                            global first row (single-slab case): wrap onto it
 *            TVDN_EDGE_HALO  row row_hi is a halo row; its axis-0 accumulator is kept locally
 *            TVDN_EDGE_ZERO  row_hi-1 is the global last row of a multi-slab Jia-Zhao run: the
 *                            wrapped axis-0 accumulator is identically zero (true while the cube's
 *                            first row is finite)
 *            TVDN_EDGE_WRAP  the same place, exact for any data: `wrap_recon` (or, when NULL, row row_hi
 *                            of recon_in) holds the CURRENT recon of global row 0, from which the wrapped
 *                            accumulator is formed as upstream forms it -- zero, or NaN where row 0 is
 *                            Inf/NaN (anisotropic.pyx:65-73: clip((a[0]-a[0]) + b[0]))
 * sums_out (device, 3 doubles): [0] sum over all axes of |b_new| (b_norm), [1] sum |recon_new -
 * recon_old|, [2] sum |recon_old|, over the slab's own rows only.
 * ---------------------------------------------------------------------------------------- */
#define TVDN_EDGE_BC 0
#define TVDN_EDGE_HALO 1
#define TVDN_EDGE_ZERO 2
#define TVDN_EDGE_WRAP 3

#define TVDN_ITER_PLAIN 0
#define TVDN_ITER_FISTA 1
#define TVDN_ITER_FISTA_D 2
#define TVDN_ITER_FISTA_D_TO_PLAIN 3

#define TVDN_SWEEP_CHAIN_LO 1
#define TVDN_SWEEP_STORE_AHEAD 2

typedef struct tvdn_iter_args {
    int32_t dtype;        /* tvdn_dtype                                            */
    int32_t ndim;         /* 3 or 4                                                */
    int64_t shape[4];     /* local block extents, first ndim entries used          */
    int64_t row_lo;       /* own rows [row_lo, row_hi) along axis 0                */
    int64_t row_hi;
    int32_t lo_mode;      /* TVDN_EDGE_*                                           */
    int32_t hi_mode;
    int32_t bc_mode;      /* 0 or 2                                                */
    int32_t mode;         /* TVDN_ITER_*                                           */
    double tk;            /* FISTA momentum ratio (tk-1)/tk_new of this iteration  */
    double tk_prev;       /* ... of the previous iteration (TVDN_ITER_FISTA_D*)    */
    double clip[4];       /* 1/lambda per axis                                     */
    double lambda_mu[4];  /* lambda/mu per axis                                    */
    const void *orig;     /* noisy input, read only                                */
    const void *recon_in;
    void *recon_out;
    const void *b_in[4];
    void *b_out[4];
    const void *d_in[4];
    void *d_out[4];
    const void *dprev_in[4]; /* d of the iteration before d_in (TVDN_ITER_FISTA_D*) */
    /* Partial sweep, for overlapping the halo exchange with the bulk of the work: advance only rows
     * [sweep_lo, sweep_hi) of the own rows (0,0 = all own rows).  Rows outside the sweep but inside
     * [row_lo,row_hi) are treated like any in-slab neighbour (read, never written).  With
     * `accumulate` != 0 the three sums are ADDED to sums_out instead of overwriting it. */
    int64_t sweep_lo;
    int64_t sweep_hi;
    int32_t accumulate;
    /* ABI 9 (the field was `reserved`, always 0).  Consecutive partial sweeps of ONE iteration over adjacent rows -- the launches
     * of a level of the streamed engine, one row each when nothing is kept in HBM -- hand the axis-0 accumulator across the cut
     * instead of forming it twice.  TVDN_SWEEP_STORE_AHEAD: the launch also stores the axis-0 output state of row sweep_hi (its
     * look-ahead computes it anyway; needs sweep_hi < row_hi and room for that row in the output ring).  TVDN_SWEEP_CHAIN_LO: the
     * launch takes the axis-0 accumulator of row sweep_lo from that stored output (b' itself, or d' and this level's d_in: the
     * very expression that formed it, anisotropic.pyx:128) instead of re-reading recon of row sweep_lo - 1 and the input state,
     * and does not store it again (needs sweep_lo > row_lo, and the previous launch of this iteration to have ended at sweep_lo
     * with STORE_AHEAD).  One plane less to read per launch: 19 -> 18 plane moves for a one-row launch of a 4-D FISTA level,
     * same bits.  Row rings only (ring_rows > 0); ignored on contiguous arrays. */
    int32_t chain;
    const void *wrap_recon; /* TVDN_EDGE_WRAP: one plane, current recon of global row 0 (NULL: row row_hi) */
    /* Row rings (ABI 3; 0 = rows are contiguous, the normal case).  With ring_rows > 0 every recon / accumulator
     * pointer is the base of a ring buffer of ring_rows row-planes in which row m of the block lives at slot
     * m % ring_rows; orig_ring_rows is the same for `orig`.  shape[0], row_lo/row_hi and the sweep rows keep their
     * meaning (row indices of a virtual array that is only partly resident): the caller keeps every row the sweep
     * touches -- [sweep_lo-1, sweep_hi] and, where an edge mode wraps, its target -- in the ring.  This is how the
     * out-of-core wavefront schedule keeps R+2 rows per iteration level without ever moving them (the host-side
     * block loop it replaces upstream: cyTVDN/cyTVDN.py:148-242 on a cube that does not fit). */
    int64_t ring_rows;
    int64_t orig_ring_rows;
    /* ABI 8.  Ring sizes of their own, 0 = ring_rows: recon_in; b_in and d_in (the state of this level); dprev_in (the state of
     * the level before); recon_out; b_out and d_out.  A "ring" longer than the cube is an array: the first levels of a streamed
     * pass read rows that are kept in HBM between passes where they are kept, and the last ones write them there, instead of
     * copying them into and out of the levels' rings (csrc/tvdn_stream_chain.hip).  What is read may also be a ring of ONE row:
     * every row is then that plane (the zeros a run's state starts from). */
    int64_t recon_in_ring_rows;
    int64_t cur_ring_rows;
    int64_t prev_ring_rows;
    int64_t recon_out_ring_rows;
    int64_t out_ring_rows;
} tvdn_iter_args;

int tvdn_iterate_fused(tvdn_ctx *ctx, const tvdn_iter_args *args, double *sums_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Many iterations on a state that already lives in HBM (compact d-rotation form): the loop of
 * cyTVDN/cyTVDN.py:148-242 without a host language round trip per iteration.  `base` carries what stays
 * fixed (dtype, shape, own rows, edge modes, bc_mode, clip, lambda_mu, orig, wrap_recon; its state
 * pointers, mode, tk and sweep fields are ignored); recon[] and S[axis][0..2] are the arrays the
 * iterations rotate through, and cur / i_* / d_form / tk_prev say which plays which role -- read on
 * entry, updated on return, so calls can be chained.  n_fista iterations with momentum ratios[i]
 * ((tk-1)/tk_new of cyTVDN.py:153-156, float64 on the host), then n_plain unaccelerated ones; iteration i
 * writes its three sums to sums_out[3*i .. 3*i+2] (device).  Asynchronous like tvdn_iterate_fused.
 * ---------------------------------------------------------------------------------------- */
typedef struct tvdn_many_args {
    tvdn_iter_args base;
    void *recon[2];      /* recon[cur] is current                                              */
    void *S[4][3];       /* per axis: d_k / d_k-1 / next (d-form) or b / next (b-form)          */
    int32_t cur;
    int32_t i_d, i_prev, i_out; /* roles inside S[axis][] while the state is in d-form          */
    int32_t i_b, i_bout;        /* ... in b-form                                                */
    int32_t d_form;             /* 1: (d_k, d_k-1) pairs; 0: b                                  */
    double tk_prev;             /* momentum ratio of the last FISTA iteration run               */
} tvdn_many_args;

int tvdn_iterate_many(tvdn_ctx *ctx, tvdn_many_args *state, int32_t n_fista, const double *ratios,
                      int32_t n_plain, double *sums_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Whole-loop entry point on HOST arrays: what denoise4D / denoise3D do between their argument
 * checks and their return (cyTVDN/cyTVDN.py:122-247, :345-435), for callers that are not Python.
 * Copies `data` to HBM once (pinned multi-lane staging, tvdn_copy_to_device), runs n_fista FISTA
 * iterations then n_plain unaccelerated ones (float64 tk recurrence of cyTVDN.py:153-156; compact
 * d-rotation state), copies recon back once.
 *   n_devices 0: everything on `device`.  n >= 1: axis 0 is cut into n slabs of (almost) equal height, slab i
 *             resident on devices[i] (entries may repeat: several slabs on one GPU); each iteration sweeps the
 *             edge rows of every slab first, then moves one recon row per neighbour device-to-device (peer
 *             copies over xGMI, on a copy stream per slab) while the interior rows are swept -- the
 *             single-process form of the slab decomposition that replaces cyTVDN/mpi.py:314-434 (the
 *             multi-process form over RCCL is cytvdn_amd.distributed.denoise_slabs).  Each slab's state must fit
 *             its device's HBM (90 % of what is free, tvdn_plan); what does not fit is streamed when stream_rows /
 *             stream_k ask for it (below): a cube beyond the HBM of ONE device through that device, and with a device
 *             list every slab through its own device from page-locked host arrays all slabs share (BASELINE configs[4]
 *             in structure, inside one process).
 *   data / recon_out may be the same array or overlap (denoise in place): the resident run uploads `data` before it
 *             writes anything; the streamed run then keeps the data term in a pinned copy of its own (one more cube
 *             of host memory, counted by tvdn_stream_host_need).
 *   sums_out  host, (n_fista+n_plain) x 3 doubles: sum|b_new|, sum|recon_new-recon_old|, sum|recon_old|
 *             per iteration over the WHOLE cube (the reference's b_norm[i] and delta_recon[i] = [1]/[2]);
 *             rows of iterations that did not run stay zero
 *   mse_out   host, n_fista+n_plain+1 doubles, or NULL; needs `reference` (sum of squared errors)
 *   use_stop  when non-zero a phase ends as soon as delta_recon (formed in the data dtype, as upstream, from
 *             the sums over all slabs) drops below `stop`; a FISTA-phase stop still falls through to the
 *             unaccelerated phase (cyTVDN.py:189-195).  A resident run looks at the sums of iteration i while
 *             iteration i+1 already runs and takes that one back if the rule was met (same bits as reading them
 *             in stream order, 7 us instead of 27 us per iteration: profiles/r06_stop_rule.jsonl); TVDN_STOP_LAG=0
 *             keeps the blocking form
 *   iters_run host, optional: number of iterations executed
 * ---------------------------------------------------------------------------------------- */
#define TVDN_MAX_DEVICES 16

/* ABI 6.  Where a tvdn_run call's time and bytes went; filled on success when tvdn_run_args.stats points at one.
 * Measurement and planning aid (bench.py reports the PCIe rates of a streamed run from it); the
 * reference has no counterpart (its loop prints nothing but progress bars, cyTVDN/cyTVDN.py:148-242). */
#define TVDN_ENGINE_RESIDENT 0
#define TVDN_ENGINE_STREAMED 1

typedef struct tvdn_run_stats {
    int32_t engine;        /* TVDN_ENGINE_*                                                                        */
    int32_t pipelined;     /* resident: 1 when the transfers ran under the first / over the last iterations        */
    int32_t stream_rows;   /* streamed: rows per chunk                                                              */
    int32_t stream_k;      /* streamed: iteration levels per PCIe round trip                                        */
    int64_t resident_rows; /* streamed: low rows of axis 0 whose state stayed in HBM between passes (0 = none)      */
    int64_t n_passes;      /* streamed: passes over the cube                                                        */
    int64_t h2d_bytes;     /* bytes that crossed PCIe towards the device during the call                            */
    int64_t d2h_bytes;     /* ... and back                                                                          */
    double setup_s;        /* entry -> first iteration queued: allocations, page-locking, placement audition        */
    double loop_s;         /* the iterations, with the transfers that run under them (a streamed run: its passes)   */
    double total_s;        /* entry -> return                                                                       */
    int32_t audition_n;    /* resident: placements of the state tried (0 / 1: the first allocation was taken)       */
    int32_t audition_kept; /* index of the one kept                                                                 */
    double audition_ms[8]; /* probe time per sweep of each candidate, in the order tried                            */
    double first_pass_s;   /* streamed: the first pass alone -- the one the page-locking of the host state runs under
                              (0 when the passes were chained: they overlap and cannot be told apart)                 */
    int32_t first_pass_iters; /* ... and the iterations it held                                                     */
    int32_t results_under_last_pass; /* streamed with resident rows: 1 when their results crossed PCIe during the last pass
                              (result array page-locked in place, no stopping rule) instead of in one piece after it   */
    int32_t state_mem;     /* ABI 7.  TVDN_MEM_*: what the run's big device block (the state, or the rings and kept rows
                              of a streamed run) is made of; TVDN_MEM_CALLER for a caller's workspace                  */
    int32_t kept_in_place; /* ABI 8.  streamed with resident rows: 0 = they entered and left the levels' rings by device copies;
                              1 = swept in place (level 0 reads them where they are kept, the last level writes them there)
                              wherever the rows next to them are kept too; 2 = all rows kept, every pass but the first
                              without a copy; 3 = all rows kept on the lean layout (rings for the levels between the
                              first and the last only, no boxes): no copy at all                                      */
    int32_t peer_check;    /* ABI 9.  a device list whose slabs sit on granules: 1 = a peer copy out of every block a
                              neighbour reads arrived intact before the run started, -1 = it did not and the slabs went on
                              plain hipMalloc blocks instead, 0 = nothing to check (one device, plain blocks)             */
    int32_t first_call;    /* ABI 9.  resident, one device: 1 when this run created its state block (no kept block fitted) */
} tvdn_run_stats;

/* ABI 6.  One slab of a cube that several PROCESSES denoise together, each streaming ITS slab through its GPU from its own
 * page-locked host memory (BASELINE configs[4] with one process per GPU; cytvdn_amd.distributed.denoise_slabs(staged=...) is
 * the caller; what replaces the tiling, per-rank load and halo patching of cyTVDN/mpi.py:131-239, :314-434).  With
 * tvdn_run_args.slab set, shape[0] / data / recon_out / reference describe this slab's OWN rows only, stream_rows / stream_k
 * must be positive and the same on every slab (stream_k at most the own rows of the smallest slab), and what crosses process
 * boundaries goes through the hooks below, all called on the calling thread, at the same points of the schedule on every slab.
 * stream_resident: interior rows of the slab (none of the stream_k a neighbour reads at a shared face) that keep their state
 * in HBM between the passes: -1 as many as fit, 0 none, n at most n (Jia-Zhao, no MSE trace); tvdn_slab_host_need says
 * beforehand how many that will be and what the slab page-locks on the host.
 * sums_out / mse_out receive this slab's share (the caller adds the slabs up); bit-identical to the one-process run. */
typedef struct tvdn_slab_io {
    int64_t global_rows;  /* rows of the WHOLE cube along axis 0                                                       */
    int64_t row0;         /* global index of this slab's first own row                                                  */
    int32_t rank, world;  /* position in the chain (Jia-Zhao) / ring (periodic) of slabs; world >= 2                    */
    int32_t first_row_nonfinite; /* Jia-Zhao: the cube's first row holds Inf / NaN (the same value on every slab)       */
    int32_t reserved;
    /* arrays[i]: a page-locked array of rows_per_array = depth + (own rows that live on the host) + depth rows, those own
     * rows at [own_lo, own_hi) -- the slab's interior rows that stay resident in HBM (stream_resident) have no slot, the
     * `depth` outermost own rows at either end always do.  Fill the `depth` halo rows next to every face this slab shares
     * with a neighbour with that neighbour's outermost `depth` own rows (rows [own_hi - depth, own_hi) of the neighbour below,
     * [own_lo, own_lo + depth) of the one above) and give it mine.  Called once for the data term before the first pass, then
     * before every later pass for recon and the accumulator state.  Non-zero return aborts the run. */
    int (*exchange)(void *user, int32_t n_arrays, void *const *arrays, int64_t rows_per_array, int64_t own_lo, int64_t own_hi,
                    int32_t depth, int64_t row_bytes);
    /* sums3: three doubles over this slab -> summed over all slabs, in place.  With use_stop once per iteration (the three sums);
     * and, when the hook is set, ONCE before the first exchange with {1 if this slab's set-up failed else 0, 0, 0}: what can
     * fail on one rank only (its host's memory, a page-locked allocation) makes every rank return the same error instead of
     * leaving the others inside the exchange until the communicator times out (ABI 7). */
    int (*allreduce)(void *user, double *sums3);
    /* Only with first_row_nonfinite: a BROADCAST, once per pass, of row 0 of every level of the pass (n_planes contiguous
     * row-planes, page-locked) from the slab that owns row 0 to every other slab.  The owner calls it with send = 1 early in
     * its pass (the callee must not wait for the takers there: they arrive late in theirs); every other slab calls it exactly
     * once per pass with send = 0 and gets the planes filled -- a slab whose sweeps reach the cube's top face (the last one,
     * and any whose stream_k-row halo reaches that far) before its first sweep there, the others at the end of the pass. */
    int (*relay_row0)(void *user, int32_t send, void *planes, int32_t n_planes, int64_t row_bytes);
    void *user;
} tvdn_slab_io;

typedef struct tvdn_run_args {
    int32_t dtype;
    int32_t ndim;
    int64_t shape[4];
    int32_t bc_mode;       /* 0 or 2 */
    int32_t device;        /* used when n_devices == 0 */
    int32_t n_fista;
    int32_t n_plain;
    int32_t use_stop;
    int32_t n_devices;     /* 0, or the number of slabs = entries of `devices` */
    double stop;
    double clip[4];        /* 1/lambda per axis, already rounded to the data dtype by the caller */
    double lambda_mu[4];   /* lambda/mu per axis, idem */
    const void *data;      /* host, C-contiguous, never written */
    const void *reference; /* host or NULL */
    void *recon_out;       /* host, same shape/dtype as data */
    double *sums_out;
    double *mse_out;
    int32_t *iters_run;
    int32_t devices[TVDN_MAX_DEVICES];
    /* Out-of-core (ABI 3).  A cube whose state does not fit its device(s) is streamed from pinned host memory with the
     * wavefront schedule: chunks of stream_rows rows, stream_k iterations per PCIe round trip, every row of every iteration
     * swept once (row rings, tvdn_iter_args.ring_rows); bit-identical to the resident run.
     * 0 / 0: never (a state beyond the HBM is refused with TVDN_ERR_UNSUPPORTED and the arithmetic in the message);
     * -1 / -1: decided here -- resident when it fits, else streamed with the library's own plan (tvdn_stream_plan);
     * both > 0: stream with exactly these.  With a device list (ABI 6) every slab is streamed through its own device: the
     * state of the whole cube lives in page-locked host arrays shared by the slabs (two sets: a pass reads one and writes the
     * other), a slab reads stream_k rows of its neighbours' state beyond each interior face from them and gives up a row per
     * level there, the slabs meet after every pass; sums, stopping rule and MSE trace are global (a cube whose first row
     * holds Inf / NaN gets the exact wrap here too: row 0 of every level goes from the first slab's thread to the last's).  A cube whose state the host cannot hold page-locked either is refused
     * before any of the caller's arrays is touched.  Both boundary conditions (periodic: the cube is swept between
     * stream_k wrapped rows at either end, and old and new host state are two sets of arrays); with use_stop one iteration per
     * pass.  `data` / `recon_out` / `reference` are page-locked in place for the duration of the call when they are
     * large (>= 256 MiB) and the runtime allows it, else staged through pinned copies; recon_out then doubles as
     * the host copy of the state. */
    int32_t stream_rows;
    int32_t stream_k;
    /* ABI 4.  host, optional, 2 entries: iterations executed in the FISTA phase and in the unaccelerated phase (with
     * use_stop either may end early; the unaccelerated phase writes its sums from slot n_fista on regardless,
     * cyTVDN.py:201).  What tells a caller which sums_out rows are real without guessing from their values. */
    int32_t *phase_iters;
    /* ABI 5.  Optional: called on the CALLING thread as the run advances, with the number of the last iteration slot
     * handed to the GPU so far (1 .. n_fista + n_plain; the unaccelerated phase counts on from n_fista even when the
     * FISTA phase stopped early) -- what upstream's progress bars show (tqdm, cyTVDN.py:150 / :203).  Launches are
     * asynchronous: without a stopping rule the count runs ahead of the GPU by the depth of its queue.  A resident run
     * whose first iterations follow the upload reports them together once the last chunk has been swept, a streamed run
     * reports a pass at a time.  Must not call back into the library. */
    void (*progress)(int32_t slots_done, void *user);
    void *progress_user;
    /* ABI 5.  Optional device memory for the state of a RESIDENT ONE-DEVICE run (ignored by device lists and streamed
     * runs): 256-byte aligned, at least tvdn_run_workspace_bytes() bytes, on the run's device, contents arbitrary;
     * NULL = the library allocates and frees its own.  Why a caller would: hipMalloc of tens of GiB takes 11 ms most
     * times and 3-5 s some times (profiles/r03_e2e_pipelined.txt: 3 of 10 back-to-back 60 GiB calls), so a process
     * that calls repeatedly keeps the memory (cytvdn_amd passes a block of torch's caching allocator); and a caller
     * that has tried several allocations passes the one that sweeps fastest -- with a workspace
     * the library's own placement audition is off. */
    void *workspace;
    int64_t workspace_bytes;
    /* ABI 6.  Optional, host: filled on success (struct above). */
    tvdn_run_stats *stats;
    /* ABI 6.  Streamed runs: the low rows of axis 0 whose state (data term, recon, accumulators) STAYS in HBM between the
     * passes instead of crossing PCIe twice per pass -- the resident + streamed hybrid.  0: none (every row streams, as
     * before ABI 6); -1: as many as fit beside the rings in 85 % of the free HBM (with stream_rows / stream_k = -1 / -1 the
     * library weighs depth against kept rows itself); n > 0: that many (at most what fits).  Jia-Zhao runs without an MSE
     * trace; ignored otherwise.  Bit-identical to the resident run whatever the split. */
    int64_t stream_resident;
    /* ABI 6.  Optional: this call is ONE slab of a multi-process streamed run (struct above). */
    const tvdn_slab_io *slab;
} tvdn_run_args;

int tvdn_run(const tvdn_run_args *args);

/* ABI 7.  Device memory for a state, composed from physical granules.  The sweep's speed on a state of tens of GiB depends
 * on the allocation it lives in: hipMalloc states of BASELINE configs[1] sweep in 11.0 ... 12.6 ms by the draw, each
 * reproducibly for as long as it is held, while the same state on a virtual range mapped from separately created 1 GiB
 * granules (hipMemCreate / hipMemMap) stays within 11.07 ... 11.45 ms whichever granules it gets and in whatever order
 * (csrc/tvdn_devmem.hip).  tvdn_run takes its own device memory from here; a caller that keeps a state
 * of its own (cytvdn_amd/engine.py does, for one slab of a multi-process run) should too.  Blocks under 2 GiB, and every
 * block when the runtime has no virtual-memory management (or TVDN_VMM=0 is set), are plain hipMalloc blocks: *kind says
 * which it was.  The reference has no counterpart (NumPy allocates its state, cyTVDN/cyTVDN.py:131-145).
 * tvdn_mem_free takes pointers of either kind (it synchronises the device, as hipFree does). */
#define TVDN_MEM_PLAIN 0
#define TVDN_MEM_GRANULES 1
#define TVDN_MEM_CALLER 2
int tvdn_mem_alloc(void **ptr, int64_t bytes, int device, int32_t *kind);
int tvdn_mem_free(void *ptr);

/* ABI 9.  The same, for a block that OTHER devices read and write as well (the slabs of a tvdn_run device list pull halo rows
 * out of their neighbours' blocks peer to peer): `peers` lists those devices, and a block on granules grants each of them
 * access (hipMemSetAccess with one descriptor per device).  A runtime that refuses the grant gets a plain hipMalloc block
 * (whose peer access hipDeviceEnablePeerAccess governs, as before ABI 9); *kind says which it was. */
int tvdn_mem_alloc_shared(void **ptr, int64_t bytes, int device, const int32_t *peers, int32_t n_peers, int32_t *kind);

/* ABI 9.  A block on granules changes size -- or, at the size it has, only its arrangement -- without being freed and
 * allocated anew: the granules it has stay (creating them is what a big block's set-up consists of), missing ones are drawn
 * like a new block's, surplus ones go back, and all of them are dealt out in a fresh random order at a NEW address (*ptr is
 * updated; contents are undefined afterwards; the device is synchronised first).  What tvdn_run does with the block it keeps
 * between runs of different sizes; tools/arrangement_search.py times one state on arrangement after arrangement with it.
 * TVDN_ERR_UNSUPPORTED: not a granule block, or the new size wants another granule size (free and allocate instead);
 * TVDN_ERR_HIP: the runtime refused a step -- the block has been given back and *ptr is NULL. */
int tvdn_mem_resize(void **ptr, int64_t bytes, int device);

/* ABI 9.  What the allocator knows about `device`.  Blocks on granules lean on two work-arounds for silent defects of ROCm
 * 7.2's virtual-memory path (csrc/tvdn_devmem.hip: stale GPU translations after a remap, flushed by a hipFree; equal granule
 * sizes), so the allocator proves them at run time: before a device hands out its first granule block a CANARY plays the
 * product's own unmap / map / flush sequence on two small granules and reads them back by kernel and by hipMemcpy.  A stale
 * word marks the device (vmm_state -1): plain hipMalloc blocks from then on, a line on stderr, results unaffected.  Every
 * hipMem* call's return is checked; `faults` counts the failures since the process started and `first_fault` keeps the first
 * one's text (call, address, size, HIP error).  The reference has no counterpart: it trusts np.zeros_like
 * (cyTVDN/cyTVDN.py:131-145); this is how its replacement earns the same trust. */
#define TVDN_CANARY_NOT_RUN 0
#define TVDN_CANARY_PASSED 1
#define TVDN_CANARY_STALE (-1)  /* a stale translation was seen: granules are off for this process */
#define TVDN_CANARY_FAILED (-2) /* the virtual-memory calls themselves failed: granules are off */
typedef struct tvdn_mem_status_out {
    int32_t vmm_state;   /* 0 no big block asked for yet, 1 granules in use, -1 plain blocks only (TVDN_VMM=0, refused, canary) */
    int32_t canary;      /* TVDN_CANARY_* */
    int32_t canary_runs; /* how often it ran (once per device unless TVDN_VMM_CANARY=always or tvdn_mem_selftest) */
    int32_t faults;      /* hipMem* / flush calls that returned an error, all devices */
    int64_t flushes;     /* TLB flushes done, all devices */
    int64_t blocks, granules, bytes; /* granule blocks alive on this device */
    int32_t last_granules, last_pool; /* the last draw on this device: granules kept / created to choose them from (the pool is
                            bounded by 90 % of the free HBM, a floor of max(4 GiB, 5 %) left free, and TVDN_HBM_LIMIT) */
    char first_fault[200];
} tvdn_mem_status_out;
int tvdn_mem_status(int device, tvdn_mem_status_out *out);
/* Runs the canary now.  TVDN_OK; TVDN_ERR_UNSUPPORTED: a stale translation was seen; TVDN_ERR_HIP: the calls failed
 * (either way the device is marked and tvdn_last_error() says what was seen). */
int tvdn_mem_selftest(int device);

/* ABI 9.  Optional: what the FIRST tvdn_run of a process pays once -- the pinned staging lanes, the first stream of each priority
 * class, the reduction scratch, the library's code object on the device, the allocator's canary: 0.1-0.15 s together -- paid
 * now, e.g. on a helper thread while the caller still reads its cube from disk.  Idempotent; TVDN_OK, or a status when no device
 * is there.  The reference has no counterpart (its first call pays for nothing of the kind). */
int tvdn_warm_up(int device);

/* ABI 7.  Bytes of the block the last one-device tvdn_run on `device` kept for the next one (0: none): device memory that
 * hipMemGetInfo reports as used but that the next tvdn_run takes over or releases, i.e. free for planning purposes
 * (cytvdn_amd/planner.py adds it to what the device has free; tvdn_plan and tvdn_stream_plan do so themselves). */
int64_t tvdn_state_kept_bytes(int device);

/* A one-device tvdn_run that allocates its own device memory -- the state of a resident run, or the rings, boxes and
 * resident rows of a streamed one -- KEEPS that block when it returns (one per device) and hands it to the next run it fits
 * (same size, or up to a quarter larger than needed; a kept block of another size is released before the new one is
 * allocated): releasing and re-allocating tens of GiB in quick succession costs about 0.7 s per hipMalloc and 1.1 s per
 * hipFree on this platform, with single stalls of several seconds, i.e. more than a 50-iteration run itself.  NOTE that the
 * block (up to most of the HBM) therefore stays allocated after tvdn_run returns: a caller that needs the device memory for
 * something else calls tvdn_release_cache() (always TVDN_OK), or sets the environment variable TVDN_KEEP_STATE=0, which
 * never keeps one.  tvdn_plan and tvdn_stream_plan count the kept block as free; device lists and runs with a `workspace`
 * release it first. */
int tvdn_release_cache(void);

/* ABI 6.  One slab of a multi-process streamed run (args->slab set, stream_rows / stream_k > 0, stream_resident as for the run):
 * the bytes of host memory this slab will page-lock, and the interior rows it will keep resident in HBM instead (none of the
 * stream_k rows a neighbour reads at a shared face; as many as fit beside the rings in 85 % of the device's free memory).  The
 * guard inside tvdn_run knows nothing of the other ranks on its host: a launcher adds these figures up per host and refuses on
 * every rank alike BEFORE any rank page-locks anything (cytvdn_amd.distributed.denoise_slabs does).  Queries the device. */
int tvdn_slab_host_need(const tvdn_run_args *args, int64_t *need_bytes, int64_t *resident_rows);

/* ABI 6.  Pure arithmetic (no device): for a slab (args->slab, shape[0] = own rows, bc_mode) whose passes are `depth` levels
 * deep and which keeps `resident_rows` interior rows in HBM, local_slot[i] of own row i = its row in the packed local arrays the
 * exchange hook sees (depth halo rows come first), or -1 when the row is resident.  The map the run itself uses. */
int tvdn_slab_row_map(const tvdn_run_args *args, int64_t depth, int64_t resident_rows, int64_t *local_slot);

/* ABI 6.  A streamed tvdn_run hands the page-locked host memory it allocated back in the BACKGROUND (unpinning 144 GiB takes
 * about 7 s; the call returns without waiting for it, and the process's exit handlers do wait).  tvdn_wait_background() returns
 * once every such release has finished -- for a caller that wants the memory back before it goes on, or that times a second
 * streamed call without the first one's teardown in it.  Always TVDN_OK. */
int tvdn_wait_background(void);

/* Bytes of device memory the state of a resident one-device run of these args takes (dtype, ndim, shape, n_fista > 0
 * are read): what tvdn_run allocates itself, or expects behind tvdn_run_args.workspace.  Pure host arithmetic. */
int tvdn_run_workspace_bytes(const tvdn_run_args *args, int64_t *bytes);

/* How a RESIDENT one-device tvdn_run overlaps its two transfers with iterations (what upstream does one after the other:
 * datacube in, cyTVDN.py:145; recon out, :244-247): out[0] = rows per chunk, out[1] = iterations that follow the upload
 * chunk by chunk, out[2] = iterations that run over the download; all 0 = plain order (upload, iterate, download).  Pure
 * host arithmetic, the very function tvdn_run asks: cubes from 256 MiB and 32 rows on, runs from 4 iterations on, eight
 * chunks, at most 8 iterations at either end.  tvdn_run applies it to Jia-Zhao runs without MSE trace whose first row is
 * finite -- with a stopping rule the start only (out[2] taken as 0; a rule met inside the iterations that followed the upload
 * has the run done again in plain order) --; recon is bit-identical to the plain order either way, the sums agree with it to rounding (1e-6
 * relative: the rows' partial sums are added in another order).  Environment: TVDN_PIPELINE=0
 * (never), "rows,k_start,k_end" (forced). */
int tvdn_pipeline_plan(int64_t n0, int32_t n_iters, int64_t cube_bytes, int32_t *out);

/* Host side of a STREAMED tvdn_run as arithmetic only -- no HIP call, no device needed, none of the caller's arrays
 * dereferenced: *need_bytes = the page-locked host memory the run would hold (data term, recon = recon_out, the
 * reference when an MSE trace is asked for, one or two accumulator-state arrays per axis, one more cube when data
 * overlaps recon_out); *avail_bytes = what the host may give: MemAvailable, capped by the physical memory, by the
 * control group's limit and by the environment variable TVDN_HOST_LIMIT ("64G", "512M", bytes).  Returns TVDN_OK when
 * need <= 80 % of avail, else TVDN_ERR_UNSUPPORTED with both numbers in tvdn_last_error().  tvdn_run makes this very
 * call before it touches anything (the reference's counterpart is check_memory, cyTVDN/cyTVDN.py:438-467, which only
 * prints). */
int tvdn_stream_host_need(const tvdn_run_args *args, int64_t *need_bytes, int64_t *avail_bytes);

/* ABI 6.  What a streamed tvdn_run of these args (dtype, ndim, shape, n_fista / n_plain, use_stop, bc_mode, reference /
 * mse_out, stream_resident are read) chooses when it may use `hbm_free_bytes` of HBM (<= 0: what args->device has free now):
 * rows per chunk, iterations per pass, resident rows, and what that costs in HBM and in page-locked host memory.  The very
 * function tvdn_run asks with stream_rows / stream_k = -1 / -1; pure host arithmetic when hbm_free_bytes is given
 * (cytvdn_amd/planner.py plans with it; the reference's counterpart is check_memory, cyTVDN/cyTVDN.py:438-467). */
typedef struct tvdn_stream_plan_out {
    int64_t rows;
    int64_t k;
    int64_t resident_rows;
    int64_t hbm_bytes;
    int64_t host_bytes;
} tvdn_stream_plan_out;

int tvdn_stream_plan(const tvdn_run_args *args, int64_t hbm_free_bytes, tvdn_stream_plan_out *out);

/* The host side of the schedule, for callers that drive tvdn_iterate_fused themselves (cytvdn_amd/engine.py does): the
 * very functions tvdn_iterate_many and tvdn_run use, so that the parity-critical logic exists once.  Pure host code.
 *   tvdn_fista_ratios   out[i] = (tk-1)/tk_new of iteration i, float64 recurrence of cyTVDN/cyTVDN.py:153-156
 *   tvdn_iter_mode      the TVDN_ITER_* form of an iteration on the compact state: FISTA iterations keep the d-form,
 *                       the first unaccelerated one after them converts it to b, later ones ping-pong b (>= 0;
 *                       TVDN_ERR_INVALID for a FISTA iteration on a b-form state)
 *   tvdn_roles_bind     points it->recon_in/out, tk, tk_prev, b/d/dprev pointers and mode at the arrays of the
 *                       iteration about to run (state->cur / i_* / d_form say which array plays which role)
 *   tvdn_roles_advance  the role rotation after that iteration (state updated in place) */
int tvdn_fista_ratios(int32_t n, double *out);
int tvdn_iter_mode(int32_t use_fista, int32_t d_form);
int tvdn_roles_bind(const tvdn_many_args *state, int32_t use_fista, double ratio, tvdn_iter_args *it);
int tvdn_roles_advance(tvdn_many_args *state, int32_t use_fista, double ratio);

/* The HBM arithmetic of the reference's check_memory (cyTVDN/cyTVDN.py:438-467) for this engine: how many
 * arrays the compact state has, how many bytes the tallest slab of an `n_slabs`-way split of axis 0 needs
 * (halo rows and array stagger included), what `device` has free, whether that fits in 90 % of it, and the
 * fewest slabs that would fit one per device of that size (0: none).  tvdn_run consults it and returns
 * TVDN_ERR_UNSUPPORTED with this arithmetic in the message when its slabs cannot fit. */
typedef struct tvdn_plan_out {
    int64_t arrays;
    int64_t bytes_per_slab;
    int64_t free_bytes;
    int32_t fits;
    int32_t min_slabs;
} tvdn_plan_out;

int tvdn_plan(int dtype, int ndim, const int64_t *shape, int fista, int n_slabs, int device,
              tvdn_plan_out *out);

/* ------------------------------------------------------------------------------------------
 * Whole-array transfers between ordinary (pageable) host memory and HBM at PCIe speed: what
 * `recon = datacube.copy()` on the way in (cyTVDN/cyTVDN.py:145) and the returned array on the way
 * out (:244-247) become when the state lives on the GPU.  Several host threads, each with two pinned
 * bounce buffers and its own HIP stream, move interleaved 16 MiB chunks; a freshly allocated
 * destination is first-touched by all of them.  Synchronous: returns when the bytes have arrived.
 * The caller synchronises the stream that produced (to_host) or will consume (to_device) the
 * device buffer.  tvdn_run uses these for its own transfers.
 * ---------------------------------------------------------------------------------------- */
int tvdn_copy_to_device(void *dst_device, const void *src_host, size_t bytes, int device);
int tvdn_copy_to_host(void *dst_host, const void *src_device, size_t bytes, int device);

/* `n` copies of `bytes_each` bytes (a multiple of 16; 16-byte aligned, non-overlapping segments) behind one or a
 * few launches.  `dst` / `src` are HOST arrays of pointers the device can dereference: HBM, or pinned host memory
 * (hipHostMalloc / a pinned torch tensor) -- the streamed engines move a chunk's rows of all state arrays across
 * PCIe with one launch per direction.  max_blocks > 0 caps the workgroups of a launch (they loop over the pieces):
 * a transfer that waits on PCIe must not occupy the wave slots the sweeps running beside it need; 0 = one
 * workgroup per 64 KiB piece (HBM to HBM).  Asynchronous on `stream`. */
int tvdn_copy_many(int32_t n, void *const *dst, const void *const *src, int64_t bytes_each, int32_t max_blocks,
                   void *stream);

/* Measurement aid: a pure stream of n_read arrays in and n_write arrays out (16-byte elements, one per thread,
 * streaming accesses, XCD-aware workgroup order) over caller-chosen device arrays of bytes_each bytes -- what HBM
 * gives the read/write mix of a sweep on the very arrays (the very physical pages) the sweep uses; tools/ceiling_vs_sweep.py
 * puts the two side by side.  Instantiated for the mixes of the fused sweep: 10/5, 6/5, 8/4, 5/4, and 1/1. */
int tvdn_stream_mix(int32_t n_read, const void *const *in, int32_t n_write, void *const *out, int64_t bytes_each,
                    void *stream);
/* The same stream in the sweep's own traversal: a workgroup owns a 4 KiB tile of a row-plane and marches `chunk` of the
 * `rows` rows (10/5 and 6/5 mixes): what that structure reaches before neighbour re-reads and arithmetic. */
int tvdn_stream_mix_march(int32_t n_read, const void *const *in, int32_t n_write, void *const *out, int64_t bytes_each,
                          int64_t rows, int32_t chunk, void *stream);

/* Synthetic input (cytvdn_amd/synth.py restated on the device, bit-identical): fills rows
 * [row0, row0+rows) of the GLOBAL cube `shape` into `out` (rows*prod(shape[1:]) elements). */
int tvdn_synth_fill(int dtype, int ndim, const int64_t *shape, uint64_t seed, int64_t row0,
                    int64_t rows, void *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TVDN_H */
