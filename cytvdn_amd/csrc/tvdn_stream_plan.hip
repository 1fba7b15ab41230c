// The streamed engine's arithmetic: what a shape (chunk height R, depth K, rows kept resident) costs in HBM and in page-locked
// host memory, what a run chooses when asked to decide, and the same figures for one slab of a multi-process run -- as
// functions the run itself calls and as C entry points a caller can plan with (include/tvdn.h: tvdn_stream_plan,
// tvdn_stream_host_need, tvdn_slab_host_need, tvdn_slab_row_map; the reference's counterpart is check_memory,
// cyTVDN/cyTVDN.py:438-467, which only prints).  Also the page-locking of a caller's result array for the resident run's
// pipelined download (host_pin_result) and tvdn_wait_background.
#include "tvdn_stream_parts.hpp"

namespace tvdn {

// A caller's RESULT array page-locked where it is, for the pipelined download of a resident run (tvdn_run.hip): huge pages asked
// for, pages touched by many threads (contents kept), one registration -- 4 GiB in 30-40 ms when the array is fresh.  The rows
// then cross PCIe straight into the caller's memory (55 GB/s) instead of through the pinned lanes and a host copy into pages
// that fault in as they are written (36 GB/s).  TVDN_ERR_* when the runtime refuses: the caller keeps the lanes.
int host_pin_result(void *user, size_t bytes)
{
    const size_t pin_min = getenv("TVDN_PIN_IN_PLACE_MIN") ? env_bytes("TVDN_PIN_IN_PLACE_MIN") : kPinInPlaceMinDefault;
    if (!user || bytes < pin_min) return TVDN_ERR_UNSUPPORTED;
    const uintptr_t lo = ((uintptr_t)user + (size_t(2) << 20) - 1) & ~((uintptr_t)(size_t(2) << 20) - 1);
    const uintptr_t hi = ((uintptr_t)user + bytes) & ~((uintptr_t)(size_t(2) << 20) - 1);
    if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
    touch_pages((char *)user, bytes, touch_threads());
    if (hipHostRegister(user, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return TVDN_ERR_HIP;
    }
    return TVDN_OK;
}

void host_unpin_result(void *user) { (void)hipHostUnregister(user); }

// Rows of HBM (planes) the schedule keeps besides the resident rows: rings of R+2 rows per level and array, the data-term
// ring(s), the staging boxes, the planes of an exact Jia-Zhao wrap and one plane of zeros (planner.wavefront_windows of the
// Python side).
int64_t stream_planes(int nd, int64_t rows, int64_t k, bool mse, bool wrap)
{
    return ((k + 1) + (k + 2) * nd) * (rows + 2) + (rows + k + 3) * (mse ? 2 : 1) + 2 * (3 + 4 * nd) * rows + 2 * (1 + 2 * nd) + (wrap ? 2 * (k + 1) : 0) + 1;
}

// With EVERY row kept and swept in place (tvdn_stream_chain.hip) only the levels between the first and the last have rings: level 0
// reads the store, the last level writes it (and with a d-form state the two levels next to them share d arrays through it), the
// rows never pass a box and `orig` is read where it is kept.  Rings: recon of levels 1 .. K-1, state of levels 1 .. K-1.
int64_t stream_planes_all_kept(int nd, int64_t rows, int64_t k, bool wrap)
{
    return (k - 1) * (1 + nd) * (rows + 2) + (wrap ? 2 * (k + 1) : 0) + 1;
}

size_t stream_device_bytes_all_kept(int nd, int64_t R, int64_t K, size_t row_bytes)
{
    auto aligned = [](size_t b) { return (b + 255) / 256 * 256; };
    return (size_t)(K - 1) * (size_t)(1 + nd) * aligned((size_t)(R + 2) * row_bytes) + (2 * (size_t)(K + 1) + 1) * aligned(row_bytes);
}

// Chunk height R, depth K and the number of rows whose state STAYS in HBM between passes (the resident + streamed hybrid).
// A pass costs max(PCIe time of the streamed rows, sweep time of all rows + the device copies of the resident rows); the
// choice minimises that per iteration over every (R, K) whose rings fit 85 % of the free HBM, the rows kept being what the
// rest of that budget holds (2 + nd x n_state arrays per row: data term, recon, accumulator state).  Without kept rows this is
// "the deepest K the sweeps can keep up with": depth and chunk height compete for the HBM (a level costs R + 2 rows per array),
// and one-row chunks buy a third more depth than two-row ones at 6 % slower sweeps.  With kept rows, depth and kept rows
// compete and the model decides.  Rates measured on MI355X (profiles/r04_stream_rates.jsonl, r04_pcie_duplex.jsonl): a row
// crosses the link in max(up / 55 GB/s, down / 42.5 GB/s) when both directions are busy (runtime copies up, an 8-workgroup copy
// kernel down; a pass of N rows at depth K takes N + K such steps: 3 x 64 rows at K = 50 modelled 13.8 s, measured 13.8 s);
// sweeps on rings at 0.82 x 5.6 TB/s of moved bytes (0.77 x in one-row launches); device copies at 4.8 TB/s.
int choose_stream_shape(int nd, int64_t n_rows, size_t row_bytes, size_t free_bytes, bool mse, bool wrap, int n_state, bool may_keep,
                        int64_t k_cap, int64_t *rows_out, int64_t *k_out, int64_t *res_out, int64_t n_iters)
{
    const int64_t budget = (int64_t)(0.85 * (double)free_bytes / (double)row_bytes);
    const double rb = (double)row_bytes;
    // what a streamed row carries per pass: the data term and the state up, the state down (recon is rebuilt on the device and
    // comes down with the last pass only, tvdn_rebuild.hip) -- with a stopping rule (k_cap == 1) recon both ways as well
    const int ships = k_cap <= 1 ? 1 : 0;
    const int n_in = 1 + ships + nd * n_state, n_out = ships + nd * n_state, moved = 3 + nd * (n_state + 1);
    k_cap = std::max<int64_t>(1, std::min<int64_t>({k_cap, 128, std::max<int64_t>(1, n_rows)}));
    // one streamed row, both directions busy: uploads and downloads by the DMA engines (round 5: downloads one copy at a time from
    // a helper thread, tvdn_stream_parts.hpp DownPump -- 55.7 GB/s up beside 47.7 down and undisturbed sweeps in
    // tools/ubench/pcie_down_kernels.hip; 42.5 down through round 4's copy kernel)
    const double row_step = std::max((double)n_in * rb / 55e9, (double)n_out * rb / 48e9);
    int64_t best_k = 0, best_r = 0, best_res = 0;
    double best_t = 0.0;
    for (int64_t r : {32, 16, 8, 4, 2, 1}) {
        if (r > 1) r = std::min<int64_t>(r, std::max<int64_t>(2, n_rows));
        // sweeps on rings, launches of r rows: 0.86 ms per 256 MiB plane and level whether r is 2, 4 or 8 (83 % of the resident
        // sweep's rate; all rows resident, profiles/r04_stream_rates.jsonl), 0.92 ms in one-row launches
        const double eff = r == 1 ? 0.80 : 0.82;  // (one-row launches: 0.89 ms per plane and level since nothing else runs kernels beside them)
        for (int64_t k = 1; k <= k_cap; ++k) {
            const int64_t planes = stream_planes(nd, r, k, mse, wrap);
            const int n_store = 2 + nd * n_state;  // arrays a kept row holds in HBM: data term, recon, state
            const bool fits = planes <= budget;     // the rings and boxes of the general layout
            const bool lean_fits = may_keep && !mse && k >= 3 && stream_planes_all_kept(nd, r, k, wrap) + (int64_t)n_store * n_rows <= budget;
            if (!fits && !lean_fits && k >= 3) break;
            if (!fits && !lean_fits) continue;
            // With the number of iterations known, only the depths a run settles on: ceil(n / k) passes of (almost) equal depth
            // (StreamRun::set_up).  80 iterations asked at k = 13 run as 7 passes of 12 -- and would size the rows kept for
            // rings of 13 levels (53 rows of 256 MiB planes kept where 56 fit).
            int64_t passes = 0;
            if (n_iters > 0) {
                passes = (n_iters + k - 1) / k;
                if ((n_iters + passes - 1) / passes != k) continue;
            }
            const double t_sweeps = (double)n_rows * (double)k * moved * rb / (5.6e12 * eff);
            auto offer = [&](int64_t res, double t) {  // t: seconds per iteration
                if (best_k == 0 || t < best_t * 0.999) {
                    best_t = t;
                    best_k = k;
                    best_r = r;
                    best_res = res;
                }
            };
            auto per_iteration = [&](double t_pass) { return passes > 0 ? t_pass * (double)passes / (double)n_iters : t_pass / (double)k; };
            // (a) nothing kept: the pipeline of a pass fills and drains over K rows; chained passes share that between them
            //     (half of it counted)
            //     -- all of it when there are fewer than three passes to chain, a third of it when three, ...: 3 x 49 levels over
            //     64 rows of 256 MiB planes 9.75 s measured (64.8 Gvoxel-iters/s), 2 x 49 drained 8.79 s
            const double fill = passes >= 3 ? (double)std::min<int64_t>(k, n_rows) / (double)passes : (passes > 0 ? (double)std::min<int64_t>(k, n_rows) : 0.5 * (double)std::min<int64_t>(k, n_rows));
            if (fits) offer(0, per_iteration(std::max(((double)n_rows + fill) * row_step, t_sweeps)));
            // (b) what the rest of the budget holds kept.  The passes are drained, so the link works both ways at once: 68 GB/s
            //     together when a pass waits for it.  When it does not, a row-plane takes 0.89 ms per level (the 0.82 x 5.6 TB/s
            //     above), a kept row 1.44 ms per pass on top (its 2 x (2 n_store - 1) plane copies between store and rings, mostly
            //     beside the sweeps: 7.1 TB/s) and a streamed row 10 ms per pass although its transfers are hidden (they share
            //     the HBM with the sweeps).  Fitted to 32 rows of 256 MiB planes, all kept, at k = 8 and 16 (profiles/
            //     r04_stream_rates.jsonl, run r4ah) and 64 rows, 80 iterations at k = 8 / 10 / 12 / 16 / 20 with 64 / 60 / 56 /
            //     47 / 37 rows kept: 5.22 / 5.50 / 5.50 / 5.84 / 6.53 s measured, 5.48 / 5.72 / 5.81 / 5.83 / 7.25 modelled
            //     (profiles/r05_hybrid_depths.jsonl).  Not with one-row chunks: (1, 20, 47 kept) ran at 46.6 Gvoxel-iters/s
            //     where (2, 12, 56 kept) ran at 57.8.
            const int64_t res = fits && may_keep && r > 1 ? std::min<int64_t>(n_rows, (budget - planes) / n_store) : 0;
            // every row kept: swept in place, no copy (0.77-0.89 ms per 256 MiB plane and level by chunk height: the rows a launch
            // reads beside its own weigh less in a taller one -- 32 rows of 256 MiB planes, 80 iterations: (R 2, K 3) 2.13 s, (4, 3)
            // 2.03, (8, 3) 1.98, (4, 9) 1.96; a pass costs ~3.5 ms to fill and drain), on the lean layout above when every pass is
            // at least three levels deep
            const bool lean = lean_fits && (passes == 0 || n_iters / passes >= 3);  // (depths of 3 and 2 are re-balanced by the run itself: a deeper K, offered as such)
            if (lean || (res >= n_rows && k >= 3)) {
                // (64 rows on the lean layout: (R 4, K 3) 4.05 s, (8, 5) 3.92, (16, 4) 3.98 -- nothing beyond eight-row chunks)
                const double eff_kept = r >= 16 ? 0.925 : (r >= 8 ? 0.935 : (r >= 4 ? 0.915 : (r == 2 ? 0.88 : 0.80)));
                offer(n_rows, (double)n_rows * moved * rb / (5.6e12 * eff_kept) + 3.5e-3 * rb / 268.4e6 * (passes > 0 ? (double)passes / (double)n_iters : 1.0 / (double)k));
            } else if (res > 0) {
                const double link = (double)(n_rows - res) * (n_in + n_out) * rb;
                // (a pass that streams anything also waits for a fifth of one row's way up and down: 63 of 64 rows kept at k = 9
                //  ran 5.53 s where all 64 at k = 8 ran 5.22)
                const double fill = n_rows > res ? 0.2 * ((double)n_in * rb / 55e9 + (double)n_out * rb / 42.5e9) : 0.0;
                const double beside = (double)res * (2 * n_store - 1) * 2.0 * rb / 7.1e12 + link / 460e9 + fill;  // per pass
                const double sweeps_it = (double)n_rows * moved * rb / (5.6e12 * 0.82);                    // per iteration
                const double per_it = passes > 0 ? (double)passes / (double)n_iters : 1.0 / (double)k;     // passes per iteration
                offer(res, std::max(link / 68e9 * per_it, sweeps_it + beside * per_it));
            }
        }
    }
    if (best_k < 1) {
        set_error("not even 2-row chunks of one iteration level fit the device: %lld planes of %zu bytes in %zu free bytes",
                  (long long)stream_planes(nd, 2, 1, mse, wrap), row_bytes, free_bytes);
        return TVDN_ERR_UNSUPPORTED;
    }
    *rows_out = best_r;
    *k_out = best_k;
    *res_out = best_res;
    return TVDN_OK;
}

// Page-locked host bytes of a streamed run that keeps the state of `res` low rows in HBM, and what the host may give.
int stream_host_need(const tvdn_run_args *a, int64_t res, int64_t *need_bytes, int64_t *avail_bytes)
{
    TVDN_REQUIRE(a != nullptr, "args is NULL");
    TVDN_REQUIRE(a->dtype == TVDN_F32 || a->dtype == TVDN_F64, "bad dtype %d", a->dtype);
    TVDN_REQUIRE(a->ndim == 3 || a->ndim == 4, "ndim must be 3 or 4, got %d", a->ndim);
    double cube = a->dtype == TVDN_F32 ? 4.0 : 8.0;
    size_t cube_b = a->dtype == TVDN_F32 ? 4 : 8;
    for (int i = 0; i < a->ndim; ++i) {
        TVDN_REQUIRE(a->shape[i] >= 1, "shape[%d] must be >= 1", i);
        cube *= (double)a->shape[i];
        cube_b *= (size_t)a->shape[i];
    }
    const int n_state = a->n_fista > 0 ? 2 : 1;
    const bool want_mse = a->mse_out != nullptr && a->reference != nullptr;
    const bool aliased = a->data && a->recon_out && cube < 9.0e18 && arrays_overlap(a->data, a->recon_out, cube_b);
    // periodic boundaries: the rows at one end are the other end's halo, uploaded late in a pass that has already sent
    // their new values home -- old and new state are then two sets of arrays instead of one updated in place
    const int twice = a->bc_mode == TVDN_BC_PERIODIC ? 2 : 1;
    const double share = res <= 0 ? 1.0 : (res >= a->shape[0] ? 0.0 : (double)(a->shape[0] - res) / (double)a->shape[0]);
    const double need = (double)((a->ndim * n_state + 1) * twice + 1 + (want_mse ? 1 : 0) + (aliased ? 1 : 0)) * cube * share;
    const size_t avail = host_available_bytes();
    if (need_bytes) *need_bytes = need < 9.0e18 ? (int64_t)need : INT64_MAX;
    if (avail_bytes) *avail_bytes = (int64_t)avail;
    if (need > 0.0 && (avail == 0 || need > 0.8 * (double)avail)) {
        set_error("a streamed run of this cube needs %.0f bytes of page-locked host memory, which exceeds what the host has "
                  "available (%zu bytes, of which 80 %% are used at most): cut it into slabs over several nodes (cytvdn_amd.plan_run)",
                  need, avail);
        return TVDN_ERR_UNSUPPORTED;
    }
    return TVDN_OK;
}

// HBM bytes of everything a streamed run keeps on the device besides resident rows -- rings of R + 2 rows per level and array,
// the data-term ring(s), two in and two out boxes, the planes of an exact wrap, the plane of zeros -- and of ONE resident row
// (data term, recon, accumulator state).  One definition: run_streamed allocates by it, run_streamed_rank sizes a slab's
// packed host arrays by it before run_streamed runs.
size_t stream_device_bytes(int nd, int n_state, bool want_mse, int64_t R, int64_t K, size_t row_bytes, size_t *per_resident_row)
{
    auto aligned = [](size_t b) { return (b + 255) / 256 * 256; };
    const int n_in = 2 + nd * n_state + (want_mse ? 1 : 0), n_out = 1 + nd * n_state, n_store = 2 + nd * n_state;
    const size_t ring_b = aligned((size_t)(R + 2) * row_bytes), oring_b = aligned((size_t)(R + K + 3) * row_bytes);
    const size_t box_b = aligned((size_t)R * row_bytes), obox_b = aligned((size_t)(R + 1) * row_bytes), plane_b = aligned(row_bytes);
    const size_t n_rings = (size_t)(K + 1) + (size_t)(K + 2) * nd;
    if (per_resident_row) *per_resident_row = (size_t)n_store * plane_b;
    return n_rings * ring_b + oring_b * (want_mse ? 2 : 1) + 2 * ((size_t)n_in * box_b + (size_t)n_out * obox_b) + 2 * (size_t)(K + 1) * plane_b + plane_b;
}

// What one slab of a multi-process streamed run holds where: the depth its passes settle on (= the halo rows it keeps of each
// neighbour), the interior rows that stay resident in HBM, the rows of its packed local host arrays.  One definition: the run
// allocates by it, tvdn_slab_host_need tells the caller beforehand (so that the ranks of one host can add up what they will
// page-lock BEFORE any of them does).  Looks at the device's free memory unless nothing can be kept anyway.
int slab_shape(const tvdn_run_args *a, int64_t R, int64_t K, int64_t *kc_out, int64_t *res_out, int64_t *local_rows_out)
{
    const tvdn_slab_io *io = a->slab;
    TVDN_REQUIRE(io != nullptr, "tvdn_run_args.slab is NULL");
    TVDN_REQUIRE(a->dtype == TVDN_F32 || a->dtype == TVDN_F64, "bad dtype %d", a->dtype);
    TVDN_REQUIRE(a->ndim == 3 || a->ndim == 4, "ndim must be 3 or 4, got %d", a->ndim);
    TVDN_REQUIRE(R >= 1 && K >= 1, "a slab run needs stream_rows >= 1 and stream_k >= 1");
    const int nd = a->ndim;
    size_t row_bytes = a->dtype == TVDN_F32 ? 4 : 8;
    for (int i = 1; i < nd; ++i) row_bytes *= (size_t)a->shape[i];
    const int64_t own = a->shape[0], N0 = io->global_rows;
    const int n_total = a->n_fista + a->n_plain;
    const int n_state = a->n_fista > 0 ? 2 : 1;
    const bool want_mse = a->mse_out != nullptr && a->reference != nullptr;
    const bool periodic = a->bc_mode == TVDN_BC_PERIODIC;
    // the depth run_streamed will settle on (its own arithmetic: clamp, number of passes, equal depths) = the halo rows kept
    int64_t kc = a->use_stop ? 1 : std::min<int64_t>({K, (int64_t)std::max(n_total, 1), N0});
    if (!a->use_stop && n_total > 0) {
        const int64_t n_pass = (n_total + kc - 1) / kc;
        kc = n_total / n_pass + (n_total % n_pass ? 1 : 0);
    }
    // Interior own rows -- none of the kc rows a neighbour reads at a shared face -- may keep their state in HBM between the
    // passes (the resident + streamed hybrid, as on one device): as many as fit beside the rings in 85 % of the free HBM, evenly
    // spread over the interior (stream_resident / TVDN_STREAM_RESIDENT cap the count; none with an MSE trace or periodic
    // boundaries).  They have no slot in the local arrays, which shrink accordingly: what makes BASELINE configs[4] fit the host
    // memory of ONE node (10 arrays x (128 + 2 k) rows of 256 MiB per rank is 3.2 TB over 8 ranks at k = 16; with 50 rows per
    // rank resident, 2.2 TB).
    const bool face_lo = periodic || io->row0 > 0, face_hi = periodic || io->row0 + own < N0;  // faces shared with a neighbour
    const int64_t interior = std::max<int64_t>(0, own - (face_lo ? kc : 0) - (face_hi ? kc : 0));
    int64_t res = 0;
    if (!want_mse && !periodic && interior > 0 && a->stream_resident != 0 && n_total > 0) {
        DeviceRestore restore;
        TVDN_HIP(hipSetDevice(a->device));
        size_t free_b = 0, total_b = 0, per_row = 0;
        TVDN_HIP(hipMemGetInfo(&free_b, &total_b));
        free_b += state_kept_bytes(a->device);
        if (const size_t cap_b = env_bytes("TVDN_HBM_LIMIT")) free_b = std::min(free_b, cap_b);
        const size_t fixed = stream_device_bytes(nd, n_state, want_mse, R, kc, row_bytes, &per_row);
        const size_t lim = (size_t)(0.85 * (double)free_b);
        res = lim > fixed ? std::min<int64_t>(interior, (int64_t)((lim - fixed) / per_row)) : 0;
        if (a->stream_resident > 0) res = std::min<int64_t>(res, a->stream_resident);
        if (const char *e = getenv("TVDN_STREAM_RESIDENT")) res = std::max<int64_t>(0, std::min<int64_t>(res, (int64_t)atoll(e)));
    }
    *kc_out = kc;
    *res_out = res;
    *local_rows_out = own + 2 * kc - res;  // packed: halo, the own rows that live on the host, halo
    return TVDN_OK;
}

}  // namespace tvdn

// Host side of a streamed run, as arithmetic only (no HIP call, no device needed, nothing of the caller's dereferenced):
// the page-locked bytes it would hold -- the data term, recon (= recon_out), the reference when an MSE trace is asked
// for, one or two accumulator-state arrays per axis, and one more cube when `data` overlaps `recon_out` (the data term
// then needs its own copy) -- against what the host may give (MemAvailable, physical memory, the control group's
// limit, TVDN_HOST_LIMIT), of which a streamed run takes 80 % at most.  The figure is the one of a run that keeps NO rows
// resident in HBM (stream_resident = 0): an upper bound for the others.
extern "C" int tvdn_stream_host_need(const tvdn_run_args *a, int64_t *need_bytes, int64_t *avail_bytes)
{
    return tvdn::stream_host_need(a, 0, need_bytes, avail_bytes);
}

// A streamed run returns its page-locked host state in the background (unpinning and unmapping 16 GiB takes 0.8 s; a run that
// held 144 GiB would spend 7 s on it before it returned).  This waits until every such release has finished: the memory is
// back with the operating system, and the next streamed call will not find the runtime busy unpinning.
extern "C" int tvdn_wait_background(void)
{
    tvdn::wait_for_releases();
    return TVDN_OK;
}

// What a streamed tvdn_run of these args would choose with `hbm_free_bytes` of HBM to work with (<= 0: ask args->device):
// chunk height, depth, resident rows; the HBM bytes of rings + boxes + resident rows; the page-locked host bytes.  Pure
// arithmetic when hbm_free_bytes is given (no device needed): cytvdn_amd/planner.py plans with it.
extern "C" int tvdn_stream_plan(const tvdn_run_args *a, int64_t hbm_free_bytes, tvdn_stream_plan_out *out)
{
    using namespace tvdn;
    TVDN_REQUIRE(a != nullptr && out != nullptr, "NULL argument");
    TVDN_REQUIRE(a->dtype == TVDN_F32 || a->dtype == TVDN_F64, "bad dtype %d", a->dtype);
    TVDN_REQUIRE(a->ndim == 3 || a->ndim == 4, "ndim must be 3 or 4, got %d", a->ndim);
    size_t row_bytes = a->dtype == TVDN_F32 ? 4 : 8;
    for (int i = 0; i < a->ndim; ++i) {
        TVDN_REQUIRE(a->shape[i] >= 1, "shape[%d] must be >= 1", i);
        if (i) row_bytes *= (size_t)a->shape[i];
    }
    if (hbm_free_bytes <= 0) {
        DeviceRestore restore;
        size_t free_b = 0, total_b = 0;
        TVDN_HIP(hipSetDevice(a->n_devices > 0 ? a->devices[0] : a->device));
        TVDN_HIP(hipMemGetInfo(&free_b, &total_b));
        // (the block the last run of this device kept is the next run's to take over: it counts as free)
        hbm_free_bytes = (int64_t)(free_b + state_kept_bytes(a->n_devices > 0 ? a->devices[0] : a->device));
    }
    const bool mse = a->mse_out != nullptr && a->reference != nullptr;
    const int n_state = a->n_fista > 0 ? 2 : 1;
    const bool keep = a->bc_mode == TVDN_BC_JIA_ZHAO && !mse && a->stream_resident != 0;
    int64_t rows = 0, k = 0, res = 0;
    const int n_total = a->n_fista + a->n_plain;
    const int rc = choose_stream_shape(a->ndim, a->shape[0], row_bytes, (size_t)hbm_free_bytes, mse, true, n_state, keep,
                                       a->use_stop ? 1 : (n_total > 0 ? n_total : 128), &rows, &k, &res, a->use_stop ? 0 : n_total);
    if (rc) return rc;
    if (a->stream_resident > 0) res = std::min<int64_t>(res, a->stream_resident);
    out->rows = rows;
    out->k = k;
    out->resident_rows = res;
    const bool lean = res >= a->shape[0] && k >= 3 && !mse && (a->use_stop || n_total <= 0 || n_total / ((n_total + k - 1) / k) >= 3);
    out->hbm_bytes = ((lean ? stream_planes_all_kept(a->ndim, rows, k, true) : stream_planes(a->ndim, rows, k, mse, true)) + res * (2 + a->ndim * n_state)) * (int64_t)row_bytes;
    int64_t need = 0, avail = 0;
    (void)stream_host_need(a, res, &need, &avail);
    out->host_bytes = need;
    return TVDN_OK;
}

// One slab of a multi-process streamed run (args->slab set, stream_rows / stream_k > 0): the bytes of host memory this slab
// will page-lock and the rows it will keep resident in HBM instead.  A rank's own guard knows nothing of the other ranks on its
// host: the caller adds these up per host and refuses, on every rank alike, before any rank page-locks anything
// (cytvdn_amd/distributed.py does).
extern "C" int tvdn_slab_host_need(const tvdn_run_args *a, int64_t *need_bytes, int64_t *resident_rows)
{
    using namespace tvdn;
    TVDN_REQUIRE(a != nullptr, "args is NULL");
    int64_t kc = 0, res = 0, local_rows = 0;
    const int rc = slab_shape(a, a->stream_rows, a->stream_k, &kc, &res, &local_rows);
    if (rc) return rc;
    size_t row_bytes = a->dtype == TVDN_F32 ? 4 : 8;
    for (int i = 1; i < a->ndim; ++i) row_bytes *= (size_t)a->shape[i];
    const int n_state = a->n_fista > 0 ? 2 : 1;
    const bool want_mse = a->mse_out != nullptr && a->reference != nullptr;
    if (need_bytes) *need_bytes = (int64_t)((size_t)(2 + a->ndim * n_state + (want_mse ? 1 : 0)) * (size_t)local_rows * row_bytes);
    if (resident_rows) *resident_rows = res;
    return TVDN_OK;
}

// Which own rows of a slab stay resident and where the others sit in its packed local arrays, as arithmetic only (no device):
// local_slot[i] for own row i = its row index in the arrays the exchange hook sees (depth halo rows first), or -1 when the row
// is one of the `resident_rows` kept in HBM.  The very map the run uses (RowMap::slab_window); exported so that the host logic
// can be checked without a GPU (tests/test_host_guard_cpu.py).
extern "C" int tvdn_slab_row_map(const tvdn_run_args *a, int64_t depth, int64_t resident_rows, int64_t *local_slot)
{
    using namespace tvdn;
    TVDN_REQUIRE(a != nullptr && a->slab != nullptr && local_slot != nullptr, "NULL argument");
    const tvdn_slab_io *io = a->slab;
    const int64_t own = a->shape[0], N0 = io->global_rows;
    TVDN_REQUIRE(own >= 1 && io->row0 >= 0 && io->row0 + own <= N0 && depth >= 1 && resident_rows >= 0, "bad slab / depth / count");
    const bool periodic = a->bc_mode == TVDN_BC_PERIODIC;
    RowMap rm;
    rm.n0 = N0;
    rm.slab_window(io->row0, io->row0 + own, periodic || io->row0 > 0, periodic || io->row0 + own < N0, depth);
    TVDN_REQUIRE(resident_rows <= rm.e1 - rm.e0, "%lld rows cannot be resident: the slab has %lld interior rows", (long long)resident_rows,
                 (long long)(rm.e1 - rm.e0));
    rm.res = resident_rows;
    for (int64_t i = 0; i < own; ++i) {
        const int64_t g = io->row0 + i;
        local_slot[i] = rm.resident(g) ? -1 : depth + i - rm.res_below(g);
    }
    return TVDN_OK;
}

