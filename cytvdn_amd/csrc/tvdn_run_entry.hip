// tvdn_run, the C entry (include/tvdn.h): argument checks, the choice between the resident engine (tvdn_run.hip), the streamed one
// (tvdn_stream*.hip) and a rank's slab, and what is kept for the next call.  Split from tvdn_run.hip in round 6 (no file of csrc/ above
// 1000 lines).
#include <algorithm>
#include <cstdlib>
#include <memory>

#include "tvdn_run_parts.hpp"

extern "C" int tvdn_run(const tvdn_run_args *a)
{
    tvdn::DeviceRestore restore;  // declared first: destroyed after every slab, stream and buffer of the run
    TVDN_REQUIRE(a != nullptr, "args is NULL");
    TVDN_REQUIRE(a->dtype == TVDN_F32 || a->dtype == TVDN_F64, "bad dtype %d", a->dtype);
    TVDN_REQUIRE(a->ndim == 3 || a->ndim == 4, "ndim must be 3 or 4, got %d", a->ndim);
    for (int i = 0; i < a->ndim; ++i) TVDN_REQUIRE(a->shape[i] >= 1, "shape[%d] must be >= 1", i);
    TVDN_REQUIRE(a->n_fista >= 0 && a->n_plain >= 0, "negative iteration count");
    TVDN_REQUIRE(a->data && a->recon_out, "data / recon_out is NULL");
    TVDN_REQUIRE(a->sums_out || a->n_fista + a->n_plain == 0, "sums_out is NULL");
    TVDN_REQUIRE(a->n_devices >= 0 && a->n_devices <= TVDN_MAX_DEVICES, "n_devices must be 0..%d", TVDN_MAX_DEVICES);
    const bool stream_auto = a->stream_rows == -1 && a->stream_k == -1;
    TVDN_REQUIRE(stream_auto || (a->stream_rows >= 0 && a->stream_k >= 0 && (a->stream_rows > 0) == (a->stream_k > 0)),
                 "stream_rows and stream_k must both be 0 (never stream), both be -1 (stream when needed) or both be positive");
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
            tvdn::set_error("no HIP device visible: the product path needs an MI355X (gfx950); there is no CPU fallback");
            return TVDN_ERR_NO_DEVICE;
        }
        if (a->n_devices == 0) TVDN_REQUIRE(a->device >= 0 && a->device < n, "device %d out of range (0..%d)", a->device, n - 1);
        for (int i = 0; i < a->n_devices; ++i)
            TVDN_REQUIRE(a->devices[i] >= 0 && a->devices[i] < n, "devices[%d] = %d out of range (0..%d)", i, a->devices[i], n - 1);
    }
    if (a->bc_mode == TVDN_BC_MIRROR) {
        tvdn::set_error("bc_mode 1 (mirror) reconstruction update reads out of bounds upstream (utils.pyx:117-120): unsupported");
        return TVDN_ERR_UNSUPPORTED;
    }
    TVDN_REQUIRE(a->bc_mode == 0 || a->bc_mode == 2, "bc_mode must be 0 or 2, got %d", a->bc_mode);
    if (a->slab) {  // one slab of a multi-process streamed run: the caller's hooks carry what crosses process boundaries
        TVDN_REQUIRE(a->stream_rows > 0 && a->stream_k > 0, "a slab of a multi-process run (tvdn_run_args.slab) is streamed: stream_rows and stream_k must be positive");
        TVDN_REQUIRE(a->n_devices <= 1, "a slab of a multi-process run uses one device");
        return tvdn::run_streamed_rank(a, a->stream_rows, a->stream_k);
    }
    if (a->stream_rows > 0) {
        const auto ts = std::chrono::steady_clock::now();
        // several devices: every slab streamed through its own GPU from host arrays all of them share (tvdn_stream.hip)
        const int rcs = a->n_devices > 1 ? tvdn::run_streamed_slabs(a, a->stream_rows, a->stream_k)
                                         : tvdn::run_streamed(a, a->stream_rows, a->stream_k, a->stream_resident);
        if (getenv("TVDN_STREAM_TIMING"))
            fprintf(stderr, "tvdn_run streamed: whole call %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - ts).count());
        return rcs;
    }
    {   // say clearly when the slabs cannot fit, instead of failing somewhere inside hipMalloc
        const int world = a->n_devices > 0 ? a->n_devices : 1;
        TVDN_REQUIRE(a->shape[0] >= world, "axis 0 (%lld rows) cannot be cut into %d slabs", (long long)a->shape[0], world);
        tvdn_plan_out pl;
        const int rc = tvdn_plan(a->dtype, a->ndim, a->shape, a->n_fista > 0, world, a->n_devices > 0 ? a->devices[0] : a->device, &pl);
        if (rc) return rc;
        int same = 0;  // slabs sharing the first device share its HBM
        for (int i = 0; i < world; ++i) same += (a->n_devices == 0 || a->devices[i] == a->devices[0]) ? 1 : 0;
        // ONE threshold (tvdn_plan's `fits`): a state beyond 90 % of the free HBM streams when asked to decide, else is refused
        // a caller that brings the state's memory (workspace, one slab) has nothing left to fit but the sums
        const bool brought = a->workspace != nullptr && world == 1;
        const bool over = !brought && pl.bytes_per_slab * same > (int64_t)(0.9 * (double)pl.free_bytes);
        if (stream_auto && over && world == 1) {
            // asked to decide: one device, state beyond its HBM -> stream it (tvdn_stream.hip; it refuses, before it
            // touches the caller's arrays, what the host cannot hold either)
            size_t row_bytes = a->dtype == TVDN_F32 ? 4 : 8;
            for (int i = 1; i < a->ndim; ++i) row_bytes *= (size_t)a->shape[i];
            int64_t rows = 0, k = 0, res = 0;
            const bool keep = a->bc_mode == TVDN_BC_JIA_ZHAO && !(a->mse_out && a->reference) && a->stream_resident != 0;
            const int rc2 = tvdn::choose_stream_shape(a->ndim, a->shape[0], row_bytes, (size_t)pl.free_bytes, a->mse_out && a->reference, true,
                                                      a->n_fista > 0 ? 2 : 1, keep, a->use_stop ? 1 : a->n_fista + a->n_plain, &rows, &k, &res,
                                                      a->use_stop ? 0 : a->n_fista + a->n_plain);
            if (rc2) return rc2;
            return tvdn::run_streamed(a, rows, k, a->stream_resident > 0 ? a->stream_resident : (keep ? res : 0));
        }
        if (stream_auto && over && world > 1) {
            // asked to decide, several devices, slabs beyond their HBM: BASELINE configs[4] in structure -- every slab streamed
            // through its own device (the depth that one slab's rings allow; no rows kept resident: the halo rows of a pass are
            // read from the shared host arrays)
            size_t row_bytes = a->dtype == TVDN_F32 ? 4 : 8;
            for (int i = 1; i < a->ndim; ++i) row_bytes *= (size_t)a->shape[i];
            int64_t rows = 0, k = 0, res = 0;
            const int rc2 = tvdn::choose_stream_shape(a->ndim, (a->shape[0] + world - 1) / world, row_bytes, (size_t)(pl.free_bytes / same),
                                                      a->mse_out && a->reference, false, a->n_fista > 0 ? 2 : 1, false,
                                                      a->use_stop ? 1 : a->n_fista + a->n_plain, &rows, &k, &res,
                                                      a->use_stop ? 0 : a->n_fista + a->n_plain);
            if (rc2) return rc2;
            return tvdn::run_streamed_slabs(a, rows, k);
        }
        if (over) {
            tvdn::set_error("state of %lld bytes per slab x %d slab(s) on device %d exceeds 90 %% of its %lld free bytes of HBM: use more "
                            "devices (fewest slabs that fit one each: %d) or the streamed engines (stream_rows / stream_k; cytvdn_amd.plan_run)",
                            (long long)pl.bytes_per_slab, same, a->n_devices > 0 ? a->devices[0] : a->device,
                            (long long)pl.free_bytes, pl.min_slabs);
            return TVDN_ERR_UNSUPPORTED;
        }
    }
    if ((a->n_devices > 1) || a->workspace) (void)tvdn_release_cache();  // several slabs / the caller's own memory: no use for a kept block
    tvdn::RunClock clk;
    const auto t0 = std::chrono::steady_clock::now();
    tvdn_run_stats stats;
    std::memset(&stats, 0, sizeof stats);
    int rc = tvdn::run_impl(a, clk, stats);
    if (rc == tvdn::kRetryPlain) {  // a stopping rule met inside the iterations that followed the upload (tvdn_run.hip): plain order
        std::memset(&stats, 0, sizeof stats);
        rc = tvdn::run_impl(a, clk, stats, false);
    }
    clk.mark("release");
    if (!rc && a->stats) {
        stats.total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        *a->stats = stats;
    }
    return rc;
}
