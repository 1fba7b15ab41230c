// The streamed engine over several slabs: a device list inside one process (run_streamed_slabs: shared page-locked host arrays,
// one thread per slab, a barrier per pass) and one slab of a multi-process run (run_streamed_rank: packed local arrays of halo +
// host rows + halo, the caller's hooks between passes -- include/tvdn.h, tvdn_slab_io).  Both drive run_streamed (tvdn_stream.hip)
// with a SlabShare; what replaces the tiling, per-rank load and halo patching of cyTVDN/mpi.py:131-239, :314-434.
#include "tvdn_stream_parts.hpp"

namespace tvdn {

// ---- a device list whose slabs do not fit their devices: every slab streamed through its own GPU ----------------------------
// BASELINE configs[4] in structure, inside ONE process (what replaces cyTVDN/mpi.py:131-239 + :314-434 on a node: tiling,
// per-rank load, halo patching): axis 0 is cut into one slab per entry of `devices`, the state of the WHOLE cube lives in
// page-locked host arrays shared by all slabs -- two sets, a pass reads one and writes the other -- and every slab streams its
// rows through its device with the wavefront schedule, reading K rows of its neighbours' state beyond each interior face from
// those same arrays (no messages: the neighbours' rows ARE in this process's memory) and giving up a row per level there.
// One host thread per slab; all of them meet after every pass.  Sums, stopping rule and MSE trace are global.  (Across
// PROCESSES the same decomposition is cytvdn_amd.distributed.denoise_slabs(staged=...).)
int run_streamed_slabs(const tvdn_run_args *a, int64_t R, int64_t K)
{
    const auto t_start = std::chrono::steady_clock::now();
    const int world = a->n_devices;
    const int nd = a->ndim;
    const size_t item = a->dtype == TVDN_F32 ? 4 : 8;
    size_t plane = 1;
    for (int i = 1; i < nd; ++i) plane *= (size_t)a->shape[i];
    const size_t row_bytes = plane * item;
    const int64_t N0 = a->shape[0];
    const size_t cube_bytes = (size_t)N0 * row_bytes;
    const int n_total = a->n_fista + a->n_plain;
    const int n_state = a->n_fista > 0 ? 2 : 1;
    const bool want_mse = a->mse_out != nullptr && a->reference != nullptr;
    const bool periodic = a->bc_mode == TVDN_BC_PERIODIC;
    TVDN_REQUIRE(world >= 2 && world <= TVDN_MAX_DEVICES, "a streamed device list needs 2..%d entries", TVDN_MAX_DEVICES);
    TVDN_REQUIRE(N0 >= world, "axis 0 (%lld rows) cannot be cut into %d slabs", (long long)N0, world);
    TVDN_REQUIRE(R >= 1 && K >= 1, "stream_rows and stream_k must be >= 1");
    if (n_total == 0) {  // nothing to iterate: the one-device path knows what to do
        tvdn_run_args one = *a;
        one.n_devices = 0;
        one.device = a->devices[0];
        return run_streamed(&one, R, K, 0);
    }
    // The exact Jia-Zhao wrap of a non-finite first row (engine.py; upstream's Inf - Inf at the top face) needs row 0 of every
    // level of a pass on the LAST slab's device: the first slab's thread leaves those planes in a mailbox early in its pass,
    // the last slab's thread picks them up when its sweeps reach the cube's top face (what tvdn_slab_io.relay_row0 is across
    // processes).
    bool exact_wrap = false;
    if (!periodic) {
        if (a->dtype == TVDN_F32) {
            const float *p0 = (const float *)a->data;
            for (size_t i = 0; i < plane && !exact_wrap; ++i) exact_wrap = !std::isfinite(p0[i]);
        } else {
            const double *p0 = (const double *)a->data;
            for (size_t i = 0; i < plane && !exact_wrap; ++i) exact_wrap = !std::isfinite(p0[i]);
        }
    }
    const bool aliased = arrays_overlap(a->data, a->recon_out, cube_bytes);
    {   // can the host hold two sets of the state page-locked?  before anything of the caller's is touched
        const double need = (double)(1 + (aliased ? 1 : 0) + 2 + 2 * nd * n_state + (want_mse ? 1 : 0)) * (double)cube_bytes;
        const size_t avail = host_available_bytes();
        if (avail == 0 || need > 0.8 * (double)avail) {
            set_error("a streamed device list keeps two sets of the state page-locked on the host: %.0f bytes, which exceeds what the host "
                      "has available (%zu bytes, of which 80 %% are used at most)", need, avail);
            return TVDN_ERR_UNSUPPORTED;
        }
    }
    // ---- host state: the caller's arrays page-locked in place where possible, the rest from huge-page memory -------------------
    HostArr orig_h, ref_h, recon0_h;
    PinnedBuf recon1;
    std::unique_ptr<PinnedBuf[]> state(new PinnedBuf[(size_t)2 * nd * n_state]);
    int rc = aliased ? orig_h.alloc(cube_bytes) : orig_h.pin_in_place(const_cast<void *>(a->data), cube_bytes, cube_bytes, false);
    if (rc) return rc;
    if (orig_h.owned) parallel_copy(orig_h.p, a->data, cube_bytes);
    if (want_mse) {
        if ((rc = ref_h.pin_in_place(const_cast<void *>(a->reference), cube_bytes, cube_bytes, false))) return rc;
        if (ref_h.owned) parallel_copy(ref_h.p, a->reference, cube_bytes);
    }
    if ((rc = recon0_h.pin_in_place(a->recon_out, cube_bytes, cube_bytes, !aliased))) return rc;
    if ((rc = recon1.alloc(cube_bytes))) return rc;
    for (int i = 0; i < 2 * nd * n_state; ++i)
        if ((rc = state[(size_t)i].alloc(cube_bytes))) return rc;
    // the set the first pass writes: chosen so that the LAST pass lands in recon_out (unknown with a stopping rule: copied then)
    const int64_t k_eff = a->use_stop ? 1 : std::min<int64_t>({K, (int64_t)n_total, N0});
    const int n_pass = a->use_stop ? n_total : (int)((n_total + k_eff - 1) / k_eff);
    const int first_new = a->use_stop ? 1 : ((n_pass - 1) & 1);

    SlabBarrier bar;
    bar.count = world;
    std::vector<char> mail;  // row 0 of every level of a pass on its way from the first slab to the last (exact wrap)
    long mail_sent = 0;
    std::vector<long> mail_taken((size_t)world, 0);
    std::vector<SlabShare> shares((size_t)world);
    std::vector<tvdn_run_args> args((size_t)world, *a);
    std::vector<std::vector<double>> sums((size_t)world, std::vector<double>((size_t)3 * n_total, 0.0));
    std::vector<std::vector<double>> mses((size_t)world, std::vector<double>((size_t)n_total + 1, 0.0));
    std::vector<tvdn_run_stats> stats((size_t)world);
    std::vector<int32_t> iters((size_t)world, 0);
    std::vector<int32_t> phases((size_t)2 * world, 0);
    std::vector<double> stop_sums((size_t)3 * world, 0.0);
    std::vector<int> last_set((size_t)world, 0), rcs((size_t)world, 0);
    std::vector<std::string> msgs((size_t)world);
    for (int r = 0; r < world; ++r) {
        SlabShare &sh = shares[(size_t)r];
        sh.index = r;
        sh.count = world;
        sh.g0 = (int64_t)r * N0 / world;
        sh.g1 = (int64_t)(r + 1) * N0 / world;
        sh.orig = orig_h.p;
        sh.ref = want_mse ? ref_h.p : nullptr;
        sh.recon[0] = recon0_h.p;
        sh.recon[1] = recon1.p;
        for (int set = 0; set < 2; ++set)
            for (int i = 0; i < nd * n_state; ++i) sh.state[set][i] = state[(size_t)set * nd * n_state + i].p;
        sh.first_new = first_new;
        sh.barrier = &bar;
        sh.exact_wrap = exact_wrap;
        if (exact_wrap)  // row 0 of every level of a pass: from the first slab's thread to every other one's, once per pass
            sh.relay_row0 = [&bar, &mail, &mail_sent, &mail_taken, row_bytes, r](int send, void *planes, int n) -> int {
                std::unique_lock<std::mutex> lk(bar.mu);  // the barrier's lock and wake-ups: a slab that fails ends the wait
                const size_t bytes = (size_t)n * row_bytes;
                if (send) {  // (every taker of the pass before has been here: the slabs meet between two passes)
                    if (mail.size() < bytes) mail.resize(bytes);
                    std::memcpy(mail.data(), planes, bytes);
                    ++mail_sent;
                } else {
                    bar.cv.wait(lk, [&] { return mail_sent > mail_taken[(size_t)r] || bar.failed; });
                    if (bar.failed) return 1;
                    std::memcpy(planes, mail.data(), bytes);
                    ++mail_taken[(size_t)r];
                }
                bar.cv.notify_all();
                return 0;
            };
        sh.stop_sums = stop_sums.data();
        sh.last_set = &last_set[(size_t)r];
        tvdn_run_args &x = args[(size_t)r];
        x.n_devices = 0;
        x.device = a->devices[r];
        x.sums_out = sums[(size_t)r].data();
        x.mse_out = want_mse ? mses[(size_t)r].data() : nullptr;
        x.iters_run = &iters[(size_t)r];
        x.phase_iters = &phases[(size_t)2 * r];
        x.stats = &stats[(size_t)r];
        x.stream_resident = 0;
        // interior rows of a slab stay in HBM between the passes as far as they fit its share of its device (none with an MSE trace,
        // periodic boundaries, or when told not to: stream_resident == 0)
        sh.keep_rows = a->stream_resident != 0 && !want_mse && !periodic;
        sh.same_device = 0;
        for (int q = 0; q < world; ++q) sh.same_device += a->devices[q] == a->devices[r] ? 1 : 0;
        if (r != 0) x.progress = nullptr;
    }
    const auto t_threads = std::chrono::steady_clock::now();
    {
        std::vector<std::thread> th;
        for (int r = 0; r < world; ++r)
            th.emplace_back([&, r] {
                DeviceRestore restore;
                rcs[(size_t)r] = run_streamed(&args[(size_t)r], R, K, 0, &shares[(size_t)r]);
                if (rcs[(size_t)r]) {
                    msgs[(size_t)r] = tvdn_last_error();
                    bar.fail(rcs[(size_t)r], msgs[(size_t)r].c_str());
                }
            });
        for (auto &t : th) t.join();
    }
    for (int r = 0; r < world; ++r)
        if (rcs[(size_t)r] && msgs[(size_t)r].find("another slab") == std::string::npos) {  // the slab that failed first-hand
            set_error("slab %d (device %d): %s", r, a->devices[r], msgs[(size_t)r].c_str());
            return rcs[(size_t)r];
        }
    for (int r = 0; r < world; ++r)
        if (rcs[(size_t)r]) {
            set_error("%s", msgs[(size_t)r].c_str());
            return rcs[(size_t)r];
        }
    const auto t_done = std::chrono::steady_clock::now();
    // ---- results home ----------------------------------------------------------------------------------------------------------
    std::memset(a->sums_out, 0, sizeof(double) * 3 * (size_t)n_total);
    for (int r = 0; r < world; ++r)
        for (size_t i = 0; i < (size_t)3 * n_total; ++i) a->sums_out[i] += sums[(size_t)r][i];
    if (want_mse) {
        std::memset(a->mse_out, 0, sizeof(double) * ((size_t)n_total + 1));
        for (int r = 0; r < world; ++r)
            for (size_t i = 0; i <= (size_t)n_total; ++i) a->mse_out[i] += mses[(size_t)r][i];
    }
    if (last_set[0] == 1)
        parallel_copy(a->recon_out, recon1.p, cube_bytes);
    else if (recon0_h.owned)
        parallel_copy(a->recon_out, recon0_h.p, cube_bytes);
    if (a->iters_run) *a->iters_run = iters[0];
    if (a->phase_iters) {
        a->phase_iters[0] = phases[0];
        a->phase_iters[1] = phases[1];
    }
    if (a->stats) {
        tvdn_run_stats &o = *a->stats;
        std::memset(&o, 0, sizeof o);
        o.engine = TVDN_ENGINE_STREAMED;
        o.stream_rows = stats[0].stream_rows;
        o.stream_k = stats[0].stream_k;
        o.n_passes = stats[0].n_passes;
        for (int r = 0; r < world; ++r) {
            o.h2d_bytes += stats[(size_t)r].h2d_bytes;
            o.d2h_bytes += stats[(size_t)r].d2h_bytes;
            o.resident_rows += stats[(size_t)r].resident_rows;  // (interior rows the slabs kept in HBM between the passes)
        }
        o.setup_s = std::chrono::duration<double>(t_threads - t_start).count();
        o.loop_s = std::chrono::duration<double>(t_done - t_threads).count();
        o.total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    }
    return TVDN_OK;
}

// ---- one slab of a multi-process streamed run (tvdn_slab_io) ---------------------------------------------------------------
// The process-per-GPU form of run_streamed_slabs: this process holds ITS slab's state in page-locked arrays of halo + own +
// halo rows (halo = the depth of a pass), streams it through its device with the same drained passes, and between passes the
// caller's `exchange` hook refreshes the halo rows from the neighbours (cytvdn_amd/distributed.py does it with
// torch.distributed: RCCL or gloo).  The state is updated in place: a pass writes its own rows k rows behind where it reads.
int run_streamed_rank(const tvdn_run_args *a, int64_t R, int64_t K)
{
    const tvdn_slab_io *io = a->slab;
    const int nd = a->ndim;
    const size_t item = a->dtype == TVDN_F32 ? 4 : 8;
    size_t plane = 1;
    for (int i = 1; i < nd; ++i) plane *= (size_t)a->shape[i];
    const size_t row_bytes = plane * item;
    const int64_t own = a->shape[0], N0 = io->global_rows;
    const int n_total = a->n_fista + a->n_plain;
    const int n_state = a->n_fista > 0 ? 2 : 1;
    const bool want_mse = a->mse_out != nullptr && a->reference != nullptr;
    const bool periodic = a->bc_mode == TVDN_BC_PERIODIC;
    TVDN_REQUIRE(io->world >= 2 && io->rank >= 0 && io->rank < io->world, "tvdn_slab_io: rank %d of %d", io->rank, io->world);
    TVDN_REQUIRE(io->exchange != nullptr, "tvdn_slab_io.exchange is NULL");
    TVDN_REQUIRE(!a->use_stop || io->allreduce != nullptr, "tvdn_slab_io.allreduce is NULL (needed with use_stop)");
    TVDN_REQUIRE(own >= 1 && io->row0 >= 0 && io->row0 + own <= N0, "slab rows [%lld, %lld) are not inside the cube's %lld rows",
                 (long long)io->row0, (long long)(io->row0 + own), (long long)N0);
    TVDN_REQUIRE(!(io->first_row_nonfinite && !periodic) || io->relay_row0 != nullptr, "tvdn_slab_io.relay_row0 is NULL (needed when the first row is not finite)");
    if (n_total == 0) {
        if (a->recon_out != a->data) std::memmove(a->recon_out, a->data, (size_t)own * row_bytes);
        if (a->iters_run) *a->iters_run = 0;
        if (a->phase_iters) a->phase_iters[0] = a->phase_iters[1] = 0;
        return TVDN_OK;
    }
    int64_t kc = 0, res = 0, local_rows = 0;
    {
        const int rcs = slab_shape(a, R, K, &kc, &res, &local_rows);
        if (rcs) return rcs;
    }
    TVDN_REQUIRE(kc <= own, "a pass of %lld levels needs %lld rows of the neighbour's state, this slab owns %lld: stream_k must not exceed the "
                 "smallest slab's rows", (long long)kc, (long long)kc, (long long)own);
    const bool face_lo = periodic || io->row0 > 0, face_hi = periodic || io->row0 + own < N0;  // faces shared with a neighbour
    RowMap rm;  // the same map run_streamed will build: which own rows are resident
    rm.n0 = N0;
    rm.slab_window(io->row0, io->row0 + own, face_lo, face_hi, kc);
    rm.res = res;
    const size_t local_bytes = (size_t)local_rows * row_bytes;
    // What can fail on ONE rank only -- its host's memory, a page-locked allocation -- fails before the first exchange, and the
    // ranks agree on it through the all-reduce hook: a rank that returned alone would leave its peers waiting inside the
    // exchange until the communicator's timeout instead of every rank raising the same error.
    int rc_local = TVDN_OK;
    {
        const double need = (double)(2 + nd * n_state + (want_mse ? 1 : 0)) * (double)local_bytes;
        const size_t avail = host_available_bytes();
        if (avail == 0 || need > 0.8 * (double)avail) {
            set_error("this slab's state needs %.0f bytes of page-locked host memory, which exceeds what the host has available (%zu bytes, "
                      "of which 80 %% are used at most; every rank on this host asks for its own)", need, avail);
            rc_local = TVDN_ERR_UNSUPPORTED;
        }
    }
    PinnedBuf orig, recon, ref;
    std::unique_ptr<PinnedBuf[]> state(new PinnedBuf[(size_t)nd * n_state]);
    int rc = TVDN_OK;
    if (!rc_local) rc_local = orig.alloc(local_bytes);
    if (!rc_local) rc_local = recon.alloc(local_bytes);
    for (int i = 0; i < nd * n_state && !rc_local; ++i) rc_local = state[(size_t)i].alloc(local_bytes);
    if (!rc_local && want_mse) rc_local = ref.alloc(local_bytes);
    if (io->allreduce) {
        double s3[3] = {rc_local ? 1.0 : 0.0, 0.0, 0.0};
        const std::string mine = rc_local ? tvdn_last_error() : "";
        if (io->allreduce(io->user, s3)) {
            set_error("the all-reduce hook of a slab run failed (set-up status)");
            return TVDN_ERR_INVALID;
        }
        if (s3[0] > 0.0 && !rc_local) {
            set_error("%d rank(s) of this run could not set up their slab (host memory or a page-locked allocation): every rank stops", (int)s3[0]);
            return TVDN_ERR_UNSUPPORTED;
        }
        if (rc_local) set_error("%s", mine.c_str());
    }
    if (rc_local) return rc_local;
    // the own rows that live on the host <-> the caller's own-row array, run by run (local slot of own row g: kc + host rows below it)
    auto own_rows_between = [&](char *local, char *user, bool to_local) {
        for (int64_t g = io->row0; g < io->row0 + own;) {
            if (rm.resident(g)) {
                ++g;
                continue;
            }
            int64_t e = g + 1;
            while (e < io->row0 + own && !rm.resident(e)) ++e;
            char *l = local + (size_t)(kc + (g - io->row0) - rm.res_below(g)) * row_bytes, *u = user + (size_t)(g - io->row0) * row_bytes;
            parallel_copy(to_local ? l : u, to_local ? u : l, (size_t)(e - g) * row_bytes);
            g = e;
        }
    };
    own_rows_between(orig.p, (char *)const_cast<void *>(a->data), true);
    if (want_mse) own_rows_between(ref.p, (char *)const_cast<void *>(a->reference), true);
    const int64_t own_hi = kc + own - res;  // local slots [kc, own_hi): the own rows on the host; the kc outermost at either end are never resident
    {   // the data term's halo rows: once
        void *arr[1] = {orig.p};
        if (io->exchange(io->user, 1, arr, local_rows, kc, own_hi, (int32_t)kc, (int64_t)row_bytes)) {
            set_error("the exchange hook of a slab run failed (data term)");
            return TVDN_ERR_INVALID;
        }
    }
    SlabShare sh;
    sh.index = 0;
    sh.count = 1;
    sh.g0 = io->row0;
    sh.g1 = io->row0 + own;
    sh.orig = orig.p;
    sh.ref = want_mse ? ref.p : nullptr;
    sh.recon[0] = sh.recon[1] = recon.p;
    for (int i = 0; i < nd * n_state; ++i) sh.state[0][i] = sh.state[1][i] = state[(size_t)i].p;
    sh.first_new = 0;
    sh.local_rows = true;
    sh.local_v0 = io->row0;  // virtual row = K + global row, the local arrays start K rows below the first own row
    sh.resident_rows = res;
    sh.own_data = (const char *)a->data;
    sh.own_recon = (char *)a->recon_out;
    sh.exact_wrap = !periodic && io->first_row_nonfinite != 0;
    std::vector<void *> swap_arrays;
    swap_arrays.push_back(recon.p);
    for (int i = 0; i < nd * n_state; ++i) swap_arrays.push_back(state[(size_t)i].p);
    sh.before_pass = [&]() -> int {
        if (io->exchange(io->user, (int32_t)swap_arrays.size(), swap_arrays.data(), local_rows, kc, own_hi, (int32_t)kc, (int64_t)row_bytes)) {
            set_error("the exchange hook of a slab run failed");
            return TVDN_ERR_INVALID;
        }
        return TVDN_OK;
    };
    if (io->allreduce) sh.allreduce = [&](double *s3) { return io->allreduce(io->user, s3); };
    if (io->relay_row0) sh.relay_row0 = [&](int send, void *planes, int n) { return io->relay_row0(io->user, send, planes, n, (int64_t)row_bytes); };
    tvdn_run_args x = *a;
    x.shape[0] = N0;  // run_streamed sees the cube; its rows outside this slab's halo are never addressed
    x.stream_resident = 0;
    x.n_devices = 0;
    x.slab = nullptr;
    rc = run_streamed(&x, R, K, 0, &sh);
    if (rc) return rc;
    if (a->stats) a->stats->resident_rows = res;
    own_rows_between(recon.p, (char *)a->recon_out, false);  // (the resident rows went home from the device: sh.own_recon)
    return TVDN_OK;
}

}  // namespace tvdn
