// tvdn_run for a cube whose state does not fit the HBM of its device: the out-of-core wavefront schedule in C++
// (include/tvdn.h, tvdn_run_args.stream_rows / stream_k) -- the ONE streamed engine of the tree (one device, a device list,
// a rank of a multi-process run): the state lives in pinned host memory, streams through the GPU once per pass in chunks of R rows, and iteration
// level j+1 trails level j by one row, so every row of every level is swept exactly once and crosses PCIe once per
// k iterations.  Every level keeps a ring of R+2 rows per array in HBM (tvdn_iter_args.ring_rows); the sweeps are
// tvdn_iterate_fused launches, so the bits are those of the resident engine.  Upstream has no counterpart: its
// arrays never leave the host (cyTVDN/cyTVDN.py:148-242 is the loop this replaces for cubes beyond HBM).
//
// Map of the engine (round 5 cut the one 2400-line file of round 4 into these; every part is host code around
// tvdn_iterate_fused launches and copies):
//   tvdn_stream_parts.hpp   page-locked host memory (PinnedBuf, HostArr, StateBlocks, the host-memory guard), rings, RowMap (which
//                           rows stay resident), SlabShare / SlabBarrier (what a slab shares with its coordinator)
//   tvdn_stream_plan.hip    what a shape costs (stream_planes, stream_device_bytes), what a run chooses (choose_stream_shape),
//                           tvdn_stream_plan / tvdn_stream_host_need / tvdn_slab_host_need / tvdn_slab_row_map, host_pin_result
//   tvdn_stream_run.hpp     StreamRun: the state of one streamed run on one device and its named steps
//   tvdn_stream.hip         (this file) run_streamed = set_up -> schedule -> finish
//   tvdn_stream_pass.hip    StreamRun::pass: one drained pass (periodic cubes, slabs): upload chunk c + 1, scatter into level 0, K sweeps
//                           trailing each other by a row, gather level K, download; exchange / all-reduce / row-0 hooks
//   tvdn_stream_chain.hip   StreamRun::chain: Jia-Zhao on one device, several passes stacked into one running row index, downloads by a
//                           copy kernel, the last pass sending resident rows' results home itself
//   tvdn_stream_slabs.hip   run_streamed_slabs (a device list: shared host arrays, one thread per slab) and run_streamed_rank (one
//                           process per GPU: packed local arrays, the caller's hooks between passes)
#include "tvdn_stream_run.hpp"

namespace tvdn {

// R rows per chunk, K iteration levels per pass; `res_req` rows keep their state in HBM between passes (-1: as many as fit
// beside the rings in 85 % of the free HBM, 0: none).
int run_streamed(const tvdn_run_args *a, int64_t R, int64_t K, int64_t res_req, const SlabShare *sh)
{
    StreamRun run(a, R, K, res_req, sh);
    return run.run();
}

int StreamRun::run()
{
    bool nothing_to_do = false;
    int rc = set_up(nothing_to_do);
    if (rc || nothing_to_do) return rc;
    if ((rc = schedule())) return rc;
    return finish();
}

void StreamRun::pack_host_rows(char *packed, char *cube, bool to_packed)
{
    for (int64_t g = 0; g < N0;) {
        if (resident(g)) {
            ++g;
            continue;
        }
        int64_t e = g + 1;
        while (e < N0 && !resident(e)) ++e;
        char *pk = packed + (size_t)rm.host_below(g) * row_bytes, *cb = cube + (size_t)g * row_bytes;
        parallel_copy(to_packed ? pk : cb, to_packed ? cb : pk, (size_t)(e - g) * row_bytes);
        g = e;
    }
}

int StreamRun::wait_staged(int64_t upto)
{
    while (true) {
        const int64_t v = staged_upto.load();
        if (v < 0) return staged_done.wait();
        if (v >= upto) return TVDN_OK;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
}

int StreamRun::meet()  // slabs of a device-list run: every pass ends at the barrier (a failed slab releases the others)
{
    if (!sh || !sh->barrier) return TVDN_OK;
    const int rcb = sh->barrier->arrive_and_wait();
    if (rcb) set_error("another slab of this run failed: %s", sh->barrier->msg.c_str());
    return rcb;
}

int StreamRun::stop_after(int slot, bool &stop)
{
    double s3[3];
    TVDN_HIP(hipMemcpy(s3, (double *)sums_d.p + 3 * (size_t)slot, sizeof s3, hipMemcpyDeviceToHost));
    if (sh && sh->allreduce) {  // a slab of its own process: the caller adds the slabs up
        const int rca = sh->allreduce(s3);
        if (rca) {
            set_error("the all-reduce hook of a slab run failed (status %d)", rca);
            return TVDN_ERR_INVALID;
        }
    } else if (sh) {  // the global criterion: the sums of every slab (written between two meetings, read between the next two)
        for (int j = 0; j < 3; ++j) sh->stop_sums[3 * sh->index + j] = s3[j];
        int rcb = meet();
        if (rcb) return rcb;
        s3[0] = s3[1] = s3[2] = 0.0;
        for (int r = 0; r < sh->count; ++r)
            for (int j = 0; j < 3; ++j) s3[j] += sh->stop_sums[3 * r + j];
        if ((rcb = meet())) return rcb;
    }
    const double delta = a->dtype == TVDN_F32 ? (double)((float)s3[1] / (float)s3[2]) : s3[1] / s3[2];
    stop = delta < a->stop;
    return TVDN_OK;
}

// ---- set-up: what fits where, host state, device block ---------------------------------------------------------------------------
int StreamRun::set_up(bool &nothing_to_do)
{
    t_start = std::chrono::steady_clock::now();
    nd = a->ndim;
    item = a->dtype == TVDN_F32 ? 4 : 8;
    plane = 1;
    for (int i = 1; i < nd; ++i) plane *= (size_t)a->shape[i];
    row_bytes = plane * item;
    N0 = a->shape[0];
    cube_bytes = (size_t)N0 * row_bytes;
    n_total = a->n_fista + a->n_plain;
    fista = a->n_fista > 0;
    n_state = fista ? 2 : 1;
    want_mse = a->mse_out != nullptr && a->reference != nullptr;
    device = a->n_devices > 0 ? a->devices[0] : a->device;
    periodic = a->bc_mode == TVDN_BC_PERIODIC;
    TVDN_REQUIRE(periodic || a->bc_mode == TVDN_BC_JIA_ZHAO, "the streamed tvdn_run handles bc_mode 0 and 2");
    TVDN_REQUIRE(R >= 1 && K >= 1, "stream_rows and stream_k must be >= 1");
    TVDN_REQUIRE(res_req >= -1, "stream_resident must be -1 (as many rows as fit), 0 (none) or a row count");
    aliased = arrays_overlap(a->data, a->recon_out, cube_bytes);
    if (n_total == 0) {  // nothing to iterate: recon = datacube.copy() (cyTVDN.py:145)
        if (a->recon_out != a->data) std::memmove(a->recon_out, a->data, cube_bytes);
        if (want_mse) {  // MSE[0] on the host in f64 (a corner nobody times)
            double t = 0.0;
            for (size_t i = 0; i < (size_t)N0 * plane; ++i) {
                const double d = a->dtype == TVDN_F32 ? (double)((const float *)a->data)[i] - (double)((const float *)a->reference)[i]
                                                      : ((const double *)a->data)[i] - ((const double *)a->reference)[i];
                t += d * d;
            }
            a->mse_out[0] = t;
        }
        if (a->iters_run) *a->iters_run = 0;
        if (a->phase_iters) a->phase_iters[0] = a->phase_iters[1] = 0;
        if (a->stats) {
            std::memset(a->stats, 0, sizeof *a->stats);
            a->stats->engine = TVDN_ENGINE_STREAMED;
        }
        nothing_to_do = true;
        return TVDN_OK;
    }
    if (a->use_stop) K = 1;  // the stopping rule needs a decision after every iteration: one level per pass
    K = std::min<int64_t>(K, std::max<int64_t>(1, n_total));
    if (periodic || sh) K = std::min<int64_t>(K, N0);  // (the wrapped rows / the neighbours' rows a pass reads are rows of this cube)
    // as many passes as this depth needs, of (almost) equal depth: 80 iterations at k = 38 are three PCIe round trips whether
    // they hold 38 + 38 + 4 levels or 27 + 27 + 26, and the shallower rings leave HBM for resident rows.  The deeper passes
    // come first and consecutive depths differ by one level at most (what the out boxes of chained passes are sized for).
    n_pass_plan = a->use_stop ? n_total : (int)((n_total + K - 1) / K);
    if (!a->use_stop) K = depth_of_pass(0);
    // Periodic boundaries along axis 0: the sweeps see a virtual cube of N0 + 2 K rows -- the cube between K wrapped rows
    // at either end, which are each other's halo -- and, as at the face between two slabs, give up one row per level at
    // the two artificial faces; the wrap itself is never swept.
    // A slab of a device-list run (sh): the same virtual rows -- the cube between K rows at either end -- of which this slab
    // owns [own0, own1); at its interior faces it reads K rows of its neighbours' state (shared host arrays) and gives up a row
    // per level, as a periodic run does at both ends.
    KX = (periodic || sh) ? K : 0, NV = N0 + 2 * KX, G0 = KX, G1 = KX + N0;
    own0 = sh ? KX + sh->g0 : G0, own1 = sh ? KX + sh->g1 : G1;  // virtual rows whose results and sums are this run's
    art_lo = periodic || (sh && sh->g0 > 0), art_hi = periodic || (sh && sh->g1 < N0);  // faces that are not the cube's own
    if (sh) res_req = sh->local_rows ? sh->resident_rows : (sh->keep_rows ? -1 : 0);  // (a rank: what its coordinator sized; a device list: what fits)
    TVDN_HIP(hipSetDevice(device));

    // ---- what fits where: rings and boxes first, then as many resident rows as asked for / as fit ---------------------------
    cap = R + 2, ocap = R + K + 3;
    n_in = 2 + nd * n_state + (want_mse ? 1 : 0), n_out = 1 + nd * n_state, n_store = 2 + nd * n_state;
    ring_b = aligned((size_t)cap * row_bytes), oring_b = aligned((size_t)ocap * row_bytes);
    box_b = aligned((size_t)R * row_bytes), plane_b = aligned(row_bytes);
    // an out box holds R + 1 rows: at the seam between two chained passes whose depths differ by one level (80 iterations in
    // three passes: 27 + 27 + 26) the last rows of one and the first rows of the next come down in the same chunk
    obox_b = aligned((size_t)(R + 1) * row_bytes);
    dev_bytes_max = stream_device_bytes(nd, n_state, want_mse, R, K, row_bytes, nullptr);
    size_t free_b = 0, total_b = 0;
    TVDN_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += state_kept_bytes(device);  // the block the last run kept is this run's to take over or to release
    if (const size_t cap_b = env_bytes("TVDN_HBM_LIMIT"))  // what the planner of the Python side counts on (tests: "48G")
        free_b = std::min(free_b, cap_b);
    // RES of the N0 rows keep their state (data term, recon, accumulators: n_store arrays) in HBM between passes: they enter
    // the rings and leave them by device copies instead of crossing PCIe.  Jia-Zhao runs without an MSE trace (the periodic
    // schedule walks a wrapped virtual cube whose ends are both streamed; the reference cube of an MSE trace stays on the host).
    rm.n0 = N0;
    // Every row kept and every pass at least three levels deep: the lean layout (tvdn_stream_plan.hip stream_planes_all_kept) --
    // rings for the levels between the first and the last only, no boxes, no data-term ring; the rows are swept in place
    // (tvdn_stream_chain.hip).  Asked for by stream_resident < 0 (as many as fit) or >= the cube's rows.
    lean_layout = false;
    {
        const char *e_inplace = getenv("TVDN_STREAM_INPLACE"), *e_res = getenv("TVDN_STREAM_RESIDENT"), *e_lean = getenv("TVDN_STREAM_LEAN");
        const bool wanted = !sh && !periodic && !want_mse && !a->use_stop && K >= 3 && (res_req < 0 || res_req >= N0) && !e_res &&
                            !(e_inplace && atoi(e_inplace) == 0) && !(e_lean && atoi(e_lean) == 0);
        if (wanted) {
            // every pass at least three levels deep: 80 iterations asked at K = 3 would be 26 passes of 3 and one of 2 -- one pass
            // fewer makes them 2 x 4 + 24 x 3 (nothing crosses PCIe between the passes of such a run: their number is free)
            int passes = n_pass_plan;
            while (passes > 1 && n_total / passes < 3) --passes;
            const int64_t k_lean = (n_total + passes - 1) / passes;
            const size_t lean_b = stream_device_bytes_all_kept(nd, R, k_lean, row_bytes);
            const size_t need_b = lean_b + (size_t)N0 * (size_t)n_store * plane_b;
            if (n_total / passes >= 3 && need_b <= (size_t)((res_req < 0 ? 0.85 : 0.92) * (double)free_b)) {
                lean_layout = true;
                n_pass_plan = passes;
                K = k_lean;  // (= depth_of_pass(0); nothing computed so far depends on K in a run that is neither periodic nor a slab)
                dev_bytes_max = lean_b;
                rm.res = N0;
            }
        }
    }
    if (lean_layout) {
        // (nothing else to decide)
    } else if (dev_bytes_max > free_b) {
        set_error("streamed run with %lld-row chunks and k = %lld needs %zu bytes of HBM, device %d has %zu free", (long long)R,
                  (long long)K, dev_bytes_max, device, free_b);
        return TVDN_ERR_UNSUPPORTED;
    }
    if (sh) {  // a slab keeps none of the rows its neighbours read
        rm.slab_window(sh->g0, sh->g1, art_lo, art_hi, K);
    }
    if (sh && sh->local_rows && res_req > 0) {  // a slab of a multi-process run: its coordinator has sized the packed local arrays for exactly this
        TVDN_REQUIRE(!want_mse && res_req <= rm.e1 - rm.e0, "a slab cannot keep %lld rows resident (%lld interior rows; none with an MSE trace)",
                     (long long)res_req, (long long)(rm.e1 - rm.e0));
        const size_t need_b = dev_bytes_max + (size_t)res_req * (size_t)n_store * plane_b;
        if (need_b > (size_t)(0.92 * (double)free_b)) {
            set_error("streamed slab with %lld resident rows needs %zu bytes of HBM, device %d has %zu free", (long long)res_req, need_b, device, free_b);
            return TVDN_ERR_UNSUPPORTED;
        }
        rm.res = res_req;
    } else if (lean_layout) {
        // (all of them: decided above)
    } else if (!periodic && !want_mse && res_req != 0) {
        auto fits = [&](double share) -> int64_t {
            const size_t lim = (size_t)(share * (double)free_b / (double)(sh ? std::max(1, sh->same_device) : 1));  // (slabs of a device list share their device)
            return lim > dev_bytes_max ? (int64_t)((lim - dev_bytes_max) / ((size_t)n_store * plane_b)) : 0;
        };
        const int64_t may_keep = sh ? std::max<int64_t>(0, rm.e1 - rm.e0) : N0;  // (a slab: none of the rows its neighbours read)
        rm.res = std::min<int64_t>(may_keep, res_req < 0 ? fits(0.85) : std::min<int64_t>(res_req, fits(0.92)));
        if (const char *e = getenv("TVDN_STREAM_RESIDENT")) rm.res = std::max<int64_t>(0, std::min<int64_t>({(int64_t)atoll(e), may_keep, fits(0.92)}));
    }
    RES = rm.res, HR = N0 - RES;  // rows in HBM / rows on the host
    if (getenv("TVDN_RUN_TIMING"))
        fprintf(stderr, "[tvdn_run streamed] device %d: %.2f GiB free (%.2f of them the kept block), rings and boxes %.2f GiB, %lld rows asked resident, %lld kept\n", device,
                (double)free_b / 1073741824.0, (double)state_kept_bytes(device) / 1073741824.0, (double)dev_bytes_max / 1073741824.0, (long long)res_req, (long long)RES);
    // BEFORE anything of the caller's is touched: can the host hold what stays there?  (page-locked: it cannot swap)
    {
        int64_t need = 0, avail = 0;
        int rc0 = sh ? TVDN_OK : stream_host_need(a, RES, &need, &avail);  // (slabs: the coordinator has asked for all of them)
        if (rc0 && g_releases_pending.load() > 0) {  // pinned memory a previous run of this process is still handing back
            wait_for_releases();
            rc0 = stream_host_need(a, RES, &need, &avail);
        }
        if (rc0) return rc0;
    }

    // Jia-Zhao wrap at the top face: exact (TVDN_EDGE_WRAP, row 0 of every level kept aside) when row 0 is not finite
    exact_wrap = false;
    if (sh) {
        exact_wrap = !periodic && sh->exact_wrap;  // (the coordinator has looked: a slab may not hold the cube's first row)
    } else if (periodic) {
        // the wrap is swept for real on the extended cube
    } else if (a->dtype == TVDN_F32) {
        const float *p0 = (const float *)a->data;
        for (size_t i = 0; i < plane && !exact_wrap; ++i) exact_wrap = !std::isfinite(p0[i]);
    } else {
        const double *p0 = (const double *)a->data;
        for (size_t i = 0; i < plane && !exact_wrap; ++i) exact_wrap = !std::isfinite(p0[i]);
    }

    // ---- host state, made available by helper threads while the first pass runs ----------------------------------------------
    // The HR rows that live on the host, of: the data term (the caller's `data` page-locked in place where possible), recon
    // (`recon_out`, idem), the reference of an MSE trace, and the accumulator state in blocks of rows (StateBlocks).  The
    // first pass needs the data term only (recon starts as a copy of it and the accumulators as zeros, cyTVDN.py:131-145:
    // both are formed on the device), so it starts as soon as that is reachable; recon and the state blocks are needed when
    // the first rows come back down, K rows later.  Periodic runs keep old and new state in two sets (second recon: recon2_h).
    // `data` may be the very array the result goes to (the resident run allows it too): the passes then write recon rows
    // over the rows a later pass would upload as the data term, which therefore gets a pinned copy of its own.
    two_sets = periodic || sh != nullptr;  // old and new host state apart
    n_sets = two_sets ? 2 : 1;
    for (int s = 0; s < n_sets; ++s) {
        sb[s].n_arr = nd * n_state;
        sb[s].n_slots = HR;
        sb[s].row_bytes = row_bytes;
        // blocks of ~4 GiB (all arrays together), never shorter than a chunk: short enough for the first one to exist when the
        // first rows come down, long enough for the per-allocation costs not to matter
        const int64_t per_row = (int64_t)sb[s].n_arr * (int64_t)row_bytes;
        sb[s].block_rows = std::max<int64_t>({R, 1, (int64_t)((int64_t(4) << 30) / std::max<int64_t>(per_row, 1))});
        sb[s].block_rows = std::min<int64_t>(sb[s].block_rows, std::max<int64_t>(HR, 1));
        sb[s].blocks.resize((size_t)sb[s].n_blocks());
    }
    host_bytes = (size_t)HR * row_bytes;
    // data partly overlapping recon_out (not the same array): a download into recon_out may hit rows of `data` that have
    // not been read yet, so every input is taken out of `data` before the first pass starts
    const bool eager = (aliased && a->recon_out != a->data) || getenv("TVDN_STREAM_EAGER") != nullptr;

    if (sh) {  // a slab of a device-list run: the host state is the coordinator's (shared, page-locked, cube rows)
        orig_h.p = sh->orig;
        recon_h.p = sh->recon[0];
        recon2_h.p = sh->recon[1];
        ref_h.p = sh->ref;
        orig_h.cube_rows = recon_h.cube_rows = recon2_h.cube_rows = ref_h.cube_rows = true;
        for (int set = 0; set < 2; ++set)
            for (int i = 0; i < nd * n_state; ++i) sb[set].flat.push_back(sh->state[set][i]);
    }
    // (The helper that page-locks the host arrays starts BEFORE the device block is asked for: a hipMalloc of most of the HBM
    //  takes 0.3 - 1.7 s when the driver has freed memory to clear first, time in which the data term gets page-locked.)
    pinner = std::thread([this, eager] {
        (void)eager;
        (void)hipSetDevice(device);
        if (HR <= 0 || sh) {
            orig_ready.raise();
            recon_ready.raise();
            recon2_ready.raise();
            return;
        }
        // 1. the data term (and the reference): inputs of the first pass
        int rcp = aliased ? orig_h.alloc(host_bytes) : orig_h.pin_in_place(const_cast<void *>(a->data), cube_bytes, host_bytes, false);
        if (!rcp && orig_h.owned) pack_host_rows(orig_h.p, (char *)const_cast<void *>(a->data), true);
        if (!rcp && want_mse) {
            rcp = ref_h.pin_in_place(const_cast<void *>(a->reference), cube_bytes, host_bytes, false);
            if (!rcp && ref_h.owned) pack_host_rows(ref_h.p, (char *)const_cast<void *>(a->reference), true);
        }
        const std::string m1 = rcp ? tvdn_last_error() : "";
        orig_ready.raise(rcp, m1);
        // 2. where the first pass's rows come down: recon (periodic: the SECOND set) and the state blocks in slot order
        if (!rcp) {
            if (periodic) {
                rcp = recon2_h.alloc(host_bytes);
                recon2_ready.raise(rcp, rcp ? tvdn_last_error() : "");
            } else {
                // (a result array that is also the input holds data: not `fresh`)
                rcp = recon_h.pin_in_place(a->recon_out, cube_bytes, host_bytes, !aliased);
                recon_ready.raise(rcp, rcp ? tvdn_last_error() : "");
                recon2_ready.raise();
            }
        }
        const int first_set = periodic ? 1 : 0;
        for (int64_t b = 0; b < sb[first_set].n_blocks() && !rcp; ++b) rcp = sb[first_set].allocate(b);
        if (periodic && !rcp) {  // 3. the first set: the second pass's target
            rcp = recon_h.pin_in_place(a->recon_out, cube_bytes, host_bytes, !aliased);
            recon_ready.raise(rcp, rcp ? tvdn_last_error() : "");
            for (int64_t b = 0; b < sb[0].n_blocks() && !rcp; ++b) rcp = sb[0].allocate(b);
        }
        if (rcp) {  // nobody waits for ever
            const std::string m = m1.empty() ? std::string(tvdn_last_error()) : m1;
            recon_ready.raise(rcp, m);
            recon2_ready.raise(rcp, m);
            for (int s = 0; s < n_sets; ++s) sb[s].fail(rcp, m);
        }
    });

    // ---- device: rings, staging boxes, resident rows, sums ---------------------------------------------------------------
    int rc = tvdn_ctx_create(&ctx.c, device);
    if (rc) return rc;
    if ((rc = make_stream(&st.main, +1))) return rc;  // three queue classes: no false ordering between sweeps, uploads and
    if ((rc = make_stream(&st.up, 0))) return rc;     // downloads whatever other streams the process holds (tvdn_common.hpp)
    if ((rc = make_stream(&st.down, -1))) return rc;
    ring_bytes = dev_bytes_max - (exact_wrap ? 0 : 2 * (size_t)(K + 1) * plane_b);
    store_b = aligned((size_t)std::max<int64_t>(RES, 1) * row_bytes);
    dev_bytes = ring_bytes + (RES > 0 ? (size_t)n_store * store_b : 0);
    // the one big device block: the block the last run of this device kept, if it fits (tvdn_run.hip state_acquire; a kept
    // block of another size is released first, so a streamed run still has the whole HBM to itself)
    mem.device = device;
    t_before_block = since(t_start);
    block_reused = false;
    // (the rings of the levels are swept like a resident state, many streams at once: the same allocator, tvdn_devmem.hip)
    TVDN_HIP(state_acquire(&mem.p, dev_bytes, &mem.bytes, device, &block_reused, true,
                           std::max(0.25, 0.05 * (double)n_total * (double)N0 * (double)row_bytes * (double)(3 + 3 * nd) / 5.5e12)));
    block_kind = dev_kind(mem.p);  // granules or a plain block (tvdn_devmem.hip)
    t_block = since(t_start);
    TVDN_HIP(hipMemsetAsync(mem.p, 0, ring_bytes, st.main));
    cursor = (char *)mem.p;
    Rw.assign((size_t)K + 1, Ring{});
    Aw.assign((size_t)(K + 2) * nd, Ring{});  // [level + 1][axis]
    if (!lean_layout) {
        for (Ring &r : Rw) r = Ring{take(ring_b), cap, row_bytes};
        for (Ring &r : Aw) r = Ring{take(ring_b), cap, row_bytes};
        Ow = Ring{take(oring_b), ocap, row_bytes};
        if (want_mse) Fw = Ring{take(oring_b), ocap, row_bytes};
        for (int h = 0; h < 2; ++h) {
            for (int i = 0; i < n_in; ++i) inbox[h][i] = take(box_b);
            for (int i = 0; i < n_out; ++i) outbox[h][i] = take(obox_b);
        }
    } else {  // rings of the levels 1 .. K-1 only; the others are never touched (a null base would show at once)
        for (Ring &r : Rw) r = Ring{nullptr, cap, row_bytes};
        for (Ring &r : Aw) r = Ring{nullptr, cap, row_bytes};
        Ow = Ring{nullptr, ocap, row_bytes};
        for (int64_t j = 1; j <= K - 1; ++j) {
            Rw[(size_t)j].base = take(ring_b);
            for (int q = 0; q < nd; ++q) A(j, q).base = take(ring_b);
        }
    }
    zero_plane = take(plane_b);  // the accumulator state a run starts from (cyTVDN.py:131-145): the first pass uploads none
    row0b_base = nullptr;
    if (exact_wrap) {
        for (int64_t j = 0; j <= K; ++j) row0.push_back(take(plane_b));
        row0b_base = take((size_t)(K + 1) * plane_b);
    }
    // resident rows: array i of the store holds them packed (slot = resident rows below): 0 data term, 1 recon, 2 + q * n_state + s state
    store.assign((size_t)n_store, nullptr);
    if (RES > 0)
        for (int i = 0; i < n_store; ++i) store[(size_t)i] = take(store_b);

    // one slot per iteration, and a last one that takes the sums of halo rows (periodic: the wrapped rows are swept too)
    TVDN_HIP(hipMalloc(&sums_d.p, sizeof(double) * 3 * (size_t)(n_total + 1)));
    TVDN_HIP(hipMemsetAsync(sums_d.p, 0, sizeof(double) * 3 * (size_t)(n_total + 1), st.main));
    discard = n_total;
    // squared errors per (slot, row): summed in row order on the host at the end
    if (want_mse) {
        TVDN_HIP(hipMalloc(&mse_d.p, sizeof(double) * (size_t)(n_total + 1) * (size_t)N0));
        TVDN_HIP(hipMemsetAsync(mse_d.p, 0, sizeof(double) * (size_t)(n_total + 1) * (size_t)N0, st.main));
    }
    for (int h = 0; h < 2; ++h)
        if ((rc = evs.make(&in_ready[h])) || (rc = evs.make(&in_free[h])) || (rc = evs.make(&out_ready[h])) || (rc = evs.make(&out_free[h]))) return rc;
    for (hipEvent_t &e : pump_ready)
        if ((rc = evs.make(&e))) return rc;

    stager = std::thread([this] {  // resident rows of the data term: pageable `data` -> store, through the library's pinned lanes
        if (sh && RES <= 0) {
            staged_upto.store(N0);
            staged_done.raise();
            return;
        }
        // (a slab of a multi-process run: the caller's array holds its OWN rows only, row g at + (g - first own row))
        const char *data_rows = !sh ? (const char *)a->data : (sh->local_rows ? sh->own_data - (size_t)sh->g0 * row_bytes : sh->orig);
        const int64_t piece = std::max<int64_t>(1, (int64_t)((size_t(1) << 30) / row_bytes));
        for (int64_t g = 0; g < N0 && RES > 0;) {
            if (!resident(g)) {
                ++g;
                continue;
            }
            int64_t e = g + 1;
            while (e < N0 && e - g < piece && resident(e)) ++e;
            const int rcs = tvdn_copy_to_device(store_row(0, g), data_rows + (size_t)g * row_bytes, (size_t)(e - g) * row_bytes, device);
            if (rcs) {
                staged_upto.store(-1);
                staged_done.raise(rcs, tvdn_last_error());
                return;
            }
            g = e;
            staged_upto.store(g);
        }
        staged_upto.store(N0);
        staged_done.raise();
    });
    if (eager) {
        if ((rc = staged_done.wait()) || (rc = orig_ready.wait())) return rc;
    }

    h_old = sh ? sh->first_new ^ 1 : 0;  // which set holds the current state (two sets: periodic runs, slabs); 0 = recon_h / sb[0]
    // row of a host array by cube row g / virtual row v (a slab of its own process addresses its local arrays by v)
    local_rows = sh && sh->local_rows;

    std::memset(&it, 0, sizeof it);
    it.dtype = a->dtype;
    it.ndim = nd;
    it.shape[0] = NV;
    for (int i = 1; i < nd; ++i) it.shape[i] = a->shape[i];
    it.row_lo = periodic ? 0 : G0;  // periodic: the virtual cube with its wrapped rows; Jia-Zhao: the cube's own faces
    it.row_hi = periodic ? NV : G1;
    it.lo_mode = TVDN_EDGE_BC;
    it.hi_mode = periodic ? TVDN_EDGE_BC : (exact_wrap ? TVDN_EDGE_WRAP : TVDN_EDGE_ZERO);
    it.bc_mode = a->bc_mode;
    it.accumulate = 1;
    it.ring_rows = cap;
    it.orig_ring_rows = ocap;
    it.orig = Ow.base;
    for (int q = 0; q < nd; ++q) {
        it.clip[q] = a->clip[q];
        it.lambda_mu[q] = a->lambda_mu[q];
    }
    one_row[0] = 1;
    for (int i = 1; i < nd; ++i) one_row[i] = a->shape[i];

    if (exact_wrap && sh && sh->relay_row0 && (rc = row0_host.alloc((size_t)(K + 1) * row_bytes))) return rc;

    // ---- one pass: `kk` iteration levels over the whole cube ------------------------------------------------------------
    d_form = fista;
    tk_prev = 0.0;
    done = 0;
    bytes_up = 0, bytes_down = 0, n_passes = 0;  // across PCIe (tvdn_run_stats)
    return TVDN_OK;
}

int StreamRun::schedule()
{
    int rc = TVDN_OK;
    // ---- the schedule: FISTA ratios in float64 on the host (cyTVDN.py:153-156), then the unaccelerated tail -------------
    ratios.assign((size_t)n_total, 0.0);
    fista_ratios(a->n_fista, ratios.data());
    for (int i = a->n_fista; i < n_total; ++i) ratios[i] = NAN;
    ran = 0, ran_phase[0] = a->n_fista, ran_phase[1] = a->n_plain;
    TVDN_HIP(hipStreamSynchronize(st.main));
    t_passes = std::chrono::steady_clock::now();
    first_pass_s = 0.0;
    // Jia-Zhao runs go through `chain`: all passes at once when they can be chained (K <= N0 - 3 R, no stopping rule;
    // TVDN_STREAM_CHAIN=0 keeps them apart), else one at a time.  Periodic runs keep the drained `pass` above.
    // Chaining pays where the link is the bound and the run is long: with every row streamed, 4 passes of 38 levels over 64
    // planes of 256 MiB run at 40.2 Gvoxel-iters/s chained against 36.2 drained (3.9 against 4.9 s per further pass); 2 passes
    // tie (38.4 / 39.4: the first pass uploads little, the last one only drains); and a run that keeps most rows resident is
    // bound by its sweeps, which the download kernel disturbs (51.6 against 57.0).  profiles/r04_stream_rates.jsonl.
    // Default: chain from 3 passes on when no row is resident; TVDN_STREAM_CHAIN=1 / 0 forces it on (where possible) / off.
    bool want_chain = RES == 0 && n_pass_plan >= 3 && !sh;
    if (const char *e = getenv("TVDN_STREAM_CHAIN")) want_chain = atoi(e) != 0;
    if (lean_layout) want_chain = false;  // (one pass at a time: what "in place" is safe for)
    const bool can_chain = want_chain && !periodic && !a->use_stop && n_pass_plan > 1 && K <= N0 - 3 * R;
    down_blocks = can_chain ? 8 : 0;
    if (const char *e = getenv("TVDN_STREAM_DOWN_BLOCKS")) down_blocks = std::max(0, atoi(e));
    // ... since round 5 by the runtime's copies one at a time from a helper thread instead (DownPump, tvdn_stream_parts.hpp): three
    // chained passes of 49 levels over 64 rows of 256 MiB planes 13.35 -> 9.75 s (47.3 -> 64.8 Gvoxel-iters/s), 55 GB/s down beside
    // 42 up while rows come down, the sweeps undisturbed.  TVDN_STREAM_DOWN_PUMP=0 keeps the copy kernel (or, with
    // TVDN_STREAM_DOWN_BLOCKS=0, the runtime's copies queued behind events).
    down_pump = true;  // (also the drained passes of periodic cubes, slabs of a device list and ranks: tvdn_stream_pass.hip)
    if (const char *e = getenv("TVDN_STREAM_DOWN_PUMP")) down_pump = atoi(e) != 0;
    if (getenv("TVDN_STREAM_DOWN_BLOCKS")) down_pump = false;
    if (down_pump) down_blocks = 0;
    if (!periodic && exact_wrap)
        for (int64_t j = 0; j <= K; ++j) row0b.push_back(row0b_base + (size_t)j * plane_b);
    const bool drained_pass = periodic || sh != nullptr;  // pass(): two sets of host state, artificial faces
    if (!a->use_stop) {
        if (!drained_pass) {
            std::vector<PassDesc> all;
            for (int i = 0; i < n_total;) {  // a pass may hold the last FISTA iterations and the first unaccelerated ones
                const int kk = depth_of_pass((int)all.size());
                all.emplace_back();
                if ((rc = describe(i, kk, ratios.data() + i, all.back()))) return rc;
                i += kk;
            }
            if (can_chain) {
                if ((rc = chain(all))) return rc;
                first_pass_s = 0.0;  // chained passes overlap: none of them can be timed alone
                if (a->progress) a->progress((int32_t)n_total, a->progress_user);
            } else {
                for (size_t q = 0; q < all.size(); ++q) {
                    std::vector<PassDesc> one(1, all[q]);
                    one[0].first = q == 0;
                    if ((rc = chain(one))) return rc;
                    if (q == 0) first_pass_s = since(t_passes);
                    if (a->progress) a->progress((int32_t)(all[q].it0 + all[q].kk), a->progress_user);
                }
            }
            ran = n_total;
        } else {
            for (int i = 0, q = 0; i < n_total; ++q) {
                const int kk = depth_of_pass(q);
                if ((rc = pass(ratios.data() + i, kk))) return rc;
                if ((rc = meet())) return rc;  // (slabs) every slab has written its rows of the new set
                if (i == 0) first_pass_s = since(t_passes);
                ran += kk;
                i += kk;
                if (a->progress) a->progress((int32_t)i, a->progress_user);
            }
        }
    } else {
        // one level per pass; phases as upstream runs them (cyTVDN.py:148-242): an early stop ends the FISTA phase, the
        // unaccelerated phase still runs, writing its sums from its own slot on
        bool very_first = true;
        for (int phase = 0; phase < 2; ++phase) {
            const int first = phase == 0 ? 0 : a->n_fista, last = phase == 0 ? a->n_fista : n_total;
            done = first;
            ran_phase[phase] = 0;
            for (int i = first; i < last; ++i) {
                if (drained_pass) {
                    if ((rc = pass(ratios.data() + i, 1))) return rc;
                    if ((rc = meet())) return rc;
                } else {
                    std::vector<PassDesc> one(1);
                    if ((rc = describe(i, 1, ratios.data() + i, one[0]))) return rc;
                    one[0].first = very_first;
                    if ((rc = chain(one))) return rc;
                }
                very_first = false;
                ++ran;
                ++ran_phase[phase];
                if (a->progress) a->progress((int32_t)(i + 1), a->progress_user);
                bool stop;
                if ((rc = stop_after(i, stop))) return rc;
                if (stop) break;
            }
        }
    }
    t_end_passes = std::chrono::steady_clock::now();
    return TVDN_OK;

}

int StreamRun::finish()
{
    int rc = TVDN_OK;
    // ---- results home -------------------------------------------------------------------------------------------------------
    if (stager.joinable()) stager.join();
    if (pinner.joinable()) pinner.join();
    for (int64_t g = 0; g < N0 && RES > 0 && !recon_direct;) {  // resident rows: store -> the caller's array, through the library's pinned lanes
        if (!resident(g)) {
            ++g;
            continue;
        }
        int64_t e = g + 1;
        while (e < N0 && resident(e)) ++e;
        // (a rank: the caller's own-row array; a slab of a device list: the set of the shared arrays its last pass wrote)
        char *home = !sh ? (char *)a->recon_out : (sh->local_rows ? sh->own_recon - (size_t)sh->g0 * row_bytes : (h_old ? sh->recon[1] : sh->recon[0]));
        rc = tvdn_copy_to_host(home + (size_t)g * row_bytes, store_row(1, g), (size_t)(e - g) * row_bytes, device);
        if (rc) return rc;
        g = e;
    }
    if (sh) {
        if (sh->last_set) *sh->last_set = h_old;  // the coordinator brings the result home (run_streamed_slabs / run_streamed_rank)
    } else if (HR > 0) {
        const HostArr &last = (periodic && h_old == 1) ? recon2_h : recon_h;  // periodic: the set the last pass wrote
        if (last.owned) pack_host_rows(last.p, (char *)a->recon_out, false);
        else if (last.p != (char *)a->recon_out) parallel_copy(a->recon_out, last.p, cube_bytes);
    }
    // (the caller's arrays and the heap never through the runtime's path for pageable memory: tvdn_hostio.hip transfer)
    if (n_total > 0 && (rc = tvdn_copy_to_host(a->sums_out, sums_d.p, sizeof(double) * 3 * (size_t)n_total, device))) return rc;
    if (want_mse) {
        std::vector<double> per_row((size_t)(n_total + 1) * (size_t)N0);
        if ((rc = tvdn_copy_to_host(per_row.data(), mse_d.p, sizeof(double) * per_row.size(), device))) return rc;
        for (int s = 0; s <= n_total; ++s) {
            double t = 0.0;
            for (int64_t g = 0; g < N0; ++g) t += per_row[(size_t)s * (size_t)N0 + (size_t)g];
            a->mse_out[s] = t;
        }
    }
    if (a->iters_run) *a->iters_run = ran;
    if (a->phase_iters) {
        a->phase_iters[0] = ran_phase[0];
        a->phase_iters[1] = ran_phase[1];
    }
    const double home_s = since(t_end_passes);
    // the device block goes back (to the cache) BEFORE the host state starts to be unpinned in the background: a hipFree issued
    // behind dozens of hipHostUnregister calls waits for them
    auto t_mark = std::chrono::steady_clock::now();
    double td[5];
    auto lap = [&](int i) {
        td[i] = since(t_mark);
        t_mark = std::chrono::steady_clock::now();
    };
    mem.release();
    lap(0);
    sums_d.release();
    mse_d.release();
    st.release();
    ctx.release();
    lap(1);
    orig_h.release();
    recon_h.release();
    ref_h.release();
    recon2_h.release();
    lap(2);
    {
        std::vector<std::unique_ptr<PinnedBuf>> all;
        for (int s = 0; s < n_sets; ++s) {
            for (auto &b : sb[s].blocks)
                if (b) all.push_back(std::move(b));
            sb[s].blocks.clear();
        }
        PinnedBuf::release_in_background(std::move(all));
    }
    lap(3);
    if (getenv("TVDN_STREAM_TIMING"))  // measurement aid: set-up apart from the passes
        fprintf(stderr, "tvdn_run streamed: set-up in detail: context and streams %.3f s, device block of %.1f GiB %s %.3f s, the rest (events, "
                "helper threads, waiting for the first inputs) %.3f s\n", t_before_block, (double)dev_bytes / 1073741824.0,
                block_reused ? "reused" : "allocated", t_block - t_before_block, std::chrono::duration<double>(t_passes - t_start).count() - t_block);
    if (getenv("TVDN_STREAM_TIMING"))
        fprintf(stderr, "tvdn_run streamed: rows %lld k %lld resident rows %lld of %lld, set-up %.3f s, passes %.3f s (first %.3f s), results home %.3f s; "
                "released: device block %.3f s, sums/streams/context %.3f s, caller's arrays unpinned %.3f s, host state handed to the background %.3f s\n",
                (long long)R, (long long)K, (long long)RES, (long long)N0, std::chrono::duration<double>(t_passes - t_start).count(),
                std::chrono::duration<double>(t_end_passes - t_passes).count(), first_pass_s, home_s, td[0], td[1], td[2], td[3]);
    if (a->stats) {
        tvdn_run_stats &s = *a->stats;
        std::memset(&s, 0, sizeof s);
        s.engine = TVDN_ENGINE_STREAMED;
        s.stream_rows = (int32_t)R;
        s.stream_k = (int32_t)K;
        s.resident_rows = RES;
        s.n_passes = n_passes;
        s.h2d_bytes = bytes_up + RES * (int64_t)row_bytes;
        s.d2h_bytes = bytes_down + (recon_direct ? 0 : RES * (int64_t)row_bytes);
        s.setup_s = std::chrono::duration<double>(t_passes - t_start).count();
        s.loop_s = std::chrono::duration<double>(t_end_passes - t_passes).count();
        s.total_s = since(t_start);
        s.first_pass_s = first_pass_s;
        s.first_pass_iters = (int32_t)depth_of_pass(0);
        s.results_under_last_pass = recon_direct ? 1 : 0;
        s.state_mem = block_kind;
        s.kept_in_place = inplace_kind;
    }
    return TVDN_OK;
}

}  // namespace tvdn
