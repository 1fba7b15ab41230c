// tvdn_run: the whole denoise loop on host arrays, for non-Python callers (include/tvdn.h).
// Mirrors cytvdn_amd/driver.py + engine.{SlabLayout,HipBackend,LocalSlabs} in C++: compact d-rotation state, one
// slab of axis 0 per entry of the device list, halo rows moved device-to-device (peer copies over xGMI) on a
// copy stream per slab while the interior rows are swept.  One host thread drives every device; nothing here
// needs Python, torch or RCCL.
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "tvdn_common.hpp"

#include "tvdn_run_parts.hpp"

namespace tvdn {

int run_impl(const tvdn_run_args *a, RunClock &clk, tvdn_run_stats &stats, bool pipeline_with_rule)
{
    const auto t_entry = std::chrono::steady_clock::now();
    auto since_entry = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count(); };
    const int nd = a->ndim;
    const size_t item = a->dtype == TVDN_F32 ? 4 : 8;
    size_t plane = 1;
    for (int i = 1; i < nd; ++i) plane *= (size_t)a->shape[i];
    const size_t row_bytes = plane * item;
    const int64_t N0 = a->shape[0];
    const int n_total = a->n_fista + a->n_plain;
    const bool fista = a->n_fista > 0;
    const int per_axis = fista ? 3 : 2;
    const bool want_mse = a->mse_out != nullptr && a->reference != nullptr;
    const bool periodic = a->bc_mode == TVDN_BC_PERIODIC;

    // ---- the slabs -----------------------------------------------------------------------------------
    int world = a->n_devices > 0 ? a->n_devices : 1;
    TVDN_REQUIRE(world <= TVDN_MAX_DEVICES, "n_devices = %d exceeds %d", world, TVDN_MAX_DEVICES);
    TVDN_REQUIRE(N0 >= world, "axis 0 (%lld rows) cannot be cut into %d slabs", (long long)N0, world);
    std::unique_ptr<Slab[]> sl(new Slab[world]);
    // the first call of a process: its staging lanes (pinned memory, streams) are set up beside the state's allocation, not after it
    struct Warm {
        std::thread t;
        ~Warm() { if (t.joinable()) t.join(); }
    } warm;
    {
        const int dev0 = a->n_devices > 0 ? a->devices[0] : a->device;
        if (!io_is_warm(dev0)) warm.t = std::thread([dev0] { io_warm(dev0); });
    }
    const bool ring = world > 1 && periodic;
    // Jia-Zhao across several slabs: the last slab closes the wrap of the reconstruction update with the axis-0
    // accumulator of global row 0, which is zero for finite data (TVDN_EDGE_ZERO).  If the cube's first row holds an
    // Inf or a NaN, upstream gets NaN there (anisotropic.pyx:65-73): the last slab then keeps row 0's current recon
    // as one more halo row, refreshed like any other, and forms the accumulator from it (TVDN_EDGE_WRAP).
    bool row0_bad = false;
    if (!periodic) {
        if (a->dtype == TVDN_F32) {
            const float *p0 = (const float *)a->data;
            for (size_t i = 0; i < plane && !row0_bad; ++i) row0_bad = !std::isfinite(p0[i]);
        } else {
            const double *p0 = (const double *)a->data;
            for (size_t i = 0; i < plane && !row0_bad; ++i) row0_bad = !std::isfinite(p0[i]);
        }
    }
    const bool exact_wrap = world > 1 && row0_bad;
    bool state_reused = false;  // the block the last run of this device kept: an audition's winner stays the winner
    for (int r = 0; r < world; ++r) {
        Slab &s = sl[r];
        s.device = a->n_devices > 0 ? a->devices[r] : a->device;
        s.g0 = (int64_t)r * N0 / world;
        s.g1 = (int64_t)(r + 1) * N0 / world;
        s.halo_lo = (world > 1 && (r > 0 || ring)) ? 1 : 0;
        s.halo_hi = (world > 1 && (r < world - 1 || ring || exact_wrap)) ? 1 : 0;
        TVDN_HIP(hipSetDevice(s.device));
        s.main_level = world == 1 ? +1 : 0;
        s.copy_level = world == 1 ? 0 : +1;
        int rc = TVDN_OK;
        RunKit kit;
        if (kit_acquire(s.device, s.main_level, s.copy_level, &kit)) {  // what the last run of this device left (tvdn_run_state.hip)
            s.ctx = kit.ctx;
            s.main = kit.main;
            s.copy = kit.copy;
            s.sums.p = kit.sums;
            s.sums.bytes = kit.sums_bytes;
        } else {
            rc = tvdn_ctx_create(&s.ctx, s.device);
            if (rc) return rc;
        }
        // Sweeps and transfers in different hardware-queue classes (tvdn_common.hpp make_stream).  One slab: the transfers
        // are the staging lanes' DMA copies (normal class), the sweeps go high.  Several slabs: the transfers are peer
        // copies of one row under the interior sweep, possibly done by copy kernels -- those go high, so that their few
        // workgroups are dispatched ahead of the sweep's many instead of after them, and the sweeps stay normal.
        if (!s.main) rc = make_stream(&s.main, s.main_level);
        if (!rc && !s.copy) rc = make_stream(&s.copy, s.copy_level);
        if (rc) return rc;
        TVDN_HIP(hipEventCreateWithFlags(&s.edge_done, hipEventDisableTiming));
        TVDN_HIP(hipEventCreateWithFlags(&s.halo_done, hipEventDisableTiming));
        // state: one allocation, 256-byte aligned arrays; everything but orig and recon[0] zeroed in one fill
        const size_t bytes = (size_t)s.rows() * row_bytes;
        const size_t stride = (bytes + 255) / 256 * 256 + 4096;  // staggered by 4 KiB, as engine.ARRAY_SKEW
        const int n_arr = 3 + nd * per_axis;
        s.state.device = s.device;
        if (a->workspace && world == 1) {  // the caller's device memory (tvdn_run_args.workspace): nothing allocated, nothing freed
            TVDN_REQUIRE(((uintptr_t)a->workspace & 255) == 0, "workspace must be 256-byte aligned");
            TVDN_REQUIRE(a->workspace_bytes >= (int64_t)(stride * (size_t)n_arr), "workspace of %lld bytes, the state needs %lld (tvdn_run_workspace_bytes)",
                         (long long)a->workspace_bytes, (long long)(stride * (size_t)n_arr));
            hipPointerAttribute_t at;  // a host pointer here would fault on the GPU: ask the runtime what it is
            if (hipPointerGetAttributes(&at, a->workspace) != hipSuccess || at.type != hipMemoryTypeDevice || at.device != s.device) {
                (void)hipGetLastError();
                set_error("workspace %p is not device memory of device %d", a->workspace, s.device);
                return TVDN_ERR_INVALID;
            }
            s.state.p = a->workspace;
            s.state.owned = false;
        } else if (world == 1) {
            // one slab: the block the last run of this device left behind, if it fits (StateCache above)
            // What a well-placed state is worth: 5 % of the time the run's sweeps will take (at 5.5 TB/s).  No floor since round 6:
            // with one of 0.25 s the first 50-iteration call of a process spent 0.25 s on spare granules to save at most 0.03 s of
            // sweeps (53 Gvoxel-iters/s where the second call does 88; without the extras 75: profiles/r06_first_call_by_budget.jsonl).
            // A block chosen from a short pool is topped up by the first run that can afford it (state_acquire -> dev_upgrade).
            const double sweeps_s = (double)n_total * (double)(stride * (size_t)n_arr) / 5.5e12;
            clk.mark("contexts, streams");
            TVDN_HIP(state_acquire(&s.state.p, stride * (size_t)n_arr, &s.state.bytes, s.device, &state_reused, false, std::max(0.02, 0.05 * sweeps_s)));
            clk.mark("state acquired");
            s.state.keep = true;
            stats.first_call = state_reused ? 0 : 1;
        }
    }
    // several slabs: every state on granules (tvdn_devmem.hip) -- also where neighbours on OTHER devices pull rows out of it
    // peer to peer: such a block grants every device of the list access (one descriptor each; since ABI 9).  That path has
    // not met a second GPU yet, so it proves itself before it is used: a row-sized peer copy out of every block that a
    // neighbour on another device will read is compared with what was written there (slab_peer_copy_check); a mismatch or a
    // refusal puts every slab on a plain hipMalloc block, whose peer access hipDeviceEnablePeerAccess governs, and says so on
    // stderr.  TVDN_VMM_PEER=0 takes the plain blocks at once (round 5's behaviour).
    if (world > 1) {
        bool one_device = true;
        for (int q = 0; q < world; ++q) one_device = one_device && sl[q].device == sl[0].device;
        const char *ep = getenv("TVDN_VMM_PEER");
        bool on_granules = one_device || !(ep && atoi(ep) == 0);
        for (int attempt = 0; attempt < 2; ++attempt) {
            for (int r = 0; r < world; ++r) {
                Slab &s = sl[r];
                TVDN_HIP(hipSetDevice(s.device));
                const size_t stride = ((size_t)s.rows() * row_bytes + 255) / 256 * 256 + 4096;
                int peers[TVDN_MAX_DEVICES], n_peers = 0;
                for (int q = 0; q < world; ++q) {
                    bool seen = sl[q].device == s.device;
                    for (int j = 0; j < n_peers; ++j) seen = seen || peers[j] == sl[q].device;
                    if (!seen) peers[n_peers++] = sl[q].device;
                }
                TVDN_HIP(state_malloc(&s.state.p, stride * (size_t)(3 + nd * per_axis), s.device, on_granules, 0.25, peers, n_peers));
            }
            // let every device address its neighbours' memory (no-op when they are the same device)
            for (int r = 0; r < world; ++r)
                for (int d : {(r + 1) % world, (r + world - 1) % world})
                    if (sl[d].device != sl[r].device) {
                        TVDN_HIP(hipSetDevice(sl[r].device));
                        (void)hipDeviceEnablePeerAccess(sl[d].device, 0);  // direct xGMI copies where the link exists;
                        (void)hipGetLastError();                            // without it the peer copy is staged, still correct
                    }
            bool granule_blocks = false;
            for (int r = 0; r < world; ++r) granule_blocks = granule_blocks || dev_kind(sl[r].state.p) == TVDN_MEM_GRANULES;
            const char *force = getenv("TVDN_PEER_CHECK");  // 1: also between slabs of ONE device (tests: the only kind a one-GPU box has)
            if (!granule_blocks || (one_device && !(force && atoi(force) == 1))) break;
            char why[200] = "";
            std::vector<PeerSlab> ps((size_t)world);
            for (int r = 0; r < world; ++r)
                ps[(size_t)r] = PeerSlab{sl[r].state.p, sl[r].device, sl[r].rows(), sl[r].row_lo(), sl[r].row_hi(), sl[r].halo_lo != 0, sl[r].halo_hi != 0, sl[r].copy};
            if (slab_peer_copy_check(ps.data(), world, row_bytes, nd, per_axis, why, sizeof why)) {
                stats.peer_check = 1;
                break;
            }
            stats.peer_check = -1;
            if (attempt == 1 || !on_granules) {
                set_error("peer copies between the slabs' device blocks do not arrive intact (%s)", why);
                return TVDN_ERR_HIP;
            }
            fprintf(stderr, "tvdn_run: peer copies out of blocks on granules failed their check (%s): the slabs go on plain hipMalloc blocks (TVDN_VMM_PEER=0)\n", why);
            for (int r = 0; r < world; ++r) {
                TVDN_HIP(hipSetDevice(sl[r].device));
                (void)dev_free(sl[r].state.p);
                sl[r].state.p = nullptr;
            }
            on_granules = false;
        }
    }
    for (int r = 0; r < world; ++r) {
        Slab &s = sl[r];
        TVDN_HIP(hipSetDevice(s.device));
        const size_t stride = ((size_t)s.rows() * row_bytes + 255) / 256 * 256 + 4096;
        const int n_arr = 3 + nd * per_axis;
        TVDN_HIP(hipMemsetAsync(s.state.p, 0, stride * (size_t)(n_arr - 2), s.main));
        std::memset(&s.roles, 0, sizeof s.roles);
        roles_reset(s.roles, fista);
        s.assign((char *)s.state.p, stride, nd, per_axis);
        s.sums.device = s.device;
        const size_t sums_bytes = sizeof(double) * 3 * (size_t)(n_total > 0 ? n_total : 1);
        if (s.sums.p && s.sums.bytes < sums_bytes) {  // (the kept buffer of a shorter run)
            TVDN_HIP(hipFree(s.sums.p));
            s.sums.p = nullptr;
        }
        if (!s.sums.p) {
            s.sums.bytes = std::max(sums_bytes, (size_t)65536);
            TVDN_HIP(hipMalloc(&s.sums.p, s.sums.bytes));
        }
        TVDN_HIP(hipMemsetAsync(s.sums.p, 0, sums_bytes, s.main));
    }

    clk.mark("contexts, state allocated");
    // ---- upload: own rows + halo rows, straight from the caller's array ---------------------------------------
    auto rows_to_device = [&](Slab &s, char *dst, const void *src_cube) -> int {
        TVDN_HIP(hipSetDevice(s.device));
        const char *src = (const char *)src_cube;
        // local row i holds global row (g0 - halo_lo + i) mod N0: at most three contiguous pieces
        int64_t i = 0;
        while (i < s.rows()) {
            const int64_t g = ((s.g0 - s.halo_lo + i) % N0 + N0) % N0;
            int64_t n = s.rows() - i;
            if (g + n > N0) n = N0 - g;
            int rc = tvdn_copy_to_device(dst + (size_t)i * row_bytes, src + (size_t)g * row_bytes, (size_t)n * row_bytes, s.device);
            if (rc) return rc;
            i += n;
        }
        return TVDN_OK;
    };
    for (int r = 0; r < world; ++r) {
        Slab &s = sl[r];
        tvdn_iter_args &it = s.roles.base;
        it.dtype = a->dtype;
        it.ndim = nd;
        it.shape[0] = s.rows();
        for (int i = 1; i < nd; ++i) it.shape[i] = a->shape[i];
        it.row_lo = s.row_lo();
        it.row_hi = s.row_hi();
        it.lo_mode = s.halo_lo ? TVDN_EDGE_HALO : TVDN_EDGE_BC;
        it.hi_mode = s.halo_hi ? ((exact_wrap && r == world - 1) ? TVDN_EDGE_WRAP : TVDN_EDGE_HALO)
                               : (world == 1 ? TVDN_EDGE_BC : TVDN_EDGE_ZERO);
        it.bc_mode = a->bc_mode;
        for (int q = 0; q < nd; ++q) {
            it.clip[q] = a->clip[q];
            it.lambda_mu[q] = a->lambda_mu[q];
        }
        it.orig = s.orig;
    }

    // ---- placement audition (one device, long runs) -----------------------------------------------------------------
    // The sweep's speed depends on which physical pages the state's allocation received (11.2 /
    // 12.1 / 12.6 ms for the same 60 GiB at identical clocks).  Before a run long enough to pay for it, further
    // candidates are allocated beside the first while they fit 80 % of the free HBM, each is timed for two sweeps on a
    // zero-filled state (the sweep is branch-free), the fastest is kept.  cytvdn_amd/engine.py HipBackend.best_of is the
    // same thing for engine.SlabRunner (slabs across ranks); TVDN_AUDITION=n overrides the count (1 = take the first allocation).
    if (world == 1) {
        Slab &s = sl[0];
        const char *e = getenv("TVDN_AUDITION");
        // a caller that brings the state's memory has chosen its placement: no audition then.  Nor when the block is the one
        // the last run kept (if that run auditioned, this IS its winner; allocating and freeing three more blocks of tens of
        // GiB per call is what the kept block exists to avoid) -- unless TVDN_AUDITION insists.
        // Nor on a block made of granules: its sweep time does not depend on which granules it got (tvdn_devmem.hip), so a second
        // candidate would cost an allocation and tell nothing; the audition remains for plain blocks (TVDN_VMM=0, a runtime
        // without virtual-memory management).
        const bool granules = s.state.owned && dev_kind(s.state.p) == TVDN_MEM_GRANULES;
        stats.state_mem = !s.state.owned ? TVDN_MEM_CALLER : (granules ? TVDN_MEM_GRANULES : TVDN_MEM_PLAIN);
        const int want = !s.state.owned ? 1 : (e ? atoi(e) : ((state_reused || granules) ? 1 : (n_total >= 800 ? 4 : (n_total >= 400 ? 3 : 1))));
        const size_t bytes = (size_t)s.rows() * row_bytes;
        const size_t stride = (bytes + 255) / 256 * 256 + 4096;
        const size_t total = stride * (size_t)(3 + nd * per_axis);
        TVDN_HIP(hipSetDevice(s.device));
        std::vector<std::unique_ptr<DevBuf>> held;  // candidates other than the one in s.state
        void *best = s.state.p;
        double best_ms = -1.0;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        constexpr int kProbeSweeps = 6;
        auto probe = [&](void *base, double *ms) -> int {
            s.assign((char *)base, stride, nd, per_axis);
            roles_reset(s.roles, fista);
            TVDN_HIP(hipMemsetAsync(base, 0, total, s.main));
            // one untimed sweep, then a whole cycle of the roles (period 6: three d arrays per axis x two recon buffers; which arrays
            // are written decides a sweep's time by +- 1.5 %, profiles/r06_data_dependence.jsonl): the block's time, not an arrangement's
            for (int i = 0; i < 1 + kProbeSweeps; ++i) {
                if (i == 1) TVDN_HIP(hipEventRecord(e0, s.main));
                tvdn_iter_args &it = s.roles.base;
                roles_bind(s.roles, fista, 0.5, it);
                it.sweep_lo = it.sweep_hi = 0;
                it.accumulate = 0;
                const int rc = tvdn_iterate_fused(s.ctx, &it, (double *)s.sums.p, s.main);
                if (rc) return rc;
                roles_advance(s.roles, fista, 0.5);
            }
            TVDN_HIP(hipEventRecord(e1, s.main));
            TVDN_HIP(hipEventSynchronize(e1));
            float t = 0.f;
            TVDN_HIP(hipEventElapsedTime(&t, e0, e1));
            *ms = t;
            return TVDN_OK;
        };
        if (want > 1 && n_total > 0) {
            TVDN_HIP(hipEventCreate(&e0));
            TVDN_HIP(hipEventCreate(&e1));
            int rc = probe(s.state.p, &best_ms);
            if (clk.on) fprintf(stderr, "tvdn_run:   candidate 0 at %p: %.3f ms per sweep\n", s.state.p, best_ms / kProbeSweeps);
            stats.audition_n = 1;
            stats.audition_ms[0] = best_ms / kProbeSweeps;
            for (int c = 1; c < want && !rc; ++c) {
                size_t free_b = 0, total_b = 0;
                TVDN_HIP(hipMemGetInfo(&free_b, &total_b));
                if ((double)total > 0.8 * (double)free_b) break;
                std::unique_ptr<DevBuf> b(new DevBuf);
                b->device = s.device;
                b->bytes = total;
                if (state_malloc(&b->p, total, s.device) != hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                double ms = 0.0;
                rc = probe(b->p, &ms);
                if (clk.on) fprintf(stderr, "tvdn_run:   candidate %d at %p: %.3f ms per sweep\n", c, b->p, ms / kProbeSweeps);
                if (!rc && c < 8) {
                    stats.audition_ms[c] = ms / kProbeSweeps;
                    stats.audition_n = c + 1;
                }
                if (!rc && ms < best_ms) {
                    best_ms = ms;
                    best = b->p;
                    stats.audition_kept = c < 8 ? c : 0;
                }
                held.push_back(std::move(b));
            }
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            if (rc) return rc;
            if (best != s.state.p)  // keep the winner in s.state (released with the slab), the former first one among the losers
                for (auto &b : held)
                    if (b->p == best) {
                        std::swap(b->p, s.state.p);
                        std::swap(b->bytes, s.state.bytes);  // the winner is what the cache keeps afterwards, the losers are freed
                    }
            held.clear();  // frees every loser
            s.assign((char *)s.state.p, stride, nd, per_axis);
            roles_reset(s.roles, fista);
            // as after the constructor: everything zero but orig and recon[0] (the last two arrays), which the upload below
            // fills on streams of its own -- a fill of those on s.main could land on top of the uploaded rows
            TVDN_HIP(hipMemsetAsync(s.state.p, 0, stride * (size_t)(nd * per_axis + 1), s.main));
            TVDN_HIP(hipMemsetAsync(s.sums.p, 0, sizeof(double) * 3 * (size_t)(n_total > 0 ? n_total : 1), s.main));
        }
    }

    clk.mark("placement audition");
    stats.setup_s = since_entry();
    // ---- pipelined transfers (one device, resident) ------------------------------------------------------------------------
    // The call takes host arrays and returns one (cyTVDN.py:19-31, :244-247): 2 x the cube over PCIe around the iterations --
    // 0.17 s of the 0.75 s a 50-iteration run of BASELINE config 2 takes.  Both transfers hide under iterations that do
    // not wait for the whole cube:
    //   start  the cube goes up in chunks of R rows (a helper thread, the library's pinned multi-lane staging); when chunk c
    //          has arrived, iteration level j advances rows [c R - (j+1), (c+1) R - (j+1)) for j = 0 .. k0-1: the wavefront
    //          of tvdn_stream.hip (level j+1 trails level j by one row, every row of every level swept once), here as partial
    //          sweeps on the RESIDENT arrays.  Two recon buffers and three rotating accumulator arrays per axis suffice because
    //          level j+2, which writes what level j+1 reads, stays two rows behind it;
    //   end    the last k1 iterations run the same way, and every chunk's finished rows go home (a second helper thread)
    //          while the chunks after it are still being swept.
    // The sweeps at the cube's top face take the Jia-Zhao zero (TVDN_EDGE_ZERO) because row 0 of the buffers belongs to a
    // later level by then: Jia-Zhao runs with a finite first row only; no MSE trace (it needs whole iterations); with a stopping
    // rule the start only (below).  TVDN_PIPELINE=0 keeps the plain order, "R,k0,k1" forces a shape (tests).
    int64_t pipe_R = 0;
    int pipe_k0 = 0, pipe_k1 = 0;
    // A stopping rule (round 6): the START is pipelined all the same -- the first k0 iterations follow the upload, their sums are
    // looked at when the last chunk has been swept, and the rest of the run looks at every iteration one behind (below).  Should
    // the rule have been met INSIDE those first iterations (the levels after it have run on, nothing to take back), the whole run
    // is done again in plain order (kRetryPlain: tvdn_run_entry.hip) -- a rule that fires within eight iterations of the start is
    // rare and the run it ends is short.  The END cannot be: which iteration is the last is not known until it has run.
    const bool lag_on = !(getenv("TVDN_STOP_LAG") && atoi(getenv("TVDN_STOP_LAG")) == 0);
    if (world == 1 && a->bc_mode == TVDN_BC_JIA_ZHAO && !want_mse && !row0_bad && (!a->use_stop || (pipeline_with_rule && lag_on))) {
        int32_t pl[3];
        pipeline_plan(N0, n_total, (int64_t)N0 * (int64_t)row_bytes, pl);
        pipe_R = pl[0];
        pipe_k0 = pl[1];
        pipe_k1 = a->use_stop ? 0 : pl[2];
    }
    const bool pipelined = pipe_R > 0;
    stats.pipelined = pipelined ? 1 : 0;
    struct LaneCap {  // fewer staging lanes while transfers run beside the launching thread (tvdn_hostio.hip)
        bool on;
        explicit LaneCap(bool o) : on(o) { if (on) io_cap_lanes(6); }
        ~LaneCap() { if (on) io_cap_lanes(0); }
    } lane_cap(pipelined);

    // the schedule as one list: iteration i is a FISTA iteration with ratio_of[i], or an unaccelerated one
    std::unique_ptr<double[]> ratios(new double[(size_t)(n_total > 0 ? n_total : 1)]);
    fista_ratios(a->n_fista, ratios.get());  // float64 on the host, cyTVDN.py:153-156
    for (int i = a->n_fista; i < n_total; ++i) ratios[i] = 0.0;
    auto is_fista = [&](int i) { return i < a->n_fista; };
    auto tick = [&](int slots_done) {  // the caller's progress bar (tvdn_run_args.progress)
        if (a->progress) a->progress((int32_t)slots_done, a->progress_user);
    };

    // K = kk levels from iteration `first` on, over the whole cube, chunk by chunk
    auto wavefront = [&](int first, int kk, const std::function<int(int64_t)> &before_chunk,
                         const std::function<int(int64_t)> &after_chunk) -> int {
        Slab &s = sl[0];
        std::vector<tvdn_many_args> snap((size_t)kk);
        tvdn_many_args m = s.roles;
        for (int j = 0; j < kk; ++j) {
            snap[(size_t)j] = m;
            roles_advance(m, is_fista(first + j), ratios[first + j]);
        }
        const int64_t n_chunks = (N0 + kk + pipe_R - 1) / pipe_R;
        for (int64_t c = 0; c < n_chunks; ++c) {
            int rc = before_chunk ? before_chunk(c) : TVDN_OK;
            if (rc) return rc;
            for (int j = 0; j < kk; ++j) {
                const int64_t lo = std::max<int64_t>(0, c * pipe_R - (j + 1)), hi = std::min<int64_t>(N0, (c + 1) * pipe_R - (j + 1));
                if (lo >= hi) continue;
                tvdn_iter_args it = snap[(size_t)j].base;
                roles_bind(snap[(size_t)j], is_fista(first + j), ratios[first + j], it);
                it.hi_mode = TVDN_EDGE_ZERO;
                it.sweep_lo = lo;
                it.sweep_hi = hi;
                it.accumulate = 1;
                rc = tvdn_iterate_fused(s.ctx, &it, (double *)s.sums.p + 3 * (size_t)(first + j), s.main);
                if (rc) return rc;
            }
            rc = after_chunk ? after_chunk(c) : TVDN_OK;
            if (rc) return rc;
        }
        const tvdn_iter_args keep = s.roles.base;
        s.roles = m;
        s.roles.base = keep;
        return TVDN_OK;
    };

    if (pipelined) {
        // ---- start: chunks go up on a helper thread; the first k0 iterations follow them ---------------------------------
        Slab &s = sl[0];
        TVDN_HIP(hipSetDevice(s.device));
        // (No wait for the fill of the state here, since round 6: it covers the arrays BEFORE `orig` -- the rotating arrays and
        // recon[1] -- and the rows that go up land in `orig` alone; whatever reads or writes the filled arrays is queued behind the
        // fill on s.main.  52 GiB of zeros take 10 ms: as long as the first chunk's way up, which now runs beside them.)
        const int64_t n_up = (N0 + pipe_R - 1) / pipe_R;
        std::mutex mu;
        std::condition_variable cv;
        std::vector<char> arrived((size_t)n_up, 0);
        int up_rc = TVDN_OK;
        std::string up_msg;  // the last-error text is per thread: a helper's failure is handed to the calling thread
        std::thread up([&] {
            for (int64_t c = 0; c < n_up; ++c) {
                const int64_t r0 = c * pipe_R, r1 = std::min((c + 1) * pipe_R, N0);
                const auto w0 = std::chrono::steady_clock::now();
                const int rc = tvdn_copy_to_device(s.orig + (size_t)r0 * row_bytes, (const char *)a->data + (size_t)r0 * row_bytes,
                                                   (size_t)(r1 - r0) * row_bytes, s.device);
                if (clk.on)
                    fprintf(stderr, "tvdn_run:   rows %lld..%lld up in %.2f ms\n", (long long)r0, (long long)r1,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
                std::lock_guard<std::mutex> lk(mu);
                if (rc) {
                    up_rc = rc;
                    up_msg = tvdn_last_error();
                    std::fill(arrived.begin(), arrived.end(), 1);
                } else {
                    arrived[(size_t)c] = 1;
                }
                cv.notify_all();
                if (rc) return;
            }
        });
        struct Joiner {
            std::thread &t;
            ~Joiner() { if (t.joinable()) t.join(); }
        } join_up{up};
        int rc = wavefront(0, pipe_k0, [&](int64_t c) -> int {
            if (c >= n_up) return TVDN_OK;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return arrived[(size_t)c] != 0; });
                if (up_rc) {
                    set_error("%s", up_msg.c_str());
                    return up_rc;
                }
            }
            const int64_t r0 = c * pipe_R, r1 = std::min((c + 1) * pipe_R, N0);  // recon = datacube.copy() (cyTVDN.py:145)
            TVDN_HIP(hipMemcpyAsync(s.recon(0) + (size_t)r0 * row_bytes, s.orig + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes,
                                    hipMemcpyDeviceToDevice, s.main));
            return TVDN_OK;
        }, nullptr);
        if (rc) return rc;
        up.join();
        clk.mark("upload + first iterations");
    }
    // Slabs on SEVERAL devices go up -- and come home, below -- side by side, a host thread per slab: every GPU has a link and
    // staging lanes of its own (tvdn_hostio.hip), and one after the other the eight slabs of BASELINE config 4 (8 GiB each) would
    // spend 1.3 s on the way up and as long on the way home where 0.2 s do, around 4.6 s of sweeps for 200 iterations.  One device
    // (or TVDN_SLAB_IO_THREADS=0): in turn, as before; =1 forces the threads (tests on one GPU: same bytes, the lanes take turns).
    bool io_threads = false;
    for (int r = 1; r < world; ++r) io_threads = io_threads || sl[r].device != sl[0].device;
    if (const char *e = getenv("TVDN_SLAB_IO_THREADS")) io_threads = world > 1 && atoi(e) != 0;
    auto each_slab = [&](const std::function<int(int)> &fn) -> int {  // fn(r) for every slab; the first failure's status and text
        if (!io_threads) {
            for (int r = 0; r < world; ++r) {
                const int rc = fn(r);
                if (rc) return rc;
            }
            return TVDN_OK;
        }
        std::vector<int> rcs((size_t)world, TVDN_OK);
        std::vector<std::string> msgs((size_t)world);
        std::vector<std::thread> th;
        for (int r = 0; r < world; ++r)
            th.emplace_back([&, r] {
                rcs[(size_t)r] = fn(r);
                if (rcs[(size_t)r]) msgs[(size_t)r] = tvdn_last_error();  // (the last-error text is per thread)
            });
        for (auto &t : th) t.join();
        for (int r = 0; r < world; ++r)
            if (rcs[(size_t)r]) {
                set_error("%s", msgs[(size_t)r].c_str());
                return rcs[(size_t)r];
            }
        return TVDN_OK;
    };
    if (!pipelined) {
        const int rc_up = each_slab([&](int r) -> int {
            Slab &s = sl[r];
            int rc = rows_to_device(s, s.orig, a->data);
            if (rc) return rc;
            TVDN_HIP(hipMemcpyAsync(s.recon(0), s.orig, (size_t)s.rows() * row_bytes, hipMemcpyDeviceToDevice, s.main));
            if (want_mse) {
                s.ref.device = s.mse.device = s.device;
                TVDN_HIP(hipMalloc(&s.ref.p, (size_t)s.rows() * row_bytes));
                rc = rows_to_device(s, (char *)s.ref.p, a->reference);
                if (rc) return rc;
                TVDN_HIP(hipMalloc(&s.mse.p, sizeof(double) * (size_t)(n_total + 1)));
                TVDN_HIP(hipMemsetAsync(s.mse.p, 0, sizeof(double) * (size_t)(n_total + 1), s.main));
            }
            return TVDN_OK;
        });
        if (rc_up) return rc_up;
        TVDN_HIP(hipSetDevice(sl[0].device));
    }
    if (!pipelined) clk.mark("upload");
    // sum of squared errors over the own rows of every slab, into mse[slot]
    auto sse_all = [&](int cur, int slot) -> int {
        for (int r = 0; r < world; ++r) {
            Slab &s = sl[r];
            TVDN_HIP(hipSetDevice(s.device));
            int64_t shp[4];
            shp[0] = s.g1 - s.g0;
            for (int i = 1; i < nd; ++i) shp[i] = a->shape[i];
            const size_t off = (size_t)s.row_lo() * row_bytes;
            int rc = tvdn_sum_square_error(s.ctx, a->dtype, nd, shp, (char *)s.ref.p + off, s.recon(cur) + off,
                                           (double *)s.mse.p + slot, s.main);
            if (rc) return rc;
        }
        return TVDN_OK;
    };
    if (want_mse) {
        int rc = sse_all(0, 0);  // MSE[0]: input against reference (cyTVDN.py:124-125)
        if (rc) return rc;
    }

    int ran = 0, ran_phase[2] = {0, 0};
    std::vector<int> void_slots;  // iterations that ran ahead of a stopping rule in vain: their trace entries go home as zeros
    auto cur_of = [&]() { return (int)sl[0].roles.cur; };  // every slab rotates in lockstep

    // one launch on slab s over rows [lo, hi) of its own rows (0, 0 = all)
    auto sweep = [&](Slab &s, int slot, bool use_fista, double ratio, int64_t lo, int64_t hi, bool accumulate) -> int {
        tvdn_iter_args &it = s.roles.base;
        roles_bind(s.roles, use_fista, ratio, it);
        it.sweep_lo = lo;
        it.sweep_hi = hi;
        it.accumulate = accumulate ? 1 : 0;
        TVDN_HIP(hipSetDevice(s.device));
        s.ctx->mirror = s.peek ? s.peek + 4 * (slot & 1) : nullptr;  // (a stopping rule that looks one iteration behind: below)
        const int rc = tvdn_iterate_fused(s.ctx, &it, (double *)s.sums.p + 3 * (size_t)slot, s.main);
        s.ctx->mirror = nullptr;
        return rc;
    };

    auto one = [&](int slot, bool use_fista, double ratio) -> int {
        const int nxt = cur_of() ^ 1;
        if (world == 1) {
            int r = sweep(sl[0], slot, use_fista, ratio, 0, 0, false);
            if (r) return r;
        } else {
            // edge blocks first, on every slab; then their outermost rows travel while the interiors are swept
            for (int r = 0; r < world; ++r) {
                Slab &s = sl[r];
                TVDN_HIP(hipSetDevice(s.device));
                TVDN_HIP(hipStreamWaitEvent(s.main, s.halo_done, 0));  // last iteration's halo rows have arrived
                // exact wrap: the last slab's copy stream READ my row 0 of the buffer these sweeps are about to rewrite
                // (two iterations later it is `nxt` again); nothing else orders that read before this write
                if (exact_wrap && r == 0) TVDN_HIP(hipStreamWaitEvent(s.main, sl[world - 1].halo_done, 0));
                const int64_t lo = s.row_lo(), hi = s.row_hi();
                if (hi - lo < 3) {
                    int rc = sweep(s, slot, use_fista, ratio, 0, 0, false);
                    if (rc) return rc;
                } else {
                    const int64_t e = edge_block(hi - lo);
                    int rc = sweep(s, slot, use_fista, ratio, lo, lo + e, false);
                    if (!rc) rc = sweep(s, slot, use_fista, ratio, hi - e, hi, true);
                    if (rc) return rc;
                }
                TVDN_HIP(hipEventRecord(s.edge_done, s.main));
            }
            for (int r = 0; r < world; ++r) {
                Slab &s = sl[r];
                TVDN_HIP(hipSetDevice(s.device));
                // pull the neighbours' fresh edge rows into my halo rows: my copy stream, after THEIR edge sweeps of
                // this iteration and after MY OWN (so every read of those halo rows by my previous iteration is over)
                TVDN_HIP(hipStreamWaitEvent(s.copy, s.edge_done, 0));
                if (s.halo_lo) {
                    Slab &l = sl[(r + world - 1) % world];
                    TVDN_HIP(hipStreamWaitEvent(s.copy, l.edge_done, 0));
                    TVDN_HIP(hipMemcpyPeerAsync(s.recon(nxt) + (size_t)(s.row_lo() - 1) * row_bytes, s.device,
                                                l.recon(nxt) + (size_t)(l.row_hi() - 1) * row_bytes, l.device, row_bytes, s.copy));
                }
                if (s.halo_hi) {
                    Slab &h = sl[(r + 1) % world];
                    TVDN_HIP(hipStreamWaitEvent(s.copy, h.edge_done, 0));
                    TVDN_HIP(hipMemcpyPeerAsync(s.recon(nxt) + (size_t)s.row_hi() * row_bytes, s.device,
                                                h.recon(nxt) + (size_t)h.row_lo() * row_bytes, h.device, row_bytes, s.copy));
                }
                TVDN_HIP(hipEventRecord(s.halo_done, s.copy));
                const int64_t lo = s.row_lo(), hi = s.row_hi();
                if (hi - lo >= 3) {
                    const int64_t e = edge_block(hi - lo);
                    if (lo + e < hi - e) {
                        int rc = sweep(s, slot, use_fista, ratio, lo + e, hi - e, true);
                        if (rc) return rc;
                    }
                }
            }
        }
        for (int r = 0; r < world; ++r) roles_advance(sl[r].roles, use_fista, ratio);
        ++ran;
        if (want_mse) return sse_all(cur_of(), slot + 1);
        return TVDN_OK;
    };

    auto stopped = [&](int slot, bool &stop) -> int {
        stop = false;
        if (!a->use_stop) return TVDN_OK;
        double tot[3] = {0.0, 0.0, 0.0};
        for (int r = 0; r < world; ++r) {  // the global criterion: sums over every slab
            double s3[3];
            TVDN_HIP(hipSetDevice(sl[r].device));
            TVDN_HIP(hipMemcpyAsync(s3, (double *)sl[r].sums.p + 3 * (size_t)slot, sizeof s3, hipMemcpyDeviceToHost, sl[r].main));
            TVDN_HIP(hipStreamSynchronize(sl[r].main));
            for (int j = 0; j < 3; ++j) tot[j] += s3[j];
        }
        const double delta = a->dtype == TVDN_F32 ? (double)delta_in_dtype<float>(tot) : delta_in_dtype<double>(tot);
        stop = delta < a->stop;
        return TVDN_OK;
    };

    auto run_ruled = [&](int first) -> int {
        // ---- a stopping rule, looked at one iteration behind ---------------------------------------------------------------
        // Upstream reads delta_recon[i] after iteration i and breaks (cyTVDN.py:189-195, :231-237).  Done in stream order that is a
        // round trip per iteration -- launch, sweep, fold, copy of three doubles, wake-up: 27-29 us on top of a sweep of 12 us for a
        // 64 x 64 x 256 cube and 70 us for BASELINE configs[0]'s shape (profiles/r06_stop_rule.jsonl: 2.7 x and 1.34 x the time of
        // the same run without a rule).  So iteration i+1 is queued BEFORE the sums of iteration i are looked at: the fold of an
        // iteration leaves them in host memory as well (tvdn_ctx::mirror), an event behind it says when, and the device never waits
        // for the host.  If iteration i did satisfy the rule, iteration i+1 has run in vain and is taken back: it read the state
        // iteration i left and wrote into the buffers that rotate out (recon[cur ^ 1], the oldest array of every axis), so putting
        // the roles back IS the state after iteration i, bit for bit; its sums and MSE slots are zeroed on the way home, as the
        // tails upstream leaves.  TVDN_STOP_LAG=0: the blocking form (tests compare the two).
        for (int r = 0; r < world; ++r) {
            Slab &s = sl[r];
            TVDN_HIP(hipSetDevice(s.device));
            TVDN_HIP(hipHostMalloc((void **)&s.peek, 8 * sizeof(double), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
            std::memset(s.peek, 0, 8 * sizeof(double));
            for (hipEvent_t &e : s.summed) TVDN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        auto post = [&](int slot) -> int {  // the folds of `slot` are queued on every slab: mark their end
            for (int r = 0; r < world; ++r) {
                TVDN_HIP(hipSetDevice(sl[r].device));
                TVDN_HIP(hipEventRecord(sl[r].summed[slot & 1], sl[r].main));
            }
            return TVDN_OK;
        };
        auto rule_met = [&](int slot, bool &stop) -> int {
            double tot[3] = {0.0, 0.0, 0.0};
            for (int r = 0; r < world; ++r) {  // the global criterion: sums over every slab, in slab order
                hipEvent_t ev = sl[r].summed[slot & 1];
                const auto t0 = std::chrono::steady_clock::now();
                for (;;) {  // short iterations: ask; long ones: sleep on it
                    const hipError_t e = hipEventQuery(ev);
                    if (e == hipSuccess) break;
                    if (e != hipErrorNotReady) TVDN_HIP(e);
                    (void)hipGetLastError();
                    if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(200)) {
                        TVDN_HIP(hipEventSynchronize(ev));
                        break;
                    }
                }
                const volatile double *pk = sl[r].peek + 4 * (slot & 1);
                for (int j = 0; j < 3; ++j) tot[j] += pk[j];
            }
            const double delta = a->dtype == TVDN_F32 ? (double)delta_in_dtype<float>(tot) : delta_in_dtype<double>(tot);
            stop = delta < a->stop;
            return TVDN_OK;
        };
        std::vector<tvdn_many_args> before((size_t)world);
        if (first >= n_total) return TVDN_OK;
        int slot = first;
        int rc = one(first, is_fista(first), ratios[first]);
        if (!rc) rc = post(first);
        if (rc) return rc;
        while (slot >= 0) {
            const int ahead = slot + 1 < n_total ? slot + 1 : -1;
            if (ahead >= 0) {
                for (int r = 0; r < world; ++r) before[(size_t)r] = sl[r].roles;
                rc = one(ahead, is_fista(ahead), ratios[ahead]);
                if (!rc) rc = post(ahead);
                if (rc) return rc;
            }
            ++ran_phase[is_fista(slot) ? 0 : 1];
            tick(slot + 1);
            bool st;
            rc = rule_met(slot, st);
            if (rc) return rc;
            if (!st) {
                slot = ahead;
                continue;
            }
            // a FISTA-phase stop falls through to the unaccelerated phase (cyTVDN.py:189-201); a stop there ends the run
            const int resume = (is_fista(slot) && a->n_plain > 0) ? a->n_fista : -1;
            if (ahead >= 0 && ahead != resume) {  // iteration `ahead` ran in vain
                for (int r = 0; r < world; ++r) sl[r].roles = before[(size_t)r];
                --ran;
                void_slots.push_back(ahead);
                if (resume >= 0) {
                    rc = one(resume, false, 0.0);
                    if (!rc) rc = post(resume);
                    if (rc) return rc;
                }
            }
            slot = resume;
        }
        return TVDN_OK;
    };

    // no stopping rule, one slab: nobody reads the sums before the end, small launches fold them in batches (tvdn_common.hpp)
    const bool defer_sums = world == 1 && !a->use_stop;
    if (defer_sums) sums_defer_begin(sl[0].ctx);
    bool recon_home = false;
    if (pipelined) {
        // ---- middle: whole sweeps; end: the last k1 levels as a wavefront, finished rows go home chunk by chunk ----------
        Slab &s = sl[0];
        ran = pipe_k0;
        if (pipe_k0 > 0) tick(pipe_k0);
        // While the middle iterations run, a helper page-locks the caller's RESULT array where it is (tvdn_stream.hip
        // host_pin_result): the finished rows then cross PCIe straight into it at the link's rate instead of through the pinned
        // lanes and a host copy into pages that fault in as they are written (512 MiB chunks: 9.4 ms against 14.8 ms).
        // 0 = not tried / refused (the lanes are used), 1 = page-locked.  TVDN_RESULT_LANES=1 keeps the lanes.
        int result_pinned = 0;
        std::thread pin_result;
        struct PinJoin {
            std::thread &t;
            int &pinned;
            void *p;
            ~PinJoin()
            {
                if (t.joinable()) t.join();
                if (pinned) host_unpin_result(p);
                pinned = 0;
            }
        } pin_join{pin_result, result_pinned, a->recon_out};
        if (pipe_k1 > 0 && getenv("TVDN_RESULT_LANES") == nullptr)
            pin_result = std::thread([&] {
                (void)hipSetDevice(s.device);
                result_pinned = host_pin_result(a->recon_out, (size_t)N0 * row_bytes) == TVDN_OK ? 1 : 0;
            });
        if (a->use_stop) {
            // the first k0 iterations have run behind the upload: was the rule met inside them?
            for (int i = 0; i < pipe_k0; ++i) {
                bool st;
                const int rc = stopped(i, st);
                if (rc) return rc;
                if (st) return kRetryPlain;
                ++ran_phase[is_fista(i) ? 0 : 1];
            }
            const int rc = run_ruled(pipe_k0);
            if (rc) return rc;
        } else {
            for (int i = pipe_k0; i < n_total - pipe_k1; ++i) {
                const int rc = one(i, is_fista(i), ratios[i]);
                if (rc) return rc;
                tick(i + 1);
            }
        }
        clk.mark("middle iterations");
        if (pipe_k1 > 0) {
            struct Job {
                hipEvent_t ev;
                int64_t r0, r1;
            };
            std::mutex mu;
            std::condition_variable cv;
            std::deque<Job> jobs;
            bool closed = false;
            int down_rc = TVDN_OK;
            std::string down_msg;
            const char *final_buf = s.recon(cur_of() ^ (pipe_k1 % 2));  // level j writes recon[cur ^ ((j + 1) % 2)]
            std::thread down([&] {
                if (pin_result.joinable()) pin_result.join();  // (started before the middle iterations: long done)
                hipStream_t direct = nullptr;
                if (result_pinned) {
                    (void)hipSetDevice(s.device);
                    if (hipStreamCreateWithFlags(&direct, hipStreamNonBlocking) != hipSuccess) {
                        (void)hipGetLastError();
                        direct = nullptr;
                    }
                }
                struct StreamGuard {
                    hipStream_t &st;
                    ~StreamGuard()
                    {
                        if (st) (void)hipStreamDestroy(st);
                    }
                } stream_guard{direct};
                for (;;) {
                    Job job;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return closed || !jobs.empty(); });
                        if (jobs.empty()) return;
                        job = jobs.front();
                        jobs.pop_front();
                    }
                    const auto w0 = std::chrono::steady_clock::now();
                    int rc = hipEventSynchronize(job.ev) == hipSuccess ? TVDN_OK : TVDN_ERR_HIP;  // the last level has written these rows
                    const auto w1 = std::chrono::steady_clock::now();
                    if (!rc && direct) {
                        if (hipMemcpyAsync((char *)a->recon_out + (size_t)job.r0 * row_bytes, final_buf + (size_t)job.r0 * row_bytes,
                                           (size_t)(job.r1 - job.r0) * row_bytes, hipMemcpyDeviceToHost, direct) != hipSuccess ||
                            hipStreamSynchronize(direct) != hipSuccess) {
                            set_error("download of rows %lld..%lld into the page-locked result array failed: %s", (long long)job.r0,
                                      (long long)job.r1, hipGetErrorString(hipGetLastError()));
                            rc = TVDN_ERR_HIP;
                        }
                    } else if (!rc) {
                        rc = tvdn_copy_to_host((char *)a->recon_out + (size_t)job.r0 * row_bytes, final_buf + (size_t)job.r0 * row_bytes,
                                               (size_t)(job.r1 - job.r0) * row_bytes, s.device);
                    }
                    if (clk.on)
                        fprintf(stderr, "tvdn_run:   rows %lld..%lld waited %.2f ms, copied in %.2f ms\n", (long long)job.r0, (long long)job.r1,
                                std::chrono::duration<double, std::milli>(w1 - w0).count(),
                                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w1).count());
                    (void)hipEventDestroy(job.ev);
                    if (rc) {
                        std::lock_guard<std::mutex> lk(mu);
                        if (!down_rc) down_msg = rc == TVDN_ERR_HIP && !*tvdn_last_error() ? "hipEventSynchronize failed" : tvdn_last_error();
                        down_rc = rc;
                    }
                }
            });
            struct Closer {
                std::thread &t;
                std::mutex &mu;
                std::condition_variable &cv;
                bool &closed;
                ~Closer()
                {
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        closed = true;
                    }
                    cv.notify_all();
                    if (t.joinable()) t.join();
                }
            };
            int rc;
            {
                Closer closer{down, mu, cv, closed};
                rc = wavefront(n_total - pipe_k1, pipe_k1, nullptr, [&](int64_t c) -> int {
                    const int64_t r0 = std::max<int64_t>(0, c * pipe_R - pipe_k1), r1 = std::min<int64_t>(N0, (c + 1) * pipe_R - pipe_k1);
                    if (r0 >= r1) return TVDN_OK;
                    hipEvent_t ev;
                    TVDN_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                    TVDN_HIP(hipEventRecord(ev, s.main));
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        jobs.push_back(Job{ev, r0, r1});
                    }
                    cv.notify_all();
                    return TVDN_OK;
                });
            }  // every queued row is home (or failed) here
            if (rc) return rc;
            if (down_rc) {
                set_error("%s", down_msg.c_str());
                return down_rc;
            }
            recon_home = true;
            tick(n_total);
            clk.mark("last iterations + download");
        }
        if (!a->use_stop) {
            ran = n_total;
            ran_phase[0] = a->n_fista;
            ran_phase[1] = a->n_plain;
        }
    } else if (a->use_stop && n_total > 0 && lag_on) {
        const int rc = run_ruled(0);
        if (rc) return rc;
    } else {
        for (int i = 0; i < a->n_fista; ++i) {
            int rc = one(i, true, ratios[i]);
            if (rc) return rc;
            ++ran_phase[0];
            tick(i + 1);
            bool st;
            rc = stopped(i, st);
            if (rc) return rc;
            if (st) break;
        }
        for (int j = 0; j < a->n_plain; ++j) {
            const int slot = j + a->n_fista;
            int rc = one(slot, false, 0.0);
            if (rc) return rc;
            ++ran_phase[1];
            tick(slot + 1);
            bool st;
            rc = stopped(slot, st);
            if (rc) return rc;
            if (st) break;
        }
    }

    if (defer_sums) {
        TVDN_HIP(hipSetDevice(sl[0].device));
        const int rcd = sums_defer_end(sl[0].ctx, sl[0].main);
        if (rcd) return rcd;
    }
    if (!pipelined) clk.mark("iterations");
    // ---- results home ---------------------------------------------------------------------------------
    if (n_total > 0) std::memset(a->sums_out, 0, sizeof(double) * 3 * (size_t)n_total);
    if (want_mse) std::memset(a->mse_out, 0, sizeof(double) * (size_t)(n_total + 1));
    std::unique_ptr<double[]> tmp(new double[3 * (size_t)(n_total > 0 ? n_total : 1) + (size_t)n_total + 1]);
    {
        const int cur_home = cur_of();
        const int rc_home = each_slab([&](int r) -> int {  // every slab's rows, side by side where the slabs sit on several devices
            Slab &s = sl[r];
            TVDN_HIP(hipSetDevice(s.device));
            TVDN_HIP(hipStreamSynchronize(s.main));
            TVDN_HIP(hipStreamSynchronize(s.copy));
            return recon_home ? TVDN_OK
                              : tvdn_copy_to_host((char *)a->recon_out + (size_t)s.g0 * row_bytes, s.recon(cur_home) + (size_t)s.row_lo() * row_bytes,
                                                  (size_t)(s.g1 - s.g0) * row_bytes, s.device);
        });
        if (rc_home) return rc_home;
    }
    for (int r = 0; r < world; ++r) {
        Slab &s = sl[r];
        TVDN_HIP(hipSetDevice(s.device));
        int rc = TVDN_OK;
        if (n_total > 0) {
            rc = tvdn_copy_to_host(tmp.get(), s.sums.p, sizeof(double) * 3 * (size_t)n_total, s.device);  // (never the runtime's path for pageable memory: tvdn_hostio.hip)
            if (rc) return rc;
            for (int i = 0; i < 3 * n_total; ++i) a->sums_out[i] += tmp[i];
        }
        if (want_mse) {
            rc = tvdn_copy_to_host(tmp.get(), s.mse.p, sizeof(double) * (size_t)(n_total + 1), s.device);
            if (rc) return rc;
            for (int i = 0; i <= n_total; ++i) a->mse_out[i] += tmp[i];
        }
    }
    for (int v : void_slots) {
        for (int j = 0; j < 3; ++j) a->sums_out[3 * (size_t)v + (size_t)j] = 0.0;
        if (want_mse) a->mse_out[v + 1] = 0.0;
    }
    if (a->iters_run) *a->iters_run = ran;
    if (a->phase_iters) {
        a->phase_iters[0] = ran_phase[0];
        a->phase_iters[1] = ran_phase[1];
    }
    clk.mark("results home");
    stats.engine = TVDN_ENGINE_RESIDENT;
    stats.loop_s = since_entry() - stats.setup_s;  // upload, iterations, download (overlapped or not)
    for (int r = 0; r < world; ++r) {
        stats.h2d_bytes += (int64_t)sl[r].rows() * (int64_t)row_bytes * (want_mse ? 2 : 1);
        stats.d2h_bytes += (int64_t)(sl[r].g1 - sl[r].g0) * (int64_t)row_bytes;
    }
    return TVDN_OK;
}

}  // namespace tvdn
