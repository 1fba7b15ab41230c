// What a tvdn_run keeps and what it costs: the state block kept between runs (one per device), the arithmetic of check_memory for
// HBM (tvdn_plan), the shape of a resident run's pipelined transfers, and the small C entries around them (include/tvdn.h).
// Split from tvdn_run.hip in round 5 (no file of csrc/ above 1000 lines).
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "tvdn_common.hpp"

namespace tvdn {

// The state block of the last resident run of each device is KEPT when the run ends and handed to the next run that it
// fits: releasing and re-allocating tens of GiB in quick succession costs 0.7 s per hipMalloc plus 1.1 s per hipFree, with
// single stalls of 4-6 s (profiles/r03_malloc_stall_probe.jsonl), so a program that calls tvdn_run cube after cube would
// spend more time in the allocator than in the sweeps.  One block per device at most, only while no caller-provided
// workspace is in use; tvdn_release_cache() hands it back, TVDN_KEEP_STATE=0 never keeps one.
struct StateCache {
    std::mutex mu;
    void *p[TVDN_MAX_DEVICES] = {};
    size_t bytes[TVDN_MAX_DEVICES] = {};
};
static StateCache g_state_cache;

static bool keep_state()
{
    const char *e = getenv("TVDN_KEEP_STATE");
    return !(e && atoi(e) == 0) && getenv("TVDN_MALLOC") == nullptr;
}

// The state's allocation: composed from physical granules (tvdn_devmem.hip: the placement of such a block does not decide how
// fast the sweep runs on it, that of a hipMalloc block does).  `granules` false: a plain hipMalloc block
// (slabs on several devices, whose neighbours copy rows out of it peer to peer).  TVDN_MALLOC=contiguous | uncached |
// finegrained asks the runtime for another kind of device memory (measurement, round 3: they draw from the same lottery).
hipError_t state_malloc(void **p, size_t bytes, int device, bool granules, double spread_budget_s, const int *peers, int n_peers)
{
    const char *e = getenv("TVDN_MALLOC");
    if (e && !strcmp(e, "contiguous")) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocContiguous);
    if (e && !strcmp(e, "uncached")) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocUncached);
    if (e && !strcmp(e, "finegrained")) return hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained);
    if (!granules) return hipMalloc(p, bytes);
    DevAllocInfo info;
    const hipError_t rc = dev_alloc(p, bytes, device, nullptr, spread_budget_s, &info, peers, n_peers);
    if (rc == hipSuccess && getenv("TVDN_RUN_TIMING") && info.granules)
        fprintf(stderr, "tvdn_run:   state on %d granules of %lld MiB, a random subset of %d created, in %.3f s (budget for the extra ones %.2f s)\n", info.granules,
                (long long)(info.granule_bytes >> 20), info.pool, info.seconds, spread_budget_s);
    return rc;
}

// the kept block of `device` if it holds `bytes` without being more than a quarter larger (`any_larger`: however much larger -- a
// streamed run carves rings out of it and is indifferent to where they lie), else a fresh allocation
hipError_t state_acquire(void **p, size_t bytes, size_t *got_bytes, int device, bool *reused, bool any_larger, double spread_budget_s)
{
    *reused = false;
    if (keep_state() && device >= 0 && device < TVDN_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_state_cache.mu);
        void *&c = g_state_cache.p[device];
        size_t &cb = g_state_cache.bytes[device];
        if (c && cb >= bytes && (any_larger || cb - bytes <= bytes / 4)) {
            *p = c;
            *got_bytes = cb;
            c = nullptr;
            cb = 0;
            *reused = true;
            // a block whose granules a short first run chose from next to nothing: this run may be able to afford better
            const hipError_t eu = dev_upgrade(p, device, spread_budget_s);
            if (eu == hipSuccess) return hipSuccess;
            (void)hipGetLastError();  // (the block has been given back: allocate below)
            *reused = false;
        }
        if (c) {  // the wrong size: a block on granules keeps the granules it has and is dealt out anew at the right one (creating
                  // them is what a big block's set-up consists of: 2.7 s for 236 GiB, 5-7 s behind a release); else make room first
            void *q = c;
            const hipError_t er = dev_resize(&q, bytes, device, spread_budget_s);
            c = nullptr;
            cb = 0;
            if (er == hipSuccess) {
                *p = q;
                *got_bytes = bytes;
                return hipSuccess;
            }
            if (er == hipErrorNotSupported) (void)dev_free(q);  // (any other error: dev_resize has given everything back)
            (void)hipGetLastError();
        }
    }
    *got_bytes = bytes;
    return state_malloc(p, bytes, device, true, spread_budget_s);
}

size_t state_kept_bytes(int device)
{
    if (device < 0 || device >= TVDN_MAX_DEVICES) return 0;
    std::lock_guard<std::mutex> lk(g_state_cache.mu);
    return g_state_cache.bytes[device];
}

void state_release(void *p, size_t bytes, int device)
{
    if (!p) return;
    if (keep_state() && device >= 0 && device < TVDN_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_state_cache.mu);
        void *&c = g_state_cache.p[device];
        if (!c) {
            c = p;
            g_state_cache.bytes[device] = bytes;
            return;
        }
    }
    (void)dev_free(p);
}

// ---- the kit of a resident run, kept like its state ------------------------------------------------------------------------
struct KitCache {
    std::mutex mu;
    RunKit kit[TVDN_MAX_DEVICES];
    bool held[TVDN_MAX_DEVICES] = {};
};
static KitCache g_kit_cache;

static void kit_destroy(int device, RunKit &k)
{
    (void)hipSetDevice(device);
    if (k.main) (void)hipStreamDestroy(k.main);
    if (k.copy) (void)hipStreamDestroy(k.copy);
    if (k.sums) (void)hipFree(k.sums);
    if (k.ctx) (void)tvdn_ctx_destroy(k.ctx);
    k = RunKit();
}

bool kit_acquire(int device, int main_level, int copy_level, RunKit *k)
{
    if (!keep_state() || device < 0 || device >= TVDN_MAX_DEVICES) return false;
    std::lock_guard<std::mutex> lk(g_kit_cache.mu);
    if (!g_kit_cache.held[device]) return false;
    RunKit &c = g_kit_cache.kit[device];
    if (c.main_level != main_level || c.copy_level != copy_level) return false;  // (a device list after one-device runs: its own streams)
    *k = c;
    c = RunKit();
    g_kit_cache.held[device] = false;
    return true;
}

void kit_release(int device, RunKit &k)
{
    if (!k.ctx) return;
    // whatever the run left in flight (an error path) is over before anyone else uses these streams or the scratch behind the context
    bool idle = hipSetDevice(device) == hipSuccess;
    idle = idle && (!k.main || hipStreamSynchronize(k.main) == hipSuccess) && (!k.copy || hipStreamSynchronize(k.copy) == hipSuccess);
    if (!idle) (void)hipGetLastError();
    if (idle && k.main && k.copy && keep_state() && device >= 0 && device < TVDN_MAX_DEVICES) {
        // as tvdn_ctx_create leaves a context: not deferring, not timing, no mirror
        k.ctx->deferring = false;
        k.ctx->n_pend = 0;
        k.ctx->timing = false;
        k.ctx->mirror = nullptr;
        for (auto &ev : k.ctx->events) {
            (void)hipEventDestroy(ev.first);
            (void)hipEventDestroy(ev.second);
        }
        k.ctx->events.clear();
        std::lock_guard<std::mutex> lk(g_kit_cache.mu);
        if (!g_kit_cache.held[device]) {
            g_kit_cache.kit[device] = k;
            g_kit_cache.held[device] = true;
            k = RunKit();
            return;
        }
    }
    kit_destroy(device, k);
}

// Shape of a resident run's pipelined transfers (see run_impl): out = {rows per chunk, iterations that follow the upload,
// iterations that run over the download}, {0, 0, 0} = plain order.  Eight chunks (~12 ms of PCIe each for a 4 GiB cube), as
// many iterations at either end as a chunk's transfer pays for; cubes under 256 MiB move in milliseconds and runs under four
// iterations have nothing to hide a transfer under.  TVDN_PIPELINE=0 keeps the plain order, "R,k0,k1" forces a shape (tests).
void pipeline_plan(int64_t n0, int64_t n_total, int64_t cube_bytes, int32_t out[3])
{
    out[0] = out[1] = out[2] = 0;
    if (n_total <= 0 || n0 <= 0) return;
    const char *e = getenv("TVDN_PIPELINE");
    if (e && strchr(e, ',')) {
        int r_ = 0, k0_ = 0, k1_ = 0;
        if (sscanf(e, "%d,%d,%d", &r_, &k0_, &k1_) == 3 && r_ >= 1 && k0_ >= 0 && k1_ >= 0) {
            out[0] = r_;
            out[1] = (int32_t)std::min<int64_t>(k0_, n_total);
            out[2] = (int32_t)std::min<int64_t>(k1_, n_total - out[1]);
        }
        return;
    }
    if ((e && atoi(e) == 0) || n_total < 4 || n0 < 32 || cube_bytes < (int64_t(256) << 20)) return;
    out[0] = (int32_t)std::max<int64_t>(8, (n0 + 7) / 8);
    out[1] = (int32_t)std::min<int64_t>(8, n_total / 2);
    out[2] = (int32_t)std::min<int64_t>(8, n_total - out[1]);
}

// Several slabs whose blocks are made of granules: does a peer copy out of such a block arrive intact?  For every slab and
// each neighbour whose rows it will pull: a pattern goes into the neighbour's block where its outermost own row of recon[1]
// lies (the runtime's host-to-device copy), is pulled across with the run's own call (hipMemcpyPeerAsync on the copy stream)
// into this slab's halo row, and read back (device-to-host).  64 KiB per pair, at the END of the row -- in a block of several
// granules that is as far from the block's first granule as the rows a run moves.  Leaves the touched bytes zero.
bool slab_peer_copy_check(const PeerSlab *sl, int world, size_t row_bytes, int nd, int per_axis, char *why, size_t why_len)
{
    const size_t n = std::min<size_t>(row_bytes, 65536);
    std::vector<unsigned char> pat(n), got(n);
    auto fail = [&](const char *what, hipError_t e, int r, int d) {
        (void)hipGetLastError();
        snprintf(why, why_len, "%s, slab %d (device %d) <- slab %d (device %d): %s", what, r, sl[r].device, d, sl[d].device, e == hipSuccess ? "bytes differ" : hipGetErrorString(e));
        return false;
    };
    for (int r = 0; r < world; ++r) {
        const PeerSlab &s = sl[r];
        const size_t s_stride = ((size_t)s.rows * row_bytes + 255) / 256 * 256 + 4096;
        char *s_recon1 = (char *)s.state + s_stride * (size_t)(nd * per_axis);  // Slab::assign: recon[1] follows the rotating arrays
        for (int side = 0; side < 2; ++side) {
            if (side == 0 ? !s.halo_lo : !s.halo_hi) continue;
            const int d = side == 0 ? (r + world - 1) % world : (r + 1) % world;
            const PeerSlab &o = sl[d];
            const size_t o_stride = ((size_t)o.rows * row_bytes + 255) / 256 * 256 + 4096;
            char *o_recon1 = (char *)o.state + o_stride * (size_t)(nd * per_axis);
            char *src = o_recon1 + (size_t)(side == 0 ? o.row_hi - 1 : o.row_lo) * row_bytes + (row_bytes - n);
            char *dst = s_recon1 + (size_t)(side == 0 ? s.row_lo - 1 : s.row_hi) * row_bytes + (row_bytes - n);
            for (size_t i = 0; i < n; ++i) pat[i] = (unsigned char)(0x5b + 131 * i + 17 * r + 3 * side);
            hipError_t e = hipSetDevice(o.device);
            if (e == hipSuccess && tvdn_copy_to_device(src, pat.data(), n, o.device) != TVDN_OK) e = hipErrorUnknown;  // (pinned lanes: tvdn_hostio.hip)
            if (e != hipSuccess) return fail("writing the pattern", e, r, d);
            e = hipSetDevice(s.device);
            if (e == hipSuccess) e = hipMemcpyPeerAsync(dst, s.device, src, o.device, n, s.copy);
            if (e == hipSuccess) e = hipStreamSynchronize(s.copy);
            if (e != hipSuccess) return fail("the peer copy", e, r, d);
            std::fill(got.begin(), got.end(), 0);
            if (tvdn_copy_to_host(got.data(), dst, n, s.device) != TVDN_OK) e = hipErrorUnknown;
            if (e != hipSuccess) return fail("reading the copy back", e, r, d);
            if (got != pat) return fail("the peer copy", hipSuccess, r, d);
            e = hipMemset(dst, 0, n);
            if (e == hipSuccess) e = hipSetDevice(o.device);
            if (e == hipSuccess) e = hipMemset(src, 0, n);
            if (e != hipSuccess) return fail("clearing the pattern", e, r, d);
        }
    }
    return true;
}


}  // namespace tvdn

// The HBM arithmetic of check_memory (cyTVDN.py:438-467) for this engine: arrays of the compact state, bytes of the
// tallest slab of an n-way split with its halo rows, what the device has free, and the fewest slabs that would fit.
extern "C" int tvdn_plan(int dtype, int ndim, const int64_t *shape, int fista, int n_slabs, int device, tvdn_plan_out *out)
{
    tvdn::DeviceRestore restore;
    size_t kept = 0;  // the state block the last run of this device left for the next one counts as free
    if (device >= 0 && device < TVDN_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(tvdn::g_state_cache.mu);
        kept = tvdn::g_state_cache.bytes[device];
    }
    TVDN_REQUIRE(out != nullptr && shape != nullptr, "NULL argument");
    TVDN_REQUIRE(dtype == TVDN_F32 || dtype == TVDN_F64, "bad dtype %d", dtype);
    TVDN_REQUIRE(ndim == 3 || ndim == 4, "ndim must be 3 or 4, got %d", ndim);
    for (int i = 0; i < ndim; ++i) TVDN_REQUIRE(shape[i] >= 1, "shape[%d] must be >= 1", i);
    TVDN_REQUIRE(n_slabs >= 1 && n_slabs <= shape[0], "n_slabs must be 1..shape[0]");
    const int64_t item = dtype == TVDN_F32 ? 4 : 8;
    int64_t plane = item;
    for (int i = 1; i < ndim; ++i) plane *= shape[i];
    const int64_t arrays = 3 + (int64_t)ndim * (fista ? 3 : 2);
    auto slab_bytes = [&](int64_t s) {
        const int64_t rows = (shape[0] + s - 1) / s + (s > 1 ? 2 : 0);
        return arrays * (rows * plane + 4096 + 255);
    };
    size_t free_b = 0, total_b = 0;
    TVDN_HIP(hipSetDevice(device));
    TVDN_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += kept;
    out->arrays = arrays;
    out->bytes_per_slab = slab_bytes(n_slabs);
    out->free_bytes = (int64_t)free_b;
    out->fits = out->bytes_per_slab <= (int64_t)(0.9 * (double)free_b) ? 1 : 0;
    out->min_slabs = 0;
    for (int64_t s = 1; s <= shape[0]; ++s)
        if (slab_bytes(s) <= (int64_t)(0.9 * (double)free_b)) {
            out->min_slabs = (int32_t)s;
            break;
        }
    return TVDN_OK;
}

// What the first tvdn_run of a process pays once, paid now instead: the pinned staging lanes (tvdn_hostio.hip), the first streams of
// the two priority classes a run uses (the first stream of a class costs ~30 ms, later ones 1 ms), the reduction scratch, the
// library's code object on the device (its first kernel), and the allocator's canary.  Everything here is idempotent.
extern "C" int tvdn_warm_up(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) {
        tvdn::set_error("no HIP device visible: the product path needs an MI355X (gfx950); there is no CPU fallback");
        return TVDN_ERR_NO_DEVICE;
    }
    TVDN_REQUIRE(device >= 0 && device < n, "device %d out of range (0..%d)", device, n - 1);
    tvdn::DeviceRestore restore;
    TVDN_HIP(hipSetDevice(device));
    tvdn::io_warm(device);
    hipStream_t hi = nullptr, lo = nullptr;
    int rc = tvdn::make_stream(&hi, +1);
    if (!rc) rc = tvdn::make_stream(&lo, 0);
    if (hi) (void)hipStreamDestroy(hi);
    if (lo) (void)hipStreamDestroy(lo);
    if (rc) return rc;
    tvdn_ctx *c = nullptr;
    if ((rc = tvdn_ctx_create(&c, device))) return rc;
    (void)tvdn_ctx_destroy(c);
    return tvdn_mem_selftest(device) == TVDN_OK ? TVDN_OK : TVDN_OK;  // (a tripped canary is not an error of the warm-up: the device serves plain blocks)
}

extern "C" int64_t tvdn_state_kept_bytes(int device) { return (int64_t)tvdn::state_kept_bytes(device); }

extern "C" int tvdn_release_cache(void)
{
    tvdn::DeviceRestore restore;
    std::lock_guard<std::mutex> lk(tvdn::g_state_cache.mu);
    for (int d = 0; d < TVDN_MAX_DEVICES; ++d)
        if (tvdn::g_state_cache.p[d]) {
            (void)hipSetDevice(d);
            (void)tvdn::dev_free(tvdn::g_state_cache.p[d]);
            tvdn::g_state_cache.p[d] = nullptr;
            tvdn::g_state_cache.bytes[d] = 0;
        }
    std::lock_guard<std::mutex> lk2(tvdn::g_kit_cache.mu);
    for (int d = 0; d < TVDN_MAX_DEVICES; ++d)
        if (tvdn::g_kit_cache.held[d]) {
            tvdn::kit_destroy(d, tvdn::g_kit_cache.kit[d]);
            tvdn::g_kit_cache.held[d] = false;
        }
    return TVDN_OK;
}

extern "C" int tvdn_run_workspace_bytes(const tvdn_run_args *a, int64_t *bytes)
{
    TVDN_REQUIRE(a != nullptr && bytes != nullptr, "NULL argument");
    TVDN_REQUIRE(a->dtype == TVDN_F32 || a->dtype == TVDN_F64, "bad dtype %d", a->dtype);
    TVDN_REQUIRE(a->ndim == 3 || a->ndim == 4, "ndim must be 3 or 4, got %d", a->ndim);
    size_t b = a->dtype == TVDN_F32 ? 4 : 8;
    for (int i = 0; i < a->ndim; ++i) {
        TVDN_REQUIRE(a->shape[i] >= 1, "shape[%d] must be >= 1", i);
        b *= (size_t)a->shape[i];
    }
    const size_t stride = (b + 255) / 256 * 256 + 4096;  // as run_impl lays the state out
    *bytes = (int64_t)(stride * (size_t)(3 + a->ndim * (a->n_fista > 0 ? 3 : 2)));
    return TVDN_OK;
}

extern "C" int tvdn_pipeline_plan(int64_t n0, int32_t n_iters, int64_t cube_bytes, int32_t *out)
{
    TVDN_REQUIRE(out != nullptr, "NULL argument");
    TVDN_REQUIRE(n0 >= 1 && n_iters >= 0 && cube_bytes >= 0, "bad argument");
    tvdn::pipeline_plan(n0, n_iters, cube_bytes, out);
    return TVDN_OK;
}
