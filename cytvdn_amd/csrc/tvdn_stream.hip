// tvdn_run for a cube whose state does not fit the HBM of its device: the out-of-core wavefront schedule in C++
// (include/tvdn.h, tvdn_run_args.stream_rows / stream_k).  Same schedule as cytvdn_amd/wavefront.py for one process:
// the state lives in pinned host memory, streams through the GPU once per pass in chunks of R rows, and iteration
// level j+1 trails level j by one row, so every row of every level is swept exactly once and crosses PCIe once per
// k iterations.  Every level keeps a ring of R+2 rows per array in HBM (tvdn_iter_args.ring_rows); the sweeps are
// tvdn_iterate_fused launches, so the bits are those of the resident engine.  Upstream has no counterpart: its
// arrays never leave the host (cyTVDN/cyTVDN.py:148-242 is the loop this replaces for cubes beyond HBM).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <memory>
#include <thread>

#include <unistd.h>

#include "tvdn_common.hpp"

namespace tvdn {

namespace {

constexpr int kHostThreads = 8;
constexpr size_t kPinInPlaceMin = size_t(256) << 20;  // bytes from which a caller's array is page-locked in place

void parallel_copy(void *dst, const void *src, size_t bytes)  // src == nullptr: zero fill
{
    const size_t piece = (bytes / kHostThreads + 4095) / 4096 * 4096;
    std::thread th[kHostThreads];
    int n = 0;
    for (size_t off = 0; off < bytes; off += piece, ++n) {
        const size_t len = std::min(piece, bytes - off);
        th[n] = std::thread([=] {
            if (src)
                std::memcpy((char *)dst + off, (const char *)src + off, len);
            else
                std::memset((char *)dst + off, 0, len);
        });
    }
    for (int i = 0; i < n; ++i) th[i].join();
}

// A cube-sized host array the GPU can reach: the caller's own memory page-locked in place when the runtime allows
// it, otherwise a pinned allocation (filled from / copied back to the caller's array by the user of this struct).
struct HostArr {
    char *p = nullptr;
    bool registered = false, owned = false;
    int pin_in_place(void *user, size_t bytes)
    {
        // Only arrays big enough to own their pages: page-locking works on whole pages, and two small arrays of the
        // caller may share one (overlapping registrations).  Small cubes are staged through pinned copies instead.
        if (bytes < kPinInPlaceMin) return alloc(bytes);
        const hipError_t e = hipHostRegister(user, bytes, hipHostRegisterDefault);
        if (e == hipSuccess) {
            p = (char *)user;
            registered = true;
            return TVDN_OK;
        }
        (void)hipGetLastError();
        if (e == hipErrorHostMemoryAlreadyRegistered) {  // already page-locked by the caller
            p = (char *)user;
            return TVDN_OK;
        }
        return alloc(bytes);
    }
    int alloc(size_t bytes)
    {
        TVDN_HIP(hipHostMalloc((void **)&p, bytes, hipHostMallocDefault));
        owned = true;
        return TVDN_OK;
    }
    ~HostArr()
    {
        if (registered) (void)hipHostUnregister(p);
        if (owned) (void)hipHostFree(p);
    }
};

struct Ring {  // `cap` row-planes; global row g lives at slot g % cap
    char *base = nullptr;
    int64_t cap = 0;
    size_t row_bytes = 0;
    char *row(int64_t g) const { return base + (size_t)(g % cap) * row_bytes; }
};

struct Events {
    std::vector<hipEvent_t> ev;
    int make(hipEvent_t *e)
    {
        TVDN_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
        ev.push_back(*e);
        return TVDN_OK;
    }
    ~Events()
    {
        for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    }
};

struct Streams {
    hipStream_t main = nullptr, up = nullptr, down = nullptr;
    ~Streams()
    {
        for (hipStream_t s : {main, up, down})
            if (s) (void)hipStreamDestroy(s);
    }
};

struct CtxHolder {
    tvdn_ctx *c = nullptr;
    ~CtxHolder()
    {
        if (c) (void)tvdn_ctx_destroy(c);
    }
};

struct DevMem {
    void *p = nullptr;
    ~DevMem()
    {
        if (p) (void)hipFree(p);
    }
};

// Host memory a streamed run may count on: what the kernel calls available, never more than the machine has, and
// never more than the memory limit of the process's control group (the limit itself, not limit minus usage: the
// usage counts page cache the kernel would give back, and a false refusal helps nobody; the check is there to stop
// requests that are wrong by factors).  0 = could not be determined.
size_t host_available_bytes()
{
    size_t avail = 0;
    const long pages = sysconf(_SC_PHYS_PAGES), page = sysconf(_SC_PAGE_SIZE);
    const size_t physical = (pages > 0 && page > 0) ? (size_t)pages * (size_t)page : 0;
    if (FILE *f = fopen("/proc/meminfo", "r")) {
        char line[256];
        while (fgets(line, sizeof line, f)) {
            unsigned long long kb = 0;
            if (sscanf(line, "MemAvailable: %llu kB", &kb) == 1) {
                avail = (size_t)kb * 1024;
                break;
            }
        }
        fclose(f);
    }
    if (avail == 0 || (physical && avail > physical)) avail = physical;
    auto read_num = [](const char *path, unsigned long long *v) -> bool {
        FILE *f = fopen(path, "r");
        if (!f) return false;
        char buf[64] = {0};
        const bool ok = fgets(buf, sizeof buf, f) != nullptr && sscanf(buf, "%llu", v) == 1;  // "max" does not parse: no limit
        fclose(f);
        return ok;
    };
    unsigned long long lim = 0;
    if (read_num("/sys/fs/cgroup/memory.max", &lim) || read_num("/sys/fs/cgroup/memory/memory.limit_in_bytes", &lim))
        if (lim > 0 && (size_t)lim < avail) avail = (size_t)lim;
    if (const char *e = getenv("TVDN_HOST_LIMIT")) {  // a cap from outside, "64G" / "512M" / bytes (the test-suite sets one)
        char *end = nullptr;
        double v = strtod(e, &end);
        if (end != e && v > 0) {
            switch (*end) {
            case 'K': case 'k': v *= 1024.0; break;
            case 'M': case 'm': v *= 1024.0 * 1024.0; break;
            case 'G': case 'g': v *= 1024.0 * 1024.0 * 1024.0; break;
            case 'T': case 't': v *= 1024.0 * 1024.0 * 1024.0 * 1024.0; break;
            default: break;
            }
            if (v < (double)avail) avail = (size_t)v;
        }
    }
    return avail;
}

bool arrays_overlap(const void *x, const void *y, size_t bytes)
{
    const uintptr_t a0 = (uintptr_t)x, b0 = (uintptr_t)y;
    return a0 < b0 + bytes && b0 < a0 + bytes;
}

// n row-plane copies inside HBM: one streaming launch when the rows are 16-byte multiples, the runtime's copies otherwise
int copy_rows(std::vector<void *> &dst, std::vector<void *> &src, size_t row_bytes, hipStream_t s)
{
    if (row_bytes % 16 == 0)
        return tvdn_copy_many((int32_t)dst.size(), dst.data(), src.data(), (int64_t)row_bytes, 0, s);
    for (size_t i = 0; i < dst.size(); ++i) TVDN_HIP(hipMemcpyAsync(dst[i], src[i], row_bytes, hipMemcpyDeviceToDevice, s));
    return TVDN_OK;
}

}  // namespace

// Rows of HBM (planes) the schedule keeps resident: planner.wavefront_windows of the Python side.
static int64_t stream_planes(int nd, int64_t rows, int64_t k, bool mse, bool wrap)
{
    return ((k + 1) + (k + 2) * nd) * (rows + 2) + (rows + k + 3) * (mse ? 2 : 1) + 2 * (3 + 4 * nd) * rows + (wrap ? k + 1 : 0);
}

// Chunk height and depth: the deepest k (<= 128) whose rings and staging boxes fit 85 % of the free HBM, taller chunks on
// ties.  Every byte the run allocates on the device is in that count; on PCIe-bound shapes depth IS speed (256 MiB planes:
// k = 30 / 38 / 44 -> 28.4 / 33.6 / 37.0 Gvoxel-iters/s, profiles/r03_outofcore_depth.jsonl), so the budget is generous.
// (a streamed pass is PCIe-bound until k ~ 100: cytvdn_amd/planner.py).
int choose_stream_shape(int nd, int64_t n_rows, size_t row_bytes, size_t free_bytes, bool mse, bool wrap, int64_t *rows_out,
                        int64_t *k_out)
{
    const int64_t budget = (int64_t)(0.85 * (double)free_bytes / (double)row_bytes);
    int64_t best_k = 0, best_r = 0;
    for (int64_t r : {32, 16, 8, 4, 2}) {
        r = std::min<int64_t>(r, std::max<int64_t>(2, n_rows));
        const int64_t slope = stream_planes(nd, r, 2, mse, wrap) - stream_planes(nd, r, 1, mse, wrap);
        int64_t k = (budget - stream_planes(nd, r, 0, mse, wrap)) / slope;
        k = std::min<int64_t>({k, 128, std::max<int64_t>(1, n_rows)});
        if (k >= 1 && stream_planes(nd, r, k, mse, wrap) <= budget && k > best_k) {
            best_k = k;
            best_r = r;
        }
    }
    if (best_k < 1) {
        set_error("not even 2-row chunks of one iteration level fit the device: %lld planes of %zu bytes in %zu free bytes",
                  (long long)stream_planes(nd, 2, 1, mse, wrap), row_bytes, free_bytes);
        return TVDN_ERR_UNSUPPORTED;
    }
    *rows_out = best_r;
    *k_out = best_k;
    return TVDN_OK;
}

}  // namespace tvdn

// Host side of a streamed run, as arithmetic only (no HIP call, no device needed, nothing of the caller's dereferenced):
// the page-locked bytes it would hold -- the data term, recon (= recon_out), the reference when an MSE trace is asked
// for, one or two accumulator-state arrays per axis, and one more cube when `data` overlaps `recon_out` (the data term
// then needs its own copy) -- against what the host may give (MemAvailable, physical memory, the control group's
// limit, TVDN_HOST_LIMIT), of which a streamed run takes 80 % at most.
extern "C" int tvdn_stream_host_need(const tvdn_run_args *a, int64_t *need_bytes, int64_t *avail_bytes)
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
    const bool aliased = a->data && a->recon_out && cube < 9.0e18 && tvdn::arrays_overlap(a->data, a->recon_out, cube_b);
    // periodic boundaries: the rows at one end are the other end's halo, uploaded late in a pass that has already sent
    // their new values home -- old and new state are then two sets of arrays instead of one updated in place
    const int twice = a->bc_mode == TVDN_BC_PERIODIC ? 2 : 1;
    const double need = (double)((a->ndim * n_state + 1) * twice + 1 + (want_mse ? 1 : 0) + (aliased ? 1 : 0)) * cube;
    const size_t avail = tvdn::host_available_bytes();
    if (need_bytes) *need_bytes = need < 9.0e18 ? (int64_t)need : INT64_MAX;
    if (avail_bytes) *avail_bytes = (int64_t)avail;
    if (avail == 0 || need > 0.8 * (double)avail) {
        tvdn::set_error("a streamed run of this cube needs %.0f bytes of page-locked host memory, which exceeds what the host has "
                        "available (%zu bytes, of which 80 %% are used at most): cut it into slabs over several nodes (cytvdn_amd.plan_run)",
                        need, avail);
        return TVDN_ERR_UNSUPPORTED;
    }
    return TVDN_OK;
}

namespace tvdn {

int run_streamed(const tvdn_run_args *a, int64_t R, int64_t K)
{
    const auto t_start = std::chrono::steady_clock::now();
    const int nd = a->ndim;
    const size_t item = a->dtype == TVDN_F32 ? 4 : 8;
    size_t plane = 1;
    for (int i = 1; i < nd; ++i) plane *= (size_t)a->shape[i];
    const size_t row_bytes = plane * item;
    const int64_t N0 = a->shape[0];
    const size_t cube_bytes = (size_t)N0 * row_bytes;
    const int n_total = a->n_fista + a->n_plain;
    const bool fista = a->n_fista > 0;
    const int n_state = fista ? 2 : 1;
    const bool want_mse = a->mse_out != nullptr && a->reference != nullptr;
    const int device = a->n_devices > 0 ? a->devices[0] : a->device;
    const bool periodic = a->bc_mode == TVDN_BC_PERIODIC;
    TVDN_REQUIRE(periodic || a->bc_mode == TVDN_BC_JIA_ZHAO, "the streamed tvdn_run handles bc_mode 0 and 2");
    TVDN_REQUIRE(R >= 1 && K >= 1, "stream_rows and stream_k must be >= 1");
    if (a->use_stop) K = 1;  // the stopping rule needs a decision after every iteration: one level per pass
    K = std::min<int64_t>(K, std::max<int64_t>(1, n_total));
    // Periodic boundaries along axis 0: the sweeps see a virtual cube of N0 + 2 K rows -- the cube between K wrapped rows
    // at either end, which are each other's halo -- and, as at the face between two slabs, give up one row per level at
    // the two artificial faces; the wrap itself is never swept (cytvdn_amd/wavefront.py does the same).
    if (periodic) K = std::min<int64_t>(K, N0);
    const int64_t KX = periodic ? K : 0, NV = N0 + 2 * KX, G0 = KX, G1 = KX + N0;
    // BEFORE anything of the caller's is touched: can the host hold the state at all?  (page-locked: it cannot swap)
    {
        int64_t need = 0, avail = 0;
        const int rc0 = tvdn_stream_host_need(a, &need, &avail);
        if (rc0) return rc0;
    }
    TVDN_HIP(hipSetDevice(device));
    // ... and do the rings fit the device?  Also before anything is page-locked or copied (the wrap planes are counted
    // whether or not they will be needed: deciding that reads the caller's first row).
    const int64_t cap = R + 2, ocap = R + K + 3;
    const int n_in = 2 + nd * n_state + (want_mse ? 1 : 0), n_out = 1 + nd * n_state;
    auto aligned = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t ring_b = aligned((size_t)cap * row_bytes), oring_b = aligned((size_t)ocap * row_bytes);
    const size_t box_b = aligned((size_t)R * row_bytes), plane_b = aligned(row_bytes);
    const size_t n_rings = (size_t)(K + 1) + (size_t)(K + 2) * nd;
    const size_t dev_bytes_max = n_rings * ring_b + oring_b * (want_mse ? 2 : 1) + 2 * (size_t)(n_in + n_out) * box_b +
                                 (size_t)(K + 1) * plane_b;
    {
        size_t free_b = 0, total_b = 0;
        TVDN_HIP(hipMemGetInfo(&free_b, &total_b));
        if (dev_bytes_max > free_b) {
            set_error("streamed run with %lld-row chunks and k = %lld needs %zu bytes of HBM, device %d has %zu free",
                      (long long)R, (long long)K, dev_bytes_max, device, free_b);
            return TVDN_ERR_UNSUPPORTED;
        }
    }

    // Jia-Zhao wrap at the top face: exact (TVDN_EDGE_WRAP, row 0 of every level kept aside) when row 0 is not finite
    bool exact_wrap = false;
    if (periodic) {
        // the wrap is swept for real on the extended cube
    } else if (a->dtype == TVDN_F32) {
        const float *p0 = (const float *)a->data;
        for (size_t i = 0; i < plane && !exact_wrap; ++i) exact_wrap = !std::isfinite(p0[i]);
    } else {
        const double *p0 = (const double *)a->data;
        for (size_t i = 0; i < plane && !exact_wrap; ++i) exact_wrap = !std::isfinite(p0[i]);
    }

    // ---- host state: orig and recon are the caller's arrays page-locked in place where possible ----------------------
    HostArr orig_h, recon_h, ref_h;
    std::unique_ptr<HostArr[]> state_h(new HostArr[(size_t)nd * 2]);
    // `data` may be the very array the result goes to (the resident run allows it too): the passes then write recon rows
    // over the rows the next pass would upload as `orig`, so the data term gets its own pinned copy first.
    const bool aliased = arrays_overlap(a->data, a->recon_out, cube_bytes);
    int rc = aliased ? orig_h.alloc(cube_bytes) : orig_h.pin_in_place(const_cast<void *>(a->data), cube_bytes);
    if (rc) return rc;
    if (orig_h.owned) parallel_copy(orig_h.p, a->data, cube_bytes);
    if (a->recon_out != a->data) {  // recon = datacube.copy() (cyTVDN.py:145)
        if (aliased)
            std::memmove(a->recon_out, a->data, cube_bytes);
        else
            parallel_copy(a->recon_out, a->data, cube_bytes);
    }
    rc = recon_h.pin_in_place(a->recon_out, cube_bytes);
    if (rc) return rc;
    if (recon_h.owned) parallel_copy(recon_h.p, a->data, cube_bytes);
    if (want_mse) {
        rc = ref_h.pin_in_place(const_cast<void *>(a->reference), cube_bytes);
        if (rc) return rc;
        if (ref_h.owned) parallel_copy(ref_h.p, a->reference, cube_bytes);
    }
    for (int q = 0; q < nd; ++q)
        for (int s = 0; s < n_state; ++s) {
            rc = state_h[(size_t)q * 2 + s].alloc(cube_bytes);
            if (rc) return rc;
            parallel_copy(state_h[(size_t)q * 2 + s].p, nullptr, cube_bytes);
        }
    // periodic: a second set (new state of a pass); Jia-Zhao runs update the one set in place
    HostArr recon2_h;
    std::unique_ptr<HostArr[]> state2_h(new HostArr[(size_t)nd * 2]);
    if (periodic) {
        if ((rc = recon2_h.alloc(cube_bytes))) return rc;
        for (int q = 0; q < nd; ++q)
            for (int s = 0; s < n_state; ++s)
                if ((rc = state2_h[(size_t)q * 2 + s].alloc(cube_bytes))) return rc;
    }
    int h_old = 0;  // which set holds the current state (periodic); 0 = recon_h / state_h
    auto recon_of = [&](int set) -> char * { return (periodic && set) ? recon2_h.p : recon_h.p; };
    auto state_of = [&](int set, int q, int s) -> char * {
        return (periodic && set) ? state2_h[(size_t)q * 2 + s].p : state_h[(size_t)q * 2 + s].p;
    };

    // ---- device: rings, staging boxes, sums ----------------------------------------------------------------------------
    CtxHolder ctx;
    rc = tvdn_ctx_create(&ctx.c, device);
    if (rc) return rc;
    Streams st;
    if ((rc = make_stream(&st.main, +1))) return rc;  // three queue classes: no false ordering between sweeps, uploads and
    if ((rc = make_stream(&st.up, 0))) return rc;     // downloads whatever other streams the process holds (tvdn_common.hpp)
    if ((rc = make_stream(&st.down, -1))) return rc;
    const size_t dev_bytes = dev_bytes_max - (exact_wrap ? 0 : (size_t)(K + 1) * plane_b);
    DevMem mem, sums_d, mse_d;
    TVDN_HIP(hipMalloc(&mem.p, dev_bytes));
    TVDN_HIP(hipMemsetAsync(mem.p, 0, dev_bytes, st.main));
    char *cursor = (char *)mem.p;
    auto take = [&](size_t b) { char *p = cursor; cursor += b; return p; };
    std::vector<Ring> Rw((size_t)K + 1);
    std::vector<Ring> Aw((size_t)(K + 2) * nd);  // [level + 1][axis]
    for (Ring &r : Rw) r = Ring{take(ring_b), cap, row_bytes};
    for (Ring &r : Aw) r = Ring{take(ring_b), cap, row_bytes};
    Ring Ow{take(oring_b), ocap, row_bytes}, Fw;
    if (want_mse) Fw = Ring{take(oring_b), ocap, row_bytes};
    char *inbox[2][12], *outbox[2][12];
    for (int h = 0; h < 2; ++h) {
        for (int i = 0; i < n_in; ++i) inbox[h][i] = take(box_b);
        for (int i = 0; i < n_out; ++i) outbox[h][i] = take(box_b);
    }
    std::vector<char *> row0;  // row 0 of every level, kept for the top face
    if (exact_wrap)
        for (int64_t j = 0; j <= K; ++j) row0.push_back(take(plane_b));
    auto A = [&](int64_t level, int q) -> Ring & { return Aw[(size_t)(level + 1) * nd + q]; };

    // one slot per iteration, and a last one that takes the sums of halo rows (periodic: the wrapped rows are swept too)
    TVDN_HIP(hipMalloc(&sums_d.p, sizeof(double) * 3 * (size_t)(n_total + 1)));
    TVDN_HIP(hipMemsetAsync(sums_d.p, 0, sizeof(double) * 3 * (size_t)(n_total + 1), st.main));
    const int discard = n_total;
    // squared errors per (slot, row): summed in row order on the host at the end
    if (want_mse) {
        TVDN_HIP(hipMalloc(&mse_d.p, sizeof(double) * (size_t)(n_total + 1) * (size_t)N0));
        TVDN_HIP(hipMemsetAsync(mse_d.p, 0, sizeof(double) * (size_t)(n_total + 1) * (size_t)N0, st.main));
    }
    Events evs;
    hipEvent_t in_ready[2], in_free[2], out_ready[2], out_free[2];
    bool in_free_set[2] = {false, false}, out_free_set[2] = {false, false};
    for (int h = 0; h < 2; ++h) {
        if ((rc = evs.make(&in_ready[h])) || (rc = evs.make(&in_free[h])) || (rc = evs.make(&out_ready[h])) ||
            (rc = evs.make(&out_free[h])))
            return rc;
    }

    tvdn_iter_args it;
    std::memset(&it, 0, sizeof it);
    it.dtype = a->dtype;
    it.ndim = nd;
    it.shape[0] = NV;
    for (int i = 1; i < nd; ++i) it.shape[i] = a->shape[i];
    it.row_lo = 0;
    it.row_hi = NV;
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
    int64_t one_row[4];
    one_row[0] = 1;
    for (int i = 1; i < nd; ++i) one_row[i] = a->shape[i];
    auto sse_row = [&](const char *x, const char *y, int slot, int64_t g) -> int {
        return tvdn_sum_square_error(ctx.c, a->dtype, nd, one_row, x, y, (double *)mse_d.p + (size_t)slot * (size_t)N0 + (size_t)g,
                                     st.main);
    };

    // ---- one pass: `kk` iteration levels over the whole cube ------------------------------------------------------------
    const int down_blocks = getenv("TVDN_STREAM_DOWN_BLOCKS") ? atoi(getenv("TVDN_STREAM_DOWN_BLOCKS")) : 0;
    bool d_form = fista;
    double tk_prev = 0.0;
    int done = 0;
    std::vector<void *> cdst, csrc;
    auto pass = [&](const double *ratios /* kk entries, NAN = unaccelerated */, int kk) -> int {
        std::vector<int> modes((size_t)kk);
        std::vector<double> tkp((size_t)kk);
        std::vector<char> forms((size_t)kk + 1);
        forms[0] = d_form;
        double prev = tk_prev;
        for (int j = 0; j < kk; ++j) {
            const bool acc = !std::isnan(ratios[j]);
            TVDN_REQUIRE(!acc || forms[j], "a FISTA iteration cannot follow an unaccelerated one");
            modes[j] = iter_mode(acc, forms[j] != 0);
            forms[j + 1] = acc;
            tkp[j] = prev;
            if (acc) prev = ratios[j];
        }
        const int n_in_state = forms[0] ? 2 : 1, n_out_state = forms[kk] ? 2 : 1;
        // rows of the (virtual) cube this pass works on, and what each level can reach at an artificial face
        const int64_t E0 = periodic ? KX - kk : 0, E1 = periodic ? G1 + kk : N0;
        auto lo_bound = [&](int64_t level) { return periodic ? E0 + level : (int64_t)0; };
        auto hi_bound = [&](int64_t level) { return periodic ? E1 - level : N0; };
        const int64_t n_chunks = (E1 - E0 + kk + R - 1) / R;
        const int h_new = periodic ? h_old ^ 1 : h_old;

        // virtual rows [v0, v1) of a host array -> box: host row = (v - KX) mod N0, i.e. up to three contiguous pieces
        auto up_rows = [&](char *box, const char *host, int64_t v0, int64_t v1) -> int {
            for (int64_t v = v0; v < v1;) {
                const int64_t hrow = ((v - KX) % N0 + N0) % N0;
                const int64_t n = std::min(v1 - v, N0 - hrow);
                TVDN_HIP(hipMemcpyAsync(box + (size_t)(v - v0) * row_bytes, host + (size_t)hrow * row_bytes, (size_t)n * row_bytes,
                                        hipMemcpyHostToDevice, st.up));
                v += n;
            }
            return TVDN_OK;
        };
        auto upload = [&](int64_t c) -> int {
            const int64_t u0 = E0 + c * R, u1 = std::min(E0 + (c + 1) * R, E1);
            if (u0 >= u1) return TVDN_OK;
            const int h = (int)(c % 2);
            if (in_free_set[h]) TVDN_HIP(hipStreamWaitEvent(st.up, in_free[h], 0));
            int i = 0, rcu;
            if ((rcu = up_rows(inbox[h][i++], orig_h.p, u0, u1))) return rcu;
            if ((rcu = up_rows(inbox[h][i++], recon_of(h_old), u0, u1))) return rcu;
            for (int q = 0; q < nd; ++q)
                for (int s = 0; s < n_in_state; ++s)
                    if ((rcu = up_rows(inbox[h][i++], state_of(h_old, q, s), u0, u1))) return rcu;
            if (want_mse && (rcu = up_rows(inbox[h][i++], ref_h.p, u0, u1))) return rcu;
            TVDN_HIP(hipEventRecord(in_ready[h], st.up));
            return TVDN_OK;
        };

        int rc2 = upload(0);
        if (rc2) return rc2;
        for (int64_t c = 0; c < n_chunks; ++c) {
            if ((rc2 = upload(c + 1))) return rc2;  // the next chunk crosses PCIe while this one is swept
            const int h = (int)(c % 2);
            const int64_t u0 = E0 + c * R, u1 = std::min(E0 + (c + 1) * R, E1);
            if (u0 < u1) {
                TVDN_HIP(hipStreamWaitEvent(st.main, in_ready[h], 0));
                cdst.clear();
                csrc.clear();
                auto scatter = [&](const Ring &rg, const char *box) {
                    for (int64_t g = u0; g < u1; ++g) {
                        cdst.push_back(rg.row(g));
                        csrc.push_back((void *)(box + (size_t)(g - u0) * row_bytes));
                    }
                };
                int i = 0;
                scatter(Ow, inbox[h][i++]);
                scatter(Rw[0], inbox[h][i++]);
                for (int q = 0; q < nd; ++q) {
                    scatter(A(0, q), inbox[h][i++]);                        // level 0: d_k (or b)
                    if (n_in_state == 2) scatter(A(-1, q), inbox[h][i++]);  // level -1: d_k-1
                }
                if (want_mse) scatter(Fw, inbox[h][i++]);
                rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
                if (rc2) return rc2;
                if (exact_wrap && u0 == 0)
                    TVDN_HIP(hipMemcpyAsync(row0[0], Rw[0].row(0), row_bytes, hipMemcpyDeviceToDevice, st.main));
                if (want_mse && done == 0)  // MSE[0]: the input against the reference (cyTVDN.py:124-125), own rows
                    for (int64_t g = std::max(u0, G0); g < std::min(u1, G1); ++g)
                        if ((rc2 = sse_row(Rw[0].row(g), Fw.row(g), 0, g - KX))) return rc2;
                TVDN_HIP(hipEventRecord(in_free[h], st.main));
                in_free_set[h] = true;
            }
            // the wavefront: level j+1 trails level j by one row
            for (int j = 0; j < kk; ++j) {
                const int64_t lo = std::max(lo_bound(j + 1), E0 + c * R - (j + 1)), hi = std::min(hi_bound(j + 1), E0 + (c + 1) * R - (j + 1));
                if (lo >= hi) continue;
                it.mode = modes[j];
                it.tk = modes[j] == TVDN_ITER_FISTA_D ? ratios[j] : 0.0;
                it.tk_prev = tkp[j];
                it.recon_in = Rw[j].base;
                it.recon_out = Rw[j + 1].base;
                it.wrap_recon = exact_wrap ? row0[j] : nullptr;
                for (int q = 0; q < nd; ++q) {
                    char *cur = A(j, q).base, *prv = A(j - 1, q).base, *nxt = A(j + 1, q).base;
                    it.b_in[q] = it.d_in[q] = it.dprev_in[q] = nullptr;
                    it.b_out[q] = it.d_out[q] = nullptr;
                    if (modes[j] == TVDN_ITER_FISTA_D) {
                        it.d_in[q] = cur; it.dprev_in[q] = prv; it.d_out[q] = nxt;
                    } else if (modes[j] == TVDN_ITER_FISTA_D_TO_PLAIN) {
                        it.d_in[q] = cur; it.dprev_in[q] = prv; it.b_out[q] = nxt;
                    } else {
                        it.b_in[q] = cur; it.b_out[q] = nxt;
                    }
                }
                // the sums count the cube's own rows once: wrapped rows (periodic) go to the discard slot
                const int64_t parts[3][2] = {{lo, std::min(hi, G0)}, {std::max(lo, G0), std::min(hi, G1)}, {std::max(lo, G1), hi}};
                for (int part = 0; part < 3; ++part) {
                    const int64_t x0 = parts[part][0], x1 = parts[part][1];
                    if (x0 >= x1) continue;
                    it.sweep_lo = x0;
                    it.sweep_hi = x1;
                    const int slot = part == 1 ? done + j : discard;
                    rc2 = tvdn_iterate_fused(ctx.c, &it, (double *)sums_d.p + 3 * (size_t)slot, st.main);
                    if (rc2) return rc2;
                    if (want_mse && part == 1)
                        for (int64_t g = x0; g < x1; ++g)
                            if ((rc2 = sse_row(Fw.row(g), Rw[j + 1].row(g), done + j + 1, g - KX))) return rc2;
                }
                if (exact_wrap && lo == 0)
                    TVDN_HIP(hipMemcpyAsync(row0[j + 1], Rw[j + 1].row(0), row_bytes, hipMemcpyDeviceToDevice, st.main));
            }
            // rows that have reached the last level go home
            const int64_t lo = std::max(G0, E0 + c * R - kk), hi = std::min(G1, E0 + (c + 1) * R - kk);
            if (lo < hi) {
                if (out_free_set[h]) TVDN_HIP(hipStreamWaitEvent(st.main, out_free[h], 0));
                cdst.clear();
                csrc.clear();
                auto gather = [&](char *box, const Ring &rg) {
                    for (int64_t g = lo; g < hi; ++g) {
                        cdst.push_back(box + (size_t)(g - lo) * row_bytes);
                        csrc.push_back(rg.row(g));
                    }
                };
                int i = 0;
                gather(outbox[h][i++], Rw[kk]);
                for (int q = 0; q < nd; ++q) {
                    gather(outbox[h][i++], A(kk, q));
                    if (n_out_state == 2) gather(outbox[h][i++], A(kk - 1, q));
                }
                rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
                if (rc2) return rc2;
                TVDN_HIP(hipEventRecord(out_ready[h], st.main));
                TVDN_HIP(hipStreamWaitEvent(st.down, out_ready[h], 0));
                const size_t off = (size_t)(lo - KX) * row_bytes, len = (size_t)(hi - lo) * row_bytes;
                i = 0;
                if (down_blocks > 0 && len % 16 == 0) {  // measurement knob: one capped copy launch writing pinned memory
                    cdst.clear();
                    csrc.clear();
                    cdst.push_back(recon_of(h_new) + off);
                    csrc.push_back(outbox[h][i++]);
                    for (int q = 0; q < nd; ++q)
                        for (int s = 0; s < n_out_state; ++s) {
                            cdst.push_back(state_of(h_new, q, s) + off);
                            csrc.push_back(outbox[h][i++]);
                        }
                    rc2 = tvdn_copy_many((int32_t)cdst.size(), cdst.data(), csrc.data(), (int64_t)len, down_blocks, st.down);
                    if (rc2) return rc2;
                } else {
                    TVDN_HIP(hipMemcpyAsync(recon_of(h_new) + off, outbox[h][i++], len, hipMemcpyDeviceToHost, st.down));
                    for (int q = 0; q < nd; ++q)
                        for (int s = 0; s < n_out_state; ++s)
                            TVDN_HIP(hipMemcpyAsync(state_of(h_new, q, s) + off, outbox[h][i++], len, hipMemcpyDeviceToHost, st.down));
                }
                TVDN_HIP(hipEventRecord(out_free[h], st.down));
                out_free_set[h] = true;
            }
        }
        TVDN_HIP(hipStreamSynchronize(st.down));
        TVDN_HIP(hipStreamSynchronize(st.main));
        TVDN_HIP(hipStreamSynchronize(st.up));
        d_form = forms[kk];
        tk_prev = prev;
        done += kk;
        h_old = h_new;
        return TVDN_OK;
    };

    // ---- the schedule: FISTA ratios in float64 on the host (cyTVDN.py:153-156), then the unaccelerated tail -------------
    std::vector<double> ratios((size_t)n_total);
    fista_ratios(a->n_fista, ratios.data());
    for (int i = a->n_fista; i < n_total; ++i) ratios[i] = NAN;
    int ran = 0, ran_phase[2] = {a->n_fista, a->n_plain};
    auto stop_after = [&](int slot, bool &stop) -> int {
        double s3[3];
        TVDN_HIP(hipMemcpy(s3, (double *)sums_d.p + 3 * (size_t)slot, sizeof s3, hipMemcpyDeviceToHost));
        const double delta = a->dtype == TVDN_F32 ? (double)((float)s3[1] / (float)s3[2]) : s3[1] / s3[2];
        stop = delta < a->stop;
        return TVDN_OK;
    };
    TVDN_HIP(hipStreamSynchronize(st.main));
    const auto t_passes = std::chrono::steady_clock::now();
    if (!a->use_stop) {
        for (int i = 0; i < n_total;) {  // a pass may hold the last FISTA iterations and the first unaccelerated ones
            const int kk = (int)std::min<int64_t>(K, n_total - i);
            if ((rc = pass(ratios.data() + i, kk))) return rc;
            ran += kk;
            i += kk;
            if (a->progress) a->progress((int32_t)i, a->progress_user);
        }
    } else {
        // one level per pass; phases as upstream runs them (cyTVDN.py:148-242): an early stop ends the FISTA phase, the
        // unaccelerated phase still runs, writing its sums from its own slot on
        for (int phase = 0; phase < 2; ++phase) {
            const int first = phase == 0 ? 0 : a->n_fista, last = phase == 0 ? a->n_fista : n_total;
            done = first;
            ran_phase[phase] = 0;
            for (int i = first; i < last; ++i) {
                if ((rc = pass(ratios.data() + i, 1))) return rc;
                ++ran;
                ++ran_phase[phase];
                if (a->progress) a->progress((int32_t)(i + 1), a->progress_user);
                bool stop;
                if ((rc = stop_after(i, stop))) return rc;
                if (stop) break;
            }
        }
    }

    if (getenv("TVDN_STREAM_TIMING")) {  // measurement aid: set-up (page-locking, first touch, rings) apart from the passes
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "tvdn_run streamed: rows %lld k %lld, set-up %.3f s, passes %.3f s\n", (long long)R, (long long)K,
                std::chrono::duration<double>(t_passes - t_start).count(), std::chrono::duration<double>(now - t_passes).count());
    }
    // ---- results home -------------------------------------------------------------------------------------------------------
    if (periodic && h_old == 1)
        parallel_copy(a->recon_out, recon2_h.p, cube_bytes);  // the last pass wrote the second set
    else if (recon_h.owned)
        parallel_copy(a->recon_out, recon_h.p, cube_bytes);
    if (n_total > 0) TVDN_HIP(hipMemcpy(a->sums_out, sums_d.p, sizeof(double) * 3 * (size_t)n_total, hipMemcpyDeviceToHost));
    if (want_mse) {
        std::vector<double> per_row((size_t)(n_total + 1) * (size_t)N0);
        TVDN_HIP(hipMemcpy(per_row.data(), mse_d.p, sizeof(double) * per_row.size(), hipMemcpyDeviceToHost));
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
    return TVDN_OK;
}

}  // namespace tvdn
