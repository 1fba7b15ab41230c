// tvdn_run for a cube whose state does not fit the HBM of its device: the out-of-core wavefront schedule in C++
// (include/tvdn.h, tvdn_run_args.stream_rows / stream_k) -- the ONE streamed engine of the tree (one device, a device list,
// a rank of a multi-process run): the state lives in pinned host memory, streams through the GPU once per pass in chunks of R rows, and iteration
// level j+1 trails level j by one row, so every row of every level is swept exactly once and crosses PCIe once per
// k iterations.  Every level keeps a ring of R+2 rows per array in HBM (tvdn_iter_args.ring_rows); the sweeps are
// tvdn_iterate_fused launches, so the bits are those of the resident engine.  Upstream has no counterpart: its
// arrays never leave the host (cyTVDN/cyTVDN.py:148-242 is the loop this replaces for cubes beyond HBM).
//
// Map of this file (it is long; every part is host code around tvdn_iterate_fused launches and copies):
//   host memory        PinnedBuf (huge-page anonymous memory + one registration; background release), HostArr (a caller's array
//                      page-locked in place or a packed pinned copy), StateBlocks (accumulator state pinned block by block under
//                      the first pass), host_available_bytes / stream_host_need (the guard), tvdn_wait_background
//   planning           stream_planes / stream_device_bytes (what a shape costs in HBM), choose_stream_shape (R, K, rows kept),
//                      tvdn_stream_plan / tvdn_stream_host_need (the same as arithmetic for callers)
//   who keeps what     RowMap (which rows stay resident; a slab's window), SlabShare / SlabBarrier (a slab of a device list or of a
//                      multi-process run), slab_shape / tvdn_slab_host_need / tvdn_slab_row_map (a rank's packed local arrays)
//   run_streamed       set-up (helper threads: `pinner` page-locks, `stager` uploads resident rows' data term; rings, boxes and
//                      store carved from one kept device block), then one of two schedules over the same rings:
//                        `pass`   one drained pass (periodic cubes, slabs): upload chunk c + 1, scatter into level 0, K sweeps
//                                 trailing each other by a row, gather level K, download; exchange / all-reduce / row-0 hooks
//                        `chain`  Jia-Zhao on one device: several passes stacked into one running row index, downloads by a
//                                 copy kernel, the last pass sending resident rows' results home itself
//                      then results home, stats, teardown in an order that does not stall
//   run_streamed_slabs a device list: shared host arrays (two sets), one thread per slab, a barrier per pass, the row-0 mailbox
//   run_streamed_rank  one process per GPU: packed local arrays of halo + host rows + halo, the caller's hooks between passes
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>

#include <sys/mman.h>
#include <unistd.h>

#include "tvdn_common.hpp"

namespace tvdn {

namespace {

constexpr int kHostThreads = 8;
constexpr size_t kPinInPlaceMinDefault = size_t(256) << 20;  // bytes from which a caller's array is page-locked in place
size_t env_bytes(const char *name);

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

// Touch every page of [p, p + bytes) from `threads` threads (writing back what is read: contents are kept).
void touch_pages(char *p, size_t bytes, int threads)
{
    if (!bytes) return;
    const size_t piece = ((bytes + threads - 1) / threads + 4095) / 4096 * 4096;
    std::vector<std::thread> th;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t len = std::min(piece, bytes - off);
        th.emplace_back([=] {
            volatile char *q = p + off;
            for (size_t o = 0; o < len; o += 4096) q[o] = q[o];
            q[len - 1] = q[len - 1];
        });
    }
    for (auto &t : th) t.join();
}

int touch_threads()
{
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hc ? hc : 4u));
}

// Releases of pinned memory run on detached threads (unregistering and unmapping 16 GiB takes 0.8 s; a run that held 144 GiB
// would spend 7 s returning it): the next streamed run waits for them before it counts the host's memory.
std::atomic<int> g_releases_pending{0};

void wait_for_releases()
{
    while (g_releases_pending.load() > 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
}

// Page-locked host memory the library owns.  Large buffers are anonymous memory with huge pages asked for, first touched by
// many threads, page-locked with ONE registration: 16 GiB in 0.13 s on the MI355X boxes of this pool, where hipHostMalloc of
// the same size takes 3.0 s and hipHostFree 2.0 s (tools/ubench/pin_probe.hip, profiles/r04_pin_probe.jsonl) -- page-locking
// used to be most of a streamed run's set-up.  Same PCIe rate either way (57.6 GB/s one way).
struct PinnedBuf {
    char *p = nullptr;
    size_t bytes = 0;
    bool mapped = false;
    int alloc(size_t b)
    {
        bytes = b;
        if (b >= (size_t(8) << 20) && !getenv("TVDN_PIN_HIPMALLOC")) {
            void *m = mmap(nullptr, b, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (m != MAP_FAILED) {
                (void)madvise(m, b, MADV_HUGEPAGE);
                touch_pages((char *)m, b, touch_threads());
                if (hipHostRegister(m, b, hipHostRegisterDefault) == hipSuccess) {
                    p = (char *)m;
                    mapped = true;
                    return TVDN_OK;
                }
                (void)hipGetLastError();
                (void)munmap(m, b);
            }
        }
        TVDN_HIP(hipHostMalloc((void **)&p, b ? b : 1, hipHostMallocDefault));
        return TVDN_OK;
    }
    void release_now()  // on the calling thread
    {
        if (!p) return;
        if (mapped) {
            (void)hipHostUnregister(p);
            (void)munmap(p, bytes);
        } else {
            (void)hipHostFree(p);
        }
        p = nullptr;
    }
    void release()
    {
        if (!p) return;
        if (bytes < (size_t(1) << 30)) {
            release_now();
            return;
        }
        std::vector<std::unique_ptr<PinnedBuf>> one;
        one.emplace_back(new PinnedBuf);
        one[0]->p = p;
        one[0]->bytes = bytes;
        one[0]->mapped = mapped;
        p = nullptr;
        release_in_background(std::move(one));
    }
    // ONE thread for a whole batch: unmapping holds the process's address-space lock, and a thread per buffer would wait for
    // it at creation (its stack is a mapping too) -- 36 blocks of 4 GiB took 5.8 s to hand over that way.
    static void release_in_background(std::vector<std::unique_ptr<PinnedBuf>> &&bufs)
    {
        if (bufs.empty()) return;
        // ... but never past the end of the process: a thread still unpinning memory while the runtime's own exit handlers
        // run would take the process down on its way out.  Registered at first use, i.e. after the runtime's handlers, so
        // it runs before them.
        static std::once_flag at_exit_once;
        std::call_once(at_exit_once, [] { std::atexit([] { wait_for_releases(); }); });
        g_releases_pending.fetch_add(1);
        auto *batch = new std::vector<std::unique_ptr<PinnedBuf>>(std::move(bufs));
        std::thread([batch] {
            for (auto &b : *batch)
                if (b) b->release_now();
            delete batch;
            g_releases_pending.fetch_sub(1);
        }).detach();
    }
    ~PinnedBuf() { release(); }
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf &) = delete;
    PinnedBuf &operator=(const PinnedBuf &) = delete;
};

// A cube-shaped host array the GPU can reach.  Either the caller's own memory page-locked in place (`cube_rows`: row g of the
// cube at p + g * row_bytes), or pinned memory of the library's holding ONLY the rows that stay on the host, packed (the h-th
// host row at p + h * row_bytes), filled from / copied back to the caller's array by the user of this struct.
struct HostArr {
    char *p = nullptr;
    bool registered = false, owned = false, cube_rows = false;
    PinnedBuf buf;
    // `fresh`: the array's contents do not matter yet (the result array): its pages are touched first, with huge pages asked
    // for, so that the registration finds them in place (registering untouched memory faults it in page by page: 1.4 s per
    // 16 GiB against 0.1 + 0.04 s)
    int pin_in_place(void *user, size_t bytes, size_t packed_bytes, bool fresh)
    {
        // Only arrays big enough to own their pages: page-locking works on whole pages, and two small arrays of the
        // caller may share one (overlapping registrations).  Small cubes are staged through pinned copies instead.
        const size_t pin_min = getenv("TVDN_PIN_IN_PLACE_MIN") ? env_bytes("TVDN_PIN_IN_PLACE_MIN") : kPinInPlaceMinDefault;  // (tests lower it)
        if (bytes < pin_min) return alloc(packed_bytes);
        if (fresh) {
            const uintptr_t lo = ((uintptr_t)user + (size_t(2) << 20) - 1) & ~((uintptr_t)(size_t(2) << 20) - 1);
            const uintptr_t hi = ((uintptr_t)user + bytes) & ~((uintptr_t)(size_t(2) << 20) - 1);
            if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
            touch_pages((char *)user, bytes, touch_threads());
        }
        const hipError_t e = hipHostRegister(user, bytes, hipHostRegisterDefault);
        if (e == hipSuccess) {
            p = (char *)user;
            registered = cube_rows = true;
            return TVDN_OK;
        }
        (void)hipGetLastError();
        if (e == hipErrorHostMemoryAlreadyRegistered) {
            // The runtime says so for ANY overlap with an existing registration, a partial one too -- and the copy kernels
            // write these addresses straight from the GPU: an unregistered page among them is a fault that kills the process
            // (no XNACK).  Page-locked by the caller only if the device can address its first AND last byte; else a pinned
            // copy of our own.
            void *d0 = nullptr, *d1 = nullptr;
            if (hipHostGetDevicePointer(&d0, user, 0) == hipSuccess && hipHostGetDevicePointer(&d1, (char *)user + bytes - 1, 0) == hipSuccess) {
                p = (char *)user;
                cube_rows = true;
                return TVDN_OK;
            }
            (void)hipGetLastError();
        }
        return alloc(packed_bytes);
    }
    int alloc(size_t packed_bytes)
    {
        const int rc = buf.alloc(packed_bytes);
        if (rc) return rc;
        p = buf.p;
        owned = true;
        return TVDN_OK;
    }
    void release()
    {
        if (registered) (void)hipHostUnregister(p);
        registered = false;
        buf.release();
    }
    ~HostArr() { release(); }
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
    void release()
    {
        for (hipStream_t *s : {&main, &up, &down})
            if (*s) {
                (void)hipStreamDestroy(*s);
                *s = nullptr;
            }
    }
    ~Streams() { release(); }
};

struct CtxHolder {
    tvdn_ctx *c = nullptr;
    void release()
    {
        if (c) (void)tvdn_ctx_destroy(c);
        c = nullptr;
    }
    ~CtxHolder() { release(); }
};

struct DevMem {
    void *p = nullptr;
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
    }
    ~DevMem() { release(); }
};

// "64G" / "512M" / bytes from the environment; 0 = not set
size_t env_bytes(const char *name)
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
    const size_t cap = env_bytes("TVDN_HOST_LIMIT");  // a cap from outside, "64G" / "512M" / bytes (the test-suite sets one)
    if (cap && cap < avail) avail = cap;
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
static int64_t stream_planes(int nd, int64_t rows, int64_t k, bool mse, bool wrap)
{
    return ((k + 1) + (k + 2) * nd) * (rows + 2) + (rows + k + 3) * (mse ? 2 : 1) + 2 * (3 + 4 * nd) * rows + 2 * (1 + 2 * nd) + (wrap ? 2 * (k + 1) : 0) + 1;
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
                        int64_t k_cap, int64_t *rows_out, int64_t *k_out, int64_t *res_out)
{
    const int64_t budget = (int64_t)(0.85 * (double)free_bytes / (double)row_bytes);
    const double rb = (double)row_bytes;
    const int n_in = 2 + nd * n_state, n_out = 1 + nd * n_state, moved = 3 + nd * (n_state + 1);
    k_cap = std::max<int64_t>(1, std::min<int64_t>({k_cap, 128, std::max<int64_t>(1, n_rows)}));
    const double row_step = std::max((double)n_in * rb / 55e9, (double)n_out * rb / 42.5e9);  // one streamed row, both directions busy
    int64_t best_k = 0, best_r = 0, best_res = 0;
    double best_t = 0.0;
    for (int64_t r : {32, 16, 8, 4, 2, 1}) {
        if (r > 1) r = std::min<int64_t>(r, std::max<int64_t>(2, n_rows));
        // sweeps on rings, launches of r rows: 0.86 ms per 256 MiB plane and level whether r is 2, 4 or 8 (83 % of the resident
        // sweep's rate; all rows resident, profiles/r04_stream_rates.jsonl), 0.92 ms in one-row launches
        const double eff = r == 1 ? 0.77 : 0.82;
        for (int64_t k = 1; k <= k_cap; ++k) {
            const int64_t planes = stream_planes(nd, r, k, mse, wrap);
            if (planes > budget) break;
            const double t_sweeps = (double)n_rows * (double)k * moved * rb / (5.6e12 * eff);
            auto offer = [&](int64_t res, double t_pass) {
                const double t = t_pass / (double)k;
                if (best_k == 0 || t < best_t * 0.999) {
                    best_t = t;
                    best_k = k;
                    best_r = r;
                    best_res = res;
                }
            };
            // (a) nothing kept: the pipeline of a pass fills and drains over K rows; chained passes share that between them
            //     (half of it counted)
            offer(0, std::max(((double)n_rows + 0.5 * (double)std::min<int64_t>(k, n_rows)) * row_step, t_sweeps));
            // (b) what the rest of the budget holds kept.  The streamed rows are spread over a pass that is drained, both
            //     directions together at 60 GB/s (round 4's first model, which the kept-row measurements were planned and
            //     verified with).  Not with one-row chunks: (1, 20, 47 kept) ran at 46.6 Gvoxel-iters/s where (2, 12, 56 kept)
            //     runs at 57.8 (profiles/r04_stream_rates.jsonl).
            const int64_t res = may_keep && r > 1 ? std::min<int64_t>(n_rows, (budget - planes) / n_in) : 0;
            if (res > 0)
                offer(res, std::max((double)(n_rows - res) * (n_in + n_out) * rb / 60e9,
                                    t_sweeps + (double)res * (n_in + n_out) * 2.0 * rb / 4.8e12));
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
static int stream_host_need(const tvdn_run_args *a, int64_t res, int64_t *need_bytes, int64_t *avail_bytes)
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
                                       a->use_stop ? 1 : (n_total > 0 ? n_total : 128), &rows, &k, &res);
    if (rc) return rc;
    if (a->stream_resident > 0) res = std::min<int64_t>(res, a->stream_resident);
    out->rows = rows;
    out->k = k;
    out->resident_rows = res;
    out->hbm_bytes = (stream_planes(a->ndim, rows, k, mse, true) + res * (2 + a->ndim * n_state)) * (int64_t)row_bytes;
    int64_t need = 0, avail = 0;
    (void)stream_host_need(a, res, &need, &avail);
    out->host_bytes = need;
    return TVDN_OK;
}

namespace tvdn {

// Threads of one streamed device-list run meet here after every pass; a slab that fails releases the others with its error.
struct SlabBarrier {
    std::mutex mu;
    std::condition_variable cv;
    int count = 1, waiting = 0;
    long generation = 0;
    int failed = 0;
    std::string msg;
    int arrive_and_wait()  // TVDN_OK, or the status of the slab that failed
    {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) return failed;
        const long gen = generation;
        if (++waiting == count) {
            waiting = 0;
            ++generation;
            cv.notify_all();
        } else {
            cv.wait(lk, [&] { return generation != gen || failed; });
        }
        return failed;
    }
    void fail(int rc, const char *m)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) {
            failed = rc;
            msg = m ? m : "";
        }
        cv.notify_all();
    }
};

struct SlabShare {
    int index = 0, count = 1;
    int64_t g0 = 0, g1 = 0;      // own rows of the cube
    char *orig = nullptr;        // page-locked, row g of the cube at + g * row_bytes (every array below too)
    char *ref = nullptr;
    char *recon[2] = {nullptr, nullptr};
    char *state[2][8] = {};      // [set][axis * n_state + s]
    int first_new = 1;           // the set the FIRST pass writes
    SlabBarrier *barrier = nullptr;
    double *stop_sums = nullptr; // [count][3]: every slab's sums of the iteration just run (stopping rule)
    int *last_set = nullptr;     // out: the set the last pass wrote
    tvdn_run_stats *stats = nullptr;
    // One slab per PROCESS (run_streamed_rank): the arrays are this process's own -- halo + own + halo rows, virtual row v at
    // + (v - local_v0) * row_bytes, both "sets" the same arrays (a pass writes its own rows k rows behind where it reads) --
    // and what crosses process boundaries goes through the caller's hooks.
    bool local_rows = false;
    int64_t local_v0 = 0;
    // ... of which `resident_rows` interior own rows (none within K of a face shared with a neighbour) keep their state in HBM
    // between passes and have NO slot in the local arrays (those are packed: local slot of virtual row v = v - local_v0 -
    // resident rows below it); their data term comes from / their result goes to the caller's own-row arrays directly
    int64_t resident_rows = 0;
    const char *own_data = nullptr;
    char *own_recon = nullptr;
    bool exact_wrap = false;                             // Jia-Zhao, first row of the cube not finite (the same on every slab)
    std::function<int()> before_pass;                    // before every pass but the first: refresh the halo rows of recon / state
    std::function<int(double *)> allreduce;              // one iteration's three sums -> over all slabs (stopping rule)
    std::function<int(int, void *, int)> relay_row0;     // (send, planes, n): row 0 of every level, first slab -> last slab
};

namespace {

// Which rows of axis 0 keep their state in HBM between the passes: `res` of the n0 rows, spread EVENLY over the cube
// (row g is one of them when floor((g+1) res / n0) > floor(g res / n0)), so that every chunk of a pass has the same share of
// rows that cross PCIe and the transfers of one chunk hide under the sweeps of the one before.  (With the resident rows in
// one piece at the low end, the rest of a pass is PCIe-bound chunk after chunk while the link idles under the resident ones:
// 35 Gvoxel-iters/s on config-5 planes where the evenly spread rows give 61; profiles/r04_stream_rates.jsonl.)
struct RowMap {
    int64_t n0 = 1, res = 0;
    int64_t e0 = 0, e1 = -1;  // the rows that may be resident: [e0, e1) (e1 < 0: the whole cube).  A slab of a multi-process run
                              // keeps the rows its neighbours read -- K at each shared face -- on the host, where the exchange
                              // hook finds them.
    int64_t res_below(int64_t g) const  // resident rows among [0, g); any g, also beyond the cube (halo rows of a slab)
    {
        const int64_t hi = e1 < 0 ? n0 : e1, n = hi - e0;
        if (n <= 0 || res <= 0) return 0;
        const int64_t x = std::min(std::max(g, e0), hi) - e0;
        return res >= n ? x : x * res / n;
    }
    bool resident(int64_t g) const { return res_below(g + 1) > res_below(g); }
    int64_t host_below(int64_t g) const { return g - res_below(g); }  // host rows among [0, g)
    int64_t host_rows() const { return n0 - res; }
    // the window of a slab [g0, g1) whose passes are `depth` levels deep: everything but the `depth` rows at a shared face
    void slab_window(int64_t g0, int64_t g1, bool shared_lo, bool shared_hi, int64_t depth)
    {
        e0 = g0 + (shared_lo ? depth : 0);
        e1 = std::max(e0, g1 - (shared_hi ? depth : 0));
    }
};

// Accumulator-state rows that live on the host, in pinned memory allocated BLOCK BY BLOCK (in row order) by a helper thread
// while the first pass is already running: the first pass writes these rows long before any pass reads them.  Indexed by
// HOST SLOT (the h-th row that lives on the host), not by cube row.
struct StateBlocks {
    int n_arr = 0;
    int64_t n_slots = 0, block_rows = 1;
    size_t row_bytes = 0;
    std::vector<std::unique_ptr<PinnedBuf>> blocks;  // one pinned allocation per block: n_arr x block_rows rows, array-major
    std::mutex mu;
    std::condition_variable cv;
    int64_t ready = 0;  // blocks [0, ready) exist
    int failed = 0;
    std::string fail_msg;
    std::vector<char *> flat;  // slab mode: array `arr` is ONE caller-provided run of n_slots rows (nothing allocated, always ready)
    int64_t n_blocks() const { return flat.empty() ? (n_slots + block_rows - 1) / block_rows : 0; }
    int64_t block_of(int64_t h) const { return flat.empty() ? h / block_rows : 0; }
    int64_t block_end(int64_t h) const { return flat.empty() ? std::min(n_slots, (block_of(h) + 1) * block_rows) : n_slots; }  // first slot of the next block
    char *row(int arr, int64_t h) const
    {
        if (!flat.empty()) return flat[(size_t)arr] + (size_t)h * row_bytes;
        const int64_t b = block_of(h);
        return blocks[(size_t)b]->p + ((size_t)arr * (size_t)block_rows + (size_t)(h - b * block_rows)) * row_bytes;
    }
    int allocate(int64_t b)  // helper thread
    {
        std::unique_ptr<PinnedBuf> pb(new PinnedBuf);
        const int rc = pb->alloc((size_t)n_arr * (size_t)block_rows * row_bytes);
        std::lock_guard<std::mutex> lk(mu);
        if (rc) {
            failed = rc;
            fail_msg = std::string("page-locking a block of host state failed: ") + tvdn_last_error();
        } else {
            blocks[(size_t)b] = std::move(pb);
            ready = b + 1;
        }
        cv.notify_all();
        return failed;
    }
    int wait_for(int64_t h)  // calling thread: until the block of host slot h exists
    {
        if (!flat.empty()) return TVDN_OK;
        const int64_t b = block_of(h);
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return failed || ready > b; });
        if (failed) {
            set_error("%s", fail_msg.c_str());
            return failed;
        }
        return TVDN_OK;
    }
    void fail(int rc, const std::string &msg)
    {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) {
            failed = rc;
            fail_msg = msg;
        }
        cv.notify_all();
    }
};

// A one-shot flag a helper thread raises (with an error, if any)
struct Flag {
    std::mutex mu;
    std::condition_variable cv;
    bool up = false;
    int rc = TVDN_OK;
    std::string msg;
    void raise(int rc_ = TVDN_OK, const std::string &m = std::string())
    {
        std::lock_guard<std::mutex> lk(mu);
        if (up) return;
        up = true;
        rc = rc_;
        msg = m;
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return up; });
        if (rc) set_error("%s", msg.c_str());
        return rc;
    }
};

struct Joiner {
    std::thread &t;
    ~Joiner()
    {
        if (t.joinable()) t.join();
    }
};

}  // namespace

// HBM bytes of everything a streamed run keeps on the device besides resident rows -- rings of R + 2 rows per level and array,
// the data-term ring(s), two in and two out boxes, the planes of an exact wrap, the plane of zeros -- and of ONE resident row
// (data term, recon, accumulator state).  One definition: run_streamed allocates by it, run_streamed_rank sizes a slab's
// packed host arrays by it before run_streamed runs.
static size_t stream_device_bytes(int nd, int n_state, bool want_mse, int64_t R, int64_t K, size_t row_bytes, size_t *per_resident_row)
{
    auto aligned = [](size_t b) { return (b + 255) / 256 * 256; };
    const int n_in = 2 + nd * n_state + (want_mse ? 1 : 0), n_out = 1 + nd * n_state, n_store = 2 + nd * n_state;
    const size_t ring_b = aligned((size_t)(R + 2) * row_bytes), oring_b = aligned((size_t)(R + K + 3) * row_bytes);
    const size_t box_b = aligned((size_t)R * row_bytes), obox_b = aligned((size_t)(R + 1) * row_bytes), plane_b = aligned(row_bytes);
    const size_t n_rings = (size_t)(K + 1) + (size_t)(K + 2) * nd;
    if (per_resident_row) *per_resident_row = (size_t)n_store * plane_b;
    return n_rings * ring_b + oring_b * (want_mse ? 2 : 1) + 2 * ((size_t)n_in * box_b + (size_t)n_out * obox_b) + 2 * (size_t)(K + 1) * plane_b + plane_b;
}

// R rows per chunk, K iteration levels per pass; `res_req` rows keep their state in HBM between passes (-1: as many as fit
// beside the rings in 85 % of the free HBM, 0: none).
int run_streamed(const tvdn_run_args *a, int64_t R, int64_t K, int64_t res_req, const SlabShare *sh)
{
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
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
    TVDN_REQUIRE(res_req >= -1, "stream_resident must be -1 (as many rows as fit), 0 (none) or a row count");
    const bool aliased = arrays_overlap(a->data, a->recon_out, cube_bytes);
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
        return TVDN_OK;
    }
    if (a->use_stop) K = 1;  // the stopping rule needs a decision after every iteration: one level per pass
    K = std::min<int64_t>(K, std::max<int64_t>(1, n_total));
    if (periodic || sh) K = std::min<int64_t>(K, N0);  // (the wrapped rows / the neighbours' rows a pass reads are rows of this cube)
    // as many passes as this depth needs, of (almost) equal depth: 80 iterations at k = 38 are three PCIe round trips whether
    // they hold 38 + 38 + 4 levels or 27 + 27 + 26, and the shallower rings leave HBM for resident rows.  The deeper passes
    // come first and consecutive depths differ by one level at most (what the out boxes of chained passes are sized for).
    const int n_pass_plan = a->use_stop ? n_total : (int)((n_total + K - 1) / K);
    auto depth_of_pass = [&](int q) { return a->use_stop ? 1 : n_total / n_pass_plan + (q < n_total % n_pass_plan ? 1 : 0); };
    if (!a->use_stop) K = depth_of_pass(0);
    // Periodic boundaries along axis 0: the sweeps see a virtual cube of N0 + 2 K rows -- the cube between K wrapped rows
    // at either end, which are each other's halo -- and, as at the face between two slabs, give up one row per level at
    // the two artificial faces; the wrap itself is never swept.
    // A slab of a device-list run (sh): the same virtual rows -- the cube between K rows at either end -- of which this slab
    // owns [own0, own1); at its interior faces it reads K rows of its neighbours' state (shared host arrays) and gives up a row
    // per level, as a periodic run does at both ends.
    const int64_t KX = (periodic || sh) ? K : 0, NV = N0 + 2 * KX, G0 = KX, G1 = KX + N0;
    const int64_t own0 = sh ? KX + sh->g0 : G0, own1 = sh ? KX + sh->g1 : G1;  // virtual rows whose results and sums are this run's
    const bool art_lo = periodic || (sh && sh->g0 > 0), art_hi = periodic || (sh && sh->g1 < N0);  // faces that are not the cube's own
    if (sh) res_req = sh->local_rows ? sh->resident_rows : 0;  // (a device list shares host arrays indexed by cube row: none kept)
    TVDN_HIP(hipSetDevice(device));

    // ---- what fits where: rings and boxes first, then as many resident rows as asked for / as fit ---------------------------
    const int64_t cap = R + 2, ocap = R + K + 3;
    const int n_in = 2 + nd * n_state + (want_mse ? 1 : 0), n_out = 1 + nd * n_state, n_store = 2 + nd * n_state;
    auto aligned = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t ring_b = aligned((size_t)cap * row_bytes), oring_b = aligned((size_t)ocap * row_bytes);
    const size_t box_b = aligned((size_t)R * row_bytes), plane_b = aligned(row_bytes);
    // an out box holds R + 1 rows: at the seam between two chained passes whose depths differ by one level (80 iterations in
    // three passes: 27 + 27 + 26) the last rows of one and the first rows of the next come down in the same chunk
    const size_t obox_b = aligned((size_t)(R + 1) * row_bytes);
    const size_t n_rings = (size_t)(K + 1) + (size_t)(K + 2) * nd;
    const size_t dev_bytes_max = stream_device_bytes(nd, n_state, want_mse, R, K, row_bytes, nullptr);
    (void)n_rings;
    size_t free_b = 0, total_b = 0;
    TVDN_HIP(hipMemGetInfo(&free_b, &total_b));
    free_b += state_kept_bytes(device);  // the block the last run kept is this run's to take over or to release
    if (const size_t cap_b = env_bytes("TVDN_HBM_LIMIT"))  // what the planner of the Python side counts on (tests: "48G")
        free_b = std::min(free_b, cap_b);
    if (dev_bytes_max > free_b) {
        set_error("streamed run with %lld-row chunks and k = %lld needs %zu bytes of HBM, device %d has %zu free", (long long)R,
                  (long long)K, dev_bytes_max, device, free_b);
        return TVDN_ERR_UNSUPPORTED;
    }
    // RES of the N0 rows keep their state (data term, recon, accumulators: n_store arrays) in HBM between passes: they enter
    // the rings and leave them by device copies instead of crossing PCIe.  Jia-Zhao runs without an MSE trace (the periodic
    // schedule walks a wrapped virtual cube whose ends are both streamed; the reference cube of an MSE trace stays on the host).
    RowMap rm;
    rm.n0 = N0;
    if (sh) {  // a slab keeps none of the rows its neighbours read
        rm.slab_window(sh->g0, sh->g1, art_lo, art_hi, K);
    }
    if (sh && res_req > 0) {  // a slab of a multi-process run: its coordinator has sized the packed local arrays for exactly this
        TVDN_REQUIRE(!want_mse && res_req <= rm.e1 - rm.e0, "a slab cannot keep %lld rows resident (%lld interior rows; none with an MSE trace)",
                     (long long)res_req, (long long)(rm.e1 - rm.e0));
        const size_t need_b = dev_bytes_max + (size_t)res_req * (size_t)n_store * plane_b;
        if (need_b > (size_t)(0.92 * (double)free_b)) {
            set_error("streamed slab with %lld resident rows needs %zu bytes of HBM, device %d has %zu free", (long long)res_req, need_b, device, free_b);
            return TVDN_ERR_UNSUPPORTED;
        }
        rm.res = res_req;
    } else if (!periodic && !want_mse && res_req != 0) {
        auto fits = [&](double share) -> int64_t {
            const size_t lim = (size_t)(share * (double)free_b);
            return lim > dev_bytes_max ? (int64_t)((lim - dev_bytes_max) / ((size_t)n_store * plane_b)) : 0;
        };
        rm.res = std::min<int64_t>(N0, res_req < 0 ? fits(0.85) : std::min<int64_t>(res_req, fits(0.92)));
        if (const char *e = getenv("TVDN_STREAM_RESIDENT")) rm.res = std::max<int64_t>(0, std::min<int64_t>({(int64_t)atoll(e), N0, fits(0.92)}));
    }
    const int64_t RES = rm.res, HR = N0 - RES;  // rows in HBM / rows on the host
    auto resident = [&](int64_t g) { return RES > 0 && rm.resident(g); };
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
    bool exact_wrap = false;
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
    HostArr orig_h, recon_h, ref_h, recon2_h;
    StateBlocks sb[2];
    const bool two_sets = periodic || sh != nullptr;  // old and new host state apart
    const int n_sets = two_sets ? 2 : 1;
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
    Flag orig_ready, recon_ready, recon2_ready, staged_done;
    std::atomic<int64_t> staged_upto{0};  // resident rows g < staged_upto have their data term in the store
    const size_t host_bytes = (size_t)HR * row_bytes;
    // data partly overlapping recon_out (not the same array): a download into recon_out may hit rows of `data` that have
    // not been read yet, so every input is taken out of `data` before the first pass starts
    const bool eager = (aliased && a->recon_out != a->data) || getenv("TVDN_STREAM_EAGER") != nullptr;
    auto host_row = [&](const HostArr &h, int64_t g) -> char * {
        return h.cube_rows ? h.p + (size_t)g * row_bytes : h.p + (size_t)rm.host_below(g) * row_bytes;
    };
    // host rows of a cube-shaped user array <-> a packed buffer, run of consecutive host rows by run
    auto pack_host_rows = [&](char *packed, char *cube, bool to_packed) {
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
    };

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
    std::thread pinner([&] {
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
    Joiner join_pinner{pinner};

    // ---- device: rings, staging boxes, resident rows, sums ---------------------------------------------------------------
    CtxHolder ctx;
    int rc = tvdn_ctx_create(&ctx.c, device);
    if (rc) return rc;
    Streams st;
    if ((rc = make_stream(&st.main, +1))) return rc;  // three queue classes: no false ordering between sweeps, uploads and
    if ((rc = make_stream(&st.up, 0))) return rc;     // downloads whatever other streams the process holds (tvdn_common.hpp)
    if ((rc = make_stream(&st.down, -1))) return rc;
    const size_t ring_bytes = dev_bytes_max - (exact_wrap ? 0 : 2 * (size_t)(K + 1) * plane_b);
    const size_t store_b = aligned((size_t)std::max<int64_t>(RES, 1) * row_bytes);
    const size_t dev_bytes = ring_bytes + (RES > 0 ? (size_t)n_store * store_b : 0);
    // the one big device block: the block the last run of this device kept, if it fits (tvdn_run.hip state_acquire; a kept
    // block of another size is released first, so a streamed run still has the whole HBM to itself)
    struct KeptBlock {
        void *p = nullptr;
        size_t bytes = 0;
        int device = 0;
        void release()
        {
            if (p) state_release(p, bytes, device);
            p = nullptr;
        }
        ~KeptBlock() { release(); }
    } mem;
    DevMem sums_d, mse_d;
    mem.device = device;
    const double t_before_block = since(t_start);
    bool block_reused = false;
    // (the rings of the levels are swept like a resident state, many streams at once: the same allocator, tvdn_devmem.hip)
    TVDN_HIP(state_acquire(&mem.p, dev_bytes, &mem.bytes, device, &block_reused, true,
                           std::max(0.25, 0.05 * (double)n_total * (double)N0 * (double)row_bytes * (double)(3 + 3 * nd) / 5.5e12)));
    const int block_kind = dev_kind(mem.p);  // granules or a plain block (tvdn_devmem.hip)
    const double t_block = since(t_start);
    TVDN_HIP(hipMemsetAsync(mem.p, 0, ring_bytes, st.main));
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
        for (int i = 0; i < n_out; ++i) outbox[h][i] = take(obox_b);
    }
    char *zero_plane = take(plane_b);  // the accumulator state a run starts from (cyTVDN.py:131-145): the first pass uploads none
    std::vector<char *> row0;          // row 0 of every level, kept for the top face (and a second set: two chained passes at a seam)
    char *row0b_base = nullptr;
    if (exact_wrap) {
        for (int64_t j = 0; j <= K; ++j) row0.push_back(take(plane_b));
        row0b_base = take((size_t)(K + 1) * plane_b);
    }
    // resident rows: array i of the store holds them packed (slot = resident rows below): 0 data term, 1 recon, 2 + q * n_state + s state
    std::vector<char *> store((size_t)n_store, nullptr);
    if (RES > 0)
        for (int i = 0; i < n_store; ++i) store[(size_t)i] = take(store_b);
    auto store_row = [&](int i, int64_t g) -> char * { return store[(size_t)i] + (size_t)rm.res_below(g) * row_bytes; };
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

    std::thread stager([&] {  // resident rows of the data term: pageable `data` -> store, through the library's pinned lanes
        if (sh && RES <= 0) {
            staged_upto.store(N0);
            staged_done.raise();
            return;
        }
        // (a slab of a multi-process run: the caller's array holds its OWN rows only, row g at + (g - first own row))
        const char *data_rows = sh ? sh->own_data - (size_t)sh->g0 * row_bytes : (const char *)a->data;
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
    Joiner join_stager{stager};
    if (eager) {
        if ((rc = staged_done.wait()) || (rc = orig_ready.wait())) return rc;
    }
    auto wait_staged = [&](int64_t upto) -> int {  // the resident rows below `upto` have their data term in the store
        while (true) {
            const int64_t v = staged_upto.load();
            if (v < 0) return staged_done.wait();
            if (v >= upto) return TVDN_OK;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    };

    int h_old = sh ? sh->first_new ^ 1 : 0;  // which set holds the current state (two sets: periodic runs, slabs); 0 = recon_h / sb[0]
    auto wait_recon = [&](int set) -> int { return ((two_sets && set) ? recon2_ready : recon_ready).wait(); };
    // row of a host array by cube row g / virtual row v (a slab of its own process addresses its local arrays by v)
    const bool local_rows = sh && sh->local_rows;
    auto local_slot = [&](int64_t v) { return (v - sh->local_v0) - rm.res_below(v - KX); };  // packed: resident rows have no slot
    auto hrow = [&](const HostArr &h, int64_t g, int64_t v) -> char * {
        return local_rows ? h.p + (size_t)local_slot(v) * row_bytes : host_row(h, g);
    };
    auto srow = [&](int set, int arr, int64_t g, int64_t v) -> char * {
        return local_rows ? sb[set].flat[(size_t)arr] + (size_t)local_slot(v) * row_bytes : sb[set].row(arr, rm.host_below(g));
    };

    tvdn_iter_args it;
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
    int64_t one_row[4];
    one_row[0] = 1;
    for (int i = 1; i < nd; ++i) one_row[i] = a->shape[i];
    auto sse_row = [&](const char *x, const char *y, int slot, int64_t g) -> int {
        return tvdn_sum_square_error(ctx.c, a->dtype, nd, one_row, x, y, (double *)mse_d.p + (size_t)slot * (size_t)N0 + (size_t)g,
                                     st.main);
    };

    PinnedBuf row0_host;  // exact wrap across processes: row 0 of every level on its way from the first slab to the last
    if (exact_wrap && sh && sh->relay_row0 && (rc = row0_host.alloc((size_t)(K + 1) * row_bytes))) return rc;

    // ---- one pass: `kk` iteration levels over the whole cube ------------------------------------------------------------
    bool d_form = fista;
    double tk_prev = 0.0;
    int done = 0;
    int64_t bytes_up = 0, bytes_down = 0, n_passes = 0;  // across PCIe (tvdn_run_stats)
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
        // The first pass of a run starts from recon = data term and all-zero accumulators (cyTVDN.py:131-145): neither is
        // uploaded -- the level-0 rows of recon are device copies of the data-term rows, those of the state copies of a plane
        // of zeros -- so the host arrays they will come down into need not exist yet.
        const bool first = n_passes == 0;
        if (sh && sh->before_pass && !first) {  // a slab of its own process: the neighbours' new rows into my halo rows
            const int rcb = sh->before_pass();
            if (rcb) return rcb;
        }
        // exact Jia-Zhao wrap across processes: the slab that owns row 0 sends row 0 of every level to the one that owns the
        // top face, once per pass (hooks of the caller; a slab in between has nothing to do with it)
        // Exact wrap over several slabs: the slab that owns row 0 hands row 0 of every level of the pass to EVERY other slab,
        // once per pass (a broadcast: every slab takes part).  Used by the slabs whose sweeps reach the cube's top face without
        // sweeping row 0 themselves -- the last slab, and any slab whose K-row halo reaches that far.
        const bool relay_send = exact_wrap && sh && sh->relay_row0 && sh->g0 == 0 && sh->g1 < N0;
        const bool relay_recv = exact_wrap && sh && sh->relay_row0 && sh->g0 > 0;
        int planes_ready = 0;
        bool relayed = false;
        // rows of the (virtual) cube this pass works on, and what each level can reach at an artificial face
        // (an artificial face -- the wrap of a periodic run, the face between two slabs -- gives up a row per level; a slab
        // whose halo would reach beyond a Jia-Zhao cube's own face stops at that face, which then is a real one)
        const int64_t E0 = art_lo ? (periodic ? own0 - kk : std::max(G0, own0 - kk)) : G0;
        const int64_t E1 = art_hi ? (periodic ? own1 + kk : std::min(G1, own1 + kk)) : G1;
        const bool shrink_lo = art_lo && (periodic || E0 > G0), shrink_hi = art_hi && (periodic || E1 < G1);
        auto lo_bound = [&](int64_t level) { return shrink_lo ? E0 + level : E0; };
        auto hi_bound = [&](int64_t level) { return shrink_hi ? E1 - level : E1; };
        const int64_t n_chunks = (E1 - E0 + kk + R - 1) / R;
        const int h_new = two_sets ? h_old ^ 1 : h_old;
        auto cube_row = [&](int64_t v) { return ((v - KX) % N0 + N0) % N0; };  // virtual row -> row of the cube

        // The host rows among the virtual rows [v0, v1) -> consecutive rows of a box, run by run: a run ends where the next
        // row is resident, where the cube wraps and where `contiguous(g, g + 1)` says the host memory is not in one piece.
        auto up_rows = [&](char *box, int64_t v0, int64_t v1, const std::function<char *(int64_t, int64_t)> &src_row,
                           const std::function<bool(int64_t)> &joins_next) -> int {
            int64_t slot = 0;
            for (int64_t v = v0; v < v1;) {
                const int64_t g = cube_row(v);
                if (resident(g)) {
                    ++v;
                    continue;
                }
                int64_t n = 1;
                while (v + n < v1 && g + n < N0 && !resident(g + n) && joins_next(g + n - 1)) ++n;
                TVDN_HIP(hipMemcpyAsync(box + (size_t)slot * row_bytes, src_row(g, v), (size_t)n * row_bytes, hipMemcpyHostToDevice, st.up));
                bytes_up += n * (int64_t)row_bytes;
                slot += n;
                v += n;
            }
            return TVDN_OK;
        };
        auto host_rows_in = [&](int64_t v0, int64_t v1) {
            int64_t n = 0;
            for (int64_t v = v0; v < v1; ++v) n += resident(cube_row(v)) ? 0 : 1;
            return n;
        };
        auto upload = [&](int64_t c) -> int {
            const int64_t u0 = E0 + c * R, u1 = std::min(E0 + (c + 1) * R, E1);
            if (u0 >= u1 || host_rows_in(u0, u1) == 0) return TVDN_OK;
            int rcu = orig_ready.wait();
            if (rcu) return rcu;
            const int h = (int)(c % 2);
            if (in_free_set[h]) TVDN_HIP(hipStreamWaitEvent(st.up, in_free[h], 0));
            auto joins = [&](const HostArr &ha) {  // in place: cube rows g and g+1 are adjacent; packed: host slots are
                return std::function<bool(int64_t)>([&ha](int64_t) { (void)ha; return true; });
            };
            int i = 0;
            if ((rcu = up_rows(inbox[h][i++], u0, u1, [&](int64_t g, int64_t v) { return hrow(orig_h, g, v); }, joins(orig_h)))) return rcu;
            if (!first) {
                if ((rcu = wait_recon(h_old))) return rcu;
                const HostArr &ro = (two_sets && h_old) ? recon2_h : recon_h;
                if ((rcu = up_rows(inbox[h][i++], u0, u1, [&](int64_t g, int64_t v) { return hrow(ro, g, v); }, joins(recon_h)))) return rcu;
                for (int q = 0; q < nd; ++q)
                    for (int s = 0; s < n_in_state; ++s) {
                        const int arr = q * n_state + s;
                        if ((rcu = up_rows(inbox[h][i++], u0, u1, [&](int64_t g, int64_t v) { return srow(h_old, arr, g, v); },
                                           [&](int64_t g) { return sb[h_old].block_of(rm.host_below(g)) == sb[h_old].block_of(rm.host_below(g + 1)); })))
                            return rcu;
                    }
            } else {
                i += 1 + nd * n_in_state;
            }
            if (want_mse && (rcu = up_rows(inbox[h][i++], u0, u1, [&](int64_t g, int64_t v) { return hrow(ref_h, g, v); }, joins(ref_h)))) return rcu;
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
                const bool from_host = host_rows_in(u0, u1) > 0;
                if (from_host) TVDN_HIP(hipStreamWaitEvent(st.main, in_ready[h], 0));
                if (first && RES > 0 && (rc2 = wait_staged(std::min<int64_t>(N0, std::max<int64_t>(0, u1 - KX))))) return rc2;
                cdst.clear();
                csrc.clear();
                // row v of ring `rg` <- array `i_store` of the store (resident rows), box `i_box` (host rows, in their order),
                // or, in the first pass, the data-term row (recon) / the plane of zeros (state)
                auto scatter = [&](const Ring &rg, int i_store, int i_box, bool from_first, bool zeros) {
                    int64_t slot = 0;
                    for (int64_t v = u0; v < u1; ++v) {
                        const int64_t g = cube_row(v);
                        const bool res_row = resident(g);
                        const char *src;
                        if (from_first && zeros)
                            src = zero_plane;
                        else if (from_first)  // recon <- data term
                            src = res_row ? store_row(0, g) : inbox[h][0] + (size_t)slot * row_bytes;
                        else
                            src = res_row ? store_row(i_store, g) : inbox[h][i_box] + (size_t)slot * row_bytes;
                        if (!res_row) ++slot;
                        cdst.push_back(rg.row(v));
                        csrc.push_back((void *)src);
                    }
                };
                int i = 0;
                scatter(Ow, 0, i++, false, false);
                scatter(Rw[0], 1, i++, first, false);
                for (int q = 0; q < nd; ++q) {
                    scatter(A(0, q), 2 + q * n_state, i++, first, true);                            // level 0: d_k (or b)
                    if (n_in_state == 2) scatter(A(-1, q), 2 + q * n_state + 1, i++, first, true);  // level -1: d_k-1
                }
                if (want_mse) scatter(Fw, -1, i++, false, false);
                rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
                if (rc2) return rc2;
                if (exact_wrap && u0 <= G0 && G0 < u1) {
                    TVDN_HIP(hipMemcpyAsync(row0[0], Rw[0].row(G0), row_bytes, hipMemcpyDeviceToDevice, st.main));
                    planes_ready = 1;
                }
                if (want_mse && first)  // MSE[0]: the input against the reference (cyTVDN.py:124-125), own rows
                    for (int64_t g = std::max(u0, own0); g < std::min(u1, own1); ++g)
                        if ((rc2 = sse_row(Rw[0].row(g), Fw.row(g), 0, g - KX))) return rc2;
                if (from_host) {
                    TVDN_HIP(hipEventRecord(in_free[h], st.main));
                    in_free_set[h] = true;
                }
            }
            // the wavefront: level j+1 trails level j by one row
            for (int j = 0; j < kk; ++j) {
                const int64_t lo = std::max(lo_bound(j + 1), E0 + c * R - (j + 1)), hi = std::min(hi_bound(j + 1), E0 + (c + 1) * R - (j + 1));
                if (lo >= hi) continue;
                if (relay_recv && !relayed && hi == G1 && E0 > G0) {  // my first sweep at the cube's top face: row 0 of every level, from its owner
                    int rcr = sh->relay_row0(0, row0_host.p, kk);
                    if (rcr) {
                        set_error("the row-0 relay of a slab run failed (status %d)", rcr);
                        return TVDN_ERR_INVALID;
                    }
                    for (int q = 0; q < kk; ++q)
                        TVDN_HIP(hipMemcpyAsync(row0[(size_t)q], row0_host.p + (size_t)q * row_bytes, row_bytes, hipMemcpyHostToDevice, st.main));
                    relayed = true;
                }
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
                const int64_t parts[3][2] = {{lo, std::min(hi, own0)}, {std::max(lo, own0), std::min(hi, own1)}, {std::max(lo, own1), hi}};
                for (int part = 0; part < 3; ++part) {
                    const int64_t p0 = parts[part][0], p1 = parts[part][1];
                    if (p0 >= p1) continue;
                    it.sweep_lo = p0;
                    it.sweep_hi = p1;
                    const int slot = part == 1 ? done + j : discard;
                    rc2 = tvdn_iterate_fused(ctx.c, &it, (double *)sums_d.p + 3 * (size_t)slot, st.main);
                    if (rc2) return rc2;
                    if (want_mse && part == 1)
                        for (int64_t g = p0; g < p1; ++g)
                            if ((rc2 = sse_row(Fw.row(g), Rw[j + 1].row(g), done + j + 1, g - KX))) return rc2;
                }
                if (exact_wrap && lo == G0) {
                    TVDN_HIP(hipMemcpyAsync(row0[j + 1], Rw[j + 1].row(G0), row_bytes, hipMemcpyDeviceToDevice, st.main));
                    planes_ready = j + 2;
                }
            }
            if (relay_send && !relayed && planes_ready >= kk) {  // planes 0 .. kk-1 are what the top face's sweeps read
                for (int q = 0; q < kk; ++q)
                    TVDN_HIP(hipMemcpyAsync(row0_host.p + (size_t)q * row_bytes, row0[(size_t)q], row_bytes, hipMemcpyDeviceToHost, st.main));
                TVDN_HIP(hipStreamSynchronize(st.main));
                const int rcr = sh->relay_row0(1, row0_host.p, kk);
                if (rcr) {
                    set_error("the row-0 relay of a slab run failed (status %d)", rcr);
                    return TVDN_ERR_INVALID;
                }
                relayed = true;
            }
            // rows that have reached the last level go home: resident rows into the store (device copies, in the same launch
            // as the gather of the others into the out box), the others across PCIe.  [lo, hi) are rows of the cube proper.
            const int64_t lo = std::max(own0, E0 + c * R - kk), hi = std::min(own1, E0 + (c + 1) * R - kk);
            if (lo < hi) {
                const bool to_host = host_rows_in(lo, hi) > 0;
                if (to_host && out_free_set[h]) TVDN_HIP(hipStreamWaitEvent(st.main, out_free[h], 0));
                cdst.clear();
                csrc.clear();
                auto gather = [&](int i_store, int i_box, const Ring &rg) {
                    int64_t slot = 0;
                    for (int64_t v = lo; v < hi; ++v) {
                        const int64_t g = v - KX;
                        const bool res_row = resident(g);
                        cdst.push_back(res_row ? store_row(i_store, g) : outbox[h][i_box] + (size_t)slot * row_bytes);
                        csrc.push_back(rg.row(v));
                        if (!res_row) ++slot;
                    }
                };
                int i = 0;
                gather(1, i++, Rw[kk]);
                for (int q = 0; q < nd; ++q) {
                    gather(2 + q * n_state, i++, A(kk, q));
                    if (n_out_state == 2) gather(2 + q * n_state + 1, i++, A(kk - 1, q));
                }
                rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
                if (rc2) return rc2;
                if (to_host) {
                    TVDN_HIP(hipEventRecord(out_ready[h], st.main));
                    TVDN_HIP(hipStreamWaitEvent(st.down, out_ready[h], 0));
                    if ((rc2 = wait_recon(h_new))) return rc2;  // the host arrays these rows land in exist (first pass: the helper may still be at it)
                    const HostArr &rh = (two_sets && h_new) ? recon2_h : recon_h;
                    // runs of host rows: consecutive cube rows (an array page-locked in place) or consecutive host slots inside
                    // one block of host state -- a run must be one piece in every destination
                    int64_t slot = 0;
                    for (int64_t g = lo - KX; g < hi - KX;) {
                        if (resident(g)) {
                            ++g;
                            continue;
                        }
                        const int64_t hs = rm.host_below(g);
                        if ((rc2 = sb[h_new].wait_for(hs))) return rc2;
                        int64_t n = 1;
                        while (g + n < hi - KX && !resident(g + n) && sb[h_new].block_of(hs + n) == sb[h_new].block_of(hs)) ++n;
                        const size_t boff = (size_t)slot * row_bytes, len = (size_t)n * row_bytes;
                        i = 0;
                        TVDN_HIP(hipMemcpyAsync(hrow(rh, g, g + KX), outbox[h][i++] + boff, len, hipMemcpyDeviceToHost, st.down));
                        for (int q = 0; q < nd; ++q)
                            for (int s = 0; s < n_out_state; ++s)
                                TVDN_HIP(hipMemcpyAsync(srow(h_new, q * n_state + s, g, g + KX), outbox[h][i++] + boff, len, hipMemcpyDeviceToHost, st.down));
                        bytes_down += (int64_t)len * (1 + (int64_t)n_out_state * nd);
                        slot += n;
                        g += n;
                    }
                    TVDN_HIP(hipEventRecord(out_free[h], st.down));
                    out_free_set[h] = true;
                }
            }
        }
        if (relay_recv && !relayed) {  // a slab that had no use for the planes still takes part in the hand-over
            const int rcr = sh->relay_row0(0, row0_host.p, kk);
            if (rcr) {
                set_error("the row-0 relay of a slab run failed (status %d)", rcr);
                return TVDN_ERR_INVALID;
            }
            relayed = true;
        }
        TVDN_HIP(hipStreamSynchronize(st.down));
        TVDN_HIP(hipStreamSynchronize(st.main));
        TVDN_HIP(hipStreamSynchronize(st.up));
        d_form = forms[kk];
        tk_prev = prev;
        done += kk;
        h_old = h_new;
        ++n_passes;
        return TVDN_OK;
    };

    // ---- chained passes (Jia-Zhao) ---------------------------------------------------------------------------------------------
    // A pass of K levels over N0 rows fills and drains its pipeline: uploads happen in its first N0 / R chunks, downloads in
    // its last N0 / R, and only the chunks in between use the link both ways -- at K = 38 on 64 rows that is 12 chunks of 52,
    // although the link carries 56 GB/s up AND 49 GB/s down at once (tools/ubench/pcie_duplex.hip, profiles/r04_pcie_duplex.jsonl).
    // Chained, pass p + 1 starts uploading in the chunk after pass p's last upload, while p's upper levels are still climbing:
    // the passes are stacked into one running row index v = p N0 + g, level j + 1 trails level j by one row of THAT index, a
    // ring slot is v mod ring, and a launch that straddles the seam between two passes is cut there (the last row of p ends
    // at the cube's top face, row 0 of p + 1 starts at its bottom one; each piece with its own iteration numbers).  What pass
    // p + 1 uploads must be home: row g comes down (p N0 + g + K) / R chunks in and goes up again ((p + 1) N0 + g) / R - 1
    // chunks in, so K <= N0 - 3 R is asked for, and the upload stream waits for the download event of that chunk.
    // One pass at a time (`chain` of one) is the drained schedule: runs with a stopping rule, and passes deeper than that.
    struct PassDesc {
        int it0 = 0, kk = 0;         // first iteration slot, levels
        std::vector<int> modes;      // TVDN_ITER_* per level
        std::vector<double> tk, tkp; // momentum ratio of the level / of the level before it
        int n_in_state = 1, n_out_state = 1;
        bool first = false;          // starts from recon = data term and zero accumulators: uploads the data term only
        bool last = false;           // the run ends with this pass (no stopping rule): resident rows send their result straight home
    };
    auto describe = [&](int it0, int kk, const double *rat, PassDesc &pd) -> int {
        pd.it0 = it0;
        pd.kk = kk;
        pd.modes.resize((size_t)kk);
        pd.tk.resize((size_t)kk);
        pd.tkp.resize((size_t)kk);
        pd.first = n_passes == 0 && it0 == 0;
        pd.last = !a->use_stop && it0 + kk == n_total;
        bool form = d_form;
        pd.n_in_state = form ? 2 : 1;
        double prev = tk_prev;
        for (int j = 0; j < kk; ++j) {
            const bool acc = !std::isnan(rat[j]);
            TVDN_REQUIRE(!acc || form, "a FISTA iteration cannot follow an unaccelerated one");
            pd.modes[(size_t)j] = iter_mode(acc, form);
            pd.tk[(size_t)j] = acc ? rat[j] : 0.0;
            pd.tkp[(size_t)j] = prev;
            form = acc;
            if (acc) prev = rat[j];
        }
        pd.n_out_state = form ? 2 : 1;
        d_form = form;  // the trackers move on: the next description continues from here
        tk_prev = prev;
        return TVDN_OK;
    };
    std::vector<char *> row0b;  // second set of top-face planes: two passes are in flight at a seam
    // How rows come down.  The runtime's hipMemcpyAsync moves a download with a DMA engine when its stream is idle and with a
    // blit kernel otherwise; two DMA transfers in opposite directions at once take 4 x as long (512 MiB down: 10.5 ms alone,
    // 42 ms beside an upload; profiles/r04_chained_trace_summary.txt -- what made chained passes LOSE in round 3), while an
    // upload by DMA beside a download by a small copy kernel runs at 54 + 42 GB/s and leaves the sweeps alone (8 workgroups;
    // with 16 or more the sweeps lose a third: tools/ubench/pcie_duplex.hip, profiles/r04_pcie_duplex.jsonl).  Chained passes
    // keep both directions busy all the time, so their downloads are a copy kernel of 8 workgroups writing the page-locked
    // host arrays directly; drained passes keep the runtime's copies.  TVDN_STREAM_DOWN_BLOCKS=n overrides (0: runtime copies).
    int down_blocks = 0;
    bool recon_direct = false, recon_direct_decided = false;  // the last pass sends the resident rows' results home itself
    auto chain = [&](std::vector<PassDesc> &ps) -> int {
        const int P = (int)ps.size();
        const int64_t V1 = (int64_t)P * N0;  // running rows that are uploaded
        int64_t kmax = 0;
        for (const PassDesc &pd : ps) kmax = std::max<int64_t>(kmax, pd.kk);
        const int64_t n_chunks = (V1 + ps[(size_t)P - 1].kk + R - 1) / R;
        std::vector<hipEvent_t> down_done((size_t)n_chunks, nullptr);
        const int bx_recon = 1, bx_ref = 2 + nd * n_state;  // fixed box numbers: 0 data term, 1 recon, 2 + q n_state + s state
        auto bx_state = [&](int q, int s) { return 2 + q * n_state + s; };
        auto ox_state = [&](int q, int s) { return 1 + q * n_state + s; };  // out boxes: 0 recon, then the state
        it.shape[0] = V1;
        // the pieces of the running rows [v0, v1) by pass: fn(pass, v_lo, v_hi)
        auto pieces = [&](int64_t v0, int64_t v1, const std::function<int(int, int64_t, int64_t)> &fn) -> int {
            v0 = std::max<int64_t>(v0, 0);
            v1 = std::min<int64_t>(v1, V1);
            for (int64_t v = v0; v < v1;) {
                const int q = (int)(v / N0);
                const int64_t e = std::min<int64_t>(v1, (int64_t)(q + 1) * N0);
                const int rcp = fn(q, v, e);
                if (rcp) return rcp;
                v = e;
            }
            return TVDN_OK;
        };
        auto host_rows_in = [&](int64_t v0, int64_t v1) {
            int64_t n = 0;
            for (int64_t v = std::max<int64_t>(v0, 0); v < std::min(v1, V1); ++v) n += resident(v % N0) ? 0 : 1;
            return n;
        };
        // host rows among cube rows [g0, g1) -> box rows from `slot` on, run by run
        auto up_rows = [&](char *box, int64_t &slot, int64_t g0, int64_t g1, const std::function<char *(int64_t)> &src_row,
                           const std::function<bool(int64_t)> &joins_next) -> int {
            for (int64_t g = g0; g < g1;) {
                if (resident(g)) {
                    ++g;
                    continue;
                }
                int64_t n = 1;
                while (g + n < g1 && !resident(g + n) && joins_next(g + n - 1)) ++n;
                TVDN_HIP(hipMemcpyAsync(box + (size_t)slot * row_bytes, src_row(g), (size_t)n * row_bytes, hipMemcpyHostToDevice, st.up));
                bytes_up += n * (int64_t)row_bytes;
                slot += n;
                g += n;
            }
            return TVDN_OK;
        };
        const std::function<bool(int64_t)> always = [](int64_t) { return true; };
        auto upload = [&](int64_t t, int64_t t_now) -> int {
            const int64_t u0 = t * R, u1 = std::min((t + 1) * R, V1);
            if (u0 >= u1 || host_rows_in(u0, u1) == 0) return TVDN_OK;
            int rcu = orig_ready.wait();
            if (rcu) return rcu;
            const int h = (int)(t % 2);
            if (in_free_set[h]) TVDN_HIP(hipStreamWaitEvent(st.up, in_free[h], 0));
            int64_t s_orig = 0, s_recon = 0, s_ref = 0;
            std::vector<int64_t> s_state((size_t)nd * 2, 0);
            rcu = pieces(u0, u1, [&](int q, int64_t v_lo, int64_t v_hi) -> int {
                const PassDesc &pd = ps[(size_t)q];
                const int64_t g0 = v_lo - (int64_t)q * N0, g1 = v_hi - (int64_t)q * N0;
                const int64_t before = s_orig;
                int r3 = up_rows(inbox[h][0], s_orig, g0, g1, [&](int64_t g) { return host_row(orig_h, g); }, always);
                if (r3) return r3;
                const int64_t n_host = s_orig - before;
                if (want_mse && (r3 = up_rows(inbox[h][bx_ref], s_ref, g0, g1, [&](int64_t g) { return host_row(ref_h, g); }, always))) return r3;
                if (pd.first) {  // recon and state are formed on the device: their box rows stay unused
                    s_recon += n_host;
                    for (int64_t &x : s_state) x += n_host;
                    return TVDN_OK;
                }
                if (q > 0) {
                    // these rows came down at the end of pass q - 1: the upload stream waits for the chunk that sent the last of them
                    const int64_t t_out = ((int64_t)(q - 1) * N0 + (g1 - 1) + ps[(size_t)q - 1].kk) / R;
                    TVDN_REQUIRE(t_out < t_now && down_done[(size_t)t_out] != nullptr,
                                 "chained passes: row %lld of pass %d is uploaded before pass %d has sent it home (k too deep to chain)",
                                 (long long)(g1 - 1), q, q - 1);
                    TVDN_HIP(hipStreamWaitEvent(st.up, down_done[(size_t)t_out], 0));
                }
                if ((r3 = wait_recon(0))) return r3;
                if ((r3 = up_rows(inbox[h][bx_recon], s_recon, g0, g1, [&](int64_t g) { return host_row(recon_h, g); }, always))) return r3;
                for (int64_t g = g0; g < g1; ++g)
                    if (!resident(g) && (r3 = sb[0].wait_for(rm.host_below(g)))) return r3;
                for (int qx = 0; qx < nd; ++qx)
                    for (int s = 0; s < n_state; ++s) {
                        int64_t &sl = s_state[(size_t)qx * 2 + s];
                        if (s >= pd.n_in_state) {
                            sl += n_host;
                            continue;
                        }
                        const int arr = qx * n_state + s;
                        if ((r3 = up_rows(inbox[h][bx_state(qx, s)], sl, g0, g1, [&](int64_t g) { return sb[0].row(arr, rm.host_below(g)); },
                                          [&](int64_t g) { return sb[0].block_of(rm.host_below(g)) == sb[0].block_of(rm.host_below(g + 1)); })))
                            return r3;
                    }
                return TVDN_OK;
            });
            if (rcu) return rcu;
            TVDN_HIP(hipEventRecord(in_ready[h], st.up));
            return TVDN_OK;
        };

        int rc2 = upload(0, 0);
        if (rc2) return rc2;
        for (int64_t t = 0; t < n_chunks; ++t) {
            if ((rc2 = upload(t + 1, t))) return rc2;  // the next chunk crosses PCIe while this one is swept
            const int h = (int)(t % 2);
            const int64_t u0 = t * R, u1 = std::min((t + 1) * R, V1);
            if (u0 < u1) {
                const bool from_host = host_rows_in(u0, u1) > 0;
                if (from_host) TVDN_HIP(hipStreamWaitEvent(st.main, in_ready[h], 0));
                cdst.clear();
                csrc.clear();
                int64_t slot = 0;
                for (int64_t v = u0; v < u1; ++v) {
                    const int q = (int)(v / N0);
                    const PassDesc &pd = ps[(size_t)q];
                    const int64_t g = v - (int64_t)q * N0;
                    const bool res_row = resident(g);
                    if (pd.first && res_row && (rc2 = wait_staged(g + 1))) return rc2;
                    auto put = [&](const Ring &rg, const char *src) {
                        cdst.push_back(rg.row(v));
                        csrc.push_back((void *)src);
                    };
                    auto boxed = [&](int bx) { return inbox[h][bx] + (size_t)slot * row_bytes; };
                    const char *o_src = res_row ? store_row(0, g) : boxed(0);
                    put(Ow, o_src);
                    put(Rw[0], pd.first ? o_src : (res_row ? store_row(1, g) : boxed(bx_recon)));
                    for (int qx = 0; qx < nd; ++qx) {
                        put(A(0, qx), pd.first ? zero_plane : (res_row ? store_row(2 + qx * n_state, g) : boxed(bx_state(qx, 0))));
                        if (pd.n_in_state == 2)
                            put(A(-1, qx), pd.first ? zero_plane : (res_row ? store_row(2 + qx * n_state + 1, g) : boxed(bx_state(qx, 1))));
                    }
                    if (want_mse) put(Fw, boxed(bx_ref));
                    if (!res_row) ++slot;
                }
                rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
                if (rc2) return rc2;
                for (int64_t v = u0; v < u1; ++v) {
                    const int q = (int)(v / N0);
                    const int64_t g = v - (int64_t)q * N0;
                    if (exact_wrap && g == 0)
                        TVDN_HIP(hipMemcpyAsync(((q & 1) ? row0b : row0)[0], Rw[0].row(v), row_bytes, hipMemcpyDeviceToDevice, st.main));
                    if (want_mse && ps[(size_t)q].it0 == 0 && ps[(size_t)q].first)  // MSE[0]: the input against the reference (cyTVDN.py:124-125)
                        if ((rc2 = sse_row(Rw[0].row(v), Fw.row(v), 0, g))) return rc2;
                }
                if (from_host) {
                    TVDN_HIP(hipEventRecord(in_free[h], st.main));
                    in_free_set[h] = true;
                }
            }
            // the wavefront: level j+1 trails level j by one running row; a launch is cut at the seam between two passes
            for (int64_t j = 0; j < kmax; ++j) {
                rc2 = pieces(t * R - (j + 1), (t + 1) * R - (j + 1), [&](int q, int64_t v_lo, int64_t v_hi) -> int {
                    const PassDesc &pd = ps[(size_t)q];
                    if (j >= pd.kk) return TVDN_OK;
                    const int mode = pd.modes[(size_t)j];
                    it.row_lo = (int64_t)q * N0;
                    it.row_hi = (int64_t)(q + 1) * N0;
                    it.sweep_lo = v_lo;
                    it.sweep_hi = v_hi;
                    it.mode = mode;
                    it.tk = pd.tk[(size_t)j];
                    it.tk_prev = pd.tkp[(size_t)j];
                    it.recon_in = Rw[(size_t)j].base;
                    it.recon_out = Rw[(size_t)j + 1].base;
                    it.wrap_recon = exact_wrap ? ((q & 1) ? row0b : row0)[(size_t)j] : nullptr;
                    for (int qx = 0; qx < nd; ++qx) {
                        char *cur = A(j, qx).base, *prv = A(j - 1, qx).base, *nxt = A(j + 1, qx).base;
                        it.b_in[qx] = it.d_in[qx] = it.dprev_in[qx] = nullptr;
                        it.b_out[qx] = it.d_out[qx] = nullptr;
                        if (mode == TVDN_ITER_FISTA_D) {
                            it.d_in[qx] = cur; it.dprev_in[qx] = prv; it.d_out[qx] = nxt;
                        } else if (mode == TVDN_ITER_FISTA_D_TO_PLAIN) {
                            it.d_in[qx] = cur; it.dprev_in[qx] = prv; it.b_out[qx] = nxt;
                        } else {
                            it.b_in[qx] = cur; it.b_out[qx] = nxt;
                        }
                    }
                    int r3 = tvdn_iterate_fused(ctx.c, &it, (double *)sums_d.p + 3 * (size_t)(pd.it0 + (int)j), st.main);
                    if (r3) return r3;
                    if (want_mse)
                        for (int64_t v = v_lo; v < v_hi; ++v)
                            if ((r3 = sse_row(Fw.row(v), Rw[(size_t)j + 1].row(v), pd.it0 + (int)j + 1, v - (int64_t)q * N0))) return r3;
                    if (exact_wrap && v_lo == (int64_t)q * N0)
                        TVDN_HIP(hipMemcpyAsync(((q & 1) ? row0b : row0)[(size_t)j + 1], Rw[(size_t)j + 1].row(v_lo), row_bytes, hipMemcpyDeviceToDevice, st.main));
                    return TVDN_OK;
                });
                if (rc2) return rc2;
            }
            // rows that have reached their pass's last level go home: resident rows into the store, the others across PCIe
            cdst.clear();
            csrc.clear();
            struct Out {
                int q;
                int64_t g0, g1, slot0;
            };
            std::vector<Out> outs;
            int64_t oslot = 0;
            for (int q = 0; q < P; ++q) {
                const PassDesc &pd = ps[(size_t)q];
                const int64_t lo = std::max<int64_t>((int64_t)q * N0, t * R - pd.kk), hi = std::min<int64_t>((int64_t)(q + 1) * N0, (t + 1) * R - pd.kk);
                if (lo >= hi) continue;
                // The run's last pass: the state of a resident row is not needed again, and its result can cross PCIe under
                // the pass (the link has room: a hybrid run uses half of it) instead of in one piece after it -- when the
                // caller's result array is page-locked in place, i.e. has a place for every row.
                if (pd.last && RES > 0 && !recon_direct_decided) {
                    if ((rc2 = wait_recon(0))) return rc2;
                    recon_direct = recon_h.cube_rows && getenv("TVDN_STREAM_HOME_AFTER") == nullptr;
                    recon_direct_decided = true;
                }
                const bool direct = pd.last && recon_direct;
                outs.push_back(Out{q, lo - (int64_t)q * N0, hi - (int64_t)q * N0, oslot});
                for (int64_t v = lo; v < hi; ++v) {
                    const int64_t g = v - (int64_t)q * N0;
                    const bool res_row = resident(g);
                    auto put = [&](int i_store, int ox, const Ring &rg) {
                        cdst.push_back(res_row ? store_row(i_store, g) : outbox[h][ox] + (size_t)oslot * row_bytes);
                        csrc.push_back(rg.row(v));
                    };
                    if (res_row && direct) {  // the result only, into the out box like a host row's
                        cdst.push_back(outbox[h][0] + (size_t)oslot * row_bytes);
                        csrc.push_back(Rw[(size_t)pd.kk].row(v));
                        ++oslot;
                        continue;
                    }
                    put(1, 0, Rw[(size_t)pd.kk]);
                    for (int qx = 0; qx < nd; ++qx) {
                        put(2 + qx * n_state, ox_state(qx, 0), A(pd.kk, qx));
                        if (pd.n_out_state == 2) put(2 + qx * n_state + 1, ox_state(qx, 1), A(pd.kk - 1, qx));
                    }
                    if (!res_row) ++oslot;
                }
            }
            TVDN_REQUIRE(oslot <= R + 1, "chained passes: %lld rows come down in one chunk, the out boxes hold %lld (depths of consecutive passes differ by more than one)",
                         (long long)oslot, (long long)(R + 1));
            if (!cdst.empty()) {
                if (oslot > 0 && out_free_set[h]) TVDN_HIP(hipStreamWaitEvent(st.main, out_free[h], 0));
                rc2 = copy_rows(cdst, csrc, row_bytes, st.main);
                if (rc2) return rc2;
            }
            if (oslot > 0) {
                TVDN_HIP(hipEventRecord(out_ready[h], st.main));
                TVDN_HIP(hipStreamWaitEvent(st.down, out_ready[h], 0));
                if ((rc2 = wait_recon(0))) return rc2;  // the host arrays these rows land in exist (first pass: the helper may still be at it)
                std::vector<void *> kd, ks;  // copies of whole rows for the copy kernel (down_blocks > 0)
                for (const Out &o : outs) {
                    const PassDesc &pd = ps[(size_t)o.q];
                    const bool direct = pd.last && recon_direct;
                    int64_t slot = o.slot0;
                    for (int64_t g = o.g0; g < o.g1;) {
                        if (resident(g)) {
                            if (direct) {  // its result went into the out box: one row, straight into the caller's array
                                char *dst = recon_h.p + (size_t)g * row_bytes;
                                const char *src = outbox[h][0] + (size_t)slot * row_bytes;
                                if (down_blocks > 0 && row_bytes % 16 == 0 && ((uintptr_t)dst & 15) == 0) {
                                    kd.push_back(dst);
                                    ks.push_back((void *)src);
                                } else {
                                    TVDN_HIP(hipMemcpyAsync(dst, src, row_bytes, hipMemcpyDeviceToHost, st.down));
                                }
                                bytes_down += (int64_t)row_bytes;
                                ++slot;
                            }
                            ++g;
                            continue;
                        }
                        const int64_t hs = rm.host_below(g);
                        if ((rc2 = sb[0].wait_for(hs))) return rc2;
                        int64_t n = 1;
                        while (g + n < o.g1 && !resident(g + n) && sb[0].block_of(hs + n) == sb[0].block_of(hs)) ++n;
                        const size_t boff = (size_t)slot * row_bytes, len = (size_t)n * row_bytes;
                        auto down = [&](char *dst, const char *src) -> int {
                            if (down_blocks > 0 && row_bytes % 16 == 0 && ((uintptr_t)dst & 15) == 0) {
                                for (int64_t r = 0; r < n; ++r) {
                                    kd.push_back(dst + (size_t)r * row_bytes);
                                    ks.push_back((void *)(src + (size_t)r * row_bytes));
                                }
                            } else {
                                TVDN_HIP(hipMemcpyAsync(dst, src, len, hipMemcpyDeviceToHost, st.down));
                            }
                            return TVDN_OK;
                        };
                        if ((rc2 = down(host_row(recon_h, g), outbox[h][0] + boff))) return rc2;
                        for (int qx = 0; qx < nd; ++qx)
                            for (int s = 0; s < pd.n_out_state; ++s)
                                if ((rc2 = down(sb[0].row(qx * n_state + s, hs), outbox[h][ox_state(qx, s)] + boff))) return rc2;
                        bytes_down += (int64_t)len * (1 + (int64_t)pd.n_out_state * nd);
                        slot += n;
                        g += n;
                    }
                }
                // ONE launch of a few workgroups writes the chunk's rows into the page-locked host arrays (see down_blocks)
                if (!kd.empty() && (rc2 = tvdn_copy_many((int32_t)kd.size(), kd.data(), ks.data(), (int64_t)row_bytes, down_blocks, st.down))) return rc2;
                TVDN_HIP(hipEventRecord(out_free[h], st.down));
                out_free_set[h] = true;
            }
            if (P > 1) {  // what a later pass's uploads wait for (also for chunks that sent nothing: the stream is in order)
                if ((rc2 = evs.make(&down_done[(size_t)t]))) return rc2;
                TVDN_HIP(hipEventRecord(down_done[(size_t)t], st.down));
            }
        }
        TVDN_HIP(hipStreamSynchronize(st.down));
        TVDN_HIP(hipStreamSynchronize(st.main));
        TVDN_HIP(hipStreamSynchronize(st.up));
        n_passes += P;
        return TVDN_OK;
    };

    // ---- the schedule: FISTA ratios in float64 on the host (cyTVDN.py:153-156), then the unaccelerated tail -------------
    std::vector<double> ratios((size_t)n_total);
    fista_ratios(a->n_fista, ratios.data());
    for (int i = a->n_fista; i < n_total; ++i) ratios[i] = NAN;
    int ran = 0, ran_phase[2] = {a->n_fista, a->n_plain};
    auto meet = [&]() -> int {  // slabs of a device-list run: every pass ends at the barrier (a failed slab releases the others)
        if (!sh || !sh->barrier) return TVDN_OK;
        const int rcb = sh->barrier->arrive_and_wait();
        if (rcb) set_error("another slab of this run failed: %s", sh->barrier->msg.c_str());
        return rcb;
    };
    auto stop_after = [&](int slot, bool &stop) -> int {
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
    };
    TVDN_HIP(hipStreamSynchronize(st.main));
    const auto t_passes = std::chrono::steady_clock::now();
    double first_pass_s = 0.0;
    // Jia-Zhao runs go through `chain`: all passes at once when they can be chained (K <= N0 - 3 R, no stopping rule;
    // TVDN_STREAM_CHAIN=0 keeps them apart), else one at a time.  Periodic runs keep the drained `pass` above.
    // Chaining pays where the link is the bound and the run is long: with every row streamed, 4 passes of 38 levels over 64
    // planes of 256 MiB run at 40.2 Gvoxel-iters/s chained against 36.2 drained (3.9 against 4.9 s per further pass); 2 passes
    // tie (38.4 / 39.4: the first pass uploads little, the last one only drains); and a run that keeps most rows resident is
    // bound by its sweeps, which the download kernel disturbs (51.6 against 57.0).  profiles/r04_stream_rates.jsonl.
    // Default: chain from 3 passes on when no row is resident; TVDN_STREAM_CHAIN=1 / 0 forces it on (where possible) / off.
    bool want_chain = RES == 0 && n_pass_plan >= 3 && !sh;
    if (const char *e = getenv("TVDN_STREAM_CHAIN")) want_chain = atoi(e) != 0;
    const bool can_chain = want_chain && !periodic && !a->use_stop && n_pass_plan > 1 && K <= N0 - 3 * R;
    down_blocks = can_chain ? 8 : 0;
    if (const char *e = getenv("TVDN_STREAM_DOWN_BLOCKS")) down_blocks = std::max(0, atoi(e));
    if (!periodic && exact_wrap)
        for (int64_t j = 0; j <= K; ++j) row0b.push_back(row0b_base + (size_t)j * plane_b);
    const bool drained_pass = periodic || sh != nullptr;  // the `pass` lambda: two sets of host state, artificial faces
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
    const auto t_end_passes = std::chrono::steady_clock::now();

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
        char *home = sh ? sh->own_recon - (size_t)sh->g0 * row_bytes : (char *)a->recon_out;  // (a slab: the caller's own-row array)
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
    TVDN_HIP(hipMemcpy(a->sums_out, sums_d.p, sizeof(double) * 3 * (size_t)n_total, hipMemcpyDeviceToHost));
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
    }
    return TVDN_OK;
}

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
        }
        o.setup_s = std::chrono::duration<double>(t_threads - t_start).count();
        o.loop_s = std::chrono::duration<double>(t_done - t_threads).count();
        o.total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    }
    return TVDN_OK;
}

// What one slab of a multi-process streamed run holds where: the depth its passes settle on (= the halo rows it keeps of each
// neighbour), the interior rows that stay resident in HBM, the rows of its packed local host arrays.  One definition: the run
// allocates by it, tvdn_slab_host_need tells the caller beforehand (so that the ranks of one host can add up what they will
// page-lock BEFORE any of them does).  Looks at the device's free memory unless nothing can be kept anyway.
static int slab_shape(const tvdn_run_args *a, int64_t R, int64_t K, int64_t *kc_out, int64_t *res_out, int64_t *local_rows_out)
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
