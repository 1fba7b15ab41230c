// Parts of the streamed engine shared by its translation units (tvdn_stream*.hip): page-locked host memory, rings, the
// bookkeeping of who keeps which rows, and what a slab of a device list / of a multi-process run shares with its coordinator.
// Moved here from tvdn_stream.hip in round 5 (no file of the engine above 1000 lines); nothing changed but the place.
#pragma once

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <deque>
#include <thread>

#include <sys/mman.h>
#include <unistd.h>

#include "tvdn_common.hpp"

namespace tvdn {

constexpr int kHostThreads = 8;
constexpr size_t kPinInPlaceMinDefault = size_t(256) << 20;  // bytes from which a caller's array is page-locked in place

inline void parallel_copy(void *dst, const void *src, size_t bytes)  // src == nullptr: zero fill
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
inline void touch_pages(char *p, size_t bytes, int threads)
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

inline int touch_threads()
{
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::max(1u, std::min(16u, hc ? hc : 4u));
}

// Releases of pinned memory run on detached threads (unregistering and unmapping 16 GiB takes 0.8 s; a run that held 144 GiB
// would spend 7 s returning it): the next streamed run waits for them before it counts the host's memory.
inline std::atomic<int> g_releases_pending{0};  // (one counter for the whole library: C++17 inline variable)

inline void wait_for_releases()
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

// Host memory a streamed run may count on: what the kernel calls available, never more than the machine has, and
// never more than the memory limit of the process's control group (the limit itself, not limit minus usage: the
// usage counts page cache the kernel would give back, and a false refusal helps nobody; the check is there to stop
// requests that are wrong by factors).  0 = could not be determined.
inline size_t host_available_bytes()
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

inline bool arrays_overlap(const void *x, const void *y, size_t bytes)
{
    const uintptr_t a0 = (uintptr_t)x, b0 = (uintptr_t)y;
    return a0 < b0 + bytes && b0 < a0 + bytes;
}

// n row-plane copies inside HBM: one streaming launch when the rows are 16-byte multiples, the runtime's copies otherwise
inline int copy_rows(std::vector<void *> &dst, std::vector<void *> &src, size_t row_bytes, hipStream_t s)
{
    if (row_bytes % 16 == 0)
        return tvdn_copy_many((int32_t)dst.size(), dst.data(), src.data(), (int64_t)row_bytes, 0, s);
    for (size_t i = 0; i < dst.size(); ++i) TVDN_HIP(hipMemcpyAsync(dst[i], src[i], row_bytes, hipMemcpyDeviceToDevice, s));
    return TVDN_OK;
}

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
    // A slab of a device list (shared arrays indexed by cube row): interior rows may stay in HBM as far as they fit beside the rings
    // in this slab's share of its device (`same_device` slabs of the list sit on it); their host rows simply go unused.
    bool keep_rows = false;
    int same_device = 1;
    const char *own_data = nullptr;
    char *own_recon = nullptr;
    bool exact_wrap = false;                             // Jia-Zhao, first row of the cube not finite (the same on every slab)
    std::function<int()> before_pass;                    // before every pass but the first: refresh the halo rows of recon / state
    std::function<int(double *)> allreduce;              // one iteration's three sums -> over all slabs (stopping rule)
    std::function<int(int, void *, int)> relay_row0;     // (send, planes, n): row 0 of every level, first slab -> last slab
};

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

// Downloads by the runtime's copies, ONE AT A TIME.  Eight hipMemcpyAsync calls queued on a stream behind an event -- a chunk's
// rows on their way home -- are served slowly inside this engine: 21 ms per 256 MiB and thousands of __amd_rocclr_copyBuffer blit
// kernels under rocprofv3 (they wait for wavefront slots the sweeps hold), 36 Gvoxel-iters/s where round 4's copy kernel of 8
// workgroups gave 47 (and cost the sweeps 5 %).  The same copies handed to an idle stream one by one -- a helper thread waits for
// a chunk's rows to be gathered, issues a copy, waits for it, issues the next -- run at 55 GB/s beside 42 GB/s of uploads and
// leave the sweeps alone: 64.8 (profiles/r05_down_pump.jsonl; tools/ubench/pcie_down_kernels.hip shows the same link rates in
// isolation: 47.7 down + 55.7 up + 5.9 TB/s of kernels).  The scheduling thread asks the helper, on the host, before it reuses an
// out box or uploads what has come down.
struct DownPump {
    struct Copy {
        void *dst;
        const void *src;
        size_t len;
    };
    struct Job {
        int64_t id = 0;
        hipEvent_t ready = nullptr;  // recorded on the sweeps' stream behind the gather of this chunk
        std::vector<Copy> copies;
    };
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Job> q;
    int64_t finished = -1;  // every job pushed with id <= finished is done (ids ascend)
    int64_t pushed = -1;
    bool stop = false, running = false;
    int rc = TVDN_OK;
    std::string msg;
    void start(int device, hipStream_t s)
    {
        running = true;
        th = std::thread([this, device, s] {
            (void)hipSetDevice(device);
            for (;;) {
                Job j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || !q.empty(); });
                    if (q.empty()) return;
                    j = std::move(q.front());
                    q.pop_front();
                }
                hipError_t e = hipEventSynchronize(j.ready);
                for (size_t i = 0; i < j.copies.size() && e == hipSuccess; ++i) {
                    e = hipMemcpyAsync(j.copies[i].dst, j.copies[i].src, j.copies[i].len, hipMemcpyDeviceToHost, s);
                    if (e == hipSuccess) e = hipStreamSynchronize(s);
                }
                std::lock_guard<std::mutex> lk(mu);
                if (e != hipSuccess && rc == TVDN_OK) {
                    rc = TVDN_ERR_HIP;
                    msg = std::string("a download of a streamed run failed: ") + hipGetErrorString(e);
                    (void)hipGetLastError();
                }
                finished = j.id;
                cv.notify_all();
            }
        });
    }
    void push(Job &&j)
    {
        std::lock_guard<std::mutex> lk(mu);
        pushed = j.id;
        q.push_back(std::move(j));
        cv.notify_all();
    }
    // every job with id <= `id` is done (the helper has waited for its copies)
    int wait(int64_t id)
    {
        std::unique_lock<std::mutex> lk(mu);
        const int64_t upto = std::min(id, pushed);
        cv.wait(lk, [&] { return finished >= upto || !running; });
        if (rc) set_error("%s", msg.c_str());
        return rc;
    }
    int drain() { return wait(INT64_MAX); }
    void shutdown()
    {
        if (!running) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        if (th.joinable()) th.join();
        running = false;
    }
    ~DownPump() { shutdown(); }
};

// ---- tvdn_stream_plan.hip: what a shape costs, and what a run may choose -------------------------------------------------------
int64_t stream_planes(int nd, int64_t rows, int64_t k, bool mse, bool wrap);
int64_t stream_planes_all_kept(int nd, int64_t rows, int64_t k, bool wrap);
size_t stream_device_bytes_all_kept(int nd, int64_t R, int64_t K, size_t row_bytes);
int stream_host_need(const tvdn_run_args *a, int64_t res, int64_t *need_bytes, int64_t *avail_bytes);
size_t stream_device_bytes(int nd, int n_state, bool want_mse, int64_t R, int64_t K, size_t row_bytes, size_t *per_resident_row);
int slab_shape(const tvdn_run_args *a, int64_t R, int64_t K, int64_t *kc_out, int64_t *res_out, int64_t *local_rows_out);

}  // namespace tvdn
