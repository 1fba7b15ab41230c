// Device memory of a state, composed from physical granules (HIP virtual-memory management).
//
// Why.  The speed of the fused sweep on a state of tens of GiB depends on WHERE in the HBM its fifteen arrays lie relative to
// each other: hipMalloc states of BASELINE config 2 held side by side in one process sweep in 11.0 ... 12.6 ms, each
// reproducibly for minutes (profiles/r03_placement_audition_*.jsonl, r05_placement_timeline_*.jsonl) -- a 12 % lottery that
// rounds 3-4 answered by allocating whole candidates and keeping the fastest.  Round 5 took it apart with a state mapped from
// 1 GiB physical granules (hipMemCreate / hipMemMap; tools/ubench/vmm_*.hip, profiles/r05_vmm_*.jsonl):
//   * the slow states move the same bytes over the same L2 channels with the same read and write latency and the same TLB
//     misses as the fast ones (profiles/r05_pmc_slow_state.txt): nothing in the kernel to tune;
//   * a state on 61 granules created one after the other -- physically one run of the HBM, like a hipMalloc block -- draws from
//     the same lottery (windows over the whole HBM: 11.13 ... 11.84 ms, repeatable to 0.01 ms); the SAME granules in another
//     order give another time (11.26 ... 11.72), so it is the arrangement, not a property of single granules;
//   * 61 granules drawn AT RANDOM from a pool spread over two to four times as much HBM sweep in 11.09 ... 11.25 ms, every one
//     of 150 draws on boxes whose contiguous windows reach 11.8 (from a pool of the state's own size: up to 11.72).
// So placement can be had by construction: create more granules than the state needs while that is cheap, keep a random
// subset in random order, give the rest back.  No probe sweeps, no second candidate.
//
// What.  dev_alloc / dev_free / dev_resize: blocks of TVDN_VMM_MIN_MIB (2 GiB; 0 = every block) or more come from granules,
// smaller ones -- and every block when the runtime refuses virtual-memory management, fails the canary below, or TVDN_VMM=0 --
// from hipMalloc.
//   * Pool.  Up to TVDN_SPREAD (3) times the block's granules are created, as far as the memory this process may take
//     (pool_room: 90 % of the free HBM, never into the last max(4 GiB, 5 %) of the device, never beyond TVDN_HBM_LIMIT) and
//     `spread_budget_s` seconds allow.  A block that GROWS (dev_resize) draws the granules it lacks the same way.
//   * Equal granules.  Every granule of a block has the same size and the block is rounded up to a whole number of them: on ROCm
//     7.2 a reservation mapped from handles of different sizes confuses the runtime's own address lookups -- hipMemcpy /
//     hipMemset into the part behind the odd handle land elsewhere while kernels see the right bytes (tools/ubench/
//     vmm_memset.hip).  1 GiB from 8 GiB on, 64 MiB below (granule_for).
//   * Stale translations.  On ROCm 7.2 hipMemUnmap / hipMemMap do not make the GPU forget the old translation of an address:
//     kernels and copies go on using the physical memory that WAS mapped there (tools/ubench/vmm_remap_check.hip), also after
//     hipMemAddressFree + a new reservation that overlaps the old one.  A hipFree of any plain block flushes the TLBs
//     (vmm_remap_flush.hip), so every map and every unmap here ends with tlb_flush(): a 4 MiB hipMalloc + hipFree.
//
// Trust (round 6).  Both work-arounds lean on behaviour nobody documents, and a failure of either is silent wrong bits.  So:
//   * Canary.  Before a device hands out its first block of granules, vmm_canary() plays the product's own sequence on two small
//     granules: write distinct patterns by kernel, unmap, map the handles the other way round, tlb_flush(), read back by kernel
//     AND by hipMemcpy, write through the new mapping, swap back, read again.  One stale word and the device is marked
//     (state -1): plain hipMalloc blocks from then on, a line on stderr -- the product degrades to round 4's lottery instead of
//     computing garbage.  TVDN_VMM_SKIP_FLUSH=1 (test knob) turns tlb_flush() into a no-op, which on ROCm 7.2 must make it trip
//     (tests/test_gpu_vmm_guard.py).  TVDN_VMM_CANARY=0 skips it, =always repeats it before every block (stress runs).
//   * A flush that cannot be done fails the map.  tlb_flush() retries its hipMalloc for 200 ms (released granules return
//     lazily); if there is still no memory, map_block gives the granules back and reports the error, and the caller gets a plain
//     block or the error -- never a range whose translations may be stale.
//   * Every hipMem* return code is looked at.  The first failure of each call leaves its name, address and size on stderr and
//     in tvdn_last_error(); tvdn_mem_status() counts them (`faults`) and keeps the first one's text.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <mutex>
#include <random>
#include <thread>
#include <unordered_map>

#include "tvdn_common.hpp"

namespace tvdn {

namespace {

struct VmmBlock {
    char *va = nullptr;
    size_t va_bytes = 0;
    size_t G = 0;  // every granule of the block has this size
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<int> peers;  // other devices that may read and write the range (slabs of a device list: tvdn_run.hip)
    size_t pool = 0;         // how many granules its own were chosen from (dev_upgrade tops a short pool up when a run can afford it)
};

struct DevVmm {
    int state = 0;   // 0 not tried, 1 works, -1 plain blocks from now on (refused by the runtime, or the canary tripped)
    int canary = 0;  // TVDN_CANARY_*: 0 not run, 1 passed, -1 a stale translation was seen, -2 the VMM calls themselves failed
    int canary_runs = 0;
    int last_granules = 0, last_pool = 0;  // the last draw: granules kept / created to choose them from
};

std::mutex g_mu;         // g_blocks, g_dev, the fault record, the shuffles' generator
std::mutex g_canary_mu;  // one canary at a time
std::unordered_map<void *, VmmBlock> g_blocks;
DevVmm g_dev[TVDN_MAX_DEVICES];
long long g_faults = 0, g_flushes = 0;
char g_first_fault[200] = "";

constexpr size_t kMiB = 1024 * 1024;

// TVDN_x_MIB: a count of MiB; `zero_ok`: "0" means 0 (TVDN_VMM_MIN_MIB=0: every block on granules), else unset
size_t env_mib(const char *name, size_t dflt, bool zero_ok = false)
{
    const char *e = getenv(name);
    if (!e || !*e) return dflt;
    char *end = nullptr;
    const long long v = strtoll(e, &end, 10);
    if (end == e || v < 0 || (v == 0 && !zero_ok)) return dflt;
    return (size_t)v;
}

double env_double(const char *name, double dflt)
{
    const char *e = getenv(name);
    if (!e) return dflt;
    char *end = nullptr;
    const double v = strtod(e, &end);
    return end == e ? dflt : v;
}

bool env_is(const char *name, const char *value)
{
    const char *e = getenv(name);
    return e && !strcmp(e, value);
}

// A HIP call of this file that failed: say which, on what, once on stderr and in the thread's last error; count it.
// (Must not be called with g_mu held.)
hipError_t fault(hipError_t e, const char *call, const void *addr, size_t bytes)
{
    if (e == hipSuccess) return e;
    (void)hipGetLastError();
    char msg[200];
    snprintf(msg, sizeof msg, "%s(%p, %zu bytes) failed: %s", call, addr, bytes, hipGetErrorString(e));
    fprintf(stderr, "tvdn_devmem: %s\n", msg);
    set_error("%s", msg);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_faults++ == 0) snprintf(g_first_fault, sizeof g_first_fault, "%s", msg);
    return e;
}

int vmm_state(int device)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_dev[device].state;
}

bool vmm_wanted(size_t bytes, int device)
{
    if (env_is("TVDN_VMM", "0")) return false;
    if (device < 0 || device >= TVDN_MAX_DEVICES || vmm_state(device) < 0) return false;
    return bytes >= env_mib("TVDN_VMM_MIN_MIB", 2048, true) * kMiB;
}

// Unmaps what is mapped, releases every handle, frees the reservation: everything is attempted whatever fails on the way; the
// first error is returned (and reported by fault()).
hipError_t release_block(VmmBlock &b, size_t mapped)
{
    hipError_t first = hipSuccess;
    auto note = [&](hipError_t e) { if (first == hipSuccess) first = e; };
    for (size_t i = 0; i < b.handles.size(); ++i) {
        if (i < mapped) note(fault(hipMemUnmap(b.va + i * b.G, b.G), "hipMemUnmap", b.va + i * b.G, b.G));
        note(fault(hipMemRelease(b.handles[i]), "hipMemRelease", b.va ? b.va + i * b.G : nullptr, b.G));
    }
    b.handles.clear();
    if (b.va) note(fault(hipMemAddressFree(b.va, b.va_bytes), "hipMemAddressFree", b.va, b.va_bytes));
    b.va = nullptr;
    return first;
}

// The GPU's TLBs hold on to translations that hipMemUnmap / hipMemMap have replaced; freeing a plain block makes them go.
// No memory for the 4 MiB right now (the HBM is filled to 85-92 % by design and released granules return lazily): wait for
// it, up to 200 ms; then the flush has NOT happened and the caller must not trust the range.
hipError_t tlb_flush()
{
    if (env_is("TVDN_VMM_SKIP_FLUSH", "1")) return hipSuccess;  // TEST KNOB: the canary has to notice (tests/test_gpu_vmm_guard.py)
    const size_t bytes = env_mib("TVDN_VMM_FLUSH_KIB", 4096) * 1024;
    hipError_t e = hipSuccess;
    for (int attempt = 0; attempt < 40; ++attempt) {
        void *d = nullptr;
        e = hipMalloc(&d, bytes);
        if (e == hipSuccess) {
            e = hipFree(d);
            std::lock_guard<std::mutex> lk(g_mu);
            ++g_flushes;
            break;
        }
        (void)hipGetLastError();
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
    }
    return fault(e, "TLB flush: hipMalloc + hipFree", nullptr, bytes);
}

// 1 GiB from 8 GiB on, 64 MiB below.  (Until round 6 "the power of two nearest below an eighth of the block": a 4.5 GiB state then
// sat on ten granules of 512 MiB, one per array, and its sweep time was one draw -- eight seeded draws side by side in one process,
// tools/granule_size_ab.py, profiles/r06_granule_size_seeds.jsonl: 3-D plain 512^3 0.811 ... 0.865 ms on 512 MiB granules (mean
// 0.832), 0.815 ... 0.829 on 64 MiB ones (mean 0.820); 128 x 64 x 128 x 128 (7.7 GiB): 1.408 ... 1.430 / mean 1.417 against
// 1.404 ... 1.414 / 1.408.  From 15 GiB on the 1 GiB granules are as good or better: 128^4 2.826 ... 2.839 against 2.806 ... 2.897.)
size_t granule_for(size_t bytes)
{
    const size_t G = bytes >= (size_t)8192 * kMiB ? 1024 * kMiB : 64 * kMiB;
    return env_mib("TVDN_GRANULE_MIB", G / kMiB) * kMiB;
}

hipMemAllocationProp granule_prop(int device)
{
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    return prop;
}

// read-write access to [va, va + bytes) for the owner and every peer (one descriptor per device)
hipError_t grant_access(char *va, size_t bytes, int device, const std::vector<int> &peers)
{
    std::vector<hipMemAccessDesc> acc;
    auto add = [&](int d) {
        for (const auto &a : acc)
            if (a.location.id == d) return;
        hipMemAccessDesc one;
        std::memset(&one, 0, sizeof one);
        one.location.type = hipMemLocationTypeDevice;
        one.location.id = d;
        one.flags = hipMemAccessFlagsProtReadWrite;
        acc.push_back(one);
    };
    add(device);
    for (int d : peers) add(d);
    return fault(hipMemSetAccess(va, bytes, acc.data(), acc.size()), acc.size() > 1 ? "hipMemSetAccess (with peers)" : "hipMemSetAccess", va, bytes);
}

// ---- the canary --------------------------------------------------------------------------------------------------------
__global__ void canary_fill(unsigned long long *p, size_t n, unsigned long long tag)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = tag ^ (i * 0x9E3779B97F4A7C15ull);
}

__global__ void canary_count(const unsigned long long *p, size_t n, unsigned long long tag, unsigned long long *bad)
{
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) c += p[i] != (tag ^ (i * 0x9E3779B97F4A7C15ull));
    if (c) atomicAdd(bad, c);
}

// The product's own remap sequence on two small granules (see the head of this file).  TVDN_CANARY_PASSED, _STALE or _FAILED.
// Everything runs on the null stream: creating a stream flushes the TLBs too and would hide what this looks for.
int vmm_canary(int device, char *what, size_t what_len)
{
    const size_t G = 8 * kMiB, n = G / 8;
    const unsigned long long tagA = 0xA1A1A1A1A1A1A1A1ull, tagB = 0xB2B2B2B2B2B2B2B2ull, tagC = 0xC3C3C3C3C3C3C3C3ull;
    unsigned long long *bad_d = nullptr;
    char *va = nullptr;
    hipMemGenericAllocationHandle_t h[2];
    int created = 0, mapped = 0;
    int verdict = TVDN_CANARY_FAILED;
    hipError_t e = hipSuccess;
    auto step = [&](hipError_t r, const char *call) {
        if (e == hipSuccess && r != hipSuccess) {
            e = r;
            (void)hipGetLastError();
            snprintf(what, what_len, "%s failed: %s", call, hipGetErrorString(r));
        }
        return e == hipSuccess;
    };
    auto map_pair = [&](int first, int second) {
        if (step(hipMemMap(va, G, 0, h[first], 0), "hipMemMap")) ++mapped;
        if (e == hipSuccess && step(hipMemMap(va + G, G, 0, h[second], 0), "hipMemMap")) ++mapped;
        if (e == hipSuccess) step(grant_access(va, 2 * G, device, {}), "hipMemSetAccess");
        if (e == hipSuccess) step(tlb_flush(), "the TLB flush");
    };
    auto unmap_pair = [&] {
        if (mapped > 0) step(hipMemUnmap(va, G), "hipMemUnmap");
        if (mapped > 1) step(hipMemUnmap(va + G, G), "hipMemUnmap");
        mapped = 0;
    };
    unsigned long long stale = 0;
    auto count = [&](const char *p, unsigned long long tag) {  // words of the slot at p that are not what `tag` was written as
        unsigned long long c = 0;
        if (!step(hipMemsetAsync(bad_d, 0, 8, nullptr), "hipMemsetAsync")) return;
        hipLaunchKernelGGL(canary_count, dim3(256), dim3(256), 0, nullptr, (const unsigned long long *)p, n, tag, bad_d);
        if (!step(hipGetLastError(), "the canary's read kernel")) return;
        if (!step(hipMemcpy(&c, bad_d, 8, hipMemcpyDeviceToHost), "hipMemcpy")) return;
        stale += c;
        // ... and the runtime's own copies: first, middle and last word of the slot
        for (size_t i : {(size_t)0, n / 2 + 3, n - 1}) {
            unsigned long long w = 0;
            if (!step(hipMemcpy(&w, p + i * 8, 8, hipMemcpyDeviceToHost), "hipMemcpy")) return;
            stale += w != (tag ^ (i * 0x9E3779B97F4A7C15ull));
        }
    };
    auto fill = [&](char *p, unsigned long long tag) {
        hipLaunchKernelGGL(canary_fill, dim3(256), dim3(256), 0, nullptr, (unsigned long long *)p, n, tag);
        if (step(hipGetLastError(), "the canary's write kernel")) step(hipStreamSynchronize(nullptr), "hipStreamSynchronize");
    };
    hipMemAllocationProp prop = granule_prop(device);
    step(hipMalloc((void **)&bad_d, 8), "hipMalloc");
    if (e == hipSuccess && !step(hipMemAddressReserve((void **)&va, 2 * G, G, nullptr, 0), "hipMemAddressReserve")) va = nullptr;
    for (int i = 0; i < 2 && e == hipSuccess; ++i)
        if (step(hipMemCreate(&h[i], G, &prop, 0), "hipMemCreate")) ++created;
    if (e == hipSuccess) map_pair(0, 1);  // A B
    if (e == hipSuccess) fill(va, tagA);
    if (e == hipSuccess) fill(va + G, tagB);
    if (e == hipSuccess) unmap_pair();
    if (e == hipSuccess) map_pair(1, 0);  // B A: a stale translation shows A's words at slot 0
    if (e == hipSuccess) count(va, tagB);
    if (e == hipSuccess) count(va + G, tagA);
    if (e == hipSuccess) fill(va, tagC);  // into B, through the new mapping
    if (e == hipSuccess) unmap_pair();
    if (e == hipSuccess) map_pair(0, 1);  // A B again
    if (e == hipSuccess) count(va, tagA);
    if (e == hipSuccess) count(va + G, tagC);
    if (e == hipSuccess) {
        verdict = stale ? TVDN_CANARY_STALE : TVDN_CANARY_PASSED;
        if (stale) snprintf(what, what_len, "%llu of %zu words read through a remapped range were the old mapping's", stale, 4 * (n + 3));
    }
    // clean up whatever was reached (errors here are reported but do not change the verdict)
    if (mapped > 0) (void)fault(hipMemUnmap(va, G), "hipMemUnmap (canary)", va, G);
    if (mapped > 1) (void)fault(hipMemUnmap(va + G, G), "hipMemUnmap (canary)", va + G, G);
    for (int i = 0; i < created; ++i) (void)fault(hipMemRelease(h[i]), "hipMemRelease (canary)", nullptr, G);
    if (va) (void)fault(hipMemAddressFree(va, 2 * G), "hipMemAddressFree (canary)", va, 2 * G);
    if (bad_d) (void)hipFree(bad_d);  // (flushes the TLBs as well)
    (void)hipGetLastError();
    return verdict;
}

// Run the canary if this device has not had one (or TVDN_VMM_CANARY=always) and record the verdict; true: granules may be used.
bool canary_allows(int device)
{
    if (env_is("TVDN_VMM_CANARY", "0")) return true;
    std::lock_guard<std::mutex> one(g_canary_mu);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_dev[device].state < 0) return false;
        if (g_dev[device].canary == TVDN_CANARY_PASSED && !env_is("TVDN_VMM_CANARY", "always")) return true;
    }
    char what[160] = "";
    const auto t0 = std::chrono::steady_clock::now();
    const int v = vmm_canary(device, what, sizeof what);
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_dev[device].canary = v;
        ++g_dev[device].canary_runs;
        if (v != TVDN_CANARY_PASSED) g_dev[device].state = -1;
    }
    if (v == TVDN_CANARY_STALE)
        fprintf(stderr,
                "tvdn_devmem: device %d: the GPU went on using a STALE TRANSLATION after hipMemUnmap + hipMemMap + the TLB flush this library relies on "
                "(%s).  Granules are off for this process: state blocks come from hipMalloc (results are unaffected; the placement lottery of "
                "DESIGN.md section 3 is back).\n",
                device, what);
    else if (v == TVDN_CANARY_FAILED)
        fprintf(stderr, "tvdn_devmem: device %d: virtual-memory management is not usable here (%s): state blocks come from hipMalloc.\n", device, what);
    else if (getenv("TVDN_RUN_TIMING"))
        fprintf(stderr, "tvdn_devmem: device %d: remap canary passed in %.2f ms\n", device, ms);
    return v == TVDN_CANARY_PASSED;
}

// ---- granules ------------------------------------------------------------------------------------------------------------
// How many granules of size G this process may hold on the device right now beyond what it already has: 90 % of the free
// HBM, never into the last max(4 GiB, 5 % of the device) -- another process (a second rank, a notebook with torch) must not be
// pushed into OOM for the duration of a pool -- and never beyond TVDN_HBM_LIMIT (what the planner may count on).
size_t pool_room(size_t G)
{
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    const size_t keep_free = std::max((size_t)4096 * kMiB, total_b / 20);
    size_t room = std::min((size_t)(0.9 * (double)free_b), free_b > keep_free ? free_b - keep_free : (size_t)0);
    if (const size_t cap = env_bytes("TVDN_HBM_LIMIT")) room = std::min(room, cap);
    return room / G;
}

// `need` granules of size G drawn at random from a pool of up to `want` created for the purpose (as far as pool_room and the
// time budget allow), the rest given back.  Twice `need` is what a random subset needs to be reliably fast (subsets of a pool
// of the block's own size: up to 11.72 ms for config 2, of 1.25 x: up to 11.57, of 2 x and more: <= 11.25;
// profiles/r05_vmm_spread_*.jsonl); a caller that gives the extras a second or more -- a state someone keeps -- gets that
// much whatever it takes, within four times its budget (creating a granule takes ~18 ms when the driver has to clear the
// memory first, so a fresh box can spend 1.5 s on little more than the block itself).
hipError_t draw_granules(size_t need, size_t G, int device, double budget_s, std::vector<hipMemGenericAllocationHandle_t> &out, size_t *pool_size)
{
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    const hipMemAllocationProp prop = granule_prop(device);
    const double factor = std::max(1.0, env_double("TVDN_SPREAD", 3.0));
    const double budget = env_double("TVDN_SPREAD_S", budget_s);
    size_t want = std::max(need, (size_t)(factor * (double)need + 0.5));
    want = std::max(need, std::min(want, pool_room(G)));
    const size_t floor_pool = budget >= 1.0 ? std::min(want, 2 * need) : need;
    std::vector<hipMemGenericAllocationHandle_t> pool;
    pool.reserve(want);
    hipError_t e = hipSuccess;
    double t_need = 0.0;
    while (pool.size() < want) {
        if (pool.size() >= need && since() - t_need > budget && (pool.size() >= floor_pool || since() - t_need > 4.0 * budget)) break;  // the extras have had their time
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, G, &prop, 0);
        if (e != hipSuccess) break;
        pool.push_back(h);
        if (pool.size() == need) t_need = since();
    }
    (void)hipGetLastError();  // a refusal among the extras only ends the pool
    if (pool.size() < need) {
        for (auto h : pool) (void)fault(hipMemRelease(h), "hipMemRelease", nullptr, G);
        return e != hipSuccess ? e : hipErrorOutOfMemory;
    }
    {
        static std::mt19937_64 rng(0x9e3779b97f4a7c15ULL ^ (unsigned long long)env_mib("TVDN_SPREAD_SEED", 0));  // a fixed sequence per process: runs repeat (the env: measurement, other draws)
        std::lock_guard<std::mutex> lk(g_mu);
        std::shuffle(pool.begin(), pool.end(), rng);
    }
    for (size_t i = need; i < pool.size(); ++i) (void)fault(hipMemRelease(pool[i]), "hipMemRelease", nullptr, G);
    if (pool_size) *pool_size = pool.size();
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_dev[device].last_granules = (int)need;
        g_dev[device].last_pool = (int)pool.size();
    }
    pool.resize(need);
    out.insert(out.end(), pool.begin(), pool.end());
    return hipSuccess;
}

// the granules of `b`, dealt out in a fresh random order, as one range the device (and its peers) may read and write
hipError_t map_block(VmmBlock &b)
{
    {
        static std::mt19937_64 rng(0x7476646eULL ^ (unsigned long long)env_mib("TVDN_SPREAD_SEED", 0));
        std::lock_guard<std::mutex> lk(g_mu);
        std::shuffle(b.handles.begin(), b.handles.end(), rng);
    }
    const size_t need = b.handles.size(), G = b.G;
    b.va_bytes = need * G;
    hipError_t e = fault(hipMemAddressReserve((void **)&b.va, b.va_bytes, G, nullptr, 0), "hipMemAddressReserve", nullptr, b.va_bytes);
    if (e != hipSuccess) {
        b.va = nullptr;
        (void)release_block(b, 0);
        return e;
    }
    size_t mapped = 0;
    for (size_t i = 0; i < need && e == hipSuccess; ++i) {
        e = fault(hipMemMap(b.va + i * G, G, 0, b.handles[i], 0), "hipMemMap", b.va + i * G, G);
        if (e == hipSuccess) ++mapped;
    }
    if (e == hipSuccess) e = grant_access(b.va, b.va_bytes, b.device, b.peers);
    if (e == hipSuccess) e = tlb_flush();  // no flush, no trust: the range may still translate to what was mapped there before
    if (e != hipSuccess) {
        (void)release_block(b, mapped);
        (void)tlb_flush();
    }
    return e;
}

hipError_t vmm_alloc(void **p, size_t bytes, int device, double spread_budget_s, DevAllocInfo *info, const std::vector<int> &peers)
{
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    VmmBlock b;
    b.G = granule_for(bytes);
    b.device = device;
    b.peers = peers;
    const size_t need = (bytes + b.G - 1) / b.G;  // equal granules only (see the head of this file)
    size_t pool_size = 0;
    hipError_t e = draw_granules(need, b.G, device, spread_budget_s, b.handles, &pool_size);
    if (e != hipSuccess) return e;
    b.pool = pool_size;
    const double t_created = since();
    e = map_block(b);
    if (e != hipSuccess) return e;
    if (getenv("TVDN_RUN_TIMING"))
        fprintf(stderr, "tvdn_devmem: %zu granules of %zu MiB (pool %zu): created in %.3f s, dealt out and mapped in %.3f s\n", need, b.G / kMiB, pool_size, t_created,
                since() - t_created);
    if (info) {
        info->granule_bytes = (int64_t)b.G;
        info->granules = (int32_t)need;
        info->pool = (int32_t)pool_size;
        info->seconds = since();
    }
    *p = b.va;
    std::lock_guard<std::mutex> lk(g_mu);
    g_blocks.emplace((void *)b.va, std::move(b));
    return hipSuccess;
}

// released granules come back once the driver has cleared them: wait (up to `seconds`) until `bytes` are free again
void wait_for_free(size_t bytes, double seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        size_t f = 0, t = 0;
        if (hipMemGetInfo(&f, &t) != hipSuccess || f >= bytes) break;
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
    (void)hipGetLastError();
}

}  // namespace

// kind: TVDN_MEM_GRANULES / TVDN_MEM_PLAIN.  The calling thread's current device must be `device`.  spread_budget_s: how long
// the creation of granules beyond the block's own may take (what a better placement is worth to the caller).  peers: other
// devices that will read and write the block (hipMemSetAccess with one descriptor per device).
hipError_t dev_alloc(void **p, size_t bytes, int device, int *kind, double spread_budget_s, DevAllocInfo *info, const int *peers, int n_peers)
{
    if (kind) *kind = TVDN_MEM_PLAIN;
    if (info) std::memset(info, 0, sizeof *info);
    if (vmm_wanted(bytes, device) && canary_allows(device)) {
        const std::vector<int> pv(peers, peers + (peers ? n_peers : 0));
        hipError_t e = vmm_alloc(p, bytes, device, spread_budget_s, info, pv);
        if (e == hipErrorOutOfMemory) {
            // granules this process released a moment ago may not be back yet (the driver clears them first): wait for them,
            // then once more without a pool
            (void)hipGetLastError();
            wait_for_free(bytes + bytes / 16, 2.0);
            e = vmm_alloc(p, bytes, device, 0.0, info, pv);
        }
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> lk(g_mu);
            if (g_dev[device].state == 0) g_dev[device].state = 1;
            if (kind) *kind = TVDN_MEM_GRANULES;
            return e;
        }
        (void)hipGetLastError();
        if (e == hipErrorOutOfMemory) return e;  // hipMalloc would find no more memory than the granules did
        std::lock_guard<std::mutex> lk(g_mu);
        if (g_dev[device].state == 0 && pv.empty()) g_dev[device].state = -1;  // a runtime without virtual-memory management: plain blocks from now on
    }
    return hipMalloc(p, bytes);
}

// A block on granules changes size instead of being freed and allocated anew (contents undefined afterwards): the granules it
// has stay -- creating 236 of them takes 2.7 s on a fresh device and 5-7 s behind a big release, mapping them 4 ms --, missing ones
// are drawn like a new block's (a random subset of a pool, draw_granules), surplus ones given back, all dealt out in a new random
// order.  hipErrorNotSupported: not such a block (or the new size wants another granule size, or plain memory): the caller frees
// and allocates.  Any other error: everything has been given back, *p is nullptr.
hipError_t dev_resize(void **p, size_t bytes, int device, double spread_budget_s)
{
    if (!p || !*p || !vmm_wanted(bytes, device)) return hipErrorNotSupported;
    VmmBlock b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_blocks.find(*p);
        if (it == g_blocks.end() || it->second.device != device || it->second.G != granule_for(bytes) || !it->second.peers.empty()) return hipErrorNotSupported;
        b = std::move(it->second);
        g_blocks.erase(it);
    }
    const size_t G = b.G, need = (bytes + G - 1) / G;
    hipError_t e = fault(hipDeviceSynchronize(), "hipDeviceSynchronize", b.va, b.va_bytes);  // nothing in flight may still touch the range when it is unmapped
    for (size_t i = 0; i < b.handles.size(); ++i) {
        const hipError_t eu = fault(hipMemUnmap(b.va + i * G, G), "hipMemUnmap", b.va + i * G, G);
        if (e == hipSuccess) e = eu;
    }
    if (b.va) {
        const hipError_t ef = fault(hipMemAddressFree(b.va, b.va_bytes), "hipMemAddressFree", b.va, b.va_bytes);
        if (e == hipSuccess) e = ef;
    }
    b.va = nullptr;
    if (e == hipSuccess && b.handles.size() < need) {
        size_t drawn_from = 0;
        const size_t missing = need - b.handles.size();
        e = draw_granules(missing, G, device, spread_budget_s, b.handles, &drawn_from);
        b.pool = b.handles.size() - missing + drawn_from;  // (what the whole was chosen from: its own granules and the new ones' pool)
    }
    if (e != hipSuccess) {  // (not enough memory for the larger block, or the runtime refused a step): everything goes back
        (void)release_block(b, 0);
        (void)tlb_flush();
        *p = nullptr;
        return e;
    }
    while (b.handles.size() > need) {
        (void)fault(hipMemRelease(b.handles.back()), "hipMemRelease", nullptr, G);
        b.handles.pop_back();
    }
    b.pool = std::max(b.pool, need);
    e = map_block(b);
    if (e != hipSuccess) {
        *p = nullptr;
        return e;
    }
    *p = b.va;
    std::lock_guard<std::mutex> lk(g_mu);
    g_blocks.emplace((void *)b.va, std::move(b));
    return hipSuccess;
}

// A kept block whose granules were chosen from a SHORT pool -- the first run of a process was a short one and gave the extras
// 5 % of its sweep time, i.e. next to nothing -- is topped up when a run that can afford it takes the block over: up to
// `spread_budget_s` seconds of further granules, a fresh random subset of its own and the new ones, the rest given back, a new
// random order at a new address (contents undefined, as after dev_resize).  Nothing happens (hipSuccess, *p unchanged) when the
// pool was already twice the block, the budget is below 50 ms, or the block is not on granules.  On an error everything has been
// given back and *p is nullptr.
hipError_t dev_upgrade(void **p, int device, double spread_budget_s)
{
    const double budget = env_double("TVDN_SPREAD_S", spread_budget_s);
    if (!p || !*p || budget < 0.05) return hipSuccess;
    VmmBlock b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_blocks.find(*p);
        if (it == g_blocks.end() || it->second.device != device || !it->second.peers.empty()) return hipSuccess;
        const size_t need = it->second.handles.size();
        if (it->second.pool >= 2 * need) return hipSuccess;
        b = std::move(it->second);
        g_blocks.erase(it);
    }
    const auto t0 = std::chrono::steady_clock::now();
    const size_t G = b.G, need = b.handles.size(), pool_before = b.pool;
    // the extras first, while the block is still whole: a draw that fails leaves it as it was
    const double factor = std::max(1.0, env_double("TVDN_SPREAD", 3.0));
    const size_t target = std::max(need, (size_t)(factor * (double)need + 0.5));
    const size_t want_extra = std::min(target - need, pool_room(G));
    const hipMemAllocationProp prop = granule_prop(device);
    std::vector<hipMemGenericAllocationHandle_t> extra;
    while (extra.size() < want_extra && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < budget) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, G, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        extra.push_back(h);
    }
    if (extra.empty()) {  // nothing to choose from: the block stays as it is
        std::lock_guard<std::mutex> lk(g_mu);
        g_blocks.emplace((void *)b.va, std::move(b));
        return hipSuccess;
    }
    hipError_t e = fault(hipDeviceSynchronize(), "hipDeviceSynchronize", b.va, b.va_bytes);
    for (size_t i = 0; i < need; ++i) {
        const hipError_t eu = fault(hipMemUnmap(b.va + i * G, G), "hipMemUnmap", b.va + i * G, G);
        if (e == hipSuccess) e = eu;
    }
    {
        const hipError_t ef = fault(hipMemAddressFree(b.va, b.va_bytes), "hipMemAddressFree", b.va, b.va_bytes);
        if (e == hipSuccess) e = ef;
    }
    b.va = nullptr;
    b.handles.insert(b.handles.end(), extra.begin(), extra.end());
    if (e != hipSuccess) {
        (void)release_block(b, 0);
        (void)tlb_flush();
        *p = nullptr;
        return e;
    }
    {
        static std::mt19937_64 rng(0x75706772ULL);
        std::lock_guard<std::mutex> lk(g_mu);
        std::shuffle(b.handles.begin(), b.handles.end(), rng);
    }
    while (b.handles.size() > need) {
        (void)fault(hipMemRelease(b.handles.back()), "hipMemRelease", nullptr, G);
        b.handles.pop_back();
    }
    b.pool = need + extra.size();
    e = map_block(b);
    if (e != hipSuccess) {
        *p = nullptr;
        return e;
    }
    if (getenv("TVDN_RUN_TIMING"))
        fprintf(stderr, "tvdn_devmem: kept block of %zu granules re-drawn from its own and %zu new ones (pool %zu -> %zu) in %.3f s\n", need, extra.size(), pool_before, b.pool,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_dev[device].last_granules = (int)need;
        g_dev[device].last_pool = (int)b.pool;
    }
    *p = b.va;
    std::lock_guard<std::mutex> lk(g_mu);
    g_blocks.emplace((void *)b.va, std::move(b));
    return hipSuccess;
}

hipError_t dev_free(void *p)
{
    if (!p) return hipSuccess;
    VmmBlock b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_blocks.find(p);
        if (it == g_blocks.end()) return hipFree(p);
        b = std::move(it->second);
        g_blocks.erase(it);
    }
    DeviceRestore restore;
    hipError_t first = fault(hipSetDevice(b.device), "hipSetDevice", p, 0);
    auto note = [&](hipError_t e) { if (first == hipSuccess) first = e; };
    note(fault(hipDeviceSynchronize(), "hipDeviceSynchronize", p, b.va_bytes));  // as hipFree does: nothing in flight may still touch the range when it is unmapped
    size_t free0 = 0, total0 = 0;
    if (hipMemGetInfo(&free0, &total0) != hipSuccess) (void)hipGetLastError();
    const size_t va_bytes = b.va_bytes;
    note(release_block(b, b.handles.size()));
    note(tlb_flush());  // ... and no translation of it may outlive it (the addresses come back with the next block)
    // hipMemRelease returns before the driver has the memory back (it clears released pages in the background): a caller that
    // plans its next block by hipMemGetInfo right away would see it still missing -- the streamed bench entry kept 53 rows
    // resident instead of 56 that way.  hipFree has come back with the memory; so does this, within reason.
    if (va_bytes >= (size_t(1) << 30)) wait_for_free(free0 + va_bytes / 10 * 9, 3.0);
    return first;
}

int dev_kind(const void *p)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_blocks.count(const_cast<void *>(p)) ? TVDN_MEM_GRANULES : TVDN_MEM_PLAIN;
}

}  // namespace tvdn

using namespace tvdn;

extern "C" int tvdn_mem_alloc(void **ptr, int64_t bytes, int device, int32_t *kind)
{
    return tvdn_mem_alloc_shared(ptr, bytes, device, nullptr, 0, kind);
}

extern "C" int tvdn_mem_alloc_shared(void **ptr, int64_t bytes, int device, const int32_t *peers, int32_t n_peers, int32_t *kind)
{
    TVDN_REQUIRE(ptr != nullptr && bytes > 0, "bad argument");
    TVDN_REQUIRE(n_peers >= 0 && n_peers <= TVDN_MAX_DEVICES && (n_peers == 0 || peers != nullptr), "bad peer list");
    DeviceRestore restore;
    TVDN_HIP(hipSetDevice(device));
    int k = TVDN_MEM_PLAIN;
    int pl[TVDN_MAX_DEVICES];
    for (int i = 0; i < n_peers; ++i) pl[i] = peers[i];
    const hipError_t e = dev_alloc(ptr, (size_t)bytes, device, &k, 1.5, nullptr, pl, n_peers);  // a state someone keeps: worth 1.5 s of looking
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("%lld bytes of device memory on device %d: %s", (long long)bytes, device, hipGetErrorString(e));
        return TVDN_ERR_HIP;
    }
    if (kind) *kind = k;
    return TVDN_OK;
}

extern "C" int tvdn_mem_free(void *ptr)
{
    const hipError_t e = dev_free(ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("releasing device memory %p: %s (the first failing call is on stderr and in tvdn_mem_status)", ptr, hipGetErrorString(e));
        return TVDN_ERR_HIP;
    }
    return TVDN_OK;
}

extern "C" int tvdn_mem_resize(void **ptr, int64_t bytes, int device)
{
    TVDN_REQUIRE(ptr != nullptr && *ptr != nullptr && bytes > 0, "bad argument");
    DeviceRestore restore;
    TVDN_HIP(hipSetDevice(device));
    const hipError_t e = dev_resize(ptr, (size_t)bytes, device, 1.5);
    if (e == hipErrorNotSupported) {
        set_error("%p is not a block on granules of device %d that %lld bytes would fit the granule size of: free it and allocate", *ptr, device, (long long)bytes);
        return TVDN_ERR_UNSUPPORTED;
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error("re-dealing a block at %lld bytes on device %d: %s (the block has been given back)", (long long)bytes, device, hipGetErrorString(e));
        return TVDN_ERR_HIP;
    }
    return TVDN_OK;
}

extern "C" int tvdn_mem_status(int device, tvdn_mem_status_out *out)
{
    TVDN_REQUIRE(out != nullptr, "NULL argument");
    TVDN_REQUIRE(device >= 0 && device < TVDN_MAX_DEVICES, "device %d out of range", device);
    std::memset(out, 0, sizeof *out);
    std::lock_guard<std::mutex> lk(g_mu);
    out->vmm_state = env_is("TVDN_VMM", "0") ? -1 : g_dev[device].state;
    out->canary = g_dev[device].canary;
    out->canary_runs = g_dev[device].canary_runs;
    out->faults = (int32_t)std::min<long long>(g_faults, INT32_MAX);
    out->flushes = g_flushes;
    for (const auto &kv : g_blocks)
        if (kv.second.device == device) {
            ++out->blocks;
            out->granules += (int64_t)kv.second.handles.size();
            out->bytes += (int64_t)kv.second.va_bytes;
        }
    out->last_granules = g_dev[device].last_granules;
    out->last_pool = g_dev[device].last_pool;
    snprintf(out->first_fault, sizeof out->first_fault, "%s", g_first_fault);
    return TVDN_OK;
}

extern "C" int tvdn_mem_selftest(int device)
{
    TVDN_REQUIRE(device >= 0 && device < TVDN_MAX_DEVICES, "device %d out of range", device);
    DeviceRestore restore;
    TVDN_HIP(hipSetDevice(device));
    char what[160] = "";
    int v;
    {
        std::lock_guard<std::mutex> one(g_canary_mu);
        v = vmm_canary(device, what, sizeof what);
        std::lock_guard<std::mutex> lk(g_mu);
        g_dev[device].canary = v;
        ++g_dev[device].canary_runs;
        if (v != TVDN_CANARY_PASSED) g_dev[device].state = -1;
    }
    if (v != TVDN_CANARY_PASSED) {
        set_error("remap canary on device %d: %s", device, what);
        return v == TVDN_CANARY_STALE ? TVDN_ERR_UNSUPPORTED : TVDN_ERR_HIP;
    }
    return TVDN_OK;
}
