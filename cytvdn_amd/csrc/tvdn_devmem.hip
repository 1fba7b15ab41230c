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
//     order give another time (11.26 ... 11.72), so it is the arrangement, not a property of single granules (a least-squares fit
//     of per-granule costs predicts nothing: its "best" and "worst" 61 of 261 both sweep in 11.18 ms);
//   * 61 granules drawn AT RANDOM from a pool spread over two to four times as much HBM sweep in 11.09 ... 11.25 ms, every one
//     of 150 draws on boxes whose contiguous windows reach 11.8 (from a pool of the state's own size: up to 11.72).
// So placement can be had by construction: create more granules than the state needs while that is cheap, keep a random
// subset in random order, give the rest back.  No probe sweeps, no second candidate; csrc/tvdn_run.hip's audition is off on
// such a block.
//
// What.  dev_alloc / dev_free: blocks of TVDN_VMM_MIN_MIB (2 GiB) or more come from granules, smaller ones and every block when
// the runtime refuses virtual-memory management (or TVDN_VMM=0) from hipMalloc.
//   * Pool.  Up to TVDN_SPREAD (3) times the block's granules are created, as far as 90 % of the free HBM and `spread_budget_s`
//     seconds allow (creating granules is instant on cleared memory and ~18 ms per GiB when the driver has to clear it first;
//     the caller says what the run is worth: tvdn_run 5 % of its expected sweep time, at least 0.25 s).
//   * Equal granules.  Every granule of a block has the same size and the block is rounded up to a whole number of them: on ROCm
//     7.2 a reservation mapped from handles of different sizes confuses the runtime's own address lookups -- hipMemcpy /
//     hipMemset into the part behind the odd handle land elsewhere while kernels see the right bytes (tools/ubench/
//     vmm_memset.hip).  1 GiB from 8 GiB on (61 mappings take 11 ms and the translation misses stay those of hipMalloc; 64 MiB
//     granules: 100 x the UTCL1 misses, the L2 TLB busy 98 % of the sweep), else the power of two nearest below an eighth of
//     the block, at least 64 MiB: at most 12.5 % more than asked for, 1.6 % for the 60 GiB of configs[1].
//   * Stale translations.  On ROCm 7.2 hipMemUnmap / hipMemMap do not make the GPU forget the old translation of an address:
//     kernels and copies go on using the physical memory that WAS mapped there (tools/ubench/vmm_remap_check.hip), also after
//     hipMemAddressFree + a new reservation that overlaps the old one.  A hipFree of any plain block flushes the TLBs
//     (vmm_remap_flush.hip), so every map and every unmap here ends with a 4 MiB hipMalloc + hipFree.
// tvdn_mem_alloc / tvdn_mem_free export the same thing (cytvdn_amd/engine.py puts a slab's state on it).
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
    int device = 0;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<size_t> sizes;
};

std::mutex g_mu;
std::unordered_map<void *, VmmBlock> g_blocks;
int g_vmm_state[TVDN_MAX_DEVICES] = {};  // 0 not tried, 1 works, -1 refused by the runtime

size_t env_mib(const char *name, size_t dflt)
{
    const char *e = getenv(name);
    if (!e) return dflt;
    const long long v = atoll(e);
    return v > 0 ? (size_t)v : dflt;
}

constexpr size_t kMiB = 1024 * 1024;

bool vmm_wanted(size_t bytes, int device)
{
    const char *e = getenv("TVDN_VMM");
    if (e && atoi(e) == 0) return false;
    if (device < 0 || device >= TVDN_MAX_DEVICES || g_vmm_state[device] < 0) return false;
    return bytes >= env_mib("TVDN_VMM_MIN_MIB", 2048) * kMiB;
}

void release_block(VmmBlock &b, size_t mapped)
{
    size_t off = 0;
    for (size_t i = 0; i < b.handles.size(); ++i) {
        if (i < mapped) (void)hipMemUnmap(b.va + off, b.sizes[i]);
        (void)hipMemRelease(b.handles[i]);
        off += b.sizes[i];
    }
    if (b.va) (void)hipMemAddressFree(b.va, b.va_bytes);
    (void)hipGetLastError();
}

// the GPU's TLBs hold on to translations that hipMemUnmap / hipMemMap have replaced; freeing a plain block makes them go
void tlb_flush()
{
    void *d = nullptr;
    if (hipMalloc(&d, 4 * kMiB) == hipSuccess) (void)hipFree(d);
    (void)hipGetLastError();
}

double env_double(const char *name, double dflt)
{
    const char *e = getenv(name);
    if (!e) return dflt;
    char *end = nullptr;
    const double v = strtod(e, &end);
    return end == e ? dflt : v;
}

size_t granule_for(size_t bytes)
{
    size_t G = 1024 * kMiB;
    while (G > 64 * kMiB && G > bytes / 8) G /= 2;
    return env_mib("TVDN_GRANULE_MIB", G / kMiB) * kMiB;
}

// the granules of `b`, dealt out in a fresh random order, as one range the device may read and write
hipError_t map_block(VmmBlock &b, size_t G, int device)
{
    {
        static std::mt19937_64 rng(0x7476646eULL);  // a fixed sequence per process: runs repeat
        std::lock_guard<std::mutex> lk(g_mu);
        std::shuffle(b.handles.begin(), b.handles.end(), rng);
    }
    const size_t need = b.handles.size();
    b.device = device;
    b.sizes.assign(need, G);
    b.va_bytes = need * G;
    hipError_t e = hipMemAddressReserve((void **)&b.va, b.va_bytes, G, nullptr, 0);
    if (e != hipSuccess) {
        b.va = nullptr;
        release_block(b, 0);
        return e;
    }
    size_t mapped = 0;
    for (size_t i = 0; i < need && e == hipSuccess; ++i) {
        e = hipMemMap(b.va + i * G, G, 0, b.handles[i], 0);
        if (e == hipSuccess) ++mapped;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc;
        std::memset(&acc, 0, sizeof acc);
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(b.va, b.va_bytes, &acc, 1);
    }
    if (e != hipSuccess) release_block(b, mapped);
    tlb_flush();
    return e;
}

hipError_t vmm_alloc(void **p, size_t bytes, int device, double spread_budget_s, DevAllocInfo *info)
{
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    const size_t G = granule_for(bytes);
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    const size_t need = (bytes + G - 1) / G;  // equal granules only (see the head of this file)
    // the pool: as many more granules as the factor, the free HBM and the time budget allow
    const double factor = std::max(1.0, env_double("TVDN_SPREAD", 3.0));
    const double budget = env_double("TVDN_SPREAD_S", spread_budget_s);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
        (void)hipGetLastError();
        free_b = 0;
    }
    size_t want = std::max(need, (size_t)(factor * (double)need + 0.5));
    want = std::max(need, std::min(want, (size_t)(0.9 * (double)free_b) / G));
    std::vector<hipMemGenericAllocationHandle_t> pool;
    pool.reserve(want);
    hipError_t e = hipSuccess;
    double t_need = 0.0;
    // Twice the block is what a random subset needs to be reliably fast (subsets of a pool of the block's own size: up to 11.72 ms for
    // config 2, of 1.25 x: up to 11.57, of 2 x and more: <= 11.25; profiles/r05_vmm_spread_*.jsonl).  A caller that gives the extras a
    // second or more -- a state someone keeps -- gets that much whatever it takes, within four times its budget (creating a
    // granule takes ~18 ms when the driver has to clear the memory first, so a fresh box can spend its 1.5 s on little more than
    // the block itself; one bench line of a fresh box read 91.2 -- 11.76 ms per sweep -- where the twenty others of the round read
    // 94.2-95.8: it left no record of its pool, but a pool cut short is the one known way to such a time).
    const size_t floor_pool = budget >= 1.0 ? std::min(want, 2 * need) : need;
    while (pool.size() < want) {
        if (pool.size() >= need && since() - t_need > budget && (pool.size() >= floor_pool || since() - t_need > 4.0 * budget)) break;  // the extras have had their time
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, G, &prop, 0);
        if (e != hipSuccess) break;
        pool.push_back(h);
        if (pool.size() == need) t_need = since();
    }
    if (pool.size() < need) {
        for (auto h : pool) (void)hipMemRelease(h);
        (void)hipGetLastError();
        return e != hipSuccess ? e : hipErrorOutOfMemory;
    }
    (void)hipGetLastError();  // a refusal among the extras only ends the pool
    const double t_created = since();
    // a random subset in random order: shuffle, keep the first `need`, give the rest back, deal the kept ones out again
    {
        static std::mt19937_64 rng(0x9e3779b97f4a7c15ULL);
        std::lock_guard<std::mutex> lk(g_mu);
        std::shuffle(pool.begin(), pool.end(), rng);
    }
    for (size_t i = need; i < pool.size(); ++i) (void)hipMemRelease(pool[i]);
    const size_t pool_size = pool.size();
    pool.resize(need);
    VmmBlock b;
    b.handles = std::move(pool);
    e = map_block(b, G, device);
    if (e != hipSuccess) return e;
    if (getenv("TVDN_RUN_TIMING"))
        fprintf(stderr, "tvdn_devmem: %zu granules of %zu MiB (pool %zu): created in %.3f s, dealt out and mapped in %.3f s\n", need, G / kMiB, pool_size, t_created,
                since() - t_created);
    if (info) {
        info->granule_bytes = (int64_t)G;
        info->granules = (int32_t)need;
        info->pool = (int32_t)pool_size;
        info->seconds = since();
    }
    *p = b.va;
    std::lock_guard<std::mutex> lk(g_mu);
    g_blocks.emplace((void *)b.va, std::move(b));
    return hipSuccess;
}

}  // namespace

// kind: TVDN_MEM_GRANULES / TVDN_MEM_PLAIN.  The calling thread's current device must be `device`.  spread_budget_s: how long
// the creation of granules beyond the block's own may take (what a better placement is worth to the caller).
hipError_t dev_alloc(void **p, size_t bytes, int device, int *kind, double spread_budget_s, DevAllocInfo *info)
{
    if (kind) *kind = TVDN_MEM_PLAIN;
    if (info) std::memset(info, 0, sizeof *info);
    if (vmm_wanted(bytes, device)) {
        const hipError_t e = vmm_alloc(p, bytes, device, spread_budget_s, info);
        if (e == hipSuccess) {
            g_vmm_state[device] = 1;
            if (kind) *kind = TVDN_MEM_GRANULES;
            return e;
        }
        (void)hipGetLastError();
        if (e == hipErrorOutOfMemory) return e;  // hipMalloc would find no more memory than the granules did
        if (g_vmm_state[device] == 0) g_vmm_state[device] = -1;  // a runtime without virtual-memory management: plain blocks from now on
    }
    return hipMalloc(p, bytes);
}

// A block on granules changes size instead of being freed and allocated anew (contents undefined afterwards): the granules it
// has stay -- creating 236 of them takes 2.7 s on a fresh device and 5-7 s behind a big release, mapping them 4 ms --, missing ones
// are created, surplus ones given back, all dealt out in a new random order.  hipErrorNotSupported: not such a block (or the new
// size wants another granule size, or plain memory): the caller frees and allocates.
hipError_t dev_resize(void **p, size_t bytes, int device)
{
    if (!p || !*p || !vmm_wanted(bytes, device)) return hipErrorNotSupported;
    VmmBlock b;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_blocks.find(*p);
        if (it == g_blocks.end() || it->second.device != device || it->second.sizes.empty() || it->second.sizes[0] != granule_for(bytes)) return hipErrorNotSupported;
        b = std::move(it->second);
        g_blocks.erase(it);
    }
    const size_t G = b.sizes[0], need = (bytes + G - 1) / G;
    (void)hipDeviceSynchronize();  // nothing in flight may still touch the range when it is unmapped
    size_t off = 0;
    for (size_t i = 0; i < b.handles.size(); ++i, off += G) (void)hipMemUnmap(b.va + off, G);
    if (b.va) (void)hipMemAddressFree(b.va, b.va_bytes);
    b.va = nullptr;
    (void)hipGetLastError();
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    hipError_t e = hipSuccess;
    while (b.handles.size() < need && e == hipSuccess) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, G, &prop, 0);
        if (e == hipSuccess) b.handles.push_back(h);
    }
    if (e != hipSuccess) {  // not enough memory for the larger block: everything goes back
        for (auto h : b.handles) (void)hipMemRelease(h);
        (void)hipGetLastError();
        tlb_flush();
        *p = nullptr;
        return e;
    }
    while (b.handles.size() > need) {
        (void)hipMemRelease(b.handles.back());
        b.handles.pop_back();
    }
    e = map_block(b, G, device);
    if (e != hipSuccess) {
        *p = nullptr;
        return e;
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
    int prev = -1;
    (void)hipGetDevice(&prev);
    (void)hipSetDevice(b.device);
    (void)hipDeviceSynchronize();  // as hipFree does: nothing in flight may still touch the range when it is unmapped
    size_t free0 = 0, total0 = 0;
    (void)hipMemGetInfo(&free0, &total0);
    release_block(b, b.handles.size());
    tlb_flush();                   // ... and no translation of it may outlive it (the addresses come back with the next block)
    // hipMemRelease returns before the driver has the memory back (it clears released pages in the background): a caller that
    // plans its next block by hipMemGetInfo right away would see it still missing -- the streamed bench entry kept 53 rows
    // resident instead of 56 that way.  hipFree has come back with the memory; so does this, within reason.
    if (b.va_bytes >= (size_t(1) << 30)) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            size_t f = 0, t = 0;
            if (hipMemGetInfo(&f, &t) != hipSuccess || f >= free0 + b.va_bytes / 10 * 9) break;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 3.0) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
        (void)hipGetLastError();
    }
    if (prev >= 0) (void)hipSetDevice(prev);
    return hipSuccess;
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
    TVDN_REQUIRE(ptr != nullptr && bytes > 0, "bad argument");
    DeviceRestore restore;
    TVDN_HIP(hipSetDevice(device));
    int k = TVDN_MEM_PLAIN;
    const hipError_t e = dev_alloc(ptr, (size_t)bytes, device, &k, 1.5, nullptr);  // a state someone keeps: worth 1.5 s of looking
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
        set_error("releasing device memory %p: %s", ptr, hipGetErrorString(e));
        return TVDN_ERR_HIP;
    }
    return TVDN_OK;
}
