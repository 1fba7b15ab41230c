// Host <-> HBM transfers of whole datacubes at PCIe speed from ordinary (pageable) host arrays.
//
// denoise3D/4D take and return NumPy arrays (cyTVDN/cyTVDN.py:19-31, :244-247: `recon = datacube.copy()` on the way
// in, a fresh array on the way out), i.e. pageable memory.  hipMemcpy from/to pageable memory stages through one
// bounce buffer with one host thread doing the copy, and a freshly allocated destination is page-faulted by that same
// thread: 10-25 GB/s where the link does 50+.  Here `kLanes` host threads each own two pinned bounce buffers and a
// HIP stream and move every kLanes-th chunk of the array on their own: memcpy into (or out of) the pinned buffer
// overlaps the DMA of the lane's other buffer and the work of the other lanes, and the first touch of a fresh
// destination is spread over all lanes.  The pinned buffers are allocated once per process and device.
//
// Both entry points are synchronous (they return when the bytes have arrived) and order themselves only against
// their own streams: the caller synchronises the stream that produced / will consume the device buffer.
#include <algorithm>
#include <atomic>
#include <mutex>
#include <thread>

#include "tvdn_stream_parts.hpp"  // PinnedBuf: page-locked host memory by registration of touched huge pages

namespace tvdn {

constexpr int kLanes = 8;
constexpr size_t kChunk = size_t(16) << 20;  // per bounce buffer

struct Lane {
    void *pin[2] = {nullptr, nullptr};
    hipStream_t stream = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
};

struct HostIo {
    int device = -1;
    std::mutex mu;  // one transfer at a time per DEVICE: its bounce buffers are shared (devices have lanes, buffers and links of their own)
    std::atomic<bool> ready{false};
    Lane lane[kLanes];
    PinnedBuf pinned;  // the lanes' bounce buffers: ONE registration (never released: the process's staging for good)
};

static HostIo g_io[16];

static std::atomic<int> g_lanes_cap{0};
static std::atomic<int> g_cap_users{0};

// Transfers that run BESIDE a thread launching sweeps (tvdn_run's pipelined start and end) use fewer staging lanes: with
// eight host threads copying through their pinned buffers the launching thread falls behind and the overlap is lost
// (config 2, 50 iterations from host memory: 0.72-0.76 s with 8 lanes, 0.62-0.64 s with 4-6; profiles/r03_e2e_pipelined.txt).
// Reference-counted: concurrent pipelined runs each hold the cap, and it lifts when the LAST of them ends (a plain
// set/reset pair let the first run to finish lift it for the others mid-flight).  n > 0 takes a hold, n == 0 drops one.
void io_cap_lanes(int n)
{
    // one lock for count and cap together: a release that sees "last user" and another thread's fresh hold must not interleave
    // (the drop of a cap that had just been taken again)
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (n > 0) {
        g_lanes_cap.store(n);
        g_cap_users.fetch_add(1);
    } else if (g_cap_users.load() <= 1) {
        g_cap_users.store(0);
        g_lanes_cap.store(0);
    } else {
        g_cap_users.fetch_sub(1);
    }
}

int make_stream(hipStream_t *s, int level)
{
    const char *e = getenv("TVDN_STREAM_PRIO");
    int least = 0, greatest = 0;
    if (level != 0 && !(e && atoi(e) == 0)) TVDN_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
    const int prio = level > 0 ? greatest : (level < 0 ? least : 0);
    if (prio == 0) {
        TVDN_HIP(hipStreamCreateWithFlags(s, hipStreamNonBlocking));
    } else {
        TVDN_HIP(hipStreamCreateWithPriority(s, hipStreamNonBlocking, prio));
    }
    return TVDN_OK;
}

static int lanes_wanted()
{
    unsigned hc = std::thread::hardware_concurrency();
    int n = hc ? (int)hc : 4;
    const char *e = getenv("TVDN_IO_LANES");
    if (e && atoi(e) > 0)
        n = atoi(e);
    else if (g_lanes_cap.load() > 0)
        n = std::min(n, g_lanes_cap.load());
    return std::max(1, std::min(n, kLanes));
}

// The bounce buffers of all lanes are ONE piece of anonymous memory with huge pages asked for, touched by many threads and
// page-locked with one hipHostRegister (PinnedBuf): 256 MiB in a few ms where sixteen hipHostMalloc calls of 16 MiB took ~100 ms
// of the first transfer of a process -- a third of what the first denoise4D of a process paid over the second (round 6,
// profiles/r06_first_call.jsonl).  Same PCIe rate either way (profiles/r04_pin_probe.jsonl).  (io.mu held.)
static int io_init(HostIo &io, int device)
{
    if (io.ready) return TVDN_OK;
    TVDN_HIP(hipSetDevice(device));
    if (!io.pinned.p) {
        const int rc = io.pinned.alloc((size_t)kLanes * 2 * kChunk);
        if (rc) return rc;
    }
    for (int l = 0; l < kLanes; ++l) {
        Lane &L = io.lane[l];
        for (int b = 0; b < 2; ++b) {
            L.pin[b] = io.pinned.p + ((size_t)l * 2 + (size_t)b) * kChunk;
            if (!L.ev[b]) TVDN_HIP(hipEventCreateWithFlags(&L.ev[b], hipEventDisableTiming));
        }
        if (!L.stream) TVDN_HIP(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
    }
    io.device = device;
    io.ready = true;
    return TVDN_OK;
}

// The staging of `device` made ready ahead of the first transfer (tvdn_run starts this on a helper thread while it allocates
// its state: two set-up costs of a process's first call that have nothing to do with each other).  Errors are left for the
// first transfer to report.
bool io_is_warm(int device) { return device >= 0 && device < 16 && g_io[device].ready.load(); }

// (Round 6 also tried to spend the wait on a warm-up: 16 MiB uploads into a scratch block until the state was allocated -- DMA
// traffic beside the driver creating granules slowed the allocation by more than the upload gained, 0.87-0.90 s against 0.82-0.84 s
// for the whole first call -- and the lane threads copying between their bounce buffers -- no difference, 0.805-0.81 s either
// way; profiles/r06_first_call_prime.jsonl.  The first five 512 MiB chunks of a process's first upload stay at 12-40 ms each where
// every later one takes 10.5.)
void io_warm(int device)
{
    if (device < 0 || device >= 16 || g_io[device].ready.load()) return;
    DeviceRestore restore;
    std::lock_guard<std::mutex> lock(g_io[device].mu);
    if (io_init(g_io[device], device) != TVDN_OK) (void)hipGetLastError();
}

// One lane of an upload: chunks l, l+n, l+2n, ... of the array.
static void lane_upload(HostIo *io, int l, int n, char *dst, const char *src, size_t bytes, size_t chunk, int *status)
{
    Lane &L = io->lane[l];
    if (hipSetDevice(io->device) != hipSuccess) { *status = TVDN_ERR_HIP; return; }
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    int buf = 0;
    bool used[2] = {false, false};
    for (size_t c = (size_t)l; c < nchunks; c += (size_t)n, buf ^= 1) {
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        if (used[buf] && hipEventSynchronize(L.ev[buf]) != hipSuccess) { *status = TVDN_ERR_HIP; return; }
        std::memcpy(L.pin[buf], src + off, len);
        if (hipMemcpyAsync(dst + off, L.pin[buf], len, hipMemcpyHostToDevice, L.stream) != hipSuccess ||
            hipEventRecord(L.ev[buf], L.stream) != hipSuccess) { *status = TVDN_ERR_HIP; return; }
        used[buf] = true;
    }
    if (hipStreamSynchronize(L.stream) != hipSuccess) *status = TVDN_ERR_HIP;
}

// One lane of a download: the DMA of the lane's next chunk runs while it copies the current one out.
static void lane_download(HostIo *io, int l, int n, char *dst, const char *src, size_t bytes, size_t chunk, int *status)
{
    Lane &L = io->lane[l];
    if (hipSetDevice(io->device) != hipSuccess) { *status = TVDN_ERR_HIP; return; }
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    auto issue = [&](size_t c, int buf) -> bool {
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        return hipMemcpyAsync(L.pin[buf], src + off, len, hipMemcpyDeviceToHost, L.stream) == hipSuccess &&
               hipEventRecord(L.ev[buf], L.stream) == hipSuccess;
    };
    int buf = 0;
    size_t c = (size_t)l;
    if (c < nchunks && !issue(c, buf)) { *status = TVDN_ERR_HIP; return; }
    for (; c < nchunks; c += (size_t)n, buf ^= 1) {
        const size_t next = c + (size_t)n;
        if (next < nchunks && !issue(next, buf ^ 1)) { *status = TVDN_ERR_HIP; return; }
        if (hipEventSynchronize(L.ev[buf]) != hipSuccess) { *status = TVDN_ERR_HIP; return; }
        const size_t off = c * chunk, len = std::min(chunk, bytes - off);
        std::memcpy(dst + off, L.pin[buf], len);
    }
}

static int transfer(bool up, void *dst, const void *src, size_t bytes, int device)
{
    TVDN_REQUIRE(device >= 0 && device < 16, "device %d out of range", device);
    TVDN_REQUIRE(bytes == 0 || (dst && src), "NULL buffer");
    if (bytes == 0) return TVDN_OK;
    DeviceRestore restore;
    // (per device since round 6: the slabs of a device list on several GPUs go up and come home side by side, each over its own link)
    std::lock_guard<std::mutex> lock(g_io[device].mu);
    // Only copies of a few hundred bytes go through the runtime's own path for pageable memory.  For anything larger the runtime
    // PINS THE CALLER'S PAGES IN PLACE for the transfer and keeps the pinned object in a small cache keyed by address and size
    // (DmaBlitManager::hsaCopyStagedOrPinned); when the memory behind that address has meanwhile been given back and mapped anew
    // -- a NumPy array freed, the heap trimmed and grown again -- the next copy from the same address finds the stale entry and the
    // GPU reads pages that are no longer mapped for it: "Memory access fault by GPU ... on address 0x56..." and SIGABRT.  That is
    // the one native abort of round 5 (one in nine suites) and of round 6's first whole suite, both inside the 2.6 MB copies of
    // the same test (profiles/r06_abort_found.txt).  Everything of the caller's therefore crosses through the library's own
    // pinned lanes, which the runtime has no reason to look up: one lane for what fits one bounce buffer.
    if (bytes <= 512) {
        TVDN_HIP(hipSetDevice(device));
        TVDN_HIP(hipMemcpy(dst, src, bytes, up ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
        return TVDN_OK;
    }
    HostIo &io = g_io[device];
    int rc = io_init(io, device);
    if (rc) return rc;
    // Chunks of a whole bounce buffer (16 MiB) for cubes of half a GiB and more; a smaller cube is cut finer, so that every lane
    // still has four chunks to overlap its memcpy with its DMA: in 16 MiB chunks a 64 MiB cube kept four lanes busy with one
    // memcpy and one DMA each, one after the other -- 2.0 ms up and 2.8 ms down where the link needs 1.3
    // (profiles/r06_call_overhead.jsonl).  TVDN_IO_CHUNK_KIB forces a size (measurement).
    size_t chunk = kChunk;
    {
        const size_t lanes = (size_t)lanes_wanted();
        const size_t fine = (bytes / (lanes * 4) + ((size_t(1) << 20) - 1)) & ~((size_t(1) << 20) - 1);
        chunk = std::min(kChunk, std::max(size_t(1) << 20, fine));
        const char *e = getenv("TVDN_IO_CHUNK_KIB");
        if (e && atoll(e) >= 64) chunk = std::min(kChunk, (size_t)atoll(e) * 1024);
    }
    const size_t nchunks = (bytes + chunk - 1) / chunk;
    const int n = (int)std::min<size_t>((size_t)lanes_wanted(), nchunks);
    int status[kLanes] = {0};
    std::thread th[kLanes];
    for (int l = 1; l < n; ++l)
        th[l] = std::thread(up ? lane_upload : lane_download, &io, l, n, (char *)dst, (const char *)src, bytes, chunk, &status[l]);
    (up ? lane_upload : lane_download)(&io, 0, n, (char *)dst, (const char *)src, bytes, chunk, &status[0]);
    for (int l = 1; l < n; ++l) th[l].join();
    for (int l = 0; l < n; ++l)
        if (status[l]) {
            set_error("host transfer lane %d failed: %s", l, hipGetErrorString(hipGetLastError()));
            return TVDN_ERR_HIP;
        }
    return TVDN_OK;
}

}  // namespace tvdn

extern "C" int tvdn_copy_to_device(void *dst_device, const void *src_host, size_t bytes, int device)
{
    return tvdn::transfer(true, dst_device, src_host, bytes, device);
}

extern "C" int tvdn_copy_to_host(void *dst_host, const void *src_device, size_t bytes, int device)
{
    return tvdn::transfer(false, dst_host, src_device, bytes, device);
}

// ---- many equal-sized copies behind one launch (HBM <-> HBM, or HBM <-> pinned host memory) ---------------------------
// A streamed run (csrc/tvdn_stream.hip; round 2's Python engine before it) moves ~150 level windows and fills / drains ~20 staging boxes per
// chunk.  As individual hipMemcpyAsync calls those are blit kernels of the runtime that reach ~1 TB/s on 512 MiB
// (rocprofv3: 81 % of the GPU time of a pass on 256 MiB planes); batched here they are one streaming kernel per group.
namespace tvdn {

constexpr int kCopyBatch = 120;                 // pointer pairs per launch: the argument block stays below 4 KiB
constexpr long long kCopyPiece = 64 * 1024;     // bytes per workgroup

struct CopyBatch {
    int n;
    long long bytes;          // per segment, a multiple of 16
    long long pieces;         // workgroups per segment
    char *dst[kCopyBatch];
    const char *src[kCopyBatch];
};

__global__ void __launch_bounds__(256) copy_many_kernel(CopyBatch b)
{
    typedef float vec_t __attribute__((ext_vector_type(4)));
    const long long total = (long long)b.n * b.pieces;
    for (long long idx = blockIdx.x; idx < total; idx += gridDim.x) {  // one trip unless the grid is capped
        const long long seg = idx / b.pieces, piece = idx % b.pieces;
        const long long off = piece * kCopyPiece;
        const long long len = (off + kCopyPiece < b.bytes) ? kCopyPiece : b.bytes - off;
        const vec_t *s = reinterpret_cast<const vec_t *>(b.src[seg] + off);
        vec_t *d = reinterpret_cast<vec_t *>(b.dst[seg] + off);
        const long long nv = len / 16;
        long long i = threadIdx.x;
        for (; i + 3 * 256 < nv; i += 4 * 256) {  // four 16-byte loads in flight per thread
            const vec_t a0 = __builtin_nontemporal_load(s + i), a1 = __builtin_nontemporal_load(s + i + 256);
            const vec_t a2 = __builtin_nontemporal_load(s + i + 512), a3 = __builtin_nontemporal_load(s + i + 768);
            __builtin_nontemporal_store(a0, d + i);
            __builtin_nontemporal_store(a1, d + i + 256);
            __builtin_nontemporal_store(a2, d + i + 512);
            __builtin_nontemporal_store(a3, d + i + 768);
        }
        for (; i < nv; i += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
    }
}

}  // namespace tvdn

extern "C" int tvdn_copy_many(int32_t n, void *const *dst, const void *const *src, int64_t bytes_each, int32_t max_blocks,
                              void *stream)
{
    using namespace tvdn;
    TVDN_REQUIRE(n >= 0 && (n == 0 || (dst && src)), "NULL argument");
    TVDN_REQUIRE(max_blocks >= 0, "max_blocks must be >= 0");
    TVDN_REQUIRE(bytes_each >= 0 && bytes_each % 16 == 0, "bytes_each must be a non-negative multiple of 16");
    if (n == 0 || bytes_each == 0) return TVDN_OK;
    const long long pieces = (bytes_each + kCopyPiece - 1) / kCopyPiece;
    for (int i0 = 0; i0 < n; i0 += kCopyBatch) {
        CopyBatch b;
        b.n = (n - i0 < kCopyBatch) ? n - i0 : kCopyBatch;
        b.bytes = bytes_each;
        b.pieces = pieces;
        for (int i = 0; i < b.n; ++i) {
            TVDN_REQUIRE(dst[i0 + i] && src[i0 + i] && aligned16(dst[i0 + i]) && aligned16(src[i0 + i]),
                         "segment %d: NULL or not 16-byte aligned", i0 + i);
            b.dst[i] = (char *)dst[i0 + i];
            b.src[i] = (const char *)src[i0 + i];
        }
        long long grid = (long long)b.n * pieces;
        TVDN_REQUIRE(grid < (1LL << 31), "copy batch too large");
        if (max_blocks > 0 && grid > max_blocks) grid = max_blocks;
        hipLaunchKernelGGL(copy_many_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, b);
        TVDN_HIP(hipGetLastError());
    }
    return TVDN_OK;
}


// ---- measurement aid: a pure N-read / M-write 16-byte stream over caller-chosen arrays -------------------------------------
// What HBM gives a kernel that does nothing but stream `n_read` arrays in and `n_write` arrays out, one 16-byte element per
// thread, streaming (non-temporal) accesses -- the practical ceiling of the fused sweep for the SAME arrays in the SAME
// physical placement (tools/ceiling_vs_sweep.py times both side by side).  Not used by any product path.
namespace tvdn {
struct MixPtrs {
    const float4 *in[12];
    float4 *out[12];
};

template <int NR, int NW>
__global__ void __launch_bounds__(256) stream_mix_kernel(MixPtrs p, long long n16)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const long long i = xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (i >= n16) return;
    f4 v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p.in[k]) + i);
    f4 s = v[0];
#pragma unroll
    for (int k = 1; k < NR; ++k) s += v[k];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        f4 o = s;
        o.x += (float)k;
        __builtin_nontemporal_store(o, reinterpret_cast<f4 *>(p.out[k]) + i);
    }
}
// The same mix over a TILE-INTERLEAVED layout (measurement: DESIGN.md section 8, "what is left for placement"): tile t of array
// slot s lives at base + (t * (NR + NW) + s) * 4 KiB, so a workgroup's 15 tiles are one contiguous 60 KiB block and the whole
// launch is one sequential stream instead of NR + NW.  TVDN_MIX_INTERLEAVED=1 makes tvdn_stream_mix run this over in[0].
template <int NR, int NW>
__global__ void __launch_bounds__(256) stream_mix_interleaved_kernel(const float4 *base, long long tiles)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const long long t = xcd_remap(blockIdx.x, gridDim.x);
    if (t >= tiles) return;
    const f4 *blk = reinterpret_cast<const f4 *>(base) + t * (long long)(NR + NW) * 256 + threadIdx.x;
    f4 v[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) v[k] = __builtin_nontemporal_load(blk + k * 256);
    f4 s = v[0];
#pragma unroll
    for (int k = 1; k < NR; ++k) s += v[k];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
        f4 o = s;
        o.x += (float)k;
        __builtin_nontemporal_store(o, const_cast<f4 *>(blk) + (NR + k) * 256);
    }
}
// The same stream in the SWEEP's traversal: a workgroup owns one 4 KiB tile of a row-plane and marches `chunk` rows along
// axis 0 (stride = one plane), workgroup ids remapped per XCD like the sweep's -- the ceiling of that structure, apart
// from neighbour re-reads and arithmetic.
template <int NR, int NW>
__global__ void __launch_bounds__(256) stream_mix_march_kernel(MixPtrs p, long long plane16, long long rows, long long tiles, int chunk)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const long long L = xcd_remap(blockIdx.x, gridDim.x);
    const long long chunk_id = L / tiles, tile = L % tiles;
    const long long u = tile * 256 + threadIdx.x;
    if (u >= plane16) return;
    const long long m0 = chunk_id * chunk, m1 = (m0 + chunk < rows) ? m0 + chunk : rows;
    for (long long m = m0; m < m1; ++m) {
        const long long i = m * plane16 + u;
        f4 v[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p.in[k]) + i);
        f4 s = v[0];
#pragma unroll
        for (int k = 1; k < NR; ++k) s += v[k];
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            f4 o = s;
            o.x += (float)k;
            __builtin_nontemporal_store(o, reinterpret_cast<f4 *>(p.out[k]) + i);
        }
    }
}
}  // namespace tvdn

extern "C" int tvdn_stream_mix_march(int32_t n_read, const void *const *in, int32_t n_write, void *const *out, int64_t bytes_each,
                                     int64_t rows, int32_t chunk, void *stream)
{
    TVDN_REQUIRE(in && out && bytes_each > 0 && rows >= 1 && chunk >= 1 && bytes_each % (16 * rows) == 0, "bad argument");
    TVDN_REQUIRE(n_read == 10 && n_write == 5 || n_read == 6 && n_write == 5, "instantiated for 10/5 and 6/5");
    tvdn::MixPtrs p;
    std::memset(&p, 0, sizeof p);
    for (int k = 0; k < n_read; ++k) p.in[k] = (const float4 *)in[k];
    for (int k = 0; k < n_write; ++k) p.out[k] = (float4 *)out[k];
    const long long plane16 = bytes_each / 16 / rows, tiles = (plane16 + 255) / 256;
    const long long grid = tiles * ((rows + chunk - 1) / chunk);
    TVDN_REQUIRE(grid < (1LL << 31), "grid too large");
    if (n_read == 10)
        hipLaunchKernelGGL((tvdn::stream_mix_march_kernel<10, 5>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, plane16, (long long)rows, tiles, chunk);
    else
        hipLaunchKernelGGL((tvdn::stream_mix_march_kernel<6, 5>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, plane16, (long long)rows, tiles, chunk);
    TVDN_HIP(hipGetLastError());
    return TVDN_OK;
}

extern "C" int tvdn_stream_mix(int32_t n_read, const void *const *in, int32_t n_write, void *const *out, int64_t bytes_each,
                               void *stream)
{
    TVDN_REQUIRE(in && out && bytes_each > 0 && bytes_each % 16 == 0, "bad argument");
    tvdn::MixPtrs p;
    std::memset(&p, 0, sizeof p);
    TVDN_REQUIRE(n_read >= 1 && n_read <= 12 && n_write >= 1 && n_write <= 12, "1..12 arrays each way");
    for (int k = 0; k < n_read; ++k) {
        TVDN_REQUIRE(in[k] && tvdn::aligned16(in[k]), "in[%d] NULL or unaligned", k);
        p.in[k] = (const float4 *)in[k];
    }
    for (int k = 0; k < n_write; ++k) {
        TVDN_REQUIRE(out[k] && tvdn::aligned16(out[k]), "out[%d] NULL or unaligned", k);
        p.out[k] = (float4 *)out[k];
    }
    const long long n16 = bytes_each / 16;
    const long long grid = (n16 + 255) / 256;
    TVDN_REQUIRE(grid < (1LL << 31), "array too long for one launch");
    if (getenv("TVDN_MIX_INTERLEAVED") && n_read == 10 && n_write == 5) {
        // measurement knob: in[0] is the base of (n_read + n_write) * bytes_each bytes, read and written tile-interleaved
        hipLaunchKernelGGL((tvdn::stream_mix_interleaved_kernel<10, 5>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream,
                           (const float4 *)in[0], grid);
        TVDN_HIP(hipGetLastError());
        return TVDN_OK;
    }
#define TVDN_MIX(NR, NW)                                                                                                           \
    if (n_read == NR && n_write == NW) {                                                                                           \
        hipLaunchKernelGGL((tvdn::stream_mix_kernel<NR, NW>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, p, n16); \
        TVDN_HIP(hipGetLastError());                                                                                               \
        return TVDN_OK;                                                                                                            \
    }
    TVDN_MIX(10, 5)  // 4-D FISTA, compact state: orig, recon, 4 x (d_k, d_k-1) in; recon, 4 x d_k+1 out
    TVDN_MIX(6, 5)   // 4-D unaccelerated
    TVDN_MIX(8, 4)   // 3-D FISTA, compact state
    TVDN_MIX(5, 4)   // 3-D unaccelerated
    TVDN_MIX(1, 1)   // plain copy
    TVDN_MIX(9, 4)   // 4-D FISTA without a stored recon (rebuilt from orig and the d pairs): orig, 4 x (d_k, d_k-1) in; 4 x d_k+1 out
#undef TVDN_MIX
    tvdn::set_error("stream mix %d in / %d out is not instantiated (10/5, 6/5, 8/4, 5/4, 1/1, 9/4)", n_read, n_write);
    return TVDN_ERR_UNSUPPORTED;
}
