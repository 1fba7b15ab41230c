// What the PCIe link of an MI355X box gives in each direction, alone and both at once, by mechanism: the runtime's
// hipMemcpyAsync (SDMA engines / blit kernels) against a copy KERNEL that reads or writes page-locked host memory directly,
// with and without an HBM-bound kernel running beside them (the sweeps of a streamed tvdn_run, csrc/tvdn_stream.hip).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/pcie_duplex.hip -o tools/ubench/pcie_duplex -lpthread
// One JSON object per experiment.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include <sys/mman.h>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));    \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

typedef float vec_t __attribute__((ext_vector_type(4)));

// grid-stride copy of n 16-byte elements
__global__ void __launch_bounds__(256) copy_kernel(vec_t *__restrict__ d, const vec_t *__restrict__ s, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n; i += stride) {
        vec_t a0 = __builtin_nontemporal_load(s + i), a1, a2, a3;
        const bool b1 = i + 256 < n, b2 = i + 512 < n, b3 = i + 768 < n;
        if (b1) a1 = __builtin_nontemporal_load(s + i + 256);
        if (b2) a2 = __builtin_nontemporal_load(s + i + 512);
        if (b3) a3 = __builtin_nontemporal_load(s + i + 768);
        __builtin_nontemporal_store(a0, d + i);
        if (b1) __builtin_nontemporal_store(a1, d + i + 256);
        if (b2) __builtin_nontemporal_store(a2, d + i + 512);
        if (b3) __builtin_nontemporal_store(a3, d + i + 768);
    }
}

// HBM-bound background work: x = x * 1.0001 over a big buffer
__global__ void __launch_bounds__(256) scale_kernel(vec_t *x, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = x[i] * 1.0001f;
}

static char *pinned(size_t bytes)
{
    char *p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    madvise(p, bytes, MADV_HUGEPAGE);
    std::vector<std::thread> th;
    for (int t = 0; t < 16; ++t)
        th.emplace_back([=] {
            for (size_t o = bytes / 16 * t; o < bytes / 16 * (t + 1); o += 4096) p[o] = 1;
        });
    for (auto &t : th) t.join();
    CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return p;
}

int main(int argc, char **argv)
{
    const size_t piece = size_t(512) << 20;  // one transfer
    const int n_piece = argc > 1 ? atoi(argv[1]) : 12;
    CK(hipSetDevice(0));
    char *h_up = pinned(piece * 2), *h_dn = pinned(piece * 2);
    char *d_up, *d_dn;
    vec_t *big;
    const size_t big_n = (size_t(4) << 30) / 16;
    CK(hipMalloc(&d_up, piece * 2));
    CK(hipMalloc(&d_dn, piece * 2));
    CK(hipMalloc(&big, big_n * 16));
    CK(hipMemset(d_dn, 1, piece * 2));
    CK(hipMemset(big, 0, big_n * 16));
    hipStream_t s_up, s_dn, s_bg;
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithPriority(&s_up, hipStreamNonBlocking, 0));
    CK(hipStreamCreateWithPriority(&s_dn, hipStreamNonBlocking, least));
    CK(hipStreamCreateWithPriority(&s_bg, hipStreamNonBlocking, greatest));

    // mode: 0 none, 1 hipMemcpyAsync, >1 copy kernel with that many workgroups
    auto issue = [&](bool up, int mode, int i) {
        const size_t off = (size_t)(i & 1) * piece;
        if (mode == 0) return;
        if (up) {
            if (mode == 1)
                CK(hipMemcpyAsync(d_up + off, h_up + off, piece, hipMemcpyHostToDevice, s_up));
            else
                hipLaunchKernelGGL(copy_kernel, dim3(mode), dim3(256), 0, s_up, (vec_t *)(d_up + off), (const vec_t *)(h_up + off), piece / 16);
        } else {
            if (mode == 1)
                CK(hipMemcpyAsync(h_dn + off, d_dn + off, piece, hipMemcpyDeviceToHost, s_dn));
            else
                hipLaunchKernelGGL(copy_kernel, dim3(mode), dim3(256), 0, s_dn, (vec_t *)(h_dn + off), (const vec_t *)(d_dn + off), piece / 16);
        }
    };
    auto experiment = [&](const char *what, int up_mode, int dn_mode, bool background) {
        CK(hipDeviceSynchronize());
        hipEvent_t u0, u1, d0, d1, b0, b1;
        for (hipEvent_t *e : {&u0, &u1, &d0, &d1, &b0, &b1}) CK(hipEventCreate(e));
        int n_bg = 0;
        CK(hipEventRecord(u0, s_up));
        CK(hipEventRecord(d0, s_dn));
        CK(hipEventRecord(b0, s_bg));
        const double t0 = now();
        for (int i = 0; i < n_piece; ++i) {
            issue(true, up_mode, i);
            issue(false, dn_mode, i);
            if (background)
                for (int j = 0; j < 6; ++j, ++n_bg) hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((big_n + 255) / 256)), dim3(256), 0, s_bg, big, big_n);
        }
        CK(hipEventRecord(u1, s_up));
        CK(hipEventRecord(d1, s_dn));
        CK(hipEventRecord(b1, s_bg));
        CK(hipDeviceSynchronize());
        const double wall = now() - t0;
        float tu = 0, td = 0, tb = 0;
        CK(hipEventElapsedTime(&tu, u0, u1));
        CK(hipEventElapsedTime(&td, d0, d1));
        CK(hipEventElapsedTime(&tb, b0, b1));
        const double gb = (double)piece * n_piece / 1e9;
        printf("{\"what\": \"%s\", \"up\": %d, \"down\": %d, \"background\": %s, \"h2d_GBps\": %.1f, \"d2h_GBps\": %.1f, \"wall_s\": %.3f", what, up_mode,
               dn_mode, background ? "true" : "false", up_mode ? gb / (tu * 1e-3) : 0.0, dn_mode ? gb / (td * 1e-3) : 0.0, wall);
        if (background) printf(", \"background_TBps\": %.2f", (double)n_bg * big_n * 32 / (tb * 1e-3) / 1e12);
        printf("}\n");
        fflush(stdout);
        for (hipEvent_t e : {u0, u1, d0, d1, b0, b1}) CK(hipEventDestroy(e));
    };
    // The streamed engine's pattern: per chunk, the compute stream waits for the upload (in_ready), records in_free which the
    // NEXT upload waits for; the download waits for the compute stream (out_ready) and records out_free which compute waits for.
    // coupling bits: 1 = downloads wait on compute events, 2 = compute waits on download events, 4 = the same two for uploads,
    // 8 = nine 1/9-size copies per chunk instead of one
    auto coupled = [&](const char *what, int coupling, int n_chunk, int kernels_per_chunk, bool up, bool down) {
        CK(hipDeviceSynchronize());
        std::vector<hipEvent_t> evs;
        auto ev = [&]() {
            hipEvent_t e;
            CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            evs.push_back(e);
            return e;
        };
        hipEvent_t u0, u1, d0, d1, b0, b1;
        for (hipEvent_t *e : {&u0, &u1, &d0, &d1, &b0, &b1}) CK(hipEventCreate(e));
        CK(hipEventRecord(u0, s_up));
        CK(hipEventRecord(d0, s_dn));
        CK(hipEventRecord(b0, s_bg));
        hipEvent_t in_free[2] = {nullptr, nullptr}, out_free[2] = {nullptr, nullptr};
        const int parts = (coupling & 8) ? 9 : 1;
        const size_t part = piece / parts / 16 * 16;
        const double t0 = now();
        for (int c = 0; c < n_chunk; ++c) {
            const int h = c & 1;
            const size_t off = (size_t)h * piece;
            hipEvent_t in_ready = nullptr;
            if (up) {
                if ((coupling & 4) && in_free[h]) CK(hipStreamWaitEvent(s_up, in_free[h], 0));
                for (int k = 0; k < parts; ++k) CK(hipMemcpyAsync(d_up + off + k * part, h_up + off + k * part, part, hipMemcpyHostToDevice, s_up));
                if (coupling & 4) {
                    in_ready = ev();
                    CK(hipEventRecord(in_ready, s_up));
                    CK(hipStreamWaitEvent(s_bg, in_ready, 0));
                }
            }
            if ((coupling & 2) && out_free[h]) CK(hipStreamWaitEvent(s_bg, out_free[h], 0));
            for (int j = 0; j < kernels_per_chunk; ++j) hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((big_n + 255) / 256)), dim3(256), 0, s_bg, big, big_n);
            if (up && (coupling & 4)) {
                in_free[h] = ev();
                CK(hipEventRecord(in_free[h], s_bg));
            }
            if (down) {
                if (coupling & 1) {
                    hipEvent_t out_ready = ev();
                    CK(hipEventRecord(out_ready, s_bg));
                    CK(hipStreamWaitEvent(s_dn, out_ready, 0));
                }
                for (int k = 0; k < parts; ++k) CK(hipMemcpyAsync(h_dn + off + k * part, d_dn + off + k * part, part, hipMemcpyDeviceToHost, s_dn));
                if (coupling & 2) {
                    out_free[h] = ev();
                    CK(hipEventRecord(out_free[h], s_dn));
                }
            }
        }
        CK(hipEventRecord(u1, s_up));
        CK(hipEventRecord(d1, s_dn));
        CK(hipEventRecord(b1, s_bg));
        CK(hipDeviceSynchronize());
        const double wall = now() - t0;
        float tu = 0, td = 0, tb = 0;
        CK(hipEventElapsedTime(&tu, u0, u1));
        CK(hipEventElapsedTime(&td, d0, d1));
        CK(hipEventElapsedTime(&tb, b0, b1));
        const double gb = (double)part * parts * n_chunk / 1e9;
        printf("{\"what\": \"%s\", \"coupling\": %d, \"kernels_per_chunk\": %d, \"h2d_GBps\": %.1f, \"d2h_GBps\": %.1f, \"wall_s\": %.3f, \"compute_TBps\": %.2f}\n", what,
               coupling, kernels_per_chunk, up ? gb / (tu * 1e-3) : 0.0, down ? gb / (td * 1e-3) : 0.0, wall,
               (double)n_chunk * kernels_per_chunk * big_n * 32 / (tb * 1e-3) / 1e12);
        fflush(stdout);
        for (hipEvent_t e : evs) CK(hipEventDestroy(e));
        for (hipEvent_t e : {u0, u1, d0, d1, b0, b1}) CK(hipEventDestroy(e));
    };
    if (argc > 2 && !strcmp(argv[2], "engine")) {
        // The streamed engine itself, stripped to its transfers: NU host arrays go up and ND come down per chunk, 512 MiB each,
        // through double-buffered boxes, with the engine's four events per chunk and `kern` HBM-bound kernels per chunk beside
        // them; the downloads land in the SAME host arrays `lag` chunks behind the uploads (in-place host state) or in others.
        const int NU = 10, ND = 9, NC = argc > 3 ? atoi(argv[3]) : 12;
        std::vector<char *> hs, hs2;
        for (int i = 0; i < NU; ++i) hs.push_back(pinned(piece * NC));
        for (int i = 0; i < ND; ++i) hs2.push_back(pinned(piece * NC));
        char *in_box[2], *out_box[2];
        for (int h = 0; h < 2; ++h) {
            CK(hipMalloc(&in_box[h], piece * NU));
            CK(hipMalloc(&out_box[h], piece * ND));
        }
        auto engine = [&](const char *what, bool up, bool down, bool inplace, int kern, int lag, bool events) {
            CK(hipDeviceSynchronize());
            std::vector<hipEvent_t> evs;
            auto ev = [&]() {
                hipEvent_t e;
                CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                evs.push_back(e);
                return e;
            };
            hipEvent_t in_free[2] = {nullptr, nullptr}, out_free[2] = {nullptr, nullptr};
            const double t0 = now();
            for (int c = 0; c < NC + lag; ++c) {
                const int h = c & 1;
                if (up && c < NC) {
                    if (events && in_free[h]) CK(hipStreamWaitEvent(s_up, in_free[h], 0));
                    for (int i = 0; i < NU; ++i) CK(hipMemcpyAsync(in_box[h] + (size_t)i * piece, hs[i] + (size_t)c * piece, piece, hipMemcpyHostToDevice, s_up));
                    if (events) {
                        hipEvent_t r = ev();
                        CK(hipEventRecord(r, s_up));
                        CK(hipStreamWaitEvent(s_bg, r, 0));
                    }
                }
                if (events && down && out_free[h]) CK(hipStreamWaitEvent(s_bg, out_free[h], 0));
                for (int j = 0; j < kern; ++j) hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((big_n + 255) / 256)), dim3(256), 0, s_bg, big, big_n);
                if (events && up && c < NC) {
                    in_free[h] = ev();
                    CK(hipEventRecord(in_free[h], s_bg));
                }
                if (down && c >= lag) {
                    if (events) {
                        hipEvent_t r = ev();
                        CK(hipEventRecord(r, s_bg));
                        CK(hipStreamWaitEvent(s_dn, r, 0));
                    }
                    for (int i = 0; i < ND; ++i)
                        CK(hipMemcpyAsync((inplace ? hs[i + 1] : hs2[i]) + (size_t)(c - lag) * piece, out_box[h] + (size_t)i * piece, piece, hipMemcpyDeviceToHost, s_dn));
                    if (events) {
                        out_free[h] = ev();
                        CK(hipEventRecord(out_free[h], s_dn));
                    }
                }
            }
            CK(hipDeviceSynchronize());
            const double wall = now() - t0;
            printf("{\"what\": \"%s\", \"up\": %s, \"down\": %s, \"in_place\": %s, \"kernels_per_chunk\": %d, \"lag_chunks\": %d, \"events\": %s, \"chunks\": %d, "
                   "\"wall_s\": %.3f, \"h2d_GB\": %.1f, \"d2h_GB\": %.1f, \"total_GBps\": %.1f}\n",
                   what, up ? "true" : "false", down ? "true" : "false", inplace ? "true" : "false", kern, lag, events ? "true" : "false", NC, wall,
                   up ? (double)piece * NU * NC / 1e9 : 0.0, down ? (double)piece * ND * NC / 1e9 : 0.0,
                   ((up ? (double)piece * NU * NC : 0.0) + (down ? (double)piece * ND * NC : 0.0)) / wall / 1e9);
            fflush(stdout);
            for (hipEvent_t e : evs) CK(hipEventDestroy(e));
        };
        for (int kern : {0, 38}) {
            engine("engine: up only", true, false, true, kern, 0, true);
            engine("engine: down only", false, true, true, kern, 0, true);
            engine("engine: both, in place, lag 19", true, true, true, kern, 19, true);
            engine("engine: both, in place, lag 2", true, true, true, kern, 2, true);
            engine("engine: both, separate arrays, lag 2", true, true, false, kern, 2, true);
            engine("engine: both, in place, lag 2, no events", true, true, true, kern, 2, false);
        }
        return 0;
    }
    if (argc > 2) {  // only the coupled experiments
        for (int kpc : {2, 8}) {
            for (int coupling : {0, 1, 2, 3, 4, 7, 15}) {
                coupled("coupled both", coupling, n_piece, kpc, true, true);
                coupled("coupled down only", coupling, n_piece, kpc, false, true);
                coupled("coupled up only", coupling, n_piece, kpc, true, false);
            }
        }
        return 0;
    }
    for (int bg = 0; bg < 2; ++bg) {
        experiment("background alone", 0, 0, bg);
        experiment("memcpy up alone", 1, 0, bg);
        experiment("memcpy down alone", 0, 1, bg);
        experiment("memcpy both", 1, 1, bg);
        for (int g : {4, 8, 16, 32, 64, 128}) {
            experiment("kernel down alone", 0, g, bg);
            experiment("kernel up alone", g, 0, bg);
        }
        for (int g : {8, 16, 32, 64}) {
            experiment("memcpy up + kernel down", 1, g, bg);
            experiment("kernel up + memcpy down", g, 1, bg);
            experiment("kernel both", g, g, bg);
        }
    }
    return 0;
}
