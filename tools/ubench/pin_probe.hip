// What page-locking host memory costs on this box, and whether it parallelises: the set-up of a streamed tvdn_run
// (csrc/tvdn_stream.hip) is dominated by it (7-15 s for 160 GiB, profiles/r03_outofcore_depth.jsonl).
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/pin_probe.hip -o tools/ubench/pin_probe -lpthread
//   tools/ubench/pin_probe [GiB per array, default 8]
// Prints one JSON object per experiment.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include <sys/mman.h>

static double now()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));               \
            (void)hipGetLastError();                                                      \
        }                                                                                 \
    } while (0)

static void par(int n, const std::function<void(int)> &f)
{
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i) th.emplace_back(f, i);
    for (auto &t : th) t.join();
}

static double h2d_GBps(void *dev, const void *host, size_t bytes)
{
    CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    const double t0 = now();
    CK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    return bytes / (now() - t0) / 1e9;
}

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
    const size_t bytes = gib << 30;
    const int T = 8;
    CK(hipSetDevice(0));
    void *dev = nullptr;
    CK(hipMalloc(&dev, size_t(1) << 30));

    // 1. ordinary memory, touched by T threads
    double t0 = now();
    char *p = (char *)malloc(bytes);
    par(T, [&](int i) { memset(p + bytes / T * i, 1, bytes / T); });
    printf("{\"what\": \"malloc + first touch, %d threads\", \"GiB\": %zu, \"s\": %.3f}\n", T, gib, now() - t0);

    // 2. register whole / unregister
    t0 = now();
    CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    double t_reg = now() - t0;
    const double bw_reg = h2d_GBps(dev, p, size_t(1) << 30);
    t0 = now();
    CK(hipHostUnregister(p));
    printf("{\"what\": \"hipHostRegister of touched memory, one call\", \"GiB\": %zu, \"s\": %.3f, \"unregister_s\": %.3f, \"h2d_GBps\": %.1f}\n", gib,
           t_reg, now() - t0, bw_reg);

    // 3. register in T slices from T threads
    t0 = now();
    par(T, [&](int i) { CK(hipHostRegister(p + bytes / T * i, bytes / T, hipHostRegisterDefault)); });
    t_reg = now() - t0;
    t0 = now();
    par(T, [&](int i) { CK(hipHostUnregister(p + bytes / T * i)); });
    printf("{\"what\": \"hipHostRegister in %d slices from %d threads\", \"GiB\": %zu, \"s\": %.3f, \"unregister_s\": %.3f}\n", T, T, gib, t_reg,
           now() - t0);

    // 3b. register in slices, one thread, one after the other (is the cost per call or per byte?)
    t0 = now();
    for (int i = 0; i < T; ++i) CK(hipHostRegister(p + bytes / T * i, bytes / T, hipHostRegisterDefault));
    t_reg = now() - t0;
    for (int i = 0; i < T; ++i) CK(hipHostUnregister(p + bytes / T * i));
    printf("{\"what\": \"hipHostRegister in %d slices, one thread\", \"GiB\": %zu, \"s\": %.3f}\n", T, gib, t_reg);
    free(p);

    // 4. register UNTOUCHED memory (np.empty: the pages do not exist yet)
    p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    t0 = now();
    CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    t_reg = now() - t0;
    CK(hipHostUnregister(p));
    munmap(p, bytes);
    printf("{\"what\": \"hipHostRegister of untouched memory, one call\", \"GiB\": %zu, \"s\": %.3f}\n", gib, t_reg);
    p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    t0 = now();
    par(T, [&](int i) { CK(hipHostRegister(p + bytes / T * i, bytes / T, hipHostRegisterDefault)); });
    t_reg = now() - t0;
    par(T, [&](int i) { CK(hipHostUnregister(p + bytes / T * i)); });
    munmap(p, bytes);
    printf("{\"what\": \"hipHostRegister of untouched memory, %d slices from %d threads\", \"GiB\": %zu, \"s\": %.3f}\n", T, T, gib, t_reg);

    // 5. hipHostMalloc: one call; then a fill by T threads; free
    void *q = nullptr;
    t0 = now();
    CK(hipHostMalloc(&q, bytes, hipHostMallocDefault));
    double t_alloc = now() - t0;
    t0 = now();
    par(T, [&](int i) { memset((char *)q + bytes / T * i, 0, bytes / T); });
    double t_fill = now() - t0;
    const double bw_hm = h2d_GBps(dev, q, size_t(1) << 30);
    t0 = now();
    CK(hipHostFree(q));
    printf("{\"what\": \"hipHostMalloc, one call\", \"GiB\": %zu, \"s\": %.3f, \"fill_s\": %.3f, \"free_s\": %.3f, \"h2d_GBps\": %.1f}\n", gib, t_alloc,
           t_fill, now() - t0, bw_hm);

    // 6. T hipHostMalloc calls of 1/T each from T threads
    std::vector<void *> qs(T, nullptr);
    t0 = now();
    par(T, [&](int i) { CK(hipHostMalloc(&qs[i], bytes / T, hipHostMallocDefault)); });
    t_alloc = now() - t0;
    t0 = now();
    par(T, [&](int i) { CK(hipHostFree(qs[i])); });
    printf("{\"what\": \"hipHostMalloc, %d calls from %d threads\", \"GiB\": %zu, \"s\": %.3f, \"free_s\": %.3f}\n", T, T, gib, t_alloc, now() - t0);

    // 7. the same one after the other
    t0 = now();
    for (int i = 0; i < T; ++i) CK(hipHostMalloc(&qs[i], bytes / T, hipHostMallocDefault));
    t_alloc = now() - t0;
    for (int i = 0; i < T; ++i) CK(hipHostFree(qs[i]));
    printf("{\"what\": \"hipHostMalloc, %d calls one after the other\", \"GiB\": %zu, \"s\": %.3f}\n", T, gib, t_alloc);

    // 8. hipHostMalloc with the NUMA-agnostic / non-coherent flags
    for (unsigned flags : {(unsigned)hipHostMallocNonCoherent, (unsigned)hipHostMallocNumaUser}) {
        t0 = now();
        if (hipHostMalloc(&q, bytes, flags) != hipSuccess) {
            (void)hipGetLastError();
            printf("{\"what\": \"hipHostMalloc flags 0x%x\", \"error\": true}\n", flags);
            continue;
        }
        t_alloc = now() - t0;
        const double bw = h2d_GBps(dev, q, size_t(1) << 30);
        CK(hipHostFree(q));
        printf("{\"what\": \"hipHostMalloc flags 0x%x\", \"GiB\": %zu, \"s\": %.3f, \"h2d_GBps\": %.1f}\n", flags, gib, t_alloc, bw);
    }
    // 9. the alternative to hipHostMalloc: anonymous memory (huge pages asked for), first touch by T2 threads, one registration
    for (int huge = 0; huge < 2; ++huge)
        for (int T2 : {8, 16, 32}) {
            t0 = now();
            p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (huge) madvise(p, bytes, MADV_HUGEPAGE);
            par(T2, [&](int i) {
                char *q0 = p + bytes / T2 * i;
                for (size_t o = 0; o < bytes / T2; o += 4096) q0[o] = 0;
            });
            const double t_touch = now() - t0;
            t0 = now();
            CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
            t_reg = now() - t0;
            const double bw = h2d_GBps(dev, p, size_t(1) << 30);
            t0 = now();
            CK(hipHostUnregister(p));
            munmap(p, bytes);
            printf("{\"what\": \"mmap%s + touch by %d threads + hipHostRegister\", \"GiB\": %zu, \"touch_s\": %.3f, \"register_s\": %.3f, \"release_s\": %.3f, \"h2d_GBps\": %.1f}\n",
                   huge ? " + MADV_HUGEPAGE" : "", T2, gib, t_touch, t_reg, now() - t0, bw);
        }
    // 10. MAP_POPULATE
    t0 = now();
    p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_POPULATE, -1, 0);
    const double t_pop = now() - t0;
    t0 = now();
    CK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    t_reg = now() - t0;
    CK(hipHostUnregister(p));
    munmap(p, bytes);
    printf("{\"what\": \"mmap MAP_POPULATE + hipHostRegister\", \"GiB\": %zu, \"populate_s\": %.3f, \"register_s\": %.3f}\n", gib, t_pop, t_reg);
    // 11. hipHostMalloc in 2 GiB blocks, each timed
    {
        std::vector<void *> bl;
        for (size_t done_b = 0; done_b < bytes; done_b += size_t(2) << 30) {
            t0 = now();
            void *b2 = nullptr;
            CK(hipHostMalloc(&b2, size_t(2) << 30, hipHostMallocDefault));
            printf("{\"what\": \"hipHostMalloc 2 GiB block\", \"n\": %zu, \"s\": %.3f}\n", bl.size(), now() - t0);
            bl.push_back(b2);
        }
        for (void *b2 : bl) CK(hipHostFree(b2));
    }
    CK(hipFree(dev));
    return 0;
}
