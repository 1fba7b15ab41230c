// Measurement aid (not product code): does hipMemCreate of 1 GiB granules go faster from several host threads at once?
// (The first tvdn_run of a process spends 0.15-0.30 s creating its state's granules and the pool around them one after the other:
// csrc/tvdn_devmem.hip draw_granules)  Per thread count: N granules created, wall time, then released.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/vmm_create_parallel.hip -o tools/ubench/vmm_create_parallel -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 96;
    const size_t G = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
    if (hipSetDevice(0) != hipSuccess) return 1;
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    for (int rep = 0; rep < 2; ++rep)
        for (int T : {1, 2, 4, 8, 16}) {
            std::vector<hipMemGenericAllocationHandle_t> h((size_t)N);
            std::vector<int> ok((size_t)N, 0);
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    (void)hipSetDevice(0);
                    for (int i = t; i < N; i += T) ok[(size_t)i] = hipMemCreate(&h[(size_t)i], G, &prop, 0) == hipSuccess;
                });
            for (auto &x : th) x.join();
            const double t1 = now();
            int n_ok = 0;
            for (int i = 0; i < N; ++i) n_ok += ok[(size_t)i];
            for (int i = 0; i < N; ++i) if (ok[(size_t)i]) (void)hipMemRelease(h[(size_t)i]);
            const double t2 = now();
            printf("{\"rep\": %d, \"threads\": %d, \"granules\": %d, \"granule_mib\": %zu, \"created\": %d, \"create_s\": %.4f, \"ms_per_granule\": %.3f, \"release_s\": %.4f}\n", rep, T, N, G >> 20, n_ok,
                   t1 - t0, 1e3 * (t1 - t0) / N, t2 - t1);
            fflush(stdout);
            // let the driver take the released memory back before the next round
            for (int k = 0; k < 400; ++k) { size_t f = 0, tt = 0; (void)hipMemGetInfo(&f, &tt); if (f > tt - ((size_t)8 << 30)) break; std::this_thread::sleep_for(std::chrono::milliseconds(10)); }
        }
    return 0;
}
