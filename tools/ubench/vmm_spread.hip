// Measurement aid (not product code): how far over the HBM must the granules of a state be spread for the sweep to run fast?
// All free HBM as 1 GiB granules in creation order; for pools = the first P granules (P = 61 ... all), the config-2 state on
// DRAWS random subsets of the pool in random order, and on the pool's evenly strided subset in order and shuffled.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/vmm_spread.hip -Iinclude -Lcytvdn_amd -ltvdn_hip -Wl,-rpath,'$ORIGIN/../../cytvdn_amd' -o tools/ubench/vmm_spread
//   tools/ubench/vmm_spread [GRANULE_MiB=1024] [DRAWS=6] [LEAVE_GiB=8]
#include "vmm_common.hpp"

int main(int argc, char **argv)
{
    const size_t g_mib = argc > 1 ? (size_t)atoll(argv[1]) : 1024;
    const int draws = argc > 2 ? atoi(argv[2]) : 6;
    const double leave_gib = argc > 3 ? atof(argv[3]) : 8.0;
    CK(hipSetDevice(0));
    const double t_start = now_s();
    State st;
    st.init();
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    const size_t G = g_mib << 20;
    const int slots = (int)((st.total + G - 1) / G);
    const int H = (int)((free_b - (size_t)(leave_gib * 1073741824.0)) / G);
    Arena ar;
    ar.create(G, st.total, H - slots);
    std::vector<int> map((size_t)slots);
    std::mt19937 rng(2026);
    auto timed = [&](const char *what, int pool, int draw) {
        const double tr = ar.remap(map);
        st.bind(ar.va);
        st.fill();
        double full, sl[kSlices];
        st.measure(3, &full, sl);
        printf("{\"set\": \"%s\", \"pool\": %d, \"draw\": %d, \"full_ms\": %.4f, \"slice_ms\": [%.4f,%.4f,%.4f,%.4f], \"remap_s\": %.4f, \"t\": %.2f}\n", what, pool, draw, full, sl[0], sl[1],
               sl[2], sl[3], tr, now_s() - t_start);
        fflush(stdout);
    };
    std::vector<int> pools;
    for (double f : {1.0, 1.25, 1.5, 2.0, 3.0, 4.0}) {
        const int P = std::min(H, (int)(slots * f + 0.5));
        if (pools.empty() || pools.back() != P) pools.push_back(P);
    }
    if (pools.back() != H) pools.push_back(H);
    for (int P : pools) {
        // in creation order (a contiguous window), evenly strided over the pool in order, strided + shuffled, random draws
        for (int s = 0; s < slots; ++s) map[(size_t)s] = (int)((long long)s * P / slots);
        timed("strided, in order", P, 0);
        std::vector<int> sh = map;
        for (int d = 0; d < 3; ++d) {
            std::shuffle(sh.begin(), sh.end(), rng);
            map = sh;
            timed("strided, shuffled", P, d);
        }
        std::vector<int> all((size_t)P);
        std::iota(all.begin(), all.end(), 0);
        for (int d = 0; d < draws; ++d) {
            std::shuffle(all.begin(), all.end(), rng);
            for (int s = 0; s < slots; ++s) map[(size_t)s] = all[(size_t)s];
            timed("random", P, d);
        }
    }
    CK(hipDeviceSynchronize());
    printf("{\"done\": true}\n");
    return 0;
}
