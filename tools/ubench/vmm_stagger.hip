// Measurement aid (not product code): the config-2 state on 1 GiB granules (each one physically contiguous, so the low 30 address
// bits of every array are known up to the granule's base), swept with different STAGGERS between consecutive arrays.  On such
// memory the relation between the fifteen streams' low address bits is exactly what the stagger says: if the sweep's speed is
// decided by how those bits collide (cache sets, channels, banks), the best stagger is a property of the chip, not of the draw.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/vmm_stagger.hip -Iinclude -Lcytvdn_amd -ltvdn_hip -Wl,-rpath,'$ORIGIN/../../cytvdn_amd' -o tools/ubench/vmm_stagger
//   tools/ubench/vmm_stagger [GRANULE_MiB=1024] [PASSES=2]
#include "vmm_common.hpp"

int main(int argc, char **argv)
{
    const size_t g_mib = argc > 1 ? (size_t)atoll(argv[1]) : 1024;
    const int passes = argc > 2 ? atoi(argv[2]) : 2;
    CK(hipSetDevice(0));
    const double t_start = now_s();
    State st;
    st.init();
    // control: one hipMalloc state with the product's stagger, timed before and after every pass (what changes with time)
    void *ctl_p = nullptr;
    CK(hipMalloc(&ctl_p, st.total));
    State ctl = st;
    ctl.bind((char *)ctl_p);
    ctl.fill();
    auto sweep_ms = [&](State &s) {
        float ms[kSlices];
        double f = 0.0;
        s.iterate(1, nullptr);
        for (int r = 0; r < 4; ++r) {
            s.iterate(1, ms);
            f += ms[0];
        }
        return f / 4.0;
    };
    const size_t K = 1024, M = 1024 * 1024;
    const std::vector<size_t> skews = {0,       256,     512,      1 * K,   2 * K,        4 * K,        8 * K,         16 * K,        32 * K,      64 * K,
                                       128 * K, 256 * K, 512 * K,  1 * M,   2 * M,        4 * M,        8 * M,         16 * M,        32 * M,      64 * M,
                                       128 * M, 4 * K + 256, 68 * K, 260 * K, 1 * M + 4 * K, 2 * M + 4 * K, 16 * M + 4 * K, 64 * M + 4 * K, 17 * M, 65 * M + 68 * K,
                                       12 * K,  20 * K,  36 * K,   1 * M + 68 * K, 3 * M, 5 * M, 9 * M, 33 * M, 129 * M, 273 * M + 4 * K};
    size_t max_skew = 0;
    for (size_t s : skews) max_skew = std::max(max_skew, s);
    State big = st;
    big.set_skew(max_skew);
    Arena ar;
    ar.create(g_mib << 20, big.total, 0);
    std::vector<int> map((size_t)ar.slots);
    std::iota(map.begin(), map.end(), 0);
    ar.remap(map);
    printf("{\"control_hipMalloc_ms\": %.4f, \"t\": %.2f}\n", sweep_ms(ctl), now_s() - t_start);
    for (int pass = 0; pass < passes; ++pass) {
        for (size_t sk : skews) {
            st.set_skew(sk);
            st.bind(ar.va);
            st.fill();
            const double v = sweep_ms(st);
            printf("{\"pass\": %d, \"skew\": %zu, \"full_ms\": %.4f, \"t\": %.2f}\n", pass, sk, v, now_s() - t_start);
            fflush(stdout);
        }
        printf("{\"control_hipMalloc_ms\": %.4f, \"t\": %.2f}\n", sweep_ms(ctl), now_s() - t_start);
    }
    CK(hipDeviceSynchronize());
    printf("{\"done\": true}\n");
    return 0;
}
