// Measurement aid (not product code): one pool of granules, the config-2 state on it alternately in CREATION ORDER (what a
// contiguous allocation looks like) and SHUFFLED (the arrays' physical distances randomised), beside a hipMalloc state as control.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/vmm_shuffle_ab.hip -Iinclude -Lcytvdn_amd -ltvdn_hip -Wl,-rpath,'$ORIGIN/../../cytvdn_amd' -o tools/ubench/vmm_shuffle_ab
//   tools/ubench/vmm_shuffle_ab [GRANULE_MiB=64] [ROUNDS=8] [PRE_MALLOC=1] [SPARE_GRANULES=0]
// PRE_MALLOC: hipMalloc states allocated BEFORE the pool (the first is the control; they also push the pool to another part of the HBM).
#include "vmm_common.hpp"

int main(int argc, char **argv)
{
    const size_t g_mib = argc > 1 ? (size_t)atoll(argv[1]) : 64;
    const int rounds = argc > 2 ? atoi(argv[2]) : 8;
    const int pre = argc > 3 ? atoi(argv[3]) : 1;
    const int spare = argc > 4 ? atoi(argv[4]) : 0;
    CK(hipSetDevice(0));
    const double t_start = now_s();
    State st;
    st.init();
    std::vector<char *> held;
    for (int i = 0; i < pre; ++i) {
        void *p = nullptr;
        const double t0 = now_s();
        if (hipMalloc(&p, st.total) != hipSuccess) { (void)hipGetLastError(); break; }
        printf("{\"hipMalloc\": %d, \"alloc_s\": %.4f}\n", i, now_s() - t0);
        held.push_back((char *)p);
    }
    Arena ar;
    ar.create(g_mib << 20, st.total, spare);
    std::vector<int> ident((size_t)ar.slots), map;
    std::iota(ident.begin(), ident.end(), 0);
    std::vector<int> all((size_t)ar.n_handles);
    std::iota(all.begin(), all.end(), 0);
    std::mt19937 rng(99);
    State ctl = st;
    if (!held.empty()) {
        ctl.bind(held[0]);
        ctl.fill();
        ctl.iterate(1, nullptr);
    }
    auto measure = [&](State &s) {
        float ms[kSlices];
        double f = 0.0;
        s.iterate(1, nullptr);
        for (int r = 0; r < 4; ++r) {
            s.iterate(1, ms);
            f += ms[0];
        }
        return f / 4.0;
    };
    for (int r = 0; r < rounds; ++r) {
        for (int which = 0; which < 3; ++which) {
            // 0: creation order; 1: the same granules shuffled; 2: reversed order (contiguous again, other direction)
            if (which == 0) map = ident;
            else if (which == 1) {
                std::shuffle(all.begin(), all.end(), rng);
                map.assign(all.begin(), all.begin() + ar.slots);
            } else {
                map = ident;
                std::reverse(map.begin(), map.end());
            }
            const double tr = ar.remap(map);
            st.bind(ar.va);
            st.fill();
            const double v = measure(st);
            const double c = held.empty() ? 0.0 : measure(ctl);
            printf("{\"round\": %d, \"order\": \"%s\", \"vmm_ms\": %.4f, \"control_malloc_ms\": %.4f, \"remap_s\": %.4f, \"t\": %.2f}\n", r,
                   which == 0 ? "creation" : (which == 1 ? "shuffled" : "reversed"), v, c, tr, now_s() - t_start);
            fflush(stdout);
        }
    }
    CK(hipDeviceSynchronize());
    printf("{\"done\": true}\n");
    return 0;
}
