// Measurement aid (not product code): does the sweep's speed follow WHICH PART of the HBM the state's granules come from?
// Takes (nearly) all free HBM as 1 GiB granules in allocation order, then times the config-2 sweep on
//   windows   61 granules in a row starting at i, for i = 0, STEP, 2 STEP, ...          (a hipMalloc-like placement)
//   mixtures  a fraction f of the slots from the window at i, the rest from the window at j, dealt out alternately
//   strided   every k-th granule of the whole pool
// If windows differ and mixtures of a slow and a fast window are as fast as the fast one (or faster), placement can be had
// by construction: compose the state from granules of different parts.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/vmm_windows.hip -Iinclude -Lcytvdn_amd -ltvdn_hip -Wl,-rpath,'$ORIGIN/../../cytvdn_amd' -o tools/ubench/vmm_windows
//   tools/ubench/vmm_windows [GRANULE_MiB=1024] [STEP=8] [LEAVE_GiB=8]
#include "vmm_common.hpp"

int main(int argc, char **argv)
{
    const size_t g_mib = argc > 1 ? (size_t)atoll(argv[1]) : 1024;
    const int step = argc > 2 ? atoi(argv[2]) : 8;
    const double leave_gib = argc > 3 ? atof(argv[3]) : 8.0;
    CK(hipSetDevice(0));
    const double t_start = now_s();
    State st;
    st.init();
    // what a plain allocation draws on this box right now: two hipMalloc states side by side, timed, then given back
    {
        std::vector<void *> held;
        for (int i = 0; i < 2; ++i) {
            void *p = nullptr;
            const double t0 = now_s();
            if (hipMalloc(&p, st.total) != hipSuccess) { (void)hipGetLastError(); break; }
            const double ta = now_s() - t0;
            held.push_back(p);
            st.bind((char *)p);
            st.fill();
            double f, sl[kSlices];
            st.measure(4, &f, sl);
            printf("{\"hipMalloc\": %d, \"alloc_s\": %.4f, \"full_ms\": %.4f}\n", i, ta, f);
        }
        CK(hipDeviceSynchronize());
        for (void *p : held) CK(hipFree(p));
        struct timespec ts = {argc > 4 ? atoi(argv[4]) : 3, 0};
        nanosleep(&ts, nullptr);
    }
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    const size_t G = g_mib << 20;
    const int slots = (int)((st.total + G - 1) / G);
    const int H = (int)((free_b - (size_t)(leave_gib * 1073741824.0)) / G);
    printf("{\"hbm_free_GiB\": %.2f, \"granule_MiB\": %zu, \"slots\": %d, \"handles\": %d}\n", free_b / 1073741824.0, g_mib, slots, H);
    Arena ar;
    ar.create(G, st.total, H - slots);
    std::vector<int> map((size_t)slots);
    auto timed = [&](const char *what, int a, int b, double f) {
        const double tr = ar.remap(map);
        st.bind(ar.va);
        st.fill();
        double full, sl[kSlices];
        st.measure(4, &full, sl);
        printf("{\"set\": \"%s\", \"a\": %d, \"b\": %d, \"f\": %.3f, \"full_ms\": %.4f, \"slice_ms\": [%.4f,%.4f,%.4f,%.4f], \"remap_s\": %.4f, \"t\": %.2f}\n", what, a, b, f, full, sl[0],
               sl[1], sl[2], sl[3], tr, now_s() - t_start);
        fflush(stdout);
        return full;
    };
    // ---- windows, twice (the second pass shows what is time and what is place) -------------------------------------------
    std::vector<std::pair<double, int>> win;
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i + slots <= H; i += step) {
            for (int s = 0; s < slots; ++s) map[(size_t)s] = i + s;
            const double t = timed("window", i, i, 1.0);
            if (pass == 1) win.push_back({t, i});
        }
    std::sort(win.begin(), win.end());
    const int fast = win.front().second, slow = win.back().second;
    printf("{\"fastest_window\": %d, \"ms\": %.4f, \"slowest_window\": %d, \"slowest_ms\": %.4f}\n", fast, win.front().first, slow, win.back().first);
    // ---- mixtures of the slowest window with the one farthest from it, and with the fastest ------------------------------
    int far = 0;
    for (auto &w : win)
        if (std::abs(w.second - slow) > std::abs(far - slow)) far = w.second;
    for (int other : {far, fast}) {
        if (std::abs(other - slow) < slots) continue;  // overlapping windows share granules
        for (double f : {0.25, 0.5, 0.75}) {
            // slot s takes from `slow` when floor((s+1) f) > floor(s f), else from `other`: evenly dealt
            int na = 0, nb = 0;
            for (int s = 0; s < slots; ++s) {
                const bool from_a = (long long)((s + 1) * f) > (long long)(s * f);
                map[(size_t)s] = from_a ? slow + na++ : other + nb++;
            }
            timed("mixture", slow, other, f);
        }
        // the same halves, but array by array: arrays (4 granules each) alternate between the two windows
        {
            int na = 0, nb = 0;
            for (int s = 0; s < slots; ++s) {
                const bool from_a = ((s / 4) % 2) == 0;
                map[(size_t)s] = from_a ? slow + na++ : other + nb++;
            }
            timed("mixture by array", slow, other, 0.5);
        }
        // ... and by row slice: slices 0,1 of every array from one window, 2,3 from the other (the streams of one moment all in one part)
        {
            int na = 0, nb = 0;
            for (int s = 0; s < slots; ++s) {
                const bool from_a = (s % 4) < 2;
                map[(size_t)s] = from_a ? slow + na++ : other + nb++;
            }
            timed("mixture by slice", slow, other, 0.5);
        }
    }
    // ---- strided over the whole pool ----------------------------------------------------------------------------------------
    for (int k : {2, 3, 4}) {
        if ((slots - 1) * k >= H) continue;
        for (int off = 0; off < k && off < 2; ++off) {
            for (int s = 0; s < slots; ++s) map[(size_t)s] = off + s * k;
            timed("strided", k, off, 0.0);
        }
    }
    // ---- a random set of the whole pool, three draws ------------------------------------------------------------------------
    std::mt19937 rng(7);
    std::vector<int> all((size_t)H);
    std::iota(all.begin(), all.end(), 0);
    for (int t = 0; t < 3; ++t) {
        std::shuffle(all.begin(), all.end(), rng);
        for (int s = 0; s < slots; ++s) map[(size_t)s] = all[(size_t)s];
        timed("random", t, 0, 0.0);
    }
    CK(hipDeviceSynchronize());
    printf("{\"done\": true}\n");
    return 0;
}
