// Measurement aid (not product code): can the state of a resident run be COMPOSED from physical granules through HIP's
// virtual-memory management (hipMemCreate / hipMemAddressReserve / hipMemMap), and does the sweep's speed belong to the
// granules (a set to choose) or to their arrangement (an order to search)?   VERDICT r4 item 1.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/vmm_probe.hip -Iinclude -Lcytvdn_amd -ltvdn_hip -Wl,-rpath,'$ORIGIN/../../cytvdn_amd' -o tools/ubench/vmm_probe
//   tools/ubench/vmm_probe [GRANULE_MiB=1024] [SPARE_GRANULES=19] [TRIALS=40]
//
// Workload: BASELINE config 2 (256x256x128x128 f32 FISTA, compact state: 15 arrays of 4 GiB + 4 KiB), laid out exactly as
// tvdn_run lays it out (csrc/tvdn_run.hip Slab::assign).  Every line of output is one JSON object.
#include "vmm_common.hpp"

// least squares with a small ridge: rows = (handles of slice s of trial t) -> slice time
static std::vector<double> solve(int n, const std::vector<std::vector<int>> &rows, const std::vector<double> &y, double *rms)
{
    std::vector<std::vector<double>> A((size_t)n, std::vector<double>((size_t)n + 1, 0.0));
    for (size_t r = 0; r < rows.size(); ++r)
        for (int i : rows[r]) {
            for (int j : rows[r]) A[(size_t)i][(size_t)j] += 1.0;
            A[(size_t)i][(size_t)n] += y[r];
        }
    double mean = 0.0;
    for (size_t r = 0; r < rows.size(); ++r) mean += y[r] / (double)rows[r].size();
    mean /= (double)rows.size();
    for (int i = 0; i < n; ++i) {  // ridge towards the mean cost
        A[(size_t)i][(size_t)i] += 1e-3;
        A[(size_t)i][(size_t)n] += 1e-3 * mean;
    }
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r)
            if (std::fabs(A[(size_t)r][(size_t)c]) > std::fabs(A[(size_t)p][(size_t)c])) p = r;
        std::swap(A[(size_t)c], A[(size_t)p]);
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            const double f = A[(size_t)r][(size_t)c] / A[(size_t)c][(size_t)c];
            if (f == 0.0) continue;
            for (int k = c; k <= n; ++k) A[(size_t)r][(size_t)k] -= f * A[(size_t)c][(size_t)k];
        }
    }
    std::vector<double> x((size_t)n);
    for (int i = 0; i < n; ++i) x[(size_t)i] = A[(size_t)i][(size_t)n] / A[(size_t)i][(size_t)i];
    double ss = 0.0;
    for (size_t r = 0; r < rows.size(); ++r) {
        double p = 0.0;
        for (int i : rows[r]) p += x[(size_t)i];
        ss += (p - y[r]) * (p - y[r]);
    }
    *rms = std::sqrt(ss / (double)rows.size());
    return x;
}

static void print_map(const char *key, const std::vector<int> &m)
{
    printf("\"%s\": [", key);
    for (size_t i = 0; i < m.size(); ++i) printf("%s%d", i ? "," : "", m[i]);
    printf("]");
}

int main(int argc, char **argv)
{
    const size_t g_mib = argc > 1 ? (size_t)atoll(argv[1]) : 1024;
    const int spare = argc > 2 ? atoi(argv[2]) : 19;
    const int trials = argc > 3 ? atoi(argv[3]) : 40;
    const int n_malloc = argc > 4 ? atoi(argv[4]) : 3;
    CK(hipSetDevice(0));
    State st;
    st.init();
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("{\"hbm_free_GiB\": %.2f, \"hbm_total_GiB\": %.2f, \"state_GiB\": %.4f}\n", free_b / 1073741824.0, total_b / 1073741824.0, st.total / 1073741824.0);

    // ---- A. the lottery as it is: the state in hipMalloc allocations held side by side ------------------------------------
    {
        std::vector<void *> held;
        for (int i = 0; i < n_malloc; ++i) {
            void *p = nullptr;
            const double t0 = now_s();
            if (hipMalloc(&p, st.total) != hipSuccess) { (void)hipGetLastError(); break; }
            const double t_alloc = now_s() - t0;
            held.push_back(p);
            st.bind((char *)p);
            st.fill();
            double f, sl[kSlices];
            st.measure(6, &f, sl);
            printf("{\"phase\": \"hipMalloc\", \"i\": %d, \"alloc_s\": %.4f, \"full_ms\": %.4f, \"slice_ms\": [%.4f,%.4f,%.4f,%.4f], \"slices_sum_ms\": %.4f}\n", i, t_alloc, f, sl[0], sl[1],
                   sl[2], sl[3], sl[0] + sl[1] + sl[2] + sl[3]);
            fflush(stdout);
        }
        CK(hipDeviceSynchronize());
        for (void *p : held) CK(hipFree(p));
    }

    // ---- B. the same state on granules ------------------------------------------------------------------------------------
    Arena ar;
    ar.create(g_mib << 20, st.total, spare);
    const int S = ar.slots, H = ar.n_handles;
    std::vector<int> map((size_t)S);
    std::iota(map.begin(), map.end(), 0);
    double tu, tm, ta;
    const double t_first = ar.remap(map, &tu, &tm, &ta);
    printf("{\"vmm\": \"first map\", \"s\": %.4f, \"map_s\": %.4f, \"access_s\": %.4f}\n", t_first, tm, ta);
    st.bind(ar.va);
    auto run = [&](const char *phase, int idx, double t_remap, bool with_map) {
        st.fill();
        double f, sl[kSlices];
        st.measure(4, &f, sl);
        printf("{\"phase\": \"%s\", \"i\": %d, \"remap_s\": %.4f, \"full_ms\": %.4f, \"slice_ms\": [%.4f,%.4f,%.4f,%.4f], \"slices_sum_ms\": %.4f", phase, idx, t_remap, f, sl[0], sl[1],
               sl[2], sl[3], sl[0] + sl[1] + sl[2] + sl[3]);
        if (with_map) {
            printf(", ");
            print_map("map", map);
        }
        printf("}\n");
        fflush(stdout);
        return std::vector<double>{f, sl[0], sl[1], sl[2], sl[3]};
    };
    run("identity", 0, t_first, false);
    run("identity", 1, 0.0, false);

    std::mt19937 rng(20261004);
    // which handles a row slice of the sweep touches: array k occupies bytes [k stride, k stride + 4 GiB)
    const size_t slice_bytes = (size_t)(kShape[0] / kSlices) * st.row_bytes;
    auto slice_handles = [&](int s) {
        std::vector<int> v;
        for (int k = 0; k < kArr; ++k) {
            const size_t b0 = (size_t)k * st.stride + (size_t)s * slice_bytes, b1 = b0 + slice_bytes - 1;
            for (size_t g = b0 / ar.G; g <= b1 / ar.G; ++g)
                if (std::find(v.begin(), v.end(), map[g]) == v.end() && (std::min(b1, (g + 1) * ar.G - 1) - std::max(b0, g * ar.G)) > ar.G / 64) v.push_back(map[g]);
        }
        return v;
    };

    // ---- C. the same granules in other orders: is it the set or the arrangement? ------------------------------------------
    const std::vector<int> first_set = map;
    for (int t = 0; t < 5; ++t) {
        std::shuffle(map.begin(), map.end(), rng);
        const double tr = ar.remap(map, &tu, &tm, &ta);
        if (t == 0) printf("{\"vmm\": \"remap all\", \"s\": %.4f, \"unmap_s\": %.4f, \"map_s\": %.4f, \"access_s\": %.4f}\n", tr, tu, tm, ta);
        run("same set, shuffled", t, tr, false);
    }

    // ---- D. random sets out of all handles: per-granule costs by least squares over the slice times ------------------------
    std::vector<std::vector<int>> rows;
    std::vector<double> y;
    std::vector<int> all((size_t)H);
    std::iota(all.begin(), all.end(), 0);
    double worst_full = 0.0, best_full = 1e30;
    for (int t = 0; t < trials; ++t) {
        std::shuffle(all.begin(), all.end(), rng);
        for (int i = 0; i < S; ++i) map[(size_t)i] = all[(size_t)i];
        const double tr = ar.remap(map);
        const std::vector<double> r = run("random set", t, tr, true);
        worst_full = std::max(worst_full, r[0]);
        best_full = std::min(best_full, r[0]);
        if (ar.G >= slice_bytes)
            for (int s = 0; s < kSlices; ++s) {
                rows.push_back(slice_handles(s));
                y.push_back(r[(size_t)s + 1]);
            }
    }
    printf("{\"phase\": \"random sets\", \"trials\": %d, \"best_full_ms\": %.4f, \"worst_full_ms\": %.4f}\n", trials, best_full, worst_full);
    if (!rows.empty()) {
        double rms = 0.0;
        const std::vector<double> c = solve(H, rows, y, &rms);
        std::vector<int> order((size_t)H);
        std::iota(order.begin(), order.end(), 0);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return c[(size_t)a] < c[(size_t)b]; });
        printf("{\"phase\": \"fit\", \"equations\": %zu, \"unknowns\": %d, \"rms_ms\": %.5f, \"cost_ms_sorted\": [", rows.size(), H, rms);
        for (int i = 0; i < H; ++i) printf("%s%.4f", i ? "," : "", c[(size_t)order[(size_t)i]]);
        printf("], ");
        print_map("handle_sorted", order);
        printf("}\n");
        // ---- E. the predicted best and worst sets, each in three orders ---------------------------------------------------
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < S; ++i) map[(size_t)i] = which == 0 ? order[(size_t)i] : order[(size_t)(H - 1 - i)];
            double pred = 0.0;
            for (int i = 0; i < S; ++i) pred += c[(size_t)map[(size_t)i]];
            for (int t = 0; t < 3; ++t) {
                if (t) std::shuffle(map.begin(), map.end(), rng);
                const double tr = ar.remap(map);
                printf("{\"predicted_slices_sum_ms\": %.4f}\n", pred);
                run(which == 0 ? "predicted best set" : "predicted worst set", t, tr, true);
            }
        }
        // ---- F. one granule at a time: the identity set with its costliest member replaced by the cheapest spare, repeatedly --
        map = first_set;
        double tr = ar.remap(map);
        std::vector<double> base = run("greedy start (identity)", 0, tr, false);
        std::vector<char> used((size_t)H, 0);
        for (int i = 0; i < S; ++i) used[(size_t)map[(size_t)i]] = 1;
        for (int step = 0; step < 8; ++step) {
            int wi = -1;
            for (int i = 0; i < S; ++i)
                if (wi < 0 || c[(size_t)map[(size_t)i]] > c[(size_t)map[(size_t)wi]]) wi = i;
            int bh = -1;
            for (int hh = 0; hh < H; ++hh)
                if (!used[(size_t)hh] && (bh < 0 || c[(size_t)hh] < c[(size_t)bh])) bh = hh;
            if (bh < 0 || c[(size_t)bh] >= c[(size_t)map[(size_t)wi]]) break;
            const double pred = c[(size_t)map[(size_t)wi]] - c[(size_t)bh];
            used[(size_t)map[(size_t)wi]] = 0;
            used[(size_t)bh] = 1;
            map[(size_t)wi] = bh;
            tr = ar.remap(map, &tu, &tm, &ta);
            printf("{\"swap_slot\": %d, \"predicted_gain_ms\": %.4f, \"one_granule_remap_s\": %.5f}\n", wi, pred, tr);
            run("greedy swap", step, tr, false);
        }
    }
    CK(hipDeviceSynchronize());
    printf("{\"done\": true}\n");
    return 0;
}
