// Measurement aid (not product code): is the "placement lottery" a property of WHERE the state lies or of WHEN it is swept?
// Holds N_MALLOC states of BASELINE config 2 in hipMalloc allocations and one on VMM granules AT THE SAME TIME and times them
// in turn for SECONDS seconds, with a time stamp per measurement.  Differences between the columns are placement; changes
// along a column are the device's state over time.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/placement_timeline.hip -Iinclude -Lcytvdn_amd -ltvdn_hip -Wl,-rpath,'$ORIGIN/../../cytvdn_amd' -o tools/ubench/placement_timeline
//   tools/ubench/placement_timeline [SECONDS=30] [N_MALLOC=2] [GRANULE_MiB=64] [IDLE_MS=0] [FREE_EVERY_S=0] [SCHEDULE=""]
// SCHEDULE "15,8,15,2,10": sweep 15 s, idle 8 s, sweep 15 s, idle 2 s, sweep 10 s (does an idle device come back slow?)
#include "vmm_common.hpp"

int main(int argc, char **argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 30.0;
    const int n_malloc = argc > 2 ? atoi(argv[2]) : 2;
    const int idle_ms = argc > 4 ? atoi(argv[4]) : 0;
    const double free_every = argc > 5 ? atof(argv[5]) : 0.0;  // > 0: every so many seconds one hipMalloc state is freed (does a release slow the others down?)
    std::vector<double> sched;  // alternating busy / idle durations
    if (argc > 6)
        for (const char *c = argv[6]; *c;) {
            sched.push_back(atof(c));
            while (*c && *c != ',') ++c;
            if (*c == ',') ++c;
        }
    CK(hipSetDevice(0));
    const double t_start = now_s();
    State st;
    st.init();
    size_t free_b = 0, total_b = 0;
    CK(hipMemGetInfo(&free_b, &total_b));
    printf("{\"hbm_free_GiB\": %.2f, \"state_GiB\": %.4f, \"t\": %.3f}\n", free_b / 1073741824.0, st.total / 1073741824.0, now_s() - t_start);
    std::vector<char *> bases;
    std::vector<const char *> names;
    static char nm[8][32];
    for (int i = 0; i < n_malloc; ++i) {
        void *p = nullptr;
        const double t0 = now_s();
        if (hipMalloc(&p, st.total) != hipSuccess) { (void)hipGetLastError(); break; }
        printf("{\"hipMalloc\": %d, \"alloc_s\": %.4f, \"t\": %.3f}\n", i, now_s() - t0, now_s() - t_start);
        bases.push_back((char *)p);
        snprintf(nm[i], sizeof nm[i], "malloc%d", i);
        names.push_back(nm[i]);
    }
    // one pool of granules per entry of GRANULE_MiB ("1024,64": a 1 GiB pool, then a 64 MiB pool), all held at once
    std::vector<Arena> arenas;
    {
        std::vector<size_t> gs;
        for (const char *c = argc > 3 ? argv[3] : "64"; *c;) {
            gs.push_back((size_t)atoll(c));
            while (*c && *c != ',') ++c;
            if (*c == ',') ++c;
        }
        arenas.resize(gs.size());
        static char vn[8][32];
        for (size_t a = 0; a < gs.size(); ++a) {
            Arena &ar = arenas[a];
            ar.create(gs[a] << 20, st.total, 0);
            std::vector<int> map((size_t)ar.slots);
            std::iota(map.begin(), map.end(), 0);
            const double t_map = ar.remap(map);
            printf("{\"vmm\": \"mapped\", \"granule_MiB\": %zu, \"s\": %.4f, \"t\": %.3f}\n", gs[a], t_map, now_s() - t_start);
            bases.push_back(ar.va);
            snprintf(vn[a], sizeof vn[a], "vmm%zu", gs[a]);
            names.push_back(vn[a]);
        }
    }
    // every state filled once (zero accumulators, synthetic data term); the roles then just keep rotating
    std::vector<State> sts(bases.size(), st);
    for (size_t i = 0; i < bases.size(); ++i) {
        sts[i].bind(bases[i]);
        sts[i].fill();
        sts[i].iterate(1, nullptr);
    }
    fflush(stdout);
    int round = 0, freed = 0;
    double next_free = free_every > 0 ? (now_s() - t_start) + free_every : 1e30;
    size_t phase = 0;
    double phase_end = sched.empty() ? 1e30 : (now_s() - t_start) + sched[0];
    while (now_s() - t_start < seconds) {
        if (now_s() - t_start >= phase_end) {
            if (++phase >= sched.size()) break;
            CK(hipDeviceSynchronize());
            printf("{\"idle_s\": %.1f, \"t\": %.3f}\n", sched[phase], now_s() - t_start);
            fflush(stdout);
            struct timespec ts = {(time_t)sched[phase], (long)((sched[phase] - (double)(time_t)sched[phase]) * 1e9)};
            nanosleep(&ts, nullptr);
            if (++phase >= sched.size()) break;
            phase_end = (now_s() - t_start) + sched[phase];
        }
        if (now_s() - t_start >= next_free && freed + 1 < (int)bases.size()) {
            CK(hipDeviceSynchronize());
            const double t0 = now_s();
            CK(hipFree(bases[(size_t)freed]));
            printf("{\"freed\": \"%s\", \"hipFree_s\": %.4f, \"t\": %.3f}\n", names[(size_t)freed], now_s() - t0, now_s() - t_start);
            bases[(size_t)freed] = nullptr;
            ++freed;
            next_free += free_every;
        }
        printf("{\"round\": %d, \"t\": %.3f", round, now_s() - t_start);
        for (size_t i = 0; i < bases.size(); ++i) {
            if (!bases[i]) continue;
            float ms[kSlices];
            double f = 0.0;
            sts[i].iterate(1, nullptr);
            for (int r = 0; r < 3; ++r) {
                sts[i].iterate(1, ms);
                f += ms[0];
            }
            printf(", \"%s\": %.4f", names[i], f / 3.0);
        }
        printf("}\n");
        fflush(stdout);
        if (idle_ms > 0) {
            CK(hipDeviceSynchronize());
            struct timespec ts = {idle_ms / 1000, (long)(idle_ms % 1000) * 1000000L};
            nanosleep(&ts, nullptr);
        }
        ++round;
    }
    CK(hipDeviceSynchronize());
    printf("{\"done\": true}\n");
    return 0;
}
