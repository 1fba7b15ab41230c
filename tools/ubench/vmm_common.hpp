// Shared by the VMM placement probes (measurement aids, not product code): the config-2 state as tvdn_run lays it out
// (State) and a virtual range composed of physical granules (Arena).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <vector>

#include "tvdn.h"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            printf("{\"error\": \"%s: %s (line %d)\"}\n", #x, hipGetErrorString(e_), __LINE__);   \
            fflush(stdout);                                                                       \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)
#define TK(x)                                                                                     \
    do {                                                                                          \
        int r_ = (x);                                                                             \
        if (r_) {                                                                                 \
            printf("{\"error\": \"%s: %s (line %d)\"}\n", #x, tvdn_last_error(), __LINE__);       \
            fflush(stdout);                                                                       \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static const int64_t kShape[4] = {256, 256, 128, 128};
static const int kNd = 4, kPerAxis = 3, kArr = 15, kSlices = 4;

struct State {
    tvdn_ctx *ctx = nullptr;
    hipStream_t s = nullptr;
    tvdn_many_args roles;
    char *base = nullptr;
    size_t stride = 0, total = 0, row_bytes = 0;
    double *sums = nullptr;
    double ratios[64];
    int it = 0;
    hipEvent_t ev[kSlices + 1];

    void init()
    {
        TK(tvdn_ctx_create(&ctx, 0));
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        CK(hipMalloc(&sums, 3 * sizeof(double)));
        row_bytes = (size_t)kShape[1] * kShape[2] * kShape[3] * 4;
        const size_t bytes = (size_t)kShape[0] * row_bytes;
        set_skew(4096);
        TK(tvdn_fista_ratios(64, ratios));
        for (auto &e : ev) CK(hipEventCreate(&e));
    }
    // consecutive arrays start `skew` bytes beyond a whole array apart (tvdn_run: 4096)
    void set_skew(size_t skew)
    {
        const size_t bytes = (size_t)kShape[0] * row_bytes;
        stride = (bytes + 255) / 256 * 256 + skew;
        total = stride * kArr;
    }
    void bind(char *b)
    {
        base = b;
        std::memset(&roles, 0, sizeof roles);
        int k = 0;
        for (int q = 0; q < kNd; ++q)
            for (int j = 0; j < kPerAxis; ++j) roles.S[q][j] = base + stride * (size_t)(k++);
        roles.recon[1] = base + stride * (size_t)(k++);
        char *orig = base + stride * (size_t)(k++);
        roles.recon[0] = base + stride * (size_t)(k++);
        roles.cur = 0;
        roles.i_d = 0; roles.i_prev = 1; roles.i_out = 2;
        roles.i_b = 0; roles.i_bout = 1;
        roles.d_form = 1;
        roles.tk_prev = 0.0;
        tvdn_iter_args &a = roles.base;
        a.dtype = TVDN_F32;
        a.ndim = kNd;
        for (int i = 0; i < 4; ++i) a.shape[i] = kShape[i];
        a.row_lo = 0; a.row_hi = kShape[0];
        a.lo_mode = TVDN_EDGE_BC; a.hi_mode = TVDN_EDGE_BC;
        a.bc_mode = TVDN_BC_JIA_ZHAO;
        const double mu[4] = {1, 1, .5, .5};
        for (int q = 0; q < 4; ++q) {
            const float lam = (float)mu[q] / 32.f;
            a.clip[q] = (double)(1.0f / lam);
            a.lambda_mu[q] = (double)(lam / (float)mu[q]);
        }
        a.orig = orig;
        it = 0;
    }
    // zero accumulators, synthetic data term, recon = data (as tvdn_run starts)
    void fill()
    {
        CK(hipMemsetAsync(base, 0, stride * (size_t)(kArr - 2), s));
        TK(tvdn_synth_fill(TVDN_F32, kNd, kShape, 20260302ull, 0, kShape[0], (void *)roles.base.orig, s));
        CK(hipMemcpyAsync(roles.recon[0], roles.base.orig, (size_t)kShape[0] * row_bytes, hipMemcpyDeviceToDevice, s));
        roles.cur = 0;
        roles.i_d = 0; roles.i_prev = 1; roles.i_out = 2;
        roles.d_form = 1;
        roles.tk_prev = 0.0;
        it = 0;
    }
    // one iteration as `n_slices` launches over equal row ranges; ms[i] = time of launch i (events between launches)
    void iterate(int n_slices, float *ms)
    {
        tvdn_iter_args a = roles.base;
        const double r = ratios[it % 48];
        TK(tvdn_roles_bind(&roles, 1, r, &a));
        const int64_t rows = kShape[0] / n_slices;
        CK(hipEventRecord(ev[0], s));
        for (int i = 0; i < n_slices; ++i) {
            a.sweep_lo = n_slices == 1 ? 0 : i * rows;
            a.sweep_hi = n_slices == 1 ? 0 : (i + 1) * rows;
            a.accumulate = i > 0;
            TK(tvdn_iterate_fused(ctx, &a, sums, s));
            CK(hipEventRecord(ev[i + 1], s));
        }
        TK(tvdn_roles_advance(&roles, 1, r));
        ++it;
        CK(hipEventSynchronize(ev[n_slices]));
        if (ms)
            for (int i = 0; i < n_slices; ++i) CK(hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
    }
    // warm, then `reps` whole sweeps and `reps` sliced iterations: mean full ms, mean per-slice ms
    void measure(int reps, double *full_ms, double *slice_ms)
    {
        for (int i = 0; i < 2; ++i) iterate(1, nullptr);
        double f = 0.0, sl[kSlices] = {0, 0, 0, 0};
        float ms[kSlices];
        for (int i = 0; i < reps; ++i) {
            iterate(1, ms);
            f += ms[0];
        }
        for (int i = 0; i < reps; ++i) {
            iterate(kSlices, ms);
            for (int j = 0; j < kSlices; ++j) sl[j] += ms[j];
        }
        *full_ms = f / reps;
        for (int j = 0; j < kSlices; ++j) slice_ms[j] = sl[j] / reps;
    }
};

struct Arena {
    size_t G = 0;
    int slots = 0, n_handles = 0;
    char *va = nullptr;
    std::vector<hipMemGenericAllocationHandle_t> h;
    std::vector<int> at;  // handle mapped at slot i (-1: none)
    hipMemAccessDesc acc;

    void create(size_t granule, size_t total, int spare)
    {
        G = granule;
        slots = (int)((total + G - 1) / G);
        n_handles = slots + spare;
        hipMemAllocationProp prop;
        std::memset(&prop, 0, sizeof prop);
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        size_t gmin = 0, grec = 0;
        CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
        CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
        printf("{\"vmm\": \"granularity\", \"min\": %zu, \"recommended\": %zu, \"granule\": %zu, \"slots\": %d, \"handles\": %d}\n", gmin, grec, G, slots, n_handles);
        fflush(stdout);
        double t0 = now_s();
        h.resize((size_t)n_handles);
        double worst = 0.0;
        for (int i = 0; i < n_handles; ++i) {
            const double a = now_s();
            CK(hipMemCreate(&h[(size_t)i], G, &prop, 0));
            worst = std::max(worst, now_s() - a);
        }
        const double t_create = now_s() - t0;
        t0 = now_s();
        CK(hipMemAddressReserve((void **)&va, (size_t)slots * G, G < (1u << 21) ? (1u << 21) : G, nullptr, 0));
        const double t_reserve = now_s() - t0;
        at.assign((size_t)slots, -1);
        std::memset(&acc, 0, sizeof acc);
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = 0;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        printf("{\"vmm\": \"create\", \"handles\": %d, \"create_s\": %.4f, \"worst_create_s\": %.4f, \"reserve_s\": %.6f, \"va\": \"%p\"}\n", n_handles, t_create, worst, t_reserve, (void *)va);
        fflush(stdout);
    }
    // map handle map[i] at slot i for all i; returns seconds (unmap + map + set access)
    double remap(const std::vector<int> &map, double *t_unmap = nullptr, double *t_map = nullptr, double *t_acc = nullptr)
    {
        CK(hipDeviceSynchronize());
        const double t0 = now_s();
        for (int i = 0; i < slots; ++i)
            if (at[(size_t)i] >= 0 && at[(size_t)i] != map[(size_t)i]) {
                CK(hipMemUnmap(va + (size_t)i * G, G));
                at[(size_t)i] = -1;
            }
        const double t1 = now_s();
        int n_new = 0;
        for (int i = 0; i < slots; ++i)
            if (at[(size_t)i] < 0) {
                CK(hipMemMap(va + (size_t)i * G, G, 0, h[(size_t)map[(size_t)i]], 0));
                ++n_new;
            }
        const double t2 = now_s();
        // access rights: one call per run of freshly mapped slots
        for (int i = 0; i < slots;) {
            if (at[(size_t)i] >= 0) { ++i; continue; }
            int j = i;
            while (j < slots && at[(size_t)j] < 0) ++j;
            CK(hipMemSetAccess(va + (size_t)i * G, (size_t)(j - i) * G, &acc, 1));
            for (int k = i; k < j; ++k) at[(size_t)k] = map[(size_t)k];
            i = j;
        }
        // ROCm 7.2: the GPU goes on using the OLD translation of an address that was unmapped and mapped again until something
        // flushes its TLBs; the virtual-memory calls do not, a hipFree does (tools/ubench/vmm_remap_flush.hip).  Without this
        // every remap in these probes was a no-op as far as the kernels were concerned.
        {
            void *d = nullptr;
            CK(hipMalloc(&d, 4 << 20));
            CK(hipFree(d));
        }
        const double t3 = now_s();
        if (t_unmap) *t_unmap = t1 - t0;
        if (t_map) *t_map = t2 - t1;
        if (t_acc) *t_acc = t3 - t2;
        (void)n_new;
        return t3 - t0;
    }
};

