// Measurement aid (not product code): VERDICT r4 item 5 -- do TWO (or more) iterations per trip through HBM pay on a RESIDENT cube?
// The fused sweep moves 15 arrays per 4-D FISTA iteration and runs at 0.9 of what a pure stream of that mix reaches, so the
// only lever left above a few per cent is to let iteration j + 1 consume iteration j's output from the 256 MiB Infinity Cache
// instead of from HBM.  The kernel already sweeps row ranges (tvdn_iter_args.sweep_lo / sweep_hi) and csrc/tvdn_run.hip already
// schedules levels that trail each other by one row (its pipelined upload): this probe runs that WAVEFRONT on a resident state --
// chunks of R rows, level j + 1 one row behind level j, K levels per trip -- for cubes of 2^30 voxels whose row-planes are
// 16 MiB (BASELINE config 2: a chunk's working set cannot stay in the cache), 4 MiB and 1 MiB (it can), beside plain whole sweeps.
//   what a trip must keep on chip between two levels: R + 2 rows x 15 arrays x plane bytes
//
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/temporal_probe.hip -Iinclude -Lcytvdn_amd -ltvdn_hip -Wl,-rpath,'$ORIGIN/../../cytvdn_amd' -o tools/ubench/temporal_probe
//   tools/ubench/temporal_probe [ITERS=16] [ONLY_SHAPE=-1] [ONLY_K=0] [ONLY_R=0]      (the ONLY_* restrict it to one case: counter runs)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "tvdn.h"

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            printf("{\"error\": \"%s: %s (line %d)\"}\n", #x, hipGetErrorString(e_), __LINE__);   \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)
#define TK(x)                                                                                     \
    do {                                                                                          \
        int r_ = (x);                                                                             \
        if (r_) {                                                                                 \
            printf("{\"error\": \"%s: %s (line %d)\"}\n", #x, tvdn_last_error(), __LINE__);       \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 16;
    const int only_shape = argc > 2 ? atoi(argv[2]) : -1, only_k = argc > 3 ? atoi(argv[3]) : 0, only_r = argc > 4 ? atoi(argv[4]) : 0;
    CK(hipSetDevice(0));
    tvdn_ctx *ctx = nullptr;
    TK(tvdn_ctx_create(&ctx, 0));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    double *sums = nullptr;
    CK(hipMalloc(&sums, 3 * sizeof(double) * (size_t)(iters + 1)));
    std::vector<double> ratios((size_t)iters);
    TK(tvdn_fista_ratios(iters, ratios.data()));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int64_t shapes[3][4] = {{256, 256, 128, 128}, {1024, 64, 128, 128}, {4096, 16, 128, 128}};
    int shape_index = -1;
    for (const auto &shape : shapes) {
        if (++shape_index != only_shape && only_shape >= 0) continue;
        const int64_t N0 = shape[0];
        const size_t row_bytes = (size_t)shape[1] * shape[2] * shape[3] * 4;
        const size_t bytes = (size_t)N0 * row_bytes, stride = bytes + 4096;
        void *base = nullptr;
        int32_t kind = 0;
        TK(tvdn_mem_alloc(&base, (int64_t)(stride * 15), 0, &kind));
        tvdn_many_args m;
        std::memset(&m, 0, sizeof m);
        int k = 0;
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 3; ++j) m.S[q][j] = (char *)base + stride * (size_t)(k++);
        m.recon[1] = (char *)base + stride * (size_t)(k++);
        char *orig = (char *)base + stride * (size_t)(k++);
        m.recon[0] = (char *)base + stride * (size_t)(k++);
        tvdn_iter_args &b = m.base;
        b.dtype = TVDN_F32;
        b.ndim = 4;
        for (int i = 0; i < 4; ++i) b.shape[i] = shape[i];
        b.row_lo = 0;
        b.row_hi = N0;
        b.lo_mode = TVDN_EDGE_BC;
        b.hi_mode = TVDN_EDGE_BC;
        b.bc_mode = TVDN_BC_JIA_ZHAO;
        const double mu[4] = {1, 1, .5, .5};
        for (int q = 0; q < 4; ++q) {
            const float lam = (float)mu[q] / 32.f;
            b.clip[q] = (double)(1.0f / lam);
            b.lambda_mu[q] = (double)(lam / (float)mu[q]);
        }
        b.orig = orig;
        auto reset = [&] {
            CK(hipMemsetAsync(base, 0, stride * 13, s));
            TK(tvdn_synth_fill(TVDN_F32, 4, shape, 20260302ull, 0, N0, orig, s));
            CK(hipMemcpyAsync(m.recon[0], orig, bytes, hipMemcpyDeviceToDevice, s));
            m.cur = 0;
            m.i_d = 0; m.i_prev = 1; m.i_out = 2;
            m.i_b = 0; m.i_bout = 1;
            m.d_form = 1;
            m.tk_prev = 0.0;
            CK(hipMemsetAsync(sums, 0, 3 * sizeof(double) * (size_t)(iters + 1), s));
        };
        // (a) plain: one whole sweep per iteration
        double plain_ms = 0.0;
        {
            reset();
            for (int pass = 0; pass < 2; ++pass) {  // the first pass warms up
                CK(hipEventRecord(e0, s));
                TK(tvdn_iterate_many(ctx, &m, iters, ratios.data(), 0, sums, s));
                CK(hipEventRecord(e1, s));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                plain_ms = ms / iters;
                if (pass == 0) reset();
            }
            double chk[3];
            CK(hipMemcpy(chk, sums + 3 * (size_t)(iters - 1), sizeof chk, hipMemcpyDeviceToHost));
            printf("{\"shape\": [%lld,%lld,%lld,%lld], \"plane_MiB\": %.1f, \"schedule\": \"whole sweeps\", \"ms_per_iteration\": %.4f, \"Gvoxel_iters_per_s\": %.2f, \"b_norm_last\": %.6e, \"state_mem\": %d}\n",
                   (long long)shape[0], (long long)shape[1], (long long)shape[2], (long long)shape[3], row_bytes / 1048576.0, plain_ms, 1073.741824 / plain_ms, chk[0], kind);
            fflush(stdout);
        }
        // (b) wavefront: K levels per trip, chunks of R rows; Jia-Zhao top face as a constant (TVDN_EDGE_ZERO: finite data)
        for (int K : {2, 4, 8}) {
            for (int R : {1, 2, 4, 8, 16, 32}) {
                if ((size_t)(R + 2) * 15 * row_bytes > ((size_t)3 << 30) || iters % K) continue;
                if ((only_k && K != only_k) || (only_r && R != only_r)) continue;
                double ms_it = 0.0, chk0 = 0.0;
                for (int pass = 0; pass < 2; ++pass) {
                    reset();
                    CK(hipEventRecord(e0, s));
                    for (int first = 0; first < iters; first += K) {
                        std::vector<tvdn_many_args> snap((size_t)K);
                        tvdn_many_args mm = m;
                        for (int j = 0; j < K; ++j) {
                            snap[(size_t)j] = mm;
                            TK(tvdn_roles_advance(&mm, 1, ratios[(size_t)(first + j)]));
                        }
                        const int64_t n_chunks = (N0 + K + R - 1) / R;
                        for (int64_t c = 0; c < n_chunks; ++c)
                            for (int j = 0; j < K; ++j) {
                                const int64_t lo = std::max<int64_t>(0, c * R - (j + 1)), hi = std::min<int64_t>(N0, (c + 1) * R - (j + 1));
                                if (lo >= hi) continue;
                                tvdn_iter_args it = snap[(size_t)j].base;
                                TK(tvdn_roles_bind(&snap[(size_t)j], 1, ratios[(size_t)(first + j)], &it));
                                it.hi_mode = TVDN_EDGE_ZERO;
                                it.sweep_lo = lo;
                                it.sweep_hi = hi;
                                it.accumulate = 1;
                                TK(tvdn_iterate_fused(ctx, &it, sums + 3 * (size_t)(first + j), s));
                            }
                        const tvdn_iter_args keep = m.base;
                        m = mm;
                        m.base = keep;
                    }
                    CK(hipEventRecord(e1, s));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    ms_it = ms / iters;
                    double chk[3];
                    CK(hipMemcpy(chk, sums + 3 * (size_t)(iters - 1), sizeof chk, hipMemcpyDeviceToHost));
                    chk0 = chk[0];
                }
                printf("{\"shape\": [%lld,%lld,%lld,%lld], \"plane_MiB\": %.1f, \"schedule\": \"wavefront\", \"levels_per_trip\": %d, \"chunk_rows\": %d, \"between_levels_MiB\": %.0f, "
                       "\"ms_per_iteration\": %.4f, \"vs_whole_sweeps\": %.3f, \"b_norm_last\": %.6e}\n",
                       (long long)shape[0], (long long)shape[1], (long long)shape[2], (long long)shape[3], row_bytes / 1048576.0, K, R, (double)(R + 2) * 15 * row_bytes / 1048576.0, ms_it,
                       ms_it / plain_ms, chk0);
                fflush(stdout);
            }
        }
        CK(hipDeviceSynchronize());
        TK(tvdn_mem_free(base));
    }
    printf("{\"done\": true}\n");
    return 0;
}
