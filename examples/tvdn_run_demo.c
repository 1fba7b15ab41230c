/*
 * Plain-C caller of the C ABI (include/tvdn.h): no Python, no torch.
 *
 *   gcc -O2 -Iinclude examples/tvdn_run_demo.c -Lcytvdn_amd -ltvdn_hip -Wl,-rpath,$PWD/cytvdn_amd -lm -o tvdn_run_demo
 *   ./tvdn_run_demo 12 10 16 32 8        # shape (4-D) and FISTA iterations
 *   ./tvdn_run_demo 12 10 16 32 8 2      # ... cut into 2 slabs: devices 0 and 1 (or both on device 0 if there is one GPU)
 *
 * Fills a cube with a deterministic pattern, runs denoise4D's loop through tvdn_run and prints the
 * traces and an FNV-1a checksum of recon (tests/test_gpu_parity.py compares it with the Python path).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "tvdn.h"

/* tvdn_run_args.progress: called on this thread with the number of iteration slots handed to the GPU so far */
static void on_progress(int32_t slots_done, void *user)
{
    fprintf(stderr, "\r%d / %d iterations queued%s", (int)slots_done, *(const int *)user, slots_done == *(const int *)user ? "\n" : "");
}

int main(int argc, char **argv)
{
    int64_t shape[4] = {12, 10, 16, 32};
    int iters = 8;
    for (int i = 0; i < 4 && i + 1 < argc; ++i) shape[i] = atoll(argv[i + 1]);
    if (argc > 5) iters = atoi(argv[5]);
    const int slabs = argc > 6 ? atoi(argv[6]) : 0;
    size_t n = (size_t)(shape[0] * shape[1] * shape[2] * shape[3]);
    float *x = malloc(n * sizeof(float)), *recon = malloc(n * sizeof(float));
    double *sums = calloc((size_t)iters * 3, sizeof(double));
    if (!x || !recon || !sums) return 2;
    uint64_t z = 88172645463325252ull;
    for (size_t i = 0; i < n; ++i) {           /* xorshift64 counts 0..15 plus a smooth ramp */
        z ^= z << 13; z ^= z >> 7; z ^= z << 17;
        x[i] = (float)(z >> 60) + 0.001f * (float)(i % 977);
    }
    const float mu[4] = {1.0f, 1.0f, 0.5f, 0.5f};
    tvdn_run_args a = {0};
    a.dtype = TVDN_F32; a.ndim = 4; a.bc_mode = TVDN_BC_JIA_ZHAO; a.device = 0;
    a.n_fista = iters; a.n_plain = 0; a.use_stop = 0;
    if (slabs > 0) {                                     /* one slab of axis 0 per device-list entry */
        const int ngpu = tvdn_device_count();
        a.n_devices = slabs > TVDN_MAX_DEVICES ? TVDN_MAX_DEVICES : slabs;
        for (int i = 0; i < a.n_devices; ++i) a.devices[i] = ngpu > 0 ? i % ngpu : 0;
    }
    for (int q = 0; q < 4; ++q) {
        a.shape[q] = shape[q];
        const float lam = mu[q] * 1.0f / 32.0f;          /* cyTVDN.py:67-68, in the data dtype */
        a.clip[q] = (double)(1.0f / lam);                /* cyTVDN.py:77 */
        a.lambda_mu[q] = (double)(lam / mu[q]);          /* cyTVDN.py:78 */
    }
    a.data = x; a.recon_out = recon; a.sums_out = sums;
    int32_t ran = 0;
    a.iters_run = &ran;
    a.progress = on_progress;
    a.progress_user = &iters;
    int rc = tvdn_run(&a);
    if (rc != TVDN_OK) {
        fprintf(stderr, "tvdn_run failed (%d): %s\n", rc, tvdn_last_error());
        return 1;
    }
    uint64_t h = 1469598103934665603ull;
    const unsigned char *p = (const unsigned char *)recon;
    for (size_t i = 0; i < n * sizeof(float); ++i) { h ^= p[i]; h *= 1099511628211ull; }
    printf("iters_run %d\n", ran);
    for (int i = 0; i < iters; ++i)
        printf("iter %d b_norm %.17g delta %.9g\n", i, sums[3 * i], (float)sums[3 * i + 1] / (float)sums[3 * i + 2]);
    printf("recon_fnv1a %016llx\n", (unsigned long long)h);
    free(x); free(recon); free(sums);
    return 0;
}
