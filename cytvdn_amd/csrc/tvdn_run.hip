// tvdn_run: the whole denoise loop on host arrays, for non-Python callers (include/tvdn.h).
// Mirrors cytvdn_amd/driver.py + engine.HipBackend (compact d-rotation state) in C++.
#include <cmath>

#include "tvdn_common.hpp"

namespace tvdn {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

template <typename T>
static T delta_in_dtype(const double s[3])
{
    // the reference divides its two dtype-width sums in the array dtype (utils.pyx:125)
    return (T)s[1] / (T)s[2];
}

static int run_impl(const tvdn_run_args *a)
{
    const int nd = a->ndim;
    const size_t item = a->dtype == TVDN_F32 ? 4 : 8;
    size_t n = 1;
    for (int i = 0; i < nd; ++i) n *= (size_t)a->shape[i];
    const size_t bytes = n * item;
    const int n_total = a->n_fista + a->n_plain;
    const bool fista = a->n_fista > 0;

    TVDN_HIP(hipSetDevice(a->device));
    tvdn_ctx *ctx = nullptr;
    int rc = tvdn_ctx_create(&ctx, a->device);
    if (rc) return rc;
    struct CtxGuard { tvdn_ctx *c; ~CtxGuard() { (void)tvdn_ctx_destroy(c); } } guard{ctx};

    const int per_axis = fista ? 3 : 2;
    DevBuf orig, recon[2], S[4][3], ref, sums, mse;
    TVDN_HIP(hipMalloc(&orig.p, bytes));
    for (auto &r : recon) TVDN_HIP(hipMalloc(&r.p, bytes));
    for (int q = 0; q < nd; ++q)
        for (int k = 0; k < per_axis; ++k) {
            TVDN_HIP(hipMalloc(&S[q][k].p, bytes));
            TVDN_HIP(hipMemsetAsync(S[q][k].p, 0, bytes, nullptr));
        }
    TVDN_HIP(hipMalloc(&sums.p, sizeof(double) * 3 * (size_t)(n_total > 0 ? n_total : 1)));
    TVDN_HIP(hipMemsetAsync(sums.p, 0, sizeof(double) * 3 * (size_t)(n_total > 0 ? n_total : 1), nullptr));
    rc = tvdn_copy_to_device(orig.p, a->data, bytes, a->device);
    if (rc) return rc;
    TVDN_HIP(hipMemcpyAsync(recon[0].p, orig.p, bytes, hipMemcpyDeviceToDevice, nullptr));
    const bool want_mse = a->mse_out != nullptr && a->reference != nullptr;
    if (want_mse) {
        TVDN_HIP(hipMalloc(&ref.p, bytes));
        rc = tvdn_copy_to_device(ref.p, a->reference, bytes, a->device);
        if (rc) return rc;
        TVDN_HIP(hipMalloc(&mse.p, sizeof(double) * (size_t)(n_total + 1)));
        TVDN_HIP(hipMemsetAsync(mse.p, 0, sizeof(double) * (size_t)(n_total + 1), nullptr));
        rc = tvdn_sum_square_error(ctx, a->dtype, nd, a->shape, orig.p, ref.p, (double *)mse.p, nullptr);
        if (rc) return rc;
    }

    tvdn_iter_args it;
    std::memset(&it, 0, sizeof it);
    it.dtype = a->dtype;
    it.ndim = nd;
    for (int i = 0; i < nd; ++i) it.shape[i] = a->shape[i];
    it.row_lo = 0;
    it.row_hi = a->shape[0];
    it.lo_mode = TVDN_EDGE_BC;
    it.hi_mode = TVDN_EDGE_BC;
    it.bc_mode = a->bc_mode;
    for (int q = 0; q < nd; ++q) {
        it.clip[q] = a->clip[q];
        it.lambda_mu[q] = a->lambda_mu[q];
    }
    it.orig = orig.p;

    int cur = 0, i_d = 0, i_prev = 1, i_out = 2, i_b = 0, i_bout = 1;
    bool d_form = fista;
    double tk = 1.0, tk_prev_ratio = 0.0;
    int ran = 0;

    auto one = [&](int slot, bool use_fista, double ratio) -> int {
        it.recon_in = recon[cur].p;
        it.recon_out = recon[cur ^ 1].p;
        it.tk = use_fista ? ratio : 0.0;
        it.tk_prev = tk_prev_ratio;
        for (int q = 0; q < nd; ++q) {
            it.b_in[q] = it.d_in[q] = it.dprev_in[q] = nullptr;
            it.b_out[q] = it.d_out[q] = nullptr;
            if (use_fista) {
                it.d_in[q] = S[q][i_d].p; it.dprev_in[q] = S[q][i_prev].p; it.d_out[q] = S[q][i_out].p;
            } else if (d_form) {
                it.d_in[q] = S[q][i_d].p; it.dprev_in[q] = S[q][i_prev].p; it.b_out[q] = S[q][i_out].p;
            } else {
                it.b_in[q] = S[q][i_b].p; it.b_out[q] = S[q][i_bout].p;
            }
        }
        it.mode = use_fista ? TVDN_ITER_FISTA_D : (d_form ? TVDN_ITER_FISTA_D_TO_PLAIN : TVDN_ITER_PLAIN);
        int r = tvdn_iterate_fused(ctx, &it, (double *)sums.p + 3 * (size_t)slot, nullptr);
        if (r) return r;
        cur ^= 1;
        if (use_fista) {
            const int t = i_prev; i_prev = i_d; i_d = i_out; i_out = t;
            tk_prev_ratio = ratio;
        } else if (d_form) {
            i_b = i_out; i_bout = i_prev; d_form = false;
        } else {
            const int t = i_b; i_b = i_bout; i_bout = t;
        }
        ++ran;
        if (want_mse) {
            r = tvdn_sum_square_error(ctx, a->dtype, nd, a->shape, ref.p, recon[cur].p, (double *)mse.p + slot + 1, nullptr);
            if (r) return r;
        }
        return TVDN_OK;
    };

    auto stopped = [&](int slot, bool &stop) -> int {
        stop = false;
        if (!a->use_stop) return TVDN_OK;
        double s[3];
        TVDN_HIP(hipMemcpy(s, (double *)sums.p + 3 * (size_t)slot, sizeof s, hipMemcpyDeviceToHost));
        const double delta = a->dtype == TVDN_F32 ? (double)delta_in_dtype<float>(s) : delta_in_dtype<double>(s);
        stop = delta < a->stop;
        return TVDN_OK;
    };

    for (int i = 0; i < a->n_fista; ++i) {
        // float64 recurrence on the host, exactly cyTVDN.py:153-156
        const double tk_new = (1.0 + std::sqrt(1.0 + 4.0 * (tk * tk))) / 2.0;
        const double ratio = (tk - 1.0) / tk_new;
        tk = tk_new;
        rc = one(i, true, ratio);
        if (rc) return rc;
        bool st;
        rc = stopped(i, st);
        if (rc) return rc;
        if (st) break;
    }
    for (int j = 0; j < a->n_plain; ++j) {
        const int slot = j + a->n_fista;
        rc = one(slot, false, 0.0);
        if (rc) return rc;
        bool st;
        rc = stopped(slot, st);
        if (rc) return rc;
        if (st) break;
    }

    TVDN_HIP(hipDeviceSynchronize());
    rc = tvdn_copy_to_host(a->recon_out, recon[cur].p, bytes, a->device);
    if (rc) return rc;
    if (n_total > 0) TVDN_HIP(hipMemcpy(a->sums_out, sums.p, sizeof(double) * 3 * (size_t)n_total, hipMemcpyDeviceToHost));
    if (want_mse) TVDN_HIP(hipMemcpy(a->mse_out, mse.p, sizeof(double) * (size_t)(n_total + 1), hipMemcpyDeviceToHost));
    if (a->iters_run) *a->iters_run = ran;
    TVDN_HIP(hipDeviceSynchronize());
    return TVDN_OK;
}

}  // namespace tvdn

extern "C" int tvdn_run(const tvdn_run_args *a)
{
    TVDN_REQUIRE(a != nullptr, "args is NULL");
    TVDN_REQUIRE(a->dtype == TVDN_F32 || a->dtype == TVDN_F64, "bad dtype %d", a->dtype);
    TVDN_REQUIRE(a->ndim == 3 || a->ndim == 4, "ndim must be 3 or 4, got %d", a->ndim);
    for (int i = 0; i < a->ndim; ++i) TVDN_REQUIRE(a->shape[i] >= 1, "shape[%d] must be >= 1", i);
    TVDN_REQUIRE(a->n_fista >= 0 && a->n_plain >= 0, "negative iteration count");
    TVDN_REQUIRE(a->data && a->recon_out, "data / recon_out is NULL");
    TVDN_REQUIRE(a->sums_out || a->n_fista + a->n_plain == 0, "sums_out is NULL");
    if (a->bc_mode == TVDN_BC_MIRROR) {
        tvdn::set_error("bc_mode 1 (mirror) reconstruction update reads out of bounds upstream (utils.pyx:117-120): unsupported");
        return TVDN_ERR_UNSUPPORTED;
    }
    TVDN_REQUIRE(a->bc_mode == 0 || a->bc_mode == 2, "bc_mode must be 0 or 2, got %d", a->bc_mode);
    return tvdn::run_impl(a);
}
