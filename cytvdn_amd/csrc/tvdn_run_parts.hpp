// The pieces of a resident tvdn_run (csrc/tvdn_run.hip: the loop; csrc/tvdn_run_entry.hip: the C entry, its checks and the choice
// of engine): device buffers that release themselves, one slab of the cube on one device, the phase clock of TVDN_RUN_TIMING.
#pragma once
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "tvdn_common.hpp"

namespace tvdn {

struct DevBuf {
    void *p = nullptr;
    int device = 0;
    bool owned = true;  // false: the caller's workspace
    size_t bytes = 0;   // size of the allocation (state blocks only)
    bool keep = false;  // a state block that goes to the cache instead of back to the driver
    ~DevBuf()
    {
        if (p && owned) {
            (void)hipSetDevice(device);
            if (keep && bytes)
                state_release(p, bytes, device);
            else
                (void)dev_free(p);
        }
    }
};

template <typename T>
inline T delta_in_dtype(const double s[3])
{
    // the reference divides its two dtype-width sums in the array dtype (utils.pyx:125)
    return (T)s[1] / (T)s[2];
}

// One slab: rows [g0, g1) of the cube plus a halo row on every interior side, resident on one device.
struct Slab {
    int device = 0;
    int64_t g0 = 0, g1 = 0, halo_lo = 0, halo_hi = 0;  // global own rows; halo rows held
    tvdn_ctx *ctx = nullptr;
    hipStream_t main = nullptr, copy = nullptr;
    hipEvent_t edge_done = nullptr, halo_done = nullptr;
    // a run with a stopping rule (run_impl, "speculation by one"): the sums of the last two iterations as the device left them in
    // host memory (two slots of four doubles), and the event behind each slot's last fold
    double *peek = nullptr;
    hipEvent_t summed[2] = {nullptr, nullptr};
    DevBuf state;  // ONE allocation: per axis 2-3 rotating arrays, recon x2, orig
    DevBuf ref, sums, mse;
    tvdn_many_args roles;  // arrays and who plays which role (tvdn_common.hpp roles_bind / roles_advance); .base = the sweep's fixed arguments
    char *orig = nullptr;
    char *recon(int i) const { return (char *)roles.recon[i]; }
    // carve one allocation into the arrays of the state: per axis 2-3 rotating arrays, recon[1], orig, recon[0]
    void assign(char *base, size_t stride, int nd, int per_axis)
    {
        int k = 0;
        for (int q = 0; q < nd; ++q)
            for (int j = 0; j < per_axis; ++j) roles.S[q][j] = base + stride * (size_t)(k++);
        roles.recon[1] = base + stride * (size_t)(k++);
        orig = base + stride * (size_t)(k++);
        roles.recon[0] = base + stride * (size_t)(k++);
        roles.base.orig = orig;
    }
    int64_t rows() const { return halo_lo + (g1 - g0) + halo_hi; }
    int64_t row_lo() const { return halo_lo; }
    int64_t row_hi() const { return halo_lo + (g1 - g0); }
    int main_level = 0, copy_level = 0;
    ~Slab()
    {
        (void)hipSetDevice(device);
        if (edge_done) (void)hipEventDestroy(edge_done);
        if (halo_done) (void)hipEventDestroy(halo_done);
        for (hipEvent_t e : summed)
            if (e) (void)hipEventDestroy(e);
        if (peek) {
            if (main) (void)hipStreamSynchronize(main);  // (an error path: a fold that still writes there)
            (void)hipHostFree(peek);
        }
        // context, streams and the sums' buffer go to the next run of this device (tvdn_run_state.hip kit_release), or are destroyed
        RunKit k;
        k.ctx = ctx;
        k.main = main;
        k.copy = copy;
        k.main_level = main_level;
        k.copy_level = copy_level;
        k.sums = sums.p;
        k.sums_bytes = sums.bytes;
        sums.p = nullptr;
        if (k.ctx) {
            kit_release(device, k);
        } else {
            if (main) (void)hipStreamDestroy(main);
            if (copy) (void)hipStreamDestroy(copy);
            if (k.sums) (void)hipFree(k.sums);
        }
    }
};

inline int64_t edge_block(int64_t own)
{
    const int64_t e = 8;  // a whole march per side: no extra look-ahead rows (engine.edge_block)
    const int64_t cap = (own - 1) / 2;
    return cap < 1 ? 1 : (e < cap ? e : cap);
}

// TVDN_RUN_TIMING=1: where a resident run's wall time goes (stderr; the device is drained at every mark, so the phases
// are honest and the total a little longer than an untimed run's).
struct RunClock {
    bool on = getenv("TVDN_RUN_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char *what)
    {
        if (!on) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "tvdn_run: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// tvdn_run.hip: the resident run itself (one device or a device list)
// `pipeline_with_rule`: a run with a stopping rule may start pipelined (first iterations behind the upload); kRetryPlain comes back
// when the rule turned out to have been met inside those iterations: the caller runs it again without
constexpr int kRetryPlain = 1 << 20;
int run_impl(const tvdn_run_args *a, RunClock &clk, tvdn_run_stats &stats, bool pipeline_with_rule = true);

}  // namespace tvdn
