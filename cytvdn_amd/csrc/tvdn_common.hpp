// Shared host/device helpers for libtvdn_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#include "../../include/tvdn.h"

namespace tvdn {

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const char *fmt, ...);

#define TVDN_HIP(expr)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            ::tvdn::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),       \
                              __FILE__, __LINE__);                                         \
            return TVDN_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

#define TVDN_REQUIRE(cond, ...)                \
    do {                                       \
        if (!(cond)) {                         \
            ::tvdn::set_error(__VA_ARGS__);    \
            return TVDN_ERR_INVALID;           \
        }                                      \
    } while (0)

// ---- canonical 4-D geometry ------------------------------------------------------------------
// A 3-D reference array (N0,N1,N2) is handled as (N0,1,N1,N2): the marching axis M stays the
// reference's axis 0, axis A is absent.  Canonical axes: 0=M, 1=A, 2=B, 3=C (C contiguous).
struct Geom {
    long long n[4];   // extents M, A, B, C
    long long st[4];  // element strides
    int nax;          // 3 or 4 regularised axes
    long long total;
};

inline Geom make_geom(int ndim, const int64_t *shape)
{
    Geom g;
    if (ndim == 4) {
        for (int i = 0; i < 4; ++i) g.n[i] = shape[i];
    } else {
        g.n[0] = shape[0];
        g.n[1] = 1;
        g.n[2] = shape[1];
        g.n[3] = shape[2];
    }
    g.st[3] = 1;
    g.st[2] = g.n[3];
    g.st[1] = g.n[3] * g.n[2];
    g.st[0] = g.n[3] * g.n[2] * g.n[1];
    g.nax = ndim;
    g.total = g.n[0] * g.n[1] * g.n[2] * g.n[3];
    return g;
}

// reference axis -> canonical axis
inline int canon_axis(int ndim, int ax) { return (ndim == 4 || ax == 0) ? ax : ax + 1; }

// ---- context ---------------------------------------------------------------------------------
constexpr int kMaxPartialBlocks = 1 << 18;  // per launch
constexpr int kPartialWidth = 4;            // doubles per block

}  // namespace tvdn

struct tvdn_ctx {
    int device;
    double *partials;  // [kMaxPartialBlocks][kPartialWidth]
    bool timing;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> events;  // pending (start, stop) pairs
};

namespace tvdn {

// ---- device-side reduction helpers -----------------------------------------------------------
// Fixed-shape tree: lane shuffles (64-wide wavefront) -> LDS across the block's waves ->
// one partial per block.  A second single-block kernel folds the partials in index order.
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

template <int NV, int BLOCK>
__device__ __forceinline__ void block_store_partials(const double (&v)[NV], double *partials)
{
    constexpr int NW = BLOCK / 64;
    __shared__ double red[NV][NW];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double s = wave_sum(v[q]);
        if (lane == 0) red[q][w] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < NW; ++i) s += red[threadIdx.x][i];
        partials[(size_t)blockIdx.x * kPartialWidth + threadIdx.x] = s;
    }
}

int launch_finalize(tvdn_ctx *ctx, int nblocks, int nv, double *out, hipStream_t s, bool accumulate = false);

}  // namespace tvdn
