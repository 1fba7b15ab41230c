// Measurement aid (not product code): what HBM rate does a pure 10-read / 9-write float4 stream
// reach on this MI355X?  That is the practical ceiling for the fused TV sweep (19 array passes).
//   hipcc -O3 --offload-arch=gfx950 stream_mix.hip -o stream_mix && ./stream_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
struct Ptrs { const f4 *in[10]; f4 *out[9]; };

template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) mix(Ptrs p, long long n4)
{
    const long long step = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += step) {
        f4 v[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            if (NT) v[k] = __builtin_nontemporal_load(p.in[k] + i); else v[k] = p.in[k][i];
        }
        f4 s = v[0];
#pragma unroll
        for (int k = 1; k < NR; ++k) { s += v[k]; }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            f4 o = s; o.x += k;
            if (NT) __builtin_nontemporal_store(o, p.out[k] + i); else p.out[k][i] = o;
        }
    }
}

// block-contiguous variant: each workgroup owns a contiguous span (like a marching tile)
template <int NR, int NW, bool NT>
__global__ void __launch_bounds__(256) mix_span(Ptrs p, long long n4, long long span)
{
    const long long b0 = (long long)blockIdx.x * span;
    for (long long j = threadIdx.x; j < span; j += 256) {
        const long long i = b0 + j;
        if (i >= n4) break;
        f4 v[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            if (NT) v[k] = __builtin_nontemporal_load(p.in[k] + i); else v[k] = p.in[k][i];
        }
        f4 s = v[0];
#pragma unroll
        for (int k = 1; k < NR; ++k) { s += v[k]; }
#pragma unroll
        for (int k = 0; k < NW; ++k) {
            f4 o = s; o.x += k;
            if (NT) __builtin_nontemporal_store(o, p.out[k] + i); else p.out[k][i] = o;
        }
    }
}

template <int NR, int NW, bool NT>
static void run(const char *name, Ptrs p, long long n4, int grid, long long span)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f, tot = 0;
    for (int it = 0; it < 6; ++it) {
        CK(hipEventRecord(e0));
        if (span) hipLaunchKernelGGL((mix_span<NR, NW, NT>), dim3((unsigned)((n4 + span - 1) / span)), dim3(256), 0, 0, p, n4, span);
        else hipLaunchKernelGGL((mix<NR, NW, NT>), dim3(grid), dim3(256), 0, 0, p, n4);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0) { tot += ms; if (ms < best) best = ms; }
    }
    double bytes = (double)n4 * 16 * (NR + NW);
    printf("%-34s NR=%2d NW=%d nt=%d grid=%7d span=%6lld : mean %.3f ms  %.0f GB/s (best %.0f)\n", name, NR, NW, (int)NT, grid, span,
           tot / 5, bytes / (tot / 5 * 1e-3) / 1e9, bytes / (best * 1e-3) / 1e9);
}

int main()
{
    const long long n = 1ll << 30, n4 = n / 4;
    Ptrs p;
    for (int k = 0; k < 10; ++k) { void *q; CK(hipMalloc(&q, n * 4)); CK(hipMemset(q, 0, n * 4)); p.in[k] = (const f4 *)q; }
    for (int k = 0; k < 9; ++k) { void *q; CK(hipMalloc(&q, n * 4)); CK(hipMemset(q, 0, n * 4)); p.out[k] = (f4 *)q; }
    CK(hipDeviceSynchronize());
    run<1, 1, false>("copy", p, n4, 256 * 8, 0);
    run<1, 1, true>("copy nt", p, n4, 256 * 8, 0);
    run<1, 1, false>("copy big grid", p, n4, 1 << 20, 0);
    run<10, 9, false>("mix gridstride 2048", p, n4, 256 * 8, 0);
    run<10, 9, false>("mix gridstride 8192", p, n4, 256 * 32, 0);
    run<10, 9, false>("mix one-elem-per-thread", p, n4, (int)(n4 / 256), 0);
    run<10, 9, true>("mix nt gridstride 2048", p, n4, 256 * 8, 0);
    run<10, 9, true>("mix nt one-elem-per-thread", p, n4, (int)(n4 / 256), 0);
    run<10, 9, false>("mix span 8192", p, n4, 0, 8192);
    run<10, 9, false>("mix span 65536", p, n4, 0, 65536);
    run<10, 9, true>("mix nt span 8192", p, n4, 0, 8192);
    run<10, 5, false>("10R 5W gridstride 8192", p, n4, 256 * 32, 0);
    run<10, 5, false>("10R 5W one-elem-per-thread", p, n4, (int)(n4 / 256), 0);
    run<10, 5, true>("10R 5W nt one-elem-per-thread", p, n4, (int)(n4 / 256), 0);
    run<10, 5, false>("10R 5W span 8192", p, n4, 0, 8192);
    run<6, 5, false>("6R 5W one-elem-per-thread", p, n4, (int)(n4 / 256), 0);
    run<1, 9, false>("1R 9W", p, n4, 256 * 8, 0);
    run<5, 5, false>("5R 5W", p, n4, 256 * 8, 0);
    return 0;
}
