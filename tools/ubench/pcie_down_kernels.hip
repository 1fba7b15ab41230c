// Measurement aid (not product code): VERDICT r4 item 4 -- can the streamed engine's downloads reach the link's rate while
// uploads (hipMemcpyAsync) and HBM-bound sweeps run beside them?  Round 4: a copy kernel of 8 workgroups x 256 threads, four
// 16-byte loads in flight per thread, writes page-locked host memory at 42.5 GB/s beside 54 GB/s of uploads; more workgroups
// cost the sweeps a third.  Variants of that kernel here: loads in flight per thread, threads per workgroup, workgroups.
//   hipcc -O2 --offload-arch=gfx950 tools/ubench/pcie_down_kernels.hip -o tools/ubench/pcie_down_kernels -lpthread
//   tools/ubench/pcie_down_kernels [CHUNKS=10]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include <sys/mman.h>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            printf("{\"error\": \"%s: %s\"}\n", #x, hipGetErrorString(e_));    \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

typedef float vec_t __attribute__((ext_vector_type(4)));

// `U` 16-byte loads in flight per thread, `T` threads per workgroup; workgroups loop over 64 KiB x (T / 256) pieces
template <int U, int T, bool NT_STORE>
__global__ void __launch_bounds__(T) down_kernel(vec_t *__restrict__ d, const vec_t *__restrict__ s, size_t n)
{
    const size_t stride = (size_t)gridDim.x * T * U;
    for (size_t i = (size_t)blockIdx.x * T * U + threadIdx.x; i < n; i += stride) {
        vec_t a[U];
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (i + (size_t)k * T < n) a[k] = __builtin_nontemporal_load(s + i + (size_t)k * T);
#pragma unroll
        for (int k = 0; k < U; ++k)
            if (i + (size_t)k * T < n) {
                if (NT_STORE)
                    __builtin_nontemporal_store(a[k], d + i + (size_t)k * T);
                else
                    d[i + (size_t)k * T] = a[k];
            }
    }
}

__global__ void __launch_bounds__(256) scale_kernel(vec_t *x, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] = x[i] * 1.0001f;
}

static char *pinned(size_t bytes)
{
    void *m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) exit(2);
    (void)madvise(m, bytes, MADV_HUGEPAGE);
    std::vector<std::thread> th;
    const size_t piece = (bytes + 15) / 16;
    for (int t = 0; t < 16; ++t)
        th.emplace_back([=] {
            for (size_t o = (size_t)t * piece; o < std::min(bytes, (size_t)(t + 1) * piece); o += 4096) ((volatile char *)m)[o] = 0;
        });
    for (auto &t : th) t.join();
    CK(hipHostRegister(m, bytes, hipHostRegisterDefault));
    return (char *)m;
}

struct Variant {
    const char *name;
    int wgs;
    void (*launch)(int wgs, vec_t *d, const vec_t *s, size_t n, hipStream_t st);
    int dma;  // arrays of a chunk that go down by hipMemcpyAsync on a second stream instead (0: none)
};
template <int U, int T, bool NT>
static void launch(int wgs, vec_t *d, const vec_t *s, size_t n, hipStream_t st)
{
    hipLaunchKernelGGL((down_kernel<U, T, NT>), dim3(wgs), dim3(T), 0, st, d, s, n);
}

int main(int argc, char **argv)
{
    const int NC = argc > 1 ? atoi(argv[1]) : 10;
    const size_t piece = (size_t)256 << 20;  // one row-plane of BASELINE config 5
    const int NU = 9, ND = 8;                // arrays up / down per chunk (round 5: recon stays on the device)
    CK(hipSetDevice(0));
    hipStream_t s_up, s_dn, s_dn2, s_bg;
    int least, greatest;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    CK(hipStreamCreateWithPriority(&s_bg, hipStreamNonBlocking, greatest));
    CK(hipStreamCreateWithFlags(&s_up, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_dn, hipStreamNonBlocking, least));
    CK(hipStreamCreateWithPriority(&s_dn2, hipStreamNonBlocking, least));
    std::vector<char *> hu, hd;
    for (int i = 0; i < NU; ++i) hu.push_back(pinned(piece * 2));
    for (int i = 0; i < ND; ++i) hd.push_back(pinned(piece * 2));
    char *in_box, *out_box, *big;
    CK(hipMalloc(&in_box, piece * NU));
    CK(hipMalloc(&out_box, piece * ND));
    const size_t big_n = ((size_t)8 << 30) / 16;
    CK(hipMalloc(&big, big_n * 16));
    CK(hipMemset(big, 0, big_n * 16));
    CK(hipMemset(out_box, 1, piece * ND));
    const Variant vs[] = {
        {"8 x 256, 4 in flight (round 4)", 8, launch<4, 256, true>},
        {"8 x 256, 8 in flight", 8, launch<8, 256, true>},
        {"8 x 256, 16 in flight", 8, launch<16, 256, true>},
        {"8 x 512, 4 in flight", 8, launch<4, 512, true>},
        {"8 x 1024, 4 in flight", 8, launch<4, 1024, true>},
        {"8 x 1024, 8 in flight", 8, launch<8, 1024, true>},
        {"4 x 1024, 8 in flight", 4, launch<8, 1024, true>},
        {"16 x 256, 4 in flight", 16, launch<4, 256, true>},
        {"8 x 256, 8 in flight, plain stores", 8, launch<8, 256, false>},
        {"2 x 1024, 16 in flight", 2, launch<16, 1024, true>},
        {"8 x 256, 4 in flight + 2 of 8 arrays by hipMemcpyAsync", 8, launch<4, 256, true>, 2},
        {"8 x 256, 4 in flight + 3 of 8 arrays by hipMemcpyAsync", 8, launch<4, 256, true>, 3},
        {"8 x 256, 4 in flight + 4 of 8 arrays by hipMemcpyAsync", 8, launch<4, 256, true>, 4},
        {"all 8 arrays by hipMemcpyAsync", 8, launch<4, 256, true>, 8},
    };
    for (int mode = 0; mode < 3; ++mode) {  // 0: downloads alone; 1: + uploads; 2: + uploads + an HBM-bound kernel stream
        for (const Variant &v : vs) {
            CK(hipDeviceSynchronize());
            hipEvent_t d0, d1, u0, u1, b0, b1;
            for (hipEvent_t *e : {&d0, &d1, &u0, &u1, &b0, &b1}) CK(hipEventCreate(e));
            const double t0 = now();
            CK(hipEventRecord(d0, s_dn));
            CK(hipEventRecord(u0, s_up));
            CK(hipEventRecord(b0, s_bg));
            int n_bg = 0;
            for (int c = 0; c < NC; ++c) {
                if (mode >= 1)
                    for (int i = 0; i < NU; ++i) CK(hipMemcpyAsync(in_box + (size_t)i * piece, hu[i] + (size_t)(c & 1) * piece, piece, hipMemcpyHostToDevice, s_up));
                if (mode >= 2)
                    for (int j = 0; j < 30; ++j, ++n_bg) hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((big_n + 255) / 256)), dim3(256), 0, s_bg, (vec_t *)big, big_n);
                for (int i = 0; i < ND; ++i) {
                    if (i < ND - v.dma)
                        v.launch(v.wgs, (vec_t *)(hd[i] + (size_t)(c & 1) * piece), (const vec_t *)(out_box + (size_t)i * piece), piece / 16, s_dn);
                    else
                        CK(hipMemcpyAsync(hd[i] + (size_t)(c & 1) * piece, out_box + (size_t)i * piece, piece, hipMemcpyDeviceToHost, s_dn2));
                }
            }
            if (v.dma) {  // the downloads end when both streams have
                hipEvent_t j;
                CK(hipEventCreate(&j));
                CK(hipEventRecord(j, s_dn2));
                CK(hipStreamWaitEvent(s_dn, j, 0));
                CK(hipEventDestroy(j));
            }
            CK(hipEventRecord(d1, s_dn));
            CK(hipEventRecord(u1, s_up));
            CK(hipEventRecord(b1, s_bg));
            CK(hipDeviceSynchronize());
            float td, tu, tb;
            CK(hipEventElapsedTime(&td, d0, d1));
            CK(hipEventElapsedTime(&tu, u0, u1));
            CK(hipEventElapsedTime(&tb, b0, b1));
            printf("{\"mode\": \"%s\", \"kernel\": \"%s\", \"d2h_GBps\": %.1f, \"h2d_GBps\": %.1f, \"background_TBps\": %.2f, \"wall_s\": %.3f}\n",
                   mode == 0 ? "down alone" : (mode == 1 ? "down + up" : "down + up + sweeps"), v.name, (double)piece * ND * NC / 1e9 / (td * 1e-3),
                   mode >= 1 ? (double)piece * NU * NC / 1e9 / (tu * 1e-3) : 0.0, mode >= 2 ? (double)n_bg * big_n * 32 / 1e12 / (tb * 1e-3) : 0.0, now() - t0);
            fflush(stdout);
            for (hipEvent_t e : {d0, d1, u0, u1, b0, b1}) CK(hipEventDestroy(e));
        }
    }
    return 0;
}
