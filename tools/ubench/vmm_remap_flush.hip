// Measurement aid (not product code): tools/ubench/vmm_remap_check.hip showed that after hipMemUnmap + hipMemMap of another
// handle at the same address the GPU goes on using the OLD translation (ROCm 7.2).  Which runtime call, if any, makes it
// drop the stale one?   hipcc -O2 --offload-arch=gfx950 tools/ubench/vmm_remap_flush.hip -o tools/ubench/vmm_remap_flush
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_fill(unsigned char *p, size_t n, unsigned char v) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = v; }
__global__ void k_count(const unsigned char *p, size_t n, unsigned char v, unsigned long long *out) { unsigned long long c = 0; for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c += p[i] == v; atomicAdd(out, c); }
static unsigned long long *g_cnt;
static int count(const char *p, size_t n, unsigned char v, unsigned long long *res)
{
    CK(hipMemset(g_cnt, 0, 8));
    hipLaunchKernelGGL(k_count, dim3(2048), dim3(256), 0, 0, (const unsigned char *)p, n, v, g_cnt);
    CK(hipMemcpy(res, g_cnt, 8, hipMemcpyDeviceToHost));
    return 0;
}
int main(int argc, char **argv)
{
    const size_t G = (argc > 1 ? (size_t)atoll(argv[1]) : 64) << 20;
    CK(hipSetDevice(0));
    CK(hipMalloc(&g_cnt, 8));
    hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc; memset(&acc, 0, sizeof acc); acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    const char *names[] = {"nothing", "hipDeviceSynchronize", "dummy hipMalloc + hipFree of 4 MiB", "dummy hipMalloc + hipFree of 1 GiB", "a dummy VMM reservation mapped and unmapped elsewhere",
                           "unmap, dummy hipMalloc + hipFree, THEN map", "hipMemAddressFree + hipMemAddressReserve of the same range", "hipDeviceReset-free: hipStreamCreate + destroy"};
    for (int variant = 0; variant < 8; ++variant) {
        char *va = nullptr;
        CK(hipMemAddressReserve((void **)&va, 2 * G, G, nullptr, 0));
        hipMemGenericAllocationHandle_t A, B;
        CK(hipMemCreate(&A, G, &prop, 0)); CK(hipMemCreate(&B, G, &prop, 0));
        CK(hipMemMap(va, G, 0, A, 0)); CK(hipMemMap(va + G, G, 0, B, 0)); CK(hipMemSetAccess(va, 2 * G, &acc, 1));
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned char *)va, G, (unsigned char)0xA1);
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (unsigned char *)va + G, G, (unsigned char)0xB1);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(va, G)); CK(hipMemUnmap(va + G, G));
        if (variant == 5) { void *d; CK(hipMalloc(&d, 4 << 20)); CK(hipFree(d)); }
        if (variant == 6) { CK(hipMemAddressFree(va, 2 * G)); char *v3 = nullptr; CK(hipMemAddressReserve((void **)&v3, 2 * G, G, va, 0)); if (v3 != va) printf("   (got %p instead of %p)\n", (void *)v3, (void *)va); va = v3; }
        CK(hipMemMap(va, G, 0, B, 0)); CK(hipMemMap(va + G, G, 0, A, 0)); CK(hipMemSetAccess(va, 2 * G, &acc, 1));
        if (variant == 1) CK(hipDeviceSynchronize());
        if (variant == 2) { void *d; CK(hipMalloc(&d, 4 << 20)); CK(hipFree(d)); }
        if (variant == 3) { void *d; CK(hipMalloc(&d, (size_t)1 << 30)); CK(hipFree(d)); }
        if (variant == 4) { char *w = nullptr; hipMemGenericAllocationHandle_t Cx; CK(hipMemAddressReserve((void **)&w, G, G, nullptr, 0)); CK(hipMemCreate(&Cx, G, &prop, 0)); CK(hipMemMap(w, G, 0, Cx, 0)); CK(hipMemSetAccess(w, G, &acc, 1)); CK(hipMemUnmap(w, G)); CK(hipMemRelease(Cx)); CK(hipMemAddressFree(w, G)); }
        if (variant == 7) { hipStream_t s; CK(hipStreamCreate(&s)); CK(hipStreamDestroy(s)); }
        unsigned long long c0, c1;
        if (count(va, G, 0xB1, &c0) || count(va + G, G, 0xA1, &c1)) return 1;
        printf("%-62s: after the swap the kernel sees %5.1f %% of slot 0 and %5.1f %% of slot 1 through the NEW mapping\n", names[variant], 100.0 * c0 / G, 100.0 * c1 / G);
        CK(hipDeviceSynchronize());
        CK(hipMemUnmap(va, G)); CK(hipMemUnmap(va + G, G)); CK(hipMemRelease(A)); CK(hipMemRelease(B));
        // leave the reservation alone (never reuse these addresses in this process)
    }
    return 0;
}
