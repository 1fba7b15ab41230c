// Measurement aid (not product code): do the runtime's own hipMemset / hipMemcpy reach every byte of a range mapped from several
// hipMemCreate handles?  With handles of ONE size they do; with an odd last handle (ROCm 7.2) the middle of the range reads back 0
// after memset, memsetAsync and even after a kernel filled it, i.e. the copy path looks the address up wrongly.  csrc/tvdn_devmem.hip
// therefore maps equal granules only.   hipcc -O2 --offload-arch=gfx950 tools/ubench/vmm_memset.hip -o tools/ubench/vmm_memset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_fill(unsigned char *p, size_t n, unsigned char v) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = v; }
int main()
{
    CK(hipSetDevice(0));
    for (size_t G : {(size_t)8 << 20, (size_t)2 << 20, (size_t)64 << 20}) {  // n = G, 3 G (equal handles) and 3 G + 2 MiB (an odd last handle)
        for (size_t n : {G, 3 * G, 3 * G + (2 << 20)}) {
            hipMemAllocationProp prop; memset(&prop, 0, sizeof prop);
            prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            char *va = nullptr;
            CK(hipMemAddressReserve((void **)&va, n, G, nullptr, 0));
            std::vector<hipMemGenericAllocationHandle_t> hs;
            size_t off = 0;
            while (off < n) { size_t s = std::min(G, n - off); hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, s, &prop, 0)); CK(hipMemMap(va + off, s, 0, h, 0)); hs.push_back(h); off += s; }
            hipMemAccessDesc acc; memset(&acc, 0, sizeof acc); acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, n, &acc, 1));
            unsigned char b[4] = {9, 9, 9, 9};
            hipError_t e = hipMemset(va, 3, n);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&b[0], va, 1, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b[1], va + n - 1, 1, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b[2], va + n / 2, 1, hipMemcpyDeviceToHost));
            printf("G %zu MiB n %zu: hipMemset -> %s; first %d last %d mid %d;", G >> 20, n, hipGetErrorString(e), b[0], b[1], b[2]);
            e = hipMemsetAsync(va, 5, n, 0);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&b[0], va, 1, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b[1], va + n - 1, 1, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b[2], va + n / 2, 1, hipMemcpyDeviceToHost));
            printf(" hipMemsetAsync -> %s; %d %d %d;", hipGetErrorString(e), b[0], b[1], b[2]);
            hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (unsigned char *)va, n, (unsigned char)7);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&b[0], va, 1, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b[1], va + n - 1, 1, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b[2], va + n / 2, 1, hipMemcpyDeviceToHost));
            printf(" kernel fill; %d %d %d;", b[0], b[1], b[2]);
            // D2D copy across granule borders
            if (n > G) { CK(hipMemset(va, 0, 16)); CK(hipMemcpy(va + G - 8, va + n - 64, 32, hipMemcpyDeviceToDevice)); CK(hipMemcpy(&b[0], va + G + 8, 1, hipMemcpyDeviceToHost)); printf(" d2d across border: %d", b[0]); }
            printf("\n");
            off = 0;
            for (size_t i = 0; i < hs.size(); ++i) { size_t s = std::min(G, n - off); CK(hipMemUnmap(va + off, s)); CK(hipMemRelease(hs[i])); off += s; }
            CK(hipMemAddressFree(va, n));
        }
    }
    return 0;
}
