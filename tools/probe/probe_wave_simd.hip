// Which SIMD does wave w of a 768-thread workgroup (160 KB LDS: one workgroup per CU) run on?  s_getreg HW_ID.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(768) void probe(unsigned *out)
{
    extern __shared__ double lds[];
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 12 + (threadIdx.x >> 6)] = hw;
    if (threadIdx.x == 9999) lds[0] = 1.0;
}
int main()
{
    unsigned *dev, h[12 * 8];
    (void)hipMalloc(&dev, sizeof(h));
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
    hipLaunchKernelGGL(probe, dim3(8), dim3(768), 140 * 1024, 0, dev);
    (void)hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 8; ++b) {
        printf("block %d: wave->simd:", b);
        for (int w = 0; w < 12; ++w) printf(" %u", (h[b * 12 + w] >> 4) & 3);
        printf("   wave slot:");
        for (int w = 0; w < 12; ++w) printf(" %u", h[b * 12 + w] & 15);
        printf("   cu %u se %u\n", (h[b * 12] >> 8) & 15, (h[b * 12] >> 13) & 7);
    }
    return 0;
}
