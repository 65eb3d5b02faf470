// Lane layout probe of v_mfma_f64_4x4x4_4b on gfx950: D = A*B per block with one-hot A and B.
// Build: hipcc -O2 --offload-arch=gfx950 probe_mfma_f64.hip -o probe_mfma_f64
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(unsigned long long *mask)
{
    const int lane = threadIdx.x;
    for (int x = 0; x < 64; ++x)
        for (int y = 0; y < 64; ++y) {
            double a = lane == x ? 1.0 : 0.0, b = lane == y ? 1.0 : 0.0;
            double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            unsigned long long m = __ballot(d != 0.0);
            if (lane == 0) mask[x * 64 + y] = m;
        }
}

int main()
{
    unsigned long long *d, h[64 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // for every A lane x: which B lanes y pair with it, and where the product lands
    for (int x = 0; x < 64; ++x) {
        printf("A lane %2d:", x);
        for (int y = 0; y < 64; ++y)
            if (h[x * 64 + y]) {
                int n = 0, first = -1;
                for (int l = 0; l < 64; ++l)
                    if (h[x * 64 + y] >> l & 1) { if (first < 0) first = l; ++n; }
                printf(" B%d->D%d%s", y, first, n > 1 ? "+" : "");
            }
        printf("\n");
    }
    return 0;
}
