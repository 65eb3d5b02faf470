// Which A block does v_mfma_f64_4x4x4_4b read under CBSZ / ABID?  B = identity per block, A encodes (block, i, k):
// D lane (16 i + 4 b + n) then shows the A element that output block b used.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int CBSZ, int ABID>
__global__ void probe(double *out)
{
    const int lane = threadIdx.x;
    const int k = lane >> 4, r = lane & 15;
    const double a = 100.0 * (r >> 2) + 10.0 * (r & 3) + k;       // A lane = 16 k + 4 blk + i
    const double b = ((r & 3) == k) ? 1.0 : 0.0;                   // B lane = 16 k + 4 blk + n: identity
    double d = 0.0;
    d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, d, CBSZ, ABID, 0);
    out[lane] = d;
}

template <int CBSZ, int ABID>
void run(double *dev)
{
    double h[64];
    hipLaunchKernelGGL((probe<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, dev);
    hipMemcpy(h, dev, sizeof(h), hipMemcpyDeviceToHost);
    printf("cbsz %d abid %d: source block of output block 0..3 =", CBSZ, ABID);
    for (int b = 0; b < 4; ++b) {
        // D lane = 16 i + 4 b + n; take i = 1, n = 2: value 100 b' + 10 + 2
        const double v = h[16 * 1 + 4 * b + 2];
        printf(" %d(%g)", (int)(v / 100.0), v);
    }
    printf("\n");
}

int main()
{
    double *dev;
    hipMalloc(&dev, 64 * sizeof(double));
    run<0, 0>(dev);
    run<1, 0>(dev); run<1, 1>(dev);
    run<2, 0>(dev); run<2, 1>(dev); run<2, 2>(dev); run<2, 3>(dev);
    hipFree(dev);
    return 0;
}
