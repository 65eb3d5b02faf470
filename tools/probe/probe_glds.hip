// build + run (gfx950 box): hipcc --offload-arch=gfx950 -O3 -o probe_glds probe_glds.hip && ./probe_glds
// where does global_load_lds_dwordx4 put its bytes?  (LDS destination base offsets 0, 16, 48, 1024; 64 lanes)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void probe(const unsigned *src, unsigned *dump, int base_bytes, int nlanes)
{
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = 0xABABABABu;
    __syncthreads();
    const int lane = threadIdx.x;
    if (lane < nlanes) {
        const unsigned *g = src + lane * 4;
        const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)((char *)lds + base_bytes));
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) dump[i] = lds[i];
}
int main()
{
    unsigned *src, *dump;
    hipMalloc(&src, 4096); hipMalloc(&dump, 8192);
    std::vector<unsigned> h(1024), o(2048);
    for (int i = 0; i < 1024; ++i) h[i] = i;
    hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
    const int bases[] = {0, 16, 48, 64, 1024, 1040};
    for (int b : bases) for (int nl : {64, 40}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 8192, 0, src, dump, b, nl);
        hipMemcpy(o.data(), dump, 8192, hipMemcpyDeviceToHost);
        int first = -1, count = 0, linear = 1;
        for (int i = 0; i < 2048; ++i) if (o[i] != 0xABABABABu) { if (first < 0) first = i; ++count; }
        for (int i = 0; i < nl * 4; ++i) if (o[b / 4 + i] != (unsigned)i) linear = 0;
        printf("base %4d lanes %2d: first written dword %d (byte %d), %d dwords written, lane-linear at base: %s; ", b, nl, first, first * 4, count, linear ? "yes" : "NO");
        printf("dwords at first: %u %u %u %u %u\n", o[first], o[first + 1], o[first + 2], o[first + 3], o[first + 4]);
    }
    return 0;
}
