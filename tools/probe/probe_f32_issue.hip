// build + run (gfx950 box): hipcc --offload-arch=gfx950 -O3 -o probe_f32_issue probe_f32_issue.hip && ./probe_f32_issue
// Issue cost of float32 VALU forms from one wave and from two waves of a SIMD: v_fma_f32, v_pk_fma_f32, v_pk_mul_f32,
// v_mul_f32 + v_fmac_f32 pairs (the rotation recurrence of dft_f32_kernel<..., CHAIN>), 64 independent instructions
// per iteration on 16 destination registers.  Cycles by s_memtime (wave 0 alone is the oldest wave and keeps its
// rate whatever else runs: the SIMD's rate is the span of all waves).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND> __global__ void stream(float *out, int iters, long long *cycles)
{
    const int l = threadIdx.x;
    float a[16]; v2f p[16];
    for (int k = 0; k < 16; ++k) { a[k] = 1.0f + l * 1e-3f + k; p[k] = v2f{a[k], a[k] * 0.5f}; }
    const float s = 0.999f, c = 1e-3f; const v2f ps = {0.999f, 0.998f}, pc = {1e-3f, 2e-3f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(s), "v"(c));
                if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(ps), "v"(pc));
                if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(ps));
                if (KIND == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(s));
                if (KIND == 4) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(s), "v"(c));
                if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "+v"(p[k]) : "v"(ps), "v"(pc));
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int k = 0; k < 16; ++k) acc += a[k] + p[k].x + p[k].y;
    out[blockIdx.x * blockDim.x + l] = acc;
    if ((l & 63) == 0) { cycles[2 * (l >> 6)] = t0; cycles[2 * (l >> 6) + 1] = t1; }
}
int main()
{
    float *out; long long *cyc, h[32];
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 256);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_mul_f32", "v_fmac_f32", "v_pk_fma_f32 op_sel/neg"};
    const int iters = 1000;
    for (int threads : {64, 256, 512, 768, 1024}) {
        for (int kind = 0; kind < 6; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(stream<0>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 1) hipLaunchKernelGGL(stream<1>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 2) hipLaunchKernelGGL(stream<2>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 3) hipLaunchKernelGGL(stream<3>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 4) hipLaunchKernelGGL(stream<4>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                if (kind == 5) hipLaunchKernelGGL(stream<5>, dim3(1), dim3(threads), 0, 0, out, iters, cyc);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, 256, hipMemcpyDeviceToHost);
            long long lo = h[0], hi = h[1];
            for (int w = 0; w < threads / 64; ++w) { if (h[2 * w] < lo) lo = h[2 * w]; if (h[2 * w + 1] > hi) hi = h[2 * w + 1]; }
            printf("%4d lanes (%d wave(s)/SIMD)  %-26s wave 0: %.2f cycles per instruction; all waves: %.2f cycles per instruction per SIMD\n",
                   threads, (threads + 255) / 256, names[kind], (double)(h[1] - h[0]) / (iters * 64.0),
                   (double)(hi - lo) / (iters * 64.0 * ((threads + 255) / 256)));
        }
    }
    return 0;
}
