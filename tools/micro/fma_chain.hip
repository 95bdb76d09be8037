// Microbenchmark: cycles per link of a dependent v_fmac_f32 chain (one wave), the critical path of
// fc_gemv_kernel.  Build: hipcc --offload-arch=gfx950 -O3 fma_chain.hip -o fma_chain.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(const float* a, const float* b, float* out, unsigned long long* ticks, int iters)
{
    float x[32], y[32];
    for (int i = 0; i < 32; ++i) { x[i] = a[(threadIdx.x + i) & 63]; y[i] = b[(threadIdx.x * 3 + i) & 63]; }
    float acc = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc = fmaf(x[i], y[i], acc);
        asm volatile("" : "+v"(acc));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) *ticks = t1 - t0;
}
int main()
{
    float *a, *b, *o; unsigned long long* t;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 256); hipMalloc(&t, 8);
    float h[64]; for (int i = 0; i < 64; ++i) h[i] = 1.0f + i * 1e-3f;
    hipMemcpy(a, h, 256, hipMemcpyHostToDevice); hipMemcpy(b, h, 256, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o, t, 4096);
        unsigned long long ticks; hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
        printf("dependent v_fmac chain: %.2f ticks per link (s_memtime/readcyclecounter units)\n", ticks / (4096.0 * 32));
    }
    return 0;
}
