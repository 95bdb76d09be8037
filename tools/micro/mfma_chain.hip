// Microbenchmark: issue interval of DEPENDENT fp32 MFMAs (the same accumulator as C and D), which bounds
// a GEMM whose K chain must stay one ordered fma chain per output (fc.0: 4736 links) at small M.
//   NACC independent accumulators per wave, 1 wave per SIMD (256 workgroups of 256 threads).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_chain.bin mfma_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool BIG>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0)
{
    const int tid = threadIdx.x;
    float a = a0 + tid * 1e-7f, b = b0;
    if (BIG) {
        f32x16 acc[NACC];
        for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        float s = 0.f;
        for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        out[blockIdx.x * 256 + tid] = s;
    } else {
        f32x4 acc[NACC];
        for (int t = 0; t < NACC; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
#pragma unroll 1
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        float s = 0.f;
        for (int t = 0; t < NACC; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
        out[blockIdx.x * 256 + tid] = s;
    }
}

template <int NACC, bool BIG> void run(float* out, int wgs_per_cu)
{
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(grid), dim3(256), 0, 0, out, 100, 1.0f, 1e-3f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BIG>), dim3(grid), dim3(256), 0, 0, out, iters, 1.0f, 1e-3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = 8.0 * iters * NACC;                     // MFMAs per wave
    printf("%s  %d acc/wave, %d wave(s)/SIMD: %.2f ns per MFMA per wave = %.1f cycles at 2.4 GHz  (k links per us per chain: %.0f)\n",
           BIG ? "32x32x2 " : "16x16x4 ", NACC, wgs_per_cu, ms * 1e6 / n, ms * 1e6 / n * 2.4,
           (BIG ? 2.0 : 4.0) * 8.0 * iters / (ms * 1e3));
}

int main()
{
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<1, false>(out, w); run<2, false>(out, w); run<4, false>(out, w);
        run<1, true>(out, w); run<2, true>(out, w);
    }
    return 0;
}
