// Microbenchmark: how fast can ONE wave per SIMD issue v_mfma_f32_32x32x2_f32, alone and with
// the LDS / global traffic pattern of the conv main loop?  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: MFMA only, 1: + 1 ds_read_b32 per MFMA (pipelined), 2: + float4 global per 20
__global__ __launch_bounds__(256, 2) void k(const float4* __restrict__ g, float* out, unsigned long long* ticks, int iters)
{
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x16 acc[5];
    for (int t = 0; t < 5; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float b[20], bn[20];
    for (int i = 0; i < 20; ++i) b[i] = lds[lane + 64 * i];
    float4 a = g[lane];
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        float4 an = a;
        if (MODE >= 2) an = g[(it & 63) * 64 + lane];
        const float* lp = lds + lane + (it & 7) * 1280;
#pragma unroll
        for (int i = 0; i < 20; ++i) {
            if (MODE >= 1) bn[i] = lp[64 * i];
            const float av = (i & 3) == 0 ? a.x : (i & 3) == 1 ? a.y : (i & 3) == 2 ? a.z : a.w;
            acc[i % 5] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[i], acc[i % 5], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (MODE >= 1) for (int i = 0; i < 20; ++i) b[i] = bn[i];
        a = an;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int t = 0; t < 5; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(int blocks_per_cu, const float4* g, float* out, unsigned long long* ticks)
{
    const int iters = 2000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 65536, 0, g, out, ticks, 100);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 65536, 0, g, out, ticks, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    const double mfmas = 20.0 * iters;                       // per wave
    const double tf = mfmas * 4096.0 * 4 * grid / (ms * 1e-3) / 1e12;
    printf("mode %d blocks/CU %d: %.1f ticks/MFMA/wave, kernel %.3f ms, %.1f TFLOP/s, ticks/s %.3f GHz\n",
           MODE, blocks_per_cu, avg / mfmas, ms, tf, avg / (ms * 1e-3) / 1e9);
}

int main()
{
    float4* g; float* out; unsigned long long* ticks;
    hipMalloc(&g, 64 * 64 * 16); hipMemset(g, 0, 64 * 64 * 16);
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&ticks, 512 * 8);
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int bpc = 1; bpc <= 2; ++bpc) { run<0>(bpc, g, out, ticks); run<1>(bpc, g, out, ticks); run<2>(bpc, g, out, ticks); }
    return 0;
}
