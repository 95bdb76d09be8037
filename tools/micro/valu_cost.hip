// What does a VALU / LDS instruction cost next to back-to-back v_mfma_f32_16x16x4_f32 from ONE wave?
// Pattern per tile: NV VALU ops (clustered or spread) + 8 MFMAs.  1 block (4 waves) per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, bool SPREAD, bool LDSRD>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* ticks, int iters)
{
    __shared__ float lds[4096];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += 256) lds[i] = 1.f + i * 1e-4f;
    __syncthreads();
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.f + lane * 0.01f + i;
    float a = 1.f + lane, b = 2.f - lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int tile = 0; tile < 5; ++tile) {
            if (LDSRD) { x[7] += lds[(lane * 2 + tile * 128 + (it & 7) * 640) & 4095]; }
            if (!SPREAD) {
#pragma unroll
                for (int v = 0; v < NV; ++v) x[v & 7] = x[v & 7] * 1.0001f + 0.5f;
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m], 0, 0, 0);
                if (SPREAD && m < NV) { x[m & 7] = x[m & 7] * 1.0001f + 0.5f; __builtin_amdgcn_sched_barrier(0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NV, bool SPREAD, bool LDSRD> void run(float* out, unsigned long long* ticks)
{
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL((k<NV, SPREAD, LDSRD>), dim3(grid), dim3(256), 0, 0, out, ticks, 10);
    hipLaunchKernelGGL((k<NV, SPREAD, LDSRD>), dim3(grid), dim3(256), 0, 0, out, ticks, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    (void)hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= grid;
    printf("NV=%d %s lds=%d: %.1f ticks per tile (8 MFMAs = 256 ideal)\n", NV, SPREAD ? "spread " : "cluster", (int)LDSRD, avg / (iters * 5.0));
}

int main()
{
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&ticks, 256 * 8);
    run<0, false, false>(out, ticks);
    run<2, false, false>(out, ticks); run<4, false, false>(out, ticks); run<6, false, false>(out, ticks); run<8, false, false>(out, ticks);
    run<2, true, false>(out, ticks); run<4, true, false>(out, ticks); run<6, true, false>(out, ticks); run<8, true, false>(out, ticks);
    run<0, false, true>(out, ticks); run<4, false, true>(out, ticks); run<4, true, true>(out, ticks);
    return 0;
}
