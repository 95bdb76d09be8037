// Microbenchmark of the Winograd main loop (wino_mfma from conv_wino.hip) for ONE block per CU:
// ticks per K-step (40 MFMAs = 1280 cycles at the matrix-pipe rate).
#include "../../deep_contact_estimator_amd/csrc/conv_wino.hip"
#include <cstdio>
#include <vector>
using namespace dce;

template <int RS, int STEPS>
__global__ __launch_bounds__(256, 2) void kloop(const float* wpack, float* out, unsigned long long* ticks, int reps)
{
    extern __shared__ __attribute__((aligned(16))) float act[];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < WACT_FLOATS; i += 256) act[i] = 1.0f + (i % 97) * 1e-3f;
    __syncthreads();
    f32x4 acc[MT][NTW][4];
    for (int a = 0; a < MT; ++a) for (int b = 0; b < NTW; ++b) for (int c = 0; c < 4; ++c) acc[a][b][c] = f32x4{0, 0, 0, 0};
    int boff[NTW];
    col_offsets<TP2, WS2>(0, lane & 15, boff);
    const float4* ap = reinterpret_cast<const float4*>(wpack) + wv * (STEPS * 128) + 2 * lane;
    const float* xrow = act + (lane >> 4) * RS;
    A8 a = load_a8(ap, 0);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) wino_mfma<RS, STEPS>(xrow, boff, ap, a, acc);
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int a2 = 0; a2 < MT; ++a2) for (int b = 0; b < NTW; ++b) for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) s += acc[a2][b][c][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

int main()
{
    float* w; float* out; unsigned long long* ticks;
    hipMalloc(&w, 4 * 32 * 128 * 16); hipMemset(w, 0, 4 * 32 * 128 * 16);
    hipMalloc(&out, 512 * 256 * 4); hipMalloc(&ticks, 512 * 8);
    auto kern = kloop<RS2, 32>;
    for (int lds : {100 * 1024, 80 * 1024}) {
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        const int bpc = lds > 81920 ? 1 : 2, grid = 256 * bpc, reps = 20;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, w, out, ticks, 2);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, w, out, ticks, reps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid);
        (void)hipMemcpy(h.data(), ticks, grid * 8, hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= grid;
        printf("blocks/CU %d: %.0f ticks per K-step (ideal 1280 x %d), %.3f ms, MFMA-pipe use %.1f %%\n",
               bpc, avg / (reps * 32.0), bpc, ms, 100.0 * 1280.0 * bpc / (avg / (reps * 32.0)));
    }
    return 0;
}
