// Can fp32 VALU FMAs run CONCURRENTLY with fp32 MFMAs (other wave, same SIMD) and add throughput?
// Block = 8 waves: waves 0-3 issue v_mfma_f32_32x32x2_f32 back to back, waves 4-7 issue v_pk_fma_f32
// (or v_fma_f32) chains.  Compare: MFMA waves alone, VALU waves alone, both together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>   // 1: MFMA only, 2: VALU only, 3: both
__global__ __launch_bounds__(512, 2) void k(float* out, int iters)
{
    const int tid = threadIdx.x, wv = tid >> 6;
    float s = 0.f;
    if (wv < 4) {
        if (MODE & 1) {
            f32x16 acc[4];
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            const float a = 1.f + tid * 1e-3f, b = 2.f - tid * 1e-3f;
#pragma unroll 1
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 3], 0, 0, 0);
            }
            for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
        }
    } else {
        if (MODE & 2) {
            v2f x[16];
            for (int i = 0; i < 16; ++i) x[i] = v2f{1.f + i + tid * 1e-3f, 0.5f + i};
            const v2f m = {1.0000001f, 0.9999999f}, c = {1e-7f, -1e-7f};
#pragma unroll 1
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 8; ++r)          // 8 x 16 packed FMAs = 256 lane-FMAs x2 = 512 FLOP/lane... per iter
#pragma unroll
                    for (int i = 0; i < 16; ++i) x[i] = __builtin_elementwise_fma(x[i], m, c);
            }
            for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y;
        }
    }
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE> void run(float* out)
{
    const int iters = 4000, grid = 256;       // one 8-wave block per CU: 1 MFMA wave + 1 VALU wave per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, out, 10);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (MODE & 1) ? 16.0 * 4096 * iters * 4 * grid : 0;              // FLOP
    const double valu = (MODE & 2) ? 8.0 * 16 * 2 * 2 * 64 * iters * 4 * grid : 0;     // FLOP
    printf("mode %d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  total %.1f TF\n", MODE, ms, mfma / ms / 1e9, valu / ms / 1e9, (mfma + valu) / ms / 1e9);
}

int main()
{
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    run<1>(out); run<2>(out); run<3>(out); run<1>(out); run<2>(out); run<3>(out);
    return 0;
}
