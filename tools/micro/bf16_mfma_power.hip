// What does the bf16 matrix pipe sustain on this board when NOTHING but v_mfma_f32_16x16x32_bf16 is issued, as a function of the
// operand data?  (DVFS: the chip clocks to its power budget; dense MFMAs on random data pull the clock down.)  Wall clock over a
// long launch; one and two waves per SIMD; operands all zero, one constant pattern, or random bf16 (eight operand pairs in rotation).
// hipcc --offload-arch=gfx950 -O3 -o bf16_mfma_power.bin bf16_mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(const uint4* ops, float* out, int iters)
{
    const int tid = threadIdx.x;
    uint4 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = ops[(i * 2) * 512 + tid]; b[i] = ops[(i * 2 + 1) * 512 + tid]; }
    f32x4 acc[10];
    for (int t = 0; t < 10; ++t) acc[t] = f32x4{0, 0, 0, 0};
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 80; ++m)
            acc[m % 10] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m & 7]), __builtin_bit_cast(bf16x8, b[(m >> 1) & 7]), acc[m % 10], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 10; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int WAVES> void run(const uint4* ops, float* out, const char* what)
{
    const int iters = 40000, grid = 256;                        // ~0.1 s per launch: long enough for the clock to settle
    hipLaunchKernelGGL((k<WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, ops, out, 2000);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, ops, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 80 * WAVES * grid, flops = mfmas * 16 * 16 * 32 * 2;
    const double ns_per_slot = ms * 1e6 / (iters * 80.0 * (WAVES / 4));
    printf("%-28s %d wave(s) per SIMD: %7.1f TFLOP/s = %.3f of 2.5 PF; %.2f ns per MFMA and SIMD = 16 cycles at %.2f GHz\n", what, WAVES / 4, flops / ms / 1e9,
           flops / ms / 1e9 / 2500.0, ns_per_slot, 16.0 / ns_per_slot);
}

int main()
{
    std::vector<unsigned short> h(16 * 512 * 8);
    uint4* ops; float* out;
    (void)hipMalloc(&ops, h.size() * 2); (void)hipMalloc(&out, 256 * 512 * 4);
    for (int kind = 0; kind < 4; ++kind) {
        srand(7);
        for (auto& v : h) {
            if (kind == 0) v = 0;
            else if (kind == 1) v = 0x3f80;                                    // 1.0
            else if (kind == 2) { float f = (rand() / (float)RAND_MAX - 0.5f) * 4.f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }    // random values, like activations / first terms
            else v = (unsigned short)rand();                                   // random bit patterns (may hold Inf / NaN)
        }
        (void)hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const char* what[] = {"operands all zero", "operands all 1.0", "random values in (-2, 2)", "random bit patterns"};
        run<4>(ops, out, what[kind]); run<8>(ops, out, what[kind]);
    }
    return 0;
}
