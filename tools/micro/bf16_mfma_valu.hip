// What does a non-MFMA instruction cost next to back-to-back v_mfma_f32_16x16x32_bf16 (16 matrix-pipe cycles each) on gfx950 --
//   (a) issued by the SAME wave between its MFMAs (NV per MFMA, of kind KIND), one wave per SIMD;
//   (b) issued by the OTHER wave of the SIMD (wave A: MFMAs only, wave B: VALU only), alone and together;
//   (c) both waves of a SIMD running the same MFMA + VALU mix.
// hipcc --offload-arch=gfx950 -O3 -o bf16_mfma_valu.bin bf16_mfma_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// KIND: 0 v_fma_f32, 1 v_cvt_pk_bf16_f32, 2 v_pk_add_f32, 3 v_and_b32, 4 ds_write_b64, 5 ds_read_b128, 6 v_mov_dpp + v_max3
template <int KIND> __device__ __forceinline__ void other(float (&x)[8], int i, char* lds, int lane)
{
    float& v = x[i & 7];
    if constexpr (KIND == 0) v = v * 1.0001f + 0.5f;
    else if constexpr (KIND == 1) { f32x2 p = {v, x[(i + 1) & 7]}; bf16x2 b = __builtin_convertvector(p, bf16x2); v = __builtin_bit_cast(float, b); }
    else if constexpr (KIND == 2) { f32x2 p = {v, x[(i + 4) & 7]}; f32x2 q = {0.5f, 0.25f}; p = p + q; v = p.x; x[(i + 4) & 7] = p.y; }
    else if constexpr (KIND == 3) v = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & 0xffff0000u);
    else if constexpr (KIND == 4) *reinterpret_cast<float2*>(lds + lane * 8 + (i & 7) * 512) = make_float2(v, v);
    else if constexpr (KIND == 5) { float4 r = *reinterpret_cast<const float4*>(lds + lane * 16 + (i & 7) * 1024); v += r.x; }
    else if constexpr (KIND == 6) { float n = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)); v = fmaxf(fmaxf(v, n), 0.f); }
}

// MODE: 0 every wave runs MFMA + NV others per MFMA; 1 waves 0-3 MFMA only, waves 4-7 others only (8-wave blocks);
//       2 as 1 but only the MFMA waves work; 3 as 1 but only the other waves work
template <int NV, int KIND, int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(float* out, unsigned long long* ticks, int iters)
{
    __shared__ __attribute__((aligned(16))) char lds[16384];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < 4096; i += WAVES * 64) reinterpret_cast<float*>(lds)[i] = 1.f + i * 1e-4f;
    __syncthreads();
    f32x4 acc[10];
    for (int t = 0; t < 10; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.f + lane * 0.01f + i;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.f + lane * 0.001f + e); b[e] = (__bf16)(2.f - lane * 0.001f); }
    const bool do_mfma = MODE == 0 || ((MODE == 1 || MODE == 2) && wv < 4);
    const bool do_other = MODE == 0 ? NV > 0 : ((MODE == 1 || MODE == 3) && wv >= 4);
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (do_mfma && (MODE != 0 ? true : true)) {
        if (MODE == 0) {
#pragma unroll 1
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 60; ++m) {
                    acc[m % 10] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 10], 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < NV; ++v) other<KIND>(x, m * NV + v, lds, lane);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < 60; ++m) acc[m % 10] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 10], 0, 0, 0);
            }
        }
    } else if (do_other) {
#pragma unroll 1
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 120; ++m) { other<KIND>(x, m, lds, lane); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int t = 0; t < 10; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * WAVES * 64 + tid] = s;
    if (lane == 0) ticks[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int NV, int KIND, int MODE, int WAVES> void run(float* out, unsigned long long* ticks, const char* what)
{
    const int iters = 500, grid = 256;
    hipLaunchKernelGGL((k<NV, KIND, MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, ticks, 10);
    hipLaunchKernelGGL((k<NV, KIND, MODE, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, ticks, iters);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid * 8);
    (void)hipMemcpy(h.data(), ticks, grid * 8 * 8, hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int b = 0; b < grid; ++b) { for (int w = 0; w < 4; ++w) lo += h[b * 8 + w]; for (int w = 4; w < WAVES; ++w) hi += h[b * 8 + w]; }
    lo /= grid * 4.0; hi = WAVES > 4 ? hi / (grid * (WAVES - 4.0)) : 0;
    if (MODE == 0) printf("%-34s NV=%d kind=%d waves/SIMD=%d: %.1f cycles per MFMA (16 ideal)%s\n", what, NV, KIND, WAVES / 4, lo / (iters * 60.0),
                          WAVES > 4 ? " [second wave the same]" : "");
    else printf("%-34s kind=%d: MFMA waves %.1f cycles per MFMA; other waves %.1f cycles per instruction\n", what, KIND,
                MODE == 3 ? 0.0 : lo / (iters * 60.0), MODE == 2 ? 0.0 : hi / (iters * 120.0));
}

int main()
{
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&ticks, 256 * 8 * 8);
    run<0, 0, 0, 4>(out, ticks, "one wave, MFMAs only");
    run<1, 0, 0, 4>(out, ticks, "one wave, own v_fma"); run<2, 0, 0, 4>(out, ticks, "one wave, own v_fma"); run<3, 0, 0, 4>(out, ticks, "one wave, own v_fma"); run<4, 0, 0, 4>(out, ticks, "one wave, own v_fma");
    run<2, 1, 0, 4>(out, ticks, "one wave, own v_cvt_pk_bf16"); run<2, 2, 0, 4>(out, ticks, "one wave, own v_pk_add_f32"); run<2, 3, 0, 4>(out, ticks, "one wave, own v_and");
    run<1, 4, 0, 4>(out, ticks, "one wave, own ds_write_b64"); run<1, 5, 0, 4>(out, ticks, "one wave, own ds_read_b128"); run<1, 6, 0, 4>(out, ticks, "one wave, own dpp+max3");
    run<0, 0, 0, 8>(out, ticks, "two waves, MFMAs only");
    run<2, 0, 0, 8>(out, ticks, "two waves, each own v_fma"); run<3, 0, 0, 8>(out, ticks, "two waves, each own v_fma"); run<4, 0, 0, 8>(out, ticks, "two waves, each own v_fma");
    run<6, 0, 0, 8>(out, ticks, "two waves, each own v_fma");
    run<0, 0, 2, 8>(out, ticks, "MFMA wave alone");
    run<0, 0, 3, 8>(out, ticks, "v_fma wave alone"); run<0, 0, 1, 8>(out, ticks, "MFMA wave + v_fma wave");
    run<0, 1, 3, 8>(out, ticks, "v_cvt_pk wave alone"); run<0, 1, 1, 8>(out, ticks, "MFMA wave + v_cvt_pk wave");
    run<0, 2, 3, 8>(out, ticks, "v_pk_add wave alone"); run<0, 2, 1, 8>(out, ticks, "MFMA wave + v_pk_add wave");
    run<0, 4, 3, 8>(out, ticks, "ds_write_b64 wave alone"); run<0, 4, 1, 8>(out, ticks, "MFMA wave + ds_write_b64 wave");
    run<0, 5, 3, 8>(out, ticks, "ds_read_b128 wave alone"); run<0, 5, 1, 8>(out, ticks, "MFMA wave + ds_read_b128 wave");
    run<0, 6, 3, 8>(out, ticks, "dpp+max3 wave alone"); run<0, 6, 1, 8>(out, ticks, "MFMA wave + dpp+max3 wave");
    return 0;
}
