// Three questions about v_mfma_f32_16x16x32_f16 on gfx950, asked before building the two-term fp16 form of the conv stack / fc.0:
//   (1) what does the matrix pipe sustain on fp16 operands (random values) next to bf16 -- same issue rate, but wider multipliers: does
//       the board clock lower?                        (wall clock over long launches, as bf16_mfma_power.hip)
//   (2) are SUBNORMAL fp16 operands honoured or flushed to zero?   (a 2^-20 x 2^10 product; a subnormal x subnormal sum)
//   (3) is a product of two fp16 terms exact in the fp32 accumulator, and do v_cvt_pk_f16_f32 / v_cvt_f32_f16 round-trip as the split needs
// hipcc --offload-arch=gfx950 -O3 -o f16_mfma_probe.bin f16_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int WAVES, bool F16>
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
        for (int m = 0; m < 80; ++m) {
            if constexpr (F16) acc[m % 10] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a[m & 7]), __builtin_bit_cast(f16x8, b[(m >> 1) & 7]), acc[m % 10], 0, 0, 0);
            else acc[m % 10] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[m & 7]), __builtin_bit_cast(bf16x8, b[(m >> 1) & 7]), acc[m % 10], 0, 0, 0);
        }
    }
    float s = 0;
    for (int t = 0; t < 10; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int WAVES, bool F16> void run(const uint4* ops, float* out, const char* what)
{
    const int iters = 40000, grid = 256;
    hipLaunchKernelGGL((k<WAVES, F16>), dim3(grid), dim3(WAVES * 64), 0, 0, ops, out, 2000);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<WAVES, F16>), dim3(grid), dim3(WAVES * 64), 0, 0, ops, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)iters * 80 * WAVES * grid, flops = mfmas * 16 * 16 * 32 * 2;
    const double ns_per_slot = ms * 1e6 / (iters * 80.0 * (WAVES / 4));
    printf("%-5s %-34s %d wave(s) per SIMD: %7.1f TFLOP/s = %.3f of 2.5 PF; 16 cycles at %.2f GHz\n", F16 ? "fp16" : "bf16", what, WAVES / 4, flops / ms / 1e9,
           flops / ms / 1e9 / 2500.0, 16.0 / ns_per_slot);
}

// (2), (3): one wave, one MFMA; lane l holds A[row l & 15][k = 8 (l >> 4) .. + 7], B[k][col l & 15] likewise; D[row 4 (l >> 4) + r][col l & 15]
__global__ void probe(const _Float16* av, const _Float16* bv, float* d)
{
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (lane & 15) == 0 ? av[8 * (lane >> 4) + e] : (_Float16)0; b[e] = (lane & 15) == 0 ? bv[8 * (lane >> 4) + e] : (_Float16)0; }
    f32x4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) d[0] = acc[0];                         // D[0][0] = sum_k av[k] bv[k]
}

__global__ void cvt_probe(const float* x, float* out, int n)
{
    const int i = threadIdx.x;
    if (2 * i + 1 >= n) return;
    const f32x2 v = {x[2 * i], x[2 * i + 1]};
    const f16x2 hh = __builtin_convertvector(v, f16x2);  // expect v_cvt_pk_f16_f32 (RNE)
    const f32x2 back = __builtin_convertvector(hh, f32x2);
    const f32x2 r = v - back;
    const f16x2 h2 = __builtin_convertvector(r, f16x2);
    const f32x2 b2 = __builtin_convertvector(h2, f32x2);
    out[4 * i] = back.x; out[4 * i + 1] = b2.x; out[4 * i + 2] = back.y; out[4 * i + 3] = b2.y;
}

int main()
{
    std::vector<unsigned short> h(16 * 512 * 8);
    uint4* ops; float* out;
    (void)hipMalloc(&ops, h.size() * 2); (void)hipMalloc(&out, 256 * 512 * 4);
    for (int f16 = 0; f16 < 2; ++f16)
        for (int kind = 0; kind < 3; ++kind) {
            srand(7);
            for (auto& v : h) {
                float f = kind == 0 ? 0.f : kind == 1 ? (rand() / (float)RAND_MAX - 0.5f) * 4.f : (rand() / (float)RAND_MAX - 0.5f) * 40000.f;
                if (f16) { _Float16 q = (_Float16)f; memcpy(&v, &q, 2); }
                else { unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
            }
            (void)hipMemcpy(ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
            const char* what[] = {"operands all zero", "random values in (-2, 2)", "random values in (-2e4, 2e4)"};
            if (f16) { run<4, true>(ops, out, what[kind]); run<8, true>(ops, out, what[kind]); run<12, true>(ops, out, what[kind]); }
            else { run<4, false>(ops, out, what[kind]); run<8, false>(ops, out, what[kind]); run<12, false>(ops, out, what[kind]); }
        }
    // (2) subnormals
    _Float16 *av, *bv; float* d;
    (void)hipMalloc(&av, 64); (void)hipMalloc(&bv, 64); (void)hipMalloc(&d, 4);
    struct Case { const char* what; float a[32], b[32]; double expect; };
    auto one = [&](const char* what, std::vector<float> a, std::vector<float> b) {
        _Float16 ha[32] = {}, hb[32] = {};
        double expect = 0;
        for (size_t i = 0; i < a.size(); ++i) { ha[i] = (_Float16)a[i]; hb[i] = (_Float16)b[i]; expect += (double)(float)ha[i] * (double)(float)hb[i]; }
        (void)hipMemcpy(av, ha, 64, hipMemcpyHostToDevice); (void)hipMemcpy(bv, hb, 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, av, bv, d);
        float got; (void)hipMemcpy(&got, d, 4, hipMemcpyDeviceToHost);
        printf("%-64s got %.9g  exact %.9g  %s\n", what, got, expect, (double)got == expect ? "EXACT" : "DIFFERENT");
    };
    one("normal x normal (1.5 x 2.25)", {1.5f}, {2.25f});
    one("subnormal (2^-20) x 2^10", {ldexpf(1.f, -20)}, {1024.f});
    one("subnormal (3 x 2^-24) x 2^12", {3 * ldexpf(1.f, -24)}, {4096.f});
    one("subnormal (2^-18) x subnormal (2^-16)", {ldexpf(1.f, -18)}, {ldexpf(1.f, -16)});
    one("smallest normal (2^-14) x 1", {ldexpf(1.f, -14)}, {1.f});
    one("full-width: 2047 x 2047 + 2045 x 2043", {2047.f, 2045.f}, {2047.f, 2043.f});
    one("65504 x 65504", {65504.f}, {65504.f});
    one("32 products of 11-bit odd values (fp32 accumulate rounds once the sum passes 2^24)", std::vector<float>(32, 2047.f), std::vector<float>(32, 2045.f));
    // (3) conversions
    const int n = 16;
    float hx[n] = {1.0f, 1.0004883f, 1.0002441f, 3.1415927f, -2.7182818f, 65504.f, 65519.9f, 65520.f, 1e-5f, 6.1e-5f, 5.9e-8f, 2.9e-8f, 12345.678f, -0.33333334f, 1e-9f, 0.f};
    float *dx, *dout; (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dout, 2 * n * 4);
    (void)hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, dout, n);
    float ho[2 * n]; (void)hipMemcpy(ho, dout, 2 * n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) {
        const _Float16 c1 = (_Float16)hx[i]; const float r = hx[i] - (float)c1; const _Float16 c2 = (_Float16)r;
        printf("x %.9g: term1 %.9g (host %.9g) term2 %.9g (host %.9g) rest %.3g\n", hx[i], ho[2 * i], (float)c1, ho[2 * i + 1], (float)c2, (double)hx[i] - ho[2 * i] - ho[2 * i + 1]);
    }
    return 0;
}
