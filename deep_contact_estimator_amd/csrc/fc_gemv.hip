// fc_gemv.hip -- the FC layers for up to 32 windows (online mode and batch_size 1 of config/inference_one_seq_params.yaml,
// batch_size 30 of config/test_params.yaml): Linear + bias + ReLU of src/contact_cnn.py:49-55.
//
// At these sizes the work is a stream of the weight matrix (fc.0: 38.8 MB) against a few activation rows, and the
// critical path is the dependent fmaf chain every output owns: 9.4 cycles per link (tools/micro/fma_chain.hip).  Round 2
// walked K as ONE chain of 4736 links (18.5 us for fc.0, the kernel measured 21 us).  With the four-range summation tree
// of fc_tree.h the ranges run side by side:
//   * every workgroup owns 8 output neurons (fc.0: 256 workgroups = one per CU; fc.3: 64);
//   * wave w of its four owns K RANGE w of the tree (fc.0: 1152 / 1152 / 1152 / 1280 links; fc.3: 384 / 768 / 384 / 512) and runs it alone, start
//     to end, with no workgroup barrier: it streams its range of the 8 weight rows and of the activation rows in
//     128-float chunks (coalesced 16-byte loads, DEPTH chunks ahead in registers), passes each chunk through a
//     wave-private LDS image (rows padded by 16 B) and lets its 64 lanes = 8 windows x 8 neurons extend their chains --
//     MW chains per lane for up to 8 MW windows, which for MW > 1 also overlap each other's fmaf latency;
//   * one barrier at the end: range sums -> LDS, wave 0 adds them in the tree's order, + bias, ReLU.
// Inside a range every lane walks K in exactly the order the MFMA kernels feed the matrix pipe (inside 8 consecutive k:
// 0,4,1,5,2,6,3,7), so the results are BIT-IDENTICAL to every other fp32 FC kernel of the library and the online path
// still reproduces dce_infer_sequence bit for bit (tests/test_gpu_parity.py).
#include "dce_kernels.h"
#include "fc_tree.h"

namespace dce {

// Chunks in flight per wave (the register ring is written out for 1, 2 and 4).  4 measured SLOWER than 2 at <= 8 windows
// (fc.0 16.5 vs 15.1 us, 45.1 vs 43.0 us per predict() at one window, profiles/r3c_gemv_depth.txt): the kernel is not short
// of bytes in flight -- a chunk's pass through LDS and its 128 chained fmaf are what a range costs.
#ifndef GV_DEPTH1
#define GV_DEPTH1 2                     // <= 8 windows
#endif
#ifndef GV_DEPTH2
#define GV_DEPTH2 2                     // 9 .. 16 windows
#endif
constexpr int GV_R = 8;                 // neurons per workgroup
constexpr int GV_CH = 128;              // floats of K per chunk
constexpr int GV_LD = GV_CH + 4;        // padded LDS row

template <int MW> constexpr int gv_lds_floats() { return FC_RANGES * (8 + 8 * MW) * GV_LD + FC_RANGES * MW * 64; }

// MW = window groups of 8: a launch covers M <= 8 MW windows.  DEPTH = chunks in flight per wave.
template <int MW, int DEPTH>
__global__ __launch_bounds__(256)
void fc_gemv_kernel(const float* __restrict__ A, const float* __restrict__ W,
                    const float* __restrict__ bias, float* __restrict__ C,
                    int M, int N, int K, int relu)
{
    static_assert(DEPTH == 1 || DEPTH == 2 || DEPTH == 4, "the register ring is written out for one, two or four chunks");
    extern __shared__ __attribute__((aligned(16))) float gv_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = K range of the tree
    const int n0 = blockIdx.x * GV_R;
    const int U = K / GV_CH;
    const int u0 = fc_tree_unit(U, wave), u1 = fc_tree_unit(U, wave + 1);
    float* img = gv_lds + wave * (8 + 8 * MW) * GV_LD;                  // this wave's image: 8 weight rows, then 8 MW activation rows
    // loader role: lane (lrow = lane / 8, lc = lane % 8) moves float4 lc + 8 j (j = 0..3) of row lrow of every chunk
    const int lrow = lane >> 3, lc = lane & 7;
    const float4* wp = reinterpret_cast<const float4*>(W + (size_t)(n0 + lrow) * K) + lc;
    const float4* ap[MW];
#pragma unroll
    for (int g = 0; g < MW; ++g) {
        const int ar = lrow + 8 * g;
        ap[g] = reinterpret_cast<const float4*>(A + (size_t)(ar < M ? ar : M - 1) * K) + lc;
    }
    // compute role: lane (m = lane / 8, r = lane % 8): window m + 8 g (g < MW) x neuron r
    const int m = lane >> 3, r = lane & 7;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;                // chain of window group g (g < MW)
    // A chunk in flight = 4 float4 of the weight row + 4 per window group, in NAMED registers (a struct of arrays
    // handed to a lambda ends up in scratch under hipcc): slot S = w<S>j, a<S>gj.
#define GV_DECL(S) float4 w##S##0, w##S##1, w##S##2, w##S##3, a##S##00, a##S##01, a##S##02, a##S##03, a##S##10, a##S##11, a##S##12, a##S##13, \
                          a##S##20, a##S##21, a##S##22, a##S##23, a##S##30, a##S##31, a##S##32, a##S##33;
#define GV_FETCH_J(S, j, o)                                                                   \
    w##S##j = wp[(o) + 8 * j]; a##S##0##j = ap0[(o) + 8 * j];                                  \
    if constexpr (MW > 1) a##S##1##j = ap1[(o) + 8 * j];                                       \
    if constexpr (MW > 2) { a##S##2##j = ap2[(o) + 8 * j]; a##S##3##j = ap3[(o) + 8 * j]; }
#define GV_FETCH(S, u)                                                                        \
    { const int o_ = ((u) < u1 ? (u) : u1 - 1) * (GV_CH / 4);   /* past the range: re-read its last chunk, unused */ \
      GV_FETCH_J(S, 0, o_) GV_FETCH_J(S, 1, o_) GV_FETCH_J(S, 2, o_) GV_FETCH_J(S, 3, o_) }
#define GV_STORE_J(S, j)                                                                      \
    *reinterpret_cast<float4*>(img + lrow * GV_LD + 4 * (lc + 8 * j)) = w##S##j;              \
    *reinterpret_cast<float4*>(img + (8 + lrow) * GV_LD + 4 * (lc + 8 * j)) = a##S##0##j;     \
    if constexpr (MW > 1) *reinterpret_cast<float4*>(img + (16 + lrow) * GV_LD + 4 * (lc + 8 * j)) = a##S##1##j; \
    if constexpr (MW > 2) { *reinterpret_cast<float4*>(img + (24 + lrow) * GV_LD + 4 * (lc + 8 * j)) = a##S##2##j; \
                            *reinterpret_cast<float4*>(img + (32 + lrow) * GV_LD + 4 * (lc + 8 * j)) = a##S##3##j; }
    auto chain8 = [](float v, const float4 x, const float4 y, const float4 p, const float4 q) {
        v = fmaf(x.x, p.x, v); v = fmaf(y.x, q.x, v);                    // k = 0, 4, 1, 5, 2, 6, 3, 7 of a group of 8
        v = fmaf(x.y, p.y, v); v = fmaf(y.y, q.y, v);
        v = fmaf(x.z, p.z, v); v = fmaf(y.z, q.z, v);
        v = fmaf(x.w, p.w, v); v = fmaf(y.w, q.w, v);
        return v;
    };
    // the 128 k of the chunk in this wave's image: the MW chains of a lane advance side by side, 8 k at a time
    auto compute = [&]() {
        const float4* wl = reinterpret_cast<const float4*>(img + r * GV_LD);
        const float4* al0 = reinterpret_cast<const float4*>(img + (8 + m) * GV_LD);
        const float4* al1 = reinterpret_cast<const float4*>(img + (16 + m) * GV_LD);
        const float4* al2 = reinterpret_cast<const float4*>(img + (24 + m) * GV_LD);
        const float4* al3 = reinterpret_cast<const float4*>(img + (32 + m) * GV_LD);
#pragma unroll
        for (int b = 0; b < GV_CH / 8; ++b) {
            const float4 p = wl[2 * b], q = wl[2 * b + 1];
            acc0 = chain8(acc0, al0[2 * b], al0[2 * b + 1], p, q);
            if constexpr (MW > 1) acc1 = chain8(acc1, al1[2 * b], al1[2 * b + 1], p, q);
            if constexpr (MW > 2) { acc2 = chain8(acc2, al2[2 * b], al2[2 * b + 1], p, q); acc3 = chain8(acc3, al3[2 * b], al3[2 * b + 1], p, q); }
        }
    };
    // chunk u: registers -> this wave's LDS image, slot refilled with chunk u + DEPTH, then the chains.  The image is
    // private to this wave, whose lanes run in lockstep and whose LDS operations complete in order.
#define GV_STEP(S, u)                                                                         \
    { GV_STORE_J(S, 0) GV_STORE_J(S, 1) GV_STORE_J(S, 2) GV_STORE_J(S, 3)                      \
      GV_FETCH(S, (u) + DEPTH)                                                                \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();  \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                   \
      compute();                                                                              \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    const float4 *ap0 = ap[0], *ap1 = ap[MW > 1 ? 1 : 0], *ap2 = ap[MW > 2 ? 2 : 0], *ap3 = ap[MW > 2 ? 3 : 0];
    GV_DECL(A) GV_DECL(B) GV_DECL(C) GV_DECL(D)
    GV_FETCH(A, u0)
    if constexpr (DEPTH >= 2) GV_FETCH(B, u0 + 1)
    if constexpr (DEPTH == 4) { GV_FETCH(C, u0 + 2) GV_FETCH(D, u0 + 3) }
    for (int u = u0; u < u1; u += DEPTH) {
        GV_STEP(A, u)
        if constexpr (DEPTH >= 2) { if (u + 1 < u1) GV_STEP(B, u + 1) }
        if constexpr (DEPTH == 4) { if (u + 2 < u1) GV_STEP(C, u + 2)
                                    if (u + 3 < u1) GV_STEP(D, u + 3) }
    }
#undef GV_STEP
#undef GV_STORE_J
#undef GV_FETCH
#undef GV_FETCH_J
#undef GV_DECL
    const float acc[4] = {acc0, acc1, acc2, acc3};
    // ---- the tree's combine: ((((0 + p0) + p1) + p2) + p3) + bias
    float* part = gv_lds + FC_RANGES * (8 + 8 * MW) * GV_LD;             // [range][group][lane]
#pragma unroll
    for (int g = 0; g < MW; ++g) part[(wave * MW + g) * 64 + lane] = acc[g];
    __syncthreads();
    if (wave == 0) {
        const float bv = bias[n0 + r];
#pragma unroll
        for (int g = 0; g < MW; ++g) {
            float tot = 0.f;
#pragma unroll
            for (int k = 0; k < FC_RANGES; ++k) tot += part[(k * MW + g) * 64 + lane];
            float v = tot + bv;
            if (relu) v = v < 0.f ? 0.f : v;                             // NaN stays NaN, as in fc_gemm.hip
            const int row = m + 8 * g;
            if (row < M) C[(size_t)row * N + n0 + r] = v;
        }
    }
}

hipError_t init_fc_gemv()
{
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemv_kernel<1, GV_DEPTH1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 gv_lds_floats<1>() * (int)sizeof(float))) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemv_kernel<2, GV_DEPTH2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 gv_lds_floats<2>() * (int)sizeof(float))) != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemv_kernel<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               gv_lds_floats<4>() * (int)sizeof(float));
}

hipError_t launch_fc_gemv(const float* A, const float* W, const float* bias, float* C,
                          int64_t M, int N, int K, int relu, hipStream_t st)
{
    if (M <= 0) return hipSuccess;
    if (M > FC_GEMV_MAX_M || N % GV_R != 0 || K % GV_CH != 0) return hipErrorInvalidValue;
    for (int r = 0; r < FC_RANGES; ++r)                                  // every wave needs a non-empty K range
        if (fc_tree_unit(K / GV_CH, r + 1) <= fc_tree_unit(K / GV_CH, r)) return hipErrorInvalidValue;
    const dim3 grid(N / GV_R), block(256);
    plan_note("fc_gemv");
    if (M <= 8)       hipLaunchKernelGGL((fc_gemv_kernel<1, GV_DEPTH1>), grid, block, gv_lds_floats<1>() * sizeof(float), st, A, W, bias, C, (int)M, N, K, relu);
    else if (M <= 16) hipLaunchKernelGGL((fc_gemv_kernel<2, GV_DEPTH2>), grid, block, gv_lds_floats<2>() * sizeof(float), st, A, W, bias, C, (int)M, N, K, relu);
    else              hipLaunchKernelGGL((fc_gemv_kernel<4, 1>), grid, block, gv_lds_floats<4>() * sizeof(float), st, A, W, bias, C, (int)M, N, K, relu);
    return hipGetLastError();
}

}  // namespace dce
