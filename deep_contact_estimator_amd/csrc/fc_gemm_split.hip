// fc_gemm_split.hip -- the FC layers for 9 .. 64 windows (batch_size 30 of config/test_params.yaml): Linear + bias +
// ReLU of src/contact_cnn.py:49-55 with the four K ranges of the summation tree (fc_tree.h) on four waves.
//
// Up to a few dozen windows an FC layer is bound by the LENGTH of the fmaf chain an output owns, not by throughput.
// The MFMA chain kernel (fc_gemm_chain.hip) walks K as one chain per 16x16 tile: 4736 links at 8.7 cycles = 17 us for
// fc.0, plus its staging.  Here a workgroup owns ONE 16 (windows) x 16 (neurons) tile of C and its four waves own the
// four ranges of the tree: every wave runs its range alone, start to end, on v_mfma_f32_16x16x4_f32 (a dependent chain
// issues back to back at 34.8 cycles per MFMA = 4 links: fc.0's longest range, 1280 links, is 4.8 us), with
//   * no workgroup barrier before the end: a wave streams ITS range of the 16 window rows and 16 weight rows in
//     128-float chunks (whole 128-byte lines per 8 lanes, a chunk ahead in registers) through a wave-private LDS image;
//   * the image stores every 8 consecutive k as [k0 k2 k4 k6 | k1 k3 k5 k7], so that lane (i, g) fetches the two
//     operands of the MFMA pair of a group (k = {0,4,1,5}[g], then +2) with ONE aligned ds_read_b64; rows are padded
//     by 16 B (16 rows x 2 lane groups = 32 distinct 8-byte bank pairs);
//   * at the end the four range sums meet in LDS and wave 0 adds them in the tree's order, + bias, ReLU.
// K order inside a range = the order every other fp32 FC kernel uses, so the results are bit-identical to theirs
// (tests: the CPU fmaf model of the tree, and the batch-size sweep).
#include "dce_kernels.h"
#include "fc_tree.h"

namespace dce {

typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
typedef float sp_f32x2 __attribute__((ext_vector_type(2)));

#ifndef SP_DEPTH
#define SP_DEPTH 1                               // chunks in flight per wave (2 measured slower: 20.3 vs 17.5 us for fc.0 with the first load map)
#endif
constexpr int SP_CH = 128;                       // floats of K per chunk
constexpr int SP_LD = SP_CH + 4;                 // padded image row (floats)
constexpr int SP_ROWS = 32;                      // 16 window rows + 16 weight rows
constexpr int SP_IMG = SP_ROWS * SP_LD;          // floats per wave image
constexpr int SP_LDS_FLOATS = FC_RANGES * SP_IMG + FC_RANGES * 256;

template <int K>
__global__ __launch_bounds__(256)
void fc_split_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                     float* __restrict__ C, int M, int N, int relu)
{
    static_assert(K % SP_CH == 0 && K / SP_CH >= FC_RANGES, "whole chunks");
    extern __shared__ __attribute__((aligned(16))) float sp_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // = K range of the tree
    const int n0 = blockIdx.x * 16, m0 = blockIdx.y * 16;
    constexpr int U = K / SP_CH;
    const int u0 = fc_tree_unit(U, wave), u1 = fc_tree_unit(U, wave + 1);
    float* img = sp_lds + wave * SP_IMG;

    // ---- loader role: lane (l8 = lane / 8, c8 = lane % 8) moves float4 c8 + 8 j (j = 0..3) of image rows l8 + 8 t (t = 0..3;
    //      rows 0..15 = windows m0.., rows 16..31 = neurons n0..) of every chunk: a load instruction covers 8 rows x 128
    //      contiguous bytes -- whole cache lines (32-byte pieces per lane over 32 rows per instruction kept the texture path
    //      busier than the chain: 17.5 us for fc.0 instead of 12)
    const int l8 = lane >> 3, c8 = lane & 7;
    const float4* gp0 = reinterpret_cast<const float4*>(A + (size_t)(m0 + l8 < M ? m0 + l8 : M - 1) * K) + c8;
    const float4* gp1 = reinterpret_cast<const float4*>(A + (size_t)(m0 + l8 + 8 < M ? m0 + l8 + 8 : M - 1) * K) + c8;
    const float4* gp2 = reinterpret_cast<const float4*>(W + (size_t)(n0 + l8) * K) + c8;
    const float4* gp3 = reinterpret_cast<const float4*>(W + (size_t)(n0 + l8 + 8) * K) + c8;
    // float4 f = c8 + 8 j of a row holds k = 4 f .. 4 f + 3: half (f & 1) of the 8-group f / 2.  The image stores a group
    // as [k0 k2 k4 k6 | k1 k3 k5 k7]: (v.x, v.z) goes to float 8 (f/2) + 2 (f&1), (v.y, v.w) four floats behind it.
    float* irow = img + l8 * SP_LD + 8 * (c8 >> 1) + 2 * (c8 & 1);          // row t: + 8 t rows; float4 j: + 32 j floats
#define SP_DECL(S) float4 v##S##00, v##S##01, v##S##02, v##S##03, v##S##10, v##S##11, v##S##12, v##S##13, \
                          v##S##20, v##S##21, v##S##22, v##S##23, v##S##30, v##S##31, v##S##32, v##S##33;
#define SP_FETCH_T(S, t, o) v##S##t##0 = gp##t[(o)]; v##S##t##1 = gp##t[(o) + 8]; v##S##t##2 = gp##t[(o) + 16]; v##S##t##3 = gp##t[(o) + 24];
#define SP_FETCH(S, u)                                                                        \
    { const int o_ = ((u) < u1 ? (u) : u1 - 1) * (SP_CH / 4);      /* past the range: re-read its last chunk, unused */ \
      SP_FETCH_T(S, 0, o_) SP_FETCH_T(S, 1, o_) SP_FETCH_T(S, 2, o_) SP_FETCH_T(S, 3, o_) }
#define SP_STORE_V(S, t, j)                                                                   \
    *reinterpret_cast<sp_f32x2*>(irow + 8 * t * SP_LD + 32 * j)     = sp_f32x2{v##S##t##j.x, v##S##t##j.z}; \
    *reinterpret_cast<sp_f32x2*>(irow + 8 * t * SP_LD + 32 * j + 4) = sp_f32x2{v##S##t##j.y, v##S##t##j.w};
#define SP_STORE_T(S, t) SP_STORE_V(S, t, 0) SP_STORE_V(S, t, 1) SP_STORE_V(S, t, 2) SP_STORE_V(S, t, 3)
#define SP_STORE(S) SP_STORE_T(S, 0) SP_STORE_T(S, 1) SP_STORE_T(S, 2) SP_STORE_T(S, 3)

    // ---- compute role: lane (i = lane % 16, g = lane / 16): operands of group q of a chunk sit at float 8 q + 2 g of
    //      row i (window) and row 16 + i (neuron); D[row = 4 g + r][col = i] = window m0 + 4 g + r, neuron n0 + i
    const int i = lane & 15, g = lane >> 4;
    const float* fa = img + i * SP_LD + 2 * g;
    const float* fb = img + (16 + i) * SP_LD + 2 * g;
    sp_f32x4 acc = {0.f, 0.f, 0.f, 0.f};

    // one chunk of the loop: image <- slot S, slot S <- chunk u + 2 (two chunks stay in flight: the stream is bound by
    // memory latency x bytes in flight per CU), then the chunk's 32 chained MFMAs
#define SP_STEP(S, u)                                                                         \
    { SP_STORE(S)                                                                             \
      SP_FETCH(S, (u) + SP_DEPTH)                                                                    \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();  \
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");                                   \
      sp_f32x2 ra[16], rb[16];                                                                \
      _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                        \
          ra[q] = *reinterpret_cast<const sp_f32x2*>(fa + 8 * q);                             \
          rb[q] = *reinterpret_cast<const sp_f32x2*>(fb + 8 * q); }                           \
      _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                        \
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[q].x, rb[q].x, acc, 0, 0, 0);         /* k = 0, 4, 1, 5 of the group */ \
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[q].y, rb[q].y, acc, 0, 0, 0); }       /* k = 2, 6, 3, 7 */ \
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }   /* every lane has read the image */
#if SP_DEPTH == 2
    SP_DECL(A) SP_DECL(B)
    SP_FETCH(A, u0)
    SP_FETCH(B, u0 + 1)
    for (int u = u0; u < u1; u += 2) {
        SP_STEP(A, u)
        if (u + 1 < u1) SP_STEP(B, u + 1)
    }
#else
    SP_DECL(A)
    SP_FETCH(A, u0)
    for (int u = u0; u < u1; ++u) SP_STEP(A, u)
#endif
#undef SP_STEP
#undef SP_STORE
#undef SP_STORE_T
#undef SP_STORE_V
#undef SP_FETCH
#undef SP_FETCH_T
#undef SP_DECL
    // ---- the tree's combine: ((((0 + p0) + p1) + p2) + p3) + bias
    float* part = sp_lds + FC_RANGES * SP_IMG;                              // [range][r][lane]
#pragma unroll
    for (int r = 0; r < 4; ++r) part[(wave * 4 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
        const float bv = bias[n0 + i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float tot = 0.f;
#pragma unroll
            for (int k = 0; k < FC_RANGES; ++k) tot += part[(k * 4 + r) * 64 + lane];
            float v = tot + bv;
            if (relu) v = v < 0.f ? 0.f : v;                                // NaN stays NaN, as in fc_gemm.hip
            const int row = m0 + 4 * g + r;
            if (row < M) C[(size_t)row * N + n0 + i] = v;
        }
    }
}

hipError_t init_fc_split()
{
    for (const void* k : {reinterpret_cast<const void*>(&fc_split_kernel<FEAT>), reinterpret_cast<const void*>(&fc_split_kernel<FC1>)}) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS_FLOATS * (int)sizeof(float));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

bool fc_split_ok(int64_t M, int N, int K)
{
    const Tuning& tu = tune();
    return M >= tu.split_min && M <= tu.split_max && N % 16 == 0 && (K == FEAT || K == FC1);
}

hipError_t launch_fc_split(const float* A, const float* W, const float* bias, float* C,
                           int64_t M, int N, int K, int relu, hipStream_t st)
{
    if (M <= 0) return hipSuccess;
    if (N % 16 || (K != FEAT && K != FC1) || M > 65535 * 16) return hipErrorInvalidValue;
    const dim3 grid(N / 16, (unsigned)((M + 15) / 16)), block(256);
    plan_note("fc_split16x16");
    if (K == FEAT) hipLaunchKernelGGL((fc_split_kernel<FEAT>), grid, block, SP_LDS_FLOATS * sizeof(float), st, A, W, bias, C, (int)M, N, relu);
    else           hipLaunchKernelGGL((fc_split_kernel<FC1>), grid, block, SP_LDS_FLOATS * sizeof(float), st, A, W, bias, C, (int)M, N, relu);
    return hipGetLastError();
}

}  // namespace dce
