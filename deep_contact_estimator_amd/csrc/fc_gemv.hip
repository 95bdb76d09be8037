// fc_gemv.hip -- the FC layers for a handful of windows (online mode, batch_size 1 of
// config/inference_one_seq_params.yaml): Linear + bias + ReLU of src/contact_cnn.py:49-55 for M <= 32 -- since the MFMA
// chain kernel (fc_gemm_chain.hip) took over from 9 windows, used for M <= 8 and for the <= 8-row remainders of row cuts.
//
// At M = 1 the 128x128 / 64x64 MFMA GEMM tiles of fc_gemm.hip leave 224+ of the 256 CUs idle and
// each active block walks its K loop alone (fc.0: 110 us on 32 workgroups).  The work is a stream
// of the weight matrix (fc.0: 38.8 MB) against a few activation rows, so here
//   * every workgroup owns 8 output neurons (fc.0: 256 workgroups = one per CU; fc.3: 64) and all
//     256 threads stream those 8 weight rows, coalesced, 128 floats of K per chunk, keeping
//     DEPTH (4 or 8) chunks per thread in flight in registers (up to 64 KB in flight per CU: the
//     stream runs at memory latency x depth, not at one K tile per barrier);
//   * chunks pass through a double-buffered LDS image (rows padded by 16 B: the 8 rows read at one
//     column fall into distinct banks) to the compute waves, whose 64 lanes are (window, neuron)
//     pairs (8 windows per wave).
// The critical path is the dependent v_fmac chain itself: 9.4 cycles per link (tools/micro/fma_chain.hip),
// 4736 links = 18.5 us for fc.0 -- the kernel measures 21 us.
//
// Results are BIT-IDENTICAL to fc_gemm_kernel: an fp32 MFMA accumulates its K elements as an
// ordered fmaf chain, so each lane walks K in exactly the order that kernel feeds the matrix pipe --
// inside every 8 consecutive K elements the order is 0,4,1,5,2,6,3,7 (32x32x2 MFMA u takes float
// u of the two 16-B slots of a lane pair) -- starting from 0 and adding the bias last.  The online
// path therefore still reproduces dce_infer_sequence bit for bit (tests/test_gpu_parity.py).
#include "dce_kernels.h"

namespace dce {

constexpr int GV_R = 8;                 // neurons per workgroup
constexpr int GV_CH = 128;              // floats of K per chunk: 8 rows x 32 float4 = one per thread
constexpr int GV_LD = GV_CH + 4;        // padded LDS row

// MW = compute waves: wave w < MW owns windows 8w..8w+7 (its 64 lanes = 8 windows x 8 neurons), so a
// launch covers M <= 8*MW windows.  DEPTH = chunks in flight per thread (1 + MW float4 each).
template <int MW, int DEPTH>   // MW in {1, 2, 4}
__global__ __launch_bounds__(256)
void fc_gemv_kernel(const float* __restrict__ A, const float* __restrict__ W,
                    const float* __restrict__ bias, float* __restrict__ C,
                    int M, int N, int K, int relu)
{
    static_assert(DEPTH == 8 || DEPTH == 4, "the ring is written out for 4 or 8 slots");
    __shared__ __attribute__((aligned(16))) float ws[2][GV_R][GV_LD];
    __shared__ __attribute__((aligned(16))) float as[2][8 * MW][GV_LD];
    const int tid = threadIdx.x, row = tid >> 5, c4 = tid & 31;
    const int n0 = blockIdx.x * GV_R;
    const float4* wp = reinterpret_cast<const float4*>(W + (size_t)(n0 + row) * K) + c4;
    // (named A pointers / ring registers, no arrays: hipcc keeps an indexed ring in scratch)
    auto arow = [&](int j) {
        const int ar = row + 8 * j;
        return reinterpret_cast<const float4*>(A + (size_t)(ar < M ? ar : M - 1) * K) + c4;
    };
    const float4 *ap0 = arow(0), *ap1 = arow(MW > 1 ? 1 : 0), *ap2 = arow(MW > 2 ? 2 : 0), *ap3 = arow(MW > 2 ? 3 : 0);
    const int nch = K / GV_CH;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = 8 * wave + (lane >> 3), r = lane & 7;
    float acc = 0.f;
    auto fetch = [&](float4& w, float4& a0, float4& a1, float4& a2, float4& a3, int c) {
        const int cc = (c < nch ? c : nch - 1) * (GV_CH / 4); // past the end: re-read the last chunk, unused
        w = wp[cc];
        a0 = ap0[cc];
        if constexpr (MW > 1) a1 = ap1[cc];
        if constexpr (MW > 2) { a2 = ap2[cc]; a3 = ap3[cc]; }
    };
    // one ring slot: hand chunk c to LDS buffer `buf`, refill the slot with chunk c + DEPTH, and let
    // the compute waves extend their 64 chains by 128 K elements.
    auto step = [&](float4& w, float4& a0, float4& a1, float4& a2, float4& a3, int c, int buf) {
        *reinterpret_cast<float4*>(&ws[buf][row][4 * c4]) = w;
        *reinterpret_cast<float4*>(&as[buf][row][4 * c4]) = a0;
        if constexpr (MW > 1) *reinterpret_cast<float4*>(&as[buf][row + 8][4 * c4]) = a1;
        if constexpr (MW > 2) {
            *reinterpret_cast<float4*>(&as[buf][row + 16][4 * c4]) = a2;
            *reinterpret_cast<float4*>(&as[buf][row + 24][4 * c4]) = a3;
        }
        fetch(w, a0, a1, a2, a3, c + DEPTH);
        __syncthreads();                                     // chunk c visible; buffer c-1 is free again
        if (wave < MW && c < nch) {                          // loader-only waves run ahead to the next barrier
            const float4* wl = reinterpret_cast<const float4*>(ws[buf][r]);
            const float4* al = reinterpret_cast<const float4*>(as[buf][m]);
            // 4 blocks of 32 K elements; block b+1 is read from LDS while the 32 dependent fmaf of
            // block b run (the chain, ~5 cycles per link, is the critical path of this kernel)
            float4 bw[2][8], ba[2][8];
#pragma unroll
            for (int j = 0; j < 8; ++j) { bw[0][j] = wl[j]; ba[0][j] = al[j]; }
#pragma unroll
            for (int b = 0; b < GV_CH / 32; ++b) {
                if (b + 1 < GV_CH / 32) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { bw[(b + 1) & 1][j] = wl[8 * (b + 1) + j]; ba[(b + 1) & 1][j] = al[8 * (b + 1) + j]; }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 p = bw[b & 1][2 * j], q = bw[b & 1][2 * j + 1];
                    const float4 x = ba[b & 1][2 * j], y = ba[b & 1][2 * j + 1];
                    acc = fmaf(x.x, p.x, acc); acc = fmaf(y.x, q.x, acc);
                    acc = fmaf(x.y, p.y, acc); acc = fmaf(y.y, q.y, acc);
                    acc = fmaf(x.z, p.z, acc); acc = fmaf(y.z, q.z, acc);
                    acc = fmaf(x.w, p.w, acc); acc = fmaf(y.w, q.w, acc);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    // issue order pinned (oldest slot first): otherwise the loop head has to wait for vmcnt(0)
#define GV_SLOT(i) w##i, a0##i, a1##i, a2##i, a3##i
#define GV_DECL(i) float4 GV_SLOT(i);
    GV_DECL(0) GV_DECL(1) GV_DECL(2) GV_DECL(3) GV_DECL(4) GV_DECL(5) GV_DECL(6) GV_DECL(7)
#define GV_FETCH(i) fetch(GV_SLOT(i), i); __builtin_amdgcn_sched_barrier(0);
    GV_FETCH(0) GV_FETCH(1) GV_FETCH(2) GV_FETCH(3)
    if constexpr (DEPTH == 8) { GV_FETCH(4) GV_FETCH(5) GV_FETCH(6) GV_FETCH(7) }
    for (int c0 = 0; c0 < nch; c0 += DEPTH) {
        step(GV_SLOT(0), c0 + 0, 0); step(GV_SLOT(1), c0 + 1, 1); step(GV_SLOT(2), c0 + 2, 0); step(GV_SLOT(3), c0 + 3, 1);
        if constexpr (DEPTH == 8) {
            step(GV_SLOT(4), c0 + 4, 0); step(GV_SLOT(5), c0 + 5, 1); step(GV_SLOT(6), c0 + 6, 0); step(GV_SLOT(7), c0 + 7, 1);
        }
    }
#undef GV_FETCH
#undef GV_DECL
#undef GV_SLOT
    if (wave < MW && m < M) {
        float v = acc + bias[n0 + r];
        if (relu) v = v < 0.f ? 0.f : v;                     // NaN stays NaN, as in fc_gemm.hip
        C[(size_t)m * N + n0 + r] = v;
    }
}

hipError_t launch_fc_gemv(const float* A, const float* W, const float* bias, float* C,
                          int64_t M, int N, int K, int relu, hipStream_t st)
{
    if (M <= 0) return hipSuccess;
    if (M > FC_GEMV_MAX_M || N % GV_R != 0 || K % GV_CH != 0) return hipErrorInvalidValue;
    const dim3 grid(N / GV_R), block(256);
    plan_note("fc_gemv");
    if (M <= 8)       hipLaunchKernelGGL((fc_gemv_kernel<1, 8>), grid, block, 0, st, A, W, bias, C, (int)M, N, K, relu);
    else if (M <= 16) hipLaunchKernelGGL((fc_gemv_kernel<2, 8>), grid, block, 0, st, A, W, bias, C, (int)M, N, K, relu);
    else              hipLaunchKernelGGL((fc_gemv_kernel<4, 4>), grid, block, 0, st, A, W, bias, C, (int)M, N, K, relu);
    return hipGetLastError();
}

}  // namespace dce
