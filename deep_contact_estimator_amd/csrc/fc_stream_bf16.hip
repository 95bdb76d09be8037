// fc_stream_bf16.hip -- fc.0 / fc.3 of the DCE_BF16_FC precision for calls of up to 256 windows (reference src/contact_cnn.py:48-54:
// Linear + ReLU, twice; BASELINE configs[4] at the reference's shipped batch sizes 1 and 30):
//     C[M,N] = act(A[M,K] W[N,K]^T + bias),  A, W bf16, K-contiguous,  fp32 accumulate,  C bf16 (h1) or fp32 (h2)
// At these sizes the layer is the stream of its weights (fc.0: 19.4 MB) past a handful of activation rows; the 64 x 64 tile GEMM the
// mode used here put 32 workgroups on that stream (44 us + 21 us per call whatever the batch; it still serves 257 .. 511 windows).
// Here one workgroup owns 16 output
// features and its eight waves deal the K-steps (32 k) out among themselves: per step a lane loads 16 bytes of one weight row
// (operand A of v_mfma_f32_16x16x32_bf16: lane (i, g) = row n0 + i, k = 8 g ..) and 16 bytes of each 16-row block of activations
// (operand B: lane (j, g) = row 16 mt + j), one MFMA per block; nothing goes through LDS until the eight partial tiles are added,
// in wave order, and leave with bias and ReLU.  Above 64 windows the grid grows a second dimension -- one workgroup per 16 features AND
// per 64 windows, each streaming its weight rows again (from L2 / the Infinity Cache: the four-block launch at 256 windows keeps 512
// workgroups busy) -- where a sixteen-block workgroup (128 KB of partial tiles) measured slower than the tile GEMM.  Every output is the same chain of operations whatever M is (the K-steps of a wave in
// order, then the waves in order): a window's h1 / h2 do not depend on how many windows share the call -- the online pushes give the
// bits of the sequence call (tests/test_gpu_parity.py::test_online_mode_bf16_fc).
#include "dce_kernels.h"

namespace dce {

namespace {

typedef __bf16 sb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float sb_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short sb_f32_to_bf16(float f)
{   // round-to-nearest-even; NaN stays NaN (quiet)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

constexpr int SB_WAVES = 8;

// MT: 16-row blocks of activations a workgroup carries (M <= 16 MT)
template <int MT, bool OUT_BF16>
__global__ __launch_bounds__(64 * SB_WAVES)
void fc_stream_bf16_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W, const float* __restrict__ bias,
                           void* __restrict__ Cv, int M, int N, int K, int relu)
{
    extern __shared__ __attribute__((aligned(16))) float sb_part[];     // [wave][mt][lane][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16, mb = blockIdx.y * (16 * MT);         // blockIdx.y: which block of 16 MT windows (65 .. 256 windows: 2 .. 4 of them)
    const uint4* wrow = reinterpret_cast<const uint4*>(W + (size_t)(n0 + i) * K) + g;          // + 4 per K-step
    const uint4* arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = mb + 16 * mt + i < M ? mb + 16 * mt + i : M - 1;   // rows past M re-read the last one (never stored)
        arow[mt] = reinterpret_cast<const uint4*>(A + (size_t)m * K) + g;
    }
    sb_f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = sb_f32x4{0.f, 0.f, 0.f, 0.f};
    const int steps = K / 32;
    constexpr int UN = 4;                                                // K-steps whose loads are in flight together
    int s = wv;
    for (; s + (UN - 1) * SB_WAVES < steps; s += UN * SB_WAVES) {
        uint4 wf[UN], af[UN][MT];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            wf[u] = wrow[4 * (s + u * SB_WAVES)];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) af[u][mt] = arow[mt][4 * (s + u * SB_WAVES)];
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sb_bf16x8, wf[u]), __builtin_bit_cast(sb_bf16x8, af[u][mt]), acc[mt], 0, 0, 0);
    }
    for (; s < steps; s += SB_WAVES) {
        const uint4 wf = wrow[4 * s];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sb_bf16x8, wf), __builtin_bit_cast(sb_bf16x8, arow[mt][4 * s]), acc[mt], 0, 0, 0);
    }
    // D layout of the 16 x 16 tile: lane (j = lane & 15, g): acc[r] = D[row 4 g + r][column j] -> feature n0 + 4 g + r of window 16 mt + j
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) *reinterpret_cast<sb_f32x4*>(sb_part + ((wv * MT + mt) * 64 + lane) * 4) = acc[mt];
    __syncthreads();
    for (int o = tid; o < 256 * MT; o += 64 * SB_WAVES) {
        const int mt = o >> 8, l = (o & 255) >> 2, r = o & 3;
        float v = sb_part[((0 * MT + mt) * 64 + l) * 4 + r];
#pragma unroll
        for (int w = 1; w < SB_WAVES; ++w) v += sb_part[((w * MT + mt) * 64 + l) * 4 + r];
        const int m = mb + 16 * mt + (l & 15), n = n0 + 4 * (l >> 4) + r;
        v += bias[n];
        if (relu) v = v < 0.f ? 0.f : v;                                 // keeps NaN like torch
        if (m < M) {
            if constexpr (OUT_BF16) static_cast<unsigned short*>(Cv)[(size_t)m * N + n] = sb_f32_to_bf16(v);
            else static_cast<float*>(Cv)[(size_t)m * N + n] = v;
        }
    }
}

template <int MT, bool OUT_BF16>
hipError_t launch_sb(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int relu, hipStream_t st)
{
    const size_t lds = (size_t)SB_WAVES * MT * 64 * 16;
    hipLaunchKernelGGL((fc_stream_bf16_kernel<MT, OUT_BF16>), dim3(N / 16, (M + 16 * MT - 1) / (16 * MT)), dim3(64 * SB_WAVES), lds, st,
                       static_cast<const unsigned short*>(A), static_cast<const unsigned short*>(W), bias, C, M, N, K, relu);
    return hipGetLastError();
}

}  // namespace

bool fc_stream_bf16_ok(int64_t M, int N, int K) { return M >= 1 && M <= 256 && N % 16 == 0 && K % 32 == 0 && K >= 32 * SB_WAVES; }

hipError_t launch_fc_stream_bf16(const void* A, const void* W, const float* bias, void* C, int out_bf16, int64_t M, int N, int K, int relu, hipStream_t st)
{
    if (!fc_stream_bf16_ok(M, N, K)) return hipErrorInvalidValue;
    plan_note("fc_stream_bf16");
    const int m = (int)M;
    if (m <= 16) return out_bf16 ? launch_sb<1, true>(A, W, bias, C, m, N, K, relu, st) : launch_sb<1, false>(A, W, bias, C, m, N, K, relu, st);
    return out_bf16 ? launch_sb<4, true>(A, W, bias, C, m, N, K, relu, st) : launch_sb<4, false>(A, W, bias, C, m, N, K, relu, st);
}

}  // namespace dce
