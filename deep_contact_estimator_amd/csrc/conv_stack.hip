// conv_stack.hip -- fused z-score + Conv1d x4 + ReLU + MaxPool1d x2 for gfx950 (MI355X).
//
// Replaces, for the inference path, what PyTorch dispatches for
//   utils/data_handler.py:55-56      per-window z-score (mean, unbiased std over time)
//   src/contact_cnn.py:61            permute (B,150,54) -> (B,54,150)
//   src/contact_cnn.py:10-26,28-44   block1 / block2 (Dropout = identity in eval)
//   src/contact_cnn.py:64            view(B,-1): feat index = c*37 + t
//
// Design (see DESIGN.md "conv stack"):
//  * One workgroup = 4 waves = NW(2) windows; two workgroups are co-resident per CU
//    (<= 80 KiB LDS and <= 256 VGPRs each), so one group's load/z-score/store phases hide
//    under the other's MFMA phases.
//  * Every activation of the two windows lives in ONE LDS buffer in [channel][position]
//    layout; a layer's whole output is held in MFMA accumulators (80 VGPRs/lane) until all
//    waves have finished reading its input, then written back IN PLACE.  HBM sees only the
//    raw sensor rows and the 4736 pooled features per window.
//  * Each conv is an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chains at the
//    fp32 peak rate): M = Cout, N = positions of both windows laid side by side with their
//    zero pads as ordinary columns, K = (tap, cin).  The B operand is read straight from the
//    activation rows (lane = position -> conflict-free ds_read_b32, the +-1 tap shift is an
//    address immediate); the A operand (weights, pre-packed per lane on the host) streams
//    from L2 as one float4 per lane per 4 K-steps.
//  * Bias is the accumulator's initial value; ReLU, zero-padding and MaxPool (adjacent
//    positions = adjacent lanes -> one DPP quad_perm) are fused in the write-back.
//
// Position layouts (p = column index of the implicit GEMM; LDS address = row*S + p + 1):
//   stage 1 (T=150): p = 2 + 152*w + t      pads p = 1,152,153,304      row stride S1 = 307
//   stage 2 (T=75):  q = 2 +  76*w + t      pads q = 1,77,153           row stride S2 = 155
// t = 0 sits on an even column in both, so pooling pairs (2j,2j+1) are lanes (2m,2m+1), and
// pooling maps stage 1 to stage 2 by q = p/2 + 1 for both windows and for the pads alike.
#include "conv_common.h"
#include <cstdlib>

namespace dce {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int S1  = 307;          // odd strides: transposing stores hit 32 distinct banks
constexpr int S2  = 155;
constexpr int ACT_FLOATS = 128 * S2 + 16;             // 19,856 floats = 79,424 B (>= 64*S1+16)
constexpr int NT  = 5;            // 32-column tiles per wave (stage 1: 10 tiles / 2, stage 2: 5)
constexpr int RED_ROW = 56;       // stage-1 rows 56.. are free until conv1 writes back
static_assert((RED_ROW * S1) % 2 == 0 && NW * 4 * 216 <= 8 * S1, "fp64 reduction scratch fits rows 56..63, 8-B aligned");
static_assert(64 * S1 + 16 <= ACT_FLOATS, "stage-1 image must fit");
constexpr int LDS_FLOATS = ACT_FLOATS + 384;          // + the four bias vectors
static_assert(LDS_FLOATS * 4 <= 80 * 1024, "two workgroups per CU");
// ------------------------------------------------------------------------------------------
// Host-side weight packing
// ------------------------------------------------------------------------------------------
static const int kCin[4]  = {54, 64, 64, 128};
static const int kCinP[4] = {56, 64, 64, 128};
static const int kCout[4] = {64, 64, 128, 128};

size_t conv_pack_floats(int l) { return (size_t)kCout[l] * 3 * kCinP[l]; }

void conv_pack_host(int l, const float* w, float* out)
{
    const int cin = kCin[l], G = kCinP[l] / 8, cout = kCout[l];
    size_t o = 0;
    for (int mt = 0; mt < cout / 32; ++mt)
        for (int g = 0; g < G; ++g)
            for (int tap = 0; tap < 3; ++tap)
                for (int lane = 0; lane < 64; ++lane)
                    for (int u = 0; u < 4; ++u) {
                        const int co = 32 * mt + (lane & 31);
                        const int ci = 8 * g + 2 * u + (lane >> 5);
                        out[o++] = ci < cin ? w[((size_t)co * cin + ci) * 3 + tap] : 0.f;
                    }
}

// ------------------------------------------------------------------------------------------
// Device helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ float swap_adjacent(float v)
{   // lane 2m <-> 2m+1 : DPP quad_perm [1,0,3,2]
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
}

// Implicit-GEMM main loop of one conv layer for one wave: 5 column tiles x one 32-row tile.
//   bsrc = act + (lane>>5)*S + (lane&31) + 32*first_tile       (this lane's B element, tap 0, cin pair 0)
//   ap   = packed weights of this wave's row tile, + lane
//   a0   = weights of channel group 0 (3 taps), loaded by the caller BEFORE the barrier that
//          precedes this layer, so their L2 latency hides under the previous layer's write-back.
// Weights of group g+1 are fetched while group g's 60 MFMAs run (register double buffer).
struct A3 { float4 t0, t1, t2; };                      // one channel group's weights, 3 taps
__device__ __forceinline__ A3 load_a(const float4* __restrict__ ap, int g)
{
    A3 a;
    a.t0 = ap[(g * 3 + 0) * 64]; a.t1 = ap[(g * 3 + 1) * 64]; a.t2 = ap[(g * 3 + 2) * 64];
    return a;
}

// B operand of one (channel group, tap) stage: 4 K-steps x 5 column tiles = 20 values per lane.
#if DCE_EXPERIMENTS      // the direct-form conv stack of round 1 ships in the experiments build only (option conv_direct=1): the product library runs the Winograd kernels (conv_wino.hip) and conv_x3.hip
template <int S>
__device__ __forceinline__ void load_b(const float* __restrict__ bp, int tap, float (&b)[4 * NT])
{
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t) b[u * NT + t] = bp[2 * u * S + 32 * t + tap];
}

__device__ __forceinline__ void mfma_stage(const float4 a4, const float (&b)[4 * NT], f32x16 (&acc)[NT])
{
    const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b[u * NT + t], acc[t], 0, 0, 0);
}

// ask the scheduler for a 1:1 MFMA / LDS-read interleave over one stage (20 MFMAs)
__device__ __forceinline__ void interleave_stage()
{
#pragma unroll
    for (int i = 0; i < 4 * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
    }
}

// Software-pipelined main loop.  A stage = (8-channel group g, tap) = 20 MFMAs.  Three B
// register sets (one per tap) rotate so that every LDS read is issued a full stage (1280 MFMA
// cycles) before its first use:
//     MFMA tap0(g) || read B2(g)      MFMA tap1(g) || read B0(g+1)      MFMA tap2(g) || read B1(g+1)
// Weights of group g+1 (A operand, L2) are fetched during group g as well.
template <int CINP, int S>
__device__ __forceinline__ void conv_mfma(const float* __restrict__ bsrc,
                                          const float4* __restrict__ ap, A3 acur,
                                          f32x16 (&acc)[NT])
{
    constexpr int G = CINP / 8;
    float b0[4 * NT], b1[4 * NT], b2[4 * NT];
    load_b<S>(bsrc, 0, b0);
    load_b<S>(bsrc, 1, b1);
#pragma unroll 1
    for (int g = 0; g < G; ++g) {
        const int gn = g + 1 < G ? g + 1 : g;                // last iteration: harmless re-read
        const A3 anxt = load_a(ap, gn);
        // keep the three weight loads HERE: left alone, the scheduler sinks them to the end of
        // the iteration and the next iteration stalls on their full L2 latency (measured: +12 %)
        __builtin_amdgcn_sched_group_barrier(0x020, 3, 0);   // 3 VMEM reads first
        const float* bp = bsrc + g * 8 * S;
        const float* bn = bsrc + gn * 8 * S;
        load_b<S>(bp, 2, b2);
        mfma_stage(acur.t0, b0, acc);
        interleave_stage();
        load_b<S>(bn, 0, b0);
        mfma_stage(acur.t1, b1, acc);
        interleave_stage();
        load_b<S>(bn, 1, b1);
        mfma_stage(acur.t2, b2, acc);
        interleave_stage();
        acur = anxt;
    }
}

__device__ __forceinline__ void acc_init_bias(const float* __restrict__ bias_lds, int m0, int h,
                                              f32x16 (&acc)[NT])
{   // bias_lds: this layer's bias vector, staged in LDS at kernel start
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bv[r] = bias_lds[m0 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = bv[r];
}

// column validity: is column p a real sample (not a pad / filler) in a layout with window
// stride WS and TLEN samples per window?
template <int WS, int TLEN>
__device__ __forceinline__ bool col_valid(int p)
{
    const int r = p - 2;
    const int t = r >= WS ? r - WS : r;
    return r >= 0 && r < NW * WS && t < TLEN;
}

// ReLU + write back a layer's accumulators in place (no pooling).
template <int S, int WS, int TLEN, bool TAPS = false>
__device__ __forceinline__ void store_plain(float* __restrict__ act, const f32x16 (&acc)[NT],
                                            int m0, int nt0, int lane, float* __restrict__ tap = nullptr, int cout = 0)
{
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int p = 32 * (nt0 + t) + j;
        const bool valid = col_valid<WS, TLEN>(p);
        if (p <= S - 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                act[co * S + p + 1] = valid ? relu_nan(acc[t][r]) : 0.f;
                if constexpr (TAPS) {
                    if (valid) { const int w = (p - 2) >= WS ? 1 : 0; tap[((size_t)w * cout + co) * TLEN + (p - 2 - WS * w)] = relu_nan(acc[t][r]); }
                }
            }
        }
    }
}

// ReLU + MaxPool1d(2,2) + write back into the stage-2 layout (q = p/2 + 1).  After the
// adjacent-lane max both lanes of a pair hold the pooled value, so the even lane stores
// accumulator row r and the odd lane row r+1: every lane stores, no per-store predication.
template <bool TAPS = false>
__device__ __forceinline__ void store_pool_stage2(float* __restrict__ act, const f32x16 (&acc)[NT],
                                                  int m0, int nt0, int lane,
                                                  float* __restrict__ tap_conv2 = nullptr, float* __restrict__ tap_pool1 = nullptr)
{
    const int j = lane & 31, h = lane >> 5, odd = j & 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int p = 32 * (nt0 + t) + (j & ~1);                 // even column of this lane's pair
        const bool valid = col_valid<152, 150>(p);
        if (p <= 304) {
            float* dst = act + (m0 + 4 * h + odd) * S2 + (p >> 1) + 2;
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float v0 = relu_nan(acc[t][r]), v1 = relu_nan(acc[t][r + 1]);
                const float o0 = fmaxf(v0, swap_adjacent(v0));   // a NaN window is NaN everywhere
                const float o1 = fmaxf(v1, swap_adjacent(v1));
                dst[((r & 3) + 8 * (r >> 2)) * S2] = valid ? (odd ? o1 : o0) : 0.f;
                if constexpr (TAPS) {
                    const int pj = 32 * (nt0 + t) + j;                       // this lane's own column
                    if (col_valid<152, 150>(pj)) {
                        const int w = (pj - 2) >= 152 ? 1 : 0, tt = pj - 2 - 152 * w;
                        tap_conv2[((size_t)w * 64 + m0 + (r & 3) + 8 * (r >> 2) + 4 * h) * 150 + tt] = v0;
                        tap_conv2[((size_t)w * 64 + m0 + ((r + 1) & 3) + 8 * ((r + 1) >> 2) + 4 * h) * 150 + tt] = v1;
                    }
                    if (valid) {
                        const int w = (p - 2) >= 152 ? 1 : 0, tt = (p - 2 - 152 * w) >> 1;
                        tap_pool1[((size_t)w * 64 + m0 + 4 * h + odd + (r & 3) + 8 * (r >> 2)) * 75 + tt] = odd ? o1 : o0;
                    }
                }
            }
        }
    }
}

// ReLU + MaxPool1d(2,2) (floor: t = 74 dropped) + flatten (c*37 + j) to HBM; same lane pairing.
template <typename FT, bool TAPS = false>   // FT = float (headline) or unsigned short (bf16 features for DCE_BF16_FC)
__device__ __forceinline__ void store_pool_feat(FT* __restrict__ feat, int64_t win0, int nvalid,
                                                const f32x16 (&acc)[NT], int m0, int lane, float* __restrict__ tap_conv4 = nullptr)
{
    const int j = lane & 31, h = lane >> 5, odd = j & 1;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if constexpr (TAPS) {
            const int pj = 32 * t + j;                                       // this lane's own column of the stage-2 layout
            if (col_valid<76, 75>(pj)) {
                const int w = (pj - 2) >= 76 ? 1 : 0, tt = pj - 2 - 76 * w;
                if (w < nvalid) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        tap_conv4[((size_t)w * 128 + m0 + (r & 3) + 8 * (r >> 2) + 4 * h) * 75 + tt] = relu_nan(acc[t][r]);
                }
            }
        }
        const int q = 32 * t + (j & ~1);
        const int r_ = q - 2;
        const int w = r_ >= 76 ? 1 : 0;
        const int tt = r_ - 76 * w;
        if (r_ >= 0 && r_ < 152 && tt < 74 && w < nvalid) {
            FT* dst = feat + (win0 + w) * FEAT + (m0 + 4 * h + odd) * 37 + (tt >> 1);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float v0 = relu_nan(acc[t][r]), v1 = relu_nan(acc[t][r + 1]);
                const float o0 = fmaxf(v0, swap_adjacent(v0));
                const float o1 = fmaxf(v1, swap_adjacent(v1));
                put_feat(dst + ((r & 3) + 8 * (r >> 2)) * 37, odd ? o1 : o0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The fused kernel
// ------------------------------------------------------------------------------------------
template <bool ZS, typename FT, bool TAPS = false>
__global__ __launch_bounds__(256, 2)
void conv_stack_kernel(const float* __restrict__ src, int64_t n, ConvPack pk, FT* __restrict__ feat, LayerTaps taps)
{
    extern __shared__ __attribute__((aligned(16))) float act[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int64_t win0 = (int64_t)blockIdx.x * NW;
    const int nvalid = (n - win0) < NW ? (int)(n - win0) : NW;

    TRACE_MARK(0);
#if DCE_TRACE
    if (tid == 0 && blockIdx.x < 4096) g_trace[blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
#endif
    // biases of the four layers -> LDS once (read back as accumulator init values)
    for (int i = tid; i < 384; i += 256) {
        const int l = i < 64 ? 0 : (i < 128 ? 1 : (i < 256 ? 2 : 3));
        const int o = i < 64 ? i : (i < 128 ? i - 64 : (i < 256 ? i - 128 : i - 256));
        act[ACT_FLOATS + i] = pk.b[l][o];
    }

    // ---- prologue: HBM -> registers -> (z-score) -> LDS, transposed to [channel][position]
    {
        float x[NW][38];
        const int64_t wstride = ZS ? CH : (int64_t)WIN * CH;
        load_windows<ZS, NW>(src + win0 * wstride, wstride, nvalid, act + RED_ROW * S1, x, tid);
        if (tid < 4 * CH) {
            const int c = tid % CH, g = tid / CH;
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int m = 0; m < 38; ++m) {
                    const int t = 4 * m + g;
                    if (t < WIN) act[c * S1 + 3 + 152 * w + t] = x[w][m];
                }
        }
        // zero pads of rows 0..53 (addresses p+1 for p in {-1,0,1,152,153,304,305}) and the two
        // filler channels 54,55 (conv1 runs K over 56 input rows with zero weights there)
        for (int i = tid; i < CH * 7; i += 256) {
            const int c = i / 7, k = i % 7;
            const int a = k < 3 ? k : (k < 5 ? 150 + k : 300 + k);      // 0,1,2,153,154,305,306
            act[c * S1 + a] = 0.f;
        }
        for (int i = tid; i < 2 * S1; i += 256) act[CH * S1 + i] = 0.f;
    }
    __syncthreads();
    TRACE_MARK(1);

    f32x16 acc[NT];
    A3 a0;
    const float* bias_lds = act + ACT_FLOATS;            // [64 | 64 | 128 | 128]

    // ---- conv1: 54(56) -> 64, stage 1
    {
        const int mt = wv & 1, nt0 = NT * (wv >> 1);
        const float4* ap1 = reinterpret_cast<const float4*>(pk.w[0]) + mt * (3 * 7 * 64) + lane;
        const float4* ap2 = reinterpret_cast<const float4*>(pk.w[1]) + mt * (3 * 8 * 64) + lane;
        a0 = load_a(ap1, 0);
        acc_init_bias(bias_lds, 32 * mt, h, acc);
        conv_mfma<56, S1>(act + h * S1 + j + 32 * nt0, ap1, a0, acc);
        TRACE_MARK(2);
        a0 = load_a(ap2, 0);                             // next layer's first weights: in flight
        __syncthreads();                                 // across the write-back
        store_plain<S1, 152, 150, TAPS>(act, acc, 32 * mt, nt0, lane, TAPS ? taps.conv1 + win0 * 64 * 150 : nullptr, 64);
        acc_init_bias(bias_lds + 64, 32 * mt, h, acc);
        __syncthreads();
        TRACE_MARK(3);
        // ---- conv2: 64 -> 64, ReLU, pool -> stage 2
        conv_mfma<64, S1>(act + h * S1 + j + 32 * nt0, ap2, a0, acc);
        TRACE_MARK(4);
    }
    {
        const int mt12 = wv & 1, nt0 = NT * (wv >> 1), mt = wv;
        const float4* ap3 = reinterpret_cast<const float4*>(pk.w[2]) + mt * (3 * 8 * 64) + lane;
        const float4* ap4 = reinterpret_cast<const float4*>(pk.w[3]) + mt * (3 * 16 * 64) + lane;
        a0 = load_a(ap3, 0);
        __syncthreads();
        store_pool_stage2<TAPS>(act, acc, 32 * mt12, nt0, lane, TAPS ? taps.conv2 + win0 * 64 * 150 : nullptr,
                                TAPS ? taps.pool1 + win0 * 64 * 75 : nullptr);
        acc_init_bias(bias_lds + 128, 32 * mt, h, acc);
        __syncthreads();
        TRACE_MARK(5);
        // ---- conv3: 64 -> 128, stage 2
        conv_mfma<64, S2>(act + h * S2 + j, ap3, a0, acc);
        TRACE_MARK(6);
        a0 = load_a(ap4, 0);
        __syncthreads();
        store_plain<S2, 76, 75, TAPS>(act, acc, 32 * mt, 0, lane, TAPS ? taps.conv3 + win0 * 128 * 75 : nullptr, 128);
        acc_init_bias(bias_lds + 256, 32 * mt, h, acc);
        __syncthreads();
        TRACE_MARK(7);
        // ---- conv4: 128 -> 128, ReLU, pool, flatten -> HBM
        conv_mfma<128, S2>(act + h * S2 + j, ap4, a0, acc);
        TRACE_MARK(8);
        store_pool_feat<FT, TAPS>(feat, win0, nvalid, acc, 32 * mt, lane, TAPS ? taps.conv4 + win0 * 128 * 75 : nullptr);
        TRACE_MARK(9);
    }
}

#if DCE_TRACE
}  // namespace dce
extern "C" int dce_debug_trace_read(unsigned long long* out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_trace), sizeof(unsigned long long) * 16 * nblocks);
}
namespace dce {
#endif

template <bool ZS, typename FT> static hipError_t grant_conv_lds()
{   // > 64 KiB of dynamic LDS has to be granted per function, per device
    const int lds = DCE_TRACE ? 100 * 1024 : LDS_FLOATS * (int)sizeof(float);
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stack_kernel<ZS, FT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

hipError_t init_conv_stack()
{
    hipError_t e;
    if ((e = grant_conv_lds<true, float>()) != hipSuccess) return e;
    if ((e = grant_conv_lds<false, float>()) != hipSuccess) return e;
    if ((e = grant_conv_lds<true, unsigned short>()) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_stack_kernel<false, float, true>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * (int)sizeof(float))) != hipSuccess) return e;
    return grant_conv_lds<false, unsigned short>();
}

#endif  // DCE_EXPERIMENTS (the direct-form kernel)

hipError_t launch_conv_wino_taps(int kernel, const float* src, int64_t n, const ConvPack& pk, float* f,
                                 const LayerTaps& taps, hipStream_t st);

// dce_conv_layer_taps: one named conv kernel family with the per-layer taps on (kernel 4 = this file's direct form)
hipError_t launch_conv_taps(int kernel, const float* windows, int64_t n, const ConvPack& pk, float* feat,
                            const LayerTaps& taps, hipStream_t st)
{
    if (kernel != 4) return launch_conv_wino_taps(kernel, windows, n, pk, feat, taps, st);
#if DCE_EXPERIMENTS
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL((conv_stack_kernel<false, float, true>), dim3((unsigned)((n + NW - 1) / NW)), dim3(256),
                       LDS_FLOATS * sizeof(float), st, windows, n, pk, feat, taps);
    return hipGetLastError();
#else
    return hipErrorInvalidValue;                       // the direct form is not in this build
#endif
}

hipError_t launch_conv_stack(const float* src, int zscore, int64_t n, const ConvPack& pk,
                             void* feat, int feat_bf16, hipStream_t st, const long long* src_row)
{
#if !DCE_EXPERIMENTS
    (void)src; (void)zscore; (void)n; (void)pk; (void)feat; (void)feat_bf16; (void)st; (void)src_row;
    return hipErrorNotSupported;
#else
    if (src_row) return hipErrorNotSupported;          // the online graph runs on the Winograd kernels only
    if (n <= 0) return hipSuccess;
    size_t lds = LDS_FLOATS * sizeof(float);
#if DCE_TRACE
    if (tune().one_per_cu) lds = 100 * 1024;      // debug: force one workgroup per CU
#endif
    const dim3 grid((unsigned)((n + NW - 1) / NW)), block(256);
    plan_note("conv_direct");
    if (feat_bf16) {
        unsigned short* f = static_cast<unsigned short*>(feat);
        if (zscore) hipLaunchKernelGGL((conv_stack_kernel<true, unsigned short>), grid, block, lds, st, src, n, pk, f, LayerTaps{});
        else        hipLaunchKernelGGL((conv_stack_kernel<false, unsigned short>), grid, block, lds, st, src, n, pk, f, LayerTaps{});
    } else {
        float* f = static_cast<float*>(feat);
        if (zscore) hipLaunchKernelGGL((conv_stack_kernel<true, float>), grid, block, lds, st, src, n, pk, f, LayerTaps{});
        else        hipLaunchKernelGGL((conv_stack_kernel<false, float>), grid, block, lds, st, src, n, pk, f, LayerTaps{});
    }
    return hipGetLastError();
#endif
}

// ------------------------------------------------------------------------------------------
// contact_dataset.__getitem__ alone: materialise z-scored windows (n,150,54)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void zscore_windows_kernel(const float* __restrict__ seq, int64_t n, float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float red[4 * 216];
    const int tid = threadIdx.x;
    const int64_t i = blockIdx.x;
    float x[1][38];
    load_windows<true, 1>(seq + i * CH, CH, 1, red, x, tid);
    if (tid < 4 * CH) {
        const int c = tid % CH, g = tid / CH;
#pragma unroll
        for (int m = 0; m < 38; ++m) {
            const int t = 4 * m + g;
            if (t < WIN) out[i * (WIN * CH) + t * CH + c] = x[0][m];
        }
    }
    (void)n;
}

hipError_t launch_zscore_windows(const float* seq_first_row, int64_t n, float* out, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(zscore_windows_kernel, dim3((unsigned)n), dim3(256), 0, st, seq_first_row, n, out);
    return hipGetLastError();
}

}  // namespace dce
