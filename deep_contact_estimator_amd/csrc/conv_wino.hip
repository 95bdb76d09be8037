// conv_wino.hip -- the conv stack with Winograd F(2,3) minimal filtering on fp32 MFMA (gfx950).
//
// Same contract as conv_stack.hip (z-score + 4x[Conv1d k3 p1 + ReLU] + 2x MaxPool1d(2) + flatten;
// reference src/contact_cnn.py:10-44,61,64 and utils/data_handler.py:55-56) with 2/3 of the matrix
// work: for an output pair (y[2m], y[2m+1]) and inputs d0..d3 = x[2m-1..2m+2] of one channel
//     m0 = (d0-d2) g0      m1 = (d1+d2) (g0+g1+g2)/2      m2 = (d2-d1) (g0-g1+g2)/2      m3 = (d1-d3) g2
//     y[2m] = m0+m1+m2     y[2m+1] = m1-m2-m3
// Summed over input channels each m_k is a GEMM  M_k[Cout, pairs] = U_k[Cout,Cin] V_k[Cin,pairs]:
// 4 GEMMs of K = Cin instead of 3 taps x K = Cin over twice the columns.  U_k is precomputed on
// the host (fp64 -> fp32); V_k is formed on the fly from two 8-byte LDS reads (2 packed adds per 8
// MFMAs); the output transform, bias (initial value of M_1), ReLU, zero padding and MaxPool (the
// two outputs of a pair live in ONE lane -> no cross-lane traffic) are fused in the write-back.
// ReLU is v_max_f32; NaN/Inf inputs are carried per window (see the prologue) and surface as all-NaN
// features, which is what torch's logits show for such a window.
// fp32 Winograd F(2,3) has transform constants 0, +-1, +-1/2 only; measured logit error vs an fp64
// evaluation is the same as the direct form's.
//
// Structure is otherwise that of conv_stack.hip: one workgroup = 4 waves = 2 windows, two
// workgroups per CU, all activations in one LDS buffer written back in place, weights streamed
// from L2 as per-lane packed float4s, the layer's whole output held in accumulators.
//
// MFMA: v_mfma_f32_16x16x4_f32 (16-column tiles: 2 windows x 38 pairs = 76 columns fit 5 tiles
// at 95 %, where 32-column tiles would waste 21 %).  A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15],
// D[row = 4*(lane>>4)+r][col = lane&15].  A K-step is 4 input channels.
// Per wave: 2 row tiles (32 output channels) x 5 column tiles x 4 Winograd components = 40
// accumulators of 4 VGPRs.
//
// LDS layout: row = channel, row stride RS; window w occupies WSEG floats: index 0 = x[-1] = 0,
// index 1+t = x[t], index T+1 (and T+2) = 0.  Pair m reads indices 2m .. 2m+3 (8-byte aligned).
//   stage 1 (T=150): WSEG 152, RS 306, 75 pairs/window       stage 2 (T=75): WSEG 78, RS 156, 38 pairs
#include "conv_wino_dev.h"

namespace dce {

// ------------------------------------------------------------------------------------------
// Host-side weight transform + packing:  [mtile_pair][kstep][lane][mt(2) x comp(4)]
// ------------------------------------------------------------------------------------------
static const int wCin[4]  = {54, 64, 64, 128};
static const int wCinP[4] = {56, 64, 64, 128};
static const int wCout[4] = {64, 64, 128, 128};

size_t conv_wino_pack_floats(int l) { return (size_t)wCout[l] * wCinP[l] * 4; }

void conv_wino_pack_host(int l, const float* w, float* out)
{
    const int cin = wCin[l], steps = wCinP[l] / 4, cout = wCout[l];
    size_t o = 0;
    for (int P = 0; P < cout / 32; ++P)
        for (int s = 0; s < steps; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int mt = 0; mt < 2; ++mt) {
                    const int co = 32 * P + 16 * mt + (lane & 15);
                    const int ci = 4 * s + (lane >> 4);
                    double g0 = 0, g1 = 0, g2 = 0;
                    if (ci < cin) {
                        g0 = w[((size_t)co * cin + ci) * 3 + 0];
                        g1 = w[((size_t)co * cin + ci) * 3 + 1];
                        g2 = w[((size_t)co * cin + ci) * 3 + 2];
                    }
                    out[o++] = (float)g0;
                    out[o++] = (float)((g0 + g1 + g2) * 0.5);
                    out[o++] = (float)((g0 - g1 + g2) * 0.5);
                    out[o++] = (float)g2;
                }
}

// ------------------------------------------------------------------------------------------
// The fused kernel
// ------------------------------------------------------------------------------------------
template <bool ZS, typename FT, bool TAPS = false>
__global__ __launch_bounds__(256, 2)
void conv_wino_kernel(const float* __restrict__ src, int64_t n, ConvPack pk, FT* __restrict__ feat,
                      const long long* __restrict__ src_row, LayerTaps taps, Gate gate = Gate{})
{
    extern __shared__ __attribute__((aligned(16))) float act[];
    if (gate_closed(gate)) return;                        // DCE_FP32_SPLIT's fallback sequence: runs only behind a launch that left the guarded range
    if (src_row) src += *src_row * CH;                    // online graph: the window start lives in device memory
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int64_t win0 = (int64_t)blockIdx.x * NW;
    const int nvalid = (n - win0) < NW ? (int)(n - win0) : NW;

    TRACE_MARK(0);
#if DCE_TRACE
    if (tid == 0 && blockIdx.x < 4096) g_trace[blockIdx.x * 16 + 10] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
#endif
    for (int i = tid; i < 384; i += 256) {
        const int l = i < 64 ? 0 : (i < 128 ? 1 : (i < 256 ? 2 : 3));
        const int o = i < 64 ? i : (i < 128 ? i - 64 : (i < 256 ? i - 128 : i - 256));
        act[WACT_FLOATS + i] = pk.b[l][o];
    }
    if (tid < NW) reinterpret_cast<int*>(act + WACT_FLOATS + 384)[tid] = 0;     // per-window NaN flags

    // ---- prologue: HBM -> registers -> (z-score) -> LDS [channel][window segment]
    int* nanflag = reinterpret_cast<int*>(act + WACT_FLOATS + 384);
    {
        float x[NW][38];
        const int64_t wstride = ZS ? CH : (int64_t)WIN * CH;
        TRACE_MARK(11);
#if WINO_EXP & 64
        load_windows<ZS, NW>(src, wstride, nvalid, act + WRED_ROW * RS1, x, tid);     // probe: every workgroup reads windows 0, 1 (L2-hot)
#else
        load_windows<ZS, NW>(src + win0 * wstride, wstride, nvalid, act + WRED_ROW * RS1, x, tid);
#endif
        // ReLU runs as v_max_f32 (which drops NaN), so NaN / Inf semantics are carried per WINDOW:
        // torch turns any non-finite input sample into all-NaN logits (every fc.0 output sums over
        // every feature); here such a window gets all-NaN features at the end instead.
        bool bad0 = false, bad1 = false;
#if !(WINO_EXP & 32)
#pragma unroll
        for (int m = 0; m < 38; ++m) {
            bad0 |= !(fabsf(x[0][m]) <= 3.0e38f);
            bad1 |= !(fabsf(x[1][m]) <= 3.0e38f);
        }
#endif
#if DCE_TRACE
        asm volatile("" :: "v"(bad0), "v"(bad1));
#endif
        TRACE_MARK(12);
        __syncthreads();                         // flags were zeroed at kernel entry; also orders the
        if (bad0) nanflag[0] = 1;                // z-score scratch reads before the pad zeroing of
        if (bad1) nanflag[1] = 1;                // rows 56.. below
        TRACE_MARK(13);
        if (tid < 4 * CH) {
            const int c = tid % CH, g = tid / CH;
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int m = 0; m < 38; ++m) {
                    const int t = 4 * m + g;
                    if (t < WIN) act[c * RS1 + w * WS1 + 1 + t] = x[w][m];
                }
        }
        // zero pads (index 0 and 151 of each window segment) of all 64 rows, and the two filler
        // input channels 54,55 (conv1 runs K over 56 rows with zero weights there)
        for (int i = tid; i < 64 * 4; i += 256) {
            const int c = i >> 2, k = i & 3;
            act[c * RS1 + (k >> 1) * WS1 + (k & 1) * (WS1 - 1)] = 0.f;
        }
        for (int i = tid; i < 2 * RS1; i += 256) act[CH * RS1 + i] = 0.f;
    }
    TRACE_MARK(14);
    __syncthreads();
    TRACE_MARK(1);
    const bool nan0 = __builtin_amdgcn_readfirstlane(nanflag[0]) != 0;
    const bool nan1 = __builtin_amdgcn_readfirstlane(nanflag[1]) != 0;

    f32x4 acc[MT][NTW][4];
    int boff[NTW];
    const float* bias_lds = act + WACT_FLOATS;            // [64 | 64 | 128 | 128]
    const float* xrow1 = act + q * RS1;
    const float* xrow2 = act + q * RS2;

    // ---- stage 1: conv1 (54->64) and conv2 (64->64, pooled): wave = row-tile pair (wv&1), column half (wv>>1)
    {
        const int P = wv & 1, nt0 = NTW * (wv >> 1), co0 = 32 * P;
        const float4* ap1 = reinterpret_cast<const float4*>(pk.ww[0]) + P * (14 * 128) + 2 * lane;
        const float4* ap2 = reinterpret_cast<const float4*>(pk.ww[1]) + P * (16 * 128) + 2 * lane;
        col_offsets<TP1, WS1>(nt0, j, boff);
        A8 a = load_a8(ap1, 0);
        wino_mfma<RS1, 14>(xrow1, boff, ap1, a, bias_lds, co0, lane, acc);
        TRACE_MARK(2);
        a = load_a8(ap2, 0);                              // next layer's first weights in flight
        __syncthreads();                                  // across the write-back
        wino_store_plain<RS1, WS1, TP1, 150, MT, NTW, NW, TAPS>(act, acc, co0, nt0, lane, TAPS ? taps.conv1 + win0 * 64 * 150 : nullptr, 64);
        __syncthreads();
        TRACE_MARK(3);
        wino_mfma<RS1, 16>(xrow1, boff, ap2, a, bias_lds + 64, co0, lane, acc);
        TRACE_MARK(4);
        const float4* ap3 = reinterpret_cast<const float4*>(pk.ww[2]) + wv * (16 * 128) + 2 * lane;
        a = load_a8(ap3, 0);
        __syncthreads();
        wino_store_pool_stage2<MT, NTW, NW, TAPS>(act, acc, co0, nt0, lane, TAPS ? taps.conv2 + win0 * 64 * 150 : nullptr,
                                                  TAPS ? taps.pool1 + win0 * 64 * 75 : nullptr);
        // stage-2 pads: index 0, 76, 77 of each window segment, all 128 rows
        for (int i = tid; i < 128 * 6; i += 256) {
            const int c = i / 6, k = i % 6;
            act[c * RS2 + (k / 3) * WS2 + (k % 3 == 0 ? 0 : 75 + k % 3)] = 0.f;
        }
        // ---- stage 2: conv3 (64->128), conv4 (128->128, pooled -> HBM): wave = row-tile pair wv
        const int co2 = 32 * wv;
        const float4* ap4 = reinterpret_cast<const float4*>(pk.ww[3]) + wv * (32 * 128) + 2 * lane;
        col_offsets<TP2, WS2>(0, j, boff);
        __syncthreads();
        TRACE_MARK(5);
        wino_mfma<RS2, 16>(xrow2, boff, ap3, a, bias_lds + 128, co2, lane, acc);
        TRACE_MARK(6);
        a = load_a8(ap4, 0);
        __syncthreads();
        wino_store_plain<RS2, WS2, TP2, 75, MT, NTW, NW, TAPS>(act, acc, co2, 0, lane, TAPS ? taps.conv3 + win0 * 128 * 75 : nullptr, 128);
        __syncthreads();
        TRACE_MARK(7);
        wino_mfma<RS2, 32>(xrow2, boff, ap4, a, bias_lds + 256, co2, lane, acc);
        TRACE_MARK(8);
        wino_store_feat<FT, MT, NTW, NW, TAPS>(feat, win0, nvalid, acc, co2, lane, nan0, nan1, TAPS ? taps.conv4 + win0 * 128 * 75 : nullptr,
                                               (size_t)((n + 1) & ~(int64_t)1) * FEAT);
        TRACE_MARK(9);
    }
}

#if DCE_EXPERIMENTS
// ------------------------------------------------------------------------------------------
// The same two-window workgroup with FOUR row tiles per wave (round 3; conv_wino_rt4_kernel).
//
// What limits the kernel above is not the matrix work but what a wave issues between its MFMAs: per column tile one
// LDS quad read, four transform ops and an address op feed only 8 MFMAs, every wave of a stage recomputes the same V
// (stage 1: twice, stage 2: four times per workgroup), and a wave that runs alone on its SIMD (its partner workgroup is
// in a prologue / write-back) issues each of those bunches in order with its MFMAs.  Here a wave holds the same 40
// accumulator tiles in another shape: TWO full column tiles x 4 row tiles + ONE column tile x 2 row tiles --
//     stage 1 (64 channels = 4 row tiles, 10 column tiles):  wave w: column tiles 2w, 2w+1 (all rows) + tile 8 + (w>>1), row pair w&1
//     stage 2 (128 channels, 5 column tiles):                wave w: row group w>>1 (64 channels); column tiles 2b, 2b+1 (b = w&1)
//                                                             + tile 4, row pair b of the group
// -- so a K-step is 3 bunches for 40 MFMAs instead of 5 (transform VALU and LDS reads -40 %), at the price of 16 weight
// registers per K-step instead of 8.  Same MFMAs, same LDS layout, same per-accumulator K order: bit-identical features.
// MEASURED (profiles/r3b_conv_experiments.txt; 4096 windows, three interleaved rounds): 440.4 us vs 424.9 us for the
// kernel above -- SLOWER, so it is an opt-in A/B variant (DCE_CONV4=1), not the default.  The phase trace says why: its
// MFMA phases are 2.6 % shorter (178.8k vs 183.5k cycles per workgroup) but the prologue and the four write-backs of the
// workgroup that shares the CU grow by 13.6k cycles (28.3k vs 22.1k; 39.5k vs 32.1k): a wave's non-MFMA instructions get
// to issue in the BREAKS of the MFMA stream of the wave it shares a SIMD with, and runs of 16 MFMAs halve the breaks.
// Splitting a full tile's run into 8 + 8 around the transform ops (-DWINO4_SPLIT=1) gives back most of it (430.1 us) and
// still does not beat five runs of 8.  The kernel is issue-bound: a workgroup pair's 238k cycles are its 199.7k cycles of
// MFMAs plus ~4 cycles for each of the ~9,400 other instructions its two waves per SIMD issue.
// Register order of a wave's four row tiles: r[0], r[1] = the row pair that also serves the half tile, r[2], r[3] = the
// other pair of the wave's 64 channels (so the half tile needs no runtime register choice).

// ------------------------------------------------------------------------------------------
#ifndef WINO4_SPLIT
#define WINO4_SPLIT 0
#endif
struct A16 { float4 r[4]; };            // this lane's weights for one K-step: [row tile][comp]

__device__ __forceinline__ A16 load_a16(const float4* __restrict__ apH, const float4* __restrict__ apO, int s)
{   // apH / apO: packed weights of the half-tile pair / the other pair, + 2*lane float4
    A16 a;
    a.r[0] = apH[s * 128]; a.r[1] = apH[s * 128 + 1];
    a.r[2] = apO[s * 128]; a.r[3] = apO[s * 128 + 1];
    return a;
}

struct Acc4 { f32x4 f[2][4][4]; f32x4 h[2][4]; };      // [full tile][row tile][comp], half tile [row tile of pair][comp]

template <int RS>
__device__ __forceinline__ void wino_step4(const float* __restrict__ xs, const float* __restrict__ xn,
                                           const int (&boff)[3], const A16& a, V4& vcur, Quad& rawb, Acc4& acc)
{
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const float* pc = nt + 2 < 3 ? xs + boff[nt + 2] : xn + boff[nt + 2 - 3];
        const float v[4] = {vcur.a.x, vcur.b.x, vcur.b.y, vcur.a.y};
        auto mfmas = [&](int c0, int c1) {
#pragma unroll
            for (int c = c0; c < c1; ++c) {
#pragma unroll
                for (int rt = 0; rt < (nt < 2 ? 4 : 2); ++rt) {
                    const float4 w4 = a.r[rt];
                    const float av = c == 0 ? w4.x : c == 1 ? w4.y : c == 2 ? w4.z : w4.w;
                    if (nt < 2) acc.f[nt][rt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[c], acc.f[nt][rt][c], 0, 0, 0);
                    else        acc.h[rt][c]     = __builtin_amdgcn_mfma_f32_16x16x4f32(av, v[c], acc.h[rt][c], 0, 0, 0);
                }
            }
        };
#if WINO4_SPLIT
        // a full tile's 16 MFMAs in two runs of 8 with the bookkeeping between them: the wave that shares the SIMD
        // (the partner workgroup's) gets to issue in the breaks of this wave's MFMA stream, and a run of 16 halves them
        const Quad rawc = load_quad2(pc);                  // tile i+2 of the (K-step, column tile) sequence
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, nt < 2 ? 2 : 4);
        __builtin_amdgcn_sched_barrier(0);
        const V4 vnxt = wino_v(rawb);                      // tile i+1
        __builtin_amdgcn_sched_barrier(0);
        if (nt < 2) mfmas(2, 4);
        __builtin_amdgcn_sched_barrier(0);
#else
        const Quad rawc = load_quad2(pc);                  // tile i+2 of the (K-step, column tile) sequence
        const V4 vnxt = wino_v(rawb);                      // tile i+1
        __builtin_amdgcn_sched_barrier(0);
        mfmas(0, 4);
        __builtin_amdgcn_sched_barrier(0);
#endif
        vcur = vnxt;
        rawb = rawc;
    }
}

// One layer's main loop (see wino_mfma): two K-steps per iteration, weights of step s+1 / s+2 requested at the top of
// step s / s+1.  rowH / rowO: first output channel of the half-tile pair / of the other pair (for the bias).
template <int RS, int STEPS>
__device__ __forceinline__ void wino_mfma4(const float* __restrict__ xrow, const int (&boff)[3],
                                           const float4* __restrict__ apH, const float4* __restrict__ apO, A16 a_even,
                                           const float* __restrict__ bias_lds, int rowH, int rowO, int lane, Acc4& acc)
{
    static_assert(STEPS % 2 == 0 && STEPS >= 4, "two K-steps per iteration");
    {
        const int q4 = 4 * (lane >> 4);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            f32x4 b;
            const int co = (rt < 2 ? rowH : rowO) + 16 * (rt & 1) + q4;
#pragma unroll
            for (int r = 0; r < 4; ++r) b[r] = bias_lds[co + r];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                acc.f[t][rt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc.f[t][rt][1] = b;
                acc.f[t][rt][2] = f32x4{0.f, 0.f, 0.f, 0.f}; acc.f[t][rt][3] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            if (rt < 2) {
                acc.h[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc.h[rt][1] = b;
                acc.h[rt][2] = f32x4{0.f, 0.f, 0.f, 0.f}; acc.h[rt][3] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    }
    V4 vcur = wino_v(load_quad2(xrow + boff[0]));
    Quad rawb = load_quad2(xrow + boff[1]);
#pragma unroll 1
    for (int s = 0; s < STEPS; s += 2) {
        const int s2 = s + 2 < STEPS ? s + 2 : s;         // last iteration: harmless re-reads
        const A16 a_odd = load_a16(apH, apO, s + 1);
        wino_step4<RS>(xrow + s * 4 * RS, xrow + (s + 1) * 4 * RS, boff, a_even, vcur, rawb, acc);
        a_even = load_a16(apH, apO, s2);
        wino_step4<RS>(xrow + (s + 1) * 4 * RS, xrow + s2 * 4 * RS, boff, a_odd, vcur, rawb, acc);
    }
}

// Visit every (column tile, row tile) of a wave's accumulators: f(m0..m3 of the 4 r, column tile index ct, first channel co)
template <class F>
__device__ __forceinline__ void acc4_for_each(const Acc4& acc, const int (&ct)[3], int rowH, int rowO, F&& f)
{
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int rt = 0; rt < (t < 2 ? 4 : 2); ++rt) {
            const int co = (rt < 2 ? rowH : rowO) + 16 * (rt & 1);
            if (t < 2) f(acc.f[t][rt], ct[t], co); else f(acc.h[rt], ct[2], co);
        }
}

template <bool ZS, typename FT, bool TAPS = false>
__global__ __launch_bounds__(256, 2)
void conv_wino_rt4_kernel(const float* __restrict__ src, int64_t n, ConvPack pk, FT* __restrict__ feat,
                          const long long* __restrict__ src_row, LayerTaps taps)
{
    extern __shared__ __attribute__((aligned(16))) float act[];
    if (src_row) src += *src_row * CH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int64_t win0 = (int64_t)blockIdx.x * NW;
    const int nvalid = (n - win0) < NW ? (int)(n - win0) : NW;

    TRACE_MARK(0);
    for (int i = tid; i < 384; i += 256) {
        const int l = i < 64 ? 0 : (i < 128 ? 1 : (i < 256 ? 2 : 3));
        const int o = i < 64 ? i : (i < 128 ? i - 64 : (i < 256 ? i - 128 : i - 256));
        act[WACT_FLOATS + i] = pk.b[l][o];
    }
    if (tid < NW) reinterpret_cast<int*>(act + WACT_FLOATS + 384)[tid] = 0;
    int* nanflag = reinterpret_cast<int*>(act + WACT_FLOATS + 384);
    {   // ---- prologue: exactly conv_wino_kernel's
        float x[NW][38];
        const int64_t wstride = ZS ? CH : (int64_t)WIN * CH;
        load_windows<ZS, NW>(src + win0 * wstride, wstride, nvalid, act + WRED_ROW * RS1, x, tid);
        bool bad0 = false, bad1 = false;
#pragma unroll
        for (int m = 0; m < 38; ++m) {
            bad0 |= !(fabsf(x[0][m]) <= 3.0e38f);
            bad1 |= !(fabsf(x[1][m]) <= 3.0e38f);
        }
        __syncthreads();
        if (bad0) nanflag[0] = 1;
        if (bad1) nanflag[1] = 1;
        if (tid < 4 * CH) {
            const int c = tid % CH, g = tid / CH;
#pragma unroll
            for (int w = 0; w < NW; ++w)
#pragma unroll
                for (int m = 0; m < 38; ++m) {
                    const int t = 4 * m + g;
                    if (t < WIN) act[c * RS1 + w * WS1 + 1 + t] = x[w][m];
                }
        }
        for (int i = tid; i < 64 * 4; i += 256) {
            const int c = i >> 2, k = i & 3;
            act[c * RS1 + (k >> 1) * WS1 + (k & 1) * (WS1 - 1)] = 0.f;
        }
        for (int i = tid; i < 2 * RS1; i += 256) act[CH * RS1 + i] = 0.f;
    }
    __syncthreads();
    TRACE_MARK(1);
    const bool nan0 = __builtin_amdgcn_readfirstlane(nanflag[0]) != 0;
    const bool nan1 = __builtin_amdgcn_readfirstlane(nanflag[1]) != 0;

    Acc4 acc;
    int boff[3];
    const float* bias_lds = act + WACT_FLOATS;
    const float* xrow1 = act + q * RS1;
    const float* xrow2 = act + q * RS2;
    // column n of a stage -> (window, pair), LDS offset of the pair (fillers past the last pair read offset 0)
    auto col1 = [&](int ct, int& w, int& m) { return ColMap<TP1, NW>::valid(16 * ct + j, w, m); };
    auto col2 = [&](int ct, int& w, int& m) { return ColMap<TP2, NW>::valid(16 * ct + j, w, m); };

    // ---- stage 1: wave w = column tiles 2w, 2w+1 (64 channels) + tile 8 + (w>>1), channel pair w&1
    {
        const int ct[3] = {2 * wv, 2 * wv + 1, 8 + (wv >> 1)};
        const int pH = wv & 1, rowH = 32 * pH, rowO = 32 * (pH ^ 1);
        const float4* ap1H = reinterpret_cast<const float4*>(pk.ww[0]) + pH * (14 * 128) + 2 * lane;
        const float4* ap1O = reinterpret_cast<const float4*>(pk.ww[0]) + (pH ^ 1) * (14 * 128) + 2 * lane;
        const float4* ap2H = reinterpret_cast<const float4*>(pk.ww[1]) + pH * (16 * 128) + 2 * lane;
        const float4* ap2O = reinterpret_cast<const float4*>(pk.ww[1]) + (pH ^ 1) * (16 * 128) + 2 * lane;
#pragma unroll
        for (int t = 0; t < 3; ++t) boff[t] = 2 * (16 * ct[t] + j);
        A16 a = load_a16(ap1H, ap1O, 0);
        wino_mfma4<RS1, 14>(xrow1, boff, ap1H, ap1O, a, bias_lds, rowH, rowO, lane, acc);
        TRACE_MARK(2);
        a = load_a16(ap2H, ap2O, 0);
        __syncthreads();
        acc4_for_each(acc, ct, rowH, rowO, [&](const f32x4 (&m4)[4], int c_t, int co) {
            int w, m;
            if (!col1(c_t, w, m)) return;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* d = act + (co + 4 * q + r) * RS1 + w * WS1 + 1 + 2 * m;
                d[0] = fmaxf((m4[0][r] + m4[1][r]) + m4[2][r], 0.f);
                d[1] = fmaxf((m4[1][r] - m4[2][r]) - m4[3][r], 0.f);
                if constexpr (TAPS) {
                    float* tp = taps.conv1 + ((win0 + w) * 64 + co + 4 * q + r) * 150 + 2 * m;
                    tp[0] = d[0]; tp[1] = d[1];
                }
            }
        });
        __syncthreads();
        TRACE_MARK(3);
        wino_mfma4<RS1, 16>(xrow1, boff, ap2H, ap2O, a, bias_lds + 64, rowH, rowO, lane, acc);
        TRACE_MARK(4);
    }
    // ---- stage 2: wave w = channel group w>>1 (64 of 128); column tiles 2b, 2b+1 + tile 4 with channel pair b (b = w&1)
    {
        const int g = wv >> 1, b = wv & 1;
        const int ct2[3] = {2 * b, 2 * b + 1, 4};
        const int rowH = 64 * g + 32 * b, rowO = 64 * g + 32 * (b ^ 1);
        const float4* ap3H = reinterpret_cast<const float4*>(pk.ww[2]) + (2 * g + b) * (16 * 128) + 2 * lane;
        const float4* ap3O = reinterpret_cast<const float4*>(pk.ww[2]) + (2 * g + (b ^ 1)) * (16 * 128) + 2 * lane;
        const float4* ap4H = reinterpret_cast<const float4*>(pk.ww[3]) + (2 * g + b) * (32 * 128) + 2 * lane;
        const float4* ap4O = reinterpret_cast<const float4*>(pk.ww[3]) + (2 * g + (b ^ 1)) * (32 * 128) + 2 * lane;
        A16 a = load_a16(ap3H, ap3O, 0);
        __syncthreads();                                  // every wave is done reading conv2's input
        {   // conv2's write-back with the stage-1 tiling: ReLU + MaxPool -> stage-2 layout
            const int ct[3] = {2 * wv, 2 * wv + 1, 8 + (wv >> 1)};
            const int pH = wv & 1, rH = 32 * pH, rO = 32 * (pH ^ 1);
            acc4_for_each(acc, ct, rH, rO, [&](const f32x4 (&m4)[4], int c_t, int co) {
                int w, m;
                if (!col1(c_t, w, m)) return;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float y0 = (m4[0][r] + m4[1][r]) + m4[2][r], y1 = (m4[1][r] - m4[2][r]) - m4[3][r];
                    act[(co + 4 * q + r) * RS2 + w * WS2 + 1 + m] = fmaxf(fmaxf(y0, y1), 0.f);
                    if constexpr (TAPS) {
                        const int64_t row = (win0 + w) * 64 + co + 4 * q + r;
                        taps.conv2[row * 150 + 2 * m] = fmaxf(y0, 0.f);
                        taps.conv2[row * 150 + 2 * m + 1] = fmaxf(y1, 0.f);
                        taps.pool1[row * 75 + m] = fmaxf(fmaxf(y0, y1), 0.f);
                    }
                }
            });
        }
        for (int i = tid; i < 128 * 6; i += 256) {       // stage-2 pads: index 0, 76, 77 of each window segment
            const int c = i / 6, k = i % 6;
            act[c * RS2 + (k / 3) * WS2 + (k % 3 == 0 ? 0 : 75 + k % 3)] = 0.f;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) boff[t] = 2 * (16 * ct2[t] + j);
        __syncthreads();
        TRACE_MARK(5);
        wino_mfma4<RS2, 16>(xrow2, boff, ap3H, ap3O, a, bias_lds + 128, rowH, rowO, lane, acc);
        TRACE_MARK(6);
        a = load_a16(ap4H, ap4O, 0);
        __syncthreads();
        acc4_for_each(acc, ct2, rowH, rowO, [&](const f32x4 (&m4)[4], int c_t, int co) {
            int w, m;
            if (!col2(c_t, w, m)) return;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float* d = act + (co + 4 * q + r) * RS2 + w * WS2 + 1 + 2 * m;
                d[0] = fmaxf((m4[0][r] + m4[1][r]) + m4[2][r], 0.f);
                d[1] = 2 * m + 1 < 75 ? fmaxf((m4[1][r] - m4[2][r]) - m4[3][r], 0.f) : 0.f;     // index 76 is a zero pad
                if constexpr (TAPS) {
                    float* tp = taps.conv3 + ((win0 + w) * 128 + co + 4 * q + r) * 75 + 2 * m;
                    tp[0] = d[0];
                    if (2 * m + 1 < 75) tp[1] = d[1];
                }
            }
        });
        __syncthreads();
        TRACE_MARK(7);
        wino_mfma4<RS2, 32>(xrow2, boff, ap4H, ap4O, a, bias_lds + 256, rowH, rowO, lane, acc);
        TRACE_MARK(8);
        const float nanv = __builtin_nanf("");
        acc4_for_each(acc, ct2, rowH, rowO, [&](const f32x4 (&m4)[4], int c_t, int co) {
            int w, m;
            if (!col2(c_t, w, m) || w >= nvalid) return;
            const bool bad = w ? nan1 : nan0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float y0 = (m4[0][r] + m4[1][r]) + m4[2][r], y1 = (m4[1][r] - m4[2][r]) - m4[3][r];
                if constexpr (TAPS) {
                    float* tp = taps.conv4 + ((win0 + w) * 128 + co + 4 * q + r) * 75 + 2 * m;
                    tp[0] = fmaxf(y0, 0.f);
                    if (2 * m + 1 < 75) tp[1] = fmaxf(y1, 0.f);
                }
                if (m < 37) put_feat(feat + (win0 + w) * FEAT + (co + 4 * q + r) * 37 + m, bad ? nanv : fmaxf(fmaxf(y0, y1), 0.f));
            }
        });
        TRACE_MARK(9);
    }
}

#endif  // DCE_EXPERIMENTS (conv_wino_rt4_kernel)

// ------------------------------------------------------------------------------------------
// One window per workgroup: the latency variant for a handful of windows (online mode, the
// reference's batch_size 1).  Same layers, same LDS layout (window segment 0 only), same per-
// accumulator K order -> the features are bit-identical to conv_wino_kernel's; only the tiling
// changes: stage 1 = one row tile x 5 column tiles per wave (75 pairs), stage 2 = two row tiles x
// 3 column tiles (38 pairs): 1752 MFMAs per wave instead of 3120 for a half-empty two-window
// workgroup, and twice as many workgroups to spread over the CUs.
// ------------------------------------------------------------------------------------------
template <bool ZS, bool TAPS = false>
__global__ __launch_bounds__(256)
void conv_wino1_kernel(const float* __restrict__ src, int64_t n, ConvPack pk, float* __restrict__ feat,
                       const long long* __restrict__ src_row, LayerTaps taps)
{
    extern __shared__ __attribute__((aligned(16))) float act[];
    if (src_row) src += *src_row * CH;                    // online graph: the window start lives in device memory
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int64_t win0 = blockIdx.x;
    if (win0 >= n) return;
    TRACE_MARK(0);

    for (int i = tid; i < 384; i += 256) {
        const int l = i < 64 ? 0 : (i < 128 ? 1 : (i < 256 ? 2 : 3));
        const int o = i < 64 ? i : (i < 128 ? i - 64 : (i < 256 ? i - 128 : i - 256));
        act[WACT_FLOATS + i] = pk.b[l][o];
    }
    int* nanflag = reinterpret_cast<int*>(act + WACT_FLOATS + 384);
    if (tid == 0) nanflag[0] = 0;
    // conv1's first PF K-steps of weights are requested before the window itself
    constexpr int PF = WINO1_PF;                          // weight prefetch depth, K-steps
    A8 ring[PF];
    const float4* ap1 = reinterpret_cast<const float4*>(pk.ww[0]) + (wv >> 1) * (14 * 128) + 2 * lane + (wv & 1);
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = load_a8<1>(ap1, i);
    {
        float x[1][38];
        const int64_t wstride = ZS ? CH : (int64_t)WIN * CH;
        load_windows<ZS, 1>(src + win0 * wstride, wstride, 1, act + WRED_ROW * RS1, x, tid);
        bool bad0 = false;
#pragma unroll
        for (int m = 0; m < 38; ++m) bad0 |= !(fabsf(x[0][m]) <= 3.0e38f);
        __syncthreads();
        if (bad0) nanflag[0] = 1;
        if (tid < 4 * CH) {
            const int c = tid % CH, g = tid / CH;
#pragma unroll
            for (int m = 0; m < 38; ++m) {
                const int t = 4 * m + g;
                if (t < WIN) act[c * RS1 + 1 + t] = x[0][m];
            }
        }
        for (int i = tid; i < 64 * 2; i += 256) act[(i >> 1) * RS1 + (i & 1) * (WS1 - 1)] = 0.f;   // x[-1], x[150]
        for (int i = tid; i < 2 * RS1; i += 256) act[CH * RS1 + i] = 0.f;                          // channels 54, 55
    }
    __syncthreads();
    TRACE_MARK(1);
    const bool nan0 = __builtin_amdgcn_readfirstlane(nanflag[0]) != 0;
    const float* bias_lds = act + WACT_FLOATS;
    const float* xrow1 = act + q * RS1;
    const float* xrow2 = act + q * RS2;
    const float4* ap2 = reinterpret_cast<const float4*>(pk.ww[1]) + (wv >> 1) * (16 * 128) + 2 * lane + (wv & 1);
    const float4* ap3 = reinterpret_cast<const float4*>(pk.ww[2]) + wv * (16 * 128) + 2 * lane;
    const float4* ap4 = reinterpret_cast<const float4*>(pk.ww[3]) + wv * (32 * 128) + 2 * lane;
    {   // ---- stage 1: wave = row tile wv (16 output channels) x all 5 column tiles
        f32x4 acc[1][5][4];
        int boff[5];
        const int co0 = 16 * wv;
        col_offsets<TP1, WS1, 5, 1>(0, j, boff);
        wino_mfma_deep<RS1, 14, 1, 5, PF, 1, 0>(xrow1, boff, ap1, ap2, ring, bias_lds, co0, lane, acc);
        TRACE_MARK(2);
        __syncthreads();
        wino_store_plain<RS1, WS1, TP1, 150, 1, 5, 1, TAPS>(act, acc, co0, 0, lane, TAPS ? taps.conv1 + win0 * 64 * 150 : nullptr, 64);
        __syncthreads();
        TRACE_MARK(3);
        wino_mfma_deep<RS1, 16, 1, 5, PF, 2, 14 % PF>(xrow1, boff, ap2, ap3, ring, bias_lds + 64, co0, lane, acc);
        TRACE_MARK(4);
        __syncthreads();
        wino_store_pool_stage2<1, 5, 1, TAPS>(act, acc, co0, 0, lane, TAPS ? taps.conv2 + win0 * 64 * 150 : nullptr,
                                              TAPS ? taps.pool1 + win0 * 64 * 75 : nullptr);
        for (int i = tid; i < 128 * 3; i += 256) {       // stage-2 pads: index 0, 76, 77
            const int c = i / 3, k = i % 3;
            act[c * RS2 + (k == 0 ? 0 : 75 + k)] = 0.f;
        }
    }
    {   // ---- stage 2: wave = row-tile pair wv (32 output channels) x 3 column tiles
        f32x4 acc[2][3][4];
        int boff[3];
        const int co2 = 32 * wv;
        col_offsets<TP2, WS2, 3, 1>(0, j, boff);
        __syncthreads();
        TRACE_MARK(5);
        wino_mfma_deep<RS2, 16, 2, 3, PF, 2, 30 % PF>(xrow2, boff, ap3, ap4, ring, bias_lds + 128, co2, lane, acc);
        TRACE_MARK(6);
        __syncthreads();
        wino_store_plain<RS2, WS2, TP2, 75, 2, 3, 1, TAPS>(act, acc, co2, 0, lane, TAPS ? taps.conv3 + win0 * 128 * 75 : nullptr, 128);
        __syncthreads();
        TRACE_MARK(7);
        wino_mfma_deep<RS2, 32, 2, 3, PF, 2, 46 % PF>(xrow2, boff, ap4, nullptr, ring, bias_lds + 256, co2, lane, acc);
        TRACE_MARK(8);
        wino_store_feat<float, 2, 3, 1, TAPS>(feat, win0, 1, acc, co2, lane, nan0, false, TAPS ? taps.conv4 + win0 * 128 * 75 : nullptr);
        TRACE_MARK(9);
    }
}

// ------------------------------------------------------------------------------------------
// The same one-window kernel on EIGHT waves (two per SIMD).  A workgroup that is alone on its CU runs its MFMA
// phases at 67 % (stage 1) / 78 % (stage 2) of the pipe rate with one wave per SIMD (tools/trace_conv1.py: 75.7k cycles
// for 56.1k of matrix-pipe work) -- the wave's own transform VALU, LDS reads and waits issue in order with its MFMAs.
// A second wave on the SIMD fills those gaps (tools/micro/wino_loop.hip: 91 -> 100 %).  Tiling, so that the two waves
// of a SIMD (w and w+4) together carry what one carried before:
//   stage 1: wave (row tile w&3, column group w>>2): column tiles {0,1,2} / {3,4}
//   stage 2: wave = row tile w (16 of the 128 output channels) x all 3 column tiles
// (row-tile PAIRS x column groups {0,1} / {2} would halve stage 2's transform VALU per MFMA, but the software pipeline
//  of wino_step needs at least two column tiles per wave)
// Same LDS layout, same per-accumulator K order: bit-identical features.
// ------------------------------------------------------------------------------------------
template <int NTW1, bool TAPS = false>
__device__ __forceinline__ void wino1x8_stage1(float* __restrict__ act, const float* __restrict__ bias_lds,
                                               const float4* ap1, const float4* ap2, const float4* ap3, A8 (&ring)[WINO1_PF],
                                               int rt, int nt0, int lane, int tid, const LayerTaps& taps, int64_t win0)
{
    constexpr int PF = WINO1_PF;
    const int j = lane & 15, q = lane >> 4;
    const float* xrow1 = act + q * RS1;
    f32x4 acc[1][NTW1][4];
    int boff[NTW1];
    const int co0 = 16 * rt;
    col_offsets<TP1, WS1, NTW1, 1>(nt0, j, boff);
    wino_mfma_deep<RS1, 14, 1, NTW1, PF, 1, 0>(xrow1, boff, ap1, ap2, ring, bias_lds, co0, lane, acc);
    TRACE_MARK(2);
    __syncthreads();
    wino_store_plain<RS1, WS1, TP1, 150, 1, NTW1, 1, TAPS>(act, acc, co0, nt0, lane, TAPS ? taps.conv1 + win0 * 64 * 150 : nullptr, 64);
    __syncthreads();
    TRACE_MARK(3);
    wino_mfma_deep<RS1, 16, 1, NTW1, PF, 1, 14 % PF>(xrow1, boff, ap2, ap3, ring, bias_lds + 64, co0, lane, acc);
    TRACE_MARK(4);
    __syncthreads();
    wino_store_pool_stage2<1, NTW1, 1, TAPS>(act, acc, co0, nt0, lane, TAPS ? taps.conv2 + win0 * 64 * 150 : nullptr,
                                             TAPS ? taps.pool1 + win0 * 64 * 75 : nullptr);
    for (int i = tid; i < 128 * 3; i += 512) {           // stage-2 pads: index 0, 76, 77
        const int c = i / 3, k = i % 3;
        act[c * RS2 + (k == 0 ? 0 : 75 + k)] = 0.f;
    }
}

template <bool ZS, bool TAPS = false>
__global__ __launch_bounds__(512)
void conv_wino1x8_kernel(const float* __restrict__ src, int64_t n, ConvPack pk, float* __restrict__ feat,
                         const long long* __restrict__ src_row, LayerTaps taps)
{
    extern __shared__ __attribute__((aligned(16))) float act[];
    if (src_row) src += *src_row * CH;                    // online graph: the window start lives in device memory
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t win0 = blockIdx.x;
    if (win0 >= n) return;
    TRACE_MARK(0);

    for (int i = tid; i < 384; i += 512) {
        const int l = i < 64 ? 0 : (i < 128 ? 1 : (i < 256 ? 2 : 3));
        const int o = i < 64 ? i : (i < 128 ? i - 64 : (i < 256 ? i - 128 : i - 256));
        act[WACT_FLOATS + i] = pk.b[l][o];
    }
    int* nanflag = reinterpret_cast<int*>(act + WACT_FLOATS + 384);
    if (tid == 0) nanflag[0] = 0;
    constexpr int PF = WINO1_PF;                          // weight prefetch depth, K-steps
    A8 ring[PF];
    const int rt = wv & 3;                                // stage 1: row tile; the packed weights hold row-tile PAIRS
    const float4* ap1 = reinterpret_cast<const float4*>(pk.ww[0]) + (rt >> 1) * (14 * 128) + 2 * lane + (rt & 1);
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = load_a8<1>(ap1, i);
    {
        float x[1][38];
        const int64_t wstride = ZS ? CH : (int64_t)WIN * CH;
        // (the load helper maps threads 0..215 to (row group, channel); the other threads' loads are never stored)
        load_windows<ZS, 1>(src + win0 * wstride, wstride, 1, act + WRED_ROW * RS1, x, tid < 256 ? tid : 255);
        bool bad0 = false;
#pragma unroll
        for (int m = 0; m < 38; ++m) bad0 |= !(fabsf(x[0][m]) <= 3.0e38f);
        __syncthreads();
        if (bad0 && tid < 4 * CH) nanflag[0] = 1;
        if (tid < 4 * CH) {
            const int c = tid % CH, g = tid / CH;
#pragma unroll
            for (int m = 0; m < 38; ++m) {
                const int t = 4 * m + g;
                if (t < WIN) act[c * RS1 + 1 + t] = x[0][m];
            }
        }
        for (int i = tid; i < 64 * 2; i += 512) act[(i >> 1) * RS1 + (i & 1) * (WS1 - 1)] = 0.f;   // x[-1], x[150]
        for (int i = tid; i < 2 * RS1; i += 512) act[CH * RS1 + i] = 0.f;                          // channels 54, 55
    }
    __syncthreads();
    TRACE_MARK(1);
    const bool nan0 = __builtin_amdgcn_readfirstlane(nanflag[0]) != 0;
    const float* bias_lds = act + WACT_FLOATS;
    const float4* ap2 = reinterpret_cast<const float4*>(pk.ww[1]) + (rt >> 1) * (16 * 128) + 2 * lane + (rt & 1);
    const float4* ap3 = reinterpret_cast<const float4*>(pk.ww[2]) + (wv >> 1) * (16 * 128) + 2 * lane + (wv & 1);
    const float4* ap4 = reinterpret_cast<const float4*>(pk.ww[3]) + (wv >> 1) * (32 * 128) + 2 * lane + (wv & 1);
    if (wv < 4) wino1x8_stage1<3, TAPS>(act, bias_lds, ap1, ap2, ap3, ring, rt, 0, lane, tid, taps, win0);
    else        wino1x8_stage1<2, TAPS>(act, bias_lds, ap1, ap2, ap3, ring, rt, 3, lane, tid, taps, win0);
    {   // ---- stage 2: wave = row tile wv (16 output channels) x 3 column tiles
        const int j = lane & 15, q = lane >> 4;
        const float* xrow2 = act + q * RS2;
        f32x4 acc[1][3][4];
        int boff[3];
        const int co2 = 16 * wv;
        col_offsets<TP2, WS2, 3, 1>(0, j, boff);
        __syncthreads();
        TRACE_MARK(5);
        wino_mfma_deep<RS2, 16, 1, 3, PF, 1, 30 % PF>(xrow2, boff, ap3, ap4, ring, bias_lds + 128, co2, lane, acc);
        TRACE_MARK(6);
        __syncthreads();
        wino_store_plain<RS2, WS2, TP2, 75, 1, 3, 1, TAPS>(act, acc, co2, 0, lane, TAPS ? taps.conv3 + win0 * 128 * 75 : nullptr, 128);
        __syncthreads();
        TRACE_MARK(7);
        wino_mfma_deep<RS2, 32, 1, 3, PF, 1, 46 % PF>(xrow2, boff, ap4, nullptr, ring, bias_lds + 256, co2, lane, acc);
        TRACE_MARK(8);
        wino_store_feat<float, 1, 3, 1, TAPS>(feat, win0, 1, acc, co2, lane, nan0, false, TAPS ? taps.conv4 + win0 * 128 * 75 : nullptr);
        TRACE_MARK(9);
    }
}

template <bool ZS, int NSEG, int NT1, int NT2, bool TAPS = false>
__global__ __launch_bounds__(512)
void conv_wino_seg_kernel(const float* __restrict__ src, int64_t n, ConvPack pk, float* __restrict__ feat,
                          const long long* __restrict__ src_row, LayerTaps taps, int chsplit = 0)
{
    extern __shared__ __attribute__((aligned(16))) float act[];
    if (src_row) src += *src_row * CH;                    // online graph: the window start lives in device memory
    // chsplit (quarter segments of up to 32 windows: 8 n <= 256 workgroups, one per CU): TWO workgroups per segment, each finishing one half of conv4's
    // output channels (41 % of the stack's MFMAs) -- four waves instead of eight on an MFMA-bound layer, on CUs that would idle; conv1..3 are computed by
    // both.  A channel's chain is the same instructions on the same wave either way: the same bits (as the latency mode's kernels, latency.hip)
    const unsigned blk = chsplit ? blockIdx.x >> 1 : blockIdx.x;
    const int64_t win0 = blk / NSEG;
    if (win0 >= n) return;
    conv_seg_body<ZS, NSEG, NT1, NT2, TAPS>(act, src, win0, (int)(blk % NSEG), pk, feat, taps, nullptr, chsplit ? (int)(blockIdx.x & 1) : -1);
}


#if DCE_TRACE
}  // namespace dce
extern "C" int dce_debug_trace_read_wino(unsigned long long* out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_trace), sizeof(unsigned long long) * 16 * nblocks);
}
namespace dce {
#endif

template <bool ZS, typename FT> static hipError_t grant_wino_lds()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_kernel<ZS, FT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, DCE_TRACE ? 100 * 1024 : WLDS_FLOATS * (int)sizeof(float));
}

hipError_t init_conv_wino()
{
    hipError_t e;
    if ((e = grant_wino_lds<true, float>()) != hipSuccess) return e;
    if ((e = grant_wino_lds<false, float>()) != hipSuccess) return e;
    if ((e = grant_wino_lds<true, unsigned short>()) != hipSuccess) return e;
    if ((e = grant_wino_lds<false, unsigned short>()) != hipSuccess) return e;
#if DCE_EXPERIMENTS      // (DCE_FP32_SPLIT's three-plane feature store and the four-wave predecessor of the one-window kernel: experiments build since round 6)
    if ((e = grant_wino_lds<true, Feat3>()) != hipSuccess) return e;
    if ((e = grant_wino_lds<false, Feat3>()) != hipSuccess) return e;
    for (const void* k : {reinterpret_cast<const void*>(&conv_wino1_kernel<true>), reinterpret_cast<const void*>(&conv_wino1_kernel<false>), reinterpret_cast<const void*>(&conv_wino1_kernel<false, true>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS_FLOATS * (int)sizeof(float))) != hipSuccess) return e;
#endif
    for (const void* k : {reinterpret_cast<const void*>(&conv_wino1x8_kernel<true>), reinterpret_cast<const void*>(&conv_wino1x8_kernel<false>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS_FLOATS * (int)sizeof(float))) != hipSuccess) return e;
    for (const void* k : {reinterpret_cast<const void*>(&conv_wino_seg_kernel<true, 2, 3, 2>), reinterpret_cast<const void*>(&conv_wino_seg_kernel<false, 2, 3, 2>),
                          reinterpret_cast<const void*>(&conv_wino_seg_kernel<true, 4, 2, 1>), reinterpret_cast<const void*>(&conv_wino_seg_kernel<false, 4, 2, 1>),
                          reinterpret_cast<const void*>(&conv_wino_seg_kernel<false, 2, 3, 2, true>), reinterpret_cast<const void*>(&conv_wino_seg_kernel<false, 4, 2, 1, true>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, HLDS_FLOATS * (int)sizeof(float))) != hipSuccess) return e;
    // the TAPS instantiations (dce_conv_layer_taps: parity tests of the layers inside the fused stack)
#if DCE_EXPERIMENTS
    for (const void* k : {reinterpret_cast<const void*>(&conv_wino_rt4_kernel<true, float>), reinterpret_cast<const void*>(&conv_wino_rt4_kernel<false, float>),
                          reinterpret_cast<const void*>(&conv_wino_rt4_kernel<true, unsigned short>), reinterpret_cast<const void*>(&conv_wino_rt4_kernel<false, unsigned short>),
                          reinterpret_cast<const void*>(&conv_wino_rt4_kernel<false, float, true>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, DCE_TRACE ? 100 * 1024 : WLDS_FLOATS * (int)sizeof(float))) != hipSuccess) return e;
#endif
    for (const void* k : {reinterpret_cast<const void*>(&conv_wino_kernel<false, float, true>), reinterpret_cast<const void*>(&conv_wino1x8_kernel<false, true>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, WLDS_FLOATS * (int)sizeof(float))) != hipSuccess) return e;
    return hipSuccess;
}

// One launch of a NAMED conv kernel family on pre-normalised windows with the per-layer taps switched on
// (kernel numbering: dce_kernels.h).  The TAPS instantiations differ from the product kernels only by the extra
// global stores next to each layer's write-back.
hipError_t launch_conv_wino_taps(int kernel, const float* src, int64_t n, const ConvPack& pk, float* f,
                                 const LayerTaps& taps, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const size_t lds = WLDS_FLOATS * sizeof(float), hl = HLDS_FLOATS * sizeof(float);
    const long long* none = nullptr;
    switch (kernel) {
    case 0: hipLaunchKernelGGL((conv_wino_kernel<false, float, true>), dim3((unsigned)((n + NW - 1) / NW)), dim3(256), lds, st, src, n, pk, f, none, taps); break;
    case 1: hipLaunchKernelGGL((conv_wino1x8_kernel<false, true>), dim3((unsigned)n), dim3(512), lds, st, src, n, pk, f, none, taps); break;
    case 2: hipLaunchKernelGGL((conv_wino_seg_kernel<false, 2, 3, 2, true>), dim3((unsigned)(2 * n)), dim3(512), hl, st, src, n, pk, f, none, taps, 0); break;
    case 3: hipLaunchKernelGGL((conv_wino_seg_kernel<false, 4, 2, 1, true>), dim3((unsigned)(4 * n)), dim3(512), hl, st, src, n, pk, f, none, taps, 0); break;
#if DCE_EXPERIMENTS
    case 6: hipLaunchKernelGGL((conv_wino_rt4_kernel<false, float, true>), dim3((unsigned)((n + NW - 1) / NW)), dim3(256), lds, st, src, n, pk, f, none, taps); break;
    case 5: hipLaunchKernelGGL((conv_wino1_kernel<false, true>), dim3((unsigned)n), dim3(256), lds, st, src, n, pk, f, none, taps); break;
#endif
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_conv_wino(const float* src, int zscore, int64_t n, const ConvPack& pk,
                            void* feat, int feat_bf16, hipStream_t st, const long long* src_row)
{
    if (n <= 0) return hipSuccess;
    const Tuning& tu = tune();
    const int64_t wino1_max = tu.wino1_max >= 0 ? tu.wino1_max : WINO1_MAX_N;
    // The two-window kernel fills the chip with rounds of 512 workgroups = 1024 windows; up to 256 windows past a round
    // would each sit alone on a CU for a lone workgroup's 65 us.  Windows are independent and every conv kernel produces
    // the same bits, so that remainder goes to the one-window / segment kernels instead (20 .. 38 us).
    const bool peel = tu.conv_peel;
    if (peel && !feat_bf16 && !src_row && !DCE_TRACE && n > 1024 && n % 1024 != 0 && n % 1024 <= wino1_max) {
        const int64_t rest = n % 1024, m = n - rest;
        const hipError_t e = launch_conv_wino(src, zscore, m, pk, feat, feat_bf16, st, nullptr);
        if (e != hipSuccess) return e;
        return launch_conv_wino(src + m * (zscore ? (int64_t)CH : (int64_t)WIN * CH), zscore, rest, pk,
                                static_cast<float*>(feat) + m * FEAT, feat_bf16, st, nullptr);
    }
    size_t lds = WLDS_FLOATS * sizeof(float);
#if DCE_TRACE
    if (tu.one_per_cu) lds = 100 * 1024;      // debug: force one workgroup per CU
#endif
    const dim3 grid((unsigned)((n + NW - 1) / NW)), block(256);
    if (!feat_bf16 && n <= wino1_max && (!DCE_TRACE || tu.trace_wino1)) {
        // at most one workgroup per CU: one window each finishes in 56 % of a two-window workgroup's time
        float* f = static_cast<float*>(feat);
        // two / four CUs per window while that leaves no CU without one
        const int64_t half_max = tu.winoh_max >= 0 ? tu.winoh_max : WINOH_MAX_N;
        const int64_t quarter_max = tu.winoq_max >= 0 ? tu.winoq_max : WINOQ_MAX_N;
        const size_t hl = HLDS_FLOATS * sizeof(float);
        if (n <= quarter_max) {
            const int chs = n <= tu.winoq_chsplit_max ? 1 : 0;        // conv4's channel halves on two workgroups while every workgroup still has a CU of its own
            plan_note(chs ? "conv_wino_quarter_ch2" : "conv_wino_quarter");
            const dim3 g((unsigned)((chs ? 8 : 4) * n));
            if (zscore) hipLaunchKernelGGL((conv_wino_seg_kernel<true, 4, 2, 1>), g, dim3(512), hl, st, src, n, pk, f, src_row, LayerTaps{}, chs);
            else        hipLaunchKernelGGL((conv_wino_seg_kernel<false, 4, 2, 1>), g, dim3(512), hl, st, src, n, pk, f, src_row, LayerTaps{}, chs);
            return hipGetLastError();
        }
        if (n <= half_max) {
            plan_note("conv_wino_half");
            if (zscore) hipLaunchKernelGGL((conv_wino_seg_kernel<true, 2, 3, 2>), dim3((unsigned)(2 * n)), dim3(512), hl, st, src, n, pk, f, src_row, LayerTaps{}, 0);
            else        hipLaunchKernelGGL((conv_wino_seg_kernel<false, 2, 3, 2>), dim3((unsigned)(2 * n)), dim3(512), hl, st, src, n, pk, f, src_row, LayerTaps{}, 0);
            return hipGetLastError();
        }
#if DCE_EXPERIMENTS
        if (!tu.wino1_w8) {                              // the four-wave predecessor of the one-window kernel (A/B)
            plan_note("conv_wino1x4");
            if (zscore) hipLaunchKernelGGL((conv_wino1_kernel<true>), dim3((unsigned)n), block, lds, st, src, n, pk, f, src_row, LayerTaps{});
            else        hipLaunchKernelGGL((conv_wino1_kernel<false>), dim3((unsigned)n), block, lds, st, src, n, pk, f, src_row, LayerTaps{});
            return hipGetLastError();
        }
#endif
        plan_note("conv_wino1x8");
        if (zscore) hipLaunchKernelGGL((conv_wino1x8_kernel<true>), dim3((unsigned)n), dim3(512), lds, st, src, n, pk, f, src_row, LayerTaps{});
        else        hipLaunchKernelGGL((conv_wino1x8_kernel<false>), dim3((unsigned)n), dim3(512), lds, st, src, n, pk, f, src_row, LayerTaps{});
        return hipGetLastError();
    }
#if DCE_EXPERIMENTS
    if (tu.conv4 > 0 && feat_bf16 != 2) {           // DCE_CONV4=1: four row tiles per wave (A/B; measured 3.8 % SLOWER end to end, see the kernel's header)
        plan_note("conv_wino2_rt4");
        if (feat_bf16) {
            unsigned short* f = static_cast<unsigned short*>(feat);
            if (zscore) hipLaunchKernelGGL((conv_wino_rt4_kernel<true, unsigned short>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
            else        hipLaunchKernelGGL((conv_wino_rt4_kernel<false, unsigned short>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
        } else {
            float* f = static_cast<float*>(feat);
            if (zscore) hipLaunchKernelGGL((conv_wino_rt4_kernel<true, float>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
            else        hipLaunchKernelGGL((conv_wino_rt4_kernel<false, float>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
        }
        return hipGetLastError();
    }
#endif
    if (feat_bf16 == 2) {                                // DCE_FP32_SPLIT (experiments build): the features leave as three bf16 planes (conv_common.h put_feat3)
#if DCE_EXPERIMENTS
        plan_note("conv_wino2_feat3");
        Feat3* f = static_cast<Feat3*>(feat);
        if (zscore) hipLaunchKernelGGL((conv_wino_kernel<true, Feat3>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
        else        hipLaunchKernelGGL((conv_wino_kernel<false, Feat3>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
        return hipGetLastError();
#else
        return hipErrorInvalidValue;
#endif
    }
    plan_note("conv_wino2");
    if (feat_bf16) {
        unsigned short* f = static_cast<unsigned short*>(feat);
        if (zscore) hipLaunchKernelGGL((conv_wino_kernel<true, unsigned short>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
        else        hipLaunchKernelGGL((conv_wino_kernel<false, unsigned short>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
    } else {
        float* f = static_cast<float*>(feat);
        if (zscore) hipLaunchKernelGGL((conv_wino_kernel<true, float>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{}, t_gate);
        else        hipLaunchKernelGGL((conv_wino_kernel<false, float>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{}, t_gate);
    }
    return hipGetLastError();
}

}  // namespace dce
