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
#include "conv_common.h"
#include <cstdlib>
#include <type_traits>

namespace dce {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WS1 = 152, RS1 = 2 * WS1 + 2, TP1 = 75;   // RS1 = 306: the transposing prologue stores
                                                        // (lane = channel) then hit 16 banks, not 2
constexpr int WS2 = 78,  RS2 = 2 * WS2, TP2 = 38;
constexpr int WACT_FLOATS = 128 * RS2;                       // 19,968 floats (>= 64*RS1 = 19,584)
constexpr int WLDS_FLOATS = WACT_FLOATS + 384 + 2;           // + biases + 2 NaN flags = 81,416 B
constexpr int WRED_ROW = 56;                                 // fp64 z-score scratch: rows 56..63 of stage 1
static_assert(64 * RS1 <= WACT_FLOATS && WLDS_FLOATS * 4 <= 80 * 1024, "two workgroups per CU");
static_assert(NW == 2, "per-window NaN flags are written for two windows");
static_assert(NW * 4 * 216 <= 8 * RS1 && (WRED_ROW * RS1) % 2 == 0, "z-score scratch fits, 8-B aligned");
constexpr int MT = 2, NTW = 5;                               // row / column tiles per wave
constexpr int WINO1_MAX_N = 256;                             // <= this many windows: conv_wino1_kernel (one window per workgroup)
#if !DCE_EXPERIMENTS && (defined(WINO_EXP) || defined(WINO_PEEL) || defined(WINO_PKINIT) || defined(WINO_RELU_ASM) || defined(WINO1_PF) || defined(WINO_INTERLEAVE) || defined(WINO_PK))
#error "the WINO_* probe / ablation macros are experiment switches: build with -DDCE_EXPERIMENTS=1"
#endif
#ifndef WINO_EXP
#define WINO_EXP 0           // bit flags for ablations / timing probes (tools/micro/wino_loop.hip); 0 in the product:
                             //   2 no LDS reads, 4 one weight line, 8 no input-transform VALU, 16 no output-transform VALU,
                             //   32 no NaN/Inf scan in the prologue, 64 every workgroup loads windows 0,1 (L2-hot source)
#endif
#ifndef WINO_PEEL
#define WINO_PEEL 0
#endif
#ifndef WINO_PKINIT
#define WINO_PKINIT 0
#endif
#ifndef WINO_RELU_ASM
#define WINO_RELU_ASM 0      // 1: ReLU of the write-backs as asm v_max_f32 -- fmaxf() on a value that came out of inline asm (the
                             // packed output transform) is preceded by a canonicalising v_max_f32 v, v, v: 40 extra per plain layer
#endif
#ifndef WINO1_PF
#define WINO1_PF 8           // one-window kernels: weight prefetch depth, K-steps
#endif
#ifndef WINO_INTERLEAVE
#define WINO_INTERLEAVE 0    // 1: deal the input-transform ops of tile i+1 out between the MFMAs of tile i.  Measured (r2k, product
                             // kernel, 3 interleaved rounds): 442.4 us vs 427.0 us bunched -- an op between two MFMAs costs more than
                             // the same op in a bunch ahead of eight back-to-back MFMAs; kept as an experiment switch only
#endif
#ifndef WINO_PK
// Winograd input transform of one column tile (4 adds per lane):
//   0: four plain v_add/v_sub_f32 (asm)   2: two hand-written v_pk_add_f32   1: compiler-chosen packed adds
// Measured in the product kernel (tools/ab_bench.py, 4096 windows, 3 interleaved rounds, r2b): 0 -> 423.5 us,
// 2 -> 436.3 us per launch.  Beside fp32 MFMAs a packed fp32 op costs more matrix-pipe issue time than the
// two plain ops it replaces (MI355X_MICROARCH.md, "price of one filler beside MFMAs"); the isolated
// one-wave loop of tools/micro/wino_loop.hip had ranked them the other way round.
#define WINO_PK 0
#endif

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
// Device helpers
// ------------------------------------------------------------------------------------------
struct A8 { float4 m0, m1; };            // this lane's weights for one K-step: [mt][comp]

template <int MTT = 2>
__device__ __forceinline__ A8 load_a8(const float4* __restrict__ ap, int s)
{   // MTT = 1: ap is pre-offset to this wave's half of the row-tile pair; m1 is never read
#if WINO_EXP & 4
    s = 0;                                               // ablation: every K-step re-reads one L1-hot line
#endif
    A8 a; a.m0 = ap[s * 128]; a.m1 = MTT == 2 ? ap[s * 128 + 1] : a.m0; return a;
}

// the 4 inputs of pair m: (x[2m-1], x[2m], x[2m+1], x[2m+2]) -- two 8-byte LDS reads
__device__ __forceinline__ float4 load_quad(const float* __restrict__ p)
{
    const float2 lo = *reinterpret_cast<const float2*>(p);
    const float2 hi = *reinterpret_cast<const float2*>(p + 2);
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

// One layer's main loop for one wave, software-pipelined by hand at column-tile granularity
// (a tile = 8 MFMAs = 256 matrix-pipe cycles):
//     tile i   : 8 MFMAs on V(i), computed one tile earlier  -> no VALU->MFMA wait states
//     tile i+1 : 4 VALU ops form V(i+1) from the raw quad loaded one tile earlier
//     tile i+2 : its raw quad is requested from LDS now
// and the next K-step's weights (2 x 16 B from L2) are requested at the top of each K-step.
// sched_barrier(0) pins this order: left alone, hipcc sinks every load to just before its use
// (measured: a lone wave then reaches only 55 % of the MFMA issue rate).
//   xrow : act + (lane>>4)*RS               (this lane's channel within the K-step)
//   boff : per column tile, this lane's float offset of pair m inside a row (w*WSEG + 2m)
//   ap   : packed weights of this wave's row-tile pair, + 2*lane float4
typedef float v2f __attribute__((ext_vector_type(2)));
struct Quad { v2f p, q; };               // (d0,d1), (d2,d3) = x[2m-1..2m+2]
struct V4 { v2f a, b; };                 // a = (v0,v3), b = (v1,v2)

__device__ __forceinline__ Quad load_quad2(const float* __restrict__ ptr)
{
    Quad r;
    r.p = *reinterpret_cast<const v2f*>(ptr);
    r.q = *reinterpret_cast<const v2f*>(ptr + 2);
    return r;
}

// Winograd input transform in exactly two packed adds:
//   a = (d0,d1) - (d2,d3) = (v0, v3)          b = (d1 + d2, d2 - d1) = (v1, v2)
__device__ __forceinline__ V4 wino_v(const Quad r)
{
    V4 v;
#if WINO_EXP & 8
    v.a = r.p; v.b = r.q; return v;                      // timing probe: no transform VALU (WRONG results)
#endif
#if WINO_PK == 2
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v.a) : "v"(r.p), "v"(r.q));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[1,0]" : "=v"(v.b) : "v"(r.p), "v"(r.q));
#elif WINO_PK == 0
    // four plain VALU ops, written as asm so that the SLP vectoriser cannot re-pack them
    float v0, v3, v1, v2;
    asm("v_sub_f32 %0, %1, %2" : "=v"(v0) : "v"(r.p.x), "v"(r.q.x));      // d0 - d2
    asm("v_sub_f32 %0, %1, %2" : "=v"(v3) : "v"(r.p.y), "v"(r.q.y));      // d1 - d3
    asm("v_add_f32 %0, %1, %2" : "=v"(v1) : "v"(r.p.y), "v"(r.q.x));      // d1 + d2
    asm("v_sub_f32 %0, %1, %2" : "=v"(v2) : "v"(r.q.x), "v"(r.p.y));      // d2 - d1
    v.a = v2f{v0, v3};
    v.b = v2f{v1, v2};
#else
    v.a = r.p - r.q;
    v.b = v2f{r.q.x, r.q.x} + v2f{r.p.y, -r.p.y};
#endif
    return v;
}

// One K-step (5 column tiles x 8 MFMAs) of the pipelined main loop; see wino_mfma.
// FIRST: the layer's first K-step -- accumulators start from the literal 0 (an inline constant of the
// MFMA's C operand) or, for component 1, from the bias (it enters y[2m] and y[2m+1] with +1), so no
// accumulator-initialisation instructions are ever issued.
template <int RS, bool FIRST, int MT = dce::MT, int NTW = dce::NTW>
__device__ __forceinline__ void wino_step(const float* __restrict__ xs, const float* __restrict__ xn,
                                          const int (&boff)[NTW], const A8 a,
                                          V4& vcur, Quad& rawb, f32x4 (&acc)[MT][NTW][4],
                                          const f32x4 (&bias)[MT], const float* __restrict__ xnn = nullptr)
{
    const float a0[4] = {a.m0.x, a.m0.y, a.m0.z, a.m0.w};
    const float a1[4] = {a.m1.x, a.m1.y, a.m1.z, a.m1.w};
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        // tile i+2 of the (K-step, column tile) sequence: this K-step, the next one, or -- one column tile per wave --
        // the one after that (xnn)
        const float* pc = nt + 2 < NTW ? xs + boff[nt + 2]
                        : nt + 2 < 2 * NTW ? xn + boff[(nt + 2 - NTW) % NTW] : xnn + boff[(nt + 2 - 2 * NTW) % NTW];
#if WINO_EXP & 2
        const Quad rawc = rawb; (void)pc;                  // experiment: no LDS reads
#else
        const Quad rawc = load_quad2(pc);                  // tile i+2
#endif
#if WINO_INTERLEAVE
        // The four transform ops of tile i+1 are dealt out between the MFMAs of tile i (one per MFMA gap,
        // order pinned by sched_barrier) instead of being issued in a bunch ahead of them: a bunch of ~9
        // non-MFMA instructions is longer than the 32-cycle shadow of the MFMA before it.
        V4 vnxt;
        float t0, t3, t1, t2;
        const float v[4] = {vcur.a.x, vcur.b.x, vcur.b.y, vcur.a.y};
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 c0 = FIRST ? (c == 1 ? bias[0] : zero) : acc[0][nt][c];
            acc[0][nt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c], v[c], c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (c == 0) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t0) : "v"(rawb.p.x), "v"(rawb.q.x));      // d0 - d2
            if (c == 1) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t3) : "v"(rawb.p.y), "v"(rawb.q.y));      // d1 - d3
            if (c == 2) asm volatile("v_add_f32 %0, %1, %2" : "=v"(t1) : "v"(rawb.p.y), "v"(rawb.q.x));      // d1 + d2
            if (c == 3) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(t2) : "v"(rawb.q.x), "v"(rawb.p.y));      // d2 - d1
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MT == 2) {
                const f32x4 c1 = FIRST ? (c == 1 ? bias[1] : zero) : acc[1][nt][c];
                acc[1][nt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], v[c], c1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        vnxt.a = v2f{t0, t3};
        vnxt.b = v2f{t1, t2};
#else
        const V4 vnxt = wino_v(rawb);                      // tile i+1
        __builtin_amdgcn_sched_barrier(0);
        const float v[4] = {vcur.a.x, vcur.b.x, vcur.b.y, vcur.a.y};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x4 c0 = FIRST ? (c == 1 ? bias[0] : zero) : acc[0][nt][c];
            acc[0][nt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[c], v[c], c0, 0, 0, 0);
            if constexpr (MT == 2) {
                const f32x4 c1 = FIRST ? (c == 1 ? bias[1] : zero) : acc[1][nt][c];
                acc[1][nt][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[c], v[c], c1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#endif
        vcur = vnxt;
        rawb = rawc;
    }
}

// One layer's main loop for one wave, software-pipelined by hand at column-tile granularity
// (a tile = 8 MFMAs = 256 matrix-pipe cycles):
//     tile i   : 8 MFMAs on V(i), computed one tile earlier  -> no VALU->MFMA wait states
//     tile i+1 : 2 packed adds form V(i+1) from the raw quad loaded one tile earlier
//     tile i+2 : its raw quad is requested from LDS now
// and the weights (2 x 16 B from L2) of K-step s+1 / s+2 are requested at the top of step s / s+1.
// sched_barrier(0) pins this order: left alone, hipcc sinks every load to just before its use
// (measured: a lone wave then reaches only 55 % of the MFMA issue rate).  Two K-steps per loop
// iteration, so the rotating registers (V, raw quad, weights) return to their starting names
// and the back-edge needs no copies.
//   xrow : act + (lane>>4)*RS               (this lane's channel within the K-step)
//   boff : per column tile, this lane's float offset of pair m inside a row (w*WSEG + 2m)
//   ap   : packed weights of this wave's row-tile pair, + 2*lane float4
template <int RS, int STEPS, int MT = dce::MT, int NTW = dce::NTW>
__device__ __forceinline__ void wino_mfma(const float* __restrict__ xrow, const int (&boff)[NTW],
                                          const float4* __restrict__ ap, A8 a_even,
                                          const float* __restrict__ bias_lds, int co0, int lane,
                                          f32x4 (&acc)[MT][NTW][4])
{
    static_assert(STEPS % 2 == 0 && STEPS >= 4, "two K-steps per iteration");
    f32x4 bias[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[mt][r] = bias_lds[co0 + 16 * mt + 4 * (lane >> 4) + r];
    V4 vcur = wino_v(load_quad2(xrow + boff[0]));         // V of tile (0,0)
    Quad rawb = load_quad2(xrow + boff[1]);               // raw of tile (0,1)
    // (peeling the first K-step pair would save these 160 moves per layer but costs more VGPRs
    //  than the 256 available at two workgroups per CU: measured 56 dwords of spill)
#if WINO_PEEL
    // the first K-step pair outside the loop, its first half with the accumulators taken from the literal 0 / the
    // bias as the MFMAs' C operand: no 160 accumulator-init moves per layer
    {
        const A8 a_odd = load_a8<MT>(ap, 1);
        wino_step<RS, true, MT, NTW>(xrow, xrow + 4 * RS, boff, a_even, vcur, rawb, acc, bias);
        a_even = load_a8<MT>(ap, 2);
        wino_step<RS, false, MT, NTW>(xrow + 4 * RS, xrow + 8 * RS, boff, a_odd, vcur, rawb, acc, bias);
    }
#pragma unroll 1
    for (int s = 2; s < STEPS; s += 2) {
#else
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
#if WINO_PKINIT
            // accumulator init on 64-bit moves: 80 instructions per layer instead of 160
            const v2f blo = {bias[mt][0], bias[mt][1]}, bhi = {bias[mt][2], bias[mt][3]};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                v2f lo, hi;
                if (c == 1) {
                    asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(lo) : "v"(blo));
                    asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(hi) : "v"(bhi));
                } else {
                    asm volatile("v_pk_mov_b32 %0, 0, 0" : "=v"(lo));
                    asm volatile("v_pk_mov_b32 %0, 0, 0" : "=v"(hi));
                }
                acc[mt][nt][c] = f32x4{lo.x, lo.y, hi.x, hi.y};
            }
#else
            acc[mt][nt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[mt][nt][1] = bias[mt];
            acc[mt][nt][2] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[mt][nt][3] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
        }
#pragma unroll 1
    for (int s = 0; s < STEPS; s += 2) {
#endif
        const int s2 = s + 2 < STEPS ? s + 2 : s;         // last iteration: harmless re-reads
        const A8 a_odd = load_a8<MT>(ap, s + 1);
        wino_step<RS, false, MT, NTW>(xrow + s * 4 * RS, xrow + (s + 1) * 4 * RS, boff, a_even, vcur, rawb, acc, bias);
        a_even = load_a8<MT>(ap, s2);
        wino_step<RS, false, MT, NTW>(xrow + (s + 1) * 4 * RS, xrow + s2 * 4 * RS, boff, a_odd, vcur, rawb, acc, bias);
    }
}

// The same layer loop for the one-window kernel, where a workgroup runs alone on its CU and a
// K-step (20-24 MFMAs = 640-768 cycles) is far shorter than an L2 / MALL round trip: the packed
// weights are prefetched PF K-steps ahead through a register ring, and the ring runs on into the
// NEXT layer's weights (ap_next) so that the write-back between two layers does not drain it.
// Fully unrolled (<= 32 K-steps), so every ring index is a compile-time constant.
// Ring slot of K-step s is (BASE + s) % PF, BASE = K-steps of all earlier layers (mod PF).
template <int RS, int STEPS, int MT, int NTW, int PF, int MTN, int BASE>   // MTN: row tiles per wave of the next layer
__device__ __forceinline__ void wino_mfma_deep(const float* __restrict__ xrow, const int (&boff)[NTW],
                                               const float4* __restrict__ ap, const float4* __restrict__ ap_next,
                                               A8 (&ring)[PF], const float* __restrict__ bias_lds, int co0, int lane,
                                               f32x4 (&acc)[MT][NTW][4])
{
    static_assert(STEPS >= PF, "ring shorter than the layer");
    f32x4 bias[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[mt][r] = bias_lds[co0 + 16 * mt + 4 * (lane >> 4) + r];
    V4 vcur = wino_v(load_quad2(xrow + boff[0]));
    Quad rawb = load_quad2(NTW > 1 ? xrow + boff[NTW > 1 ? 1 : 0] : xrow + 4 * RS + boff[0]);      // tile 1 of the sequence
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            acc[mt][nt][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[mt][nt][1] = bias[mt];
            acc[mt][nt][2] = f32x4{0.f, 0.f, 0.f, 0.f};
            acc[mt][nt][3] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const A8 a = ring[(BASE + s) % PF];
        if (s + PF < STEPS) ring[(BASE + s) % PF] = load_a8<MT>(ap, s + PF);
        else if (ap_next)   ring[(BASE + s) % PF] = load_a8<MTN>(ap_next, s + PF - STEPS);
        const int sn = s + 1 < STEPS ? s + 1 : s;          // last steps: harmless re-reads
        const int snn = s + 2 < STEPS ? s + 2 : STEPS - 1;
        wino_step<RS, false, MT, NTW>(xrow + s * 4 * RS, xrow + sn * 4 * RS, boff, a, vcur, rawb, acc, bias, xrow + snn * 4 * RS);
    }
}

// Columns of the implicit GEMMs.  One window per workgroup: column n = pair n.  TWO windows: column n = (TP + 1) w + m --
// each window brings one DUMMY column (m = TP) behind its TP pairs.  A window segment is WSEG = 2 (TP + 1) floats wide
// (152 = 2 x 76, 78 = 2 x 39), so pair m of window w sits at float 2 n of the row for EVERY column: the offsets of a wave's
// column tiles differ by compile-time constants (32 floats per tile) and reach the LDS reads as instruction immediates
// instead of one v_add per tile (78 K-steps x 4 VALU instructions per wave: this kernel is issue-bound).
// The dummy column and the fillers past the last one read whatever lies there (still inside the workgroup's LDS) and feed
// accumulators that are never stored; the tile counts do not change (152 columns = 10 tiles, 78 = 5).
template <int TP, int NWIN> struct ColMap {
    static constexpr int TPC = TP + (NWIN > 1 ? 1 : 0);               // columns per window
    __device__ static __forceinline__ bool valid(int n, int& w, int& m)
    {
        w = (NWIN > 1 && n >= TPC) ? 1 : 0;
        m = n - w * TPC;
        return n < NWIN * TPC && m < TP;
    }
};
template <int TP, int WSEG, int NTW = dce::NTW, int NWIN = NW>
__device__ __forceinline__ void col_offsets(int nt0, int j, int (&boff)[NTW])
{
    static_assert(NWIN == 1 || WSEG == 2 * (TP + 1), "two windows: the segment width makes the column map linear");
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = 16 * (nt0 + nt) + j;
        if constexpr (NWIN > 1) boff[nt] = 2 * (16 * nt0 + j) + 32 * nt;      // = 2 n, as base + constant
        else boff[nt] = n < TP ? 2 * n : 0;
    }
}

// Output transform of two accumulator rows at once (r, r+1 of an MFMA result quad sit in consecutive registers):
//   y0 = (m0 + m1) + m2,  y1 = (m1 - m2) - m3   as four v_pk_add_f32 for two pairs instead of eight scalar adds.
// A write-back is ~340 non-MFMA instructions per wave that issue into the breaks of the partner workgroup's MFMA stream
// (profiles/r3b_conv_experiments.txt): fewer instructions, shorter write-backs.  Same operations in the same order per
// element: the same bits.
__device__ __forceinline__ void wino_out2(const f32x4& m0, const f32x4& m1, const f32x4& m2, const f32x4& m3, int h, v2f& y0, v2f& y1)
{
    const v2f a0 = h ? v2f{m0[2], m0[3]} : v2f{m0[0], m0[1]}, a1 = h ? v2f{m1[2], m1[3]} : v2f{m1[0], m1[1]};
    const v2f a2 = h ? v2f{m2[2], m2[3]} : v2f{m2[0], m2[1]}, a3 = h ? v2f{m3[2], m3[3]} : v2f{m3[0], m3[1]};
    y0 = (a0 + a1) + a2;
    // (hipcc selects v_pk_add_f32 for the sums but leaves the differences as scalar v_sub_f32: spelled out)
    v2f u;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(u) : "v"(a1), "v"(a2));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(y1) : "v"(u), "v"(a3));
}

// output transform + ReLU + in-place write-back, no pooling (conv1: T=150, conv3: T=75)
template <int RS, int WSEG, int TP, int T, int MT = dce::MT, int NTW = dce::NTW, int NWIN = NW, bool TAPS = false>
__device__ __forceinline__ void wino_store_plain(float* __restrict__ act, const f32x4 (&acc)[MT][NTW][4],
                                                 int co0, int nt0, int lane, float* __restrict__ tap = nullptr, int cout = 0)
{   // tap (TAPS only): this layer's (windows, cout, T) block of the first window of the workgroup
    const int j = lane & 15, q = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = 16 * (nt0 + nt) + j;
        int w, m;
        if (ColMap<TP, NWIN>::valid(n, w, m)) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v2f y0, y1;
                    wino_out2(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3], h, y0, y1);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 2 * h + e;
                        float* d = act + (co0 + 16 * mt + 4 * q + r) * RS + w * WSEG + 1 + 2 * m;
#if WINO_EXP & 16
                        d[0] = acc[mt][nt][0][r]; d[1] = acc[mt][nt][1][r];     // timing probe: no output-transform VALU (WRONG results)
#else
#if WINO_RELU_ASM
                        float r0, r1;
                        asm("v_max_f32 %0, 0, %1" : "=v"(r0) : "v"(y0[e]));
                        asm("v_max_f32 %0, 0, %1" : "=v"(r1) : "v"(y1[e]));
                        d[0] = r0;
                        d[1] = (T % 2 == 0 || 2 * m + 1 < T) ? r1 : 0.f;
#else
                        d[0] = fmaxf(y0[e], 0.f);
                        d[1] = (T % 2 == 0 || 2 * m + 1 < T) ? fmaxf(y1[e], 0.f) : 0.f;   // index T+1 is a zero pad
#endif
#endif
                        if constexpr (TAPS) {
                            float* tp = tap + ((size_t)w * cout + co0 + 16 * mt + 4 * q + r) * T + 2 * m;
                            tp[0] = d[0];
                            if (2 * m + 1 < T) tp[1] = d[1];
                        }
                    }
                }
        }
    }
}

// output transform + ReLU + MaxPool1d(2,2) -> stage-2 layout (conv2)
template <int MT = dce::MT, int NTW = dce::NTW, int NWIN = NW, bool TAPS = false, int R2 = RS2, int W2 = WS2>
__device__ __forceinline__ void wino_store_pool_stage2(float* __restrict__ act, const f32x4 (&acc)[MT][NTW][4],
                                                       int co0, int nt0, int lane,
                                                       float* __restrict__ tap_conv2 = nullptr, float* __restrict__ tap_pool1 = nullptr)
{
    const int j = lane & 15, q = lane >> 4;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = 16 * (nt0 + nt) + j;
        int w, m;
        if (ColMap<TP1, NWIN>::valid(n, w, m)) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v2f y0, y1;
                    wino_out2(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3], h, y0, y1);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int r = 2 * h + e;
                        // max(relu(y0), relu(y1)) = max(y0, y1, 0)
#if WINO_EXP & 16
                        act[(co0 + 16 * mt + 4 * q + r) * R2 + w * W2 + 1 + m] = acc[mt][nt][0][r];
#else
                        act[(co0 + 16 * mt + 4 * q + r) * R2 + w * W2 + 1 + m] = fmaxf(fmaxf(y0[e], y1[e]), 0.f);
#endif
                        if constexpr (TAPS) {
                            const size_t row = (size_t)w * 64 + co0 + 16 * mt + 4 * q + r;
                            tap_conv2[row * 150 + 2 * m] = fmaxf(y0[e], 0.f);
                            tap_conv2[row * 150 + 2 * m + 1] = fmaxf(y1[e], 0.f);
                            tap_pool1[row * 75 + m] = fmaxf(fmaxf(y0[e], y1[e]), 0.f);
                        }
                    }
                }
        }
    }
}

// output transform + ReLU + MaxPool1d(2,2) (pairs 0..36; t = 74 dropped) + flatten c*37+j -> HBM
template <typename FT, int MT = dce::MT, int NTW = dce::NTW, int NWIN = NW, bool TAPS = false>
__device__ __forceinline__ void wino_store_feat(FT* __restrict__ feat, int64_t win0, int nvalid,
                                                const f32x4 (&acc)[MT][NTW][4], int co0, int lane,
                                                bool nan0, bool nan1, float* __restrict__ tap_conv4 = nullptr, size_t plane_elems = 0)
{
    const int j = lane & 15, q = lane >> 4;
    // one 64-bit address per column tile; the 8 (row tile, r) outputs of a lane sit at compile-time offsets from it
    // (global_store immediates): every VALU instruction outside the MFMA loops is paid for by the MFMA stream of
    // the workgroup sharing the SIMDs
    const float nanv = __builtin_nanf("");
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int n = 16 * nt + j;
        int w, m;
        const bool colv = ColMap<TP2, NWIN>::valid(n, w, m);
        if constexpr (TAPS) {
            if (colv && w < nvalid) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m0 = acc[mt][nt][0][r], m1 = acc[mt][nt][1][r];
                        const float m2 = acc[mt][nt][2][r], m3 = acc[mt][nt][3][r];
                        float* tp = tap_conv4 + ((size_t)w * 128 + co0 + 16 * mt + 4 * q + r) * 75 + 2 * m;
                        tp[0] = fmaxf((m0 + m1) + m2, 0.f);
                        if (2 * m + 1 < 75) tp[1] = fmaxf((m1 - m2) - m3, 0.f);
                    }
            }
        }
        if (colv && m < 37 && w < nvalid) {
            FT* base = feat + (std::is_same<FT, Feat3>::value ? 0 : (win0 + w) * FEAT + (co0 + 4 * q) * 37 + m);
            const bool bad = w ? nan1 : nan0;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    v2f y0, y1;
                    wino_out2(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3], h, y0, y1);
                    float vv[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
#if WINO_EXP & 16
                        vv[e] = acc[mt][nt][0][2 * h + e];
#else
                        vv[e] = fmaxf(fmaxf(y0[e], y1[e]), 0.f);
#endif
                        if (bad) vv[e] = nanv;
                    }
                    if constexpr (std::is_same<FT, Feat3>::value) {
                        const int k0 = (co0 + 4 * q + 16 * mt + 2 * h) * 37 + m;
                        put_feat3(base, plane_elems, win0 + w, k0, k0 + 37, vv[0], vv[1]);
                    } else {
                        put_feat(base + (16 * mt + 2 * h) * 37, vv[0]);
                        put_feat(base + (16 * mt + 2 * h + 1) * 37, vv[1]);
                    }
                }
        }
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
#endif  // DCE_EXPERIMENTS

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

// ------------------------------------------------------------------------------------------
// A SEGMENT of a window per workgroup (<= 128 windows: two CUs per window; <= 64: four).  The layers are local in time,
// so a window can be cut with halos and no exchange: segment sg of NSEG computes the features j = a4..b4 of every
// channel (conv4 pairs a4..b4; pair m of a layer = its outputs 2m, 2m+1) from the rows that reach them,
//     conv3 pairs  a3..b3 = a4-1 .. b4+1          (clipped to 0..37)
//     conv2 pairs  a2..b2 = 2 a3 - 1 .. 2 b3 + 2  (the pooled positions conv3 reads; clipped to 0..74)
//     conv1 pairs  a1..b1 = a2-1 .. b2+1          (clipped to 0..74)
// halves: 42 / 41 / 20 / 19 pairs -> 3 column tiles in stage 1 and 2 in stage 2 instead of 5 and 3 (63 % of a window's
// MFMAs per workgroup); quarters: <= 26 / 24 / 11 / 10 pairs -> 2 and 1 column tiles (37 %).  Columns past a range
// compute on a copy of its last pair and are never stored; the window's true edges keep their zero pads wherever they
// fall into a segment (t = -1; t = 150, 151 in stage 1; t = 75, 76 in stage 2).  Every output is still one accumulator's
// chain over the same K order: bit-identical features.  (The z-score needs the whole window's statistics, so every
// segment loads all 150 rows; only the rows of its own range go to LDS.)
// LDS coordinates: local index L of a row holds x[tb + L]; stage 1: tb1 = 2 a1 - 1, stage 2: tb2 = 2 a3 - 1.
// Waves (eight, two per SIMD; w and w+4 share one): stage 1 = (row tile w&3, column group w>>2), stage 2 = row tile w x NT2
// column tiles.
// ------------------------------------------------------------------------------------------
constexpr int RS1H = 98, RS2H = 70;                          // row strides (floats) of the two stages (halves: 86 / 42 used)
constexpr int HACT_FLOATS = 128 * RS2H;                      // 8960 (>= 64 * RS1H = 6272)
constexpr int HLDS_FLOATS = HACT_FLOATS + 384 + 2 + 864 + 2; // + biases + NaN flag (+pad) + fp64 z-score scratch (8-B aligned)
constexpr int WINOH_MAX_N = 128, WINOQ_MAX_N = 64;

// stage 1 of a segment for one wave: row tile rt x NTW column tiles starting at column tile nt0 (column n <-> pair a + n)
template <int NTW, bool TAPS = false>
__device__ __forceinline__ void seg_stage1(float* __restrict__ act, const float* __restrict__ bias_lds,
                                           const float4* ap1, const float4* ap2, const float4* ap3, A8 (&ring)[WINO1_PF],
                                           int rt, int nt0, int lane, int tid, int a1, int b1, int a2, int b2, int tb1, int tb2,
                                           const LayerTaps& taps, int64_t win0)
{
    constexpr int PF = WINO1_PF;
    const int j = lane & 15, q = lane >> 4;
    f32x4 acc[1][NTW][4];
    int boff[NTW];
    const int co0 = 16 * rt;
    const float* xrow1 = act + q * RS1H;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) { const int m = a1 + 16 * (nt0 + nt) + j; boff[nt] = 2 * (m < b1 ? m : b1) - 1 - tb1; }
    wino_mfma_deep<RS1H, 14, 1, NTW, PF, 1, 0>(xrow1, boff, ap1, ap2, ring, bias_lds, co0, lane, acc);
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int m = a1 + 16 * (nt0 + nt) + j;
        if (m <= b1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[0][nt][0][r], m1 = acc[0][nt][1][r], m2 = acc[0][nt][2][r], m3 = acc[0][nt][3][r];
                float* d = act + (co0 + 4 * q + r) * RS1H + 2 * m - tb1;
                d[0] = fmaxf((m0 + m1) + m2, 0.f);
                d[1] = fmaxf((m1 - m2) - m3, 0.f);
                if constexpr (TAPS) {      // (halo pairs are written by both neighbouring segments: the same bits)
                    float* tp = taps.conv1 + (win0 * 64 + co0 + 4 * q + r) * 150 + 2 * m;
                    tp[0] = d[0]; tp[1] = d[1];
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) { const int m = a2 + 16 * (nt0 + nt) + j; boff[nt] = 2 * (m < b2 ? m : b2) - 1 - tb1; }
    wino_mfma_deep<RS1H, 16, 1, NTW, PF, 1, 14 % PF>(xrow1, boff, ap2, ap3, ring, bias_lds + 64, co0, lane, acc);
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        const int m = a2 + 16 * (nt0 + nt) + j;
        if (m <= b2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float m0 = acc[0][nt][0][r], m1 = acc[0][nt][1][r], m2 = acc[0][nt][2][r], m3 = acc[0][nt][3][r];
                act[(co0 + 4 * q + r) * RS2H + m - tb2] = fmaxf(fmaxf((m0 + m1) + m2, (m1 - m2) - m3), 0.f);
                if constexpr (TAPS) {
                    const int64_t row = win0 * 64 + co0 + 4 * q + r;
                    taps.conv2[row * 150 + 2 * m] = fmaxf((m0 + m1) + m2, 0.f);
                    taps.conv2[row * 150 + 2 * m + 1] = fmaxf((m1 - m2) - m3, 0.f);
                    taps.pool1[row * 75 + m] = fmaxf(fmaxf((m0 + m1) + m2, (m1 - m2) - m3), 0.f);
                }
            }
        }
    }
    // stage-2 edges where they fall into the segment: x[-1], x[75], x[76], all 128 rows (nobody else writes them)
    for (int i = tid; i < 128 * 3; i += 512) {
        const int c = i / 3, k = i % 3, L = (k == 0 ? -1 : 74 + k) - tb2;
        if (L >= 0 && L < RS2H) act[c * RS2H + L] = 0.f;
    }
}

template <bool ZS, int NSEG, int NT1, int NT2, bool TAPS = false>
__global__ __launch_bounds__(512)
void conv_wino_seg_kernel(const float* __restrict__ src, int64_t n, ConvPack pk, float* __restrict__ feat,
                          const long long* __restrict__ src_row, LayerTaps taps)
{
    static_assert((NSEG == 2 && NT1 == 3 && NT2 == 2) || (NSEG == 4 && NT1 == 2 && NT2 == 1), "column tiles per segment count");
    extern __shared__ __attribute__((aligned(16))) float act[];
    if (src_row) src += *src_row * CH;                    // online graph: the window start lives in device memory
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
    const int64_t win0 = blockIdx.x / NSEG;
    const int sg = blockIdx.x % NSEG;
    if (win0 >= n) return;
    // the segment's pair ranges (wave-uniform integers)
    const int a4 = NSEG == 2 ? (sg ? 19 : 0) : (sg == 0 ? 0 : 1 + 9 * sg), b4 = NSEG == 2 ? (sg ? 36 : 18) : 9 + 9 * sg;
    const int a3 = a4 > 0 ? a4 - 1 : 0, b3 = b4 + 1 < 37 ? b4 + 1 : 37;
    const int a2 = 2 * a3 - 1 > 0 ? 2 * a3 - 1 : 0, b2 = 2 * b3 + 2 < 74 ? 2 * b3 + 2 : 74;
    const int a1 = a2 > 0 ? a2 - 1 : 0, b1 = b2 + 1 < 74 ? b2 + 1 : 74;
    const int tb1 = 2 * a1 - 1, tb2 = 2 * a3 - 1;

    for (int i = tid; i < 384; i += 512) {
        const int l = i < 64 ? 0 : (i < 128 ? 1 : (i < 256 ? 2 : 3));
        const int o = i < 64 ? i : (i < 128 ? i - 64 : (i < 256 ? i - 128 : i - 256));
        act[HACT_FLOATS + i] = pk.b[l][o];
    }
    int* nanflag = reinterpret_cast<int*>(act + HACT_FLOATS + 384);
    if (tid == 0) nanflag[0] = 0;
    constexpr int PF = WINO1_PF;
    A8 ring[PF];
    const int rt = wv & 3;
    const float4* ap1 = reinterpret_cast<const float4*>(pk.ww[0]) + (rt >> 1) * (14 * 128) + 2 * lane + (rt & 1);
    const float4* ap2 = reinterpret_cast<const float4*>(pk.ww[1]) + (rt >> 1) * (16 * 128) + 2 * lane + (rt & 1);
    const float4* ap3 = reinterpret_cast<const float4*>(pk.ww[2]) + (wv >> 1) * (16 * 128) + 2 * lane + (wv & 1);
    const float4* ap4 = reinterpret_cast<const float4*>(pk.ww[3]) + (wv >> 1) * (32 * 128) + 2 * lane + (wv & 1);
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = load_a8<1>(ap1, i);
    {
        float x[1][38];
        const int64_t wstride = ZS ? CH : (int64_t)WIN * CH;
        load_windows<ZS, 1>(src + win0 * wstride, wstride, 1, act + HACT_FLOATS + 388, x, tid < 256 ? tid : 255);
        bool bad0 = false;
#pragma unroll
        for (int m = 0; m < 38; ++m) bad0 |= !(fabsf(x[0][m]) <= 3.0e38f);
        __syncthreads();
        if (bad0 && tid < 4 * CH) nanflag[0] = 1;
        if (tid < 4 * CH) {
            const int c = tid % CH, g = tid / CH;
#pragma unroll
            for (int m = 0; m < 38; ++m) {
                const int t = 4 * m + g, L = t - tb1;
                if (t < WIN && L >= 0 && L < RS1H) act[c * RS1H + L] = x[0][m];
            }
        }
        // the window's true edges where they fall into the segment: x[-1], x[150], x[151]; filler channels 54, 55
        for (int i = tid; i < 64 * 3; i += 512) {
            const int c = i / 3, k = i % 3, L = (k == 0 ? -1 : 149 + k) - tb1;
            if (L >= 0 && L < RS1H) act[c * RS1H + L] = 0.f;
        }
        for (int i = tid; i < 2 * RS1H; i += 512) act[CH * RS1H + i] = 0.f;
    }
    __syncthreads();
    const bool nan0 = __builtin_amdgcn_readfirstlane(nanflag[0]) != 0;
    const float* bias_lds = act + HACT_FLOATS;

    // ---- stage 1: conv1, conv2; wave (row tile w&3, column group w>>2): halves = column tiles {0,1} / {2}, quarters = {0} / {1}
    //      (waves w and w+4 share a SIMD: every SIMD carries NT1 column tiles of one row tile)
    if constexpr (NSEG == 4) seg_stage1<1, TAPS>(act, bias_lds, ap1, ap2, ap3, ring, rt, wv >> 2, lane, tid, a1, b1, a2, b2, tb1, tb2, taps, win0);
    else if (wv < 4)         seg_stage1<2, TAPS>(act, bias_lds, ap1, ap2, ap3, ring, rt, 0, lane, tid, a1, b1, a2, b2, tb1, tb2, taps, win0);
    else                     seg_stage1<1, TAPS>(act, bias_lds, ap1, ap2, ap3, ring, rt, 2, lane, tid, a1, b1, a2, b2, tb1, tb2, taps, win0);
    // ---- stage 2 (all eight waves): conv3, conv4 for row tile wv x NT2 column tiles
    {
        f32x4 acc[1][NT2][4];
        int boff[NT2];
        const int co2 = 16 * wv;
        const float* xrow2 = act + q * RS2H;
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) { const int m = a3 + 16 * nt + j; boff[nt] = 2 * (m < b3 ? m : b3) - 1 - tb2; }
        __syncthreads();
        wino_mfma_deep<RS2H, 16, 1, NT2, PF, 1, 30 % PF>(xrow2, boff, ap3, ap4, ring, bias_lds + 128, co2, lane, acc);
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const int m = a3 + 16 * nt + j;
            if (m <= b3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m0 = acc[0][nt][0][r], m1 = acc[0][nt][1][r], m2 = acc[0][nt][2][r], m3 = acc[0][nt][3][r];
                    float* d = act + (co2 + 4 * q + r) * RS2H + 2 * m - tb2;
                    d[0] = fmaxf((m0 + m1) + m2, 0.f);
                    d[1] = 2 * m + 1 < 75 ? fmaxf((m1 - m2) - m3, 0.f) : 0.f;        // x[75] is a zero pad
                    if constexpr (TAPS) {
                        float* tp = taps.conv3 + (win0 * 128 + co2 + 4 * q + r) * 75 + 2 * m;
                        tp[0] = d[0];
                        if (2 * m + 1 < 75) tp[1] = d[1];
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) { const int m = a4 + 16 * nt + j; boff[nt] = 2 * (m < b4 ? m : b4) - 1 - tb2; }
        wino_mfma_deep<RS2H, 32, 1, NT2, PF, 1, 46 % PF>(xrow2, boff, ap4, nullptr, ring, bias_lds + 256, co2, lane, acc);
        const float nanv = __builtin_nanf("");
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const int m = a4 + 16 * nt + j;
            if (m <= b4) {
                float* base = feat + win0 * FEAT + (co2 + 4 * q) * 37 + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m0 = acc[0][nt][0][r], m1 = acc[0][nt][1][r], m2 = acc[0][nt][2][r], m3 = acc[0][nt][3][r];
                    const float v = fmaxf(fmaxf((m0 + m1) + m2, (m1 - m2) - m3), 0.f);
                    base[r * 37] = nan0 ? nanv : v;
                    if constexpr (TAPS) {      // conv4 before the pool: pairs a4..b4 (t = 74, which the pool drops, is never computed here)
                        float* tp = taps.conv4 + (win0 * 128 + co2 + 4 * q + r) * 75 + 2 * m;
                        tp[0] = fmaxf((m0 + m1) + m2, 0.f);
                        tp[1] = fmaxf((m1 - m2) - m3, 0.f);
                    }
                }
            }
        }
    }
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
    if ((e = grant_wino_lds<true, Feat3>()) != hipSuccess) return e;
    if ((e = grant_wino_lds<false, Feat3>()) != hipSuccess) return e;
    for (const void* k : {reinterpret_cast<const void*>(&conv_wino1_kernel<true>), reinterpret_cast<const void*>(&conv_wino1_kernel<false>),
                          reinterpret_cast<const void*>(&conv_wino1x8_kernel<true>), reinterpret_cast<const void*>(&conv_wino1x8_kernel<false>)})
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
    for (const void* k : {reinterpret_cast<const void*>(&conv_wino_kernel<false, float, true>), reinterpret_cast<const void*>(&conv_wino1_kernel<false, true>),
                          reinterpret_cast<const void*>(&conv_wino1x8_kernel<false, true>)})
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
    case 2: hipLaunchKernelGGL((conv_wino_seg_kernel<false, 2, 3, 2, true>), dim3((unsigned)(2 * n)), dim3(512), hl, st, src, n, pk, f, none, taps); break;
    case 3: hipLaunchKernelGGL((conv_wino_seg_kernel<false, 4, 2, 1, true>), dim3((unsigned)(4 * n)), dim3(512), hl, st, src, n, pk, f, none, taps); break;
#if DCE_EXPERIMENTS
    case 6: hipLaunchKernelGGL((conv_wino_rt4_kernel<false, float, true>), dim3((unsigned)((n + NW - 1) / NW)), dim3(256), lds, st, src, n, pk, f, none, taps); break;
#endif
    case 5: hipLaunchKernelGGL((conv_wino1_kernel<false, true>), dim3((unsigned)n), dim3(256), lds, st, src, n, pk, f, none, taps); break;
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
            plan_note("conv_wino_quarter");
            if (zscore) hipLaunchKernelGGL((conv_wino_seg_kernel<true, 4, 2, 1>), dim3((unsigned)(4 * n)), dim3(512), hl, st, src, n, pk, f, src_row, LayerTaps{});
            else        hipLaunchKernelGGL((conv_wino_seg_kernel<false, 4, 2, 1>), dim3((unsigned)(4 * n)), dim3(512), hl, st, src, n, pk, f, src_row, LayerTaps{});
            return hipGetLastError();
        }
        if (n <= half_max) {
            plan_note("conv_wino_half");
            if (zscore) hipLaunchKernelGGL((conv_wino_seg_kernel<true, 2, 3, 2>), dim3((unsigned)(2 * n)), dim3(512), hl, st, src, n, pk, f, src_row, LayerTaps{});
            else        hipLaunchKernelGGL((conv_wino_seg_kernel<false, 2, 3, 2>), dim3((unsigned)(2 * n)), dim3(512), hl, st, src, n, pk, f, src_row, LayerTaps{});
            return hipGetLastError();
        }
        const bool w8 = tu.wino1_w8;
        if (w8) {
            plan_note("conv_wino1x8");
            if (zscore) hipLaunchKernelGGL((conv_wino1x8_kernel<true>), dim3((unsigned)n), dim3(512), lds, st, src, n, pk, f, src_row, LayerTaps{});
            else        hipLaunchKernelGGL((conv_wino1x8_kernel<false>), dim3((unsigned)n), dim3(512), lds, st, src, n, pk, f, src_row, LayerTaps{});
            return hipGetLastError();
        }
        plan_note("conv_wino1x4");
        if (zscore) hipLaunchKernelGGL((conv_wino1_kernel<true>), dim3((unsigned)n), block, lds, st, src, n, pk, f, src_row, LayerTaps{});
        else        hipLaunchKernelGGL((conv_wino1_kernel<false>), dim3((unsigned)n), block, lds, st, src, n, pk, f, src_row, LayerTaps{});
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
    if (feat_bf16 == 2) {                                // DCE_FP32_SPLIT: the features leave as three bf16 planes (conv_common.h put_feat3)
        plan_note("conv_wino2_feat3");
        Feat3* f = static_cast<Feat3*>(feat);
        if (zscore) hipLaunchKernelGGL((conv_wino_kernel<true, Feat3>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
        else        hipLaunchKernelGGL((conv_wino_kernel<false, Feat3>), grid, block, lds, st, src, n, pk, f, src_row, LayerTaps{});
        return hipGetLastError();
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
