// conv_wino_dev.h -- the device side of the Winograd F(2,3) conv stack (conv_wino.hip has the algorithm and the layouts): constants,
// MFMA main loops, write-backs, and the body of the SEGMENT workgroup (a half / a quarter of a window with halos) as a device function,
// shared by conv_wino.hip (the kernels of the batch path) and latency.hip (the one-window service kernel of the latency mode).
#pragma once
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

// The segment workgroup (512 threads, HLDS_FLOATS floats of LDS at `act`): segment sg of window win0 -> its range of the features.
//   XREG = false: the window is read from src (raw sequence rows if ZS, else (n,150,54) pre-normalised windows), as load_windows does;
//   XREG = true : the caller holds this thread's RAW samples (rows t = 4 m + g of channel c, tid = 54 g + c < 216) in xin and the window
//                 is z-scored here (ZS) -- the latency mode's service kernel keeps the last 150 samples in LDS (latency.hip).
//   COHERENT    : the features are stored with agent-scope (write-through) stores: another workgroup of the SAME kernel reads them
//   XPOSE (> 0) : ... in the micro-batch latency kernel's layout (latency_mb.hip): the features in the order k' = 128 t + channel (not torch's flatten
//                 37 channel + t: the four channels an accumulator register quad holds are then neighbours, and the kernel's fc.0 weights are packed in
//                 the same order), k' of window w at ((k' >> 2) * XPOSE + w) * 4 + (k' & 3) -- the four k' of an MFMA step side by side, XPOSE windows
//                 per 16-byte column -- instead of row w of a (windows, 4736) matrix: one 16-byte store per lane and column tile
template <bool ZS, int NSEG, int NT1, int NT2, bool TAPS = false, bool XREG = false, bool COHERENT = false, int XPOSE = 0>
__device__ __forceinline__ void conv_seg_body(float* __restrict__ act, const float* __restrict__ src, int64_t win0, int sg, const ConvPack& pk,
                                              float* __restrict__ feat, const LayerTaps& taps, const float (*xin)[38] = nullptr, int chalf = -1)
{   // chalf (latency mode): 0 / 1 = this workgroup finishes only output channels 64 chalf .. 64 chalf + 63 of conv4 (a second workgroup on the
    // same segment takes the other half: conv1..3 are computed by both -- on CUs that would idle -- and conv4, 41 % of the stack's MFMAs, is halved)
    static_assert((NSEG == 2 && NT1 == 3 && NT2 == 2) || (NSEG == 4 && NT1 == 2 && NT2 == 1), "column tiles per segment count");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, q = lane >> 4;
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
        if constexpr (XREG) {
#pragma unroll
            for (int m = 0; m < 38; ++m) x[0][m] = xin[0][m];
            load_windows<ZS, 1, 2>(nullptr, 0, 1, act + HACT_FLOATS + 388, x, tid < 256 ? tid : 255);
        } else load_windows<ZS, 1>(src + win0 * wstride, wstride, 1, act + HACT_FLOATS + 388, x, tid < 256 ? tid : 255);
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
        if (chalf >= 0 && (wv >> 2) != chalf) return;                  // (no barrier follows: the other half's waves are done)
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) { const int m = a4 + 16 * nt + j; boff[nt] = 2 * (m < b4 ? m : b4) - 1 - tb2; }
        wino_mfma_deep<RS2H, 32, 1, NT2, PF, 1, 46 % PF>(xrow2, boff, ap4, nullptr, ring, bias_lds + 256, co2, lane, acc);
        const float nanv = __builtin_nanf("");
#pragma unroll
        for (int nt = 0; nt < NT2; ++nt) {
            const int m = a4 + 16 * nt + j;
            if (m <= b4) {
                float* base = feat + win0 * FEAT + (co2 + 4 * q) * 37 + m;
                if constexpr (XPOSE > 0) {
                    // k' = 128 m + channel: this lane's four channels are ONE quad -- one 16-byte agent-scope (sc1, write-through) store
                    typedef unsigned xp_u32x4 __attribute__((ext_vector_type(4)));
                    xp_u32x4 pk4;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float m0 = acc[0][nt][0][r], m1 = acc[0][nt][1][r], m2 = acc[0][nt][2][r], m3 = acc[0][nt][3][r];
                        const float v = fmaxf(fmaxf((m0 + m1) + m2, (m1 - m2) - m3), 0.f);
                        pk4[r] = __float_as_uint(nan0 ? nanv : v);
                    }
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(feat, 0, (FEAT / 4) * XPOSE * 16, 0x00027000);
                    __builtin_amdgcn_raw_buffer_store_b128(pk4, rs, (unsigned)(((m * 32 + (co2 >> 2) + q) * XPOSE + (int)win0) * 16), 0, 16);
                    continue;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float m0 = acc[0][nt][0][r], m1 = acc[0][nt][1][r], m2 = acc[0][nt][2][r], m3 = acc[0][nt][3][r];
                    const float v = fmaxf(fmaxf((m0 + m1) + m2, (m1 - m2) - m3), 0.f);
                    if constexpr (COHERENT) __hip_atomic_store(base + r * 37, nan0 ? nanv : v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else base[r * 37] = nan0 ? nanv : v;
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


}  // namespace dce
