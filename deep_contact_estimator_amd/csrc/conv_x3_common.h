// conv_x3_common.h -- device helpers shared by the two conv-stack kernels on three-term bf16 operands: conv_x3.hip (one window per
// workgroup: mid-size batches, layer taps) and conv_x3p.hip (two windows per workgroup, a phase apart: chip-filling batches).
// LDS layout of the activations, slot swizzle, the split into three terms, a layer's GEMM for one wave, the in-place write-back.
#pragma once
#include "conv_common.h"

namespace dce {

typedef __bf16 cx_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 cx_f16x8 __attribute__((ext_vector_type(8)));
typedef float cx_f32x4 __attribute__((ext_vector_type(4)));

constexpr int CX_ROWS1 = 16 * 10 + 2, CX_ROWS2 = 16 * 5 + 2;      // LDS rows a stage reads: every column tile x every tap
constexpr int CX_PLANE = CX_ROWS2 * 256;                          // 20,992 B >= 162 rows x 128 B
constexpr int CX_LDS = 3 * CX_PLANE;                              // 62,976 B
static_assert(CX_ROWS1 * 128 <= CX_PLANE && 2 * CX_LDS <= 160 * 1024 - 2048, "two workgroups per CU");
constexpr int CX_NT = 5;                                          // column tiles per wave

static const int cxCin[4]  = {54, 64, 64, 128};
static const int cxCinP[4] = {64, 64, 64, 128};
static const int cxCout[4] = {64, 64, 128, 128};

// Swizzle of the 16-byte slots of a row.  ds_read_b128 serves 16 lanes a cycle -- lanes {0-3, 12-15, 20-27} and {4-11, 16-19,
// 28-31} of each half wave -- over 64 banks = sixteen 16-byte positions; here lane = (row offset j = lane & 15, slot offset
// g = lane >> 4), and the three taps start at rows = 0, 1, 2 (mod 16).  These two functions make all sixteen positions of every
// cycle distinct for all three alignments (exhaustive check of the lane grouping above; with (row >> 1) & 7 / row & 15 -- the
// swizzles of the GEMM kernels, whose reads start at multiples of 32 rows -- taps 1 and 2 ran into two-way conflicts:
// SQ_LDS_BANK_CONFLICT 2.3e7 of 6.7e7 LDS cycles per launch).  Both repeat every 8 rows, so they do not depend on the column tile.
template <int ROWB> __device__ __forceinline__ int cx_swz(int row) { return ROWB == 128 ? row & 7 : (row & 7) << 1; }

// byte offset of channel ch (bf16) of row `row` inside a plane
template <int ROWB> __device__ __forceinline__ int cx_addr(int row, int ch)
{
    return row * ROWB + (((ch >> 3) ^ cx_swz<ROWB>(row)) << 4) + (ch & 7) * 2;
}

// three terms of two values: p[k] = (term k of v0) | (term k of v1) << 16
__device__ __forceinline__ void cx_split2(float v0, float v1, unsigned (&p)[3])
{
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
#if defined(CX_SCALAR_SPLIT) && CX_SCALAR_SPLIT
    {   // plain v_sub_f32 for the exact remainders (written as instructions: the optimiser re-packs neighbouring subtractions into v_pk_add_f32)
        auto sub = [](float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p[k] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{v0, v1}, b2));
            if (k < 2) { v0 = sub(v0, __builtin_bit_cast(float, p[k] << 16)); v1 = sub(v1, __builtin_bit_cast(float, p[k] & 0xffff0000u)); }
        }
        return;
    }
#endif
    const f2 v = {v0, v1};
    const b2 t1 = __builtin_convertvector(v, b2);
    const f2 r1 = v - __builtin_convertvector(t1, f2);
    const b2 t2 = __builtin_convertvector(r1, b2);
    const f2 r2 = r1 - __builtin_convertvector(t2, f2);
    const b2 t3 = __builtin_convertvector(r2, b2);
    p[0] = __builtin_bit_cast(unsigned, t1); p[1] = __builtin_bit_cast(unsigned, t2); p[2] = __builtin_bit_cast(unsigned, t3);
}

// One layer's GEMM for one wave: acc[rt][ct] += sum over K-steps s = (channel block kb, tap) of W(rt, s) x X(ct, s).
//   ROWB : bytes per LDS row of the layer's input (2 x input channels)      NKB : 32-channel blocks of K
//   xrow : cx_lds + (16 ct0 + j) * ROWB  (this lane's row of column tile 0, tap 0)     sw[tap] = swz(16 ct0 + j + tap)
//   wp   : this wave's packed weights: [step][row tile (2)][plane (3)][lane (64)] x 16 bytes
//   pre  : (PRE) the weight fragments of step 0, requested by the caller ahead of the barrier in front of this layer
struct CxW { uint4 f[2][3]; };                                        // one K-step's weight fragments of a wave: [row tile][plane]
__device__ __forceinline__ void cx_fetch_w0(const uint4* __restrict__ wp, CxW& w)
{
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int p = 0; p < 3; ++p) w.f[rt][p] = wp[(rt * 3 + p) * 64];
}
//   ILV  : the 21 fragment requests of step s+1 are dealt out between the MFMAs of step s (two MFMAs, one request, ...) instead of
//          going out in a bunch ahead of them, during which the matrix pipe runs dry (~220 cycles of a 960-cycle K-step)
//   NT   : terms per operand: 3 = fp32-grade products (six MFMAs each); 2 = ~17 significant bits (a1 b1 + a1 b2 + a2 b1: three MFMAs),
//          for the precision whose features leave rounded to bf16 anyway.  The packed weights keep their three planes either way.
//   F16  : (NT = 2 only; conv_h2.hip) the terms are fp16 -- 11 + 11 significand bits, scaled operands -- on v_mfma_f32_16x16x32_f16, and the
//          packed weights hold two planes
template <int ROWB, int NKB, bool PRE = false, bool ILV = false, int NT = 3, bool F16 = false>
__device__ __forceinline__ void cx_layer(const char* __restrict__ xrow, const int (&sw)[3], int g,
                                         const uint4* __restrict__ wp, cx_f32x4 (&acc)[2][CX_NT], const CxW* pre = nullptr)
{
    constexpr int S = 3 * NKB;
    constexpr int WPL = F16 ? 2 : 3;                                  // planes of the weight pack
    static_assert(NT == 2 || NT == 3, "two or three terms");
    static_assert(!F16 || NT == 2, "fp16 terms come in pairs");
    uint4 af[2][2][NT], bf[2][CX_NT][NT];                             // [buffer][...][plane]
    auto fetch = [&](int s, int b) {                                  // s, b compile-time at every call
        const int kb = s / 3, tap = s % 3;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int p = 0; p < NT; ++p) af[b][rt][p] = (PRE && s == 0) ? pre->f[rt][p] : wp[((s * 2 + rt) * WPL + p) * 64];
        const char* x = xrow + tap * ROWB + (((4 * kb + g) ^ sw[tap]) << 4);
#pragma unroll
        for (int ct = 0; ct < CX_NT; ++ct)
#pragma unroll
            for (int p = 0; p < NT; ++p) bf[b][ct][p] = *reinterpret_cast<const uint4*>(x + ct * 16 * ROWB + p * CX_PLANE);
    };
    // (Requesting the NEXT layer's first weight fragments before the write-back, so that its barriers do not stand in front
    //  of an L2 round trip, was tried: 24 more live registers, 84-100 B of scratch, 394 us instead of 350; again in round 4 with
    //  the kernel at 192 registers and no scratch: 296.1 us instead of 294.2 (profiles/r4h_ab_conv_x3_persist.txt) -- the partner
    //  workgroup's MFMA phase already covers that round trip.)
    fetch(0, 0);
    if constexpr (ILV) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const int b = s & 1;
        if (s + 1 < S) fetch(s + 1, b ^ 1);
        // (left alone, hipcc sinks every fetch to just before its first use and the MFMA stream waits on each of them:
        //  388 us per 4096 windows instead of 350)
        if constexpr (!ILV) __builtin_amdgcn_sched_barrier(0);
        // six terms per product, small ones first; consecutive MFMAs go to different accumulators
        constexpr int NP = NT == 3 ? 6 : 3;
        constexpr int TA[6] = {NT == 3 ? 0 : 1, NT == 3 ? 2 : 0, NT == 3 ? 1 : 0, 0, 1, 0}, TB[6] = {NT == 3 ? 2 : 0, NT == 3 ? 0 : 1, NT == 3 ? 1 : 0, 1, 0, 0};
#pragma unroll
        for (int t = 0; t < NP; ++t)
#pragma unroll
            for (int ct = 0; ct < CX_NT; ++ct)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    if constexpr (F16) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                        __builtin_bit_cast(cx_f16x8, af[b][rt][TA[t]]), __builtin_bit_cast(cx_f16x8, bf[b][ct][TB[t]]), acc[rt][ct], 0, 0, 0);
                    else acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(cx_bf16x8, af[b][rt][TA[t]]), __builtin_bit_cast(cx_bf16x8, bf[b][ct][TB[t]]), acc[rt][ct], 0, 0, 0);
                }
        if constexpr (ILV) {
            if (s + 1 < S) {                                          // the order of this K-step's region: weights (L2) first, then the LDS reads
#pragma unroll
                for (int i = 0; i < 2 * NT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
#pragma unroll
                for (int i = 0; i < NT * CX_NT; ++i) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
                __builtin_amdgcn_sched_group_barrier(0x008, 2 * CX_NT * NP - 2 * (2 * NT + NT * CX_NT), 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ float cx_neighbour(float v)                // the value of lane ^ 1 (the other column of the pool pair)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1 /* quad_perm [1,0,3,2] */, 0xf, 0xf, true));
}

// bias + ReLU (+ MaxPool over column pairs) of a wave's tiles -> three-term planes of the next layer's input, in LDS
//   T : columns of this layer; POOL: the next layer sees T / 2 positions
template <int ROWB_OUT, bool POOL, int T, bool TAPS = false, int NT = 3>
__device__ __forceinline__ void cx_store(char* __restrict__ lds, const cx_f32x4 (&acc)[2][CX_NT],
                                         int co0, int ct0, int j, int g,
                                         float* __restrict__ tap = nullptr, float* __restrict__ tap_pool = nullptr)
{   // tap / tap_pool (TAPS only): this window's (cout, T) block of the layer's output after ReLU / its (cout, T/2) pooled block
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int co = co0 + 16 * rt + 4 * g;
#pragma unroll
        for (int ct = 0; ct < CX_NT; ++ct) {
            const int t = 16 * (ct0 + ct) + j;
            float v[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};      // the bias is the accumulators' initial value
            if constexpr (TAPS) {
                if (t < T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tap[(size_t)(co + r) * T + t] = fmaxf(v[r], 0.f);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = POOL ? fmaxf(fmaxf(v[r], cx_neighbour(v[r])), 0.f) : fmaxf(v[r], 0.f);
            const bool ok = POOL ? ((j & 1) == 0 && (t >> 1) < T / 2) : t < T;
            const int row = (POOL ? (t >> 1) : t) + 1;
            if constexpr (TAPS && POOL) {
                if (ok)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tap_pool[(size_t)(co + r) * (T / 2) + (t >> 1)] = v[r];
            }
            unsigned lo[3], hi[3];
            cx_split2(v[0], v[1], lo);
            cx_split2(v[2], v[3], hi);
            if (ok) {
                char* d = lds + cx_addr<ROWB_OUT>(row, co);
#pragma unroll
                for (int p = 0; p < NT; ++p) *reinterpret_cast<uint2*>(d + p * CX_PLANE) = make_uint2(lo[p], hi[p]);
            }
        }
    }
}


}  // namespace dce
