// conv_h2.hip -- the conv stack (z-score + 4 x [Conv1d k3 p1 + ReLU] + 2 x MaxPool1d(2) + flatten; reference
// src/contact_cnn.py:10-44,61,64 and utils/data_handler.py:55-56) for the precision DCE_FP32_F16X2: fp32-tolerance results from TWO fp16
// terms per operand on v_mfma_f32_16x16x32_f16 -- three MFMAs per product where conv_x3.hip's exact three-term bf16 split needs six.
//
// Arithmetic.  An operand a (activation or weight) is first scaled by a power of two, a' = a * 2^s (exact), then cut into
//     h1 = fp16(a')   (round to nearest even, 11 significand bits)        h2 = fp16(a' - h1)   (the remainder, exact in fp32, again 11 bits)
// so a' = h1 + h2 up to 2^-22 |a'| (two bits short of fp32's 24), and a product is  a'b' ~= h1 k1 + h1 k2 + h2 k1  -- each of the three an
// EXACT fp32 number inside the MFMA's fp32 accumulator (11 x 11 bits); the dropped h2 k2 is <= 2^-22 |a'b'|.  Measured against an fp64
// evaluation the operand rounding costs 0.014 of the logit tolerance (|d| <= 1e-5 max|ref| + 1e-4 |ref|), ten times less than the fp32
// accumulation that every precision of this library shares (tools/emulate_f16x2.py; a two-term bf16 split, 8 + 8 bits, sits AT that
// tolerance, which is why DCE_FP32_SPLIT pays for a third term).
// Range.  fp16 spans 2^-24 .. 65504, so the scales matter -- and they are chosen so that no input can leave the range:
//   * weights: one power of two per layer, fixed when the weights are finalised: max|w| * 2^sw in [2^14, 2^15);
//   * activations: one power of two per WINDOW and LAYER, chosen by the kernel from the layer's largest output: every wave takes the
//     maximum of its accumulators (DPP), one LDS atomic per wave, read back behind the barrier that the in-place write-back needs anyway;
//     the layer then leaves scaled so that its largest value lies in [2^14, 2^15).  Values down to 2^-38 of the window's maximum keep
//     their first term, the matrix pipe honours fp16 subnormals (tools/micro/f16_mfma_probe.hip), what is lost below that is < 2^-39 of
//     the maximum.  The bias rides in as the accumulators' initial value, scaled alike; a window whose activations are so small
//     that the scaled bias would leave fp32's comfortable range (2^60) gets a smaller scale instead (smax: its products are then far
//     below the bias anyway).  The features leave with their scale exponent in feat_scale[window]; fc_gemm_h2.hip takes it off per row.
//   There is no range guard and no fallback in this precision: nothing an input can do moves an operand out of fp16's range
//   (non-finite inputs: NaN features, as everywhere).  A window's result depends on that window alone.
// Kernel: conv_x3.hip's NT = 2 form (one window per four-wave workgroup, two LDS planes = 42 KB -> three workgroups per CU, weights
// streamed from L2 as per-lane packs, features straight from the accumulators in the K order t' * 128 + c) -- see that file for the
// tiling; conv_x3_common.h holds the shared device code.
#include <cfloat>
#include <climits>
#include <algorithm>
#include <cmath>
#include <cstring>

#ifndef CX_ILV
#define CX_ILV 1
#endif
#if defined(H2_EXP) && !DCE_EXPERIMENTS
#error "H2_EXP is a timing probe: build with -DDCE_EXPERIMENTS=1"
#endif
#include "conv_x3_common.h"

namespace dce {

constexpr int H2_SMAX = 180;                                       // scale exponents live in [-114, 180]: 2^127 .. 2^-149 are brought to [2^14, 2^15) (subnormals: to below it)
typedef _Float16 hx_f16x2 __attribute__((ext_vector_type(2)));
typedef float hx_f32x2 __attribute__((ext_vector_type(2)));

// ---- host: scales and packs -------------------------------------------------------------------------------------------------------
// sw with max|w| * 2^sw in [2^14, 2^15) (0 for an all-zero tensor); INT_MIN for a tensor with a non-finite entry
int h2_weight_shift(const float* w, size_t n)
{
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) {
        if (!std::isfinite(w[i])) return INT_MIN;
        m = std::fmax(m, std::fabs(w[i]));
    }
    if (m == 0.f) return 0;
    return 14 - std::ilogb(m);                                         // -113 (the largest fp32 weights are scaled DOWN into fp16's range) .. 163 (fp32 subnormals: ldexp is exact)
}

// the largest scale exponent a layer's INPUT may carry: bias * 2^(S + sw) stays below 2^60 (nothing else bounds it: the exponents are integers
// that only ever enter v_ldexp_f32; H2_SMAX covers fp32's smallest subnormal)
int h2_input_smax(const float* bias, size_t n, int sw)
{
    float m = 0.f;
    for (size_t i = 0; i < n; ++i) {
        if (!std::isfinite(bias[i])) return INT_MIN;
        m = std::fmax(m, std::fabs(bias[i]));
    }
    int s = H2_SMAX;
    if (m > 0.f) s = std::min(s, 60 - sw - (std::ilogb(m) + 1));
    return s < -114 ? -114 : s;
}

static inline unsigned short h2_bits(_Float16 h) { unsigned short u; memcpy(&u, &h, 2); return u; }
// the two fp16 terms of x * 2^sw
void h2_split_host(float x, int sw, unsigned short& t1, unsigned short& t2)
{
    const float v = std::ldexp(x, sw);
    const _Float16 h1 = (_Float16)v;
    const float r = v - (float)h1;
    t1 = h2_bits(h1); t2 = h2_bits((_Float16)r);
}

// a layer's weights (cout, cin, 3) fp32 -> [row-tile pair P][step s = 3 kb + tap][row tile (2)][term (2)][lane (64)][8 fp16], as conv_x3_pack_host
size_t conv_h2_pack_halfs(int l) { return (size_t)cxCout[l] * cxCinP[l] * 3 * 2; }
void conv_h2_pack_host(int l, const float* w, int sw, unsigned short* out)
{
    const int cin = cxCin[l], nkb = cxCinP[l] / 32, cout = cxCout[l];
    size_t o = 0;
    for (int P = 0; P < cout / 32; ++P)
        for (int s = 0; s < 3 * nkb; ++s)
            for (int rt = 0; rt < 2; ++rt)
                for (int p = 0; p < 2; ++p)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = 32 * P + 16 * rt + (lane & 15), ci = 32 * (s / 3) + 8 * (lane >> 4) + e, tap = s % 3;
                            const float v = ci < cin ? w[((size_t)co * cin + ci) * 3 + tap] : 0.f;
                            unsigned short t[2];
                            h2_split_host(v, sw, t[0], t[1]);
                            out[o++] = t[p];
                        }
}

// rows x K fp32 -> fc_gemm_h2.hip's operand layout: [row][K-tile of 32][term (2)][32 fp16] (a row's K-tile = one 128-byte line)
void fc_h2_pack_host(const float* w, size_t rows, size_t K, int sw, unsigned short* out)
{
    for (size_t r = 0; r < rows; ++r)
        for (size_t k = 0; k < K; ++k) {
            unsigned short t1, t2;
            h2_split_host(w[r * K + k], sw, t1, t2);
            unsigned short* d = out + r * 2 * K + (k >> 5) * 64 + (k & 31);
            d[0] = t1; d[32] = t2;
        }
}

// ---- device -----------------------------------------------------------------------------------------------------------------------
namespace {

// the largest of a wave's non-negative values, wave-uniform (v_max_f32 drops NaN)
__device__ __forceinline__ float hx_wave_max(float v)
{
    auto mx = [](float a, int b) { return fmaxf(a, __builtin_bit_cast(float, b)); };
    const int i0 = __builtin_bit_cast(int, v);
    v = mx(v, __builtin_amdgcn_update_dpp(0, i0, 0xB1, 0xf, 0xf, true));                                      // quad_perm [1,0,3,2]
    v = mx(v, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));              // quad_perm [2,3,0,1]
    v = mx(v, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, true));             // row_ror:4
    v = mx(v, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, true));             // row_ror:8
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// t with m * 2^t in [2^14, 2^15) for the bit pattern of a non-negative float m (0 / subnormal: 141; Inf: 0)
__device__ __forceinline__ int hx_shift(unsigned mbits)
{
    const int e = (int)((mbits >> 23) & 0xffu);
    return e == 0xff ? 0 : 141 - e;
}
__device__ __forceinline__ int hx_clamp(int s, int hi) { return s > hi ? hi : s < -114 ? -114 : s; }   // (14 - 127: the largest fp32 values still land in fp16's range)

// the two fp16 terms of two (already scaled) values: p[k] = (term k of v0) | (term k of v1) << 16
__device__ __forceinline__ void hx_split2(float v0, float v1, unsigned (&p)[2])
{
    auto sub = [](float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };   // (plain v_sub_f32: see CX_SCALAR_SPLIT)
    const hx_f16x2 h = __builtin_convertvector(hx_f32x2{v0, v1}, hx_f16x2);                                   // v_cvt_pk_f16_f32, nearest-even
    p[0] = __builtin_bit_cast(unsigned, h);
    const float r0 = sub(v0, (float)h[0]), r1 = sub(v1, (float)h[1]);                                         // exact
    p[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hx_f32x2{r0, r1}, hx_f16x2));
}

// ... of two values times the power of two sc (wave-uniform, an SGPR): v_fma_mixlo/hi_f16 multiply and round to fp16 in one instruction (the
// product with a power of two is exact, so this is the one rounding of v_cvt_pk_f16_f32), v_fma_mix_f32 forms v * sc - h1 with the fp16
// term read in place (exact): five instructions per pair where ldexp + convert + convert back + subtract + convert take eight
__device__ __forceinline__ void hx_split2s(float v0, float v1, float sc, unsigned (&p)[2])
{
    unsigned h;
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v0), "s"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(v1), "s"(sc));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v0), "s"(sc), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v1), "s"(sc), "v"(h));
    p[0] = h;
    p[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(hx_f32x2{r0, r1}, hx_f16x2));
}
__device__ __forceinline__ float hx_pow2(int t) { return __builtin_bit_cast(float, (unsigned)(127 + t) << 23); }      // -126 <= t <= 127

// a wave's largest output of a layer (after ReLU; the pool cannot raise it), columns t < T only, -> one LDS atomic
template <int T>
__device__ __forceinline__ void hx_layer_max(const cx_f32x4 (&acc)[2][CX_NT], int ct0, int j, int lane, unsigned* word)
{
    float m = 0.f;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < CX_NT; ++ct) {
            float q = fmaxf(fmaxf(acc[rt][ct][0], acc[rt][ct][1]), fmaxf(acc[rt][ct][2], acc[rt][ct][3]));
            if (16 * (ct0 + ct) + 15 >= T) q = 16 * (ct0 + ct) + j < T ? q : 0.f;                            // (the last column tile only)
            m = fmaxf(m, q);
        }
    m = hx_wave_max(m);
    if (lane == 0) {
        const unsigned bits = __builtin_bit_cast(unsigned, m) & 0x7fffffffu;              // (non-negative floats order like their bit patterns)
        __hip_atomic_fetch_max(word, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// ReLU (+ MaxPool over column pairs) of a wave's tiles, times 2^shift, -> the two-term planes of the next layer's input, in LDS (in place)
//   TAPS (dce_conv_layer_taps, parity tests): this window's (cout, T) block of the layer's output after ReLU / its (cout, T/2) pooled block also
//   go to HBM in fp32, unscaled (the accumulators carry 2^-unscale)
template <int ROWB_OUT, bool POOL, int T, bool TAPS = false>
__device__ __forceinline__ void hx_store(char* __restrict__ lds, const cx_f32x4 (&acc)[2][CX_NT], int co0, int ct0, int j, int g, int shift,
                                         float* __restrict__ tap = nullptr, float* __restrict__ tap_pool = nullptr, int unscale = 0)
{
    const float sc = hx_pow2(shift);                                   // (next_shift keeps |shift| <= 126)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int co = co0 + 16 * rt + 4 * g;
#pragma unroll
        for (int ct = 0; ct < CX_NT; ++ct) {
            const int t = 16 * (ct0 + ct) + j;
            float v[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};      // the (scaled) bias is the accumulators' initial value
#if defined(H2_EXP) && (H2_EXP & 1)
            // timing probe (WRONG results; experiments build): the un-pooled layers' write-back from lanes j < 8 only -- the sixteen rows a ds_write_b64
            // group of sixteen lanes touches fall on eight 16-byte slots, a two-way bank conflict; eight rows do not conflict.  Same instructions, no replays:
            // what the replays cost (profiles/r6j_conv_h2_bank_conflicts.txt)
            const bool ok = POOL ? ((j & 1) == 0 && (t >> 1) < T / 2) : (t < T && j < 8);
#else
            const bool ok = POOL ? ((j & 1) == 0 && (t >> 1) < T / 2) : t < T;
#endif
            const int row = (POOL ? (t >> 1) : t) + 1;
            if constexpr (TAPS) {
                if (t < T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tap[(size_t)(co + r) * T + t] = __builtin_ldexpf(fmaxf(v[r], 0.f), unscale);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = POOL ? fmaxf(fmaxf(v[r], cx_neighbour(v[r])), 0.f) : fmaxf(v[r], 0.f);
            if constexpr (TAPS && POOL) {
                if (ok)
#pragma unroll
                    for (int r = 0; r < 4; ++r) tap_pool[(size_t)(co + r) * (T / 2) + (t >> 1)] = __builtin_ldexpf(v[r], unscale);
            }
            unsigned lo[2], hi[2];
            hx_split2s(v[0], v[1], sc, lo);
            hx_split2s(v[2], v[3], sc, hi);
            if (ok) {
                char* d = lds + cx_addr<ROWB_OUT>(row, co);
#pragma unroll
                for (int p = 0; p < 2; ++p) *reinterpret_cast<uint2*>(d + p * CX_PLANE) = make_uint2(lo[p], hi[p]);
            }
        }
    }
}

}  // namespace

// TAPS (dce_conv_layer_taps kernel 8, parity tests): every layer's output ALSO goes to HBM in fp32, unscaled, and so do the features, as (n, 4736)
// fp32 in the reference's flatten order c * 37 + t'
// OUT32 (mid-size batches, below fc_gemm_h2.hip's threshold: the FC layers then run on the fp32 kernels): the features leave unscaled, as
// (n, 4736) fp32 in the reference's flatten order, through LDS and 16-byte stores; nothing else is written
// OUTBF (DCE_BF16_FC with the option bf16_conv_h2: BASELINE configs[4] as it is written -- bf16 on the FC layers, conv results of fp32 grade): the
// features leave rounded to bf16 (nearest-even of the unscaled value), (n, 4736) in the K order t' * 128 + c, straight from the accumulators --
// bf16 has fp32's range: no scale, no maximum, no barrier
template <bool ZS, bool TAPS, bool OUT32 = false, bool OUTBF = false>
__global__ __launch_bounds__(256, 3)
void conv_h2_kernel(const float* __restrict__ src, int64_t n, ConvPackH2 pk, unsigned short* __restrict__ feat2, int* __restrict__ feat_scale,
                    LayerTaps taps, float* __restrict__ feat32)
{
    constexpr int LDSB = 2 * CX_PLANE;
    extern __shared__ __attribute__((aligned(16))) char cx_lds[];
    __shared__ unsigned hx_pro[4], hx_max[4];
    const int64_t win = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    (void)n;

    const uint4* w0 = reinterpret_cast<const uint4*>(pk.w[0]) + lane;
    const uint4* w1 = reinterpret_cast<const uint4*>(pk.w[1]) + lane;
    const uint4* w2 = reinterpret_cast<const uint4*>(pk.w[2]) + lane;
    const uint4* w3 = reinterpret_cast<const uint4*>(pk.w[3]) + lane;

    TRACE_MARK(0);
    // ---- prologue: the window (z-scored if ZS), its largest magnitude -> the input's scale, two-term planes [t + 1][channel] (channels 54..63 and
    //      the pad rows zero)
    float x[1][38];
    float2 v[16];
    float tmax = 0.f;
    bool bad = false;
    if constexpr (ZS) {
        load_windows<ZS, 1, 1>(src + win * (int64_t)CH, 0, 1, nullptr, x, tid);
        load_windows<ZS, 1, 2>(src, 0, 1, reinterpret_cast<float*>(cx_lds), x, tid);
        const int gq = tid / CH;
#pragma unroll
        for (int m = 0; m < 38; ++m) {
            bad |= !(fabsf(x[0][m]) <= FLT_MAX);
            if (4 * m + gq < WIN) tmax = fmaxf(tmax, fabsf(x[0][m]));
        }
        if (tid >= 4 * CH) tmax = 0.f;                                 // (threads beyond the 216 loaders hold no samples)
    } else {
        const float2* wsrc = reinterpret_cast<const float2*>(src + win * (int64_t)(WIN * CH));
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = wsrc[tid + 256 * q < WIN * CH / 2 ? tid + 256 * q : 0];
        cc_f32x2 nz = {0.f, 0.f};                                      // x * 0 is 0 for a finite x and NaN for Inf / NaN
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            nz = __builtin_elementwise_fma(cc_f32x2{v[q].x, v[q].y}, cc_f32x2{0.f, 0.f}, nz);
            tmax = fmaxf(tmax, fmaxf(fabsf(v[q].x), fabsf(v[q].y)));
        }
        bad = !(nz.x == 0.f) || !(nz.y == 0.f);
    }
    {
        const bool wbad = __builtin_amdgcn_ballot_w64(bad) != 0;
        const float wmax = hx_wave_max(tmax);
        if (lane == 0) hx_pro[wv] = wbad ? 0xffffffffu : (__builtin_bit_cast(unsigned, wmax) & 0x7fffffffu);
        if (tid < 4) hx_max[tid] = 0u;
    }
    __syncthreads();                                                   // (also: every thread is done with the z-score scratch)
    unsigned pmax = hx_pro[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) pmax = pmax > hx_pro[q] ? pmax : hx_pro[q];
    pmax = __builtin_amdgcn_readfirstlane(pmax);
    const int window_bad = pmax == 0xffffffffu;
    int S = hx_clamp(hx_shift(pmax), pk.smax[0]);                     // the scale exponent of the current layer's INPUT
    {   // zero fill: 2624 x 16 bytes = 10 full rounds of the workgroup + 64
        uint4* z = reinterpret_cast<uint4*>(cx_lds) + tid;
#pragma unroll
        for (int r = 0; r < LDSB / 16 / 256; ++r) z[256 * r] = make_uint4(0, 0, 0, 0);
        if (tid < LDSB / 16 % 256) z[256 * (LDSB / 16 / 256)] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    if constexpr (ZS) {
        if (tid < 4 * CH) {
            const int c = tid % CH, gq = tid / CH;
#pragma unroll
            for (int m = 0; m < 38; m += 2) {                          // rows t = 4 m + gq and 4 (m + 1) + gq
                unsigned p[2];
                hx_split2(__builtin_ldexpf(x[0][m], S), __builtin_ldexpf(x[0][m + 1], S), p);
                const int t0 = 4 * m + gq, t1 = t0 + 4;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    *reinterpret_cast<unsigned short*>(cx_lds + k * CX_PLANE + cx_addr<128>(t0 + 1, c)) = (unsigned short)p[k];
                    if (t1 < WIN) *reinterpret_cast<unsigned short*>(cx_lds + k * CX_PLANE + cx_addr<128>(t1 + 1, c)) = (unsigned short)(p[k] >> 16);
                }
            }
        }
    } else {
        int t = tid / 27, c2 = tid % 27;                               // pair i = tid + 256 q: row i / 27, channels 2 (i % 27), + 1
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            unsigned p[2];
            hx_split2(__builtin_ldexpf(v[q].x, S), __builtin_ldexpf(v[q].y, S), p);
            if (tid + 256 * q < WIN * CH / 2) {
                char* d = cx_lds + cx_addr<128>(t + 1, 2 * c2);
#pragma unroll
                for (int k = 0; k < 2; ++k) *reinterpret_cast<unsigned*>(d + k * CX_PLANE) = p[k];
            }
            t += 9; c2 += 13;                                          // 256 = 9 x 27 + 13
            if (c2 >= 27) { c2 -= 27; t += 1; }
        }
    }
    __syncthreads();
    TRACE_MARK(1);

    cx_f32x4 acc[2][CX_NT];
    auto bias_acc = [&](const float* __restrict__ bias, int co0, int e) {      // the accumulators start from the bias of their four channels, in the products' scale 2^e
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + co0 + 16 * rt + 4 * g);
            const cx_f32x4 b = {__builtin_ldexpf(bv.x, e), __builtin_ldexpf(bv.y, e), __builtin_ldexpf(bv.z, e), __builtin_ldexpf(bv.w, e)};
#pragma unroll
            for (int ct = 0; ct < CX_NT; ++ct) acc[rt][ct] = b;
        }
    };
    // behind a layer's MFMAs: this wave's largest output -> hx_max[l]; behind the barrier: the shift that brings the layer's largest output
    // to [2^14, 2^15), held to the next layer's smax; S moves on to the next layer's input
    auto next_shift = [&](int l) {
        const unsigned m = __builtin_amdgcn_readfirstlane(hx_max[l]);
        const int e = S + pk.sw[l];                                    // the accumulators' scale exponent
        int t = hx_clamp(e + hx_shift(m), pk.smax[l + 1]) - e;
        t = t > 126 ? 126 : t < -126 ? -126 : t;                       // the multiplier 2^t is a float (beyond: a layer whose largest output is below 2^-112 of its products' scale -- any scale serves it)
        S = e + t;
        return t;
    };

    // ---- stage 1 (T = 150, 64 channels in and out): wave = row-tile pair wv & 1, column tiles 5 (wv >> 1) ..
    {
        const int P = wv & 1, ct0 = 5 * (wv >> 1), base = 16 * ct0 + j;
        const int sw[3] = {cx_swz<128>(base), cx_swz<128>(base + 1), cx_swz<128>(base + 2)};
        const char* xrow = cx_lds + base * 128;
        bias_acc(pk.b[0], 32 * P, S + pk.sw[0]);
        cx_layer<128, 2, false, CX_ILV != 0, 2, true>(xrow, sw, g, w0 + (size_t)P * (6 * 2 * 2 * 64), acc);
        hx_layer_max<WIN>(acc, ct0, j, lane, &hx_max[0]);
        TRACE_MARK(2);
        __syncthreads();                                               // every wave has read conv1's input; every wave's maximum is in
        { const int e = S + pk.sw[0];
          hx_store<128, false, WIN, TAPS>(cx_lds, acc, 32 * P, ct0, j, g, next_shift(0), TAPS ? taps.conv1 + win * 64 * 150 : nullptr, nullptr, -e); }
        __syncthreads();
        TRACE_MARK(3);
        bias_acc(pk.b[1], 32 * P, S + pk.sw[1]);
        cx_layer<128, 2, false, CX_ILV != 0, 2, true>(xrow, sw, g, w1 + (size_t)P * (6 * 2 * 2 * 64), acc);
        hx_layer_max<WIN>(acc, ct0, j, lane, &hx_max[1]);
        TRACE_MARK(4);
        __syncthreads();
        { const int e = S + pk.sw[1];                                 // pooled: rows 1..75 of the stage-2 layout (64 channels)
          hx_store<128, true, WIN, TAPS>(cx_lds, acc, 32 * P, ct0, j, g, next_shift(1), TAPS ? taps.conv2 + win * 64 * 150 : nullptr, TAPS ? taps.pool1 + win * 64 * 75 : nullptr, -e); }
        if (tid < 8 * 2) reinterpret_cast<uint4*>(cx_lds + (tid >> 3) * CX_PLANE + 76 * 128)[tid & 7] = make_uint4(0, 0, 0, 0);   // row 76 = right pad
        __syncthreads();
        TRACE_MARK(5);
    }
    // ---- stage 2 (T = 75): wave = row-tile pair wv, all five column tiles
    {
        const int sw[3] = {cx_swz<128>(j), cx_swz<128>(j + 1), cx_swz<128>(j + 2)};
        bias_acc(pk.b[2], 32 * wv, S + pk.sw[2]);
        cx_layer<128, 2, false, CX_ILV != 0, 2, true>(cx_lds + j * 128, sw, g, w2 + (size_t)wv * (6 * 2 * 2 * 64), acc);
        hx_layer_max<75>(acc, 0, j, lane, &hx_max[2]);
        TRACE_MARK(6);
        __syncthreads();
        { const int e = S + pk.sw[2];                                 // 128 channels: 256-byte rows, rows 1..75
          hx_store<256, false, 75, TAPS>(cx_lds, acc, 32 * wv, 0, j, g, next_shift(2), TAPS ? taps.conv3 + win * 128 * 75 : nullptr, nullptr, -e); }
        if (tid < 32 * 2) {                                                            // rows 0 and 76 of the new layout = the zero padding
            const int p = tid >> 5, r = (tid >> 4) & 1, s = tid & 15;
            reinterpret_cast<uint4*>(cx_lds + p * CX_PLANE + (r ? 76 : 0) * 256)[s] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        TRACE_MARK(7);
        const int sw4[3] = {cx_swz<256>(j), cx_swz<256>(j + 1), cx_swz<256>(j + 2)};
        bias_acc(pk.b[3], 32 * wv, S + pk.sw[3]);
        cx_layer<256, 4, false, CX_ILV != 0, 2, true>(cx_lds + j * 256, sw4, g, w3 + (size_t)wv * (12 * 2 * 2 * 64), acc);
        if constexpr (OUTBF) {
            static_assert(!TAPS && !OUT32, "one feature format per instantiation");
            const int e4 = S + pk.sw[3];
            unsigned short* const outb = feat2 + (size_t)win * FEAT;
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int co = 32 * wv + 16 * rt + 4 * g;
#pragma unroll
                for (int ct = 0; ct < CX_NT; ++ct) {
                    const int t = 16 * ct + j;
                    float q[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[r] = __builtin_ldexpf(fmaxf(fmaxf(q[r], cx_neighbour(q[r])), 0.f), -e4);
                    unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(hx_f32x2{q[0], q[1]}, bf2));
                    unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(hx_f32x2{q[2], q[3]}, bf2));
                    if (window_bad) lo = hi = 0x7fc07fc0u;              // a non-finite sample: NaN features
                    if ((j & 1) == 0 && (t >> 1) < 37) *reinterpret_cast<uint2*>(outb + (t >> 1) * 128 + co) = make_uint2(lo, hi);
                }
            }
            TRACE_MARK(9);
            return;
        }
        if constexpr (!OUT32) hx_layer_max<75>(acc, 0, j, lane, &hx_max[3]);
        TRACE_MARK(8);
        __syncthreads();                                               // every wave's maximum is in (the one barrier the features' common scale costs; OUT32: conv4's input is dead)
        if constexpr (OUT32) {
            static_assert(!TAPS, "the taps ride on the product route");
            const int e4 = S + pk.sw[3];
            float* const fl = reinterpret_cast<float*>(cx_lds);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int co = 32 * wv + 16 * rt + 4 * g;
#pragma unroll
                for (int ct = 0; ct < CX_NT; ++ct) {
                    const int t = 16 * ct + j;
                    float q[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) q[r] = __builtin_ldexpf(fmaxf(fmaxf(q[r], cx_neighbour(q[r])), 0.f), -e4);
                    if ((j & 1) == 0 && (t >> 1) < 37) {
                        float* d = fl + co * 37 + (t >> 1);
                        d[0] = q[0]; d[37] = q[1]; d[74] = q[2]; d[111] = q[3];
                    }
                }
            }
            __syncthreads();
            const float nn = __builtin_nanf("");
            for (int q = tid; q < FEAT / 4; q += 256) {
                float4 val = reinterpret_cast<const float4*>(cx_lds)[q];
                if (window_bad) val = make_float4(nn, nn, nn, nn);
                reinterpret_cast<float4*>(feat32 + (size_t)win * FEAT)[q] = val;
            }
            TRACE_MARK(9);
            return;
        }
        // ---- conv4 + ReLU + MaxPool (t = 74 dropped) straight from the accumulators to HBM in the K order k' = t' * 128 + c: a lane holds four
        //      consecutive channels of one pooled position = 8 bytes per term; a row's K-tile of 32 is one 128-byte line [term 1 | term 2]
        const int e4 = S + pk.sw[3];                                   // conv4's accumulators carry 2^e4
        const float scf = hx_pow2(next_shift(3));                      // (S is now the features' scale exponent)
        if (tid == 0) feat_scale[win] = window_bad ? 0 : S;
        unsigned short* const out = feat2 + (size_t)win * (2 * FEAT);
        if (window_bad) {                                              // (wave-uniform) a non-finite sample: NaN in every term of the window's features
            for (int q = tid; q < 2 * FEAT / 8; q += 256) reinterpret_cast<uint4*>(out)[q] = make_uint4(0x7e007e00u, 0x7e007e00u, 0x7e007e00u, 0x7e007e00u);
            if constexpr (TAPS) {
                for (int q = tid; q < FEAT; q += 256) feat32[(size_t)win * FEAT + q] = __builtin_nanf("");
            }
            TRACE_MARK(9);
            return;
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int co = 32 * wv + 16 * rt + 4 * g;
#pragma unroll
            for (int ct = 0; ct < CX_NT; ++ct) {
                const int t = 16 * ct + j;
                float q[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};
                if constexpr (TAPS) {
                    if (t < 75)
#pragma unroll
                        for (int r = 0; r < 4; ++r) taps.conv4[((size_t)win * 128 + co + r) * 75 + t] = __builtin_ldexpf(fmaxf(q[r], 0.f), -e4);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) q[r] = fmaxf(fmaxf(q[r], cx_neighbour(q[r])), 0.f);
                if constexpr (TAPS) {
                    if ((j & 1) == 0 && (t >> 1) < 37)
#pragma unroll
                        for (int r = 0; r < 4; ++r) feat32[(size_t)win * FEAT + (co + r) * 37 + (t >> 1)] = __builtin_ldexpf(q[r], -e4);
                }
                unsigned lo[2], hi[2];
                hx_split2s(q[0], q[1], scf, lo);
                hx_split2s(q[2], q[3], scf, hi);
                if ((j & 1) == 0 && (t >> 1) < 37) {
                    const int k = (t >> 1) * 128 + co;
#pragma unroll
                    for (int p = 0; p < 2; ++p) *reinterpret_cast<uint2*>(out + (k >> 5) * 64 + p * 32 + (k & 31)) = make_uint2(lo[p], hi[p]);
                }
            }
        }
        TRACE_MARK(9);
    }
}

hipError_t init_conv_h2()
{
    hipError_t e;
    for (const void* k : {reinterpret_cast<const void*>(&conv_h2_kernel<true, false>), reinterpret_cast<const void*>(&conv_h2_kernel<false, false>),
                          reinterpret_cast<const void*>(&conv_h2_kernel<false, true>),
                          reinterpret_cast<const void*>(&conv_h2_kernel<true, false, true>), reinterpret_cast<const void*>(&conv_h2_kernel<false, false, true>),
                          reinterpret_cast<const void*>(&conv_h2_kernel<true, false, false, true>), reinterpret_cast<const void*>(&conv_h2_kernel<false, false, false, true>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * CX_PLANE)) != hipSuccess) return e;
    return hipSuccess;
}

// feat2: (n, 148 K-tiles, 2 terms, 32) fp16; feat_scale: (n) scale exponents
hipError_t launch_conv_h2(const float* src, int zscore, int64_t n, const ConvPackH2& pk, unsigned short* feat2, int* feat_scale, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    constexpr int L2T = 2 * CX_PLANE;
    plan_note("conv_h2");
    if (zscore) hipLaunchKernelGGL((conv_h2_kernel<true, false>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, feat2, feat_scale, LayerTaps{}, nullptr);
    else        hipLaunchKernelGGL((conv_h2_kernel<false, false>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, feat2, feat_scale, LayerTaps{}, nullptr);
    return hipGetLastError();
}

// the same stack with (n, 4736) fp32 features out: DCE_FP32_F16X2 at batches below fc_gemm_h2.hip's threshold
hipError_t launch_conv_h2_f32(const float* src, int zscore, int64_t n, const ConvPackH2& pk, float* feat, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    constexpr int L2T = 2 * CX_PLANE;
    plan_note("conv_h2_f32");
    if (zscore) hipLaunchKernelGGL((conv_h2_kernel<true, false, true>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, nullptr, nullptr, LayerTaps{}, feat);
    else        hipLaunchKernelGGL((conv_h2_kernel<false, false, true>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, nullptr, nullptr, LayerTaps{}, feat);
    return hipGetLastError();
}

// ... with (n, 4736) bf16 features out in the K order t' * 128 + c: DCE_BF16_FC with the option bf16_conv_h2 (its fc.0 takes the K-permuted bf16 weights)
hipError_t launch_conv_h2_bf16(const float* src, int zscore, int64_t n, const ConvPackH2& pk, unsigned short* feat, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    constexpr int L2T = 2 * CX_PLANE;
    plan_note("conv_h2_bf16_permk");
    if (zscore) hipLaunchKernelGGL((conv_h2_kernel<true, false, false, true>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, feat, nullptr, LayerTaps{}, nullptr);
    else        hipLaunchKernelGGL((conv_h2_kernel<false, false, false, true>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, feat, nullptr, LayerTaps{}, nullptr);
    return hipGetLastError();
}

// dce_conv_layer_taps, kernel 8: pre-normalised windows through conv_h2_kernel with every layer's output (and the fp32 features) written out
hipError_t launch_conv_h2_taps(const float* windows, int64_t n, const ConvPackH2& pk, unsigned short* feat2, int* feat_scale, float* feat32,
                               const LayerTaps& taps, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL((conv_h2_kernel<false, true>), dim3((unsigned)n), dim3(256), 2 * CX_PLANE, st, windows, n, pk, feat2, feat_scale, taps, feat32);
    return hipGetLastError();
}

}  // namespace dce

// test hook (tests/test_f16x2.py): the host split, as the conv / fc.0 weights get it
extern "C" int dce_debug_split_h2(const float* x, size_t n, unsigned short* terms)
{
    const int sw = dce::h2_weight_shift(x, n);
    if (sw == INT_MIN) return sw;
    for (size_t i = 0; i < n; ++i) dce::h2_split_host(x[i], sw, terms[i], terms[n + i]);
    return sw;
}
