// conv_x3.hip -- the conv stack (z-score + 4 x [Conv1d k3 p1 + ReLU] + 2 x MaxPool1d(2) + flatten; reference
// src/contact_cnn.py:10-44,61,64 and utils/data_handler.py:55-56) with fp32 results on the bf16 matrix pipe: every fp32
// operand -- activations and weights -- enters the MFMAs as THREE bf16 terms (a = a1 + a2 + a3 exactly), six bf16 x bf16
// MFMAs per product, fp32 accumulate (fc_gemm_x3.hip has the arithmetic).  Used by the DCE_FP32_SPLIT precision from 128 windows per
// call (features out as three bf16 planes for fc_gemm_x3.hip, or as fp32 below that kernel's threshold) and by DCE_BF16_FC (features
// out rounded to bf16: the first term of the last split; on two-term operands, see NT = 2 below); never by the default DCE_FP32 precision.
//
// Why not Winograd here: its input transform would have to produce three-term operands on the fly (six more VALU
// operations per transformed value, on a kernel that is already bound by what it issues between MFMAs).  In the direct form
// the operands of a layer ARE the previous layer's outputs: they are split once, in the write-back, and every MFMA operand
// is one 16-byte LDS read.  The direct form issues 1.5x the MACs of F(2,3); six bf16 MFMAs per product at 16x the fp32 rate
// still make it 2.4x fewer matrix-pipe cycles than the fp32 Winograd kernel (28.8k per SIMD and window against 49.9k x 2 / 2).
//
// One workgroup = 4 waves = ONE window; 63 KB of LDS -> two workgroups per CU.  Activations live in LDS as three bf16
// planes, [position][channel] (position p = t + 1; rows 0 and T + 1 are the zero padding): the B operand of
// v_mfma_f32_16x16x32_bf16 -- lane (j = column = position, g): 8 consecutive channels -- is then ONE ds_read_b128 per plane,
// for every tap (tap k reads row t + k).  16-byte slot s of row r is stored at s ^ swz(r) (cx_swz below) so that the rows a
// read touches spread over all banks for every tap; the swizzle of row t + k + 16 ct does not depend on the column tile ct,
// so a K-step needs one address register and 15 immediates.
// GEMM per layer: M = Cout (16-row tiles), N = positions (16-column tiles: 150 -> 10, 75 -> 5), K = 3 taps x Cin in steps
// of 32 channels.  A wave holds 2 row tiles x 5 column tiles (40 accumulator registers); per K-step it reads 6 weight
// fragments (packed per lane on the host, three planes, streamed from L2) and 15 activation fragments and issues 60 MFMAs;
// the fragments of step s+1 are requested before the MFMAs of step s.  Write-back (in place, between two barriers; the bias
// is the accumulators' initial value): ReLU, MaxPool (neighbouring columns sit in neighbouring lanes: one DPP quad_perm), split
// into the three terms (v_cvt_pk_bf16_f32), 8-byte stores.  conv4 + pool go through LDS once more and leave as 16-byte stores.
//
// NT = 2 (round 4; DCE_BF16_FC only, at every batch size and in the online pushes; plan conv_x2_bf16*): the same kernel on TWO terms per
// operand -- a1 b1 + a1 b2 + a2 b1, three MFMAs per product, ~17 significant bits in front of features that leave rounded to bf16 (8 bits).
// The third plane is neither written nor read (the packed weights keep theirs, the kernel skips it): 42 KB of LDS and 146 registers
// -> THREE workgroups per CU, i.e. three independent waves per SIMD to cover each other's prologues and write-backs: 170 us per 4096
// windows against 308 for NT = 3 in the same mode, 0.84 of the matrix pipe while workgroups are resident (the board's ceiling for a
// dense bf16 stream), HBM traffic 1.02 x algorithmic; the mode's error against an fp64 evaluation is the same with two terms and with
// three (profiles/r4h_bf16_terms_audit.json).  DCE_FP32_SPLIT keeps NT = 3: its contract is the fp32 tolerance.
#include <cfloat>
#include <cstring>
#include <type_traits>

#ifndef CX_ILV
#define CX_ILV 1        // the fragment requests of K-step s + 1 dealt out between the MFMAs of step s instead of going out in a bunch ahead of them (A/B: -DCX_ILV=0; 319.5 -> 312.4 us per 4096 windows)
#endif
#ifndef CX_SCALAR_SPLIT
#define CX_SCALAR_SPLIT 1   // the split's exact remainders on v_sub_f32 instead of v_pk_add_f32 (A/B: -DCX_SCALAR_SPLIT=0; 319.5 -> 311.7 us, both: 308.4; same bits)
#endif

#include "conv_x3_common.h"

namespace dce {

// Host: a layer's weights (cout, cin, 3) fp32 -> [row-tile pair P][step s = 3 kb + tap][row tile (2)][plane (3)][lane (64)][8 bf16]:
// lane (i, g) of row tile rt holds W[32 P + 16 rt + i][32 kb + 8 g + e][tap], e = 0..7 (zero beyond cin), as its three terms
size_t conv_x3_pack_halfs(int l) { return (size_t)cxCout[l] * cxCinP[l] * 3 * 3; }

void conv_x3_pack_host(int l, const float* w, unsigned short* out)
{
    const int cin = cxCin[l], nkb = cxCinP[l] / 32, cout = cxCout[l];
    size_t o = 0;
    for (int P = 0; P < cout / 32; ++P)
        for (int s = 0; s < 3 * nkb; ++s)
            for (int rt = 0; rt < 2; ++rt)
                for (int p = 0; p < 3; ++p)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int e = 0; e < 8; ++e) {
                            const int co = 32 * P + 16 * rt + (lane & 15), ci = 32 * (s / 3) + 8 * (lane >> 4) + e, tap = s % 3;
                            const float v = ci < cin ? w[((size_t)co * cin + ci) * 3 + tap] : 0.f;
                            unsigned short t[3];
                            split3_host(&v, 1, 0, t);
                            out[o++] = t[p];
                        }
}

template <bool ZS, bool TAPS = false, int OUT = 0, bool PERMK = false, bool PERSIST = false, int NT = 3>
__global__ __launch_bounds__(256, NT == 2 ? 3 : 2)
void conv_x3_kernel(const float* __restrict__ src, int64_t n, ConvPackX3 pk, unsigned short* __restrict__ feat3, size_t plane_elems,
                    LayerTaps taps = LayerTaps{}, float* __restrict__ feat32 = nullptr, const long long* __restrict__ src_row = nullptr,
                    GuardArgs guard = GuardArgs{})
{   // guard (DCE_FP32_SPLIT on pre-normalised windows; dce_kernels.h GuardArgs): the load stage also takes the window's largest magnitude;
    // a window above x_hi (a layer's operand could reach bf16's largest finite value: the first term of a split would round to Inf) or
    // entirely below x_lo (third terms would go subnormal) writes the launch's generation to guard.word[0] -- the gated DCE_FP32 sequence
    // behind this launch then recomputes it -- and counts itself in guard.word[1].
    // TAPS (dce_conv_layer_taps, parity tests): every layer's output also goes to HBM in fp32, and so do the features.
    // OUT: 0 the features leave as three bf16 planes (fc_gemm_x3.hip's layout); 1 as (n, 4736) fp32 in feat32 (batches below
    // the split-bf16 fc.0 kernel's threshold, whose fc.0 runs on the fp32 kernels); 2 as (n, 4736) bf16, round-to-nearest-even,
    // in feat3 (the DCE_BF16_FC precision, whose FC layers take bf16 operands)
    // PERSIST (with PERMK only; experiments build, DCE_X3_PERSIST=1 -- measured 2-4 % SLOWER than one workgroup per window, with and
    // without a start offset between the two workgroups of a CU: profiles/r4h_ab_conv_x3_persist.txt): gridDim.x workgroups walk the windows blockIdx.x, + gridDim.x, ..; the next window's samples are
    // requested when conv3's MFMAs are through, so that their HBM latency passes under conv3's write-back and conv4 instead of
    // opening the next prologue (the loads sit in registers: 38 per thread z-scored, 32 pre-normalised)
    // NT = 2 (DCE_BF16_FC only, OUT = 2): two-term operands, three MFMAs per product -- ~17 significant bits through the four layers,
    // 2^8 finer than the bf16 rounding the features leave with; two planes of LDS (42 KB) and ~140 registers: three workgroups per CU
    static_assert(!PERSIST || (PERMK && !TAPS), "the persistent form has the register-only feature tail");
    static_assert(NT == 3 || (OUT == 2 && !TAPS), "two-term operands only where the features are rounded to bf16");
    constexpr int LDSB = NT * CX_PLANE;
    if (src_row) src += *src_row * CH;                                 // online graph: the window start lives in device memory
    extern __shared__ __attribute__((aligned(16))) char cx_lds[];
    float x[1][38];
    float2 v[16];
    auto request = [&](int64_t w, int tid_) {
        if constexpr (ZS) load_windows<ZS, 1, 1>(src + w * (int64_t)CH, 0, 1, nullptr, x, tid_);
        else {
            // pre-normalised window: 4050 pairs of neighbouring channels, 16 per thread -- one 8-byte load, one split, three
            // 4-byte LDS stores each (the per-channel mapping of load_windows costs 38 loads and 114 two-byte stores per thread)
            const float2* wsrc = reinterpret_cast<const float2*>(src + w * (int64_t)(WIN * CH));
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = wsrc[tid_ + 256 * q < WIN * CH / 2 ? tid_ + 256 * q : 0];
        }
    };
    if constexpr (PERSIST) request(blockIdx.x, threadIdx.x);
    for (int64_t win = blockIdx.x;;) {
    int tid = threadIdx.x;
    if constexpr (PERSIST) asm volatile("" : "+v"(tid));              // (or the address arithmetic of every phase is hoisted out of the loop and spilled)
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;

    const uint4* w0 = reinterpret_cast<const uint4*>(pk.w[0]) + lane;
    const uint4* w1 = reinterpret_cast<const uint4*>(pk.w[1]) + lane;
    const uint4* w2 = reinterpret_cast<const uint4*>(pk.w[2]) + lane;
    const uint4* w3 = reinterpret_cast<const uint4*>(pk.w[3]) + lane;

    TRACE_MARK(0);
    // ---- prologue: the window (z-scored if ZS) -> three-term planes, [t + 1][channel], channels 54..63 and the pad rows zero
    if constexpr (!PERSIST) request(win, tid);
    int window_bad;
    if constexpr (ZS) {
        load_windows<ZS, 1, 2>(src, 0, 1, reinterpret_cast<float*>(cx_lds), x, tid);
        bool bad = false;
#pragma unroll
        for (int m = 0; m < 38; ++m) bad |= !(fabsf(x[0][m]) <= FLT_MAX);
        window_bad = __syncthreads_or(bad);                            // (also: every thread is done with the z-score scratch)
        for (int i = tid; i < LDSB / 16; i += 256) reinterpret_cast<uint4*>(cx_lds)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        if (tid < 4 * CH) {
            const int c = tid % CH, gq = tid / CH;
#pragma unroll
            for (int m = 0; m < 38; m += 2) {                          // rows t = 4 m + gq and 4 (m + 1) + gq
                unsigned p[3];
                cx_split2(x[0][m], x[0][m + 1], p);
                const int t0 = 4 * m + gq, t1 = t0 + 4;
#pragma unroll
                for (int k = 0; k < NT; ++k) {
                    *reinterpret_cast<unsigned short*>(cx_lds + k * CX_PLANE + cx_addr<128>(t0 + 1, c)) = (unsigned short)p[k];
                    if (t1 < WIN) *reinterpret_cast<unsigned short*>(cx_lds + k * CX_PLANE + cx_addr<128>(t1 + 1, c)) = (unsigned short)(p[k] >> 16);
                }
            }
        }
    } else {
        int t = tid / 27, c2 = tid % 27;                               // pair i = tid + 256 q: row i / 27, channels 2 (i % 27), + 1
        {   // zero fill: 3936 x 16 bytes = 15 full rounds of the workgroup + 96 (constant offsets: no loop bookkeeping)
            uint4* z = reinterpret_cast<uint4*>(cx_lds) + tid;
#pragma unroll
            for (int r = 0; r < LDSB / 16 / 256; ++r) z[256 * r] = make_uint4(0, 0, 0, 0);
            if (tid < LDSB / 16 % 256) z[256 * (LDSB / 16 / 256)] = make_uint4(0, 0, 0, 0);
        }
        // non-finite scan: x * 0 is 0 for a finite x and NaN for Inf / NaN (16 packed FMAs; a chain of compares compiles
        // to five instructions per value)
        cc_f32x2 nz = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 16; ++q) nz = __builtin_elementwise_fma(cc_f32x2{v[q].x, v[q].y}, cc_f32x2{0.f, 0.f}, nz);
        if (guard.word == nullptr) window_bad = __syncthreads_or(!(nz.x == 0.f) || !(nz.y == 0.f));      // (also: the zero fill is complete)
        else {
            // the range guard rides on the same barrier: four facts per thread -> per wave (ballots) -> per window (one LDS word per wave)
            __shared__ unsigned cx_flags[4];
            float tmax = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tmax = fmaxf(tmax, fmaxf(fabsf(v[q].x), fabsf(v[q].y)));        // (v_max drops NaN; Inf is above every x_hi)
            unsigned wbits = 0;
            if (__builtin_amdgcn_ballot_w64(!(nz.x == 0.f) || !(nz.y == 0.f)) != 0) wbits |= 1u;        // a non-finite sample
            if (__builtin_amdgcn_ballot_w64(tmax > guard.x_hi) != 0) wbits |= 2u;                         // a sample above the guarded range
            if (__builtin_amdgcn_ballot_w64(tmax >= guard.x_lo) != 0) wbits |= 4u;                        // a sample of ordinary size
            if (__builtin_amdgcn_ballot_w64(tmax > 0.f) != 0) wbits |= 8u;                                // a non-zero sample
            if (lane == 0) cx_flags[wv] = wbits;
            __syncthreads();                                           // (also: the zero fill is complete)
            const unsigned all = cx_flags[0] | cx_flags[1] | cx_flags[2] | cx_flags[3];
            window_bad = (int)(all & 1u);
            // (a non-finite window is NaN on both routes: no fallback for it)
            if (!(all & 1u) && ((all & 2u) || ((all & 8u) && !(all & 4u))) && tid == 0) {
                __hip_atomic_store(guard.word, guard.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicAdd(guard.word + 1, 1u);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            unsigned p[3];
            cx_split2(v[q].x, v[q].y, p);
            if (tid + 256 * q < WIN * CH / 2) {
                char* d = cx_lds + cx_addr<128>(t + 1, 2 * c2);
#pragma unroll
                for (int k = 0; k < NT; ++k) *reinterpret_cast<unsigned*>(d + k * CX_PLANE) = p[k];
            }
            t += 9; c2 += 13;                                          // 256 = 9 x 27 + 13
            if (c2 >= 27) { c2 -= 27; t += 1; }
        }
    }
    __syncthreads();
    TRACE_MARK(1);

    cx_f32x4 acc[2][CX_NT];
    auto bias_acc = [&](const float* __restrict__ bias, int co0) {     // the accumulators start from the bias of their four channels
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const float4 bv = *reinterpret_cast<const float4*>(bias + co0 + 16 * rt + 4 * g);
#pragma unroll
            for (int ct = 0; ct < CX_NT; ++ct) acc[rt][ct] = cx_f32x4{bv.x, bv.y, bv.z, bv.w};
        }
    };

    // ---- stage 1 (T = 150, 64 channels in and out): wave = row-tile pair wv & 1, column tiles 5 (wv >> 1) ..
    {
        const int P = wv & 1, ct0 = 5 * (wv >> 1), base = 16 * ct0 + j;
        const int sw[3] = {cx_swz<128>(base), cx_swz<128>(base + 1), cx_swz<128>(base + 2)};
        const char* xrow = cx_lds + base * 128;
        bias_acc(pk.b[0], 32 * P);
        cx_layer<128, 2, false, CX_ILV != 0, NT>(xrow, sw, g, w0 + (size_t)P * (6 * 2 * 3 * 64), acc);
        TRACE_MARK(2);
        __syncthreads();                                               // every wave has read conv1's input
        cx_store<128, false, WIN, TAPS, NT>(cx_lds, acc, 32 * P, ct0, j, g, TAPS ? taps.conv1 + win * 64 * 150 : nullptr);
        __syncthreads();
        TRACE_MARK(3);
        bias_acc(pk.b[1], 32 * P);
        cx_layer<128, 2, false, CX_ILV != 0, NT>(xrow, sw, g, w1 + (size_t)P * (6 * 2 * 3 * 64), acc);
        TRACE_MARK(4);
        __syncthreads();
        cx_store<128, true, WIN, TAPS, NT>(cx_lds, acc, 32 * P, ct0, j, g, TAPS ? taps.conv2 + win * 64 * 150 : nullptr, TAPS ? taps.pool1 + win * 64 * 75 : nullptr);     // pooled: rows 1..75 of the stage-2 layout (64 channels)
        if (tid < 8 * NT) reinterpret_cast<uint4*>(cx_lds + (tid >> 3) * CX_PLANE + 76 * 128)[tid & 7] = make_uint4(0, 0, 0, 0);   // row 76 = right pad
        __syncthreads();
        TRACE_MARK(5);
    }
    // ---- stage 2 (T = 75): wave = row-tile pair wv, all five column tiles
    {
        const int sw[3] = {cx_swz<128>(j), cx_swz<128>(j + 1), cx_swz<128>(j + 2)};
        bias_acc(pk.b[2], 32 * wv);
        cx_layer<128, 2, false, CX_ILV != 0, NT>(cx_lds + j * 128, sw, g, w2 + (size_t)wv * (6 * 2 * 3 * 64), acc);
        TRACE_MARK(6);
        if constexpr (PERSIST) {
            if (win + gridDim.x < n) request(win + gridDim.x, tid);
        }
        __syncthreads();
        cx_store<256, false, 75, TAPS, NT>(cx_lds, acc, 32 * wv, 0, j, g, TAPS ? taps.conv3 + win * 128 * 75 : nullptr);      // 128 channels: 256-byte rows, rows 1..75
        if (tid < 32 * NT) {                                                   // rows 0 and 76 of the new layout = the zero padding
            const int p = tid >> 5, r = (tid >> 4) & 1, s = tid & 15;
            reinterpret_cast<uint4*>(cx_lds + p * CX_PLANE + (r ? 76 : 0) * 256)[s] = make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
        TRACE_MARK(7);
        const int sw4[3] = {cx_swz<256>(j), cx_swz<256>(j + 1), cx_swz<256>(j + 2)};
        bias_acc(pk.b[3], 32 * wv);
        cx_layer<256, 4, false, CX_ILV != 0, NT>(cx_lds + j * 256, sw4, g, w3 + (size_t)wv * (12 * 2 * 3 * 64), acc);
        TRACE_MARK(8);
        if constexpr (PERMK) {
            // ---- conv4 + ReLU + MaxPool (t = 74 dropped) straight from the accumulators to HBM in the K order k' = t' * 128 + c: a lane
            //      holds four consecutive channels of one pooled position = 8 bytes per plane, so the features need neither the trip
            //      through LDS nor its two barriers (the reference's flatten order c * 37 + t' would be 2-byte stores: 9.6k cycles per
            //      workgroup).  fc.0's weights for this path have their K axis permuted the same way (dce_finalize_weights,
            //      fc_perm_k_host): no product changes, only the order of a summation that claims no bit pattern.
            static_assert(OUT != 1 && !TAPS, "the fp32 features of the mid-size batches and the taps keep the reference's order");
            unsigned short* const out = OUT == 2 ? feat3 + (size_t)win * FEAT : feat3 + (size_t)(win >> 1) * (2 * FEAT) + (int)(win & 1) * 32;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int co = 32 * wv + 16 * rt + 4 * g;
#pragma unroll
                for (int ct = 0; ct < CX_NT; ++ct) {
                    const int t = 16 * ct + j;
                    float v[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaxf(v[r], cx_neighbour(v[r])), 0.f);
                    unsigned lo[3], hi[3];
                    cx_split2(v[0], v[1], lo);
                    cx_split2(v[2], v[3], hi);
                    if (window_bad) {                                  // a non-finite sample: NaN in every term of the window's features
#pragma unroll
                        for (int p = 0; p < 3; ++p) lo[p] = hi[p] = 0x7fc07fc0u;
                    }
                    if ((j & 1) == 0 && (t >> 1) < 37) {
                        const int k = (t >> 1) * 128 + co;
                        if constexpr (OUT == 2) *reinterpret_cast<uint2*>(out + k) = make_uint2(lo[0], hi[0]);
                        else {
#pragma unroll
                            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(out + p * plane_elems + (k >> 5) * 64 + (k & 31)) = make_uint2(lo[p], hi[p]);
                        }
                    }
                }
            }
            TRACE_MARK(9);
            if constexpr (PERSIST) {
                win += gridDim.x;
                if (win >= n) return;
                __syncthreads();                                       // every wave is through conv4's input before the next window overwrites it
                continue;
            } else return;
        }
        // ---- conv4 + bias + ReLU + MaxPool (t = 74 dropped) + flatten k = c * 37 + t' -> three planes [k] in LDS (the layer's
        //      input is dead once every wave is through its MFMAs), then 16-byte stores into fc_gemm_x3.hip's layout: 7 per
        //      thread, where storing from the accumulators' layout took 120 two-byte stores per lane (9.6k cycles per workgroup)
        __syncthreads();
        const float nanv = __builtin_nanf("");
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int co = 32 * wv + 16 * rt + 4 * g;
#pragma unroll
            for (int ct = 0; ct < CX_NT; ++ct) {
                const int t = 16 * ct + j;
                float v[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};
                if constexpr (TAPS) {
                    if (t < 75)
#pragma unroll
                        for (int r = 0; r < 4; ++r) taps.conv4[((size_t)win * 128 + co + r) * 75 + t] = fmaxf(v[r], 0.f);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaxf(v[r], cx_neighbour(v[r])), 0.f);
                if constexpr (TAPS) {
                    if ((j & 1) == 0 && (t >> 1) < 37)
#pragma unroll
                        for (int r = 0; r < 4; ++r) feat32[(size_t)win * FEAT + (co + r) * 37 + (t >> 1)] = window_bad ? nanv : v[r];
                }
                if constexpr (OUT == 1) {
                    if ((j & 1) == 0 && (t >> 1) < 37) {
                        float* d = reinterpret_cast<float*>(cx_lds) + co * 37 + (t >> 1);
                        d[0] = v[0]; d[37] = v[1]; d[74] = v[2]; d[111] = v[3];
                    }
                } else if constexpr (OUT == 2) {
                    unsigned lo[3], hi[3];                             // term 1 = the value rounded to bf16 (nearest-even)
                    cx_split2(v[0], v[1], lo);
                    cx_split2(v[2], v[3], hi);
                    if ((j & 1) == 0 && (t >> 1) < 37) {
                        unsigned short* d = reinterpret_cast<unsigned short*>(cx_lds) + co * 37 + (t >> 1);
                        d[0] = (unsigned short)lo[0];  d[37] = (unsigned short)(lo[0] >> 16);
                        d[74] = (unsigned short)hi[0]; d[111] = (unsigned short)(hi[0] >> 16);
                    }
                } else {
                    unsigned lo[3], hi[3];
                    cx_split2(v[0], v[1], lo);
                    cx_split2(v[2], v[3], hi);
                    if ((j & 1) == 0 && (t >> 1) < 37) {
                        unsigned short* d = reinterpret_cast<unsigned short*>(cx_lds) + co * 37 + (t >> 1);
#pragma unroll
                        for (int p = 0; p < 3; ++p) {
                            d[p * FEAT] = (unsigned short)lo[p];       d[p * FEAT + 37] = (unsigned short)(lo[p] >> 16);
                            d[p * FEAT + 74] = (unsigned short)hi[p];  d[p * FEAT + 111] = (unsigned short)(hi[p] >> 16);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if constexpr (OUT == 2) {
            for (int q = tid; q < FEAT / 8; q += 256) {
                uint4 val = reinterpret_cast<const uint4*>(cx_lds)[q];
                if (window_bad) val = make_uint4(0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u);
                reinterpret_cast<uint4*>(feat3 + (size_t)win * FEAT)[q] = val;
            }
            TRACE_MARK(9);
            return;
        }
        if constexpr (OUT == 1) {
            const float nn = __builtin_nanf("");
            for (int q = tid; q < FEAT / 4; q += 256) {
                float4 val = reinterpret_cast<const float4*>(cx_lds)[q];
                if (window_bad) val = make_float4(nn, nn, nn, nn);
                reinterpret_cast<float4*>(feat32 + (size_t)win * FEAT)[q] = val;
            }
            TRACE_MARK(9);
            return;
        }
        // row `win` of a pair-interleaved plane: runs of 32 k (64 bytes) at stride 128 bytes
        unsigned short* out = feat3 + (size_t)(win >> 1) * (2 * FEAT) + (int)(win & 1) * 32;
        for (int q = tid; q < 3 * (FEAT / 8); q += 256) {
            const int p = q / (FEAT / 8), k8 = (q % (FEAT / 8)) * 8;
            uint4 val = *reinterpret_cast<const uint4*>(cx_lds + ((size_t)p * FEAT + k8) * 2);
            if (window_bad) val = make_uint4(0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u, 0x7fc07fc0u);      // a non-finite window: NaN in every term
            *reinterpret_cast<uint4*>(out + p * plane_elems + (k8 >> 5) * 64 + (k8 & 31)) = val;
        }
    }
    TRACE_MARK(9);
    return;
    }
}

#if DCE_EXPERIMENTS
// the persistent form's grid: two workgroups per CU (what the LDS admits), every one with the same number of windows +- 1
static unsigned persist_grid(int64_t n)
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        cus = 256;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    return (unsigned)(n < 2 * (int64_t)cus ? n : 2 * (int64_t)cus);
}
#endif

hipError_t init_conv_x3()
{
    // Product library (round 6): the TWO-term bf16 form alone (DCE_BF16_FC at up to 256 windows per launch, taps, online pushes).  The three-term forms
    // -- DCE_FP32_SPLIT's conv stack and the bf16-FC mode's round-3 stack -- live in the experiments build: fp32_f16x2 (conv_h2.hip) holds the same
    // contract at 1.13 - 2.0 x their speed at every size (profiles/r6h_retire_split_sweep.txt).
    hipError_t e = hipSuccess;
    for (const void* k : {reinterpret_cast<const void*>(&conv_x3_kernel<true, false, 2, false, false, 2>), reinterpret_cast<const void*>(&conv_x3_kernel<false, false, 2, false, false, 2>),
                          reinterpret_cast<const void*>(&conv_x3_kernel<true, false, 2, true, false, 2>), reinterpret_cast<const void*>(&conv_x3_kernel<false, false, 2, true, false, 2>),
#if DCE_EXPERIMENTS
                          reinterpret_cast<const void*>(&conv_x3_kernel<true>), reinterpret_cast<const void*>(&conv_x3_kernel<false>), reinterpret_cast<const void*>(&conv_x3_kernel<false, true>),
                          reinterpret_cast<const void*>(&conv_x3_kernel<true, false, 1>), reinterpret_cast<const void*>(&conv_x3_kernel<false, false, 1>),
                          reinterpret_cast<const void*>(&conv_x3_kernel<true, false, 2>), reinterpret_cast<const void*>(&conv_x3_kernel<false, false, 2>),
                          reinterpret_cast<const void*>(&conv_x3_kernel<true, false, 0, true>), reinterpret_cast<const void*>(&conv_x3_kernel<false, false, 0, true>),
                          reinterpret_cast<const void*>(&conv_x3_kernel<true, false, 2, true>), reinterpret_cast<const void*>(&conv_x3_kernel<false, false, 2, true>),
                          reinterpret_cast<const void*>(&conv_x3_kernel<true, false, 0, true, true>), reinterpret_cast<const void*>(&conv_x3_kernel<false, false, 0, true, true>),
                          reinterpret_cast<const void*>(&conv_x3_kernel<true, false, 2, true, true>), reinterpret_cast<const void*>(&conv_x3_kernel<false, false, 2, true, true>),
#endif
                          })
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, CX_LDS)) != hipSuccess) return e;
    return e;
}

// dce_conv_layer_taps, kernel 7: pre-normalised windows through conv_x3_kernel with every layer's output (and the fp32
// features) written out
hipError_t launch_conv_x3_taps(const float* windows, int64_t n, const ConvPackX3& pk, unsigned short* feat3, float* feat32,
                               const LayerTaps& taps, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
#if DCE_EXPERIMENTS
    const size_t plane_elems = (size_t)((n + 1) & ~(int64_t)1) * FEAT;
    hipLaunchKernelGGL((conv_x3_kernel<false, true>), dim3((unsigned)n), dim3(256), CX_LDS, st, windows, n, pk, feat3, plane_elems, taps, feat32);
    return hipGetLastError();
#else
    (void)windows; (void)pk; (void)feat3; (void)feat32; (void)taps; (void)st;
    return hipErrorInvalidValue;                       // (the three-term stack: experiments build)
#endif
}

// the same stack with (n, 4736) fp32 features out: DCE_FP32_SPLIT at batches below the split-bf16 fc.0 kernel's threshold
hipError_t launch_conv_x3_f32(const float* src, int zscore, int64_t n, const ConvPackX3& pk, float* feat, hipStream_t st, const GuardArgs& guard)
{
    if (n <= 0) return hipSuccess;
#if DCE_EXPERIMENTS
    plan_note("conv_x3_f32");
    if (zscore) hipLaunchKernelGGL((conv_x3_kernel<true, false, 1>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, nullptr, (size_t)0, LayerTaps{}, feat);
    else        hipLaunchKernelGGL((conv_x3_kernel<false, false, 1>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, nullptr, (size_t)0, LayerTaps{}, feat, nullptr, guard);
    return hipGetLastError();
#else
    (void)src; (void)zscore; (void)pk; (void)feat; (void)st; (void)guard;
    return hipErrorInvalidValue;
#endif
}

// ... with (n, 4736) bf16 features out: the DCE_BF16_FC precision
hipError_t launch_conv_x3_bf16(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat, hipStream_t st, int permk, int terms,
                               const long long* src_row)
{
    if (n <= 0) return hipSuccess;
    if (terms == 2) {                                  // two-term operands (three MFMAs per product), three workgroups per CU
        constexpr int L2T = 2 * CX_PLANE;
        plan_note(permk ? "conv_x2_bf16_permk" : "conv_x2_bf16");
        if (permk) {
            if (zscore) hipLaunchKernelGGL((conv_x3_kernel<true, false, 2, true, false, 2>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
            else        hipLaunchKernelGGL((conv_x3_kernel<false, false, 2, true, false, 2>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
        } else {
            if (zscore) hipLaunchKernelGGL((conv_x3_kernel<true, false, 2, false, false, 2>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
            else        hipLaunchKernelGGL((conv_x3_kernel<false, false, 2, false, false, 2>), dim3((unsigned)n), dim3(256), L2T, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
        }
        return hipGetLastError();
    }
#if !DCE_EXPERIMENTS
    return hipErrorInvalidValue;                       // (three-term bf16 stack: experiments build)
#else
    if (permk == 2) {                                  // ... from persistent workgroups (measured 2-4 % slower: profiles/r4h_ab_conv_x3_persist.txt)
        plan_note("conv_x3_bf16_permk_persist");
        if (zscore) hipLaunchKernelGGL((conv_x3_kernel<true, false, 2, true, true>), dim3(persist_grid(n)), dim3(256), CX_LDS, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
        else        hipLaunchKernelGGL((conv_x3_kernel<false, false, 2, true, true>), dim3(persist_grid(n)), dim3(256), CX_LDS, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
        return hipGetLastError();
    }
    if (permk) {                                       // features in the K order t' * 128 + c, straight from the accumulators
        plan_note("conv_x3_bf16_permk");
        if (zscore) hipLaunchKernelGGL((conv_x3_kernel<true, false, 2, true>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
        else        hipLaunchKernelGGL((conv_x3_kernel<false, false, 2, true>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
        return hipGetLastError();
    }
    plan_note("conv_x3_bf16");
    if (zscore) hipLaunchKernelGGL((conv_x3_kernel<true, false, 2>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
    else        hipLaunchKernelGGL((conv_x3_kernel<false, false, 2>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, feat, (size_t)0, LayerTaps{}, nullptr, src_row);
    return hipGetLastError();
#endif
}

hipError_t launch_conv_x3(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat3, hipStream_t st, int permk, const GuardArgs& guard)
{
#if !DCE_EXPERIMENTS
    (void)src; (void)zscore; (void)n; (void)pk; (void)feat3; (void)st; (void)permk; (void)guard;
    return hipErrorInvalidValue;                       // (DCE_FP32_SPLIT's conv stack: experiments build)
#else
    if (n <= 0) return hipSuccess;
    const size_t plane_elems = (size_t)((n + 1) & ~(int64_t)1) * FEAT;
#if DCE_EXPERIMENTS
    if (permk == 2) {
        plan_note("conv_x3_permk_persist");
        if (zscore) hipLaunchKernelGGL((conv_x3_kernel<true, false, 0, true, true>), dim3(persist_grid(n)), dim3(256), CX_LDS, st, src, n, pk, feat3, plane_elems, LayerTaps{}, nullptr);
        else        hipLaunchKernelGGL((conv_x3_kernel<false, false, 0, true, true>), dim3(persist_grid(n)), dim3(256), CX_LDS, st, src, n, pk, feat3, plane_elems, LayerTaps{}, nullptr);
        return hipGetLastError();
    }
#endif
    if (permk) {
        plan_note("conv_x3_permk");
        if (zscore) hipLaunchKernelGGL((conv_x3_kernel<true, false, 0, true>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, feat3, plane_elems, LayerTaps{}, nullptr);
        else        hipLaunchKernelGGL((conv_x3_kernel<false, false, 0, true>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, feat3, plane_elems, LayerTaps{}, nullptr, nullptr, guard);
        return hipGetLastError();
    }
    plan_note("conv_x3");
    if (zscore) hipLaunchKernelGGL((conv_x3_kernel<true>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, feat3, plane_elems, LayerTaps{}, nullptr);
    else        hipLaunchKernelGGL((conv_x3_kernel<false>), dim3((unsigned)n), dim3(256), CX_LDS, st, src, n, pk, feat3, plane_elems, LayerTaps{}, nullptr, nullptr, guard);
    return hipGetLastError();
#endif
}

}  // namespace dce

#if DCE_TRACE
extern "C" int dce_debug_trace_read_x3(unsigned long long* out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_trace), sizeof(unsigned long long) * 16 * nblocks);
}
#endif
