// conv_common.h -- device helpers shared by the two conv-stack kernels (direct and Winograd).
#pragma once
#include "dce_kernels.h"

namespace dce {

#ifndef DCE_TRACE
#define DCE_TRACE 0
#endif
#if DCE_TRACE
// debug build only (build.build_variant('trace', ['-DDCE_TRACE=1'])): per-workgroup phase
// timestamps (s_memtime) + HW_ID, read back with dce_debug_trace_read() (tools/trace_conv.py)
static __device__ unsigned long long g_trace[4096 * 16];   // one per translation unit
#define TRACE_MARK(k) do { if (tid == 0 && blockIdx.x < 4096) \
        g_trace[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define TRACE_MARK(k) do {} while (0)
#endif

#ifndef DCE_ZS_RECIPROCAL
#define DCE_ZS_RECIPROCAL 1
#endif

constexpr int NW = 2;             // windows per workgroup (two workgroups per CU)

__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }   // keeps NaN like torch

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f)
{   // round-to-nearest-even; NaN stays NaN
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ void put_feat(float* p, float v) { *p = v; }
__device__ __forceinline__ void put_feat(unsigned short* p, float v) { *p = f32_to_bf16_rne(v); }

// DCE_FP32_SPLIT: the features as THREE bf16 planes, v = t1 + t2 + t3 exactly, in the pair-interleaved layout the
// split-bf16 fc.0 kernel reads (fc_gemm_x3.hip: element (r, k) of a plane at (r >> 1) * 2K + (k >> 5) * 64 + (r & 1) * 32
// + (k & 31); plane stride = rows rounded up to even, times K).
struct Feat3 { unsigned short u; };
typedef __bf16 cc_bf16x2 __attribute__((ext_vector_type(2)));
typedef float cc_f32x2 __attribute__((ext_vector_type(2)));
// two values at once: v_cvt_pk_bf16_f32 (round-to-nearest-even, as split3_kernel and the host routine round) turns a pair
// into its three terms in 9 VALU instructions; the six 16-bit stores go to (row, k0) and (row, k1)
__device__ __forceinline__ void put_feat3(Feat3* planes, size_t plane_elems, int64_t row, int k0, int k1, float v0, float v1)
{
    const cc_f32x2 v = {v0, v1};
    const cc_bf16x2 t1 = __builtin_convertvector(v, cc_bf16x2);
    const cc_f32x2 r1 = v - __builtin_convertvector(t1, cc_f32x2);          // exact
    const cc_bf16x2 t2 = __builtin_convertvector(r1, cc_bf16x2);
    const cc_f32x2 r2 = r1 - __builtin_convertvector(t2, cc_f32x2);         // exact
    const cc_bf16x2 t3 = __builtin_convertvector(r2, cc_bf16x2);
    const unsigned p1 = __builtin_bit_cast(unsigned, t1), p2 = __builtin_bit_cast(unsigned, t2), p3 = __builtin_bit_cast(unsigned, t3);
    unsigned short* base = reinterpret_cast<unsigned short*>(planes) + (size_t)(row >> 1) * (2 * FEAT) + (int)(row & 1) * 32;
    unsigned short* d0 = base + (k0 >> 5) * 64 + (k0 & 31);
    unsigned short* d1 = base + (k1 >> 5) * 64 + (k1 & 31);
    d0[0] = (unsigned short)p1; d0[plane_elems] = (unsigned short)p2; d0[2 * plane_elems] = (unsigned short)p3;
    d1[0] = (unsigned short)(p1 >> 16); d1[plane_elems] = (unsigned short)(p2 >> 16); d1[2 * plane_elems] = (unsigned short)(p3 >> 16);
}

// Load NWIN windows (rows t = 4m + g of channel c per thread, tid = g*54 + c < 216) and, if ZS,
// z-score them per channel over time exactly as utils/data_handler.py:55-56 does on fp32 data:
//   (x - mean) / std,  mean = sum/150 rounded to fp32,  std = sqrt(sum((x-mean)^2)/149) (unbiased,
//   no epsilon) rounded to fp32.
// Both reductions run in fp64 (full-rate on CDNA4, ~300 ops per thread): with sensor offsets
// of O(1-10) and spreads of O(0.01) one fp32 ulp of the mean is already 5e-5 standard
// deviations, so the mean must be the correctly rounded one, not an fp32 running sum.
// red: >= NWIN*4*216 floats of LDS scratch (two sets of 216 doubles per window).
// STAGE: 0 both halves; 1 only the loads (a persistent workgroup issues them a layer ahead); 2 only the arithmetic on x.
template <bool ZS, int NWIN, int STAGE = 0>
__device__ __forceinline__ void load_windows(const float* __restrict__ src, int64_t win_stride,
                                             int nvalid, float* __restrict__ red,
                                             float (&x)[NWIN][38], int tid)
{
    const int c = tid % CH, g = tid / CH;
    const bool loader = tid < 4 * CH;
    // Every load is issued unconditionally from an always-valid address: under a per-element
    // runtime condition hipcc wraps each load in its own saveexec/branch block (~7 instructions per
    // element, all of which queue behind the partner workgroup's MFMAs).  Threads >= 216 re-read
    // row group 0 and a missing window (odd n) re-reads window 0; neither is ever stored.  Only
    // the last row group (t = 148..151) has rows past the window: those two read 0.
    const int gl = loader ? g : 0;
    if constexpr (STAGE != 2) {
#pragma unroll
        for (int w = 0; w < NWIN; ++w) {
            const float* wsrc = src + (w < nvalid ? w : 0) * win_stride + gl * CH + c;
#pragma unroll
            for (int m = 0; m < 37; ++m) x[w][m] = wsrc[4 * m * CH];
            x[w][37] = gl < 2 ? wsrc[148 * CH] : 0.f;             // rows 148 + gl
        }
    }
    if (ZS && STAGE != 1) {
        double* dred = reinterpret_cast<double*>(red);
#pragma unroll
        for (int w = 0; w < NWIN; ++w) {
            double s = 0.0;
#pragma unroll
            for (int m = 0; m < 38; ++m) s += (double)x[w][m];       // rows t >= 150 were loaded as 0
            if (loader) dred[w * 216 + tid] = s;
        }
        __syncthreads();
        float mean[NWIN];
#pragma unroll
        for (int w = 0; w < NWIN; ++w) {
            const double* r = dred + w * 216 + c;
            const double mu = loader ? ((r[0] + r[54]) + (r[108] + r[162])) / 150.0 : 0.0;
            mean[w] = (float)mu;
            double q = 0.0;
#pragma unroll
            for (int m = 0; m < 38; ++m) {
                const double d = (double)x[w][m] - mu;
                q += (4 * m + g < WIN) ? d * d : 0.0;
            }
            if (loader) dred[(NWIN + w) * 216 + tid] = q;
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < NWIN; ++w) {
            const double* r = dred + (NWIN + w) * 216 + c;
            const float sd = loader ? (float)sqrt(((r[0] + r[54]) + (r[108] + r[162])) / 149.0) : 1.f;
#if DCE_ZS_RECIPROCAL
            // one correctly rounded reciprocal per channel, then a multiply per sample: within
            // 1 ulp of the reference's (x - mean) / std and ~10x fewer VALU ops than 38 divisions
            const float inv = 1.f / sd;
#pragma unroll
            for (int m = 0; m < 38; ++m) x[w][m] = (w < nvalid) ? (x[w][m] - mean[w]) * inv : 0.f;
#else
#pragma unroll
            for (int m = 0; m < 38; ++m) x[w][m] = (w < nvalid) ? (x[w][m] - mean[w]) / sd : 0.f;
#endif
        }
    }
}

}  // namespace dce
