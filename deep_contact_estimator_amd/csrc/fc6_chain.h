// fc6_chain.h -- fc.6 (512 -> 16 logits, reference src/contact_cnn.py:56-57) as a FIXED summation tree that
// every kernel of the library evaluates with the same instructions, so that the logits of a window are the
// same bits whichever kernel sequence its batch size selects (fused into the fc.3 GEMM for chip-filling
// batches, a stand-alone tail kernel behind the GEMV / tile GEMMs, the online mode):
//
//   logit[j] = ((((((((p0 + p1) + p2) + p3) + p4) + p5) + p6) + p7) + b[j]),
//   p_c      = chain over the 64 inputs k of chunk c (columns 64c .. 64c+63 of ReLU(fc.3)),  started at 0,
//              walked by 16 v_mfma_f32_16x16x4_f32 in the order  u = 0..3, e = 0..3 :  k = 16u + e + {0, 4, 8, 12}
//
// (an fp32 MFMA is an ordered fp32 fma chain over its K index; a 64-term chunk is what one 128x64 GEMM tile of
// fc.3 holds, which is what lets the GEMM's epilogue finish the chunk without a second pass over h2).
// One wave evaluates a 16-row x 16-class tile of one chunk: A[i][k] = h2[row i][k], B[k][j] = W3[j][k].
#pragma once
#include "dce_kernels.h"

namespace dce {

typedef float fc6_f32x4 __attribute__((ext_vector_type(4)));
constexpr int FC6_CHUNK = 64, FC6_NCHUNK = FC2 / FC6_CHUNK;       // 8 chunks of 64 inputs

// this lane's W3 operands of chunk c: lane (j = lane & 15, g = lane >> 4) holds W3[j][64c + 16u + 4g + e] in bw[u][e]
__device__ __forceinline__ void fc6_load_w3(const float* __restrict__ W3, int chunk, int lane, float4 (&bw)[4])
{
    const float* p = W3 + (lane & 15) * FC2 + chunk * FC6_CHUNK + 4 * (lane >> 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) bw[u] = *reinterpret_cast<const float4*>(p + 16 * u);
}

// p_c for 16 rows x 16 classes.  rows: LDS, row i of the tile at rows + i*ld, its chunk columns at [col0, col0+64)
// (col0 and ld multiples of 4 floats).  Result: D[row = 4*(lane>>4) + r][class = lane & 15] in element r.
__device__ __forceinline__ fc6_f32x4 fc6_chunk_mfma(const float* __restrict__ rows, int ld, int col0, int lane,
                                                   const float4 (&bw)[4])
{
    const float* ap = rows + (lane & 15) * ld + col0 + 4 * (lane >> 4);
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const float4*>(ap + 16 * u);
    fc6_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, bw[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, bw[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, bw[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, bw[u].w, acc, 0, 0, 0);
    }
    return acc;
}

// the fixed combine of the 8 chunk sums of one (window, class)
__device__ __forceinline__ float fc6_combine(const float (&p)[FC6_NCHUNK], float bias)
{
    float v = p[0];
#pragma unroll
    for (int c = 1; c < FC6_NCHUNK; ++c) v = v + p[c];
    return v + bias;
}

// torch.max(output, 1) over the 16 logits of a window: first maximum; a NaN wins, the first NaN first
__device__ __forceinline__ int fc6_argmax16(const float* __restrict__ l)
{
    int best = 0;
    bool nan_seen = false;
    for (int k = 0; k < NCLS; ++k)
        if (!nan_seen && l[k] != l[k]) { best = k; nan_seen = true; }
    if (!nan_seen)
        for (int k = 1; k < NCLS; ++k)
            if (l[k] > l[best]) best = k;                // strict >: ties -> lowest index
    return best;
}

}  // namespace dce
