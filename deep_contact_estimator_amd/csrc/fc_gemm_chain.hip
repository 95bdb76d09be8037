// fc_gemm_chain.hip -- fc.0 / fc.3 (Linear + bias + ReLU, reference src/contact_cnn.py:49-55) for batches that are
// too small to fill the chip with GEMM tiles: 9 .. a few hundred windows (config/test_params.yaml's batch_size 30).
//
// What bounds a layer at these sizes is not the matrix pipe's throughput but the LENGTH of one output's fma chain: every
// kernel of this library walks K as one ordered chain (that is what makes a window's logits the same bits at every batch
// size), fc.0's chain has 4736 links, and a dependent chain advances at
//     v_fmac_f32                 9.4 cycles per link        (tools/micro/fma_chain.hip;  the GEMV kernel, fc_gemv.hip)
//     v_mfma_f32_32x32x2_f32    34   cycles per link        (64 cycles, 2 links;         the tile GEMMs, fc_gemm.hip)
//     v_mfma_f32_16x16x4_f32     8.7 cycles per link        (35 cycles, 4 links;  tools/micro/mfma_chain.hip: a dependent
//                                                            chain issues back to back, one accumulator per wave suffices)
// So: one 16x16 output tile per wave, ONE wave per SIMD (a second wave on the SIMD would halve each chain's speed), a
// workgroup of four waves = a 32 x 32 block of C, operands streamed through registers (a ring of K chunks in flight, as in
// the GEMV kernel) into a double-buffered LDS image.  fc.0 for up to 128 windows is 1024 chains on 1024 SIMDs: 17 us of
// chain, 30.5 us per launch measured (profiles/r2z_latency.txt), where the GEMV kernel takes 39 us at 30 windows (it is
// LDS-bound: every fma reads both operands from LDS; an MFMA reads 512 B for 1024 of them) and the 64x64 tile GEMM 93 us at
// 64 .. 256 windows.  What is left above the chain is the staging work that issues in order with it (timing probes: no LDS
// stores 23.1 us, no ring refills 24.7 us, neither 22.3 us); staging in four waves of their own (8 waves per workgroup,
// 128 KB LDS) measured slower (33.8 us), ds_write2_b32 stores without the register shuffles too (4-way conflicts, 32.6 us).
//
// K order = that of every other FC kernel of the library (0,4,1,5,2,6,3,7 inside each 8 consecutive k, chunks ascending,
// start from 0, bias added last): v_mfma_f32_16x16x4_f32 accumulates its four k in lane-group order, so the first MFMA of a
// group of 8 k takes k = {0,4,1,5} from lane groups 0..3 and the second {2,6,3,7}.  To let a lane fetch its two values
// (k and k+2) with one aligned 8-byte LDS read, the staging threads store each group of 8 k as [k0 k2 k4 k6 | k1 k3 k5 k7]
// (pure register naming between the 16-byte global loads and the 16-byte LDS stores, no VALU work); 16-byte slots of a
// 512-byte LDS row are XOR-swizzled with the row number, so that the ds_read_b64 of 16 rows x one slot is conflict-free.
// Results are bit-identical to the GEMV / tile / phased kernels (tests/test_gpu_parity.py).
#include "dce_kernels.h"
#include <cstdlib>

namespace dce {

typedef float ch_f32x4 __attribute__((ext_vector_type(4)));
typedef float ch_f32x2 __attribute__((ext_vector_type(2)));

constexpr int CH_K = 128;                       // floats of K per chunk (a 512-byte LDS row = 32 slots of 16 B)
constexpr int CH_ROWS = 64;                     // LDS rows per buffer: 32 rows of A (windows) + 32 rows of W (neurons)
constexpr int CH_BUF_BYTES = CH_ROWS * CH_K * 4;            // 32 KB
constexpr int CH_LDS_BYTES = 2 * CH_BUF_BYTES;              // double buffer: 64 KB
constexpr int CH_DEPTH = 4;                     // K chunks in flight per thread (8 float4 each)

struct ChSlot { ch_f32x4 x[4], y[4]; };        // one chunk's share of a thread: 4 groups of 8 k (x = k0..3, y = k4..7)

// Staging loads are asm (wave-uniform 64-bit base in SGPRs + a 32-bit lane offset + an immediate) with counted
// s_waitcnt vmcnt, as in fc_gemm_small_kernel: written as C++ loads, hipcc sinks the ring's loads to their first use
// (measured on the first version of this kernel: chunks 1..3 were requested only after chunk 0 had arrived and been
// staged), and a ring that is not in flight is no ring.  tests/test_abi.py guards the pattern in the disassembly.
#define CH_LD(dst, voff, sbase, imm) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(voff), "s"(sbase))
#define CH_FETCH(S, sbase)                                                                           \
    do { CH_LD(S.x[0], voff, sbase, 0);   CH_LD(S.y[0], voff, sbase, 16);                            \
         CH_LD(S.x[1], voff, sbase, 128); CH_LD(S.y[1], voff, sbase, 144);                           \
         CH_LD(S.x[2], voff, sbase, 256); CH_LD(S.y[2], voff, sbase, 272);                           \
         CH_LD(S.x[3], voff, sbase, 384); CH_LD(S.y[3], voff, sbase, 400); } while (0)
// the oldest chunk of the ring has landed once at most 8 * (CH_DEPTH - 1) loads are outstanding; the "+v" operands
// tie the wait to the registers the loads write
#define CH_WAIT(N, S)                                                                                \
    asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(S.x[0]), "+v"(S.y[0]), "+v"(S.x[1]), "+v"(S.y[1]), \
                                             "+v"(S.x[2]), "+v"(S.y[2]), "+v"(S.x[3]), "+v"(S.y[3]))

template <int K>                                 // 4736 (fc.0) or 2048 (fc.3): the chunk loop is laid out at compile time
__global__ __launch_bounds__(256)
void fc_gemm_chain_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                          float* __restrict__ C, int M, int N, int relu)
{
    static_assert(CH_DEPTH == 4, "the ring below is written out for four slots (vmcnt(24) = 8 loads x 3 younger chunks)");
    extern __shared__ __attribute__((aligned(16))) unsigned char ch_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;

    // ---- staging role: thread (row = tid/4, q = tid%4) moves the 32-byte groups q, q+4, q+8, q+12 of its row;
    //      waves 0,1 stage the 32 rows of A, waves 2,3 the 32 rows of W (a wave-uniform base pointer)
    const int srow = tid >> 2, q = tid & 3;
    const int arow = m0 + srow < M ? m0 + srow : M - 1;
    const unsigned voff = (unsigned)((wv < 2 ? arow : n0 + srow - 32) * (size_t)K * 4 + 32 * q);
    const char* sb = reinterpret_cast<const char*>(wv < 2 ? A : W);
    constexpr int nch = K / CH_K;
    static_assert(K % CH_K == 0 && nch > CH_DEPTH, "whole chunks, more of them than ring slots");
    // LDS byte offsets of this thread's 8 slots inside a buffer (slot index XOR row)
    int st_off[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h) st_off[j][h] = srow * (CH_K * 4) + (((2 * (q + 4 * j) + h) ^ (srow & 15)) << 4);
    // each group of 8 k is stored as [k0 k2 k4 k6 | k1 k3 k5 k7]
    auto stage_half = [&](const ChSlot& s, int buf, int j, int h) {
        unsigned char* b = ch_lds + buf * CH_BUF_BYTES;
        *reinterpret_cast<float4*>(b + st_off[j][h]) = h == 0 ? make_float4(s.x[j].x, s.x[j].z, s.y[j].x, s.y[j].z)
                                                               : make_float4(s.x[j].y, s.x[j].w, s.y[j].y, s.y[j].w);
    };
    auto stage = [&](const ChSlot& s, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { stage_half(s, buf, j, 0); stage_half(s, buf, j, 1); }
    };

    // ---- compute role: wave (wm, wn) owns C[m0 + 16 wm .. +16][n0 + 16 wn .. +16]; lane (i, g):
    //      A[i][k] and W[i][k] operands of lane group g sit in slot 2s + (g>>1), half g&1 of row i
    const int wm = wv >> 1, wn = wv & 1, i = lane & 15, g = lane >> 4;
    const int t = i ^ (g >> 1);
    int a_off[8], b_off[8];                                    // byte offsets for slices s = 0..7 (s + 8: +256 B)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int o = (((2 * s) ^ t) << 4) + (g & 1) * 8;
        a_off[s] = (16 * wm + i) * (CH_K * 4) + o;
        b_off[s] = (32 + 16 * wn + i) * (CH_K * 4) + o;
    }
    ch_f32x2 fa[2][8], fb[2][8];                               // fragments of half a chunk (8 slices), two sets
    auto rd = [&](int set, int buf, int half, int s) {
        const unsigned char* b = ch_lds + buf * CH_BUF_BYTES + half * 256;
        fa[set][s] = *reinterpret_cast<const ch_f32x2*>(b + a_off[s]);
        fb[set][s] = *reinterpret_cast<const ch_f32x2*>(b + b_off[s]);
    };
    auto load_half = [&](int set, int buf, int half) {
#pragma unroll
        for (int s = 0; s < 8; ++s) rd(set, buf, half, s);
    };
    ch_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto mfma2 = [&](int set, int s) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][s].x, fb[set][s].x, acc, 0, 0, 0);    // k = 0,4,1,5
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][s].y, fb[set][s].y, acc, 0, 0, 0);    // k = 2,6,3,7
    };

    // chunk c lives in ring slot c % 4; loads past the last chunk re-read it (never staged)
    auto cbase = [&](int c) { return sb + (size_t)(c < nch ? c : nch - 1) * (CH_K * 4); };
    ChSlot r0, r1, r2, r3;
    { const char* p = cbase(0); CH_FETCH(r0, p); }
    { const char* p = cbase(1); CH_FETCH(r1, p); }
    { const char* p = cbase(2); CH_FETCH(r2, p); }
    { const char* p = cbase(3); CH_FETCH(r3, p); }
    CH_WAIT(24, r0);
    stage(r0, 0);
    { const char* p = cbase(4); CH_FETCH(r0, p); }
    __syncthreads();
    load_half(0, 0, 0);
    // iteration c: chunk c+1 goes to LDS, the second half of chunk c is read while its first half runs on the matrix
    // pipe, and the first half of chunk c+1 is read (behind the barrier that publishes it) while the second half runs.
    // Every iteration is the same straight-line code: past the last chunk the ring re-reads it and the "next chunk"
    // that is staged and half-read is never used.  The MFMA chain is the critical path (an MFMA holds the issue port
    // for 4 of its 35 cycles), so everything else is dealt out between its links, slice by slice, and
    // sched_barrier(0) keeps hipcc from bunching it up again (left alone it sinks every LDS read to just before its
    // MFMA and the chain waits out the LDS latency 16 times per chunk).
#define CH_SB __builtin_amdgcn_sched_barrier(0)
#define CH_A(c, NEXT, s, j, h) mfma2(0, s); rd(1, (c) & 1, 1, s); stage_half(NEXT, ((c) + 1) & 1, j, h);
#define CH_ITER(c, NEXT)                                                                              \
    {                                                                                                 \
        const char* p_ = cbase((c) + 1 + CH_DEPTH);                                                   \
        CH_WAIT(24, NEXT); CH_SB;                                                                     \
        CH_A(c, NEXT, 0, 0, 0) CH_SB;                                                                 \
        CH_A(c, NEXT, 1, 0, 1) CH_LD(NEXT.x[0], voff, p_, 0);   CH_LD(NEXT.y[0], voff, p_, 16);  CH_SB; \
        CH_A(c, NEXT, 2, 1, 0) CH_SB;                                                                 \
        CH_A(c, NEXT, 3, 1, 1) CH_LD(NEXT.x[1], voff, p_, 128); CH_LD(NEXT.y[1], voff, p_, 144); CH_SB; \
        CH_A(c, NEXT, 4, 2, 0) CH_SB;                                                                 \
        CH_A(c, NEXT, 5, 2, 1) CH_LD(NEXT.x[2], voff, p_, 256); CH_LD(NEXT.y[2], voff, p_, 272); CH_SB; \
        CH_A(c, NEXT, 6, 3, 0) CH_SB;                                                                 \
        CH_A(c, NEXT, 7, 3, 1) CH_LD(NEXT.x[3], voff, p_, 384); CH_LD(NEXT.y[3], voff, p_, 400); CH_SB; \
        __syncthreads(); CH_SB;                                                                       \
        _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) { mfma2(1, s_); rd(0, ((c) + 1) & 1, 0, s_); CH_SB; } \
    }
    constexpr int groups = nch / CH_DEPTH, rem = nch % CH_DEPTH;
#pragma unroll 1
    for (int c0 = 0; c0 < groups * CH_DEPTH; c0 += CH_DEPTH) {
        CH_ITER(c0 + 0, r1) CH_ITER(c0 + 1, r2) CH_ITER(c0 + 2, r3) CH_ITER(c0 + 3, r0)
    }
    if constexpr (rem > 0) CH_ITER(groups * CH_DEPTH + 0, r1)
    if constexpr (rem > 1) CH_ITER(groups * CH_DEPTH + 1, r2)
    if constexpr (rem > 2) CH_ITER(groups * CH_DEPTH + 2, r3)
#undef CH_ITER
#undef CH_A
#undef CH_SB
    // The ring's tail loads (chunks past the end) are still in flight and the compiler does not know it: their
    // destination registers must stay allocated until they have landed, or a late load overwrites whatever the
    // epilogue put there (first version: a store address -> memory aperture violation).
#define CH_REGS(S) "+v"(S.x[0]), "+v"(S.y[0]), "+v"(S.x[1]), "+v"(S.y[1]), "+v"(S.x[2]), "+v"(S.y[2]), "+v"(S.x[3]), "+v"(S.y[3])
    asm volatile("s_waitcnt vmcnt(0)" : CH_REGS(r0), CH_REGS(r1));
    asm volatile("" : CH_REGS(r2), CH_REGS(r3));
#undef CH_REGS
    // D[row = 4 g + r][col = i]
    const int n = n0 + 16 * wn + i;
    const float bv = bias[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * wm + 4 * g + r;
        if (m < M) {
            float v = acc[r] + bv;
            if (relu) v = v < 0.f ? 0.f : v;                   // NaN stays NaN, as in fc_gemm.hip
            C[(size_t)m * N + n] = v;
        }
    }
}

hipError_t init_fc_gemm_chain()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_chain_kernel<FEAT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_chain_kernel<FC1>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES);
}

// window counts served by this kernel: fc.0 (K = 4736) [DCE_CHAIN_MIN (9), DCE_CHAIN_MAX (640)], fc.3 (K = 2048)
// [DCE_CHAIN_MIN, DCE_CHAIN_MAX3 (2048)]; 0 as a maximum switches it off for that layer.  Measured (r2z): two
// workgroups fit a CU, a "round" of 512 workgroups takes ~45 us for fc.0 and ~12 us for fc.3, so fc.0 (64 workgroups per
// 32 windows) beats the 137 us of one round of 128x64 phased tiles up to 2.5 rounds, and fc.3 (16 per 32 windows) beats
// the fused 64x... phased kernel (71 us incl. combine) up to 2048 windows (47 + 7.5 us with the tail kernel).
bool fc_gemm_chain_ok(int64_t M, int N, int K)
{
    static const int64_t lo = getenv("DCE_CHAIN_MIN") ? atoll(getenv("DCE_CHAIN_MIN")) : 9;
    static const int64_t hi = getenv("DCE_CHAIN_MAX") ? atoll(getenv("DCE_CHAIN_MAX")) : 640;
    static const int64_t hi3 = getenv("DCE_CHAIN_MAX3") ? atoll(getenv("DCE_CHAIN_MAX3")) : 2048;
    return M >= lo && M <= (K == FEAT ? hi : hi3) && N % 32 == 0 && (K == FEAT || K == FC1);
}

hipError_t launch_fc_gemm_chain(const float* A, const float* W, const float* bias, float* C,
                                int64_t M, int N, int K, int relu, hipStream_t st)
{
    if (M <= 0) return hipSuccess;
    if (N % 32 || (K != FEAT && K != FC1) || M > 65535 * 32) return hipErrorInvalidValue;
    const dim3 grid(N / 32, (unsigned)((M + 31) / 32)), block(256);
    if (K == FEAT) hipLaunchKernelGGL(fc_gemm_chain_kernel<FEAT>, grid, block, CH_LDS_BYTES, st, A, W, bias, C, (int)M, N, relu);
    else           hipLaunchKernelGGL(fc_gemm_chain_kernel<FC1>, grid, block, CH_LDS_BYTES, st, A, W, bias, C, (int)M, N, relu);
    return hipGetLastError();
}

}  // namespace dce
