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
#include "fc_tree.h"
#include <cstdlib>

namespace dce {

typedef float ch_f32x4 __attribute__((ext_vector_type(4)));
typedef float ch_f32x2 __attribute__((ext_vector_type(2)));

constexpr int CH_K = 128;                       // floats of K per chunk (a 512-byte LDS row = 32 slots of 16 B)
constexpr int CH_LDS_BYTES = 2 * 64 * CH_K * 4; // double buffer of up to 64 rows (32 of A + BN of W): 64 KB
constexpr int CH_DEPTH = 4;                     // K chunks in flight per thread (2 * NGT float4 each)

// Staging loads are asm (wave-uniform 64-bit base in SGPRs + a 32-bit lane offset + an immediate) with counted
// s_waitcnt vmcnt, as in fc_gemm_small_kernel: written as C++ loads, hipcc sinks the ring's loads to their first use
// (measured on the first version of this kernel: chunks 1..3 were requested only after chunk 0 had arrived and been
// staged), and a ring that is not in flight is no ring.  tests/test_abi.py guards the pattern in the disassembly.
#define CH_LD(dst, voff, sbase, imm) \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #imm : "=v"(dst) : "v"(voff), "s"(sbase))

template <int NGT> struct ChSlot { ch_f32x4 x[NGT], y[NGT]; };   // one chunk's share of a thread: NGT groups of 8 k (x = k0..3, y = k4..7)

// "the oldest chunk of the ring has landed": at most 2 NGT (CH_DEPTH - 1) loads outstanding; the "+v" operands tie the
// wait to the registers the loads write (and keep them allocated until then)
__device__ __forceinline__ void ch_wait_oldest(ChSlot<4>& s)
{
    asm volatile("s_waitcnt vmcnt(24)" : "+v"(s.x[0]), "+v"(s.y[0]), "+v"(s.x[1]), "+v"(s.y[1]),
                                         "+v"(s.x[2]), "+v"(s.y[2]), "+v"(s.x[3]), "+v"(s.y[3]));
}
__device__ __forceinline__ void ch_wait_oldest(ChSlot<3>& s)
{
    asm volatile("s_waitcnt vmcnt(18)" : "+v"(s.x[0]), "+v"(s.y[0]), "+v"(s.x[1]), "+v"(s.y[1]), "+v"(s.x[2]), "+v"(s.y[2]));
}
__device__ __forceinline__ void ch_hold(ChSlot<4>& a, ChSlot<4>& b, bool wait)
{
    if (wait) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x[0]), "+v"(a.y[0]), "+v"(a.x[1]), "+v"(a.y[1]), "+v"(a.x[2]), "+v"(a.y[2]), "+v"(a.x[3]), "+v"(a.y[3]),
                                                  "+v"(b.x[0]), "+v"(b.y[0]), "+v"(b.x[1]), "+v"(b.y[1]), "+v"(b.x[2]), "+v"(b.y[2]), "+v"(b.x[3]), "+v"(b.y[3]));
    else      asm volatile("" : "+v"(a.x[0]), "+v"(a.y[0]), "+v"(a.x[1]), "+v"(a.y[1]), "+v"(a.x[2]), "+v"(a.y[2]), "+v"(a.x[3]), "+v"(a.y[3]),
                                "+v"(b.x[0]), "+v"(b.y[0]), "+v"(b.x[1]), "+v"(b.y[1]), "+v"(b.x[2]), "+v"(b.y[2]), "+v"(b.x[3]), "+v"(b.y[3]));
}
__device__ __forceinline__ void ch_hold(ChSlot<3>& a, ChSlot<3>& b, bool wait)
{
    if (wait) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.x[0]), "+v"(a.y[0]), "+v"(a.x[1]), "+v"(a.y[1]), "+v"(a.x[2]), "+v"(a.y[2]),
                                                  "+v"(b.x[0]), "+v"(b.y[0]), "+v"(b.x[1]), "+v"(b.y[1]), "+v"(b.x[2]), "+v"(b.y[2]));
    else      asm volatile("" : "+v"(a.x[0]), "+v"(a.y[0]), "+v"(a.x[1]), "+v"(a.y[1]), "+v"(a.x[2]), "+v"(a.y[2]),
                                "+v"(b.x[0]), "+v"(b.y[0]), "+v"(b.x[1]), "+v"(b.y[1]), "+v"(b.x[2]), "+v"(b.y[2]));
}

// K = 4736 (fc.0) or 2048 (fc.3): the chunk loop is laid out at compile time.  BN = 32: every wave owns a 16x16 tile of
// a 32x32 block; BN = 16 (<= 64 windows, where even the narrow blocks leave CUs idle): a 32x16 block, waves 0,1 compute,
// all four stage -- 48 rows of operands per chunk instead of 64, i.e. a quarter less of the staging work that is this
// kernel's overhead above the chain.
template <int K, int BN>
__global__ __launch_bounds__(256)
void fc_gemm_chain_kernel(const float* __restrict__ A, const float* __restrict__ W, const float* __restrict__ bias,
                          float* __restrict__ C, int M, int N, int relu)
{
    static_assert(CH_DEPTH == 4 && (BN == 32 || BN == 16), "the ring below is written out for four slots");
    constexpr int NROWS = 32 + BN, NGT = NROWS / 16, BUF = NROWS * CH_K * 4;
    using Slot = ChSlot<NGT>;
    extern __shared__ __attribute__((aligned(16))) unsigned char ch_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * BN;

    // ---- staging role: thread (r0 = tid/16, gi = tid%16) moves the 32-byte group gi of rows r0, r0+16 (A: windows
    //      m0 + ..), r0+32 and, for BN = 32, r0+48 (W: neurons n0 + ..) of every chunk: 16 lanes cover one row's 512 B
    const int r0 = tid >> 4, gi = tid & 15;
    const size_t rowb = (size_t)K * 4;
    const unsigned voffA0 = (unsigned)((m0 + r0 < M ? m0 + r0 : M - 1) * rowb + 32 * gi);
    const unsigned voffA1 = (unsigned)((m0 + r0 + 16 < M ? m0 + r0 + 16 : M - 1) * rowb + 32 * gi);
    const unsigned voffW0 = (unsigned)((n0 + r0) * rowb + 32 * gi);
    const unsigned voffW1 = (unsigned)((n0 + r0 + (BN == 32 ? 16 : 0)) * rowb + 32 * gi);
    constexpr int nch = K / CH_K;
    static_assert(K % CH_K == 0 && nch > CH_DEPTH, "whole chunks, more of them than ring slots");
    // chunk c lives in ring slot c % 4; loads past the last chunk re-read it (never used)
    auto fetch = [&](Slot& s, int c) {
        const size_t o = (size_t)(c < nch ? c : nch - 1) * (CH_K * 4);
        const char* pa = reinterpret_cast<const char*>(A) + o;
        const char* pw = reinterpret_cast<const char*>(W) + o;
        CH_LD(s.x[0], voffA0, pa, 0); CH_LD(s.y[0], voffA0, pa, 16);
        CH_LD(s.x[1], voffA1, pa, 0); CH_LD(s.y[1], voffA1, pa, 16);
        CH_LD(s.x[2], voffW0, pw, 0); CH_LD(s.y[2], voffW0, pw, 16);
        if constexpr (NGT == 4) { CH_LD(s.x[3], voffW1, pw, 0); CH_LD(s.y[3], voffW1, pw, 16); }
    };
    // each group of 8 k is stored as [k0 k2 k4 k6 | k1 k3 k5 k7]; 16-byte slot index XOR row (row & 15 = r0 for all of
    // a thread's rows); group j sits 16 rows = 8 KB behind group j-1
    const int st0 = r0 * (CH_K * 4) + (((2 * gi + 0) ^ r0) << 4), st1 = r0 * (CH_K * 4) + (((2 * gi + 1) ^ r0) << 4);
    auto stage_half = [&](const Slot& s, int buf, int j, int h) {
        unsigned char* b = ch_lds + buf * BUF + j * (16 * CH_K * 4);
        *reinterpret_cast<float4*>(b + (h == 0 ? st0 : st1)) = h == 0 ? make_float4(s.x[j].x, s.x[j].z, s.y[j].x, s.y[j].z)
                                                                      : make_float4(s.x[j].y, s.x[j].w, s.y[j].y, s.y[j].w);
    };

    // ---- compute role: wave (wm = w&1, wn = w>>1) owns C[m0 + 16 wm .. +16][n0 + 16 wn .. +16] (BN = 16: waves 2,3 run
    //      the same instruction stream on rows that belong to nobody and store nothing); lane (i, g):
    //      A[i][k] and W[i][k] operands of lane group g sit in slot 2s + (g>>1), half g&1 of row i
    const int wm = wv & 1, wn = wv >> 1, i = lane & 15, g = lane >> 4;
    const int t = i ^ (g >> 1);
    int a_off[8], b_off[8];                                    // byte offsets for slices s = 0..7 (s + 8: +256 B)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int o = (((2 * s) ^ t) << 4) + (g & 1) * 8;
        a_off[s] = (16 * wm + i) * (CH_K * 4) + o;
        b_off[s] = (32 + (BN == 32 ? 16 * wn : 0) + i) * (CH_K * 4) + o;
    }
    ch_f32x2 fa[2][8], fb[2][8];                               // fragments of half a chunk (8 slices), two sets
    auto rd = [&](int set, int buf, int half, int s) {
        const unsigned char* b = ch_lds + buf * BUF + half * 256;
        fa[set][s] = *reinterpret_cast<const ch_f32x2*>(b + a_off[s]);
        fb[set][s] = *reinterpret_cast<const ch_f32x2*>(b + b_off[s]);
    };
    ch_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, tot = {0.f, 0.f, 0.f, 0.f};     // running K range / finished ranges (fc_tree.h)
    constexpr int cut1 = fc_tree_unit(K / 128, 1), cut2 = fc_tree_unit(K / 128, 2), cut3 = fc_tree_unit(K / 128, 3);   // first chunk of ranges 1..3
    static_assert(CH_K == 128, "the tree's ranges are whole chunks");
    auto mfma2 = [&](int set, int s) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][s].x, fb[set][s].x, acc, 0, 0, 0);    // k = 0,4,1,5
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][s].y, fb[set][s].y, acc, 0, 0, 0);    // k = 2,6,3,7
    };

    Slot r0_, r1_, r2_, r3_;
    fetch(r0_, 0); fetch(r1_, 1); fetch(r2_, 2); fetch(r3_, 3);
    ch_wait_oldest(r0_);
#pragma unroll
    for (int j = 0; j < NGT; ++j) { stage_half(r0_, 0, j, 0); stage_half(r0_, 0, j, 1); }
    fetch(r0_, 4);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 8; ++s) rd(0, 0, 0, s);
    // iteration c: chunk c+1 goes to LDS, the second half of chunk c is read while its first half runs on the matrix
    // pipe, and the first half of chunk c+1 is read (behind the barrier that publishes it) while the second half runs.
    // Every iteration is the same straight-line code: past the last chunk the ring re-reads it and the "next chunk"
    // that is staged and half-read is never used.  The MFMA chain is the critical path (an MFMA holds the issue port
    // for 4 of its 35 cycles), so everything else is dealt out between its links, slice by slice, and
    // sched_barrier(0) keeps hipcc from bunching it up again (left alone it sinks every LDS read to just before its
    // MFMA and the chain waits out the LDS latency 16 times per chunk).
#define CH_SB __builtin_amdgcn_sched_barrier(0)
#define CH_A(c, NEXT, s, j, h) mfma2(0, s); rd(1, (c) & 1, 1, s); if constexpr ((j) < NGT) stage_half(NEXT, ((c) + 1) & 1, j, h);
#define CH_ITER(c, NEXT)                                                                              \
    {                                                                                                 \
        const size_t o_ = (size_t)((c) + 5 < nch ? (c) + 5 : nch - 1) * (CH_K * 4);                   \
        const char* pa_ = reinterpret_cast<const char*>(A) + o_;                                      \
        const char* pw_ = reinterpret_cast<const char*>(W) + o_;                                      \
        ch_wait_oldest(NEXT); CH_SB;                                                                  \
        /* a new K range of the summation tree starts with chunk c: written as selects on a wave-uniform condition -- a   \
           branch here makes hipcc rotate the loop (body ahead of its header), which the linear asm-load guard of          \
           tests/test_abi.py cannot follow */                                                                              \
        { const bool cut_ = ((c) == cut1) | ((c) == cut2) | ((c) == cut3);                                                 \
          const ch_f32x4 s_ = tot + acc;                                                                                   \
          _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { tot[e_] = cut_ ? s_[e_] : tot[e_]; acc[e_] = cut_ ? 0.f : acc[e_]; } } \
        CH_A(c, NEXT, 0, 0, 0) CH_SB;                                                                 \
        CH_A(c, NEXT, 1, 0, 1) CH_LD(NEXT.x[0], voffA0, pa_, 0); CH_LD(NEXT.y[0], voffA0, pa_, 16); CH_SB; \
        CH_A(c, NEXT, 2, 1, 0) CH_SB;                                                                 \
        CH_A(c, NEXT, 3, 1, 1) CH_LD(NEXT.x[1], voffA1, pa_, 0); CH_LD(NEXT.y[1], voffA1, pa_, 16); CH_SB; \
        CH_A(c, NEXT, 4, 2, 0) CH_SB;                                                                 \
        CH_A(c, NEXT, 5, 2, 1) CH_LD(NEXT.x[2], voffW0, pw_, 0); CH_LD(NEXT.y[2], voffW0, pw_, 16); CH_SB; \
        CH_A(c, NEXT, 6, 3, 0) CH_SB;                                                                 \
        CH_A(c, NEXT, 7, 3, 1) if constexpr (NGT == 4) { CH_LD(NEXT.x[NGT - 1], voffW1, pw_, 0); CH_LD(NEXT.y[NGT - 1], voffW1, pw_, 16); } CH_SB; \
        __syncthreads(); CH_SB;                                                                       \
        _Pragma("unroll") for (int s_ = 0; s_ < 8; ++s_) { mfma2(1, s_); rd(0, ((c) + 1) & 1, 0, s_); CH_SB; } \
    }
    constexpr int groups = nch / CH_DEPTH, rem = nch % CH_DEPTH;
#pragma unroll 1
    for (int c0 = 0; c0 < groups * CH_DEPTH; c0 += CH_DEPTH) {
        CH_ITER(c0 + 0, r1_) CH_ITER(c0 + 1, r2_) CH_ITER(c0 + 2, r3_) CH_ITER(c0 + 3, r0_)
    }
    if constexpr (rem > 0) CH_ITER(groups * CH_DEPTH + 0, r1_)
    if constexpr (rem > 1) CH_ITER(groups * CH_DEPTH + 1, r2_)
    if constexpr (rem > 2) CH_ITER(groups * CH_DEPTH + 2, r3_)
#undef CH_ITER
#undef CH_A
#undef CH_SB
    // The ring's tail loads (chunks past the end) are still in flight and the compiler does not know it: their
    // destination registers must stay allocated until they have landed, or a late load overwrites whatever the
    // epilogue put there (first version: a store address -> memory aperture violation).
    ch_hold(r0_, r1_, true);
    ch_hold(r2_, r3_, false);
    // D[row = 4 g + r][col = i]
    if (BN == 16 && wn != 0) return;
    const int n = n0 + 16 * wn + i;
    const float bv = bias[n];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 16 * wm + 4 * g + r;
        if (m < M) {
            float v = (tot[r] + acc[r]) + bv;
            if (relu) v = v < 0.f ? 0.f : v;                   // NaN stays NaN, as in fc_gemm.hip
            C[(size_t)m * N + n] = v;
        }
    }
}

hipError_t init_fc_gemm_chain()
{
    for (const void* k : {reinterpret_cast<const void*>(&fc_gemm_chain_kernel<FEAT, 32>), reinterpret_cast<const void*>(&fc_gemm_chain_kernel<FEAT, 16>),
                          reinterpret_cast<const void*>(&fc_gemm_chain_kernel<FC1, 32>), reinterpret_cast<const void*>(&fc_gemm_chain_kernel<FC1, 16>)}) {
        const hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, CH_LDS_BYTES);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// window counts served by this kernel: fc.0 (K = 4736) [DCE_CHAIN_MIN (9), DCE_CHAIN_MAX (640)], fc.3 (K = 2048)
// [DCE_CHAIN_MIN, DCE_CHAIN_MAX3 (2048)]; 0 as a maximum switches it off for that layer.  Measured (r2z): two
// workgroups fit a CU, a "round" of 512 workgroups takes ~45 us for fc.0 and ~12 us for fc.3, so fc.0 (64 workgroups per
// 32 windows) beats the 137 us of one round of 128x64 phased tiles up to 2.5 rounds, and fc.3 (16 per 32 windows) beats
// the fused 64x... phased kernel (71 us incl. combine) up to 2048 windows (47 + 7.5 us with the tail kernel).
bool fc_gemm_chain_ok(int64_t M, int N, int K)
{
    const Tuning& tu = tune();
    const int64_t lo = tu.chain_min, hi = tu.chain_max, hi3 = tu.chain_max3;
    // per-lane global offsets are 32-bit (row * K * 4 bytes): keep every row of the launch inside 4 GiB
    if ((uint64_t)M * (uint64_t)K * 4 >= (1ull << 32)) return false;
    return M >= lo && M <= (K == FEAT ? hi : hi3) && N % 32 == 0 && (K == FEAT || K == FC1);
}

hipError_t launch_fc_gemm_chain(const float* A, const float* W, const float* bias, float* C,
                                int64_t M, int N, int K, int relu, hipStream_t st)
{
    if (M <= 0) return hipSuccess;
    if (N % 32 || (K != FEAT && K != FC1) || M > 65535 * 32) return hipErrorInvalidValue;
    if ((uint64_t)M * (uint64_t)K * 4 >= (1ull << 32)) return hipErrorInvalidValue;     // 32-bit per-lane offsets (voffA = row*K*4)
    // 32x16 blocks while even they leave CUs idle (<= 64 windows: fc.0 256 workgroups, fc.3 64): a quarter less staging per chain
    const int64_t bn16_max = tune().chain_bn16_max;
    const dim3 block(256);
    plan_note(M <= bn16_max ? "fc_chain32x16" : "fc_chain32x32");
    if (M <= bn16_max) {
        const dim3 grid(N / 16, (unsigned)((M + 31) / 32));
        if (K == FEAT) hipLaunchKernelGGL((fc_gemm_chain_kernel<FEAT, 16>), grid, block, CH_LDS_BYTES, st, A, W, bias, C, (int)M, N, relu);
        else           hipLaunchKernelGGL((fc_gemm_chain_kernel<FC1, 16>), grid, block, CH_LDS_BYTES, st, A, W, bias, C, (int)M, N, relu);
    } else {
        const dim3 grid(N / 32, (unsigned)((M + 31) / 32));
        if (K == FEAT) hipLaunchKernelGGL((fc_gemm_chain_kernel<FEAT, 32>), grid, block, CH_LDS_BYTES, st, A, W, bias, C, (int)M, N, relu);
        else           hipLaunchKernelGGL((fc_gemm_chain_kernel<FC1, 32>), grid, block, CH_LDS_BYTES, st, A, W, bias, C, (int)M, N, relu);
    }
    return hipGetLastError();
}

}  // namespace dce
