// fc_gemm.hip -- the three nn.Linear layers of contact_cnn on gfx950 (MI355X).
//
// Replaces what PyTorch dispatches for  src/contact_cnn.py:47-58  (fc.0 + ReLU, fc.3 + ReLU,
// fc.6; Dropout = identity in eval) and for the loop epilogue  src/inference_one_seq.py:26-27
// (torch.max(output,1) + decimal2binary, :59-62).
//
//  * fc_gemm_kernel: C[M,N] = act(A[M,K] W[N,K]^T + b) on v_mfma_f32_32x32x2_f32 -- exact fp32
//    (k-ordered fmaf chain, deterministic) at the fp32 peak rate.  128x128x32 block tile,
//    4 waves as 2x2, each 64x64 (2x2 MFMA tiles, 64 accumulator VGPRs), double-buffered LDS
//    with register staging (global_load_dwordx4 of tile t+1 issued before the MFMAs of tile t,
//    ds_write_b128 after), rows padded to 36 floats so the ds_read_b128 fragment reads are
//    bank-conflict free.  Both operands are K-contiguous (PyTorch's [out][in] layout needs no
//    repack): a lane reads 4 consecutive k of its row and feeds 4 MFMAs; lanes 0-31 / 32-63
//    take k-quads 0 / 1 of each 8-wide K slice for A and B alike.
//  * blockIdx -> tile map is XCD-aware: the 64 blocks co-resident on one XCD (32 CUs x 2) form
//    an 8x8 (or 16x4) super-tile, so each A row-panel and W column-panel is fetched into that
//    XCD's L2 once per 8 (4) consumers.
//  * fc3_tail_kernel: fc.6 as the fixed summation tree of fc6_chain.h (8 chunk chains on
//    v_mfma_f32_16x16x4_f32, combined in order) over h2 rows staged in LDS, then argmax with torch.max
//    semantics (first maximum; a NaN wins, first NaN first) and the 4-bit unpack, MSB = leg 0.
//    Chip-filling fp32 batches never come here: their fc.3 GEMM finishes the chunk chains in its
//    epilogue (fc_gemm_phased.hip) and fc6_combine_kernel adds them up.
#include "dce_kernels.h"
#include "fc6_chain.h"
#include "fc_tree.h"
#include <cstdlib>
#include <cstring>

namespace dce {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One K-tile is 128 BYTES of K per row in either precision (32 floats / 64 bf16); LDS rows are
// padded to 144 B so the 16-B fragment reads of 16 consecutive rows hit 16 distinct slots.
constexpr int KT_BYTES = 128, LDR = KT_BYTES + 16;             // (64-byte K-tiles: 4 blocks/CU but -14 %)
constexpr int CPR = KT_BYTES / 16;                               // 16-B columns per staged row

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float relu_nan(float v) { return v < 0.f ? 0.f : v; }

__device__ __forceinline__ unsigned short f32_to_bf16_rne(float f)
{   // round-to-nearest-even; NaN stays NaN (quiet)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

template <int TM, int TN, int WGN = 2> struct GemmCfg {
    static constexpr int NT = 128 * WGN;                         // threads: 2 x WGN waves
    static constexpr int RPP = NT / CPR;                         // rows per staging pass
    static constexpr int BM = 64 * TM, BN = 32 * TN * WGN;        // block tile
    static constexpr int A_BYTES = BM * LDR, B_BYTES = BN * LDR;
    static constexpr int LDS_BYTES = 2 * (A_BYTES + B_BYTES);
};

// C[M,N] = act(A[M,K] W[N,K]^T + bias).  TM x TN MFMA 32x32 tiles per wave; block = 2x2 waves.
//   BF16 = false: A, W fp32; v_mfma_f32_32x32x2_f32  (exact fp32)            -- the headline path
//   BF16 = true : A, W bf16; v_mfma_f32_32x32x16_bf16, fp32 accumulate       -- DCE_BF16_FC
//   OUT_BF16    : store C as bf16 (input of the next bf16 GEMM) instead of fp32
//   WGN         : waves along N (2: 256 threads, 2 waves/SIMD at 2 blocks/CU; 4: 512 threads, 4 waves/SIMD)
template <int TM, int TN, bool BF16, bool OUT_BF16, int WGN = 2>
__global__ __launch_bounds__(128 * WGN, WGN)
void fc_gemm_kernel(const void* __restrict__ Av, const void* __restrict__ Wv,
                    const float* __restrict__ bias, void* __restrict__ Cv,
                    int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2, Gate gate = Gate{})
{
    using Cfg = GemmCfg<TM, TN, WGN>;
    if (gate_closed(gate)) return;                       // DCE_FP32_SPLIT's fallback sequence (dce_kernels.h Gate)
    constexpr int BM = Cfg::BM, BN = Cfg::BN, RPP = Cfg::RPP;
    constexpr int SA = BM / RPP, SB = BN / RPP;          // 16-B pieces staged per thread per K-tile
    constexpr int ES = BF16 ? 2 : 4;                     // element size
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                     // [2][BM][LDR]
    char* Bs = smem + 2 * Cfg::A_BYTES;                  // [2][BN][LDR]
    const char* A = static_cast<const char*>(Av);
    const char* W = static_cast<const char*>(Wv);

    // ---- XCD-aware tile assignment (speed only; any placement is correct)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const int sid = (li >> 6) * 8 + xcd;                 // super-tile id
    const int within = li & 63;
    const int sn = 1 << sn_log2, sm = 64 >> sn_log2;     // super-tile = sm x sn tiles
    const int nsn = ntiles >> sn_log2;                   // super-tiles along N
    const int tm = (sid / nsn) * sm + (within >> sn_log2);
    const int tn = (sid % nsn) * sn + (within & (sn - 1));
    if (tm >= mtiles) return;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wv / WGN) * 32 * TM, wn = (wv % WGN) * 32 * TN;
    const int i = lane & 31, h = lane >> 5;

    // staging: thread -> (row = tid/8 + 32*s, 16-byte column tid%8)
    const int srow = tid / CPR, sk4 = tid % CPR;
    const size_t rowb = (size_t)K * ES;                  // bytes per operand row
    const char* ag[SA];
    const char* bg[SB];
#pragma unroll
    for (int s = 0; s < SA; ++s) {
        int ra = m0 + srow + RPP * s;
        ra = ra < M ? ra : M - 1;                        // clamp: rows >= M are computed, never stored
        ag[s] = A + (size_t)ra * rowb + 16 * sk4;
    }
#pragma unroll
    for (int s = 0; s < SB; ++s) bg[s] = W + (size_t)(n0 + srow + RPP * s) * rowb + 16 * sk4;
    const int sdst = srow * LDR + 16 * sk4;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fp32: the fixed summation tree of fc_tree.h -- `tot` collects the finished K ranges, acc the running one
    f32x16 tot[BF16 ? 1 : TM][BF16 ? 1 : TN];
    if constexpr (!BF16) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) tot[a][b][r] = 0.f;
    }
    auto fold = [&]() {
        if constexpr (!BF16) {
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { tot[a][b][r] += acc[a][b][r]; acc[a][b][r] = 0.f; }
        }
    };
    const FcTree tree = fc_tree(BF16 ? 1 : K, KT_BYTES / 4);

    static_assert(SA == SB && (SA == 4 || SA == 2 || SA == 1), "staging registers are named: 1, 2 or 4 per operand");
    {   // first K-tile: plain loads
        float4 t[SA + SB];
#pragma unroll
        for (int s = 0; s < SA; ++s) t[s] = *reinterpret_cast<const float4*>(ag[s]);
#pragma unroll
        for (int s = 0; s < SB; ++s) t[SA + s] = *reinterpret_cast<const float4*>(bg[s]);
#pragma unroll
        for (int s = 0; s < SA; ++s) *reinterpret_cast<float4*>(As + sdst + RPP * s * LDR) = t[s];
#pragma unroll
        for (int s = 0; s < SB; ++s) *reinterpret_cast<float4*>(Bs + sdst + RPP * s * LDR) = t[SA + s];
    }
    __syncthreads();
    v4f ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;          // staging registers (native vectors: asm operands)

    const int KT = (int)(rowb / KT_BYTES);
    const int fa = (wm + i) * LDR + 16 * h;              // this lane's fragment row, 16-B half h
    const int fb = (wn + i) * LDR + 16 * h;
    // The staging loads are issued as inline asm: written as plain loads, LLVM sinks them (at IR
    // level, below any sched_barrier) to just before the ds_writes at the end of the iteration, and
    // every K-tile then waits out their full L2/HBM latency.  The asm loads are invisible to the
    // compiler's s_waitcnt bookkeeping, so the wait before their first use is explicit and names
    // every destination register (cdna_hip_programming.md 5.7, form ii).  Loads past the last tile
    // re-read it (unconditional loads keep the registers out of scratch and the loop branch-free).
#define DCE_GLOAD16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr))
#define DCE_ISSUE(P, tile)                                                                        \
    { const size_t koff = (size_t)((tile) < KT ? (tile) : KT - 1) * KT_BYTES;                     \
      DCE_GLOAD16(P##a0, ag[0] + koff); DCE_GLOAD16(P##b0, bg[0] + koff);                         \
      if constexpr (SA >= 2) { DCE_GLOAD16(P##a1, ag[1] + koff); DCE_GLOAD16(P##b1, bg[1] + koff); } \
      if constexpr (SA == 4) { DCE_GLOAD16(P##a2, ag[2] + koff); DCE_GLOAD16(P##a3, ag[3] + koff);    \
                               DCE_GLOAD16(P##b2, bg[2] + koff); DCE_GLOAD16(P##b3, bg[3] + koff); } }
    // wait until at most N4 / N2 / N1 (for 4 / 2 / 1 pieces per operand) younger loads are outstanding,
    // then hand set P to LDS buffer `buf`
#define DCE_WAIT_WRITE(P, buf, N4, N2, N1)                                                        \
    { char* ad = As + (buf) * Cfg::A_BYTES + sdst;                                                \
      char* bd = Bs + (buf) * Cfg::B_BYTES + sdst;                                                \
      if constexpr (SA == 4) {                                                                    \
          asm volatile("s_waitcnt vmcnt(" #N4 ")" : "+v"(P##a0), "+v"(P##a1), "+v"(P##a2), "+v"(P##a3), \
                                                   "+v"(P##b0), "+v"(P##b1), "+v"(P##b2), "+v"(P##b3)); \
          *reinterpret_cast<v4f*>(ad + 2 * RPP * LDR) = P##a2; *reinterpret_cast<v4f*>(ad + 3 * RPP * LDR) = P##a3; \
          *reinterpret_cast<v4f*>(bd + 2 * RPP * LDR) = P##b2; *reinterpret_cast<v4f*>(bd + 3 * RPP * LDR) = P##b3; \
      } else if constexpr (SA == 2) {                                                             \
          asm volatile("s_waitcnt vmcnt(" #N2 ")" : "+v"(P##a0), "+v"(P##a1), "+v"(P##b0), "+v"(P##b1)); \
      } else {                                                                                    \
          asm volatile("s_waitcnt vmcnt(" #N1 ")" : "+v"(P##a0), "+v"(P##b0));                    \
      }                                                                                           \
      *reinterpret_cast<v4f*>(ad) = P##a0; *reinterpret_cast<v4f*>(bd) = P##b0;                   \
      if constexpr (SA >= 2) {                                                                    \
          *reinterpret_cast<v4f*>(ad + RPP * LDR) = P##a1; *reinterpret_cast<v4f*>(bd + RPP * LDR) = P##b1; \
      } }

    auto compute = [&](int cur) {
        const char* as = As + cur * Cfg::A_BYTES + fa;
        const char* bs = Bs + cur * Cfg::B_BYTES + fb;
#pragma unroll
        for (int kq = 0; kq < KT_BYTES / 32; ++kq) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *reinterpret_cast<const float4*>(as + 32 * a * LDR + 32 * kq);
#pragma unroll
            for (int b = 0; b < TN; ++b) bf[b] = *reinterpret_cast<const float4*>(bs + 32 * b * LDR + 32 * kq);
            if constexpr (BF16) {
                // lane (i,h) holds k = 16*kq + 8*h .. +7 of its row: the 32x32x16 fragment
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, af[a]), __builtin_bit_cast(bf16x8, bf[b]), acc[a][b], 0, 0, 0);
            } else {
                // lanes 0-31 / 32-63 take k-quads 0 / 1 of each 8-wide K slice: 4 MFMAs of K=2
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            const float av = u == 0 ? af[a].x : u == 1 ? af[a].y : u == 2 ? af[a].z : af[a].w;
                            const float bv = u == 0 ? bf[b].x : u == 1 ? bf[b].y : u == 2 ? bf[b].z : bf[b].w;
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                        }
            }
        }
    };

    // one tile of lookahead.  (Two tiles ahead for the bf16 path, whose K-tile is only 512 MFMA cycles per
    // wave, measured fc.0 0.0985 vs 0.0957 ms and fc.3 0.021 vs 0.024 ms: a wash -- that path is bound by
    // LDS bandwidth, 128 B/clk/CU at one ds_read_b128 per MFMA, not by load latency.)
    for (int kt = 0; kt < KT; ++kt) {
        DCE_ISSUE(r, kt + 1)
        if (!BF16 && fc_tree_cut(tree, kt)) fold();           // a new K range starts with this tile
        compute(kt & 1);
        DCE_WAIT_WRITE(r, (kt & 1) ^ 1, 0, 0, 0)
        __syncthreads();
    }
#undef DCE_WAIT_WRITE
#undef DCE_ISSUE
#undef DCE_GLOAD16

    // ---- epilogue: bias + (ReLU) ; D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn + 32 * b + i;
        const float bv = bias[col];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = (BF16 ? acc[a][b][r] : tot[a][b][r] + acc[a][b][r]) + bv;
                if (relu) v = relu_nan(v);
                if (row < M) {
                    if constexpr (OUT_BF16) static_cast<unsigned short*>(Cv)[(size_t)row * N + col] = f32_to_bf16_rne(v);
                    else static_cast<float*>(Cv)[(size_t)row * N + col] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// 64x64 tiles for batches that do not fill the chip (33 .. ~3000 windows: at most one block per CU).
// Nothing then overlaps a block's K-tile with another's, and a K-tile (16 MFMAs per wave, ~0.45 us)
// is shorter than the memory round trip of its successor: with one tile of lookahead fc.0 walked
// K at 0.75 us per tile.  Here the staging loads run GS_DEPTH tiles ahead through a register ring
// (asm loads + counted vmcnt, as above).  Tiling, K order and epilogue are fc_gemm_kernel<1,1>'s:
// bit-identical results.
// ------------------------------------------------------------------------------------------
constexpr int GS_DEPTH = 4;

__global__ __launch_bounds__(256, 2)
void fc_gemm_small_kernel(const float* __restrict__ Af, const float* __restrict__ Wf,
                          const float* __restrict__ bias, float* __restrict__ C,
                          int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2, Gate gate = Gate{})
{
    using Cfg = GemmCfg<1, 1, 2>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, RPP = Cfg::RPP;
    if (gate_closed(gate)) return;
    static_assert(BM / RPP == 2 && BN / RPP == 2 && KT_BYTES == 128, "two 16-byte pieces per operand per thread");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;
    char* Bs = smem + 2 * Cfg::A_BYTES;
    const char* A = reinterpret_cast<const char*>(Af);
    const char* W = reinterpret_cast<const char*>(Wf);

    const int bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const int sid = (li >> 6) * 8 + xcd;
    const int within = li & 63;
    const int sn = 1 << sn_log2, sm = 64 >> sn_log2;
    const int nsn = ntiles >> sn_log2;
    const int tm = (sid / nsn) * sm + (within >> sn_log2);
    const int tn = (sid % nsn) * sn + (within & (sn - 1));
    if (tm >= mtiles) return;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32;
    const int i = lane & 31, h = lane >> 5;
    const int srow = tid / CPR, sk4 = tid % CPR;
    const size_t rowb = (size_t)K * 4;
    int r0 = m0 + srow, r1 = m0 + srow + RPP;
    r0 = r0 < M ? r0 : M - 1; r1 = r1 < M ? r1 : M - 1;
    const char* ag0 = A + (size_t)r0 * rowb + 16 * sk4;
    const char* ag1 = A + (size_t)r1 * rowb + 16 * sk4;
    const char* bg0 = W + (size_t)(n0 + srow) * rowb + 16 * sk4;
    const char* bg1 = W + (size_t)(n0 + srow + RPP) * rowb + 16 * sk4;
    const int sdst = srow * LDR + 16 * sk4;

    f32x16 acc, tot;                                     // running K range / finished ranges (fc_tree.h)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; tot[r] = 0.f; }
    const int KT = (int)(rowb / KT_BYTES);               // a multiple of GS_DEPTH (checked by the launcher)
    const FcTree tree = fc_tree(K, KT_BYTES / 4);

    *reinterpret_cast<float4*>(As + sdst) = *reinterpret_cast<const float4*>(ag0);
    *reinterpret_cast<float4*>(As + sdst + RPP * LDR) = *reinterpret_cast<const float4*>(ag1);
    *reinterpret_cast<float4*>(Bs + sdst) = *reinterpret_cast<const float4*>(bg0);
    *reinterpret_cast<float4*>(Bs + sdst + RPP * LDR) = *reinterpret_cast<const float4*>(bg1);

    // ring slot of tile t = t % 4; four named register quads (a0, a1, b0, b1) per slot
    v4f q0a0, q0a1, q0b0, q0b1, q1a0, q1a1, q1b0, q1b1, q2a0, q2a1, q2b0, q2b1, q3a0, q3a1, q3b0, q3b1;
    // wave-uniform 64-bit base (SGPRs, advanced by SALU) + constant 32-bit lane offsets: the K walk issues no VALU
    // instruction (a VALU instruction here breaks the back-to-back MFMA stream of the block sharing the SIMDs --
    // fc_gemm_phased.hip measured ~34 matrix-pipe cycles per add)
    const char* sAb = A + (size_t)m0 * rowb;
    const char* sWb = W + (size_t)n0 * rowb;
    const unsigned oa0 = (unsigned)((size_t)(r0 - m0) * rowb + 16 * sk4), oa1 = (unsigned)((size_t)(r1 - m0) * rowb + 16 * sk4);
    const unsigned ob0 = (unsigned)((size_t)srow * rowb + 16 * sk4), ob1 = (unsigned)((size_t)(srow + RPP) * rowb + 16 * sk4);
#define GS_LD(dst, voff, sbase) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase))
#define GS_LOAD(S, tile)                                                                          \
    { const size_t ko = (size_t)((tile) < KT ? (tile) : KT - 1) * KT_BYTES;                       \
      GS_LD(q##S##a0, oa0, sAb + ko); GS_LD(q##S##a1, oa1, sAb + ko); GS_LD(q##S##b0, ob0, sWb + ko); GS_LD(q##S##b1, ob1, sWb + ko); }
    GS_LOAD(1, 1) GS_LOAD(2, 2) GS_LOAD(3, 3)
    __syncthreads();

    const int fa = (wm + i) * LDR + 16 * h, fb = (wn + i) * LDR + 16 * h;
    auto compute = [&](int cur) {
        const char* as = As + cur * Cfg::A_BYTES + fa;
        const char* bs = Bs + cur * Cfg::B_BYTES + fb;
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const float4 af = *reinterpret_cast<const float4*>(as + 32 * kq);
            const float4 bf = *reinterpret_cast<const float4*>(bs + 32 * kq);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
    };
    // one K-tile: refill the slot tile kt left free with tile kt+4, multiply tile kt out of LDS, then
    // hand tile kt+1 (the oldest of the four in flight: 12 younger loads may stay outstanding) to LDS
#define GS_STEP(SFREE, SNEXT, kt)                                                                 \
    { GS_LOAD(SFREE, (kt) + GS_DEPTH)                                                             \
      if (fc_tree_cut(tree, (kt))) { _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) { tot[r_] += acc[r_]; acc[r_] = 0.f; } } \
      compute((kt) & 1);                                                                          \
      asm volatile("s_waitcnt vmcnt(12)" : "+v"(q##SNEXT##a0), "+v"(q##SNEXT##a1), "+v"(q##SNEXT##b0), "+v"(q##SNEXT##b1)); \
      char* ad = As + (((kt) + 1) & 1) * Cfg::A_BYTES + sdst;                                     \
      char* bd = Bs + (((kt) + 1) & 1) * Cfg::B_BYTES + sdst;                                     \
      *reinterpret_cast<v4f*>(ad) = q##SNEXT##a0; *reinterpret_cast<v4f*>(ad + RPP * LDR) = q##SNEXT##a1; \
      *reinterpret_cast<v4f*>(bd) = q##SNEXT##b0; *reinterpret_cast<v4f*>(bd + RPP * LDR) = q##SNEXT##b1; \
      __syncthreads(); }
    for (int kt = 0; kt < KT; kt += GS_DEPTH) {
        GS_STEP(0, 1, kt) GS_STEP(1, 2, kt + 1) GS_STEP(2, 3, kt + 2) GS_STEP(3, 0, kt + 3)
    }
    // the ring's trailing (clamped) loads: waited for while their registers are still live (hipcc does
    // not know loads are in flight into them and would otherwise reuse them for the epilogue)
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(q0a0), "+v"(q0a1), "+v"(q0b0), "+v"(q0b1), "+v"(q1a0), "+v"(q1a1), "+v"(q1b0), "+v"(q1b1),
                                        "+v"(q2a0), "+v"(q2a1), "+v"(q2b0), "+v"(q2b1), "+v"(q3a0), "+v"(q3a1), "+v"(q3b0), "+v"(q3b1));
#undef GS_STEP
#undef GS_LOAD
#undef GS_LD

    const int col = n0 + wn + i;
    const float bv = bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = (tot[r] + acc[r]) + bv;
        if (relu) v = relu_nan(v);
        if (row < M) C[(size_t)row * N + col] = v;
    }
}

template <int TM, int TN, bool BF16, bool OUT_BF16, int WGN = 2>
static hipError_t grant_lds()
{
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_kernel<TM, TN, BF16, OUT_BF16, WGN>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<TM, TN, WGN>::LDS_BYTES);
}

constexpr int TAIL_WINDOWS = 16;       // fc3_tail_kernel: windows per 256-thread block (16 lanes per window)
constexpr int TAIL_H2_LD = FC2 + 4;     // 516 floats: the 16 rows of a ds_read_b128 start 4 banks apart
constexpr int TAIL_PART_FLOATS = FC6_NCHUNK * TAIL_WINDOWS * NCLS;
constexpr int TAIL_LDS_BYTES = (TAIL_WINDOWS * TAIL_H2_LD + TAIL_PART_FLOATS + TAIL_WINDOWS * NCLS) * (int)sizeof(float);
__global__ void fc3_tail_kernel(const float*, const float*, const float*, int64_t, float*, int32_t*, uint8_t*,
                                unsigned*, unsigned, unsigned*, uint8_t*, Gate);

hipError_t init_fc_gemm()
{
    hipError_t e;
    if ((e = grant_lds<2, 2, false, false>()) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc3_tail_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, TAIL_LDS_BYTES)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_small_kernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, GemmCfg<1, 1, 2>::LDS_BYTES)) != hipSuccess) return e;
    if ((e = grant_lds<1, 1, false, false>()) != hipSuccess) return e;
    if ((e = grant_lds<2, 2, true, true>()) != hipSuccess) return e;
    if ((e = grant_lds<2, 2, true, false>()) != hipSuccess) return e;
    if ((e = grant_lds<1, 1, true, true>()) != hipSuccess) return e;
    if ((e = init_fc_gemm_phased()) != hipSuccess) return e;
    if ((e = init_fc_gemm_chain()) != hipSuccess) return e;
    if ((e = init_fc_gemv()) != hipSuccess) return e;
    if ((e = init_fc_split()) != hipSuccess) return e;
    return grant_lds<1, 1, true, false>();
}

template <int TM, int TN, bool BF16, bool OUT_BF16, int WGN = 2>
static hipError_t launch_gemm_cfg(const void* A, const void* W, const float* bias, void* C,
                                  int64_t M, int N, int K, int relu, hipStream_t st)
{
    using Cfg = GemmCfg<TM, TN, WGN>;
    const int mtiles = (int)((M + Cfg::BM - 1) / Cfg::BM), ntiles = N / Cfg::BN;
    int sn_log2 = 3;                                   // super-tile 8 x 8 ...
    while ((1 << sn_log2) > ntiles) --sn_log2;         // ... or (64/ntiles) x ntiles when N is narrow
    const int sm = 64 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 64;
    plan_note(BF16 ? (TM == 2 ? "fc_tile128_bf16" : "fc_tile64_bf16") : (TM == 2 ? "fc_tile128" : "fc_tile64"));
    hipLaunchKernelGGL((fc_gemm_kernel<TM, TN, BF16, OUT_BF16, WGN>), dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, st,
                       A, W, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2, t_gate);
    return hipGetLastError();
}

static hipError_t launch_fc_gemm_one(const float* A, const float* W, const float* bias, float* C,
                                     int64_t M, int N, int K, int relu, hipStream_t st, bool remainder = false);

// Rows are independent and every kernel of the library produces the same bits for a row, so a batch may be cut by rows and
// each piece given to the kernel that suits its size.  The phased GEMMs run one workgroup per CU: a launch lasts whole
// ROUNDS of 256 tiles (fc.0: 4096 rows of 256x128 tiles = 525 us, 1024 rows of 128x64 tiles = 137 us), and a batch that
// ends just past a round pays a full one for the tail -- 4200 windows = 2 rounds of the big tile (1050 us) or 5 of the
// small one (685 us).  So: whole big rounds, then whole small rounds, then the remainder on its own (chain kernel up to 640
// rows, GEMV up to 8): 4200 windows = 525 + 31 us.  (DCE_GEMM_PEEL=0 switches the cut off.)
hipError_t launch_fc_gemm(const float* A, const float* W, const float* bias, float* C,
                          int64_t M, int N, int K, int relu, hipStream_t st)
{
    if (M <= 0) return hipSuccess;
    if (N % 128 || K % 32 || M > (1 << 30)) return hipErrorInvalidValue;
    const bool peel = tune().gemm_peel;
    const int64_t rows1 = (int64_t)(256 / (N / 64)) * 128, rows2 = (int64_t)(256 / (N / 128)) * 256;   // rows of one round
    if (peel && N <= 2048 && 256 % (N / 64) == 0 && M > rows1 && M % rows1 != 0 && fc_gemm_phased_ok(rows1, N, K, 0)) {
        const int64_t big = fc_gemm_phased_ok(rows2, N, K, 0) ? M / rows2 * rows2 : 0;
        const int64_t small = (M - big) / rows1 * rows1, rest = M - big - small;
        hipError_t e = hipSuccess;
        if (big) e = launch_fc_gemm_one(A, W, bias, C, big, N, K, relu, st);
        if (e == hipSuccess && small) e = launch_fc_gemm_one(A + big * K, W, bias, C + big * N, small, N, K, relu, st);
        if (e == hipSuccess && rest) e = launch_fc_gemm_one(A + (big + small) * K, W, bias, C + (big + small) * N, rest, N, K, relu, st, true);
        return e;
    }
    return launch_fc_gemm_one(A, W, bias, C, M, N, K, relu, st);
}

static hipError_t launch_fc_gemm_one(const float* A, const float* W, const float* bias, float* C,
                                     int64_t M, int N, int K, int relu, hipStream_t st, bool remainder)
{
    // a handful of rows as the remainder of a cut: the weight-streaming GEMV
    if (remainder && M <= 8 && N % 8 == 0 && K % 128 == 0) return launch_fc_gemv(A, W, bias, C, M, N, K, relu, st);
    // 9 .. ~100 windows: the four K ranges of the summation tree side by side on four waves (fc_gemm_split.hip)
    if (fc_split_ok(M, N, K)) return launch_fc_split(A, W, bias, C, M, N, K, relu, st);
    // a few dozen to a few hundred windows: chain-latency kernel, one 16x16 tile per wave (fc_gemm_chain.hip)
    if (fc_gemm_chain_ok(M, N, K)) return launch_fc_gemm_chain(A, W, bias, C, M, N, K, relu, st);
    // chip-filling sizes: one phased workgroup per CU (fc_gemm_phased.hip); same K order, same bits
    if (fc_gemm_phased_ok(M, N, K, 0)) return launch_fc_gemm_phased(A, W, bias, C, 0, 0, M, N, K, relu, st);
    // (In between, a no-LDS kernel -- one 16x16 / 32x32 output tile per wave on v_mfma_f32_16x16x4_f32, operands streamed
    //  from L2 straight into MFMA registers, bit-identical -- was built and measured in round 2: 1.4x - 3.2x SLOWER than the
    //  tile kernels below at 64 .. 2048 windows (fc.0 at 512 windows 518 vs 162 us: fragment-shaped 16-B-per-row loads keep
    //  the texture path busy; profiles/r2l_latency_*.txt).  The LDS-tiled kernels stay.)
    // 128x128 tiles when they alone fill the chip (512 resident blocks), else 64x64
    const int64_t big_blocks = ((M + 127) / 128) * (N / 128);
    if (big_blocks >= 384) return launch_gemm_cfg<2, 2, false, false>(A, W, bias, C, M, N, K, relu, st);
    const bool deep = tune().gemm_small_deep;
    const int64_t small_blocks = ((M + 63) / 64) * (N / 64);
    if (deep && small_blocks <= 512 && (K * 4 / KT_BYTES) % GS_DEPTH == 0) {
        using Cfg = GemmCfg<1, 1, 2>;
        const int mtiles = (int)((M + Cfg::BM - 1) / Cfg::BM), ntiles = N / Cfg::BN;
        int sn_log2 = 3;
        while ((1 << sn_log2) > ntiles) --sn_log2;
        const int sm = 64 >> sn_log2, nsn = ntiles >> sn_log2;
        const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
        const int grid = ((nsuper + 7) / 8) * 8 * 64;
        plan_note("fc_tile64_deep");
        hipLaunchKernelGGL(fc_gemm_small_kernel, dim3(grid), dim3(256), Cfg::LDS_BYTES, st,
                           A, W, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2, t_gate);
        return hipGetLastError();
    }
    return launch_gemm_cfg<1, 1, false, false>(A, W, bias, C, M, N, K, relu, st);
}

hipError_t launch_fc_gemm_bf16(const void* A, const void* W, const float* bias, void* C, int out_bf16,
                               int64_t M, int N, int K, int relu, hipStream_t st)
{
    if (M <= 0) return hipSuccess;
    if (N % 128 || K % 64 || M > (1 << 30)) return hipErrorInvalidValue;
    if (fc_gemm_phased_ok(M, N, K, 1)) return launch_fc_gemm_phased(A, W, bias, C, 1, out_bf16, M, N, K, relu, st);
    if (tune().bf16_stream && fc_stream_bf16_ok(M, N, K)) return launch_fc_stream_bf16(A, W, bias, C, out_bf16, M, N, K, relu, st);
    const int64_t big_blocks = ((M + 127) / 128) * (N / 128);
    if (big_blocks >= 384)
        return out_bf16 ? launch_gemm_cfg<2, 2, true, true>(A, W, bias, C, M, N, K, relu, st)
                        : launch_gemm_cfg<2, 2, true, false>(A, W, bias, C, M, N, K, relu, st);
    return out_bf16 ? launch_gemm_cfg<1, 1, true, true>(A, W, bias, C, M, N, K, relu, st)
                    : launch_gemm_cfg<1, 1, true, false>(A, W, bias, C, M, N, K, relu, st);
}

// ------------------------------------------------------------------------------------------
// fc.6 (512 -> 16) + torch.max(output,1) + decimal2binary
// ------------------------------------------------------------------------------------------

// decimal2binary (reference src/inference_one_seq.py:59-62): class -> 4 bits, MSB = leg 0
__device__ __forceinline__ uchar4 contact_bits(int best)
{
    uchar4 c;
    c.x = (best >> 3) & 1; c.y = (best >> 2) & 1; c.z = (best >> 1) & 1; c.w = best & 1;
    return c;
}

__global__ __launch_bounds__(256)
void fc3_tail_kernel(const float* __restrict__ h2, const float* __restrict__ W3,
                     const float* __restrict__ b3, int64_t n, float* __restrict__ logits,
                     int32_t* __restrict__ pred, uint8_t* __restrict__ contacts,
                     unsigned* __restrict__ done_flag, unsigned done_seq, unsigned* __restrict__ seq_counter,
                     uint8_t* __restrict__ packed, Gate gate)
{
    if (gate_closed(gate)) return;                       // DCE_FP32_SPLIT's fallback sequence (dce_kernels.h Gate)
#if DCE_EXPERIMENTS
    if (gate.taken && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(gate.taken, 1u);     // ... ran: counted (dce_split_guard_info)
#endif
    // dynamic LDS (43 KB): the 16 h2 rows of the current window tile [16][516] | chunk sums [8][16][16] | logits [16][16]
    extern __shared__ __attribute__((aligned(16))) float tsm[];
    float* h2s = tsm;
    float* part = tsm + TAIL_WINDOWS * TAIL_H2_LD;
    float (*lg)[NCLS] = reinterpret_cast<float (*)[NCLS]>(part + TAIL_PART_FLOATS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave w owns chunks 2w and 2w+1 of every window tile; its W3 operands stay in registers (L2-resident 32 KB)
    float4 bw[2][4];
    fc6_load_w3(W3, 2 * wv, lane, bw[0]);
    fc6_load_w3(W3, 2 * wv + 1, lane, bw[1]);
    const int cls = tid & 15, wl = tid >> 4;
    const float bv = b3[cls];
    for (int64_t base = (int64_t)blockIdx.x * TAIL_WINDOWS; base < n;
         base += (int64_t)gridDim.x * TAIL_WINDOWS) {
        __syncthreads();                                 // lg / part / h2s free again
        {   // the tile's 16 h2 rows -> LDS, coalesced, 8 loads in flight per thread
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e4 = tid + 256 * i;
                const int64_t row = base + e4 / (FC2 / 4);
                v[i] = reinterpret_cast<const float4*>(h2 + (row < n ? row : n - 1) * FC2)[e4 % (FC2 / 4)];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int e4 = tid + 256 * i;
                *reinterpret_cast<float4*>(h2s + (e4 / (FC2 / 4)) * TAIL_H2_LD + 4 * (e4 % (FC2 / 4))) = v[i];
            }
        }
        __syncthreads();
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {                 // chunk sums p_c of the 16 windows x 16 classes
            const int c = 2 * wv + cc;
            const fc6_f32x4 acc = fc6_chunk_mfma(h2s, TAIL_H2_LD, c * FC6_CHUNK, lane, bw[cc]);
#pragma unroll
            for (int r = 0; r < 4; ++r) part[(c * TAIL_WINDOWS + 4 * (lane >> 4) + r) * NCLS + (lane & 15)] = acc[r];
        }
        __syncthreads();
        const int64_t win = base + wl;
        float p[FC6_NCHUNK];
#pragma unroll
        for (int c = 0; c < FC6_NCHUNK; ++c) p[c] = part[(c * TAIL_WINDOWS + wl) * NCLS + cls];
        const float acc = fc6_combine(p, bv);
        lg[wl][cls] = acc;
        if (win < n && logits) logits[win * NCLS + cls] = acc;
        if (win < n && packed) *reinterpret_cast<float*>(packed + win * PACKED_ROW + 4 * cls) = acc;
        __syncthreads();
        if (tid < TAIL_WINDOWS && base + tid < n) {
            const int best = fc6_argmax16(lg[tid]);
            if (pred) pred[base + tid] = best;
            if (contacts) {
                uchar4 c;
                c.x = (best >> 3) & 1; c.y = (best >> 2) & 1; c.z = (best >> 1) & 1; c.w = best & 1;
                *reinterpret_cast<uchar4*>(contacts + (base + tid) * 4) = c;
            }
            if (packed) *reinterpret_cast<uchar4*>(packed + (base + tid) * PACKED_ROW + 4 * NCLS) = contact_bits(best);
        }
    }
    if (done_flag) {
        // online mode (one block, outputs in pinned host memory): publish completion to a host that
        // polls the flag instead of paying for a D2H copy and a stream synchronisation
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            if (seq_counter) done_seq = *seq_counter = *seq_counter + 1;    // graph launches: the count lives on the device
            __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------------
// The last step behind the fused fc.3 + fc.6-chunk GEMM epilogue (fc_gemm_phased.hip): add the 8 chunk sums
// of every (window, class) in the fixed order, + bias -> logits, argmax, contact bits.
//   part: [8 chunks][part_rows][16] floats
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void fc6_combine_kernel(const float* __restrict__ part, int64_t part_rows, const float* __restrict__ b3, int64_t n,
                        float* __restrict__ logits, int32_t* __restrict__ pred, uint8_t* __restrict__ contacts,
                        uint8_t* __restrict__ packed, Gate gate)
{
    __shared__ float lg[16][NCLS];
    if (gate_closed(gate)) return;                       // DCE_FP32_SPLIT's fallback sequence (dce_kernels.h Gate)
#if DCE_EXPERIMENTS
    if (gate.taken && blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(gate.taken, 1u);
#endif
    const int tid = threadIdx.x, cls = tid & 15, wl = tid >> 4;
    const int64_t base = (int64_t)blockIdx.x * 16, win = base + wl;
    const int64_t row = win < n ? win : n - 1;
    float p[FC6_NCHUNK];
#pragma unroll
    for (int c = 0; c < FC6_NCHUNK; ++c) p[c] = part[(c * part_rows + row) * NCLS + cls];
    const float v = fc6_combine(p, b3[cls]);
    lg[wl][cls] = v;
    if (win < n && logits) logits[win * NCLS + cls] = v;
    if (win < n && packed) *reinterpret_cast<float*>(packed + win * PACKED_ROW + 4 * cls) = v;
    __syncthreads();
    if (tid < 16 && base + tid < n) {
        const int best = fc6_argmax16(lg[tid]);
        if (pred) pred[base + tid] = best;
        if (contacts) {
            uchar4 c;
            c.x = (best >> 3) & 1; c.y = (best >> 2) & 1; c.z = (best >> 1) & 1; c.w = best & 1;
            *reinterpret_cast<uchar4*>(contacts + (base + tid) * 4) = c;
        }
        if (packed) *reinterpret_cast<uchar4*>(packed + (base + tid) * PACKED_ROW + 4 * NCLS) = contact_bits(best);
    }
}

hipError_t launch_fc6_combine(const float* part, int64_t part_rows, const float* b3, int64_t n,
                              float* logits, int32_t* pred, uint8_t* contacts, hipStream_t st, uint8_t* packed)
{
    if (n <= 0) return hipSuccess;
    plan_note("fc6_combine");
    hipLaunchKernelGGL(fc6_combine_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st,
                       part, part_rows, b3, n, logits, pred, contacts, packed, t_gate);
    return hipGetLastError();
}

hipError_t launch_fc3_tail(const float* h2, const float* W3, const float* b3, int64_t n,
                           float* logits, int32_t* pred, uint8_t* contacts, hipStream_t st,
                           unsigned* done_flag, unsigned done_seq, unsigned* seq_counter, uint8_t* packed)
{
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n + TAIL_WINDOWS - 1) / TAIL_WINDOWS;
    if (blocks > 1024) blocks = 1024;
    plan_note("fc3_tail");
    hipLaunchKernelGGL(fc3_tail_kernel, dim3((unsigned)blocks), dim3(256), TAIL_LDS_BYTES, st,
                       h2, W3, b3, n, logits, pred, contacts, blocks == 1 ? done_flag : nullptr, done_seq, seq_counter, packed, t_gate);
    return hipGetLastError();
}

// online mode: the new sample travels in the kernel arguments (no H2D copy) into its row of the
// device-resident sample buffer
__global__ __launch_bounds__(64)
void online_append_kernel(float* __restrict__ row, OnlineSample s)
{
    if (threadIdx.x < CH) row[threadIdx.x] = s.v[threadIdx.x];
}

hipError_t launch_online_append(float* row, const OnlineSample& s, hipStream_t st)
{
    hipLaunchKernelGGL(online_append_kernel, dim3(1), dim3(64), 0, st, row, s);
    return hipGetLastError();
}

// The graph form: the sample is read from pinned host memory, the cursor kept in device memory;
// when the buffer is full its last 149 rows move to the front first.
__global__ __launch_bounds__(256)
void online_append_state_kernel(float* __restrict__ ring, OnlineState* __restrict__ state,
                                const float* __restrict__ sample_host)
{
    const int tid = threadIdx.x;
    int cur = state->cursor;                             // (read by every thread before thread 0 updates it)
    __syncthreads();
    if (cur == ONLINE_ROWS) {
        for (int i = tid; i < (WIN - 1) * CH; i += 256) ring[i] = ring[(ONLINE_ROWS - (WIN - 1)) * CH + i];
        cur = WIN - 1;
        __syncthreads();
    }
    if (tid < CH) ring[cur * CH + tid] = sample_host[tid];
    if (tid == 0) { state->cursor = cur + 1; state->src_row = (long long)cur + 1 - WIN; }
}

hipError_t launch_online_append_state(float* ring, OnlineState* state, const float* sample_host, hipStream_t st)
{
    hipLaunchKernelGGL(online_append_state_kernel, dim3(1), dim3(256), 0, st, ring, state, sample_host);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// 16x16 class confusion counts C[gt][pred] -- the sufficient statistic of every metric the
// reference's src/test.py:19-70 prints (per-leg 2x2 confusion, FN/FP rates, precision, Jaccard,
// class / leg accuracy).  Integer histogram: LDS atomics per block, one global atomic per bin.
// ------------------------------------------------------------------------------------------
// HBM-bound integer work: 12 B per window (i32 prediction + i64 label).  VEC: four windows per thread
// per trip through 16-byte loads (one for the predictions, two for the labels), two trips in flight.
template <bool VEC>
__global__ __launch_bounds__(256)
void confusion16_kernel(const int32_t* __restrict__ pred, const int64_t* __restrict__ label,
                        int64_t n, unsigned long long* __restrict__ counts)
{
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    auto tally = [&](int p, int64_t g) {
        if (p >= 0 && p < 16 && g >= 0 && g < 16) atomicAdd(&h[(int)g * 16 + p], 1u);
    };
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, nthr = (int64_t)gridDim.x * 256;
    if (VEC) {
        const int64_t n4 = n / 4;
        const int4* p4 = reinterpret_cast<const int4*>(pred);
        const longlong2* g2 = reinterpret_cast<const longlong2*>(label);
        int64_t i = tid;
        for (; i + nthr < n4; i += 2 * nthr) {             // two independent trips per iteration
            const int4 pa = p4[i], pb = p4[i + nthr];
            const longlong2 ga0 = g2[2 * i], ga1 = g2[2 * i + 1];
            const longlong2 gb0 = g2[2 * (i + nthr)], gb1 = g2[2 * (i + nthr) + 1];
            tally(pa.x, ga0.x); tally(pa.y, ga0.y); tally(pa.z, ga1.x); tally(pa.w, ga1.y);
            tally(pb.x, gb0.x); tally(pb.y, gb0.y); tally(pb.z, gb1.x); tally(pb.w, gb1.y);
        }
        if (i < n4) {
            const int4 pa = p4[i];
            const longlong2 ga0 = g2[2 * i], ga1 = g2[2 * i + 1];
            tally(pa.x, ga0.x); tally(pa.y, ga0.y); tally(pa.z, ga1.x); tally(pa.w, ga1.y);
        }
        if (tid < (n & 3)) tally(pred[4 * n4 + tid], label[4 * n4 + tid]);
    } else {
        for (int64_t i = tid; i < n; i += nthr) tally(pred[i], label[i]);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

hipError_t launch_confusion16(const int32_t* pred, const int64_t* label, int64_t n,
                              unsigned long long* counts, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const bool vec = (reinterpret_cast<uintptr_t>(pred) % 16 == 0) && (reinterpret_cast<uintptr_t>(label) % 16 == 0);
    int64_t blocks = ((vec ? (n + 3) / 4 : n) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (vec) hipLaunchKernelGGL(confusion16_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, pred, label, n, counts);
    else     hipLaunchKernelGGL(confusion16_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, pred, label, n, counts);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// (n,68)-byte packed rows (the gather's wire format) -> the reference's three arrays.  HBM-bound byte work:
// 68 B in, 72 B out per window; one thread per (window, 4-byte word), the 17th word carries the contact bits.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void unpack_results_kernel(const uint8_t* __restrict__ packed, int64_t n, float* __restrict__ logits,
                           int32_t* __restrict__ pred, uint8_t* __restrict__ contacts)
{
    const int64_t words = n * (PACKED_ROW / 4);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / (PACKED_ROW / 4);
        const int w = (int)(i - row * (PACKED_ROW / 4));
        const unsigned v = reinterpret_cast<const unsigned*>(packed)[i];          // rows are 4-byte aligned: word i of the buffer
        if (w < NCLS) { if (logits) logits[row * NCLS + w] = __uint_as_float(v); }
        else {
            if (contacts) reinterpret_cast<unsigned*>(contacts)[row] = v;
            if (pred) pred[row] = (int)(((v & 1u) << 3) | (((v >> 8) & 1u) << 2) | (((v >> 16) & 1u) << 1) | ((v >> 24) & 1u));
        }
    }
}

hipError_t launch_unpack_results(const uint8_t* packed, int64_t n, float* logits, int32_t* pred, uint8_t* contacts, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    int64_t blocks = (n * (PACKED_ROW / 4) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(unpack_results_kernel, dim3((unsigned)blocks), dim3(256), 0, st, packed, n, logits, pred, contacts);
    return hipGetLastError();
}

}  // namespace dce
