// fc_gemm_phased.hip -- the FC-layer GEMMs at chip-filling sizes on gfx950, fp32 and bf16:
//     C[M,N] = act(A[M,K] W[N,K]^T + bias),  A, W K-contiguous (PyTorch's [out][in]),  fp32 accumulate
// (reference src/contact_cnn.py:48-54: fc.0 + ReLU, fc.3 + ReLU).
//   fp32: v_mfma_f32_32x32x2_f32, K walked exactly as fc_gemm.hip / fc_gemv.hip walk it (inside every 8
//         consecutive k: 0,4,1,5,2,6,3,7) -> the same bits as those kernels;
//   bf16: v_mfma_f32_32x32x16_bf16 (DCE_BF16_FC, BASELINE configs[4]).
//
// Structure: ONE workgroup per CU, 8 waves = two groups of four.  Waves w and w+4 share a SIMD and sit
// in different groups; the groups run ONE PHASE APART:
//     phase p   : group 0  math(tile t)   | group 1  load(tile t)      -- workgroup barrier --
//     phase p+1 : group 0  load(tile t+1) | group 1  math(tile t)      -- workgroup barrier --
// "load" = read this wave's fragments of one K-tile from LDS into registers and issue its share of the
// global->LDS loads of tile t+2; "math" = nothing but the tile's MFMAs.  Every SIMD's matrix pipe always has
// one wave in its math phase, and a barrier costs it one hand-over instead of a stall of the whole block
// (the tile kernel of fc_gemm.hip leaves two unrelated blocks per CU to cover each other's barriers by
// chance: 0.87 of the fp32 peak on fc.0; this one is built so that the cover is there by construction).
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KB per wave-instruction): no staging
//     VGPRs, no ds_write pass; three LDS buffers, loads run two K-tiles ahead of the math;
//   * bf16 needs the large tile for another reason: at bf16 rate a 128x128 tile moves 64 B/clk/CU out of
//     L2 and through ds_write_b128 (79 B/clk/CU); 256x128 per CU asks for 47 B/clk.
// (s_setprio(1) around the math phase measured +-0 on both precisions, r2 A/B: the load-phase wave's handful of
// instructions never starves the math wave.)
// LDS image: a K-tile is (BM + BN) rows x 128 or 256 B of K; LDS-DMA writes lane-linear (1 KB = 8 or 4 rows
// per wave-instruction), so the bank swizzle lives in the per-lane GLOBAL address: 16-byte column c of row r
// is stored at slot c ^ swz(r) (PhCfg::swz); the fragment reads (lane = row, fixed logical column) apply the
// same XOR and are conflict-free for ds_read_b128's 16-lane groups.
#include "dce_kernels.h"
#include "fc6_chain.h"
#include "fc_tree.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

#ifndef PH_TRACE
#define PH_TRACE 0
#endif
#if defined(PH_EXP) && !DCE_EXPERIMENTS
#error "PH_EXP timing probes give wrong results: experiments build only"
#endif
#ifndef PH_FOLD_MFMA
#define PH_FOLD_MFMA 0       // 1: the summation tree's fold clears the accumulators with an MFMA of zeros instead of 16 v_mov per tile
                             // (measured: fc.0 533.1 vs 532.1 us, no gain -- the fold's cost is the 32 v_pk_add_f32, which wait for
                             // breaks in the partner wave's MFMA stream and delay the phase barrier: ~1 us per fold, three per launch;
                             // with the tree's cuts switched off altogether (-DFC_TREE_OFF, wrong results) fc.0 528.8 / fc.3 65.8 us)
#endif


namespace dce {

#if PH_TRACE
// debug build (-DPH_TRACE=1): s_memtime at four points of every phase pair, first 64 K-tiles, every wave of block 0
__device__ unsigned long long g_ph_trace[8 * 64 * 4];
#define PH_MARK(k) do { if (blockIdx.x == 0 && (tid & 63) == 0 && t < 64) g_ph_trace[(wid * 64 + t) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define PH_MARK(k) do {} while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int PH_NBUF = 3;

// wave tile (32 TM) x (32 TN), waves 4 (M) x 2 (N); ROWB = bytes of K per row per K-tile (128 or 256)
template <int TM, int TN, int ROWB> struct PhCfg {
    static constexpr int BM = 4 * 32 * TM, BN = 2 * 32 * TN, ROWS = BM + BN;
    static constexpr int TILE = ROWS * ROWB, LDS = PH_NBUF * TILE;
    static constexpr int SLOTS = ROWB / 16, CROWS = 64 / SLOTS;    // 16-byte slots per row; rows per 1 KB chunk
    static constexpr int NA = BM / CROWS / 8, NW = BN / CROWS / 8; // 1 KB chunks per wave per K-tile: of A, of W
    static constexpr int KQ = ROWB / 32;                           // 32-byte column pairs (fragment loads per row per tile)
    static_assert(ROWB == 64 || ROWB == 128 || ROWB == 256, "");
    static_assert((NA == 4 && NW == 2) || (NA == 2 && NW == 1), "issue_tile is written for 4+2 and 2+1 chunks");
    static_assert(LDS <= 160 * 1024, "three K-tiles in LDS");
    // bank swizzle: 16-byte column c of row r is stored at slot c ^ swz(r); conflict-free for ds_read_b128's
    // lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} (lane = row): 8 slots need (r>>1)&7, 16 slots r&15
    // (4 slots -- 64-byte rows, the 32-k K-tiles of round 5's two-workgroups-per-CU experiment: (r >> 2) & 3, as fc_gemm_x3.hip found for its 64-byte rows)
    __device__ static constexpr int swz(int r) { return SLOTS == 4 ? (r >> 2) & 3 : SLOTS == 8 ? (r >> 1) & 7 : r & 15; }
};

__device__ __forceinline__ unsigned short f32_to_bf16(float f)
{   // round-to-nearest-even; NaN stays NaN (quiet)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// The 1 KB chunks this wave brings in per K-tile: chunk j lands at lds0 + j*8 KB (+ lane*16); the first NA
// come from the A panel (wave-uniform base sA), the rest from the W panel (sW); v[j] = per-lane byte offsets.
// M0 carries the LDS destination; it is compiler-reserved, so it is saved and restored inside the statement.
#define PH_GLDS(vreg, sreg) "global_load_lds_dwordx4 " vreg ", " sreg "\n\t"
#define PH_NEXT "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
template <int NA, int NW>
__device__ __forceinline__ void issue_tile(unsigned lds0, const char* sA, const char* sW, const unsigned (&v)[NA + NW])
{
    unsigned keep;
    if constexpr (NA == 4) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     PH_GLDS("%4", "%2") PH_NEXT PH_GLDS("%5", "%2") PH_NEXT PH_GLDS("%6", "%2") PH_NEXT PH_GLDS("%7", "%2") PH_NEXT
                     PH_GLDS("%8", "%3") PH_NEXT PH_GLDS("%9", "%3")
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(lds0), "s"(sA), "s"(sW), "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5])
                     : "memory");
    } else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     PH_GLDS("%4", "%2") PH_NEXT PH_GLDS("%5", "%2") PH_NEXT PH_GLDS("%6", "%3")
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(lds0), "s"(sA), "s"(sW), "v"(v[0]), "v"(v[1]), "v"(v[2])
                     : "memory");
    }
}
#undef PH_GLDS
#undef PH_NEXT

// End of a phase: this wave's LDS-DMA of all but the newest tile has landed (each wave issues NG pieces per
// tile, so "at most NG outstanding" = everything older is in LDS), its own fragment reads have returned (so
// the buffer they came from may be refilled after the barrier), then the workgroup barrier.
template <int NG> __device__ __forceinline__ void phase_end(bool more)
{
    static_assert(NG == 6 || NG == 3, "");
    if (more) {
        if constexpr (NG == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else                   asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// 32-bit LDS byte address of a __shared__ object (what M0 takes for LDS-DMA)
__device__ __forceinline__ unsigned lds_addr(const void* p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}

}  // namespace

// FUSE6 (fp32, 128 x 64 tile, N = 512: fc.3): the block's 64 output columns are exactly one chunk of fc.6's
// summation tree (fc6_chain.h), so the epilogue finishes that chunk -- h2 tile -> LDS -> 16 MFMAs per 16 rows
// -> chunk sums to `part` ([8][part_rows][16]) -- and h2 itself goes to HBM only when Cv != NULL (taps).
template <bool BF16, bool OUT_BF16, int TM, int TN, int ROWB, bool FUSE6, bool LOCKSTEP>
__global__ __launch_bounds__(512, ROWB == 64 ? 4 : 2)       // (64-byte K-tiles: 3 x 24 KB of LDS and <= 128 VGPRs -> TWO workgroups per CU)
void fc_gemm_phased_kernel(const void* __restrict__ Av, const void* __restrict__ Wv,
                           const float* __restrict__ bias, void* __restrict__ Cv,
                           int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2,
                           const float* __restrict__ W3 = nullptr, float* __restrict__ part = nullptr, long long part_rows = 0,
                           Gate gate = Gate{})
{
    static_assert(!FUSE6 || (!OUT_BF16 && TM == 1 && TN == 1), "the fused fc.6 epilogue is fc.3's 128x64 tile with fp32 output");
    if (gate_closed(gate)) return;                       // DCE_FP32_SPLIT's fallback sequence (dce_kernels.h Gate)
    using Cfg = PhCfg<TM, TN, ROWB>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, NG = Cfg::NA + Cfg::NW, KQ = Cfg::KQ;
    constexpr int ES = BF16 ? 2 : 4;                     // operand element size
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- XCD-aware tile assignment (speed only): the 32 blocks co-resident on one XCD form an sm x sn super-tile
    const int bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const int sid = (li >> 5) * 8 + xcd;
    const int within = li & 31;
    const int sn = 1 << sn_log2, sm = 32 >> sn_log2;
    const int nsn = ntiles >> sn_log2;
    const int tm = (sid / nsn) * sm + (within >> sn_log2);
    const int tn = (sid % nsn) * sn + (within & (sn - 1));
    if (tm >= mtiles) return;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;                            // phase group; waves w and w+4 share a SIMD
    const int wm = (wid & 3) * 32 * TM, wn = grp * 32 * TN;   // this wave's corner of the block tile
    const int i = lane & 31, h = lane >> 5;

    // ---- global -> LDS: chunk c = wid + 8 j of the stacked tile; chunks [0, BM/8) are A rows, then W rows
    const size_t rowb = (size_t)K * ES;
    unsigned voff[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int c = wid + 8 * j;
        const int r = Cfg::CROWS * c + lane / Cfg::SLOTS;   // row of the stacked tile
        const int slot = lane % Cfg::SLOTS;              // 16-byte slot this lane fills
        const int col = slot ^ Cfg::swz(r);              // logical 16-byte column that lives there
        int grow = j < Cfg::NA ? r : r - BM;             // row inside the A / W panel
        if (j < Cfg::NA && m0 + grow >= M) grow = M - 1 - m0;   // rows past M re-read the last one (never stored)
        voff[j] = (unsigned)(grow * rowb + 16 * col);
    }
#if defined(PH_EXP) && (PH_EXP & 1)
    // timing probe (wrong results; needs DCE_EXPERIMENTS): every workgroup reads the SAME A and W rows -- 3.6 MB, L2-resident after
    // the first launch -- so that the operand loads see L2-hit latency instead of the fabric's
    const char* sA = static_cast<const char*>(Av);
    const char* sW = static_cast<const char*>(Wv);
#else
    const char* sA = static_cast<const char*>(Av) + (size_t)m0 * rowb;
    const char* sW = static_cast<const char*>(Wv) + (size_t)n0 * rowb;
#endif
    const unsigned lds_wave = lds_addr(smem) + wid * 1024;     // chunk `wid` of buffer 0

    // ---- fragment reads: lane (i, h) reads row (wave corner + 32 a + i), logical 16-byte column 2 kq + h
    //      fp32: k = 8 kq + 4 h + (0..3) -> four K=2 MFMAs;   bf16: k = 16 kq + 8 h + (0..7) -> one K=16 MFMA
    const int sw = Cfg::swz(i);                          // wave corners are multiples of 32 rows: swz(row) = swz(i)
    int fo[KQ];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) fo[kq] = 16 * ((2 * kq + h) ^ sw);
    const int arow = (wm + i) * ROWB, brow = (BM + wn + i) * ROWB;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fp32: the fixed summation tree of fc_tree.h -- `tot` collects the finished K ranges, acc the running one; a cut
    // costs a wave TM*TN*16 adds + moves ahead of one tile's MFMAs, three times per tile of C
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
            // tot += acc (8 v_pk_add_f32 per 32x32 tile), acc = 0.  PH_FOLD_MFMA=1 clears on the matrix pipe instead (an MFMA
            // of zeros with the literal 0 as C; the asm is opaque to hipcc's hazard recogniser, hence the spelled-out wait states).
            const float z = 0.f;
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    tot[a][b] += acc[a][b];
#if PH_FOLD_MFMA
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %1, 0" : "=&v"(acc[a][b]) : "v"(z));
#else
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[a][b][r] = z;
#endif
                }
#if PH_FOLD_MFMA
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#endif
        }
    };
    const FcTree tree = fc_tree(BF16 ? 1 : K, ROWB / 4);

    float4 bw6[4];                                       // FUSE6: this lane's W3 operands of the block's chunk (= column tile tn)
    if constexpr (FUSE6) fc6_load_w3(W3, tn, lane, bw6);

    const int KT = (int)(rowb / ROWB);                   // >= 4 (checked by the launcher)

    // fragments of one K-tile: KQ x (TM + TN) x 16 bytes per lane
    // LDS byte offsets of this lane's fragment columns, one set per buffer, made opaque so that they STAY in
    // registers: recomputed inside the loop they are VALU instructions of the load phase, and a VALU instruction
    // of the wave that shares a SIMD with a wave streaming MFMAs waits for that wave's next MFMA to issue
    // (64-byte K-tiles run under a 128-register budget: one address per kq, the buffer -- 24 KB apart, inside ds_read_b128's 16-bit
    //  offset field -- folded into the instruction as an immediate)
    constexpr int PAB = ROWB == 64 ? 1 : PH_NBUF;
    static_assert(ROWB != 64 || 2 * Cfg::TILE + 32 * ROWB * (TM > TN ? TM : TN) < 65536, "buffer + block offsets fit the LDS instructions' immediate");
    unsigned pa[PAB][KQ], pb[PAB][KQ];
#pragma unroll
    for (int b = 0; b < PAB; ++b)
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            pa[b][kq] = b * Cfg::TILE + arow + fo[kq];
            pb[b][kq] = b * Cfg::TILE + brow + fo[kq];
            asm volatile("" : "+v"(pa[b][kq]), "+v"(pb[b][kq]));
        }
    // fragments of one K-tile: KQ x (TM + TN) x 16 bytes per lane; `b` must be a compile-time constant at every call
    auto load_frags = [&](int b, float4 (&af)[KQ][TM], float4 (&bf)[KQ][TN]) {
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[kq][a] = *reinterpret_cast<const float4*>(smem + pa[PAB == 1 ? 0 : b][kq] + (PAB == 1 ? b * Cfg::TILE : 0) + a * 32 * ROWB);
#pragma unroll
            for (int c = 0; c < TN; ++c) bf[kq][c] = *reinterpret_cast<const float4*>(smem + pb[PAB == 1 ? 0 : b][kq] + (PAB == 1 ? b * Cfg::TILE : 0) + c * 32 * ROWB);
        }
    };
    auto math = [&](const float4 (&af)[KQ][TM], const float4 (&bf)[KQ][TN]) {
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            if constexpr (BF16) {
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(bf16x8, af[kq][a]), __builtin_bit_cast(bf16x8, bf[kq][b]), acc[a][b], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b) {
                            const float av = u == 0 ? af[kq][a].x : u == 1 ? af[kq][a].y : u == 2 ? af[kq][a].z : af[kq][a].w;
                            const float bv = u == 0 ? bf[kq][b].x : u == 1 ? bf[kq][b].y : u == 2 ? bf[kq][b].z : bf[kq][b].w;
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[a][b], 0, 0, 0);
                        }
            }
        }
    };

    if constexpr (LOCKSTEP) {
        // ---- the lockstep schedule: all eight waves in step, ONE barrier per K-tile, nothing but MFMAs between two
        // barriers on the critical path: each wave keeps TWO register sets of fragments -- while it multiplies tile t
        // out of one set it reads tile t+1's fragments from LDS into the other and issues its share of the global->LDS
        // loads of tile t+3.  The two waves of a SIMD share the matrix pipe during the whole tile.
        //   buffer of tile t+3 = buffer of tile t: its fragments were read during tile t-1 and returned (lgkmcnt(0))
        //   before the barrier that ended tile t-1.  Tile t+1 is read during tile t: every wave waited for its pieces
        //   of t+1 ("vmcnt(NG)": all but the newest tile landed; the newest then was t+2) before the barrier that ended
        //   tile t-1.
        float4 afA[KQ][TM], bfA[KQ][TN], afB[KQ][TM], bfB[KQ][TN];
        issue_tile<Cfg::NA, Cfg::NW>(lds_wave, sA, sW, voff);
        issue_tile<Cfg::NA, Cfg::NW>(lds_wave + Cfg::TILE, sA + ROWB, sW + ROWB, voff);
        issue_tile<Cfg::NA, Cfg::NW>(lds_wave + 2 * Cfg::TILE, sA + 2 * ROWB, sW + 2 * ROWB, voff);
        if constexpr (NG == 6) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");    // tile 0 landed for everybody
        else                   asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        load_frags(0, afA, bfA);
        phase_end<NG>(true);                             // tile 1 landed; tile 0's fragments are in registers
        auto step = [&](int t, const float4 (&afC)[KQ][TM], const float4 (&bfC)[KQ][TN], float4 (&afN)[KQ][TM], float4 (&bfN)[KQ][TN]) {
            const bool more = t + 3 < KT;
            if (more) {
                const size_t ko = (size_t)(t + 3) * ROWB;
                issue_tile<Cfg::NA, Cfg::NW>(lds_wave + (t % 3) * Cfg::TILE, sA + ko, sW + ko, voff);
            }
            if (t + 1 < KT) { const int nb = (t + 1) % 3; if (nb == 0) load_frags(0, afN, bfN); else if (nb == 1) load_frags(1, afN, bfN); else load_frags(2, afN, bfN); }
            __builtin_amdgcn_sched_barrier(0);           // the LDS reads go out ahead of the MFMAs, not behind them
            if (!BF16 && fc_tree_cut(tree, t)) fold();
            math(afC, bfC);
            phase_end<NG>(more);
        };
        int t = 0;
        for (; t + 2 <= KT; t += 2) {
            step(t, afA, bfA, afB, bfB);
            step(t + 1, afB, bfB, afA, bfA);
        }
        if (t < KT) step(t, afA, bfA, afB, bfB);
    } else {
    // prologue: tiles 0 and 1 in flight; tile 0 landed for everybody before the first phase
    issue_tile<Cfg::NA, Cfg::NW>(lds_wave, sA, sW, voff);
    issue_tile<Cfg::NA, Cfg::NW>(lds_wave + Cfg::TILE, sA + ROWB, sW + ROWB, voff);
    phase_end<NG>(true);
    if (grp == 1) phase_end<NG>(true);                   // group 1 runs one phase behind group 0

    // Ordering of LDS-DMA against fragment reads.  Number the phases p = 0, 1, 2, ...; a barrier ends each.
    //   group 0: load(t) in phase 2t,   math(t) in phase 2t+1        group 1: load(t) in 2t+1, math(t) in 2t+2
    // Tile t+2 goes to buffer (t+2) % 3 = the buffer of tile t-1.  Each wave issues its six pieces of tile t+2 at the
    // start of ITS load(t): group 0 in phase 2t, group 1 in phase 2t+1.
    //   WAR: the last reader of tile t-1 is group 1's load(t-1), phase 2t-1; it ends with lgkmcnt(0) + barrier, so
    //        every fragment read of that buffer has RETURNED before phase 2t starts.
    //   RAW: the first reader of tile t+2 is group 0's load(t+2), phase 2t+4.  A wave's "vmcnt(6)" at a phase end
    //        means all but its six newest pieces have landed: group 0's pieces of t+2 are retired at the end of phase
    //        2t+2 (it issued t+3 at the start of that phase), group 1's at the end of phase 2t+3 -- both followed by
    //        a barrier before phase 2t+4.  Once nothing is left to issue (t+2 >= KT) the waits become vmcnt(0).
    // (the loop is unrolled by three so that the buffer of every tile is a compile-time constant)
    auto ktile = [&](int t, auto bufc) {
        constexpr int buf = decltype(bufc)::value, nbuf = (buf + 2) % 3;     // buffer of tile t / of tile t+2
        // ---- load phase: loads of tile t+2, fragments of tile t
        const bool more = t + 2 < KT;
        if (more) {
            const size_t ko = (size_t)(t + 2) * ROWB;
            issue_tile<Cfg::NA, Cfg::NW>(lds_wave + nbuf * Cfg::TILE, sA + ko, sW + ko, voff);
        }
        float4 af[KQ][TM], bf[KQ][TN];
        PH_MARK(0);
        load_frags(buf, af, bf);
        PH_MARK(1);
        phase_end<NG>(more);
        // ---- math phase
        PH_MARK(2);
        math(af, bf);
        PH_MARK(3);
        phase_end<NG>(more);
    };
    // The summation tree's cuts fall on whole rounds of this loop (fc_tree.h: multiples of three K-tiles), so the K walk
    // is four plain loops with one accumulator fold between them -- no per-tile test between a barrier and the MFMAs.
    const int seg_end[FC_RANGES] = {(!BF16 && tree.b1 > 0 && tree.b1 % 3 == 0) ? tree.b1 : 0, (!BF16 && tree.b2 > 0 && tree.b2 % 3 == 0) ? tree.b2 : 0,
                                    (!BF16 && tree.b3 > 0 && tree.b3 % 3 == 0) ? tree.b3 : 0, KT};
    int t = 0;
#pragma unroll 1
    for (int seg = 0; seg < FC_RANGES; ++seg) {
        const int te = seg_end[seg];
        if (seg + 1 < FC_RANGES && te <= t) continue;    // no cut here (bf16, or a shape without the tree)
#pragma unroll 1
        for (; t < te; t += 3) {
            ktile(t, std::integral_constant<int, 0>{});
            if (t + 1 < KT) ktile(t + 1, std::integral_constant<int, 1>{});
            if (t + 2 < KT) ktile(t + 2, std::integral_constant<int, 2>{});
        }
        if (seg + 1 < FC_RANGES) fold();
    }
    // (Measured alternatives, all slower than this fold behind the range's last barrier -- fc.0 530 us, fc.3 67 us: a run-time
    //  test for the cut in every tile, before its MFMAs 540 / 75 us, after them 535 / 78 us; the range's last round peeled with the
    //  fold compiled in behind its third tile's MFMAs -- where the partner wave issues no MFMA -- 570 / 76 us: twice the loop code.)
    if (grp == 0) phase_end<NG>(false);                  // same number of barriers for both groups
    }

    if constexpr (FUSE6) {
        // ---- fused epilogue.  Every wave is past its last fragment read (the barrier above), LDS is free.
        constexpr int HLD = 68;                          // h2 tile [128][68] floats
        float* ht = reinterpret_cast<float*>(smem);
        const int col_l = wn + i;
        const float bv = bias[n0 + col_l];
        float* C = static_cast<float*>(Cv);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row_l = wm + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = (BF16 ? acc[0][0][r] : tot[0][0][r] + acc[0][0][r]) + bv;
            v = v < 0.f ? 0.f : v;                       // fc.3's ReLU; keeps NaN like torch
            ht[row_l * HLD + col_l] = v;
            if (C && m0 + row_l < M) C[(size_t)(m0 + row_l) * N + n0 + col_l] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const fc6_f32x4 p = fc6_chunk_mfma(ht + 16 * wid * HLD, HLD, 0, lane, bw6);   // wave w: rows 16w .. 16w+15
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * wid + 4 * (lane >> 4) + r;
            if (row < M) part[((size_t)tn * part_rows + row) * NCLS + (lane & 15)] = p[r];
        }
        return;
    }

    // ---- epilogue: bias + (ReLU); D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    auto store_tile = [&](auto full) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int col = n0 + wn + 32 * b + i;
            const float bv = bias[col];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float v = (BF16 ? acc[a][b][r] : tot[a][b][r] + acc[a][b][r]) + bv;
                    if (relu) v = v < 0.f ? 0.f : v;              // keeps NaN like torch
                    if (decltype(full)::value || row < M) {
                        if constexpr (OUT_BF16) static_cast<unsigned short*>(Cv)[(size_t)row * N + col] = f32_to_bf16(v);
                        else static_cast<float*>(Cv)[(size_t)row * N + col] = v;
                    }
                }
        }
    };
    if (m0 + BM <= M) store_tile(std::true_type{});      // whole tile in range: no per-store predicate
    else store_tile(std::false_type{});
}


#if DCE_EXPERIMENTS
// ------------------------------------------------------------------------------------------------------------------
// fc_gemm_ki_kernel -- the bf16 256 x 128 tile with the two wave groups dealing the K-TILES out between them (round 4).
// In the phased kernel above both groups multiply the SAME K-tile (each its half of the tile's columns): a math phase is 16 MFMAs
// per wave = 512 cycles at bf16 rate, and the hand-over at each of the two phase changes per K-tile costs as much again
// (profiles/r3c_trace_gemm.txt: 1024 cycles of MFMAs in a 1850-cycle period).  Here group 0 owns the even K-tiles and group 1 the odd
// ones, each for the WHOLE 256 x 128 tile (wave tile 64 x 128: 32 MFMAs = 1024 cycles per math phase), still one phase apart:
//     phase 2k   : group 0  load(tile 2k)   | group 1  math(tile 2k-1)     -- workgroup barrier --
//     phase 2k+1 : group 0  math(tile 2k)   | group 1  load(tile 2k+1)     -- workgroup barrier --
// so the same two hand-overs now stand beside twice the MFMAs.  The group that loads tile t also issues the whole LDS-DMA of tile
// t+2 (12 pieces per wave) into the buffer of tile t-1, which the OTHER group finished reading a phase ago; it waits for those pieces
// at the end of its own math(t).  The two groups' partial sums over their halves of K meet once, after the loop: each wave hands
// two of its four column blocks to its partner wave through LDS (the tile buffers are free by then) and finishes the other two.
// Another summation order than the phased kernel's: bf16 precision only, which claims no bit pattern.
// MEASURED (profiles/r4o_gemm_ki.txt): the loop is what it was built to be -- 3045 cycles per pair of K-tiles against 3700 (a load phase
// is now ~1000 cycles: 24 fragment reads and 12 LDS-DMA instructions issued beside the other group's MFMA stream, the CU's address
// path alone takes 768 cycles for a tile's 48 KB) -- and the launch takes 71 us against 69: ~128k cycles at 1.80 GHz against ~150k at
// 2.17 GHz.  What the denser loop gains the board's power management takes back in clock, as with fc_gemm_x3.hip and with the conv
// stack at 32768 windows: this GEMM sits at ~1.15 PFLOP/s whichever way its phases are cut.  Experiments build only (DCE_GEMM_KI=1).
template <bool OUT_BF16>
__global__ __launch_bounds__(512, 2)
void fc_gemm_ki_kernel(const void* __restrict__ Av, const void* __restrict__ Wv, const float* __restrict__ bias, void* __restrict__ Cv,
                       int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2)
{
    using Cfg = PhCfg<2, 2, 128>;
    constexpr int BM = Cfg::BM, KQ = Cfg::KQ, TM = 2, TN = 4, ROWB = 128, TILE = Cfg::TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const int sid = (li >> 5) * 8 + xcd;
    const int within = li & 31;
    const int sn = 1 << sn_log2, sm = 32 >> sn_log2;
    const int nsn = ntiles >> sn_log2;
    const int tm = (sid / nsn) * sm + (within >> sn_log2);
    const int tn = (sid % nsn) * sn + (within & (sn - 1));
    if (tm >= mtiles) return;
    const int m0 = tm * BM, n0 = tn * Cfg::BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2, w4 = wid & 3;
    const int wm = w4 * 64;
    const int i = lane & 31, h = lane >> 5;

    // global -> LDS: the loading group's wave w4 brings chunks w4 + 4 m (m = 0..11) of the tile's 48: two issue_tile calls, chunks
    // w4 + 8 j and w4 + 4 + 8 j -- 32 rows further down both panels, same slots (swz(r + 32) = swz(r)): one set of per-lane offsets
    const size_t rowb = (size_t)K * 2;
    unsigned voff[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int c = w4 + 8 * j;
        const int r = Cfg::CROWS * c + lane / Cfg::SLOTS;
        const int col = (lane % Cfg::SLOTS) ^ Cfg::swz(r);
        const int grow = j < Cfg::NA ? r : r - BM;       // (whole tiles only: the launcher checks M % 256 == 0)
        voff[j] = (unsigned)(grow * rowb + 16 * col);
    }
    const char* sA = static_cast<const char*>(Av) + (size_t)m0 * rowb;
    const char* sW = static_cast<const char*>(Wv) + (size_t)n0 * rowb;
    const unsigned lds_wave = lds_addr(smem) + w4 * 1024;
    auto issue = [&](int t) {
        const size_t ko = (size_t)t * ROWB;
        const unsigned dst = lds_wave + (unsigned)(t % 3) * TILE;
        issue_tile<4, 2>(dst, sA + ko, sW + ko, voff);
        issue_tile<4, 2>(dst + 4 * 1024, sA + ko + 32 * rowb, sW + ko + 32 * rowb, voff);
    };

    const int sw = Cfg::swz(i);
    unsigned pa[KQ];                                      // A rows of this wave; the W rows sit (BM - wm) rows further down: an immediate
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
        pa[kq] = (wm + i) * ROWB + 16 * ((2 * kq + h) ^ sw);
        asm volatile("" : "+v"(pa[kq]));
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int KT = (int)(rowb / ROWB);                    // >= 4, even or odd
    float4 af[KQ][TM], bf[KQ][TN];
    auto load_frags = [&](unsigned boff) {               // boff: byte offset of the tile's buffer (wave-uniform, run time: both groups run this one loop)
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const char* base = smem + (boff + pa[kq]);
#pragma unroll
            for (int a = 0; a < TM; ++a) af[kq][a] = *reinterpret_cast<const float4*>(base + a * 32 * ROWB);
#pragma unroll
            for (int c = 0; c < TN; ++c) bf[kq][c] = *reinterpret_cast<const float4*>(base + (BM - wm + c * 32) * ROWB);
        }
    };
    auto math = [&]() {
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kq][a]), __builtin_bit_cast(bf16x8, bf[kq][b]), acc[a][b], 0, 0, 0);
    };
    auto bar = [&](bool drain) {                         // end of a phase: own fragment reads returned (and, behind a math phase, own LDS-DMA landed)
        __builtin_amdgcn_sched_barrier(0);               // (MFMAs carry no memory dependence: without this the scheduler moves them across the barrier)
        if (drain) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else       asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    // prologue: each group brings its first tile; everybody waits for both
    { const int t = 62; PH_MARK(0); }
    if (grp < KT) issue(grp);
    bar(true);
    if (grp == 1) bar(false);                             // group 1 runs one phase behind group 0
    int buf = grp;                                        // tile t lives in buffer t % 3
#pragma unroll 1
    for (int t = grp; t < KT; t += 2) {
        PH_MARK(0);
        if (t + 2 < KT) issue(t + 2);                    // into the buffer of tile t-1: the other group read it a phase ago
        load_frags((unsigned)buf * TILE);
        PH_MARK(1);
        bar(false);
        PH_MARK(2);
        math();
        PH_MARK(3);
        bar(true);
        buf = buf == 0 ? 2 : buf - 1;                    // (buf + 2) % 3
    }
    // the group that ran out of tiles first keeps in step with the other's remaining phases: both execute KT + 1 phases' barriers
    {
        const int mine = (KT - grp + 1) / 2;             // tiles this group processed: 2 barriers each (+ 1 for group 1's delay)
        const int done = 2 * mine + grp, total = KT + 1;
        for (int p = done; p < total; ++p) bar(false);
    }

    // ---- the two halves of K meet: group 0 finishes column blocks 0, 1 and group 1 blocks 2, 3; each hands the other two to its
    //      partner wave (same w4) through LDS: [w4][to group][a][b'][quarter][lane] x 16 bytes = 128 KB (the tile buffers are free)
    {
        { const int t = 62; PH_MARK(1); }
        float4* ex = reinterpret_cast<float4*>(smem);
        const bool g0 = grp == 0;                        // (selects, not a run-time index into acc: that would put the accumulators in scratch)
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v;
                    v.x = g0 ? acc[a][2 + b][4 * q] : acc[a][b][4 * q];
                    v.y = g0 ? acc[a][2 + b][4 * q + 1] : acc[a][b][4 * q + 1];
                    v.z = g0 ? acc[a][2 + b][4 * q + 2] : acc[a][b][4 * q + 2];
                    v.w = g0 ? acc[a][2 + b][4 * q + 3] : acc[a][b][4 * q + 3];
                    ex[((((w4 * 2 + (1 - grp)) * TM + a) * 2 + b) * 4 + q) * 64 + lane] = v;
                }
        bar(false);
        { const int t = 62; PH_MARK(2); }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + 32 * ((g0 ? 0 : 2) + b) + i;
            const float bv = bias[col];
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 o = ex[((((w4 * 2 + grp) * TM + a) * 2 + b) * 4 + q) * 64 + lane];
                    const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * q + e;
                        const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float v = ((g0 ? acc[a][b][r] : acc[a][2 + b][r]) + ov[e]) + bv;
                        if (relu) v = v < 0.f ? 0.f : v;
                        if constexpr (OUT_BF16) static_cast<unsigned short*>(Cv)[(size_t)row * N + col] = f32_to_bf16(v);
                        else static_cast<float*>(Cv)[(size_t)row * N + col] = v;
                    }
                }
        }
        { const int t = 62; PH_MARK(3); }
    }
}

#endif  // DCE_EXPERIMENTS (fc_gemm_ki_kernel)

#if DCE_EXPERIMENTS
// ------------------------------------------------------------------------------------------------------------------
// fc_gemm_pipe_kernel -- the bf16 256 x 128 tile WITHOUT workgroup barriers in its K loop (round 4; DCE_GEMM=pipe).
// The phased schedule above hands the matrix pipe from one wave group to the other twice per K-tile; at fp32 rate that hand-over is
// 6 % of a phase, at bf16 rate (a 64-k K-tile = 512 cycles of MFMAs per wave) it is a third (profiles/r3c_trace_gemm.txt), and the
// kernel does not wait for memory (profiles/r4i_gemm_probes.txt).  Here all eight waves run the same loop over 32-k K-tiles (64-byte
// rows, 24 KB per tile, SIX LDS buffers), each wave with two register sets of fragments (tile u multiplies while tile u+1 is read),
// and what the barriers guaranteed is tracked per LDS buffer by two counters in LDS that only ever count up:
//     landed[b] : waves whose share of the tile now in buffer b has arrived (s_waitcnt vmcnt, then ds_add)    -> RAW, before reading
//     freed[b]  : waves whose fragment reads of the tile in buffer b have returned (lgkmcnt(0), then ds_add)  -> WAR, before refilling
// A wave that finds a counter short spins on it (ds_read + s_sleep) with half of the tile's MFMAs already issued and with the other
// wave of its SIMD free to go on: nobody waits for the slowest wave at a fixed point of every tile.  Step u of a wave (fragments of
// tile u in `cur`, requested during step u-1):
//     S0  lgkmcnt(0): cur has arrived (= this wave is through with buffer u % 6)          -> freed[u % 6] += 1
//     S1  the four MFMAs of k-quarter 0
//     S2  wait freed[(u - LAG) % 6] (posted LAG + 1/2 tiles ago)  -> LDS-DMA of tile u + 6 - LAG into that buffer
//         vmcnt: own share of tile u+2 has landed (issued 4 - LAG tiles ago)               -> landed[(u+2) % 6] += 1
//         wait landed[(u+1) % 6] (posted one tile ago)
//     S3  the four MFMAs of k-quarter 1, the eight fragment reads of tile u+1 dealt out between them
// MEASURED (profiles/r4j_gemm_pipe.txt): correct and bit-equal to the phased kernel only with the own-share wait one tile more
// conservative than the instruction count says (with "all but the 4 newest tiles" the results are wrong and differ from run to run:
// the counter protocol re-derives clean, so the suspicion -- not proven -- is that a satisfied vmcnt is not yet visibility of an
// LDS-DMA's bytes to another wave's ds_read; the phased kernel has a workgroup barrier between the two), and SLOWER than the phased kernel either way: 103-116 us against 68 for fc.0.  Per 32-k
// tile a wave has 256 cycles of MFMAs and two LDS round trips of polling; the polls of the two waves of a SIMD do not hide behind each
// other's MFMAs.  Experiments build only.
#ifndef PP_LAG
#define PP_LAG 0
#endif
template <bool OUT_BF16>
__global__ __launch_bounds__(512, 2)
void fc_gemm_pipe_kernel(const void* __restrict__ Av, const void* __restrict__ Wv, const float* __restrict__ bias, void* __restrict__ Cv,
                         int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2)
{
    constexpr int BM = 256, BN = 128, ROWB = 64, TILE = (BM + BN) * ROWB, NBUF = 6, NG = 3, KQ = 2, TM = 2, TN = 2, LAG = PP_LAG;
    static_assert(NBUF * TILE + 64 <= 160 * 1024 && (LAG == 0 || LAG == 1), "");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const int sid = (li >> 5) * 8 + xcd;
    const int within = li & 31;
    const int sn = 1 << sn_log2, sm = 32 >> sn_log2;
    const int nsn = ntiles >> sn_log2;
    const int tm = (sid / nsn) * sm + (within >> sn_log2);
    const int tn = (sid % nsn) * sn + (within & (sn - 1));
    if (tm >= mtiles) return;
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wid & 3) * 64, wn = (wid >> 2) * 64;
    const int i = lane & 31, h = lane >> 5;

    // global -> LDS: 1 KB chunk c = 16 rows x 64 B of the stacked tile (rows 0..255 A, 256..383 W); wave wid brings chunks wid, wid + 8
    // (A) and wid + 16 (W); lane l fills slot l % 4 of row 16 c + l / 4 with the logical 16-byte column slot ^ swz(row), swz(r) = (r >> 2) & 3
    const size_t rowb = (size_t)K * 2;
    unsigned voff[NG];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        const int r = 16 * (wid + 8 * j) + lane / 4;
        const int col = (lane % 4) ^ ((r >> 2) & 3);
        const int grow = j < 2 ? r : r - BM;             // (the launcher sends only whole tiles here: M % 256 == 0)
        voff[j] = (unsigned)(grow * rowb + 16 * col);
    }
    const char* sA = static_cast<const char*>(Av) + (size_t)m0 * rowb;
    const char* sW = static_cast<const char*>(Wv) + (size_t)n0 * rowb;
    const unsigned lds_wave = lds_addr(smem) + wid * 1024;
    const unsigned cnt = lds_addr(smem) + NBUF * TILE;                  // freed[6] at +0, landed[6] at +32
    if (tid < 16) reinterpret_cast<unsigned*>(smem + NBUF * TILE)[tid] = 0u;
    __syncthreads();

    // fragment reads: lane (i, h) reads row (wave corner + 32 a + i), logical column 2 kq + h: k = 16 kq + 8 h + (0..7)
    const int sw = (i >> 2) & 3;                          // wave corners are multiples of 32 rows: swz(row) = swz(i)
    unsigned pa[KQ], pb[KQ];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
        const int fo = 16 * ((2 * kq + h) ^ sw);
        pa[kq] = (wm + i) * ROWB + fo;
        pb[kq] = (BM + wn + i) * ROWB + fo;
        asm volatile("" : "+v"(pa[kq]), "+v"(pb[kq]));
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    auto cnt_add = [&](unsigned addr) {
        if (lane == 0) asm volatile("ds_add_u32 %0, %1" :: "v"(addr), "v"(1u) : "memory");
    };
    auto cnt_wait = [&](unsigned addr, unsigned want) {
        for (;;) {
            unsigned v;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
            if ((unsigned)__builtin_amdgcn_readfirstlane((int)v) >= want) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    auto vm_wait = [&](int newer) {                       // at most `newer` tiles' worth of this wave's LDS-DMA still in flight
        newer = newer > 0 ? newer - 1 : 0;              // one tile more than the count says: see MEASURED above
        if (newer >= 4)      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (newer == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (newer == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (newer == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    auto load_frags = [&](auto bufc, float4 (&af)[KQ][TM], float4 (&bf)[KQ][TN]) {
        constexpr int b = decltype(bufc)::value;
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
#pragma unroll
            for (int a = 0; a < TM; ++a) af[kq][a] = *reinterpret_cast<const float4*>(smem + b * TILE + pa[kq] + a * 32 * ROWB);
#pragma unroll
            for (int c = 0; c < TN; ++c) bf[kq][c] = *reinterpret_cast<const float4*>(smem + b * TILE + pb[kq] + c * 32 * ROWB);
        }
    };
    auto math_q = [&](const float4 (&af)[KQ][TM], const float4 (&bf)[KQ][TN], int kq) {
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[kq][a]), __builtin_bit_cast(bf16x8, bf[kq][b]), acc[a][b], 0, 0, 0);
    };

    const int KT = (int)(rowb / ROWB);                    // >= 8 (checked by the launcher)
    float4 afA[KQ][TM], bfA[KQ][TN], afB[KQ][TM], bfB[KQ][TN];
#pragma unroll
    for (int u = 0; u < NBUF - LAG; ++u) issue_tile<2, 1>(lds_wave + u * TILE, sA + u * ROWB, sW + u * ROWB, voff);
    vm_wait(NBUF - LAG - 2);                              // tiles 0 and 1 of this wave have landed
    cnt_add(cnt + 32);
    cnt_add(cnt + 36);
    cnt_wait(cnt + 32, 8u);
    load_frags(std::integral_constant<int, 0>{}, afA, bfA);

    auto step = [&](int u, auto bufc, const float4 (&afC)[KQ][TM], const float4 (&bfC)[KQ][TN], float4 (&afN)[KQ][TM], float4 (&bfN)[KQ][TN]) {
        constexpr int buf = decltype(bufc)::value, nb = (buf + 1) % NBUF, b2 = (buf + 2) % NBUF, fb = (buf + NBUF - LAG) % NBUF;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                   // S0
        cnt_add(cnt + 4 * buf);
        __builtin_amdgcn_sched_barrier(0);
        math_q(afC, bfC, 0);                                                               // S1
        __builtin_amdgcn_sched_barrier(0);
        const int last = u + NBUF - LAG < KT ? u + NBUF - LAG : KT - 1;                    // newest tile in flight after this step's issue
        if (u >= LAG && u + NBUF - LAG < KT) {                                             // S2
            cnt_wait(cnt + 4 * fb, 8u * (unsigned)((u - LAG) / NBUF + 1));
            const size_t ko = (size_t)(u + NBUF - LAG) * ROWB;
            issue_tile<2, 1>(lds_wave + fb * TILE, sA + ko, sW + ko, voff);
        }
        if (u + 2 < KT) {
            vm_wait(last - (u + 2));
            cnt_add(cnt + 32 + 4 * b2);
        }
        if (u + 1 < KT) {
            cnt_wait(cnt + 32 + 4 * nb, 8u * (unsigned)((u + 1) / NBUF + 1));
            __builtin_amdgcn_sched_barrier(0);
            load_frags(std::integral_constant<int, nb>{}, afN, bfN);                       // S3
            math_q(afC, bfC, 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); }
        } else {
            __builtin_amdgcn_sched_barrier(0);
            math_q(afC, bfC, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    for (int u = 0; u < KT; u += 6) {
        step(u, std::integral_constant<int, 0>{}, afA, bfA, afB, bfB);
        if (u + 1 < KT) step(u + 1, std::integral_constant<int, 1>{}, afB, bfB, afA, bfA);
        if (u + 2 < KT) step(u + 2, std::integral_constant<int, 2>{}, afA, bfA, afB, bfB);
        if (u + 3 < KT) step(u + 3, std::integral_constant<int, 3>{}, afB, bfB, afA, bfA);
        if (u + 4 < KT) step(u + 4, std::integral_constant<int, 4>{}, afA, bfA, afB, bfB);
        if (u + 5 < KT) step(u + 5, std::integral_constant<int, 5>{}, afB, bfB, afA, bfA);
    }

#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn + 32 * b + i;
        const float bv = bias[col];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                float v = acc[a][b][r] + bv;
                if (relu) v = v < 0.f ? 0.f : v;
                if constexpr (OUT_BF16) static_cast<unsigned short*>(Cv)[(size_t)row * N + col] = f32_to_bf16(v);
                else static_cast<float*>(Cv)[(size_t)row * N + col] = v;
            }
    }
}

#endif  // DCE_EXPERIMENTS

// tile 2 = 256 x 128 with 128-byte K-tiles; tile 1 = 128 x 64 with 256-byte K-tiles (same 48 KB per K-tile,
// twice the MFMAs per phase that 128-byte K-tiles would give the small wave tile)
template <int T> struct PhTile;
template <> struct PhTile<2> { static constexpr int TM = 2, TN = 2, ROWB = 128; };
template <> struct PhTile<1> { static constexpr int TM = 1, TN = 1, ROWB = 256; };
// tile 3 (bf16 only, option bf16_k32=1, EXPERIMENTS BUILD; round 5's one experiment on this GEMM): the 256 x 128 tile with 32-k K-tiles --
// 3 x 24 KB of LDS and 128 registers, so that TWO workgroups share a CU and each one's phase hand-overs are covered by the other's MFMA
// phase (what took conv_x3 from 308 to 171 us); it needs >= 512 tiles per launch (8192 windows of fc.0).  MEASURED (profiles/r5i_gemm_k32.txt):
// bit-identical and SLOWER -- fc.0 at 8192 windows 164-166 us against 134 (0.385 against 0.474 of 2.5 PF), at 32768 windows 677-679 against
// 567-568.  The counters say why: twice the waves are resident (SQ_WAVE_CYCLES 2.84e8 against 1.12e8) for the same MFMA cycles in 1.47x
// the time; the 64-byte row pieces of a 32-k K-tile double the L2 requests (TCC_REQ 2.95e7 against 1.51e7) through a per-CU address path
// that the 64-k kernel already keeps busy 768 of 1024 cycles; waves wait 2.2x as long (SQ_WAIT_INST_ANY).  Two workgroups do not cover each
// other here because what they wait for is the same LDS-DMA path.  With this the bf16 GEMM is closed.
template <> struct PhTile<3> { static constexpr int TM = 2, TN = 2, ROWB = 64; };

// Schedule: the two-groups-one-phase-apart loop ships for both precisions; DCE_GEMM=lockstep selects the other loop
// (kept as a tested A/B variant).  Measured (r2q, interleaved rounds of the bench step):
//   fp32: phased 529 us (fc.0) / 65 (fc.3), lockstep 600 / 82.  With all eight waves in step, both waves of every
//         SIMD stand at the one barrier together -- and each wave's fragment reads sit inside its own MFMA stream;
//         one phase apart, half of the waves are always early and a math phase is nothing but MFMAs.
//   bf16: phased 68-71 us (fc.0), lockstep 74-75.  End to end the two trade places with the board's clock governor:
//         in 3-second runs on one box the step was 5 % FASTER with the slower lockstep GEMM (7.89 vs 7.52 M windows/s:
//         after the denser phased GEMM the fp32 conv stack ran 452 instead of 423 us), in 6000-step runs on another
//         box the conv stack ran 419 us behind either and phased won by 0.8 % (8.13 vs 8.06 M).  The kernel-level
//         result is the stable one; it decides.
static bool use_lockstep(bool /*bf16*/)
{
    return DCE_EXPERIMENTS && tune().gemm_lockstep;       // (the lockstep schedule -- the one with register spills -- is instantiated in the experiments build only)
}

template <bool BF16, bool OUT_BF16, int T> static hipError_t grant_phased()
{
    using P = PhTile<T>;
#if DCE_EXPERIMENTS
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_phased_kernel<BF16, OUT_BF16, P::TM, P::TN, P::ROWB, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PhCfg<P::TM, P::TN, P::ROWB>::LDS);
    if (e != hipSuccess) return e;
#endif
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_phased_kernel<BF16, OUT_BF16, P::TM, P::TN, P::ROWB, false, false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, PhCfg<P::TM, P::TN, P::ROWB>::LDS);
}

hipError_t init_fc_gemm_phased()
{
    hipError_t e;
#if DCE_EXPERIMENTS
    for (const void* k : {reinterpret_cast<const void*>(&fc_gemm_ki_kernel<true>), reinterpret_cast<const void*>(&fc_gemm_ki_kernel<false>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, PhCfg<2, 2, 128>::LDS)) != hipSuccess) return e;
#endif
#if DCE_EXPERIMENTS
    for (const void* k : {reinterpret_cast<const void*>(&fc_gemm_pipe_kernel<true>), reinterpret_cast<const void*>(&fc_gemm_pipe_kernel<false>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 384 * 64 + 64)) != hipSuccess) return e;
#endif
    for (const void* k : {
#if DCE_EXPERIMENTS
                          reinterpret_cast<const void*>(&fc_gemm_phased_kernel<false, false, 1, 1, 256, true, true>),
                          reinterpret_cast<const void*>(&fc_gemm_phased_kernel<true, false, 1, 1, 256, true, true>),
#endif
                          reinterpret_cast<const void*>(&fc_gemm_phased_kernel<false, false, 1, 1, 256, true, false>),
                          reinterpret_cast<const void*>(&fc_gemm_phased_kernel<true, false, 1, 1, 256, true, false>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, PhCfg<1, 1, 256>::LDS)) != hipSuccess) return e;
    if ((e = grant_phased<false, false, 2>()) != hipSuccess) return e;
    if ((e = grant_phased<false, false, 1>()) != hipSuccess) return e;
    if ((e = grant_phased<true, true, 2>()) != hipSuccess) return e;
    if ((e = grant_phased<true, false, 2>()) != hipSuccess) return e;
    if ((e = grant_phased<true, true, 1>()) != hipSuccess) return e;
#if DCE_EXPERIMENTS
    if ((e = grant_phased<true, true, 3>()) != hipSuccess) return e;
#endif
    return grant_phased<true, false, 1>();
}

// Tile choice: 256 x 128 when those tiles alone fill the chip (>= 192 of them), else 128 x 64 from 128 tiles up;
// 0 = this shape stays on the tile kernels of fc_gemm.hip.  DCE_GEMM=tile forces 0 (A/B).
static int phased_tile(int64_t M, int N, int K, int es)
{
    const Tuning& tu = tune();
    const bool off = tu.gemm_tile;
    const int min_tiles = tu.phased_min_tiles, min_tiles1 = tu.phased_min_tiles1, tmin = tu.phased_min;   // tmin 2: only the 256x128 tile
    if (off || (size_t)K * es % 256 || (size_t)K * es < 3 * 256 || M > (1 << 30)) return 0;
    if ((size_t)256 * K * es + 128 >= (1ull << 32)) return 0;                     // per-lane offsets are 32-bit
    // One workgroup per CU: a launch takes ceil(tiles / 256) rounds, and a round of 256 x 128 tiles lasts 3.83 rounds of
    // 128 x 64 tiles (fc.0: 525 vs 137 us).  Between the powers of two the small tile's finer rounds win: 3000 windows
    // are 192 big tiles = one round, 525 us, or 768 small ones = three rounds, 411 us.  (DCE_PHASED_COST=0: the first
    // tile size, from the large one down, that fills its minimum -- the rule before this cost model.)
    const bool cost_model = tu.phased_cost;
    int best = 0;
    double best_cost = 0.0;
    for (int t = 2; t >= tmin; --t) {
        const int bm = 128 * t, bn = 64 * t;
        if (N % bn) continue;
        const int nt = N / bn;
        if ((nt & (nt - 1)) != 0) continue;                                       // super-tile map wants a power of two
        // 256 x 128 tiles must fill the chip; the 128 x 64 tile still beats the tile kernels on half of it
        // (fc.0 at 512 windows: 128 tiles, 139 vs 160 us; at 64 tiles it loses, 135 vs 95 us)
        const int64_t tiles = ((M + bm - 1) / bm) * nt;
        if (tiles < (t == 2 ? min_tiles : min_tiles1)) continue;
        if (!cost_model) return t;
        const double cost = (double)((tiles + 255) / 256) * (t == 2 ? 3.83 : 1.0);
        if (best == 0 || cost < best_cost) { best = t; best_cost = cost; }
    }
    return best;
}

bool fc_gemm_phased_ok(int64_t M, int N, int K, int bf16) { return phased_tile(M, N, K, bf16 ? 2 : 4) != 0; }

template <bool BF16, bool OUT_BF16, int T>
static hipError_t launch_phased_cfg(const void* A, const void* W, const float* bias, void* C,
                                    int64_t M, int N, int K, int relu, hipStream_t st)
{
    using P = PhTile<T>;
    using Cfg = PhCfg<P::TM, P::TN, P::ROWB>;
    const int mtiles = (int)((M + Cfg::BM - 1) / Cfg::BM), ntiles = N / Cfg::BN;
    // the 32 blocks of one XCD form an sm x sn super-tile: each A panel enters (ntiles/sn) L2s, each W panel
    // (mtiles/sm).  fc.0 (A = 2 W bytes, 16 x 16 tiles): 4 x 8 -> 2 A + 4 W = 310 MB of fabric reads per launch,
    // against 388 MB for 8 x 4 or 2 x 16.  DCE_PHASED_SN overrides log2(sn) (A/B).
    int sn_log2 = tune().phased_sn;
    while ((1 << sn_log2) > ntiles) --sn_log2;         // (32/ntiles) x ntiles when N is narrow
    const int sm = 32 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 32;
#if DCE_EXPERIMENTS
    if constexpr (BF16 && T == 2) {
        if (tune().gemm_ki && !use_lockstep(true) && M % 256 == 0 && K % 64 == 0 && K >= 256) {       // K-tiles dealt out between the wave groups (whole tiles only)
            plan_note("fc_ki256x128");
            hipLaunchKernelGGL((fc_gemm_ki_kernel<OUT_BF16>), dim3(grid), dim3(512), Cfg::LDS, st, A, W, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
            return hipGetLastError();
        }
    }
    if constexpr (BF16 && T == 2) {
        if (tune().gemm_pipe && !use_lockstep(true) && M % 256 == 0 && K % 32 == 0 && K >= 256) {       // no workgroup barriers in the K loop (fc_gemm_pipe_kernel; whole tiles only)
            plan_note("fc_pipe256x128");
            hipLaunchKernelGGL((fc_gemm_pipe_kernel<OUT_BF16>), dim3(grid), dim3(512), 6 * 384 * 64 + 64, st, A, W, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
            return hipGetLastError();
        }
    }
#endif
    // (the note names the operand type too: bench.py prices a stage against the pipe its kernel family runs on)
    if (BF16) plan_note(use_lockstep(BF16) ? (T == 2 ? "fc_lockstep256x128_bf16" : "fc_lockstep128x64_bf16") : (T == 3 ? "fc_phased256x128_k32_bf16" : T == 2 ? "fc_phased256x128_bf16" : "fc_phased128x64_bf16"));
    else plan_note(use_lockstep(BF16) ? (T == 2 ? "fc_lockstep256x128" : "fc_lockstep128x64") : (T == 2 ? "fc_phased256x128" : "fc_phased128x64"));
#if DCE_EXPERIMENTS
    if (use_lockstep(BF16))
        hipLaunchKernelGGL((fc_gemm_phased_kernel<BF16, OUT_BF16, P::TM, P::TN, P::ROWB, false, true>), dim3(grid), dim3(512), Cfg::LDS, st,
                           A, W, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2, nullptr, nullptr, 0ll);
    else
#endif
        hipLaunchKernelGGL((fc_gemm_phased_kernel<BF16, OUT_BF16, P::TM, P::TN, P::ROWB, false, false>), dim3(grid), dim3(512), Cfg::LDS, st,
                           A, W, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2, nullptr, nullptr, 0ll, t_gate);
    return hipGetLastError();
}

hipError_t launch_fc_gemm_phased(const void* A, const void* W, const float* bias, void* C, int bf16, int out_bf16,
                                 int64_t M, int N, int K, int relu, hipStream_t st)
{
    const int t = phased_tile(M, N, K, bf16 ? 2 : 4);
    if (t == 0 || (!bf16 && out_bf16)) return hipErrorInvalidValue;
    if (!bf16) return t == 2 ? launch_phased_cfg<false, false, 2>(A, W, bias, C, M, N, K, relu, st)
                             : launch_phased_cfg<false, false, 1>(A, W, bias, C, M, N, K, relu, st);
#if DCE_EXPERIMENTS
    // (A/B, bf16_k32=1) the large tile with 32-k K-tiles, two workgroups per CU: needs two tiles per CU
    if (t == 2 && out_bf16 && tune().bf16_k32 && !use_lockstep(true) && ((M + 255) / 256) * (N / 128) >= 512 && K % 32 == 0 && K >= 128)
        return launch_phased_cfg<true, true, 3>(A, W, bias, C, M, N, K, relu, st);
#endif
    if (out_bf16) return t == 2 ? launch_phased_cfg<true, true, 2>(A, W, bias, C, M, N, K, relu, st)
                                : launch_phased_cfg<true, true, 1>(A, W, bias, C, M, N, K, relu, st);
    return t == 2 ? launch_phased_cfg<true, false, 2>(A, W, bias, C, M, N, K, relu, st)
                  : launch_phased_cfg<true, false, 1>(A, W, bias, C, M, N, K, relu, st);
}

// fc.3 + fc.6 chunk sums in one launch: the 128 x 64 phased tile, when it is the tile fc.3 would get anyway
bool fc23_fused_ok(int64_t M, int bf16)
{
    // Fused only where fc.3 gets the 128 x 64 tile anyway (one round of tiles, 3072..~12000 windows).  Longer chunks
    // take the 256 x 128 tile + the stand-alone tail: forcing the small tile there (DCE_FC23=always) measured 0.8 %
    // slower end to end on the 1e6-window sequence (3.849 vs 3.879 M windows/s) -- twice the phase hand-overs cost
    // more than h2's HBM round trip.  DCE_FC23=split: never fused (A/B).
    const int mode = tune().fc23;
    if (mode == 1) return false;
    const int t = phased_tile(M, FC2, FC1, bf16 ? 2 : 4);
    if (t == 1) return true;
    return mode == 2 && t == 2 && ((M + 127) / 128) * FC6_NCHUNK >= 192;
}

hipError_t launch_fc23_fused(const void* h1, const void* W2, const float* b2, const float* W3, int bf16,
                             float* part, int64_t part_rows, float* h2_out, int64_t M, hipStream_t st)
{
    using Cfg = PhCfg<1, 1, 256>;
    static_assert(Cfg::BN == FC6_CHUNK && FC2 / Cfg::BN == FC6_NCHUNK, "one column tile of fc.3 = one chunk of fc.6");
    const int mtiles = (int)((M + Cfg::BM - 1) / Cfg::BM), ntiles = FC2 / Cfg::BN;
    const int sn_log2 = 2, sm = 32 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 32;
    void* h2v = static_cast<void*>(h2_out);
    const long long pr = (long long)part_rows;
#define FC23_LAUNCH(BF, LS) hipLaunchKernelGGL((fc_gemm_phased_kernel<BF, false, 1, 1, 256, true, LS>), dim3(grid), dim3(512), Cfg::LDS, st, \
                                               h1, W2, b2, h2v, (int)M, FC2, FC1, 1, mtiles, ntiles, sn_log2, W3, part, pr, t_gate)
    if (bf16) plan_note(use_lockstep(false) ? "fc23_fused_lockstep128x64_bf16" : "fc23_fused_phased128x64_bf16");
    else plan_note(use_lockstep(false) ? "fc23_fused_lockstep128x64" : "fc23_fused_phased128x64");
#if DCE_EXPERIMENTS
    if (use_lockstep(bf16)) { if (bf16) FC23_LAUNCH(true, true); else FC23_LAUNCH(false, true); }
    else
#endif
    { if (bf16) FC23_LAUNCH(true, false); else FC23_LAUNCH(false, false); }
#undef FC23_LAUNCH
    return hipGetLastError();
}

}  // namespace dce

#if PH_TRACE
extern "C" int dce_debug_phase_trace_read(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_ph_trace), sizeof(unsigned long long) * 8 * 64 * 4);
}
#endif
