// fc_gemm_h2.hip -- fc.0 (Linear + bias + ReLU, reference src/contact_cnn.py:48-49) at chip-filling sizes for the precision
// DCE_FP32_F16X2: both operands as TWO fp16 terms each, three v_mfma_f32_32x32x16_f16 per block product (h1 k1 + h1 k2 + h2 k1, each exact
// in the fp32 accumulator) where fc_gemm_x3.hip's exact three-term bf16 split issues six.  conv_h2.hip has the arithmetic and the scales:
// a row of A (one window's features) carries its own power-of-two scale 2^row_scale[m], the weights one scale 2^sw for the layer; the
// epilogue takes both off (v_ldexp_f32, exact) before the bias.
//
// Kernel: the phased schedule of fc_gemm_x3.hip / fc_gemm_phased.hip (one workgroup per CU, 8 waves = two groups of four a PHASE apart,
// LDS-DMA, math phases that are nothing but MFMAs), cut for two terms:
//   * workgroup tile 256 x 128, wave tile 64 x 64 (2 x 2 blocks of 32 x 32), K-tile = 32 k;
//   * operands in HBM as [row][K-tile][term (2)][32 fp16]: a row's K-tile is ONE 128-byte line [h1 x 32 | h2 x 32], so the LDS-DMA asks
//     L2 for whole lines without the pair interleaving fc_gemm_x3.hip's 64-byte plane segments need;
//   * a K-tile in LDS = (256 + 128) rows x 128 B = 48 KB; two buffers.  The LDS-DMA of tile u+1 is issued by group 0 alone at the
//     start of its load(u): twelve 1 KB pieces per wave;
//   * per wave and K-tile: 16 ds_read_b128 (64 VGPRs) feed 24 MFMAs = 768 cycles of the matrix pipe;
//   * rows are eight 16-byte slots; slot s of row r holds logical column s ^ ((r >> 1) & 7) (logical column = 4 term + k / 8): the sixteen
//     rows a ds_read_b128 serves per cycle -- two rows share a 128-byte half of the bank space -- then sit in eight different slots.
#include "dce_kernels.h"
#include <type_traits>

namespace dce {

typedef float h2_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h2_f16x8 __attribute__((ext_vector_type(8)));

#ifndef H2_TRACE
#define H2_TRACE 0
#endif

namespace {

constexpr int H2_BM = 256, H2_BN = 128, H2_KT = 32, H2_ROWS = H2_BM + H2_BN;
constexpr int H2_ROWB = 128;                                      // bytes of a row's K-tile: two terms x 32 k x 2 B
constexpr int H2_TILE = H2_ROWS * H2_ROWB, H2_LDS = 2 * H2_TILE;  // 48 KB; 96 KB
constexpr int H2_NCH = H2_TILE / 1024 / 4;                        // 1 KB pieces per ISSUING wave (the four of group 0) per K-tile: 12
constexpr int H2_NQA = H2_BM / 8 / 4;                             // ... of which from the A panel: 8 (a piece = 8 rows)
static_assert(H2_NCH == 12 && H2_NQA == 8 && (H2_NCH - H2_NQA) * 4 * 8 == H2_BN, "piece deal");

__device__ __forceinline__ unsigned h2_lds_addr(const void* p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ constexpr int h2_swz(int r) { return (r >> 1) & 7; }

// one 1 KB piece of a K-tile: 64 lanes x 16 bytes from base + v (per-lane byte offset) to LDS at M0 (+ lane * 16); M0 then steps on by
// 4 KB to the wave's next piece (fc_gemm_x3.hip: x3_piece)
__device__ __forceinline__ unsigned h2_m0_begin(unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1" : "=&s"(keep) : "s"(lds) : "memory");
    return keep;
}
__device__ __forceinline__ void h2_piece(const char* base, unsigned v)
{
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %1, %0\n\ts_add_u32 m0, m0, 0x1000" :: "s"(base), "v"(v) : "memory");
}
__device__ __forceinline__ void h2_m0_end(unsigned keep)
{
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep) : "memory");
}

}  // namespace

__global__ __launch_bounds__(512, 2)
void fc_gemm_h2_kernel(const unsigned short* __restrict__ A2, const int* __restrict__ row_scale, const unsigned short* __restrict__ W2, int sw,
                       const float* __restrict__ bias, float* __restrict__ C,
                       int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2)
{
    constexpr int BM = H2_BM, BN = H2_BN, ROWB = H2_ROWB, TILE = H2_TILE, NCH = H2_NCH, NQA = H2_NQA;
    extern __shared__ __attribute__((aligned(16))) char h2_smem[];
    // ---- XCD-aware tile assignment, as fc_gemm_phased.hip: the 32 blocks of one XCD form an sm x sn super-tile
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
    const int grp = wid >> 2;                                // phase group; waves w and w+4 share a SIMD
    const int wm = (wid & 3) * 64, wn = grp * 64;            // this wave's corner of the block tile
    const int i = lane & 31, h = lane >> 5;

    // ---- global -> LDS, issued by the four waves of group 0 only.  Wave w' = wid & 3 brings pieces c = w' + 4 j, j = 0..11: j < 8: A rows
    //      8 (w' + 4 j) .., else W rows 8 (w' + 4 (j - 8)) ..; lane (lr = lane / 8, slot = lane % 8) fills slot `slot` of row lr of the piece
    //      with the logical column slot ^ swz(row)
    const size_t rowb = (size_t)K * 4;                                   // bytes of one row: K x two terms x 2 B
    unsigned voff[NCH];
    {
        const int lr = lane >> 3, slot = lane & 7, w4 = wid & 3;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int r = 8 * (w4 + 4 * (j < NQA ? j : j - NQA)) + lr;   // row inside the A panel / the W panel (both start at multiples of 16 rows)
            const int col = slot ^ h2_swz(r);
            int grow = r;
            if (j < NQA && m0 + grow >= M) grow = M - 1 - m0;             // rows past M re-read the last one (never stored)
            voff[j] = (unsigned)((size_t)grow * rowb + 16 * col);
        }
    }
    const char* sA = reinterpret_cast<const char*>(A2) + (size_t)m0 * rowb;
    const char* sW = reinterpret_cast<const char*>(W2) + (size_t)n0 * rowb;
    constexpr size_t KSTEP = ROWB;                                        // bytes from one K-tile to the next in HBM
    const unsigned lds_wave = h2_lds_addr(h2_smem) + (wid & 3) * 1024;    // piece w' of buffer 0; piece j lands 4 j KB behind it
    auto issue = [&](unsigned lds0, size_t ko) {                          // the twelve pieces of one K-tile, in a bunch
        const unsigned keep = h2_m0_begin(lds0);
#pragma unroll
        for (int j = 0; j < NCH; ++j) h2_piece((j < NQA ? sA : sW) + ko, voff[j]);
        h2_m0_end(keep);
    };

    // ---- fragment reads: lane (i, h) reads row (corner + 32 blk + i), logical column 4 term + 2 kq + h  (k = 16 kq + 8 h + 0..7)
    const int swz = h2_swz(i);
    unsigned fa[2][2][2], fb[2][2][2];                                    // [buffer][term][kq]: byte offsets of block 0
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                const int o = 16 * ((4 * p + 2 * kq + h) ^ swz);
                fa[b][p][kq] = b * TILE + (wm + i) * ROWB + o;
                fb[b][p][kq] = b * TILE + (BM + wn + i) * ROWB + o;
                asm volatile("" : "+v"(fa[b][p][kq]), "+v"(fb[b][p][kq]));    // stay in registers (see fc_gemm_phased.hip)
            }

    h2_f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 af[2][2][2], bf[2][2][2];                                      // fragments of one K-tile: [kq][term][block]
    auto load_frags = [&](int buf) {                                      // buf is a compile-time constant at every call
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    af[kq][p][blk] = *reinterpret_cast<const float4*>(h2_smem + fa[buf][p][kq] + blk * 32 * ROWB);
                    bf[kq][p][blk] = *reinterpret_cast<const float4*>(h2_smem + fb[buf][p][kq] + blk * 32 * ROWB);
                }
    };
    auto math = [&]() {                                                   // three terms per block pair, small ones first; consecutive MFMAs go to different accumulators
        constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(h2_f16x8, af[kq][TA[t]][a]), __builtin_bit_cast(h2_f16x8, bf[kq][TB[t]][b]), acc[a][b], 0, 0, 0);
    };

    const int KT = K / H2_KT;                                             // >= 2 (checked by the launcher)
    // Phases p = 0, 1, 2, ...; a workgroup barrier ends each (fc_gemm_x3.hip has the hazard argument).
    //   group 0: load(u) in phase 2u, math(u) in 2u+1            group 1: load(u) in 2u+1, math(u) in 2u+2
    if (grp == 0) {
        issue(lds_wave, 0);
        issue(lds_wave + TILE, KSTEP);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");                 // tile 0 landed
    }
    asm volatile("s_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");                 // group 1 sits out phase 0
    auto ktile = [&](int u, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        // ---- load phase
        if (grp == 0 && u >= 1 && u + 1 < KT) issue(lds_wave + (buf ^ 1) * TILE, (size_t)(u + 1) * KSTEP);
        load_frags(buf);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- math phase
        math();
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
#pragma unroll 1
    for (int u = 0; u < KT; u += 2) {
        ktile(u, std::integral_constant<int, 0>{});
        if (u + 1 < KT) ktile(u + 1, std::integral_constant<int, 1>{});
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");                 // same number of barriers for both groups

    // ---- epilogue: off with the scales (row's and layer's), bias + (ReLU); D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    auto store_tile = [&](auto full) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            int ex[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (!decltype(full)::value && row >= M) row = M - 1;
                ex[r] = -(row_scale[row] + sw);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = n0 + wn + 32 * b + i;
                const float bv = bias[col];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float v = __builtin_ldexpf(acc[a][b][r], ex[r]) + bv;
                    if (relu) v = v < 0.f ? 0.f : v;                      // keeps NaN like torch
                    if (decltype(full)::value || row < M) C[(size_t)row * N + col] = v;
                }
            }
        }
    };
    if (m0 + BM <= M) store_tile(std::true_type{});
    else store_tile(std::false_type{});
}



#if DCE_EXPERIMENTS
// ---- (experiments build, option h2_ksplit=1; measured 5 % SLOWER than the N-split kernel above: 203.7 against 194.1 us per 4096 windows by HIP
// events, 214.5 against 203.8 under the tracer, profiles/r5q_f16x2_ksplit_ab.txt -- fewer, longer phases and a quarter less LDS traffic do not
// buy time on a kernel that runs at the clock the board grants it) the same GEMM with the K-TILES dealt out between the two wave groups: a wave owns 64 x 128 of the tile -- 2 x 4 blocks,
// 128 accumulator registers -- for every OTHER K-tile, so a phase is 48 MFMAs against 24 fragment reads per wave (0.5 per MFMA; the N-split form
// above: 24 against 16, 0.67, and a load phase as long as the math phase beside it), and a K-tile's fragments are read by four waves instead of
// eight (96 KB of LDS reads per 48 KB tile instead of 128).  Three buffers: tile p + 2 is issued by the group that loads tile p, in phase p, and
// first read in phase p + 2.  At the end the groups exchange halves of their accumulators through LDS (fixed order: even tiles + odd tiles) and
// each finishes two of the four column blocks.
__global__ __launch_bounds__(512, 2)
void fc_gemm_h2k_kernel(const unsigned short* __restrict__ A2, const int* __restrict__ row_scale, const unsigned short* __restrict__ W2, int sw,
                        const float* __restrict__ bias, float* __restrict__ C,
                        int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2)
{
    constexpr int BM = H2_BM, ROWB = H2_ROWB, TILE = H2_TILE, NCH = H2_NCH, NQA = H2_NQA;
    extern __shared__ __attribute__((aligned(16))) char h2_smem[];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const int sid = (li >> 5) * 8 + xcd;
    const int within = li & 31;
    const int sn = 1 << sn_log2, sm = 32 >> sn_log2;
    const int nsn = ntiles >> sn_log2;
    const int tm = (sid / nsn) * sm + (within >> sn_log2);
    const int tn = (sid % nsn) * sn + (within & (sn - 1));
    if (tm >= mtiles) return;
    const int m0 = tm * BM, n0 = tn * H2_BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;                                // K-tile parity of this wave's group; waves w and w+4 share a SIMD
    const int wm = (wid & 3) * 64;                           // this wave's rows of the block tile (all 128 columns)
    const int i = lane & 31, h = lane >> 5;

    // (pieces: ONE per-lane offset for the A panel and one for the W panel -- a piece's 8 rows start 32 rows behind the previous one's, which moves
    //  the wave-uniform base, not the lanes; rows past M are READ (the feature buffer is padded by a tile of rows) and never stored)
    const size_t rowb = (size_t)K * 4;
    unsigned voffA, voffW;
    {
        const int lr = lane >> 3, slot = lane & 7, w4 = wid & 3;
        const int r = 8 * w4 + lr;                                        // row of piece 0; swz(r + 32 j) = swz(r)
        voffA = voffW = (unsigned)((size_t)r * rowb + 16 * (slot ^ h2_swz(r)));
        asm volatile("" : "+v"(voffA), "+v"(voffW));
    }
    const char* sA = reinterpret_cast<const char*>(A2) + (size_t)m0 * rowb;
    const char* sW = reinterpret_cast<const char*>(W2) + (size_t)n0 * rowb;
    const unsigned lds_wave = h2_lds_addr(h2_smem) + (wid & 3) * 1024;
    auto issue = [&](unsigned lds0, size_t ko) {
        const unsigned keep = h2_m0_begin(lds0);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (j < NQA) h2_piece(sA + ko + (size_t)j * 32 * rowb, voffA);
            else         h2_piece(sW + ko + (size_t)(j - NQA) * 32 * rowb, voffW);
        }
        h2_m0_end(keep);
    };

    const int swz = h2_swz(i);
    unsigned fa[2][2], fb[2][2];                                          // [term][kq]: byte offsets of block 0 in buffer 0
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int kq = 0; kq < 2; ++kq) {
            const int o = 16 * ((4 * p + 2 * kq + h) ^ swz);
            fa[p][kq] = (wm + i) * ROWB + o;
            fb[p][kq] = (BM + i) * ROWB + o;
            asm volatile("" : "+v"(fa[p][kq]), "+v"(fb[p][kq]));
        }

    h2_f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 af[2][2][2], bf[2][2][4];                                      // [kq][term][block]
    auto load_frags = [&](unsigned bo) {                                  // bo: the buffer's byte offset
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) af[kq][p][blk] = *reinterpret_cast<const float4*>(h2_smem + (fa[p][kq] + bo) + blk * 32 * ROWB);
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) bf[kq][p][blk] = *reinterpret_cast<const float4*>(h2_smem + (fb[p][kq] + bo) + blk * 32 * ROWB);
            }
    };
    auto math = [&]() {
        constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
        for (int kq = 0; kq < 2; ++kq)
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                            __builtin_bit_cast(h2_f16x8, af[kq][TA[t]][a]), __builtin_bit_cast(h2_f16x8, bf[kq][TB[t]][b]), acc[a][b], 0, 0, 0);
    };

    const int KT = K / H2_KT;                                             // even, >= 4 (checked by the launcher)
    // Phase p = 0 .. KT: group p & 1 issues tile p + 2 and reads tile p's fragments, the other group multiplies tile p - 1.
    //   WAR: the buffer of tile p + 2 held tile p - 1, read in phase p - 1, which ended with lgkmcnt(0) + barrier;
    //   RAW: tile p + 2 is first read in phase p + 2; its issuing group waits vmcnt(0) at the end of its math phase p + 1, ahead of that barrier.
    issue(lds_wave + grp * TILE, (size_t)grp * ROWB);                     // tiles 0 (group 0) and 1 (group 1)
    if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // group 1 sits out phase 0 (tile 1 has landed when it ends)
    unsigned bu = grp, bn = grp == 0 ? 2 : 0;                             // buffers of tile u and of tile u + 2
#pragma unroll 1
    for (int u = grp; u < KT; u += 2) {
        // ---- load phase (phase u)
        if (u + 2 < KT) issue(lds_wave + bn * TILE, (size_t)(u + 2) * ROWB);
        load_frags(bu * TILE);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        // ---- math phase (phase u + 1)
        math();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        bu = bn; bn = bn == 0 ? 2 : bn - 1;                               // (u + 2) % 3, (u + 4) % 3: 0 -> 2 -> 1 -> 0
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");                 // same number of barriers for both groups

    // ---- the groups exchange halves: group g keeps column blocks 2g, 2g + 1 and sends the other two (64 floats per lane) through LDS; then
    //      the epilogue on this wave's two column blocks.  (G is the group as a compile-time constant: a run-time index into the accumulators
    //      would put them in scratch.)
    auto finish = [&](auto gc) {
        constexpr int G = decltype(gc)::value;
        float* const x = reinterpret_cast<float*>(h2_smem) + ((size_t)wid * 64 * 64);      // this wave's 16 KB: [64 values][64 lanes]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) x[((a * 2 + b) * 16 + r) * 64 + lane] = acc[a][2 * (1 - G) + b][r];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float* const y = reinterpret_cast<const float*>(h2_smem) + ((size_t)(wid ^ 4) * 64 * 64);   // the partner wave (same rows, other group)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mine = acc[a][2 * G + b][r], theirs = y[((a * 2 + b) * 16 + r) * 64 + lane];
                    acc[a][2 * G + b][r] = G == 0 ? mine + theirs : theirs + mine;        // even tiles + odd tiles, whoever adds
                }
        auto store_tile = [&](auto full) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                int ex[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (!decltype(full)::value && row >= M) row = M - 1;
                    ex[r] = -(row_scale[row] + sw);
                }
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int col = n0 + 64 * G + 32 * b + i;
                    const float bv = bias[col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float v = __builtin_ldexpf(acc[a][2 * G + b][r], ex[r]) + bv;
                        if (relu) v = v < 0.f ? 0.f : v;
                        if (decltype(full)::value || row < M) C[(size_t)row * N + col] = v;
                    }
                }
            }
        };
        if (m0 + BM <= M) store_tile(std::true_type{});
        else store_tile(std::false_type{});
    };
    if (grp == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
}
#endif

hipError_t init_fc_gemm_h2()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_h2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS);
#if DCE_EXPERIMENTS
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_h2k_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * H2_TILE);
#endif
    return e;
}

int fc_gemm_h2_pad_rows() { return H2_BM; }                       // rows the A operand's buffer holds beyond M (read by ragged tiles, never used)

// 256 x 128 tiles must fill the chip (as the phased fp32 kernel asks of its large tile)
bool fc_gemm_h2_ok(int64_t M, int N, int K)
{
    if (N % H2_BN || K % H2_KT || K < 2 * H2_KT || M <= 0) return false;
    const int nt = N / H2_BN;
    if ((nt & (nt - 1)) != 0) return false;
    if ((size_t)(M > N ? M : N) * K * 4 >= (1ull << 32)) return false;              // per-lane offsets are 32-bit
    return ((M + H2_BM - 1) / H2_BM) * nt >= tune().x3_min_tiles;
}

hipError_t launch_fc_gemm_h2(const unsigned short* A2, const int* row_scale, const unsigned short* W2, int sw, const float* bias, float* C,
                             int64_t M, int N, int K, int relu, hipStream_t st)
{
    if (!fc_gemm_h2_ok(M, N, K)) return hipErrorInvalidValue;
    const int mtiles = (int)((M + H2_BM - 1) / H2_BM), ntiles = N / H2_BN;
    int sn_log2 = tune().phased_sn;
    while ((1 << sn_log2) > ntiles) --sn_log2;
    const int sm = 32 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 32;
#if DCE_EXPERIMENTS
    // (the K-split form reads whole tiles of A rows: the caller's buffer is padded by H2_BM rows -- fc_gemm_h2_pad_rows)
    if (tune().h2_ksplit && (K / H2_KT) % 2 == 0 && K / H2_KT >= 4) {
        plan_note("fc_h2k_256x128");
        hipLaunchKernelGGL(fc_gemm_h2k_kernel, dim3(grid), dim3(512), 3 * H2_TILE, st, A2, row_scale, W2, sw, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
        return hipGetLastError();
    }
#endif
    plan_note("fc_h2_256x128");
    hipLaunchKernelGGL(fc_gemm_h2_kernel, dim3(grid), dim3(512), H2_LDS, st, A2, row_scale, W2, sw, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
    return hipGetLastError();
}

}  // namespace dce
