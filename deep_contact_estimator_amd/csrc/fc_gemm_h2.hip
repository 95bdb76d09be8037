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
#include "fc6_chain.h"
#include <type_traits>

namespace dce {

typedef float h2_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h2_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 h2_bf16x8 __attribute__((ext_vector_type(8)));

#ifndef H2_TRACE
#define H2_TRACE 0
#endif
#ifndef H2K_NOSPILL
#define H2K_NOSPILL 0        // (experiments A/B, -DH2K_NOSPILL=1) the fc.0 K-split variant's final exchange one row block at a time: no scratch (it refuted the scratch reading of round 5's fault)
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
// ("scc" in the clobber list is the fix of round 5's intermittent GPU memory fault: the s_add_u32 on m0 rewrites SCC, and without the clobber the scheduler
//  was free to put this statement between the s_add_u32 and the s_addc_u32 that form the NEXT piece's 64-bit base -- the carry was lost, and a piece whose
//  rows lie beyond a multiple of 4 GB that the operand buffer happens to cross was fetched from 4 GB below.  Seen in the assembly of all three K-split
//  instantiations, the shipped fc.3 one among them; DESIGN.md 4.6, profiles/r6k_*.  -DH2_SCC_UNDECLARED=1: the old statement, for the A/B.)
#ifndef H2_SCC_UNDECLARED
#define H2_SCC_UNDECLARED 0
#elif H2_SCC_UNDECLARED && !DCE_EXPERIMENTS
#error "H2_SCC_UNDECLARED reproduces a bug: experiments builds only"
#endif
__device__ __forceinline__ void h2_piece(const char* base, unsigned v)
{
#if H2_SCC_UNDECLARED
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %1, %0\n\ts_add_u32 m0, m0, 0x1000" :: "s"(base), "v"(v) : "memory");
#else
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %1, %0\n\ts_add_u32 m0, m0, 0x1000" :: "s"(base), "v"(v) : "memory", "scc");
#endif
}
__device__ __forceinline__ void h2_m0_end(unsigned keep)
{
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep) : "memory");
}

// the scale exponent of a row of h1 from the scale exponent of its features (see fc_gemm_h2_kernel<OUT2>): h1 * 2^s1 < 2^15
__device__ __forceinline__ int h2_h1_scale(int sfeat, int eW, int eB)
{
    const int ea = eW + 15 - sfeat, eb = (ea > eB ? ea : eB) + 1, s1 = 15 - eb;
    return s1 > 180 ? 180 : s1 < -114 ? -114 : s1;
}
// the two fp16 terms of two (already scaled) values: p[k] = (term k of v0) | (term k of v1) << 16   (conv_h2.hip: hx_split2)
__device__ __forceinline__ void h2_split2(float v0, float v1, unsigned (&p)[2])
{
    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f16x2 hh = __builtin_convertvector(f32x2{v0, v1}, f16x2);
    p[0] = __builtin_bit_cast(unsigned, hh);
    p[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{v0 - (float)hh[0], v1 - (float)hh[1]}, f16x2));
}

}  // namespace

// OUT2 (fc.0 in front of an fc.3 that takes two-term operands too): C leaves as two fp16 terms of h1 * 2^s1, [row][N / 32][term][32] in H1, with
// s1 = the row's scale exponent in h1_scale[row].  h1's largest entry is not known before the last column tile is done, so s1 comes from a bound:
// |h1| <= ||feat||_2 max_n ||W1_n||_2 + max|b1| <= sqrt(K) 2^(15 - row_scale) max_n ||W1_n||_2 + max|b1| < 2^eb, eb = max(eW + 15 - row_scale, eB) + 1
// (eW, eB: the two static terms' exponents, from dce_finalize_weights) -- h1 * 2^(15 - eb) is below 2^15 whatever the data; the bound is ~2^6 loose
// on ordinary data (Cauchy-Schwarz: sqrt(K) / 4; features: a third of them at their maximum), which leaves the largest entry near 2^9: first terms
// normal down to 2^-23 of it, fp16 subnormals (honoured by the matrix pipe) below.
template <bool OUT2>
__global__ __launch_bounds__(512, 2)
void fc_gemm_h2_kernel(const unsigned short* __restrict__ A2, const int* __restrict__ row_scale, const unsigned short* __restrict__ W2, int sw,
                       const float* __restrict__ bias, float* __restrict__ C,
                       int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2,
                       unsigned short* __restrict__ H1 = nullptr, int* __restrict__ h1_scale = nullptr, int eW = 0, int eB = 0)
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
        // (sched_barrier: MFMAs touch no memory, so without it hipcc moves two thirds of the math phase up between the fragment reads of the load
        //  phase -- across the barrier -- and the two groups no longer cover each other: the first build of this kernel ran that way, 194 us)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- math phase
        math();
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
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
            int ex[16], s1[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (!decltype(full)::value && row >= M) row = M - 1;
                const int sf = row_scale[row];
                ex[r] = -(sf + sw);
                if constexpr (OUT2) {
                    s1[r] = h2_h1_scale(sf, eW, eB);
                    if (tn == 0 && wn == 0 && i == 0 && (decltype(full)::value || m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h < M)) h1_scale[row] = s1[r];
                }
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int col = n0 + wn + 32 * b + i;
                const float bv = bias[col];
                if constexpr (OUT2) {
                    unsigned short* const hb = H1 + (size_t)(col >> 5) * 64 + (col & 31);
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {                     // rows r, r + 1 are neighbours: one split for the pair
                        const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float v0 = __builtin_ldexpf(acc[a][b][r], ex[r]) + bv, v1 = __builtin_ldexpf(acc[a][b][r + 1], ex[r + 1]) + bv;
                        if (relu) { v0 = v0 < 0.f ? 0.f : v0; v1 = v1 < 0.f ? 0.f : v1; }
                        unsigned p[2];
                        h2_split2(__builtin_ldexpf(v0, s1[r]), __builtin_ldexpf(v1, s1[r + 1]), p);
                        if (decltype(full)::value || row < M) { unsigned short* d = hb + (size_t)row * (2 * N); d[0] = (unsigned short)p[0]; d[32] = (unsigned short)p[1]; }
                        if (decltype(full)::value || row + 1 < M) { unsigned short* d = hb + (size_t)(row + 1) * (2 * N); d[0] = (unsigned short)(p[0] >> 16); d[32] = (unsigned short)(p[1] >> 16); }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                        float v = __builtin_ldexpf(acc[a][b][r], ex[r]) + bv;
                        if (relu) v = v < 0.f ? 0.f : v;                  // keeps NaN like torch
                        if (decltype(full)::value || row < M) C[(size_t)row * N + col] = v;
                    }
                }
            }
        }
    };
    if (m0 + BM <= M) store_tile(std::true_type{});
    else store_tile(std::false_type{});
}

// ---- the K-TILES dealt out between the two wave groups: a wave owns (32 AB) x BN of the tile -- AB x BB blocks -- for every OTHER K-tile, and the
// groups add their halves at the end (fixed order: even tiles + odd tiles).  Three LDS buffers: tile p + 2 is issued by the group that loads tile p,
// in phase p, and first read in phase p + 2.  Two uses:
//   * fc.3 + fc.6's chunk sums (H2Fc3: 128 x 64 tile = one chunk of fc.6's summation tree, 256 tiles at 4096 windows, 64 k per phase as two 32-k
//     sub-tiles, 32 x 64 wave tiles: 24 MFMAs against 24 fragment reads per wave and phase -- with an N split between the groups a wave's 32 x 32
//     would read 8 fragments for 6 MFMAs), epilogue as fc_gemm_phased.hip's FUSE6: h2 tile -> LDS -> fc6_chunk_mfma -> `part`;
//   * (experiments build only, option h2_ksplit) fc.0 on 64 x 128 wave tiles (H2KCfg<256, 128, 2, 4, 1>), 48 MFMAs against 24 reads per phase: no faster than the
//     N-split kernel above (174.0 against 176.9 us per 4096 windows, profiles/r5q_f16x2_ksplit_ab.txt) -- and the ONE kernel of either build that needs scratch:
//     128 accumulators + 96 fragment registers leave no room for its final exchange, 27 VGPRs spill (H2K_NOSPILL=1: the exchange one row block at a time, none).
//     Round 5 saw it abort with a GPU memory fault in one process of ten.  Round 6: NOT the scratch (the no-spill build faults alike, profiles/r6k_*) -- the lost
//     carry described at h2_piece, which every instantiation of this template had in its issue() (the bases of consecutive pieces are formed around the asm
//     statements), the shipped fc.3 one included.  Fixed there; DESIGN.md 4.6.
template <int BM_, int BN_, int AB_, int BB_, int NSUB_> struct H2KCfg {
    static constexpr int BM = BM_, BN = BN_, AB = AB_, BB = BB_, NSUB = NSUB_;
    static constexpr int R = BM + BN, SUBT = R * H2_ROWB, TILE = NSUB * SUBT, LDS = 3 * TILE;
    static constexpr int NA = BM / 32, NW = BN / 32, NCH = NSUB * (NA + NW);      // 1 KB pieces per wave (all eight issue) and K-tile
    static constexpr int KP = H2_KT * NSUB;                                      // k per phase
    static_assert(BM == 4 * 32 * AB && BN == 32 * BB && BB % 2 == 0 && TILE == 48 * 1024 && NCH == 12, "four waves of a group cover BM; a 48 KB tile");
};
using H2KFc3 = H2KCfg<128, 64, 1, 2, 2>;
#if DCE_EXPERIMENTS
using H2KFc0 = H2KCfg<256, 128, 2, 4, 1>;                                 // round 6: back in the experiments build, to trace round 5's intermittent fault
#endif

//   BF16 (the bf16-FC mode's fc.3, option bf16_fc3_ksplit): ONE bf16 term per operand -- a row's 128 bytes are 64 k of it, a phase is NSUB x 64 k,
//   one MFMA per fragment pair, no scales -- on the same schedule; h1 and W2 are the mode's row-major bf16 arrays
template <class Cfg, bool FUSE6, bool BF16 = false>
__global__ __launch_bounds__(512, 2)
void fc_gemm_h2k_kernel(const unsigned short* __restrict__ A2, const int* __restrict__ row_scale, const unsigned short* __restrict__ W2, int sw,
                        const float* __restrict__ bias, float* __restrict__ C,
                        int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2,
                        const float* __restrict__ W6 = nullptr, float* __restrict__ part = nullptr, long long part_rows = 0)
{
    constexpr int BM = Cfg::BM, BN = Cfg::BN, AB = Cfg::AB, BB = Cfg::BB, NSUB = Cfg::NSUB, ROWB = H2_ROWB, SUBT = Cfg::SUBT, TILE = Cfg::TILE;
    static_assert(!FUSE6 || (BM == 128 && BN == FC6_CHUNK && AB == 1 && BB == 2), "the fused fc.6 epilogue is fc.3's 128 x 64 tile");
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
    const int m0 = tm * BM, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;                                // K-tile parity of this wave's group; waves w and w+4 share a SIMD
    const int wm = (wid & 3) * 32 * AB;                      // this wave's rows of the block tile (all BN columns)
    const int i = lane & 31, h = lane >> 5;

    // (pieces: ONE per-lane offset for the A panel and one for the W panel -- a piece's 8 rows start 32 rows behind the previous one's, which moves
    //  the wave-uniform base, not the lanes; rows past M are READ (the operand's buffer is padded by a tile of rows) and never stored)
    const size_t rowb = (size_t)K * (BF16 ? 2 : 4);
    unsigned voffA, voffW;
    {
        const int lr = lane >> 3, slot = lane & 7, w4 = wid & 3;
        const int r = 8 * w4 + lr;                                        // row of piece 0; swz(r + 32 j) = swz(r)
        voffA = voffW = (unsigned)((size_t)r * rowb + 16 * (slot ^ h2_swz(r)));
        asm volatile("" : "+v"(voffA), "+v"(voffW));
    }
    const char* sA = reinterpret_cast<const char*>(A2) + (size_t)m0 * rowb;
    const char* sW = reinterpret_cast<const char*>(W2) + (size_t)n0 * rowb;
    const unsigned lds_wave = h2_lds_addr(h2_smem) + (wid & 3) * 1024;   // piece w' of buffer 0; piece q lands 4 q KB behind it: sub-tile by sub-tile, A rows then W rows
    auto issue = [&](unsigned lds0, size_t ko) {
        const unsigned keep = h2_m0_begin(lds0);
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
            for (int j = 0; j < Cfg::NA; ++j) h2_piece(sA + ko + sub * ROWB + (size_t)j * 32 * rowb, voffA);
#pragma unroll
            for (int j = 0; j < Cfg::NW; ++j) h2_piece(sW + ko + sub * ROWB + (size_t)j * 32 * rowb, voffW);
        }
        h2_m0_end(keep);
    };

    const int swz = h2_swz(i);
    unsigned fa[4], fb[4];                                                // byte offsets of block 0, sub-tile 0, buffer 0, for slot pair c = 2 term + kq (two fp16 terms) / c = kq of 64 k (BF16): logical column 2 c + h
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int o = 16 * ((2 * c + h) ^ swz);
        fa[c] = (wm + i) * ROWB + o;
        fb[c] = (BM + i) * ROWB + o;
        asm volatile("" : "+v"(fa[c]), "+v"(fb[c]));
    }

    h2_f32x16 acc[AB][BB];
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float4 af[NSUB][4][AB], bf[NSUB][4][BB];                              // [sub-tile][slot pair c][block]
    auto load_frags = [&](unsigned bo) {                                  // bo: the buffer's byte offset
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
#pragma unroll
                for (int blk = 0; blk < AB; ++blk) af[sub][c][blk] = *reinterpret_cast<const float4*>(h2_smem + (fa[c] + bo) + sub * SUBT + blk * 32 * ROWB);
#pragma unroll
                for (int blk = 0; blk < BB; ++blk) bf[sub][c][blk] = *reinterpret_cast<const float4*>(h2_smem + (fb[c] + bo) + sub * SUBT + blk * 32 * ROWB);
            }
    };
    auto math = [&]() {
        if constexpr (BF16) {
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int a = 0; a < AB; ++a)
#pragma unroll
                        for (int b = 0; b < BB; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                __builtin_bit_cast(h2_bf16x8, af[sub][c][a]), __builtin_bit_cast(h2_bf16x8, bf[sub][c][b]), acc[a][b], 0, 0, 0);
        } else {
            constexpr int TA[3] = {1, 0, 0}, TB[3] = {0, 1, 0};
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
                for (int kq = 0; kq < 2; ++kq)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int a = 0; a < AB; ++a)
#pragma unroll
                            for (int b = 0; b < BB; ++b)
                                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                    __builtin_bit_cast(h2_f16x8, af[sub][2 * TA[t] + kq][a]), __builtin_bit_cast(h2_f16x8, bf[sub][2 * TB[t] + kq][b]), acc[a][b], 0, 0, 0);
        }
    };

    const int KT = K / (BF16 ? 2 * Cfg::KP : Cfg::KP);                    // even, >= 4 (checked by the launchers); a BF16 phase holds twice the k
    constexpr size_t KSTEP = (size_t)ROWB * NSUB;
    // Phase p = 0 .. KT: group p & 1 issues tile p + 2 and reads tile p's fragments, the other group multiplies tile p - 1.
    //   WAR: the buffer of tile p + 2 held tile p - 1, read in phase p - 1, which ended with lgkmcnt(0) + barrier;
    //   RAW: tile p + 2 is first read in phase p + 2; its issuing group waits vmcnt(0) at the end of its math phase p + 1, ahead of that barrier.
    issue(lds_wave + grp * TILE, (size_t)grp * KSTEP);                    // tiles 0 (group 0) and 1 (group 1)
    if (grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");      // group 1 sits out phase 0 (tile 1 has landed when it ends)
    unsigned bu = grp, bn = grp == 0 ? 2 : 0;                             // buffers of tile u and of tile u + 2
#pragma unroll 1
    for (int u = grp; u < KT; u += 2) {
        // ---- load phase (phase u)
        if (u + 2 < KT) issue(lds_wave + bn * TILE, (size_t)(u + 2) * KSTEP);
        load_frags(bu * TILE);
        __builtin_amdgcn_sched_barrier(0);                                // (see fc_gemm_h2_kernel: the MFMAs stay behind the barrier)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- math phase (phase u + 1)
        math();
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bu = bn; bn = bn == 0 ? 2 : bn - 1;                               // (u + 2) % 3, (u + 4) % 3: 0 -> 2 -> 1 -> 0
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");                 // same number of barriers for both groups

    // ---- the groups exchange halves: group g keeps column blocks g BB/2 .. and sends the others through LDS; then the epilogue on what it keeps.
    //      (G is the group as a compile-time constant: a run-time index into the accumulators would put them in scratch.)
    auto finish = [&](auto gc) {
        constexpr int G = decltype(gc)::value, HB = BB / 2, XF = AB * HB * 16;      // floats per lane that cross
        float* const x = reinterpret_cast<float*>(h2_smem) + ((size_t)wid * XF * 64);      // this wave's area: [XF values][64 lanes]
        const float* const y = reinterpret_cast<const float*>(h2_smem) + ((size_t)(wid ^ 4) * XF * 64);   // the partner wave (same rows, other group)
        if constexpr (!FUSE6 && H2K_NOSPILL) {
            // (round 6, experiments: the exchange and the store ONE row block at a time -- 32 values cross per step instead of 64 and their addresses die
            //  before the next block's are formed: no spill, no scratch; it faulted like the spilling build -- the fault was never the scratch)
#pragma unroll
            for (int a = 0; a < AB; ++a) {
#pragma unroll
                for (int b = 0; b < HB; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) x[((a * HB + b) * 16 + r) * 64 + lane] = acc[a][HB * (1 - G) + b][r];
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
                for (int b = 0; b < HB; ++b) {
                    const int col = n0 + 32 * (HB * G + b) + i;
                    const float bv = bias[col];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                        const float mine = acc[a][HB * G + b][r], theirs = y[((a * HB + b) * 16 + r) * 64 + lane];
                        const float sum = G == 0 ? mine + theirs : theirs + mine;
                        float v = __builtin_ldexpf(sum, BF16 ? 0 : -(row_scale[row < M ? row : M - 1] + sw)) + bv;
                        if (relu) v = v < 0.f ? 0.f : v;
                        if (row < M) C[(size_t)row * N + col] = v;
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < AB; ++a)
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) x[((a * HB + b) * 16 + r) * 64 + lane] = acc[a][HB * (1 - G) + b][r];
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int a = 0; a < AB; ++a)
#pragma unroll
            for (int b = 0; b < HB; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float mine = acc[a][HB * G + b][r], theirs = y[((a * HB + b) * 16 + r) * 64 + lane];
                    acc[a][HB * G + b][r] = G == 0 ? mine + theirs : theirs + mine;       // even tiles + odd tiles, whoever adds
                }
        if constexpr (FUSE6) {
            // ---- fused epilogue (fc.3): scales off, bias + ReLU, h2 tile -> LDS (behind the exchange area), this block's chunk of fc.6's summation
            //      tree -> `part` (fc6_chain.h; as fc_gemm_phased.hip's FUSE6); h2 itself to C only when C != NULL (taps)
            constexpr int HLD = 68;                          // h2 tile [128][68] floats
            float* const ht = reinterpret_cast<float*>(h2_smem + 64 * 1024);
            float4 bw6[4];
            fc6_load_w3(W6, tn, lane, bw6);
            const int col_l = 32 * G + i;
            const float bv = bias[n0 + col_l];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row_l = wm + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int row = m0 + row_l < M ? m0 + row_l : M - 1;
                float v = (BF16 ? acc[0][G][r] : __builtin_ldexpf(acc[0][G][r], -(row_scale[row] + sw))) + bv;
                v = v < 0.f ? 0.f : v;                       // fc.3's ReLU; keeps NaN like torch
                ht[row_l * HLD + col_l] = v;
                if (C && m0 + row_l < M) C[(size_t)(m0 + row_l) * N + n0 + col_l] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const fc6_f32x4 p6 = fc6_chunk_mfma(ht + 16 * wid * HLD, HLD, 0, lane, bw6);   // wave w: rows 16w .. 16w+15
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 16 * wid + 4 * (lane >> 4) + r;
                if (row < M) part[((size_t)tn * part_rows + row) * NCLS + (lane & 15)] = p6[r];
            }
        } else {
            auto store_tile = [&](auto full) {
#pragma unroll
                for (int a = 0; a < AB; ++a) {
                    int ex[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (!decltype(full)::value && row >= M) row = M - 1;
                        ex[r] = BF16 ? 0 : -(row_scale[row] + sw);
                    }
#pragma unroll
                    for (int b = 0; b < HB; ++b) {
                        const int col = n0 + 32 * (HB * G + b) + i;
                        const float bv = bias[col];
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                            float v = __builtin_ldexpf(acc[a][HB * G + b][r], ex[r]) + bv;
                            if (relu) v = v < 0.f ? 0.f : v;
                            if (decltype(full)::value || row < M) C[(size_t)row * N + col] = v;
                        }
                    }
                }
            };
            if (m0 + BM <= M) store_tile(std::true_type{});
            else store_tile(std::false_type{});
        }
    };
    if (grp == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
}

hipError_t init_fc_gemm_h2()
{
    hipError_t e;
    for (const void* k : {reinterpret_cast<const void*>(&fc_gemm_h2_kernel<false>), reinterpret_cast<const void*>(&fc_gemm_h2_kernel<true>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, H2_LDS)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_h2k_kernel<H2KFc3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H2KFc3::LDS)) != hipSuccess) return e;
#if DCE_EXPERIMENTS
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_h2k_kernel<H2KFc0, false>), hipFuncAttributeMaxDynamicSharedMemorySize, H2KFc0::LDS)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_h2k_kernel<H2KFc3, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H2KFc3::LDS)) != hipSuccess) return e;
#endif
    return hipSuccess;
}

int fc_gemm_h2_pad_rows() { return H2_BM; }                       // rows an A operand's buffer holds beyond M (read by ragged tiles of the K-split kernels, never used)

// min_tiles: 256 x 128 tiles the launch must have (fc.0: option h2_min_tiles; fc.3 on this kernel: a chip-filling launch, x3_min_tiles)
bool fc_gemm_h2_ok(int64_t M, int N, int K, int min_tiles)
{
    if (N % H2_BN || K % H2_KT || K < 2 * H2_KT || M <= 0) return false;
    const int nt = N / H2_BN;
    if ((nt & (nt - 1)) != 0) return false;
    if ((size_t)(M > N ? M : N) * K * 4 >= (1ull << 32)) return false;              // per-lane offsets are 32-bit
    return ((M + H2_BM - 1) / H2_BM) * nt >= min_tiles;
}

//   H1 != NULL: h1 leaves as two fp16 terms [row][N / 32][2][32] with its row scales in h1_scale (the operand of the fc.3 kernels below); eW, eB:
//   exponents of sqrt(K) max_n ||W_n||_2 and of max|bias| (fc_gemm_h2_kernel<OUT2>)
hipError_t launch_fc_gemm_h2(const unsigned short* A2, const int* row_scale, const unsigned short* W2, int sw, const float* bias, float* C,
                             int64_t M, int N, int K, int relu, hipStream_t st, unsigned short* H1, int* h1_scale, int eW, int eB)
{
    if (!fc_gemm_h2_ok(M, N, K, 1)) return hipErrorInvalidValue;
    const int mtiles = (int)((M + H2_BM - 1) / H2_BM), ntiles = N / H2_BN;
    int sn_log2 = tune().phased_sn;
    while ((1 << sn_log2) > ntiles) --sn_log2;
    const int sm = 32 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 32;
#if DCE_EXPERIMENTS
    if (!H1 && tune().h2_ksplit && (K / H2_KT) % 2 == 0 && K / H2_KT >= 4) {
        plan_note("fc_h2k_256x128");
        hipLaunchKernelGGL((fc_gemm_h2k_kernel<H2KFc0, false>), dim3(grid), dim3(512), H2KFc0::LDS, st, A2, row_scale, W2, sw, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
        return hipGetLastError();
    }
#endif
    plan_note(H1 ? "fc_h2_256x128_out2" : "fc_h2_256x128");
    if (H1) hipLaunchKernelGGL((fc_gemm_h2_kernel<true>), dim3(grid), dim3(512), H2_LDS, st, A2, row_scale, W2, sw, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2, H1, h1_scale, eW, eB);
    else    hipLaunchKernelGGL((fc_gemm_h2_kernel<false>), dim3(grid), dim3(512), H2_LDS, st, A2, row_scale, W2, sw, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
    return hipGetLastError();
}

// fc.3 (+ReLU) on two-term fp16 operands with fc.6's chunk sums finished in the epilogue: h1 = [M + pad][64][2][32] fp16 with its row scales
// (fc_gemm_h2_kernel<OUT2>), W2p = fc.3's weights in the same layout times 2^sw; part: [8][part_rows][16] chunk sums; h2_out: NULL or (M, 512) fp32
bool fc23_h2_ok(int64_t M)
{   // up to one and a half rounds of 128 x 64 tiles (12288 windows); longer launches put fc.3 on the 256 x 128 kernel + the tail (as the fp32 path does)
    return M > 0 && M <= 12288 && (size_t)(M + H2KFc3::BM) * FC1 * 4 < (1ull << 32);
}
hipError_t launch_fc23_fused_h2(const unsigned short* h1, const int* h1_scale, const unsigned short* W2p, int sw, const float* b2, const float* W3,
                                float* part, int64_t part_rows, float* h2_out, int64_t M, hipStream_t st)
{
    static_assert(H2KFc3::BN == FC6_CHUNK && FC2 / H2KFc3::BN == FC6_NCHUNK && FC1 % (2 * H2KFc3::KP) == 0 && FC1 / H2KFc3::KP >= 4, "one column tile of fc.3 = one chunk of fc.6; an even number of K-tiles");
    if (M <= 0) return hipSuccess;
    const int mtiles = (int)((M + H2KFc3::BM - 1) / H2KFc3::BM), ntiles = FC2 / H2KFc3::BN;
    const int sn_log2 = 2, sm = 32 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 32;
    plan_note("fc23_fused_h2_128x64");
    hipLaunchKernelGGL((fc_gemm_h2k_kernel<H2KFc3, true>), dim3(grid), dim3(512), H2KFc3::LDS, st, h1, h1_scale, W2p, sw, b2, h2_out,
                       (int)M, FC2, FC1, 1, mtiles, ntiles, sn_log2, W3, part, (long long)part_rows);
    return hipGetLastError();
}

// (experiments build, option bf16_fc3_ksplit=1: measured NO faster than fc_gemm_phased.hip's fused 128 x 64 tile -- 17.5 against 17.8 us per 4096
// windows, tools/ab_bf16_fc3.sh: 3.4 us of MFMAs under ~14 us of per-tile chain, first-tile latency, fc.6 epilogue and launch, whatever the schedule)
// the bf16-FC mode's fc.3 + fc.6 chunk sums on the same kernel (one bf16 term per operand): h1 (M + pad, 2048) and W2 (512, 2048)
// row-major bf16; rows of h1 past M are read (the mode's h1 buffer is fp32-sized: twice what its bf16 rows need) and never used
hipError_t launch_fc23_fused_bf16k(const void* h1, const void* W2, const float* b2, const float* W3, float* part, int64_t part_rows, float* h2_out, int64_t M, hipStream_t st)
{
    static_assert(FC1 % (4 * H2KFc3::KP) == 0 && FC1 / (2 * H2KFc3::KP) >= 4, "an even number of 128-k phases");
    if (M <= 0) return hipSuccess;
    const int mtiles = (int)((M + H2KFc3::BM - 1) / H2KFc3::BM), ntiles = FC2 / H2KFc3::BN;
    const int sn_log2 = 2, sm = 32 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 32;
#if DCE_EXPERIMENTS
    plan_note("fc23_fused_bf16k_128x64");
    hipLaunchKernelGGL((fc_gemm_h2k_kernel<H2KFc3, true, true>), dim3(grid), dim3(512), H2KFc3::LDS, st, static_cast<const unsigned short*>(h1), nullptr,
                       static_cast<const unsigned short*>(W2), 0, b2, h2_out, (int)M, FC2, FC1, 1, mtiles, ntiles, sn_log2, W3, part, (long long)part_rows);
    return hipGetLastError();
#else
    (void)h1; (void)W2; (void)b2; (void)W3; (void)part; (void)part_rows; (void)h2_out; (void)st; (void)grid;
    return hipErrorInvalidValue;
#endif
}

}  // namespace dce
