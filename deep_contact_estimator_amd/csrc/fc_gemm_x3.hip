// fc_gemm_x3.hip -- fc.0 (Linear + bias + ReLU, reference src/contact_cnn.py:48-49) at chip-filling sizes with the fp32
// operands carried as THREE bf16 terms each and multiplied on the bf16 matrix pipe (precision DCE_FP32_SPLIT).
//
// gfx950 has no TF32 / xf32, and its fp32 MFMA runs at 1/16 of the bf16 rate (157 TF vs 2.5 PF).  An fp32 number is
// exactly a1 + a2 + a3 with bf16 terms (8 + 8 + 8 significand bits: a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16(a - a1 - a2)),
// so a product a*b is the sum of nine bf16 x bf16 products, each EXACT in the fp32 accumulator of v_mfma_f32_32x32x16_bf16.
// The three smallest (a2 b3, a3 b2, a3 b3 < 2^-32 |ab|) lie below the fp32 rounding error of the sum and are dropped:
//     a b  ~=  a1 b3 + a3 b1 + a2 b2 + a1 b2 + a2 b1 + a1 b1        six MFMAs at 16x the fp32 rate = 2.7x an fp32 MFMA.
// What is left is fp32 accumulation in a different association than the fmaf chain of the fp32 kernels: the logits differ
// from theirs in the last bits (NOT bit-identical; same error against an fp64 evaluation -- tests/test_x3_gpu.py holds this
// mode to the same fp32 tolerance as the fp32 path, tools/check_full_parity.py reports both).  Opt-in; the fp32 MFMA path
// stays the default and the headline.
//
// Kernel: the schedule of fc_gemm_phased.hip (one workgroup per CU, 8 waves = two groups of four a PHASE apart, LDS-DMA,
// math phases that are nothing but MFMAs) re-cut for 48 MFMAs per K-tile:
//   * workgroup tile 256 x 128, wave tile 64 x 64 (2 x 2 blocks of 32 x 32), K-tile = 32 k;
//   * a K-tile in LDS = 3 planes x (256 + 128) rows x 64 B = 72 KB; TWO buffers (144 KB).  The LDS-DMA of tile u+1 is
//     issued by group 0 alone at the start of its load(u) -- the phase from which the buffer of tile u-1 is free -- and
//     is first read two phases later;
//   * per wave and K-tile: 24 ds_read_b128 (3 planes x 4 blocks x 2 k-groups, 96 VGPRs) feed 48 MFMAs = 1536 cycles of
//     the matrix pipe: 0.5 LDS reads per MFMA where the plain bf16 kernel needs 1 (its wave tile re-reads each operand
//     for one MFMA, here each is used three times), and phases 3x as long for the same hand-over;
//   * rows are 64 B = four 16-byte slots; slot s of row r holds logical column s ^ ((r >> 2) & 3): ds_read_b128 serves
//     16 lanes (= rows) a cycle, 256 B over 64 banks, and the four rows of a group that share r & 3 -- the same 64 bytes
//     of the bank space -- then sit in four different slots (SQ_LDS_BANK_CONFLICT 0; (r >> 1) & 3 measured two-way
//     conflicts on every fragment read, and 2 % of the launch).
// Operands in HBM: A3 = three bf16 planes of the (M, K) features written by split3_kernel, W3 = three planes of the (N, K)
// weights split on the host when the weights are finalised; both pair-interleaved (x3_off).
#include "dce_kernels.h"
#include "fc6_chain.h"
#include <cstring>
#include <type_traits>

namespace dce {

typedef float x3_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));

#ifndef X3_ISSUE
#define X3_ISSUE 0           // who brings the K-tiles in: 0 group 0, in a bunch at the start of its load phase; 1 group 1, one piece per
                             // MFMA gap of its math phase (A/B, see the main loop)
#endif
#ifndef X3_EXP
#define X3_EXP 0             // timing probes (WRONG results): 1 no LDS-DMA after the first two tiles, 2 no fragment reads after the first, 4 no barriers' counter waits
#endif
#ifndef X3_TRACE
#define X3_TRACE 0           // debug build: s_memtime at four points of every K-tile, first 32 K-tiles, every wave of block 0
#endif
#if X3_TRACE
__device__ unsigned long long g_x3_trace[8 * 32 * 4];
#ifndef X3_TRACE_BLOCK
#define X3_TRACE_BLOCK 0
#endif
#ifndef X3_TRACE_STRIDE
#define X3_TRACE_STRIDE 1    // record K-tiles 0, S, 2S, ... (32 of them)
#endif
#define X3_MARK(k) do { if (blockIdx.x == X3_TRACE_BLOCK && lane == 0 && u % X3_TRACE_STRIDE == 0 && u / X3_TRACE_STRIDE < 32) \
                            x3_marks[(wid * 32 + u / X3_TRACE_STRIDE) * 4 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define X3_MARK(k) do {} while (0)
#endif

namespace {

// Tile configuration.  fc.0: 256 x 128 with 32-k K-tiles (64-byte rows, planes PAIR-INTERLEAVED in HBM, x3_off); fc.3 (round 5): 128 x 64
// with 64-k K-tiles (128-byte rows = whole lines, planes row-major) -- 256 tiles at 4096 windows, the same 72 KB per K-tile and the
// same eighteen 1 KB pieces per issuing wave.
template <int BM_, int BN_, int KT_> struct X3Cfg {
    static constexpr int BM = BM_, BN = BN_, KT = KT_, ROWS = BM + BN;
    static constexpr int ROWB = 2 * KT;                          // bytes of K per row per plane
    static constexpr int PLANE = ROWS * ROWB, TILE = 3 * PLANE, LDS = 2 * TILE;
    static constexpr int SL = ROWB / 16, CR = 64 / SL;            // 16-byte slots per row; rows per 1 KB chunk
    static constexpr int NCH = TILE / 1024 / 4;                   // 1 KB chunks per ISSUING wave (the four of group 0) per K-tile
    static constexpr int NQ = NCH / 3, NQA = BM / CR / 4;         // chunks per plane per wave; of which from the A panel
    static constexpr int WTM = BM / 4, WTN = BN / 2, AB = WTM / 32, BB = WTN / 32, KQ = KT / 16;
    static constexpr bool ILV = ROWB == 64;                       // pair-interleaved operand planes (x3_off)
    static_assert(TILE == 4 * NCH * 1024 && NCH == 18 && LDS <= 160 * 1024, "four waves x eighteen chunks; two K-tiles in LDS");
    static_assert(NQA * 4 * CR == BM && (NQ - NQA) * 4 * CR == BN, "chunk deal");
    __device__ static constexpr int swz(int r) { return SL == 4 ? (r >> 2) & 3 : (r >> 1) & 7; }
};
using X3Fc0 = X3Cfg<256, 128, 32>;
using X3Fc3 = X3Cfg<128, 64, 64>;
constexpr int X3_BM = X3Fc0::BM, X3_BN = X3Fc0::BN, X3_KT = X3Fc0::KT, X3_LDS = X3Fc0::LDS;

__device__ __forceinline__ unsigned x3_lds_addr(const void* p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}

// round-to-nearest-even fp32 -> bf16 (NaN stays NaN, quiet)
__host__ __device__ __forceinline__ unsigned short x3_bf16(float f)
{
    unsigned u;
#ifdef __HIP_DEVICE_COMPILE__
    u = __float_as_uint(f);
#else
    memcpy(&u, &f, 4);
#endif
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__host__ __device__ __forceinline__ float x3_f32(unsigned short b)
{
    const unsigned u = (unsigned)b << 16;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
// x = t1 + t2 + t3 (+ less than 2^-24 |x|)
__host__ __device__ __forceinline__ void x3_split(float x, unsigned short& t1, unsigned short& t2, unsigned short& t3)
{
    t1 = x3_bf16(x);
    const float r = x - x3_f32(t1);          // exact (Sterbenz-style: t1 is x rounded to 8 bits)
    t2 = x3_bf16(r);
    const float r2 = r - x3_f32(t2);         // exact
    t3 = x3_bf16(r2);
}

// Operand planes in HBM are stored PAIR-INTERLEAVED: element (r, k) of a (rows, K) plane sits at
//     (r >> 1) * 2K + (k >> 5) * 64 + (r & 1) * 32 + (k & 31)
// so that the 64-byte K-tile segments of two neighbouring rows form one 128-byte line: the LDS-DMA's eight lanes per row
// pair then ask L2 for whole lines (row-major planes gave 4.4e7 64-byte requests per fc.0 launch).
__host__ __device__ __forceinline__ size_t x3_off(size_t r, size_t k, size_t K)
{
    return (r >> 1) * 2 * K + (k >> 5) * 64 + (r & 1) * 32 + (k & 31);
}

// one 1 KB piece of a K-tile: 64 lanes x 16 bytes from base + v (per-lane byte offset) to LDS at M0 (+ lane * 16); M0 then
// steps on by 4 KB to the wave's next chunk.  M0 is compiler-reserved: x3_m0_begin saves it and loads the first
// destination, x3_m0_end puts it back; between the two only MFMAs and pieces are issued (sched_barrier keeps it so).
__device__ __forceinline__ unsigned x3_m0_begin(unsigned lds)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1" : "=&s"(keep) : "s"(lds) : "memory");
    return keep;
}
__device__ __forceinline__ void x3_piece(const char* base, unsigned v)
{
    asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %1, %0\n\ts_add_u32 m0, m0, 0x1000" :: "s"(base), "v"(v) : "memory", "scc");
}
__device__ __forceinline__ void x3_m0_end(unsigned keep)
{
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep) : "memory");
}

}  // namespace

// fp32 (rows, cols) -> three bf16 planes [3][rows][cols]; eight elements per thread
#if DCE_EXPERIMENTS
__global__ __launch_bounds__(256)
void split3_kernel(const float* __restrict__ x, unsigned short* __restrict__ planes, size_t n8, size_t plane_elems, int cols)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const size_t r = 8 * i / cols, k = 8 * i % cols;                       // eight consecutive k of one row (cols % 8 == 0)
    const float4 lo = reinterpret_cast<const float4*>(x)[2 * i], hi = reinterpret_cast<const float4*>(x)[2 * i + 1];
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    unsigned short t[3][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) x3_split(v[e], t[0][e], t[1][e], t[2][e]);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        uint4 o;
        o.x = t[p][0] | ((unsigned)t[p][1] << 16); o.y = t[p][2] | ((unsigned)t[p][3] << 16);
        o.z = t[p][4] | ((unsigned)t[p][5] << 16); o.w = t[p][6] | ((unsigned)t[p][7] << 16);
        *reinterpret_cast<uint4*>(planes + p * plane_elems + x3_off(r, k, cols)) = o;
    }
}

// Cfg: tile configuration (X3Fc0 / X3Fc3).  OUT3: C leaves as three bf16 planes, row-major [3][M rounded up to even][N] (fc.0 in front of an
// fc.3 that takes three-term operands) instead of fp32.  FUSE6 (X3Fc3 only): the block's 64 output columns are one chunk of fc.6's
// summation tree (fc6_chain.h), finished in the epilogue exactly as fc_gemm_phased.hip's FUSE6 does -- chunk sums to `part`, h2 itself to
// C only when C != NULL (taps).
#endif

template <class Cfg, bool OUT3, bool FUSE6>
__global__ __launch_bounds__(512, 2)
void fc_gemm_x3_kernel(const unsigned short* __restrict__ A3, const unsigned short* __restrict__ W3,
                       const float* __restrict__ bias, float* __restrict__ C,
                       int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2,
                       const float* __restrict__ W6 = nullptr, float* __restrict__ part = nullptr, long long part_rows = 0)
{
    constexpr int BM = Cfg::BM, BN = Cfg::BN, ROWB = Cfg::ROWB, PLANE = Cfg::PLANE, TILE = Cfg::TILE, NCH = Cfg::NCH, NQ = Cfg::NQ, NQA = Cfg::NQA;
    constexpr int AB = Cfg::AB, BB = Cfg::BB, KQ = Cfg::KQ;
    static_assert(!FUSE6 || (BM == 128 && BN == FC6_CHUNK && AB == 1 && BB == 1 && !OUT3), "the fused fc.6 epilogue is fc.3's 128 x 64 tile");
    extern __shared__ __attribute__((aligned(16))) char x3_smem[];
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
    const int wm = (wid & 3) * Cfg::WTM, wn = grp * Cfg::WTN;    // this wave's corner of the block tile
    const int i = lane & 31, h = lane >> 5;

    // ---- global -> LDS, issued by the four waves of group 0 only (at the start of their load phase, see below).  Wave w' =
    //      wid & 3 brings chunks c = w' + 4 j, j = 0..17: plane j / 6; q = j % 6 < 4: A rows CR (w' + 4 q) .., else W rows
    //      CR (w' + 4 (q - 4)) ..; lane (lr = lane / SL, slot = lane % SL) fills slot `slot` of row lr of the chunk with the
    //      logical column slot ^ swz(row).
    const size_t rowb = (size_t)K * 2;                                   // bytes of one row; (ILV) a row PAIR spans 2 rowb (x3_off)
    const size_t planeA = (size_t)((M + 1) & ~1) * rowb, planeW = (size_t)N * rowb;
    unsigned voff[NCH];
    {
        const int lr = lane / Cfg::SL, slot = lane % Cfg::SL, w4 = wid & 3;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int p = j / NQ, q = j % NQ;
            const int r = Cfg::CR * (w4 + 4 * (q < NQA ? q : q - NQA)) + lr;   // row inside the A panel / the W panel
            const int col = slot ^ Cfg::swz(r);                           // panels start at multiples of 32 rows: swz(r) = swz of the tile row
            if (q < NQA) {
                int grow = r;
                if (m0 + grow >= M) grow = M - 1 - m0;                    // rows past M re-read the last one (never stored); m0 is even
                voff[j] = Cfg::ILV ? (unsigned)(p * planeA + (size_t)(grow >> 1) * 2 * rowb + (grow & 1) * 64 + 16 * col)
                                   : (unsigned)(p * planeA + (size_t)grow * rowb + 16 * col);
            } else {
                voff[j] = Cfg::ILV ? (unsigned)(p * planeW + (size_t)(r >> 1) * 2 * rowb + (r & 1) * 64 + 16 * col)
                                   : (unsigned)(p * planeW + (size_t)r * rowb + 16 * col);
            }
        }
    }
    const char* sA = reinterpret_cast<const char*>(A3) + (size_t)m0 * rowb;
    const char* sW = reinterpret_cast<const char*>(W3) + (size_t)n0 * rowb;
    constexpr size_t KSTEP = Cfg::ILV ? 2 * ROWB : ROWB;                  // bytes from one K-tile to the next in HBM
    const unsigned lds_wave = x3_lds_addr(x3_smem) + (wid & 3) * 1024;    // chunk w' of buffer 0; piece j lands 4 j KB behind it
    auto piece = [&](int j, size_t ko) {                                  // j is a compile-time constant at every call; pieces go in order
        x3_piece((j % NQ < NQA ? sA : sW) + ko, voff[j]);
    };

    // ---- fragment reads: lane (i, h) reads row (corner + 32 blk + i), logical column 2 kq + h  (k = 16 kq + 8 h + 0..7)
    const int sw = Cfg::swz(i);
    unsigned fa[2][KQ], fb[2][KQ];                                        // [buffer][kq]: byte offsets of plane 0, block 0
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const int o = 16 * ((2 * kq + h) ^ sw);
            fa[b][kq] = b * TILE + (wm + i) * ROWB + o;
            fb[b][kq] = b * TILE + (BM + wn + i) * ROWB + o;
            asm volatile("" : "+v"(fa[b][kq]), "+v"(fb[b][kq]));          // stay in registers (see fc_gemm_phased.hip)
        }

    x3_f32x16 acc[AB][BB];
#pragma unroll
    for (int a = 0; a < AB; ++a)
#pragma unroll
        for (int b = 0; b < BB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // fragments of one K-tile: [kq][plane][block] x 16 bytes per lane, A and W
    float4 af[KQ][3][AB], bf[KQ][3][BB];
    auto load_frags = [&](int buf) {                                      // buf is a compile-time constant at every call
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int blk = 0; blk < AB; ++blk) af[kq][p][blk] = *reinterpret_cast<const float4*>(x3_smem + fa[buf][kq] + p * PLANE + blk * 32 * ROWB);
#pragma unroll
                for (int blk = 0; blk < BB; ++blk) bf[kq][p][blk] = *reinterpret_cast<const float4*>(x3_smem + fb[buf][kq] + p * PLANE + blk * 32 * ROWB);
            }
    };
    // the MFMAs of a K-tile.  X3_ISSUE == 1: with `deal` (wave-uniform) the eighteen pieces of a later K-tile go out one
    // behind each of MFMAs 1 .. 18 -- ONE instruction stream for both cases (two copies of the MFMA chain behind a branch
    // made hipcc shuttle the 64 accumulator registers between them with v_mov_b64 every K-tile).
    auto math = [&](bool deal, unsigned lds0, size_t ko) {
        // six terms per block pair, small ones first; consecutive MFMAs go to different accumulators
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#if X3_ISSUE == 1
        unsigned keep = 0;
        if (deal) keep = x3_m0_begin(lds0);
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < AB; ++a)
#pragma unroll
                    for (int b = 0; b < BB; ++b) {
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                            __builtin_bit_cast(x3_bf16x8, af[kq][TA[t]][a]), __builtin_bit_cast(x3_bf16x8, bf[kq][TB[t]][b]), acc[a][b], 0, 0, 0);
#if X3_ISSUE == 1
                        const int n = ((kq * 6 + t) * AB + a) * BB + b;
                        if (n < NCH) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (deal && !(X3_EXP & 1)) piece(n, ko);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (n == NCH) {
                            if (deal) x3_m0_end(keep);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#endif
                    }
        (void)deal; (void)lds0; (void)ko;
    };
    // the eighteen pieces of one K-tile, in a bunch
    auto issue = [&](unsigned lds0, size_t ko) {
        const unsigned keep = x3_m0_begin(lds0);
#pragma unroll
        for (int j = 0; j < NCH; ++j) if (!(X3_EXP & 1) || ko < 2 * KSTEP) piece(j, ko);
        x3_m0_end(keep);
    };

#if X3_TRACE
    unsigned long long* x3_marks = reinterpret_cast<unsigned long long*>(x3_smem + Cfg::LDS);      // 8 KB behind the two tiles
#endif
    const int KT = K / Cfg::KT;                                           // >= 2 (checked by the launcher)
    // Phases p = 0, 1, 2, ...; a workgroup barrier ends each.
    //   group 0: load(u) in phase 2u, math(u) in 2u+1            group 1: load(u) in 2u+1, math(u) in 2u+2
    // Tile u lives in buffer u & 1 and is read in phases 2u and 2u+1.  Group 0 issues ALL of tile u+1 at the start of its
    // load(u), phase 2u (u >= 1; tiles 0 and 1 up front): a load phase is 24 LDS reads against the other group's 48 MFMAs,
    // so the 18 pieces ride in its slack.  (Measured alternatives: every wave issuing 9 pieces at the start of phase 2u --
    // group 1 then stands ~350 cycles in front of its MFMAs; group 1 issuing one piece per MFMA gap of its math phase --
    // its 48 MFMAs take 2250 cycles instead of 1790.)
    //   WAR: the buffer's previous tile u-1 was last read in phase 2u-1, which ended with lgkmcnt(0) + barrier;
    //   RAW: first read in phase 2u+2; group 0 waits vmcnt(0) at the end of its math(u), phase 2u+1, ahead of that barrier.
    // Both groups run the same instruction stream (group 1 one barrier behind); only the issue and the counter waits differ,
    // behind wave-uniform branches.
    if (grp == (X3_ISSUE == 1 ? 1 : 0)) {
        issue(lds_wave, 0);
        issue(lds_wave + TILE, KSTEP);
        asm volatile("s_waitcnt vmcnt(18)" ::: "memory");                 // tile 0 landed
    }
    asm volatile("s_barrier" ::: "memory");
    if (grp == 1) asm volatile("s_barrier" ::: "memory");                 // group 1 sits out phase 0
    auto ktile = [&](int u, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#if X3_ISSUE == 1
        // (A/B) group 1 deals tile u+2 out over its math(u), phase 2u+2, and waits for it at the end of its load(u+1)
        X3_MARK(0);
        if (!(X3_EXP & 2) || u == 0) load_frags(buf);
        if (grp == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        X3_MARK(1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        X3_MARK(2);
        math(grp == 1 && u + 2 < KT, lds_wave + buf * TILE, (size_t)(u + 2) * KSTEP);
        X3_MARK(3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
        // ---- load phase
        if (grp == 0 && u >= 1 && u + 1 < KT) issue(lds_wave + (buf ^ 1) * TILE, (size_t)(u + 1) * KSTEP);
        X3_MARK(0);
        if (!(X3_EXP & 2) || u == 0) load_frags(buf);
        X3_MARK(1);
        // (sched_barrier, round 5: MFMAs carry no memory dependence, and without it hipcc of ROCm 7.2 moves two thirds of the math phase up between the
        //  fragment reads of the load phase, across the barrier -- found in the ISA while building fc_gemm_h2.hip, which inherited it; fc_gemm_phased.hip
        //  has carried the same fence since round 2)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // ---- math phase
        X3_MARK(2);
        math(false, 0, 0);
        X3_MARK(3);
        __builtin_amdgcn_sched_barrier(0);
        if (grp == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
#pragma unroll 1
    for (int u = 0; u < KT; u += 2) {
        ktile(u, std::integral_constant<int, 0>{});
        if (u + 1 < KT) ktile(u + 1, std::integral_constant<int, 1>{});
    }
    if (grp == 0) asm volatile("s_barrier" ::: "memory");                 // same number of barriers for both groups
#if X3_TRACE
    if (blockIdx.x == X3_TRACE_BLOCK && lane == 0)
        for (int q = 0; q < 32 * 4; ++q) g_x3_trace[wid * 128 + q] = x3_marks[wid * 128 + q];
#endif

    if constexpr (FUSE6) {
        // ---- fused epilogue (fc.3): bias + ReLU, h2 tile -> LDS, this block's chunk of fc.6's summation tree -> `part`.  Every wave is
        //      past its last fragment read (the barrier above), LDS is free.
        constexpr int HLD = 68;                          // h2 tile [128][68] floats
        float* ht = reinterpret_cast<float*>(x3_smem);
        float4 bw6[4];
        fc6_load_w3(W6, tn, lane, bw6);
        const int col_l = wn + i;
        const float bv = bias[n0 + col_l];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row_l = wm + (r & 3) + 8 * (r >> 2) + 4 * h;
            float v = acc[0][0][r] + bv;
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

    // ---- epilogue: bias + (ReLU); D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    auto store_tile = [&](auto full) {
#pragma unroll
        for (int b = 0; b < BB; ++b) {
            const int col = n0 + wn + 32 * b + i;
            const float bv = bias[col];
#pragma unroll
            for (int a = 0; a < AB; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float v = acc[a][b][r] + bv;
                    if (relu) v = v < 0.f ? 0.f : v;                      // keeps NaN like torch
                    if (decltype(full)::value || row < M) {
                        if constexpr (OUT3) {                             // three bf16 planes, row-major: the next layer's three-term operand
                            unsigned short t1, t2, t3;
                            x3_split(v, t1, t2, t3);
                            unsigned short* const p0 = reinterpret_cast<unsigned short*>(C);        // (wave-uniform base + 32-bit element offsets: one address register per store)
                            const unsigned pl = (unsigned)((M + 1) & ~1) * (unsigned)N, idx = (unsigned)row * (unsigned)N + (unsigned)col;
                            p0[idx] = t1; p0[pl + idx] = t2; p0[2 * pl + idx] = t3;
                        } else C[(size_t)row * N + col] = v;
                    }
                    if constexpr (OUT3) __builtin_amdgcn_sched_barrier(0);  // (one value at a time: left alone hipcc splits all 64 at once and spills)
                }
        }
    };
    if (m0 + BM <= M) store_tile(std::true_type{});
    else store_tile(std::false_type{});
}

// Round 6: DCE_FP32_SPLIT left the product library (fp32_f16x2 holds the same contract at 1.13 - 2.0 x its speed at every size, profiles/r6h_retire_split_sweep.txt):
// the kernels of this file are instantiated in the experiments build only; the product's launchers refuse.
hipError_t init_fc_gemm_x3()
{
#if !DCE_EXPERIMENTS
    return hipSuccess;
#else
    hipError_t e;
    for (const void* k : {reinterpret_cast<const void*>(&fc_gemm_x3_kernel<X3Fc0, false, false>), reinterpret_cast<const void*>(&fc_gemm_x3_kernel<X3Fc0, true, false>),
                          reinterpret_cast<const void*>(&fc_gemm_x3_kernel<X3Fc3, false, true>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, X3_LDS + (X3_TRACE ? 8192 : 0))) != hipSuccess) return e;
    return hipSuccess;
#endif
}

// 256 x 128 tiles must fill the chip (as the phased fp32 kernel asks of its large tile)
bool fc_gemm_x3_ok(int64_t M, int N, int K)
{
    if (!DCE_EXPERIMENTS) return false;
    if (N % X3_BN || K % X3_KT || K < 2 * X3_KT || M <= 0) return false;
    const int nt = N / X3_BN;
    if ((nt & (nt - 1)) != 0) return false;
    if (3ull * (size_t)(M > N ? M : N) * K * 2 >= (1ull << 32)) return false;       // per-lane offsets are 32-bit, planes included
    return ((M + X3_BM - 1) / X3_BM) * nt >= tune().x3_min_tiles;
}

// fc.3 + fc.6 chunk sums on three-term operands: where the fp32 path fuses them (one round of 128 x 64 tiles) and the planes' offsets fit
bool fc23_x3_ok(int64_t M)
{
    return DCE_EXPERIMENTS && tune().x3_fc3 && fc23_fused_ok(M, 0) && 3ull * (size_t)((M + 1) & ~(int64_t)1) * FC1 * 2 < (1ull << 32);
}

// rows x cols fp32 -> three ROW-MAJOR planes [3][rows][cols] (fc.3's weights: its 64-k K-tiles are whole 128-byte lines as they are)
void split3_rows_host(const float* x, size_t rows, size_t cols, unsigned short* planes)
{
    const size_t n = rows * cols;
    for (size_t e = 0; e < n; ++e) x3_split(x[e], planes[e], planes[n + e], planes[2 * n + e]);
}

// rows x cols fp32 -> three pair-interleaved planes (x3_off); rows even.  cols = 0: flat (planes[p][k], the test hook)
void split3_host(const float* x, size_t rows, size_t cols, unsigned short* planes)
{
    const size_t n = rows * (cols ? cols : 1);
    for (size_t r = 0; r < rows; ++r)
        for (size_t k = 0; k < (cols ? cols : 1); ++k) {
            const size_t src = cols ? r * cols + k : r, dst = cols ? x3_off(r, k, cols) : r;
            x3_split(x[src], planes[dst], planes[n + dst], planes[2 * n + dst]);
        }
}

hipError_t launch_split3(const float* x, unsigned short* planes, int64_t rows, int cols, hipStream_t st)
{
    const size_t n = (size_t)rows * cols;
    if (n == 0) return hipSuccess;
#if DCE_EXPERIMENTS
    if (cols % 32) return hipErrorInvalidValue;
    plan_note("split3");
    // plane stride = an even number of rows (the kernel pads M to even)
    hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n / 8 + 255) / 256)), dim3(256), 0, st, x, planes, n / 8,
                       (size_t)((rows + 1) & ~(int64_t)1) * cols, cols);
    return hipGetLastError();
#else
    (void)x; (void)planes; (void)st;
    return hipErrorInvalidValue;
#endif
}

hipError_t launch_fc_gemm_x3(const unsigned short* A3, const unsigned short* W3, const float* bias, void* C,
                             int64_t M, int N, int K, int relu, hipStream_t st, int out_planes)
{
    if (!fc_gemm_x3_ok(M, N, K)) return hipErrorInvalidValue;
#if !DCE_EXPERIMENTS
    (void)A3; (void)W3; (void)bias; (void)C; (void)relu; (void)st; (void)out_planes;
    return hipErrorInvalidValue;
#else
    const int mtiles = (int)((M + X3_BM - 1) / X3_BM), ntiles = N / X3_BN;
    int sn_log2 = tune().phased_sn;
    while ((1 << sn_log2) > ntiles) --sn_log2;
    const int sm = 32 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 32;
    plan_note("fc_x3_256x128");
    float* Cf = static_cast<float*>(C);
    if (out_planes) hipLaunchKernelGGL((fc_gemm_x3_kernel<X3Fc0, true, false>), dim3(grid), dim3(512), X3_LDS + (X3_TRACE ? 8192 : 0), st, A3, W3, bias, Cf, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
    else            hipLaunchKernelGGL((fc_gemm_x3_kernel<X3Fc0, false, false>), dim3(grid), dim3(512), X3_LDS + (X3_TRACE ? 8192 : 0), st, A3, W3, bias, Cf, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
    return hipGetLastError();
#endif
}

// fc.3 (+ReLU) on three-term operands with fc.6's chunk sums finished in the epilogue: h1p = three row-major bf16 planes [3][M even][2048]
// (fc_gemm_x3's OUT3 epilogue), W2p = three planes [3][512][2048] (split3_rows_host); part: [8][part_rows][16] chunk sums; h2_out: NULL or (M,512) fp32
hipError_t launch_fc23_fused_x3(const unsigned short* h1p, const unsigned short* W2p, const float* b2, const float* W3, float* part, int64_t part_rows,
                                float* h2_out, int64_t M, hipStream_t st)
{
    static_assert(X3Fc3::BN == FC6_CHUNK && FC2 / X3Fc3::BN == FC6_NCHUNK && FC1 % X3Fc3::KT == 0, "one column tile of fc.3 = one chunk of fc.6");
    if (M <= 0) return hipSuccess;
#if !DCE_EXPERIMENTS
    (void)h1p; (void)W2p; (void)b2; (void)W3; (void)part; (void)part_rows; (void)h2_out; (void)st;
    return hipErrorInvalidValue;
#else
    const int mtiles = (int)((M + X3Fc3::BM - 1) / X3Fc3::BM), ntiles = FC2 / X3Fc3::BN;
    const int sn_log2 = 2, sm = 32 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 32;
    plan_note("fc23_fused_x3_128x64");
    hipLaunchKernelGGL((fc_gemm_x3_kernel<X3Fc3, false, true>), dim3(grid), dim3(512), X3Fc3::LDS + (X3_TRACE ? 8192 : 0), st, h1p, W2p, b2, h2_out,
                       (int)M, FC2, FC1, 1, mtiles, ntiles, sn_log2, W3, part, (long long)part_rows);
    return hipGetLastError();
#endif
}

}  // namespace dce

// test hook (tests/test_x3_gpu.py): the host split, as fc.0's weights get it
extern "C" void dce_debug_split3(const float* x, size_t n, unsigned short* planes) { dce::split3_host(x, n, 0, planes); }

#if X3_TRACE
extern "C" int dce_debug_x3_trace_read(unsigned long long* out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_x3_trace), sizeof(unsigned long long) * 8 * 32 * 4);
}
#endif
