// conv_x3p.hip -- the conv stack on three-term bf16 operands (conv_x3.hip has the arithmetic; reference src/contact_cnn.py:10-44,61,64
// and utils/data_handler.py:55-56) for chip-filling batches: one persistent workgroup of EIGHT waves per CU works on TWO windows, and
// every wave carries one window's non-MFMA work inside the other window's MFMA stream.
//
// Why (profiles/r4a_micro_bf16_mfma_valu.txt, r4a_trace_phase_shifted.txt, r4b_trace_one_window_ksteps.txt; gfx950):
//   * beside a wave that issues v_mfma_f32_16x16x32_bf16 back to back, the OTHER wave of the SIMD gets one VALU instruction in 8-13
//     cycles: a write-back phase next to a conv phase crawls, whether the two waves belong to two free-running workgroups (conv_x3.hip:
//     86k cycles per window pair for 57.6k cycles of MFMAs) or to two groups of one workgroup held a fixed number of phases apart (first
//     cut of this file: 80-85k);
//   * a wave's own VALU work between its own MFMAs is hidden (three instructions per MFMA), and with the fragment requests dealt out
//     between the MFMAs two such waves keep a SIMD's matrix pipe full over the K-steps of a layer;
//   * but one window's layers depend on each other: with one window per workgroup (second cut: ping-pong LDS buffers, a tile's
//     write-back behind its last MFMA) the write-backs, layer starts and barriers stay exposed -- 43.7k cycles per window.
// So a wave holds the accumulators of BOTH windows (2 x 5 tiles of 16 x 16: stage 1 row tile wv & 3, column tiles 5 (wv >> 2)..;
// stage 2 row tile wv) and a phase is one layer's K loop for one window (30 MFMAs per K-step, 3 weight + 15 activation fragments)
// with the other window's pending write-back -- ReLU, MaxPool, split into three terms, LDS stores, in place: its layer was read to
// the end in the phase before -- dealt out over the K-steps, one tile per step.  X runs three phases ahead of Y:
//
//   phase            1         2         3               4         5         6         7         8
//   MFMAs            conv1 X   conv4 Y-  conv2 X         conv1 Y   conv3 X   conv2 Y   conv4 X   conv3 Y
//   rides along      store3 Y- store1 X  features Y-     store2 X  store1 Y  store3 X  store2 Y  features X
//                                        + prologue Y                                            + prologue X+
//
// (Y- = the previous pair's second window, X+ = the next pair's first), one barrier per phase.  The next window of a slot arrives by
// LDS-DMA (global_load_lds_dwordx4, 32 pieces of 1 KB) in a staging buffer the two slots use alternately, and its prologue --
// z-score, split, stores into the slot's planes, which nobody reads then -- rides in phase 3 / 8 like a write-back.  The features
// leave straight from the accumulators, in the order k' = t' * 128 + c (a lane holds four consecutive channels of one pooled
// position: one 8-byte store per tile and plane) instead of the reference's flatten order k = c * 37 + t'; fc.0's weights for this
// path are stored with their K axis permuted the same way (dce_finalize_weights), which changes no product, only the order of an
// fp32-grade / bf16-input summation that claims no bit pattern (DCE_FP32_SPLIT, DCE_BF16_FC).
// The pipeline fills and drains on windows that do not exist (three phases per launch and workgroup): their results go nowhere.
// LDS: 2 x 62,976 B of activation planes + 32 KB staging + flags = 158.8 KB: one workgroup per CU, two waves per SIMD.
#include "conv_x3_common.h"
#include <cfloat>

#if DCE_EXPERIMENTS

namespace dce {

#if DCE_TRACE
// debug build: per workgroup, wave 0's clock at the start of every phase of its last window pair and in front of every barrier
// (tools/trace_conv_x3p.py)
static __device__ unsigned long long g_trace_p[1024 * 16];
#ifndef CXP_TRACE_Q
#define CXP_TRACE_Q -1                                                // which window pair of a workgroup is traced (-1: every pair, i.e. the last one stays)
#endif
#define CXP_T(k) do { if (lane0 == 0 && wv == 0 && blockIdx.x < 1024 && (CXP_TRACE_Q < 0 || q == CXP_TRACE_Q + ((k) < 6 ? 1 : 0))) g_trace_p[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define CXP_T(k) do {} while (0)
#endif
#if DCE_TRACE == 2
// ... and with -DDCE_TRACE=2 every wave's clock at the K-step boundaries of one phase (slots [wave][0..15]; tools/trace_conv_x3p_k.py)
static __device__ unsigned long long g_trace_k[256 * 8 * 16];
#define CXP_TK(k) do { if (KTRACE && __lane_id() == 0 && blockIdx.x < 256) g_trace_k[(blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define CXP_TK(k) do {} while (0)
#endif

#ifndef CXP_PRIO
#define CXP_PRIO 1       // a wave's issue priority falls as it advances through a layer's K-steps (A/B: -DCXP_PRIO=0)
#endif
#ifndef CXP_EXP
#define CXP_EXP 0        // timing probes (WRONG results): 2 no side work (write-backs, prologues, next layer's first weights), 4 no feature stores, 8 no prologue steps after the first window, 16 no write-back stages
#endif

namespace {

constexpr int CXP_RAW = 32 * 1024;                                // staging of one raw window (32,400 B), 32 pieces of 1 KB
constexpr int CXP_OFF_B = CX_LDS, CXP_OFF_RAW = 2 * CX_LDS, CXP_OFF_FLAG = CXP_OFF_RAW + CXP_RAW;
constexpr int CXP_LDS = CXP_OFF_FLAG + 64;
static_assert(CXP_LDS <= 160 * 1024, "one workgroup per CU");
constexpr int WIN_BYTES = WIN * CH * 4;
// A lane that has nothing to store (a column past the layer's end, the odd column of a pool pair, a row past the window) stores
// to this offset of its plane instead of branching around the store -- a branch would cut the K-step's scheduling region in two.
// It is row 162 of the 128-byte layouts (past the 162 rows a stage reads) and row 81 of the 256-byte layout (read only by columns
// whose outputs are dropped).
constexpr int CXP_DUMP = 81 * 256;
static_assert(CXP_DUMP >= CX_ROWS1 * 128 && CXP_DUMP + 128 <= CX_PLANE, "");

__device__ __forceinline__ unsigned cxp_lds_addr(const void* p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}

// this wave's LDS stores are done, then the workgroup barrier (requests to global memory stay in flight)
__device__ __forceinline__ void cxp_barrier()
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void cxp_barrier_dma()                 // ... and everything this wave requested has landed (LDS-DMA)
{
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// a wave's share of a window's LDS-DMA: pieces 4 wave .. 4 wave + 3 (1 KB each: 64 lanes x 16 bytes, LDS address = M0 + 16 lane).
// The last piece is cut at the window's end: its surplus lanes re-read the last 16 bytes (into staging nobody reads).
// M0 is compiler-reserved: saved and restored inside each statement.
__device__ __forceinline__ void cxp_issue_dma(unsigned lds_dst, const char* gsrc, unsigned voff)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned off = voff + k * 1024;
        off = off < (unsigned)(WIN_BYTES - 16) ? off : (unsigned)(WIN_BYTES - 16);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_dst + k * 1024), "v"(off), "s"(gsrc) : "memory");
    }
}

// A tile's way out of the accumulators, cut into six stages so that each can follow one MFMA of the tile behind it (the compiler
// would otherwise issue a tile's 50 instructions in one run behind that tile's MFMAs): 0 ReLU (+ MaxPool over the column pair held
// by lanes j, j ^ 1), 1-3 the three terms (v_cvt_pk_bf16_f32 and an exact subtraction each), 4-5 the stores.
// v - t for the exact remainders of the split: plain v_sub_f32 (written as an instruction: the optimiser re-packs two neighbouring
// fp32 subtractions into v_pk_add_f32, which holds a SIMD's issue port four times as long next to MFMAs --
// profiles/r4c_micro_bf16_mfma_mix.txt)
__device__ __forceinline__ float cxp_sub(float a, float b)
{
    float r;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// three terms of two values (cx_split2 with scalar subtractions): p[k] = (term k of v0) | (term k of v1) << 16
__device__ __forceinline__ void cxp_split2(float v0, float v1, unsigned (&p)[3])
{
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const b2 t = __builtin_convertvector(f2{v0, v1}, b2);
        p[k] = __builtin_bit_cast(unsigned, t);
        if (k < 2) { v0 = cxp_sub(v0, __builtin_bit_cast(float, p[k] << 16)); v1 = cxp_sub(v1, __builtin_bit_cast(float, p[k] & 0xffff0000u)); }
    }
}

struct CxpTail {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    f2 ra, rb;                                                        // the four values, then their remainders
    char* at;                                                         // where the tile's next term goes (set with the first term)
    // (a stage's results are pinned where the stage stands: sched_barrier orders machine instructions, but before that the
    //  optimiser sinks side-effect-free arithmetic down to its first use)
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(ra), "+v"(rb)); }
    template <bool POOL> __device__ __forceinline__ void relu(const cx_f32x4& a)
    {
        float v[4] = {a[0], a[1], a[2], a[3]};                        // (the bias is the accumulators' initial value)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = POOL ? fmaxf(fmaxf(v[r], cx_neighbour(v[r])), 0.f) : fmaxf(v[r], 0.f);
        ra = f2{v[0], v[1]}; rb = f2{v[2], v[3]};
        pin();
    }
    __device__ __forceinline__ uint2 term(bool more)                  // the next term of the four values; `more`: the remainders stay in ra / rb
    {
        const unsigned lo = __builtin_bit_cast(unsigned, __builtin_convertvector(ra, b2)), hi = __builtin_bit_cast(unsigned, __builtin_convertvector(rb, b2));
        if (more) {
            ra = f2{cxp_sub(ra.x, __builtin_bit_cast(float, lo << 16)), cxp_sub(ra.y, __builtin_bit_cast(float, lo & 0xffff0000u))};
            rb = f2{cxp_sub(rb.x, __builtin_bit_cast(float, hi << 16)), cxp_sub(rb.y, __builtin_bit_cast(float, hi & 0xffff0000u))};
            pin();
        }
        return make_uint2(lo, hi);
    }
};

// stages of a layer's tile -> three-term planes of the next layer's input (LDS buffer `out`): 0 ReLU / pool, 1-3 a term and its store
//   co: first of the lane's four channels; t: the lane's column;  T: columns of this layer; POOL: the next layer sees T / 2 positions
template <int ROWB_OUT, bool POOL, int T>
__device__ __forceinline__ void cxp_store_stage(CxpTail& tl, int stage, char* __restrict__ out, const cx_f32x4& a, int co, int t, int j)
{
    if (stage == 0) tl.relu<POOL>(a);
    else if (stage <= 3) {
        if (stage == 1) {
            const bool ok = POOL ? ((j & 1) == 0 && (t >> 1) < T / 2) : t < T;
            const int row = (POOL ? (t >> 1) : t) + 1;
            const int at = cx_addr<ROWB_OUT>(row, co);                   // (both arms computed first: a select, not a branch)
            tl.at = out + (ok ? at : CXP_DUMP);
            asm volatile("" : "+v"(tl.at));                              // (computed here, once per tile -- not hoisted to the top of the phase for all five)
        }
        *reinterpret_cast<uint2*>(tl.at + (stage - 1) * CX_PLANE) = tl.term(stage < 3);
    }
}

// One phase for one wave: acc[ct] = bias + sum over K-steps s = (channel block kb, tap) of W(s) x X(ct, s), with `side(s)` -- the
// other window's pending work -- dealt out over the K-steps.
//   ROWB : bytes per LDS row of the layer's input (2 x input channels)      NKB : 32-channel blocks of K
//   xrow : in + (16 ct0 + j) * ROWB  (this lane's row of column tile 0, tap 0)     sw[tap] = swz(16 ct0 + j + tap)
//   wp   : this wave's row tile of the packed weights (+ lane): fragment (step s, plane p) at wp[(6 s + p) * 64]
//   apre : the weight fragments of step 0, requested by the caller ahead of the barrier in front of this phase
template <int ROWB, int NKB, bool KTRACE = false, class Side>
__device__ __forceinline__ void cxp_layer(const char* __restrict__ xrow, const int (&sw)[3], int g, const uint4* __restrict__ wp,
                                          const uint4 (&apre)[3], const float4& bias, cx_f32x4 (&acc)[CX_NT], Side&& side)
{
    constexpr int S = 3 * NKB;
    // activation fragments (LDS): requested one K-step ahead, two buffers; weight fragments (L2, a K-step of 30 MFMAs is shorter
    // than a loaded L2's answer): two K-steps ahead, three buffers
    uint4 af[3][3], bf[2][CX_NT][3];                                  // [buffer][...][plane]
    auto fetch_w = [&](int s) {                                       // s compile-time at every call
#pragma unroll
        for (int p = 0; p < 3; ++p) af[s % 3][p] = s == 0 ? apre[p] : wp[(s * 6 + p) * 64];
    };
    auto fetch_x = [&](int s) {
        const int kb = s / 3, tap = s % 3;
        const char* x = xrow + tap * ROWB + (((4 * kb + g) ^ sw[tap]) << 4);
#pragma unroll
        for (int ct = 0; ct < CX_NT; ++ct)
#pragma unroll
            for (int p = 0; p < 3; ++p) bf[s & 1][ct][p] = *reinterpret_cast<const uint4*>(x + ct * 16 * ROWB + p * CX_PLANE);
    };
    // six terms per product, small ones first
    constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
    for (int ct = 0; ct < CX_NT; ++ct) acc[ct] = cx_f32x4{bias.x, bias.y, bias.z, bias.w};
    CXP_TK(0);
    fetch_w(0); fetch_w(1); fetch_x(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < S; ++s) {
        CXP_TK(1 + s);
        // The SIMD's issue port goes to the older of its two waves whenever both have an instruction ready: left alone, waves
        // 0..3 run ahead, finish a phase 2.5k cycles early and leave waves 4..7 to finish alone at a single wave's rate.  A
        // priority that falls with the K-step lets the wave that is behind win the port instead.
        if (CXP_PRIO) switch (3 - (4 * s) / S) {                      // (the builtin wants a literal; the switch folds once the loop is unrolled)
            case 3: __builtin_amdgcn_s_setprio(3); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
        }
        if (s + 2 < S) fetch_w(s + 2);
        if (s + 1 < S) fetch_x(s + 1);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int ct = 0; ct < CX_NT; ++ct)                        // consecutive MFMAs go to different accumulators
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cx_bf16x8, af[s % 3][TA[t]]),
                                                                  __builtin_bit_cast(cx_bf16x8, bf[s & 1][ct][TB[t]]), acc[ct], 0, 0, 0);
        if (!(CXP_EXP & 2)) side(s);
        // the order of this K-step's region: an MFMA, one of the 18 fragment requests (weights first: L2), and what there is of
        // the side work
        if (s + 2 < S) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
        if (s + 1 < S) {
#pragma unroll
            for (int i = 0; i < 3 * CX_NT; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 30 - (s + 2 < S ? 3 : 0) - (s + 1 < S ? 3 * CX_NT : 0); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (CXP_PRIO) __builtin_amdgcn_s_setprio(0);
    CXP_TK(1 + S);
}

// the six stages of tile `tile` of a pending write-back, dealt out over a phase's K-steps: one tile per step when the phase has
// 6 K-steps, half a tile when it has 12
template <int S, class Stage> __device__ __forceinline__ void cxp_deal(int s, Stage&& stage)
{
    if constexpr (S == 6) {
        if (s < CX_NT) {
#pragma unroll
            for (int st = 0; st < 6; ++st) stage(s, st);
        }
    } else {
        if (s < 2 * CX_NT) {
#pragma unroll
            for (int st = 0; st < 3; ++st) stage(s >> 1, 3 * (s & 1) + st);
        }
    }
}

}  // namespace

// OUT: 0 the features leave as three bf16 planes (fc_gemm_x3.hip's pair-interleaved layout), 2 as (n, 4736) bf16 (term 1 = the
// value rounded to nearest-even); both in the order k' = t' * 128 + c.
template <bool ZS, int OUT>
__global__ __launch_bounds__(512, 2)
void conv_x3p_kernel(const float* __restrict__ src, int64_t n, ConvPackX3 pk, unsigned short* __restrict__ feat, size_t plane_elems,
                     unsigned short* __restrict__ dump)
{
    extern __shared__ __attribute__((aligned(16))) char cxp_lds[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const bufX = cxp_lds;
    char* const bufY = cxp_lds + CXP_OFF_B;
    const char* const raw = cxp_lds + CXP_OFF_RAW;
    int* const flags = reinterpret_cast<int*>(cxp_lds + CXP_OFF_FLAG);         // non-finite window: [slot X / Y][parity of the pair's index in this workgroup]
    const int64_t win_stride = ZS ? CH : WIN * CH;                             // floats between consecutive windows
    const int64_t npairs = (n + 1) >> 1;
    const int Q = (int)((npairs - blockIdx.x + gridDim.x - 1) / gridDim.x);    // window pairs of this workgroup (>= 1): pair blockIdx.x + gridDim.x q = windows 2 pair, 2 pair + 1
    const unsigned dma_dst = cxp_lds_addr(raw) + wv * 4096, dma_voff = wv * 4096 + lane0 * 16;
    auto window_of = [&](int q, int slot) { return 2 * ((int64_t)blockIdx.x + (int64_t)gridDim.x * q) + slot; };
    auto window_src = [&](int q, int slot) {                                   // (a window that does not exist -- odd n, pipeline fill / drain -- reads the last one)
        int64_t w = window_of(q, slot);
        w = w < n ? w : n - 1;
        return reinterpret_cast<const char*>(src + w * win_stride);
    };
    // Every phase derives its lane-dependent addresses from an opaque copy of the lane id: the loop's body is the same in every
    // iteration, and left alone the compiler hoists the address arithmetic of ALL phases out of the loop (1.5 KB of scratch).
#define CXP_LANE int lane = lane0; asm volatile("" : "+v"(lane)); const int j = lane & 15, g = lane >> 4; (void)j; (void)g

    // The prologue of a window, staged in `raw`, into its slot's buffer as three-term planes [t + 1][channel]: 150 rows x 32 channel
    // pairs (pairs 27..31 = channels 54..63 = 0).  A wave owns 4 pairs over all rows (lane = (row mod 16, pair)): the z-score's two
    // reductions (utils/data_handler.py:55-56: mean and unbiased std per channel over the 150 rows; fp64 as in load_windows) stay
    // inside the wave -- 10 rows in the lane, then four shuffles -- and need no barrier.  Per item: one 8-byte read, one split, three
    // 4-byte stores.  Cut into 12 steps, two per K-step of the phase that carries it: 0-2 the sums, 3-5 the squares (both only with the
    // z-score; the rows are read again rather than kept: registers), 6-10 two rows each, 11 the non-finite flag.
    float pmean0 = 0.f, pmean1 = 0.f, pinv0 = 1.f, pinv1 = 1.f, pnz = 0.f;
    double pacc0 = 0.0, pacc1 = 0.0, psq0 = 0.0, psq1 = 0.0;          // sums, then the means; sums of squared deviations
    float2 pc[2] = {};                                                 // the rows the next step works on
    const char* psrc = raw;                                            // this lane's pair in row r0 of the staged window
    char* pdst = bufX;                                                 // ... and its place in row r0 + 1 of the slot's first plane
    auto prologue_step = [&](int step, char* __restrict__ buf, int* flag) {
        // rows of the three passes handled in this step: [a, b).  Without the z-score only the last pass exists, a row per step.
        constexpr int P1[13] = {0, 5, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10};       // sums
        constexpr int P2[13] = {0, 0, 0, 5, 10, 10, 10, 10, 10, 10, 10, 10, 10};         // squared deviations
        constexpr int P3z[14] = {0, 0, 0, 0, 0, 2, 3, 5, 6, 8, 9, 10, 10, 10};           // z-scored rows out
        constexpr int P3n[14] = {0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 10, 10};           // pre-normalised rows out
        auto across_rows = [](double x) {                                      // sum over the 16 lanes (row mod 16) that hold this channel pair
            x += __shfl_xor(x, 4); x += __shfl_xor(x, 8); x += __shfl_xor(x, 16);
            return x + __shfl_xor(x, 32);
        };
        if (step == 0) {
            CXP_LANE;
            const int pr = 4 * wv + (lane & 3), r0 = lane >> 2;                // this lane: channel pair pr, rows r0 + 16 m
            psrc = raw + r0 * (CH * 4) + (pr < 27 ? pr : 0) * 8;
            pdst = buf + cx_addr<128>(r0 + 1, 2 * pr);                         // (row r0 + 1 + 16 m has the swizzle of row r0 + 1: + 2048 m)
            // rows 0 and 151 = the zero padding (threads 0..47; the others store their zeros to the dump row)
            const int at = (tid >> 4) * CX_PLANE + ((tid >> 3) & 1) * (151 * 128) + (tid & 7) * 16;
            *reinterpret_cast<uint4*>(buf + (tid < 48 ? at : CXP_DUMP)) = make_uint4(0, 0, 0, 0);
            pnz = 0.f; pacc0 = 0.0; pacc1 = 0.0; psq0 = 0.0; psq1 = 0.0;
        }
        const bool real = 4 * wv + (lane0 & 3) < 27, last_in = (lane0 >> 2) < WIN - 144;      // (row r0 + 144 exists for r0 < 6)
        auto row = [&](int m) {
            float2 v = *reinterpret_cast<const float2*>(psrc + m * (16 * CH * 4));
            if (m == 9) { v.x = last_in ? v.x : 0.f; v.y = last_in ? v.y : 0.f; }
            return v;
        };
        if constexpr (ZS) {
#pragma unroll
            for (int m = P1[step]; m < P1[step + 1]; ++m) { const float2 v = row(m); pacc0 += (double)v.x; pacc1 += (double)v.y; }
            if (step == 1) { pacc0 = across_rows(pacc0) / 150.0; pacc1 = across_rows(pacc1) / 150.0; }       // the means
#pragma unroll
            for (int m = P2[step]; m < P2[step + 1]; ++m) {
                const float2 v = row(m);
                const double d0 = (double)v.x - pacc0, d1 = (double)v.y - pacc1;
                psq0 += (m < 9 || last_in) ? d0 * d0 : 0.0; psq1 += (m < 9 || last_in) ? d1 * d1 : 0.0;
            }
            if (step == 3) {
                pmean0 = (float)pacc0; pmean1 = (float)pacc1;
                pinv0 = 1.f / (float)sqrt(across_rows(psq0) / 149.0); pinv1 = 1.f / (float)sqrt(across_rows(psq1) / 149.0);
            }
        }
        // (a row is read one step ahead of its use: the wave issues in order, and a read consumed where it stands makes it wait for
        //  the LDS -- behind the fifteen fragment requests in the queue -- ten times per window: 4k cycles per prologue)
        auto P3 = [&](int i) { return ZS ? P3z[i] : P3n[i]; };
#pragma unroll
        for (int m = P3(step); m < P3(step + 1); ++m) {
            const float2 v = pc[m - P3(step)];
            float x0 = v.x, x1 = v.y;
            if (ZS) { x0 = (x0 - pmean0) * pinv0; x1 = (x1 - pmean1) * pinv1; }
            // non-finite scan: x * 0 is 0 for a finite x and NaN otherwise
            pnz = __builtin_fmaf(x0, 0.f, __builtin_fmaf(x1, 0.f, pnz));
            unsigned p[3];
            cxp_split2(real ? x0 : 0.f, real ? x1 : 0.f, p);
            char* d = pdst + m * 2048;
            if (m == 9) d = last_in ? d : buf + CXP_DUMP;
#pragma unroll
            for (int k = 0; k < 3; ++k) *reinterpret_cast<unsigned*>(d + k * CX_PLANE) = p[k];
        }
#pragma unroll
        for (int m = P3(step + 1); m < P3(step + 2); ++m) pc[m - P3(step + 1)] = row(m);
        // (the instruction itself: handed an atomic with a lane-dependent operand, the compiler reduces it over the wave first -- in a
        //  scalar loop over the 64 lanes, 5k cycles in the middle of the last K-step; the LDS serialises the 64 lanes in 64 cycles)
        if (step == 11) asm volatile("ds_or_b32 %0, %1" :: "v"(cxp_lds_addr(flag)), "v"((real && !(pnz == 0.f)) ? 1 : 0) : "memory");
    };

    // ---- start: the first window of slot X lands, its prologue runs on its own
    if (tid < 4) flags[tid] = 0;
    cxp_issue_dma(dma_dst, window_src(0, 0), dma_voff);
    cxp_barrier_dma();
#pragma unroll
    for (int step = 0; step < 12; ++step) prologue_step(step, bufX, &flags[0]);
    cxp_barrier();

    // this wave's row tile of the packed weights ([row-tile pair][step][row tile (2)][plane (3)][lane (64)] x 16 bytes)
    const int R1 = wv & 3, h1 = wv >> 2;                                       // stage 1: row tile R1, column tiles 5 h1 ..
    const uint4* const w0 = reinterpret_cast<const uint4*>(pk.w[0]) + ((R1 >> 1) * 36 + (R1 & 1) * 3) * 64;
    const uint4* const w1 = reinterpret_cast<const uint4*>(pk.w[1]) + ((R1 >> 1) * 36 + (R1 & 1) * 3) * 64;
    const uint4* const w2 = reinterpret_cast<const uint4*>(pk.w[2]) + ((wv >> 1) * 36 + (wv & 1) * 3) * 64;
    const uint4* const w3 = reinterpret_cast<const uint4*>(pk.w[3]) + ((wv >> 1) * 72 + (wv & 1) * 3) * 64;
    uint4 apre[3];
    float4 bpre;
    auto pre_layer = [&](const uint4* __restrict__ wp, const float* __restrict__ bias, int co0) {      // next phase's first weight fragments + bias: requested now
        CXP_LANE;
#pragma unroll
        for (int p = 0; p < 3; ++p) apre[p] = wp[lane + p * 64];
        bpre = *reinterpret_cast<const float4*>(bias + co0 + 4 * g);
    };
    cx_f32x4 accX[CX_NT], accY[CX_NT];
#pragma unroll
    for (int ct = 0; ct < CX_NT; ++ct) accY[ct] = cx_f32x4{0.f, 0.f, 0.f, 0.f};
    CxpTail tl;
    pre_layer(w0, pk.b[0], 16 * R1);

    // the phases' building blocks (lane ids come from the caller's opaque copy)
    auto conv_stage1 = [&](const char* buf, const uint4* w, cx_f32x4 (&acc)[CX_NT], int lane, auto&& side) {       // conv1 / conv2: 128-byte rows, T = 150
        const int j = lane & 15, g = lane >> 4, base = 16 * 5 * h1 + j;
        const int sw[3] = {cx_swz<128>(base), cx_swz<128>(base + 1), cx_swz<128>(base + 2)};
        cxp_layer<128, 2>(buf + base * 128, sw, g, w + lane, apre, bpre, acc, side);
    };
    auto conv_stage1_traced = [&](const char* buf, const uint4* w, cx_f32x4 (&acc)[CX_NT], int lane, auto&& side) {
        const int j = lane & 15, g = lane >> 4, base = 16 * 5 * h1 + j;
        const int sw[3] = {cx_swz<128>(base), cx_swz<128>(base + 1), cx_swz<128>(base + 2)};
        cxp_layer<128, 2, true>(buf + base * 128, sw, g, w + lane, apre, bpre, acc, side);
    };
    auto conv3 = [&](const char* buf, cx_f32x4 (&acc)[CX_NT], int lane, auto&& side) {
        const int j = lane & 15, g = lane >> 4;
        const int sw[3] = {cx_swz<128>(j), cx_swz<128>(j + 1), cx_swz<128>(j + 2)};
        cxp_layer<128, 2>(buf + j * 128, sw, g, w2 + lane, apre, bpre, acc, side);
    };
    auto conv4 = [&](const char* buf, cx_f32x4 (&acc)[CX_NT], int lane, auto&& side) {
        const int j = lane & 15, g = lane >> 4;
        const int sw[3] = {cx_swz<256>(j), cx_swz<256>(j + 1), cx_swz<256>(j + 2)};
        cxp_layer<256, 4>(buf + j * 256, sw, g, w3 + lane, apre, bpre, acc, side);
    };
    // pending write-backs: conv1 (store1), conv2 + pool (store2: rows 1..75 of the stage-2 layout, 64 channels; row 76 = padding),
    // conv3 (store3: 128 channels, 256-byte rows 1..75; rows 0 and 76 = padding), in place
    auto store1 = [&](char* buf, const cx_f32x4 (&acc)[CX_NT], int lane, int ct, int st) {
        const int j = lane & 15, g = lane >> 4;
        cxp_store_stage<128, false, WIN>(tl, st, buf, acc[ct], 16 * R1 + 4 * g, 16 * (5 * h1 + ct) + j, j);
    };
    auto store2 = [&](char* buf, const cx_f32x4 (&acc)[CX_NT], int lane, int ct, int st) {
        const int j = lane & 15, g = lane >> 4;
        cxp_store_stage<128, true, WIN>(tl, st, buf, acc[ct], 16 * R1 + 4 * g, 16 * (5 * h1 + ct) + j, j);
    };
    auto store3 = [&](char* buf, const cx_f32x4 (&acc)[CX_NT], int lane, int ct, int st) {
        const int j = lane & 15, g = lane >> 4;
        cxp_store_stage<256, false, 75>(tl, st, buf, acc[ct], 16 * wv + 4 * g, 16 * ct + j, j);
    };
    auto zero_rows = [&](char* buf, int rowb, int last) {                      // rows 0 and `last` of every plane (threads past the rows' slots: dump row)
        const int slots = rowb / 16, per = 2 * slots, p = tid / per, r = (tid / slots) & 1, sl = tid % slots;
        const int at = p * CX_PLANE + (r ? last : 0) * rowb + sl * 16;
        *reinterpret_cast<uint4*>(buf + (tid < 3 * per ? at : CXP_DUMP)) = make_uint4(0, 0, 0, 0);
    };
    // conv4 + pool -> features of window `win` (nothing is stored for a window that does not exist)
    size_t fstride = 0;                                                        // bytes between a tile's planes in HBM (0 on the dump line)
    auto features = [&](const cx_f32x4 (&acc)[CX_NT], int lane, int64_t win, int ct, int st) {
        const int j = lane & 15, g = lane >> 4;
        if (st == 0) tl.relu<true>(acc[ct]);                                   // MaxPool over (t, t + 1), ReLU; t = 74 has no partner and is dropped
        else if (st <= (OUT == 2 ? 1 : 3)) {                                   // (bf16 features: the value rounded, no remainders)
            // No lane sits out (a store under an exec mask is a branch, and a branch cuts the K-step's scheduling region in two):
            // the odd column of a pool pair holds the pair's maximum too and stores the same 8 bytes to the same address as its
            // even neighbour; the three positions past t' = 36 and windows that do not exist store to a dump line.
            if (st == 1) {
                const int t = 16 * ct + j, k = (t >> 1) * 128 + 16 * wv + 4 * g;
                unsigned short* const at = OUT == 2 ? feat + (size_t)win * FEAT + k : feat + (size_t)(win >> 1) * (2 * FEAT) + (int)(win & 1) * 32 + (k >> 5) * 64 + (k & 31);
                const bool ok = (t >> 1) < 37 && win < n;
                tl.at = reinterpret_cast<char*>(ok ? at : dump);
                fstride = ok ? plane_elems * 2 : 0;
                asm volatile("" : "+v"(tl.at), "+v"(fstride));
            }
            const uint2 v = tl.term(OUT != 2 && st < 3);
            if (!(CXP_EXP & 4) || v.x == 0x12345678u) *reinterpret_cast<uint2*>(tl.at + (st - 1) * fstride) = v;
        }
    };
    auto nan_features = [&](int lane, int64_t win) {                           // a non-finite sample: NaN in every term of the window's features
        const int j = lane & 15, g = lane >> 4;
#pragma unroll
        for (int ct = 0; ct < CX_NT; ++ct) {
            const int t = 16 * ct + j, k = (t >> 1) * 128 + 16 * wv + 4 * g;
            if ((j & 1) == 0 && (t >> 1) < 37 && win < n) {
                if constexpr (OUT == 2) *reinterpret_cast<uint2*>(feat + (size_t)win * FEAT + k) = make_uint2(0x7fc07fc0u, 0x7fc07fc0u);
                else {
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        *reinterpret_cast<uint2*>(feat + p * plane_elems + (size_t)(win >> 1) * (2 * FEAT) + (int)(win & 1) * 32 + (k >> 5) * 64 + (k & 31)) = make_uint2(0x7fc07fc0u, 0x7fc07fc0u);
                }
            }
        }
    };

    for (int q = 0; q <= Q; ++q) {
        const int64_t winX = q < Q ? window_of(q, 0) : n, winYp = q > 0 ? window_of(q - 1, 1) : n;      // (n = does not exist)
        int* const flagX = flags + (q & 1), *const flagY = flags + 2 + (q & 1), *const flagYp = flags + 2 + ((q + 1) & 1);
        // ================= phase 1: conv1 X; store3 Y- rides along; slot Y's next window is requested =================
        CXP_T(0);
        if (q < Q) cxp_issue_dma(dma_dst, window_src(q, 1), dma_voff);         // (the staging buffer was read in the phase before)
        if (tid == 0) *flagY = 0;
        {   CXP_LANE;
            conv_stage1(bufX, w0, accX, lane, [&](int s) {
                cxp_deal<6>(s, [&](int ct, int st) { store3(bufY, accY, lane, ct, st); });
                if (s == 5) { zero_rows(bufY, 256, 76); pre_layer(w3, pk.b[3], 16 * wv); }
            }); }
        CXP_T(1);
        cxp_barrier();
        // ================= phase 2: conv4 Y-; store1 X =================
        CXP_T(2);
        {   CXP_LANE;
            conv4(bufY, accY, lane, [&](int s) {
                cxp_deal<12>(s, [&](int ct, int st) { store1(bufX, accX, lane, ct, st); });
                if (s == 10) pre_layer(w1, pk.b[1], 16 * R1);
            }); }
        CXP_T(3);
        cxp_barrier_dma();                                                     // (slot Y's next window has landed: every wave waits for its own pieces)
        // ================= phase 3: conv2 X; features Y- and the prologue of Y =================
        CXP_T(4);
        {   CXP_LANE;
            const bool bad = *flagYp != 0;
            conv_stage1_traced(bufX, w1, accX, lane, [&](int s) {
                cxp_deal<6>(s, [&](int ct, int st) { features(accY, lane, winYp, ct, st); });
                if (!(CXP_EXP & 8)) { prologue_step(2 * s, bufY, flagY); prologue_step(2 * s + 1, bufY, flagY); }
                if (s == 4) pre_layer(w0, pk.b[0], 16 * R1);
            });
            if (__builtin_expect(bad, 0)) nan_features(lane, winYp); }
        CXP_T(5);
        cxp_barrier();
#if DCE_TRACE == 2
        { constexpr bool KTRACE = true; CXP_TK(9); }
#endif
        if (q == Q) break;
        // ================= phase 4: conv1 Y; store2 X; slot X's next window is requested =================
        CXP_T(6);
        if (q + 1 < Q) cxp_issue_dma(dma_dst, window_src(q + 1, 0), dma_voff);
        if (tid == 0) flags[(q + 1) & 1] = 0;
        {   CXP_LANE;
            conv_stage1(bufY, w0, accY, lane, [&](int s) {
                cxp_deal<6>(s, [&](int ct, int st) { store2(bufX, accX, lane, ct, st); });
                if (s == 5) { zero_rows(bufX, 128, 76); pre_layer(w2, pk.b[2], 16 * wv); }
            }); }
        CXP_T(7);
        cxp_barrier();
        // ================= phase 5: conv3 X; store1 Y =================
        CXP_T(8);
        {   CXP_LANE;
            conv3(bufX, accX, lane, [&](int s) {
                cxp_deal<6>(s, [&](int ct, int st) { store1(bufY, accY, lane, ct, st); });
                if (s == 4) pre_layer(w1, pk.b[1], 16 * R1);
            }); }
        CXP_T(9);
        cxp_barrier();
        // ================= phase 6: conv2 Y; store3 X =================
        CXP_T(10);
        {   CXP_LANE;
            conv_stage1(bufY, w1, accY, lane, [&](int s) {
                cxp_deal<6>(s, [&](int ct, int st) { store3(bufX, accX, lane, ct, st); });
                if (s == 5) { zero_rows(bufX, 256, 76); pre_layer(w3, pk.b[3], 16 * wv); }
            }); }
        CXP_T(11);
        cxp_barrier();
        // ================= phase 7: conv4 X; store2 Y =================
        CXP_T(12);
        {   CXP_LANE;
            conv4(bufX, accX, lane, [&](int s) {
                cxp_deal<12>(s, [&](int ct, int st) { store2(bufY, accY, lane, ct, st); });
                if (s == 10) { zero_rows(bufY, 128, 76); pre_layer(w2, pk.b[2], 16 * wv); }
            }); }
        CXP_T(13);
        cxp_barrier_dma();                                                     // (slot X's next window has landed)
        // ================= phase 8: conv3 Y; features X and the prologue of X+ =================
        CXP_T(14);
        {   CXP_LANE;
            const bool bad = *flagX != 0;
            conv3(bufY, accY, lane, [&](int s) {
                cxp_deal<6>(s, [&](int ct, int st) { features(accX, lane, winX, ct, st); });
                if (!(CXP_EXP & 8)) { prologue_step(2 * s, bufX, flags + ((q + 1) & 1)); prologue_step(2 * s + 1, bufX, flags + ((q + 1) & 1)); }
                if (s == 4) pre_layer(w0, pk.b[0], 16 * R1);
            });
            if (__builtin_expect(bad, 0)) nan_features(lane, winX); }
        CXP_T(15);
        cxp_barrier();
    }
#undef CXP_LANE
}

namespace {
int cxp_num_cu()
{
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        else cus = 256;
    }
    return cus;
}
template <bool ZS, int OUT>
hipError_t cxp_launch(const float* src, int64_t n, const ConvPackX3& pk, unsigned short* feat, size_t plane_elems, hipStream_t st)
{
    static unsigned short* dump = nullptr;                            // 256 bytes nobody reads (one per process: see conv_x3p_kernel's feature stores)
    if (!dump) { hipError_t e = hipMalloc(&dump, 256); if (e != hipSuccess) return e; }
    const int64_t cus = cxp_num_cu(), npairs = (n + 1) / 2;
    const unsigned grid = (unsigned)(npairs < cus ? npairs : cus);
    hipLaunchKernelGGL((conv_x3p_kernel<ZS, OUT>), dim3(grid), dim3(512), CXP_LDS, st, src, n, pk, feat, plane_elems, dump);
    return hipGetLastError();
}
}  // namespace

void fc_perm_k_host(const float* w, size_t rows, float* out)
{
    for (size_t o = 0; o < rows; ++o)
        for (int c = 0; c < 128; ++c)
            for (int t = 0; t < 37; ++t) out[o * FEAT + t * 128 + c] = w[o * FEAT + c * 37 + t];
}

hipError_t init_conv_x3p()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3p_kernel<false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, CXP_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3p_kernel<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, CXP_LDS);
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3p_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, CXP_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_x3p_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, CXP_LDS);
}

// features as three bf16 planes in fc_gemm_x3.hip's layout, K order k' = t' * 128 + c
hipError_t launch_conv_x3p(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat3, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    const size_t plane_elems = (size_t)((n + 1) & ~(int64_t)1) * FEAT;
    plan_note("conv_x3p");
    return zscore ? cxp_launch<true, 0>(src, n, pk, feat3, plane_elems, st) : cxp_launch<false, 0>(src, n, pk, feat3, plane_elems, st);
}

// features as (n, 4736) bf16, K order k' = t' * 128 + c (the DCE_BF16_FC precision)
hipError_t launch_conv_x3p_bf16(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat, hipStream_t st)
{
    if (n <= 0) return hipSuccess;
    plan_note("conv_x3p_bf16");
    return zscore ? cxp_launch<true, 2>(src, n, pk, feat, 0, st) : cxp_launch<false, 2>(src, n, pk, feat, 0, st);
}

}  // namespace dce

#if DCE_TRACE == 2
extern "C" int dce_debug_trace_read_x3p_k(unsigned long long* out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_trace_k), sizeof(unsigned long long) * 8 * 16 * nblocks);
}
#endif
#if DCE_TRACE
extern "C" int dce_debug_trace_read_x3p(unsigned long long* out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_trace_p), sizeof(unsigned long long) * 16 * nblocks);
}

#endif

#else   // !DCE_EXPERIMENTS: the default library carries neither the kernel nor its dispatch (dce_api.hip never selects it)

namespace dce {
void fc_perm_k_host(const float* w, size_t rows, float* out)
{
    for (size_t o = 0; o < rows; ++o)
        for (int c = 0; c < 128; ++c)
            for (int t = 0; t < 37; ++t) out[o * FEAT + t * 128 + c] = w[o * FEAT + c * 37 + t];
}
hipError_t init_conv_x3p() { return hipSuccess; }
hipError_t launch_conv_x3p(const float*, int, int64_t, const ConvPackX3&, unsigned short*, hipStream_t) { return hipErrorNotSupported; }
hipError_t launch_conv_x3p_bf16(const float*, int, int64_t, const ConvPackX3&, unsigned short*, hipStream_t) { return hipErrorNotSupported; }
}  // namespace dce

#endif  // DCE_EXPERIMENTS
