// conv_x3p.hip -- the conv stack on three-term bf16 operands (conv_x3.hip has the arithmetic; reference src/contact_cnn.py:10-44,61,64
// and utils/data_handler.py:55-56) for chip-filling batches: one persistent workgroup of EIGHT waves per CU works on one window at a
// time, and every wave's non-MFMA work rides inside its own MFMA stream.
//
// Why (profiles/r4a_micro_bf16_mfma_valu.txt, gfx950): beside a wave that issues v_mfma_f32_16x16x32_bf16 back to back, the OTHER wave
// of the SIMD gets one VALU instruction in 8-13 cycles -- a write-back phase next to a conv phase crawls, whether the two waves belong
// to two free-running workgroups (conv_x3.hip: 86k cycles per window pair for 57.6k cycles of MFMAs) or to two groups of one workgroup
// held a fixed number of phases apart (first cut of this file, profiles/r4a_trace_phase_shifted.txt: 80-85k).  Two waves that EACH mix
// MFMAs with their own VALU work keep the matrix pipe full with up to four VALU instructions per MFMA.  So:
//   * 8 waves, one 16-row tile x five 16-column tiles each (stage 1: row tile wv & 3, column tiles 5 (wv >> 2)..; stage 2: row tile
//     wv): per K-step 3 weight fragments (L2) + 15 activation fragments (LDS) for 30 MFMAs, requests dealt out between the MFMAs.
//   * the activations ping-pong between two LDS buffers (conv1: A -> B, conv2 + pool: B -> A, conv3: A -> B, conv4: B -> features), so a
//     write-back needs no barrier in front of it and starts while the wave's MFMAs still run: the last K-step goes tile by tile, and
//     a tile's bias/ReLU/pool/split/stores issue between the MFMAs of the tiles behind it.  One barrier per layer.
//   * the NEXT window arrives by LDS-DMA (global_load_lds_dwordx4, 32 pieces of 1 KB, requested at the start of conv1) and its
//     prologue -- z-score, split into three terms, stores into buffer A, which nobody reads then -- is dealt out over the twelve
//     K-steps of conv4.  No HBM latency, no 63 KB zero fill (only the padding rows and channels 54..63 are written).
//   * the features leave straight from the accumulators, in the order k' = t' * 128 + c (a lane holds four consecutive channels of
//     one pooled position: one 8-byte store per tile and plane) instead of the reference's flatten order k = c * 37 + t'; fc.0's
//     weights for this path are stored with their K axis permuted the same way (dce_finalize_weights), which changes no product, only
//     the order of an fp32-grade / bf16-input summation that claims no bit pattern (DCE_FP32_SPLIT, DCE_BF16_FC).
// LDS: 2 x 62,976 B of activation planes + 32 KB staging + flags = 158.8 KB: one workgroup per CU, two waves per SIMD.
#include "conv_x3_common.h"
#include <cfloat>

namespace dce {

#if DCE_TRACE
// debug build: per workgroup, wave 0's clock at the start of every layer of its last window and in front of every barrier
// (tools/trace_conv_x3p.py)
static __device__ unsigned long long g_trace_p[1024 * 16];
#define CXP_T(k) do { if (lane0 == 0 && wv == 0 && blockIdx.x < 1024) g_trace_p[blockIdx.x * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define CXP_T(k) do {} while (0)
#endif
#if DCE_TRACE == 2
// ... and with -DDCE_TRACE=2 every wave's clock at conv1's K-step boundaries, its tail and its barrier (slots [wave][0..15])
static __device__ unsigned long long g_trace_k[256 * 8 * 16];
#define CXP_TK(k) do { if (KTRACE && __lane_id() == 0 && blockIdx.x < 256) g_trace_k[(blockIdx.x * 8 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * 16 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define CXP_TK(k) do {} while (0)
#endif

#ifndef CXP_PRIO
#define CXP_PRIO 1       // a wave's issue priority falls as it advances through a layer's K-steps (A/B: -DCXP_PRIO=0)
#endif
#ifndef CXP_EXP
#define CXP_EXP 0        // timing probes (WRONG results): 1 no tile tails (write-backs / feature stores), 2 no side work (prologue of the next window, next layer's first weights)
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
struct CxpTail {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    f2 ra, rb;
    unsigned lo[3], hi[3];
    // (a stage's results are pinned where the stage stands: sched_barrier orders machine instructions, but before that the
    //  optimiser sinks side-effect-free arithmetic down to its first use -- the stores of stages 4 and 5)
    __device__ __forceinline__ void pin() { asm volatile("" : "+v"(ra), "+v"(rb)); }
    template <bool POOL> __device__ __forceinline__ void relu(const cx_f32x4& a)
    {
        float v[4] = {a[0], a[1], a[2], a[3]};                        // (the bias is the accumulators' initial value)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = POOL ? fmaxf(fmaxf(v[r], cx_neighbour(v[r])), 0.f) : fmaxf(v[r], 0.f);
        ra = f2{v[0], v[1]}; rb = f2{v[2], v[3]};
        pin();
    }
    __device__ __forceinline__ void term(int k)                       // term k of the four values; the remainders stay in ra / rb
    {
        const b2 ta = __builtin_convertvector(ra, b2), tb = __builtin_convertvector(rb, b2);
        lo[k] = __builtin_bit_cast(unsigned, ta); hi[k] = __builtin_bit_cast(unsigned, tb);
        if (k < 2) { ra -= __builtin_convertvector(ta, f2); rb -= __builtin_convertvector(tb, f2); pin(); }
        else asm volatile("" : "+v"(lo[2]), "+v"(hi[2]));
    }
};

// stages of a layer's tile -> three-term planes of the next layer's input (LDS buffer `out`)
//   co: first of the lane's four channels; t: the lane's column;  T: columns of this layer; POOL: the next layer sees T / 2 positions
template <int ROWB_OUT, bool POOL, int T>
__device__ __forceinline__ void cxp_store_stage(CxpTail& tl, int stage, char* __restrict__ out, const cx_f32x4& a, int co, int t, int j)
{
    if (stage == 0) tl.relu<POOL>(a);
    else if (stage <= 3) tl.term(stage - 1);
    else {
        const bool ok = POOL ? ((j & 1) == 0 && (t >> 1) < T / 2) : t < T;
        const int row = (POOL ? (t >> 1) : t) + 1;
        char* d = out + (ok ? cx_addr<ROWB_OUT>(row, co) : CXP_DUMP);
        if (stage == 4) {
            *reinterpret_cast<uint2*>(d) = make_uint2(tl.lo[0], tl.hi[0]);
            *reinterpret_cast<uint2*>(d + CX_PLANE) = make_uint2(tl.lo[1], tl.hi[1]);
        } else *reinterpret_cast<uint2*>(d + 2 * CX_PLANE) = make_uint2(tl.lo[2], tl.hi[2]);
    }
}

// One layer for one wave: acc[ct] = bias + sum over K-steps s = (channel block kb, tap) of W(s) x X(ct, s), then the six stages
// `tail(ct, stage)` of every tile as soon as its last MFMA is out (the last K-step runs tile by tile), with `side(s)` -- unrelated
// work of the caller -- dealt out over the K-steps.
//   ROWB : bytes per LDS row of the layer's input (2 x input channels)      NKB : 32-channel blocks of K
//   xrow : in + (16 ct0 + j) * ROWB  (this lane's row of column tile 0, tap 0)     sw[tap] = swz(16 ct0 + j + tap)
//   wp   : this wave's row tile of the packed weights (+ lane): fragment (step s, plane p) at wp[(6 s + p) * 64]
//   apre : the weight fragments of step 0, requested by the caller ahead of the barrier in front of this layer
template <int ROWB, int NKB, bool KTRACE = false, class Side, class Tail>
__device__ __forceinline__ void cxp_layer(const char* __restrict__ xrow, const int (&sw)[3], int g, const uint4* __restrict__ wp,
                                          const uint4 (&apre)[3], const float4& bias, cx_f32x4 (&acc)[CX_NT], Side&& side, Tail&& tail)
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
    auto mfma = [&](int s, int ct, int t) {
        acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cx_bf16x8, af[s % 3][TA[t]]), __builtin_bit_cast(cx_bf16x8, bf[s & 1][ct][TB[t]]),
                                                          acc[ct], 0, 0, 0);
    };
#pragma unroll
    for (int ct = 0; ct < CX_NT; ++ct) acc[ct] = cx_f32x4{bias.x, bias.y, bias.z, bias.w};
    CXP_TK(0);
    fetch_w(0); fetch_w(1); fetch_x(0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < S - 1; ++s) {
        CXP_TK(1 + s);
        // The SIMD's issue port goes to the older of its two waves whenever both have an instruction ready: left alone, waves
        // 0..3 run ahead (60 % of the pipe), finish a layer 2.5k cycles early and leave waves 4..7 to finish alone at a single
        // wave's rate.  A priority that falls with the K-step lets the wave that is behind win the port instead.
        if (CXP_PRIO) switch (3 - (4 * s) / S) {                      // (the builtin wants a literal; the switch folds once the loop is unrolled)
            case 3: __builtin_amdgcn_s_setprio(3); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            default: __builtin_amdgcn_s_setprio(0); break;
        }
        if (s + 2 < S) fetch_w(s + 2);
        fetch_x(s + 1);
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int ct = 0; ct < CX_NT; ++ct) mfma(s, ct, t);        // consecutive MFMAs go to different accumulators
        if (!(CXP_EXP & 2)) side(s);
        // the order of this K-step's region: an MFMA, one of the 18 fragment requests (weights first: L2), and what there is of
        // the caller's side work
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 3 * CX_NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
#pragma unroll
        for (int i = 0; i < 30 - 3 - 3 * CX_NT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    {   // last K-step, tile by tile: stage i of a tile's tail follows MFMA i of the tile behind it
        CXP_TK(S);
        if (CXP_PRIO) __builtin_amdgcn_s_setprio(0);
        if (!(CXP_EXP & 2)) side(S - 1);
#pragma unroll
        for (int t = 0; t < 6; ++t) mfma(S - 1, 0, t);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ct = 1; ct < CX_NT; ++ct)
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                mfma(S - 1, ct, t);
                if (!(CXP_EXP & 1)) tail(ct - 1, t);
                __builtin_amdgcn_sched_barrier(0);
            }
        CXP_TK(S + 1);
#pragma unroll
        for (int t = 0; t < 6; ++t) if (!(CXP_EXP & 1)) tail(CX_NT - 1, t);
        CXP_TK(S + 2);
    }
}

}  // namespace

// OUT: 0 the features leave as three bf16 planes (fc_gemm_x3.hip's pair-interleaved layout), 2 as (n, 4736) bf16 (term 1 = the
// value rounded to nearest-even); both in the order k' = t' * 128 + c.
template <bool ZS, int OUT>
__global__ __launch_bounds__(512, 2)
void conv_x3p_kernel(const float* __restrict__ src, int64_t n, ConvPackX3 pk, unsigned short* __restrict__ feat, size_t plane_elems)
{
    extern __shared__ __attribute__((aligned(16))) char cxp_lds[];
    const int tid = threadIdx.x, lane0 = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const bufA = cxp_lds;
    char* const bufB = cxp_lds + CXP_OFF_B;
    const char* const raw = cxp_lds + CXP_OFF_RAW;
    int* const flags = reinterpret_cast<int*>(cxp_lds + CXP_OFF_FLAG);         // non-finite window: [parity of the window's index in this workgroup]
    const int64_t win_stride = ZS ? CH : WIN * CH;                             // floats between consecutive windows
    const int Q = (int)((n - blockIdx.x + gridDim.x - 1) / gridDim.x);         // windows of this workgroup (>= 1): blockIdx.x + gridDim.x q
    const unsigned dma_dst = cxp_lds_addr(raw) + wv * 4096, dma_voff = wv * 4096 + lane0 * 16;
    auto window_of = [&](int q) { return (int64_t)blockIdx.x + (int64_t)gridDim.x * q; };
    auto window_src = [&](int q) { return reinterpret_cast<const char*>(src + window_of(q) * win_stride); };
    // Every layer derives its lane-dependent addresses from an opaque copy of the lane id: the window loop's body is the same in
    // every iteration, and left alone the compiler hoists the address arithmetic of ALL layers out of the loop (1.5 KB of scratch).
#define CXP_LANE int lane = lane0; asm volatile("" : "+v"(lane)); const int j = lane & 15, g = lane >> 4; (void)j; (void)g

    // The prologue of window q, staged in `raw`, into buffer A as three-term planes [t + 1][channel]: 150 rows x 32 channel pairs
    // (pairs 27..31 = channels 54..63 = 0).  A wave owns 4 pairs over all rows (lane = (row mod 16, pair)): the z-score's two
    // reductions (utils/data_handler.py:55-56: mean and unbiased std per channel over the 150 rows; fp64 as in load_windows) stay
    // inside the wave -- 10 rows in the lane, then four shuffles -- and need no barrier.  Per item: one 8-byte read, one split, three
    // 4-byte stores.  Cut into 12 steps so that conv4's K loop can carry it (step = K-step; the first window runs them back to back).
    float2 pv[10];
    float pmean0 = 0.f, pmean1 = 0.f, pinv0 = 1.f, pinv1 = 1.f, pnz = 0.f;
    double psum0 = 0.0, psum1 = 0.0;
    auto prologue_step = [&](int step, int q) {
        CXP_LANE;
        const int pr = 4 * wv + (lane & 3), r0 = lane >> 2;                    // rows r0 + 16 m
        const bool real = pr < 27;
        auto across_rows = [](double x) {                                      // sum over the 16 lanes (row mod 16) that hold this channel pair
            x += __shfl_xor(x, 4); x += __shfl_xor(x, 8); x += __shfl_xor(x, 16);
            return x + __shfl_xor(x, 32);
        };
        if (step == 0) {
            const char* s = raw + r0 * (CH * 4) + (real ? pr : 0) * 8;
#pragma unroll
            for (int m = 0; m < 10; ++m) pv[m] = *reinterpret_cast<const float2*>(s + (r0 + 16 * m < WIN ? m : 0) * (16 * CH * 4));
            // rows 0 and 151 = the zero padding (threads 0..47; the others store their zeros to the dump row)
            *reinterpret_cast<uint4*>(bufA + (tid < 48 ? (tid >> 4) * CX_PLANE + ((tid >> 3) & 1) * (151 * 128) + (tid & 7) * 16 : CXP_DUMP)) = make_uint4(0, 0, 0, 0);
            pnz = 0.f;
        }
        if (ZS && step == 1) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                const bool in = r0 + 16 * m < WIN;
                s0 += in ? (double)pv[m].x : 0.0;
                s1 += in ? (double)pv[m].y : 0.0;
            }
            psum0 = across_rows(s0) / 150.0; psum1 = across_rows(s1) / 150.0;      // the means
        }
        if (ZS && step == 2) {
            double q0 = 0.0, q1 = 0.0;
#pragma unroll
            for (int m = 0; m < 10; ++m) {
                const bool in = r0 + 16 * m < WIN;
                const double d0 = (double)pv[m].x - psum0, d1 = (double)pv[m].y - psum1;
                q0 += in ? d0 * d0 : 0.0;
                q1 += in ? d1 * d1 : 0.0;
            }
            pmean0 = (float)psum0; pmean1 = (float)psum1;
            pinv0 = 1.f / (float)sqrt(across_rows(q0) / 149.0); pinv1 = 1.f / (float)sqrt(across_rows(q1) / 149.0);
        }
        if (step >= 2) {                                                        // rows m = step - 2 (steps 2..11)
            const int m = step - 2;
            float x0 = pv[m].x, x1 = pv[m].y;
            if (ZS) { x0 = (x0 - pmean0) * pinv0; x1 = (x1 - pmean1) * pinv1; }
            const int row = r0 + 16 * m;
            // non-finite scan: x * 0 is 0 for a finite x and NaN otherwise (rows past the window re-read row r0: no harm)
            pnz = __builtin_fmaf(x0, 0.f, __builtin_fmaf(x1, 0.f, pnz));
            unsigned p[3];
            cx_split2(real ? x0 : 0.f, real ? x1 : 0.f, p);
            char* d = bufA + (row < WIN ? cx_addr<128>(row + 1, 2 * pr) : CXP_DUMP);
#pragma unroll
            for (int k = 0; k < 3; ++k) *reinterpret_cast<unsigned*>(d + k * CX_PLANE) = p[k];
            if (step == 11) __hip_atomic_fetch_or(&flags[q & 1], (real && !(pnz == 0.f)) ? 1 : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    // ---- start: the first window lands, its prologue runs on its own
    if (tid == 0) { flags[0] = 0; flags[1] = 0; }
    cxp_issue_dma(dma_dst, window_src(0), dma_voff);
    cxp_barrier_dma();
#pragma unroll
    for (int step = 0; step < 12; ++step) prologue_step(step, 0);
    cxp_barrier();

    // this wave's row tile of the packed weights ([row-tile pair][step][row tile (2)][plane (3)][lane (64)] x 16 bytes)
    const int R1 = wv & 3, h1 = wv >> 2;                                       // stage 1: row tile R1, column tiles 5 h1 ..
    const uint4* const w0 = reinterpret_cast<const uint4*>(pk.w[0]) + ((R1 >> 1) * 36 + (R1 & 1) * 3) * 64;
    const uint4* const w1 = reinterpret_cast<const uint4*>(pk.w[1]) + ((R1 >> 1) * 36 + (R1 & 1) * 3) * 64;
    const uint4* const w2 = reinterpret_cast<const uint4*>(pk.w[2]) + ((wv >> 1) * 36 + (wv & 1) * 3) * 64;
    const uint4* const w3 = reinterpret_cast<const uint4*>(pk.w[3]) + ((wv >> 1) * 72 + (wv & 1) * 3) * 64;
    uint4 apre[3];
    float4 bpre;
    auto pre_layer = [&](const uint4* __restrict__ wp, const float* __restrict__ bias, int co0) {      // next layer's first weight fragments + bias: requested now
        CXP_LANE;
#pragma unroll
        for (int p = 0; p < 3; ++p) apre[p] = wp[lane + p * 64];
        bpre = *reinterpret_cast<const float4*>(bias + co0 + 4 * g);
    };
    cx_f32x4 acc[CX_NT];
    CxpTail tl;
    pre_layer(w0, pk.b[0], 16 * R1);

    for (int q = 0; q < Q; ++q) {
        const int64_t win = window_of(q);
        // ================= conv1: A -> B (64 channels, T = 150); the next window is requested =================
        CXP_T(0);
        if (q + 1 < Q) cxp_issue_dma(dma_dst, window_src(q + 1), dma_voff);    // (the staging buffer was read during the previous conv4)
        if (tid == 0) flags[(q + 1) & 1] = 0;
        {   CXP_LANE;
            const int base = 16 * 5 * h1 + j;
            const int sw[3] = {cx_swz<128>(base), cx_swz<128>(base + 1), cx_swz<128>(base + 2)};
            cxp_layer<128, 2, true>(bufA + base * 128, sw, g, w0 + lane, apre, bpre, acc,
                              [&](int s) { if (s == 4) pre_layer(w1, pk.b[1], 16 * R1); },      // (apre / bpre are consumed at the top of a layer)
                              [&](int ct, int st) { cxp_store_stage<128, false, WIN>(tl, st, bufB, acc[ct], 16 * R1 + 4 * g, 16 * (5 * h1 + ct) + j, j); });
            if (tid < 48) reinterpret_cast<uint4*>(bufB + (tid >> 4) * CX_PLANE + ((tid >> 3) & 1) * (151 * 128))[tid & 7] = make_uint4(0, 0, 0, 0);
        }
        CXP_T(1);
        cxp_barrier();
#if DCE_TRACE == 2
        { constexpr bool KTRACE = true; CXP_TK(9); }
#endif
        // ================= conv2 + pool: B -> A rows 1..75 of the stage-2 layout (64 channels), rows 0 and 76 = padding =================
        CXP_T(2);
        {   CXP_LANE;
            const int base = 16 * 5 * h1 + j;
            const int sw[3] = {cx_swz<128>(base), cx_swz<128>(base + 1), cx_swz<128>(base + 2)};
            cxp_layer<128, 2>(bufB + base * 128, sw, g, w1 + lane, apre, bpre, acc,
                              [&](int s) { if (s == 4) pre_layer(w2, pk.b[2], 16 * wv); },
                              [&](int ct, int st) { cxp_store_stage<128, true, WIN>(tl, st, bufA, acc[ct], 16 * R1 + 4 * g, 16 * (5 * h1 + ct) + j, j); });
            if (tid < 48) reinterpret_cast<uint4*>(bufA + (tid >> 4) * CX_PLANE + ((tid >> 3) & 1) * (76 * 128))[tid & 7] = make_uint4(0, 0, 0, 0);
        }
        CXP_T(3);
        cxp_barrier();
        // ================= conv3: A -> B (128 channels: 256-byte rows 1..75, rows 0 and 76 = padding) =================
        CXP_T(4);
        {   CXP_LANE;
            const int sw[3] = {cx_swz<128>(j), cx_swz<128>(j + 1), cx_swz<128>(j + 2)};
            cxp_layer<128, 2>(bufA + j * 128, sw, g, w2 + lane, apre, bpre, acc,
                              [&](int s) { if (s == 4) pre_layer(w3, pk.b[3], 16 * wv); },
                              [&](int ct, int st) { cxp_store_stage<256, false, 75>(tl, st, bufB, acc[ct], 16 * wv + 4 * g, 16 * ct + j, j); });
            if (tid < 96) reinterpret_cast<uint4*>(bufB + (tid >> 5) * CX_PLANE + ((tid >> 4) & 1) * (76 * 256))[tid & 15] = make_uint4(0, 0, 0, 0);
        }
        CXP_T(5);
        cxp_barrier_dma();                                                     // (the next window has landed: every wave waits for its own pieces)
        // ================= conv4 + pool -> features; the next window's prologue into buffer A rides along =================
        CXP_T(6);
        {   CXP_LANE;
            const bool bad = flags[q & 1] != 0;
            unsigned short* const out = OUT == 2 ? feat + (size_t)win * FEAT : feat + (size_t)(win >> 1) * (2 * FEAT) + (int)(win & 1) * 32;
            const int sw[3] = {cx_swz<256>(j), cx_swz<256>(j + 1), cx_swz<256>(j + 2)};
            // (behind the last window the prologue steps run on stale staging data into a buffer nobody reads: a branch around them
            //  would cut the K-steps' scheduling regions in two)
            cxp_layer<256, 4>(bufB + j * 256, sw, g, w3 + lane, apre, bpre, acc,
                              [&](int s) { prologue_step(s, q + 1); if (s == 10) pre_layer(w0, pk.b[0], 16 * R1); },
                              [&](int ct, int st) {
                if (st == 0) tl.relu<true>(acc[ct]);                           // MaxPool over (t, t + 1), ReLU; t = 74 has no partner and is dropped
                else if (st <= 3) tl.term(st - 1);
                else {
                    const int t = 16 * ct + j, k = (t >> 1) * 128 + 16 * wv + 4 * g;
                    unsigned short* const d = OUT == 2 ? out + k : out + (k >> 5) * 64 + (k & 31);
                    if ((j & 1) == 0 && (t >> 1) < 37) {
                        if (st == 4) {
                            *reinterpret_cast<uint2*>(d) = make_uint2(tl.lo[0], tl.hi[0]);
                            if constexpr (OUT != 2) *reinterpret_cast<uint2*>(d + plane_elems) = make_uint2(tl.lo[1], tl.hi[1]);
                        } else if constexpr (OUT != 2) *reinterpret_cast<uint2*>(d + 2 * plane_elems) = make_uint2(tl.lo[2], tl.hi[2]);
                    }
                }
            });
            if (__builtin_expect(bad, 0)) {                                    // a non-finite sample: NaN in every term of the window's features
#pragma unroll
                for (int ct = 0; ct < CX_NT; ++ct) {
                    const int t = 16 * ct + j, k = (t >> 1) * 128 + 16 * wv + 4 * g;
                    if ((j & 1) == 0 && (t >> 1) < 37) {
                        if constexpr (OUT == 2) *reinterpret_cast<uint2*>(out + k) = make_uint2(0x7fc07fc0u, 0x7fc07fc0u);
                        else {
#pragma unroll
                            for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(out + p * plane_elems + (k >> 5) * 64 + (k & 31)) = make_uint2(0x7fc07fc0u, 0x7fc07fc0u);
                        }
                    }
                }
            }
        }
        CXP_T(7);
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
    const int64_t cus = cxp_num_cu();
    const unsigned grid = (unsigned)(n < cus ? n : cus);
    hipLaunchKernelGGL((conv_x3p_kernel<ZS, OUT>), dim3(grid), dim3(512), CXP_LDS, st, src, n, pk, feat, plane_elems);
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

#if DCE_TRACE
extern "C" int dce_debug_trace_read_x3p(unsigned long long* out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_trace_p), sizeof(unsigned long long) * 16 * nblocks);
}
#if DCE_TRACE == 2
extern "C" int dce_debug_trace_read_x3p_k(unsigned long long* out, int nblocks)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_trace_k), sizeof(unsigned long long) * 8 * 16 * nblocks);
}
#endif
#endif
