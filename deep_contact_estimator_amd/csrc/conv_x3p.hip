// conv_x3p.hip -- the conv stack on three-term bf16 operands (conv_x3.hip has the arithmetic and the per-layer GEMM; reference
// src/contact_cnn.py:10-44,61,64 and utils/data_handler.py:55-56) for chip-filling batches: TWO windows per workgroup, a fixed
// number of phases apart, so that one window's write-back always runs beside the other window's MFMAs.
//
// Why: with one window per workgroup and two free-running workgroups per CU (conv_x3.hip) a SIMD's two waves are in their
// non-MFMA phases (prologue, write-backs, feature output: ~2,900 instructions per wave and window, most of them waiting on each
// other's results, on LDS or on HBM) at the same time a third of the time: 86k cycles per window pair for 57.6k cycles of MFMAs.
// Here one workgroup = 8 waves = two groups of four; group 1 runs the same instruction stream as group 0, three phases later:
//
//   phase of a window        0: out(prev) + in    1: conv1   2: store   3: conv2   4: store+pool   5: conv3   6: store   7: conv4
//   group 0 at step g        g mod 8
//   group 1 at step g        (g - 3) mod 8        -> an MFMA phase of one group always meets a non-MFMA phase of the other
//
// and every phase ends in ONE workgroup barrier, which is what keeps the two groups in step.  A SIMD holds one wave of each group.
//   * persistent: a workgroup per CU walks over its window pairs; the NEXT window of a group arrives by LDS-DMA
//     (global_load_lds_dwordx4, 32 pieces of 1 KB) in a staging buffer the two groups use alternately: requested at the start of
//     the group's phase 6, consumed in its phase 0 -- the HBM latency of the prologue is gone, and so is its 63 KB zero fill
//     (only the padding rows and channels 54..63 are written).
//   * the first weight fragments and the bias of a layer are requested in the write-back phase in front of it, ahead of the barrier:
//     a conv phase starts on its activation reads alone.
//   * phase 0 has no barrier inside: the features leave straight from the accumulators, in the order k' = t' * 128 + c (a lane
//     holds four consecutive channels of one pooled position: one 8-byte store per tile and plane) instead of the reference's
//     flatten order k = c * 37 + t'; fc.0's weights for this path are stored with their K axis permuted the same way
//     (dce_finalize_weights), which changes no product and only the order of an fp32-grade / bf16-input summation that
//     claims no bit pattern (DCE_FP32_SPLIT, DCE_BF16_FC).
// LDS: 2 x 62,976 B of activation planes + 32 KB staging + flags = 158.8 KB: one workgroup per CU, two waves per SIMD.
#include "conv_x3_common.h"
#include <cfloat>

namespace dce {

#if DCE_TRACE
// debug build: per workgroup and group, the start of every phase (slot 2 ph) and the end of its work, in front of the barrier
// (slot 2 ph + 1), of the last window; tools/trace_conv_x3p.py
static __device__ unsigned long long g_trace_p[1024 * 64];
#define CXP_T(k) do { if (lane0 == 0 && wv == 0 && blockIdx.x < 1024) g_trace_p[blockIdx.x * 64 + grp * 32 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define CXP_T(k) do {} while (0)
#endif

#ifndef CXP_ILV
#define CXP_ILV 1        // fragment requests dealt out between the MFMAs (A/B: -DCXP_ILV=0)
#endif
#ifndef CXP_PRIO
#define CXP_PRIO 1       // the group in a conv phase outranks the group in a write-back phase on the SIMD's issue port (A/B: -DCXP_PRIO=0)
#endif

namespace {

constexpr int CXP_RAW = 32 * 1024;                                // staging of one raw window (32,400 B), 32 pieces of 1 KB
constexpr int CXP_OFF_RAW = 2 * CX_LDS, CXP_OFF_FLAG = CXP_OFF_RAW + CXP_RAW;
constexpr int CXP_LDS = CXP_OFF_FLAG + 64;
static_assert(CXP_LDS <= 160 * 1024, "one workgroup per CU");
constexpr int WIN_BYTES = WIN * CH * 4;

__device__ __forceinline__ unsigned cxp_lds_addr(const void* p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}

// end of a phase: this wave's LDS stores are done, then the workgroup barrier (requests to global memory stay in flight)
__device__ __forceinline__ void cxp_phase_end()
{
    __builtin_amdgcn_sched_barrier(0);
    if (CXP_PRIO) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void cxp_phase_end_dma()               // ... and everything this wave requested has landed (LDS-DMA)
{
    __builtin_amdgcn_sched_barrier(0);
    if (CXP_PRIO) __builtin_amdgcn_s_setprio(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// a wave's share of a window's LDS-DMA: pieces 8 wv .. 8 wv + 7 (1 KB each: 64 lanes x 16 bytes, LDS address = M0 + 16 lane).
// The last piece is cut at the window's end: its surplus lanes re-read the last 16 bytes (into staging nobody reads).
// M0 is compiler-reserved: saved and restored inside each statement.
__device__ __forceinline__ void cxp_issue_dma(unsigned lds_dst, const char* gsrc, unsigned voff)
{
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsigned off = voff + k * 1024;
        off = off < (unsigned)(WIN_BYTES - 16) ? off : (unsigned)(WIN_BYTES - 16);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(lds_dst + k * 1024), "v"(off), "s"(gsrc) : "memory");
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
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wv = wave & 3;
    // Every phase derives its lane-dependent addresses from an opaque copy of the lane id: the window loop's body is the same
    // in every iteration, and left alone the compiler hoists the address arithmetic of ALL phases out of the loop (1.5 KB of scratch).
#define CXP_LANE int lane = lane0; asm volatile("" : "+v"(lane)); const int j = lane & 15, g = lane >> 4, gtid = lane + 64 * wv; (void)j; (void)g; (void)gtid
    char* const lds = cxp_lds + grp * CX_LDS;                          // this group's activation planes
    const char* const raw = cxp_lds + CXP_OFF_RAW;
    int* const flags = reinterpret_cast<int*>(cxp_lds + CXP_OFF_FLAG) + 2 * grp;      // non-finite window: [parity of the window's index in this group]
    const int64_t win_stride = ZS ? CH : WIN * CH;                     // floats between consecutive windows
    const int64_t npairs = (n + 1) >> 1;
    const int Q = (int)((npairs - blockIdx.x + gridDim.x - 1) / gridDim.x);            // window pairs of this workgroup (>= 1)
    const unsigned dma_dst = cxp_lds_addr(raw) + wv * 8192, dma_voff = wv * 8192 + lane0 * 16;
    auto window_of = [&](int q) { return 2 * ((int64_t)blockIdx.x + (int64_t)gridDim.x * q) + grp; };
    auto window_src = [&](int q) {                                     // (a pair's second window may not exist: odd n -> recompute the last one, store nothing)
        int64_t w = window_of(q);
        w = w < n ? w : n - 1;
        return reinterpret_cast<const char*>(src + w * win_stride);
    };

    // ---- start: group 0's first window lands before step 0; group 1 runs three steps behind and asks for its first window at
    //      step 1, where the steady state asks for it too (its phase 6), after group 0 has read the staging buffer in step 0
    if ((tid & 255) == 0) { flags[0] = 0; flags[1] = 0; }
    if (grp == 0) { cxp_issue_dma(dma_dst, window_src(0), dma_voff); cxp_phase_end_dma(); }
    else {
        cxp_phase_end();
        cxp_phase_end();
        cxp_issue_dma(dma_dst, window_src(0), dma_voff);
        cxp_phase_end();
        cxp_phase_end_dma();
    }
    cx_f32x4 acc[2][CX_NT];
    CxW wpre;
    float4 bpre[2];
    auto pre_layer = [&](const uint4* __restrict__ wp, const float* __restrict__ bias, int co0, int g) {      // next layer's first fragments + bias: requested now
        cx_fetch_w0(wp, wpre);
        bpre[0] = *reinterpret_cast<const float4*>(bias + co0 + 4 * g);
        bpre[1] = *reinterpret_cast<const float4*>(bias + co0 + 16 + 4 * g);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto bias_acc = [&]() {                                            // the accumulators start from the bias of their four channels
        if (CXP_PRIO) __builtin_amdgcn_s_setprio(3);                   // (a conv phase begins here and ends at its barrier)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int ct = 0; ct < CX_NT; ++ct) acc[rt][ct] = cx_f32x4{bpre[rt].x, bpre[rt].y, bpre[rt].z, bpre[rt].w};
    };
    // this wave's packed weights: stage 1 (T = 150, 64 channels): wave = row-tile pair wv & 1, column tiles 5 (wv >> 1) ..;
    // stage 2 (T = 75): row-tile pair wv, all five column tiles
    const int P1 = wv & 1, ct1 = 5 * (wv >> 1);
    const uint4* const w0 = reinterpret_cast<const uint4*>(pk.w[0]) + (size_t)P1 * (6 * 2 * 3 * 64);
    const uint4* const w1 = reinterpret_cast<const uint4*>(pk.w[1]) + (size_t)P1 * (6 * 2 * 3 * 64);
    const uint4* const w2 = reinterpret_cast<const uint4*>(pk.w[2]) + (size_t)wv * (6 * 2 * 3 * 64);
    const uint4* const w3 = reinterpret_cast<const uint4*>(pk.w[3]) + (size_t)wv * (12 * 2 * 3 * 64);

    for (int q = 0; q <= Q; ++q) {
        // ================= phase 0: the previous window's features out; this window in =================
        if (q < Q) { CXP_T(0); }
        CXP_LANE;
        if (q > 0) {
            const int64_t win = window_of(q - 1);
            const bool bad = flags[(q - 1) & 1] != 0;
            if (win < n) {
                unsigned short* const out = OUT == 2 ? feat + (size_t)win * FEAT
                                                     : feat + (size_t)(win >> 1) * (2 * FEAT) + (int)(win & 1) * 32;
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const int co = 32 * wv + 16 * rt + 4 * g;
#pragma unroll
                    for (int ct = 0; ct < CX_NT; ++ct) {
                        const int t = 16 * ct + j;
                        float v[4] = {acc[rt][ct][0], acc[rt][ct][1], acc[rt][ct][2], acc[rt][ct][3]};
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaxf(fmaxf(v[r], cx_neighbour(v[r])), 0.f);       // MaxPool over (t, t + 1), ReLU; t = 74 has no partner and is dropped
                        unsigned lo[3], hi[3];
                        cx_split2(v[0], v[1], lo);
                        cx_split2(v[2], v[3], hi);
                        if (__builtin_expect(bad, 0)) {                // a non-finite sample: NaN in every term of the window's features
#pragma unroll
                            for (int p = 0; p < 3; ++p) lo[p] = hi[p] = 0x7fc07fc0u;
                        }
                        if ((j & 1) == 0 && (t >> 1) < 37) {
                            const int k = (t >> 1) * 128 + co;
                            if constexpr (OUT == 2) *reinterpret_cast<uint2*>(out + k) = make_uint2(lo[0], hi[0]);
                            else {
#pragma unroll
                                for (int p = 0; p < 3; ++p)
                                    *reinterpret_cast<uint2*>(out + p * plane_elems + (k >> 5) * 64 + (k & 31)) = make_uint2(lo[p], hi[p]);
                            }
                        }
                    }
                }
            }
            if (q == Q) break;
        }
        pre_layer(w0 + lane, pk.b[0], 32 * P1, g);
        {   // the staged window -> three-term planes [t + 1][channel]: 150 rows x 32 channel pairs (pairs 27..31 = channels 54..63 = 0).
            // A wave owns 8 pairs over all rows (lane = (row mod 8, pair)): the z-score's two reductions (utils/data_handler.py:55-56:
            // mean and unbiased std per channel over the 150 rows; fp64 as in load_windows) stay inside the wave -- 19 rows in
            // the lane, then three shuffles -- and need no barrier.  Per item: one 8-byte read, one split, three 4-byte stores.
            const int pr = 8 * wv + (lane & 7), r0 = lane >> 3;
            const bool real = pr < 27;
            const char* s = raw + r0 * (CH * 4) + (real ? pr : 0) * 8;
            float2 v[19];
#pragma unroll
            for (int m = 0; m < 19; ++m) v[m] = *reinterpret_cast<const float2*>(s + (r0 + 8 * m < WIN ? m : 0) * (8 * CH * 4));
            if constexpr (ZS) {
                auto across_rows = [](double x) {                      // sum over the 8 lanes (row mod 8) that hold this channel pair
                    x += __shfl_xor(x, 8);
                    x += __shfl_xor(x, 16);
                    return x + __shfl_xor(x, 32);
                };
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int m = 0; m < 19; ++m) {
                    const bool in = r0 + 8 * m < WIN;
                    s0 += in ? (double)v[m].x : 0.0;
                    s1 += in ? (double)v[m].y : 0.0;
                }
                const double mu0 = across_rows(s0) / 150.0, mu1 = across_rows(s1) / 150.0;
                double q0 = 0.0, q1 = 0.0;
#pragma unroll
                for (int m = 0; m < 19; ++m) {
                    const bool in = r0 + 8 * m < WIN;
                    const double d0 = (double)v[m].x - mu0, d1 = (double)v[m].y - mu1;
                    q0 += in ? d0 * d0 : 0.0;
                    q1 += in ? d1 * d1 : 0.0;
                }
                const float mean0 = (float)mu0, mean1 = (float)mu1;
                const float inv0 = 1.f / (float)sqrt(across_rows(q0) / 149.0), inv1 = 1.f / (float)sqrt(across_rows(q1) / 149.0);
#pragma unroll
                for (int m = 0; m < 19; ++m) { v[m].x = (v[m].x - mean0) * inv0; v[m].y = (v[m].y - mean1) * inv1; }
            }
            cc_f32x2 nz = {0.f, 0.f};                                  // non-finite scan: x * 0 is 0 for a finite x and NaN otherwise
#pragma unroll
            for (int m = 0; m < 19; ++m) nz = __builtin_elementwise_fma(cc_f32x2{v[m].x, v[m].y}, cc_f32x2{0.f, 0.f}, nz);
            if (real && (!(nz.x == 0.f) || !(nz.y == 0.f))) flags[q & 1] = 1;
#pragma unroll
            for (int m = 0; m < 19; ++m) {
                const int row = r0 + 8 * m;
                unsigned p[3];
                cx_split2(real ? v[m].x : 0.f, real ? v[m].y : 0.f, p);
                if (row < WIN) {
                    char* d = lds + cx_addr<128>(row + 1, 2 * pr);
#pragma unroll
                    for (int k = 0; k < 3; ++k) *reinterpret_cast<unsigned*>(d + k * CX_PLANE) = p[k];
                }
            }
            // rows 0 and 151 = the zero padding
            if (gtid < 48) reinterpret_cast<uint4*>(lds + (gtid >> 4) * CX_PLANE + ((gtid >> 3) & 1) * (151 * 128))[gtid & 7] = make_uint4(0, 0, 0, 0);
        }
        CXP_T(1);
        cxp_phase_end();
        // ================= phase 1: conv1 =================
        CXP_T(2);
        bias_acc();
        {   CXP_LANE;
            const int base = 16 * ct1 + j;
            const int sw[3] = {cx_swz<128>(base), cx_swz<128>(base + 1), cx_swz<128>(base + 2)};
            cx_layer<128, 2, true, CXP_ILV != 0>(lds + base * 128, sw, g, w0 + lane, acc, &wpre); }
        CXP_T(3);
        cxp_phase_end();
        // ================= phase 2: store =================
        CXP_T(4);
        {   CXP_LANE;
            if (gtid == 0) flags[(q + 1) & 1] = 0;                     // the next window's flag (last read in phase 0, next set in the next phase 0)
            pre_layer(w1 + lane, pk.b[1], 32 * P1, g);
            cx_store<128, false, WIN>(lds, acc, 32 * P1, ct1, j, g); }
        CXP_T(5);
        cxp_phase_end();
        // ================= phase 3: conv2 =================
        CXP_T(6);
        bias_acc();
        {   CXP_LANE;
            const int base = 16 * ct1 + j;
            const int sw[3] = {cx_swz<128>(base), cx_swz<128>(base + 1), cx_swz<128>(base + 2)};
            cx_layer<128, 2, true, CXP_ILV != 0>(lds + base * 128, sw, g, w1 + lane, acc, &wpre); }
        CXP_T(7);
        cxp_phase_end();
        // ================= phase 4: store + pool: rows 1..75 of the stage-2 layout (64 channels), row 76 = right pad =================
        CXP_T(8);
        {   CXP_LANE;
            pre_layer(w2 + lane, pk.b[2], 32 * wv, g);
            cx_store<128, true, WIN>(lds, acc, 32 * P1, ct1, j, g);
            if (gtid < 24) reinterpret_cast<uint4*>(lds + (gtid >> 3) * CX_PLANE + 76 * 128)[gtid & 7] = make_uint4(0, 0, 0, 0); }
        CXP_T(9);
        cxp_phase_end();
        // ================= phase 5: conv3 =================
        CXP_T(10);
        bias_acc();
        {   CXP_LANE;
            const int sw[3] = {cx_swz<128>(j), cx_swz<128>(j + 1), cx_swz<128>(j + 2)};
            cx_layer<128, 2, true, CXP_ILV != 0>(lds + j * 128, sw, g, w2 + lane, acc, &wpre); }
        CXP_T(11);
        cxp_phase_end();
        // ================= phase 6: ask for the next window; store (128 channels: 256-byte rows 1..75, rows 0 and 76 = padding) =================
        CXP_T(12);
        if (q + 1 < Q) cxp_issue_dma(dma_dst, window_src(q + 1), dma_voff);
        {   CXP_LANE;
            pre_layer(w3 + lane, pk.b[3], 32 * wv, g);
            cx_store<256, false, 75>(lds, acc, 32 * wv, 0, j, g);
            if (gtid < 96) {
                const int p = gtid >> 5, r = (gtid >> 4) & 1, sl = gtid & 15;
                reinterpret_cast<uint4*>(lds + p * CX_PLANE + (r ? 76 : 0) * 256)[sl] = make_uint4(0, 0, 0, 0);
            } }
        CXP_T(13);
        cxp_phase_end();
        // ================= phase 7: conv4 =================
        CXP_T(14);
        bias_acc();
        {   CXP_LANE;
            const int sw[3] = {cx_swz<256>(j), cx_swz<256>(j + 1), cx_swz<256>(j + 2)};
            cx_layer<256, 4, true, CXP_ILV != 0>(lds + j * 256, sw, g, w3 + lane, acc, &wpre); }
        CXP_T(15);
        cxp_phase_end_dma();                                           // (the next window has landed: every wave waits for its own pieces)
    }
#undef CXP_LANE
    cxp_phase_end();                                                   // closes the last phase 0
    if (grp == 0) { cxp_phase_end(); cxp_phase_end(); cxp_phase_end(); }
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
    const int64_t npairs = (n + 1) / 2;
    const int64_t cus = cxp_num_cu();
    const unsigned grid = (unsigned)(npairs < cus ? npairs : cus);
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
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(dce::g_trace_p), sizeof(unsigned long long) * 64 * nblocks);
}
#endif
