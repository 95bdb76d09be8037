// latency_mb.hip -- the LATENCY MODE for MICRO-BATCHES (option latency=1; DCE_FP32 contexts): 2 .. 32 windows through the whole net in ONE kernel.
//
// The reference's test script ships batch_size 30 (config/test_params.yaml:9; the loop of src/test.py:83-104,126), where the batch path's four launches
// cost 20.3 + 19.7 + 12.6 + 7.7 us of device time (profiles/r5z_latency.txt: a quarter-window conv segment per workgroup, fc.0 streaming its 38.8 MB
// behind the conv kernel's end, fc.3, the tail).  latency.hip serves ONE window per kernel with vectors; here one grid of 256 co-resident workgroups
// (one per CU, 84 KB of LDS each) carries a batch of n <= 32 windows through the same roles on MFMA tiles:
//   * conv role, workgroups 0 .. NC-1: the quarter-window segments of conv_wino_dev.h, one per workgroup (NC = 4 n), or -- up to 16 windows -- two
//     workgroups per segment, each finishing one half of conv4's output channels as in latency.hip (NC = 8 n <= 128).  Features -> HBM (write-through).
//   * fc.0 role, workgroups NC .. NC+127: 16 neurons each.  A wave holds ITS 18 or 19 granules of 32 k of the 16 rows in registers -- 152 VGPRs, the B
//     operands of v_mfma_f32_16x16x4_f32 -- requested before any feature exists: the 38.8 MB weight stream runs under the conv role.  Then the features of
//     16 windows stream past (A operands, one MFMA row tile), the eight waves' partial sums meet in LDS in a fixed order, bias + ReLU -> h1.
//     17 .. 32 windows are two row tiles: the second one (windows 16 ..) of neuron tile j runs on the workgroup 128 blocks away (same XCD) -- a conv workgroup
//     once its segment is stored, or an idle one -- which fetches the tile's 303 KB a second time, from L2 / the Infinity Cache (second_tile below).
//   * fc.3 (a tile = 16 neurons x 16 windows): the same scheme over K = 2048, then the tile's share of fc.6 -- partial logits over its 16 neurons for its
//     windows.  Up to 12 windows on idle CUs (weights requested at the start), up to 16 on the conv workgroups once their segment is done, from 17 on the
//     first 64 workgroups of the fc.0 role (32 neuron tiles x 2 row tiles).
//   * the last step on workgroup 0: the 32 partial logits per (window, class) added in order + bias, torch.max(output, 1), decimal2binary.
// The features and h1 cross in the QUAD layout [k / 4][window (32)][4 k]: what an MFMA A operand wants lane by lane (see MbTile::run).  The features' k runs
// position-major here (k' = 128 t + channel; fc.0's weight pack is in the same order): a conv lane's four channels are one quad, one 16-byte store.
// Hand-overs: data (features, h1, partial logits) in ordinary device memory, written with agent-scope (write-through, sc1) stores and read with agent-scope
// loads (served by L2, never by a CU's L1); behind them one 64-bit FLAG per producer in fine-grained memory that takes the request's number once the
// producer's stores are acknowledged (s_waitcnt vmcnt(0) + workgroup barrier).  A consumer's first wave polls the flags it needs; every wait has a
// deadline, and a kernel that runs into one raises the mailbox's error word and leaves (as latency.hip).
// Numerics: every sum is fp32 on the fp32 matrix pipe (exact fp32 fma chains) -- a wave's 576 / 608-k (256-k) chain, then eight partial sums in fixed order:
// deterministic, NOT the batch path's bits (its summation tree is another association); held to the fp32 tolerance against the CPU restatement.
#include "conv_wino_dev.h"
#include "fc6_chain.h"
#include <cstring>

namespace dce {

namespace {

constexpr int MB_FC0 = FC1 / 16;                                  // fc.0 workgroups: 128 tiles of 16 neurons
constexpr int MB_FC3 = FC2 / 16;                                  // fc.3 tiles of 16 neurons: 32
constexpr int MB_LDS = 84 * 1024;                                 // > half of the CU's 160 KB: one workgroup per CU
constexpr int MB_G0 = (FEAT / 32 + 7) / 8, MB_G3 = FC1 / 32 / 8;    // granules of 32 k a wave holds at most: fc.0 19 (148 = 4 x 19 + 4 x 18), fc.3 8
static_assert(MB_FC0 == 128 && MB_FC3 == 32 && FEAT % 32 == 0 && MB_G0 == 19 && MB_G3 * 8 * 32 == FC1, "the deal of K over the eight waves");
static_assert(HLDS_FLOATS * 4 + 64 <= MB_LDS && 8 * 2 * 256 * 4 + 4096 <= MB_LDS, "segment image / reduction scratch fit");
static_assert(4 * LATMB_MAX_N <= 128 && 8 * (LATMB_MAX_N / 2) <= 128 && LATMB_MAX_N * NCLS <= 512, "conv workgroups beside the 128 of fc.0; one thread per (window, class)");

typedef float mb_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned mb_u32x4 __attribute__((ext_vector_type(4)));
#ifndef MB_AUX
#define MB_AUX 16                                                 // cache policy of the A-operand loads: 16 = sc1 (agent scope)
#endif

__device__ __forceinline__ void mb_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// everything this workgroup stored has been acknowledged by the memory side before its flag takes the request's number
__device__ __forceinline__ void mb_post(unsigned long long* flag, unsigned long long seq, int tid)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave 0 polls flags[0 .. count) until all carry seq (count <= 128: two per lane); false: deadline (100 MHz ticks)
__device__ __forceinline__ bool mb_wait(const unsigned long long* flags, int count, unsigned long long seq, unsigned long long deadline, int* lflag, int tid)
{
    if (tid < 64) {
        const unsigned long long t0 = wall_clock64();
        int ok = 1;
        for (unsigned it = 0;; ++it) {
            bool mine = true;
            for (int i = tid; i < count; i += 64) mine = mine && __hip_atomic_load(flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == seq;
            if (__builtin_amdgcn_ballot_w64(!mine) == 0) break;
            if ((it & 15) == 15 && wall_clock64() - t0 > deadline) { ok = 0; break; }
            __builtin_amdgcn_s_sleep(2);
        }
        if (tid == 0) *lflag = ok;
    }
    __syncthreads();
    const int ok = *lflag;
    __syncthreads();                                                   // (the word is free again)
    return ok != 0;
}

// One 16-neuron tile of a Linear layer for up to 16 MT windows: out[m][16 j + i] = act(sum_k A[m][k] W[16 j + i][k] + bias).
// K is dealt out in GRANULES of 32 k: wave w owns granules g0(w) .. g0(w) + ng(w) - 1 (fc.0: 148 granules = 4 x 19 + 4 x 18; fc.3: 64 = 8 x 8).  Lane
// (i = lane & 15, g = lane >> 4) holds W[16 j + i][32 G + 16 (e >> 2) + 4 g + (e & 3)], e = 0..7, of every granule G of its wave -- two 16-byte pieces, each
// load instruction reading 64 contiguous bytes of a row -- and reads the same k of window m = lane & 15 of each row tile: MFMA step (G, e) multiplies the
// four k = 32 G + 16 (e >> 2) + 4 g' + (e & 3), g' = 0..3 -- every k once.  The weights are loaded by load() BEFORE the producers' flags are awaited: 8 NG VGPRs (152 for fc.0) that the stream of
// the layer's weights fills while the producers still compute.  Result: red[(16 mt + m) * 16 + i] (LDS), bias and ReLU applied.
template <int NG> struct MbTile {
    float4 wr[NG][2];
    int g0, ng, rot;
    // Wp: the layer's weights PACKED for this kernel (latmb_pack_host): [tile j][wave w][granule slot (NG)][half (2)][lane (64)][4 floats] -- a load
    // instruction reads ONE contiguous KB, a wave 2 NG of them back to back, a tile 16 NG KB: the 38.8 MB of fc.0 leave HBM as 128 long runs (from the
    // (2048, 4736) matrix itself a load instruction took 64 bytes from each of 16 rows 18.9 KB apart: 2.5 TB/s cold, profiles/r6f_latmb_cold.txt)
    __device__ __forceinline__ void load(const float* __restrict__ Wp, int K, int j, int w, int lane)
    {
        const int total = K / 32, lo = total / 8, extra = total % 8;       // waves 0 .. extra-1 take lo + 1 granules
        ng = lo + (w < extra ? 1 : 0);
        g0 = w * lo + (w < extra ? w : extra);
        // Tile j walks its wave's granules from a start of its own (step q takes granule (q + rot) mod ng): at any moment the 128 tiles then ask L2
        // for different lines of the operand rows instead of all for the same ones.  The order of a wave's chain depends on j alone: deterministic.
        rot = (j * 7) % ng;
        // lane g's 8 k of a granule: 4 g .. 4 g + 3 and 16 + 4 g .. 16 + 4 g + 3
        const float4* wp = reinterpret_cast<const float4*>(Wp) + (size_t)(j * 8 + w) * NG * 128 + lane;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            const int pg = granule(q);
            wr[q][0] = wp[pg * 128]; wr[q][1] = wp[pg * 128 + 64];
        }
    }
    __device__ __forceinline__ int granule(int q) const                  // (wave-uniform; a wave one granule short: step NG - 1 repeats a granule and is never multiplied)
    {
        const int p = (q < ng ? q : ng - 1) + rot;
        return p >= ng ? p - ng : p;
    }
    template <int MT>
    __device__ __forceinline__ void run(const float* __restrict__ A, int K, int n, const float* __restrict__ bias, int j, int w, int lane, int tid, float* __restrict__ red, int mt0 = 0) const
    {
        constexpr int PF = MT == 2 ? 4 : 7;                                // A operands requested this many granules ahead
        static_assert(PF < NG, "prefetch ring shorter than the wave's K range");
        // A operands: 16-byte agent-scope (sc1) buffer loads -- served by L2, never by this CU's L1; the producers stored write-through.  A lives in the
        // QUAD layout [k / 4][window (32)][4]: lane (m, g) reads the four k of quad 8 G + g (and 8 G + 4 + g) of window m -- sixteen neighbouring lanes read
        // 256 contiguous bytes.  (In a (windows, K) matrix the sixteen lanes next to each other would read sixteen different rows: one cache line per lane
        // and load, 0.4 us per window and tile -- the first build of this kernel, profiles/r6e_latmb.txt.)  Windows past n hold whatever an earlier call
        // left there: a row of A feeds its own row of the result only, and those rows are never stored.
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (K / 4) * LATMB_MAX_N * 16, 0x00027000);
        unsigned ao[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) ao[mt] = (unsigned)((((8 * g0 + (lane >> 4)) * LATMB_MAX_N) + 16 * (mt0 + mt) + (lane & 15)) * 16);      // mt0: the first row tile of this call
        auto lda = [&](int mt, int q, float4 (&d)[2]) {
            const int so = 8 * LATMB_MAX_N * 16 * granule(q);              // (scalar: the instruction's soffset)
            const mb_u32x4 lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ao[mt], so, MB_AUX);
            const mb_u32x4 hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, ao[mt] + 4 * LATMB_MAX_N * 16, so, MB_AUX);
            d[0] = make_float4(__uint_as_float(lo[0]), __uint_as_float(lo[1]), __uint_as_float(lo[2]), __uint_as_float(lo[3]));
            d[1] = make_float4(__uint_as_float(hi[0]), __uint_as_float(hi[1]), __uint_as_float(hi[2]), __uint_as_float(hi[3]));
        };
        mb_f32x4 acc[MT];
        float4 av[PF][MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = mb_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < PF; ++q)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) lda(mt, q, av[q][mt]);
#pragma unroll
        for (int q = 0; q < NG; ++q) {
            float4 cur[MT][2];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                cur[mt][0] = av[q % PF][mt][0]; cur[mt][1] = av[q % PF][mt][1];
                if (q + PF < NG) lda(mt, q + PF, av[q % PF][mt]);
            }
            __builtin_amdgcn_sched_barrier(0);                             // (the requests of granule q + PF go out AHEAD of granule q's MFMAs: left alone, hipcc sinks them to their use)
            if (q < NG - 1 || q < ng) {                                    // (wave-uniform: the last granule of a wave that is one short)
#pragma unroll
                for (int e = 0; e < 8; ++e)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float4 xv = cur[mt][e >> 2], yv = wr[q][e >> 2];
                        const float x = (e & 3) == 0 ? xv.x : (e & 3) == 1 ? xv.y : (e & 3) == 2 ? xv.z : xv.w;
                        const float y = (e & 3) == 0 ? yv.x : (e & 3) == 1 ? yv.y : (e & 3) == 2 ? yv.z : yv.w;
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[mt], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // D[m = 4 g + r][i] in element r.  The eight waves' partial sums -> LDS, added in wave order
        float* part = red + 512;                                           // [wave][mt][m][i]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) part[((w * MT + mt) * 16 + 4 * (lane >> 4) + r) * 16 + (lane & 15)] = acc[mt][r];
        __syncthreads();
        if (tid < MT * 256) {
            const int mt = tid >> 8, mi = tid & 255;
            float v = part[(mt * 16) * 16 + mi];
#pragma unroll
            for (int ww = 1; ww < 8; ++ww) v += part[((ww * MT + mt) * 16) * 16 + mi];
            v += bias[16 * j + (tid & 15)];
            red[tid] = v < 0.f ? 0.f : v;                                  // ReLU; keeps NaN, as the batch kernels and torch do
        }
        __syncthreads();
    }
};

#define MB_TRACE(k) do { if (a.trace && tid == 0) __hip_atomic_store(a.trace + (k), (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)

}  // namespace

// ZS: raw sequence rows (window i = rows i .. i + 149, z-score fused) instead of pre-normalised windows
template <bool ZS>
__global__ __launch_bounds__(512)
void latency_mb_kernel(LatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = a.mb_n, b = blockIdx.x;
    const int per = a.mb_chalf ? 8 : 4, NC = per * n;                      // conv workgroups
    const unsigned long long seq = a.seq, spin = a.deadline_ticks;
    unsigned long long* const fconv = a.mb_flags;                          // [128] conv workgroup b has stored its features
    unsigned long long* const ffc0 = a.mb_flags + 128;                     // [128] fc.0 tile j has stored its columns of h1
    unsigned long long* const ffc3 = a.mb_flags + 256;                     // [64]  fc.3 tile t has stored its partial logits
    unsigned long long* const ffc0b = a.mb_flags + 576;                    // [128] 17 .. 32 windows: fc.0 tile j of the SECOND row tile (windows 16 ..) has stored its columns of h1
    const bool two = n > 16;                                               // two row tiles of 16 windows
    int* const lflag = reinterpret_cast<int*>(lds + MB_LDS / 4 - 4);
    auto fail_out = [&]() { if (tid == 0) __hip_atomic_store(&a.mbox->error, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); };

    // fc.3 tiles first, first + stride, ..: false = a wait ran into its deadline
    const int ntile = MB_FC3 * (n > 16 ? 2 : 1);
    const bool dedicated = 256 - MB_FC0 - NC >= ntile;                     // enough idle CUs behind the fc.0 role: they take fc.3, with its weights requested at the start
    auto fc3_tiles = [&](int first, int stride) -> bool {
        int waited = 0;
        for (int t3 = first; t3 < ntile; t3 += stride) {
            const int nt = t3 & (MB_FC3 - 1), mt0 = t3 >> 5;                    // neuron tile, row tile
            MbTile<MB_G3> t;
            t.load(a.mb_w2, FC1, nt, w, lane);
            const float w3v = a.w3[(size_t)(tid & 15) * FC2 + 16 * nt + ((tid >> 4) & 15)];      // thread (cls = tid & 15, i = (tid >> 4) & 15): W3[cls][16 nt + i]
            if (!((waited >> mt0) & 1)) { if (!mb_wait(mt0 ? ffc0b : ffc0, MB_FC0, seq, spin, lflag, tid)) { fail_out(); return false; } waited |= 1 << mt0; }
            if (t3 == (two ? MB_FC3 : 0)) MB_TRACE(8);                    // (17 .. 32 windows: the second row tile, whose fc.0 ran behind a conv segment)
            t.run<1>(a.mb_h1, FC1, n, a.b2, nt, w, lane, tid, lds, mt0);
            // lds[m * 16 + i] = ReLU(fc.3)[16 mt0 + m][16 nt + i].  Partial logits of the tile: thread (m, cls) walks its 16 neurons in order
            {
                float* w3t = lds + 1024;                                       // [i][cls]
                if (tid < 256) w3t[((tid >> 4) & 15) * 16 + (tid & 15)] = w3v;
                __syncthreads();
                const int m = tid >> 4, cls = tid & 15;
                if (tid < 256 && 16 * mt0 + m < n) {
                    float p = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) p = fmaf(lds[m * 16 + i], w3t[i * 16 + cls], p);
                    mb_st(a.mb_plt + ((size_t)nt * LATMB_MAX_N + 16 * mt0 + m) * NCLS + cls, p);
                }
            }
            mb_post(&ffc3[t3], seq, tid);
        }
        return true;
    };

    // red[m * 16 + i] = h1[16 mt0 + m][16 j + i] -> the quad layout fc.3's tiles read
    auto store_h1 = [&](int j, int mt0) {
        const int m = 16 * mt0 + (tid >> 4), k = 16 * j + (tid & 15);
        if (tid < 256 && m < n) mb_st(a.mb_h1 + ((size_t)(k >> 2) * LATMB_MAX_N + m) * 4 + (k & 3), lds[tid]);      // (one row tile: 256 values)
    };
    // 17 .. 32 windows: fc.0's SECOND row tile (windows 16 ..) of neuron tile j2 on the 128 workgroups outside the fc.0 role -- the conv workgroups once their
    // segment is stored, and the idle ones.  The fc.0 role then multiplies 16 windows, not 32: at 32 windows its 2368 MFMAs per workgroup were 7.9 us of the fp32
    // matrix pipe on half the chip while the other half waited for h1.  The tile's 303 KB of weights come a second time, from L2 / the Infinity Cache (the fc.0
    // workgroup of the same tile sits 128 blocks away: the same XCD), requested before the wait for the other segments' features.
    auto second_tile = [&]() -> bool {
        const int j2 = (b - NC) & (MB_FC0 - 1);
        MbTile<MB_G0> t;
        t.load(a.mb_w1, FEAT, j2, w, lane);
        if (!mb_wait(fconv, NC, seq, spin, lflag, tid)) { fail_out(); return false; }
        t.run<1>(a.mb_feat, FEAT, n, a.b1, j2, w, lane, tid, lds, 1);
        store_h1(j2, 1);
        mb_post(&ffc0b[j2], seq, tid);
        return true;
    };

    if (b >= NC) {
        if (b >= NC + MB_FC0) {                                            // idle CUs: fc.3 of a small batch if there are enough of them; a second row tile of fc.0 from 17 windows
            if (dedicated) (void)fc3_tiles(b - NC - MB_FC0, 256 - MB_FC0 - NC);
            else if (two) (void)second_tile();
            return;
        }
        // ======================================================================== fc.0 role: tile j = 16 neurons
        const int j = b - NC;
        if (a.fc_delay_ticks) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < a.fc_delay_ticks) __builtin_amdgcn_s_sleep(8); }
        MbTile<MB_G0> t;
        const unsigned long long t_req = a.trace ? wall_clock64() : 0ull;
        t.load(a.mb_w1, FEAT, j, w, lane);
        if (j == 0) MB_TRACE(4);
        if (a.trace) {                                                     // traced runs: when every wave of this workgroup has its 303 KB of fc.0 (the whole stream: the latest of the 128 tiles)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_store(a.mb_flags + 320 + j, t_req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(a.mb_flags + 448 + j, (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (j == 0) MB_TRACE(5);
        }
        if (!mb_wait(fconv, NC, seq, spin, lflag, tid)) { fail_out(); return; }
        if (j == 0) MB_TRACE(6);
        t.run<1>(a.mb_feat, FEAT, n, a.b1, j, w, lane, tid, lds);          // (17 .. 32 windows: the first row tile; the second one runs on the other 128 workgroups)
        store_h1(j, 0);
        mb_post(&ffc0[j], seq, tid);
        if (j == 0) MB_TRACE(7);
        if (two) (void)fc3_tiles(j, MB_FC0);                               // fc.3's 64 tiles on the first 64 of this role: done first, their fc.3 weights arrive under the wait for h1
        return;
    }

    // ============================================================================ conv role: segment (b % per) of window b / per
    {
        const int win = b / per, r = b % per;
        const int sg = a.mb_chalf ? r >> 1 : r, chalf = a.mb_chalf ? (r & 1) : -1;
        if (b == 0) MB_TRACE(1);
        conv_seg_body<ZS, 4, 2, 1, false, false, true, LATMB_MAX_N>(lds, a.src, win, sg, a.pk, a.mb_feat, LayerTaps{}, nullptr, chalf);
        if (b == 0) MB_TRACE(2);
        if (a.trace && b == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); MB_TRACE(14); __syncthreads(); MB_TRACE(15); }   // (traced runs: wave 0's stores acknowledged; every wave of the workgroup here)
        mb_post(&fconv[b], seq, tid);                                      // (with chalf the waves of the other channel half came back early and wait at its barrier)
        if (b == 0) MB_TRACE(3);
    }
    // ---- fc.3 on the conv workgroups (unless idle CUs took it: `dedicated`): a tile = 16 neurons x 16 windows (32 neuron tiles x one or two row tiles: with 17 .. 32 windows the two row tiles of
    //      a neuron tile go to two workgroups -- 1024 MFMAs each instead of 2048 on one), tile t3 on workgroup t3, t3 + NC, ..; then the tile's share
    //      of fc.6: partial logits over its 16 neurons for its windows
    if (two) { if (!second_tile()) return; }
    else if (!dedicated && !fc3_tiles(b, NC < ntile ? NC : ntile)) return;

    if (b != 0) return;
    // ============================================================================ workgroup 0: logits, torch.max(output, 1), decimal2binary
    MB_TRACE(9);
    if (!mb_wait(ffc3, ntile, seq, spin, lflag, tid)) { fail_out(); return; }
    MB_TRACE(10);
    {
        const int m = tid >> 4, cls = tid & 15;
        float* lg = lds;                                                   // [m][cls]
        if (m < n) {
            float pv[MB_FC3];                                              // (all 32 requests in flight at once, then the ordered sum)
#pragma unroll
            for (int t3 = 0; t3 < MB_FC3; ++t3) pv[t3] = __builtin_nontemporal_load(a.mb_plt + ((size_t)t3 * LATMB_MAX_N + m) * NCLS + cls);
            float v = pv[0];
#pragma unroll
            for (int t3 = 1; t3 < MB_FC3; ++t3) v += pv[t3];
            v += a.b3[cls];
            lg[tid] = v;
            if (a.logits) a.logits[tid] = v;
            if (a.packed) reinterpret_cast<float*>(a.packed + (size_t)m * PACKED_ROW)[cls] = v;
        }
        __syncthreads();
        if (tid < n) {
            const int best = fc6_argmax16(lg + tid * NCLS);
            const uchar4 cb = make_uchar4((best >> 3) & 1, (best >> 2) & 1, (best >> 1) & 1, best & 1);
            if (a.pred) a.pred[tid] = best;
            if (a.contacts) reinterpret_cast<uchar4*>(a.contacts)[tid] = cb;
            if (a.packed) *reinterpret_cast<uchar4*>(a.packed + (size_t)tid * PACKED_ROW + 4 * NCLS) = cb;
        }
    }
    MB_TRACE(11);
}

// (rows, K) Linear weights -> the kernel's order: out[((((j * 8 + w) * NG + s) * 2 + h) * 64 + lane) * 4 + e] = W[16 j + (lane & 15)][col(32 (g0(w) + s) + 16 h + 4 (lane >> 4) + e)]
// for granule slot s < ng(w) of wave w (K / 32 granules dealt out over eight waves: the first K / 32 % 8 waves take one more); unused slots stay zero.
// chan > 0 (fc.0): the kernel's k' runs position-major over the conv stack's output, k' = chan t + channel (conv_wino_dev.h XPOSE), torch's flatten
// channel-major: col(k') = (k' % chan) (K / chan) + k' / chan; chan = 0: col(k') = k'
size_t latmb_pack_floats(int rows, int K) { return (size_t)(rows / 16) * 8 * ((K / 32 + 7) / 8) * 512; }
void latmb_pack_host(const float* W, int rows, int K, float* out, int chan)
{
    const int total = K / 32, lo = total / 8, extra = total % 8, NG = (total + 7) / 8;
    memset(out, 0, latmb_pack_floats(rows, K) * sizeof(float));
    for (int j = 0; j < rows / 16; ++j)
        for (int w = 0; w < 8; ++w) {
            const int ng = lo + (w < extra ? 1 : 0), g0 = w * lo + (w < extra ? w : extra);
            for (int s = 0; s < ng; ++s)
                for (int h = 0; h < 2; ++h)
                    for (int lane = 0; lane < 64; ++lane) {
                        const float* row = W + (size_t)(16 * j + (lane & 15)) * K;
                        const int k0 = 32 * (g0 + s) + 16 * h + 4 * (lane >> 4);
                        float* dst = out + ((((size_t)(j * 8 + w) * NG + s) * 2 + h) * 64 + lane) * 4;
                        for (int e = 0; e < 4; ++e) { const int k = k0 + e; dst[e] = row[chan > 0 ? (k % chan) * (K / chan) + k / chan : k]; }
                    }
        }
}

hipError_t init_latency_mb()
{
    hipError_t e;
    for (const void* k : {reinterpret_cast<const void*>(&latency_mb_kernel<false>), reinterpret_cast<const void*>(&latency_mb_kernel<true>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, MB_LDS)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_latency_mb(int zscore, const LatArgs& a, hipStream_t st)
{
    if (a.mb_n < 2 || a.mb_n > LATMB_MAX_N) return hipErrorInvalidValue;
    plan_note(zscore ? "latency_mb_zs" : "latency_mb");
    if (zscore) hipLaunchKernelGGL(latency_mb_kernel<true>, dim3(256), dim3(512), MB_LDS, st, a);
    else        hipLaunchKernelGGL(latency_mb_kernel<false>, dim3(256), dim3(512), MB_LDS, st, a);
    return hipGetLastError();
}

}  // namespace dce
