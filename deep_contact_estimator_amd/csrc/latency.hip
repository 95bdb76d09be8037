// latency.hip -- the LATENCY MODE of the path (option latency=1; DCE_FP32 contexts): ONE window through the whole net in ONE kernel.
//
// The reference ships batch_size 1 (config/inference_one_seq_params.yaml:10; the runner of README.md:67-83 evaluates one window per
// new sample), where the batch path's four launches cost 17.6 + 12.8 + 7.6 + 4.8 us of device time: a quarter-window conv segment per
// workgroup, then fc.0 streaming its 38.8 MB of weights behind four 1200-link fmaf chains, fc.3, fc.6.  Here one grid of 256
// workgroups -- one per CU, all co-resident (84 KB of LDS each) -- splits into two ROLES:
//   * workgroups 0..7, "conv":  the four quarter-window segments of conv_wino_dev.h (z-score + conv1..4 + pools -> features), two
//                               workgroups per segment: both compute conv1..3 of it, each finishes one half of conv4's output channels;
//   * workgroups 8..239, "fc.0": 2048 neurons = 192 x 9 + 40 x 8 rows;
//   * workgroups 240..255, "fc.3": 32 neurons each; the first of them also fc.6 + argmax + contact bits.
// An fc workgroup holds the weights of ITS neurons in registers (fc.0: 9 rows x 4736 floats over 512 threads = 108 VGPRs) -- they do not
// depend on the window, so they are requested BEFORE the features exist and their 38.8 MB stream hides under the conv role's 17 us (one
// shot), or stay resident from request to request (service: fc.0 then costs no memory traffic at all).  Every neuron's sum is split over
// the lanes (K / 512 per thread) and folded by a fixed tree -- DPP inside a row of 16 lanes, LDS across rows and waves, then 32 ordered
// adds -- so results are deterministic but NOT the batch path's bits (its summation tree, fc_tree.h, buys bit-identity across batch
// sizes with four long chains); the mode is held to the fp32 tolerance against the CPU restatement instead (tests/test_round5_gpu.py).
// The layers meet through fine-grained device memory without any cache maintenance (see below): the features behind an arrival
// counter of the eight conv workgroups, h1 and h2 as (value, request number) words that their readers poll.  Every wait has a deadline; a kernel that runs into one raises the mailbox's error word and leaves.
//
// Two forms of the same kernel:
//   one shot (dce_forward_windows / dce_infer_sequence with n = 1): source window in device memory, results to device pointers,
//     stream-ordered like any other launch;
//   service  (dce_online_push): the kernel stays resident; the conv workgroups poll a mailbox in pinned host memory for the next
//     sample (216 bytes), keep the last 150 samples in LDS, and the first fc workgroup writes the estimate back to the mailbox: no launch,
//     no copy, no stream operation per push.  It leaves on a quit request, or by itself after `idle` without one (the host relaunches
//     it on the next push; the sample history also lives in device memory and is reloaded).
#include "conv_wino_dev.h"
#include "fc6_chain.h"

namespace dce {

namespace {

constexpr int LAT_CONV = 8;                                   // conv workgroups: four quarter-window segments x two halves of conv4's output channels
constexpr int LAT_FC0 = 232, LAT_FC3 = 16, LAT_GRID = LAT_CONV + LAT_FC0 + LAT_FC3;
constexpr int LAT_FC0_9 = FC1 - 8 * LAT_FC0;                  // fc.0 workgroups that carry 9 rows (192); the others 8
constexpr int LAT_R = 9;                                      // fc.0 rows per workgroup at most
constexpr int LAT_N3 = FC2 / LAT_FC3;                         // fc.3 neurons per fc.3 workgroup (32)
constexpr int LAT_S = 3;                                      // float4 slots of a 4736-float row per thread: index tid + 512 s < 1184
constexpr int LAT_HIST = WIN * CH;                            // the service's sample history, floats
constexpr int LAT_LDS = 84 * 1024;                            // > half of the CU's 160 KB: one workgroup per CU
static_assert((HLDS_FLOATS + LAT_HIST) * 4 + 64 <= LAT_LDS, "segment image + sample history fit");
static_assert(LAT_GRID == 256 && LAT_FC0_9 >= 0 && LAT_FC0_9 <= LAT_FC0 && 9 * LAT_FC0_9 + 8 * (LAT_FC0 - LAT_FC0_9) == FC1 && LAT_N3 * LAT_FC3 == FC2 &&
              FEAT / 4 <= 512 * LAT_S && FC1 / 4 == 512, "deal of the neurons");

template <int CTRL> __device__ __forceinline__ float lat_dpp(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// the sum over a row of 16 lanes, in every lane of the row: pairs, quads, halves (row_half_mirror), the row (row_mirror)
__device__ __forceinline__ float lat_row_sum(float v)
{
    v += lat_dpp<0xB1>(v); v += lat_dpp<0x4E>(v); v += lat_dpp<0x141>(v); v += lat_dpp<0x140>(v);
    return v;
}
__device__ __forceinline__ float lat_dot4(float4 a, float4 b, float acc) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc)))); }

// NV per-thread partial sums -> NV totals (in LDS red[0..NV)): rows of 16 lanes by DPP, the 32 rows of the workgroup in order
template <int NV>
__device__ __forceinline__ void lat_block_sum(const float (&v)[NV], float* __restrict__ red, int tid)
{
    const int lane = tid & 63, row = tid >> 4;               // 32 rows of 16 lanes
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float s = lat_row_sum(v[i]);
        if ((lane & 15) == 0) red[NV + i * 32 + row] = s;
    }
    __syncthreads();
    if (tid < NV) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 32; ++r) t += red[NV + tid * 32 + r];
        red[tid] = t;
    }
    __syncthreads();
}

// ---- hand-overs between workgroups WITHOUT cache maintenance.  An agent-scope release / acquire fence on this part is a write-back /
// invalidate of the XCD's whole L2 (buffer_wbl2 / buffer_inv sc1): measured, three hand-overs built on them made a one-window call
// 81 us, and a mailbox poll with acquire semantics invalidated the L2 under the conv role's weights on every iteration.  Instead the
// few KB that cross workgroups -- features, h1, h2, the counters -- live in FINE-GRAINED device memory (uncached in L2) and are moved by
// agent-scope relaxed atomics (sc1: performed at the memory side), ordered by nothing more than s_waitcnt + the workgroup barrier:
//   producer: stores -> s_waitcnt vmcnt(0) (each acknowledged by the memory side) -> barrier -> counter += 1
//   consumer: counter reached -> barrier -> loads
__device__ __forceinline__ float lat_ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float4 lat_ld4(const float* p)
{
    const unsigned long long a = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float4(__uint_as_float((unsigned)a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned)b), __uint_as_float((unsigned)(b >> 32)));
}
__device__ __forceinline__ void lat_st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lat_stores_done() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

// arrive: everything this workgroup stored has been acknowledged before the count moves
__device__ __forceinline__ void lat_arrive(unsigned long long* ctr, int tid)
{
    lat_stores_done();
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until *ctr >= target: 1 reached, 0 the service is quitting, -1 deadline (ticks of the 100 MHz wall clock)
__device__ __forceinline__ int lat_wait(const unsigned long long* ctr, unsigned long long target, const unsigned* quit,
                                        unsigned long long deadline, int* __restrict__ flag, int tid)
{
    if (tid == 0) {
        int ok = 1;
        const unsigned long long t0 = wall_clock64();
        for (unsigned it = 0; __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++it) {
            if ((it & 15) == 15) {
                if (quit && __hip_atomic_load(quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                if (wall_clock64() - t0 > deadline) { ok = -1; break; }
            }
            __builtin_amdgcn_s_sleep(4);
        }
        *flag = ok;
    }
    __syncthreads();
    const int ok = *flag;
    __syncthreads();                                         // (the flag word is free again)
    return ok;
}

// h1 and h2 cross workgroups as (value, tag) PAIRS in one 64-bit word, tag = the request's number: a consumer thread polls exactly the
// words it needs until they carry this request's tag -- no counter (252 arrivals on one address were 3.9 us of the 29), no barrier.
__device__ __forceinline__ void lat_st_ll(unsigned long long* p, float v, unsigned tag)
{
    __hip_atomic_store(p, ((unsigned long long)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int N>
__device__ __forceinline__ bool lat_ld_ll(const unsigned long long* p, unsigned tag, float (&out)[N], const unsigned* quit, unsigned long long deadline)
{
    const unsigned long long t0 = wall_clock64();
    for (unsigned it = 0;; ++it) {
        unsigned long long wv[N];
#pragma unroll
        for (int i = 0; i < N; ++i) wv[i] = __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < N; ++i) ok = ok && (unsigned)(wv[i] >> 32) == tag;
        if (ok) {
#pragma unroll
            for (int i = 0; i < N; ++i) out[i] = __uint_as_float((unsigned)wv[i]);
            return true;
        }
        if ((it & 31) == 31) {
            if (quit && __hip_atomic_load(quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
            if (wall_clock64() - t0 > deadline) return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

#define LAT_TRACE(k) do { if (a.trace && tid == 0) __hip_atomic_store(a.trace + (k), (unsigned long long)wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)

}  // namespace

// MODE 0: one shot, pre-normalised window; 1: one shot, raw rows (z-score fused); 2: service
template <int MODE>
__global__ __launch_bounds__(512)
void latency_kernel(LatArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    int* flag = reinterpret_cast<int*>(lds + HLDS_FLOATS + LAT_HIST);          // one word behind both roles' images
    LatSync* const sy = a.sync;
    constexpr bool SERVICE = MODE == 2;
    const unsigned long long spin = a.deadline_ticks;

    if (blockIdx.x < LAT_CONV) {
        // ======================================================================== conv role: segment blockIdx.x of the window
        const int sg = blockIdx.x >> 1, chalf = blockIdx.x & 1;                 // segment, half of conv4's output channels
        const bool lead = blockIdx.x == 0;
        if constexpr (!SERVICE) {
            if (lead) LAT_TRACE(1);
            conv_seg_body<MODE == 1, 4, 2, 1, false, false, true>(lds, a.src, 0, sg, a.pk, a.feat, LayerTaps{}, nullptr, chalf);
            if (lead) LAT_TRACE(2);
            lat_arrive(&sy->feat, tid);                                         // (the waves of the other channel half came back early and wait at its barrier)
            if (lead) LAT_TRACE(3);
            return;
        } else {
            float* hist = lds + HLDS_FLOATS;                                    // [150][54], row (head + t) % 150 = sample t of the window
            LatMailbox* const mb = a.mbox;
            int head = a.hist_state[0], count = a.hist_state[1];
            for (int i = tid; i < LAT_HIST; i += 512) hist[i] = a.hist[i];
            // The host posts its first request once `alive` is up, and the lead then advances the device copy of the history (a.hist, a.hist_state).
            // A conv workgroup that was dispatched late -- another stream held its CU -- must not find that copy already advanced: it would apply the
            // request it then sees a second time and keep a history one row off for the life of the service.  So every conv workgroup ARRIVES on a start
            // counter once its snapshot is in its registers / LDS, and the lead raises `alive` only when all eight have (round 5's advice).
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(&sy->started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lead && tid == 0) {
                const unsigned long long t0 = wall_clock64();
                bool up = true;
                while (__hip_atomic_load(&sy->started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)LAT_CONV) {
                    if (wall_clock64() - t0 > 500000000ull) { up = false; break; }      // 5 s: the host's own limit for the start (it then reports the failure)
                    __builtin_amdgcn_s_sleep(8);
                }
                if (up) __hip_atomic_store(&mb->alive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __syncthreads();
            unsigned last = a.req_base;
            for (;;) {
                // ---- the next request: thread 0 polls the mailbox (and, beside workgroup 0, the quit word workgroup 0 raises when it
                //      has waited `idle` in vain: only one workgroup decides that, or the four could disagree)
                if (tid == 0) {
                    int kind = -1;
                    const unsigned long long t0 = wall_clock64();
                    for (;;) {
                        const unsigned r = __hip_atomic_load(&mb->req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (r != last) { last = r; kind = (int)__hip_atomic_load(&mb->kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
                        if (lead ? (wall_clock64() - t0 > a.idle_ticks)
                                    : (__hip_atomic_load(&sy->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { kind = 2; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                    flag[0] = kind; flag[1] = (int)last;
                }
                __syncthreads();
                const int kind = flag[0];
                const unsigned req = (unsigned)flag[1];
                __syncthreads();
                if (kind == 2) break;
                if (lead) LAT_TRACE(0);
                // ---- the sample -> the history (every conv workgroup keeps its own copy; workgroup 0 also keeps the device copy a
                //      relaunched service starts from)
                if (tid < CH) {
                    const float v = __hip_atomic_load(&mb->sample[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (the host wrote it before the request number)
                    hist[head * CH + tid] = v;
                    if (lead) a.hist[head * CH + tid] = v;
                }
                head = head + 1 == WIN ? 0 : head + 1;
                count = count < WIN ? count + 1 : WIN;
                if (lead && tid == 0) { a.hist_state[0] = head; a.hist_state[1] = count; }
                __syncthreads();
                if (tid == 0) __hip_atomic_store(&mb->ack[blockIdx.x], req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // this workgroup has taken the sample
                if (kind != 1 || count < WIN) continue;
                // ---- the window's raw samples this thread owns (load_windows' map: rows t = 4 m + g of channel c), then the segment
                float x[1][38];
                {
                    const int t216 = tid < 4 * CH ? tid : 0, c = t216 % CH, g = t216 / CH;
#pragma unroll
                    for (int m = 0; m < 38; ++m) {
                        const int t = 4 * m + g;
                        int r = head + t; r = r >= WIN ? r - WIN : r;
                        x[0][m] = t < WIN ? hist[(r < WIN ? r : 0) * CH + c] : 0.f;
                    }
                }
                if (lead) LAT_TRACE(1);
                conv_seg_body<true, 4, 2, 1, false, true, true>(lds, nullptr, 0, sg, a.pk, a.feat, LayerTaps{}, x, chalf);
                if (lead) LAT_TRACE(2);
                lat_arrive(&sy->feat, tid);
                if (lead) LAT_TRACE(3);
            }
            // ---- leaving: release the fc role, tell the host
            if (tid == 0) {
                __hip_atomic_store(&sy->quit, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (lead) { lat_stores_done(); __hip_atomic_store(&mb->alive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            }
            return;
        }
    }

    // ============================================================================ fc.0 role: workgroups 4 .. 239
    constexpr unsigned ERR = 1u;
    auto fail_out = [&]() {                                                        // a wait ended without its data: deadline (error) or the service is quitting
        if (tid == 0 && !(SERVICE && __hip_atomic_load(&sy->quit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))
            __hip_atomic_store(&a.mbox->error, ERR, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    float* red = lds;                                                              // reduction scratch of lat_block_sum
    const int w = blockIdx.x - LAT_CONV;                                           // 0 .. 251
    if (w < LAT_FC0) {
        const int r0 = w < LAT_FC0_9 ? 9 * w : 9 * LAT_FC0_9 + 8 * (w - LAT_FC0_9), nr = w < LAT_FC0_9 ? 9 : 8;
        // ---- this workgroup's rows of fc.0 -> registers (they do not depend on the window: requested before the features exist)
        if (a.fc_delay_ticks) { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < a.fc_delay_ticks) __builtin_amdgcn_s_sleep(8); }   // (A/B: let the conv role's first loads go ahead of this 38.8 MB stream)
        float4 w1[LAT_R][LAT_S];
        bool valid[LAT_S];
#pragma unroll
        for (int s = 0; s < LAT_S; ++s) valid[s] = tid + 512 * s < FEAT / 4;
#pragma unroll
        for (int r = 0; r < LAT_R; ++r) {
            const float4* row = reinterpret_cast<const float4*>(a.w1 + (size_t)(r0 + (r < nr ? r : nr - 1)) * FEAT);
#pragma unroll
            for (int s = 0; s < LAT_S; ++s) {
                const float4 v = row[valid[s] ? tid + 512 * s : tid];
                w1[r][s] = valid[s] ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float bias1 = tid < nr ? a.b1[r0 + tid] : 0.f;
        for (unsigned long long seq = a.seq;; ++seq) {
            const int ok = lat_wait(&sy->feat, LAT_CONV * seq, SERVICE ? &sy->quit : nullptr, SERVICE ? ~0ull : spin, flag, tid);
            if (ok <= 0) { if (ok < 0) fail_out(); return; }
            if (w == 0) LAT_TRACE(4);
            float acc[LAT_R];
            float4 x[LAT_S];
#pragma unroll
            for (int s = 0; s < LAT_S; ++s) x[s] = valid[s] ? lat_ld4(a.feat + 4 * (tid + 512 * s)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < LAT_R; ++r) {
                float v = 0.f;
#pragma unroll
                for (int s = 0; s < LAT_S; ++s) v = lat_dot4(w1[r][s], x[s], v);
                acc[r] = v;
            }
            lat_block_sum<LAT_R>(acc, red, tid);
            if (tid < nr) {
                const float v = red[tid] + bias1;
                lat_st_ll(a.h1 + r0 + tid, v < 0.f ? 0.f : v, (unsigned)seq);      // (keeps NaN, as the batch kernels and torch do)
            }
            if (w == 0) LAT_TRACE(5);
            if constexpr (!SERVICE) return;
            __syncthreads();                                                       // (red is free again)
        }
    }

    // ============================================================================ fc.3 role: workgroups 240 .. 255, 32 neurons each; the first also fc.6
    // (h1 is needed whole by every fc.3 workgroup: on 128 workgroups of 4 neurons the 128 x 2048 words polled per round congested the
    //  memory side -- the last of them had h1 4.9 us after the first; 16 readers do not)
    const int u = w - LAT_FC0;                                                     // 0 .. 15
    float4 w2[LAT_N3];
#pragma unroll
    for (int i = 0; i < LAT_N3; ++i) w2[i] = reinterpret_cast<const float4*>(a.w2 + (size_t)(LAT_N3 * u + i) * FC1)[tid];
    // fc.6 is folded into this role: thread (class j = tid / 32, neuron i = tid % 32) holds W3[j][32 u + i]; the workgroup sends its 16
    // PARTIAL logits (over its 32 neurons) instead of its 32 h2 values, and the first fc.3 workgroup adds the 16 x 16 partials in order:
    // the last hand-over is then followed by 16 adds, not by a 512-term layer
    const float w3 = a.w3[(size_t)(tid >> 5) * FC2 + LAT_N3 * u + (tid & 31)];
    const float bias2 = tid < LAT_N3 ? a.b2[LAT_N3 * u + tid] : 0.f;
    const float bias3 = (u == 0 && tid < NCLS) ? a.b3[tid] : 0.f;
    for (unsigned long long seq = a.seq;; ++seq) {
        const unsigned tag = (unsigned)seq;
        // ---- fc.3: every thread waits for the four h1 values IT multiplies (the service: for as long as it takes -- the request may not have come yet)
        {
            float h[4];
            const bool got = lat_ld_ll<4>(a.h1 + 4 * tid, tag, h, SERVICE ? &sy->quit : nullptr, SERVICE ? ~0ull : spin);
            if (__syncthreads_or(!got)) { fail_out(); return; }
            if (u == 0) LAT_TRACE(6);
            const float4 hv = make_float4(h[0], h[1], h[2], h[3]);
            float acc[LAT_N3];
#pragma unroll
            for (int i = 0; i < LAT_N3; ++i) acc[i] = lat_dot4(w2[i], hv, 0.f);
            lat_block_sum<LAT_N3>(acc, red, tid);
            float* h2s = red + 2048;                                               // this workgroup's 32 values of ReLU(fc.3)
            if (tid < LAT_N3) {
                const float v = red[tid] + bias2;
                h2s[tid] = v < 0.f ? 0.f : v;
            }
            __syncthreads();
            const float part = lat_row_sum(w3 * h2s[tid & 31]);                    // class tid / 32: the two rows of its 32 lanes
            if ((tid & 15) == 0) red[tid >> 4] = part;
            __syncthreads();
            if (tid < NCLS) lat_st_ll(a.h2 + NCLS * u + tid, red[2 * tid] + red[2 * tid + 1], tag);     // (the h2 words now carry partial logits [u][class])
        }
        if (u == 0) LAT_TRACE(7);
        if (u != 0) { if constexpr (SERVICE) { __syncthreads(); continue; } else return; }
        // ---- the 16 x 16 partial logits -> logits, torch.max(output, 1), decimal2binary
        {
            float pv[1] = {0.f};
            const bool got = tid < LAT_FC3 * NCLS ? lat_ld_ll<1>(a.h2 + tid, tag, pv, SERVICE ? &sy->quit : nullptr, spin) : true;
            if (__syncthreads_or(!got)) { fail_out(); return; }
            LAT_TRACE(8);
            if (tid < LAT_FC3 * NCLS) red[tid] = pv[0];
            __syncthreads();
            float* lg = red + 512;
            if (tid < NCLS) {
                float v = red[tid];
#pragma unroll
                for (int k = 1; k < LAT_FC3; ++k) v += red[NCLS * k + tid];
                lg[tid] = v + bias3;
            }
            __syncthreads();
            const int best = fc6_argmax16(lg);                                     // (every thread: 16 compares)
            const uchar4 cb = make_uchar4((best >> 3) & 1, (best >> 2) & 1, (best >> 1) & 1, best & 1);
            if constexpr (SERVICE) {
                // the two lines of the mailbox, one 16-lane store each (wave 0: lanes 0..15 line A, lanes 16..31 line B)
                const unsigned num = a.done_base + (unsigned)(seq - a.seq) + 1u;
                if (tid < 32) {
                    const int k = tid & 15;
                    unsigned word;
                    if (tid < 16) word = k < 15 ? __float_as_uint(lg[k]) : num;
                    else          word = k == 0 ? __float_as_uint(lg[15]) : k == 1 ? (unsigned)best : k == 2 ? __builtin_bit_cast(unsigned, cb) : k == 15 ? num : 0u;
                    unsigned* dst = reinterpret_cast<unsigned*>(tid < 16 ? static_cast<void*>(&a.mbox->a) : static_cast<void*>(&a.mbox->b)) + k;
                    __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                LAT_TRACE(10);
            } else {
                if (tid < NCLS) {
                    if (a.logits) a.logits[tid] = lg[tid];
                    if (a.packed) reinterpret_cast<float*>(a.packed)[tid] = lg[tid];
                }
                if (tid == 0) {
                    if (a.pred) *a.pred = best;
                    if (a.contacts) *reinterpret_cast<uchar4*>(a.contacts) = cb;
                    if (a.packed) *reinterpret_cast<uchar4*>(a.packed + 4 * NCLS) = cb;
                    LAT_TRACE(9);
                }
            }
        }
        if constexpr (!SERVICE) return;
        __syncthreads();
    }
}

hipError_t init_latency()
{
    hipError_t e;
    for (const void* k : {reinterpret_cast<const void*>(&latency_kernel<0>), reinterpret_cast<const void*>(&latency_kernel<1>),
                          reinterpret_cast<const void*>(&latency_kernel<2>)})
        if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, LAT_LDS)) != hipSuccess) return e;
    return hipSuccess;
}

int latency_grid() { return LAT_GRID; }

hipError_t launch_latency(int mode, const LatArgs& a, hipStream_t st)
{
    plan_note(mode == 2 ? "latency_service" : mode == 1 ? "latency_one_zs" : "latency_one");
    if (mode == 0)      hipLaunchKernelGGL(latency_kernel<0>, dim3(LAT_GRID), dim3(512), LAT_LDS, st, a);
    else if (mode == 1) hipLaunchKernelGGL(latency_kernel<1>, dim3(LAT_GRID), dim3(512), LAT_LDS, st, a);
    else                hipLaunchKernelGGL(latency_kernel<2>, dim3(LAT_GRID), dim3(512), LAT_LDS, st, a);
    return hipGetLastError();
}

}  // namespace dce
