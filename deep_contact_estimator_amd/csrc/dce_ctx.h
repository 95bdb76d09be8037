// dce_ctx.h -- the context behind the opaque dce_ctx of include/dce.h, shared by the translation units that
// implement the C ABI (dce_api.hip: the path; dce_comm.hip: the RCCL exchange).  Internal to libdce.so.
#pragma once
#include "../../include/dce.h"
#include "dce_kernels.h"

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

namespace dce { extern thread_local std::string g_create_error; }   // message of a failing call that has no ctx

struct dce_ctx {
    int device = 0;
    int64_t max_batch = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t xstream_ev = nullptr;
    hipStream_t xfer_stream = nullptr;     // host-buffer callers: chunk copies overlap the kernels of the
    hipEvent_t ring_ev[3][3] = {};         // neighbouring chunks; per ring slot: staged-in, computed, staged-out
    std::string err;

    std::vector<float> host_w[14];         // staged state_dict (host, PyTorch layout)
    bool have[14] = {};
    bool finalized = false;
    int precision = DCE_FP32;
    bool split_alias = false;              // product library: the context was finalized as DCE_FP32_SPLIT and runs DCE_FP32_F16X2 (dce_last_plan says so)
    std::vector<const char*> plan;         // kernel families launched by the most recent kernel sequence (dce_last_plan)
    dce::Tuning tuning;                    // the A/B switches: dce_create_ex's option string, else DCE_TUNE, over the defaults
    dce::Tuning gate_tuning;               // ... the same without the small-batch kernel families and row cuts: what a gated DCE_FP32 fallback sequence runs

    // DCE_FP32_SPLIT's range guard (dce_finalize_weights computes it; include/dce.h dce_split_guard_info)
    struct SplitGuard {
        bool refused = false;              // the checkpoint itself leaves the guarded range: the mode runs the DCE_FP32 kernels
        float x_hi = 0.f, x_lo = 0.f;      // a pre-normalised window passes if max|x| <= x_hi and (max|x| >= x_lo or the window is all zero)
        double gain[6] = {}, offs[6] = {}; // max|activation of layer l| <= gain[l] X + offs[l] for inputs |x| <= X (conv1..4, fc.0, fc.3)
        std::string reason;
    } guard;
    unsigned* d_guard = nullptr;           // device: [0] generation of the last launch that saw a window out of range, [1] such windows so far, [2] gated fallback sequences that ran
    unsigned guard_gen = 0;                // generation of the current guarded launch
    unsigned guard_launches = 0;           // guarded launches so far
    bool gate_on = false;                  // the kernel sequence being enqueued is the gated DCE_FP32 fallback

    float* d_weights = nullptr;            // one allocation: conv packs, biases, fc weights
    dce::ConvPack pk{};
    const float *fc1w = nullptr, *fc1b = nullptr, *fc2w = nullptr, *fc2b = nullptr,
                *fc3w = nullptr, *fc3b = nullptr;
    const void *fc1w_bf16 = nullptr, *fc2w_bf16 = nullptr;   // DCE_BF16_FC only
    const void* fc1w_bf16p = nullptr;                        // ... fc.0 with its K axis in conv_x3p.hip's feature order (t' * 128 + c)

    float *feat = nullptr, *h1 = nullptr, *h2 = nullptr;   // scratch, max_batch rows each
    unsigned short* feat3 = nullptr;                        // DCE_FP32_SPLIT: the features as three bf16 planes [3][n][4736]
    dce::ConvPackX3 pkx3{};                                      // ... and the conv weights, packed per lane (conv_x3.hip)
    const unsigned short* fc1w_x3 = nullptr;                // ... and fc.0's weights, [3][2048][4736] (inside d_weights, or fc1w_x3_own: made on first use when the permuted copy is the one in use)
    unsigned short* fc1w_x3_own = nullptr;
    const unsigned short* fc2w_x3 = nullptr;                // ... fc.3's weights, three row-major planes [3][512][2048] (inside d_weights)
    unsigned short* h1p = nullptr;                          // ... h1 as three row-major bf16 planes [3][max_batch + 1][2048] (fc.0's epilogue writes them, fc.3 reads them)
    const unsigned short* fc1w_x3p = nullptr;               // ... and the same with the K axis in conv_x3p.hip's feature order
    dce::ConvPackH2 pkh2{};                                 // DCE_FP32_F16X2: the conv weights as two fp16 terms, packed per lane, with their scales (conv_h2.hip)
    const unsigned short* fc1w_h2 = nullptr;               // ... fc.0's weights [2048][148][2][32] fp16, K in the order t' * 128 + c, times 2^fc1_sw
    int fc1_sw = 0;
    int* feat_scale = nullptr;                             // ... the scale exponent of every window's features (max_batch)
    bool h2_refused = false;                               // ... a non-finite weight: the precision runs the DCE_FP32 kernels
    const unsigned short* fc2w_h2 = nullptr;               // ... fc.3's weights [512][64][2][32] fp16 times 2^fc2_sw
    int fc2_sw = 0, fc1_eW = 0, fc1_eB = 0;                // ... and the two static exponents of h1's row-scale bound (fc_gemm_h2.hip OUT2)
    unsigned short* h1h = nullptr;                         // ... h1 as two fp16 terms [max_batch + pad][64][2][32] (fc.0's epilogue writes it, fc.3 reads it)
    int* h1_scale = nullptr;                               // ... with its row scale exponents
    float* part = nullptr;                                 // fc.6 chunk sums [8][max_batch][16] (fused fc.3 epilogue)
    bool want_feat = false;                                // dce_forward_taps: DCE_FP32_SPLIT keeps the fp32 features (split by a kernel of its own)
    bool want_h1 = false;                                  // dce_forward_taps: h1 is wanted in fp32 (DCE_FP32_SPLIT then keeps fc.3 on the fp32 kernels)
    bool want_h2 = false;                                  // dce_forward_taps: the fused epilogue also writes h2

    // staging for host-pointer callers: a ring of RING_SLOTS chunk-sized slots (run_all), so that
    // the device footprint is bounded by max_batch, not by the length of the caller's input
    float* d_in = nullptr;   size_t d_in_bytes = 0;
    float* d_logits = nullptr; int32_t* d_pred = nullptr; uint8_t* d_contacts = nullptr; uint8_t* d_packed = nullptr;
    size_t d_out_rows = 0;

    // online mode: linear buffer of ONLINE_ROWS sample rows; the live window is its last 150 rows
    float* d_ring = nullptr;
    int64_t ring_rows = 0;
    float* h_online_pin = nullptr;         // pinned, device-visible: logits(16) | pred | contacts | ... | flag
    unsigned online_seq = 0;
    unsigned* done_flag = nullptr;         // set around an online push: the tail kernel publishes done_seq there
    unsigned done_seq = 0;
    // online mode as one hipGraph launch per sample (constant launch parameters; see dce_kernels.h)
    dce::OnlineState* d_online_state = nullptr;
    const long long* src_row_dev = nullptr;   // set around the graph's kernel sequence
    unsigned* seq_counter_dev = nullptr;
    hipGraph_t online_graph = nullptr;
    hipGraphExec_t online_exec = nullptr;
    int online_mode = -1;                  // -1 undecided, 0 direct launches, 1 graph
    bool online_state_dirty = true;        // device state must be zeroed before the next push

    // latency mode (latency.hip; option latency=1)
    float* lat_x = nullptr;                // fine-grained device memory: one row of features | h1 | h2 | the arrival counters
    dce::LatSync* lat_sync = nullptr;      // (inside lat_x)
    unsigned long long* lat_trace = nullptr;   // DCE_LAT_TRACE: pinned, phase stamps of the last request
    dce::LatMailbox* lat_mbox = nullptr;   // pinned host
    float* lat_hist = nullptr; int* lat_hist_state = nullptr;    // device: the service's sample history
    hipStream_t lat_stream = nullptr;      // the resident service kernel runs here
    bool lat_running = false;
    unsigned long long lat_seq = 0;        // estimates requested since the counters were last zeroed
    unsigned lat_req = 0;                  // mailbox request number
    int lat_count = 0;                     // samples in the history (host mirror)
    float *lat_mb_feat = nullptr, *lat_mb_h1 = nullptr, *lat_mb_plt = nullptr;   // micro-batch form (latency_mb.hip): features, h1, partial logits of up to 32 windows
    unsigned long long* lat_mb_flags = nullptr;                                  // ... its producer flags (fine-grained device memory)
    const float *lat_mb_w1 = nullptr, *lat_mb_w2 = nullptr;                      // ... fc.0 / fc.3 in its per-lane order (inside d_weights; latency contexts of DCE_FP32)
    unsigned long long lat_mb_seq = 0;     // ... requests so far
    int64_t call_total = 0;                // windows of the API call being served (latency plans serve whole calls only: a window's bits must not depend on where a chunk boundary fell)

    // multi-GPU (dce_comm.hip): one RCCL communicator per ctx, collectives on comm_stream behind the ctx stream
    void* comm = nullptr;                  // ncclComm_t
    int comm_rank = 0, comm_world = 0;
    hipStream_t comm_stream = nullptr;
    hipEvent_t comm_ready = nullptr;       // recorded on the ctx stream: the send buffer is complete
    hipEvent_t comm_done[2] = {};          // recorded on comm_stream after the gathers of even / odd index
    int64_t comm_issued = 0;               // asynchronous gathers issued so far

    // profiling
    int prof_period = 0;                   // 0 = off, k = time every k-th kernel sequence
    int64_t prof_tick = 0;
    bool prof = false;                     // events are recorded for the CURRENT sequence
    std::vector<hipEvent_t> ev_pool;
    struct Span { int slot; hipEvent_t a, b; };
    std::vector<Span> spans;
    double prof_ms[DCE_PROFILE_SLOTS] = {};
    int64_t prof_n[DCE_PROFILE_SLOTS] = {};
};

int dce_internal_quiesce(dce_ctx* c);     // dce_api.hip: the latency mode's resident kernel, if any, leaves (and a deadline error of it is reported)

inline int fail(dce_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else dce::g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail((c), DCE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

// Every entry point binds the calling thread to the ctx's device for the duration of the call and
// puts the caller's device back on return (a single-process multi-GPU host must not find its
// current device changed by a call into this library).
struct DeviceGuard {
    int prev = -1; bool switched = false; hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { err = hipSetDevice(dev); switched = err == hipSuccess; }
    }
    ~DeviceGuard() { if (switched && prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DEVICE_GUARD(c)                                                                         \
    DeviceGuard dev_guard_((c)->device);                                                        \
    if (dev_guard_.err != hipSuccess)                                                           \
        return fail((c), DCE_ERR_HIP, "hipSetDevice(%d) failed: %s", (c)->device, hipGetErrorString(dev_guard_.err))

