/*
 * dce.h -- C ABI of libdce.so: the MI355X (gfx950) sliding-window contact-state
 * inference path.
 *
 * The reference (UMich-CURLY/deep-contact-estimator) has no FFI layer: its hot path
 * is reached through three Python-level contracts.  Each entry point below names the
 * reference call it stands in for, so that a maintainer can bind it with ctypes (see
 * INTEGRATION.md) behind the unchanged Python surface:
 *
 *   contact_cnn() + load_state_dict(...)      src/contact_cnn.py:7-58,
 *                                             src/inference_one_seq.py:153-156
 *   model(input_data) -> (B,16) logits        src/contact_cnn.py:60-66,
 *                                             src/inference_one_seq.py:25, src/test.py:87
 *   contact_dataset.__getitem__ (z-score)     utils/data_handler.py:32-61
 *   inference() loop: argmax + bit-unpack     src/inference_one_seq.py:19-30,59-62
 *
 * Conventions
 *   - plain C types only; no C++/torch types cross this boundary.
 *   - every function returns 0 (DCE_OK) or a negative dce_status; nothing throws or
 *     aborts.  dce_last_error(ctx) gives a human-readable message for the last failure.
 *   - a ctx is bound to ONE device and ONE stream and is not re-entrant.  Multi-GPU is
 *     one ctx per device (one process per GPU in this repo).
 *   - "on_device" flags say whether the caller's data pointers are device (HIP) or host
 *     pointers.  Host data is staged chunk by chunk (max_batch windows) through a ctx-owned ring
 *     of three device slots on a second stream, under the neighbouring chunks' kernels; the
 *     copies go straight from / to the caller's memory (pageable or pinned), the device
 *     footprint is bounded by max_batch, not by n.  On ANY error return no copy into the
 *     caller's buffers is still in flight.
 *   - the calling thread's current HIP device is unchanged on return from every entry point.
 *   - any output pointer may be NULL to skip that output.
 *   - all launches are asynchronous on the ctx stream when on_device != 0; with host
 *     pointers the call returns after the results have landed in host memory.
 */
#ifndef DCE_H
#define DCE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCE_WINDOW      150   /* window_size in the YAML configs; hard-wired in the model (4736 = 128*37) */
#define DCE_CHANNELS    54    /* utils/mat2numpy.py:73  q12 qd12 acc3 omega3 p12 v12 */
#define DCE_CLASSES     16    /* src/contact_cnn.py:56-57 */
#define DCE_LEGS        4     /* src/inference_one_seq.py:59-62 */
#define DCE_FEAT        4736  /* src/contact_cnn.py:48  128 channels * 37 */
#define DCE_FC1         2048
#define DCE_FC2         512

typedef enum dce_status {
    DCE_OK              =  0,
    DCE_ERR_ARG         = -1,  /* NULL / out-of-range / wrong shape */
    DCE_ERR_HIP         = -2,  /* a HIP runtime call failed; message has hipGetErrorString */
    DCE_ERR_STATE       = -3,  /* called in the wrong order (e.g. forward before finalize) */
    DCE_ERR_KEY         = -4,  /* unknown or duplicate state_dict key */
    DCE_ERR_NOMEM       = -5,
    DCE_ERR_COMM        = -6   /* RCCL is unavailable or one of its calls failed; message has ncclGetErrorString */
} dce_status;

typedef enum dce_precision {
    DCE_FP32            = 0,   /* fp32 MFMA everywhere (exact fp32 fmaf chains) -- the headline path */
    DCE_BF16_FC         = 1,   /* BASELINE configs[4]: bf16 operands (fp32 accumulate) on fc.0 / fc.3, whose inputs -- the features and h1 -- are
                                  rounded to bf16 (8 significant bits); the conv stack in front of that rounding keeps results of fp32 grade.
                                  In launches of more than 256 windows it runs on the fp16 matrix pipe with every operand as TWO fp16 terms of
                                  the value times a per-window power of two (22 significant bits, three MFMAs per product; csrc/conv_h2.hip,
                                  see DCE_FP32_F16X2); up to 256 windows (one-window calls, online pushes) and in the taps on two bf16 terms
                                  (~17 bits; csrc/conv_x3.hip, NT = 2), whose results do not depend on the size of the launch -- the mode's
                                  error against an fp64 evaluation is that of its bf16 FC operands either way (profiles/r4h_bf16_terms_audit.json).
                                  Option: bf16_conv_h2=0 (two bf16 terms at every size: the default of rounds 4-5; round 3's three-term stack
                                  is an experiments-build option).  CONTRACT of the mode: logits within 6e-3 of the largest logit
                                  of an fp32 / fp64 evaluation (the price of 8-bit operands on fc.0 / fc.3), and as close to the CPU
                                  restatement of the mode (oracle_forward_windows_bf16fc: the two differ where a value sits on a bf16
                                  rounding boundary -- <= 3e-3 on every fixture and fuzz set, 4.2e-3 worst over 1e6 logits with the two-term
                                  bf16 stack, <= 2e-3 with the fp32-grade ones); argmax equal wherever the top-2 margin exceeds 1e-2 of the
                                  largest logit; bench.py reports the step with both conv stacks */
    DCE_FP32_SPLIT      = 2,   /* RETIRED from the product library in round 6: libdce.so runs DCE_FP32_F16X2 for this value (same contract against
                                  the reference, 1.13 - 2.0 x the speed at every launch size, no range guard needed; profiles/r6h_retire_split_sweep.txt)
                                  and dce_last_plan begins with "fp32_split_is_fp32_f16x2".  The precision itself -- fp32 results with the conv stack
                                  and fc.0 on the bf16 matrix pipe, every fp32 operand as three bf16 terms (a = a1 + a2 + a3 exactly), six MFMAs per
                                  product, behind a RANGE GUARD (static per-layer bounds at dce_finalize_weights, a per-window check in the conv
                                  kernel and a gated DCE_FP32 kernel sequence behind every guarded launch; dce_split_guard_info) -- lives on in the
                                  experiments build (libdce_experiments.so: csrc/conv_x3.hip, csrc/fc_gemm_x3.hip), with its tests */
    DCE_FP32_F16X2      = 3    /* fp32-TOLERANCE results with the conv stack (from 128 windows per launch), fc.0 and fc.3 (from 1281; below
                                  that the DCE_FP32 kernels) on the fp16 matrix pipe: every operand is scaled by a power of two and
                                  enters as TWO fp16 terms (11 + 11 significand bits), three MFMAs per product, fp32 accumulate
                                  (csrc/conv_h2.hip, csrc/fc_gemm_h2.hip; 2.5 x DCE_FP32's throughput at 4096 windows).  Not fp32 operands -- 22 of their 24 bits -- but the same contract
                                  against the reference as DCE_FP32: the operand rounding costs 0.014 of the logit tolerance, a tenth of what
                                  fp32 accumulation costs every precision here.  The scales -- per layer for the weights, per WINDOW and layer
                                  for the activations, chosen by the kernel from the layer's largest output -- keep every operand inside
                                  fp16's range whatever the input: no range guard, no fallback, and a window's result depends on that
                                  window alone.  A checkpoint with a non-finite weight runs the DCE_FP32 kernels.  Opt-in */
} dce_precision;
/* Batch-size regimes.  DCE_FP32 gives a window the same bits whatever the size of the call it arrives in (one fixed summation tree in
 * every kernel family; latency=1 contexts: whole calls of up to 32 windows take one-kernel forms with another summation order, see below).  The two
 * other precisions pick kernels by the number of windows in a launch (a call of more than max_batch windows is several launches: the last one may
 * fall into another regime); the table is kPlanRows in csrc/dce_api.hip.  DCE_BF16_FC: one conv kernel up to 256 windows, another above (both far inside the mode's band), FC
 * kernels by size -- up to 256 windows per launch one weight-streaming kernel whose results do not depend on the number of windows (an
 * online push gives the bits of a sequence call in launches of <= 256), above that tile / phased GEMMs with other fp32 summation orders:
 * an h1 value at a bf16 rounding boundary may round the other way, <= 2e-2 of the largest logit.  DCE_FP32_F16X2: below 128 windows the
 * DCE_FP32 kernels, from 128 the two-term fp16 conv stack, from 1281 fc.0 / fc.3 on two-term operands too (launches past 12288 windows: fc.3 on
 * another tile) -- within a regime a window's bits depend on that window alone; between regimes they differ by fp32 rounding (<= 2e-5 of the
 * largest logit).  Tested:
 * tests/test_round4_gpu.py::test_batch_size_regimes_stay_within_the_mode_tolerance.  A caller that needs call-size invariance uses DCE_FP32. */

typedef struct dce_ctx dce_ctx;   /* opaque; owns device weights, scratch and (by default) a stream */

/* Library/ABI version, for binding sanity checks. */
int  dce_abi_version(void);

/* How this libdce.so was built: bit set of DCE_BUILD_*.  The default library is 0: product kernels only.  EXPERIMENTS: it also
 * carries the variants that were measured slower and kept for A/B (csrc/dce_kernels.h; python -m deep_contact_estimator_amd.build
 * --experiments), whose switches (DCE_CONV4, DCE_GEMM=lockstep, DCE_X3_PAIR) the default library ignores.  TRACE: phase time
 * stamps inside the kernels.  ASAN: the host side is instrumented with AddressSanitizer + UBSan (build --asan). */
#define DCE_BUILD_EXPERIMENTS 1
#define DCE_BUILD_TRACE       2
#define DCE_BUILD_ASAN        4
int  dce_build_flags(void);

/* Number of visible HIP devices, or a negative dce_status. */
int  dce_device_count(void);

/* contact_cnn().to(device): create a context on device_id.  max_batch bounds the
 * windows per forward call (scratch for conv features / FC activations is sized from
 * it: 4736+2048+512 floats per window); calls with more windows are chunked inside. */
int  dce_create(dce_ctx** out, int device_id, int64_t max_batch);
/* ... with the A/B switches of DESIGN.md's appendix as an option string "key=value,key=value" (NULL: the environment variable
 * DCE_TUNE; both over the built-in defaults).  ONE table names the keys (kTuneKeys, csrc/dce_api.hip); an unknown key or a malformed
 * value is DCE_ERR_ARG; keys of variants that only the experiments build contains are accepted and ignored by the default library. */
int  dce_create_ex(dce_ctx** out, int device_id, int64_t max_batch, const char* options);
void dce_destroy(dce_ctx* ctx);

/* Run on a caller-provided hipStream_t (e.g. torch's current stream) instead of the
 * ctx-owned one.  use_own != 0 restores the ctx-owned (non-blocking) stream and ignores
 * hip_stream; otherwise hip_stream is used as given -- NULL means HIP's null stream, which
 * is what torch.cuda.current_stream().cuda_stream is for torch's default stream. */
int  dce_set_stream(dce_ctx* ctx, void* hip_stream, int use_own);

/* load_state_dict: one call per state_dict key of contact_cnn (src/contact_cnn.py:8-58):
 *   block1.0.weight (64,54,3)   block1.0.bias (64)    block1.2.weight (64,64,3)    block1.2.bias (64)
 *   block2.0.weight (128,64,3)  block2.0.bias (128)   block2.2.weight (128,128,3)  block2.2.bias (128)
 *   fc.0.weight (2048,4736)     fc.0.bias (2048)      fc.3.weight (512,2048)       fc.3.bias (512)
 *   fc.6.weight (16,512)        fc.6.bias (16)
 * host is a contiguous fp32 HOST array in PyTorch layout; it is copied. */
int  dce_load_weight(dce_ctx* ctx, const char* key, const float* host,
                     const int64_t* shape, int ndim);

/* model.eval(): check all 14 keys are present, repack into kernel layouts, upload. */
int  dce_finalize_weights(dce_ctx* ctx, int precision /* dce_precision */);

/* model(input_data) + torch.max(output,1) + decimal2binary:
 * windows (n,150,54) fp32, ALREADY z-scored (what DataLoader yields) ->
 * logits (n,16) f32, pred (n) i32 argmax (ties -> lowest index), contacts (n,4) u8
 * (MSB first: class 9 -> 1,0,0,1). */
int  dce_forward_windows(dce_ctx* ctx, const float* windows, int64_t n, int on_device,
                         float* logits, int32_t* pred, uint8_t* contacts);

/* contact_dataset + DataLoader + inference(): raw sequence (T,54) fp32 -> results for
 * the T-149 sliding windows; output row j belongs to data row j+149.  Each window is
 * z-scored per channel over time (mean, unbiased std, no epsilon) inside the kernel.
 * window must be 150. */
int  dce_infer_sequence(dce_ctx* ctx, const float* seq, int64_t T, int window, int on_device,
                        float* logits, int32_t* pred, uint8_t* contacts);

/* The same two calls with the results as ONE array of DCE_PACKED_ROW-byte rows -- the 16 fp32 logits of a window
 * followed by its 4 contact bits (pred is the four bits read MSB first, src/inference_one_seq.py:59-62 is a
 * bijection on 0..15) -- written in that form by the last kernel of the path.  This is the row format of the
 * multi-GPU gather below: what a rank computes is what travels, no repacking pass. */
#define DCE_PACKED_ROW  68
/* A DEVICE `packed` pointer must be 4-byte aligned (the kernels write and read the rows as 32-bit words; rows are 68 bytes, so
 * an aligned buffer keeps every row aligned): DCE_ERR_ARG otherwise.  Host pointers need no alignment. */
int  dce_forward_windows_packed(dce_ctx* ctx, const float* windows, int64_t n, int on_device, uint8_t* packed);
int  dce_infer_sequence_packed(dce_ctx* ctx, const float* seq, int64_t T, int window, int on_device, uint8_t* packed);
/* (n,68) packed rows -> logits (n,16) f32, pred (n) i32, contacts (n,4) u8 (any may be NULL); device pointers run
 * as one kernel on the ctx stream, host pointers are unpacked on the host (ctx may then be NULL). */
int  dce_unpack_results(dce_ctx* ctx, const uint8_t* packed, int64_t n, int on_device,
                        float* logits, int32_t* pred, uint8_t* contacts);

/* contact_dataset.__getitem__ for windows [first, first+n): materialise the z-scored
 * windows (n,150,54) from a raw (T,54) sequence (utils/data_handler.py:55-56). */
int  dce_zscore_windows(dce_ctx* ctx, const float* seq, int64_t T, int64_t first, int64_t n,
                        int on_device, float* windows_out);

/* Per-layer taps for parity tests: run ONE batch of pre-normalised windows and copy out
 * intermediate activations (device or host pointers per on_device; any may be NULL):
 *   feat (n,4736) = block2 output flattened channel-major (src/contact_cnn.py:64)
 *   h1   (n,2048) = ReLU(fc.0)      h2 (n,512) = ReLU(fc.3)
 * fp32 arrays; in the DCE_BF16_FC mode feat and h1 ARE bf16 (the operands of the bf16 GEMMs) and the taps hand out
 * their uint16 bit patterns -- (n,4736) / (n,2048) uint16 -- while h2 and logits stay fp32. */
int  dce_forward_taps(dce_ctx* ctx, const float* windows, int64_t n, int on_device,
                      void* feat, void* h1, float* h2, float* logits);

/* Parity-test hook for the layers INSIDE the fused conv stack (reference src/contact_cnn.py:10-26,28-44; the
 * forward hooks of tests/golden/make_golden.py): run ONE named conv kernel family on n <= 64 pre-normalised HOST
 * windows with per-layer taps switched on and return the post-ReLU activations in PyTorch layout (HOST arrays):
 *   conv1 (n,64,150)  conv2 (n,64,150, before the pool)  pool1 (n,64,75)  conv3 (n,128,75)
 *   conv4 (n,128,75, before the pool)  feat (n,4736) = pool2 flattened.
 * kernel: 0 two-window Winograd workgroup (the chip-filling kernel), 1 one window on eight waves, 2 half-window
 * segments, 3 quarter-window segments, 4 direct form (DCE_CONV=direct), 5 one window on four waves, 6 the two-window
 * workgroup with four row tiles per wave (DCE_CONV4=1, an A/B variant of 0), 7 the three-term bf16 conv stack of the
 * DCE_FP32_SPLIT precision (conv_x3.hip; contexts finalised with that precision only).  The segment kernels (2, 3) never
 * compute conv4's t = 74 (MaxPool drops it): that column reads NaN.
 * The tapped kernels are the product kernels instantiated with the extra stores; a DCE_BF16_FC context is refused (DCE_ERR_STATE). */
int  dce_conv_layer_taps(dce_ctx* ctx, const float* windows, int64_t n, int kernel,
                         float* conv1, float* conv2, float* pool1, float* conv3, float* conv4, float* feat);

/* "Next" row after the path (reference src/test.py:19-70,102-104; src/inference_one_seq.py:48-54):
 * accumulate the 16x16 class confusion counts  counts[gt*16 + pred] += 1  over n windows.
 * pred: (n) i32 as written by dce_forward_windows / dce_infer_sequence; labels: (n) or (n,1) i64
 * decimal contact labels (utils/mat2numpy.py:199-200); counts: 256 x i64, ACCUMULATED (zero it
 * first).  Every metric the reference prints is a function of this matrix (see metrics.py);
 * across GPUs the matrices simply add (one all-reduce of 2 KB).  Classes outside [0,16) are skipped. */
int  dce_confusion_counts(dce_ctx* ctx, const int32_t* pred, const int64_t* labels, int64_t n,
                          int on_device, int64_t* counts);

/* Online mode (the reference README.md:67-83 describes a real-time runner that evaluates one
 * window per new sample at robot rate; its code is not in the reference tree): the ctx keeps the
 * last 150 samples in a device-resident ring; every push appends one (54,) fp32 HOST sample and,
 * once 150 samples are present, evaluates the newest window exactly as dce_infer_sequence would
 * (same kernels, z-score fused), so pushing a sequence row by row reproduces dce_infer_sequence's
 * rows bit for bit.  Returns 1 when outputs were written (HOST pointers, any may be NULL),
 * 0 while the ring is still filling, < 0 on error.  dce_online_reset empties the ring.
 * Latency path: no H2D/D2H copies and no stream synchronisation -- the sample is read by the first
 * kernel from pinned host memory, the estimate is written by the last kernel to pinned host memory
 * followed by a sequence number the call polls (~60 us per push on MI355X). */
/* LATENCY MODE (option "latency=1" of dce_create_ex; DCE_FP32 contexts; csrc/latency.hip) -- the reference ships batch_size 1
 * (config/inference_one_seq_params.yaml:10) and 30 (config/test_params.yaml:9).  A dce_forward_windows / dce_infer_sequence call (and their _packed
 * forms) of ONE window is then ONE kernel of 256 co-resident workgroups (stream-ordered like any launch; 26 us instead of 43), a call of 2 .. 32
 * windows likewise (csrc/latency_mb.hip: 32 us up to 16 windows, 39.5 us at 30 instead of 43 / 51; option latency_mb=0 keeps those on the batch path's
 * four launches).  Only WHOLE calls take these kernels: a call cut into chunks (more windows than max_batch) runs the batch kernels for every chunk, so
 * that a window's bits never depend on where a chunk boundary fell.  dce_online_push is served by the one-window
 * kernel in RESIDENT form: started by the first push on a stream of its own, it takes every sample from a mailbox in pinned host memory
 * and answers through it (no launch, copy or stream operation per push: ~30 us per push from C instead of ~57).  It leaves when any other
 * device-touching entry point of the context is called (forward / sequence / taps / z-score / unpack / confusion counts / the exchange calls), on dce_online_reset / dce_destroy, or by itself after latency_idle_ms (250) without a sample; the
 * next push restarts it with the sample history intact.  Results: inside the fp32 tolerance of the reference, deterministic, NOT the bits of
 * the batch path (another summation order).  The mode needs the whole device -- one workgroup per CU, 256 CUs, dce_create_ex refuses a
 * smaller one; other work on the device waits while a launch or the resident kernel runs; one latency-mode stream per device.  Every
 * hand-over inside the kernel has a 50 ms deadline: a launch that ran into it (CUs held by someone else) is reported as DCE_ERR_HIP by
 * the next call on the context (or by dce_sync). */
int  dce_online_reset(dce_ctx* ctx);
int  dce_online_push(dce_ctx* ctx, const float* sample, float* logits, int32_t* pred, uint8_t* contacts);

/* Multi-GPU (not in the reference, which is single-device: src/inference_one_seq.py:139; BASELINE.json configs[3]).
 * Windows are independent (utils/data_handler.py:55-57), so the GPUs of a node take contiguous window ranges (a
 * 149-row halo of the sequence per boundary, weights replicated; one process and one ctx per GPU) and the ONLY
 * exchange is collecting the results: one RCCL gather over xGMI of every rank's packed rows to the root.
 *
 *   dce_comm_get_unique_id  rank 0: a fresh ncclUniqueId; the host carries its 128 bytes to the other ranks (file,
 *                           environment, any store) -- no MPI, no torch needed.
 *   dce_comm_init           every rank (collective, blocks until all `world` ranks have called): ncclCommInitRank on
 *                           the ctx's device.  RCCL is bound at run time: $DCE_RCCL_LIB, else the librccl already in
 *                           the process, else the system one.
 *   dce_comm_info           what RCCL reports for the communicator (ncclCommUserRank / ncclCommCount / ncclGetVersion)
 *                           and which library was bound.
 *   dce_gather_results      rank r's packed_local (n_local,68) DEVICE rows -> the root's packed_all, rank blocks in rank
 *                           order (rows_per_rank[g] rows of rank g; NULL = every rank holds n_local rows): ONE
 *                           ncclGather, or -- when the shards differ in length -- one ncclGroup of ncclSend/ncclRecv
 *                           with the true sizes.  Runs on a ctx-owned communication stream behind everything queued on
 *                           the ctx stream so far.  async == 0: work queued on the ctx stream afterwards waits for it
 *                           (stream-ordered, no host block).  async != 0: it overlaps the kernels queued after it; work
 *                           queued after THIS call waits only for the gather issued by the PREVIOUS async call, so a
 *                           caller that alternates two send buffers (and two receive buffers on the root) never
 *                           overwrites rows in flight.  packed_all may be NULL off the root.
 *   dce_allreduce_counts    sum the 256 confusion counts of dce_confusion_counts over the ranks (ncclAllReduce, int64).
 *   dce_comm_sync           block until every exchange issued on this ctx has completed; reports asynchronous RCCL errors.
 *   dce_comm_destroy        ncclCommDestroy (dce_destroy does it too). */
#define DCE_COMM_ID_BYTES 128
int  dce_comm_get_unique_id(uint8_t id[DCE_COMM_ID_BYTES]);
int  dce_comm_init(dce_ctx* ctx, int rank, int world, const uint8_t id[DCE_COMM_ID_BYTES]);
int  dce_comm_info(dce_ctx* ctx, int* rank, int* world, int* rccl_version, char* library, int library_len);
int  dce_gather_results(dce_ctx* ctx, const uint8_t* packed_local, int64_t n_local, uint8_t* packed_all,
                        const int64_t* rows_per_rank, int root, int async);
int  dce_allreduce_counts(dce_ctx* ctx, int64_t* counts, int on_device);
int  dce_comm_sync(dce_ctx* ctx);
int  dce_comm_destroy(dce_ctx* ctx);

/* Kernel timing with HIP events on the ctx's stream, for bench.py's roofline block.
 * on = k > 0 records an event pair around each of the four kernels of every k-th kernel sequence
 * (k = 1: every one; an event costs ~4 us of stream time, so a sparse sample keeps the timed
 * region honest), on = 0 stops.  dce_profile_read synchronises and returns the accumulated
 * milliseconds and launch counts since the last reset:
 * slot 0 = conv stack, 1 = fc.0 GEMM, 2 = fc.3 GEMM (with fc.6's chunk sums in its epilogue at chip-filling
 * batches), 3 = fc.6 + argmax + contact bits (combine kernel behind the fused epilogue, tail kernel otherwise). */
#define DCE_PROFILE_SLOTS 4
int  dce_profile_enable(dce_ctx* ctx, int on);
int  dce_profile_read(dce_ctx* ctx, double ms[DCE_PROFILE_SLOTS], int64_t launches[DCE_PROFILE_SLOTS], int reset);

/* DCE_FP32_SPLIT's range guard as it stands for this context (synchronises the ctx stream to read the device-side counters). */
typedef struct dce_split_guard {
    int      precision;             /* dce_precision of the context */
    int      enabled;               /* finalised with DCE_FP32_SPLIT and the guard not switched off (option split_guard=0) */
    int      refused;               /* the checkpoint leaves the guarded range: every call runs the DCE_FP32 kernels (reason says why) */
    float    x_hi, x_lo;            /* a pre-normalised window passes if max|x| <= x_hi and (max|x| >= x_lo or it is all zero) */
    float    z_max;                 /* 149 / sqrt(150): what a z-scored window can reach */
    double   gain[6], offs[6];      /* max|activation| of conv1..4, fc.0, fc.3 <= gain X + offs for inputs |x| <= X */
    uint32_t guarded_launches;      /* launches that carried the per-window check */
    uint32_t windows_out_of_range;  /* windows that failed it */
    uint32_t fallbacks_run;         /* gated DCE_FP32 sequences that ran because of them */
    char     reason[256];
} dce_split_guard;
int  dce_split_guard_info(dce_ctx* ctx, dce_split_guard* out);

/* Which kernels the most recent kernel sequence of this ctx launched, as space-separated family names in launch order
 * (e.g. "conv_wino2 fc_phased256x128 fc23_fused_phased128x64 fc6_combine"): lets a test assert that a batch size or an
 * A/B switch (dce_create_ex's option string) selected the kernel it means to check. */
int  dce_last_plan(dce_ctx* ctx, char* out, int out_len);

/* Block until everything queued on the ctx stream has finished. */
int  dce_sync(dce_ctx* ctx);

/* Message for the last failing call on this ctx (or on creation when ctx == NULL). */
const char* dce_last_error(dce_ctx* ctx);

/* Test hook: the three bf16 terms (a = t1 + t2 + t3 exactly for every normal fp32 a, round-to-nearest-even each) the
 * DCE_FP32_SPLIT kernels split their operands into -- the host routine that prepares fc.0's weights; planes = [3][n] uint16
 * (tests/test_x3_gpu.py::test_split_terms_are_exact).  No device involved. */
void dce_debug_split3(const float* x, size_t n, unsigned short* planes);

/* Test hook of DCE_FP32_F16X2 (csrc/conv_h2.hip): the scale exponent sw the weights x[0..n) get (max|x| * 2^sw in [2^14, 2^15); INT_MIN for a
 * tensor with a non-finite entry) and the two fp16 terms of every x * 2^sw -- terms = [2][n] uint16, t1 = fp16(x 2^sw) round-to-nearest-even,
 * t2 = fp16(x 2^sw - t1) -- the host routine that prepares the conv / fc.0 weights.  No device involved.  Returns sw. */
int  dce_debug_split_h2(const float* x, size_t n, unsigned short* terms);

/* Debug hook of the latency mode (option latency=1, contexts created with DCE_LAT_TRACE set): 16 stamps of the device's 100 MHz wall
 * clock taken by the last request -- [0] request seen, [1] window ready, [2] conv segment 0 done, [3] its arrival posted, [4] features
 * seen by fc workgroup 0, [5] its fc.0 rows done, [6] h1 complete, [7] its fc.3 rows done, [8] h2 complete, [9]/[10] results written.
 * After a call of 2 .. 32 windows (the micro-batch kernel, csrc/latency_mb.hip): [1] conv workgroup 0 starts, [2] its segment done, [3] its flag posted,
 * [4] fc.0 tile 0 has requested its weights, [5] they have landed, [6] it has seen every conv flag, [7] its columns of h1 posted, [8] fc.3 tile 0 has seen
 * every fc.0 flag, [9] workgroup 0 waits for the partial logits, [10] has them, [11] results written; [12] / [13] the earliest request and the latest
 * landing of fc.0's weights over the 128 tiles (the 38.8 MB stream that runs under the conv role). */
int  dce_debug_latency_trace(dce_ctx* ctx, unsigned long long stamps[16]);

/* Test hooks of the memory-safety tests (csrc/dev_alloc.hip; option guard_alloc=1 | 2 of dce_create_ex): a context created with the option keeps EVERY
 * device buffer it owns in a mapping of its own whose last (1) / first (2) byte abuts an unmapped page, so that a kernel access one element outside any
 * buffer is a GPU page fault at that access.  These two give a CALLER-owned device buffer (windows, sequences, results passed with on_device = 1) the same
 * placement (the placement of `ctx`; a plain hipMalloc for a context without the option).  tests/test_guard_alloc_gpu.py, tools/guard_stress.py. */
int  dce_debug_alloc(dce_ctx* ctx, size_t bytes, void** device_ptr);
int  dce_debug_free(dce_ctx* ctx, void* device_ptr);

#ifdef __cplusplus
}
#endif
#endif /* DCE_H */
