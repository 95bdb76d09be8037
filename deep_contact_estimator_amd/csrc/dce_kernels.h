// dce_kernels.h -- launch interface between the C ABI (dce_api.hip) and the gfx950 kernels.
// Internal to libdce.so; the public boundary is include/dce.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

// -DDCE_EXPERIMENTS=1 (python -m deep_contact_estimator_amd.build --experiments -> libdce_experiments.so, select with DCE_LIB): the
// kernel variants that were built, measured slower than what ships and kept for the record and the A/B -- the four-row-tile
// Winograd workgroup (DCE_CONV4=1), the lockstep schedule of the phased GEMM (DCE_GEMM=lockstep), the paired three-term conv stack
// (DCE_X3_PAIR=1) -- and the timing-probe macros.  The default libdce.so contains none of them: their switches are ignored there.
#ifndef DCE_EXPERIMENTS
#define DCE_EXPERIMENTS 0
#endif

namespace dce {

// ---- geometry of contact_cnn (reference src/contact_cnn.py:8-58) ------------------------
constexpr int WIN = 150, CH = 54, NCLS = 16, FEAT = 4736, FC1 = 2048, FC2 = 512;

// ---- A/B switches.  ONE table (kTuneKeys, dce_api.hip) names every switch; a context takes its values from the
// option string of dce_create_ex (or, without one, the environment variable DCE_TUNE) -- "key=value,key=value" -- and hands them to the
// launchers through a thread-local pointer that every C-ABI entry point sets for the duration of the call, so that two contexts of
// one process (tests, tools/race_screen.py) can run different kernel variants side by side.  Keys of variants that exist in the
// experiments build only are accepted and ignored by the product library; an unknown key fails dce_create.
struct Tuning {
    // -- FC GEMMs at chip-filling sizes (fc_gemm_phased.hip)
    bool gemm_tile = false;                                 // gemm_tile=1: round 1's tile kernels instead of the phased ones (bit-identical: the reference of test_phased_gemm_equals_tile_kernels)
    int phased_min_tiles = 192, phased_min_tiles1 = 128;    // tiles a launch needs for the 256x128 / 128x64 phased tile
    int phased_min = 1;                                     // 2: only the 256x128 tile
    bool phased_cost = true;                                // 0: tile minimum only, no rounds model
    int phased_sn = 3;                                      // log2 of the XCD super-tile's N extent
    int fc23 = 0;                                           // 1: fc.3 never with the fused fc.6 chunk sums, 2: always
    bool gemm_peel = true, conv_peel = true;                // 0: no row cuts of a batch past whole rounds of phased tiles / two-window conv workgroups
    bool gemm_small_deep = true;                            // 0: 64x64 tile GEMM without the deep staging ring
    // -- small and mid-size batches
    bool gemv = true;                                       // 0: no weight-streaming GEMV for <= 8 windows
    long long split_min = 9, split_max = 64;                // windows served by the four-range MFMA kernel (fc_gemm_split.hip)
    long long chain_min = 9, chain_max = 640, chain_max3 = 2048, chain_bn16_max = 64;   // ... by the MFMA chain kernel (fc.0 / fc.3; 32x16 blocks up to)
    int winoq_chsplit_max = 32;                             // ... up to this many windows the quarter-segment conv kernel runs TWO workgroups per segment (one half of conv4's output channels each; 8 n <= 256 workgroups); 0: one
    long long wino1_max = -1, winoh_max = -1, winoq_max = -1;   // windows up to which the one-window / half-window / quarter-window conv kernels run (-1: kernel default 256 / 128 / 64)
    bool wino1_w8 = true;                                   // 0: the four-wave predecessor of the one-window kernel
    bool online_graph = false, online_direct = false;       // online pushes as ONE captured hipGraph launch / as plain launches with per-push parameters
    bool latency = false;                                   // 1: the LATENCY MODE (latency.hip; DCE_FP32 only): one-window calls as one kernel of 256 co-resident workgroups, online pushes served by a resident kernel that polls a mailbox in pinned memory.  Inside the fp32 tolerance, NOT the batch path's bits
    int latency_fc_delay = 100;                             // 10 ns ticks by which the fc.0 role starts its 38.8 MB weight stream behind the conv role (one-window calls: 30.1 us with 0, 28.0 with 1 us, 28.6 / 28.9 with 3 / 6 us -- the conv role's first weight fragments would queue behind the stream; profiles/r5h_latency_ab.txt)
    bool latency_mb = true;                                 // ... calls of 2 .. 32 windows as ONE kernel too (latency_mb.hip: conv segments, fc.0 with register-resident weights on MFMA tiles, fc.3 + fc.6 on the conv workgroups); 0: the batch path's four launches
    int latency_mb_chalf = 16;                              // ... up to this many windows two conv workgroups share a quarter segment (each half of conv4's output channels)
    int latency_idle_ms = 250;                              // ... how long the resident kernel waits for the next sample before it leaves by itself
    // -- DCE_BF16_FC
    bool bf16_stream = true;                                // 0: fc.0 / fc.3 at <= 256 windows on the 64x64 tile GEMM instead of fc_stream_bf16.hip
    long long x3_bf16_min = 1;                              // windows from which the mode's conv stack runs on conv_x3.hip (below: the fp32 kernels, features rounded on the store)
    bool bf16_conv_h2 = true;                               // the mode's conv stack on conv_h2.hip -- two FP16 terms per operand with per-window scales: results of fp32 grade (22-bit operands) at three MFMAs per product, i.e. BASELINE configs[4] as it is written ("conv stays fp32") -- in launches of at least bf16_conv_h2_min windows; 0: conv_x3.hip on two bf16 terms (~17 bits) at every size
    long long bf16_conv_h2_min = 257;                       // (up to 256 windows -- one-window calls, online pushes -- the mode stays on conv_x3.hip's two-term form, whose results do not depend on the size of the launch)
    int x3_bf16_terms = 2;                                  // 3: the mode's conv stack on three-term operands (six MFMAs per product) -- BASELINE configs[4] as written ("conv stays fp32"-grade)
    // -- DCE_FP32_SPLIT
    bool x3_conv = true;                                    // 0: keep the fp32 Winograd conv stack (three-plane feature store) instead of conv_x3.hip (both bf16-pipe precisions)
    long long x3_conv_min = 128;                            // windows from which the conv stack runs on conv_x3.hip also below fc.0's threshold (fp32 features out)
    int x3_min_tiles = 192;                                 // 256x128 tiles a launch needs for fc_gemm_x3.hip
    bool x3_unfused = false;                                // 1: fp32 features + split3 kernel instead of the conv kernel's three-plane output
    bool x3_permk = true;                                   // 0: conv_x3.hip's features through LDS in the reference's flatten order instead of straight out in the order t' * 128 + c
    bool x3_fc3 = false;                                    // 1: fc.3 on three-term operands too (fc_gemm_x3.hip's 128 x 64 tile with the fused fc.6 epilogue; h1 leaves fc.0 as three planes).  Built and parity-green in round 5, and NOT faster: 64.3 us against 71 for the fp32 MFMA kernel, +9.5 us on fc.0's epilogue (profiles/r5j_split_fc3.txt) -- with 32 x 32 wave tiles a K-tile's LDS traffic is fc.0's for half its MFMAs
    // -- DCE_FP32_F16X2
    int h2_min_tiles = 96;                                  // 256x128 tiles a launch needs for fc.0 on fc_gemm_h2.hip (96: from 1281 windows -- below that the fp32 FC kernels behind conv_h2_f32 are as fast; one round of tiles takes fc.0 ~170 us whatever their number)
    bool h2_fc3 = true;                                     // 0: fc.3 + fc.6 chunk sums on the fp32 kernels instead of two-term fp16 operands (fc_gemm_h2k_kernel<H2KFc3>; h1 then leaves fc.0 in fp32)
    bool split_guard = true;                                // 0: no range guard (static bound at finalize, per-window exponent check + fp32 fallback): the round-4 behaviour, for the A/B and the audit
    // -- experiments build only (ignored by the product library)
    bool gemm_lockstep = false, gemm_pipe = false, gemm_ki = false;
    bool bf16_k32 = false;                                  // fc.0's 256x128 bf16 tile with 32-k K-tiles, two workgroups per CU, from 512 tiles per launch (fc_gemm_phased.hip PhTile<3>; round 5: measured 22 % slower)
    int conv4 = 0;
    bool x3_persist = false, x3_pair = false;
    long long x3_persist_min = 1024, x3_pair_min = 1024;
    bool conv_direct = false;                               // the direct-form conv stack of round 1 (conv_stack.hip)
    bool bf16_fc3_ksplit = false;                           // DCE_BF16_FC: the fused fc.3 + fc.6 on fc_gemm_h2k_kernel<H2KFc3, FUSE6, BF16> (K-tiles dealt out between the wave groups, 128-k phases) instead of fc_gemm_phased.hip's 128x64 tile (round 5: measured no faster, 17.5 against 17.8 us)
    bool h2_ksplit = false;                                 // DCE_FP32_F16X2: fc.0's K-tiles dealt out between the two wave groups (fc_gemm_h2k_kernel<H2KFc0>: 64 x 128 wave tiles, three LDS buffers; round 5: no faster, and the variant whose intermittent fault round 6 traced)
    bool one_per_cu = false, trace_wino1 = false;           // trace builds
    // -- memory-safety tests (dev_alloc.hip)
    int guard_mask = -1;                                    // ... which buffer groups (bit 0 weights, 1 feat, 2 h1, 3 h2, 4 part, 5 staged input, 6 staged results, 7 feat3, 8 two-/three-term h1 + scales, 9 small state, 10 online ring)
    int guard_alloc = 0;                                    // 1 / 2: every device buffer of the context in a mapping of its own whose last / first byte abuts an unmapped page; 3: a 4 GB address line inside every buffer (csrc/dev_alloc.hip)
};
// parses "key=value,..." over `t`; false + message on an unknown key or a malformed value
bool tuning_parse(const char* spec, Tuning& t, char* err, int err_len);
extern thread_local const Tuning* t_tuning;                  // the calling ctx's switches (nullptr: process defaults)
const Tuning& tune();
struct TuningScope {                                         // RAII: entry points bind their ctx's switches
    const Tuning* prev;
    explicit TuningScope(const Tuning* t) : prev(t_tuning) { t_tuning = t; }
    ~TuningScope() { t_tuning = prev; }
};

// ---- DCE_FP32_SPLIT's range guard.  A three-term split a = a1 + a2 + a3 is exact for |a| below bf16's largest finite number
// (3.3895e38 < fp32's 3.4028e38: above it a1 rounds to Inf) and while a3 stays a normal number.  dce_finalize_weights bounds every
// layer's activations for inputs |x| <= X (static: sums of |w|) and derives the largest safe X; z-scored windows are bounded by
// construction (|z| <= 149 / sqrt(150)), pre-normalised windows are checked by the conv kernel's load stage, per window.  A launch that
// saw a window outside [x_lo, x_hi] writes its generation to word[0]; the DCE_FP32 kernel sequence enqueued behind it is GATED on that
// word: every workgroup of it reads the word first and returns unless it holds this launch's generation.
struct GuardArgs { unsigned* word = nullptr; unsigned gen = 0; float x_hi = 0.f, x_lo = 0.f; };
#if DCE_EXPERIMENTS
struct Gate { const unsigned* word = nullptr; unsigned gen = 0; unsigned* taken = nullptr; };
extern thread_local Gate t_gate;                             // set around the gated fallback sequence; {} = no gate
__device__ __forceinline__ bool gate_closed(const Gate& g) { return g.word != nullptr && *g.word != g.gen; }
#else
// (product library, round 6: no gated launch exists -- DCE_FP32_SPLIT lives in the experiments build.  The fp32 kernels keep the parameter as an EMPTY
//  type, so that one source serves both builds; the test below folds to false and leaves no instruction behind.)
struct Gate {};
extern thread_local Gate t_gate;
__device__ __forceinline__ constexpr bool gate_closed(const Gate&) { return false; }
#endif

// ---- device buffers of a context (dev_alloc.hip): hipMalloc / hipFree, or -- guard != 0 -- a mapping per buffer that ends (1) / starts (2) at an unmapped page
hipError_t dev_alloc_raw(void** out, size_t bytes, int guard);
hipError_t dev_free_raw(void* p);
template <class T> inline hipError_t dev_alloc(T** out, int guard, size_t bytes) { return dev_alloc_raw(reinterpret_cast<void**>(out), bytes, guard); }
template <class T> inline hipError_t dev_free(const T* p) { return dev_free_raw(const_cast<void*>(static_cast<const void*>(p))); }

// ---- which kernels a call ran: every launcher notes the kernel family it picked; dce_last_plan returns the notes of
// the ctx's most recent kernel sequence (tests assert that an A/B switch or a batch size really selected the kernel
// they mean to exercise)
extern thread_local std::vector<const char*>* t_plan;
inline void plan_note(const char* kernel) { if (t_plan) t_plan->push_back(kernel); }

// ---- packed conv weights (built once in dce_finalize_weights) ---------------------------
// Layer l has CinPad (multiple of 8) input rows and Cout output channels.  The implicit
// GEMM runs K in the order (channel group of 8, tap, channel); a K-step is 2 channels (MFMA 32x32x2), a group is
// 4 steps = 8 channels = one float4 per lane.  Element [mtile][g][tap][lane][u] holds
//   w[cout = 32*mtile + (lane&31)][cin = 8*g + 2*u + (lane>>5)][tap]     (0 if cin >= Cin)
struct ConvPack {
    const float* w[4];   // direct-form packed weights per layer (device)
    const float* ww[4];  // Winograd F(2,3)-transformed packed weights per layer (device)
    const float* b[4];   // bias per layer (device), PyTorch order
};
size_t conv_pack_floats(int layer);                                  // floats in layer's pack
void   conv_pack_host(int layer, const float* w_torch, float* out);  // [Cout][Cin][3] -> pack

// Winograd F(2,3) variant (conv_wino.hip): U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) per (cout,cin),
// element [mtile_pair][kstep][lane][mt(2)][comp(4)] = U_comp[32*pair + 16*mt + (lane&15)][4*kstep + (lane>>4)]
size_t conv_wino_pack_floats(int layer);
void   conv_wino_pack_host(int layer, const float* w_torch, float* out);
hipError_t init_conv_wino();
//   src_row (optional, device memory): the kernel adds *src_row rows to src -- the online graph keeps the
//   position of the live window on the device so that its launch parameters never change
hipError_t launch_conv_wino(const float* src, int zscore, int64_t n, const ConvPack& pk,
                            void* feat, int feat_bf16, hipStream_t st, const long long* src_row = nullptr);

// Per-layer taps of the fused conv stack for parity tests (dce_conv_layer_taps): post-ReLU activations in PyTorch
// layout, one block per window -- conv1 (n,64,150), conv2 (n,64,150) before the pool, pool1 (n,64,75), conv3 (n,128,75),
// conv4 (n,128,75) before the pool.  The write-backs of TAPS instantiations of the conv kernels store them next to
// their LDS / feature stores; the product instantiations carry no trace of it.
struct LayerTaps { float *conv1, *conv2, *pool1, *conv3, *conv4; };
// kernel: 0 two-window Winograd, 1 one-window x 8 waves, 2 half-window segments, 3 quarter-window segments,
//         4 direct form, 5 one-window x 4 waves, 6 two-window Winograd with four row tiles per wave (DCE_CONV4=1),
//         7 (dce_api.hip; DCE_FP32_SPLIT contexts only) conv_x3.hip: three-term bf16 operands
hipError_t launch_conv_taps(int kernel, const float* windows, int64_t n, const ConvPack& pk, float* feat,
                            const LayerTaps& taps, hipStream_t st);

// Per-device one-time setup (dynamic-LDS grants); call after hipSetDevice.
hipError_t init_conv_stack();
hipError_t init_fc_gemm();

// Fused z-score + conv1..conv4 + ReLU + 2x MaxPool for n windows -> feat (n,4736).
//   zscore != 0: src is a raw (T,54) sequence; window i = rows [first+i, first+i+150)
//   zscore == 0: src is (n,150,54) pre-normalised windows, window i at src + i*8100
//   feat_bf16 == 1: feat is (n,4736) bf16 (round-to-nearest-even) for the bf16 FC path; == 2 (Winograd two-window kernel
//   only): three bf16 planes, v = t1 + t2 + t3, in fc_gemm_x3.hip's layout (DCE_FP32_SPLIT)
hipError_t launch_conv_stack(const float* src, int zscore, int64_t n, const ConvPack& pk,
                             void* feat, int feat_bf16, hipStream_t st, const long long* src_row = nullptr);

// z-scored windows only: out (n,150,54) from seq rows [first, first+n+149)
hipError_t launch_zscore_windows(const float* seq_first_row, int64_t n, float* out, hipStream_t st);

// C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]); fp32 MFMA, N % 128 == 0, K % 32 == 0.
hipError_t launch_fc_gemm(const float* A, const float* W, const float* bias, float* C,
                          int64_t M, int N, int K, int relu, hipStream_t st);

// The same layer for M <= 32 windows (used up to 8: online mode / batch_size 1; from 9 windows the chain kernel below is
// faster): weight-streaming GEMV on all CUs,
// bit-identical to launch_fc_gemm (same K order).  N % 8 == 0, K % 128 == 0.
constexpr int FC_GEMV_MAX_M = 32;
hipError_t init_fc_gemv();
hipError_t launch_fc_gemv(const float* A, const float* W, const float* bias, float* C,
                          int64_t M, int N, int K, int relu, hipStream_t st);

// The same layers for 9 .. a few hundred windows (fc_gemm_chain.hip): one 16x16 output tile per wave on
// v_mfma_f32_16x16x4_f32, one wave per SIMD -- bound by the length of an output's fma chain, not by throughput.
// Bit-identical to the other fp32 FC kernels (same K order).  N % 32 == 0, K % 128 == 0.
hipError_t init_fc_gemm_chain();
bool       fc_gemm_chain_ok(int64_t M, int N, int K);
hipError_t launch_fc_gemm_chain(const float* A, const float* W, const float* bias, float* C,
                                int64_t M, int N, int K, int relu, hipStream_t st);

// The same layers for 9 .. ~100 windows (fc_gemm_split.hip): one 16x16 tile of C per workgroup, the four K ranges of the
// summation tree on its four waves -- bound by the longest range's MFMA chain (fc.0: 1280 links), not by 4736.
hipError_t init_fc_split();
bool       fc_split_ok(int64_t M, int N, int K);
hipError_t launch_fc_split(const float* A, const float* W, const float* bias, float* C,
                           int64_t M, int N, int K, int relu, hipStream_t st);

// Same GEMM on bf16 operands (v_mfma_f32_32x32x16_bf16, fp32 accumulate): A[M,K], W[N,K] bf16,
// C fp32 or bf16 (out_bf16).  N % 128 == 0, K % 64 == 0.  DCE_BF16_FC precision only.
hipError_t launch_fc_gemm_bf16(const void* A, const void* W, const float* bias, void* C, int out_bf16,
                               int64_t M, int N, int K, int relu, hipStream_t st);

// fc.0 with fp32 operands carried as three bf16 terms each (fc_gemm_x3.hip, precision DCE_FP32_SPLIT): A3 = [3][M][K],
// W3 = [3][N][K] bf16 planes (launch_split3 / split3_host make them), C fp32.  256x128 tiles that fill the chip only.
hipError_t init_fc_gemm_x3();
bool       fc_gemm_x3_ok(int64_t M, int N, int K);
void       split3_host(const float* x, size_t rows, size_t cols, unsigned short* planes);
hipError_t launch_split3(const float* x, unsigned short* planes, int64_t rows, int cols, hipStream_t st);
//   out_planes: C leaves as three ROW-MAJOR bf16 planes [3][M rounded up to even][N] (unsigned short), the next layer's three-term operand
hipError_t launch_fc_gemm_x3(const unsigned short* A3, const unsigned short* W3, const float* bias, void* C,
                             int64_t M, int N, int K, int relu, hipStream_t st, int out_planes = 0);
// fc.3 + fc.6 chunk sums on three-term operands (round 5): the 128 x 64 tile of the same kernel with the fused fc.6 epilogue
bool       fc23_x3_ok(int64_t M);
void       split3_rows_host(const float* x, size_t rows, size_t cols, unsigned short* planes);
hipError_t launch_fc23_fused_x3(const unsigned short* h1p, const unsigned short* W2p, const float* b2, const float* W3, float* part, int64_t part_rows,
                                float* h2_out, int64_t M, hipStream_t st);

// The conv stack with three-term bf16 operands (conv_x3.hip, precision DCE_FP32_SPLIT): direct-form implicit GEMM on
// v_mfma_f32_16x16x32_bf16, one window per workgroup; features leave as the three planes fc_gemm_x3.hip reads.
struct ConvPackX3 { const unsigned short* w[4]; const float* b[4]; };
size_t     conv_x3_pack_halfs(int layer);
void       conv_x3_pack_host(int layer, const float* w, unsigned short* out);
hipError_t init_conv_x3();
//   permk: the features leave straight from the accumulators in the K order k' = t' * 128 + c (no LDS staging, no barriers); fc.0
//   behind it then takes weights whose K axis is permuted the same way (fc_perm_k_host)
hipError_t launch_conv_x3(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat3, hipStream_t st, int permk = 0, const GuardArgs& guard = GuardArgs{});
hipError_t launch_conv_x3_bf16(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat, hipStream_t st, int permk = 0, int terms = 3, const long long* src_row = nullptr);
hipError_t launch_conv_x3_f32(const float* src, int zscore, int64_t n, const ConvPackX3& pk, float* feat, hipStream_t st, const GuardArgs& guard = GuardArgs{});
hipError_t launch_conv_x3_taps(const float* windows, int64_t n, const ConvPackX3& pk, unsigned short* feat3, float* feat32,
                               const LayerTaps& taps, hipStream_t st);
// The same stack for chip-filling batches (conv_x3p.hip): one persistent workgroup per CU, two windows a fixed three phases
// apart (one's write-back beside the other's MFMAs), next window by LDS-DMA.  The features leave in the K order
// k' = t' * 128 + c (not the reference's flatten order c * 37 + t'): fc.0 behind it takes weights whose K axis is permuted
// the same way (fc_perm_k_host).
hipError_t init_conv_x3p();
hipError_t launch_conv_x3p(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat3, hipStream_t st);
hipError_t launch_conv_x3p_bf16(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat, hipStream_t st);
// out[o][t' * 128 + c] = w[o][c * 37 + t'] for the (rows, 4736) fc.0 weight
void       fc_perm_k_host(const float* w, size_t rows, float* out);

// ---- DCE_FP32_F16X2 (conv_h2.hip, fc_gemm_h2.hip): fp32-tolerance results from TWO fp16 terms per operand (three MFMAs per product), every
// operand scaled by a power of two -- per layer for the weights (sw, fixed at dce_finalize_weights), per window and layer for the
// activations (chosen by the conv kernel from the layer's largest output) -- so that no input can leave fp16's range.
struct ConvPackH2 {
    const unsigned short* w[4];   // per-lane packs of the two-term weights (conv_h2_pack_host)
    const float* b[4];            // biases, PyTorch order
    int sw[4];                    // weight scale exponents: max|w_l| * 2^sw in [2^14, 2^15)
    int smax[5];                  // the largest scale exponent the INPUT of conv layer l may carry (the scaled bias stays below 2^60); [4]: the features'
};
int        h2_weight_shift(const float* w, size_t n);                 // INT_MIN: a non-finite weight (the precision then runs the DCE_FP32 kernels)
int        h2_input_smax(const float* bias, size_t n, int sw);        // INT_MIN: a non-finite bias
size_t     conv_h2_pack_halfs(int layer);
void       conv_h2_pack_host(int layer, const float* w, int sw, unsigned short* out);
void       fc_h2_pack_host(const float* w, size_t rows, size_t K, int sw, unsigned short* out);      // [row][K-tile of 32][term (2)][32 fp16]
hipError_t init_conv_h2();
//   feat2: (n, 2 x 4736) fp16 in fc_gemm_h2.hip's operand layout, K in the order t' * 128 + c; feat_scale: (n) scale exponents of the rows
hipError_t launch_conv_h2(const float* src, int zscore, int64_t n, const ConvPackH2& pk, unsigned short* feat2, int* feat_scale, hipStream_t st);
//   ... with (n, 4736) fp32 features out, unscaled, in the reference's flatten order (mid-size batches: the FC layers on the fp32 kernels)
hipError_t launch_conv_h2_f32(const float* src, int zscore, int64_t n, const ConvPackH2& pk, float* feat, hipStream_t st);
//   ... with (n, 4736) bf16 features out (nearest-even of the unscaled values), K in the order t' * 128 + c: DCE_BF16_FC with the option bf16_conv_h2
hipError_t launch_conv_h2_bf16(const float* src, int zscore, int64_t n, const ConvPackH2& pk, unsigned short* feat, hipStream_t st);
//   ... with every layer's output and the features (n, 4736, the reference's flatten order) also written out in fp32, unscaled (dce_conv_layer_taps kernel 8)
hipError_t launch_conv_h2_taps(const float* windows, int64_t n, const ConvPackH2& pk, unsigned short* feat2, int* feat_scale, float* feat32,
                               const LayerTaps& taps, hipStream_t st);
// fc.0 on two-term fp16 operands: C = act(A W^T * 2^-(row_scale[m] + sw) + bias), 256 x 128 tiles that fill the chip only
hipError_t init_fc_gemm_h2();
bool       fc_gemm_h2_ok(int64_t M, int N, int K, int min_tiles);
int        fc_gemm_h2_pad_rows();                                     // rows A2's buffer must hold beyond M (ragged tiles read them)
//   H1 != NULL: h1 leaves as two fp16 terms [row][N / 32][2][32] with its row scales in h1_scale (the operand of fc.3 below) instead of fp32 in C
hipError_t launch_fc_gemm_h2(const unsigned short* A2, const int* row_scale, const unsigned short* W2, int sw, const float* bias, float* C,
                             int64_t M, int N, int K, int relu, hipStream_t st, unsigned short* H1 = nullptr, int* h1_scale = nullptr, int eW = 0, int eB = 0);
// fc.3 + fc.6 chunk sums on two-term fp16 operands (the 128 x 64 tile of the K-split kernel with the fused fc.6 epilogue)
bool       fc23_h2_ok(int64_t M);
//   ... and the bf16-FC mode's fc.3 on the same kernel, one bf16 term per operand (h1, W2 row-major bf16)
hipError_t launch_fc23_fused_bf16k(const void* h1, const void* W2, const float* b2, const float* W3, float* part, int64_t part_rows, float* h2_out, int64_t M, hipStream_t st);
hipError_t launch_fc23_fused_h2(const unsigned short* h1, const int* h1_scale, const unsigned short* W2p, int sw, const float* b2, const float* W3,
                                float* part, int64_t part_rows, float* h2_out, int64_t M, hipStream_t st);

// The same GEMMs at chip-filling sizes (fc_gemm_phased.hip): one workgroup per CU, 256x128 or 128x64 tiles,
// LDS-DMA staging, two wave groups one phase apart; fp32 (bit-identical to the tile kernels: same K order) and
// bf16.  launch_fc_gemm / launch_fc_gemm_bf16 dispatch here when fc_gemm_phased_ok(M, N, K, bf16).
hipError_t init_fc_gemm_phased();
// fc_stream_bf16.hip: fc.0 / fc.3 of DCE_BF16_FC for calls of up to 256 windows (weights streamed past the activations, eight waves deal the K-steps)
bool fc_stream_bf16_ok(int64_t M, int N, int K);
hipError_t launch_fc_stream_bf16(const void* A, const void* W, const float* bias, void* C, int out_bf16, int64_t M, int N, int K, int relu, hipStream_t st);
bool       fc_gemm_phased_ok(int64_t M, int N, int K, int bf16);
hipError_t launch_fc_gemm_phased(const void* A, const void* W, const float* bias, void* C, int bf16, int out_bf16,
                                 int64_t M, int N, int K, int relu, hipStream_t st);

// fc.3 (+ReLU) with fc.6's chunk sums finished in the GEMM epilogue (fp32, chip-filling batches; fc6_chain.h):
// A = h1 (M,2048), W2 (512,2048) -- fp32, or bf16 in the DCE_BF16_FC mode (h2 and everything after it stay fp32);
// b2; part: [8][part_rows][16] chunk sums out; h2_out: NULL, or (M,512) fp32 for taps.
bool       fc23_fused_ok(int64_t M, int bf16);
hipError_t launch_fc23_fused(const void* h1, const void* W2, const float* b2, const float* W3, int bf16,
                             float* part, int64_t part_rows, float* h2_out, int64_t M, hipStream_t st);
// ... and the combine behind it: logits = ordered sum of the 8 chunk sums + b3, argmax, contact bits
hipError_t launch_fc6_combine(const float* part, int64_t part_rows, const float* b3, int64_t n,
                              float* logits, int32_t* pred, uint8_t* contacts, hipStream_t st, uint8_t* packed = nullptr);

// logits = h2 * W3^T + b3 (same summation tree, fc6_chain.h) ; argmax (first max, NaN-first) ; 4-bit unpack (MSB = leg 0)
// done_flag (optional, single-block launches only): a system-scope release store of done_seq after the
// outputs, for a host that polls instead of synchronising the stream (online mode).
//   seq_counter (optional, device memory, with done_flag): publish ++*seq_counter instead of done_seq.
hipError_t launch_fc3_tail(const float* h2, const float* W3, const float* b3, int64_t n,
                           float* logits, int32_t* pred, uint8_t* contacts, hipStream_t st,
                           unsigned* done_flag = nullptr, unsigned done_seq = 0, unsigned* seq_counter = nullptr,
                           uint8_t* packed = nullptr);
// packed: optional (n,68)-byte rows -- 16 fp32 logits followed by the 4 contact bits -- the row format of the multi-GPU
// gather (dce_gather_results), written by the same kernels so that no repacking pass exists.
constexpr int PACKED_ROW = 68;
// (n,68) packed rows -> logits (n,16) f32, pred (n) i32, contacts (n,4) u8 (any may be NULL)
hipError_t launch_unpack_results(const uint8_t* packed, int64_t n, float* logits, int32_t* pred, uint8_t* contacts, hipStream_t st);

// online mode: write one (54,) sample, carried in the kernel arguments, to its row of the sample buffer
struct OnlineSample { float v[54]; };
hipError_t launch_online_append(float* row, const OnlineSample& s, hipStream_t st);

// online mode as ONE hipGraph launch per sample: every kernel of the push has constant launch
// parameters because the moving parts live in memory -- the sample in pinned host memory (read by the
// append kernel), the write cursor / window start / sequence number in this device-resident state.
struct OnlineState { long long src_row; int cursor; unsigned seq; };
constexpr int ONLINE_ROWS = 4096;     // rows of the sample buffer; the last 149 move to the front when it is full
hipError_t launch_online_append_state(float* ring, OnlineState* state, const float* sample_host, hipStream_t st);

// ---- latency mode (latency.hip; option latency=1): one window through the whole net in ONE kernel of 256 co-resident workgroups
struct LatSync { unsigned long long feat; unsigned quit, started; };           // fine-grained device memory: arrivals of the conv workgroups (monotonic: request s waits for 8 s), the service's quit word, conv workgroups of the service that hold their snapshot of the history (zeroed with the rest before every start)
struct LatMailbox {                                                            // pinned host memory, device-visible, coherent
    unsigned req;              // host -> device: number of the newest request (written LAST, release)
    unsigned kind;             //   0 append the sample, 1 append + estimate, 2 quit
    float    sample[54];
    unsigned ack[8];           // device -> host: conv workgroup i has taken request ack[i] (all eight: the sample slot is free)
    unsigned alive;            //   1 while the service kernel runs
    unsigned error;            //   a wait ran into its deadline (one shot or service)
    // The estimate: TWO 64-byte lines, each written by ONE 16-lane store and each carrying the estimate's number in its last word.  The
    // device's writes to host memory may arrive line by line in any order (PCIe relaxed ordering: a separate "done" word was seen by the
    // host ahead of the logits' line); a line, though, arrives whole: the host waits until BOTH lines carry the number it expects.
    struct alignas(64) LineA { float logits[15]; unsigned tag; } a;
    struct alignas(64) LineB { float logit15; int pred; unsigned char contacts[4]; unsigned pad[12]; unsigned tag; } b;
};
static_assert(sizeof(LatMailbox::LineA) == 64 && sizeof(LatMailbox::LineB) == 64, "one line each");
struct LatArgs {
    ConvPack pk;
    const float *w1, *b1, *w2, *b2, *w3, *b3;                                  // fc.0 / fc.3 / fc.6 in PyTorch layout
    float* feat;                                                               // fine-grained device memory: one row of features,
    unsigned long long *h1, *h2;                                               // ... h1 / h2 as (value, tag) words: tag = the number of the request that wrote them
    LatSync* sync;
    LatMailbox* mbox;
    unsigned long long seq;                                                    // number of this (first) request (>= 1; counted from the last zeroing of the exchange memory)
    unsigned long long deadline_ticks, idle_ticks;                             // 100 MHz ticks: every wait's deadline; service: how long to wait for a request
    // one shot
    const float* src; float* logits; int32_t* pred; uint8_t* contacts; uint8_t* packed;
    // service
    float* hist; int* hist_state;                                              // device copy of the last 150 samples [150][54] and {head, count}
    unsigned req_base, done_base;                                              // mailbox numbers at launch (requests posted / estimates delivered so far)
    unsigned long long fc_delay_ticks;                                         // (A/B) the fc.0 role starts its weight stream this much later
    unsigned long long* trace;                                                 // NULL, or 16 wall-clock stamps of the last request (tools/latency_mode.py)
    // micro-batch form (latency_mb.hip): 2 .. LATMB_MAX_N windows in one launch
    int mb_n, mb_chalf;                                                        // windows; 1: two conv workgroups per quarter segment (n <= 16)
    const float *mb_w1, *mb_w2;                                                // fc.0 / fc.3 packed per (tile, wave, granule, lane) (latmb_pack_host)
    float *mb_feat, *mb_h1, *mb_plt;                                           // device memory: features [4736 / 4][32][4], h1 [2048 / 4][32][4] (latency_mb.hip's quad layout), partial logits [32 tiles][32][16]
    unsigned long long* mb_flags;                                              // fine-grained device memory: [128] conv + [128] fc.0 + [64] fc.3 producer flags (the request's number), [320 .. 576) traced runs' stamps, [576 .. 704) fc.0's second row tile
};
constexpr int LATMB_MAX_N = 32;
size_t     latmb_pack_floats(int rows, int K);
void       latmb_pack_host(const float* W, int rows, int K, float* out, int chan);      // chan > 0: columns re-ordered position-major (fc.0: 128)
hipError_t init_latency_mb();
hipError_t launch_latency_mb(int zscore, const LatArgs& a, hipStream_t st);
hipError_t init_latency();
int        latency_grid();                                                     // workgroups = CUs the mode needs
hipError_t launch_latency(int mode /* 0 one window, pre-normalised; 1 raw rows (z-score fused); 2 service */, const LatArgs& a, hipStream_t st);

// counts[gt*16 + pred] += 1 over n (pred, label) pairs; out-of-range classes are skipped
hipError_t launch_confusion16(const int32_t* pred, const int64_t* label, int64_t n,
                              unsigned long long* counts, hipStream_t st);

}  // namespace dce
