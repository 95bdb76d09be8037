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

// ---- A/B switches: read from the environment ONCE PER CONTEXT in dce_create and handed to the
// launchers through a thread-local pointer that every C-ABI entry point sets for the duration of the call, so that two
// contexts of one process (tests, tools/race_screen.py) can run different kernel variants side by side.
struct Tuning {
    bool gemm_tile = false, gemm_lockstep = false;          // DCE_GEMM=tile | lockstep   (default: phased)
    bool gemm_ki = false;                                   // DCE_GEMM_KI=1 (experiments build): the bf16 256 x 128 tile on fc_gemm_ki_kernel (the two wave groups deal the K-tiles out between them: math phases twice as long; same launch time at a lower clock)
    bool gemm_pipe = false;                                 // DCE_GEMM=pipe (experiments build): the bf16 256 x 128 tile on fc_gemm_pipe_kernel (LDS counters instead of workgroup barriers in the K loop; measured slower)
    int phased_min_tiles = 192, phased_min_tiles1 = 128;    // DCE_PHASED_MIN_TILES, DCE_PHASED_MIN_TILES1
    int phased_min = 1;                                     // DCE_GEMM_PHASED_MIN (2: only the 256x128 tile)
    bool phased_cost = true;                                // DCE_PHASED_COST=0: tile minimum only, no rounds model
    int phased_sn = 3;                                      // DCE_PHASED_SN: log2 of the super-tile's N extent
    int fc23_mode = 0;                                      // DCE_FC23=split (1) | always (2)
    bool gemm_peel = true, conv_peel = true;                // DCE_GEMM_PEEL=0, DCE_CONV_PEEL=0
    bool bf16_stream = true;                                // DCE_BF16_STREAM=0: DCE_BF16_FC's fc.0 / fc.3 at <= 256 windows on the 64 x 64 tile GEMM (44 + 21 us per call) instead of fc_stream_bf16.hip
    bool gemm_small_deep = true;                            // DCE_GEMM_SMALL=0
    long long split_min = 9, split_max = 64;                 // DCE_SPLIT_MIN / DCE_SPLIT_MAX: windows served by the four-range MFMA kernel (fc_gemm_split.hip)
    long long chain_min = 9, chain_max = 640, chain_max3 = 2048, chain_bn16_max = 64;   // DCE_CHAIN_*
    long long wino1_max = -1, winoh_max = -1, winoq_max = -1;   // DCE_WINO1_MAX / DCE_WINOH_MAX / DCE_WINOQ_MAX (-1: kernel default)
    bool wino1_w8 = true;                                   // DCE_WINO1_WAVES=4 -> false
    bool one_per_cu = false, trace_wino1 = false;           // DCE_ONE_PER_CU, DCE_TRACE_WINO1 (trace builds)
    bool x3_conv = true;                                    // DCE_X3_CONV=0: DCE_FP32_SPLIT keeps the fp32 Winograd conv stack (three-plane feature output) instead of conv_x3.hip (A/B)
    long long x3_conv_min = 128;                            // DCE_X3_CONV_MIN: from this many windows the mode's conv stack runs on conv_x3.hip also BELOW the fc.0 threshold (fp32 features out)
    bool x3_unfused = false;                                // DCE_X3_UNFUSED: fp32 features + split3 kernel instead of the conv kernel's three-plane output (A/B)
    bool x3_permk = true;                                   // DCE_X3_PERMK=0: conv_x3.hip's features go through LDS into the reference's flatten order (A/B) instead of straight out in the order t' * 128 + c
    long long x3_bf16_min = 1;                              // DCE_X3_BF16_MIN: DCE_BF16_FC's conv stack runs on conv_x3.hip from this many windows (default: always -- one window per workgroup is also the fastest form at batch 1: 18.7 us against 20.8 for the fp32 quarter-window kernel and 62 for the two-window kernel the mode used below 128 windows)
    int x3_bf16_terms = 2;                                  // DCE_X3_BF16_TERMS=3: DCE_BF16_FC's conv stack on three-term operands (six MFMAs per product) as DCE_FP32_SPLIT's; default two terms (three MFMAs, ~17 significant bits ahead of the features' 8-bit rounding: same error against an fp64 evaluation, profiles/r4h_bf16_terms_audit.json; 308 -> 171 us per 4096 windows)
    bool x3_persist = false;                                // DCE_X3_PERSIST=1 (experiments build): conv_x3.hip as persistent workgroups (two per CU) that request the next window's samples a layer ahead; measured 2-4 % slower
    long long x3_persist_min = 1024;                        // DCE_X3_PERSIST_MIN: windows per launch from which they do
    bool x3_pair = false;                                   // DCE_X3_PAIR=1: chip-filling batches on conv_x3p.hip (two windows per 8-wave workgroup; measured 5-9 % SLOWER than conv_x3.hip, kept for the record and the A/B) instead of conv_x3.hip
    long long x3_pair_min = 1024;                           // DCE_X3_PAIR_MIN: windows per launch from which conv_x3p.hip runs
    int x3_min_tiles = 192;                                 // DCE_X3_MIN_TILES: 256x128 tiles a launch needs for the split-bf16 fc.0 kernel
    int conv4 = 0;                                          // DCE_CONV4=1: four row tiles per wave in the two-window conv kernel (A/B; slower)
};
Tuning tuning_from_env();
extern thread_local const Tuning* t_tuning;                  // the calling ctx's switches (nullptr: process defaults)
const Tuning& tune();
struct TuningScope {                                         // RAII: entry points bind their ctx's switches
    const Tuning* prev;
    explicit TuningScope(const Tuning* t) : prev(t_tuning) { t_tuning = t; }
    ~TuningScope() { t_tuning = prev; }
};

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
hipError_t launch_fc_gemm_x3(const unsigned short* A3, const unsigned short* W3, const float* bias, float* C,
                             int64_t M, int N, int K, int relu, hipStream_t st);

// The conv stack with three-term bf16 operands (conv_x3.hip, precision DCE_FP32_SPLIT): direct-form implicit GEMM on
// v_mfma_f32_16x16x32_bf16, one window per workgroup; features leave as the three planes fc_gemm_x3.hip reads.
struct ConvPackX3 { const unsigned short* w[4]; const float* b[4]; };
size_t     conv_x3_pack_halfs(int layer);
void       conv_x3_pack_host(int layer, const float* w, unsigned short* out);
hipError_t init_conv_x3();
//   permk: the features leave straight from the accumulators in the K order k' = t' * 128 + c (no LDS staging, no barriers); fc.0
//   behind it then takes weights whose K axis is permuted the same way (fc_perm_k_host)
hipError_t launch_conv_x3(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat3, hipStream_t st, int permk = 0);
hipError_t launch_conv_x3_bf16(const float* src, int zscore, int64_t n, const ConvPackX3& pk, unsigned short* feat, hipStream_t st, int permk = 0, int terms = 3, const long long* src_row = nullptr);
hipError_t launch_conv_x3_f32(const float* src, int zscore, int64_t n, const ConvPackX3& pk, float* feat, hipStream_t st);
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

// counts[gt*16 + pred] += 1 over n (pred, label) pairs; out-of-range classes are skipped
hipError_t launch_confusion16(const int32_t* pred, const int64_t* label, int64_t n,
                              unsigned long long* counts, hipStream_t st);

}  // namespace dce
