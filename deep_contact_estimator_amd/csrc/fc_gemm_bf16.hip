// fc_gemm_bf16.hip -- fc.0 of the DCE_BF16_FC precision mode (BASELINE configs[4]) on gfx950:
//     C[M,N] = act(A[M,K] W[N,K]^T + bias),  A, W bf16 (K-contiguous, PyTorch's [out][in]),  fp32 accumulate
// on v_mfma_f32_32x32x16_bf16 (reference src/contact_cnn.py:48-50 with bf16 operands).
//
// Why a second GEMM kernel: at bf16 rate the 128x128-tile kernel of fc_gemm.hip is bound by operand
// movement, not by the matrix pipe -- per 512 matrix-pipe cycles a block stages 32 KB through VGPRs
// and ds_write_b128 (~79 B/clk/CU) and reads 64 KB back, and two such blocks per CU ask L2 for
// 64 B/clk/CU.  This kernel cuts all three:
//   * 256 x 128 block tile (one block per CU at 4096 x 2048): 48 KB of operands per 1024 pipe
//     cycles = 47 B/clk/CU from L2;
//   * operands go global -> LDS directly (global_load_lds_dwordx4, 1 KB per wave-instruction): no
//     staging VGPRs, no ds_write pass; three 48 KB LDS buffers, two K-tiles of loads in flight;
//   * 8 waves = two groups of four (waves w and w+4 share a SIMD and sit in different groups) that
//     run ONE PHASE APART: while a group issues its 16 MFMAs of a K-tile (512 cycles), the other
//     group reads its fragments of the next step from LDS and issues the loads of tile t+2; a
//     workgroup barrier ends every phase.  The matrix pipe of every SIMD always has one wave in
//     its math phase.
// LDS image: a K-tile is 384 rows (256 of A, 128 of W) x 128 B; LDS-DMA writes lane-linear (8 rows
// per 1 KB wave-instruction), so the bank swizzle lives in the per-lane GLOBAL address: 16-byte
// column c of row r is stored at slot c ^ ((r >> 1) & 7); the fragment reads (lane = row, fixed
// logical column) apply the same XOR and are conflict-free for ds_read_b128's 16-lane groups.
#include "dce_kernels.h"
#include <type_traits>

namespace dce {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int PH_BM = 256, PH_BN = 128, PH_ROWS = PH_BM + PH_BN;   // stacked rows of one K-tile
constexpr int PH_ROWB = 128;                                       // bytes of K per row per tile (64 bf16)
constexpr int PH_TILE = PH_ROWS * PH_ROWB;                         // 49,152 B
constexpr int PH_NBUF = 3;
constexpr int PH_LDS = PH_NBUF * PH_TILE;                          // 147,456 B
constexpr int PH_GLDS = PH_ROWS / 8 / 8;                           // 1 KB chunks per wave per K-tile = 6

__device__ __forceinline__ unsigned short f32_to_bf16(float f)
{   // round-to-nearest-even; NaN stays NaN (quiet)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// the six 1 KB chunks this wave brings in per K-tile: chunk j lands at lds0 + j*8 KB (+ lane*16).
// sA / sW: wave-uniform bases of the block's A / W panel at this K-tile; v0..v5: per-lane byte offsets.
// M0 carries the LDS destination; it is compiler-reserved, so it is saved and restored inside the statement.
__device__ __forceinline__ void issue_tile(unsigned lds0, const char* sA, const char* sW,
                                           unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned v4, unsigned v5)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %4, %2\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %5, %2\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %6, %2\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %7, %2\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %8, %3\n\t"
                 "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %9, %3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds0), "s"(sA), "s"(sW), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(v4), "v"(v5)
                 : "memory");
}

// end of a phase: this wave's LDS-DMA of all but the newest N pieces has landed, its own fragment reads have
// returned (so the buffer they came from may be refilled after the barrier), then the workgroup barrier
#define PH_WAIT_BARRIER(N) asm volatile("s_waitcnt vmcnt(" #N ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

// 32-bit LDS byte address of a __shared__ object (what M0 takes for LDS-DMA)
__device__ __forceinline__ unsigned lds_addr(const void* p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)p;
}

}  // namespace

template <bool OUT_BF16>
__global__ __launch_bounds__(512, 2)
void fc_gemm_bf16_phased_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ W,
                                const float* __restrict__ bias, void* __restrict__ Cv,
                                int M, int N, int K, int relu, int mtiles, int ntiles, int sn_log2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // ---- XCD-aware tile assignment (speed only): the 32 blocks co-resident on one XCD form an sm x sn super-tile
    const int bid = blockIdx.x;
    const int xcd = bid & 7, li = bid >> 3;
    const int sid = (li >> 5) * 8 + xcd;
    const int within = li & 31;
    const int sn = 1 << sn_log2, sm = 32 >> sn_log2;
    const int nsn = ntiles >> sn_log2;
    const int tm = (sid / nsn) * sm + (within >> sn_log2);
    const int tn = (sid % nsn) * sn + (within & (sn - 1));
    if (tm >= mtiles) return;
    const int m0 = tm * PH_BM, n0 = tn * PH_BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wid >> 2;                            // phase group; waves w and w+4 share a SIMD
    const int wm = (wid & 3) * 64, wn = grp * 64;        // this wave's 64 x 64 corner of the block tile
    const int i = lane & 31, h = lane >> 5;

    // ---- global -> LDS: chunk c = wid + 8 j (j = 0..5) of the stacked tile; chunks 0..31 are A rows, 32..47 W rows
    const size_t rowb = (size_t)K * 2;
    unsigned voff[PH_GLDS];
#pragma unroll
    for (int j = 0; j < PH_GLDS; ++j) {
        const int c = wid + 8 * j;
        const int r = 8 * c + (lane >> 3);               // row of the stacked tile
        const int slot = lane & 7;                       // 16-byte slot this lane fills
        const int col = slot ^ ((r >> 1) & 7);           // logical 16-byte column that lives there
        int grow = c < 32 ? r : r - PH_BM;               // row inside the A / W panel
        if (c < 32 && m0 + grow >= M) grow = M - 1 - m0; // rows past M re-read the last one (never stored)
        voff[j] = (unsigned)(grow * rowb + 16 * col);
    }
    const char* sA = reinterpret_cast<const char*>(A) + (size_t)m0 * rowb;
    const char* sW = reinterpret_cast<const char*>(W) + (size_t)n0 * rowb;
    const unsigned lds_base = lds_addr(smem);
    const unsigned lds_wave = lds_base + wid * 1024;     // chunk wid of buffer 0

    // ---- fragment reads: lane (i, h) reads row (w? + 32 a + i), logical column 2 ks + h
    const int sw = (i >> 1) & 7;
    int fo[4];                                           // swizzled 16-byte column offset per k-step
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = 16 * ((2 * ks + h) ^ sw);
    const int arow = (wm + i) * PH_ROWB, brow = (PH_BM + wn + i) * PH_ROWB;

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int KT = K / 64;                               // >= 3 (checked by the launcher)
    // prologue: tiles 0 and 1 in flight; tile 0 landed for everybody before the first phase
    issue_tile(lds_wave, sA, sW, voff[0], voff[1], voff[2], voff[3], voff[4], voff[5]);
    issue_tile(lds_wave + PH_TILE, sA + PH_ROWB, sW + PH_ROWB, voff[0], voff[1], voff[2], voff[3], voff[4], voff[5]);
    PH_WAIT_BARRIER(6);
    if (grp == 1) PH_WAIT_BARRIER(6);                    // group 1 runs one phase behind group 0

    int buf = 0, nbuf = 2;                               // buffer of tile t / of tile t+2
    for (int t = 0; t < KT; ++t) {
        // ---- load phase: loads of tile t+2, fragments of tile t
        const bool more = t + 2 < KT;
        if (more) {
            const size_t ko = (size_t)(t + 2) * PH_ROWB;
            issue_tile(lds_wave + nbuf * PH_TILE, sA + ko, sW + ko, voff[0], voff[1], voff[2], voff[3], voff[4], voff[5]);
        }
        const char* tb = smem + buf * PH_TILE;
        float4 af[4][2], bf[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                af[ks][a] = *reinterpret_cast<const float4*>(tb + arow + a * 32 * PH_ROWB + fo[ks]);
                bf[ks][a] = *reinterpret_cast<const float4*>(tb + brow + a * 32 * PH_ROWB + fo[ks]);
            }
        if (more) PH_WAIT_BARRIER(6); else PH_WAIT_BARRIER(0);
        // ---- math phase
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                        __builtin_bit_cast(bf16x8, af[ks][a]), __builtin_bit_cast(bf16x8, bf[ks][b]), acc[a][b], 0, 0, 0);
        if (more) PH_WAIT_BARRIER(6); else PH_WAIT_BARRIER(0);
        buf = buf == 2 ? 0 : buf + 1;
        nbuf = nbuf == 2 ? 0 : nbuf + 1;
    }
    if (grp == 0) PH_WAIT_BARRIER(0);                    // same number of barriers for both groups

    // ---- epilogue: bias + (ReLU); D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    auto store_tile = [&](auto full) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int col = n0 + wn + 32 * b + i;
            const float bv = bias[col];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + wm + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float v = acc[a][b][r] + bv;
                    if (relu) v = v < 0.f ? 0.f : v;              // keeps NaN like torch
                    if (decltype(full)::value || row < M) {
                        if constexpr (OUT_BF16) static_cast<unsigned short*>(Cv)[(size_t)row * N + col] = f32_to_bf16(v);
                        else static_cast<float*>(Cv)[(size_t)row * N + col] = v;
                    }
                }
        }
    };
    if (m0 + PH_BM <= M) store_tile(std::true_type{});   // whole tile in range: no per-store predicate
    else store_tile(std::false_type{});
}

hipError_t init_fc_gemm_bf16()
{
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_bf16_phased_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&fc_gemm_bf16_phased_kernel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, PH_LDS);
}

// true if this shape is one the phased kernel covers well (enough 256 x 128 tiles to fill the chip)
bool fc_gemm_bf16_phased_ok(int64_t M, int N, int K)
{
    if (N % PH_BN || K % 64 || K < 3 * 64 || M > (1 << 30)) return false;
    const int nt = N / PH_BN;
    if ((nt & (nt - 1)) != 0) return false;                                     // super-tile map wants a power of two
    if ((size_t)PH_BM * K * 2 + 16 * 8 >= (1ull << 32)) return false;          // per-lane offsets are 32-bit
    return ((M + PH_BM - 1) / PH_BM) * (N / PH_BN) >= 192;
}

hipError_t launch_fc_gemm_bf16_phased(const void* A, const void* W, const float* bias, void* C, int out_bf16,
                                      int64_t M, int N, int K, int relu, hipStream_t st)
{
    const int mtiles = (int)((M + PH_BM - 1) / PH_BM), ntiles = N / PH_BN;
    int sn_log2 = 2;                                   // super-tile 8 x 4 ...
    while ((1 << sn_log2) > ntiles) --sn_log2;         // ... or (32/ntiles) x ntiles when N is narrow
    const int sm = 32 >> sn_log2, nsn = ntiles >> sn_log2;
    const int nsuper = ((mtiles + sm - 1) / sm) * nsn;
    const int grid = ((nsuper + 7) / 8) * 8 * 32;
    const unsigned short* a = static_cast<const unsigned short*>(A);
    const unsigned short* w = static_cast<const unsigned short*>(W);
    if (out_bf16) hipLaunchKernelGGL(fc_gemm_bf16_phased_kernel<true>, dim3(grid), dim3(512), PH_LDS, st,
                                     a, w, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
    else          hipLaunchKernelGGL(fc_gemm_bf16_phased_kernel<false>, dim3(grid), dim3(512), PH_LDS, st,
                                     a, w, bias, C, (int)M, N, K, relu, mtiles, ntiles, sn_log2);
    return hipGetLastError();
}

}  // namespace dce
