// What does ONE instruction of a given kind cost when it sits between back-to-back v_mfma_f32_16x16x32_bf16 of the same wave, with
// one and with two waves per SIMD?  Wall clock (HIP events) against the MFMA-only loop of the same shape, so that neither the
// unit of s_memtime nor DVFS enters: cost = (t(NV) - t(0)) / (MFMAs x NV) in units of t(0) / MFMAs (= one MFMA slot, 16 cycles).
// hipcc --offload-arch=gfx950 -O3 -o bf16_mfma_mix.bin bf16_mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Kind { K_AND, K_ADDU, K_LSHL, K_MAXF, K_CNDMASK, K_CVTPK, K_PKADD, K_SUBF, K_FMA, K_DPP, K_DSW64, K_DSW32, K_DSR128, K_GLD, K_GST, K_NOP, K_WAIT, K_SMOV, K_GST_CONTIG, K_GST_DUP256, K_GST_SEG32, K_GST4_CONTIG, K_NKINDS };
static const char* kKind[] = {"v_and_b32", "v_add_u32", "v_lshlrev_b32", "v_max_f32", "v_cndmask_b32", "v_cvt_pk_bf16_f32", "v_pk_add_f32", "v_sub_f32", "v_fma_f32",
                              "v_mov_b32_dpp", "ds_write_b64", "ds_write_b32", "ds_read_b128", "global_load_dwordx4", "global_store_dwordx2", "s_nop 0", "s_waitcnt (met)", "s_mov_b32", "store x2 contiguous 512B", "store x2 256B, lanes paired", "store x2 8 x 32B @256B", "store x4 contiguous 1KB"};

template <int KIND> __device__ __forceinline__ void other(float (&x)[8], int i, unsigned ldsaddr, const float* gp, float4 (&sink)[4], const char* gp2, const char* gp3, const char* gp4, const char* gp5)
{
    float& v = x[i & 7];
    float& w = x[(i + 3) & 7];
    if constexpr (KIND == K_AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(v));
    else if constexpr (KIND == K_ADDU) asm volatile("v_add_u32 %0, 3, %0" : "+v"(v));
    else if constexpr (KIND == K_LSHL) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v));
    else if constexpr (KIND == K_MAXF) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v) : "v"(w));
    else if constexpr (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v) : "v"(w));
    else if constexpr (KIND == K_CVTPK) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v) : "v"(w));
    else if constexpr (KIND == K_PKADD) { typedef float f2 __attribute__((ext_vector_type(2))); f2 a = {v, w}; asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(a)); v = a.x; w = a.y; }
    else if constexpr (KIND == K_SUBF) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v) : "v"(w));
    else if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v) : "v"(w));
    else if constexpr (KIND == K_DPP) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v) : "v"(w));
    else if constexpr (KIND == K_DSW64) asm volatile("ds_write_b64 %0, %1" :: "v"(ldsaddr + (i & 7) * 512), "v"(*reinterpret_cast<double*>(&x[(i & 3) * 2])) : "memory");
    else if constexpr (KIND == K_DSW32) asm volatile("ds_write_b32 %0, %1" :: "v"(ldsaddr + (i & 7) * 512), "v"(v) : "memory");
    else if constexpr (KIND == K_DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(sink[i & 3]) : "v"(ldsaddr * 2 + (i & 7) * 1024) : "memory");
    else if constexpr (KIND == K_GLD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sink[i & 3]) : "v"(gp + (i & 7) * 256) : "memory");
    else if constexpr (KIND == K_GST) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(gp + (i & 7) * 256), "v"(*reinterpret_cast<double*>(&x[(i & 3) * 2])) : "memory");
    else if constexpr (KIND == K_GST_CONTIG) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(gp2 + (i & 7) * 4096), "v"(*reinterpret_cast<double*>(&x[(i & 3) * 2])) : "memory");
    else if constexpr (KIND == K_GST_DUP256) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(gp3 + (i & 7) * 4096), "v"(*reinterpret_cast<double*>(&x[(i & 3) * 2])) : "memory");
    else if constexpr (KIND == K_GST_SEG32) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(gp4 + (i & 7) * 4096), "v"(*reinterpret_cast<double*>(&x[(i & 3) * 2])) : "memory");
    else if constexpr (KIND == K_GST4_CONTIG) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(gp5 + (i & 7) * 4096), "v"(*reinterpret_cast<f32x4*>(&x[(i & 1) * 4])) : "memory");
    else if constexpr (KIND == K_NOP) asm volatile("s_nop 0");
    else if constexpr (KIND == K_WAIT) asm volatile("s_waitcnt vmcnt(63) lgkmcnt(15)");
    else if constexpr (KIND == K_SMOV) { int t; asm volatile("s_mov_b32 %0, 5" : "=s"(t)); }
}

template <int NV, int KIND, int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void k(float* out, float* gbuf, int iters)
{
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += WAVES * 64) reinterpret_cast<float*>(lds)[i] = 1.f + i * 1e-4f;
    __syncthreads();
    f32x4 acc[5];
    for (int t = 0; t < 5; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = 1.f + lane * 0.01f + i;
    float4 sink[4] = {};
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(1.f + lane * 0.001f + e); b[e] = (__bf16)(2.f - lane * 0.001f); }
    const unsigned ldsaddr = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) char*)lds + lane * 8 + (tid >> 6) * 4096;
    float* gp = gbuf + (size_t)(blockIdx.x * WAVES * 64 + tid) * 4;
    const char* wbase = reinterpret_cast<const char*>(gbuf) + (size_t)(blockIdx.x * WAVES + (tid >> 6)) * 32768;      // 32 KB per wave
    const char* gp2 = wbase + lane * 8;                                        // 512 contiguous bytes
    const char* gp3 = wbase + (((lane & 15) >> 1) * 4 + (lane >> 4)) * 8;     // 256 contiguous bytes, lanes j, j ^ 1 the same 8
    const char* gp4 = wbase + ((lane & 15) >> 1) * 256 + (lane >> 4) * 8;     // 8 segments of 32 bytes at a stride of 256
    const char* gp5 = wbase + lane * 16;                                       // 1 KB contiguous
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 30; ++m) {
            acc[m % 5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[m % 5], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < (m % 6 == 0 ? NV : 0); ++v) other<KIND>(x, it * 8 + m * NV + v, ldsaddr, gp, sink, gp2, gp3, gp4, gp5);      // five per 30 MFMAs
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == K_DSR128 || KIND == K_GLD) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int t = 0; t < 5; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += x[i];
    for (int i = 0; i < 4; ++i) s += sink[i].x + sink[i].w;
    out[blockIdx.x * WAVES * 64 + tid] = s;
}

template <int NV, int KIND, int WAVES> float run(float* out, float* gbuf)
{
    const int iters = 2000, grid = 256;
    hipLaunchKernelGGL((k<NV, KIND, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, gbuf, 20);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, KIND, WAVES>), dim3(grid), dim3(WAVES * 64), 0, 0, out, gbuf, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / (iters * 30.f * (WAVES / 4));           // ns per MFMA slot of a SIMD
}

template <int KIND> void kind(float* out, float* gbuf, float base1, float base2)
{
    const float a1 = run<1, KIND, 4>(out, gbuf), a2 = run<2, KIND, 4>(out, gbuf), a3 = run<3, KIND, 4>(out, gbuf);
    const float b1 = run<1, KIND, 8>(out, gbuf), b2 = run<2, KIND, 8>(out, gbuf), b3 = run<3, KIND, 8>(out, gbuf);
    printf("%-28s MFMA slots per instruction (1, 2, 3 in a bunch per 6 MFMAs): one wave/SIMD %.2f %.2f %.2f   two waves/SIMD %.2f %.2f %.2f\n", kKind[KIND],
           (a1 / base1 - 1) * 6, (a2 / base1 - 1) * 3, (a3 / base1 - 1) * 2, (b1 / base2 - 1) * 6, (b2 / base2 - 1) * 3, (b3 / base2 - 1) * 2);
}

int main()
{
    float *out, *gbuf;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&gbuf, (size_t)256 * 8 * 32768 + 65536);
    (void)hipMemset(gbuf, 0, (size_t)256 * 8 * 32768 + 65536);
    const float base1 = run<0, K_AND, 4>(out, gbuf), base2 = run<0, K_AND, 8>(out, gbuf);
    printf("MFMAs only: %.2f ns per MFMA (one wave per SIMD), %.2f ns (two waves per SIMD; 16 cycles at 2.4 GHz = 6.67 ns)\n", base1, base2);
    kind<K_GST>(out, gbuf, base1, base2); kind<K_GLD>(out, gbuf, base1, base2); kind<K_DSW64>(out, gbuf, base1, base2); kind<K_DSR128>(out, gbuf, base1, base2); kind<K_PKADD>(out, gbuf, base1, base2);
    kind<K_GST_CONTIG>(out, gbuf, base1, base2); kind<K_GST_DUP256>(out, gbuf, base1, base2); kind<K_GST_SEG32>(out, gbuf, base1, base2); kind<K_GST4_CONTIG>(out, gbuf, base1, base2);
    return 0;
}
