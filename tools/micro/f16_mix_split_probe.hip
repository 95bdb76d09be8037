// Does the five-instruction form of the scaled two-term split -- v_fma_mixlo_f16 / v_fma_mixhi_f16 (multiply by a power of two and round to fp16 in one
// instruction), v_fma_mix_f32 (v * sc - h1 with the fp16 term read in place), v_cvt_pk_f16_f32 -- return the bits of the reference sequence
// (fp32 multiply, round-to-nearest-even conversion, exact subtraction, conversion)?  conv_h2.hip: hx_split2s.  Answer on MI355X: 0 mismatches.
// hipcc --offload-arch=gfx950 -O3 -o f16_mix_split_probe.bin f16_mix_split_probe.hip
#include <hip/hip_runtime.h>
__device__ __forceinline__ void split2s(float v0, float v1, float sc, unsigned (&p)[2])
{
    unsigned h, r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(v0), "s"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(v1), "s"(sc));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(v0), "s"(sc), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(v1), "s"(sc), "v"(h));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    p[0] = h;
    p[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f2{__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1)}, h2));
}
__global__ void k(const float* x, unsigned* out, float sc, int n)
{
    int i = threadIdx.x;
    if (2 * i + 1 >= n) return;
    unsigned p[2];
    split2s(x[2 * i], x[2 * i + 1], __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc))), p);
    out[2 * i] = p[0]; out[2 * i + 1] = p[1];
}
#include <cstdio>
#include <cmath>
#include <cstring>
int main()
{
    const int n = 64; float hx[n]; 
    for (int i = 0; i < n; ++i) hx[i] = (i % 7 == 0 ? 0.f : (float)(sin(i * 1.3) * 1000.0 * pow(2.0, (i % 11) - 5)));
    hx[5] = 3.1415927e-3f; hx[6] = 1e-9f; hx[9] = 12345.678f;
    float *dx; unsigned* dout; hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    const float sc = 0.25f;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dx, dout, sc, n);
    unsigned ho[n]; hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n / 2; ++i) for (int e = 0; e < 2; ++e) {
        float v = hx[2 * i + e] * sc;
        _Float16 h1 = (_Float16)v; float r = v - (float)h1; _Float16 h2 = (_Float16)r;
        unsigned short a = (unsigned short)(ho[2 * i] >> (16 * e)), b = (unsigned short)(ho[2 * i + 1] >> (16 * e));
        unsigned short ea, eb; memcpy(&ea, &h1, 2); memcpy(&eb, &h2, 2);
        if (a != ea || b != eb) { ++bad; printf("x %g: got %04x %04x want %04x %04x\n", hx[2 * i + e], a, b, ea, eb); }
    }
    printf("mismatches: %d\n", bad);
    return 0;
}
