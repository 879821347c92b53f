#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* out, float scale, int ovfl) {
    int i = threadIdx.x;
    if (ovfl) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1");     // MODE.FP16_OVFL: saturate instead of inf / NaN
    float a = in[2 * i], b = in[2 * i + 1];
    unsigned v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    out[4 * i] = v;
    unsigned w = 0;
    w = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, w, false);
    out[4 * i + 1] = w;
    s16x2 o = {0, 0};
    o = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(o, a, b, scale, false);
    out[4 * i + 2] = __builtin_bit_cast(unsigned, o);
    s16x2 p = {0, 0};
    p = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(p, a, b, scale, false);
    out[4 * i + 3] = __builtin_bit_cast(unsigned, p);
}
__global__ void m(const long* a, const long* b, float* c) {
    f32x16 z;
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    z = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(a[threadIdx.x], b[threadIdx.x], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a[threadIdx.x], b[threadIdx.x], z, 0, 0, 0);
    z = __builtin_amdgcn_mfma_f32_32x32x16_fp8_bf8(a[threadIdx.x], b[threadIdx.x], z, 0, 0, 0);
    for (int i = 0; i < 16; ++i) c[threadIdx.x * 16 + i] = z[i];
}
int main() {
    const int N = 64;
    float h[2 * N];
    float vals[] = {0.f, 1.f, -1.f, 0.3f, 448.f, 449.f, 480.f, 1000.f, 1e6f, 57344.f, 60000.f, 1e-3f, 2e-3f, 1.5e-5f, 7e-6f, 3.2e-2f, 0.0155f, 0.0156f, 1.0625f, 1.125f, 1.1875f, 17.f, 240.f, INFINITY, NAN, -500.f};
    int nv = sizeof(vals) / 4;
    for (int i = 0; i < 2 * N; ++i) h[i] = vals[i % nv];
    float* din; unsigned* dout;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dout, N * 16);
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    for (float sc : {1.f, -1.f, 0.25f, 4.f, 3.f}) {
        const int ovfl = sc < 0.f;
        if (ovfl) sc = 1.f;
        k<<<1, N>>>(din, dout, sc, ovfl);
        unsigned ho[4 * N];
        hipMemcpy(ho, dout, sizeof(ho), hipMemcpyDeviceToHost);
        printf("scale %g fp16_ovfl %d\n", sc, ovfl);
        for (int i = 0; i < nv; ++i) {
            int t = i / 2, e = i % 2;
            printf("  %12g  fp8 %02x  bf8 %02x  sfp8 %02x  sbf8 %02x\n", vals[i], (ho[4 * t] >> (8 * e)) & 0xff, (ho[4 * t + 1] >> (8 * e)) & 0xff, (ho[4 * t + 2] >> (8 * e)) & 0xff, (ho[4 * t + 3] >> (8 * e)) & 0xff);
        }
    }
    return 0;
}
