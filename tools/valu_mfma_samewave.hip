// How much vector-ALU work hides behind the matrix pipe on gfx950, measured in shader cycles (s_memtime) per MFMA:
// every wave runs a fixed stream of  v_mfma_f32_32x32x16_bf16 + N independent VALU instructions  (inline assembly, so the
// order is exactly as written), with ONE wave per SIMD (256-thread workgroups) or TWO (512-thread workgroups: what the fused
// kernels run), one workgroup per CU, every CU busy.  N sweeps 0..16; the VALU instruction is v_fma_f32 (4-cycle issue class),
// v_cvt_pk_bf16_f32, or v_cndmask_b32 / v_and_b32 (what the epilogues are made of).  Also: the VALU stream alone.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_mfma_samewave.hip -o /tmp/vms && /tmp/vms
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ACCV: the MFMA accumulators in architectural VGPRs ("v") instead of accumulation VGPRs ("a")
template <int N, int KIND, bool MFMA, bool ACCV = false>
__global__ void k(int iters, unsigned long long* cyc, float* sink) {
    f32x16 acc0 = (f32x16)(0.f), acc1 = (f32x16)(0.f);
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * threadIdx.x); b[j] = (__bf16)(0.5f); }
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 0.001f + j;
    const float c1 = 1.0001f, c2 = 0.5f;
    const unsigned m = 0xffff0000u;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MFMA) {
                if (ACCV) {
                    if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc1) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b));
                } else {
                    if (u & 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc1) : "v"(a), "v"(b));
                    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc0) : "v"(a), "v"(b));
                }
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
                float& r = x[(u * N + j) % 16];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2));
                else if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r) : "v"(c1));
                else if (KIND == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(m));
                else if (KIND == 4) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(r));
                else if (KIND == 5) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(r) : "s"(0x00010001u));
                else if (KIND == 6) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(r) : "v"(c1));
                else if (KIND == 7) asm volatile("v_cvt_scalef32_pk_fp8_bf16 %0, %1, %2" : "+v"(r) : "v"(c1), "s"(0.25f));
                else if (KIND == 8) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(r) : "v"(c1));
                else if (KIND == 9) asm volatile("v_cvt_scalef32_pk_bf8_bf16 %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2));
                else asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c1));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x % 64 == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += x[j] + acc0[j] + acc1[j];
    if (s == 12345.678f) *sink = s;
}

template <int N, int KIND, bool MFMA, bool ACCV = false>
static double run(int threads, int iters, unsigned long long* dcyc, float* sink) {
    const int blocks = 256, waves = blocks * threads / 64;
    hipLaunchKernelGGL((k<N, KIND, MFMA, ACCV>), dim3(blocks), dim3(threads), 0, 0, 200, dcyc, sink);       // warm-up
    hipLaunchKernelGGL((k<N, KIND, MFMA, ACCV>), dim3(blocks), dim3(threads), 0, 0, iters, dcyc, sink);
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), dcyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[waves / 2] / (iters * 8.0);             // median wave: cycles per (MFMA + N VALU) group
}

template <int KIND>
static void sweep(const char* name, unsigned long long* dcyc, float* sink) {
    const int it = 4000;
    printf("%s: cycles per group of [1 MFMA + N x %s], median wave;  (VALU alone)\n", name, name);
    printf("   N   1 wave/SIMD   2 waves/SIMD |  alone 1w   alone 2w\n");
#define ROW(N) printf("  %2d   %9.1f   %11.1f  | %8.1f   %8.1f\n", N, run<N, KIND, true>(256, it, dcyc, sink), run<N, KIND, true>(512, it, dcyc, sink), \
                      N ? run<N, KIND, false>(256, it, dcyc, sink) : 0.0, N ? run<N, KIND, false>(512, it, dcyc, sink) : 0.0);
    ROW(0) ROW(2) ROW(4) ROW(6) ROW(8) ROW(10) ROW(12) ROW(16)
#undef ROW
}

int main() {
    unsigned long long* dcyc; float* sink;
    (void)hipMalloc(&dcyc, 256 * 8 * sizeof(unsigned long long));
    (void)hipMalloc(&sink, 4);
    {   // the same with the accumulators in architectural VGPRs (what the compiler picks when it has the registers)
        const int it = 4000;
        printf("v_fma_f32, accumulators in a / in v registers: cycles per group of [1 MFMA + N x v_fma_f32]\n   N   a: 1 w/SIMD  2 w/SIMD |  v: 1 w/SIMD  2 w/SIMD\n");
#define ROWV(N) printf("  %2d   %9.1f  %9.1f  |  %9.1f  %9.1f\n", N, run<N, 0, true, false>(256, it, dcyc, sink), run<N, 0, true, false>(512, it, dcyc, sink), \
                       run<N, 0, true, true>(256, it, dcyc, sink), run<N, 0, true, true>(512, it, dcyc, sink));
        ROWV(0) ROWV(2) ROWV(4) ROWV(6) ROWV(8) ROWV(12)
#undef ROWV
    }
    sweep<0>("v_fma_f32", dcyc, sink);
    sweep<1>("v_cvt_pk_bf16_f32", dcyc, sink);
    sweep<2>("v_and_b32", dcyc, sink);
    sweep<3>("v_cndmask_b32", dcyc, sink);
    sweep<4>("v_pk_max_i16", dcyc, sink);
    sweep<5>("v_pk_min_u16", dcyc, sink);
    sweep<6>("v_lshl_or_b32", dcyc, sink);
    sweep<7>("v_cvt_scalef32_pk_fp8_bf16", dcyc, sink);
    sweep<8>("v_pk_mul_lo_u16", dcyc, sink);
    sweep<9>("v_cvt_scalef32_pk_bf8_bf16", dcyc, sink);
    return 0;
}
