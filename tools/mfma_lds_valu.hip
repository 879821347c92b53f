// Does vector-ALU work still hide behind the matrix pipe when the MFMA's A operand arrives from LDS through a register ring, as in the
// fused kernels?  Each wave runs [ds_read_b128 of the fragment RING-1 steps ahead; s_waitcnt for the current fragment; 1 MFMA on one of
// two accumulators; N independent VALU] per step, one or two waves per SIMD, every CU busy; the VALU instruction is v_pk_max_i16 (4.4
// cycles of SIMD time) or v_fma_f32.  Compare with tools/valu_mfma_samewave.hip (operands in registers).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_valu.hip -o /tmp/mlv && /tmp/mlv
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int N, int RING, int KIND, bool LDSA>
__global__ void k(int iters, unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + i;
    __syncthreads();
    f32x16 acc0 = (f32x16)(0.f), acc1 = (f32x16)(0.f);
    u32x4 b = {0x3f003f00u, 0x3f003f00u, 0x3f003f00u, 0x3f003f00u};
    unsigned x[16];
    for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 3 + j;
    const float c1 = 1.0001f, c2 = 0.5f;
    const char* base = lds + lane * 16;
    u32x4 A[RING];
#pragma unroll
    for (int g = 0; g < RING - 1; ++g) A[g] = *reinterpret_cast<const u32x4*>(base + g * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {              // 32 fragments of 1 KiB: one weight image
            if (LDSA) A[(u + RING - 1) % RING] = *reinterpret_cast<const u32x4*>(base + ((u + RING - 1) % 32) * 1024);
            const u32x4 a = LDSA ? A[u % RING] : b;
            if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                unsigned& r = x[(u * N + j) % 16];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r) : "v"(c1), "v"(c2));
                else asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(r));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += __builtin_bit_cast(float, x[j]) + acc0[j] + acc1[j];
    if (s == 12345.678f) *sink = s;
}

template <int N, int RING, int KIND, bool LDSA>
static double run(int threads, unsigned long long* dcyc, float* sink) {
    const int blocks = 256, waves = blocks * threads / 64, iters = 200;
    hipLaunchKernelGGL((k<N, RING, KIND, LDSA>), dim3(blocks), dim3(threads), 32768 + 4096, 0, 20, dcyc, sink);
    hipLaunchKernelGGL((k<N, RING, KIND, LDSA>), dim3(blocks), dim3(threads), 32768 + 4096, 0, iters, dcyc, sink);
    std::vector<unsigned long long> h(waves);
    (void)hipMemcpy(h.data(), dcyc, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[waves / 2] / (iters * 32.0);
}

int main() {
    unsigned long long* dcyc; float* sink;
    (void)hipMalloc(&dcyc, 256 * 8 * sizeof(unsigned long long));
    (void)hipMalloc(&sink, 4);
    printf("cycles per step of [ds_read_b128 ahead + wait + 1 MFMA + N x VALU], median wave\n");
    printf("                              |  A from LDS, ring 4  |  A from LDS, ring 8  |  A in registers\n");
    printf("   VALU          N            |   1 w/SIMD  2 w/SIMD |   1 w/SIMD  2 w/SIMD |   1 w/SIMD  2 w/SIMD\n");
#define ROW(KIND, NAME, N) printf("   %-12s %2d            |  %9.1f %9.1f |  %9.1f %9.1f |  %9.1f %9.1f\n", NAME, N, \
        run<N, 4, KIND, true>(256, dcyc, sink), run<N, 4, KIND, true>(512, dcyc, sink), run<N, 8, KIND, true>(256, dcyc, sink), run<N, 8, KIND, true>(512, dcyc, sink), \
        run<N, 4, KIND, false>(256, dcyc, sink), run<N, 4, KIND, false>(512, dcyc, sink));
    ROW(0, "v_fma_f32", 0) ROW(0, "v_fma_f32", 3) ROW(0, "v_fma_f32", 5) ROW(0, "v_fma_f32", 8)
    ROW(1, "v_pk_max_i16", 3) ROW(1, "v_pk_max_i16", 5) ROW(1, "v_pk_max_i16", 8)
    return 0;
}
