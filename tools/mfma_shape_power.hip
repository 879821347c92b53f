// Which MFMA shape does the chip run faster AT ITS POWER LIMIT inside an instruction mix like the storing forward's?  Every CU runs two
// waves per SIMD of  [1 ds_read_b128 (A fragment, shared by two column tiles) ; 2 x v_mfma_f32_32x32x16_bf16  |  4 x v_mfma_f32_16x16x32_bf16
// (the same 32 x 64 x 16 product) ; NV packed-16-bit vector instructions ; one 1 KiB non-temporal store every 8 steps]  for ~0.2 s and the
// WALL time per step is compared: cycles per FLOP are equal for the two shapes (guide), so a difference is the clock the power cap allows.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_power.hip -o /tmp/msp && /tmp/msp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int NV, bool STORES>
__global__ __launch_bounds__(512, 2) void k(int iters, char* out, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 32768 / 4; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (i & 255);
    __syncthreads();
    f32x16 a32[2] = {(f32x16)(0.f), (f32x16)(0.f)};
    f32x4 a16[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a16[i] = (f32x4)(0.f);
    u32x4 b0 = {0x3f003f00u + lane, 0x3f003f00u, 0x3f003f00u, 0x3f003f00u}, b1 = {0x3e003e00u, 0x3e003e00u + lane, 0x3e003e00u, 0x3e003e00u};
    unsigned x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 3 + j;
    const char* base = lds + lane * 16;
    char* dst = out + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 65536 + lane * 16;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(base + u * 1024);
            if (SHAPE == 32) {
                a32[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b0), a32[0], 0, 0, 0);
                a32[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b1), a32[1], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    a16[(u & 1) * 4 + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, q & 1 ? b1 : b0), a16[(u & 1) * 4 + q], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_pk_max_i16 %0, %0, 0" : "+v"(x[(u * NV + j) % 16]));
            if (STORES && (u & 7) == 7) {
                const u32x4 v = {x[0], x[1], x[2], x[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(dst + ((i * 4 + (u >> 3)) & 63) * 1024));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += __builtin_bit_cast(float, x[j]) + a32[0][j] + a32[1][j];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a16[i][0] + a16[i][3];
    if (s == 12345.678f) *sink = s;
}

template <int SHAPE, int NV, bool STORES>
static double run(char* out, float* sink) {
    const int blocks = 256, iters = 6000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SHAPE, NV, STORES>), dim3(blocks), dim3(512), 32768, 0, 300, out, sink);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<SHAPE, NV, STORES>), dim3(blocks), dim3(512), 32768, 0, iters, out, sink);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (iters * 32.0);          // ns per step (per wave; two waves per SIMD run concurrently)
}

int main() {
    char* out; float* sink;
    (void)hipMalloc(&out, (size_t)256 * 8 * 65536); (void)hipMalloc(&sink, 4);
    printf("ns per step of [ds_read_b128 + (2 x 32x32x16 | 4 x 16x16x32) + NV x v_pk_max_i16 (+ 1 KiB store per 8 steps)], 256 CUs x 8 waves, ~0.2 s runs\n");
    printf("   NV  stores |  32x32x16   16x16x32   ratio\n");
#define ROW(NV, ST) { const double a = run<32, NV, ST>(out, sink), b = run<16, NV, ST>(out, sink); printf("   %2d  %-6s | %9.2f  %9.2f   %.3f\n", NV, ST ? "yes" : "no", a, b, a / b); fflush(stdout); }
    ROW(0, false) ROW(4, false) ROW(10, false) ROW(10, true) ROW(16, true)
    return 0;
}
