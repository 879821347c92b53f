// Does vector-ALU work of one wave overlap with MFMA work of ANOTHER wave on the same SIMD (gfx950)?
// 8 waves per workgroup, one workgroup per CU: waves 0-3 (one per SIMD) run a chain of MFMAs, waves 4-7 (their SIMD
// mates) a chain of v_fma_f32.  Timed: MFMA waves alone, VALU waves alone, both.  Also the same mix inside ONE wave.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_mfma_overlap.hip -o /tmp/overlap && /tmp/overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>   // 0: 32x32x16 bf16 (8 passes), 1: 32x32x2 f32 (16 passes)
__device__ __forceinline__ void mfma_loop(int n, float* out) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = (f32x16)(0.f);
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + threadIdx.x); b[j] = (__bf16)(0.5f); }
    const float fa = 1.0f + threadIdx.x, fb = 0.5f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
            else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[c], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int j = 0; j < 16; ++j) s += acc[c][j];
    if (s == 12345.678f) *out = s;
}
__device__ __forceinline__ void valu_loop(int n, float* out) {
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f);
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += x[j];
    if (s == 12345.678f) *out = s;
}
template <int KIND>
__global__ __launch_bounds__(512) void k(int n_mfma, int n_valu, int mode, float* out) {
    const int wave = threadIdx.x >> 6;
    if (mode == 3) {                    // one wave per SIMD does both, interleaved by the compiler / hardware
        if (wave < 4) {
            f32x16 acc[4];
            for (int c = 0; c < 4; ++c) acc[c] = (f32x16)(0.f);
            bf16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + threadIdx.x); b[j] = (__bf16)(0.5f); }
            const float fa = 1.0f + threadIdx.x, fb = 0.5f;
            float x[8];
            for (int j = 0; j < 8; ++j) x[j] = threadIdx.x * 0.001f + j;
            const int per = n_valu / n_mfma;          // v_fma groups of 8 per 4 MFMAs
            for (int i = 0; i < n_mfma; ++i) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
                    else acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[c], 0, 0, 0);
                    for (int r = 0; r < per; ++r) {
#pragma unroll
                        for (int j = 2 * c; j < 2 * c + 2; ++j) x[j] = __builtin_fmaf(x[j], 1.0001f, 0.5f);
                    }
                }
            }
            float s = 0.f;
            for (int c = 0; c < 4; ++c) for (int j = 0; j < 16; ++j) s += acc[c][j];
            for (int j = 0; j < 8; ++j) s += x[j];
            if (s == 12345.678f) *out = s;
        }
        return;
    }
    if (wave < 4) { if (mode & 1) mfma_loop<KIND>(n_mfma, out); }
    else { if (mode & 2) valu_loop(n_valu, out); }
}
template <int KIND>
static float run(int nm, int nv, int mode, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, nm, nv, mode, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, nm, nv, mode, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float* out; hipMalloc(&out, 4);
    const int nm = 20000;                         // x4 MFMAs per wave
    for (int kind = 0; kind < 2; ++kind) {
        const int cyc = kind == 0 ? 32 : 64;      // cycles per MFMA
        const int nv = nm * 4 * cyc / 4 / 8;      // x8 v_fma per iteration, 4 cycles each: the same nominal time as the MFMA chain
        float a = kind == 0 ? run<0>(nm, nv, 1, out) : run<1>(nm, nv, 1, out);
        float b = kind == 0 ? run<0>(nm, nv, 2, out) : run<1>(nm, nv, 2, out);
        float c = kind == 0 ? run<0>(nm, nv, 3 - 0, out) : run<1>(nm, nv, 3, out);
        float both = 0;
        {   // mode 1|2 = both kinds of waves
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, nm, nv, 1 | 2 | 4, out); else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, nm, nv, 1 | 2 | 4, out);
            hipEventRecord(e0);
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, nm, nv, 1 | 2 | 4, out); else hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, nm, nv, 1 | 2 | 4, out);
            hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&both, e0, e1);
        }
        printf("%s: MFMA waves alone %.3f ms | VALU waves alone %.3f ms | both (different waves, same SIMD) %.3f ms | one wave interleaved %.3f ms\n",
               kind == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_32x32x2_f32  ", a, b, both, c);
    }
    return 0;
}
