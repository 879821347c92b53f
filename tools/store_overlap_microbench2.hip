// Which ingredient of the fused kernels stops global stores from overlapping with the arithmetic (tools/store_overlap_microbench.hip:
// plain v_fma blocks overlap with stores almost perfectly)?  Same walk -- 8 waves per CU, per tile 4 "layers" x 4 "row tiles", two
// 1 KiB stores per row tile -- with the arithmetic of a row tile = 16 v_mfma_f32_32x32x16_bf16 and, step by step: (B) a workgroup
// barrier per layer, (D) a 32 KiB LDS-DMA per layer into a double buffer with the kernels' counted wait in front of the barrier,
// (L) the A operands read from that buffer (ds_read_b128).   hipcc --offload-arch=gfx950 -O3 ... -o /tmp/som2 && /tmp/som2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <bool STORE, bool MMA, bool BAR, bool DMA, bool LDSRD, int EPI = 0>
__global__ __launch_bounds__(512, 2) void k(char* dst, const char* wsrc, long ntiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc0 = (f32x16)(0.f), acc1 = (f32x16)(0.f);
    u32x4 a = {0x3f803f80u, 0x3f003f00u, 0x3e803e80u, (unsigned)lane}, b = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    int cur = 0;
    const long ngroups = (ntiles + 7) / 8;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long tile = g * 8 + wave;
        for (int l = 0; l < 4; ++l) {
            if (DMA)
                for (int c = wave; c < 32; c += 8)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + l * 32768 + c * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(smem + (cur ^ 1) * 32768 + c * 1024), 16, 0, 0);
            for (int m = 0; m < 4; ++m) {
                if (MMA) {
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        if (LDSRD) a = *reinterpret_cast<const u32x4*>(smem + cur * 32768 + (m * 8 + ks) * 1024 + lane * 16);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc1, 0, 0, 0);
                    }
                }
                // (E) an epilogue on the vector ALU that consumes the accumulators and produces the store data: EPI instructions per value
                unsigned e0 = 0u, e1 = 0u;
                if (EPI) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        unsigned u0 = __builtin_bit_cast(unsigned, acc0[i]), u1 = __builtin_bit_cast(unsigned, acc1[i]);
#pragma unroll
                        for (int r = 0; r < EPI; ++r) {
                            asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u0) : "v"(0x7fffffffu), "v"((unsigned)(r + i)));
                            asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u1) : "v"(0x7fffffffu), "v"((unsigned)(r + i)));
                        }
                        e0 ^= u0; e1 ^= u1;
                    }
                    acc0 = (f32x16)(__builtin_bit_cast(float, e0 & 0x3fffffffu)); acc1 = (f32x16)(__builtin_bit_cast(float, e1 & 0x3fffffffu));
                }
                if (STORE && tile < ntiles) {
                    char* p = dst + ((long)l * ntiles + tile) * 8192 + m * 2048 + lane * 16;
                    u32x4 v = {__builtin_bit_cast(unsigned, acc0[m]) ^ e0, __builtin_bit_cast(unsigned, acc1[m]) ^ e1, 3u, (unsigned)lane};
                    __builtin_nontemporal_store(v, (u32x4*)p);
                    __builtin_nontemporal_store(v, (u32x4*)(p + 1024));
                }
            }
            if (DMA) {
                if (STORE) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (BAR) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
            cur ^= 1;
        }
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) s += acc0[j] + acc1[j];
    if (s == 12345.678f) *sink = s;
}
template <bool STORE, bool MMA, bool BAR, bool DMA, bool LDSRD, int EPI = 0>
static float run(char* d, const char* w, long ntiles, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<STORE, MMA, BAR, DMA, LDSRD, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<STORE, MMA, BAR, DMA, LDSRD, EPI>), dim3(256), dim3(512), 65536, 0, d, w, ntiles, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    return best;
}
#define ROW(name, B, D, L) printf("%-34s mfma alone %.2f ms | with stores %.2f ms   (stores alone %.2f: max %.2f, sum %.2f)\n", name, \
    run<false, true, B, D, L>(d, w, ntiles, sink), run<true, true, B, D, L>(d, w, ntiles, sink), st, \
    st > run<false, true, B, D, L>(d, w, ntiles, sink) ? st : run<false, true, B, D, L>(d, w, ntiles, sink), st + run<false, true, B, D, L>(d, w, ntiles, sink));
int main() {
    const long ntiles = 196608;
    char *d, *w; float* sink;
    if (hipMalloc(&d, (size_t)ntiles * 4 * 8192) != hipSuccess) return 1;
    hipMalloc(&w, 4 * 32768); hipMemset(w, 0, 4 * 32768); hipMalloc(&sink, 4);
    const float st = run<true, false, false, false, false>(d, w, ntiles, sink);
    ROW("16 MFMA per row tile", false, false, false)
    ROW("+ barrier per layer", true, false, false)
    ROW("+ 32 KiB LDS-DMA per layer", true, true, false)
    ROW("+ A operands from LDS", true, true, true)
#define ROWE(name, E) { const float c0 = run<false, true, true, true, true, E>(d, w, ntiles, sink), c1 = run<true, true, true, true, true, E>(d, w, ntiles, sink); \
    printf("%-34s compute alone %.2f ms | with stores %.2f ms   (stores alone %.2f: max %.2f, sum %.2f)\n", name, c0, c1, st, st > c0 ? st : c0, st + c0); }
    ROWE("+ epilogue, 1 VALU per value", 1)
    ROWE("+ epilogue, 2 VALU per value", 2)
    ROWE("+ epilogue, 3 VALU per value", 3)
    ROWE("+ epilogue, 4 VALU per value", 4)
    return 0;
}
