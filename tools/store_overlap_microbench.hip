// Do a wave's global stores overlap with its own arithmetic on MI355X?  256 persistent workgroups x W waves; every wave alternates
// a block of independent v_fma_f32 (SPIN x 8 accumulators: a known number of VALU issue cycles) with two 1 KiB stores, walking
// its own tiles (layer-major or tile-major addresses are equivalent: tools/store_layout_microbench.hip).  Reported: the time of
// the arithmetic alone, of the stores alone, and of both -- sum or maximum?  Variants: non-temporal / plain stores, 4 or 8 waves
// per CU, and the stores of a block issued from the MIDDLE of the arithmetic instead of behind it.
//   hipcc --offload-arch=gfx950 -O3 tools/store_overlap_microbench.hip -o /tmp/som && /tmp/som
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT, bool STORE, bool ALU, bool MID>
__global__ __launch_bounds__(512) void k(char* dst, long ntiles, int spin, float* sink) {
    const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    float x[8];
    for (int j = 0; j < 8; ++j) x[j] = lane * 0.001f + j;
    const long ngroups = (ntiles + nw - 1) / nw;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long tile = g * nw + wave;
        if (tile >= ntiles) continue;
        for (int l = 0; l < 4; ++l)
            for (int m = 0; m < 4; ++m) {
                char* p = dst + ((long)l * ntiles + tile) * 8192 + m * 2048 + lane * 16;
                if (ALU) for (int i = 0; i < spin / 2; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(1.0001f), "v"(0.5f));
                if (STORE && MID) {
                    v[0] = __builtin_bit_cast(unsigned, x[0]);
                    if (NT) { __builtin_nontemporal_store(v, (u32x4*)p); __builtin_nontemporal_store(v, (u32x4*)(p + 1024)); }
                    else { *(u32x4*)p = v; *(u32x4*)(p + 1024) = v; }
                }
                if (ALU) for (int i = 0; i < spin - spin / 2; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(1.0001f), "v"(0.5f));
                if (STORE && !MID) {
                    v[0] = __builtin_bit_cast(unsigned, x[0]);
                    if (NT) { __builtin_nontemporal_store(v, (u32x4*)p); __builtin_nontemporal_store(v, (u32x4*)(p + 1024)); }
                    else { *(u32x4*)p = v; *(u32x4*)(p + 1024) = v; }
                }
            }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) s += x[j];
    if (s == 12345.678f) *sink = s;
}
template <bool NT, bool STORE, bool ALU, bool MID>
static float run(char* d, long ntiles, int spin, int threads, float* sink) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k<NT, STORE, ALU, MID>), dim3(256), dim3(threads), 0, 0, d, ntiles, spin, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    return best;
}
int main() {
    const long ntiles = 196608;
    char* d; float* sink;
    if (hipMalloc(&d, (size_t)ntiles * 4 * 8192) != hipSuccess) return 1;
    hipMalloc(&sink, 4);
    const double gb = (double)ntiles * 4 * 8192 / 1e9;
    for (int threads : {512, 256})
        for (int spin : {16, 48, 96}) {
            const float alu = run<true, false, true, false>(d, ntiles, spin, threads, sink);
            const float st_nt = run<true, true, false, false>(d, ntiles, spin, threads, sink), st_pl = run<false, true, false, false>(d, ntiles, spin, threads, sink);
            const float both_nt = run<true, true, true, false>(d, ntiles, spin, threads, sink), both_pl = run<false, true, true, false>(d, ntiles, spin, threads, sink);
            const float mid_nt = run<true, true, true, true>(d, ntiles, spin, threads, sink);
            printf("%d waves/CU, %3d x 8 fma per 2 KiB: alu %.2f ms | stores %.2f nt %.2f plain (%.1f TB/s) | both %.2f nt %.2f plain | stores from the middle %.2f nt  [max %.2f, sum %.2f]\n",
                   threads / 64, spin, alu, st_nt, st_pl, gb / st_nt, both_nt, both_pl, mid_nt, alu > st_nt ? alu : st_nt, alu + st_nt);
        }
    return 0;
}
