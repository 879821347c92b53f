// Does the ORDER in which a kernel's 1 KiB store instructions walk HBM matter on MI355X?  256 persistent workgroups x 8 waves;
// every wave owns 64-sample tiles (tile = group * 8 + wave, groups round-robin over the workgroups, as the fused kernels do)
// and writes, per tile, L blocks of 2 x 4 KiB (two 32-sample halves x four 1 KiB row tiles) -- the e5m2 output gradients of
// one net's layers -- either in tile-major records [tile][half][layer][row tile] (records of 2 x REC bytes, what round 1
// used) or in layer-major arrays [layer][tile][half][row tile].  Optional ALU work between the stores of a layer.
//   hipcc --offload-arch=gfx950 -O3 tools/store_layout_microbench.hip -o /tmp/slm && /tmp/slm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int L = 4, REC = 41088;      // bytes of a 32-sample record in the tile-major layout (two nets' blocks + scale record)

template <bool LAYER_MAJOR>
__global__ __launch_bounds__(512) void k(char* dst, long ntiles, int spin) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 v = {1u, 2u, 3u, (unsigned)lane};
    float x = lane * 0.001f;
    const long ngroups = (ntiles + 7) / 8;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long tile = g * 8 + wave;
        if (tile >= ntiles) continue;
        for (int l = 0; l < L; ++l) {
            for (int m = 0; m < 4; ++m) {
                for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
                for (int c = 0; c < 2; ++c) {
                    char* p = LAYER_MAJOR ? dst + ((long)l * ntiles * 2 + tile * 2 + c) * 4096 + m * 1024 + lane * 16
                                          : dst + (tile * 2 + c) * (long)REC + l * 4096 + m * 1024 + lane * 16;
                    v[0] = __builtin_bit_cast(unsigned, x);
                    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
                }
            }
        }
    }
}
int main() {
    const long ntiles = 196608;                       // 65 536 rays x 192 samples
    const size_t bytes = (size_t)ntiles * 2 * REC;
    char* d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int spin : {0, 64, 256}) {
        for (int lm = 0; lm < 2; ++lm) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(a);
                if (lm) hipLaunchKernelGGL(k<true>, dim3(256), dim3(512), 0, 0, d, ntiles, spin);
                else hipLaunchKernelGGL(k<false>, dim3(256), dim3(512), 0, 0, d, ntiles, spin);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (rep && ms < best) best = ms;
            }
            const double gb = (double)ntiles * L * 8192 / 1e9;
            printf("spin %3d  %-12s  %.2f GB in %.3f ms = %.2f TB/s\n", spin, lm ? "layer-major" : "tile-major", gb, best, gb / best);
        }
    }
    return 0;
}
