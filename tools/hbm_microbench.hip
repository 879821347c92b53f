// HBM streaming microbenchmark for MI355X (gfx950): what the scratch traffic of the backward kernels can reach.
//   hipcc --offload-arch=gfx950 -O3 tools/hbm_microbench.hip -o gpurun_out/hbm_microbench && gpurun_out/hbm_microbench
// Patterns mirror the kernels: every wave moves 1 KiB per instruction (16 B/lane), grid-stride persistent blocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __global__ void __launch_bounds__(512) k_write(f4* dst, size_t n) {
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) {
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}
template <bool NT> __global__ void __launch_bounds__(512) k_read(const f4* src, size_t n, float* sink) {
    f4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512) {
        f4 v = NT ? __builtin_nontemporal_load(src + i) : src[i];
        a += v;
    }
    if (a.x + a.y + a.z + a.w == 12345.678f) *sink = a.x;
}
// half of the blocks read one buffer while the other half writes another (B1 of chunk k+1 beside wgrad of chunk k)
__global__ void __launch_bounds__(512) k_mixed(const f4* src, f4* dst, size_t n, float* sink) {
    const unsigned half = gridDim.x / 2;
    if (blockIdx.x & 1) {
        f4 v = {1.f, 2.f, 3.f, 4.f};
        for (size_t i = (size_t)(blockIdx.x >> 1) * 512 + threadIdx.x; i < n; i += (size_t)half * 512) __builtin_nontemporal_store(v, dst + i);
    } else {
        f4 a = {0, 0, 0, 0};
        for (size_t i = (size_t)(blockIdx.x >> 1) * 512 + threadIdx.x; i < n; i += (size_t)half * 512) a += __builtin_nontemporal_load(src + i);
        if (a.x + a.y + a.z + a.w == 12345.678f) *sink = a.x;
    }
}

template <class F> static double time_ms(F f, int reps) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
    const size_t bytes = (size_t)6 << 30, n = bytes / 16;
    f4 *x, *y; float* sink;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(x, 0, bytes)); CK(hipMemset(y, 0, bytes));
    for (int blocks : {256, 512, 1024, 2048}) {
        double w0 = time_ms([&] { k_write<false><<<blocks, 512>>>(x, n); }, 5);
        double w1 = time_ms([&] { k_write<true><<<blocks, 512>>>(x, n); }, 5);
        double r0 = time_ms([&] { k_read<false><<<blocks, 512>>>(x, n, sink); }, 5);
        double r1 = time_ms([&] { k_read<true><<<blocks, 512>>>(x, n, sink); }, 5);
        double m = time_ms([&] { k_mixed<<<blocks, 512>>>(x, y, n, sink); }, 5);
        printf("blocks %4d  6 GiB  write %.2f TB/s  write-nt %.2f  read %.2f  read-nt %.2f  mixed(read 6 + write 6 GiB) %.2f TB/s total\n", blocks,
               bytes / w0 * 1e-9, bytes / w1 * 1e-9, bytes / r0 * 1e-9, bytes / r1 * 1e-9, 2.0 * bytes / m * 1e-9);
    }
    return 0;
}
