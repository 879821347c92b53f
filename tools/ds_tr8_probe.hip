// What ds_read_b64_tr_b8 (gfx950) delivers: LDS is filled with byte (a & 255) at every address a of a 4 KiB window plus a second
// pattern (a >> 8); every lane supplies an address and prints the 8 bytes it receives -- from which the lane -> (row, column) map of
// the hardware transpose follows (the guide documents the 16-bit form only).
//   hipcc --offload-arch=gfx950 -O3 tools/ds_tr8_probe.hip -o /tmp/tr8 && /tmp/tr8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void k(const int* addr, unsigned* out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = pattern ? (unsigned char)(i >> 4) : (unsigned char)(i & 255);
    __syncthreads();
    const int a = addr[threadIdx.x];
    i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(lds + a));
    out[threadIdx.x * 2] = (unsigned)v[0];
    out[threadIdx.x * 2 + 1] = (unsigned)v[1];
}
int main() {
    int h[64]; unsigned o[128];
    int* d; unsigned* dout;
    (void)hipMalloc(&d, sizeof(h)); (void)hipMalloc(&dout, sizeof(o));
    for (int test = 0; test < 2; ++test) {
        // lane l supplies 64 * l (test 0: rows of 64 bytes, every lane its own row) or 16 * l + (test pattern)
        for (int l = 0; l < 64; ++l) h[l] = test == 0 ? 64 * l : 16 * l;
        (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
        for (int pat = 0; pat < 2; ++pat) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, dout, pat);
            (void)hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
            printf("addresses %s, LDS byte = %s:\n", test == 0 ? "64*lane" : "16*lane", pat ? "addr>>4" : "addr&255");
            for (int l = 0; l < 64; ++l) {
                printf("  lane %2d:", l);
                for (int b = 0; b < 8; ++b) printf(" %3u", (o[2 * l + b / 4] >> (8 * (b % 4))) & 255u);
                printf("\n");
            }
        }
    }
    return 0;
}
