// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 (what nca_wgrad_bf16's 8-bit path relies on): which (lane, byte) of the A / B
// operands is which (row | column, k), how the e8m0 block scales apply, the format codes (cbsz / blgp) and the accumulator layout.
//   hipcc --offload-arch=gfx950 -O2 tools/mx_mfma_probe.hip -o /tmp/mxp && /tmp/mxp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int CBSZ, int BLGP>
__global__ void k(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* c) {
    f32x16 z;
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    z = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], z, CBSZ, BLGP, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
    for (int i = 0; i < 16; ++i) c[threadIdx.x * 16 + i] = z[i];
}
static uint8_t a8[64][32], b8[64][32];
static int sa[64], sb[64];
static float hc[1024];
static void *da, *db, *dsa, *dsb; static float* dc;
template <int CBSZ, int BLGP> static void run() {
    hipMemcpy(da, a8, 2048, hipMemcpyHostToDevice); hipMemcpy(db, b8, 2048, hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
    k<CBSZ, BLGP><<<1, 64>>>((const i32x8*)da, (const i32x8*)db, (const int*)dsa, (const int*)dsb, dc);
    hipMemcpy(hc, dc, 4096, hipMemcpyDeviceToHost);
}
static void nz(const char* what) {
    int n = 0; printf("%s: nonzero outputs:", what);
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if (hc[l * 16 + i] != 0.f) { if (n < 6) printf(" (lane %d reg %d)=%g", l, i, hc[l * 16 + i]); ++n; }
    printf("  [%d in all]\n", n);
}
int main() {
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc((void**)&dc, 4096);
    for (int l = 0; l < 64; ++l) sa[l] = sb[l] = 127;
    // 1. all ones, both as e4m3 (0x38), formats 0/0
    memset(a8, 0x38, sizeof a8); memset(b8, 0x38, sizeof b8);
    run<0, 0>(); printf("1. ones x ones, e4m3 x e4m3, scales 127: out[0] = %g (64 expected)\n", hc[0]);
    // 1b. A as e5m2 ones (0x3c) with cbsz = 1
    memset(a8, 0x3c, sizeof a8);
    run<1, 0>(); printf("1b. e5m2 ones (cbsz 1) x e4m3 ones: out[0] = %g (64 expected)\n", hc[0]);
    // 2. one A element
    for (int L : {0, 5, 37}) for (int J : {0, 9, 31}) {
        memset(a8, 0, sizeof a8); a8[L][J] = 0x3c; memset(b8, 0x38, sizeof b8);
        run<1, 0>(); char w[64]; snprintf(w, 64, "2. A lane %d byte %d", L, J); nz(w);
    }
    // 3. one B element
    for (int L : {0, 5, 37}) {
        memset(a8, 0x3c, sizeof a8); memset(b8, 0, sizeof b8); b8[L][3] = 0x38;
        run<1, 0>(); char w[64]; snprintf(w, 64, "3. B lane %d byte 3", L); nz(w);
    }
    // 4. k pairing: A (lane 0, byte J) against every B (lane 0 | 32, byte J')
    for (int LA : {0, 32}) for (int J : {0, 5, 17, 31}) {
        printf("4. A lane %d byte %d pairs with B", LA, J);
        for (int LB : {0, 32}) for (int Jp = 0; Jp < 32; ++Jp) {
            memset(a8, 0, sizeof a8); memset(b8, 0, sizeof b8); a8[LA][J] = 0x3c; b8[LB][Jp] = 0x38;
            run<1, 0>();
            bool any = false; for (int q = 0; q < 1024; ++q) any |= hc[q] != 0.f;
            if (any) printf(" (lane %d byte %d)", LB, Jp);
        }
        printf("\n");
    }
    // 5. scales
    memset(a8, 0x3c, sizeof a8); memset(b8, 0x38, sizeof b8);
    for (int l = 0; l < 64; ++l) { sa[l] = l < 32 ? 128 : 127; sb[l] = 127; }
    run<1, 0>(); printf("5. scale_a 128 on lanes 0-31, 127 on 32-63: out[0] = %g (96 if the scale is per lane half / K block)\n", hc[0]);
    for (int l = 0; l < 64; ++l) { sa[l] = 127; sb[l] = (l % 32 == 0) ? 129 : 127; }
    run<1, 0>(); printf("5b. scale_b 129 on lanes 0 and 32 only: out(lane 0, reg 0) = %g, out(lane 1, reg 0) = %g (256 and 64 if per column)\n", hc[0], hc[16]);
    for (int l = 0; l < 64; ++l) { sa[l] = 127 | (125 << 8); sb[l] = 127; }
    run<1, 0>(); printf("5c. scale_a bytes {127, 125, ..} with opsel 0: out[0] = %g (64: byte 0 is the one used)\n", hc[0]);
    // 6. values: one A element x (code) times B ones, and one B element x A ones
    for (int l = 0; l < 64; ++l) sa[l] = sb[l] = 127;
    for (int code : {0x42, 0xc4, 0x34, 0x3e, 0x7b}) {
        memset(a8, 0, sizeof a8); a8[0][0] = (uint8_t)code; memset(b8, 0x38, sizeof b8);
        run<1, 0>(); printf("6. A e5m2 code 0x%02x x 1: %g\n", code, hc[0]);
    }
    for (int code : {0x3c, 0xc4, 0x20, 0x7e, 0x01}) {
        memset(b8, 0, sizeof b8); b8[0][0] = (uint8_t)code; memset(a8, 0x3c, sizeof a8);
        run<1, 0>(); printf("6. B e4m3 code 0x%02x x 1: %g\n", code, hc[0]);
    }
    // 7. larger scales
    memset(a8, 0x3c, sizeof a8); memset(b8, 0x38, sizeof b8);
    for (int e : {124, 129, 100, 160}) {
        for (int l = 0; l < 64; ++l) { sa[l] = e; sb[l] = 127; }
        run<1, 0>(); printf("7. scale_a %d: out[0] = %g (64 * 2^%d)\n", e, hc[0], e - 127);
    }
    for (int l = 0; l < 64; ++l) { sa[l] = 127 + (l / 32 ? -3 : 2); sb[l] = 127 + (l / 32 ? 1 : -1); }
    run<1, 0>(); printf("7b. first-probe scales: out[0] = %g (32 * 2 + 32 / 4 = 72)\n", hc[0]);
    return 0;
}
