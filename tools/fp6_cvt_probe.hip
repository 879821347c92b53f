// What v_cvt_scalef32_pk32_{fp6,bf6}_bf16 and their inverses do on gfx950 (6-bit staging): 32 bf16 values of a lane
// <-> 6 dwords of e2m3 ("fp6") / e3m2 ("bf6") under the lane's power-of-two scale.  Prints the round trip of a set of values per
// scale and compares it with the emulation the rounding-ablation kernels use (nca_kernels_f32.hip, abl_fp6_group): grid, rounding
// to nearest even, saturation, subnormals, the scale's direction (pack divides, unpack multiplies), what the mantissa of the scale
// operand does.      hipcc --offload-arch=gfx950 -O2 tools/fp6_cvt_probe.hip -o /tmp/fp6_probe && /tmp/fp6_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
typedef __bf16 bf16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
template <bool BF6>
__global__ void rt(const float* in, const float* scale, float* out, unsigned* bits) {
    const int l = threadIdx.x;
    bf16x32 v;
    for (int i = 0; i < 32; ++i) v[i] = (__bf16)in[l * 32 + i];
    u32x6 p;
    if (BF6) p = __builtin_amdgcn_cvt_scalef32_pk32_bf6_bf16(v, scale[l]); else p = __builtin_amdgcn_cvt_scalef32_pk32_fp6_bf16(v, scale[l]);
    for (int i = 0; i < 6; ++i) bits[l * 6 + i] = p[i];
    bf16x32 w;
    if (BF6) w = __builtin_amdgcn_cvt_scalef32_pk32_bf16_bf6(p, scale[l]); else w = __builtin_amdgcn_cvt_scalef32_pk32_bf16_fp6(p, scale[l]);
    for (int i = 0; i < 32; ++i) out[l * 32 + i] = (float)w[i];
}
static float emu(float x, float s, bool bf6) {
    const int MB = bf6 ? 2 : 3, EMIN = bf6 ? -2 : 0; const float VMAX = bf6 ? 28.f : 7.5f;
    float v = fabsf(x) / s; int ev; frexpf(v, &ev); ev -= 1; if (v == 0.f || ev < EMIN) ev = EMIN;
    float step = ldexpf(1.f, ev - MB); float r = rintf(v / step) * step; if (r > VMAX) r = VMAX;
    return copysignf(r * s, x);
}
int main() {
    const int N = 64;
    float h[N * 32], sc[N], o[N * 32]; unsigned b[N * 6];
    const float vals[32] = {0.f, 1.f, -1.f, 0.3f, 0.0625f, 0.03f, 0.09375f, 0.1875f, 0.8f, 0.9375f, 1.0625f, 1.125f, 1.1875f, 1.9375f, 2.125f, 3.9f,
                            4.25f, 4.75f, 5.3f, 7.25f, 7.5f, 7.75f, 8.f, 9.f, 15.f, 26.f, 28.f, 29.f, 30.f, 31.f, 100.f, -1000.f};
    const float scales[8] = {1.f, 0.25f, 4.f, 1.5f, 1.99f, 0.0009765625f, 1024.f, 3.f};
    for (int l = 0; l < N; ++l) { sc[l] = scales[l % 8]; for (int i = 0; i < 32; ++i) h[l * 32 + i] = vals[i] * (l % 8 == 5 ? 0.0009765625f : (l % 8 == 6 ? 1024.f : 1.f)); }
    float *din, *dsc, *dout; unsigned* db;
    hipMalloc(&din, sizeof(h)); hipMalloc(&dsc, sizeof(sc)); hipMalloc(&dout, sizeof(o)); hipMalloc(&db, sizeof(b));
    hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice); hipMemcpy(dsc, sc, sizeof(sc), hipMemcpyHostToDevice);
    for (int bf6 = 0; bf6 < 2; ++bf6) {
        if (bf6) rt<true><<<1, N>>>(din, dsc, dout, db); else rt<false><<<1, N>>>(din, dsc, dout, db);
        hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(b, db, sizeof(b), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 8; ++l) {
            const float s_eff = ldexpf(1.f, ilogbf(sc[l]));          // hypothesis: only the scale's exponent counts
            printf("%s scale %g:", bf6 ? "bf6(e3m2)" : "fp6(e2m3)", sc[l]);
            for (int i = 0; i < 32; ++i) {
                const float x = (float)(__bf16)h[l * 32 + i], e = emu(x, s_eff, bf6);
                if (e != o[l * 32 + i]) ++bad;
                if (l < 2) printf(" %g>%g%s", x, o[l * 32 + i], e != o[l * 32 + i] ? "(!)" : "");
            }
            printf("\n");
        }
        printf("  mismatches against the emulation (scale = 2^floor(log2 scale)): %d of 256;  lane 0 dwords %08x %08x %08x %08x %08x %08x\n", bad, b[0], b[1], b[2], b[3], b[4], b[5]);
    }
    return 0;
}
