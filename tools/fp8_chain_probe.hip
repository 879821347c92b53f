// Groundwork for fp8 contractions in the layer chain: does the accumulator layout of one layer feed the NEXT layer's
// v_mfma_scale_f32_32x32x64_f8f6f4 as directly as it feeds v_mfma_f32_32x32x16_bf16 today?  One wave, one 32-sample column tile,
// width 128: the previous layer's outputs sit in 4 x 16 accumulator registers per lane (lane = (sample n, half h), register i of row
// tile m = feature 32m + 8(i>>2) + 4h + (i&3)); each row tile's 16 values become 16 e4m3 bytes (byte i = register i -- the 16 bytes
// the storing forward already writes per lane), and k-step ks of the next layer takes {bytes of row tile 2ks, bytes of row tile 2ks+1}
// as its 32-byte B fragment as it stands.  The weight image holds, for lane (row r, half h) and byte b, W[r][feature of (b, h)]:
// the contraction pairs A and B bytes of equal (h, b), so the feature permutation lives in the image alone (as in the bf16 path).
// Exact integers / quarter values throughout: the result must equal the CPU product bit for bit.
//   hipcc --offload-arch=gfx950 -O2 tools/fp8_chain_probe.hip -o /tmp/f8c && /tmp/f8c
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__host__ __device__ inline int feat_of(int m, int i, int h) { return 32 * m + 8 * (i >> 2) + 4 * h + (i & 3); }

__global__ void k(const float* Hin /* [128][32] */, const uint8_t* Aimg /* [4 mo][2 ks][64 lanes][32 B] */, float* Y /* [128][32] */) {
    const int lane = threadIdx.x, n = lane & 31, h = lane >> 5;
    // the previous layer's accumulators of this lane, then their e4m3 bytes (4 dwords per row tile)
    int q[4][4];
    for (int m = 0; m < 4; ++m)
        for (int w = 0; w < 4; ++w) {
            int v = 0;
            v = __builtin_amdgcn_cvt_pk_fp8_f32(Hin[feat_of(m, 4 * w + 0, h) * 32 + n], Hin[feat_of(m, 4 * w + 1, h) * 32 + n], v, false);
            v = __builtin_amdgcn_cvt_pk_fp8_f32(Hin[feat_of(m, 4 * w + 2, h) * 32 + n], Hin[feat_of(m, 4 * w + 3, h) * 32 + n], v, true);
            q[m][w] = v;
        }
    for (int mo = 0; mo < 4; ++mo) {
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int ks = 0; ks < 2; ++ks) {
            const i32x8 B = {q[2 * ks][0], q[2 * ks][1], q[2 * ks][2], q[2 * ks][3], q[2 * ks + 1][0], q[2 * ks + 1][1], q[2 * ks + 1][2], q[2 * ks + 1][3]};
            const i32x8 A = *reinterpret_cast<const i32x8*>(Aimg + ((mo * 2 + ks) * 64 + lane) * 32);
            acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, acc, 0 /* A: e4m3 */, 0 /* B: e4m3 */, 0, 127, 0, 127);
        }
        for (int i = 0; i < 16; ++i) Y[feat_of(mo, i, h) * 32 + n] = acc[i];
    }
}

static uint8_t e4m3_of(float v) {        // exact for the values used here (multiples of 1/4 up to 15.75 in magnitude)
    if (v == 0.f) return 0;
    const uint8_t s = v < 0 ? 0x80 : 0;
    float a = std::fabs(v);
    int e = 0;
    while (a >= 2.f) { a *= 0.5f; ++e; }
    while (a < 1.f) { a *= 2.f; --e; }
    const int mant = (int)std::lround((a - 1.f) * 8.f);
    return s | (uint8_t)(((e + 7) << 3) | mant);
}

int main() {
    std::vector<float> H(128 * 32), W(128 * 128), Yref(128 * 32), Y(128 * 32);
    unsigned r = 12345u;
    auto rnd = [&]() { r = r * 1664525u + 1013904223u; return (int)(r >> 16); };
    for (auto& v : H) { const int t = rnd() % 9; v = t < 3 ? 0.f : 0.25f * (float)(t - 2); }          // post-ReLU: zeros and 0.25 .. 1.5
    for (auto& v : W) v = 0.25f * (float)(rnd() % 13 - 6);                                              // -1.5 .. 1.5
    for (int o = 0; o < 128; ++o)
        for (int n = 0; n < 32; ++n) {
            float s = 0.f;
            for (int f = 0; f < 128; ++f) s += W[o * 128 + f] * H[f * 32 + n];
            Yref[o * 32 + n] = s;
        }
    // weight image: row tile mo, k-step ks, lane (r, h), byte b -> W[32 mo + r][feature of row tile 2ks + (b >> 4), register b & 15, half h]
    std::vector<uint8_t> A(4 * 2 * 64 * 32);
    for (int mo = 0; mo < 4; ++mo)
        for (int ks = 0; ks < 2; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int b = 0; b < 32; ++b) {
                    const int rr = lane & 31, h = lane >> 5;
                    // NOTE the output rows of a 32 x 32 tile come back in accumulator order as well: row rr of the MFMA is row rr of W's tile
                    A[((mo * 2 + ks) * 64 + lane) * 32 + b] = e4m3_of(W[(32 * mo + rr) * 128 + feat_of(2 * ks + (b >> 4), b & 15, h)]);
                }
    float *dH, *dY; uint8_t* dA;
    (void)hipMalloc(&dH, H.size() * 4); (void)hipMalloc(&dY, Y.size() * 4); (void)hipMalloc(&dA, A.size());
    (void)hipMemcpy(dH, H.data(), H.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dH, dA, dY);
    (void)hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int o = 0; o < 128; ++o)
        for (int n = 0; n < 32; ++n)
            if (Y[o * 32 + n] != Yref[o * 32 + n] && bad++ < 8) printf("  Y[%d][%d] = %g, expected %g\n", o, n, Y[o * 32 + n], Yref[o * 32 + n]);
    // (the MFMA's 32 output rows map to accumulator registers as rows 8(i>>2) + 4h + (i&3): feat_of(mo, i, h) - 32 mo; row rr of the A operand is that row)
    printf("%s: %d of %d outputs differ\n", bad ? "MISMATCH" : "exact", bad, 128 * 32);
    return bad != 0;
}
