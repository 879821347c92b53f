// nca_kernels_f32.hip -- gfx950 kernels of the f32 (parity) path.
//
//   nca_pack_f32      natural flat parameters -> MFMA-ordered LDS images
//   nca_fused_f32<F,BWD>  one pass over ray-ordered samples:
//         BWD=false: query point -> positional encoding -> static MLP -> dynamic MLP ->
//                    activation -> per-ray partial sums            (model_helpers.py:115-129)
//         BWD=true : the same recompute, then the backward sweep (dgrad) of each net; hidden layer
//                    inputs H and output gradients D go to the scratch in register order (quads of
//                    four rows per lane, 1 KiB per store, issued inside the contraction that consumes
//                    them), the encoded input row-major; ReLU masks stay in LDS
//   nca_wgrad_f32     dW = D * H^T over the sample axis (split over workgroups), bias sums
//   nca_reduce_f32    fixed-order sum of the split slabs -> natural flat gradients
//   nca_pix_f32       pix = I0 - sum of the per-tile partial ray sums
//
// Data layout: activations are transposed, H[feature][sample]: a wave owns 32 consecutive samples
// of one ray (lane&31 = sample, lane>>5 = k-half), a 32x32 accumulator tile per 32 features.  A
// layer's output registers are the next layer's B operand in place (see nca_layout.hpp).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <cstdlib>
#include "nca_kernels.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4e __attribute__((ext_vector_type(4)));

#define NCA_HALF_PI_F 1.57079637050628662109375f   // fl32(0.5 * pi), the constant the reference adds
#define NCA_HALF_PI_D 1.57079632679489661923
#define NCA_TWO_PI_F 6.283185482025146484375f      // fl32(2 * pi)

// Rounding ablations (tools/ablation_build.sh; 0 in every shipped library -- nca_build_info() reports the mask): the
// parity kernels with ONE of the bf16 mode's roundings switched on, to find which of them costs held-out PSNR.  1: encoded input
// features, 2: hidden-layer weights, 4: hidden activations (after ReLU), 8: output gradients of the dgrad chain, 16: layer-0 weights
// -- each rounded to NCA_ABL_MANT significant bits (8 = bf16, 11 = f16's precision without its range) --, 32: the layer inputs as the
// weight-gradient kernel reads them (what is STORED; the chain keeps f32) to 4 significant bits (e4m3's), 64: the stored output
// gradients to 3 (e5m2's), 128: the TRANSPOSED weight images only (what the dgrad chain multiplies with) to 4 bits (e4m3's).
// 256 / 512: the stored hidden-layer inputs / output gradients as SIX-bit floats (e2m3 / e3m2, the MX fp6 / bf6 formats) under one
// power-of-two scale per lane and pair of row tiles (32 values: what one v_cvt_scalef32_pk32_{fp6,bf6} would convert) -- DESIGN.md 7's
// 6-bit staging, asked of the PSNR gate before any kernel is written.  (Not the last hidden layer's output, which the f32 backward
// reads back into its chain, and not the encoded input block: those keep bit 32's treatment.)
#ifndef NCA_ABL
#define NCA_ABL 0
#endif
#ifndef NCA_ABL_MANT
#define NCA_ABL_MANT 8
#endif
template <int BIT>
__device__ __forceinline__ float abl(float x) {
    if constexpr ((NCA_ABL & BIT) != 0) {
        constexpr int MANT = (BIT == 32 || BIT == 128) ? 4 : (BIT == 64 ? 3 : NCA_ABL_MANT), DROP = 24 - MANT;
        const unsigned u = __float_as_uint(x);
        return __uint_as_float((u + ((1u << (DROP - 1)) - 1u) + ((u >> DROP) & 1u)) & ~((1u << DROP) - 1u));     // round to nearest even
    } else return x;
}
int nca_kernels_ablation_mask() { return NCA_ABL | (NCA_ABL ? NCA_ABL_MANT << 16 : 0); }

// x3 split (see the x3 section below): exact three-way bf16 split of f32 values
typedef float x3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 x3_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 x3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned x3_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void x3_split_pair(float lo, float hi, unsigned& p0, unsigned& p1, unsigned& p2) {
    const x3_f32x2 v = {lo, hi};
    const x3_bf16x2 q1 = __builtin_convertvector(v, x3_bf16x2);
    const x3_f32x2 r1 = v - __builtin_convertvector(q1, x3_f32x2);
    const x3_bf16x2 q2 = __builtin_convertvector(r1, x3_bf16x2);
    const x3_f32x2 r2 = r1 - __builtin_convertvector(q2, x3_f32x2);
    const x3_bf16x2 q3 = __builtin_convertvector(r2, x3_bf16x2);
    p0 = __builtin_bit_cast(unsigned, q1);
    p1 = __builtin_bit_cast(unsigned, q2);
    p2 = __builtin_bit_cast(unsigned, q3);
}
// piece p (0..2) of one value, as the pack kernel needs it
__device__ __forceinline__ unsigned x3_piece(float w, int p) {
    unsigned a, b, c;
    x3_split_pair(w, 0.f, a, b, c);
    return (p == 0 ? a : (p == 1 ? b : c)) & 0xffffu;
}

// ------------------------------------------------------------------------------------------
// pack
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_f32_body(const NcaLayout& y, const float* __restrict__ prm_, float* __restrict__ out) {
#if NCA_ABL & (2 | 16)
    // (ablation: weights of the F-wide layers as the bf16 mode sees them; biases, Wo and bo stay f32 there too)
    struct Rounded {
        const float* p; NcaLayout y;
        __device__ float operator[](int i) const {
            for (int j = 0; j < y.NL; ++j)
                if (i >= y.layer[j].w_off && i < y.layer[j].b_off) return (j == 0) ? abl<16>(p[i]) : abl<2>(p[i]);
            return p[i];
        }
    } prm{prm_, y};
#else
    const float* __restrict__ prm = prm_;
#endif
    const uint32_t total = y.packed_bytes / 4u;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        float v = 0.f;
        const uint32_t byte = e * 4u;
        for (int j = 0; j < y.NL; ++j) {
            const NcaLayerL& l = y.layer[j];
            const bool skip = l.kind == NCA_IN_SKIP;
            if (skip && byte >= l.img2_off && byte < l.img2_off + l.img2_bytes) {
                // second stage of a skip layer: hidden-part k-steps, then (last layer) Wo and bo
                uint32_t q = (byte - l.img2_off) / 4u;
                const uint32_t wcount = (uint32_t)(l.ksteps - l.ksteps_enc) * 64u * (uint32_t)y.MT;
                const uint32_t tail = 2u * (uint32_t)y.MT * 16u;
                if (q < wcount) {
                    int m = q % y.MT, lane = (q / y.MT) % 64, s = q / (y.MT * 64);
                    int r = lane & 31, h = lane >> 5;
                    v = prm[l.w_off + (32 * m + r) * l.K + y.K0 + nca_kidx_hidden(s, h)];
                } else if (j == y.NL - 1) {
                    q -= wcount;
                    if (q < tail) {
                        int i = q % 16, m = (q / 16) % y.MT, h = q / (16 * y.MT);
                        v = prm[y.wo_off + 32 * m + nca_rho(i) + 4 * h];
                    } else if (q == tail) {
                        v = prm[y.bo_off];
                    }
                }
            }
            if (y.x3) {
                // x3 images: A-fragment planes of bf16 pieces (nca_layout.hpp); a dword holds elements j = 2u, 2u + 1
                const int KH = nca_x3_kh(y.F);                             // k-steps per sub-stage
                const uint32_t nfr = 3u * y.MT * KH * 256u;                // dwords of fragments in a sub-stage
                const uint32_t tailn = 2u * (uint32_t)y.MT * 16u;
                bool done = false;
                for (int sub = 0; sub < 2 && !done; ++sub) {
                    for (int tr = 0; tr < 2 && !done; ++tr) {
                        if (tr == 0 && l.kind != NCA_IN_HID) continue;     // forward images of encoded / skip layers stay f32
                        if (tr == 1 && l.kind == NCA_IN_ENC) continue;
                        const uint32_t base = tr ? (sub ? l.imgT2_off : l.imgT_off) : (sub ? l.img2_off : l.img_off);
                        const uint32_t bytes = tr ? (sub ? l.imgT2_bytes : l.imgT_bytes) : (sub ? l.img2_bytes : l.img_bytes);
                        if (!bytes || byte < base || byte >= base + bytes) continue;
                        done = true;
                        uint32_t q = (byte - base) / 4u;
                        if (q < nfr) {
                            const int pc = q / (y.MT * KH * 256), r = q % (y.MT * KH * 256);
                            const int mm = r / (KH * 256), ks = sub * KH + (r / 256) % KH, ln = (r % 256) / 4, u = r % 4;
                            const int o = 32 * mm + (ln & 31), h = ln >> 5;
                            const int k0 = nca_x3_kidx(ks, h, 2 * u), k1 = nca_x3_kidx(ks, h, 2 * u + 1);
                            float w0, w1;
                            if (!tr) { w0 = prm[l.w_off + o * l.K + k0]; w1 = prm[l.w_off + o * l.K + k1]; }
                            else {
                                const int col0 = l.kind == NCA_IN_SKIP ? y.K0 : 0;
                                w0 = abl<128>(prm[l.w_off + k0 * l.K + col0 + o]); w1 = abl<128>(prm[l.w_off + k1 * l.K + col0 + o]);
                            }
                            v = __uint_as_float(x3_piece(w0, pc) | (x3_piece(w1, pc) << 16));
                        } else if (!tr) {
                            q -= nfr;
                            const bool two = y.MT >= 2;
                            // sub-stage 0: bias tail (then Wo, bo if the layer has ONE sub-stage and is the last); sub-stage 1: Wo, bo
                            bool want_wo = false;
                            if (sub == 0) {
                                if (q < tailn) {
                                    int i = q % 16, m = (q / 16) % y.MT, h = q / (16 * y.MT);
                                    v = prm[l.b_off + 32 * m + nca_rho(i) + 4 * h];
                                } else if (!two && j == y.NL - 1) { q -= tailn; want_wo = true; }
                            } else if (j == y.NL - 1) want_wo = true;
                            if (want_wo) {
                                if (q < tailn) {
                                    int i = q % 16, m = (q / 16) % y.MT, h = q / (16 * y.MT);
                                    v = prm[y.wo_off + 32 * m + nca_rho(i) + 4 * h];
                                } else if (q == tailn) v = prm[y.bo_off];
                            }
                        }
                    }
                }
                if (done) continue;
            }
            if (byte >= l.img_off && byte < l.img_off + l.img_bytes) {
                uint32_t q = (byte - l.img_off) / 4u;
                const uint32_t wcount = (uint32_t)(skip ? l.ksteps_enc : l.ksteps) * 64u * (uint32_t)y.MT;
                const uint32_t tail = 2u * (uint32_t)y.MT * 16u;
                if (q < wcount) {
                    int m = q % y.MT, lane = (q / y.MT) % 64, s = q / (y.MT * 64);
                    int r = lane & 31, h = lane >> 5, k;
                    if (s < l.ksteps_enc) {
                        int ia, ib;
                        nca_enc_pair(y, s, &ia, &ib);
                        k = h ? ib : ia;
                    } else {
                        k = (l.kind == NCA_IN_SKIP ? y.K0 : 0) + nca_kidx_hidden(s - l.ksteps_enc, h);
                    }
                    v = k >= 0 ? prm[l.w_off + (32 * m + r) * l.K + k] : 0.f;
                } else if (q < wcount + tail) {
                    q -= wcount;
                    int i = q % 16, m = (q / 16) % y.MT, h = q / (16 * y.MT);
                    v = prm[l.b_off + 32 * m + nca_rho(i) + 4 * h];
                } else if (j == y.NL - 1 && !skip) {
                    q -= wcount + tail;
                    if (q < tail) {
                        int i = q % 16, m = (q / 16) % y.MT, h = q / (16 * y.MT);
                        v = prm[y.wo_off + 32 * m + nca_rho(i) + 4 * h];
                    } else if (q == tail) {
                        v = prm[y.bo_off];
                    }
                }
            }
            if (!y.x3 && l.imgT_bytes && byte >= l.imgT_off && byte < l.imgT_off + l.imgT_bytes) {
                uint32_t q = (byte - l.imgT_off) / 4u;
                int m = q % y.MT, lane = (q / y.MT) % 64, s = q / (y.MT * 64);
                int r = lane & 31, h = lane >> 5;
                int col0 = l.kind == NCA_IN_SKIP ? y.K0 : 0;
                v = prm[l.w_off + nca_kidx_hidden(s, h) * l.K + col0 + 32 * m + r];
            }
        }
        out[e] = v;
    }
}
__global__ void nca_pack_f32(NcaLayout y, const float* __restrict__ prm, float* __restrict__ out) { pack_f32_body(y, prm, out); }
// both nets of a composite render in one launch: blockIdx.y = net
struct NcaPack2Args { NcaLayout y[2]; const float* prm[2]; void* out[2]; };
__global__ void nca_pack2_f32(NcaPack2Args a) { pack_f32_body(a.y[blockIdx.y], a.prm[blockIdx.y], static_cast<float*>(a.out[blockIdx.y])); }

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
template <int MT> struct AVec;
template <> struct AVec<1> { typedef float T; };
template <> struct AVec<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <> struct AVec<4> { typedef float T __attribute__((ext_vector_type(4))); };

template <int MT>
__device__ __forceinline__ void load_a(const float* p, float (&a)[MT]) {
    typename AVec<MT>::T v = *reinterpret_cast<const typename AVec<MT>::T*>(p);
    if constexpr (MT == 1) a[0] = v;
    else {
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = v[m];
    }
}

__device__ __forceinline__ float act_fwd(int act, float x) {
    // torch.nn.Softplus(beta=1, threshold=20) / Sigmoid / hardtanh(softplus, 0, 1)
    if (act == NCA_ACT_SIGMOID) return 1.f / (1.f + expf(-x));
    float sp = x > 20.f ? x : log1pf(expf(x));
    if (act == NCA_ACT_CLAMP) sp = fminf(fmaxf(sp, 0.f), 1.f);
    return sp;
}
__device__ __forceinline__ float act_bwd(int act, float x) {
    if (act == NCA_ACT_SIGMOID) { float s = 1.f / (1.f + expf(-x)); return s * (1.f - s); }
    float d;
    if (x > 20.f) d = 1.f; else { float z = expf(x); d = z / (z + 1.f); }
    if (act == NCA_ACT_CLAMP) {
        float sp = x > 20.f ? x : log1pf(expf(x));
        if (!(sp > 0.f && sp < 1.f)) d = 0.f;
    }
    return d;
}

// sum over the 32 lanes of each wave half
__device__ __forceinline__ float half_sum(float v) {
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ double half_sum(double v) {
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}

// One hidden-width contraction acc[m] += W[m-tile][k] * B[k] over the F/2 k-steps of an image whose B
// operands are the previous layer's accumulator registers.  The A fragment of step s+1 is read
// while step s's MFMAs issue; the scheduling barrier keeps the compiler from hoisting dozens of LDS
// reads (and their registers) to the top of the unrolled loop.
// scratch blocks are written once and read once by another kernel: stream them past the L2, which then keeps the
// weight images the per-layer LDS DMA re-reads (measured: 3.91 -> 3.58 ms per launch)
__device__ __forceinline__ void store_quad(float* p, float a, float b, float c, float d) {
    const f32x4e v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4e*>(p));
}


// ------------------------------------------------------------------------------------------
// x3: hidden-width contractions on the bf16 matrix cores with f32 accuracy (see nca_wgrad_f32x3 below for the
// arithmetic: x = x1 + x2 + x3 exactly, six piece products of weight >= 2^-16, f32 accumulation).  Here the weights come
// pre-split from the packed image (three bf16 A-fragment planes per sub-stage, nca_layout.hpp) and the activations are
// split when they are packed into B fragments: registers 8 s + 2 u, + 1 of row tile t -> word u of k-step 2 t + s.
// ------------------------------------------------------------------------------------------
// One sub-stage: k-steps [K0, K0 + KH) of the contraction for all MT row tiles.  `sub` = the sub-stage image + lane * 16.
// The B fragments of ONE k-step are split out of the previous layer's accumulator registers (h) right before their use
// (12 registers instead of 48 per half, and the split's VALU work is spread between the MFMAs); the A fragments of the
// next (k-step, row-tile pair) are requested before the current MFMAs issue.
template <int MT, int KH, int K0>
__device__ __forceinline__ void x3_sub(const char* __restrict__ sub, const f32x16 (&h)[MT], f32x16 (&acc)[MT]) {
#ifndef NCA_X3_RG
#define NCA_X3_RG 2
#endif
    constexpr int RG = MT >= 2 ? NCA_X3_RG : 1, NG = MT / RG, NIT = KH * NG;
    x3_u32x4 A[2][3][RG], B[3];
    auto load = [&](int it, x3_u32x4 (&a)[3][RG]) {
        const int kk = it / NG, m0 = (it % NG) * RG;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int mm = 0; mm < RG; ++mm) a[p][mm] = *reinterpret_cast<const x3_u32x4*>(sub + ((p * MT + m0 + mm) * KH + kk) * 1024);
    };
    load(0, A[0]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int kk = it / NG, m0 = (it % NG) * RG;
        if (it % NG == 0) {
            const int t = (K0 + kk) >> 1, s2 = (K0 + kk) & 1;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                unsigned b0, b1, b2;
                x3_split_pair(h[t][8 * s2 + 2 * u], h[t][8 * s2 + 2 * u + 1], b0, b1, b2);
                B[0][u] = b0; B[1][u] = b1; B[2][u] = b2;
            }
        }
        if (it + 1 < NIT) load(it + 1, A[(it + 1) & 1]);
        // small products first; consecutive MFMAs alternate accumulators
#define X3_MMA(I, J)                                                                                                       \
        _Pragma("unroll") for (int mm = 0; mm < RG; ++mm)                                                               \
            acc[m0 + mm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(x3_bf16x8, A[it & 1][I][mm]), __builtin_bit_cast(x3_bf16x8, B[J]), acc[m0 + mm], 0, 0, 0);
        X3_MMA(1, 1) X3_MMA(2, 0) X3_MMA(0, 2) X3_MMA(1, 0) X3_MMA(0, 1) X3_MMA(0, 0)
#undef X3_MMA
        __builtin_amdgcn_sched_barrier(0);
    }
}
// NCA_ABL bits 256 / 512: a group of 32 values (two row tiles of one lane; 16 at width 32) as fp6 under one power-of-two scale:
// BF6 = false: e2m3 (magnitudes 0, 1/8 .. 7/8, 1 .. 7.5), BF6 = true: e3m2 (0, 1/16 .. 3/16, 1/4 .. 28); round to nearest even,
// saturating; the scale puts the group's maximum into the top binade (one binade up if it would round past the largest value).
template <bool BF6, int N>
__device__ __forceinline__ void abl_fp6_group(const f32x16 (&h)[N], f32x16 (&q)[N]) {
    constexpr int TOP = BF6 ? 4 : 2, MBITS = BF6 ? 2 : 3, EMIN = BF6 ? -2 : 0;
    constexpr float VMAX = BF6 ? 28.f : 7.5f, LIMIT = BF6 ? 30.f : 7.75f;
    float gmax = 0.f;
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) gmax = fmaxf(gmax, fabsf(h[m][i]));
    int e = (int)((__float_as_uint(gmax) >> 23) & 255u) - 127 - TOP;
    if (e < -120) e = -120;                                       // (zero / denormal groups: everything rounds to 0 below)
    if (gmax * __uint_as_float((unsigned)(127 - e) << 23) >= LIMIT) ++e;
    const float s = __uint_as_float((unsigned)(127 + e) << 23), inv = __uint_as_float((unsigned)(127 - e) << 23);
#pragma unroll
    for (int m = 0; m < N; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = fabsf(h[m][i]) * inv;                 // exact (power of two)
            int ev = (int)((__float_as_uint(v) >> 23) & 255u) - 127;
            if (ev < EMIN) ev = EMIN;
            const float step = __uint_as_float((unsigned)(127 + ev - MBITS) << 23), istep = __uint_as_float((unsigned)(127 - ev + MBITS) << 23);
            float r = rintf(v * istep) * step;
            if (r > VMAX) r = VMAX;
            q[m][i] = copysignf(r * s, h[m][i]);
        }
}
template <int MT, int ABL_BIT>
__device__ __forceinline__ void x3_store_block(float* st, const f32x16 (&h)[MT]) {
    if constexpr ((ABL_BIT == 32 && (NCA_ABL & 256) != 0) || (ABL_BIT == 64 && (NCA_ABL & 512) != 0)) {
        constexpr int G = MT >= 2 ? 2 : 1;
#pragma unroll
        for (int m0 = 0; m0 < MT; m0 += G) {
            f32x16 in[G], q[G];
#pragma unroll
            for (int g = 0; g < G; ++g) in[g] = h[m0 + g];
            abl_fp6_group<ABL_BIT == 64, G>(in, q);
#pragma unroll
            for (int g = 0; g < G; ++g)
#pragma unroll
                for (int k = 0; k < 4; ++k) store_quad(st + ((m0 + g) * 4 + k) * 256, q[g][4 * k], q[g][4 * k + 1], q[g][4 * k + 2], q[g][4 * k + 3]);
        }
        return;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int g = 0; g < 4; ++g) store_quad(st + (m * 4 + g) * 256, abl<ABL_BIT>(h[m][4 * g]), abl<ABL_BIT>(h[m][4 * g + 1]), abl<ABL_BIT>(h[m][4 * g + 2]), abl<ABL_BIT>(h[m][4 * g + 3]));
}

// Backward: the B operands are also what the weight-gradient kernel needs (a layer input H or an output gradient D), and
// they sit untouched in registers for the whole contraction -- so their 4 MT quad stores (1 KiB each) are issued here,
// one every fourth k-step, and drain under the MFMAs instead of in a burst of all eight waves before the layer barrier.
template <int MT>
__device__ __forceinline__ void hidden_steps(const float* __restrict__ ih, const f32x16 (&b)[MT], f32x16 (&acc)[MT],
                                             float* st = nullptr, bool do_store = false) {
    constexpr int NS = 16 * MT;
    float av[2][MT];
    load_a<MT>(ih, av[0]);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 1 < NS) load_a<MT>(ih + (s + 1) * 64 * MT, av[(s + 1) & 1]);
        if ((s & 3) == 0 && do_store) {
            const int qd = s >> 2, m = qd >> 2, g = qd & 3;       // quad g of row tile m: registers 4g .. 4g+3
            store_quad(st + qd * 256, b[m][4 * g], b[m][4 * g + 1], b[m][4 * g + 2], b[m][4 * g + 3]);
        }
        const float bop = b[s >> 4][s & 15];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][m], bop, acc[m], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Encoded input generator: calls step(s, a, b) once per k-step with the two features of the pair
// (a feeds lane-half 0, b lane-half 1), in the order fixed by nca_enc_pair.
//   bands:   sin(2^k x) and sin(fl32(2^k x + fl32(pi/2)))  (model/CPPN.py:121-123), times window[k].
//            Evaluated by angle doubling in f64 from one sincos per coordinate; the reference's
//            rounded "+pi/2" is reproduced exactly through eps = fl32(xb + c) - xb - pi/2.
//   fourier: sin/cos(fl32(fl32(2pi * x) * g))                (model/CPPN.py:115-118)
template <class Step>
__device__ __forceinline__ void enc_steps(const NcaLayout& y, const float (&p)[3], const float* __restrict__ win,
                                          const float* __restrict__ four, const float* __restrict__ lat, Step&& step) {
    int s = 0;
    if (y.enc_mode != NCA_ENC_FOURIER) {
        step(s++, p[0], p[1]);
        step(s++, p[2], 0.f);
    }
    if (y.enc_mode == NCA_ENC_BANDS) {
        double sn[3], cs[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) sincos((double)p[c], &sn[c], &cs[c]);
        float scl = 1.f;
        for (int k = 0; k < y.L; ++k) {
            const float w = win[k];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float xb = p[c] * scl;                      // exact (power of two)
                const float t = __fadd_rn(xb, NCA_HALF_PI_F);     // the reference's rounded argument
                const double eps = ((double)t - (double)xb) - NCA_HALF_PI_D;
                const double e2 = eps * eps;
                const double cf = cs[c] * (1.0 - 0.5 * e2) - sn[c] * (eps - eps * e2 * (1.0 / 6.0));
                step(s++, w * (float)sn[c], w * (float)cf);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double s2 = 2.0 * sn[c] * cs[c];
                const double c2 = 1.0 - 2.0 * sn[c] * sn[c];
                sn[c] = s2; cs[c] = c2;
            }
            scl *= 2.f;
        }
    } else if (y.enc_mode == NCA_ENC_FOURIER) {
        const int n = 3 * y.L;
        for (int i = 0; i < n; ++i) {
            const int c = i % 3;
            const float pc = c == 0 ? p[0] : (c == 1 ? p[1] : p[2]);
            const float v = __fmul_rn(__fmul_rn(NCA_TWO_PI_F, pc), four[i]);
            double sv, cv;
            sincos((double)v, &sv, &cv);
            step(s++, (float)sv, (float)cv);
        }
    }
    for (int u = 0; 2 * u < y.T; ++u) {
        const float a = lat[2 * u];
        const float b = (2 * u + 1 < y.T) ? lat[2 * u + 1] : 0.f;
        step(s++, a, b);
    }
}

// ------------------------------------------------------------------------------------------
// fused forward / backward-dgrad kernel
// ------------------------------------------------------------------------------------------
template <int F>
struct FusedCfg {
    static constexpr int MT = F / 32;
    static constexpr int IMG_MAX = NCA_MAX_KSTEPS * 64 * MT * 4 + 2 * (2 * MT * 16 * 4) + 16;   // largest image: k-steps + bias + Wo/bo tails
    static constexpr int BUF_BYTES = (IMG_MAX + 1023) & ~1023;
};

// small per-net constants staged once per workgroup into LDS (so that no ordinary global load sits
// between a weight DMA and its consumer): band window, fourier coefficients, time latents
#define NCA_CONST_WIN 16
#define NCA_CONST_FOUR 48
#define NCA_CONST_LAT 2048
#define NCA_CONST_NET_FLOATS (NCA_CONST_WIN + NCA_CONST_FOUR + NCA_CONST_LAT)
#define NCA_CONST_BYTES (2 * NCA_CONST_NET_FLOATS * 4)

// LDS-DMA one weight image (whole 1 KiB pieces, round-robin over the waves) into `dst`
__device__ __forceinline__ void stage_issue(const NcaStage& st, char* dst, int wave, int lane) {
    const int npiece = (int)(st.bytes >> 10);
    const char* src = reinterpret_cast<const char*>(st.ptr) + lane * 16;
    for (int c = wave; c < npiece; c += NCA_WAVES) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)c * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
    }
}
__device__ __forceinline__ void stage_publish() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// Backward stages of hidden layers issue exactly NST scratch stores AFTER the weight DMA of the stage (and every load
// issued in between has been consumed by then).  VMEM operations retire in issue order, so waiting until NST remain
// outstanding waits for the DMA but not for those stores: the 16 KiB a wave writes per layer drain under the next layer.
template <int NST>
__device__ __forceinline__ void stage_publish_counted(bool stores_issued) {
    if (stores_issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// Kernel modes as in the bf16 kernel (nca_kernels.hpp): 0 forward, 1 recompute backward (one scratch for H and D),
// 2 forward that also stores the input block / layer inputs / ReLU masks / raw outputs, 3 backward from that store.
#ifndef NCA_F32_MINBLOCKS
#define NCA_F32_MINBLOCKS 2     // (1 with NCA_WAVES=4: one 512-register wave per SIMD -- tools/variant_build_all.sh, timing experiment)
#endif
// SKIP: some net of the launch has a skip layer (CPPN with num_late_layers > 0, model/CPPN.py:53-58, 102-106): its encoded-input pass runs with the
// previous layer's output live.  The nets the reference ships have none; their instantiation (SKIP = false) has no encoding inside the
// hidden-layer loop and needs no spill for it (profiles/r06_kernel_resources.txt).
template <int F, int MODE, bool X3, bool SKIP>
__global__ __launch_bounds__(NCA_NT, NCA_F32_MINBLOCKS) void nca_fused_f32(const NcaFusedArgs a) {
    constexpr bool BWD = MODE == NCA_KM_BWD || MODE == NCA_KM_BWD_STORED;
    constexpr bool STORE = MODE == NCA_KM_BWD || MODE == NCA_KM_FWD_STORE;
    constexpr bool RECOMP = MODE != NCA_KM_BWD_STORED;
    constexpr bool FSTORE = MODE == NCA_KM_FWD_STORE, STORED = MODE == NCA_KM_BWD_STORED;
    constexpr int MT = FusedCfg<F>::MT;
    constexpr int BUF = FusedCfg<F>::BUF_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [buf0][buf1][constants of both nets][per-wave output-layer gradient sums: WAVES x 2 nets x (F+1)]
    // [buf0][buf1][constants of both nets][per-wave output-layer gradient sums][per-wave ReLU masks: 8 B per lane and layer]
    float* cst = reinterpret_cast<float*>(smem + 2 * BUF);
    const int cnf = a.const_net_floats;
    float* osum = cst + 2 * cnf;
    char* const maskbase = reinterpret_cast<char*>(osum + NCA_WAVES * 2 * (F + 1));
    float* const wos = reinterpret_cast<float*>(maskbase);      // mode 3: [Wo | bo] of both nets (no last-layer image in LDS)
    constexpr int WOS = 2 * MT * 16 + 16;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lj = lane & 31, lh = lane >> 5;

    for (int net = 0; net < a.nnets; ++net) {
        const NcaNetArgs& na = a.net[net];
        float* c = cst + net * cnf;
        if (na.win) for (int i = tid; i < na.lay.L; i += NCA_NT) c[i] = na.win[i];
        if (na.four) for (int i = tid; i < 3 * na.lay.L; i += NCA_NT) c[NCA_CONST_WIN + i] = na.four[i];
        if (na.lat) for (int i = tid; i < na.lay.P * na.lay.T; i += NCA_NT) c[NCA_CONST_WIN + NCA_CONST_FOUR + i] = na.lat[i];
    }
    if (BWD) {
        for (int i = tid; i < NCA_WAVES * 2 * (F + 1); i += NCA_NT) osum[i] = 0.f;
    }
    if (STORED)
        for (int net = 0; net < a.nnets; ++net)
            for (int i = tid; i < 2 * MT * 16 + 1; i += NCA_NT) wos[net * WOS + i] = a.net[net].wo_src[i];
    __syncthreads();

    // stage 0 -> buffer 0
    stage_issue(a.stage[0], smem, wave, lane);
    stage_publish();
    int cur = 0, si = 0;

    const int64_t ngroups = (a.ntiles + NCA_WAVES - 1) / NCA_WAVES;
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t tile = grp * NCA_WAVES + wave;
        const bool tvalid = tile < a.ntiles;
        const int64_t tl = tvalid ? tile : a.ntiles - 1;

        // ---- per-lane sample: query point, phase (model_helpers.py:117-122) ------------------
        int64_t ray = 0, n = 0;
        int smp = 0;
        bool valid;
        float p[3];
        if (a.mode == NCA_MODE_RAYS) {
            ray = a.ray0 + tl / a.nchunk;
            const int chunk = (int)(tl % a.nchunk);
            smp = chunk * 32 + lj;
            valid = tvalid && smp < a.S;
            if (smp >= a.S) smp = a.S - 1;
            n = ray * a.S + smp;
            const float zz = RECOMP ? a.z[ray * a.zs_r + smp] : 0.f;
            if (!RECOMP) {
                p[0] = p[1] = p[2] = 0.f;
            } else if (a.ray_is_f64) {
                const double* o = reinterpret_cast<const double*>(a.origins) + ray * 3;
                const double* d = reinterpret_cast<const double*>(a.dirs) + ray * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c] = (float)__dadd_rn(o[c], __dmul_rn(d[c], (double)zz));
            } else {
                const float* o = reinterpret_cast<const float*>(a.origins) + ray * 3;
                const float* d = reinterpret_cast<const float*>(a.dirs) + ray * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c] = __fadd_rn(o[c], __fmul_rn(d[c], zz));
            }
        } else {
            n = a.n0 + tl * 32 + lj;
            valid = tvalid && n < a.N;
            if (n >= a.N) n = a.N - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = RECOMP ? a.pts[n * 3 + c] : 0.f;
        }
        int ph = 0;
        if (RECOMP && a.phase) {
            ph = a.mode == NCA_MODE_RAYS ? a.phase[ray * a.ps_r + (int64_t)smp * a.ps_s] : a.phase[n];
        }
        // backward scratch is tile-major: scratch[tile][row][32 columns].  A wave owns one tile, so every
        // row it touches sits at a compile-time offset (row * 128 B) from one per-lane base; lane-half h
        // owns rows rho(i)+4h and carries those 4 rows in its base.
        // The input block and the layer inputs live in the H region (indexed by the tile's position in the whole batch
        // when a storing forward wrote it), the output gradients in the D region of this launch (mode 1: the same).
        const int64_t tg = tl + a.tile0;
        float* const tcol = STORE ? a.scratch + tg * a.rows_total * 32 + lj : nullptr;

        float raw[2] = {0.f, 0.f};

#pragma unroll
        for (int net = 0; net < 2; ++net) {
            if (net >= a.nnets) break;
            const NcaNetArgs& na = a.net[net];
            const NcaLayout& y = na.lay;
            int phc = ph < 0 ? 0 : (ph >= y.P ? y.P - 1 : ph);
            const float* cnet = cst + net * cnf;
            const float* cwin = cnet;
            const float* cfour = cnet + NCA_CONST_WIN;
            const float* lat = y.T > 0 ? cnet + NCA_CONST_WIN + NCA_CONST_FOUR + phc * y.T : nullptr;
            float* const hc = STORE ? tcol + na.row0 * 32 : nullptr;         // the row-major input block: this lane's column
            // hidden blocks (H, D) are stored as the accumulators sit in registers: [row tile][register quad][lane][4 floats],
            // quad g of lane (r, h) = rows 32 m + 8 g + 4 h + 0..3 of sample r -- one 1 KiB store per wave instruction.
            // hf: this net's rows of the H region (input block first), df: its D blocks
            float* const hf = (BWD || STORE) ? a.scratch + (tg * a.rows_total + na.row0) * 32 + lane * 4 : nullptr;
            float* const df = BWD ? reinterpret_cast<float*>(a.dscratch) + (tl * a.d_total + na.drow0) * 32 + lane * 4 : nullptr;
            // storing forward / backward from the store: masks [tile][net][layer][lane][8 B], raw outputs [tile][net][32]
            char* const mglob = (FSTORE || STORED) ? a.mstore + (((tg * 2 + net) * a.mstore_layers) * 64 + lane) * 8 : nullptr;
            float* const rglob = (FSTORE || STORED) ? a.rstore + (tg * 2 + net) * 32 + lj : nullptr;

            f32x16 hprev[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) hprev[m] = (f32x16)(0.f);

            // Gradient wrt the raw output, output-layer parameter gradients and D_{NL-1}: expects the last hidden layer's
            // output in hprev and this net's raw output in raw[net]; leaves D_{NL-1} in hprev.  `wo` = [Wo | bo] in LDS.
            auto last_layer_grads = [&](const float* wo) __attribute__((always_inline)) {
                        // gradient wrt the raw output of this net
                        float g;
                        if (a.mode == NCA_MODE_RAYS && !a.g_raw) {
                            const float* gs = net == 0 ? a.g_sig_s : a.g_sig_d;
                            const double gsig = gs ? (double)gs[n] : 0.0;
                            const double gp = a.g_pix[ray] * a.dists[smp];
                            const double dsig = a.single ? (gsig - gp * (double)a.scale) : (gsig - gp) * (double)a.scale;
                            g = (float)dsig * act_bwd(a.act, raw[net]);
                        } else {
                            g = a.g_raw[n];
                        }
                        if (!valid) g = 0.f;
                        // dWo[f] = sum_n g H[f][n]: reduce-scatter over the 32 lanes of each wave half, at most 32
                        // values at a time (two row tiles), so that lane lj ends up owning flat index lj*per+e
                        {
                            float* orow = osum + (wave * 2 + net) * (F + 1);
                            constexpr int GM = MT >= 2 ? 2 : 1;          // row tiles per pass
                            constexpr int NV = GM * 16;
#pragma unroll
                            for (int m0 = 0; m0 < MT; m0 += GM) {
                                float v[NV];
#pragma unroll
                                for (int q = 0; q < NV; ++q) v[q] = g * hprev[m0 + (q >> 4)][q & 15];
                                int cnt = NV;
#pragma unroll
                                for (int d = 16; d >= 1; d >>= 1) {
                                    if (cnt >= 2) {
                                        const int hn = cnt / 2;
                                        const bool up = (lj & d) != 0;
#pragma unroll
                                        for (int i = 0; i < NV / 2; ++i) {
                                            if (i < hn) {
                                                // both halves in named registers first: otherwise the select is folded
                                                // into a lane-varying index into v[] (a 32-way compare/select chain)
                                                float lo = v[i], hi = v[i + hn];
                                                asm volatile("" : "+v"(lo), "+v"(hi));
                                                const float keep = up ? hi : lo;
                                                const float send = up ? lo : hi;
                                                v[i] = keep + __shfl_xor(send, d);
                                            }
                                        }
                                        cnt = hn;
                                    } else {
                                        v[0] += __shfl_xor(v[0], d);
                                    }
                                }
                                if (NV == 32) {
                                    orow[32 * (m0 + (lj >> 4)) + nca_rho(lj & 15) + 4 * lh] += v[0];
                                } else if ((lj & 1) == 0) {          // NV == 16: index lj>>1, complete in both lanes of a pair
                                    orow[32 * m0 + nca_rho((lj >> 1) & 15) + 4 * lh] += v[0];
                                }
                            }
                            const float gsum = half_sum(lh == 0 ? g : 0.f);
                            if (lane == 0) orow[F] += gsum;
                        }
                        // D_{NL-1} = Wo * g masked by ReLU
                        // (stored by the dgrad sweep, under the contraction that consumes it)
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) hprev[m][i] = hprev[m][i] > 0.f ? abl<8>(wo[(lh * MT + m) * 16 + i] * g) : 0.f;
                        }
            };

            // ================= forward (recompute) ==========================================
            // One layer.  FIRST (layer 0, always an encoded-input layer) is its own instantiation, called ahead of the loop over the
            // others: inside ONE loop over all layers the previous layer's output (hprev, 16 MT registers that layer 0 never reads) is
            // live across the loop header and sits in registers through the f64 encoding of layer 0 -- that, not the contractions,
            // is where the f32 kernels spilled 56 - 119 VGPRs (tools/isa_spills.py: the scratch operations sit between the chains).
            if (RECOMP) {
                if constexpr (SKIP) {          // (a net with a skip layer encodes inside the loop anyway: one loop over all layers, as before round 6)
                    for (int jj = 0; jj < y.NL; ++jj) {
                        constexpr bool FIRST = false;
#define NCA_LAYER_KIND l.kind
#include "nca_f32_layer.inc"
#undef NCA_LAYER_KIND
                    }
                } else {
                    // (the lambda is defined HERE, not ahead of the branch: a by-reference closure that exists in the SKIP instantiation too --
                    // even unused -- takes the addresses of hprev / raw / cur / si and cost that kernel 306 instead of 119 spilled registers)
                    auto forward_layer = [&](int jj, auto first_c) __attribute__((always_inline)) {
                        constexpr bool FIRST = decltype(first_c)::value;
#define NCA_LAYER_KIND (FIRST ? (int)NCA_IN_ENC : (int)NCA_IN_HID)          /* layer 0 is the encoded-input layer by construction (nca_build_layout); without a skip layer every other one is hidden-width */
#include "nca_f32_layer.inc"
#undef NCA_LAYER_KIND
                    };
                    forward_layer(0, std::true_type{});
                    for (int jj = 1; jj < y.NL; ++jj) forward_layer(jj, std::false_type{});
                }
            }

            if (STORED) {
                // the forward state comes from the store: raw output and the last hidden layer's output in register order
                raw[net] = *rglob;
                const float* hl = hf + (y.K0rows_pad + (y.NL - 1) * F) * 32;
                asm volatile("" : "+v"(hl));
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4e v = __builtin_nontemporal_load(reinterpret_cast<const f32x4e*>(hl + (m * 4 + q) * 256));
                        hprev[m][4 * q] = v[0]; hprev[m][4 * q + 1] = v[1]; hprev[m][4 * q + 2] = v[2]; hprev[m][4 * q + 3] = v[3];
                    }
                last_layer_grads(wos + net * WOS);
            }

            // ================= backward sweep (dgrad) =======================================
            if (BWD) {
                // hprev holds D_{NL-1}.  For jj = NL-1 .. 1:  D_{jj-1} = relu'(H_jj) .* (W_jj^T D_jj)
                for (int jj = y.NL - 1; jj >= 1; --jj) {
                    const int nsi = (si + 1 == a.nstages) ? 0 : si + 1;
                    stage_issue(a.stage[nsi], smem + (cur ^ 1) * BUF, wave, lane);
                    const float* imgl = reinterpret_cast<const float*>(smem + cur * BUF) + lane * MT;
                    f32x16 acc[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = (f32x16)(0.f);
                    int nsi_d = nsi;
                    if (X3) {
                        constexpr int KH = MT >= 2 ? MT : 2 * MT;
                        if (tvalid) x3_store_block<MT, 64>(df + jj * F * 32, hprev);                                                // D_jj
                        x3_sub<MT, KH, 0>(smem + cur * BUF + lane * 16, hprev, acc);
                        if (MT >= 2) {
                            stage_publish_counted<4 * MT>(tvalid);      // the D_jj stores are younger than this sub-stage's DMA
                            cur ^= 1;
                            si = nsi;
                            nsi_d = (si + 1 == a.nstages) ? 0 : si + 1;
                            stage_issue(a.stage[nsi_d], smem + (cur ^ 1) * BUF, wave, lane);
                            x3_sub<MT, KH, KH>(smem + cur * BUF + lane * 16, hprev, acc);
                        }
                    } else {
                        hidden_steps<MT>(imgl, hprev, acc, df + jj * F * 32, tvalid);         // stores D_jj
                    }
                    // mask with the ReLU pattern of layer jj's input (= output of layer jj-1)
                    const float* hh = hf + (y.K0rows_pad + (jj - 1) * F) * 32;
                    asm volatile("" : "+v"(hh));
                    uint2 mv = make_uint2(0u, 0u);
                    const bool bits = STORED || a.mask_layers > 0;           // mask bits at hand (else: re-read the layer input)
                    if (STORED) {
                        const unsigned long long w = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(mglob + (jj - 1) * 512));
                        mv = make_uint2((unsigned)w, (unsigned)(w >> 32));
                    } else if (a.mask_layers > 0) {
                        mv = *reinterpret_cast<const uint2*>(maskbase + ((wave * a.mask_layers + (jj - 1)) * 64 + lane) * 8);
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const unsigned fld = ((m >> 1) ? mv.y : mv.x) >> (16 * (m & 1));
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (!bits) hv = *reinterpret_cast<const float4*>(hh + (m * 4 + q) * 256);     // fallback: masks did not fit in LDS
                            const bool on[4] = {bits ? ((fld >> (4 * q)) & 1u) != 0u : hv.x > 0.f,
                                                bits ? ((fld >> (4 * q + 1)) & 1u) != 0u : hv.y > 0.f,
                                                bits ? ((fld >> (4 * q + 2)) & 1u) != 0u : hv.z > 0.f,
                                                bits ? ((fld >> (4 * q + 3)) & 1u) != 0u : hv.w > 0.f};
#pragma unroll
                            for (int k = 0; k < 4; ++k) hprev[m][4 * q + k] = on[k] ? abl<8>(acc[m][4 * q + k]) : 0.f;
                        }
                    }
                    if (X3 && MT >= 2) stage_publish();           // (the second sub-stage's DMA was issued after the D_jj stores)
                    else stage_publish_counted<4 * MT>(tvalid);   // D_jj stores
                    cur ^= 1;
                    si = nsi_d;
                }
                if (tvalid) {     // D_0 has no consumer in this kernel: stored here, drains under the next net / tile
                    float* dd = df;
                    asm volatile("" : "+v"(dd));
                    if constexpr ((NCA_ABL & 512) != 0) x3_store_block<MT, 64>(dd, hprev);
                    else
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            store_quad(dd + (m * 4 + q) * 256, abl<64>(hprev[m][4 * q]), abl<64>(hprev[m][4 * q + 1]), abl<64>(hprev[m][4 * q + 2]), abl<64>(hprev[m][4 * q + 3]));
                }
            }
        }  // nets

        // ================= epilogue =========================================================
        if (!BWD) {
            if (a.mode == NCA_MODE_RAYS && !a.raw_only) {
                // render_volume_density[_composite] (model_helpers.py:72-97)
                double term;
                if (a.single) {
                    const float sa = act_fwd(a.act, raw[0]);
                    if (valid && lh == 0) a.sig_s[n] = sa;
                    term = ((double)sa * a.dists[smp]) * (double)a.scale;
                } else {
                    const float ss = __fmul_rn(act_fwd(a.act, raw[0]), a.scale);
                    const float sd = __fmul_rn(act_fwd(a.act, raw[1]), a.scale);
                    if (valid && lh == 0) { a.sig_s[n] = ss; a.sig_d[n] = sd; }
                    term = (double)__fadd_rn(ss, sd) * a.dists[smp];
                }
                if (!(valid && lh == 0)) term = 0.0;
                term = half_sum(term);
                if (lane == 0 && tvalid) a.part[tile] = term;
            } else {
                if (valid && lh == 0) a.raw_out[n] = raw[0];
            }
        }
    }  // tile groups

    if (BWD) {
        // per-workgroup output-layer gradient partials, fixed wave order
        __syncthreads();
        for (int i = tid; i < 2 * (F + 1); i += NCA_NT) {
            float s = 0.f;
            for (int w = 0; w < NCA_WAVES; ++w) s += osum[(w * 2) * (F + 1) + i];
            float* dst = a.oslab + (int64_t)blockIdx.x * 2 * (F + 1) + i;
            *dst = a.accumulate ? *dst + s : s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// wgrad: slab[q] (+)= D[rows o][cols n in split q] * H[rows i][cols n]^T
// ------------------------------------------------------------------------------------------
#define WG_PITCH 33
__global__ __launch_bounds__(256) void nca_wgrad_f32(const NcaWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                 // [2][128][33]
    float* Bs = As + 2 * 128 * WG_PITCH;                        // [2][128][33]
    const NcaWgradJob job = a.job[blockIdx.y];
    const int q = blockIdx.x;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int F = job.F, MT = F / 32;
    const int brows = job.b_rows_pad;                            // multiple of 32, <= 128
    const int CT = brows / 32;
    // D blocks come from the D region of this launch, H / input blocks from the H region (the same buffer in the
    // recompute backward; the whole batch's store after a storing forward)
    const float* __restrict__ Ag = a.scratch + job.d_row0 * 32;      // + tile * rows_total * 32
    const float* __restrict__ Bg = a.scratch_b + (a.tile0_b * a.rows_total_b + job.b_row0) * 32;
    const int64_t tstride = a.rows_total * 32, tstride_b = a.rows_total_b * 32;

    const int64_t ntile = a.ntiles;
    const int64_t per = (ntile + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = (int64_t)q * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x16)(0.f);
    float bsum = 0.f;

    const int lrow = tid >> 3, lc4 = tid & 7;                    // loader: rows lrow + 32 i, float4 chunk lc4
    float4 pa[4], pb[4];
    // An operand tile is ONE contiguous block of rows x 128 B, read as 256 threads x float4 per 32 rows.  Hidden blocks
    // (every D, every H but the encoded input) are in the fused kernel's register order: float4 #(4 m + g) * 64 + l
    // holds rows 32 m + 8 g + 4 (l >> 5) + 0..3 of sample l & 31; the encoded-input block is row-major [row][32].
    const bool bfrag = job.b_frag != 0;
    auto issue = [&](int64_t t) {
        const float* at = Ag + t * tstride + tid * 4;
        const float* bt = Bg + t * tstride_b + tid * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4e va = 32 * i < F ? __builtin_nontemporal_load(reinterpret_cast<const f32x4e*>(at + i * 1024)) : (f32x4e){0.f, 0.f, 0.f, 0.f};
            const f32x4e vb = 32 * i < brows ? __builtin_nontemporal_load(reinterpret_cast<const f32x4e*>(bt + i * 1024)) : (f32x4e){0.f, 0.f, 0.f, 0.f};
            pa[i] = make_float4(va[0], va[1], va[2], va[3]);
            pb[i] = make_float4(vb[0], vb[1], vb[2], vb[3]);
        }
    };
    const int fr = 8 * (tid >> 6) + 4 * ((tid >> 5) & 1), fc = tid & 31;      // fragment layout: first row within the row tile, sample
    auto commit = [&](int buf) {
        float* ad = As + buf * 128 * WG_PITCH;
        float* bd = Bs + buf * 128 * WG_PITCH;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float* x = ad + (32 * i + fr) * WG_PITCH + fc;
            x[0] = pa[i].x; x[WG_PITCH] = pa[i].y; x[2 * WG_PITCH] = pa[i].z; x[3 * WG_PITCH] = pa[i].w;
            if (bfrag) {
                float* z = bd + (32 * i + fr) * WG_PITCH + fc;
                z[0] = pb[i].x; z[WG_PITCH] = pb[i].y; z[2 * WG_PITCH] = pb[i].z; z[3 * WG_PITCH] = pb[i].w;
            } else {
                float* z = bd + (lrow + 32 * i) * WG_PITCH + lc4 * 4;
                z[0] = pb[i].x; z[1] = pb[i].y; z[2] = pb[i].z; z[3] = pb[i].w;
            }
        }
    };

    int buf = 0;
    if (t0 < t1) { issue(t0); commit(0); }
    __syncthreads();
    for (int64_t t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        if (more) issue(t + 1);
        if (wave < MT) {
            const float* ar = As + buf * 128 * WG_PITCH + (32 * wave + lr) * WG_PITCH + lh;
            const float* br = Bs + buf * 128 * WG_PITCH + lr * WG_PITCH + lh;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float av = ar[2 * s];
                bsum += av;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < CT) {
                        const float bv = br[c * 32 * WG_PITCH + 2 * s];
                        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[c], 0, 0, 0);
                    }
                }
            }
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    if (wave < MT) {
        float* slab = a.slab + (int64_t)q * a.slab_stride;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c < CT) {
                const int colb = 32 * c + lr;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int o = 32 * wave + nca_rho(i) + 4 * lh;
                    float* dst = nullptr;
                    if (colb < job.ncols_w) dst = slab + job.out_off + (int64_t)o * job.out_ld + job.out_col0 + colb;
                    else if (colb < job.ncols_w + job.P) dst = slab + job.onehot_off + o * job.P + (colb - job.ncols_w);
                    if (dst) *dst = a.accumulate ? *dst + acc[c][i] : acc[c][i];
                }
            }
        }
        if (job.bias_off >= 0) {
            bsum += __shfl_xor(bsum, 32);
            if (lh == 0) {
                float* dst = slab + job.bias_off + 32 * wave + lr;
                *dst = a.accumulate ? *dst + bsum : bsum;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// wgrad, f32 accuracy on the bf16 matrix cores.  Every f32 operand x is split EXACTLY into three bf16 pieces
// x = x1 + x2 + x3 (8 + 8 + 8 significant bits, by truncation and exact subtraction), and D * H^T is accumulated in f32
// from the six piece products of weight <= 2^-16 (x1 y1, x1 y2, x2 y1, x1 y3, x3 y1, x2 y2); the dropped ones are below
// 2^-24 of |x||y|, i.e. under the rounding of an f32 FMA chain (measured against f64 on random data: 6e-8 vs 4e-7 for
// plain f32).  Products of bf16 pieces are exact in f32, so the result does not depend on bf16 rounding anywhere.
// Six 32-cycle v_mfma_f32_32x32x16_bf16 replace eight 64-cycle v_mfma_f32_32x32x2_f32: 2.7x less matrix-pipe time.
// The split happens once per element while the operand tile is staged into LDS (three bf16 planes [feature][sample],
// 80 B pitch = conflict-free 16 B fragment reads); same jobs, slabs and tile walk as nca_wgrad_f32.
// ------------------------------------------------------------------------------------------
typedef __bf16 w3_bf16x8 __attribute__((ext_vector_type(8)));
#define W3_PITCH 80
#define W3_PLANE (128 * W3_PITCH)
#define W3_OPER (3 * W3_PLANE)

typedef float w3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 w3_bf16x2 __attribute__((ext_vector_type(2)));
// two f32 -> three bf16x2 dwords, x = x1 + x2 + x3 exactly: round to bf16 (v_cvt_pk_bf16_f32), subtract exactly
// (v_pk_add_f32; the residual of a rounding is representable), twice; the third residual has <= 8 significant bits
__device__ __forceinline__ void split3(w3_f32x2 v, uint32_t (&p)[3]) {
    const w3_bf16x2 q1 = __builtin_convertvector(v, w3_bf16x2);
    const w3_f32x2 r1 = v - __builtin_convertvector(q1, w3_f32x2);
    const w3_bf16x2 q2 = __builtin_convertvector(r1, w3_bf16x2);
    const w3_f32x2 r2 = r1 - __builtin_convertvector(q2, w3_f32x2);
    const w3_bf16x2 q3 = __builtin_convertvector(r2, w3_bf16x2);
    p[0] = __builtin_bit_cast(uint32_t, q1);
    p[1] = __builtin_bit_cast(uint32_t, q2);
    p[2] = __builtin_bit_cast(uint32_t, q3);
}

template <int CT>      // 32-row tiles of the H block: one instantiation each keeps the MFMA section free of branches
__device__ __forceinline__ void wgrad3_body(const NcaWgradArgs& a, const NcaWgradJob& job, char* smem) {
    char* As = smem;                       // D tile: 3 planes [128 features][32 samples] bf16
    char* Bs = smem + W3_OPER;             // H tile
    const int q = blockIdx.x;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    const int F = job.F, MT = F / 32;
    const float* __restrict__ Ag = a.scratch + job.d_row0 * 32;
    const float* __restrict__ Bg = a.scratch_b + (a.tile0_b * a.rows_total_b + job.b_row0) * 32;
    const int64_t tstride = a.rows_total * 32, tstride_b = a.rows_total_b * 32;

    const int64_t ntile = a.ntiles;
    const int64_t per = (ntile + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = (int64_t)q * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;

    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[c] = (f32x16)(0.f);
    float bs[4][4];                        // bias gradient: this thread's four rows of every row tile, summed over its sample
#pragma unroll
    for (int i = 0; i < 4; ++i) { bs[i][0] = bs[i][1] = bs[i][2] = bs[i][3] = 0.f; }

    // Loader: 256 threads x float4 per 32 rows, 1 KiB per wave instruction.  Hidden blocks (every D, every H but the
    // encoded input) are in the fused kernels' register order: float4 #(4 m + g) * 64 + l holds rows 32 m + 8 g + 4 (l >> 5)
    // + 0..3 of sample l & 31; the encoded-input block is row-major [row][32 samples].
    // Samples are paired (s, s + 16) into bf16x2 dwords -- the contraction index may be permuted as long as both operands
    // agree -- because the partners then sit in lanes l and l + 32 and ONE v_permlane32_swap moves two registers; and the
    // lanes of one half (0..15 / 16..31) are given rows 4 apart, which with the 80 B pitch lands their dwords in disjoint
    // bank halves (rows 2 apart, as a lane ^ 1 exchange would give, cost 11 % of the kernel in bank conflicts).
    const bool bfrag = job.b_frag != 0;
    const int up = lane >> 5;                                   // 0: keeps rows +0, +1 / samples 4c, 4c + 1; 1: rows +2, +3 / ...
    const int hq = (lane >> 4) & 1, fj = lane & 15;             // fragment order: row half, sample pair (fj, fj + 16)
    const int frag_src = (wave * 64 + 32 * hq + fj + 16 * up) * 4;              // floats inside a 32-row tile
    const int frag_dst = (8 * wave + 4 * hq + 2 * up) * W3_PITCH + fj * 4;      // bytes inside a plane
    const int rc = lane & 3, rrow = 8 * wave + ((lane >> 2) & 7);              // row-major: float4 rc + 4 up of row rrow
    const int rows_src = (rrow * 8 + rc + 4 * up) * 4;
    const int rows_dst = rrow * W3_PITCH + (4 * rc + 2 * up) * 4;
    struct Regs { f32x4e a[4], b[4]; };
    Regs R0;
    auto issue = [&](int64_t t, Regs& r) {
        const float* at = Ag + t * tstride + frag_src;
        const float* bt = Bg + t * tstride_b + (bfrag ? frag_src : rows_src);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < MT) r.a[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4e*>(at + i * 1024));
            if (i < CT) r.b[i] = __builtin_nontemporal_load(reinterpret_cast<const f32x4e*>(bt + i * 1024));
        }
    };
    // (x, z) and (y, w) of this lane's float4 against the partner lane's: lanes < 32 end up with elements 0, 1 of both,
    // lanes >= 32 with elements 2, 3 -- as (own sample group, the one 16 samples on) in both cases
    auto swap_split = [&](const f32x4e& v, uint32_t (&p0)[3], uint32_t (&p1)[3]) {
        const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
        const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
        split3((w3_f32x2){__uint_as_float(s0[0]), __uint_as_float(s0[1])}, p0);
        split3((w3_f32x2){__uint_as_float(s1[0]), __uint_as_float(s1[1])}, p1);
    };
    auto put_frag = [&](char* base, int i, const f32x4e& v) {     // two rows x one sample pair
        uint32_t p0[3], p1[3];
        swap_split(v, p0, p1);
        char* dst = base + 32 * i * W3_PITCH + frag_dst;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            *reinterpret_cast<uint32_t*>(dst + pl * W3_PLANE) = p0[pl];
            *reinterpret_cast<uint32_t*>(dst + pl * W3_PLANE + W3_PITCH) = p1[pl];
        }
    };
    auto put_rows = [&](char* base, int i, const f32x4e& v) {     // one row x two neighbouring sample pairs
        uint32_t p0[3], p1[3];
        swap_split(v, p0, p1);
        char* dst = base + 32 * i * W3_PITCH + rows_dst;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<uint2*>(dst + pl * W3_PLANE) = make_uint2(p0[pl], p1[pl]);
    };
    auto commit = [&](const Regs& r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < MT) {
                bs[i][0] += r.a[i][0]; bs[i][1] += r.a[i][1]; bs[i][2] += r.a[i][2]; bs[i][3] += r.a[i][3];
                put_frag(As, i, r.a[i]);
            }
            if (i < CT) {
                if (bfrag) put_frag(Bs, i, r.b[i]);
                else put_rows(Bs, i, r.b[i]);
            }
        }
    };
    auto compute = [&]() {
        if (wave < MT) {
            const char* ar = As + (32 * wave + lr) * W3_PITCH + lh * 16;
            const char* br = Bs + lr * W3_PITCH + lh * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                w3_bf16x8 A[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) A[p] = *reinterpret_cast<const w3_bf16x8*>(ar + p * W3_PLANE + ks * 32);
#pragma unroll
                for (int c0 = 0; c0 < CT; c0 += 2) {          // two column tiles at a time: consecutive MFMAs alternate accumulators
                    w3_bf16x8 B[2][3];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            if (c0 + c < CT) B[c][p] = *reinterpret_cast<const w3_bf16x8*>(br + p * W3_PLANE + (c0 + c) * 32 * W3_PITCH + ks * 32);
                    // small products first
#define W3_MMA(I, J)                                                                                                   \
                    _Pragma("unroll") for (int c = 0; c < 2; ++c)                                                   \
                        if (c0 + c < CT) acc[c0 + c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[I], B[c][J], acc[c0 + c], 0, 0, 0);
                    W3_MMA(1, 1) W3_MMA(2, 0) W3_MMA(0, 2) W3_MMA(1, 0) W3_MMA(0, 1) W3_MMA(0, 0)
#undef W3_MMA
                }
            }
        }
    };

    // tile t sits in LDS while t + 1 is in flight (two tiles in flight were measured: no gain, the kernel is bound by
    // the SIMDs' MFMA + VALU issue time, not by latency)
    if (t0 < t1) { issue(t0, R0); commit(R0); }
    __syncthreads();
    for (int64_t t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        if (more) issue(t + 1, R0);
        compute();
        __syncthreads();                 // every wave is done with the planes
        if (more) commit(R0);
        __syncthreads();
    }

    if (wave < MT) {
        float* slab = a.slab + (int64_t)q * a.slab_stride;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int colb = 32 * c + lr;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int o = 32 * wave + nca_rho(i) + 4 * lh;
                float* dst = nullptr;
                if (colb < job.ncols_w) dst = slab + job.out_off + (int64_t)o * job.out_ld + job.out_col0 + colb;
                else if (colb < job.ncols_w + job.P) dst = slab + job.onehot_off + o * job.P + (colb - job.ncols_w);
                if (dst) *dst = a.accumulate ? *dst + acc[c][i] : acc[c][i];
            }
        }
    }
    if (job.bias_off >= 0) {
        float* slab = a.slab + (int64_t)q * a.slab_stride;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sum = bs[i][j];                          // over the 32 samples: lane bits 0..3 and 5
                sum += __shfl_xor(sum, 32); sum += __shfl_xor(sum, 8); sum += __shfl_xor(sum, 4);
                sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 1);
                if (fj == 0 && up == 0 && i < MT) {
                    float* dst = slab + job.bias_off + 32 * i + 8 * wave + 4 * hq + j;
                    *dst = a.accumulate ? *dst + sum : sum;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void nca_wgrad_f32x3(const NcaWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const NcaWgradJob job = a.job[blockIdx.y];
    switch (job.b_rows_pad / 32) {
        case 1: wgrad3_body<1>(a, job, smem); break;
        case 2: wgrad3_body<2>(a, job, smem); break;
        case 3: wgrad3_body<3>(a, job, smem); break;
        default: wgrad3_body<4>(a, job, smem); break;
    }
}

// ------------------------------------------------------------------------------------------
// reduce: natural flat gradients from the split slabs (fixed summation order)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void reduce_body(const NcaReduceArgs& a, int64_t blk) {
    const int64_t e = blk * 256 + threadIdx.x;
    if (e >= a.n_total) return;
    const int net = e < a.n_params[0] ? 0 : 1;
    const int64_t le = net == 0 ? e : e - a.n_params[0];
    const NcaReduceNet& rn = a.net[net];
    float* out = rn.grads + le;
    if (le >= rn.wo_off || le < rn.lat_count) return;       // Wo / bo and the latents: nca_reduce_small_f32 (long sums, few outputs)
    // eight independent partial sums (splits q .. q+7 mod 8) keep several loads in flight;
    // the combination order is fixed, so the result is still bit-reproducible
    float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const float* sp = a.slab + rn.slab_off + le;
    const int64_t stride = a.slab_stride;
    // rows of this column: the rebuilding jobs of the last F-wide layer may run over more splits than the others
    // (two explicit ranges, W then b, as the scaling below: nothing here assumes the bias follows the weight in the flat buffer)
    const int64_t F2t = (int64_t)rn.F * rn.tl_K;
    const bool tail_w = rn.tail_from_sums && le >= rn.tl_w_off && le < rn.tl_w_off + F2t;
    const bool tail_b = rn.tail_from_sums && le >= rn.tl_b_off && le < rn.tl_b_off + rn.F;
    const bool tail_col = tail_w || tail_b;
    int nsum = tail_col ? rn.n_split : rn.n_split_std;
    int q = 0;
    for (; q + 8 <= nsum; q += 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u) s8[u] += sp[(int64_t)(q + u) * stride];
    }
    for (; q < nsum; ++q) s8[0] += sp[(int64_t)q * stride];
    float r = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
    // the last F-wide layer under tail_from_sums: leave the sum over the splits in slab row 0 (this thread is the only reader of its
    // column) -- nca_reduce_small_f32, launched next, forms dWo from S and s summed over the splits and reads 129 values per output
    // instead of 129 x n_split (it was 83 us of one wave per output walking the slabs: as long at 1 024 rays per step as at 65 536)
    if (tail_col) a.slab[rn.slab_off + le] = r;
    // the last F-wide layer's sums were formed without the factor Wo[f] of their output row (nca_layout.hpp)
    if (tail_w) r *= rn.params[rn.wo_off + (le - rn.tl_w_off) / rn.tl_K];
    else if (tail_b) r *= rn.params[rn.wo_off + (le - rn.tl_b_off)];
    *out = r;
}

// The few outputs with long sums -- Wo / bo (over the fused kernel's per-workgroup partials) and the time latents (over
// the F rows of the one-hot block) -- one WAVE per output: lane-strided partial sums, then a fixed xor tree.
__global__ __launch_bounds__(256) void nca_reduce_small_f32(const NcaReduceArgs a) {
    const int lane = threadIdx.x & 63;
    int64_t id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int net = 0; net < 2; ++net) {
        const NcaReduceNet& rn = a.net[net];
        if (!rn.grads) continue;
        const int64_t cnt = rn.lat_count + rn.F + 1;
        if (id >= cnt) { id -= cnt; continue; }
        float s = 0.f;
        float* out;
        if (id < rn.lat_count) {
            // time_latents[p][t] = sum_f W0[f][Kenc+t] * Dsum[f][p]; Dsum was reduced over the splits into slab 0 by
            // nca_onehot_sum_f32 (launched before this kernel)
            const int pp = (int)(id / rn.T), t = (int)(id % rn.T);
            for (int f = lane; f < rn.F; f += 64) s = fmaf(rn.params[rn.w0_off + f * rn.K0 + rn.Kenc + t], a.slab[rn.onehot_off + f * rn.P + pp], s);
            out = rn.grads + id;
        } else {
            const int k = (int)(id - rn.lat_count);           // 0..F (F = bias)
            if (rn.tail_from_sums && k < rn.F) {
                // dWo[k] = sum_kk bf16(W[k][kk]) S[k][kk] + b[k] s[k], S and s summed over the splits (nca_layout.hpp); the forward
                // multiplied with the bf16-rounded weights, so those are the ones the identity holds for
                // (S[k][.] and s[k] summed over the splits sit in slab row 0: nca_reduce_f32 left them there)
                const float* wrow = rn.params + rn.tl_w_off + (int64_t)k * rn.tl_K;
                for (int kk = lane; kk < rn.tl_K; kk += 64) {
                    const float wq = __uint_as_float(((__float_as_uint(wrow[kk]) + 0x7fffu + ((__float_as_uint(wrow[kk]) >> 16) & 1u)) & 0xffff0000u));
                    s = fmaf(wq, a.slab[rn.slab_off + rn.tl_w_off + (int64_t)k * rn.tl_K + kk], s);
                }
                if (lane == 0) s = fmaf(rn.params[rn.tl_b_off + k], a.slab[rn.slab_off + rn.tl_b_off + k], s);
            } else {
                for (int w = lane; w < rn.n_wg; w += 64) s += a.oslab[(int64_t)w * a.oslab_stride + net * (rn.F + 1) + k];
            }
            out = rn.grads + rn.wo_off + k;
        }
        s += __shfl_xor(s, 32); s += __shfl_xor(s, 16); s += __shfl_xor(s, 8); s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
        if (lane == 0) *out = s;
        return;
    }
}

// Dsum[f][p] = sum over splits of the one-hot block, written in place into slab 0 (fixed order)
__device__ __forceinline__ void onehot_body(const NcaReduceArgs& a, int64_t blk) {
    const int64_t e = blk * 256 + threadIdx.x;
    for (int net = 0; net < 2; ++net) {
        const NcaReduceNet& rn = a.net[net];
        const int64_t cnt = (int64_t)rn.F * rn.P;
        if (e >= cnt || !rn.grads) continue;
        float* p0 = a.slab + rn.onehot_off + e;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        int q = 0;
        for (; q + 4 <= rn.n_split_std; q += 4) {         // (the one-hot block belongs to the layer-0 jobs)
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += p0[(int64_t)(q + u) * a.slab_stride];
        }
        for (; q < rn.n_split_std; ++q) s4[0] += p0[(int64_t)q * a.slab_stride];
        *p0 = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
}

// ONE launch for the three independent sums of a backward's tail: workgroups [0, nb_red) sum the split slabs into the natural gradients,
// [nb_red, nb_red + nb_hot) sum the one-hot block over the splits, the rest (a one-chunk backward from a store) add the tile records'
// output-bias sums to oslab.  nca_reduce_small_f32, launched next, reads what all three left.  (At the reference's default batch a step is
// ~0.7 ms: three launches of ~5 us each were 2 % of it.)
__global__ __launch_bounds__(256) void nca_reduce_f32(const NcaReduceArgs a, int nb_red, int nb_hot) {
    __shared__ float part[256];
    const int b = blockIdx.x;
    if (b < nb_red) reduce_body(a, b);
    else if (b < nb_red + nb_hot) onehot_body(a, b - nb_red);
    else nca_tile_record_sum(a.rec_region, a.rec_tile_bytes, a.rec_off, a.rec_ntiles, a.rec_net0, a.rec_net1, a.rec_F, a.rec_oslab, b - nb_red - nb_hot, a.rec_nwg, part);
}

// pix[r] = I0[r] - sum_c part[r][c]   (model_helpers.py:82 / 95)
__global__ void nca_pix_f32(int64_t R, int nchunk, const float* __restrict__ I0, const double* __restrict__ part, double* __restrict__ pix) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double s = 0.0;
    for (int c = 0; c < nchunk; ++c) s += part[r * nchunk + c];
    pix[r] = (double)I0[r] - s;
}

// ------------------------------------------------------------------------------------------
// launchers (called from nca_api.cpp)
// ------------------------------------------------------------------------------------------
template <int F, int MODE, bool X3, bool SKIP>
static hipError_t launch_fused_mode_s(const NcaFusedArgs& a_in, int grid, hipStream_t st) {
    constexpr bool bwd = MODE == NCA_KM_BWD || MODE == NCA_KM_BWD_STORED;
    NcaFusedArgs a = a_in;
    // constant area sized to the latents actually present (the cap of 2048 floats per net is rarely needed) ...
    int lat = 0;
    for (int n = 0; n < a.nnets; ++n) { const int v = a.net[n].lay.P * a.net[n].lay.T; lat = v > lat ? v : lat; }
    if (lat > NCA_CONST_LAT) return hipErrorInvalidValue;
    a.const_net_floats = NCA_CONST_WIN + NCA_CONST_FOUR + ((lat + 15) & ~15);
    size_t lds = 2 * FusedCfg<F>::BUF_BYTES + 2 * (size_t)a.const_net_floats * 4 + (bwd ? NCA_WAVES * 2 * (F + 1) * sizeof(float) : 0);
    // ... which leaves room for the ReLU masks of the recomputed layers (8 B per lane and layer): the dgrad sweep
    // then reads no H back from the scratch.  If they do not fit, it falls back to re-reading H.
    a.mask_layers = 0;
    if (MODE == NCA_KM_BWD) {
        int ml = 0;
        for (int n = 0; n < a.nnets; ++n) ml = a.net[n].lay.NL - 1 > ml ? a.net[n].lay.NL - 1 : ml;
        const size_t need = (size_t)NCA_WAVES * ml * 512;
        if (ml > 0 && lds + need <= 160 * 1024) { a.mask_layers = ml; lds += need; }
    }
    if (MODE == NCA_KM_BWD_STORED) lds += 2 * (2 * FusedCfg<F>::MT * 16 + 16) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_fused_f32<F, MODE, X3, SKIP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((nca_fused_f32<F, MODE, X3, SKIP>), dim3(grid), dim3(NCA_NT), lds, st, a);
    return hipGetLastError();
}
template <int F, int MODE, bool X3>
static hipError_t launch_fused_mode(const NcaFusedArgs& a, int grid, hipStream_t st) {
    bool skip = false;
    for (int n = 0; n < a.nnets; ++n)
        for (int j = 0; j < a.net[n].lay.NL; ++j) skip = skip || a.net[n].lay.layer[j].kind == NCA_IN_SKIP;
    return skip ? launch_fused_mode_s<F, MODE, X3, true>(a, grid, st) : launch_fused_mode_s<F, MODE, X3, false>(a, grid, st);
}
template <int F, bool X3>
static hipError_t launch_fused_x(const NcaFusedArgs& a, int kmode, int grid, hipStream_t st) {
    switch (kmode) {
        case NCA_KM_FWD: return launch_fused_mode<F, NCA_KM_FWD, X3>(a, grid, st);
        case NCA_KM_BWD: return launch_fused_mode<F, NCA_KM_BWD, X3>(a, grid, st);
        case NCA_KM_FWD_STORE: return launch_fused_mode<F, NCA_KM_FWD_STORE, X3>(a, grid, st);
        case NCA_KM_BWD_STORED: return launch_fused_mode<F, NCA_KM_BWD_STORED, X3>(a, grid, st);
    }
    return hipErrorInvalidValue;
}
template <int F>
static hipError_t launch_fused_t(const NcaFusedArgs& a, int kmode, int grid, hipStream_t st) {
    for (int n = 1; n < a.nnets; ++n) if (a.net[n].lay.x3 != a.net[0].lay.x3) return hipErrorInvalidValue;
    return a.net[0].lay.x3 ? launch_fused_x<F, true>(a, kmode, grid, st) : launch_fused_x<F, false>(a, kmode, grid, st);
}

hipError_t nca_launch_fused_f32(int F, const NcaFusedArgs& a, int kmode, int grid, hipStream_t st) {
    switch (F) {
        case 32: return launch_fused_t<32>(a, kmode, grid, st);
        case 64: return launch_fused_t<64>(a, kmode, grid, st);
        case 128: return launch_fused_t<128>(a, kmode, grid, st);
    }
    return hipErrorInvalidValue;
}

hipError_t nca_launch_pack_f32(const NcaLayout& y, const float* prm, void* out, hipStream_t st) {
    const int total = (int)(y.packed_bytes / 4u);
    const int grid = (total + 255) / 256;
    hipLaunchKernelGGL(nca_pack_f32, dim3(grid > 1024 ? 1024 : grid), dim3(256), 0, st, y, prm, reinterpret_cast<float*>(out));
    return hipGetLastError();
}

hipError_t nca_launch_pack2_f32(const NcaLayout& ya, const float* prm_a, void* out_a, const NcaLayout& yb, const float* prm_b, void* out_b, hipStream_t st) {
    NcaPack2Args a;
    a.y[0] = ya; a.y[1] = yb; a.prm[0] = prm_a; a.prm[1] = prm_b; a.out[0] = out_a; a.out[1] = out_b;
    const int total = (int)((ya.packed_bytes > yb.packed_bytes ? ya.packed_bytes : yb.packed_bytes) / 4u);
    const int grid = (total + 255) / 256;
    hipLaunchKernelGGL(nca_pack2_f32, dim3(grid > 1024 ? 1024 : grid, 2), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t nca_launch_wgrad_f32(const NcaWgradArgs& a, int nsplit, hipStream_t st) {
    static const bool plain = getenv("NCA_WGRAD_F32") != nullptr && getenv("NCA_WGRAD_F32")[0] == 'p';   // NCA_WGRAD_F32=plain: A/B switch
    if (!plain) {
        const size_t lds3 = 2 * W3_OPER;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_wgrad_f32x3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
        hipLaunchKernelGGL(nca_wgrad_f32x3, dim3(nsplit, a.njobs), dim3(256), lds3, st, a);
        return hipGetLastError();
    }
    const size_t lds = 4 * 128 * WG_PITCH * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_wgrad_f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nca_wgrad_f32, dim3(nsplit, a.njobs), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t nca_launch_reduce_f32(const NcaReduceArgs& a, hipStream_t st) {
    int64_t hot = 0;
    for (int n = 0; n < 2; ++n) if (a.net[n].grads && (int64_t)a.net[n].F * a.net[n].P > hot) hot = (int64_t)a.net[n].F * a.net[n].P;
    const int nb_hot = (int)((hot + 255) / 256), nb_red = (int)((a.n_total + 255) / 256), nb_rec = a.rec_region ? a.rec_nwg : 0;
    hipLaunchKernelGGL(nca_reduce_f32, dim3(nb_red + nb_hot + nb_rec), dim3(256), 0, st, a, nb_red, nb_hot);
    int64_t small = 0;
    for (int n = 0; n < 2; ++n) if (a.net[n].grads) small += a.net[n].lat_count + a.net[n].F + 1;
    if (small > 0) hipLaunchKernelGGL(nca_reduce_small_f32, dim3((int)((small + 3) / 4)), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t nca_launch_pix_f32(int64_t R, int nchunk, const float* I0, const double* part, double* pix, hipStream_t st) {
    const int grid = (int)((R + 255) / 256);
    hipLaunchKernelGGL(nca_pix_f32, dim3(grid), dim3(256), 0, st, R, nchunk, I0, part, pix);
    return hipGetLastError();
}
