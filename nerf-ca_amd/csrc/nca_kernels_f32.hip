// nca_kernels_f32.hip -- gfx950 kernels of the f32 (parity) path.
//
//   nca_pack_f32      natural flat parameters -> MFMA-ordered LDS images
//   nca_fused_f32<F,BWD>  one pass over ray-ordered samples:
//         BWD=false: query point -> positional encoding -> static MLP -> dynamic MLP ->
//                    activation -> per-ray partial sums            (model_helpers.py:115-129)
//         BWD=true : the same recompute, then the backward sweep (dgrad) of each net; layer
//                    inputs H and output gradients D are written feature-major for the wgrad
//   nca_wgrad_f32     dW = D * H^T over the sample axis (split over workgroups), bias sums
//   nca_reduce_f32    fixed-order sum of the split slabs -> natural flat gradients
//   nca_pix_f32       pix = I0 - sum of the per-tile partial ray sums
//
// Data layout: activations are transposed, H[feature][sample]: a wave owns 32 consecutive samples
// of one ray (lane&31 = sample, lane>>5 = k-half), a 32x32 accumulator tile per 32 features.  A
// layer's output registers are the next layer's B operand in place (see nca_layout.hpp).
#include <hip/hip_runtime.h>
#include "nca_kernels.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NCA_HALF_PI_F 1.57079637050628662109375f   // fl32(0.5 * pi), the constant the reference adds
#define NCA_HALF_PI_D 1.57079632679489661923
#define NCA_TWO_PI_F 6.283185482025146484375f      // fl32(2 * pi)

// ------------------------------------------------------------------------------------------
// pack
// ------------------------------------------------------------------------------------------
__global__ void nca_pack_f32(NcaLayout y, const float* __restrict__ prm, float* __restrict__ out) {
    const uint32_t total = y.packed_bytes / 4u;
    for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        float v = 0.f;
        const uint32_t byte = e * 4u;
        for (int j = 0; j < y.NL; ++j) {
            const NcaLayerL& l = y.layer[j];
            if (byte >= l.img_off && byte < l.img_off + l.img_bytes) {
                uint32_t q = (byte - l.img_off) / 4u;
                const uint32_t wcount = (uint32_t)l.ksteps * 64u * (uint32_t)y.MT;
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
                } else if (j == y.NL - 1) {
                    q -= wcount + tail;
                    if (q < tail) {
                        int i = q % 16, m = (q / 16) % y.MT, h = q / (16 * y.MT);
                        v = prm[y.wo_off + 32 * m + nca_rho(i) + 4 * h];
                    } else if (q == tail) {
                        v = prm[y.bo_off];
                    }
                }
            }
            if (l.imgT_bytes && byte >= l.imgT_off && byte < l.imgT_off + l.imgT_bytes) {
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
            const float v = __fmul_rn(__fmul_rn(NCA_TWO_PI_F, p[c]), four[i]);
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
    static constexpr int BUF_BYTES = (IMG_MAX + 255) & ~255;
    static constexpr int PF = (BUF_BYTES + NCA_NT * 16 - 1) / (NCA_NT * 16);       // 16-byte prefetch registers per thread
};

template <int F, bool BWD>
__global__ __launch_bounds__(NCA_NT, 2) void nca_fused_f32(const NcaFusedArgs a) {
    constexpr int MT = FusedCfg<F>::MT;
    constexpr int BUF = FusedCfg<F>::BUF_BYTES;
    constexpr int PF = FusedCfg<F>::PF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // [buf0][buf1][per-wave output-layer gradient sums: WAVES x 2 nets x (F+1)]
    float* osum = reinterpret_cast<float*>(smem + 2 * BUF);

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lj = lane & 31, lh = lane >> 5;

    if (BWD) {
        for (int i = tid; i < NCA_WAVES * 2 * (F + 1); i += NCA_NT) osum[i] = 0.f;
    }

    // stage 0 -> buffer 0
    {
        const uint4* src = reinterpret_cast<const uint4*>(a.stage[0].ptr);
        const int n16 = (int)(a.stage[0].bytes >> 4);
        for (int i = tid; i < n16; i += NCA_NT) reinterpret_cast<uint4*>(smem)[i] = src[i];
    }
    __syncthreads();
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
            const float zz = a.z[ray * a.zs_r + smp];
            if (a.ray_is_f64) {
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
            for (int c = 0; c < 3; ++c) p[c] = a.pts[n * 3 + c];
        }
        int ph = 0;
        if (a.phase) {
            ph = a.mode == NCA_MODE_RAYS ? a.phase[ray * a.ps_r + (int64_t)smp * a.ps_s] : a.phase[n];
        }
        const int64_t col = tl * 32 + lj;   // column in the backward scratch

        float raw[2] = {0.f, 0.f};

#pragma unroll
        for (int net = 0; net < 2; ++net) {
            if (net >= a.nnets) break;
            const NcaNetArgs& na = a.net[net];
            const NcaLayout& y = na.lay;
            int phc = ph < 0 ? 0 : (ph >= y.P ? y.P - 1 : ph);
            const float* lat = y.T > 0 ? na.lat + phc * y.T : nullptr;
            float* const hs = BWD ? a.scratch + na.row0 * a.Nc : nullptr;   // this net's scratch rows

            f32x16 hprev[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) hprev[m] = (f32x16)(0.f);

            // ================= forward (recompute) ==========================================
            for (int jj = 0; jj < y.NL; ++jj) {
                const NcaLayerL& l = y.layer[jj];
                // prefetch the next image into registers
                const int nsi = (si + 1 == a.nstages) ? 0 : si + 1;
                uint4 pf[PF];
                {
                    const uint4* src = reinterpret_cast<const uint4*>(a.stage[nsi].ptr);
                    const int n16 = (int)(a.stage[nsi].bytes >> 4);
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const int idx = tid + i * NCA_NT;
                        pf[i] = idx < n16 ? src[idx] : make_uint4(0, 0, 0, 0);
                    }
                }
                const float* img = reinterpret_cast<const float*>(smem + cur * BUF);
                const float* imgl = img + lane * MT;
                const float* tail = img + l.ksteps * 64 * MT;

                f32x16 acc[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[m][i] = tail[(lh * MT + m) * 16 + i];

                if (l.kind != NCA_IN_HID) {
                    float* const henc = hs;   // rows [0, K0rows_pad)
                    enc_steps(y, p, na.win, na.four, lat, [&](int s, float fa, float fb) {
                        const float bop = lh ? fb : fa;
                        float av[MT];
                        load_a<MT>(imgl + s * 64 * MT, av);
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bop, acc[m], 0, 0, 0);
                        if (BWD && jj == 0 && tvalid) {
                            int ia, ib;
                            nca_enc_pair(y, s, &ia, &ib);
                            const int row = lh ? ib : ia;
                            if (row >= 0) henc[(int64_t)row * a.Nc + col] = bop;
                        }
                    });
                    if (BWD && jj == 0 && y.P > 0 && tvalid) {
                        // one-hot phase rows: their "weight gradient" is sum_n [phase_n = p] D0[:, n]
                        for (int pp = lh; pp < y.P; pp += 2) henc[(int64_t)(y.K0 + pp) * a.Nc + col] = (pp == phc) ? 1.f : 0.f;
                    }
                }
                if (l.kind != NCA_IN_ENC) {
                    const float* ih = imgl + l.ksteps_enc * 64 * MT;
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float av[MT];
                            load_a<MT>(ih + (16 * t + i) * 64 * MT, av);
                            const float bop = hprev[t][i];
#pragma unroll
                            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bop, acc[m], 0, 0, 0);
                        }
                }
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int i = 0; i < 16; ++i) hprev[m][i] = fmaxf(acc[m][i], 0.f);

                if (BWD && jj + 1 < y.NL && tvalid) {
                    // input of layer jj+1, feature-major (waves past the last tile write nothing)
                    float* hh = hs + (int64_t)(y.K0rows_pad + jj * F) * a.Nc + col;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 16; ++i) hh[(int64_t)(32 * m + nca_rho(i) + 4 * lh) * a.Nc] = hprev[m][i];
                }

                if (jj == y.NL - 1) {
                    // output layer F -> 1 from the image tail (model/CPPN.py:108)
                    const float* wo = tail + 2 * MT * 16;
                    float part = 0.f;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 16; ++i) part = fmaf(wo[(lh * MT + m) * 16 + i], hprev[m][i], part);
                    part += __shfl_xor(part, 32);
                    raw[net] = part + wo[2 * MT * 16];

                    if (BWD) {
                        // gradient wrt the raw output of this net
                        float g;
                        if (a.mode == NCA_MODE_RAYS) {
                            const float* gs = net == 0 ? a.g_sig_s : a.g_sig_d;
                            const double gsig = gs ? (double)gs[n] : 0.0;
                            const double gp = a.g_pix[ray] * a.dists[smp];
                            const double dsig = a.single ? (gsig - gp * (double)a.scale) : (gsig - gp) * (double)a.scale;
                            g = (float)dsig * act_bwd(a.act, raw[net]);
                        } else {
                            g = a.g_raw[n];
                        }
                        if (!valid) g = 0.f;
                        // dWo[f] = sum_n g H[f][n] : reduce-scatter over the 32 lanes of each half
                        float v[MT * 16];
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int i = 0; i < 16; ++i) v[m * 16 + i] = g * hprev[m][i];
                        {
                            constexpr int NV = MT * 16;
                            int cnt = NV;
#pragma unroll
                            for (int d = 16; d >= 1; d >>= 1) {
                                if (cnt >= 2) {
                                    const int hn = cnt / 2;
                                    const bool up = (lj & d) != 0;
#pragma unroll
                                    for (int i = 0; i < NV / 2; ++i) {
                                        if (i < hn) {
                                            const float keep = up ? v[i + hn] : v[i];
                                            const float send = up ? v[i] : v[i + hn];
                                            v[i] = keep + __shfl_xor(send, d);
                                        }
                                    }
                                    cnt = hn;
                                } else {
                                    v[0] += __shfl_xor(v[0], d);
                                }
                            }
                            // lane now owns `per` consecutive flat indices starting at lane-dependent base
                            constexpr int per = NV >= 32 ? NV / 32 : 1;
                            float* orow = osum + (wave * 2 + net) * (F + 1);
                            if (NV >= 32) {
#pragma unroll
                                for (int e = 0; e < per; ++e) {
                                    const int idx = lj * per + e;
                                    const int f = 32 * (idx >> 4) + nca_rho(idx & 15) + 4 * lh;
                                    orow[f] += v[e];
                                }
                            } else {
                                // NV == 16 (F = 32): after 4 halvings lanes (lj>>1) own index lj>>1, fully summed after d=1
                                if ((lj & 1) == 0) {
                                    const int idx = lj >> 1;
                                    const int f = nca_rho(idx & 15) + 4 * lh;
                                    orow[f] += v[0];
                                }
                            }
                            const float gsum = half_sum(lh == 0 ? g : 0.f);
                            if (lane == 0) orow[F] += gsum;
                        }
                        // D_{NL-1} = Wo * g masked by ReLU
                        float* dd = hs + (int64_t)(y.K0rows_pad + (y.NL - 1) * F + (y.NL - 1) * F) * a.Nc + col;
#pragma unroll
                        for (int m = 0; m < MT; ++m)
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const float dv = hprev[m][i] > 0.f ? wo[(lh * MT + m) * 16 + i] * g : 0.f;
                                hprev[m][i] = dv;
                                if (tvalid) dd[(int64_t)(32 * m + nca_rho(i) + 4 * lh) * a.Nc] = dv;
                            }
                    }
                }

                // publish the prefetched image into the other buffer
                {
                    uint4* dst = reinterpret_cast<uint4*>(smem + (cur ^ 1) * BUF);
                    const int n16 = (int)(a.stage[nsi].bytes >> 4);
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const int idx = tid + i * NCA_NT;
                        if (idx < n16) dst[idx] = pf[i];
                    }
                }
                __syncthreads();
                cur ^= 1;
                si = nsi;
            }

            // ================= backward sweep (dgrad) =======================================
            if (BWD) {
                // hprev holds D_{NL-1}.  For jj = NL-1 .. 1:  D_{jj-1} = relu'(H_jj) .* (W_jj^T D_jj)
                for (int jj = y.NL - 1; jj >= 1; --jj) {
                    const int nsi = (si + 1 == a.nstages) ? 0 : si + 1;
                    uint4 pf[PF];
                    {
                        const uint4* src = reinterpret_cast<const uint4*>(a.stage[nsi].ptr);
                        const int n16 = (int)(a.stage[nsi].bytes >> 4);
#pragma unroll
                        for (int i = 0; i < PF; ++i) {
                            const int idx = tid + i * NCA_NT;
                            pf[i] = idx < n16 ? src[idx] : make_uint4(0, 0, 0, 0);
                        }
                    }
                    const float* imgl = reinterpret_cast<const float*>(smem + cur * BUF) + lane * MT;
                    f32x16 acc[MT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) acc[m] = (f32x16)(0.f);
#pragma unroll
                    for (int t = 0; t < MT; ++t)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            float av[MT];
                            load_a<MT>(imgl + (16 * t + i) * 64 * MT, av);
                            const float bop = hprev[t][i];
#pragma unroll
                            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m], bop, acc[m], 0, 0, 0);
                        }
                    // mask with the stored input of layer jj (= output of layer jj-1), store D_{jj-1}
                    const float* hh = hs + (int64_t)(y.K0rows_pad + (jj - 1) * F) * a.Nc + col;
                    float* dd = hs + (int64_t)(y.K0rows_pad + (y.NL - 1) * F + (jj - 1) * F) * a.Nc + col;
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int64_t ro = (int64_t)(32 * m + nca_rho(i) + 4 * lh) * a.Nc;
                            const float dv = hh[ro] > 0.f ? acc[m][i] : 0.f;
                            hprev[m][i] = dv;
                            if (tvalid) dd[ro] = dv;
                        }
                    {
                        uint4* dst = reinterpret_cast<uint4*>(smem + (cur ^ 1) * BUF);
                        const int n16 = (int)(a.stage[nsi].bytes >> 4);
#pragma unroll
                        for (int i = 0; i < PF; ++i) {
                            const int idx = tid + i * NCA_NT;
                            if (idx < n16) dst[idx] = pf[i];
                        }
                    }
                    __syncthreads();
                    cur ^= 1;
                    si = nsi;
                }
            }
        }  // nets

        // ================= epilogue =========================================================
        if (!BWD) {
            if (a.mode == NCA_MODE_RAYS) {
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
    const float* __restrict__ Ag = a.scratch + (int64_t)job.d_row0 * a.Nc;
    const float* __restrict__ Bg = a.scratch + (int64_t)job.b_row0 * a.Nc;

    const int64_t ntile = a.Nc / 32;
    const int64_t per = (ntile + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = (int64_t)q * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;

    f32x16 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = (f32x16)(0.f);
    float bsum = 0.f;

    const int lrow = tid >> 3, lc4 = tid & 7;                    // loader: rows lrow + 32 i, float4 chunk lc4
    float4 pa[4], pb[4];
    auto issue = [&](int64_t t) {
        const int64_t c0 = t * 32 + lc4 * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lrow + 32 * i;
            pa[i] = r < F ? *reinterpret_cast<const float4*>(Ag + (int64_t)r * a.Nc + c0) : make_float4(0, 0, 0, 0);
            pb[i] = r < brows ? *reinterpret_cast<const float4*>(Bg + (int64_t)r * a.Nc + c0) : make_float4(0, 0, 0, 0);
        }
    };
    auto commit = [&](int buf) {
        float* ad = As + buf * 128 * WG_PITCH;
        float* bd = Bs + buf * 128 * WG_PITCH;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = lrow + 32 * i;
            float* x = ad + r * WG_PITCH + lc4 * 4;
            x[0] = pa[i].x; x[1] = pa[i].y; x[2] = pa[i].z; x[3] = pa[i].w;
            float* z = bd + r * WG_PITCH + lc4 * 4;
            z[0] = pb[i].x; z[1] = pb[i].y; z[2] = pb[i].z; z[3] = pb[i].w;
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
// reduce: natural flat gradients from the split slabs (fixed summation order)
// ------------------------------------------------------------------------------------------
__global__ void nca_reduce_f32(const NcaReduceArgs a) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n_total) return;
    const int net = e < a.n_params[0] ? 0 : 1;
    const int64_t le = net == 0 ? e : e - a.n_params[0];
    const NcaReduceNet& rn = a.net[net];
    float* out = rn.grads + le;
    if (le >= rn.wo_off) {
        // Wo / bo: sum the fused kernel's per-workgroup partials
        const int k = (int)(le - rn.wo_off);   // 0..F (F = bias)
        float s = 0.f;
        for (int w = 0; w < a.n_wg; ++w) s += a.oslab[(int64_t)w * a.oslab_stride + net * (rn.F + 1) + k];
        *out = s;
        return;
    }
    if (le < rn.lat_count) {
        // time_latents[p][t]: dL = sum_f W0[f][Kenc+t] * sum_q Dsum_q[f][p]
        const int pp = (int)(le / rn.T), t = (int)(le % rn.T);
        float s = 0.f;
        for (int f = 0; f < rn.F; ++f) {
            float d = 0.f;
            for (int q = 0; q < a.n_split; ++q) d += a.slab[(int64_t)q * a.slab_stride + rn.onehot_off + f * rn.P + pp];
            s = fmaf(rn.params[rn.w0_off + f * rn.K0 + rn.Kenc + t], d, s);
        }
        *out = s;
        return;
    }
    float s = 0.f;
    for (int q = 0; q < a.n_split; ++q) s += a.slab[(int64_t)q * a.slab_stride + rn.slab_off + le];
    *out = s;
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
template <int F>
static hipError_t launch_fused_t(const NcaFusedArgs& a, bool bwd, int grid, hipStream_t st) {
    const size_t lds = 2 * FusedCfg<F>::BUF_BYTES + (bwd ? NCA_WAVES * 2 * (F + 1) * sizeof(float) : 0);
    if (bwd) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_fused_f32<F, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((nca_fused_f32<F, true>), dim3(grid), dim3(NCA_NT), lds, st, a);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_fused_f32<F, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((nca_fused_f32<F, false>), dim3(grid), dim3(NCA_NT), lds, st, a);
    }
    return hipGetLastError();
}

hipError_t nca_launch_fused_f32(int F, const NcaFusedArgs& a, bool bwd, int grid, hipStream_t st) {
    switch (F) {
        case 32: return launch_fused_t<32>(a, bwd, grid, st);
        case 64: return launch_fused_t<64>(a, bwd, grid, st);
        case 128: return launch_fused_t<128>(a, bwd, grid, st);
    }
    return hipErrorInvalidValue;
}

hipError_t nca_launch_pack_f32(const NcaLayout& y, const float* prm, void* out, hipStream_t st) {
    const int total = (int)(y.packed_bytes / 4u);
    const int grid = (total + 255) / 256;
    hipLaunchKernelGGL(nca_pack_f32, dim3(grid > 1024 ? 1024 : grid), dim3(256), 0, st, y, prm, reinterpret_cast<float*>(out));
    return hipGetLastError();
}

hipError_t nca_launch_wgrad_f32(const NcaWgradArgs& a, int nsplit, hipStream_t st) {
    const size_t lds = 4 * 128 * WG_PITCH * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_wgrad_f32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nca_wgrad_f32, dim3(nsplit, a.njobs), dim3(256), lds, st, a);
    return hipGetLastError();
}

hipError_t nca_launch_reduce_f32(const NcaReduceArgs& a, hipStream_t st) {
    const int grid = (int)((a.n_total + 255) / 256);
    hipLaunchKernelGGL(nca_reduce_f32, dim3(grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

hipError_t nca_launch_pix_f32(int64_t R, int nchunk, const float* I0, const double* part, double* pix, hipStream_t st) {
    const int grid = (int)((R + 255) / 256);
    hipLaunchKernelGGL(nca_pix_f32, dim3(grid), dim3(256), 0, st, R, nchunk, I0, part, pix);
    return hipGetLastError();
}
