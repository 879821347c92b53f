// nca_kernels_loss.hip -- pixel loss + D2NeRF-style regularisers of train/model_helpers.py:189-262 and
// the loss assembly of train/run_composite.py:287-292, forward AND gradient in one pass per ray:
//
//   terms[]  = loss, pixel, blendw mean, sigma maxima, favor, static/dynamic ray entropy + ray sums,
//              occlusion, l1, l2                                  (the reference's 11-tuple + loss)
//   g_pix[r], g_sig_s[r,s], g_sig_d[r,s] = d loss / d (pix, sigma_s, sigma_d)
//
// One wave per ray (a ray's S samples are strided over the 64 lanes); three wave reductions per ray.
// Arithmetic follows the reference's dtypes: the blend-weight entropy runs in f32 (sigma is f32), every
// term that touches `dists` runs in f64.  Sums over rays: per-wave values -> per-block partials ->
// one finishing block, all in fixed order (bit-reproducible, no atomics).
#include <hip/hip_runtime.h>
#include "nca_kernels.hpp"
#include "nca_rng.hpp"

#define LOSS_WAVES 4
#define LOSS_NT (64 * LOSS_WAVES)
enum { T_LOSS = 0, T_PIXEL, T_BLENDW, T_SMAX, T_DMAX, T_FAVOR, T_SENT, T_SSUM, T_DENT, T_DSUM, T_OCCL, T_L1, T_L2, T_COUNT };
#define NPART 12   // per-block partial sums (maxima handled separately)

__device__ __forceinline__ double wsum(double v) {
    v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 32)); v = fmaxf(v, __shfl_xor(v, 16)); v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 4)); v = fmaxf(v, __shfl_xor(v, 2)); v = fmaxf(v, __shfl_xor(v, 1));
    return v;
}

__global__ __launch_bounds__(LOSS_NT) void nca_loss_rays(const NcaLossArgs a) {
    __shared__ double sh[LOSS_WAVES][NPART];
    __shared__ float shm[LOSS_WAVES][2];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * LOSS_WAVES + wave;
    double part[NPART];
#pragma unroll
    for (int i = 0; i < NPART; ++i) part[i] = 0.0;
    float mx_s = 0.f, mx_d = 0.f;

    // this step's weights: by value, or from the device vector a graph replay refreshes -- or, term-gradient mode (a.term_grads: the
    // backward of compute_losses as an autograd function), the upstream gradient of each of the reference's eleven return values
    // [blendw, max_s, max_d, favor, s_entropy, s_sum, d_entropy, d_sum, occl, l1, l2] (the maxima carry none, model_helpers.py:189-198)
    const bool tgm = a.term_grads != nullptr;
    const double w_favor = tgm ? a.term_grads[3] : (a.weights_dev ? a.weights_dev[0] : a.w_favor), w_dent = tgm ? a.term_grads[6] : (a.weights_dev ? a.weights_dev[1] : a.w_dent);
    const double w_occl = tgm ? a.term_grads[8] : (a.weights_dev ? a.weights_dev[2] : a.w_occl), w_l1 = tgm ? a.term_grads[9] : (a.weights_dev ? a.weights_dev[3] : a.w_l1);
    const double w_l2 = tgm ? a.term_grads[10] : w_l1;
    const double w_bw = tgm ? a.term_grads[0] : 0.0, w_sent = tgm ? a.term_grads[4] : 0.0, w_ssum = tgm ? a.term_grads[5] : 0.0, w_dsum = tgm ? a.term_grads[7] : 0.0;
    if (r < a.R) {
        const float* ss = a.sig_s + r * a.S;
        const float* sd = a.sig_d + r * a.S;
        const float skew = (float)a.skew;
        // ---- pass 1: ray sums, favor entropy, l2, maxima -------------------------------------------------
        double Ms = 0.0, Md = 0.0, l2 = 0.0, fav = 0.0, bwsum = 0.0;
        for (int s = lane; s < a.S; s += 64) {
            const float vs = ss[s], vd = sd[s];
            const double dl = a.dists[s];
            const double ms = (double)vs * dl, md = (double)vd * dl;
            Ms += ms; Md += md; l2 += ms * ms;
            mx_s = fmaxf(mx_s, vs); mx_d = fmaxf(mx_d, vd);
            // compute_ratio / compute_blendw_loss in f32 (model_helpers.py:189-204)
            const float bw = vd / (vs + vd + 1e-10f);
            bwsum += (double)bw;
            float b = skew == 1.f ? bw : powf(bw, skew);
            b = fminf(fmaxf(b, 1e-19f), 1.f - 1e-19f);
            const float rb = fmaxf(1.f - b, 1e-19f);
            fav += (double)(-(b * logf(b) + rb * logf(rb)));
        }
        Ms = wsum(Ms); Md = wsum(Md); l2 = wsum(l2); fav = wsum(fav); bwsum = wsum(bwsum);
        // ---- pass 2: ray entropies (compute_sigma_s_ray_loss, model_helpers.py:206-224) ------------------
        const double Mcs = fmax(Ms, 1e-19), Mcd = fmax(Md, 1e-19);
        double es = 0.0, ed = 0.0, qp = 0.0, qps = 0.0;       // (qps: the static field's sum of q p, term-gradient mode only)
        // q = ln(pd + eps) + pd / (pd + eps) of this lane's samples, kept for the gradient pass (an f64 logarithm is the most
        // expensive thing in this kernel): up to QKEEP * 64 samples per ray, beyond that the gradient pass recomputes
        constexpr int QKEEP = 8;
        double qkeep[QKEEP];
#pragma unroll
        for (int j = 0; j < QKEEP; ++j) qkeep[j] = 0.0;
#pragma unroll
        for (int j = 0; j < QKEEP; ++j) {
            const int s = lane + 64 * j;
            if (s < a.S) {
                const double dl = a.dists[s];
                const double ps = (double)ss[s] * dl / Mcs, pd = (double)sd[s] * dl / Mcd;
                const double lgs = log(ps + 1e-10);
                es += ps * lgs;
                if (tgm) qps += (lgs + ps / (ps + 1e-10)) * ps;
                const double lg = log(pd + 1e-10);
                ed += pd * lg;
                const double q = lg + pd / (pd + 1e-10);
                qkeep[j] = q;
                qp += q * pd;
            }
        }
        for (int s = lane + 64 * QKEEP; s < a.S; s += 64) {
            const double dl = a.dists[s];
            const double ps = (double)ss[s] * dl / Mcs, pd = (double)sd[s] * dl / Mcd;
            const double lgs = log(ps + 1e-10);
            es += ps * lgs;
            if (tgm) qps += (lgs + ps / (ps + 1e-10)) * ps;
            const double lg = log(pd + 1e-10);
            ed += pd * lg;
            qp += (lg + pd / (pd + 1e-10)) * pd;
        }
        es = wsum(es); ed = wsum(ed); qp = wsum(qp);
        if (tgm) qps = wsum(qps);
        const double wr = a.wpix[r];
        const int mask_s = Ms < a.mask_thre ? 0 : 1;
        int mask_d = Md < a.mask_thre ? 0 : 1;
        if (a.use_weighting && wr > 1.0 + a.weighted_thresh) mask_d = 1;
        // pix: the caller's, or formed here from the forward's per-tile ray sums (same order as nca_pix_f32: bit-identical)
        double pix_r = 0.0;
        if (a.ray_part) {
            double sum = 0.0;
            for (int c = 0; c < a.ray_nchunk; ++c) sum += a.ray_part[r * a.ray_nchunk + c];
            pix_r = (double)a.ray_I0[r] - sum;
            if (a.pix_out && lane == 0) a.pix_out[r] = pix_r;
        } else if (a.pix) pix_r = a.pix[r];
        const double diff = (a.ray_part || a.pix) ? pix_r - a.gt[r] : 0.0;      // (no pixel term in term-gradient mode: weighted_MSELoss is its own function)
        // ---- gradients ------------------------------------------------------------------------------------
        const double wm = a.unit_mse ? 1.0 : wr;           // (fine pass: unit pixel weights, weighted regularisers)
        if (a.g_pix && lane == 0) a.g_pix[r] = 2.0 * wm * diff * a.inv_R;
        if (a.g_sig_s) {
            float* gs = a.g_sig_s + r * a.S;
            float* gd = a.g_sig_d + r * a.S;
            const float fscale = (float)(w_favor * a.inv_R / (double)a.S);
            const double escale = w_dent * a.inv_R * (double)mask_d / Mcd;
            const bool unclipped = Md >= 1e-19;      // d clip(M)/dM
            // term-gradient mode: the terms the assembled loss never weights (blend-weight mean, static ray entropy, the two ray sums)
            const float bscale = (float)(w_bw * a.inv_R / (double)a.S);
            const double escale_s = w_sent * a.inv_R * (double)mask_s / Mcs;
            const bool unclipped_s = Ms >= 1e-19;
            auto grad_of = [&](int s, bool have_q, double qk) __attribute__((always_inline)) {
                const float vs = ss[s], vd = sd[s];
                const double dl = a.dists[s];
                // favor: F = -(b ln b + rb ln rb), b = clip(bw^skew), rb = clip(1 - b)
                const float T = vs + vd + 1e-10f;
                const float bw = vd / T;
                const float braw = skew == 1.f ? bw : powf(bw, skew);
                float dFdbw = 0.f;
                if (braw >= 1e-19f && braw <= 1.f - 1e-19f) {
                    const float b = braw;
                    const float rbraw = 1.f - b;
                    float dF = -(logf(b) + 1.f);
                    if (rbraw >= 1e-19f) dF += logf(rbraw) + 1.f;
                    const float dbdbw = skew == 1.f ? 1.f : skew * powf(bw, skew - 1.f);
                    dFdbw = dF * dbdbw;
                }
                const float gf = fscale * dFdbw + bscale;           // (bscale = 0 outside term-gradient mode)
                const float g_s_f = gf * (-vd / (T * T));
                const float g_d_f = gf * ((T - vd) / (T * T));
                // dynamic ray entropy: E = -sum p ln(p+eps); dE/dm_j = (-q_j + [unclipped] sum_s q_s p_s) / M
                const double pd = (double)vd * dl / Mcd;
                const double q = have_q ? qk : log(pd + 1e-10) + pd / (pd + 1e-10);
                const double g_d_e = escale * dl * (-q + (unclipped ? qp : 0.0));
                const double g_d_o = w_occl * a.inv_R * dl;
                double g_s_l = w_l1 * (dl + 2.0 * (double)vs * dl * dl);
                double g_s_x = 0.0, g_d_x = 0.0, dd_x = 0.0;       // term-gradient mode's further terms (and their share of d / d dists)
                if (tgm) {
                    g_s_l = w_l1 * dl + w_l2 * (2.0 * (double)vs * dl * dl);
                    g_s_x = w_ssum * a.inv_R * dl;
                    g_d_x = w_dsum * a.inv_R * dl;
                    dd_x = w_ssum * a.inv_R * (double)vs + w_dsum * a.inv_R * (double)vd;
                    if (w_sent != 0.0) {
                        const double ps = (double)vs * dl / Mcs;
                        const double qs = log(ps + 1e-10) + ps / (ps + 1e-10);
                        const double e = escale_s * (-qs + (unclipped_s ? qps : 0.0));
                        g_s_x += e * dl;
                        dd_x += e * (double)vs;
                    }
                }
                gs[s] = g_s_f + (float)(g_s_l + g_s_x);
                gd[s] = g_d_f + (float)(g_d_e + g_d_o + g_d_x);
                // d loss / d dists[s] of this ray: every term above reaches dists through m = sigma * dists (swap the factor), the
                // pixel term through pix = I0 - sum (sigma_s + sigma_d) dists
                if (a.dists_work)
                    a.dists_work[r * a.S + s] = -(2.0 * wm * diff * a.inv_R) * ((double)vs + (double)vd)
                                                + escale * (double)vd * (-q + (unclipped ? qp : 0.0)) + w_occl * a.inv_R * (double)vd
                                                + (tgm ? w_l1 * (double)vs + w_l2 * (2.0 * (double)vs * (double)vs * dl) + dd_x
                                                       : w_l1 * ((double)vs + 2.0 * (double)vs * (double)vs * dl));
            };
#pragma unroll
            for (int j = 0; j < QKEEP; ++j)
                if (lane + 64 * j < a.S) grad_of(lane + 64 * j, true, qkeep[j]);
            for (int s = lane + 64 * QKEEP; s < a.S; s += 64) grad_of(s, false, 0.0);
        }
        if (lane == 0) {
            part[0] = wm * diff * diff;                 // pixel (sum over rays; scaled by inv_R at the end)
            part[1] = bwsum / (double)a.S;              // mean_s blendw
            part[2] = fav / (double)a.S;                // mean_s entropy
            part[3] = (double)mask_s * -es;
            part[4] = Ms;
            part[5] = (double)mask_d * -ed;
            part[6] = Md;
            part[7] = Md;                               // occlusion with the all-ones mask = ray sum
            part[8] = Ms;                               // l1
            part[9] = l2;
        }
    }
    mx_s = wmax(mx_s); mx_d = wmax(mx_d);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NPART; ++i) sh[wave][i] = part[i];
        shm[wave][0] = mx_s; shm[wave][1] = mx_d;
    }
    __syncthreads();
    if (threadIdx.x < NPART) {
        double s = 0.0;
        for (int w = 0; w < LOSS_WAVES; ++w) s += sh[w][threadIdx.x];
        a.partials[(int64_t)blockIdx.x * (NPART + 2) + threadIdx.x] = s;
    } else if (threadIdx.x < NPART + 2) {
        const int k = threadIdx.x - NPART;
        float m = 0.f;
        for (int w = 0; w < LOSS_WAVES; ++w) m = fmaxf(m, shm[w][k]);
        a.partials[(int64_t)blockIdx.x * (NPART + 2) + threadIdx.x] = (double)m;
    }
}

// One block of 1024 threads sums the per-block partials in a fixed order: thread t takes blocks t, t + 1024, ... (all 14
// columns of a block in one pass, 112 contiguous bytes per block), then the wave and the 16 waves are folded in fixed trees.
#define LOSS_FIN_NT 1024
__global__ __launch_bounds__(LOSS_FIN_NT) void nca_loss_finish(const NcaLossArgs a, int nblocks) {
    __shared__ double sh[LOSS_FIN_NT / 64][NPART + 2];
    double v[NPART + 2];
#pragma unroll
    for (int k = 0; k < NPART + 2; ++k) v[k] = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += LOSS_FIN_NT) {
        const double* row = a.partials + (int64_t)b * (NPART + 2);
#pragma unroll
        for (int k = 0; k < NPART + 2; ++k) v[k] = k < NPART ? v[k] + row[k] : fmax(v[k], row[k]);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NPART + 2; ++k) {
        double x = v[k];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const double y = __shfl_xor(x, d);
            x = k < NPART ? x + y : fmax(x, y);
        }
        if (lane == 0) sh[wave][k] = x;
    }
    __syncthreads();
    double res[NPART + 2];
#pragma unroll
    for (int k = 0; k < NPART + 2; ++k) {
        double x = sh[0][k];
        for (int w = 1; w < LOSS_FIN_NT / 64; ++w) x = k < NPART ? x + sh[w][k] : fmax(x, sh[w][k]);
        res[k] = x;
    }
    if (threadIdx.x == 0) {
        const double iR = a.inv_R;
        double* t = a.terms;
        t[T_PIXEL] = res[0] * iR;
        t[T_BLENDW] = res[1] * iR;
        t[T_FAVOR] = res[2] * iR;
        t[T_SENT] = res[3] * iR;
        t[T_SSUM] = res[4] * iR;
        t[T_DENT] = res[5] * iR;
        t[T_DSUM] = res[6] * iR;
        t[T_OCCL] = res[7] * iR;
        t[T_L1] = res[8];
        t[T_L2] = res[9];
        t[T_SMAX] = res[NPART];
        t[T_DMAX] = res[NPART + 1];
        const double w_favor = a.weights_dev ? a.weights_dev[0] : a.w_favor, w_dent = a.weights_dev ? a.weights_dev[1] : a.w_dent;
        const double w_occl = a.weights_dev ? a.weights_dev[2] : a.w_occl, w_l1 = a.weights_dev ? a.weights_dev[3] : a.w_l1;
        t[T_LOSS] = t[T_PIXEL] + w_favor * t[T_FAVOR] + w_dent * t[T_DENT] + w_occl * t[T_OCCL] + w_l1 * t[T_L2] + w_l1 * t[T_L1];
        if (a.terms_f32)
            for (int k = 0; k < T_COUNT; ++k) a.terms_f32[k] = (float)t[k];
    }
}

// g_dists[s] = sum over rays of the per-ray contributions, one block per sample index, fixed order (thread t takes rays t, t + 256, ...;
// then a fixed tree)
__global__ __launch_bounds__(256) void nca_loss_dists_sum(const NcaLossArgs a) {
    __shared__ double sh[256];
    const int s = blockIdx.x;
    double v = 0.0;
    for (int64_t r = threadIdx.x; r < a.R; r += 256) v += a.dists_work[r * a.S + s];
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int d = 128; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.g_dists[s] = sh[0];
}

// weighted_MSELoss (train/model_helpers.py:284-288): out = (pred - gt)^2 * w, elementwise (the caller takes the mean), and its backward
template <typename T>
__global__ void nca_wsqerr_fwd(int64_t R, const T* __restrict__ pred, const T* __restrict__ gt, const T* __restrict__ w, T* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const T d = pred[r] - gt[r];
    out[r] = d * d * w[r];
}
template <typename T>
__global__ void nca_wsqerr_bwd(int64_t R, const T* __restrict__ pred, const T* __restrict__ gt, const T* __restrict__ w, const T* __restrict__ g_out,
                               T* __restrict__ g_pred, T* __restrict__ g_gt, T* __restrict__ g_w) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const T d = pred[r] - gt[r], g = g_out[r];
    const T gp = (T)2 * d * w[r] * g;
    if (g_pred) g_pred[r] = gp;
    if (g_gt) g_gt[r] = -gp;
    if (g_w) g_w[r] = d * d * g;
}
hipError_t nca_launch_wsqerr(int64_t R, bool f64, const void* pred, const void* gt, const void* w, void* out, hipStream_t st) {
    const int grid = (int)((R + 255) / 256);
    if (f64) hipLaunchKernelGGL(nca_wsqerr_fwd<double>, dim3(grid), dim3(256), 0, st, R, (const double*)pred, (const double*)gt, (const double*)w, (double*)out);
    else hipLaunchKernelGGL(nca_wsqerr_fwd<float>, dim3(grid), dim3(256), 0, st, R, (const float*)pred, (const float*)gt, (const float*)w, (float*)out);
    return hipGetLastError();
}
hipError_t nca_launch_wsqerr_bwd(int64_t R, bool f64, const void* pred, const void* gt, const void* w, const void* g_out, void* g_pred, void* g_gt, void* g_w, hipStream_t st) {
    const int grid = (int)((R + 255) / 256);
    if (f64) hipLaunchKernelGGL(nca_wsqerr_bwd<double>, dim3(grid), dim3(256), 0, st, R, (const double*)pred, (const double*)gt, (const double*)w, (const double*)g_out,
                                (double*)g_pred, (double*)g_gt, (double*)g_w);
    else hipLaunchKernelGGL(nca_wsqerr_bwd<float>, dim3(grid), dim3(256), 0, st, R, (const float*)pred, (const float*)gt, (const float*)w, (const float*)g_out,
                            (float*)g_pred, (float*)g_gt, (float*)g_w);
    return hipGetLastError();
}

hipError_t nca_launch_loss(const NcaLossArgs& a, hipStream_t st) {
    const int nblocks = (int)((a.R + LOSS_WAVES - 1) / LOSS_WAVES);
    hipLaunchKernelGGL(nca_loss_rays, dim3(nblocks), dim3(LOSS_NT), 0, st, a);
    hipLaunchKernelGGL(nca_loss_finish, dim3(1), dim3(LOSS_FIN_NT), 0, st, a, nblocks);
    if (a.g_dists) hipLaunchKernelGGL(nca_loss_dists_sum, dim3(a.S), dim3(256), 0, st, a);
    return hipGetLastError();
}

int64_t nca_loss_partials_bytes(int64_t R) {
    const int64_t nblocks = (R + LOSS_WAVES - 1) / LOSS_WAVES;
    return nblocks * (NPART + 2) * (int64_t)sizeof(double);
}

// ------------------------------------------------------------------------------------------
// stand-alone compositing of raw fields the caller already holds: render_volume_density_composite /
// render_volume_density (train/model_helpers.py:72-97), one wave per ray, forward and backward
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float cact_fwd(int act, float x) {
    if (act == NCA_ACT_SIGMOID) return 1.f / (1.f + expf(-x));
    float sp = x > 20.f ? x : log1pf(expf(x));
    if (act == NCA_ACT_CLAMP) sp = fminf(fmaxf(sp, 0.f), 1.f);
    return sp;
}
__device__ __forceinline__ float cact_bwd(int act, float x) {
    if (act == NCA_ACT_SIGMOID) { float s = 1.f / (1.f + expf(-x)); return s * (1.f - s); }
    float d;
    if (x > 20.f) d = 1.f; else { float z = expf(x); d = z / (z + 1.f); }
    if (act == NCA_ACT_CLAMP) {
        float sp = x > 20.f ? x : log1pf(expf(x));
        if (!(sp > 0.f && sp < 1.f)) d = 0.f;
    }
    return d;
}

__global__ __launch_bounds__(LOSS_NT) void nca_composite_fwd_k(const NcaCompositeArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * LOSS_WAVES + wave;
    if (r >= a.R) return;
    double acc = 0.0;
    for (int s = lane; s < a.S; s += 64) {
        const int64_t n = r * a.S + s;
        if (a.single) {
            const float sa = cact_fwd(a.act, a.raw_s[n]);
            a.sig_s[n] = sa;                                           // un-scaled (model_helpers.py:90)
            acc += ((double)sa * a.dists[s]) * (double)a.scale;
        } else {
            const float ss = __fmul_rn(cact_fwd(a.act, a.raw_s[n]), a.scale);
            const float sd = __fmul_rn(cact_fwd(a.act, a.raw_d[n]), a.scale);
            a.sig_s[n] = ss;
            a.sig_d[n] = sd;
            acc += (double)__fadd_rn(ss, sd) * a.dists[s];
        }
    }
    acc = wsum(acc);
    if (lane == 0) a.pix[r] = (double)a.I0[r] - acc;
}

__global__ __launch_bounds__(256) void nca_composite_bwd_k(const NcaCompositeArgs a) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= a.R * a.S) return;
    const int64_t r = n / a.S;
    const int s = (int)(n % a.S);
    const double gp = (a.g_pix ? a.g_pix[r] : 0.0) * a.dists[s];
    const double gs = a.g_sig_s ? (double)a.g_sig_s[n] : 0.0;
    if (a.single) {
        a.g_raw_s[n] = (float)(gs - gp * (double)a.scale) * cact_bwd(a.act, a.raw_s[n]);
    } else {
        const double gd = a.g_sig_d ? (double)a.g_sig_d[n] : 0.0;
        a.g_raw_s[n] = (float)((gs - gp) * (double)a.scale) * cact_bwd(a.act, a.raw_s[n]);
        a.g_raw_d[n] = (float)((gd - gp) * (double)a.scale) * cact_bwd(a.act, a.raw_d[n]);
    }
}

hipError_t nca_launch_composite(const NcaCompositeArgs& a, bool bwd, hipStream_t st) {
    if (bwd) {
        const int64_t n = a.R * a.S;
        hipLaunchKernelGGL(nca_composite_bwd_k, dim3((int)((n + 255) / 256)), dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL(nca_composite_fwd_k, dim3((int)((a.R + LOSS_WAVES - 1) / LOSS_WAVES)), dim3(LOSS_NT), 0, st, a);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Adam + LinearLR (train/run_composite.py:209-215, 307-308): the element-wise update of torch.optim.Adam's
// default path (lerp, addcmul, sqrt/ bias-correction, addcdiv) with the step-dependent scalars derived from
// a device-resident step counter, so the launch can sit in a captured graph.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nca_adam_k(const NcaAdamArgs a) {
    const int64_t done = *a.step;                                  // optimiser steps taken before this one
    const double t = (double)(done + 1);
    const double bc1 = 1.0 - pow(a.beta1, t), bc2 = 1.0 - pow(a.beta2, t);
    const double frac = a.lr_total_iters > 0 ? fmin((double)done, (double)a.lr_total_iters) / (double)a.lr_total_iters : 1.0;
    const double lr = a.lr * (1.0 + (a.lr_end_factor - 1.0) * frac);  // LinearLR, start_factor = 1, closed form
    const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2), eps = (float)a.eps;
    const float w1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, w2 = (float)(1.0 - a.beta2);
    const int seg = blockIdx.y;
    float* __restrict__ p = a.params[seg];
    const float* __restrict__ g = a.grads[seg];
    float* __restrict__ m = a.exp_avg[seg];
    float* __restrict__ v = a.exp_avg_sq[seg];
#pragma unroll 4
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n[seg]; i += (int64_t)gridDim.x * 256) {
        const float gi = g[i];
        const float mi = m[i] + w1 * (gi - m[i]);
        const float vi = v[i] * b2 + w2 * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] = p[i] - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    }
    // the tick: every workgroup read step[0] when it started; the last one to finish increments it (and the training iteration), and
    // leaves the arrival counter step[1] at zero for the next launch -- one launch instead of an update and a one-thread kernel
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned total = gridDim.x * gridDim.y;
        unsigned* arrived = reinterpret_cast<unsigned*>(a.step + 1);
        if (atomicAdd(arrived, 1u) == total - 1u) {
            *arrived = 0u;
            a.step[0] = done + 1;
            if (a.iter_counter) *a.iter_counter += 1;
        }
    }
}

hipError_t nca_launch_adam(const NcaAdamArgs& a, hipStream_t st) {
    int64_t nmax = 0;
    for (int s = 0; s < a.n_seg; ++s) nmax = a.n[s] > nmax ? a.n[s] : nmax;
    // few, fat workgroups: the launch ends with ONE same-address atomic per workgroup (the arrival count of the in-kernel tick), and those
    // serialise at ~25 ns each -- 602 of them were 15 us of a 20 us launch at the default nets' 77 056 parameters (profiles/r06_small_batch_trace.txt);
    // 64 workgroups per segment, four elements in flight per thread, move the same 1.2 MB in a few microseconds
    int gx = (int)((nmax + 1023) / 1024);
    gx = gx < 1 ? 1 : (gx > 64 ? 64 : gx);
    hipLaunchKernelGGL(nca_adam_k, dim3(gx, a.n_seg), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Per-step batch preparation: gather of the sampled rays out of the resident table (run_composite.py:262-273) and the stratified
// depth jitter with its interval lengths (model_helpers.py:3-12, 73-74).  Thread r copies ray ids[r]; the first S threads also
// jitter one depth each.  Arithmetic in the reference's order and precision (the library is compiled with -ffp-contract=off).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nca_prepare_batch_k(int64_t R, int S, const int64_t* __restrict__ ids, const double* __restrict__ table,
                                                           const int64_t* __restrict__ phases, int64_t n_rows, int32_t* __restrict__ bad_ids,
                                                           const float* __restrict__ depth, const float* __restrict__ t_rand,
                                                           double* __restrict__ o, double* __restrict__ d, double* __restrict__ gt, double* __restrict__ w,
                                                           int32_t* __restrict__ ph, float* __restrict__ z, double* __restrict__ dists) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    auto jitter = [&](int k) {
        // mid = 0.5 * (z[1:] + z[:-1]); hi = cat(mid, z[-1:]); lo = cat(z[:1], mid); z' = lo + (hi - lo) * t
        const float hi = k + 1 < S ? __fmul_rn(0.5f, __fadd_rn(depth[k + 1], depth[k])) : depth[S - 1];
        const float lo = k > 0 ? __fmul_rn(0.5f, __fadd_rn(depth[k], depth[k - 1])) : depth[0];
        return __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), t_rand[k]));
    };
    if (i < S) {
        const int k = (int)i;
        const float zk = jitter(k);
        z[k] = zk;
        dists[k] = k + 1 < S ? (double)__fsub_rn(jitter(k + 1), zk) : 1e-10;
    }
    if (i < R) {
        int64_t id = ids[i];
        if (n_rows > 0 && (id < 0 || id >= n_rows)) {          // never reaches memory: clamped, and counted for the caller
            if (bad_ids) atomicAdd(bad_ids, 1);
            id = id < 0 ? 0 : n_rows - 1;
        }
        const double* row = table + id * 12;
#pragma unroll
        for (int c = 0; c < 3; ++c) { o[i * 3 + c] = row[c]; d[i * 3 + c] = row[3 + c]; }
        gt[i] = row[6];
        w[i] = row[9];
        ph[i] = (int32_t)phases[id];
    }
}
hipError_t nca_launch_prepare_batch(int64_t R, int S, const int64_t* ids, const double* table, const int64_t* phases, int64_t n_rows, int32_t* bad_ids,
                                    const float* depth, const float* t_rand,
                                    double* o, double* d, double* gt, double* w, int32_t* ph, float* z, double* dists, hipStream_t st) {
    const int64_t n = R > S ? R : S;
    hipLaunchKernelGGL(nca_prepare_batch_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, R, S, ids, table, phases, n_rows, bad_ids, depth, t_rand, o, d, gt, w, ph, z, dists);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Per-step sampling and schedules on the device (include/nerfca_hip.h "per-step batch sampling"; csrc/nca_rng.hpp): the importance
// sampling of run_composite.py:250-260, the jitter draw of model_helpers.py:8, the FreeNeRF windows of CPPN.py:144-159 and the
// four linear_param_decay weights of run_composite.py:276-279, all as functions of (seed, iteration).
// ------------------------------------------------------------------------------------------
// (wave-uniform by construction -- one address for every lane -- and told so: everything that depends only on (seed, iteration), the
// permutation keys above all, then runs on the scalar ALU once per wave instead of per lane)
__device__ __forceinline__ int64_t sampler_iter(const NcaSampler& s) {
    int64_t d = s.iter_dev ? *s.iter_dev : (int64_t)0;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uint64_t)d), hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)d >> 32));
    return s.n_iter + (int64_t)(((uint64_t)hi << 32) | lo);
}

// the ray id of slot `slot` of the global batch
__device__ __forceinline__ int64_t draw_ray_id(const NcaSampler& s, int64_t it, int64_t slot, int half, const NcaPermKeys& keys) {
    const NcaU4 r = nca_rng_words(s.seed, it, NCA_RNG_STREAM_IDS, (uint64_t)slot);
    if (s.n_var > 0 && s.n_var_ids > 0) {
        const bool var = nca_perm((uint64_t)slot, (uint64_t)s.R_global, half, keys) < (uint64_t)s.n_var;
        return var ? s.var_ids[nca_rng_below(r.x, r.y, (uint64_t)s.n_var_ids)] : s.non_var_ids[nca_rng_below(r.x, r.y, (uint64_t)s.n_non_var_ids)];
    }
    return (int64_t)nca_rng_below(r.x, r.y, (uint64_t)s.n_rows);
}

__global__ __launch_bounds__(256) void nca_draw_ray_ids_k(const NcaSampler s, int64_t slot0, int64_t R, int64_t* __restrict__ ids) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const int64_t it = sampler_iter(s);
    const NcaPermKeys keys = nca_perm_keys(s.seed, it);
    ids[i] = draw_ray_id(s, it, slot0 + i, nca_perm_half_bits((uint64_t)s.R_global), keys);
}
__global__ __launch_bounds__(256) void nca_draw_uniform_k(const NcaSampler s, int stream_id, int64_t n, float* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = nca_rng_unit(nca_rng_words(s.seed, sampler_iter(s), stream_id, (uint64_t)i).x);
}
hipError_t nca_launch_draw_ray_ids(const NcaSampler& s, int64_t slot0, int64_t R, int64_t* ids, hipStream_t st) {
    hipLaunchKernelGGL(nca_draw_ray_ids_k, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, s, slot0, R, ids);
    return hipGetLastError();
}
hipError_t nca_launch_draw_uniform(const NcaSampler& s, int stream_id, int64_t n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(nca_draw_uniform_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s, stream_id, n, out);
    return hipGetLastError();
}

// update_freq_mask_alpha (model/CPPN.py:144-159; nerf-ca_amd/schedules.py:freq_mask): element k of the band window at iteration `it`, in the
// host schedule's arithmetic -- pointer = (L * it) / max + start in f64, clip to [1e-8, 1 - 1e-8] in f64, rounded to f32
__device__ __forceinline__ float free_window_at(const NcaWindowSched& w, int64_t it, int k) {
    if (it >= w.decay_steps) return 1.f;
    const double pointer = (double)((int64_t)w.L * it) / (double)w.decay_steps + (double)w.window_start;
    const int64_t whole = (int64_t)pointer;
    double m = k < whole ? 1.0 : (k == whole ? pointer - (double)whole : 0.0);
    m = fmin(fmax(m, 1e-8), 1.0 - 1e-8);
    return (float)m;
}
// linear_param_decay (train/model_helpers.py:264-269) in the host's f64 arithmetic
__device__ __forceinline__ double linear_decay_at(const NcaWeightSched& w, int64_t it) {
    if (it < w.delay) return 0.0;
    const double a = fmin((double)(it - w.delay) / (double)w.steps, 1.0);
    return (1.0 - a) * w.start + a * w.end;
}

__global__ __launch_bounds__(256) void nca_begin_step_k(const NcaBeginArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t it = sampler_iter(a.s);
    const int S = a.S;
    auto t_of = [&](int k) { return a.t_rand_in ? a.t_rand_in[k] : nca_rng_unit(nca_rng_words(a.s.seed, it, NCA_RNG_STREAM_JITTER, (uint64_t)k).x); };
    auto jitter = [&](int k, float t) {
        // mid = 0.5 * (z[1:] + z[:-1]); hi = cat(mid, z[-1:]); lo = cat(z[:1], mid); z' = lo + (hi - lo) * t   (model_helpers.py:3-12)
        const float hi = k + 1 < S ? __fmul_rn(0.5f, __fadd_rn(a.depth[k + 1], a.depth[k])) : a.depth[S - 1];
        const float lo = k > 0 ? __fmul_rn(0.5f, __fadd_rn(a.depth[k], a.depth[k - 1])) : a.depth[0];
        return __fadd_rn(lo, __fmul_rn(__fsub_rn(hi, lo), t));
    };
    if (i < S) {
        const int k = (int)i;
        const float tk = t_of(k);
        const float zk = jitter(k, tk);
        a.z[k] = zk;
        a.dists[k] = k + 1 < S ? (double)__fsub_rn(jitter(k + 1, t_of(k + 1)), zk) : 1e-10;
        if (a.t_rand_out) a.t_rand_out[k] = tk;
    }
    if (blockIdx.x == gridDim.x - 1) {        // schedules: the last workgroup's threads (windows: 64 per vector; weights: threads 0..3 of the last 64)
        const int t = threadIdx.x;
        if (t < 64 * a.sch.n_windows) {
            const NcaWindowSched& w = a.sch.window[t >> 6];
            const int k = t & 63;
            if (w.kind == NCA_WINDOW_FREE && w.out && k < w.L) w.out[k] = free_window_at(w, it, k);
        }
        if (a.sch.weights_out && t >= 252) a.sch.weights_out[t - 252] = linear_decay_at(a.sch.weight[t - 252], it);
    }
    if (i < a.R) {
        int64_t id;
        if (a.ids_in) id = a.ids_in[i];
        else {
            const NcaPermKeys keys = nca_perm_keys(a.s.seed, it);
            id = draw_ray_id(a.s, it, a.slot0 + i, nca_perm_half_bits((uint64_t)a.s.R_global), keys);
        }
        if (a.ids_out) a.ids_out[i] = id;
        if (a.s.n_rows > 0 && (id < 0 || id >= a.s.n_rows)) {          // never reaches memory: clamped, and counted for the caller
            if (a.bad_ids) atomicAdd(a.bad_ids, 1);
            id = id < 0 ? 0 : a.s.n_rows - 1;
        }
        const double* row = a.table + id * 12;
#pragma unroll
        for (int c = 0; c < 3; ++c) { a.o[i * 3 + c] = row[c]; a.d[i * 3 + c] = row[3 + c]; }
        a.gt[i] = row[6];
        a.w[i] = row[9];
        a.ph[i] = (int32_t)a.phases[id];
    }
}
hipError_t nca_launch_begin_step(const NcaBeginArgs& a, hipStream_t st) {
    const int64_t n = a.R > a.S ? a.R : a.S;
    hipLaunchKernelGGL(nca_begin_step_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Fine-pass depths (train/model_helpers.py:131-148 + sample_pdf 162-187): per ray, the jump of the total density
// between neighbouring coarse samples -- normalised by the BATCH-wide maximum (:139) -- is the weight of the bin
// between their mid-points; n_fine depths are drawn by inverse-transform sampling of that piecewise-constant pdf and
// merged with the coarse depths into one sorted vector.  One wave per ray: wave prefix scan for the CDF, binary
// search in LDS for searchsorted(right=True), bitonic sort of the draws, rank merge with the (sorted) coarse depths.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float fine_total(const NcaFineArgs& a, int64_t r, int k) {
    const float s = a.sig_s[r * a.S + k];
    return a.sig_d ? __fadd_rn(s, a.sig_d[r * a.S + k]) : s;
}

// pass 1: maximum jump per block (the leading 1e-10 of every ray's weight vector is folded into the finishing step)
__global__ __launch_bounds__(LOSS_NT) void nca_fine_max_k(const NcaFineArgs a) {
    __shared__ float sh[LOSS_WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * LOSS_WAVES + wave;
    float m = 0.f;
    if (r < a.R)
        for (int k = 1 + lane; k < a.S; k += 64) m = fmaxf(m, fabsf(__fsub_rn(fine_total(a, r, k), fine_total(a, r, k - 1))));
    m = wmax(m);
    if (lane == 0) sh[wave] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float v = sh[0];
        for (int w = 1; w < LOSS_WAVES; ++w) v = fmaxf(v, sh[w]);
        a.partial_max[blockIdx.x] = v;
    }
}
__global__ __launch_bounds__(256) void nca_fine_max_finish_k(const NcaFineArgs a, int nblocks) {
    __shared__ float sh[256];
    float v = 1e-10f;
    for (int b = threadIdx.x; b < nblocks; b += 256) v = fmaxf(v, a.partial_max[b]);
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + st]);
        __syncthreads();
    }
    if (threadIdx.x == 0) *a.jmax = sh[0];
}

__device__ __forceinline__ float wscan_incl(float v, int lane) {     // inclusive prefix sum over the 64 lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(v, d);
        if (lane >= d) v += o;
    }
    return v;
}

// pass 2: one wave per ray.  LDS per wave: cdf[S-1] | draws[npad] | zc[S]
__global__ __launch_bounds__(LOSS_NT) void nca_fine_sample_k(const NcaFineArgs a, int npad) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * LOSS_WAVES + wave;
    if (r >= a.R) return;                                   // no block-wide barrier below: waves are independent
    const int S = a.S, NF = a.n_fine, NB = S - 1;            // NB bins (mid-points) = length of the cdf
    float* cdf = reinterpret_cast<float*>(fsm) + (size_t)wave * (NB + npad + S);
    float* dr = cdf + NB;
    float* zc = dr + npad;
    const float jmax = *a.jmax;
    for (int k = lane; k < S; k += 64) zc[k] = a.z[k];

    // weights of bins 1 .. S-2 (weights[..., 1:-1]) + 1e-5, their sum
    float sum = 0.f;
    for (int j = lane; j < S - 2; j += 64) {
        const float w = __fadd_rn(__fdiv_rn(fabsf(__fsub_rn(fine_total(a, r, j + 1), fine_total(a, r, j))), jmax), 1e-5f);
        cdf[j + 1] = w;                                      // staged in place
        sum += w;
    }
    sum = (float)wsum((double)sum);
    __builtin_amdgcn_wave_barrier();
    // cdf = [0, cumsum(w / sum)]
    float carry = 0.f;
    for (int base = 0; base < S - 2; base += 64) {
        const int j = base + lane;
        const float p = j < S - 2 ? __fdiv_rn(cdf[j + 1], sum) : 0.f;
        const float inc = wscan_incl(p, lane) + carry;
        if (j < S - 2) cdf[j + 1] = inc;
        carry = __shfl(inc, 63);
    }
    if (lane == 0) cdf[0] = 0.f;
    __builtin_amdgcn_wave_barrier();

    // inverse-transform sampling (searchsorted(right=True): number of cdf entries <= u)
    for (int j = lane; j < npad; j += 64) {
        float smp = __int_as_float(0x7f800000);              // +inf padding sorts to the end
        if (j < NF) {
            const float u = a.u[r * NF + j];
            int lo = 0, hi = NB;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
            const int below = lo - 1 < 0 ? 0 : lo - 1, above = lo > NB - 1 ? NB - 1 : lo;
            const float c0 = cdf[below], c1 = cdf[above];
            const float b0 = 0.5f * (zc[below + 1] + zc[below]), b1 = 0.5f * (zc[above + 1] + zc[above]);
            float den = c1 - c0;
            if (den < 1e-5f) den = 1.f;
            smp = b0 + (u - c0) / den * (b1 - b0);
        }
        dr[j] = smp;
    }
    __builtin_amdgcn_wave_barrier();
    // bitonic sort of the draws (npad = power of two >= 64)
    for (int k = 2; k <= npad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = lane; i < npad; i += 64) {
                const int l = i ^ j;
                if (l > i) {
                    const float x = dr[i], y = dr[l];
                    const bool up = (i & k) == 0;
                    if ((x > y) == up) { dr[i] = y; dr[l] = x; }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    // rank merge; torch.sort of cat([draws, coarse]) is stable, so a draw precedes an equal coarse depth
    float* out = a.z_all + r * (int64_t)(S + NF);
    for (int i = lane; i < S; i += 64) {
        const float v = zc[i];
        int lo = 0, hi = NF;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (dr[mid] <= v) lo = mid + 1; else hi = mid; }
        out[i + lo] = v;
    }
    for (int j = lane; j < NF; j += 64) {
        const float v = dr[j];
        int lo = 0, hi = S;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (zc[mid] < v) lo = mid + 1; else hi = mid; }
        out[j + lo] = v;
    }
}

int64_t nca_fine_partials(int64_t R) { return (R + LOSS_WAVES - 1) / LOSS_WAVES; }

// stage 1: batch-wide maximum of the density jumps -> *a.jmax;  stage 2: sampling with the maximum found at a.jmax
// (a caller that shards the batch over ranks all-reduces MAX between the two)
hipError_t nca_launch_fine_max(const NcaFineArgs& a, hipStream_t st) {
    const int nblocks = (int)nca_fine_partials(a.R);
    hipLaunchKernelGGL(nca_fine_max_k, dim3(nblocks), dim3(LOSS_NT), 0, st, a);
    hipLaunchKernelGGL(nca_fine_max_finish_k, dim3(1), dim3(256), 0, st, a, nblocks);
    return hipGetLastError();
}
hipError_t nca_launch_fine_sample(const NcaFineArgs& a, hipStream_t st) {
    const int nblocks = (int)nca_fine_partials(a.R);
    int npad = 64;
    while (npad < a.n_fine) npad <<= 1;
    const size_t lds = (size_t)LOSS_WAVES * ((a.S - 1) + npad + a.S) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_fine_sample_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nca_fine_sample_k, dim3(nblocks), dim3(LOSS_NT), lds, st, a, npad);
    return hipGetLastError();
}
hipError_t nca_launch_fine(const NcaFineArgs& a, hipStream_t st) {
    hipError_t e = nca_launch_fine_max(a, st);
    return e != hipSuccess ? e : nca_launch_fine_sample(a, st);
}

// ------------------------------------------------------------------------------------------
// Backward of the fine-pass depths (the reference leaves them in the autograd graph, model_helpers.py:135-146): from
// d loss / d z_all back to the coarse densities.  One wave per ray, the forward's own arithmetic redone for the cdf and the
// searches (same operations in the same order: the same bins are hit), then
//   g_zp[j]   = g_zall[position of draw j after the stable sort of cat([draws, coarse])]
//   zp        = b0 + (u - c0) / den (b1 - b0):   d/dc0 = (u - c1) / den^2 (b1 - b0),  d/dc1 = -(u - c0) / den^2 (b1 - b0);
//               where den < 1e-5 was replaced by 1:  d/dc0 = -(b1 - b0), d/dc1 = 0   (torch.where passes no gradient to it)
//   cdf[k]    = sum_{i<k} pdf[i]  ->  g_pdf[i] = sum_{k>i} g_cdf[k]   (suffix sums, gathered per cdf entry in a fixed order)
//   pdf = wp / sum(wp),  wp[i] = |jump_{i+1}| / jmax + 1e-5,  jump_k = tot[k] - tot[k-1]
// g_tot[s] = sgn(jump_s) g_w[s] - sgn(jump_{s+1}) g_w[s+1].  The maximum jmax is one element of the BATCH: each ray leaves its
// part of d loss / d jmax and its count of jumps that attain it; the second stage hands (sum / count) to those jumps.
// LDS per wave: cdf[S-1] | gcdf[S-1] | wp[S] | zc[S] | zp[NF] | clo[NF] | chi[NF] | ibelow[NF] | iabove[NF]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(LOSS_NT) void nca_fine_bwd_k(const NcaFineBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char fsm[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * LOSS_WAVES + wave;
    if (r >= a.R) return;                                   // waves are independent (no block-wide barrier)
    const int S = a.S, NF = a.n_fine, NB = S - 1, NW = S - 2;
    float* cdf = reinterpret_cast<float*>(fsm) + (size_t)wave * (2 * NB + 2 * S + 5 * NF);
    float* gcdf = cdf + NB;
    float* wp = gcdf + NB;
    float* zc = wp + S;
    float* zp = zc + S;
    float* clo = zp + NF;
    float* chi = clo + NF;
    int* ibel = reinterpret_cast<int*>(chi + NF);
    int* iabo = ibel + NF;
    const float jmax = *a.jmax;
    NcaFineArgs fa{};
    fa.S = S; fa.sig_s = a.sig_s; fa.sig_d = a.sig_d;
    for (int k = lane; k < S; k += 64) zc[k] = a.z[k];
    // forward, as nca_fine_sample_k
    float sum = 0.f;
    for (int j = lane; j < NW; j += 64) {
        const float w = __fadd_rn(__fdiv_rn(fabsf(__fsub_rn(fine_total(fa, r, j + 1), fine_total(fa, r, j))), jmax), 1e-5f);
        cdf[j + 1] = w;
        wp[j] = w;
        sum += w;
    }
    sum = (float)wsum((double)sum);
    __builtin_amdgcn_wave_barrier();
    float carry = 0.f;
    for (int base = 0; base < NW; base += 64) {
        const int j = base + lane;
        const float p = j < NW ? __fdiv_rn(cdf[j + 1], sum) : 0.f;
        const float inc = wscan_incl(p, lane) + carry;
        if (j < NW) cdf[j + 1] = inc;
        carry = __shfl(inc, 63);
    }
    if (lane == 0) cdf[0] = 0.f;
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < NF; j += 64) {
        const float u = a.u[r * NF + j];
        int lo = 0, hi = NB;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] <= u) lo = mid + 1; else hi = mid; }
        const int below = lo - 1 < 0 ? 0 : lo - 1, above = lo > NB - 1 ? NB - 1 : lo;
        const float c0 = cdf[below], c1 = cdf[above];
        const float b0 = 0.5f * (zc[below + 1] + zc[below]), b1 = 0.5f * (zc[above + 1] + zc[above]);
        const float den = c1 - c0, db = b1 - b0;
        const bool repl = den < 1e-5f;
        const float dd = repl ? 1.f : den;
        zp[j] = b0 + (u - c0) / dd * db;
        ibel[j] = below; iabo[j] = above;
        clo[j] = repl ? -db : (u - c1) / (den * den) * db;       // d zp / d c0
        chi[j] = repl ? 0.f : -(u - c0) / (den * den) * db;      // d zp / d c1
    }
    __builtin_amdgcn_wave_barrier();
    // upstream gradient of every draw: its place in the stable sort of cat([draws, coarse])
    const float* gz = a.g_zall + r * (int64_t)(S + NF);
    for (int j = lane; j < NF; j += 64) {
        const float v = zp[j];
        int pos = 0;
        for (int k = 0; k < NF; ++k) { const float o = zp[k]; pos += (o < v || (o == v && k < j)) ? 1 : 0; }
        int lo = 0, hi = S;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (zc[mid] < v) lo = mid + 1; else hi = mid; }
        const float g = gz[pos + lo];
        clo[j] *= g;
        chi[j] *= g;
    }
    __builtin_amdgcn_wave_barrier();
    // g_cdf[k]: gather over the draws in index order (deterministic)
    for (int k = lane; k < NB; k += 64) {
        float g = 0.f;
        for (int j = 0; j < NF; ++j) {
            if (ibel[j] == k) g += clo[j];
            if (iabo[j] == k) g += chi[j];
        }
        gcdf[k] = g;
    }
    __builtin_amdgcn_wave_barrier();
    // g_pdf[i] = sum_{k = i+1}^{NB-1} g_cdf[k]: suffix sums, chunks of 64 from the end; kept in gcdf[i + 1] (entry of pdf i)
    float tailsum = 0.f;
    float dot = 0.f;                                      // sum_i g_pdf[i] wp[i]
    for (int top = NW; top > 0; top -= 64) {              // pdf indices [top - 64, top)
        const int i = top - 1 - lane;                     // lane 0 takes the highest index
        const float v = i >= 0 ? gcdf[i + 1] : 0.f;
        const float inc = wscan_incl(v, lane) + tailsum;  // sum of g_cdf[i+1 .. NB-1]
        if (i >= 0) { gcdf[i + 1] = inc; dot += inc * wp[i]; }
        tailsum = __shfl(inc, 63);
    }
    dot = (float)wsum((double)dot);
    __builtin_amdgcn_wave_barrier();
    // g_wp -> g_w (jumps 1 .. S-2) -> g_tot; d loss / d jmax
    float gm = 0.f, cnt = 0.f;
    float* gt = a.g_tot + r * (int64_t)S;
    for (int s0 = lane; s0 < S; s0 += 64) {
        // g_w of jump s0 and of jump s0 + 1 (0 outside 1 .. S-2)
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = s0 + e;
            if (k >= 1 && k <= S - 2) {
                const float gwp = gcdf[k] / sum - dot / (sum * sum);       // pdf index k - 1 lives in gcdf[k]
                const float gw = gwp / jmax;
                const float jump = __fsub_rn(fine_total(fa, r, k), fine_total(fa, r, k - 1));
                const float sg = jump > 0.f ? 1.f : (jump < 0.f ? -1.f : 0.f);
                acc += e == 0 ? sg * gw : -sg * gw;
                if (e == 0) gm -= gwp * fabsf(jump) / (jmax * jmax);
            }
        }
        gt[s0] = acc;
        if (s0 >= 1 && fabsf(__fsub_rn(fine_total(fa, r, s0), fine_total(fa, r, s0 - 1))) == jmax) cnt += 1.f;
    }
    gm = (float)wsum((double)gm);
    cnt = (float)wsum((double)cnt);
    if (lane == 0) { a.gmax_part[r] = gm; a.cnt_part[r] = cnt; }
}

// second stage: every jump that attains the batch-wide maximum receives gmax_each
__global__ __launch_bounds__(LOSS_NT) void nca_fine_bwd_max_k(const NcaFineBwdArgs a) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * LOSS_WAVES + wave;
    if (r >= a.R) return;
    const float jmax = *a.jmax, ge = *a.gmax_each;
    NcaFineArgs fa{};
    fa.S = a.S; fa.sig_s = a.sig_s; fa.sig_d = a.sig_d;
    float* gt = a.g_tot + r * (int64_t)a.S;
    for (int s0 = lane; s0 < a.S; s0 += 64) {
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int k = s0 + e;
            if (k >= 1 && k <= a.S - 1) {
                const float jump = __fsub_rn(fine_total(fa, r, k), fine_total(fa, r, k - 1));
                if (fabsf(jump) == jmax) {
                    const float sg = jump > 0.f ? 1.f : (jump < 0.f ? -1.f : 0.f);
                    acc += e == 0 ? sg * ge : -sg * ge;
                }
            }
        }
        if (acc != 0.f) gt[s0] += acc;
    }
}

hipError_t nca_launch_fine_bwd(const NcaFineBwdArgs& a, hipStream_t st) {
    const int nblocks = (int)nca_fine_partials(a.R);
    const size_t lds = (size_t)LOSS_WAVES * (2 * (a.S - 1) + 2 * a.S + 5 * a.n_fine) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_fine_bwd_k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(nca_fine_bwd_k, dim3(nblocks), dim3(LOSS_NT), lds, st, a);
    return hipGetLastError();
}
hipError_t nca_launch_fine_bwd_max(const NcaFineBwdArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(nca_fine_bwd_max_k, dim3((int)nca_fine_partials(a.R)), dim3(LOSS_NT), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// d loss / d depth per sample (fine pass of the reference, model_helpers.py:131-158: the sampled depths are NOT detached, so
// the fine losses reach them through  p = o + d z  ->  positional encoding  ->  first layer).  One wave per 32-sample tile of
// the f32 backward: lane (r, h) reads its 64 values of D_0 exactly as the dgrad kernel's lanes stored them (row
// 32 m + 8 q + e + 4 h of quad (m, q)), forms its half of G = W0[:, encoded columns]^T D_0 from the weights staged in LDS, the
// halves are added, and the encoding's derivative is applied in f64:
//   bands:   dp_c = G[c] + sum_k w_k 2^k ( cos(xb) G[3+6k+c] + cos(fl32(xb + fl32(pi/2))) G[6+6k+c] ),  xb = fl32(p_c 2^k)
//   fourier: dp_c = sum_{i = c mod 3} 2 pi g_i ( cos(v_i) G[i] - sin(v_i) G[3L+i] ),  v_i = fl32(fl32(2 pi p_c) g_i)
//   none:    dp_c = G[c];                       dz = sum_c d_c dp_c, summed over the nets of the render.
// ------------------------------------------------------------------------------------------
// G = W0[:, encoded columns]^T D_0 runs on the matrix cores (v_mfma_f32_32x32x2_f32, 3 row tiles = 96 encoded rows): D_0 is
// loaded into registers in the accumulator layout the dgrad kernel held it in -- which IS the B operand of that MFMA, k-step
// 16 t + i = register i of row tile t (rows rho(i) and rho(i) + 4 from the two lane halves) -- and the A operand comes from an
// LDS image [k-step][lane][4]: W0[o = 32 (s >> 4) + rho(s & 15) + 4 h][f = 32 m + r].  Lane (r, h) then holds G for the rows
// 32 m + rho(i) + 4 h of sample r; the encoding's derivative is linear in G, so each half adds its rows and the halves are
// summed.  sin / cos of all bands: one f64 sincos per coordinate + the double-angle recurrence, as in the forward.
#define ZG_MT 3
typedef float zg_f32x16 __attribute__((ext_vector_type(16)));
template <int F>
__global__ __launch_bounds__(256) void nca_zgrad_f32(const NcaZgradArgs a) {
    constexpr int MT = F / 32, NS = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) float zw[];        // [NS][64][4]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    for (int pass = 0; pass < 2 * a.nnets; ++pass) {
        const int net = pass >> 1, src = pass & 1;
        const NcaZgradNet& nn = a.net[net];
        if (src >= nn.nsrc) continue;              // (uniform over the block)
        const bool first = pass == 0;              // the first pass writes g_z, the others add to it
        __syncthreads();
        for (int i = threadIdx.x; i < NS * 64 * 4; i += 256) {
            const int m = i & 3, ln = (i >> 2) & 63, sidx = i >> 8;
            const int o = 32 * (sidx >> 4) + ((sidx & 15) & 3) + 8 * ((sidx & 15) >> 2) + 4 * (ln >> 5);
            const int f = 32 * m + (ln & 31);
            zw[i] = (m < ZG_MT && f < nn.Kenc) ? nn.w[src][(int64_t)o * nn.ldw[src] + f] : 0.f;
        }
        __syncthreads();
        for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < a.ntiles; tile += (int64_t)gridDim.x * 4) {
            const int64_t ray = a.ray0 + tile / a.nchunk;
            int smp = (int)(tile % a.nchunk) * 32 + lr;
            const bool valid = smp < a.S;
            if (!valid) smp = a.S - 1;
            // D block of this tile in accumulator order: D[t][i] = row 32 t + rho(i) + 4 h of sample r
            float D[MT][16];
            if (a.bf16) {
                const char* db = reinterpret_cast<const char*>(a.dscratch) + tile * a.d_total + nn.drow[src] + lane * 16;
#pragma unroll
                for (int ks = 0; ks < 2 * MT; ++ks) {     // fragment k-step ks, element j <-> register 8 (ks & 1) + j of row tile ks >> 1
                    const uint4 w4 = *reinterpret_cast<const uint4*>(db + ks * 1024);
                    const unsigned ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        D[ks >> 1][8 * (ks & 1) + 2 * u] = __uint_as_float(ww[u] << 16);
                        D[ks >> 1][8 * (ks & 1) + 2 * u + 1] = __uint_as_float(ww[u] & 0xffff0000u);
                    }
                }
            } else {
                const float* df = a.dscratch + (tile * a.d_total + nn.drow[src]) * 32 + lane * 4;
#pragma unroll
                for (int mq = 0; mq < 4 * MT; ++mq) {
                    const float4 dv = *reinterpret_cast<const float4*>(df + mq * 256);
                    D[mq >> 2][4 * (mq & 3)] = dv.x; D[mq >> 2][4 * (mq & 3) + 1] = dv.y;
                    D[mq >> 2][4 * (mq & 3) + 2] = dv.z; D[mq >> 2][4 * (mq & 3) + 3] = dv.w;
                }
            }
            zg_f32x16 G[ZG_MT];
#pragma unroll
            for (int m = 0; m < ZG_MT; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) G[m][i] = 0.f;
            const float4* img = reinterpret_cast<const float4*>(zw) + lane;
#pragma unroll
            for (int sidx = 0; sidx < NS; ++sidx) {
                const float4 av = img[sidx * 64];
                const float b = D[sidx >> 4][sidx & 15];
                G[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b, G[0], 0, 0, 0);
                G[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b, G[1], 0, 0, 0);
                G[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b, G[2], 0, 0, 0);
            }
            // G of natural row rr (compile-time) if this lane half holds it, else 0
            auto pick = [&](int rr) __attribute__((always_inline)) -> double {
                const int m = rr >> 5, x = rr & 31, hb = (x >> 2) & 1, i = (x & 3) + 4 * (x >> 3);
                return lh == hb ? (double)G[m][i] : 0.0;
            };
            // the query point, as the forward forms it (model_helpers.py:117-120)
            const float zz = a.z[ray * a.zs_r + smp];
            float p[3];
            double dd[3];
            if (a.ray_is_f64) {
                const double* o = reinterpret_cast<const double*>(a.origins) + ray * 3;
                const double* d = reinterpret_cast<const double*>(a.dirs) + ray * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) { p[c] = (float)__dadd_rn(o[c], __dmul_rn(d[c], (double)zz)); dd[c] = d[c]; }
            } else {
                const float* o = reinterpret_cast<const float*>(a.origins) + ray * 3;
                const float* d = reinterpret_cast<const float*>(a.dirs) + ray * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) { p[c] = __fadd_rn(o[c], __fmul_rn(d[c], zz)); dd[c] = (double)d[c]; }
            }
            double dp[3] = {0.0, 0.0, 0.0};
            if (nn.enc_mode == NCA_ENC_FOURIER) {
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    if (i < 3 * nn.L) {
                        const int c = i % 3;
                        const float coef = nn.four[i];
                        const float v = __fmul_rn(__fmul_rn(6.283185482025146484375f, p[c]), coef);
                        const double dv = 6.283185482025146484375 * (double)coef;
                        double sv, cv;
                        sincos((double)v, &sv, &cv);
                        // rows i (sin) and 3 L + i (cos): the second index is not a compile-time constant -> both halves via LDS-free select
                        double gs = pick(i), gc = 0.0;
#pragma unroll
                        for (int LL = 1; LL <= 16; ++LL) if (nn.L == LL && 3 * LL + i < 96) gc = pick(3 * LL + i);
                        dp[c] += dv * (cv * gs - sv * gc);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3; ++c) dp[c] = pick(c);
                if (nn.enc_mode == NCA_ENC_BANDS) {
                    double sn[3], cs[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) sincos((double)p[c], &sn[c], &cs[c]);
#pragma unroll
                    for (int k = 0; k < 15; ++k) {
                        if (k < nn.L) {
                            const double wk = nn.win ? (double)nn.win[k] : 1.0;
                            const float sc = ldexpf(1.f, k);
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                // d/dx sin(xb) = cos(xb);  d/dx sin(fl32(xb + fl32(pi/2))) = cos(xb + pi/2 + eps) = -(sin(xb) cos eps + cos(xb) sin eps)
                                const float xb = __fmul_rn(p[c], sc);
                                const float u = __fadd_rn(xb, 1.57079637050628662109375f);
                                const double eps = ((double)u - (double)xb) - 1.57079632679489661923;
                                const double e2 = eps * eps;
                                const double se = eps * (1.0 - e2 * (1.0 / 6.0)), ce = 1.0 - e2 * (0.5 - e2 * (1.0 / 24.0));
                                const double dcos = -(sn[c] * ce + cs[c] * se);
                                dp[c] += wk * (double)sc * (cs[c] * pick(3 + 6 * k + c) + dcos * pick(6 + 6 * k + c));
                                const double s2 = 2.0 * sn[c] * cs[c], c2 = 1.0 - 2.0 * sn[c] * sn[c];
                                sn[c] = s2; cs[c] = c2;
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) dp[c] += __shfl_xor(dp[c], 32);
            const double gz = dd[0] * dp[0] + dd[1] * dp[1] + dd[2] * dp[2];
            if (valid && lh == 0) {
                float* dst = a.g_z + ray * a.S + smp;
                *dst = first ? (float)gz : *dst + (float)gz;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// d loss / d latent INPUT per point (Temporal.query_time with latent vectors that require a gradient, model/Temporal.py:113-136:
// autograd gives every point its own  W0[:, Kenc .. Kenc + T)^T D_0[:, n]).  One wave per 32-sample tile of the point backward's
// chunk scratch: lane (r, h) holds its rows of D_0 of sample r (read exactly as nca_zgrad_f32 reads them, f32 quads or bf16
// fragments), multiplies with the latent columns of W0 staged in LDS, and the two lane halves are added.  T * F / 2 FMAs per lane.
// ------------------------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(256) void nca_latgrad_f32(const NcaLatgradArgs a) {
    constexpr int MT = F / 32;
    extern __shared__ __attribute__((aligned(16))) float lw[];        // [F][T]: W0[o][Kenc + t]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    for (int i = threadIdx.x; i < F * a.T; i += 256) lw[i] = a.w0[(int64_t)(i / a.T) * a.ldw + a.Kenc + (i % a.T)];
    __syncthreads();
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < a.ntiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t n = a.n0 + tile * 32 + lr;
        float D[MT][16];
        if (a.bf16) {
            const char* db = reinterpret_cast<const char*>(a.dscratch) + tile * a.d_total + a.drow + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 2 * MT; ++ks) {
                const uint4 w4 = *reinterpret_cast<const uint4*>(db + ks * 1024);
                const unsigned ww[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    D[ks >> 1][8 * (ks & 1) + 2 * u] = __uint_as_float(ww[u] << 16);
                    D[ks >> 1][8 * (ks & 1) + 2 * u + 1] = __uint_as_float(ww[u] & 0xffff0000u);
                }
            }
        } else {
            const float* df = a.dscratch + (tile * a.d_total + a.drow) * 32 + lane * 4;
#pragma unroll
            for (int mq = 0; mq < 4 * MT; ++mq) {
                const float4 dv = *reinterpret_cast<const float4*>(df + mq * 256);
                D[mq >> 2][4 * (mq & 3)] = dv.x; D[mq >> 2][4 * (mq & 3) + 1] = dv.y;
                D[mq >> 2][4 * (mq & 3) + 2] = dv.z; D[mq >> 2][4 * (mq & 3) + 3] = dv.w;
            }
        }
        for (int t = 0; t < a.T; ++t) {
            float s = 0.f;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = 32 * m + (i & 3) + 8 * (i >> 2) + 4 * lh;          // accumulator register i of row tile m (both scratch formats)
                    s = fmaf(lw[row * a.T + t], D[m][i], s);
                }
            s += __shfl_xor(s, 32);
            if (lh == 0 && n < a.N) a.g_lat[n * a.T + t] = s;
        }
    }
}
template <int F>
static hipError_t launch_latgrad(const NcaLatgradArgs& a, hipStream_t st) {
    const int lds = F * a.T * (int)sizeof(float);
    int64_t blocks = (a.ntiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(nca_latgrad_f32<F>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}
hipError_t nca_launch_latgrad_f32(int F, const NcaLatgradArgs& a, hipStream_t st) {
    switch (F) {
        case 32: return launch_latgrad<32>(a, st);
        case 64: return launch_latgrad<64>(a, st);
        case 128: return launch_latgrad<128>(a, st);
    }
    return hipErrorInvalidValue;
}

template <int F>
static hipError_t launch_zgrad(const NcaZgradArgs& a, hipStream_t st) {
    const int lds = 16 * (F / 32) * 64 * 4 * (int)sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_zgrad_f32<F>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int64_t blocks = (a.ntiles + 3) / 4;
    if (blocks > 512) blocks = 512;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(nca_zgrad_f32<F>, dim3((unsigned)blocks), dim3(256), lds, st, a);
    return hipGetLastError();
}
hipError_t nca_launch_zgrad_f32(const NcaZgradArgs& a, hipStream_t st) {
    for (int n = 1; n < a.nnets; ++n) if (a.net[n].F != a.net[0].F) return hipErrorInvalidValue;
    switch (a.net[0].F) {
        case 32: return launch_zgrad<32>(a, st);
        case 64: return launch_zgrad<64>(a, st);
        case 128: return launch_zgrad<128>(a, st);
    }
    return hipErrorInvalidValue;
}
