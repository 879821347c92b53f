// nca_kernels_wide.hip -- gfx950 kernels of the GENERAL path (nca_wide.hpp): nets wider than 128 units or with other channel counts
// than 3 -> 1, layer by layer with row-major activations in HBM.
//
//   nca_wide_encode   query point (rays or points) -> positional encoding (model/CPPN.py:112-135) -> X0 rows
//   nca_wide_pack     natural weights -> fan-in-padded [F][Kp] images
//   nca_wide_gemm<K>  128 x 128 x 16 LDS-tiled f32 GEMM on v_mfma_f32_32x32x2_f32; K = forward (+ bias, ReLU), dgrad (x ReLU mask), wgrad
//                     (contraction over the samples, split over workgroups)
//   nca_wide_head_*   the output layer (F -> num_output_channels) and its backward
//   nca_wide_colsum   bias gradients / output-layer weight gradients: (weighted) column sums, split over workgroups
//   nca_wide_reduce   fixed-order sum of the splits -> natural flat gradient
//
// Every sum over samples has a fixed order (splits by position, tree inside a workgroup by thread index): results are bit-identical run
// to run, as on the fused path.
#include <hip/hip_runtime.h>
#include "nca_kernels.hpp"
#include "nca_wide.hpp"

typedef float wf32x16 __attribute__((ext_vector_type(16)));

#define NCA_HALF_PI_F 1.57079637050628662109375f   // fl32(0.5 * pi), the constant the reference adds
#define NCA_HALF_PI_D 1.57079632679489661923
#define NCA_TWO_PI_F 6.283185482025146484375f      // fl32(2 * pi)

// ------------------------------------------------------------------------------------------ encode
// One thread per sample.  Same arithmetic as the fused f32 kernels' enc_steps (nca_kernels_f32.hip): bands by angle doubling in f64 from
// one sincos per coordinate, the reference's rounded "+ pi/2" reproduced through eps = fl32(xb + c) - xb - pi/2.
__global__ __launch_bounds__(256) void nca_wide_encode(const NcaWideEncArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.rows) return;
    float* row = a.X0 + i * a.K0p;
    if (i >= a.n_valid) {
        for (int k = 0; k < a.K0p; ++k) row[k] = 0.f;
        return;
    }
    const int64_t n = a.n0 + i;
    const int C = a.g.C;
    float p[NCA_WIDE_MAX_C];
    int ph = 0;
    if (a.g.mode == NCA_MODE_RAYS) {
        const int64_t ray = n / a.g.S;
        const int smp = (int)(n - ray * a.g.S);
        const float zz = a.g.z[ray * a.g.zs_r + smp];
        if (a.g.ray_is_f64) {
            const double* o = reinterpret_cast<const double*>(a.g.origins) + ray * 3;
            const double* d = reinterpret_cast<const double*>(a.g.dirs) + ray * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = (float)__dadd_rn(o[c], __dmul_rn(d[c], (double)zz));
        } else {
            const float* o = reinterpret_cast<const float*>(a.g.origins) + ray * 3;
            const float* d = reinterpret_cast<const float*>(a.g.dirs) + ray * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = __fadd_rn(o[c], __fmul_rn(d[c], zz));
        }
        if (a.g.phase) ph = a.g.phase[ray * a.g.ps_r + (int64_t)smp * a.g.ps_s];
    } else {
        for (int c = 0; c < C; ++c) p[c] = a.g.pts[n * C + c];
        if (a.g.phase) ph = a.g.phase[n];
    }
    int k = 0;
    if (a.enc_mode != NCA_ENC_FOURIER)
        for (int c = 0; c < C; ++c) row[k++] = p[c];
    if (a.enc_mode == NCA_ENC_BANDS) {
        double sn[NCA_WIDE_MAX_C], cs[NCA_WIDE_MAX_C];
        for (int c = 0; c < C; ++c) sincos((double)p[c], &sn[c], &cs[c]);
        float scl = 1.f;
        for (int b = 0; b < a.L; ++b) {
            const float w = a.win ? a.win[b] : 1.f;
            for (int c = 0; c < C; ++c) {
                const float xb = p[c] * scl;                      // exact (power of two)
                const float t = __fadd_rn(xb, NCA_HALF_PI_F);     // the reference's rounded argument
                const double eps = ((double)t - (double)xb) - NCA_HALF_PI_D;
                const double e2 = eps * eps;
                const double cf = cs[c] * (1.0 - 0.5 * e2) - sn[c] * (eps - eps * e2 * (1.0 / 6.0));
                row[k + c] = w * (float)sn[c];
                row[k + C + c] = w * (float)cf;
            }
            k += 2 * C;
            for (int c = 0; c < C; ++c) {
                const double s2 = 2.0 * sn[c] * cs[c];
                const double c2 = 1.0 - 2.0 * sn[c] * sn[c];
                sn[c] = s2; cs[c] = c2;
            }
            scl *= 2.f;
        }
    } else if (a.enc_mode == NCA_ENC_FOURIER) {
        const int nf = C * a.L;
        for (int q = 0; q < nf; ++q) {
            const float v = __fmul_rn(__fmul_rn(NCA_TWO_PI_F, p[q % C]), a.four[q]);
            double sv, cv;
            sincos((double)v, &sv, &cv);
            row[q] = (float)sv;
            row[nf + q] = (float)cv;
        }
        k = 2 * nf;
    }
    if (a.T > 0) {
        const int phc = ph < 0 ? 0 : (ph >= a.P ? a.P - 1 : ph);
        for (int t = 0; t < a.T; ++t) row[k++] = a.lat[phc * a.T + t];
        for (int q = 0; q < a.P; ++q) row[k++] = (a.onehot && q == phc) ? 1.f : 0.f;
    }
    for (; k < a.K0p; ++k) row[k] = 0.f;
}
hipError_t nca_launch_wide_encode(const NcaWideEncArgs& a, hipStream_t st) {
    if (a.rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(nca_wide_encode, dim3((unsigned)((a.rows + 255) / 256)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ pack
__global__ __launch_bounds__(256) void nca_wide_pack(const NcaWideLayout y, const float* __restrict__ prm, float* __restrict__ out) {
    const int j = blockIdx.y;
    if (j == y.NL) {                                              // biases, output layer
        const int64_t nb = (int64_t)y.NL * y.F, nw = (int64_t)y.Cout * y.F, total = y.packed_floats - y.pb_off;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
            float v = 0.f;
            if (i < nb) v = prm[y.layer[i / y.F].b_off + i % y.F];
            else if (i < nb + nw) v = prm[y.wo_off + (i - nb)];
            else if (i < nb + nw + y.Cout) v = prm[y.bo_off + (i - nb - nw)];
            out[y.pb_off + i] = v;
        }
        return;
    }
    const NcaWideLayer& l = y.layer[j];
    const int64_t total = (int64_t)y.F * l.Kp;
    float* dst = out + l.pw_off;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int f = (int)(i / l.Kp), c = (int)(i - (int64_t)f * l.Kp);
        int nat = -1;                                         // natural column of padded column c
        if (l.kind == NCA_IN_SKIP) nat = c < y.K0 ? c : (c >= y.K0p ? y.K0 + (c - y.K0p) : -1);
        else if (c < l.K) nat = c;
        dst[i] = nat >= 0 ? prm[l.w_off + (int64_t)f * l.K + nat] : 0.f;
    }
}
hipError_t nca_launch_wide_pack(const NcaWideLayout& y, const float* prm, float* out, hipStream_t st) {
    hipLaunchKernelGGL(nca_wide_pack, dim3(64, (unsigned)y.NL + 1), dim3(256), 0, st, y, prm, out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ GEMM
// Workgroup: 256 threads = 4 waves in 2 x 2, each wave a 64 x 64 block of the 128 x 128 tile = 2 x 2 MFMA blocks of 32 x 32 (64 accumulator
// registers).  Operand tiles sit in LDS k-major, As[k][r] / Bs[k][c] with a row pitch of 132 floats: lane (j, h) of a k-step reads As[2 s + h][r0 + j]
// -- 32 consecutive floats per half wave, no bank conflict on the reads (the four-scalar stash of an operand that is k-contiguous in memory does conflict two ways:
// 3 - 5 % of a launch by SQ_LDS_BANK_CONFLICT; the row-major alternative with ds_read_b128 operand reads was measured and is slower).  Global loads are
// float4 along whichever index is contiguous in memory; the next k-slab is fetched into registers while the current one is multiplied (one LDS buffer pair, two
// barriers per slab).
#define WG_BM 128
#define WG_BN 128
#define WG_BK 16
#define WG_PITCH 132

template <bool KC>      // KC: element (r, k) at base[r * ld + k]; else at base[k * ld + r]
__device__ __forceinline__ void wg_fetch(const float* __restrict__ base, int64_t ld, int64_t r0, int64_t rlim, int64_t k0, int tid, float4 (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        if constexpr (KC) {
            const int r = idx >> 2, kq = idx & 3;
            v[i] = (r0 + r < rlim) ? *reinterpret_cast<const float4*>(base + (r0 + r) * ld + k0 + 4 * kq) : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            const int k = idx >> 5, rq = idx & 31;
            v[i] = (r0 + 4 * rq < rlim) ? *reinterpret_cast<const float4*>(base + (k0 + k) * ld + r0 + 4 * rq) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}
template <bool KC>
__device__ __forceinline__ void wg_stash(float* __restrict__ s, int tid, const float4 (&v)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + 256 * i;
        if constexpr (KC) {
            const int r = idx >> 2, kq = idx & 3;
            s[(4 * kq + 0) * WG_PITCH + r] = v[i].x;
            s[(4 * kq + 1) * WG_PITCH + r] = v[i].y;
            s[(4 * kq + 2) * WG_PITCH + r] = v[i].z;
            s[(4 * kq + 3) * WG_PITCH + r] = v[i].w;
        } else {
            const int k = idx >> 5, rq = idx & 31;
            *reinterpret_cast<float4*>(s + k * WG_PITCH + 4 * rq) = v[i];
        }
    }
}

template <int KIND>
__global__ __launch_bounds__(256, KIND == NCA_WG_WGRAD ? 4 : 2) void nca_wide_gemm(const NcaWideGemmArgs a) {          // (all three run four workgroups per CU; wgrad needs telling to stay within 128 registers)
    constexpr bool AKC = KIND != NCA_WG_WGRAD, BKC = KIND == NCA_WG_FWD;
    __shared__ __attribute__((aligned(16))) float As[WG_BK * WG_PITCH];
    __shared__ __attribute__((aligned(16))) float Bs[WG_BK * WG_PITCH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lj = lane & 31, lh = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    // XCD-aware tile order.  Workgroups are dealt to the 8 XCDs round-robin by their linear id, and each XCD has its own L2: the workgroups that read the
    // SAME operand tile -- the column tiles of one row tile (forward, dgrad: the A rows), all output tiles of one sample split (wgrad: both operands) -- are
    // made neighbours ON ONE XCD (ids l, l + 8, l + 16, ...) instead of neighbours in id, which would spread them over the XCDs (at 256 units, per 262 144-row launch:
    // forward 472 -> 241 MB read, dgrad 812 -> 549, wgrad 1 074 -> 537; profiles/r06_wide_pmc.json).  Groups are dealt to the XCDs in turn when their number is a
    // multiple of 8; otherwise in id order.
    const int nrt = (int)((a.rows + WG_BM - 1) / WG_BM), nct = (int)((a.cols + WG_BN - 1) / WG_BN);
    const int gs = KIND == NCA_WG_WGRAD ? nrt * nct : nct;                       // workgroups per group
    const int64_t ng = KIND == NCA_WG_WGRAD ? a.nsplit : nrt;                   // groups
    const int64_t lid = blockIdx.x;
    int64_t group;
    int member;
    if ((ng & 7) == 0) {
        const int64_t q = lid >> 3;
        member = (int)(q % gs);
        group = (q / gs) * 8 + (lid & 7);
    } else {
        member = (int)(lid % gs);
        group = lid / gs;
    }
    const int rt = KIND == NCA_WG_WGRAD ? member % nrt : (int)group, ct = KIND == NCA_WG_WGRAD ? member / nrt : member;
    const int64_t zsplit = KIND == NCA_WG_WGRAD ? group : 0;
    const int64_t r0 = (int64_t)rt * WG_BM, c0 = (int64_t)ct * WG_BN;

    // contraction range of this workgroup: [kb, ke) over the concatenation of A's two segments
    const int64_t ktot = a.ka[0] + a.ka[1];
    int64_t kb = 0, ke = ktot;
    if (KIND == NCA_WG_WGRAD) {
        const int64_t per = ((ktot + a.nsplit - 1) / a.nsplit + WG_BK - 1) / WG_BK * WG_BK;
        kb = zsplit * per;
        ke = kb + per < ktot ? kb + per : ktot;
    }
    wf32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (wf32x16)(0.f);

    auto fetch = [&](int64_t k, float4 (&va)[2], float4 (&vb)[2]) __attribute__((always_inline)) {
        const int seg = k >= a.ka[0] ? 1 : 0;          // (ka[0] is a multiple of 16: a slab never straddles the segments)
        const int64_t kk = seg ? k - a.ka[0] : k;
        wg_fetch<AKC>(a.A[seg], a.lda[seg], r0, a.rows, kk, tid, va);
        wg_fetch<BKC>(a.B, a.ldb, c0, a.cols, k, tid, vb);
    };
    const bool sums = KIND == NCA_WG_WGRAD && a.rowsum && ct == 0;          // (uniform per workgroup)
    float rsum = 0.f;
    if (kb < ke) {
        float4 va[2], vb[2];
        fetch(kb, va, vb);
        for (int64_t k = kb; k < ke; k += WG_BK) {
            __syncthreads();                           // the previous slab's reads are done
            wg_stash<AKC>(As, tid, va);
            wg_stash<BKC>(Bs, tid, vb);
            __syncthreads();
            if (k + WG_BK < ke) fetch(k + WG_BK, va, vb);
            if (sums && tid < WG_BM) {
#pragma unroll
                for (int kk = 0; kk < WG_BK; ++kk) rsum += As[kk * WG_PITCH + tid];          // (slab by slab, k ascending: a fixed order)
            }
#pragma unroll
            for (int s = 0; s < WG_BK / 2; ++s) {
                const float* ar = As + (2 * s + lh) * WG_PITCH + wr * 64 + lj;
                const float* br = Bs + (2 * s + lh) * WG_PITCH + wc * 64 + lj;
                const float a0 = ar[0], a1 = ar[32], b0 = br[0], b1 = br[32];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    }
    if (sums && tid < WG_BM && r0 + tid < a.rows) a.rowsum[zsplit * a.split_stride + r0 + tid] = rsum;
    // epilogue: register v of lane (j, h) of block (bi, bj) = C[r0 + 64 wr + 32 bi + 8 (v >> 2) + 4 h + (v & 3)][c0 + 64 wc + 32 bj + j].  One 64-bit base per lane,
    // 32-bit offsets inside the tile (a tile spans at most 128 rows of at most 2^20 floats)
    float* Cb = a.C + (KIND == NCA_WG_WGRAD ? zsplit * a.split_stride : 0);
    const int64_t rl = r0 + wr * 64 + 4 * lh, cl = c0 + wc * 64 + lj;          // this lane's first row / column
    float* const cp = Cb + rl * a.ldc + cl;
    const int ldc = (int)a.ldc;
    uint32_t* const mbp = (KIND != NCA_WG_WGRAD && a.maskbits) ? a.maskbits + (((int64_t)rt * nct + ct) * 256 + tid) * 2 : nullptr;
    uint32_t mb[2] = {0xffffffffu, 0xffffffffu};
    if (KIND == NCA_WG_DGRAD && mbp) {
        const uint2 t = *reinterpret_cast<const uint2*>(mbp);
        mb[0] = t.x; mb[1] = t.y;
    }
    uint32_t mo[2] = {0u, 0u};
    const int rleft = (int)(a.rows - rl < 128 ? a.rows - rl : 128), cleft = (int)(a.cols - cl < 128 ? a.cols - cl : 128);          // valid offsets: < these
#pragma unroll
    for (int bj = 0; bj < 2; ++bj) {
        if (bj * 32 >= cleft) continue;
        float bias = 0.f;
        if (KIND == NCA_WG_FWD && a.bias) bias = a.bias[cl + bj * 32];
#pragma unroll
        for (int bi = 0; bi < 2; ++bi) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int ro = bi * 32 + 8 * (v >> 2) + (v & 3);
                if (ro >= rleft) continue;
                float x = acc[bi][bj][v];
                if (KIND == NCA_WG_FWD) {
                    x += bias;
                    if (a.relu) x = x > 0.f ? x : 0.f;
                    if (x > 0.f) mo[bi] |= 1u << (bj * 16 + v);
                } else if (KIND == NCA_WG_DGRAD) {
                    x = (mb[bi] >> (bj * 16 + v)) & 1u ? x : 0.f;
                }
                cp[ro * ldc + bj * 32] = x;
            }
        }
    }
    if (KIND == NCA_WG_FWD && mbp) *reinterpret_cast<uint2*>(mbp) = make_uint2(mo[0], mo[1]);
}
hipError_t nca_launch_wide_gemm(int kind, const NcaWideGemmArgs& a, hipStream_t st) {
    if (a.rows <= 0 || a.cols <= 0) return hipSuccess;
    const int64_t tiles = ((a.rows + WG_BM - 1) / WG_BM) * ((a.cols + WG_BN - 1) / WG_BN);
    const dim3 grid((unsigned)(tiles * (kind == NCA_WG_WGRAD ? a.nsplit : 1)));
    if (kind == NCA_WG_FWD) hipLaunchKernelGGL(nca_wide_gemm<NCA_WG_FWD>, grid, dim3(256), 0, st, a);
    else if (kind == NCA_WG_DGRAD) hipLaunchKernelGGL(nca_wide_gemm<NCA_WG_DGRAD>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(nca_wide_gemm<NCA_WG_WGRAD>, grid, dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ output layer
// 16 lanes per row (float4 each, F / 64 rounds), four rows per wave pass
__global__ __launch_bounds__(256) void nca_wide_head_fwd(int64_t n_valid, int F, int Cout, const float* __restrict__ H, const float* __restrict__ Wo,
                                                         const float* __restrict__ bo, float* __restrict__ raw) {
    const int sub = threadIdx.x & 15;
    const int64_t row = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    const bool ok = row < n_valid;
    const float* h = H + (ok ? row : 0) * F;
    for (int o = 0; o < Cout; ++o) {
        const float* w = Wo + (int64_t)o * F;
        float s = 0.f;
        for (int f = 4 * sub; f < F; f += 64) {
            const float4 hv = *reinterpret_cast<const float4*>(h + f);
            const float4 wv = *reinterpret_cast<const float4*>(w + f);
            s += hv.x * wv.x;
            s += hv.y * wv.y;
            s += hv.z * wv.z;
            s += hv.w * wv.w;
        }
        s += __shfl_xor(s, 8);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 1);
        if (ok && sub == 0) raw[row * Cout + o] = s + bo[o];
    }
}
hipError_t nca_launch_wide_head_fwd(int64_t n_valid, int F, int Cout, const float* H, const float* Wo, const float* bo, float* raw, hipStream_t st) {
    if (n_valid <= 0) return hipSuccess;
    hipLaunchKernelGGL(nca_wide_head_fwd, dim3((unsigned)((n_valid * 16 + 255) / 256)), dim3(256), 0, st, n_valid, F, Cout, H, Wo, bo, raw);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void nca_wide_head_bwd(int64_t n_valid, int64_t rows, int F, int Cout, const float* __restrict__ H, const float* __restrict__ Wo,
                                                         const float* __restrict__ g, float* __restrict__ D) {
    const int fq = F >> 2;
    const int64_t total = rows * fq;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / fq;
        const int f = (int)(i - n * fq) * 4;
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < n_valid) {
            const float4 hv = *reinterpret_cast<const float4*>(H + n * F + f);
            for (int o = 0; o < Cout; ++o) {
                const float gv = g[n * Cout + o];
                const float4 wv = *reinterpret_cast<const float4*>(Wo + (int64_t)o * F + f);
                d.x += gv * wv.x; d.y += gv * wv.y; d.z += gv * wv.z; d.w += gv * wv.w;
            }
            d.x = hv.x > 0.f ? d.x : 0.f;
            d.y = hv.y > 0.f ? d.y : 0.f;
            d.z = hv.z > 0.f ? d.z : 0.f;
            d.w = hv.w > 0.f ? d.w : 0.f;
        }
        *reinterpret_cast<float4*>(D + n * F + f) = d;
    }
}
hipError_t nca_launch_wide_head_bwd(int64_t n_valid, int64_t rows, int F, int Cout, const float* H, const float* Wo, const float* g, float* D, hipStream_t st) {
    if (rows <= 0) return hipSuccess;
    const int64_t total = rows * (F / 4);
    const int64_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(nca_wide_head_bwd, dim3((unsigned)(nb < 65536 ? nb : 65536)), dim3(256), 0, st, n_valid, rows, F, Cout, H, Wo, g, D);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ column sums
// grid (ceil(F / 64), nsplit); thread (cq, rl) = 16 column quads x 16 row lanes; rows of split s: [s * per, (s + 1) * per)
__global__ __launch_bounds__(256) void nca_wide_colsum(int64_t rows, int F, int Cout, const float* __restrict__ X, int64_t ldx, const float* __restrict__ g,
                                                       float* __restrict__ out, int64_t split_stride, float* __restrict__ gsum) {
    __shared__ float4 red[16][16];
    __shared__ float gred[256];
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int f = blockIdx.x * 64 + 4 * cq;
    const int nsplit = gridDim.y;
    const int64_t per = (rows + nsplit - 1) / nsplit;
    const int64_t nb = (int64_t)blockIdx.y * per, ne = nb + per < rows ? nb + per : rows;
    float* o_base = out + (int64_t)blockIdx.y * split_stride;
    const int nout = g ? Cout : 1;
    for (int o = 0; o < nout; ++o) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        float gs = 0.f;
        if (f < F) {
            for (int64_t n = nb + rl; n < ne; n += 16) {
                const float4 x = *reinterpret_cast<const float4*>(X + n * ldx + f);
                const float w = g ? g[n * Cout + o] : 1.f;
                s.x += w * x.x; s.y += w * x.y; s.z += w * x.z; s.w += w * x.w;
            }
        }
        if (gsum && blockIdx.x == 0)
            for (int64_t n = nb + threadIdx.x; n < ne; n += 256) gs += g[n * Cout + o];
        red[rl][cq] = s;
        gred[threadIdx.x] = gs;
        __syncthreads();
        for (int d = 8; d >= 1; d >>= 1) {
            if (rl < d) {
                float4 u = red[rl][cq];
                const float4 t = red[rl + d][cq];
                u.x += t.x; u.y += t.y; u.z += t.z; u.w += t.w;
                red[rl][cq] = u;
            }
            __syncthreads();
        }
        if (gsum && blockIdx.x == 0) {
            for (int d = 128; d >= 1; d >>= 1) {
                if ((int)threadIdx.x < d) gred[threadIdx.x] += gred[threadIdx.x + d];
                __syncthreads();
            }
            if (threadIdx.x == 0) gsum[(int64_t)blockIdx.y * split_stride + o] = gred[0];
        }
        if (rl == 0 && f < F) *reinterpret_cast<float4*>(o_base + (int64_t)o * F + f) = red[0][cq];
        __syncthreads();
    }
}
hipError_t nca_launch_wide_colsum(int64_t rows, int F, int Cout, const float* X, int64_t ldx, const float* g, float* out, int64_t split_stride, float* gsum, int nsplit,
                                  hipStream_t st) {
    hipLaunchKernelGGL(nca_wide_colsum, dim3((unsigned)((F + 63) / 64), (unsigned)nsplit), dim3(256), 0, st, rows, F, Cout, X, ldx, g, out, split_stride, gsum);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ reduce
// one thread per natural parameter (behind them F * P threads for the per-phase sums of D_0): grads[i] += sum over the splits, in split order
__global__ __launch_bounds__(256) void nca_wide_reduce(const NcaWideReduceArgs a) {
    const NcaWideLayout& y = a.y;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t n_e = (int64_t)y.F * y.P;
    if (i >= y.n_params + n_e) return;
    int64_t src = -1;
    if (i >= y.n_params) {                                    // E[f][p] = sum_n D_0[n][f] onehot[n][p]: the one-hot columns of layer 0's weight gradient
        const int64_t e = i - y.n_params;
        const int f = (int)(e / y.P), p = (int)(e - (int64_t)f * y.P);
        src = y.layer[0].pw_off + (int64_t)f * y.K0p + y.K0 + p;
    } else if (i < y.lat_off + (int64_t)y.P * y.T) {
        return;                                               // time latents: nca_wide_latgrad
    } else if (i >= y.bo_off) {
        src = y.pbo_off + (i - y.bo_off);
    } else if (i >= y.wo_off) {
        src = y.pwo_off + (i - y.wo_off);
    } else {
        for (int j = 0; j < y.NL; ++j) {
            const NcaWideLayer& l = y.layer[j];
            if (i >= l.w_off && i < l.b_off) {
                const int64_t e = i - l.w_off;
                const int f = (int)(e / l.K), c = (int)(e - (int64_t)f * l.K);
                src = l.pw_off + (int64_t)f * l.Kp + nca_wide_col(y, l, c);
                break;
            }
            if (i >= l.b_off && i < l.b_off + y.F) {
                src = y.pb_off + (int64_t)j * y.F + (i - l.b_off);
                break;
            }
        }
    }
    if (src < 0) return;
    float s = 0.f;
    for (int k = 0; k < a.nsplit; ++k) s += a.slab[(int64_t)k * a.split_stride + src];
    if (i >= y.n_params) a.esum[i - y.n_params] = s;
    else a.grads[i] += s;
}
// d loss / d time_latents[p][t] += sum_f W0[f][Kenc + t] E[f][p]
__global__ __launch_bounds__(64) void nca_wide_latgrad(const NcaWideReduceArgs a) {
    const NcaWideLayout& y = a.y;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= y.P * y.T) return;
    const int p = i / y.T, t = i - p * y.T;
    const float* w0 = a.packed + y.layer[0].pw_off;
    float s = 0.f;
    for (int f = 0; f < y.F; ++f) s += w0[(int64_t)f * y.K0p + y.Kenc + t] * a.esum[(int64_t)f * y.P + p];
    a.grads[y.lat_off + i] += s;
}
hipError_t nca_launch_wide_reduce(const NcaWideReduceArgs& a, hipStream_t st) {
    const int64_t n = a.y.n_params + (int64_t)a.y.F * a.y.P;
    hipLaunchKernelGGL(nca_wide_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (a.y.T > 0) {
        hipLaunchKernelGGL(nca_wide_latgrad, dim3((unsigned)((a.y.P * a.y.T + 63) / 64)), dim3(64), 0, st, a);
        e = hipGetLastError();
    }
    return e;
}
