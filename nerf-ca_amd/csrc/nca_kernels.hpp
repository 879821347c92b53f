// nca_kernels.hpp -- kernel argument blocks and launcher prototypes (internal).
#pragma once
#include <hip/hip_runtime.h>
#include "nca_layout.hpp"

#ifndef NCA_WAVES
#define NCA_WAVES 8     // waves per workgroup of the fused kernels: 2 per SIMD (tools/variant_build_all.sh w4 "-DNCA_WAVES=4": one per SIMD, for timing)
#endif
#define NCA_NT (64 * NCA_WAVES)
#define NCA_LDS_BYTES 163840   // LDS of a gfx950 compute unit (one workgroup of the fused kernels owns it)

enum { NCA_MODE_RAYS = 0, NCA_MODE_POINTS = 1 };

struct NcaStage {
    const void* ptr;   // image in the packed buffer
    uint32_t bytes;    // multiple of 16
    uint32_t lds_off;  // bf16 kernels with resident images: where this image sits in LDS (images back to back, 16-byte granularity)
};

struct NcaNetArgs {
    NcaLayout lay;
    const float* win;    // f32[L] band weights
    const float* four;   // f32[3L] fourier coefficients (or null)
    const float* lat;    // f32[P*T] time latents (or null)
    int64_t row0;        // first scratch row of this net (backward only); bf16: BYTE offset of its input/H blocks in a tile
    int64_t drow0;       // bf16: byte offset of its output-gradient (D) blocks in a tile of the D region
    const float* wo_src; // bf16 stored-forward backward: the [Wo | bo] tail of the packed last-layer image (global)
};

// kernel modes of the fused kernels
enum { NCA_KM_FWD = 0,          // forward
       NCA_KM_BWD = 1,          // recompute + output-layer gradients + dgrad; H, D and the input block go to ONE scratch
       NCA_KM_FWD_STORE = 2,    // forward that also writes the input block, every layer input, ReLU masks and raw outputs
       NCA_KM_BWD_STORED = 3,   // f32: output-layer gradients + dgrad from that store (no recompute); D to the chunk scratch
                                // (4 was the bf16-staged backward with one layer's weight gradient on chip: retired in round 4 with bf16 staging)
       NCA_KM_BWD_NR = 5 };     // bf16, from its (8-bit staged) store: NO recompute -- raw outputs and ReLU masks of all layers come
                                // from the store, D_{NL-1} = mask (Wo x g); the output layer's weight gradient is the wgrad kernel's

struct NcaFusedArgs {
    int32_t mode, nnets;
    int64_t ntiles;      // wave tiles (32 samples each) in this launch
    // rays
    int64_t ray0;        // first ray of this launch
    int32_t S, nchunk;   // samples per ray, tiles per ray
    int32_t ray_is_f64, act, single;
    float scale;
    const void* origins;
    const void* dirs;
    const int32_t* phase;
    int64_t ps_r, ps_s;
    const float* z;
    int64_t zs_r;
    const double* dists;
    // points
    int64_t N, n0;
    const float* pts;
    // outputs (forward)
    double* part;        // [ntiles] per-tile partial ray sums
    float* sig_s;
    float* sig_d;
    float* raw_out;
    // backward
    const double* g_pix;
    const float* g_sig_s;
    const float* g_sig_d;
    const float* g_raw;
    float* scratch;      // [tile][rows_total x 32 floats]: layer inputs H and output gradients D of each 32-sample tile
                         // (f32: input block row-major [row][32], hidden blocks [row tile][quad][lane][4]; bf16: see nca_bf_tile_bytes)
    int64_t rows_total;  // f32: scratch rows per 32-sample tile over all nets;  bf16: BYTES per 32-sample tile
    int32_t net_base;    // a launch that carries ONE net of a two-net render (nnets == 1): its index (0 static, 1 dynamic) for
                         // the per-net upstream gradient, mask region and output-layer partial slot
    int32_t share_enc;   // bf16, two nets with the same encoding (mode, bands, the SAME window / coefficient vectors): net 0 (static)
                         // does not store its input block -- net 1's is a superset (+ latents, one-hot phase slots) and net 0's
                         // layer-0 weight-gradient job reads that one
    // bf16 two-region addressing (NCA_KM_BWD: dscratch == scratch, d_total == rows_total, tile0 == 0)
    char* dscratch;      // D region of this launch: [local tile][d_total bytes]
    int64_t d_total;     // bytes per 32-sample tile of the D region
    int64_t tile0;       // wave-tile index of this launch's first tile inside the H region (stored forward: global)
    char* mstore;        // stored forward: ReLU masks [wave tile][net][layer][1 KiB]
    float* rstore;       // stored forward: raw outputs [wave tile][net][64]
    int32_t mstore_layers; // layers per net in mstore
    int32_t h8;          // the forward store holds the hidden blocks 0..NL-3 as e4m3 (nca_bf_hoff): what the storing forward
                         // writes (S8 instantiation) and where the backward finds the last layer's input
    int64_t dscale_off;  // S8 backward: byte offset of the inverse-scale record inside a tile of the D region
    float* oslab;        // [grid][2][F+1] output-layer gradient partials
    int32_t accumulate;  // add to oslab instead of overwriting (ray chunks after the first)
    int32_t nstages;
    int32_t mask_layers; // backward: ReLU masks of this many layers per wave are kept in LDS (0: re-read H)
    int32_t const_net_floats; // f32 kernels: floats per net of the LDS constant area (window, fourier, latents), set by the launcher
    int32_t ctr_off;     // bf16, resident images: byte offset in LDS of the workgroup's tile counter (set by the launcher) -- the waves claim
                         // tiles as they finish (the second wave of a SIMD runs ~25 % slower than the first)
    int32_t res_bytes;   // bf16: > 0 = every weight image of this launch stays resident in LDS (stage[i].lds_off), this many bytes
                         // in all
    int32_t res_total;   // host only: bytes of all images of the launch laid back to back (build_stages)
    int32_t split;       // bf16 forward, rays mode, one net per launch: 1 = static net (writes sig_s only), 2 = dynamic net (reads
                         // sig_s, writes sig_d and the per-tile ray sums)
    int32_t expand_last; // mode 5 with bf16 output gradients (the bf16 store): the last hidden layer's block is not written -- one word per
                         // sample, bf16(g) in both halves, where it would start; the weight-gradient job rebuilds it from the mask bits
    int32_t raw_only;    // rays mode, forward: write the raw net output to raw_out[n] instead of compositing
                         // (rays mode, backward: a non-null g_raw replaces the compositing chain rule)
    NcaNetArgs net[2];
    NcaStage stage[NCA_MAX_STAGES];
};

struct NcaWgradJob {
    int32_t F;            // rows of D
    int32_t b_rows_pad;   // rows of H rounded up to 32
    int32_t b_frag;       // f32: the H block is in the fused kernel's register order (hidden blocks), not row-major (input block)
    int64_t d_row0, b_row0;
    int32_t ncols_w;      // H rows that are real weight columns
    int32_t P;            // further H rows that are one-hot phase rows
    int64_t out_off;      // slab offset (floats) of W
    int32_t out_ld, out_col0;
    int64_t onehot_off;   // slab offset of the [F][P] one-hot block
    int64_t bias_off;     // slab offset of the bias gradient, or -1
    int32_t b_row_bytes;  // bf16 path: bytes of one sample row of the H block (d_row0/b_row0 are BYTE offsets in a tile there)
    int32_t is_enc, T;
    int32_t fourier_L;    // bf16 input block of a fourier net: slots are (sin_i, cos_i) interleaved; 0 otherwise
    int32_t d8, h8;       // bf16 path: the D block is e5m2 scaled by the wave tile's power of two / the H block is e4m3 x 2^NCA_H8_LOG2
    int32_t net;          // ... which of the tile's two inverse scales applies
    int32_t expand;       // mode 5, e5m2: the job's D block is not in the D region -- it is relu'(H_{NL-1}) x one byte per sample, rebuilt
                          // from the forward's mask bits and the sample's e5m2 byte (the first 128 bytes of the block's place: u32[32])
    int64_t dscale_off;   // ... byte offset of the inverse-scale record inside a tile of the D region
    int64_t mask_off;     // expand: byte offset of the layer's mask fragment inside a wave tile of the mask store
};

struct NcaWgradArgs {
    const float* scratch;   // [tile][rows_total][32]   (bf16: the D region, bytes)
    int64_t rows_total, ntiles;
    const float* scratch_b; // bf16: the region of the H / input blocks (== scratch in recompute mode) ...
    int64_t rows_total_b;   // ... its bytes per 32-sample tile ...
    int64_t tile0_b;        // ... and the 32-sample-tile index of this launch's first tile inside it
    const char* mask;       // the forward store's mask bits (expand jobs) and their bytes per wave tile
    int64_t mask_stride;
    float* slab;
    int64_t slab_stride;
    int32_t accumulate, njobs;
    int32_t nsplit_std, nsplit_x;   // e5m2 staging: splits of the regular jobs / of the `expand` jobs (>=; slab rows)
    NcaWgradJob job[NCA_MAX_JOBS];
};

struct NcaReduceNet {
    float* grads;
    const float* params;
    int64_t slab_off;     // where this net's natural block starts inside a slab
    int64_t onehot_off;
    int32_t F, T, P, K0, Kenc, w0_off;
    int64_t lat_count, wo_off;
    int32_t tail_from_sums; // bf16 with fp8 staging: the slabs hold S = sum relu' g H^T and s = sum relu' g of the LAST F-wide layer
    int32_t tl_K;           // (nca_layout.hpp): its gradients are Wo[f] S, Wo[f] s, and dWo[f] = <bf16(W[f]), S[f]> + b[f] s[f]; tl_K = that layer's fan-in
    int64_t tl_w_off, tl_b_off;   // natural offsets of that layer's W (F*F) and b (F)
    // per net, because the two nets' weight-gradient launches may be sized differently (one of them runs beside the other net's dgrad
    // launch on part of the chip, NCA_OPT_OVERLAP_CUS):
    int32_t n_split;        // slab rows of the columns the rebuilding (`expand`) jobs write -- the last F-wide layer under tail_from_sums
    int32_t n_split_std;    // slab rows of every other column (<= n_split; the rows beyond hold nothing for them and are never read)
    int32_t n_wg;           // workgroups of this net's dgrad launch = rows of oslab that hold its output-layer partials
    int32_t pad2;
};

struct NcaReduceArgs {
    int64_t n_total;
    int64_t n_params[2];
    float* slab;             // (not const: nca_reduce_f32 / nca_onehot_sum_f32 leave sums over the splits in slab row 0 for nca_reduce_small_f32,
                             //  which must be launched after them on the same stream)
    int64_t slab_stride;
    const float* oslab;
    int64_t oslab_stride;
    NcaReduceNet net[2];
    // the tile records of a one-chunk backward from a store (mode 5), summed into oslab's output-bias slots by extra workgroups of the FIRST
    // reduce launch instead of a launch of their own (rec_region == null: nothing to do / nca_sum_tile_records ran per chunk)
    const char* rec_region;
    int64_t rec_tile_bytes, rec_off, rec_ntiles;
    int32_t rec_net0, rec_net1, rec_F, rec_nwg;
    float* rec_oslab;
};
// body of nca_sum_tile_records (bf16 TU) / of the first reduce launch's extra workgroups (f32 TU): workgroup `wg` of `n_wg` adds the per-tile sums of
// d loss / d raw (tile order: thread t takes tiles t * n_wg + wg, + 256 n_wg, ...; then a fixed tree) to its output-bias slot of oslab
__device__ __forceinline__ void nca_tile_record_sum(const char* dregion, int64_t wave_tile_bytes, int64_t dscale_off, int64_t ntiles, int net0, int net1, int F,
                                                    float* oslab, int wg, int n_wg, float* part /* __shared__ float[256] */) {
    for (int net = net0; net < net1; ++net) {
        float s = 0.f;
        for (int64_t t = (int64_t)threadIdx.x * n_wg + wg; t < ntiles; t += 256 * (int64_t)n_wg)
            s += reinterpret_cast<const float*>(dregion + t * wave_tile_bytes + dscale_off)[2 + net];
        part[threadIdx.x] = s;
        __syncthreads();
        for (int d = 128; d >= 1; d >>= 1) {
            if ((int)threadIdx.x < d) part[threadIdx.x] += part[threadIdx.x + d];
            __syncthreads();
        }
        if (threadIdx.x == 0) oslab[(int64_t)wg * 2 * (F + 1) + net * (F + 1) + F] += part[0];
        __syncthreads();
    }
}

struct NcaLossArgs {
    int64_t R;
    int32_t S, use_weighting;
    double skew, mask_thre, weighted_thresh;
    double w_favor, w_dent, w_occl, w_l1, inv_R;
    const double* weights_dev;
    int32_t unit_mse, pad_;
    const double* pix; const double* gt; const double* wpix;
    const float* sig_s; const float* sig_d; const double* dists;
    double* terms; double* g_pix; float* g_sig_s; float* g_sig_d;
    double* partials;
    double* g_dists;      // [S] or null
    double* dists_work;   // [R * S] per-ray d loss / d dists
    const double* term_grads;   // null, or DEVICE f64[11]: term-gradient mode (NcaLoss.term_grads)
    const double* ray_part;     // null, or the forward's per-tile ray sums [R][ray_nchunk]: pix is formed here (NcaLoss.ray_part)
    const float* ray_I0;
    double* pix_out;
    int32_t ray_nchunk, pad2_;
    float* terms_f32;           // null, or f32[13]: the terms once more as floats
};
struct NcaCompositeArgs {
    int64_t R;
    int32_t S, act, single;
    float scale;
    const float* raw_s; const float* raw_d; const float* I0; const double* dists;
    double* pix; float* sig_s; float* sig_d;
    const double* g_pix; const float* g_sig_s; const float* g_sig_d; float* g_raw_s; float* g_raw_d;
};
hipError_t nca_launch_composite(const NcaCompositeArgs& a, bool bwd, hipStream_t st);
hipError_t nca_launch_loss(const NcaLossArgs& a, hipStream_t st);
hipError_t nca_launch_wsqerr(int64_t R, bool f64, const void* pred, const void* gt, const void* w, void* out, hipStream_t st);
hipError_t nca_launch_wsqerr_bwd(int64_t R, bool f64, const void* pred, const void* gt, const void* w, const void* g_out, void* g_pred, void* g_gt, void* g_w, hipStream_t st);
struct NcaAdamArgs {
    double lr, beta1, beta2, eps, lr_end_factor;
    int64_t lr_total_iters;
    int32_t n_seg;
    int64_t n[4];
    float* params[4]; const float* grads[4]; float* exp_avg[4]; float* exp_avg_sq[4];
    int64_t* step;           // i64[2]: steps taken, arrival counter of the launch's workgroups
    int64_t* iter_counter;   // null, or incremented with step[0]
};
hipError_t nca_launch_adam(const NcaAdamArgs& a, hipStream_t st);
hipError_t nca_launch_prepare_batch(int64_t R, int S, const int64_t* ids, const double* table, const int64_t* phases, int64_t n_rows, int32_t* bad_ids,
                                    const float* depth, const float* t_rand,
                                    double* o, double* d, double* gt, double* w, int32_t* ph, float* z, double* dists, hipStream_t st);
// nca_begin_step: nca_prepare_batch + the device-side draws and schedules (include/nerfca_hip.h)
struct NcaBeginArgs {
    NcaSampler s;
    int64_t slot0, R;
    int32_t S, pad;
    NcaSchedules sch;
    const int64_t* ids_in; const float* t_rand_in;
    const double* table; const int64_t* phases; int32_t* bad_ids; const float* depth;
    double* o; double* d; double* gt; double* w; int32_t* ph; float* z; double* dists;
    int64_t* ids_out; float* t_rand_out;
};
hipError_t nca_launch_begin_step(const NcaBeginArgs& a, hipStream_t st);
hipError_t nca_launch_draw_ray_ids(const NcaSampler& s, int64_t slot0, int64_t R, int64_t* ids, hipStream_t st);
hipError_t nca_launch_draw_uniform(const NcaSampler& s, int stream_id, int64_t n, float* out, hipStream_t st);
struct NcaFineArgs {
    int64_t R;
    int32_t S, n_fine;
    const float* sig_s; const float* sig_d; const float* z; const float* u;
    float* z_all;
    float* partial_max;   // [ceil(R / 4)]
    float* jmax;          // [1]
};
// backward of the fine-pass depths: d loss / d (sigma_s + sigma_d) of the coarse fields from d loss / d z_all
struct NcaFineBwdArgs {
    int64_t R;
    int32_t S, n_fine;
    const float* sig_s; const float* sig_d; const float* z; const float* u;
    const float* jmax;        // [1] the batch-wide maximum the forward sampled with
    const float* g_zall;      // [R, S + n_fine]
    float* g_tot;             // [R, S]  (written)
    float* gmax_part;         // [R] per-ray part of d loss / d jmax
    float* cnt_part;          // [R] per-ray number of jumps that attain jmax
    const float* gmax_each;   // [1] second stage: d loss / d jmax divided by the number of elements that attain it (all ranks)
};
hipError_t nca_launch_fine_bwd(const NcaFineBwdArgs& a, hipStream_t st);
hipError_t nca_launch_fine_bwd_max(const NcaFineBwdArgs& a, hipStream_t st);
hipError_t nca_launch_fine(const NcaFineArgs& a, hipStream_t st);
hipError_t nca_launch_fine_max(const NcaFineArgs& a, hipStream_t st);
hipError_t nca_launch_fine_sample(const NcaFineArgs& a, hipStream_t st);
int64_t nca_fine_partials(int64_t R);
int64_t nca_loss_partials_bytes(int64_t R);

// d loss / d depth of every sample (the reference differentiates the fine pass through its sampled depths: query point ->
// positional encoding -> first layer; train/model_helpers.py:147-148): from the first layer's output gradient D_0 that the f32
// backward leaves in its chunk scratch
struct NcaZgradNet {
    // the layers that read the encoded input: the first one and, in a net with late layers, the skip layer (cat[enc, h],
    // model/CPPN.py:99-103) -- natural weight [F][ldw] whose first Kenc columns meet the encoding, and the first row of the
    // layer's D block inside a tile of the D region
    const float* w[2];
    int32_t ldw[2];
    int64_t drow[2];
    int32_t nsrc, F, enc_mode, L, Kenc, pad;
    const float* win;     // band weights (or null)
    const float* four;    // fourier coefficients (or null)
};
struct NcaZgradArgs {
    int32_t nnets, S, nchunk, ray_is_f64;
    int64_t ntiles, ray0;
    const void* origins; const void* dirs;
    const float* z; int64_t zs_r;
    const float* dscratch; int64_t d_total;      // D region of this launch: f32 [tile][d_total rows][32]; bf16 [tile][d_total BYTES],
                                                 // blocks of fragments [k-step][lane][8 bf16] (drow = byte offset of the block)
    int32_t bf16, pad;                           // (ntiles / nchunk count 32-sample tiles in both precisions)
    float* g_z;                                  // [R][S]
    NcaZgradNet net[2];
};
hipError_t nca_launch_zgrad_f32(const NcaZgradArgs& a, hipStream_t st);
// d loss / d latent input per point from the D_0 blocks of a point backward's chunk scratch (nca_mlp_bwd with g_latents)
struct NcaLatgradArgs {
    int64_t ntiles, n0, N;          // 32-sample tiles of this chunk; index of its first point; points in all
    int32_t T, Kenc, ldw, bf16;
    const float* w0;                // natural W0 [F][ldw]
    const float* dscratch; int64_t d_total, drow;      // as NcaZgradArgs (f32: rows of 32 floats; bf16: bytes)
    float* g_lat;                   // [N][T]
};
hipError_t nca_launch_latgrad_f32(int F, const NcaLatgradArgs& a, hipStream_t st);
hipError_t nca_launch_pack_f32(const NcaLayout& y, const float* prm, void* out, hipStream_t st);
// both nets of a composite render in one launch (blockIdx.y = net)
hipError_t nca_launch_pack2_f32(const NcaLayout& ya, const float* prm_a, void* out_a, const NcaLayout& yb, const float* prm_b, void* out_b, hipStream_t st);
hipError_t nca_launch_pack2_bf16(const NcaLayout& ya, const float* prm_a, void* out_a, const NcaLayout& yb, const float* prm_b, void* out_b, hipStream_t st);
hipError_t nca_launch_fused_f32(int F, const NcaFusedArgs& a, int kmode, int grid, hipStream_t st);
hipError_t nca_launch_wgrad_f32(const NcaWgradArgs& a, int nsplit, hipStream_t st);
hipError_t nca_launch_reduce_f32(const NcaReduceArgs& a, hipStream_t st);
hipError_t nca_launch_pack_bf16(const NcaLayout& y, const float* prm, void* out, hipStream_t st);
hipError_t nca_launch_fused_bf16(int F, const NcaFusedArgs& a, int kmode, int grid, hipStream_t st, bool s8 = false);
// LDS bytes a fused bf16 launch of this mode needs NEXT TO its weight images (constants, output-layer partials)
size_t nca_fused_bf16_lds_other(int F, int kmode);
// mode 5: adds the per-tile sums of d loss / d raw in the tile records (fixed order) into the workgroups' output-bias slots of oslab
// (nets net0 .. net1 - 1 of the tile records; n_wg = the workgroups of THEIR dgrad launch)
hipError_t nca_launch_sum_tile_records(const char* dregion, int64_t wave_tile_bytes, int64_t dscale_off, int64_t ntiles, int net0, int net1, int F, float* oslab, int n_wg,
                                       hipStream_t st);
// waves_per_wg: 1 = one-wave workgroups (four share a CU's LDS); 4 = one 256-thread workgroup per CU (e5m2 staging only) -- the same
// waves doing the same work, but a CU then holds EITHER this kernel or a fused kernel's workgroup, whichever order two concurrent
// launches are dispatched in
hipError_t nca_launch_wgrad_bf16(int F, const NcaWgradArgs& a, int nsplit, hipStream_t st, int waves_per_wg = 1);
// the timing-experiment mask the bf16 kernels were compiled with (0 in every shipped library; round 3's timing-only builds: tools/r03_experiments.sh elim)
int nca_kernels_exp_mask();
int nca_kernels_variant_mask();
// rounding-ablation mask of the f32 kernels (NCA_ABL; 0 in every shipped library)
int nca_kernels_ablation_mask();
hipError_t nca_launch_pix_f32(int64_t R, int nchunk, const float* I0, const double* part, double* pix, hipStream_t st);
