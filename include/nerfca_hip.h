/*
 * nerfca_hip.h -- C ABI of libnerfca_hip.so: the MI355X (gfx950) implementation of NeRF-CA's
 * ray-sampling -> (static + dynamic) MLP -> log-space X-ray compositing path.
 *
 * The reference (kirstenmaas/NeRF-CA @ 2024-10-22) is pure Python/PyTorch and has no FFI of its
 * own; every entry point below therefore names the reference *Python* interface it replaces
 * (paths relative to the reference root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - buffers are caller-owned, contiguous, 16-byte aligned; the library allocates nothing that
 *     outlives a call except a small pool of timing events (nca_timing_*) and, per calling thread, one side stream with two events
 *     (NCA_OPT_OVERLAP_CUS: created at the first overlapped backward, never otherwise);
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*) and the call returns
 *     without synchronising; no exceptions cross the ABI;
 *   - return value 0 = ok, negative = error (NCA_E_*); nca_last_error() gives the message of the
 *     calling thread's last failure.  Unsupported configurations are errors, never fallbacks.
 *
 * Index bookkeeping (bit-exact with the reference): sample n = r*S + s (ray-major,
 * train/model_helpers.py:118-122), sigma outputs are [R,S] row-major.
 */
#ifndef NERFCA_HIP_H
#define NERFCA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NCA_ABI_VERSION 12

enum {
    NCA_OK = 0,
    NCA_E_INVALID = -1,     /* bad argument / inconsistent sizes */
    NCA_E_UNSUPPORTED = -2, /* configuration the kernels do not implement */
    NCA_E_HIP = -3,         /* a HIP runtime call failed */
    NCA_E_WORKSPACE = -4    /* workspace too small */
};

/* positional-encoding families of CPPN.pos_enc / Temporal.pos_enc (model/CPPN.py:112-135) */
enum {
    NCA_ENC_NONE = 0,    /* pos_enc == 'none': features = x                                  */
    NCA_ENC_BANDS = 1,   /* any windowed / plain mode: [x, sin(2^k x), sin(2^k x + pi/2)] * w */
    NCA_ENC_FOURIER = 2  /* 'fourier': [sin(2 pi x g), cos(2 pi x g)]                         */
};

/* output activations of get_activation_func (train/model_helpers.py:63-70) */
enum { NCA_ACT_SIGMOID = 0, NCA_ACT_SOFTPLUS = 1, NCA_ACT_CLAMP = 2 };

/* arithmetic of the MLP contractions */
enum {
    NCA_PREC_F32 = 0, /* parity mode, 1e-5 rel vs the reference: f32 operands and f32 accumulation; the hidden-layer contractions run on
                         the bf16 matrix cores from EXACT three-way bf16 splits of both operands (six v_mfma_f32_32x32x16_bf16 per f32
                         product block: the dropped products are below an f32 FMA chain's own rounding), layer 0 and the output layer on
                         v_mfma_f32_32x32x2_f32 */
    NCA_PREC_BF16 = 1 /* bf16 MFMA operands, f32 accumulate: throughput mode                  */
};

/* One coordinate MLP: model/CPPN.py:6-69 (T == 0) or model/Temporal.py:6-93 (T > 0).
 * Parameters live in ONE flat f32 buffer in nn.Module.parameters() order:
 *   [time_latents P*T]  W0[F,K0] b0[F]  {W_i[F,F] b_i[F]} x n_hidden
 *   [Wskip[F,F+K0] bskip[F]  {W[F,F] b[F]} x (n_late-1)]   Wo[1,F] bo[1]
 * with K0 = enc features + T. */
/* Two families of kernels run a net.  The FUSED kernels keep a sample's activations in registers through the whole net: F = 32, 64 or 128, three
 * input channels, one output channel, both precisions -- every net the reference ships.  The GENERAL kernels (ABI 12) run what those cannot hold --
 * F any multiple of 16 up to 1024 (model/CPPN.py:40-65 takes any num_filters), 1..8 input and output channels -- layer by layer with the
 * activations in HBM: f32 only (v_mfma_f32_32x32x2_f32), a forward store only when EVERY net of the batch is theirs (NCA_STORE_GENERAL), no depth gradients, and a workspace for every call
 * (nca_render_fwd_workspace_nets, nca_mlp_fwd_workspace).  A ray batch may mix the two: each net leaves its raw field, one compositing kernel follows.
 * Their packed image (nca_pack_weights) is [fan-in-padded weights | biases | output layer]. */
typedef struct NcaNet {
    int32_t F;        /* num_filters: 32, 64 or 128 (fused kernels); a multiple of 16 in [16, 1024] (general kernels: F > 128, or see `reserved`) */
    int32_t n_hidden; /* num_early_layers (F->F layers after the input layer) */
    int32_t n_late;   /* num_late_layers (CPPN only; skip connection); both precisions since ABI 11 */
    int32_t enc_mode; /* NCA_ENC_*                                            */
    int32_t L;        /* pos_enc_basis                                        */
    int32_t T;        /* num_time_dim; 0 for the static net                   */
    int32_t P;        /* rows of time_latents (10 in the reference)           */
    int32_t reserved; /* 0 = three input channels, one output channel.  ABI 12: bits 0..7 num_input_channels (0 = 3), bits 8..15
                         num_output_channels (0 = 1), bit 16 (NCA_NET_GENERAL) = run on the general kernels whatever the width;
                         anything but 0 selects the general kernels.  Parameter order with C_out outputs: ... Wo[C_out,F] bo[C_out];
                         encoded width K0 = C (none), C (1 + 2 L) (bands), 2 C L (fourier; coefficients f32[C L]) + T */
} NcaNet;
#define NCA_NET_CHANNELS(c_in, c_out) (((c_in) == 3 ? 0 : (c_in)) | (((c_out) == 1 ? 0 : (c_out)) << 8))
#define NCA_NET_GENERAL 0x10000

/* Per-call planner options (NcaRays.plan_opts): a field that is not NCA_OPT_UNSET replaces the process-wide tunable of the same name
 * (nca_set_option) for THIS call only -- two trainers, or two threads, of one process then do not see each other's settings.  A
 * backward should be given the options of its forward (the store's format travels in NcaRays.store_format either way). */
#define NCA_OPT_UNSET INT64_MIN
typedef struct NcaPlanOpts {
    int64_t stage_fp8;                /* NCA_OPT_STAGE_FP8                */
    int64_t stage_fp8_min_tiles;      /* NCA_OPT_STAGE_FP8_MIN_TILES      */
    int64_t resident_min_tiles;       /* NCA_OPT_RESIDENT_MIN_TILES       */
    int64_t wgrad_rebuild_weight_pct; /* NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT */
    int64_t overlap_cus;              /* NCA_OPT_OVERLAP_CUS (ABI 10)     */
    int64_t bf16_store;               /* NCA_OPT_BF16_STORE (ABI 10)      */
} NcaPlanOpts;
#define NCA_PLAN_OPTS_INIT {NCA_OPT_UNSET, NCA_OPT_UNSET, NCA_OPT_UNSET, NCA_OPT_UNSET, NCA_OPT_UNSET, NCA_OPT_UNSET}   /* "every field unset": a zeroed struct is NOT that (0 is a value of every option) */
struct NcaPlan;

/* A batch of rays and the per-step sampling state: the arguments of
 * obtain_train_predictions_iter / _static (train/model_helpers.py:99-160). */
typedef struct NcaRays {
    int64_t R;              /* rays                                                          */
    int32_t S;              /* samples per ray                                               */
    int32_t ray_is_f64;     /* origins/dirs are double (the real script) or float            */
    const void* origins;    /* [R,3]                                                         */
    const void* dirs;       /* [R,3] (not normalised, train/proj_helpers.py:83)              */
    const int32_t* phase;   /* heart phase ids in [0,P); element (r,s) at r*stride_r+s*stride_s */
    int64_t phase_stride_r;
    int64_t phase_stride_s;
    const float* z;         /* jittered depths: [S] (z_stride_r = 0) or [R,S] (fine pass)    */
    int64_t z_stride_r;
    const double* dists;    /* [S] interval lengths incl. the 1e-10 tail (model_helpers.py:73) */
    const float* I0;        /* [R] initial log-intensities                                   */
    int32_t act;            /* NCA_ACT_*                                                     */
    int32_t single_field;   /* 0: composite (sigma scaled);  1: render_volume_density (one net, sigma un-scaled) */
    float scale;            /* scale_value, 1e-2                                             */
    int32_t store_format;   /* backward from a forward store: the value nca_render_fwd returned when it wrote that store
                               (NCA_STORE_*); ignored by the forward and by a backward without a store */
    const NcaPlanOpts* plan_opts; /* HOST pointer or NULL: per-call planner options (above)                                   */
    struct NcaPlan* plan_out;     /* HOST pointer or NULL: filled with what the planner decided in THIS call (the forward fills
                                     the fwd_* fields and wave_tiles, a backward the rest; the other fields are left as they are) --
                                     the per-caller counterpart of the process-wide nca_last_plan() */
} NcaRays;

/* What a storing forward left in its store (the return value of nca_render_fwd; 0 = it wrote no store).  The backward is told
 * through NcaRays.store_format and lays the store out from THAT, not from the process-wide options, which may have changed in
 * between; a value that does not fit the nets of the call is NCA_E_INVALID. */
enum {
    NCA_STORE_NONE = 0,
    NCA_STORE_F32 = 1,        /* f32 mode: every layer input, ReLU masks, raw outputs                         */
    NCA_STORE_RESERVED2 = 2,  /* (ABI <= 8: bf16 mode with bf16 staging -- retired; never returned, refused by the backward)          */
    NCA_STORE_FP8 = 3,        /* bf16 mode: layer inputs as e4m3, masks of all layers, raw outputs               */
    NCA_STORE_BF16 = 4,       /* bf16 mode with NCA_OPT_STAGE_FP8 = 0 (ABI 10): layer inputs as bf16 fragments, masks of all layers, raw
                                 outputs -- nothing in 8 bits, nothing recomputed: the backward writes bf16 output gradients and the
                                 weight gradient contracts bf16 x bf16                                                */
    NCA_STORE_GENERAL = 5,    /* every net of the batch on the general kernels (ABI 12): X0 and every layer's output of every sample, f32, by runs of whole rays, and the
                                 raw fields -- what autograd keeps in the reference; the backward recomputes nothing                                   */
    NCA_STORE_KIND_MASK = 15,
    NCA_STORE_SHARED_ENC = 16 /* flag: both nets share one stored input block (same encoding vectors)          */
};

int nca_abi_version(void);
const char* nca_last_error(void);

/* ---- parameter bookkeeping ------------------------------------------------------------ */

/* Number of f32 parameters of `net` (== sum(p.numel() for p in module.parameters())). */
int64_t nca_param_count(const NcaNet* net);
/* Bytes of the MFMA-ordered weight image produced by nca_pack_weights. */
int64_t nca_packed_bytes(const NcaNet* net, int32_t prec);
/* Re-order the flat natural parameters into the LDS images the kernels stream
 * (run before every forward: optimisers update the parameters in place). */
int nca_pack_weights(const NcaNet* net, const float* params, void* packed, int32_t prec, void* stream);
/* The same for the two nets of a composite render in ONE launch (ABI 11): at the reference's default batch (1 024 rays x 500 samples,
 * train/composite.txt:25,40) a step is ~0.7 ms and every launch of a few microseconds counts. */
int nca_pack_weights2(const NcaNet* net_a, const float* params_a, void* packed_a,
                      const NcaNet* net_b, const float* params_b, void* packed_b, int32_t prec, void* stream);

/* ---- ray path: replaces obtain_train_predictions_iter/_static + get_predictions_* +
 *      render_volume_density[_composite] (train/model_helpers.py:28-160) ------------------ */

/* Forward.  net_d/packed_d/win_d may be NULL when rays->single_field == 1.
 *   pix   f64[R]    = I0 - sum_s (sigma_s + sigma_d) * dists.  NULL (ABI 11; composite or single-field renders of nets of one width): the
 *         forward leaves only the per-tile ray sums in `work` -- f64[R][nchunk], nchunk = ceil(S / 64) in bf16, ceil(S / 32) in f32 -- and
 *         the caller hands them to nca_loss_fwd_bwd (NcaLoss.ray_part), which forms pix itself: one launch less per step.
 *   sig_s f32[R,S], sig_d f32[R,S]  (activation * scale; un-scaled in single_field mode)
 *   work  scratch of nca_render_fwd_workspace() bytes.
 *   store NULL, or a caller-owned buffer of nca_render_store_bytes() bytes: the forward then also leaves there what the
 *         reference's autograd graph keeps (train/run_composite.py:306) -- f32: every layer input, the ReLU masks and the
 *         raw outputs; bf16: the layer inputs as e4m3, the ReLU masks of every layer and the raw outputs (NCA_OPT_STAGE_FP8 = 0:
 *         no store at all) -- and nca_render_bwd given the same buffer does not recompute the layers.
 *         nca_render_store_bytes() returns 0 where this is not available (nets of different width, nets without a hidden
 *         layer, a general-kernel net beside a fused-kernel one): pass NULL there.  Every net on the general kernels (ABI 12): the store holds
 *         X0, every layer's output and ReLU bit masks by runs of whole rays, and the raw fields, f32 (NCA_STORE_GENERAL).
 * Returns a negative error code, or >= 0: the format of the store it wrote (NCA_STORE_*, 0 = none) -- hand it to the backward
 * in NcaRays.store_format. */
int64_t nca_render_fwd_workspace(const NcaRays* rays);
/* The same when a net of the batch runs on the general kernels (ABI 12): a chunk of activations on top ([rows][K0p + 2 F] floats, at most 2^18 rows,
 * fewer under `max_bytes` > 0).  Equal to nca_render_fwd_workspace() for nets of the fused kernels. */
int64_t nca_render_fwd_workspace_nets(const NcaRays* rays, const NcaNet* net_s, const NcaNet* net_d, int32_t prec, int64_t max_bytes);
int64_t nca_render_store_bytes(const NcaRays* rays, const NcaNet* net_s, const NcaNet* net_d, int32_t prec);
int nca_render_fwd(const NcaRays* rays, int32_t prec,
                   const NcaNet* net_s, const void* packed_s, const float* win_s, const float* four_s,
                   const NcaNet* net_d, const void* packed_d, const float* win_d, const float* four_d,
                   const float* latents_d, /* = params_d (time_latents are its first P*T floats) */
                   double* pix, float* sig_s, float* sig_d, void* work, int64_t work_bytes,
                   void* store, int64_t store_bytes, void* stream);

/* Backward: with recompute (store == NULL), or from the store the forward of the SAME rays, weights and windows left
 * (rays->store_format = that forward's return value).
 * Upstream gradients g_pix f64[R], g_sig_s/g_sig_d f32[R,S] (NULL = 0).
 * Writes (overwrites) grads_s / grads_d, flat f32 in the natural parameter order.
 * `params_*` are the natural flat parameters (needed for the latent gradient). */
int64_t nca_render_bwd_workspace(const NcaRays* rays, const NcaNet* net_s, const NcaNet* net_d, int32_t prec,
                                 int64_t max_bytes);
int nca_render_bwd(const NcaRays* rays, int32_t prec,
                   const NcaNet* net_s, const void* packed_s, const float* win_s, const float* four_s, const float* params_s,
                   const NcaNet* net_d, const void* packed_d, const float* win_d, const float* four_d, const float* params_d,
                   const double* g_pix, const float* g_sig_s, const float* g_sig_d,
                   float* grads_s, float* grads_d, void* work, int64_t work_bytes,
                   const void* store, int64_t store_bytes, void* stream);
/* The same, and d loss / d depth of every sample into g_depth f32[R,S] (NULL: as nca_render_bwd).  The reference's fine pass
 * does not detach its sampled depths: the fine losses reach them through the query point, the positional encoding and the
 * first layer -- and the skip layer, where a net has one -- (train/model_helpers.py:146-148), and through them the coarse
 * nets.  Nets of one width. */
int nca_render_bwd_depth(const NcaRays* rays, int32_t prec,
                   const NcaNet* net_s, const void* packed_s, const float* win_s, const float* four_s, const float* params_s,
                   const NcaNet* net_d, const void* packed_d, const float* win_d, const float* four_d, const float* params_d,
                   const double* g_pix, const float* g_sig_s, const float* g_sig_d,
                   float* grads_s, float* grads_d, float* g_depth, void* work, int64_t work_bytes,
                   const void* store, int64_t store_bytes, void* stream);

/* ---- point path: replaces CPPN.forward / Temporal.forward_composite on arbitrary points
 *      (model/CPPN.py:88-110, model/Temporal.py:138-151) ---------------------------------- */
int nca_mlp_fwd(const NcaNet* net, int32_t prec, const void* packed, const float* win, const float* four,
                const float* params, int64_t N, const float* pts /*[N,3]*/, const int32_t* phase /*[N] or NULL*/,
                float* raw /*[N]*/, void* stream);
/* ABI 12: the forward with a workspace -- what a net on the general kernels needs (nca_mlp_fwd refuses it with NCA_E_WORKSPACE); any other net runs as
 * nca_mlp_fwd and nca_mlp_fwd_workspace returns 0.  pts f32[N, C_in], raw f32[N, C_out]; g_raw / grads of nca_mlp_bwd likewise. */
int64_t nca_mlp_fwd_workspace(const NcaNet* net, int32_t prec, int64_t N, int64_t max_bytes);
int nca_mlp_fwd_ws(const NcaNet* net, int32_t prec, const void* packed, const float* win, const float* four,
                   const float* params, int64_t N, const float* pts, const int32_t* phase, float* raw,
                   void* work, int64_t work_bytes, void* stream);
int64_t nca_mlp_bwd_workspace(const NcaNet* net, int32_t prec, int64_t N, int64_t max_bytes);
/* g_latents: NULL, or f32[N, T] (ABI 9): d loss / d latent INPUT of every point -- what autograd hands back through
 * Temporal.query_time's latent vectors (model/Temporal.py:113-136), one row per point whichever points share a table row; `grads`
 * still carries the table-row sums in its first P*T floats. */
int nca_mlp_bwd(const NcaNet* net, int32_t prec, const void* packed, const float* win, const float* four,
                const float* params, int64_t N, const float* pts, const int32_t* phase, const float* g_raw /*[N]*/,
                float* grads, float* g_latents, void* work, int64_t work_bytes, void* stream);

/* ---- stand-alone compositing of raw fields: render_volume_density_composite / render_volume_density
 *      (train/model_helpers.py:72-97), as the reference's evaluation code calls them
 *      (train/run_composite.py:361, 407-413).  raw_* f32[R,S]; `single_field`, `act`, `scale` as in NcaRays. */
int nca_composite_fwd(int64_t R, int32_t S, int32_t act, int32_t single_field, float scale,
                      const float* raw_s, const float* raw_d, const float* I0, const double* dists,
                      double* pix, float* sig_s, float* sig_d, void* stream);
int nca_composite_bwd(int64_t R, int32_t S, int32_t act, int32_t single_field, float scale,
                      const float* raw_s, const float* raw_d, const double* dists,
                      const double* g_pix, const float* g_sig_s, const float* g_sig_d,
                      float* g_raw_s, float* g_raw_d, void* stream);

/* ---- losses: replaces weighted_MSELoss + compute_losses + the loss assembly and their autograd
 *      (train/model_helpers.py:189-262, 284-288; train/run_composite.py:276-292) ------------------ */
typedef struct NcaLoss {
    int64_t R;               /* rays of this batch (this rank's slice under data parallelism)          */
    int32_t S;
    int32_t use_weighting;   /* entro_use_weighting                                                     */
    double skew;             /* skewness_val                                                            */
    double mask_thre;        /* entro_mask_thre                                                         */
    double weighted_thresh;  /* entro_weighted_thresh                                                   */
    double w_favor, w_dent, w_occl, w_l1;  /* this step's weights (linear_param_decay, run_composite.py:276-279) */
    double inv_R;            /* 1 / GLOBAL ray count: mean-type terms are sums over local rays times this */
    const double* weights_dev; /* NULL, or DEVICE f64[4] = {w_favor, w_dent, w_occl, w_l1} read by the kernels instead of
                                  the four fields above: a captured HIP graph can be replayed with new weights          */
    int32_t unit_mse;        /* 1: the pixel term uses unit weights while the regularisers keep `wpix` -- the fine pass's
                                weighted_pixs_ones (train/run_composite.py:296-299)                                      */
    int32_t reserved;
    double* g_dists;         /* NULL, or DEVICE f64[S]: d loss / d dists -- through the ray sums pix = I0 - sum sigma dists (given the
                                pixel term of THIS call) and through every regulariser that contains sigma * dists.  The reference's
                                fine pass differentiates through the interval lengths of ray 0 (train/model_helpers.py:150); needs the
                                three gradient outputs and `dists_work`                                                    */
    double* dists_work;      /* DEVICE f64[R * S] scratch for g_dists (per-ray contributions, summed over rays in a fixed order) */
    const double* term_grads;/* NULL, or DEVICE f64[11] (ABI 10): TERM-GRADIENT MODE -- the backward of compute_losses as an autograd function
                                (train/model_helpers.py:250-262).  Element i is the upstream gradient of the reference's i-th return value
                                [blendw mean, sigma_s max, sigma_d max, favor, s_entropy, s_sum, d_entropy, d_sum, occl, l1, l2] (the maxima are
                                computed under no_grad there: their entries are ignored); g_sig_s / g_sig_d (and g_dists) then are
                                sum_i term_grads[i] d term_i / d sigma -- every term, not only the six the training loss weights.  There is
                                no pixel term in this mode: pix, gt and g_pix may be NULL (weighted_MSELoss is nca_weighted_sq_err);
                                the four weights above and weights_dev are not read                                                          */
    /* ABI 11 -- a zeroed tail keeps ABI 10 behaviour */
    const double* ray_part;  /* NULL, or DEVICE f64[R][ray_nchunk]: the per-tile ray sums a forward called with pix = NULL left in its workspace;
                                the loss kernel then forms pix[r] = I0[r] - sum_c ray_part[r][c] itself (same order as the forward's own kernel:
                                bit-identical) and, if pix_out is given, stores it.  The `pix` argument of nca_loss_fwd_bwd is then ignored  */
    const float* ray_I0;     /* DEVICE f32[R] (with ray_part)                                                                               */
    double* pix_out;         /* NULL or DEVICE f64[R] (with ray_part)                                                                       */
    int32_t ray_nchunk;
    int32_t reserved2;
    float* terms_f32;        /* NULL, or DEVICE f32[NCA_T_COUNT]: the terms once more, rounded to f32 -- what rides behind the flat gradient in a
                                sharded step's ONE all-reduce (train/run_composite.py:310: the early-stop pair is terms 8 and 5)             */
} NcaLoss;
enum { NCA_T_LOSS = 0, NCA_T_PIXEL, NCA_T_BLENDW, NCA_T_SIG_S_MAX, NCA_T_SIG_D_MAX, NCA_T_FAVOR, NCA_T_S_ENTROPY, NCA_T_S_SUM,
       NCA_T_D_ENTROPY, NCA_T_D_SUM, NCA_T_OCCL, NCA_T_L1, NCA_T_L2, NCA_T_COUNT };
int64_t nca_loss_workspace(int64_t R);
/* terms f64[NCA_T_COUNT]; gradients g_pix f64[R], g_sig_s/g_sig_d f32[R,S] (all three NULL = values only; term-gradient mode: g_pix NULL).
 * pix, gt, wpix are f64[R]; sig_s, sig_d f32[R,S]; dists f64[S]. */
int nca_loss_fwd_bwd(const NcaLoss* desc, const double* pix, const double* gt, const double* wpix,
                     const float* sig_s, const float* sig_d, const double* dists,
                     double* terms, double* g_pix, float* g_sig_s, float* g_sig_d,
                     void* work, int64_t work_bytes, void* stream);

/* weighted_MSELoss.forward (train/model_helpers.py:284-288): out[r] = (pred[r] - gt[r])^2 * w[r] (the caller takes the mean,
 * train/run_composite.py:287) and its backward; all arrays of one dtype, f64 (is_f64 = 1: what the real script's f64 ray table gives) or
 * f32; g_pred / g_gt / g_w may each be NULL. */
int nca_weighted_sq_err(int64_t R, int32_t is_f64, const void* pred, const void* gt, const void* w, void* out, void* stream);
int nca_weighted_sq_err_bwd(int64_t R, int32_t is_f64, const void* pred, const void* gt, const void* w, const void* g_out,
                            void* g_pred, void* g_gt, void* g_w, void* stream);

/* ---- fine-pass depths: the sampling half of the hierarchical pass of obtain_train_predictions_iter
 *      (train/model_helpers.py:131-148) with sample_pdf (:162-187): weights = |jump of sigma_s + sigma_d| / batch-wide
 *      max, inverse-transform sampling of n_fine depths per ray from the injected uniform draws u[R, n_fine], and
 *      sort(cat[fine, coarse]).  sig_d may be NULL (single field).  z f32[S] is the coarse depth vector shared by
 *      all rays (ascending); z_all f32[R, S + n_fine].  Needs S >= 3. ------------------------------------ */
int64_t nca_fine_depths_workspace(int64_t R);
int nca_fine_depths(int64_t R, int32_t S, int32_t n_fine, const float* sig_s, const float* sig_d, const float* z,
                    const float* u, float* z_all, void* work, int64_t work_bytes, void* stream);
/* Backward of the three calls above (the reference does not detach the sampled depths, train/model_helpers.py:135-146).
 * nca_fine_depths_bwd: g_tot f32[R,S] = d loss / d (sigma_s + sigma_d) through sample_pdf and the sort with the maximum held
 * fixed; gmax_part f32[R] / cnt_part f32[R] = per-ray parts of d loss / d wmax and of the number of jumps that attain wmax.
 * The caller sums both over rays (and ranks) and passes gmax_each = sum(gmax_part) / count to nca_fine_depths_bwd_max, which
 * adds it to the jumps that attain the maximum (torch.max distributes its gradient evenly over ties). */
int nca_fine_depths_bwd(int64_t R, int32_t S, int32_t n_fine, const float* sig_s, const float* sig_d, const float* z,
                        const float* u, const float* wmax, const float* g_z_all, float* g_tot, float* gmax_part, float* cnt_part,
                        void* stream);
int nca_fine_depths_bwd_max(int64_t R, int32_t S, const float* sig_s, const float* sig_d, const float* wmax, const float* gmax_each,
                            float* g_tot, void* stream);
/*      The same in two stages for a batch that is sharded over ranks: the weights are normalised by the maximum over
 *      the WHOLE batch (model_helpers.py:139), so a rank computes the maximum of its rays into the device scalar
 *      wmax f32[1], all-reduces it (MAX) and samples with the result. */
int nca_fine_weight_max(int64_t R, int32_t S, const float* sig_s, const float* sig_d, float* wmax, void* work, int64_t work_bytes,
                        void* stream);
int nca_fine_depths_given_max(int64_t R, int32_t S, int32_t n_fine, const float* sig_s, const float* sig_d, const float* z,
                              const float* u, const float* wmax, float* z_all, void* stream);

/* ---- per-step batch preparation: the ray gather of train/run_composite.py:262-273 (rays_train[ids] split into origins, directions,
 *      pixel values and loss weights; the rays' heart phases) and randomize_depth + the interval lengths of
 *      train/model_helpers.py:3-12, 73-74, in one launch instead of ~20 small torch kernels.  Bit-exact with the torch ops:
 *      z' = lo + (hi - lo) * t with mid = 0.5 * (z[1:] + z[:-1]) in f32; dists[i] = z'[i+1] - z'[i] (f32 subtraction, widened) and the
 *      1e-10 tail in the ray table's dtype.
 *      table f64[N,4,3] (rows: origin, direction, pixel x3, weight x3); phases i64[N]; ids i64[R].
 *      n_rows = N: an id outside [0, N) does not reach memory -- it is clamped into the table and, if `bad_ids` (device i32[1],
 *      caller-zeroed, may be NULL) is given, counted there; the caller reads the counter when it can afford a synchronisation
 *      (CompositeTrainer.early_stop() / evaluate() do, and raise; bench.py checks at the end of every timed record).  With bad_ids = NULL the
 *      clamping is SILENT: the step trains on row 0 or row N - 1 instead of failing as NumPy's IndexError does in the reference -- pass the
 *      counter.  n_rows <= 0 = unknown: ids are trusted as in ABI <= 8.
 *      Outputs: o, d f64[R,3]; gt, w f64[R]; ph i32[R]; z f32[S]; dists f64[S].  depth / t_rand f32[S].  ------------------------------ */
int nca_prepare_batch(int64_t R, int32_t S, const int64_t* ids, const double* table, const int64_t* phases, int64_t n_rows, int32_t* bad_ids,
                      const float* depth, const float* t_rand,
                      double* o, double* d, double* gt, double* w, int32_t* ph, float* z, double* dists, void* stream);

/* ---- per-step batch sampling and schedules ON THE DEVICE (ABI 11): the importance sampling of train/run_composite.py:250-260
 *      (img_sample_size - n_var ids drawn with replacement from the non-variance rays, n_var from the variance rays, shuffled; or
 *      uniform over the table when var_sample_perc == 0), the stratified depth jitter's uniform draw (train/model_helpers.py:8), the
 *      FreeNeRF band windows (model/CPPN.py:144-159) and the four loss weights (linear_param_decay, train/model_helpers.py:264-269;
 *      train/run_composite.py:276-279) as functions of (seed, iteration): counter-based Philox4x32-10 streams (csrc/nca_rng.hpp), so a rank
 *      draws only ITS slots of the global batch and a captured HIP graph replays a whole training run with NO host work per step -- the
 *      iteration is n_iter + *iter_dev (iter_dev: DEVICE i64[1] or NULL).  The reference's own host draws (NumPy's global generator) are not
 *      reproducible by anyone; what is reproduced is their distribution -- exactly n_var variance slots in a uniformly random arrangement,
 *      i.i.d. uniform draws per slot -- and callers that hold the reference's draws inject them (ids_in / t_rand_in).  ------------------ */
typedef struct NcaSampler {
    uint64_t seed;
    int64_t n_iter;             /* iteration = n_iter + (iter_dev ? *iter_dev : 0)                                        */
    const int64_t* iter_dev;
    int64_t R_global;           /* img_sample_size: slots of the GLOBAL batch (the arrangement is drawn over all of them) */
    int64_t n_var;              /* slots that draw from var_ids (int(var_sample_perc / 100 * img_sample_size)); 0: every slot draws from [0, n_rows) */
    const int64_t* var_ids;     /* DEVICE i64[n_var_ids]     (train/data_helpers.py: var_ray_ids)                          */
    int64_t n_var_ids;
    const int64_t* non_var_ids; /* DEVICE i64[n_non_var_ids]                                                              */
    int64_t n_non_var_ids;
    int64_t n_rows;             /* rows of the ray table                                                                  */
} NcaSampler;
/* ids i64[R] = the ray ids of slots slot0 .. slot0 + R - 1 of the global batch */
int nca_draw_ray_ids(const NcaSampler* s, int64_t slot0, int64_t R, int64_t* ids, void* stream);
/* out f32[n]: uniform [0, 1) draws of stream `stream_id` (2 = the depth jitter; >= 16 free for callers) of the sampler's iteration */
int nca_draw_uniform(const NcaSampler* s, int32_t stream_id, int64_t n, float* out, void* stream);

enum { NCA_WINDOW_NONE = 0, NCA_WINDOW_FREE = 1 };
typedef struct NcaWindowSched {
    int32_t kind;               /* NCA_WINDOW_FREE: update_freq_mask_alpha (model/CPPN.py:144-159) -- bands below the moving pointer 1, the band under
                                   it the fractional part, the rest 1e-8, everything 1 from decay_steps on; bit-identical to the host schedule */
    int32_t L;                  /* pos_enc_basis */
    int32_t window_start;       /* pos_enc_window_start */
    int32_t reserved;
    int64_t decay_steps;        /* *_pos_enc_window_decay_steps */
    float* out;                 /* DEVICE f32[L] */
} NcaWindowSched;
typedef struct NcaWeightSched { double start, end; int64_t steps, delay; } NcaWeightSched;   /* linear_param_decay(iter, start, end, steps, delay) */
typedef struct NcaSchedules {
    int32_t n_windows;          /* 0 .. 4 */
    int32_t reserved;
    NcaWindowSched window[4];
    NcaWeightSched weight[4];   /* favor_s, dynamic entropy, occlusion, l1 (train/run_composite.py:276-279) */
    double* weights_out;        /* NULL or DEVICE f64[4]: NcaLoss.weights_dev of the step */
} NcaSchedules;
/* nca_prepare_batch with everything that changes from step to step made on the device, ONE launch: ids (drawn, or ids_in), the gather, the
 * jitter draw (or t_rand_in) with the jittered depths and interval lengths, the band windows and loss weights of the iteration.
 *   ids_in / t_rand_in  NULL = draw; else the caller's i64[R] / f32[S] (parity tests inject the reference's own draws)
 *   ids_out / t_rand_out  NULL, or where the drawn values are also stored
 *   sched  NULL = no schedules; the other arguments as nca_prepare_batch */
int nca_begin_step(const NcaSampler* s, int64_t slot0, int64_t R, int32_t S, const NcaSchedules* sched,
                   const int64_t* ids_in, const float* t_rand_in,
                   const double* table, const int64_t* phases, int32_t* bad_ids, const float* depth,
                   double* o, double* d, double* gt, double* w, int32_t* ph, float* z, double* dists,
                   int64_t* ids_out, float* t_rand_out, void* stream);

/* ---- optimiser: torch.optim.Adam(lr) + LinearLR(start_factor=1, end_factor, total_iters) of
 *      train/run_composite.py:209-215, 307-308, as one launch over up to NCA_ADAM_MAX_SEG flat buffers.
 *      `step` is DEVICE i64[2] (ABI 11; both zero before the first call): step[0] = optimiser steps taken so far -- the kernel uses
 *      t = step[0] + 1 for the bias corrections and lr = cfg.lr * factor(step[0]) (the scheduler is stepped after the optimiser in the
 *      reference), and the LAST workgroup to finish increments it (step[1] counts the finished workgroups of the launch and is zero again
 *      when it ends: one launch instead of an update and a one-thread tick), so a captured graph replays the whole schedule.
 *      NcaAdam.iter_counter (DEVICE i64[1] or NULL) is incremented with it: the training iteration nca_begin_step reads. -------- */
enum { NCA_ADAM_MAX_SEG = 4 };
typedef struct NcaAdam {
    double lr;               /* base learning rate                                              */
    double beta1, beta2, eps;/* torch defaults 0.9, 0.999, 1e-8; no weight decay, no amsgrad    */
    double lr_end_factor;    /* LinearLR end_factor (1.0 = constant lr)                         */
    int64_t lr_total_iters;  /* LinearLR total_iters                                            */
    int64_t* iter_counter;   /* ABI 11: NULL, or DEVICE i64[1] incremented together with step[0]  */
} NcaAdam;
/* n, params, grads, exp_avg, exp_avg_sq: HOST arrays of n_seg entries (device pointers, f32). */
int nca_adam_step(const NcaAdam* cfg, int32_t n_seg, const int64_t* n, float* const* params, const float* const* grads,
                  float* const* exp_avg, float* const* exp_avg_sq, int64_t* step, void* stream);

/* ---- process-wide tunables: A/B switches of the planner, also the hook with which tests force a kernel path at sizes the
 *      oracle can afford.  Both calls return NCA_OK or NCA_E_INVALID; values persist until changed.  A planner decision that
 *      shapes a forward store is taken ONCE, by the forward, and travels to the backward in NcaRays.store_format. -------------- */
enum {
    NCA_OPT_RESERVED0 = 0,        /* (ABI <= 8: NCA_OPT_ONCHIP_MIN_TILES of the retired bf16-staged backward; get / set return NCA_E_INVALID) */
    NCA_OPT_STAGE_FP8 = 1,        /* bf16 mode: whether the forward may leave a STORE for its backward.  The store is 8-bit staged: the layer
                                     inputs and output gradients that only the weight-gradient kernel reads cross HBM as 8-bit floats (inputs
                                     e4m3, gradients e5m2 scaled by a power of two per 64-sample tile; f32 accumulation; the MLP contractions
                                     themselves stay bf16), the store also holds the raw outputs and the masks of every layer, and the backward
                                     recomputes nothing.  1 = always; 0 = never: nca_render_store_bytes returns 0, a forward that is handed a
                                     store leaves it untouched and returns NCA_STORE_NONE, and the backward recomputes the layers -- bf16
                                     operands everywhere, nothing in 8 bits (BASELINE configs[1] "as written"; the bf16-STAGED store of
                                     ABI <= 8 was retired in round 4: never better in held-out PSNR, 1.75 x slower);
                                     -1 = by batch size: a store when the batch has at least NCA_OPT_STAGE_FP8_MIN_TILES wave tiles.  The
                                     default is -1 with a threshold of 0, i.e. the store is UNCONDITIONAL unless the caller sets one of the
                                     two.  nca_last_plan().stage_fp8 says what ran.  Initial value from NCA_STAGE_FP8 (0 / 1) */
    NCA_OPT_RESIDENT_MIN_TILES = 2, /* bf16 mode: run the fused kernels with ONE net per launch and all of that net's weight images
                                     resident in LDS (no per-layer weight DMA, no workgroup barrier in the tile loop) when the images fit
                                     (width 128: the input layer + 4 hidden layers forward, 4 transposed images backward) and the batch has at least this many 64-sample wave
                                     tiles.  A two-net render then takes two forward launches (the second composites with the first
                                     one's sigma).  0 = always, -1 = never; default 8 * 8 waves * CUs (NCA_RESIDENT=0 -> never,
                                     =force -> always).  Results are bit-identical to the streaming kernels */
    NCA_OPT_STAGE_FP8_MIN_TILES = 3, /* threshold of NCA_OPT_STAGE_FP8 = -1, in 64-sample wave tiles of the whole batch (>= 0; default 0 = always) */
    NCA_OPT_WGRAD_REBUILD_WEIGHT_PCT = 4, /* fp8 staging: the weight-gradient launch is ONE round of one-wave jobs, so its slowest wave is
                                     the launch; the jobs that rebuild their output-gradient block from mask bits take more cycles per
                                     tile than the others and get this many percent of the others' sample splits (100 .. 200; default
                                     115 = the cycle ratio measured on MI355X with round 3's clock-probe build, tools/r03_experiments.sh; initial value from NCA_WGRAD_W).
                                     A constant rather than a calibration at first use: the splits fix the summation order, and with it
                                     the bits of the gradient -- tools/calibrate_wgrad.py times the candidates on a given box */
    NCA_OPT_OVERLAP_CUS = 5,      /* bf16 mode, backward from the 8-bit staged store of a two-net ray batch with resident weight images (the bench path):
                                     the static net's weight-gradient launch (HBM-bound) runs BESIDE the dynamic net's dgrad launch (issue-bound) on
                                     a second stream of the library -- fork and join are events on the caller's stream, so a stream capture records
                                     both branches.  The value is the number of compute units whose wave slots the overlapped weight-gradient launch
                                     is sized for (its one-round grid); the dgrad launch beside it gets the others.  A compute unit holds a workgroup
                                     of EITHER kernel (both take more than half of its LDS), so the two grids add up to the chip whichever is
                                     dispatched first.  The dynamic net's weight gradient follows the join on the whole chip.  0 = off: one
                                     weight-gradient launch for both nets after both dgrad launches (ABI <= 9).  The value fixes the split of the
                                     sample sums, hence the bits of the gradient (run to run they are identical either way).  Initial value from
                                     NCA_OVERLAP_CUS; nca_last_plan().overlap_cus says what ran */
    NCA_OPT_BF16_STORE = 6,       /* bf16 mode, batches for which the 8-bit staged store is off (NCA_OPT_STAGE_FP8 = 0, or below its threshold): 1 (default) =
                                     the forward may still leave a store, with the layer inputs as BF16 fragments (NCA_STORE_BF16; masks of every layer
                                     and raw outputs as in the 8-bit store) -- the backward then recomputes nothing (mode 5), writes bf16 output
                                     gradients and the weight gradient contracts bf16 x bf16: BASELINE configs[1] "as written", nothing in 8 bits;
                                     0 = no store, the backward recomputes the layers (mode 1; also what runs when the caller passes no store).
                                     Initial value from NCA_BF16_STORE */
    NCA_OPT_COUNT
};
int nca_get_option(int32_t opt, int64_t* value);
int nca_set_option(int32_t opt, int64_t value);

/* What the planner decided in the process's last nca_render_fwd and last nca_render_bwd[_depth] / nca_mlp_bwd (bench.py labels
 * its line with it; tests assert the path they mean to exercise).  Process-wide, not per thread: a PyTorch backward runs on the
 * autograd engine's thread. */
typedef struct NcaPlan {
    int32_t fwd_store_format;     /* NCA_STORE_* | flags of the last forward (0: no store)                              */
    int32_t fwd_launches;         /* fused launches of that forward (2: one per net, weight images resident in LDS)     */
    int32_t fwd_resident;         /* 1: resident weight images                                                          */
    int32_t bwd_kernel_mode;      /* 1 recompute, 3 from the f32 store, 5 from the bf16 mode's 8-bit staged store (nothing recomputed) */
    int32_t bwd_resident;
    int32_t bwd_launches_per_chunk; /* fused dgrad launches per ray chunk                                              */
    int32_t bwd_onchip;           /* always 0 (ABI <= 8: the retired bf16-staged backward's on-chip layer)              */
    int32_t stage_fp8;            /* that backward staged 8-bit blocks                                                  */
    int32_t wgrad_jobs, wgrad_splits, wgrad_splits_rebuild;
    int32_t chunks;               /* ray chunks of that backward                                                        */
    int64_t wave_tiles;           /* wave tiles of the whole batch of the last call                                     */
    int32_t overlap_cus;          /* NCA_OPT_OVERLAP_CUS as that backward ran it (0: one weight-gradient launch for both nets)   */
    int32_t overlap_forked;       /* 1: the overlapped launch went to the library's second stream; 0: the same launches in a row on the caller's
                                     (the second stream did not exist yet and the caller's stream was capturing)                  */
    int64_t reserved[3];
} NcaPlan;
int nca_last_plan(NcaPlan* out);
/* Static description of the build: target, ABI, and the timing-experiment mask the kernels were compiled with ("NCA_EXP=0" in
 * every shipped library; the timing-only builds of rounds 2 - 3, whose results are wrong by construction, come from the tag r03-kernels: tools/r03_experiments.sh). */
const char* nca_build_info(void);

/* ---- in-library kernel timing (HIP events on the launch stream), used by bench.py -------- */
enum { NCA_K_PACK = 0, NCA_K_FWD = 1, NCA_K_BWD_DGRAD = 2, NCA_K_BWD_WGRAD = 3, NCA_K_BWD_REDUCE = 4, NCA_K_LOSS = 5, NCA_K_ADAM = 6, NCA_K_COUNT = 7 };
int nca_timing_enable(int32_t on);
/* Synchronises the recorded events and returns accumulated milliseconds and launch count. */
int nca_timing_read(int32_t kind, double* total_ms, int64_t* launches);
int nca_timing_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* NERFCA_HIP_H */
