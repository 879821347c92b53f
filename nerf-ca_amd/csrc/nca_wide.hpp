// nca_wide.hpp -- the GENERAL kernels (internal): nets the fused kernels do not cover -- more than 128 units per layer, or other channel
// counts than 3 -> 1 (model/CPPN.py:40-65 takes any num_filters / num_input_channels / num_output_channels).
//
// The fused kernels keep a sample's activations in registers through the whole net; that stops at 128 units (512 VGPRs, 160 KB of LDS).
// Here a net runs LAYER BY LAYER with its activations in HBM, row-major [sample][unit]: one LDS-tiled f32 GEMM per layer on
// v_mfma_f32_32x32x2_f32 (f32 operands, f32 accumulation: the parity arithmetic of the output layer and layer 0 of the fused f32 kernels),
// forward, dgrad and wgrad being the same kernel with other operand strides.  Slower per FLOP than the fused path (every layer's input and
// output cross HBM once each way) and meant for what that path cannot hold.
//
//   X0 block  [sample][K0p]   columns: encoded input (natural order) | time latents | one-hot phase (backward only) | zero pad to 16
//   H_j       [sample][F]     output of layer j (after ReLU); F is a multiple of 16
//   M_j       ReLU bit masks of H_j, j < NL - 1 (backward only): 2 KiB per 128 x 128 tile, written by the forward GEMM, read by the dgrad GEMM
//   Wp_j      [F][Kp_j]       layer j's weight with its fan-in padded: layer 0 K0p; hidden F; skip K0p + F (encoded part first, CPPN.py:102)
//   packed    [Wp_0 .. Wp_{NL-1} | biases [NL][F] | Wo [Cout][F] | bo [Cout]]   (nca_pack_weights)
// Rows are padded to a multiple of 128 per chunk; padded rows hold zeros in X0 and in every output gradient.
#pragma once
#include <hip/hip_runtime.h>
#include "nca_layout.hpp"

#define NCA_WIDE_MAX_F 1024
#define NCA_WIDE_MAX_C 8          // input / output channels
#define NCA_WIDE_ROWS 128         // row granularity of a chunk (the GEMM tile)

struct NcaWideLayer {
    int32_t kind;                 // NCA_IN_*
    int32_t K, Kp;                // natural / padded fan-in
    int32_t w_off, b_off;         // natural flat offsets (floats)
    int64_t pw_off;               // floats from the packed base: Wp [F][Kp]
};
struct NcaWideLayout {
    int32_t F, NL, C, Cout;
    int32_t enc_mode, L, T, P;
    int32_t Kenc, K0, K0p;        // K0 = Kenc + T natural columns of the encoded input; K0p = round_up(K0 + P, 16)
    int32_t lat_off, wo_off, bo_off, n_params;
    int64_t pb_off, pwo_off, pbo_off;     // the packed image behind the layers' Wp: biases [NL][F], Wo [Cout][F], bo [Cout] (floats from its base) --
    int64_t packed_floats;                // a forward needs nothing but the image (and the latent table); a backward's split slabs have the SAME layout
    NcaWideLayer layer[NCA_MAX_LAYERS];
};

// channel counts ride in NcaNet.reserved: bits 0..7 num_input_channels (0 = 3), bits 8..15 num_output_channels (0 = 1)
NCA_HD inline int nca_net_cin(const NcaNet& n) { return (n.reserved & 0xff) ? (n.reserved & 0xff) : 3; }
NCA_HD inline int nca_net_cout(const NcaNet& n) { return ((n.reserved >> 8) & 0xff) ? ((n.reserved >> 8) & 0xff) : 1; }
// does this net run on the general kernels?
NCA_HD inline bool nca_net_is_wide(const NcaNet& n) { return n.F > 128 || nca_net_cin(n) != 3 || nca_net_cout(n) != 1 || (n.reserved & 0x10000) != 0; }

inline int nca_build_layout_wide(const NcaNet& n, NcaWideLayout* out, const char** why) {
    NcaWideLayout y{};
    if (n.F < 16 || n.F > NCA_WIDE_MAX_F || (n.F & 15)) { *why = "general kernels: num_filters must be a multiple of 16 in [16, 1024] (the host pads with zero-weight units)"; return NCA_E_UNSUPPORTED; }
    if (n.n_hidden < 0 || n.n_late < 0 || 1 + n.n_hidden + n.n_late > NCA_MAX_LAYERS) { *why = "too many layers"; return NCA_E_UNSUPPORTED; }
    if (n.T < 0 || n.T > 32 || (n.T > 0 && n.P <= 0) || n.P > 64) { *why = "num_time_dim must be in [0,32], phases in [1,64]"; return NCA_E_UNSUPPORTED; }
    if (n.T > 0 && n.n_late > 0) { *why = "Temporal with num_late_layers > 0 has no output in the reference (Temporal.py:128-135)"; return NCA_E_UNSUPPORTED; }
    if (n.enc_mode < 0 || n.enc_mode > 2 || n.L < 0 || n.L > 16) { *why = "bad positional encoding"; return NCA_E_UNSUPPORTED; }
    if (n.enc_mode != NCA_ENC_NONE && n.L == 0) { *why = "pos_enc_basis == 0 with an encoding: use NCA_ENC_NONE"; return NCA_E_INVALID; }
    y.C = nca_net_cin(n); y.Cout = nca_net_cout(n);
    if (y.C > NCA_WIDE_MAX_C || y.Cout > NCA_WIDE_MAX_C) { *why = "general kernels: at most 8 input and 8 output channels"; return NCA_E_UNSUPPORTED; }
    y.F = n.F; y.NL = 1 + n.n_hidden + n.n_late;
    y.enc_mode = n.enc_mode; y.L = n.L; y.T = n.T; y.P = n.T > 0 ? n.P : 0;
    y.Kenc = n.enc_mode == NCA_ENC_NONE ? y.C : (n.enc_mode == NCA_ENC_BANDS ? y.C * (1 + 2 * n.L) : 2 * y.C * n.L);
    y.K0 = y.Kenc + n.T;
    y.K0p = (y.K0 + y.P + 15) / 16 * 16;
    int off = y.P * y.T;
    int64_t poff = 0;
    y.lat_off = 0;
    for (int j = 0; j < y.NL; ++j) {
        NcaWideLayer& l = y.layer[j];
        if (j == 0) { l.kind = NCA_IN_ENC; l.K = y.K0; l.Kp = y.K0p; }
        else if (n.n_late > 0 && j == 1 + n.n_hidden) { l.kind = NCA_IN_SKIP; l.K = y.K0 + y.F; l.Kp = y.K0p + y.F; }
        else { l.kind = NCA_IN_HID; l.K = y.F; l.Kp = y.F; }
        l.w_off = off; off += y.F * l.K;
        l.b_off = off; off += y.F;
        l.pw_off = poff; poff += (int64_t)y.F * l.Kp;
    }
    y.wo_off = off; off += y.Cout * y.F;
    y.bo_off = off; off += y.Cout;
    y.n_params = off;
    y.pb_off = poff; poff += (int64_t)y.NL * y.F;
    y.pwo_off = poff; poff += (int64_t)y.Cout * y.F;
    y.pbo_off = poff; poff += y.Cout;
    y.packed_floats = (poff + 3) / 4 * 4;
    *out = y;
    return NCA_OK;
}
// natural column j of layer l -> column of Wp
NCA_HD inline int nca_wide_col(const NcaWideLayout& y, const NcaWideLayer& l, int j) { return (l.kind == NCA_IN_SKIP && j >= y.K0) ? y.K0p + (j - y.K0) : j; }

// where a chunk's samples come from
struct NcaWideGeom {
    int32_t mode;                 // NCA_MODE_RAYS / NCA_MODE_POINTS
    int32_t S, ray_is_f64, C;
    const void* origins; const void* dirs;
    const float* z; int64_t zs_r;
    const int32_t* phase; int64_t ps_r, ps_s;
    const float* pts;             // [N][C]
};
struct NcaWideEncArgs {
    NcaWideGeom g;
    int64_t n0, n_valid, rows;    // first sample of the chunk, samples in it, rows of X0 to write (the rest zeros)
    int32_t enc_mode, L, Kenc, T, P, K0, K0p, onehot;
    const float* win; const float* four; const float* lat;
    float* X0;
};
hipError_t nca_launch_wide_encode(const NcaWideEncArgs& a, hipStream_t st);
hipError_t nca_launch_wide_pack(const NcaWideLayout& y, const float* prm, float* out, hipStream_t st);

// C[r][c] (+ epilogue) = sum_k A(r, k) B(c, k), k over seg 0 then seg 1 of A (B's k runs on)
enum { NCA_WG_FWD = 0,            // A [r][k] k contiguous, B [c][k] k contiguous;  + bias[c], ReLU (or not)
       NCA_WG_DGRAD = 1,          // A [r][k],              B [k][c] c contiguous;  x ReLU mask bits of the layer input (maskbits)
       NCA_WG_WGRAD = 2 };        // A [k][r] r contiguous, B [k][c];  contraction split over gridDim.z, partial sums to C + z * split_stride
struct NcaWideGemmArgs {
    const float* A[2]; int64_t lda[2]; int64_t ka[2];     // ka multiples of 16 (seg 1 may be empty)
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int64_t rows, cols;           // extents: loads beyond them read as zero, stores beyond them are dropped
    const float* bias; int32_t relu;
    uint32_t* maskbits;           // ReLU bit masks of a [rows][cols] layer output, per 128 x 128 tile and thread two words (bit (2 bi + bj) 16 + v = accumulator register v of MFMA
                                  // block (bi, bj) is positive): NCA_WG_FWD writes them (null: not), NCA_WG_DGRAD -- whose output has the same shape and tiling -- multiplies
                                  // by them instead of reading the layer output back (64 loads per lane and rows x cols x 4 bytes less per launch)
    int64_t split_stride;         // NCA_WG_WGRAD
    float* rowsum;                // NCA_WG_WGRAD: null, or where the sums of A over the contraction go -- rowsum[z * split_stride + r] = sum_k A(r, k) of split z (the
                                  // bias gradient sum_n D[n][f] rides along with the weight gradient: the column-0 workgroups add up the A slabs they stage anyway)
    int32_t nsplit, pad;
};
hipError_t nca_launch_wide_gemm(int kind, const NcaWideGemmArgs& a, hipStream_t st);

// output layer: raw[n][o] = <H[n], Wo[o]> + bo[o]
hipError_t nca_launch_wide_head_fwd(int64_t n_valid, int F, int Cout, const float* H, const float* Wo, const float* bo, float* raw, hipStream_t st);
// D[n][f] = (H[n][f] > 0) sum_o g[n][o] Wo[o][f]  (rows >= n_valid: 0; `rows` rows written)
hipError_t nca_launch_wide_head_bwd(int64_t n_valid, int64_t rows, int F, int Cout, const float* H, const float* Wo, const float* g, float* D, hipStream_t st);
// weighted column sums over the rows of a chunk, split over gridDim.y:  out[s][o][f] = sum_{n in split s} g[n][o] X[n][f]   (g == null: Cout = 1, weights 1)
// and gsum[s][o] = sum_n g[n][o] (null: not formed)
hipError_t nca_launch_wide_colsum(int64_t rows, int F, int Cout, const float* X, int64_t ldx, const float* g, float* out, int64_t split_stride, float* gsum, int nsplit,
                                  hipStream_t st);

// fixed-order sum of the split slabs into the natural gradient (added to what is there)
struct NcaWideReduceArgs {
    NcaWideLayout y;
    const float* slab; int64_t split_stride; int32_t nsplit, pad;      // a split is laid out like the packed image
    const float* packed;          // (latent gradient: W0's latent columns)
    float* grads;                 // natural
    float* esum;                  // [F][P] scratch: per-phase sums of D_0 (T > 0)
};
hipError_t nca_launch_wide_reduce(const NcaWideReduceArgs& a, hipStream_t st);
