// nca_layout.hpp -- geometry shared by host code and kernels: where each layer's parameters sit in
// the natural flat buffer, and how they are re-ordered into the images the MFMA loops stream.
//
// Orientation used everywhere: activations are kept TRANSPOSED, H[feature][sample], with the sample
// on the MFMA lane (column) and the features in accumulator registers.  A layer is D = W * H:
//   A operand  = weights   W[out-feature row][k]      (streamed from an LDS image)
//   B operand  = H[k][sample]                          (the previous layer's accumulator registers,
//                                                       used in place: no lane movement, no LDS)
// For v_mfma_f32_32x32x2_f32 the accumulator register i of lane-half h holds row rho(i)+4h, so the
// k order of a hidden layer is permuted: k-step s = 16t+i reads rows 32t + rho(i) + 4h (h = 0,1).
#pragma once
#include <stdint.h>
#include "../../include/nerfca_hip.h"

#if defined(__HIPCC__)
#define NCA_HD __host__ __device__
#else
#define NCA_HD          // (the host-only unit tests of the layout include this header with a plain C++ compiler)
#endif

#define NCA_MAX_LAYERS 12
#define NCA_MAX_STAGES 48
#define NCA_MAX_JOBS 40
#define NCA_MAX_KSTEPS 64   // k-steps one LDS weight image may hold (a hidden layer of width 128 uses all 64)

enum { NCA_IN_ENC = 0, NCA_IN_HID = 1, NCA_IN_SKIP = 2 };

struct NcaLayerL {
    int32_t kind;        // NCA_IN_*
    int32_t K;           // natural fan-in
    int32_t ksteps;      // MFMA k-steps of the whole layer
    int32_t ksteps_enc;  // of which: encoded-input steps (ENC / SKIP layers)
    int32_t w_off;       // natural flat offsets (floats)
    int32_t b_off;
    uint32_t img_off;    // forward image  (bytes from the packed base)
    uint32_t img_bytes;  // incl. bias tail (and Wo/bo tail on the last layer)
    uint32_t imgT_off;   // transposed image for dgrad (hidden part only); 0 bytes on layer 0
    uint32_t imgT_bytes;
    uint32_t img2_off;   // SKIP layers stream in two stages: `img` = encoded part + bias tail, `img2` = hidden part (+ Wo tail)
    uint32_t img2_bytes; // x3 hidden layers of width >= 64 likewise: `img` = first half of the k-steps + bias tail, `img2` = the rest (+ Wo tail)
    uint32_t imgT2_off;  // x3: second half of the k-steps of the transposed image
    uint32_t imgT2_bytes;
};

struct NcaLayout {
    int32_t F, MT, NL;          // width, row tiles (F/32), number of F-wide layers
    int32_t enc_mode, L, T, P;
    int32_t Kenc;               // encoded coordinate features (3, 3+6L or 6L)
    int32_t K0;                 // Kenc + T
    int32_t enc_steps;          // k-steps that carry the K0 inputs
    int32_t K0rows;             // rows of the stored input block: K0 (+ P one-hot phase rows when T > 0)
    int32_t K0rows_pad;         // rounded up to 32
    int32_t lat_off, wo_off, bo_off, n_params;
    uint32_t packed_bytes, max_img_bytes;
    int32_t x3;                 // f32 path: hidden-width contractions on the bf16 matrix cores from exact 3-way bf16 splits (below)
    int32_t reserved;
    NcaLayerL layer[NCA_MAX_LAYERS];
};

NCA_HD inline int nca_rho(int i) { return (i & 3) + 8 * (i >> 2); }
// natural row of hidden k-step s, lane-half h
NCA_HD inline int nca_kidx_hidden(int s, int h) { return 32 * (s >> 4) + nca_rho(s & 15) + 4 * h; }

// natural input-feature indices (a for lane-half 0, b for lane-half 1) of encoded k-step s; -1 = zero pad
NCA_HD inline void nca_enc_pair(const NcaLayout& y, int s, int* ia, int* ib) {
    int base = 0;
    if (y.enc_mode == NCA_ENC_FOURIER) {
        int n = 3 * y.L;
        if (s < n) { *ia = s; *ib = n + s; return; }
        base = n;
    } else {
        if (s == 0) { *ia = 0; *ib = 1; return; }
        if (s == 1) { *ia = 2; *ib = -1; return; }
        base = 2;
        if (y.enc_mode == NCA_ENC_BANDS) {
            int n = 3 * y.L;
            if (s < 2 + n) {
                int k = (s - 2) / 3, c = (s - 2) % 3;
                *ia = 3 + 6 * k + c;
                *ib = 6 + 6 * k + c;
                return;
            }
            base = 2 + n;
        }
    }
    int u = s - base;
    *ia = y.Kenc + 2 * u;
    *ib = (2 * u + 1 < y.T) ? y.Kenc + 2 * u + 1 : -1;
}

// x3 images (f32 path, hidden-width contractions): weights split EXACTLY into three bf16 pieces w = w1 + w2 + w3 and laid
// out as A fragments of v_mfma_f32_32x32x16_bf16: [piece][row tile][k-step of the sub-stage][lane][8 bf16].  Lane l of
// k-step ks holds output row 32 m + (l & 31) and, as element j, the input feature nca_x3_kidx(ks, l >> 5, j) -- the order
// in which the previous layer's accumulator registers are packed into B fragments (registers 8 (ks & 1) + j of row tile
// ks >> 1, rows rho(.) + 4 h).  A layer of width >= 64 streams as two sub-stages of HALF THE K-STEPS each (all row
// tiles), so that an image (3 x 16 KiB + tail at width 128) fits the LDS double buffer and only half of the activations
// have to sit split in registers at a time.
NCA_HD inline int nca_x3_kidx(int ks, int h, int j) { return 32 * (ks >> 1) + nca_rho(8 * (ks & 1) + j) + 4 * h; }
NCA_HD inline int nca_x3_kh(int F) { return F >= 64 ? F / 32 : F / 16; }           // k-steps per sub-stage
NCA_HD inline uint32_t nca_x3_sub_bytes(int F) { return 3u * (uint32_t)(F / 32) * (uint32_t)nca_x3_kh(F) * 1024u; }

// sizes of the f32 images
NCA_HD inline uint32_t nca_img_w_bytes(int ksteps, int MT) { return (uint32_t)ksteps * 64u * (uint32_t)MT * 4u; }
NCA_HD inline uint32_t nca_img_tail_bytes(int MT) { return 2u * (uint32_t)MT * 16u * 4u; }  // bias (or Wo) in accumulator order

inline int nca_build_layout(const NcaNet& n, NcaLayout* out, const char** why, bool x3 = false) {
    NcaLayout y{};
    y.x3 = x3 ? 1 : 0;
    if (!(n.F == 32 || n.F == 64 || n.F == 128)) { *why = "num_filters must be 32, 64 or 128"; return NCA_E_UNSUPPORTED; }
    if (n.n_hidden < 0 || n.n_late < 0 || 1 + n.n_hidden + n.n_late > NCA_MAX_LAYERS) { *why = "too many layers"; return NCA_E_UNSUPPORTED; }
    if (n.T < 0 || n.T > 32 || (n.T > 0 && n.P <= 0) || n.P > 64) { *why = "num_time_dim must be in [0,32], phases in [1,64]"; return NCA_E_UNSUPPORTED; }
    if (n.T > 0 && n.n_late > 0) { *why = "Temporal with num_late_layers > 0 has no output in the reference (Temporal.py:128-135)"; return NCA_E_UNSUPPORTED; }
    if (n.enc_mode < 0 || n.enc_mode > 2 || n.L < 0 || n.L > 16) { *why = "bad positional encoding"; return NCA_E_UNSUPPORTED; }
    if (n.enc_mode != NCA_ENC_NONE && n.L == 0) { *why = "pos_enc_basis == 0 with an encoding: use NCA_ENC_NONE"; return NCA_E_INVALID; }
    y.F = n.F; y.MT = n.F / 32; y.NL = 1 + n.n_hidden + n.n_late;
    y.enc_mode = n.enc_mode; y.L = n.L; y.T = n.T; y.P = n.T > 0 ? n.P : 0;
    y.Kenc = n.enc_mode == NCA_ENC_NONE ? 3 : (n.enc_mode == NCA_ENC_BANDS ? 3 + 6 * n.L : 6 * n.L);
    y.K0 = y.Kenc + n.T;
    int coord_steps = n.enc_mode == NCA_ENC_NONE ? 2 : (n.enc_mode == NCA_ENC_BANDS ? 2 + 3 * n.L : 3 * n.L);
    y.enc_steps = coord_steps + (n.T + 1) / 2;
    y.K0rows = y.K0 + y.P;
    y.K0rows_pad = (y.K0rows + 31) / 32 * 32;
    if (y.K0rows_pad > 128) { *why = "encoded input wider than 128 rows"; return NCA_E_UNSUPPORTED; }
    int off = 0;
    y.lat_off = 0;
    off += y.P * y.T;
    uint32_t boff = 0, maxb = 0;
    for (int j = 0; j < y.NL; ++j) {
        NcaLayerL& l = y.layer[j];
        if (j == 0) { l.kind = NCA_IN_ENC; l.K = y.K0; l.ksteps_enc = y.enc_steps; l.ksteps = y.enc_steps; }
        else if (n.n_late > 0 && j == 1 + n.n_hidden) { l.kind = NCA_IN_SKIP; l.K = y.K0 + y.F; l.ksteps_enc = y.enc_steps; l.ksteps = y.enc_steps + y.F / 2; }
        else { l.kind = NCA_IN_HID; l.K = y.F; l.ksteps_enc = 0; l.ksteps = y.F / 2; }
        if (l.ksteps_enc > NCA_MAX_KSTEPS || l.ksteps - l.ksteps_enc > NCA_MAX_KSTEPS) { *why = "a layer stage needs more than 64 MFMA k-steps (encoded input too wide)"; return NCA_E_UNSUPPORTED; }
        l.w_off = off; off += y.F * l.K;
        l.b_off = off; off += y.F;
        const uint32_t wo_tail = (j == y.NL - 1) ? nca_img_tail_bytes(y.MT) + 16u : 0u;
        l.img_off = boff;
        if (l.kind == NCA_IN_SKIP) {
            l.img_bytes = nca_img_w_bytes(l.ksteps_enc, y.MT) + nca_img_tail_bytes(y.MT);                 // encoded part + bias
            boff += (l.img_bytes + 1023u) & ~1023u;
            l.img2_off = boff;
            l.img2_bytes = nca_img_w_bytes(l.ksteps - l.ksteps_enc, y.MT) + wo_tail;                       // hidden part (+ Wo, bo)
            boff += (l.img2_bytes + 1023u) & ~1023u;
            if (l.img2_bytes > maxb) maxb = l.img2_bytes;
        } else if (x3 && l.kind == NCA_IN_HID) {
            const uint32_t sub = nca_x3_sub_bytes(y.F);
            const bool two = y.MT >= 2;
            l.img_bytes = sub + nca_img_tail_bytes(y.MT) + (two ? 0u : wo_tail);                             // first half + bias
            boff += (l.img_bytes + 1023u) & ~1023u;
            l.img2_off = two ? boff : 0u;
            l.img2_bytes = two ? sub + wo_tail : 0u;                                                         // second half (+ Wo, bo)
            if (two) boff += (l.img2_bytes + 1023u) & ~1023u;
            if (l.img2_bytes > maxb) maxb = l.img2_bytes;
        } else {
            l.img_bytes = nca_img_w_bytes(l.ksteps, y.MT) + nca_img_tail_bytes(y.MT) + wo_tail;
            boff += (l.img_bytes + 1023u) & ~1023u;   // images are DMA'd to LDS in 1 KiB pieces
            l.img2_off = 0; l.img2_bytes = 0;
        }
        if (l.img_bytes > maxb) maxb = l.img_bytes;
    }
    for (int j = 0; j < y.NL; ++j) {
        NcaLayerL& l = y.layer[j];
        if (l.kind == NCA_IN_ENC) { l.imgT_off = 0; l.imgT_bytes = 0; continue; }
        l.imgT_off = boff;
        l.imgT2_off = 0; l.imgT2_bytes = 0;
        if (x3) {
            l.imgT_bytes = nca_x3_sub_bytes(y.F);
            boff += (l.imgT_bytes + 1023u) & ~1023u;
            if (y.MT >= 2) {
                l.imgT2_off = boff;
                l.imgT2_bytes = nca_x3_sub_bytes(y.F);
                boff += (l.imgT2_bytes + 1023u) & ~1023u;
            }
        } else {
            l.imgT_bytes = nca_img_w_bytes(y.F / 2, y.MT);
            boff += (l.imgT_bytes + 1023u) & ~1023u;
        }
        if (l.imgT_bytes > maxb) maxb = l.imgT_bytes;
    }
    y.wo_off = off; off += y.F;
    y.bo_off = off; off += 1;
    y.n_params = off;
    y.packed_bytes = boff;
    y.max_img_bytes = maxb;
    *out = y;
    return NCA_OK;
}

// ------------------------------------------------------------------------------------------
// bf16 path (v_mfma_f32_32x32x16_bf16): a k-step is 16 features, a wave owns 64 samples (two 32-column
// tiles).  The encoded input of layer 0 lives in fixed SLOTS so that every register index in the
// kernel is a compile-time constant:
//     slot 0..Kenc-1   encoded coordinates in natural order        (Kenc <= 80 with latents, <= 96 without;
//                      fourier: slot 2i = sin_i, 2i+1 = cos_i)
//     slot 80..80+T-1  time latents                                (T <= 16)
//     slot 96..96+P-1  one-hot phase rows (backward scratch only)  (P <= 16)
// Hidden layers read the previous accumulator tiles as B operands: k-step 2t+s, element j of lane
// half h is feature 32t + 16s + 8(j>>2) + 4h + (j&3).
// ------------------------------------------------------------------------------------------
#define NCA_BF_K0SLOTS 96      // layer-0 input slots fed to the MFMAs (6 k-steps)
#define NCA_BF_ENCROWS 112     // slots of the stored input block (adds the one-hot rows)
#define NCA_BF_LAT_SLOT 80
#define NCA_BF_HOT_SLOT 96

NCA_HD inline int nca_bf_kidx_hidden(int ks, int h, int j) { return 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }
// natural input index of a layer-0 slot (-1: padding)
NCA_HD inline int nca_bf_slot_to_nat(const NcaLayout& y, int slot) {
    if (y.enc_mode == NCA_ENC_FOURIER && slot < y.Kenc) return (slot & 1) ? 3 * y.L + (slot >> 1) : (slot >> 1);   // (sin_i, cos_i) interleaved
    if (slot < y.Kenc) return slot;
    if (slot >= NCA_BF_LAT_SLOT && slot < NCA_BF_LAT_SLOT + y.T) return y.Kenc + (slot - NCA_BF_LAT_SLOT);
    return -1;
}

inline int nca_build_layout_bf16(const NcaNet& n, NcaLayout* out, const char** why) {
    int rc = nca_build_layout(n, out, why);
    if (rc != NCA_OK) return rc;
    NcaLayout& y = *out;
    if (n.T > 16 || y.P > 16) { *why = "bf16 path: num_time_dim and phases must be <= 16"; return NCA_E_UNSUPPORTED; }
    if (y.Kenc > (n.T > 0 ? NCA_BF_LAT_SLOT : NCA_BF_K0SLOTS)) { *why = "bf16 path: encoded input too wide (pos_enc_basis <= 12 with latents, <= 15 without)"; return NCA_E_UNSUPPORTED; }
    uint32_t boff = 0, maxb = 0;
    const uint32_t tail = nca_img_tail_bytes(y.MT);
    for (int j = 0; j < y.NL; ++j) {
        NcaLayerL& l = y.layer[j];
        const uint32_t wo_tail = (j == y.NL - 1) ? tail + 16u : 0u;
        l.img_off = boff;
        l.img2_off = 0; l.img2_bytes = 0;
        if (l.kind == NCA_IN_SKIP) {
            // a skip layer (CPPN with num_late_layers > 0, model/CPPN.py:53-58, 102-106) reads cat[encoded input, h]: two images, two LDS stages --
            // `img` = the encoded part (the layer-0 slots: 6 k-steps) + the bias tail, `img2` = the hidden part (F / 16 k-steps) (+ [Wo | bo] on the last layer)
            l.ksteps_enc = NCA_BF_K0SLOTS / 16;
            l.ksteps = l.ksteps_enc + y.F / 16;
            l.img_bytes = (uint32_t)y.MT * (uint32_t)l.ksteps_enc * 1024u + tail;
            boff += (l.img_bytes + 1023u) & ~1023u;
            l.img2_off = boff;
            l.img2_bytes = (uint32_t)y.MT * (uint32_t)(y.F / 16) * 1024u + wo_tail;
            boff += (l.img2_bytes + 1023u) & ~1023u;
            if (l.img2_bytes > maxb) maxb = l.img2_bytes;
        } else {
            l.ksteps = (j == 0) ? NCA_BF_K0SLOTS / 16 : y.F / 16;
            l.ksteps_enc = (j == 0) ? l.ksteps : 0;
            l.img_bytes = (uint32_t)y.MT * (uint32_t)l.ksteps * 1024u + tail + wo_tail;
            boff += (l.img_bytes + 1023u) & ~1023u;
        }
        if (l.img_bytes > maxb) maxb = l.img_bytes;
    }
    for (int j = 0; j < y.NL; ++j) {
        NcaLayerL& l = y.layer[j];
        l.imgT2_off = 0; l.imgT2_bytes = 0;
        if (j == 0) { l.imgT_off = 0; l.imgT_bytes = 0; continue; }
        l.imgT_off = boff;                                  // (a skip layer: the transposed image of its HIDDEN part -- the sweep needs no gradient of the encoded input)
        l.imgT_bytes = (uint32_t)y.MT * (uint32_t)(y.F / 16) * 1024u;
        boff += l.imgT_bytes;
        if (l.imgT_bytes > maxb) maxb = l.imgT_bytes;
    }
    y.packed_bytes = boff;
    y.max_img_bytes = maxb;
    return NCA_OK;
}
// does the net have a skip layer (bf16 mode: streaming kernels, their SKIP instantiation)
NCA_HD inline bool nca_has_skip(const NcaLayout& y) {
    for (int j = 0; j < y.NL; ++j) if (y.layer[j].kind == NCA_IN_SKIP) return true;
    return false;
}

// The bf16 mode's forward store (8-bit staging, NCA_OPT_STAGE_FP8): the blocks that only the weight-gradient kernel reads cross HBM as
// 8-bit floats, and the backward recomputes nothing.  (A bf16-STAGED store -- 64 x F bytes per block, last layer recomputed -- existed
// until round 3; without a store the recompute backward keeps bf16 blocks in its own scratch: nca_bf_tile_bytes.)
//   hidden block j = output of layer j = input of layer j+1, j = 0..NL-2, behind the input block of a tile of the forward store:
//       e4m3 (32 x F bytes, [row tile][lane][16 B]: byte i = accumulator register i, value x 2^NCA_H8_LOG2); the store also holds
//       the ReLU masks of ALL NL layers and the raw outputs -- the backward reads masks and raw outputs only, the weight-gradient
//       kernel reads the blocks
//   output-gradient block l = 0..NL-1 of a tile of the backward's D region:
//       e5m2 (32 x F bytes, same byte order, value x the tile's power-of-two scale), or bf16 fragments when the depth-gradient
//       kernel is to read D_0.  Under fp8 staging block NL-1 holds relu'(H_{NL-1}) g WITHOUT the factor Wo[f]: the weight-gradient
//       kernel then leaves S[f][k] = sum_n relu' g H_{NL-2}[k] and s[f] = sum_n relu' g, from which the reduce kernel forms
//           dW_{NL-1}[f][k] = Wo[f] S[f][k],  db_{NL-1}[f] = Wo[f] s[f],  dWo[f] = sum_k W_{NL-1}[f][k] S[f][k] + b_{NL-1}[f] s[f]
//       -- the last identity because sum_n g relu(z) = sum_n g relu'(z) z with z = W_{NL-1} H_{NL-2} + b_{NL-1}: the output layer's
//       weight gradient needs neither the layer's input in the store nor a pass of its own.
//       As e5m2 that block has ONE distinct byte per sample, e5m2(g x scale), at the features whose mask bit is set: it is not
//       stored.  Its place in the tile keeps, in its first 128 bytes, u32[32] = that byte x 0x00010001 per sample of the 32-sample
//       tile, and the weight-gradient job rebuilds the fragments from them and the forward's mask bits (wgrad_job_mx, EXPAND).
//       Behind all nets' blocks of a 32-sample tile a 128-byte record; the first record of a 64-sample wave tile holds, per net,
//       the INVERSE scale of the tile (f32[2]) and the tile's sum of d loss / d raw (f32[2]: the output layer's bias gradient,
//       added up over the tiles in tile order by nca_sum_tile_records, whichever wave ran the tile)
#define NCA_H8_LOG2 2
#define NCA_D8_LOG2 4            // the tile's largest |d loss / d raw| is scaled into [2^4, 2^5)
#define NCA_D8_REC_BYTES 128
// the input block of a 32-sample tile: bf16 fragments [k-step 0..6][lane][16 B] (6 k-steps of layer-0 slots + the one-hot phase
// slots), or, under fp8 staging, e4m3 (x 2^NCA_H8_LOG2) [32-slot tile 0..3][lane][16 B] -- byte 8 a + j of lane (r, h) = slot
// 32 t + 16 a + 8 h + j of sample r: what the weight-gradient kernel rounds the bf16 block to anyway
NCA_HD inline int64_t nca_bf_ebytes(bool h8) { return h8 ? 32 * 128 : 32 * (int64_t)NCA_BF_ENCROWS * 2; }
NCA_HD inline int64_t nca_bf_hoff(const NcaLayout& y, int j, bool h8) { return (int64_t)j * (h8 ? 32 : 64) * y.F; }
NCA_HD inline int64_t nca_bf_hbytes(const NcaLayout& y, bool h8) { return y.NL < 2 ? 0 : nca_bf_hoff(y, y.NL - 1, h8); }
NCA_HD inline int64_t nca_bf_doff(const NcaLayout& y, int l, bool d8) { return (int64_t)l * (d8 ? 32 : 64) * y.F; }
NCA_HD inline int64_t nca_bf_dbytes(const NcaLayout& y, bool d8) { return nca_bf_doff(y, y.NL, d8); }

// bytes of one 32-sample tile of the bf16 backward scratch for this net:
//   [input block 32 x 112][inputs of layers 1..NL-1: 32 x F each][output gradients D_0..D_{NL-1}: 32 x F each]
NCA_HD inline int64_t nca_bf_tile_bytes(const NcaLayout& y) {
    return 32 * (int64_t)NCA_BF_ENCROWS * 2 + (int64_t)(2 * y.NL - 1) * 32 * y.F * 2;
}
