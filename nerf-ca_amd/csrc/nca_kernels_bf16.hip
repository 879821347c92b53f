// nca_kernels_bf16.hip -- gfx950 kernels of the bf16 (throughput) path: bf16 MFMA operands
// (v_mfma_f32_32x32x16_bf16), f32 accumulation, f32 master weights, f32 encoding arithmetic.
//
//   nca_pack_bf16        natural flat f32 parameters -> bf16 MFMA-ordered LDS images (+ f32 bias tails)
//   nca_fused_bf16<F,MODE,S8,RES>  a wave owns 64 consecutive samples of one ray: lane = sample for the
//         encoding, then two 32-column tiles for the MFMAs (v_permlane32_swap builds both operand
//         tiles from the per-lane features).  Layers run row-tile-outer: the 32x32 accumulator pair of
//         row tile m is finished, ReLU'd, packed pairwise to bf16 and becomes k-steps 2m, 2m+1 of the
//         next layer's B operand -- activations never leave registers.
//         MODE 0 forward; 2 storing forward (e4m3 layer inputs, ReLU masks, raw outputs to the caller's store);
//         5 backward from that store (no recompute: output-layer gradients, dgrad sweep, e5m2 output gradients
//         to the chunk scratch); 1 recompute backward without a store (bf16 blocks to one scratch).
//   nca_wgrad_bf16<F,D8>  dW = D * H^T over samples: one wave per (layer, sample split) keeps dW of a whole layer in
//         256 accumulator registers; operand tiles come HBM -> LDS by LDS-DMA into a per-wave ring, are transposed by
//         MFMAs against an identity (samples onto K) and contracted -- D8: e5m2 x e4m3 on the MX-fp8 path, one
//         v_mfma_scale_f32_32x32x64_f8f6f4 per 32x32 block of dW and 64-sample tile; else bf16.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "nca_kernels.hpp"

// This file holds ONLY the code that runs in the shipped library.  The timing-only elimination builds (NCA_EXP) and the measured-and-
// switched-off variants of rounds 2 and 3 (NCA_BF_PIPE / NCA_BF_PIPE2, NCA_WGRAD_TR, NCA_ONCHIP_NR, NCA_CHAIN8, the s_setprio / lock /
// split-pipeline experiments) live in the repository's history: tools/r03_experiments.sh checks the round-3 kernel sources out (git tag
// r03-kernels) and builds any of them there; DESIGN.md 4.4 / 7 has their numbers.
int nca_kernels_exp_mask() { return 0; }
int nca_kernels_variant_mask() { return NCA_WAVES << 8; }       // (bits 0..3 named the round-3 A/B macros: all gone, all 0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

// fp8 staging (nca_layout.hpp): four f32 -> one dword of e4m3 / e5m2 bytes, value / 2^floor(log2 div) (v_cvt_scalef32_pk_*: the
// scaling is part of the conversion; round to nearest even; with MODE.FP16_OVFL set -- s8_mode() -- out-of-range values
// saturate to +-448 / +-57344 instead of becoming NaN / inf: measured, tools/fp8_cvt_probe.hip)
__device__ __forceinline__ unsigned cvt4_e4m3(float a, float b, float c, float d, float div) {
    s16x2 v = {0, 0};
    v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(v, a, b, div, false);
    v = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(v, c, d, div, true);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned cvt4_e5m2(float a, float b, float c, float d, float div) {
    s16x2 v = {0, 0};
    v = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(v, a, b, div, false);
    v = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(v, c, d, div, true);
    return __builtin_bit_cast(unsigned, v);
}
// the same from a packed bf16 pair (two values per instruction; the pair has been rounded to bf16 already).
// (the conversions write one half of a register and keep the other.  Both halves get written, so the start value is immaterial: a
// zero costs a v_mov per word -- 8 per row tile of an epilogue that is bound by vector-instruction issue -- while an empty asm
// "defines" a register without an instruction)
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 undef_pair() {
    int v;
    asm volatile("" : "=v"(v));     // (volatile: one definition per use, or the compiler shares one and copies it)
    return __builtin_bit_cast(s16x2, v);
}
__device__ __forceinline__ unsigned cvt4_e4m3_pk(unsigned lo, unsigned hi, float div) {
    s16x2 v = undef_pair();
    v = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(v, __builtin_bit_cast(bf16x2v, lo), div, false);
    v = __builtin_amdgcn_cvt_scalef32_pk_fp8_bf16(v, __builtin_bit_cast(bf16x2v, hi), div, true);
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned cvt4_e5m2_pk(unsigned lo, unsigned hi, float div) {
    s16x2 v = undef_pair();
    v = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(v, __builtin_bit_cast(bf16x2v, lo), div, false);
    v = __builtin_amdgcn_cvt_scalef32_pk_bf8_bf16(v, __builtin_bit_cast(bf16x2v, hi), div, true);
    return __builtin_bit_cast(unsigned, v);
}
// Epilogue arithmetic on PACKED bf16 pairs (the vector ALU, not the matrix pipe, bounds these kernels: ~6 plain vector
// instructions hide behind one 32x32x16 MFMA, tools/valu_mfma_samewave.hip, and a row tile has 2 values per lane and MFMA):
// ReLU as one v_pk_max_i16 (rounding to bf16 commutes with ReLU; negative floats and -0 are negative integers), "is positive"
// as one v_pk_min_u16 against 1 (a ReLU output is >= +0).
// (Inline assembly on purpose: written with vector types the compiler splits the pair again -- two single conversions and a
// v_perm_b32 per pack, compare + select per half for the minimum: 5 instructions where 1 is meant.  The operands are VGPRs
// written by vector-ALU instructions, never straight MFMA results, so no MFMA-to-VALU wait states are involved.)
__device__ __forceinline__ unsigned pack2_pk(float lo, float hi) {       // pack2 whose result the optimiser must keep as ONE dword
    unsigned w = pack2(lo, hi);
    asm("" : "+v"(w));
    return w;
}
__device__ __forceinline__ unsigned relu_pk(unsigned w) {
    unsigned r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(w));
    return r;
}
__device__ __forceinline__ unsigned pos_pk(unsigned w) {
    unsigned r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(w), "s"(0x00010001u));
    return r;
}
// a packed bf16 pair with the halves kept whose flag (0 / 1 per half) is set: ONE multiply (x 1 keeps the bits, x 0 clears them)
// (NOT as inline assembly: the compiler's hazard recognizer does not count an asm statement as a vector-ALU write, and a matrix
// instruction that reads the result within two wait states gets the register's OLD contents -- seen with exactly this multiply)
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned keep_pk(unsigned w, unsigned flags01) {
    return __builtin_bit_cast(unsigned, (u16x2)(__builtin_bit_cast(u16x2, w) * __builtin_bit_cast(u16x2, flags01)));
}
// (a << k) | b in one instruction (the compiler pairs two shifts with a v_or3 instead: 1.5 per term).  Inline assembly: only for
// values that no matrix instruction reads (see keep_pk) -- these are the mask words, which go to memory.
template <int K>
__device__ __forceinline__ unsigned shl_or(unsigned a, unsigned b) {
    unsigned r;
    asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(K), "v"(b));
    return r;
}
// ReLU of eight packed bf16 pairs (one 32-feature row tile of one column tile: fragment words 0..7) and their "is positive" bits as
// one field -- bit k / 16+k = low / high half of word k -- in ONE asm block of 8 v_pk_max_i16 + 8 v_pk_min_u16 + 7 v_lshl_or_b32.
// As 23 separate asm statements the compiler's hazard recognizer put an s_nop between every producer and the asm that reads it (it
// must assume an asm writes or reads a register half, the dst_sel forwarding hazard): 26 s_nop per row tile, a quarter of the
// epilogue's issue slots, for full-dword instructions that need none.  The inputs come from compiler-visible conversions (the
// MFMA-to-VALU wait states are the compiler's), the outputs are first read by the NEXT layer's matrix instructions, a whole
// row-tile epilogue later at the least (see keep_pk for why that distance matters).
__device__ __forceinline__ unsigned relu_mask8(unsigned (&w)[8]) {
    unsigned f, t1, t2, t3;
    asm("v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0\n\tv_pk_max_i16 %2, %2, 0\n\tv_pk_max_i16 %3, %3, 0\n\t"
        "v_pk_max_i16 %4, %4, 0\n\tv_pk_max_i16 %5, %5, 0\n\tv_pk_max_i16 %6, %6, 0\n\tv_pk_max_i16 %7, %7, 0\n\t"
        "v_pk_min_u16 %8, %0, %12\n\tv_pk_min_u16 %9, %1, %12\n\tv_pk_min_u16 %10, %2, %12\n\tv_pk_min_u16 %11, %3, %12\n\t"
        "v_lshl_or_b32 %8, %9, 1, %8\n\tv_pk_min_u16 %9, %4, %12\n\t"
        "v_lshl_or_b32 %8, %10, 2, %8\n\tv_pk_min_u16 %10, %5, %12\n\t"
        "v_lshl_or_b32 %8, %11, 3, %8\n\tv_pk_min_u16 %11, %6, %12\n\t"
        "v_lshl_or_b32 %8, %9, 4, %8\n\tv_pk_min_u16 %9, %7, %12\n\t"
        "v_lshl_or_b32 %8, %10, 5, %8\n\tv_lshl_or_b32 %8, %11, 6, %8\n\tv_lshl_or_b32 %8, %9, 7, %8"
        : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]), "=&v"(f), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "s"(0x00010001u));
    return f;
}
// the ReLU alone (layers whose mask nobody stores)
__device__ __forceinline__ void relu8(unsigned (&w)[8]) {
    asm("v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0\n\tv_pk_max_i16 %2, %2, 0\n\tv_pk_max_i16 %3, %3, 0\n\t"
        "v_pk_max_i16 %4, %4, 0\n\tv_pk_max_i16 %5, %5, 0\n\tv_pk_max_i16 %6, %6, 0\n\tv_pk_max_i16 %7, %7, 0"
        : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
}
// the field alone, of words that are ReLU outputs already (the last layer packs its f32 ReLU)
__device__ __forceinline__ unsigned mask8(const unsigned (&w)[8]) {
    unsigned f, t1, t2, t3;
    asm("v_pk_min_u16 %0, %4, %12\n\tv_pk_min_u16 %1, %5, %12\n\tv_pk_min_u16 %2, %6, %12\n\tv_pk_min_u16 %3, %7, %12\n\t"
        "v_lshl_or_b32 %0, %1, 1, %0\n\tv_pk_min_u16 %1, %8, %12\n\t"
        "v_lshl_or_b32 %0, %2, 2, %0\n\tv_pk_min_u16 %2, %9, %12\n\t"
        "v_lshl_or_b32 %0, %3, 3, %0\n\tv_pk_min_u16 %3, %10, %12\n\t"
        "v_lshl_or_b32 %0, %1, 4, %0\n\tv_pk_min_u16 %1, %11, %12\n\t"
        "v_lshl_or_b32 %0, %2, 5, %0\n\tv_lshl_or_b32 %0, %3, 6, %0\n\tv_lshl_or_b32 %0, %1, 7, %0"
        : "=&v"(f), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(w[0]), "v"(w[1]), "v"(w[2]), "v"(w[3]), "v"(w[4]), "v"(w[5]), "v"(w[6]), "v"(w[7]), "s"(0x00010001u));
    return f;
}

__device__ __forceinline__ void s8_mode() { asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }

__device__ __forceinline__ float bf_lo(unsigned w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
__device__ __forceinline__ bf16x8 frag(u32x4 x) { return __builtin_bit_cast(bf16x8, x); }
// scratch blocks are written once and read once by another kernel: stream them past the caches
__device__ __forceinline__ void store_nt(char* p, u32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
}
__device__ __forceinline__ u32x4 load_nt(const char* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
// ReLU as ONE integer max on the bit pattern (negative floats and -0 are negative integers): no canonicalising
// v_max x,x,x in front as fmaxf would get, and -- unlike an inline-asm v_max_f32 -- visible to the compiler's hazard
// recogniser, which must put the wait states between an MFMA and the first VALU read of its result.
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// ------------------------------------------------------------------------------------------
// pack
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pack_bf16_body(const NcaLayout& y, const float* __restrict__ prm, unsigned* __restrict__ out) {
    const uint32_t total = y.packed_bytes / 4u;
    for (uint32_t wd = blockIdx.x * blockDim.x + threadIdx.x; wd < total; wd += gridDim.x * blockDim.x) {
        unsigned v = 0u;
        const uint32_t byte = wd * 4u;
        for (int j = 0; j < y.NL; ++j) {
            const NcaLayerL& l = y.layer[j];
            const bool skip = l.kind == NCA_IN_SKIP;
            // image 1: the whole layer -- or, of a skip layer, its encoded part (layer-0 slots) -- + the bias tail (+ [Wo | bo] on a last layer that is not a skip layer)
            if (byte >= l.img_off && byte < l.img_off + l.img_bytes) {
                const int nks = skip ? l.ksteps_enc : l.ksteps;
                const uint32_t wbytes = (uint32_t)y.MT * (uint32_t)nks * 1024u;
                const uint32_t off = byte - l.img_off;
                const uint32_t tail = 2u * (uint32_t)y.MT * 16u;   // floats
                if (off < wbytes) {
                    float two[2];
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const uint32_t e = off / 2u + e2;            // bf16 element index
                        const int jj = e % 8, lane = (e / 8) % 64, ks = (e / 512) % nks, m = e / (512 * nks);
                        const int r = lane & 31, h = lane >> 5;
                        int k;
                        if (j == 0 || skip) k = nca_bf_slot_to_nat(y, 16 * ks + 8 * h + jj);
                        else k = nca_bf_kidx_hidden(ks, h, jj);
                        two[e2] = k >= 0 ? prm[l.w_off + (32 * m + r) * l.K + k] : 0.f;
                    }
                    v = pack2(two[0], two[1]);
                } else {
                    uint32_t q = (off - wbytes) / 4u;
                    float f = 0.f;
                    if (q < tail) {
                        const int i = q % 16, m = (q / 16) % y.MT, h = q / (16 * y.MT);
                        f = prm[l.b_off + 32 * m + nca_rho(i) + 4 * h];
                    } else if (j == y.NL - 1 && !skip) {
                        q -= tail;
                        if (q < tail) {
                            const int i = q % 16, m = (q / 16) % y.MT, h = q / (16 * y.MT);
                            f = prm[y.wo_off + 32 * m + nca_rho(i) + 4 * h];
                        } else if (q == tail) {
                            f = prm[y.bo_off];
                        }
                    }
                    v = __builtin_bit_cast(unsigned, f);
                }
            }
            // image 2 of a skip layer: its hidden part (columns K0 .. K0 + F - 1 of the natural weight) (+ [Wo | bo] on the last layer)
            if (skip && byte >= l.img2_off && byte < l.img2_off + l.img2_bytes) {
                const int KS = y.F / 16;
                const uint32_t wbytes = (uint32_t)y.MT * (uint32_t)KS * 1024u;
                const uint32_t off = byte - l.img2_off;
                const uint32_t tail = 2u * (uint32_t)y.MT * 16u;
                if (off < wbytes) {
                    float two[2];
                    for (int e2 = 0; e2 < 2; ++e2) {
                        const uint32_t e = off / 2u + e2;
                        const int jj = e % 8, lane = (e / 8) % 64, ks = (e / 512) % KS, m = e / (512 * KS);
                        const int r = lane & 31, h = lane >> 5;
                        two[e2] = prm[l.w_off + (32 * m + r) * l.K + y.K0 + nca_bf_kidx_hidden(ks, h, jj)];
                    }
                    v = pack2(two[0], two[1]);
                } else if (j == y.NL - 1) {
                    uint32_t q = (off - wbytes) / 4u;
                    float f = 0.f;
                    if (q < tail) {
                        const int i = q % 16, m = (q / 16) % y.MT, h = q / (16 * y.MT);
                        f = prm[y.wo_off + 32 * m + nca_rho(i) + 4 * h];
                    } else if (q == tail) {
                        f = prm[y.bo_off];
                    }
                    v = __builtin_bit_cast(unsigned, f);
                }
            }
            if (l.imgT_bytes && byte >= l.imgT_off && byte < l.imgT_off + l.imgT_bytes) {
                const uint32_t off = byte - l.imgT_off;
                const int KS = y.F / 16;
                const int col0 = skip ? y.K0 : 0;             // (a skip layer: the transposed image of its hidden part)
                float two[2];
                for (int e2 = 0; e2 < 2; ++e2) {
                    const uint32_t e = off / 2u + e2;
                    const int jj = e % 8, lane = (e / 8) % 64, ks = (e / 512) % KS, m = e / (512 * KS);
                    const int r = lane & 31, h = lane >> 5;
                    two[e2] = prm[l.w_off + nca_bf_kidx_hidden(ks, h, jj) * l.K + col0 + 32 * m + r];
                }
                v = pack2(two[0], two[1]);
            }
        }
        out[wd] = v;
    }
}
__global__ void nca_pack_bf16(NcaLayout y, const float* __restrict__ prm, unsigned* __restrict__ out) { pack_bf16_body(y, prm, out); }
// both nets of a composite render in one launch: blockIdx.y = net
struct NcaPack2ArgsB { NcaLayout y[2]; const float* prm[2]; void* out[2]; };
__global__ void nca_pack2_bf16(NcaPack2ArgsB a) { pack_bf16_body(a.y[blockIdx.y], a.prm[blockIdx.y], static_cast<unsigned*>(a.out[blockIdx.y])); }

// ------------------------------------------------------------------------------------------
// shared device helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float act_fwd_b(int act, float x) {
    if (act == NCA_ACT_SIGMOID) return 1.f / (1.f + expf(-x));
    float sp = x > 20.f ? x : log1pf(expf(x));
    if (act == NCA_ACT_CLAMP) sp = fminf(fmaxf(sp, 0.f), 1.f);
    return sp;
}
__device__ __forceinline__ float act_bwd_b(int act, float x) {
    if (act == NCA_ACT_SIGMOID) { float s = 1.f / (1.f + expf(-x)); return s * (1.f - s); }
    float d;
    if (x > 20.f) d = 1.f; else { float z = expf(x); d = z / (z + 1.f); }
    if (act == NCA_ACT_CLAMP) {
        float sp = x > 20.f ? x : log1pf(expf(x));
        if (!(sp > 0.f && sp < 1.f)) d = 0.f;
    }
    return d;
}
__device__ __forceinline__ float half_sum_b(float v) {
    v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ double wave_sum_b(double v) {
    v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}

template <int F>
struct BfCfg {
    static constexpr int MT = F / 32;
    static constexpr int KS = F / 16;
    static constexpr int KS0 = NCA_BF_K0SLOTS / 16;
    static constexpr int KSMAX = KS > KS0 ? KS : KS0;
    static constexpr int IMG_MAX = KSMAX * MT * 1024 + 2 * (2 * MT * 16 * 4) + 16;
    static constexpr int BUF_BYTES = (IMG_MAX + 1023) & ~1023;
};

#define NCA_CONST_WIN 16
#define NCA_CONST_FOUR 48
#define NCA_CONST_LAT 256       // the bf16 path takes at most 16 phases x 16 latent dimensions (nca_build_layout_bf16)
#define NCA_CONST_NET_FLOATS (NCA_CONST_WIN + NCA_CONST_FOUR + NCA_CONST_LAT)
// LDS constant area: encoding constants of both nets (modes that encode) or the waves' ReLU-mask slots (backward from a store)
__host__ __device__ constexpr int bf_const_bytes(int kmode) {
    return kmode == NCA_KM_BWD_NR ? NCA_WAVES * 2048 : 2 * NCA_CONST_NET_FLOATS * 4;
}
// mode 5: [Wo | bo] of both nets in accumulator order, f32, behind the output-layer partials
__host__ __device__ constexpr int bf_wo_floats(int F) { return 2 * (F / 32) * 16 + 16; }

__device__ __forceinline__ void stage_issue_b(const NcaStage& st, char* dst, int wave, int lane) {
    const int npiece = (int)(st.bytes >> 10);
    const char* src = reinterpret_cast<const char*>(st.ptr) + lane * 16;
    for (int c = wave; c < npiece; c += NCA_WAVES) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)c * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + c * 1024), 16, 0, 0);
    }
}
__device__ __forceinline__ void stage_publish_b() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// Backward stages issue exactly NST scratch stores AFTER the weight DMA of the stage.  VMEM operations
// retire in issue order, so waiting until NST remain outstanding waits for the DMA (and everything
// older) but not for those stores.  The raw barrier avoids the vmcnt(0) a __syncthreads() would add.
template <int NST>
__device__ __forceinline__ void stage_publish_counted(bool stores_issued) {
    if (stores_issued) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the wave's global stores / loads
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// acc0/acc1 += A(row tile m of the layer image) * B, with the A fragments of the whole layer image streamed through a
// ring of registers:
// the image is [row tile][k-step][lane][16 B], i.e. contiguous in the global step g = m * NKS + ks, so the read
// of step g + PF is issued while step g multiplies -- also across row-tile boundaries, where the ReLU/pack
// epilogue then hides the LDS latency of the next row tile's first fragments.
#ifndef NCA_BF_PF
#define NCA_BF_PF 3
#endif
constexpr int NCA_BF_RING = 4;
// How the two waves of a SIMD share it, and why the row-tile loops below are plain (MFMA block, then epilogue, one row tile at a
// time): every re-ordering measured in rounds 2 and 3 -- software pipelining of the row tiles, the deferred epilogue in six pieces
// behind the next row tile's MFMA pairs, s_setprio around the MFMAs, a forced anti-phase of the two waves, the epilogue of one column
// tile behind the MFMAs of the other -- left the STEP where it was (the chip runs these kernels at its power cap; the
// same instructions and bytes cost the same time in any order).  What is left is fewer instructions and fewer bytes.
static_assert(NCA_BF_PF >= 1 && NCA_BF_PF < NCA_BF_RING, "prefetch distance must fit the ring");
// RING registers, prefetch distance RING - 1 (a ring of 4 in every mode)
template <int NKS, int MTOT, int RING>
__device__ __forceinline__ void ring_prime(const char* imgl, u32x4 (&A)[RING]) {
#pragma unroll
    for (int g = 0; g < RING - 1; ++g)
        if (g < MTOT * NKS) A[g % RING] = *reinterpret_cast<const u32x4*>(imgl + g * 1024);
}
template <int NKS, int MTOT, int NB, int RING>
__device__ __forceinline__ void mma_rowtile_ring(const char* imgl, int m, u32x4 (&A)[RING], const u32x4 (&B)[2][NB],
                                                 f32x16& acc0, f32x16& acc1) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int g = m * NKS + ks, nx = g + RING - 1;
        if (nx < MTOT * NKS) A[nx % RING] = *reinterpret_cast<const u32x4*>(imgl + nx * 1024);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(A[g % RING]), frag(B[0][ks]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(A[g % RING]), frag(B[1][ks]), acc1, 0, 0, 0);
        // keep every LDS read and MFMA inside its own step (the scheduler otherwise sinks the reads next to their
        // use and the prefetch distance is lost); vector/scalar ALU and global memory instructions may still move
        __builtin_amdgcn_sched_barrier(0x2 | 0x4 | 0x10 | 0x400);
    }
}

// acc0/acc1 += A(row tile m of an image of NKS k-steps) * B without the ring: the two images of a skip layer (SKIP instantiations)
template <int NKS, int NB>
__device__ __forceinline__ void mma_rowtile_plain(const char* imgl, int m, const u32x4 (&B)[2][NB], f32x16& acc0, f32x16& acc1) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const u32x4 A = *reinterpret_cast<const u32x4*>(imgl + (m * NKS + ks) * 1024);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(A), frag(B[0][ks]), acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(A), frag(B[1][ks]), acc1, 0, 0, 0);
    }
}

// identity fragment of k-step s for the transposing product Z = X^T * E: element j of lane (c,h) is
// E[k = 16s + 8h + j][column c of the 32-wide output tile tcol] = 1 iff 32*tcol + c == that k.
__device__ __forceinline__ u32x4 ident_frag(int kbase, int lc) {
    // kbase = 16s + 8h - 32*tcol ; element j is 1 when kbase + j == lc
    u32x4 r;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned lo = (kbase + 2 * w == lc) ? 0x3f80u : 0u;
        const unsigned hi = (kbase + 2 * w + 1 == lc) ? 0x3f800000u : 0u;
        r[w] = lo | hi;
    }
    return r;
}

// transpose a [32 samples][32*NT features] sample-major bf16 block (already loaded as A fragments:
// lane (r = sample, h), k-step s: features 16s+8h..+7) into NT accumulator tiles with rows = samples,
// column = feature on the lane, packed as two k-steps (16 samples each) of MFMA operands.
template <int NT>
__device__ __forceinline__ void transpose_block(const u32x4 (&X)[2 * NT], int lc, int lh, u32x4 (&T)[NT][2], float (*colsum)[NT]) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        // features of tile t live in k-steps 2t and 2t+1 of X
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const u32x4 E = ident_frag(16 * s + 8 * lh, lc);
            z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(X[2 * t + s]), frag(E), z, 0, 0, 0);
        }
        if (colsum) {
            float cs = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) cs += z[i];
            (*colsum)[t] += cs;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            T[t][0][u] = pack2(z[2 * u], z[2 * u + 1]);
            T[t][1][u] = pack2(z[8 + 2 * u], z[8 + 2 * u + 1]);
        }
    }
}

// The same for an 8-bit block ([row tile][lane][16 B], byte i = accumulator register i: the low / high 8 bytes are the two
// k-steps of the tile): the transposing products run on the fp8 matrix path against an 8-bit identity (x 1 is exact in any
// format) and come out as f32, which (MUL) is multiplied by `mul` -- the wave tile's inverse scale for e5m2 output
// gradients -- before the column sums and the bf16 packing.  E5M2: e5m2 bytes (else e4m3).
template <bool E5M2>
__device__ __forceinline__ long ident8(int kbase, int lc) {
    const int j = lc - kbase;                       // byte j is 1.0 when kbase + j == lc
    const unsigned long one = E5M2 ? 0x3cul : 0x38ul;
    return (j >= 0 && j < 8) ? (long)(one << (8 * j)) : 0l;
}
template <int NT, int NX, bool E5M2, bool MUL>
__device__ __forceinline__ void transpose_block8(const u32x4 (&X)[NX], int lc, int lh, float mul, u32x4 (&T)[NT][2], float (*colsum)[NT]) {
    static_assert(NX >= NT, "one 16-byte fragment per 32-feature tile");
    const long E0 = ident8<E5M2>(8 * lh, lc), E1 = ident8<E5M2>(16 + 8 * lh, lc);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x16 z;
#pragma unroll
        for (int i = 0; i < 16; ++i) z[i] = 0.f;
        const long lo = (long)(((unsigned long)X[t][1] << 32) | X[t][0]), hi = (long)(((unsigned long)X[t][3] << 32) | X[t][2]);
        if (E5M2) {
            z = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(lo, E0, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8(hi, E1, z, 0, 0, 0);
        } else {
            z = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(lo, E0, z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(hi, E1, z, 0, 0, 0);
        }
        if (MUL) {
#pragma unroll
            for (int i = 0; i < 16; ++i) z[i] *= mul;
        }
        if (colsum) {
            float cs = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) cs += z[i];
            (*colsum)[t] += cs;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            T[t][0][u] = pack2(z[2 * u], z[2 * u + 1]);
            T[t][1][u] = pack2(z[8 + 2 * u], z[8 + 2 * u + 1]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// fused forward / backward-dgrad kernel
// ------------------------------------------------------------------------------------------
// S8: 8-bit staging -- the storing forward (mode 2, always S8) writes the input block and the hidden blocks as e4m3, the backward from the store (mode 5)
// writes D_0..D_{NL-2} as e5m2 scaled by a power of two per 64-sample tile (nca_layout.hpp)
// RES (modes 0, 2, 5 with ONE net per launch): every weight image of the launch is RESIDENT in LDS (NcaStage::lds_off, loaded once
// per workgroup); the tile loop then has no weight DMA, no counted waits and no workgroup barrier -- the eight waves drift apart
// and fill each other's epilogue and store slots.  The streaming variant double-buffers one image per stage behind a barrier.
#ifndef NCA_BF_MINBLOCKS
#define NCA_BF_MINBLOCKS 2     // (1 with NCA_WAVES=4: up to 512 registers per wave)
#endif
// SKIP (streaming kernels only): some net of the launch has a skip layer (CPPN with num_late_layers > 0, model/CPPN.py:53-58, 102-106) -- that
// layer reads cat[encoded input, h]: the encoding is formed again (the registers cannot hold it across the early layers), its two images
// (encoded part, hidden part) occupy BOTH halves of the LDS double buffer while the layer computes, and the next layer's image arrives after
// it.  The nets the reference ships have none; their instantiations (SKIP = false) hold none of this.
template <int F, int MODE, bool S8, bool RES, bool SKIP = false>
__global__ __launch_bounds__(NCA_NT, NCA_BF_MINBLOCKS) void nca_fused_bf16(const NcaFusedArgs a) {
    static_assert(!(SKIP && RES), "a net with a skip layer runs the streaming kernels");
    constexpr bool NR = MODE == NCA_KM_BWD_NR;                                // from a store with fp8 staging: nothing is recomputed
    constexpr int RINGK = NCA_BF_RING;                                        // A-fragment ring of the layer contractions
    static_assert(MODE == NCA_KM_FWD || MODE == NCA_KM_BWD || MODE == NCA_KM_FWD_STORE || NR, "bf16 modes: 0 forward, 1 recompute backward, 2 storing forward, 5 backward from the store");
    // (MODE 2 with !S8: the BF16 store of round 5 -- layer inputs as bf16 fragments, masks of every layer, raw outputs: what mode 5 and
    // the bf16 weight-gradient jobs need when nothing may be staged in 8 bits, NCA_STORE_BF16)
    constexpr bool BWD = MODE == NCA_KM_BWD || NR;                            // output-layer gradients + dgrad sweep
    constexpr bool STORE = MODE == NCA_KM_BWD || MODE == NCA_KM_FWD_STORE;    // writes the input block and the layer inputs
    constexpr bool RECOMP = !NR;                                              // runs the forward layers
    constexpr bool FSTORE = MODE == NCA_KM_FWD_STORE;
    constexpr int MT = BfCfg<F>::MT, KS = BfCfg<F>::KS, KS0 = BfCfg<F>::KS0, KSMAX = BfCfg<F>::KSMAX;
    constexpr int BUF = BfCfg<F>::BUF_BYTES;
    constexpr int HB = 32 * F * 2;                          // bytes of one hidden scratch block (32 samples x F)
    constexpr int EB = (FSTORE && S8) ? 32 * 128 : 32 * NCA_BF_ENCROWS * 2;      // bytes of the input block (fp8 staging: e4m3, nca_bf_ebytes)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(!RES || MODE == NCA_KM_FWD || MODE == NCA_KM_FWD_STORE || NR, "resident images: forward and backward from the store");
    const int wbytes = RES ? a.res_bytes : 2 * BUF;             // weight images: all of them / the double buffer
    constexpr int CONSTB = bf_const_bytes(MODE);
    float* cst = reinterpret_cast<float*>(smem + wbytes);
    float* osum = reinterpret_cast<float*>(smem + wbytes + CONSTB);
    // ReLU masks of the recomputed layers: [wave][layer][lane][16 B] (2 bits per packed bf16 pair)
    char* const maskbase = smem + wbytes + CONSTB + NCA_WAVES * 2 * (F + 1) * 4;
    float* const wo_lds = reinterpret_cast<float*>(maskbase);          // mode 5 (no other use of that area there)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 31, lh = lane >> 5;
    if (S8) s8_mode();

    if (RES) {
        // the images sit back to back at 16-byte granularity while the DMA moves whole 1 KiB pieces: the padding behind an image
        // lands on the start of the next one (behind the last: on the constant area), so they go in one after the other
        for (int i = 0; i < a.nstages; ++i) {
            stage_issue_b(a.stage[i], smem + a.stage[i].lds_off, wave, lane);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    for (int net = 0; RECOMP && net < a.nnets; ++net) {
        const NcaNetArgs& na = a.net[net];
        float* c = cst + net * NCA_CONST_NET_FLOATS;
        if (na.win) for (int i = tid; i < na.lay.L; i += NCA_NT) c[i] = na.win[i];
        if (na.four) for (int i = tid; i < 3 * na.lay.L; i += NCA_NT) c[NCA_CONST_WIN + i] = na.four[i];
        if (na.lat) for (int i = tid; i < na.lay.P * na.lay.T; i += NCA_NT) c[NCA_CONST_WIN + NCA_CONST_FOUR + i] = na.lat[i];
    }
    if (BWD) for (int i = tid; i < NCA_WAVES * 2 * (F + 1); i += NCA_NT) osum[i] = 0.f;
    if (RES && tid == 0) *reinterpret_cast<int*>(smem + a.ctr_off) = 0;
    if (NR)
        for (int net = 0; net < a.nnets; ++net)
            for (int i = tid; i < 2 * MT * 16 + 1; i += NCA_NT) wo_lds[net * bf_wo_floats(F) + i] = a.net[net].wo_src[i];
    __syncthreads();
    if (!RES) {
        stage_issue_b(a.stage[0], smem, wave, lane);
        stage_publish_b();
    }
    int cur = 0, si = 0;

    const int64_t ngroups = (a.ntiles + NCA_WAVES - 1) / NCA_WAVES;
    // This workgroup's tiles: groups blockIdx.x, blockIdx.x + gridDim.x, ... of 8.  Streaming kernels: wave w takes tile w of every
    // group (the waves meet at every stage's barrier anyway).  Resident images: no barrier couples the waves, and the second wave
    // of a SIMD gets the issue slots the first leaves -- it runs ~25 % slower -- so each wave CLAIMS its next tile from a counter
    // in LDS.  Nothing that is summed across tiles stays in a wave, so the results do not depend on who ran which tile.
    constexpr bool DYN = RES;
    const int64_t my_groups = (int64_t)blockIdx.x < ngroups ? (ngroups - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    for (int64_t it = 0;; ++it) {
        int64_t grp;
        int wslot = wave;
        if (DYN) {
            int k = 0;
            if (lane == 0) k = __hip_atomic_fetch_add(static_cast<int*>(__builtin_assume_aligned(smem + a.ctr_off, 16)), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            k = __builtin_amdgcn_readfirstlane(k);
            if (k >= my_groups * NCA_WAVES) break;
            grp = blockIdx.x + (int64_t)(k / NCA_WAVES) * gridDim.x;
            wslot = k % NCA_WAVES;
        } else {
            grp = blockIdx.x + it * gridDim.x;
            if (grp >= ngroups) break;
        }
        const int64_t tile = grp * NCA_WAVES + wslot;          // 64-sample tile
        const bool tvalid = tile < a.ntiles;
        const int64_t tl = tvalid ? tile : a.ntiles - 1;

        // ---- per-lane sample (lane = sample) --------------------------------------------------------
        int64_t ray = 0, n = 0;
        int smp = 0;
        bool valid;
        float p[3];
        if (a.mode == NCA_MODE_RAYS) {
            ray = a.ray0 + tl / a.nchunk;
            smp = (int)(tl % a.nchunk) * 64 + lane;
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
            n = a.n0 + tl * 64 + lane;
            valid = tvalid && n < a.N;
            if (n >= a.N) n = a.N - 1;
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = RECOMP ? a.pts[n * 3 + c] : 0.f;
        }
        // second launch of a split render: the static net's sigma, written by the first launch
        float ss_other = 0.f;
        if (!BWD && a.split == 2 && valid) ss_other = a.sig_s[n];
        int ph = 0;
        if (RECOMP && a.phase) ph = a.mode == NCA_MODE_RAYS ? a.phase[ray * a.ps_r + (int64_t)smp * a.ps_s] : a.phase[n];

        // Scratch: two 32-sample tiles per wave, fragment-major blocks (see nca_bf_tile_bytes).  The input block and the
        // layer inputs live in the H region (indexed by the tile's position in the whole batch when a stored forward
        // wrote it), the output gradients in the D region of this launch; the recompute backward uses one for both.
        // A wave without a tile (the batch's last group) works on the last tile once more.  Where a region is only WRITTEN -- the
        // storing forward's store, the D region of a backward from the store -- it writes that copy to its own slot: both
        // regions end in slack slots up to the next multiple of the 8 waves, so the hot loops store without a predicate.
        const int64_t tg = (FSTORE ? tile : tl) + a.tile0;
        char* const t32 = (BWD || STORE) ? reinterpret_cast<char*>(a.scratch) + (tg * 2) * a.rows_total : nullptr;   // rows_total = bytes per 32-sample tile
        char* const d32 = BWD ? a.dscratch + ((NR ? tile : tl) * 2) * a.d_total : nullptr;

        float raw[2] = {0.f, 0.f};

#pragma unroll
        for (int net = 0; net < 2; ++net) {
            if (net >= a.nnets) break;
            const NcaNetArgs& na = a.net[net];
            const NcaLayout& y = na.lay;
            const int phc = ph < 0 ? 0 : (ph >= y.P ? y.P - 1 : ph);
            const float* cnet = cst + net * NCA_CONST_NET_FLOATS;
            char* const nb = (BWD || STORE) ? t32 + na.row0 : nullptr;   // this net's input/H blocks inside a tile (byte offset)
            char* const db = BWD ? d32 + na.drow0 : nullptr;             // this net's D blocks
            const bool lds_mask = MODE == NCA_KM_BWD && a.mask_layers >= y.NL - 1;
            char* const mwave = maskbase + (wave * a.mask_layers) * 1024 + lane * 16;
            // stored forward: ReLU masks [wave tile][net][layer][lane][16 B]
            char* const mglob = (FSTORE || NR) ? a.mstore + ((tg * 2 + net + a.net_base) * a.mstore_layers) * 1024 + lane * 16 : nullptr;
            // Backward from the store: the ReLU masks of layer L travel HBM -> LDS by LDS-DMA into slot L & 1 of the wave's two
            // 1 KiB slots (the constant area is idle here: nothing is encoded), one layer ahead of their use and right behind the
            // weight DMA of a stage, so the stage's counted vmcnt covers them.  (As plain loads into registers they made the
            // compiler wait for vmcnt(0) at the top of every layer -- loads and stores retire in order, so that waited for all
            // of the previous layer's D stores: no store ever overlapped the next layer's MFMAs.)
            char* const mslot = reinterpret_cast<char*>(cst) + wave * 2048;
            auto mask_dma = [&](int L) __attribute__((always_inline)) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(mglob + L * 1024),
                                                 (__attribute__((address_space(3))) void*)(mslot + (L & 1) * 1024), 16, 0, 0);
            };

            // ================= encoding, lane = sample ===================================================
            // (a lambda: layer 0 calls it once per tile and net; a skip layer -- SKIP instantiations -- calls it again for its encoded part)
            u32x4 B[2][KSMAX];
            auto encode = [&](u32x4 (&B)[2][KSMAX], const bool allow_store) __attribute__((always_inline)) {
                float fe[NCA_BF_K0SLOTS];
#pragma unroll
                for (int i = 0; i < NCA_BF_K0SLOTS; ++i) fe[i] = 0.f;
                // The counts the unrolled slot loops below compare against, re-read from their registers per tile: as loop invariants the
                // compiler forms all ~50 comparison masks ONCE, ahead of the tile loop, parks them in lanes of a vector register (they
                // do not fit the scalar file) and fetches each with a v_readlane per tile -- a vector issue slot apiece in a kernel that
                // is bound by those (tools/isa_spills.py); compared per tile they are scalar-ALU instructions and live in no register.
                int encL = y.L, encT = y.T;
                asm volatile("" : "+s"(encL), "+s"(encT));
                if (y.enc_mode == NCA_ENC_FOURIER) {
                    // [sin(2 pi x g_i), cos(2 pi x g_i)] interleaved into slots 2i, 2i+1 (model/CPPN.py:115-118)
#pragma unroll
                    for (int i = 0; i < NCA_BF_K0SLOTS / 2; ++i) {
                        if (i < 3 * encL) {
                            const int c = i % 3;
                            const float pc = c == 0 ? p[0] : (c == 1 ? p[1] : p[2]);
                            const float v = __fmul_rn(__fmul_rn(6.283185482025146484375f, pc), cnet[NCA_CONST_WIN + i]);
                            sincosf(v, &fe[2 * i], &fe[2 * i + 1]);
                        }
                    }
                } else {
                    fe[0] = p[0]; fe[1] = p[1]; fe[2] = p[2];
                }
                if (y.enc_mode == NCA_ENC_BANDS) {
                    float sn[3], cs[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) sincosf(p[c], &sn[c], &cs[c]);
#pragma unroll
                    for (int k = 0; k < 15; ++k) {
                        if (k < encL) {
                            const float w = cnet[k];
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                fe[3 + 6 * k + c] = w * sn[c];
                                fe[6 + 6 * k + c] = w * cs[c];
                                const float s2 = 2.f * sn[c] * cs[c];
                                const float c2 = 1.f - 2.f * sn[c] * sn[c];
                                sn[c] = s2; cs[c] = c2;
                            }
                        }
                    }
                }
                if (encT > 0) {
                    const float* lat = cnet + NCA_CONST_WIN + NCA_CONST_FOUR + phc * encT;
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (t < encT) fe[NCA_BF_LAT_SLOT + t] = lat[t];
                }
                unsigned fp[NCA_BF_K0SLOTS / 2];
#pragma unroll
                for (int w = 0; w < NCA_BF_K0SLOTS / 2; ++w) fp[w] = pack2(fe[2 * w], fe[2 * w + 1]);

                // per k-step: words 8s..8s+3 = k-half 0, 8s+4..8s+7 = k-half 1 of MY sample.  One
                // half-swap per word yields the operand of tile 0 (samples 0..31) and tile 1 (32..63).
#pragma unroll
                for (int s = 0; s < KS0; ++s)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        auto r = __builtin_amdgcn_permlane32_swap(fp[8 * s + w], fp[8 * s + 4 + w], false, false);
                        B[0][s][w] = r[0];
                        B[1][s][w] = r[1];
                    }
                const bool store_in = allow_store && STORE && (FSTORE || tvalid) && !(a.share_enc && net + a.net_base == 0);
                if (store_in) {
                    // input block of both column tiles, fragment-major [k-step][lane][16 B]: the layer-0
                    // operands as they sit in registers, then one k-step of one-hot phase slots
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        char* blk = nb + c * a.rows_total + lane * 16;
                        const int pc = __shfl(phc, 32 * c + lr);          // phase of sample 32c + r
                        u32x4 hot;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const unsigned lo = (y.P > 0 && pc == 8 * lh + 2 * w) ? 0x3f80u : 0u;
                            const unsigned hi = (y.P > 0 && pc == 8 * lh + 2 * w + 1) ? 0x3f800000u : 0u;
                            hot[w] = lo | hi;
                        }
                        if (FSTORE && S8) {
                            // e4m3 (x 2^NCA_H8_LOG2), two k-steps per 16 bytes: [32-slot tile][lane][16 B] (nca_bf_ebytes)
                            static_assert(KS0 == 6, "input block: six k-steps of layer-0 slots + one of one-hot slots");
                            constexpr float DIV = 1.f / (float)(1 << NCA_H8_LOG2);
                            const u32x4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const u32x4 lo = t < 3 ? B[c][2 * t] : hot, hi = t < 3 ? B[c][2 * t + 1] : zero;
                                const u32x4 q = {cvt4_e4m3_pk(lo[0], lo[1], DIV), cvt4_e4m3_pk(lo[2], lo[3], DIV), cvt4_e4m3_pk(hi[0], hi[1], DIV), cvt4_e4m3_pk(hi[2], hi[3], DIV)};
                                store_nt(blk + t * 1024, q);
                            }
                        } else {
#pragma unroll
                            for (int s = 0; s < KS0; ++s) store_nt(blk + s * 1024, B[c][s]);
                            store_nt(blk + KS0 * 1024, hot);
                        }
                    }
                }
            };
            if (RECOMP) encode(B, true);

            // S8: inverse of the power of two by which this tile's output gradients are scaled on their way to e5m2 (the chain
            // itself stays unscaled bf16: the scaling is part of the conversion)
            float inv_s = 1.f;
            // ---------- gradient wrt the raw output, output-layer parameter gradients, D_{NL-1}: expects the last layer's
            // packed output in B and its raw output in raw[net]; leaves D_{NL-1} in B.  `wo` = [Wo | bo] in LDS
            auto last_layer_grads = [&](const float* wo) __attribute__((always_inline)) {
                    // ---------- gradient wrt the raw output, output-layer parameter gradients, D_{NL-1} ----------
                    float g;
                    if (a.mode == NCA_MODE_RAYS && !a.g_raw) {
                        const float* gs = net + a.net_base == 0 ? a.g_sig_s : a.g_sig_d;
                        const double gsig = gs ? (double)gs[n] : 0.0;
                        const double gp = a.g_pix[ray] * a.dists[smp];
                        const double dsig = a.single ? (gsig - gp * (double)a.scale) : (gsig - gp) * (double)a.scale;
                        g = (float)dsig * act_bwd_b(a.act, raw[net]);
                    } else {
                        g = a.g_raw[n];
                    }
                    if (!valid) g = 0.f;
                    if (S8 && NR) {
                        // the dgrad chain is linear in g, so the largest |g| of the tile sizes every D_l of the tile: 2^e <= max < 2^(e+1)
                        // is scaled to 2^NCA_D8_LOG2 (exponent arithmetic; an all-zero tile gets the smallest inverse scale)
                        float am = fabsf(g);
                        am = fmaxf(am, __shfl_xor(am, 32)); am = fmaxf(am, __shfl_xor(am, 16)); am = fmaxf(am, __shfl_xor(am, 8));
                        am = fmaxf(am, __shfl_xor(am, 4)); am = fmaxf(am, __shfl_xor(am, 2)); am = fmaxf(am, __shfl_xor(am, 1));
                        int eb = (int)(__float_as_uint(am) >> 23) - NCA_D8_LOG2;
                        eb = eb < 1 ? 1 : (eb > 254 ? 254 : eb);
                        inv_s = __uint_as_float((unsigned)eb << 23);
                        if (lane == 0 && tvalid) reinterpret_cast<float*>(d32 + a.dscale_off)[net + a.net_base] = inv_s;
                    }
                    // g of tile c for BOTH lane halves: [g.lower|g.lower] and [g.upper|g.upper]
                    const float gc[2] = {__shfl(g, lr), __shfl(g, lr + 32)};
                    float* orow = osum + (wave * 2 + net) * (F + 1);
                    char* const dblk = db + nca_bf_doff(y, y.NL - 1, S8 && NR);
                    u32x4 Bn[2][2 * MT];
                    if (NR) {
                        // D_{NL-1} = relu'(H_{NL-1}) (Wo x g) feeds the sweep; the block that goes to the weight-gradient kernel is
                        // relu'(H_{NL-1}) g WITHOUT Wo (nca_layout.hpp: the reduce kernel puts Wo back and gets dWo from the same sums).
                        // The layer's mask bits arrived by DMA (requested at the top of the net; the loads of raw / g since then
                        // have returned, and the queue is in order)
                        asm volatile("s_waitcnt vmcnt(1)" ::: "memory");          // (1: the scale record of lane 0)
                        const u32x4 mv = *reinterpret_cast<const u32x4*>(mslot + ((y.NL - 1) & 1) * 1024 + lane * 16);
                        if (y.NL >= 2) mask_dma(y.NL - 2);
                        const unsigned gpk[2] = {pack2_pk(gc[0], gc[0]), pack2_pk(gc[1], gc[1])};        // bf16(g) in both halves
#pragma unroll
                        for (int m = 0; m < MT; ++m) {
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                const unsigned fld = mv[2 * c + (m >> 1)] >> (8 * (m & 1));
#pragma unroll
                                for (int s2 = 0; s2 < 2; ++s2) {
                                    u32x4 dw, ds;
#pragma unroll
                                    for (int u = 0; u < 4; ++u) {
                                        const float a0 = wo[(lh * MT + m) * 16 + 8 * s2 + 2 * u] * gc[c];
                                        const float a1 = wo[(lh * MT + m) * 16 + 8 * s2 + 2 * u + 1] * gc[c];
                                        const unsigned on = (fld >> (4 * s2 + u)) & 0x00010001u;
                                        dw[u] = keep_pk(pack2_pk(a0, a1), on);
                                        if (!S8) ds[u] = keep_pk(gpk[c], on);
                                    }
                                    Bn[c][2 * m + s2] = dw;
                                    if (!S8 && !a.expand_last) store_nt(dblk + c * a.d_total + lane * 16 + (2 * m + s2) * 1024, ds);
                                }
                            }
                        }
                        if (S8) {
                            // e5m2: the block itself is not written -- the weight-gradient kernel rebuilds it from the forward's mask
                            // bits and the sample's byte (wgrad_job_mx, EXPAND), left here as byte * 0x00010001 where the block of
                            // the sample's 32-sample tile would start: u32[32], half lh of the wave writes tile lh
                            const unsigned gb = lh ? gpk[1] : gpk[0];
                            *reinterpret_cast<unsigned*>(dblk + lh * a.d_total + lr * 4) = cvt4_e5m2_pk(gb, gb, inv_s) & 0x00ff00ffu;
                        } else if (a.expand_last) {
                            // bf16 output gradients with the last block rebuilt likewise (the bf16 store): bf16(g) in both halves of a word
                            // per sample, where the block of the sample's 32-sample tile would start (wgrad_job, EXPAND)
                            *reinterpret_cast<unsigned*>(dblk + lh * a.d_total + lr * 4) = lh ? gpk[1] : gpk[0];
                        }
                    }
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        if (NR) break;
                        // H_last of this row tile back to f32 (both column tiles), in accumulator register order
                        float hv[2][16];
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                hv[c][2 * u] = bf_lo(B[c][2 * m][u]);          hv[c][2 * u + 1] = bf_hi(B[c][2 * m][u]);
                                hv[c][8 + 2 * u] = bf_lo(B[c][2 * m + 1][u]);  hv[c][8 + 2 * u + 1] = bf_hi(B[c][2 * m + 1][u]);
                            }
                        // dWo[f] = sum_n g H[f][n]: reduce-scatter the 16 values over the 32 lanes of each half
                        float v[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) v[i] = gc[0] * hv[0][i] + gc[1] * hv[1][i];
                        int cnt = 16;
#pragma unroll
                        for (int d = 16; d >= 1; d >>= 1) {
                            if (cnt >= 2) {
                                const int hn = cnt / 2;
                                const bool up = (lr & d) != 0;
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    if (i < hn) {
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
                        if ((lr & 1) == 0) orow[32 * m + nca_rho((lr >> 1) & 15) + 4 * lh] += v[0];
                        // D_{NL-1} = relu'(H_last) * Wo * g, packed as the dgrad B operand, stored sample-major
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            float dv[16];
#pragma unroll
                            for (int i = 0; i < 16; ++i) dv[i] = hv[c][i] > 0.f ? wo[(lh * MT + m) * 16 + i] * gc[c] : 0.f;
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                Bn[c][2 * m][u] = pack2(dv[2 * u], dv[2 * u + 1]);
                                Bn[c][2 * m + 1][u] = pack2(dv[8 + 2 * u], dv[8 + 2 * u + 1]);
                            }
                            if (tvalid) {
                                char* fp2 = dblk + c * a.d_total + lane * 16;
                                store_nt(fp2 + (2 * m) * 1024, Bn[c][2 * m]);
                                store_nt(fp2 + (2 * m + 1) * 1024, Bn[c][2 * m + 1]);
                            }
                        }
                    }
                    const float gsum = half_sum_b(g) ;            // sum over the 32 lanes of each half
                    const float gtot = gsum + __shfl_xor(gsum, 32);
                    if (NR) {          // the tile's sum goes to its record (the weight-gradient kernel adds the tiles up in tile order)
                        if (lane == 0 && tvalid) reinterpret_cast<float*>(d32 + a.dscale_off)[2 + net + a.net_base] = gtot;
                    } else if (lane == 0) orow[F] += gtot;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int k = 0; k < 2 * MT; ++k) B[c][k] = Bn[c][k];
            };

            // ================= layers (forward / recompute) ===============================================
            float part[2] = {0.f, 0.f};       // output-layer partial dot per column tile
            if (NR) {
                // nothing to recompute: the raw output and the masks of every layer are in the store
                mask_dma(y.NL - 1);
                raw[net] = a.rstore[((tg * 2 + net + a.net_base) * 64) + lane];
                last_layer_grads(wo_lds + net * bf_wo_floats(F));
            }
            for (int jj = NR ? y.NL : 0; jj < y.NL; ++jj) {
                const int nsi = (si + 1 == a.nstages) ? 0 : si + 1;
                // (a skip layer: the stage after its first image is its own second image -- both halves of the double buffer hold this layer)
                const bool skipl = SKIP && y.layer[jj].kind == NCA_IN_SKIP;
                if (!RES) stage_issue_b(a.stage[nsi], smem + (cur ^ 1) * BUF, wave, lane);
                // Resident images: the layer's image offset and k-step count follow from the width alone (build_stages lays the forward
                // images of the launch's one net back to back: layer 0, then the hidden-width layers; nca_build_layout_bf16).  Read from
                // the argument block instead -- y.layer[jj], a.stage[si] with run-time indices -- they were two dependent scalar loads
                // from memory at every layer boundary of every tile, ~1 500 cycles per boundary in the round-3 timeline
                constexpr int IMG0 = MT * KS0 * 1024 + 2 * MT * 16 * 4, IMGH = MT * KS * 1024 + 2 * MT * 16 * 4;
                const char* img = RES ? smem + (jj == 0 ? 0 : IMG0 + (jj - 1) * IMGH) : smem + cur * BUF;
                const int nks = (jj == 0 || skipl) ? KS0 : KS;
                const float* tail = reinterpret_cast<const float*>(img + MT * nks * 1024);
                // [Wo | bo]: behind the bias tail -- of a skip layer: behind the k-steps of its second image
                const char* img2 = smem + (cur ^ 1) * BUF;
                const float* wo_tail = skipl ? reinterpret_cast<const float*>(img2 + MT * KS * 1024) : tail + 2 * MT * 16;
                const bool last = jj == y.NL - 1;
                u32x4 Benc[2][KSMAX];
                if constexpr (SKIP) {
                    if (skipl) {
                        encode(Benc, false);           // the layer-0 operand once more (nothing is stored: the input block is in the store already)
                        stage_publish_b();             // the second image has landed
                    }
                }
                const bool h8 = S8 && FSTORE;                                     // fp8 staging: the layer outputs go to the store as e4m3 (the last layer: its mask only)
                char* const hblk = STORE ? nb + EB + nca_bf_hoff(y, jj, S8 && FSTORE) : nullptr;          // input block of layer jj+1
                u32x4 Bn[2][2 * MT];
                unsigned mw[2][2] = {{0u, 0u}, {0u, 0u}};     // mask words: [column tile][row-tile pair]
                const char* imgl = img + lane * 16;
                // One instantiation per (k-step count, last layer?, e4m3 output?): the row-tile loop has no branch in it, so the
                // scheduler may move the epilogue's vector ALU work (pack, ReLU, mask bits, conversion: ~100 instructions per row
                // tile) into the shadow of the MFMAs around it.
                // The storing forward writes unconditionally: a wave without a tile recomputes the batch's last tile and writes
                // it to the slack tile slots behind the store (store_plan rounds the tile count up to the 8 waves).
                const bool st_ok = FSTORE ? true : tvalid;
                auto rowtiles = [&](auto nks_c, auto last_c, auto h8_c, auto skl_c) __attribute__((always_inline)) {
                constexpr int NKS = decltype(nks_c)::value;
                constexpr bool LAST = decltype(last_c)::value, H8 = decltype(h8_c)::value, SKL = decltype(skl_c)::value;
                u32x4 A[RINGK];
                if constexpr (!SKL) ring_prime<NKS, MT, RINGK>(imgl, A);
                auto epilogue = [&](int m, f32x16& acc0, f32x16& acc1) __attribute__((always_inline)) {
                    if (LAST) {
                        // the output layer's dot product takes the f32 activations
#pragma unroll
                        for (int i = 0; i < 16; ++i) { acc0[i] = relu1(acc0[i]); acc1[i] = relu1(acc1[i]); }
                        const float* wo = wo_tail;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const float w = wo[(lh * MT + m) * 16 + i];
                            part[0] = fmaf(w, acc0[i], part[0]);
                            part[1] = fmaf(w, acc1[i], part[1]);
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            Bn[0][2 * m][u] = pack2(acc0[2 * u], acc0[2 * u + 1]);
                            Bn[0][2 * m + 1][u] = pack2(acc0[8 + 2 * u], acc0[8 + 2 * u + 1]);
                            Bn[1][2 * m][u] = pack2(acc1[2 * u], acc1[2 * u + 1]);
                            Bn[1][2 * m + 1][u] = pack2(acc1[8 + 2 * u], acc1[8 + 2 * u + 1]);
                        }
                    }
                    // words 0..7 of a (row tile, column tile): fragment 2m holds accumulator registers 0..7, fragment 2m + 1 registers 8..15
                    const bool want_mask = STORE && (!LAST || FSTORE);          // a forward store holds the masks of EVERY layer (mode 5 recomputes none)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        unsigned w[8];
                        if (LAST) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) { w[u] = Bn[c][2 * m][u]; w[4 + u] = Bn[c][2 * m + 1][u]; }
                        } else {
                            // round to bf16 (compiler-visible conversions: the wait states behind the MFMAs are the compiler's), then
                            // ReLU -- and the mask bits -- on the packed pairs
                            const f32x16& acc = c == 0 ? acc0 : acc1;
#pragma unroll
                            for (int u = 0; u < 4; ++u) { w[u] = pack2(acc[2 * u], acc[2 * u + 1]); w[4 + u] = pack2(acc[8 + 2 * u], acc[8 + 2 * u + 1]); }
                        }
                        // bit k / 16+k of a field: low / high bf16 of packed word k (k = 4*(fragment&1) + u) is > 0
                        if (want_mask) mw[c][m >> 1] |= (LAST ? mask8(w) : relu_mask8(w)) << (8 * (m & 1));
                        else if (!LAST) relu8(w);
                        if (!LAST) {
#pragma unroll
                            for (int u = 0; u < 4; ++u) { Bn[c][2 * m][u] = w[u]; Bn[c][2 * m + 1][u] = w[4 + u]; }
                        }
                    }
                    if (STORE && !LAST && H8) {
                        // e4m3 of the bf16 activations (x 2^NCA_H8_LOG2): byte i = register i, [row tile][lane][16 B]
                        constexpr float DIV = 1.f / (float)(1 << NCA_H8_LOG2);
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            u32x4 q;
#pragma unroll
                            for (int w = 0; w < 4; ++w) q[w] = cvt4_e4m3_pk(Bn[c][2 * m + (w >> 1)][2 * (w & 1)], Bn[c][2 * m + (w >> 1)][2 * (w & 1) + 1], DIV);
                            if (st_ok) store_nt(hblk + c * a.rows_total + lane * 16 + m * 1024, q);
                        }
                    } else if (STORE && !LAST) {
                        // the next layer's B-operand fragments exactly as they sit in registers: 1 KiB per
                        // wave instruction, [k-step][lane][16 B] (feature order inside a tile is the
                        // accumulator->operand order; the wgrad un-permutes when it writes dW)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            char* fp2 = hblk + c * a.rows_total + lane * 16;
                            if (st_ok) {
                                store_nt(fp2 + (2 * m) * 1024, Bn[c][2 * m]);
                                store_nt(fp2 + (2 * m + 1) * 1024, Bn[c][2 * m + 1]);
                            }
                        }
                    }
                };
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x16 acc0, acc1;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const float b = tail[(lh * MT + m) * 16 + i]; acc0[i] = b; acc1[i] = b; }
                    if constexpr (SKL) {       // cat[encoded input, h]: the encoded part from the first image, the hidden part from the second
                        mma_rowtile_plain<KS0, KSMAX>(imgl, m, Benc, acc0, acc1);
                        mma_rowtile_plain<KS, KSMAX>(img2 + lane * 16, m, B, acc0, acc1);
                    } else mma_rowtile_ring<NKS, MT, KSMAX, RINGK>(imgl, m, A, B, acc0, acc1);
                    epilogue(m, acc0, acc1);
                    __builtin_amdgcn_sched_barrier(0);       // one row tile at a time: without the fence the scheduler overlaps row tiles and spills
                }
                };
                {
                    const std::integral_constant<int, KS0> ks0c{};
                    const std::integral_constant<int, KS> ksc{};
                    const std::true_type yes{};
                    const std::false_type no{};
                    bool done_skip = false;
                    if constexpr (SKIP) {
                        if (skipl) {
                            done_skip = true;
                            if constexpr (S8 && FSTORE) { if (last) rowtiles(ks0c, yes, yes, yes); else rowtiles(ks0c, no, yes, yes); }
                            else { if (last) rowtiles(ks0c, yes, no, yes); else rowtiles(ks0c, no, no, yes); }
                        }
                    }
                    if (done_skip) {}
                    else if constexpr (S8 && FSTORE) {
                        if (jj == 0) { if (last) rowtiles(ks0c, yes, yes, no); else rowtiles(ks0c, no, yes, no); }
                        else { if (last) rowtiles(ksc, yes, yes, no); else rowtiles(ksc, no, yes, no); }
                    } else {
                        if (jj == 0) { if (last) rowtiles(ks0c, yes, no, no); else rowtiles(ks0c, no, no, no); }
                        else { if (last) rowtiles(ksc, yes, no, no); else rowtiles(ksc, no, no, no); }
                    }
                }
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int k = 0; k < 2 * MT; ++k) B[c][k] = Bn[c][k];
                if (lds_mask && !last) {
                    u32x4 mv = {mw[0][0], mw[0][1], mw[1][0], mw[1][1]};
                    *reinterpret_cast<u32x4*>(mwave + jj * 1024) = mv;
                }
                if (FSTORE) {
                    u32x4 mv = {mw[0][0], mw[0][1], mw[1][0], mw[1][1]};
                    store_nt(mglob + jj * 1024, mv);
                }

                if (last) {
                    const float* wo = wo_tail;
                    const float bo = wo[2 * MT * 16];
                    // column tile c of lane (r,h) is sample 32c + r: sum the two halves, then lane = sample picks its tile
                    const float r0 = part[0] + __shfl_xor(part[0], 32) + bo;
                    const float r1 = part[1] + __shfl_xor(part[1], 32) + bo;
                    raw[net] = lh ? r1 : r0;
                    // the raw output goes to the store as well ([wave tile][net][64] f32): the backward from a store recomputes nothing
                    if (FSTORE) a.rstore[((tg * 2 + net + a.net_base) * 64) + lane] = raw[net];
                }

                if (BWD && last) last_layer_grads(wo_tail);

                // at least 4 MT stores follow the weight DMA of every storing stage: H (plus a mask store in the storing
                // forward, which it then also waits for), or D_{NL-1} on the last layer of both backward modes; the last
                // layer of the storing forward stores nothing
                if (skipl) {
                    // both halves of the double buffer held this layer: once every wave is through with them, the next stage's image goes where
                    // the first image was (no prefetch across a skip layer)
                    lds_barrier();
                    const int nsi2 = (nsi + 1 == a.nstages) ? 0 : nsi + 1;
                    stage_issue_b(a.stage[nsi2], smem + cur * BUF, wave, lane);
                    stage_publish_b();
                    si = nsi2;
                    continue;
                }
                if (RES) {}                                             // nothing to publish, nothing to wait for
                else if (S8 && FSTORE && h8 && !last) stage_publish_counted<2 * MT>(true);   // 2 MT 8-bit stores (+ the mask store)
                else if ((STORE && !last) || (BWD && last)) stage_publish_counted<4 * MT>(FSTORE || tvalid);
                else stage_publish_b();
                cur ^= 1;
                si = nsi;
            }

            // ================= backward sweep (dgrad) =====================================================
            if (BWD) {
                // Two layers per trip: each reads one operand array and leaves the next layer's operand in the other -- copying 64
                // registers back per layer was a sixth of the sweep's vector instructions.
                static_assert(KSMAX >= 2 * MT, "a layer's output fragments fit the operand array");
                u32x4 B2[2][KSMAX];
                auto sweep_layer = [&](int jj, u32x4 (&Bin)[2][KSMAX], u32x4 (&Bout)[2][KSMAX]) __attribute__((always_inline)) {
                    const int nsi = (si + 1 == a.nstages) ? 0 : si + 1;
                    if (NR) {
                        // this layer's masks were requested before the D stores of the step before (the output layer's step or the
                        // previous iteration): at most those stores are younger
                        // (S8: the output layer's step stores one word per lane and lane 0's record, not a block)
                        constexpr int NSTM = S8 ? 2 * MT : 4 * MT;
                        // (the output layer's step may have stored a whole bf16 block -- the depth-gradient path -- or one word per lane: 2 covers both)
                        if (jj == y.NL - 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTM) : "memory");
                    }
                    if (!RES) stage_issue_b(a.stage[nsi], smem + (cur ^ 1) * BUF, wave, lane);
                    if (NR && jj >= 2) mask_dma(jj - 2);            // for the next iteration; this one's arrived under the previous stage
                    // (resident: the transposed images of layers NL-1 .. 1 back to back, one per sweep step -- see the forward)
                    const char* img = RES ? smem + si * (MT * KS * 1024) : smem + cur * BUF;
                    const char* const hblk = nb + EB + (jj - 1) * HB;                       // input of layer jj (mask)
                    char* const dblk = db + nca_bf_doff(y, jj - 1, S8 && NR);            // D_{jj-1}
                    const bool wr_d = tvalid;
                    u32x4 mv = {0u, 0u, 0u, 0u};
                    if (lds_mask) mv = *reinterpret_cast<const u32x4*>(mwave + (jj - 1) * 1024);
                    if (NR) mv = *reinterpret_cast<const u32x4*>(mslot + ((jj - 1) & 1) * 1024 + lane * 16);
                    const bool bits = lds_mask || NR;         // mask bits at hand (else: re-read the layer input)
                    u32x4 A[RINGK];
                    const char* imgl = img + lane * 16;
                    ring_prime<KS, MT, RINGK>(imgl, A);
                    // From a store the D blocks are written without a predicate (slack tile slots behind the D region take the copies
                    // of waves that have no tile)
                    const bool st_ok = NR ? true : wr_d;
                    auto epilogue = [&](int m, const f32x16& acc0, const f32x16& acc1) __attribute__((always_inline)) {
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const char* hp = hblk + c * a.rows_total + lane * 16;
                            char* dp = dblk + c * a.d_total + lane * 16;
                            u32x4 q8 = {0u, 0u, 0u, 0u};           // S8: the 16 masked values of this (row tile, column tile) as e5m2 bytes
#pragma unroll
                            for (int s2 = 0; s2 < 2; ++s2) {
                                // word u of fragment 2m+s2 holds accumulator registers 8 s2 + 2u, +1.  Round to bf16 first, then zero
                                // the masked halves of the packed pair with ONE and: the pair's two mask bits sit 16 apart, times
                                // 0xffff they are the pair's and-mask
                                u32x4 hw = {0u, 0u, 0u, 0u};
                                if (!bits) hw = *reinterpret_cast<const u32x4*>(hp + (2 * m + s2) * 1024);
                                const unsigned fld = mv[2 * c + (m >> 1)] >> (8 * (m & 1));
                                u32x4 dw;
#pragma unroll
                                for (int u = 0; u < 4; ++u) {
                                    const float a0 = c == 0 ? acc0[8 * s2 + 2 * u] : acc1[8 * s2 + 2 * u];
                                    const float a1 = c == 0 ? acc0[8 * s2 + 2 * u + 1] : acc1[8 * s2 + 2 * u + 1];
                                    const unsigned two = bits ? (fld >> (4 * s2 + u)) & 0x00010001u : pos_pk(relu_pk(hw[u]));
                                    dw[u] = keep_pk(pack2_pk(a0, a1), two);
                                }
                                Bout[c][2 * m + s2] = dw;
                                if (S8 && NR) {
                                    q8[2 * s2] = cvt4_e5m2_pk(dw[0], dw[1], inv_s);
                                    q8[2 * s2 + 1] = cvt4_e5m2_pk(dw[2], dw[3], inv_s);
                                } else if (st_ok) store_nt(dp + (2 * m + s2) * 1024, dw);
                            }
                            if (S8 && NR && st_ok) store_nt(dp + m * 1024, q8);
                        }
                    };
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        f32x16 acc0, acc1;
#pragma unroll
                        for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
                        mma_rowtile_ring<KS, MT, KSMAX, RINGK>(imgl, m, A, Bin, acc0, acc1);
                        epilogue(m, acc0, acc1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!RES) stage_publish_counted<(S8 && NR) ? 2 * MT : 4 * MT>(st_ok);              // D stores
                    cur ^= 1;
                    si = nsi;
                };
                for (int jj = y.NL - 1; jj >= 1; jj -= 2) {
                    sweep_layer(jj, B, B2);
                    if (jj - 1 >= 1) sweep_layer(jj - 1, B2, B);
                }
            }
        }  // nets

        // ================= epilogue (lane = sample) =========================================================
        if (!BWD) {
            if (a.mode == NCA_MODE_RAYS && !a.raw_only) {
                double term;
                if (a.split) {
                    // one net per launch (resident weights): launch 1 = static net, writes its scaled sigma and nothing else;
                    // launch 2 = dynamic net, composites with the sigma of launch 1
                    const float sg = __fmul_rn(act_fwd_b(a.act, raw[0]), a.scale);
                    if (a.split == 1) {
                        if (valid) a.sig_s[n] = sg;
                        continue;
                    }
                    if (valid) a.sig_d[n] = sg;
                    term = (double)__fadd_rn(ss_other, sg) * a.dists[smp];
                } else if (a.single) {
                    const float sa = act_fwd_b(a.act, raw[0]);
                    if (valid) a.sig_s[n] = sa;
                    term = ((double)sa * a.dists[smp]) * (double)a.scale;
                } else {
                    const float ss = __fmul_rn(act_fwd_b(a.act, raw[0]), a.scale);
                    const float sd = __fmul_rn(act_fwd_b(a.act, raw[1]), a.scale);
                    if (valid) { a.sig_s[n] = ss; a.sig_d[n] = sd; }
                    term = (double)__fadd_rn(ss, sd) * a.dists[smp];
                }
                if (!valid) term = 0.0;
                term = wave_sum_b(term);
                if (lane == 0 && tvalid) a.part[tile] = term;
            } else {
                if (valid) a.raw_out[n] = raw[0];
            }
        }
    }  // tile groups
    if (BWD) {
        __syncthreads();
        for (int i = tid; i < a.nnets * (F + 1); i += NCA_NT) {
            float s = 0.f;
            for (int w = 0; w < NCA_WAVES; ++w) s += osum[(w * 2) * (F + 1) + i];
            float* dst = a.oslab + (int64_t)blockIdx.x * 2 * (F + 1) + a.net_base * (F + 1) + i;
            *dst = a.accumulate ? *dst + s : s;
        }
    }
}

// ------------------------------------------------------------------------------------------
// wgrad, no LDS.  One wave = one (job, split): dW of the whole layer in accumulators.
// ------------------------------------------------------------------------------------------
// dW / db of one (job, split) -> its slab, natural [o][i] order
template <int F, int NTB, bool EXPAND = false>
__device__ __forceinline__ void wgrad_write(const f32x16 (&acc)[F / 32][NTB], const float (&bsum)[F / 32], const NcaWgradJob& job, float* slab, int accumulate, int lane) {
    constexpr int MT = F / 32;
    const int lc = lane & 31, lh = lane >> 5;
    // Hidden blocks hold features in accumulator->operand order: position c = 16s+8a+4b+e of a 32-wide
    // tile is feature 16s+8b+4a+e (bits 3 and 2 swapped).  The input block is in natural slot order.
    auto unperm = [](int c) { return (c & 0x13) | ((c & 8) >> 1) | ((c & 4) << 1); };
    // A D block rebuilt from mask bits (wgrad_job_mx, EXPAND) has its bytes in mask-bit order: position p = 16e+8h+4s+u is
    // feature 16s + 8(u>>1) + 4h + 2(u&1) + e.
    auto drow = [&](int p) { return EXPAND ? ((p & 4) << 2) | ((p & 2) << 2) | ((p & 8) >> 1) | ((p & 1) << 1) | (p >> 4) : unperm(p); };
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int c = 0; c < NTB; ++c) {
            const int slot = job.is_enc ? 32 * c + lc : 32 * c + unperm(lc);      // H column (layer-0 slot or hidden feature)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int o = 32 * m + drow(nca_rho(i) + 4 * lh);
                float* dst = nullptr;
                if (job.is_enc) {
                    if (slot < job.ncols_w) dst = slab + job.out_off + (int64_t)o * job.out_ld + (job.fourier_L ? ((slot & 1) ? 3 * job.fourier_L + (slot >> 1) : (slot >> 1)) : slot);
                    else if (slot >= NCA_BF_LAT_SLOT && slot < NCA_BF_LAT_SLOT + job.T) dst = slab + job.out_off + (int64_t)o * job.out_ld + job.ncols_w + (slot - NCA_BF_LAT_SLOT);
                    else if (slot >= NCA_BF_HOT_SLOT && slot < NCA_BF_HOT_SLOT + job.P) dst = slab + job.onehot_off + o * job.P + (slot - NCA_BF_HOT_SLOT);
                } else if (slot < job.ncols_w) {
                    dst = slab + job.out_off + (int64_t)o * job.out_ld + job.out_col0 + slot;      // (out_col0: the hidden part of a skip layer's weight)
                }
                if (dst) *dst = accumulate ? *dst + acc[m][c][i] : acc[m][c][i];
            }
        }
        if (job.bias_off >= 0) {
            // column sums of the transposed D tile = sum over samples; the two lane halves hold disjoint samples
            const float b = bsum[m] + __shfl_xor(bsum[m], 32);
            if (lh == 0) {
                float* dst = slab + job.bias_off + 32 * m + drow(lc);
                *dst = accumulate ? *dst + b : b;
            }
        }
    }
}

// NTB = 32-column tiles of the H block (F/32 for hidden inputs, 4 for the 112-wide input block).  D8: every D block of the launch
// is e5m2 scaled per wave tile (fp8 staging, nca_layout.hpp); H8: the job's H block is e4m3 x 2^NCA_H8_LOG2 (the input block and
// the last layer's input stay bf16).
//
// Operand tiles travel HBM -> LDS by LDS-DMA (global_load_lds, no registers) into a ring of NSLOT tile slots per wave, NSLOT - 1
// tiles ahead of the MFMAs: one wave per SIMD (its 256 dW accumulators fill the AGPR file) keeps 24-32 KiB in flight without
// holding them in VGPRs -- a CU needs ~50 KiB outstanding to pull its share of 6.5 TB/s at ~2 us of loaded latency, and one
// 8 KiB 8-bit tile ahead in registers gave 4.5 TB/s.  Fragments are [lane][16 B] in HBM and land in LDS as they are, so the
// reads back (ds_read_b128, lane * 16) are conflict-free.
template <int F, int NTB, bool D8, bool H8>
struct WgradRing {
    static constexpr int MT = F / 32;
    static constexpr int ND = D8 ? MT : 2 * MT, NH = H8 ? NTB : 2 * NTB, FR = ND + NH;      // 1 KiB fragments per tile
    static constexpr int NSLOT = FR <= 8 ? 4 : (FR <= 12 ? 3 : 2);
    static constexpr int BYTES = NSLOT * FR * 1024;
};
constexpr int NCA_WGRAD_LDS = 36 * 1024;

// EXPAND (bf16 output gradients, the last hidden layer's job of a backward from the bf16 store): the job's D block relu'(H_{NL-1}) g is
// NOT read -- per sample it holds one distinct value, bf16(g), at the features whose mask bit is set.  A tile brings the lane's two mask
// words (8 B of the forward's store) and the sample's word (bf16(g) in both halves; left by the dgrad kernel where the block would
// start) by three 4-byte LDS-DMAs into the first KiB of its slot and rebuilds the eight fragments as the dgrad kernel formed them
// (keep_pk on the packed pair): 12 B per sample instead of 2 F.
template <int F, int NTB, bool D8, bool H8, bool EXPAND = false>
__device__ __forceinline__ void wgrad_job(const NcaWgradArgs& a, const NcaWgradJob& job, int q, int nsplit, int lane, char* ring) {
    using R = WgradRing<F, NTB, D8, H8>;
    static_assert(!EXPAND || (!D8 && !H8), "the rebuilt block is a bf16 one (e5m2: wgrad_job_mx)");
    constexpr int MT = F / 32, ND_ = R::ND, NH_ = R::NH;
    constexpr int FR = EXPAND ? 1 + NH_ : R::FR;                       // 1 KiB pieces of a slot: an expand tile's D part is three 256-byte rows
    constexpr int NDMA = EXPAND ? 3 + NH_ : R::FR;                     // vector-memory operations per tile
    constexpr int NSLOT = EXPAND ? (FR <= 8 ? 4 : (FR <= 12 ? 3 : 2)) : R::NSLOT;
    static_assert(NSLOT * FR * 1024 <= NCA_WGRAD_LDS && (NSLOT - 1) * NDMA <= 63, "ring does not fit the wave's LDS share / counted wait");
    constexpr float HINV = 1.f / (float)(1 << NCA_H8_LOG2);
    const int lc = lane & 31, lh = lane >> 5;
    const int64_t per = (a.ntiles + nsplit - 1) / nsplit;
    const int64_t t0 = (int64_t)q * per, t1 = (t0 + per < a.ntiles) ? t0 + per : a.ntiles;
    const char* base = reinterpret_cast<const char*>(a.scratch);          // D region of this launch
    const char* base_b = reinterpret_cast<const char*>(a.scratch_b);      // region of the input / H blocks (may be the same)
    const int brow = job.b_row_bytes;
    f32x16 acc[MT][NTB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < NTB; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][c][i] = 0.f;
    float bsum[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) bsum[m] = 0.f;

    // every tile issues exactly FR DMA instructions (a fragment beyond the input block's 112 slots re-reads fragment 0 and is
    // zeroed after the read back), so "all but the youngest (NSLOT - 1) FR" is a compile-time vmcnt
    auto issue = [&](int64_t t, int slot) {
        const char* dp = base + t * a.rows_total + job.d_row0 + lane * 16;            // rows_total = bytes per 32-sample tile
        const char* bp = base_b + (t + a.tile0_b) * a.rows_total_b + job.b_row0 + lane * 16;
        char* dst = ring + slot * (FR * 1024);
        constexpr int DP = EXPAND ? 1 : ND_;                               // 1 KiB pieces of the slot's D part
        if constexpr (EXPAND) {
            // the lane's two mask words of this 32-sample tile (column tile t & 1 of wave tile t >> 1 of the store) and the sample's word
            const char* mp = a.mask + ((t + a.tile0_b) >> 1) * a.mask_stride + job.mask_off + lane * 16 + ((t + a.tile0_b) & 1) * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)mp, (__attribute__((address_space(3))) void*)dst, 4, 0, 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)mp, (__attribute__((address_space(3))) void*)(dst + 252), 4, 4, 2);   // (LDS dst + 256)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + t * a.rows_total + job.d_row0 + (lane & 31) * 4),
                                             (__attribute__((address_space(3))) void*)(dst + 512), 4, 0, 2);
        } else {
#pragma unroll
            for (int s = 0; s < ND_; ++s)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dp + s * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + s * 1024), 16, 0, 2);
        }
#pragma unroll
        for (int s = 0; s < NH_; ++s)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bp + ((H8 || s * 32 + 32 <= brow) ? s : 0) * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + (DP + s) * 1024), 16, 0, 2);
    };
    auto scale_of = [&](int64_t t) {      // the wave tile's inverse scale sits in the first of its two 32-sample records
        return D8 ? reinterpret_cast<const float*>(base + (t & ~(int64_t)1) * a.rows_total + job.dscale_off)[job.net] : 1.f;
    };
    const int64_t n = t1 > t0 ? t1 - t0 : 0;
#pragma unroll
    for (int p = 0; p < NSLOT - 1; ++p)
        if (p < n) issue(t0 + p, p);
    float sc = n > 0 ? scale_of(t0) : 1.f;
    for (int64_t i = 0; i < n; ++i) {
        const float nsc = i + 1 < n ? scale_of(t0 + i + 1) : 1.f;
        if (i + NSLOT - 1 < n) {
            issue(t0 + i + NSLOT - 1, (int)((i + NSLOT - 1) % NSLOT));       // into the slot tile i - 1 was read from (its reads have been consumed)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 1) * NDMA) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* slot = ring + (int)(i % NSLOT) * (FR * 1024) + lane * 16;
        constexpr int DP = EXPAND ? 1 : ND_;
        u32x4 XD[ND_], XH[NH_];
        if constexpr (EXPAND) {
            const char* sm = slot - lane * 12;
            const unsigned mw[2] = {*reinterpret_cast<const unsigned*>(sm), *reinterpret_cast<const unsigned*>(sm + 256)};
            const unsigned g2 = *reinterpret_cast<const unsigned*>(sm + 512);
            // fragment 2m + s2, word u: the packed pair's two mask bits sit 16 apart in the field of row tile m (as the dgrad kernel formed it)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const unsigned fld = mw[m >> 1] >> (8 * (m & 1));
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                    for (int u = 0; u < 4; ++u) XD[2 * m + s2][u] = keep_pk(g2, (fld >> (4 * s2 + u)) & 0x00010001u);
            }
        } else {
#pragma unroll
            for (int s = 0; s < ND_; ++s) XD[s] = *reinterpret_cast<const u32x4*>(slot + s * 1024);
        }
#pragma unroll
        for (int s = 0; s < NH_; ++s) XH[s] = (H8 || s * 32 + 32 <= brow) ? *reinterpret_cast<const u32x4*>(slot + (DP + s) * 1024) : (u32x4){0, 0, 0, 0};
        u32x4 TD[MT][2], TH[NTB][2];
        if constexpr (D8) transpose_block8<MT, ND_, true, true>(XD, lc, lh, H8 ? sc * HINV : sc, TD, &bsum);      // (an e4m3 partner's scale rides along)
        else transpose_block<MT>(XD, lc, lh, TD, &bsum);
        if constexpr (H8) transpose_block8<NTB, NH_, false, !D8>(XH, lc, lh, HINV, TH, nullptr);
        else transpose_block<NTB>(XH, lc, lh, TH, nullptr);
        // dW[o][i] += sum_n D^T-form[o][n] * H-form[n][i]: both operands are accumulator-layout tiles
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < NTB; ++c)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    acc[m][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(TD[m][s]), frag(TH[c][s]), acc[m][c], 0, 0, 0);
        sc = nsc;
    }
    if (D8 && H8) {          // the H scale rode on the D tiles: the bias sums (column sums of those tiles) carry it too
#pragma unroll
        for (int m = 0; m < MT; ++m) bsum[m] *= (float)(1 << NCA_H8_LOG2);
    }

    wgrad_write<F, NTB>(acc, bsum, job, a.slab + (int64_t)q * a.slab_stride, a.accumulate, lane);
}

// The 8-bit weight gradient (every D block of the launch is e5m2: fp8 staging).  A 64-sample wave tile is contracted by ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 per 32 x 32 block of dW: A = D^T (e5m2), B = H^T (e4m3), K = the wave tile's 64 samples, the
// tile's power-of-two inverse scale (and the 2^-NCA_H8_LOG2 of the layer inputs) as the A operand's e8m0 block scale -- twice
// the bf16 rate, no scaling or packing on the vector ALU.  The operands need the sample index on K, i.e. the blocks
// transposed: as before by MFMAs against an 8-bit identity (x 1 is exact), whose f32 results are re-encoded as bytes (exact
// for D and for e4m3 H blocks; the bf16 input block and a bf16 last-layer input are rounded to e4m3 here).  Lane (feature
// position, half h) then holds 16 samples of each of the wave tile's two 32-sample halves = its 32 K values; both operands
// use the same K order, which is all the contraction needs (tools/mx_mfma_probe.hip: lane maps, scales, formats).
// (the conversions write one half of a register and keep the other.  Both halves get written, so the start value is immaterial: a
// zero costs a v_mov per word, a live input ties the result to that input's register and costs a v_mov into the MFMA operand
// tuple afterwards -- an empty asm "defines" a register without an instruction, which the allocator places inside the tuple)
__device__ __forceinline__ int any_vgpr() {
    int v;
    asm volatile("" : "=v"(v));     // (volatile: one definition per use, or the compiler shares one and copies it)
    return v;
}
__device__ __forceinline__ unsigned z4_e5m2(float a, float b, float c, float d) {
    int v = any_vgpr();
    v = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, v, true);
    return (unsigned)v;
}
__device__ __forceinline__ unsigned z4_e4m3(float a, float b, float c, float d) {
    int v = any_vgpr();
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

// Four 32 x 32 byte tiles transposed by MFMAs against the 8-bit identity, results in ARCHITECTURAL registers: the compiler's own
// MFMAs put their results into accumulation registers -- all 256 of which hold dW here -- and then moves the tiles (and the dW
// blocks they displace) back and forth with v_accvgpr_read / _write: 672 such moves per wave tile, more than every other
// vector instruction of the loop together.  Inline assembly keeps the products in VGPRs (C = the constant 0: no zeroing either).
// The hardware does not interlock a vector-ALU read behind an MFMA write, and the compiler does not know these are MFMAs: the
// wait states are spelled out (2nd product of a tile on the 1st: back to back; VALU read after the last 8-pass product: 11,
// 21 given here once per four tiles).
template <bool E5M2>
__device__ __forceinline__ void transpose4_vgpr(const u32x4 (&x)[4], long E0, long E1, f32x16 (&z)[4]) {
    long lo[4], hi[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        lo[t] = (long)(((unsigned long)x[t][1] << 32) | x[t][0]);
        hi[t] = (long)(((unsigned long)x[t][3] << 32) | x[t][2]);
    }
    if (E5M2)
        asm volatile("v_mfma_f32_32x32x16_bf8_bf8 %0, %4, %12, 0\n\tv_mfma_f32_32x32x16_bf8_bf8 %1, %5, %12, 0\n\t"
                     "v_mfma_f32_32x32x16_bf8_bf8 %2, %6, %12, 0\n\tv_mfma_f32_32x32x16_bf8_bf8 %3, %7, %12, 0\n\t"
                     "v_mfma_f32_32x32x16_bf8_bf8 %0, %8, %13, %0\n\tv_mfma_f32_32x32x16_bf8_bf8 %1, %9, %13, %1\n\t"
                     "v_mfma_f32_32x32x16_bf8_bf8 %2, %10, %13, %2\n\tv_mfma_f32_32x32x16_bf8_bf8 %3, %11, %13, %3\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 4"
                     : "=&v"(z[0]), "=&v"(z[1]), "=&v"(z[2]), "=&v"(z[3])
                     : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(E0), "v"(E1));
    else
        asm volatile("v_mfma_f32_32x32x16_fp8_fp8 %0, %4, %12, 0\n\tv_mfma_f32_32x32x16_fp8_fp8 %1, %5, %12, 0\n\t"
                     "v_mfma_f32_32x32x16_fp8_fp8 %2, %6, %12, 0\n\tv_mfma_f32_32x32x16_fp8_fp8 %3, %7, %12, 0\n\t"
                     "v_mfma_f32_32x32x16_fp8_fp8 %0, %8, %13, %0\n\tv_mfma_f32_32x32x16_fp8_fp8 %1, %9, %13, %1\n\t"
                     "v_mfma_f32_32x32x16_fp8_fp8 %2, %10, %13, %2\n\tv_mfma_f32_32x32x16_fp8_fp8 %3, %11, %13, %3\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 4"
                     : "=&v"(z[0]), "=&v"(z[1]), "=&v"(z[2]), "=&v"(z[3])
                     : "v"(lo[0]), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[0]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "v"(E0), "v"(E1));
}

// Sum of an accumulator's 16 registers on ADJACENT register pairs (v_pk_add_f32 straight on the accumulator: 7 + 1 instructions).
// Written as a plain loop the compiler pairs the sums of two accumulators instead and gathers every operand pair with two v_mov:
// 128 moves per wave tile, a fifth of the loop's instructions -- and the loop is issue-bound (one wave per SIMD).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float colsum16(const f32x16& z) {
    f32x2 p0 = __builtin_shufflevector(z, z, 0, 1) + __builtin_shufflevector(z, z, 2, 3);
    f32x2 p1 = __builtin_shufflevector(z, z, 4, 5) + __builtin_shufflevector(z, z, 6, 7);
    f32x2 p2 = __builtin_shufflevector(z, z, 8, 9) + __builtin_shufflevector(z, z, 10, 11);
    f32x2 p3 = __builtin_shufflevector(z, z, 12, 13) + __builtin_shufflevector(z, z, 14, 15);
    p0 += p1; p2 += p3; p0 += p2;
    return p0[0] + p0[1];
}

// fragment S (1 KiB) of a block: global src + 1024 S -> LDS to + 1024 S (the immediate offset moves both addresses), non-temporal
template <int S>
__device__ __forceinline__ void dma_piece(const char* src, char* to) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)to, 16, S * 1024, 2);
}

// EXPAND (mode 5, the last hidden layer's job): the D block relu'(H_{NL-1}) g is NOT read -- per sample it has one distinct byte,
// e5m2(g), at the features whose mask bit is set.  A tile brings the lane's two mask words (8 B of the forward's store) and the
// sample's byte (as byte * 0x00010001; left by the dgrad kernel where the block would start) by three 4-byte LDS-DMAs into the
// first KiB of its slot, and the fragment bytes are rebuilt on the vector ALU, 4 bytes in 4 instructions: nibble, x 0x204081 &
// 0x01010101 (one bit per byte), packed 16-bit multiply by the byte.  The nibbles come in mask-bit order, so the rows of dW come
// out in the order wgrad_write<EXPAND> undoes.  9 B per sample instead of F.
template <int F, int NTB, bool H8, bool EXPAND = false>
__device__ __forceinline__ void wgrad_job_mx(const NcaWgradArgs& a, const NcaWgradJob& job, int q, int nsplit, int lane, char* ring) {
    using R = WgradRing<F, NTB, true, H8>;
    constexpr int MT = F / 32, ND_ = EXPAND ? 1 : R::ND, NH_ = R::NH, FR = ND_ + NH_;
    constexpr int NSLOT = EXPAND ? (F == 128 ? 6 : 4) : 4;       // (an expand tile is 5 KiB: more of them in flight)
    constexpr int NDMA = (EXPAND ? 3 : ND_) + NH_ + 1;              // vector-memory operations per tile: the fragments and the wave tile's scale
    constexpr int SC0 = NSLOT * FR * 1024;    // the scales' 256 bytes per slot, behind the fragments
    static_assert(SC0 + NSLOT * 256 <= NCA_WGRAD_LDS && NSLOT >= 2, "ring does not fit the wave's LDS share");
    static_assert(!EXPAND || H8, "expand jobs read an e4m3 H block");
    static_assert((NSLOT - 1) * NDMA <= 63, "counted wait");
    const int lc = lane & 31, lh = lane >> 5;
    // whole wave tiles per split: the two 32-sample halves of a wave tile share one scale and one MFMA
    const int64_t per = (((a.ntiles + nsplit - 1) / nsplit) + 1) & ~(int64_t)1;
    const int64_t t0 = (int64_t)q * per, t1 = (t0 + per < a.ntiles) ? t0 + per : a.ntiles;
    const char* base = reinterpret_cast<const char*>(a.scratch);
    const char* base_b = reinterpret_cast<const char*>(a.scratch_b);
    const int brow = job.b_row_bytes;
    f32x16 acc[MT][NTB];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int c = 0; c < NTB; ++c)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][c][i] = 0.f;
    float bsum[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) bsum[m] = 0.f;

    // Tiles are issued in order, so every address stream is a running wave-uniform pointer advanced by its stride (a few scalar adds
    // per tile; indexed by the tile number the loop spent ~150 scalar instructions per tile on 64-bit multiplies -- with one wave
    // per SIMD they are issue cycles like any other).  PAR = the tile's parity, known at every call site: the scale record (and
    // an expand job's mask words) belong to the wave tile, i.e. move on after the odd tile.
    static_assert(NSLOT == 2 || NSLOT == 4 || NSLOT == 6, "the tile issued NSLOT - 1 ahead has the other parity");
    const int64_t n = t1 > t0 ? t1 - t0 : 0;
    const char* p_d = base + t0 * a.rows_total + job.d_row0;
    const char* p_h = base_b + (t0 + a.tile0_b) * a.rows_total_b + job.b_row0;
    const char* p_sc = base + t0 * a.rows_total + job.dscale_off + job.net * 4;                 // (t0 is even)
    const char* p_m = EXPAND ? a.mask + ((t0 + a.tile0_b) >> 1) * a.mask_stride + job.mask_off : nullptr;
    int wslot = 0;
    auto issue = [&](auto PARC) __attribute__((always_inline)) {
        constexpr int PAR = decltype(PARC)::value;
        char* dst = ring + wslot * (FR * 1024);
        // The inverse scale of the tile's wave tile (in the record of its first 32-sample tile), one copy per lane, comes through the
        // ring as well: a register load gets the compiler's s_waitcnt vmcnt(0) in front of its first use, which drains every tile in
        // flight once per wave tile (-0.55 ms per launch at the bench size without it).  EVERY load of the loop is non-temporal:
        // loads of different cache policies return out of order with each other, and the counted wait below assumes order.
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p_sc, (__attribute__((address_space(3))) void*)(ring + SC0 + wslot * 256), 4, 0, 2);
        // (the instruction's immediate offset moves the global AND the LDS address: one address pair per block, no 64-bit add per fragment)
        static_assert(H8, "fragment s of the H block exists (the bf16 input block has 7 of 8)");
        if constexpr (EXPAND) {
            const char* mp = p_m + lane * 16 + PAR * 8;       // the lane's two mask words of this column tile
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)mp, (__attribute__((address_space(3))) void*)dst, 4, 0, 2);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)mp, (__attribute__((address_space(3))) void*)(dst + 252), 4, 4, 2);   // (LDS dst + 256)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p_d + (lane & 31) * 4), (__attribute__((address_space(3))) void*)(dst + 512), 4, 0, 2);
        } else {
            const char* dp = p_d + lane * 16;
            dma_piece<0>(dp, dst);
            if constexpr (ND_ > 1) dma_piece<1>(dp, dst);
            if constexpr (ND_ > 2) dma_piece<2>(dp, dst);
            if constexpr (ND_ > 3) dma_piece<3>(dp, dst);
        }
        static_assert(ND_ <= 4 && NH_ <= 4, "immediate offsets reach 4 KiB");
        const char* bp = p_h + lane * 16;
        dma_piece<0>(bp, dst + ND_ * 1024);
        if constexpr (NH_ > 1) dma_piece<1>(bp, dst + ND_ * 1024);
        if constexpr (NH_ > 2) dma_piece<2>(bp, dst + ND_ * 1024);
        if constexpr (NH_ > 3) dma_piece<3>(bp, dst + ND_ * 1024);
        p_d += a.rows_total;
        p_h += a.rows_total_b;
        if constexpr (PAR) {
            p_sc += 2 * a.rows_total;
            if constexpr (EXPAND) p_m += a.mask_stride;
        }
        wslot = wslot + 1 == NSLOT ? 0 : wslot + 1;
    };
    using Even = std::integral_constant<int, 0>;
    using Odd = std::integral_constant<int, 1>;
    if (0 < n) issue(Even{});
    if constexpr (NSLOT >= 4) {
        if (1 < n) issue(Odd{});
        if (2 < n) issue(Even{});
    }
    if constexpr (NSLOT == 6) {
        if (3 < n) issue(Odd{});
        if (4 < n) issue(Even{});
    }
    int rslot = 0;                                            // the slot of the tile being consumed
    const long ED0 = ident8<true>(8 * lh, lc), ED1 = ident8<true>(16 + 8 * lh, lc);          // e5m2 identity (D blocks)
    const long EH0 = ident8<false>(8 * lh, lc), EH1 = ident8<false>(16 + 8 * lh, lc);        // e4m3 identity (e4m3 H blocks)
    const u32x4 EB0 = ident_frag(8 * lh, lc), EB1 = ident_frag(16 + 8 * lh, lc);             // bf16 identity (bf16 H blocks)
    for (int64_t i = 0; i < n; i += 2) {
        i32x8 PA[MT], PB[NTB];
        float sc = 0.f;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int64_t ii = i + half;                     // (n is even: the launcher checks that the launch covers whole wave tiles)
            if (ii + NSLOT - 1 < n) {
                if (half == 0) issue(Odd{}); else issue(Even{});       // tile ii + NSLOT - 1
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 1) * NDMA) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (half == 0) sc = *reinterpret_cast<const float*>(ring + SC0 + rslot * 256 + lane * 4);
            const char* slot = ring + rslot * (FR * 1024) + lane * 16;
            rslot = rslot + 1 == NSLOT ? 0 : rslot + 1;
            u32x4 xd[MT];
            if constexpr (EXPAND) {
                const char* sm = slot - lane * 12;
                const unsigned mw[2] = {*reinterpret_cast<const unsigned*>(sm), *reinterpret_cast<const unsigned*>(sm + 256)};
                const unsigned g2 = *reinterpret_cast<const unsigned*>(sm + 512);
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const unsigned nib = __builtin_amdgcn_ubfe(mw[m >> 1], 8 * (m & 1) + 4 * (w & 1) + 16 * (w >> 1), 4);
                        xd[m][w] = keep_pk(__umul24(nib, 0x00204081u) & 0x01010101u, g2);
                    }
            } else {
#pragma unroll
                for (int m = 0; m < MT; ++m) xd[m] = *reinterpret_cast<const u32x4*>(slot + m * 1024);
            }
            if constexpr (MT == 4) {                         // D: e5m2 bytes -> transposed, bias sums, bytes again
                f32x16 z[4];
                transpose4_vgpr<true>(xd, ED0, ED1, z);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    bsum[m] = fmaf(colsum16(z[m]), sc, bsum[m]);
#pragma unroll
                    for (int w = 0; w < 4; ++w) PA[m][4 * half + w] = (int)z4_e5m2(z[m][4 * w], z[m][4 * w + 1], z[m][4 * w + 2], z[m][4 * w + 3]);
                }
            } else
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const u32x4 x = xd[m];
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8((long)(((unsigned long)x[1] << 32) | x[0]), ED0, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_32x32x16_bf8_bf8((long)(((unsigned long)x[3] << 32) | x[2]), ED1, z, 0, 0, 0);
                bsum[m] = fmaf(colsum16(z), sc, bsum[m]);
#pragma unroll
                for (int w = 0; w < 4; ++w) PA[m][4 * half + w] = (int)z4_e5m2(z[4 * w], z[4 * w + 1], z[4 * w + 2], z[4 * w + 3]);
            }
            if constexpr (H8 && NTB == 4) {                  // H: e4m3 bytes -> transposed bytes
                u32x4 x[4];
                f32x16 z[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) x[c] = *reinterpret_cast<const u32x4*>(slot + (ND_ + c) * 1024);
                transpose4_vgpr<false>(x, EH0, EH1, z);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int w = 0; w < 4; ++w) PB[c][4 * half + w] = (int)z4_e4m3(z[c][4 * w], z[c][4 * w + 1], z[c][4 * w + 2], z[c][4 * w + 3]);
            } else
#pragma unroll
            for (int c = 0; c < NTB; ++c) {                  // H: e4m3 bytes (or bf16 pairs, rounded to e4m3 x 2^NCA_H8_LOG2 here) -> transposed bytes
                f32x16 z;
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = 0.f;
                if constexpr (H8) {
                    const u32x4 x = *reinterpret_cast<const u32x4*>(slot + (ND_ + c) * 1024);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8((long)(((unsigned long)x[1] << 32) | x[0]), EH0, z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8((long)(((unsigned long)x[3] << 32) | x[2]), EH1, z, 0, 0, 0);
#pragma unroll
                    for (int w = 0; w < 4; ++w) PB[c][4 * half + w] = (int)z4_e4m3(z[4 * w], z[4 * w + 1], z[4 * w + 2], z[4 * w + 3]);
                } else {
                    const bool ok0 = (2 * c) * 32 + 32 <= brow, ok1 = (2 * c + 1) * 32 + 32 <= brow;
                    u32x4 x0 = *reinterpret_cast<const u32x4*>(slot + (ND_ + 2 * c) * 1024), x1 = *reinterpret_cast<const u32x4*>(slot + (ND_ + 2 * c + 1) * 1024);
                    if (!ok0) x0 = (u32x4){0u, 0u, 0u, 0u};
                    if (!ok1) x1 = (u32x4){0u, 0u, 0u, 0u};
                    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(x0), frag(EB0), z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(x1), frag(EB1), z, 0, 0, 0);
                    constexpr float DIV = 1.f / (float)(1 << NCA_H8_LOG2);
#pragma unroll
                    for (int w = 0; w < 4; ++w) PB[c][4 * half + w] = (int)cvt4_e4m3(z[4 * w], z[4 * w + 1], z[4 * w + 2], z[4 * w + 3], DIV);
                }
            }
        }
        // the wave tile's inverse scale as an e8m0 exponent, with the layer inputs' 2^-NCA_H8_LOG2
        const int sa = (int)(__float_as_uint(sc) >> 23) - NCA_H8_LOG2;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int c = 0; c < NTB; ++c)
                acc[m][c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(PA[m], PB[c], acc[m][c], 1 /* A: e5m2 */, 0 /* B: e4m3 */, 0, sa, 0, 127);
    }
    wgrad_write<F, NTB, EXPAND>(acc, bsum, job, a.slab + (int64_t)q * a.slab_stride, a.accumulate, lane);
}

// Mode 5: the output layer's bias gradient = sum over the wave tiles of the per-tile sums of d loss / d raw that the dgrad kernel left
// in the tile records (f32[2] behind the inverse scales) -- added in TILE order (workgroup b: tiles b, b + grid, ...; fixed tree),
// whichever wave ran which tile, into workgroup b's bias slot of the output-layer partials (the dgrad launch before it left the
// slots' shares at 0; the reduce kernel adds the slots up).
__global__ __launch_bounds__(256) void nca_sum_tile_records(const char* dregion, int64_t wave_tile_bytes, int64_t dscale_off, int64_t ntiles, int net0, int net1, int F,
                                                            float* oslab) {
    __shared__ float part[256];
    nca_tile_record_sum(dregion, wave_tile_bytes, dscale_off, ntiles, net0, net1, F, oslab, (int)blockIdx.x, (int)gridDim.x, part);
}
// n_wg: the workgroups of the dgrad launch of nets net0 .. net1 - 1 = the rows of oslab that hold their partials
hipError_t nca_launch_sum_tile_records(const char* dregion, int64_t wave_tile_bytes, int64_t dscale_off, int64_t ntiles, int net0, int net1, int F, float* oslab, int n_wg,
                                       hipStream_t st) {
    hipLaunchKernelGGL(nca_sum_tile_records, dim3(n_wg), dim3(256), 0, st, dregion, wave_tile_bytes, dscale_off, ntiles, net0, net1, F, oslab);
    return hipGetLastError();
}

// NW waves per workgroup (e5m2 staging: 1 or 4; each wave is its own (job, split) with its own ring -- nothing is shared, there is no barrier)
template <int F, bool D8, int NW = 1>
__global__ __launch_bounds__(64 * NW, 1) void nca_wgrad_bf16(const NcaWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wring[];
    // (NW > 1: the compact 1-D grid of working (job, split) pairs in both cases)
    const int wave = NW > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0;
    const int lane = NW > 1 ? (int)threadIdx.x & 63 : (int)threadIdx.x;
    char* ring = wring + wave * NCA_WGRAD_LDS;
    // e5m2 staging: a 1-D grid of exactly the working (job, split) pairs, job after job -- expand jobs have nsplit_x splits, the others
    // nsplit_std.  (A 2-D grid with idle workgroups does not do: the grid must fit ONE round of the chip's 4 x CUs slots, workgroups go
    // to the XCDs round-robin, and idle ones unbalance that by enough to push a few working ones into a second round: 8.5 instead of 4.6 ms.)
    int jy = blockIdx.y, qx = (int)blockIdx.x * NW + wave, ns = gridDim.x;
    if constexpr (D8) {
        jy = 0;
        for (;; ++jy) {
            ns = a.job[jy].expand ? a.nsplit_x : a.nsplit_std;
            if (qx < ns || jy + 1 >= a.njobs) break;
            qx -= ns;
        }
        if (qx >= ns) return;
    } else if constexpr (NW > 1) {          // bf16 output gradients: every job runs over nsplit_std splits
        ns = a.nsplit_std;
        jy = qx / ns;
        qx -= jy * ns;
        if (jy >= a.njobs) return;
    }
    const NcaWgradJob job = a.job[jy];
    // (at F = 128 the 112-slot input block and a hidden block have the same shape: one body serves both)
    if constexpr (D8) {
        s8_mode();
        // (a rebuilding job over the INPUT block: the encoded part of a last layer that is a skip layer -- four 32-slot tiles whatever the width)
        if (job.expand) { if (F != 128 && job.is_enc) wgrad_job_mx<F, 4, true, true>(a, job, qx, ns, lane, ring); else wgrad_job_mx<F, F / 32, true, true>(a, job, qx, ns, lane, ring); }
        else if (F != 128 && job.is_enc && job.h8) wgrad_job_mx<F, 4, true>(a, job, qx, ns, lane, ring);
        else wgrad_job_mx<F, F / 32, true>(a, job, qx, ns, lane, ring);         // (e5m2 D blocks come with e4m3 H blocks)
    } else {
        if (job.expand) {          // (bf16 output gradients: a hidden block has F / 32 column tiles, the input block -- a skip layer's encoded part -- four)
            if constexpr (!D8) { if (job.is_enc) wgrad_job<F, 4, false, false, true>(a, job, qx, ns, lane, ring); else wgrad_job<F, F / 32, false, false, true>(a, job, qx, ns, lane, ring); }
        }
        else if (F != 128 && job.is_enc && job.h8) wgrad_job<F, 4, D8, true>(a, job, qx, ns, lane, ring);
        else if (job.h8) wgrad_job<F, F / 32, D8, true>(a, job, qx, ns, lane, ring);
        else if (F != 128 && job.is_enc) wgrad_job<F, 4, D8, false>(a, job, qx, ns, lane, ring);
        else wgrad_job<F, F == 128 ? 4 : F / 32, D8, false>(a, job, qx, ns, lane, ring);
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
template <int F, int MODE, bool S8, bool RES = false, bool SKIP = false>
static hipError_t launch_fused_bf_mode_s(const NcaFusedArgs& a, int grid, hipStream_t st) {
    constexpr bool bwd = MODE == NCA_KM_BWD || MODE == NCA_KM_BWD_NR;
    size_t lds = (RES ? (size_t)a.res_bytes : 2 * BfCfg<F>::BUF_BYTES) + bf_const_bytes(MODE);
    if (bwd) lds += NCA_WAVES * 2 * (F + 1) * sizeof(float);
    if (MODE == NCA_KM_BWD_NR) lds += 2 * bf_wo_floats(F) * sizeof(float);
    static thread_local NcaFusedArgs b;
    const NcaFusedArgs* pa = &a;
    if (RES) {
        if (a.res_bytes <= 0 || a.nstages <= 0) return hipErrorInvalidValue;
        b = a;
        b.ctr_off = (int32_t)lds;            // the workgroup's tile counter
        lds += 16;
        pa = &b;
        const size_t dma_end = (size_t)a.stage[a.nstages - 1].lds_off + a.stage[a.nstages - 1].bytes;       // whole 1 KiB pieces
        if (dma_end > lds) lds = dma_end;
        if (lds > (size_t)NCA_LDS_BYTES) return hipErrorInvalidValue;
    }
    if (MODE == NCA_KM_BWD) lds += (size_t)NCA_WAVES * a.mask_layers * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_fused_bf16<F, MODE, S8, RES, SKIP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((nca_fused_bf16<F, MODE, S8, RES, SKIP>), dim3(grid), dim3(NCA_NT), lds, st, *pa);
    return hipGetLastError();
}
template <int F, int MODE, bool S8, bool RES = false>
static hipError_t launch_fused_bf_mode(const NcaFusedArgs& a, int grid, hipStream_t st) {
    bool skip = false;
    for (int n = 0; n < a.nnets; ++n) skip = skip || nca_has_skip(a.net[n].lay);
    if constexpr (RES || MODE == NCA_KM_BWD_NR) {
        // (resident images: the planner never chooses them for a net with a skip layer; the backward from a store runs no forward layer --
        // its sweep takes the skip layer's hidden-part image like any other)
        if (RES && skip) return hipErrorInvalidValue;
        return launch_fused_bf_mode_s<F, MODE, S8, RES, false>(a, grid, st);
    } else {
        return skip ? launch_fused_bf_mode_s<F, MODE, S8, false, true>(a, grid, st) : launch_fused_bf_mode_s<F, MODE, S8, false, false>(a, grid, st);
    }
}
template <int F>
static hipError_t launch_fused_bf(const NcaFusedArgs& a, int kmode, int grid, hipStream_t st, bool s8) {
    if (a.res_bytes > 0) {         // resident weight images (one net per launch; the host has checked that they fit)
        switch (kmode) {
            case NCA_KM_FWD: return launch_fused_bf_mode<F, NCA_KM_FWD, false, true>(a, grid, st);
            case NCA_KM_FWD_STORE: return s8 ? launch_fused_bf_mode<F, NCA_KM_FWD_STORE, true, true>(a, grid, st) : launch_fused_bf_mode<F, NCA_KM_FWD_STORE, false, true>(a, grid, st);
            case NCA_KM_BWD_NR: return s8 ? launch_fused_bf_mode<F, NCA_KM_BWD_NR, true, true>(a, grid, st) : launch_fused_bf_mode<F, NCA_KM_BWD_NR, false, true>(a, grid, st);
        }
        return hipErrorInvalidValue;
    }
    switch (kmode) {
        case NCA_KM_FWD: return launch_fused_bf_mode<F, NCA_KM_FWD, false>(a, grid, st);
        case NCA_KM_BWD: return launch_fused_bf_mode<F, NCA_KM_BWD, false>(a, grid, st);
        case NCA_KM_FWD_STORE: return s8 ? launch_fused_bf_mode<F, NCA_KM_FWD_STORE, true>(a, grid, st) : launch_fused_bf_mode<F, NCA_KM_FWD_STORE, false>(a, grid, st);
        case NCA_KM_BWD_NR: return s8 ? launch_fused_bf_mode<F, NCA_KM_BWD_NR, true>(a, grid, st) : launch_fused_bf_mode<F, NCA_KM_BWD_NR, false>(a, grid, st);
    }
    return hipErrorInvalidValue;
}

size_t nca_fused_bf16_lds_other(int F, int kmode) {
    const bool bwd = kmode == NCA_KM_BWD || kmode == NCA_KM_BWD_NR;
    return bf_const_bytes(kmode) + (bwd ? NCA_WAVES * 2 * (F + 1) * sizeof(float) : 0) + (kmode == NCA_KM_BWD_NR ? 2 * bf_wo_floats(F) * sizeof(float) : 0) + 16;       // (+ the tile counter of a resident launch)
}

hipError_t nca_launch_fused_bf16(int F, const NcaFusedArgs& a, int kmode, int grid, hipStream_t st, bool s8) {
    switch (F) {
        case 32: return launch_fused_bf<32>(a, kmode, grid, st, s8);
        case 64: return launch_fused_bf<64>(a, kmode, grid, st, s8);
        case 128: return launch_fused_bf<128>(a, kmode, grid, st, s8);
    }
    return hipErrorInvalidValue;
}

hipError_t nca_launch_pack2_bf16(const NcaLayout& ya, const float* prm_a, void* out_a, const NcaLayout& yb, const float* prm_b, void* out_b, hipStream_t st) {
    NcaPack2ArgsB a;
    a.y[0] = ya; a.y[1] = yb; a.prm[0] = prm_a; a.prm[1] = prm_b; a.out[0] = out_a; a.out[1] = out_b;
    const int total = (int)((ya.packed_bytes > yb.packed_bytes ? ya.packed_bytes : yb.packed_bytes) / 4u);
    const int grid = (total + 255) / 256;
    hipLaunchKernelGGL(nca_pack2_bf16, dim3(grid > 1024 ? 1024 : grid, 2), dim3(256), 0, st, a);
    return hipGetLastError();
}
hipError_t nca_launch_pack_bf16(const NcaLayout& y, const float* prm, void* out, hipStream_t st) {
    const int total = (int)(y.packed_bytes / 4u);
    const int grid = (total + 255) / 256;
    hipLaunchKernelGGL(nca_pack_bf16, dim3(grid > 1024 ? 1024 : grid), dim3(256), 0, st, y, prm, reinterpret_cast<unsigned*>(out));
    return hipGetLastError();
}

hipError_t nca_launch_wgrad_bf16(int F, const NcaWgradArgs& a, int nsplit, hipStream_t st, int waves_per_wg) {
    const bool d8 = a.njobs > 0 && a.job[0].d8 != 0;          // one format for every D block of a launch
    for (int j = 0; j < a.njobs; ++j)
        if ((a.job[j].d8 != 0) != d8 || (d8 && !a.job[j].h8)) return hipErrorInvalidValue;
    if (d8 && (a.ntiles & 1)) return hipErrorInvalidValue;        // whole wave tiles (two 32-sample tiles share a scale and an MFMA)
    if (waves_per_wg != 1 && waves_per_wg != 4) return hipErrorInvalidValue;
    dim3 grid(nsplit, a.njobs), block(64);
    if (d8) {          // one wave per working (job, split) pair
        if (a.nsplit_std <= 0 || a.nsplit_x < a.nsplit_std) return hipErrorInvalidValue;
        int total = 0;
        for (int j = 0; j < a.njobs; ++j) total += a.job[j].expand ? a.nsplit_x : a.nsplit_std;
        grid = dim3((total + waves_per_wg - 1) / waves_per_wg, 1);
    } else if (waves_per_wg == 4) {
        if (a.nsplit_std != nsplit) return hipErrorInvalidValue;
        grid = dim3((nsplit * a.njobs + 3) / 4, 1);
    }
    constexpr int L = NCA_WGRAD_LDS;          // the wave's ring of tile slots: four waves (four one-wave workgroups, or one workgroup of four) share a CU's 160 KiB
    if (waves_per_wg == 4) {
#define NCA_WG4(FF, DD)                                                                                                                            \
    do {                                                                                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nca_wgrad_bf16<FF, DD, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * L); \
        hipLaunchKernelGGL((nca_wgrad_bf16<FF, DD, 4>), grid, dim3(256), 4 * L, st, a);                                                            \
    } while (0)
        switch (F) {
            case 32: if (d8) NCA_WG4(32, true); else NCA_WG4(32, false); break;
            case 64: if (d8) NCA_WG4(64, true); else NCA_WG4(64, false); break;
            case 128: if (d8) NCA_WG4(128, true); else NCA_WG4(128, false); break;
            default: return hipErrorInvalidValue;
        }
#undef NCA_WG4
        return hipGetLastError();
    }
    switch (F) {
        case 32: if (d8) hipLaunchKernelGGL((nca_wgrad_bf16<32, true>), grid, block, L, st, a); else hipLaunchKernelGGL((nca_wgrad_bf16<32, false>), grid, block, L, st, a); break;
        case 64: if (d8) hipLaunchKernelGGL((nca_wgrad_bf16<64, true>), grid, block, L, st, a); else hipLaunchKernelGGL((nca_wgrad_bf16<64, false>), grid, block, L, st, a); break;
        case 128: if (d8) hipLaunchKernelGGL((nca_wgrad_bf16<128, true>), grid, block, L, st, a); else hipLaunchKernelGGL((nca_wgrad_bf16<128, false>), grid, block, L, st, a); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
