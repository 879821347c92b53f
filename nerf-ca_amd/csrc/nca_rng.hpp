// nca_rng.hpp -- the counter-based random streams of the per-step batch sampler (nca_draw_ray_ids / nca_draw_uniform / nca_begin_step).
//
// The reference draws its batch with NumPy's global generator on the host (train/run_composite.py:250-260: two np.random.choice with
// replacement, a concatenation, np.random.shuffle) and its depth jitter with torch.rand (train/model_helpers.py:8).  Neither stream can be
// reproduced bit for bit by anyone (NumPy's global state), so what has to match is the DISTRIBUTION: exactly n_var slots of the batch hold an
// i.i.d. uniform draw from the variance-ray ids, the others an i.i.d. uniform draw from the rest, in a uniformly random arrangement.
//
// Here every number is a pure function of (seed, iteration, stream, index) -- Philox4x32-10 (Salmon et al., SC'11) -- so that
//   * a rank draws ITS slots of the global batch without drawing the others (ray sharding: same batch on every rank, no broadcast),
//   * a captured HIP graph replays the draw with the iteration read from a device counter (no host work per step),
//   * the host-launched step, the graph step and the tests' NumPy restatement (tests/philox_ref.py) agree to the bit.
// "Concatenate and shuffle" of two i.i.d. groups is a uniformly random placement of n_var marks on R slots with an i.i.d. draw per slot: slot i
// is a variance slot iff perm(i) < n_var for a keyed bijection perm of [0, R) (a 6-round Feistel network over the next even power of two,
// cycle-walked into range) -- O(1) per slot, no sort, exactly n_var marks.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NCA_RNG_HD __host__ __device__ inline
#else
#define NCA_RNG_HD inline
#endif

enum { NCA_RNG_STREAM_IDS = 0, NCA_RNG_STREAM_PERM = 1, NCA_RNG_STREAM_JITTER = 2, NCA_RNG_STREAM_USER = 16 };

struct NcaU4 { uint32_t x, y, z, w; };

NCA_RNG_HD NcaU4 nca_philox4x32_10(NcaU4 c, uint32_t k0, uint32_t k1) {
    for (int i = 0; i < 10; ++i) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        NcaU4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// the four words of (seed, iteration, stream, index): counter = (index lo, index hi, iteration lo, iteration hi ^ stream << 24), key = seed
NCA_RNG_HD NcaU4 nca_rng_words(uint64_t seed, int64_t n_iter, int stream, uint64_t index) {
    NcaU4 c;
    c.x = (uint32_t)index;
    c.y = (uint32_t)(index >> 32);
    c.z = (uint32_t)(uint64_t)n_iter;
    c.w = (uint32_t)((uint64_t)n_iter >> 32) ^ ((uint32_t)stream << 24);
    return nca_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// uniform integer in [0, n): the high 64 bits of a 64 x 64-bit product (bias <= n / 2^64)
NCA_RNG_HD uint64_t nca_rng_below(uint32_t lo, uint32_t hi, uint64_t n) {
    const uint64_t r = ((uint64_t)hi << 32) | lo;
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul64hi(r, n);
#else
    return (uint64_t)(((unsigned __int128)r * n) >> 64);
#endif
}

// uniform float in [0, 1) with 24 random bits (torch.rand's resolution for float32)
NCA_RNG_HD float nca_rng_unit(uint32_t w) { return (float)(w >> 8) * 5.9604644775390625e-08f; }

NCA_RNG_HD uint32_t nca_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

// keyed bijection of [0, n), n >= 1: six Feistel rounds over 2 * half bits (the next even power of two >= n), cycle-walked into range.
// keys: six round keys (two Philox calls of the step's PERM stream).  Terminates: the walk follows one cycle of a permutation that starts
// inside [0, n) and must come back to it.
struct NcaPermKeys { uint32_t k[6]; };
NCA_RNG_HD NcaPermKeys nca_perm_keys(uint64_t seed, int64_t n_iter) {
    const NcaU4 a = nca_rng_words(seed, n_iter, NCA_RNG_STREAM_PERM, 0), b = nca_rng_words(seed, n_iter, NCA_RNG_STREAM_PERM, 1);
    NcaPermKeys k;
    k.k[0] = a.x; k.k[1] = a.y; k.k[2] = a.z; k.k[3] = a.w; k.k[4] = b.x; k.k[5] = b.y;
    return k;
}
NCA_RNG_HD int nca_perm_half_bits(uint64_t n) {
    int bits = 1;
    while (bits < 62 && ((uint64_t)1 << bits) < n) ++bits;
    return (bits + 1) / 2;           // 2 * half >= bits
}
NCA_RNG_HD uint64_t nca_perm(uint64_t i, uint64_t n, int half, const NcaPermKeys& key) {
    const uint64_t mask = ((uint64_t)1 << half) - 1;
    uint64_t x = i;
    do {
        uint64_t l = x >> half, r = x & mask;
        for (int q = 0; q < 6; ++q) {
            const uint64_t f = (uint64_t)nca_mix32((uint32_t)r ^ key.k[q]) & mask;          // (half <= 31: r fits 32 bits)
            const uint64_t nl = r;
            r = l ^ f;
            l = nl;
        }
        x = (l << half) | r;
    } while (x >= n);
    return x;
}
