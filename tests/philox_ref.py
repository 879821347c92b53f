"""TEST INFRASTRUCTURE: NumPy restatement of nerf-ca_amd/csrc/nca_rng.hpp (Philox4x32-10 streams of (seed, iteration, stream, index),
the keyed Feistel bijection, the slot -> ray id rule of nca_draw_ray_ids) -- the checker of the device-side batch sampler.  Index
bookkeeping is integer work: the GPU tests compare bit for bit.  (The reference draws with NumPy's global generator on the host,
train/run_composite.py:250-260; what it fixes is the distribution, which tests/test_device_step.py checks on these streams.)"""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
STREAM_IDS, STREAM_PERM, STREAM_JITTER = 0, 1, 2
U32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Arrays of uint64 holding 32-bit words -> four arrays of 32-bit words (as uint64)."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & U32 for c in (c0, c1, c2, c3))
    k0, k1 = np.uint64(k0) & U32, np.uint64(k1) & U32
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c0, np.uint64(M1) * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ k0
        n1 = p1 & U32
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ k1
        n3 = p0 & U32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0, k1 = (k0 + np.uint64(W0)) & U32, (k1 + np.uint64(W1)) & U32
    return c0, c1, c2, c3


def rng_words(seed, n_iter, stream, index):
    index = np.asarray(index, dtype=np.uint64)
    it = np.uint64(n_iter & 0xFFFFFFFFFFFFFFFF)
    c2 = np.full(index.shape, it & U32, dtype=np.uint64)
    c3 = np.full(index.shape, ((it >> np.uint64(32)) & U32) ^ np.uint64((stream << 24) & 0xFFFFFFFF), dtype=np.uint64)
    seed = seed & 0xFFFFFFFFFFFFFFFF
    return philox4x32_10(index & U32, index >> np.uint64(32), c2, c3, seed & 0xFFFFFFFF, seed >> 32)


def below(lo, hi, n):
    """High 64 bits of (hi:lo) x n, exactly (Python integers)."""
    return np.array([(((int(h) << 32) | int(l)) * int(n)) >> 64 for l, h in zip(lo, hi)], dtype=np.int64)


def unit(w):
    return ((np.asarray(w, dtype=np.uint64) >> np.uint64(8)).astype(np.float32) * np.float32(5.9604644775390625e-08)).astype(np.float32)


def mix32(x):
    x = np.asarray(x, dtype=np.uint64) & U32
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x85EBCA6B)) & U32
    x ^= x >> np.uint64(13); x = (x * np.uint64(0xC2B2AE35)) & U32
    x ^= x >> np.uint64(16)
    return x


def perm_keys(seed, n_iter):
    a = rng_words(seed, n_iter, STREAM_PERM, np.array([0, 1]))
    return [int(a[0][0]), int(a[1][0]), int(a[2][0]), int(a[3][0]), int(a[0][1]), int(a[1][1])]


def half_bits(n):
    bits = 1
    while bits < 62 and (1 << bits) < n:
        bits += 1
    return (bits + 1) // 2


def perm(i, n, keys):
    """The keyed bijection of [0, n) applied to the array i."""
    half = half_bits(n)
    mask = np.uint64((1 << half) - 1)
    x = np.asarray(i, dtype=np.uint64).copy()
    todo = np.ones(x.shape, dtype=bool)
    while todo.any():
        l, r = x[todo] >> np.uint64(half), x[todo] & mask
        for q in range(6):
            f = mix32(r ^ np.uint64(keys[q])) & mask
            l, r = r, l ^ f
        x[todo] = (l << np.uint64(half)) | r
        todo &= x >= np.uint64(n)
    return x.astype(np.int64)


def ray_ids(seed, n_iter, R_global, n_var, var_ids, non_var_ids, n_rows, slot0=0, count=None):
    """nca_draw_ray_ids: the ids of slots slot0 .. slot0 + count - 1 of the global batch."""
    count = R_global - slot0 if count is None else count
    slots = np.arange(slot0, slot0 + count, dtype=np.uint64)
    w = rng_words(seed, n_iter, STREAM_IDS, slots)
    if n_var > 0 and var_ids is not None and len(var_ids) > 0:
        is_var = perm(slots, R_global, perm_keys(seed, n_iter)) < n_var
        iv, inv = below(w[0], w[1], len(var_ids)), below(w[0], w[1], len(non_var_ids))
        return np.where(is_var, np.asarray(var_ids, dtype=np.int64)[iv], np.asarray(non_var_ids, dtype=np.int64)[inv])
    return below(w[0], w[1], n_rows)


def uniform(seed, n_iter, n, stream=STREAM_JITTER):
    return unit(rng_words(seed, n_iter, stream, np.arange(n, dtype=np.uint64))[0])
