"""TEST INFRASTRUCTURE (oracle): pure-Python restatement of the random subset liblrge draws with --seed.

Follows liblrge/src/lib.rs:189-204 (`unique_random_set`): `StdRng::seed_from_u64(seed)` then
`rand::seq::index::sample(&mut rng, n, k)`.  The arithmetic is in crates that are not under /root/reference
(Cargo.lock:1001-1029: rand 0.9.4, rand_chacha 0.9.0, rand_core 0.9.5); this file restates their published
algorithms independently of include/lrge_rand.hpp (different language, integer-masking instead of fixed-width
types) so that the two can be checked against each other.  PARITY UNPINNED for seed -> subset (no seeded known
answer in the reference, no Rust toolchain here); the ChaCha block function is pinned by the published zero-key
key streams in tests/test_sampling.py.  Only tests/ may import this module.
"""
import struct

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


def _rotl(v, c):
    return ((v << c) & M32) | (v >> (32 - c))


def chacha_block(key_words, counter, rounds, stream=0):
    """64-byte ChaCha block as 16 u32 words; 64-bit counter in words 12-13, 64-bit stream id in 14-15."""
    s = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + \
        [counter & M32, (counter >> 32) & M32, stream & M32, (stream >> 32) & M32]
    x = list(s)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & M32; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & M32; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & M32 for a, b in zip(x, s)]


def seed_bytes_from_u64(state):
    """rand_core SeedableRng::seed_from_u64: PCG32 (XSH-RR) output, advanced before each draw, little-endian."""
    out = b""
    for _ in range(8):
        state = (state * 6364136223846793005 + 11634580027462260723) & M64
        xorshifted = (((state >> 18) ^ state) >> 27) & M32
        rot = state >> 59
        x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) % 32))) & M32
        out += struct.pack("<I", x)
    return out


class StdRng:
    """rand 0.9 StdRng = ChaCha12Rng, consumed 32 bits at a time."""

    def __init__(self, seed32):
        self.key = struct.unpack("<8I", seed32)
        self.counter = 0
        self.buf = []

    @classmethod
    def seed_from_u64(cls, seed):
        return cls(seed_bytes_from_u64(seed))

    def next_u32(self):
        if not self.buf:
            self.buf = chacha_block(self.key, self.counter, 12)
            self.counter += 1
        return self.buf.pop(0)


def sample_single_inclusive(rng, low, high):
    """UniformInt<u32>::sample_single_inclusive: widening multiply + Canon's one-step bias reduction."""
    rng_range = (high - low + 1) & M32
    if rng_range == 0:
        return rng.next_u32()
    m = rng.next_u32() * rng_range
    result, lo_order = m >> 32, m & M32
    if lo_order > ((-rng_range) & M32):
        new_hi = (rng.next_u32() * rng_range) >> 32
        if lo_order + new_hi > M32:
            result += 1
    return (low + result) & M32


def uniform_sample(rng, low, high_exclusive):
    """Uniform::<u32>::new(low, high).sample(rng): Lemire's method with rejection."""
    rng_range = high_exclusive - low
    thresh = ((-rng_range) & M32) % rng_range
    while True:
        m = rng.next_u32() * rng_range
        if (m & M32) >= thresh:
            return low + (m >> 32)


def choose_algorithm(length, amount):
    import numpy as np
    f = np.float32
    j = 0 if length < 500000 else 1
    if amount < 163:
        c0 = (f(1.6), f(8.0) / f(45.0))
        c1 = (f(10.0), f(70.0) / f(9.0))
        m4 = c0[j] * f(amount)
        if amount > 11 and f(length) < (c1[j] + m4) * f(amount):
            return "inplace"
        return "floyd"
    c = (f(270.0), f(330.0) / f(9.0))
    return "inplace" if f(length) < c[j] * f(amount) else "rejection"


def index_sample(rng, length, amount):
    algo = choose_algorithm(length, amount)
    if algo == "floyd":
        idx = []
        for j in range(length - amount, length):
            t = sample_single_inclusive(rng, 0, j)
            if t in idx:
                idx[idx.index(t)] = j
            idx.append(t)
        return idx
    if algo == "inplace":
        idx = list(range(length))
        for i in range(amount):
            j = sample_single_inclusive(rng, i, length - 1)
            idx[i], idx[j] = idx[j], idx[i]
        return idx[:amount]
    seen, idx = set(), []
    for _ in range(amount):
        pos = uniform_sample(rng, 0, length)
        while pos in seen:
            pos = uniform_sample(rng, 0, length)
        seen.add(pos)
        idx.append(pos)
    return idx


def unique_random_set(k, n, seed):
    if k > n:
        raise ValueError("Cannot generate %d unique values from a range of 0 to %d" % (k, n))
    return index_sample(StdRng.seed_from_u64(seed), n, k)
