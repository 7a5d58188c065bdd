// lrge_rand.hpp -- the random subset liblrge draws with --seed, restated for the C++ host side.
//
// liblrge/src/lib.rs:189-204 (`unique_random_set`): StdRng::seed_from_u64(seed) (or entropy), then
// rand::seq::index::sample(&mut rng, n, k) collected as Vec<u32>.  The arithmetic lives in third-party crates that
// are not under /root/reference (Cargo.lock:1001-1029): rand 0.9.4, rand_chacha 0.9.0, rand_core 0.9.5.  Their
// published algorithms, restated here:
//   * rand_core `SeedableRng::seed_from_u64`: a PCG32 stream (multiplier 6364136223846793005, increment
//     11634580027462260723, XSH-RR output) fills the 32-byte seed, 4 little-endian bytes per step;
//   * rand `StdRng` = rand_chacha `ChaCha12Rng`: ChaCha with 12 rounds, 256-bit key = seed, 64-bit block counter in
//     state words 12-13, 64-bit stream id 0 in words 14-15; `next_u32` hands out the key stream as little-endian
//     words in order (the 4-block buffering of rand_chacha is invisible for a u32-only consumer);
//   * rand `seq::index::sample` for lengths that fit u32: Floyd's combination algorithm, an in-place partial
//     Fisher-Yates, or rejection sampling with a hash set, chosen from (length, amount) by the constants below;
//   * rand `distr::uniform` for u32: `sample_single_inclusive` = one widening multiply plus Canon's one-step bias
//     reduction (used by `random_range`, i.e. Floyd and in-place); `Uniform::sample` = Lemire's widening multiply
//     with rejection below `(2^32 - range) % range` (used by the rejection sampler).
//
// PARITY UNPINNED for seed -> subset: the reference holds no seeded known answer (lib.rs:227-240 only checks that
// two calls agree) and no Rust toolchain exists here.  What is pinned: the ChaCha block function against the
// published ChaCha20 / ChaCha12 zero-key key streams (tests/test_sampling.py), and this file against an independent
// Python restatement (oracle/rand09.py) over every algorithm branch.
#pragma once
#include <cstdint>
#include <cstring>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

namespace lrge {
namespace rand09 {

inline uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

// One 64-byte ChaCha block (`rounds` = 12 for StdRng) as 16 little-endian words.
inline void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    std::memcpy(x, s, sizeof x);
    auto qr = [&](int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
    };
    for (int r = 0; r < rounds; r += 2) {
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15);
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

class StdRng {   // rand 0.9 StdRng = ChaCha12Rng
    uint32_t key_[8];
    uint64_t counter_ = 0;
    uint32_t buf_[16];
    int index_ = 16;

public:
    explicit StdRng(const uint8_t seed[32]) {
        for (int i = 0; i < 8; ++i)
            key_[i] = (uint32_t)seed[4 * i] | (uint32_t)seed[4 * i + 1] << 8 | (uint32_t)seed[4 * i + 2] << 16 | (uint32_t)seed[4 * i + 3] << 24;
    }
    // rand_core SeedableRng::seed_from_u64
    static StdRng seed_from_u64(uint64_t state) {
        uint8_t seed[32];
        for (int i = 0; i < 8; ++i) {
            state = state * 6364136223846793005ull + 11634580027462260723ull;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
            seed[4 * i] = (uint8_t)x; seed[4 * i + 1] = (uint8_t)(x >> 8); seed[4 * i + 2] = (uint8_t)(x >> 16); seed[4 * i + 3] = (uint8_t)(x >> 24);
        }
        return StdRng(seed);
    }
    // StdRng::from_rng(&mut rand::rng()): 32 bytes of OS entropy
    static StdRng from_entropy() {
        std::random_device rd;
        uint8_t seed[32];
        for (int i = 0; i < 8; ++i) { uint32_t v = rd(); std::memcpy(seed + 4 * i, &v, 4); }
        return StdRng(seed);
    }
    uint32_t next_u32() {
        if (index_ == 16) { chacha_block(key_, counter_++, 0, 12, buf_); index_ = 0; }
        return buf_[index_++];
    }
};

// UniformInt<u32>::sample_single_inclusive (rand 0.9, `unbiased` feature off)
inline uint32_t sample_single_inclusive(StdRng &rng, uint32_t low, uint32_t high) {
    uint32_t range = high - low + 1;
    if (range == 0) return rng.next_u32();
    uint64_t m = (uint64_t)rng.next_u32() * range;
    uint32_t result = (uint32_t)(m >> 32), lo_order = (uint32_t)m;
    if (lo_order > (uint32_t)(0u - range)) {
        uint32_t new_hi = (uint32_t)(((uint64_t)rng.next_u32() * range) >> 32);
        result += ((uint64_t)lo_order + new_hi) >> 32 ? 1u : 0u;
    }
    return low + result;
}

// Uniform::<u32>::new(0, length).sample(rng): Lemire with rejection
struct UniformU32 {
    uint32_t low, range, thresh;
    UniformU32(uint32_t lo, uint32_t hi_exclusive) : low(lo), range(hi_exclusive - lo) { thresh = range ? (uint32_t)(0u - range) % range : 0; }
    uint32_t sample(StdRng &rng) const {
        if (range == 0) return rng.next_u32();
        for (;;) {
            uint64_t m = (uint64_t)rng.next_u32() * range;
            if ((uint32_t)m >= thresh) return low + (uint32_t)(m >> 32);
        }
    }
};

inline std::vector<uint32_t> sample_floyd(StdRng &rng, uint32_t length, uint32_t amount) {
    std::vector<uint32_t> idx;
    idx.reserve(amount);
    for (uint32_t j = length - amount; j < length; ++j) {
        uint32_t t = sample_single_inclusive(rng, 0, j);
        for (auto &v : idx) if (v == t) { v = j; break; }
        idx.push_back(t);
    }
    return idx;
}

inline std::vector<uint32_t> sample_inplace(StdRng &rng, uint32_t length, uint32_t amount) {
    std::vector<uint32_t> idx(length);
    for (uint32_t i = 0; i < length; ++i) idx[i] = i;
    for (uint32_t i = 0; i < amount; ++i) {
        uint32_t j = sample_single_inclusive(rng, i, length - 1);
        std::swap(idx[i], idx[j]);
    }
    idx.resize(amount);
    return idx;
}

inline std::vector<uint32_t> sample_rejection(StdRng &rng, uint32_t length, uint32_t amount) {
    UniformU32 distr(0, length);
    std::unordered_set<uint32_t> cache;
    cache.reserve(amount);
    std::vector<uint32_t> idx;
    idx.reserve(amount);
    for (uint32_t i = 0; i < amount; ++i) {
        uint32_t pos = distr.sample(rng);
        while (!cache.insert(pos).second) pos = distr.sample(rng);
        idx.push_back(pos);
    }
    return idx;
}

enum class Algo { Floyd, Inplace, Rejection };

// rand::seq::index::sample's choice for length <= u32::MAX (f32 arithmetic, as published)
inline Algo choose_algorithm(uint32_t length, uint32_t amount) {
    int j = length < 500000u ? 0 : 1;
    if (amount < 163) {
        const float C[2][2] = {{1.6f, 8.0f / 45.0f}, {10.0f, 70.0f / 9.0f}};
        float amount_fp = (float)amount;
        float m4 = C[0][j] * amount_fp;
        if (amount > 11 && (float)length < (C[1][j] + m4) * amount_fp) return Algo::Inplace;
        return Algo::Floyd;
    }
    const float C[2] = {270.0f, 330.0f / 9.0f};
    return (float)length < C[j] * (float)amount ? Algo::Inplace : Algo::Rejection;
}

inline std::vector<uint32_t> index_sample(StdRng &rng, uint32_t length, uint32_t amount) {
    if (amount > length) throw std::invalid_argument("`amount` of samples must be less than or equal to `length`");
    switch (choose_algorithm(length, amount)) {
    case Algo::Floyd: return sample_floyd(rng, length, amount);
    case Algo::Inplace: return sample_inplace(rng, length, amount);
    default: return sample_rejection(rng, length, amount);
    }
}

}  // namespace rand09

// lib.rs:189-204
inline std::vector<uint32_t> unique_random_set(size_t k, uint32_t n, std::optional<uint64_t> seed) {
    if (k > n) throw std::invalid_argument("Cannot generate " + std::to_string(k) + " unique values from a range of 0 to " + std::to_string(n));
    rand09::StdRng rng = seed ? rand09::StdRng::seed_from_u64(*seed) : rand09::StdRng::from_entropy();
    return rand09::index_sample(rng, n, (uint32_t)k);
}

}  // namespace lrge
