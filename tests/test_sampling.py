"""Sub-sampling with --seed (liblrge/src/lib.rs:189-204, twoset.rs:153-155,632-652, ava.rs:130-134).

The product path is `lrge_hip_unique_random_set` (include/lrge_rand.hpp, C++); the checker is the independent
pure-Python restatement oracle/rand09.py.  The ChaCha block function is pinned by the published zero-key key
streams; seed -> subset is unpinned against real rand 0.9.4 (no seeded known answer in the reference)."""
import ctypes as C
import struct
import numpy as np
import pytest

from lrge_amd import _ffi
from lrge_amd.twoset import split_into_sets, unique_random_set
from oracle import rand09

# zero key, zero nonce, block 0: ChaCha20 (RFC 7539 section 2.3.2 family / draft-agl-tls-chacha20poly1305 TC1) and the
# 12- and 8-round variants (draft-strombergson-chacha-test-vectors, TC1, 256-bit key)
KEYSTREAM = {
    20: "76b8e0ada0f13d90405d6ae55386bd28bdd219b8a08ded1aa836efcc8b770dc7da41597c5157488d7724e03fb8d84a376a43b8f41518a11cc387b669b2ee6586",
    12: "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f0564f879d27ae3c02ce82834acfa8c793a629f2ca0de6919610be82f411326be",
    8: "3e00ef2f895f40d67f5bb8e81f09a5a12c840ec3ce9a7f3b181be188ef711a1e984ce172b9216f419f445367456d5619314a42a3da86b001387bfdb80e0cfe42",
}


def _block(key, counter, rounds):
    k = np.asarray(key, dtype=np.uint32)
    out = np.zeros(16, dtype=np.uint32)
    assert _ffi.lib().lrge_hip_chacha_block(k.ctypes.data, counter, rounds, out.ctypes.data) == 0
    return out


@pytest.mark.parametrize("rounds", [20, 12, 8])
def test_chacha_block_known_answers(rounds):
    assert _block([0] * 8, 0, rounds).tobytes().hex() == KEYSTREAM[rounds]
    assert struct.pack("<16I", *rand09.chacha_block([0] * 8, 0, rounds)).hex() == KEYSTREAM[rounds]


def test_chacha_block_counter_and_key_match_the_restatement():
    rng = np.random.Generator(np.random.PCG64(5))
    for _ in range(20):
        key = [int(v) for v in rng.integers(0, 2**32, size=8)]
        ctr = int(rng.integers(0, 2**63)) * 2 + 1
        assert list(_block(key, ctr, 12)) == rand09.chacha_block(key, ctr, 12)


# (k, n): every branch of index::sample's choice -- Floyd (k < 12; large n), in-place (small n / large k),
# rejection (k >= 163, n large), on both sides of the 500 000 switch -- and the edges k = 0, k = n, n = 1
CASES = [(0, 10), (1, 1), (5, 100), (11, 12), (12, 100), (12, 2000), (20, 100), (162, 4000), (162, 60000), (163, 40000),
         (163, 50000), (100, 499999), (100, 500000), (100, 700000), (150, 3000000), (1000, 1000), (5000, 30000),
         (15000, 400000), (15000, 549000), (15000, 551000), (15000, 1000000), (3, 2**32 - 1)]


@pytest.mark.parametrize("k,n", CASES)
def test_unique_random_set_matches_the_restatement(k, n):
    for seed in (0, 1, 42, 2**64 - 1):
        got = unique_random_set(k, n, seed)
        assert got.dtype == np.uint32 and len(got) == k
        assert got.tolist() == rand09.unique_random_set(k, n, seed), (k, n, seed)
        assert len(set(got.tolist())) == k and (k == 0 or int(got.max()) < n)


def test_every_algorithm_branch_is_exercised():
    algos = {rand09.choose_algorithm(n, k) for k, n in CASES}
    assert algos == {"floyd", "inplace", "rejection"}
    for j_lo, j_hi in ((100, 499999), (100, 500000)):          # both columns of the constant tables
        assert rand09.choose_algorithm(j_hi, j_lo) in ("floyd", "inplace")
    # the CLI defaults (-T 10000 -Q 5000) on a typical run: rejection sampling above ~550 k reads, in-place below
    assert rand09.choose_algorithm(549000, 15000) == "inplace" and rand09.choose_algorithm(551000, 15000) == "rejection"


def test_reference_unit_tests_of_unique_random_set():
    """lib.rs:210-263: size, range, uniqueness; same seed -> same set; no seed -> (almost surely) different sets;
    k > n refused."""
    for _ in range(200):
        r = unique_random_set(5, 100, None)
        assert len(r) == 5 and all(int(x) < 100 for x in r) and len(set(r.tolist())) == 5
    assert unique_random_set(5, 1000000, 42).tolist() == unique_random_set(5, 1000000, 42).tolist()
    assert unique_random_set(5, 10000000, None).tolist() != unique_random_set(5, 10000000, None).tolist()
    with pytest.raises(ValueError, match="Cannot generate"):
        unique_random_set(10, 5, None)
    out = (C.c_uint32 * 1)()
    assert _ffi.lib().lrge_hip_unique_random_set(10, 5, 1, 0, out) == _ffi.ERR_INVALID


def test_targets_are_the_last_sampled_indices():
    """twoset.rs:153-155 + split_into_hashsets (:632-652): pop from the end -> the last T indices are the targets."""
    idx = unique_random_set(30, 1000, 7)
    t, q = split_into_sets(idx, 20)
    assert t == set(idx[10:].tolist()) and q == set(idx[:10].tolist())


def test_seed_expansion_is_pcg32():
    """rand_core seed_from_u64: the first word for state 0 is the XSH-RR output of one LCG step from 0."""
    inc = 11634580027462260723
    xs = (((inc >> 18) ^ inc) >> 27) & 0xFFFFFFFF
    rot = inc >> 59
    first = ((xs >> rot) | (xs << (32 - rot))) & 0xFFFFFFFF
    assert rand09.seed_bytes_from_u64(0)[:4] == struct.pack("<I", first)
    assert len(rand09.seed_bytes_from_u64(12345)) == 32
