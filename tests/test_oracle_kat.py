"""CPU: pin the oracle against every known-answer test the reference holds for this path
(SURVEY.md section 8c): estimate.rs:163-342, mapping.rs:423-492."""
import math

import numpy as np
import pytest

INF = float("inf")


def test_per_read_estimate_kat(oracle):
    # estimate.rs:305-322
    assert oracle.per_read_estimate(100, 200.0, 1000, 100, 10) == 2910.0
    # estimate.rs:325-342
    assert oracle.per_read_estimate(100, 200.0, 1000, 0, 10) == INF


MEDIAN_KATS = [
    ([1, 3, 5, 7, 9], 5.0), ([3, 1, 7, 5, 9], 5.0), ([1, 3, 5, 7], 4.0), ([10], 10.0),
    ([-3, 1, 0, 3, -1], 0.0), ([1, 2, 3, INF], 2.5), ([-INF, 1, 2, 3], 1.5), ([-INF, 1, 2, INF], 1.5),
    ([INF, INF], INF), ([-INF, -INF], -INF), ([-1, -INF, 0, 1, INF], 0.0),
]


@pytest.mark.parametrize("data,expect", MEDIAN_KATS)
def test_median_kats(oracle, data, expect):
    # estimate.rs:163-265 -- median() itself does not filter infinities
    lo, med, hi = oracle.median(np.array(data, dtype=np.float32), finite=False)
    assert lo is None and hi is None and med == np.float32(expect)


def test_median_empty(oracle):
    assert oracle.median(np.zeros(0, dtype=np.float32), finite=False) == (None, None, None)


def test_median_with_quantiles(oracle):
    # estimate.rs:267-275
    r = oracle.median(np.arange(1, 11, dtype=np.float32), False, 0.15, 0.65)
    assert r == (np.float32(2.35), np.float32(5.5), np.float32(6.85))
    # estimate.rs:277-295
    d = np.array([1, 2, 3, 4, 5, 6, INF, INF, INF, INF], dtype=np.float32)
    assert oracle.median(d, False, 0.15, 0.65) == (np.float32(2.35), np.float32(5.5), np.float32(INF))


def test_median_finite_filter_and_nan_quirk(oracle):
    d = np.array([1, INF, INF], dtype=np.float32)
    assert oracle.median(d, True) == (None, np.float32(1.0), None)
    lo, med, hi = oracle.median(d, False)        # inf*0 = NaN is reference behaviour
    assert med == np.float32(INF) or math.isnan(med)


def test_is_internal_kats(oracle):
    # mapping.rs:423-492: (qlen, qs, qe, rev, tlen, ts, te, ratio) -> expected
    assert oracle.is_internal(390, 46, 317, False, 278, 4, 275, 0.2)
    assert oracle.is_internal(298, 1, 297, False, 398, 54, 350, 0.2)
    assert not oracle.is_internal(390, 0, 355, False, 418, 39, 394, 0.05)


def test_inverse_predicate_differs(oracle):
    # twoset.rs:493-517: the inverse predicate skips LARGE overhangs (sense opposite to is_internal)
    assert oracle.inverse_skip(1000, 400, 600, False, 1000, 400, 600, 0.2)      # overhang 800 > 40
    assert not oracle.inverse_skip(1000, 0, 600, False, 700, 100, 700, 0.2)     # overhang 0
    assert oracle.is_internal(1000, 0, 600, False, 700, 100, 700, 0.2)          # ... which is_internal drops


def test_hash64_invertible_range(oracle):
    mask = (1 << 30) - 1
    vals = {oracle.lib().lo_hash64(i, mask) for i in range(5000)}
    assert len(vals) == 5000 and max(vals) <= mask


def test_sketch_basics(oracle):
    rng = np.random.Generator(np.random.PCG64(1))
    seq = bytes(np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 5000)])
    for k, hpc in ((15, False), (19, True)):
        mz = oracle.sketch(seq, 5, k, rid=7, is_hpc=hpc)
        assert len(mz) > 0
        assert np.all(np.diff(mz["y"].astype(np.int64)) > 0)          # output sorted by position
        assert np.all(mz["y"] >> 32 == 7)
        span = mz["x"] & 0xff
        assert np.all(span >= k) if hpc else np.all(span == k)
        # reverse complement gives the same hashes on the opposite strand
        comp = bytes.maketrans(b"ACGT", b"TGCA")
        rc = seq.translate(comp)[::-1]
        mz2 = oracle.sketch(rc, 5, k, rid=7, is_hpc=hpc)
        assert set(mz["x"].tolist()) == set(mz2["x"].tolist())
    assert len(oracle.sketch(b"ACGTACGTAC", 5, 15)) == 0                 # shorter than k
    assert len(oracle.sketch(b"N" * 100, 5, 15)) == 0


def test_sort_policies_agree_on_distinct_keys(oracle):
    rng = np.random.Generator(np.random.PCG64(3))
    a = np.zeros(5000, dtype=oracle.MM128)
    a["x"] = rng.permutation(5000).astype(np.uint64) * np.uint64(977)
    a["y"] = np.arange(5000, dtype=np.uint64)
    s0 = oracle.sort128x(a, oracle.SORT_STABLE)
    s1 = oracle.sort128x(a, oracle.SORT_MM2)
    assert np.array_equal(s0, s1)
    assert np.all(np.diff(s0["x"].astype(np.int64)) > 0)
    # with ties the stable policy keeps input order; the radix emulation is a permutation of it
    a["x"] = rng.integers(0, 50, 5000).astype(np.uint64)
    s0 = oracle.sort128x(a, oracle.SORT_STABLE)
    s1 = oracle.sort128x(a, oracle.SORT_MM2)
    assert np.array_equal(s0["x"], s1["x"])
    for v in range(50):
        ys = s0["y"][s0["x"] == v]
        assert np.all(np.diff(ys.astype(np.int64)) > 0)
        assert sorted(s1["y"][s1["x"] == v].tolist()) == ys.tolist()


def test_twoset_estimate_recovers_genome_size(oracle, tiny_ont):
    ds = tiny_ont
    T = oracle.ReadSet(ds.t.seqs(), ds.t.names)
    Q = oracle.ReadSet(ds.q.seqs(), ds.q.names)
    opt = oracle.make_opt(oracle.PRESET_AVA_ONT, dual=True)
    ix = oracle.Index(T, opt)
    assert ix.mid_occ >= 10
    rc, counts, has = ix.twoset_counts(Q, threads=4)
    assert rc == 0 and has.sum() == Q.n
    avg = np.float32(ds.t.lens().sum()) / np.float32(ds.t.n)
    est = [oracle.per_read_estimate(int(l), float(avg), ds.t.n, int(c), 100) for l, c in zip(ds.q.lens(), counts)]
    lo, med, hi = oracle.median(np.array(est, dtype=np.float32), True, 0.15, 0.65)
    assert 0.7 * 200_000 < med < 1.4 * 200_000
    # tie policy: minimap2's unstable radix order vs the stable order give the same counts here
    opt2 = oracle.make_opt(oracle.PRESET_AVA_ONT, dual=True, sort_mode=oracle.SORT_MM2)
    rc2, counts2, _ = oracle.Index(T, opt2).twoset_counts(Q, threads=4)
    assert np.array_equal(counts, counts2)


def test_ava_counts_symmetric(oracle):
    from lrge_amd import synth
    g, reads, _ = synth.make_config("tiny_ava")
    R = oracle.ReadSet(reads.seqs(), reads.names)
    opt = oracle.make_opt(oracle.PRESET_AVA_ONT, dual=False)
    ix = oracle.Index(R, opt)
    rc, counts = ix.ava_counts(threads=4)
    assert rc == 0 and counts.sum() % 2 == 0 and counts.sum() > 0
    # brute force from the per-read mappings, with the dual flag ON as cross-check of NO_DUAL pruning
    opt_d = oracle.make_opt(oracle.PRESET_AVA_ONT, dual=True)
    ixd = oracle.Index(R, opt_d)
    pairs = set()
    for q in range(R.n):
        for r in ixd.map(R.seq(q), R.names[q]):
            if int(r["rid"]) != q:
                pairs.add((min(q, int(r["rid"])), max(q, int(r["rid"]))))
    ref = np.zeros(R.n, dtype=np.int64)
    for a, b in pairs:
        ref[a] += 1; ref[b] += 1
    # dual=yes sees every pair from both sides, so it can only find a superset of the NO_DUAL pairs
    assert np.all(ref >= counts)
    assert (ref - counts).sum() <= 0.05 * ref.sum() + 4


def test_duplicate_ids_rejected(oracle):
    from lrge_amd import synth
    g, reads, _ = synth.make_config("tiny_ava", scale=0.2)
    names = list(reads.names); names[3] = names[1]
    R = oracle.ReadSet(reads.seqs(), names)
    ix = oracle.Index(R, oracle.make_opt(oracle.PRESET_AVA_ONT, dual=False))
    rc, _ = ix.ava_counts(threads=2)
    assert rc == -7
