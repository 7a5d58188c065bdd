"""Property-based parity (hypothesis): random small genomes -- with planted repeats, low-complexity stretches and
N runs -- random reads on both strands with substitutions / indels, random presets and modes.  Whatever comes
out, the device must report exactly the oracle's counts (two-set, inverse, all-vs-all, with and without the
internal-overlap filter) and exactly its chains.  Derandomised, so a failure reproduces."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

pytestmark = pytest.mark.gpu

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b


def _genome(rng, n, n_repeats, rep_len, lowc):
    g = _ACGT[rng.integers(0, 4, size=n)]
    if n_repeats and n > 4 * rep_len:
        unit = g[:rep_len].copy()
        for _ in range(n_repeats):                      # interspersed copies: high-occurrence minimizers, wrong-locus chains
            p = int(rng.integers(0, n - rep_len))
            g[p:p + rep_len] = unit
    if lowc and n > 2000:                               # a homopolymer and a dinucleotide stretch (HPC, tandem seeds)
        p = int(rng.integers(0, n - 400)); g[p:p + 150] = ord("A")
        p = int(rng.integers(0, n - 400)); g[p:p + 200] = np.tile(np.frombuffer(b"CA", dtype=np.uint8), 100)
    return g


def _reads(rng, g, n_reads, lo, hi, err, n_rate, prefix):
    seqs, names = [], []
    for i in range(n_reads):
        L = int(rng.integers(lo, hi + 1)); L = min(L, len(g))
        s0 = int(rng.integers(0, len(g) - L + 1))
        src = g[s0:s0 + L]
        if rng.integers(0, 2):
            src = _COMP[src[::-1]]
        u = rng.random(L)
        b = src.copy()
        sub = u < err * 0.5
        b[sub] = _ACGT[rng.integers(0, 4, size=int(sub.sum()))]
        reps = np.ones(L, dtype=np.int64)
        reps[(u >= err * 0.5) & (u < err * 0.75)] = 0   # deletions
        reps[(u >= err * 0.75) & (u < err)] = 2         # insertions (duplicated base)
        o = np.repeat(b, reps)
        if n_rate:
            o[rng.random(len(o)) < n_rate] = ord("N")
        if len(o) == 0:
            o = src[:1].copy()
        seqs.append(o.tobytes())
        names.append(b"%s%04d" % (prefix, i))
    return seqs, names


CASE = st.fixed_dictionaries(dict(
    seed=st.integers(0, 2**31 - 1), glen=st.integers(1500, 24000), n_rep=st.integers(0, 6), rep_len=st.integers(60, 700),
    lowc=st.booleans(), nq=st.integers(1, 14), nt=st.integers(2, 40), lo=st.integers(40, 900), span=st.integers(0, 4000),
    err=st.sampled_from([0.0, 0.005, 0.03, 0.08, 0.15]), n_rate=st.sampled_from([0.0, 0.0, 0.002]),
    preset=st.sampled_from(["ont", "pb"]), internal=st.booleans()))


def _check(ctx, oracle, c, skip=0, iters=0):
    from lrge_amd import engine
    from conftest import to_arrays
    rng = np.random.Generator(np.random.PCG64(c["seed"]))
    g = _genome(rng, c["glen"], c["n_rep"], c["rep_len"], c["lowc"])
    qseqs, qnames = _reads(rng, g, c["nq"], c["lo"], c["lo"] + c["span"], c["err"], c["n_rate"], b"q")
    tseqs, tnames = _reads(rng, g, c["nt"], c["lo"], c["lo"] + c["span"], c["err"], c["n_rate"], b"t")
    preset = 1 if c["preset"] == "pb" else 0
    opreset = oracle.PRESET_AVA_PB if preset else oracle.PRESET_AVA_ONT
    F = c["internal"]

    def up(seqs, ranks):
        b, o = to_arrays(seqs)
        return ctx.upload(b, o, ranks)
    qr, tr = engine.name_ranks(qnames, tnames)
    Qd, Td = up(qseqs, qr), up(tseqs, tr)
    ix = engine.Index(ctx, Td, preset)
    To, Qo = oracle.ReadSet(tseqs, tnames), oracle.ReadSet(qseqs, qnames)
    ixo = oracle.Index(To, oracle.make_opt(opreset, dual=True))
    if skip: ixo.opt.max_chain_skip = skip
    if iters: ixo.opt.max_chain_iter = iters
    st_ = ix.stats()
    assert (st_["n_minimizers"], st_["n_keys"], st_["mid_occ"]) == (ixo.n_minimizers, ixo.n_keys, ixo.mid_occ)
    # two-set forward
    counts, has = ix.overlap_twoset(Qd, remove_internal=F)
    rc, ec, eh = ixo.twoset_counts(Qo, remove_internal=F, threads=4)
    assert np.array_equal(counts, ec) and np.array_equal(has, eh), ("twoset", c)
    # inverse: the same index, the queries streamed
    inv = ix.overlap_inverse(Qd, remove_internal=F)
    rc, einv = ixo.inverse_counts(Qo, remove_internal=F, threads=4)
    assert np.array_equal(inv, einv), ("inverse", c)
    # chains of the two-set run (every field the PAF needs)
    cols = ["query", "target", "rev", "score", "cnt", "qs", "qe", "rs", "re", "mlen", "blen"]
    got = ix.chains(Qd, dual=True)
    got = np.stack([got[k].astype(np.int64) for k in cols], axis=1) if len(got) else np.zeros((0, len(cols)), np.int64)
    exp = []
    for qi, (s, nm) in enumerate(zip(qseqs, qnames)):
        if len(s) == 0:
            continue
        for r in ixo.map(s, nm):
            exp.append([qi] + [int(r[k]) for k in ["rid", "rev", "score", "cnt", "qs", "qe", "rs", "re", "mlen", "blen"]])
    exp = np.array(exp, dtype=np.int64).reshape(-1, len(cols))
    key = lambda a: a[np.lexsort(a.T[::-1])]
    assert got.shape == exp.shape and np.array_equal(key(got), key(exp)), ("chains", c)
    ix.free()
    # all-vs-all over the targets
    ar, _ = engine.name_ranks(tnames, tnames)
    Ad = up(tseqs, ar)
    ixa = engine.Index(ctx, Ad, preset)
    ixoa = oracle.Index(To, oracle.make_opt(opreset, dual=False))
    if skip: ixoa.opt.max_chain_skip = skip
    if iters: ixoa.opt.max_chain_iter = iters
    ava = ixa.overlap_ava(remove_internal=F)
    rc, eava = ixoa.ava_counts(remove_internal=F, threads=4)
    assert np.array_equal(ava, eava), ("ava", c)
    ixa.free()


@settings(max_examples=120, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
@given(CASE)
def test_random_small_sets_match_the_oracle(ctx, oracle, c):
    _check(ctx, oracle, c)


def test_property_cases_are_not_vacuous(ctx, oracle):
    """The generator above must actually produce overlapping reads (a fixed sample of its parameter space)."""
    from lrge_amd import engine
    from conftest import to_arrays
    total = 0
    for seed in range(6):
        rng = np.random.Generator(np.random.PCG64(seed))
        g = _genome(rng, 12000, 3, 300, True)
        qseqs, qnames = _reads(rng, g, 8, 600, 3000, 0.05, 0.0, b"q")
        tseqs, tnames = _reads(rng, g, 30, 600, 3000, 0.05, 0.0, b"t")
        qr, tr = engine.name_ranks(qnames, tnames)
        b, o = to_arrays(qseqs); Qd = ctx.upload(b, o, qr)
        b, o = to_arrays(tseqs); Td = ctx.upload(b, o, tr)
        ix = engine.Index(ctx, Td, 0)
        total += int(ix.overlap_twoset(Qd)[0].sum())
        ix.free()
    assert total > 100


@pytest.mark.parametrize("env", [
    {"LRGE_HIP_CHAIN": "lpg", "LRGE_HIP_DEBUG_MAX_SKIP": "100000", "LRGE_HIP_LPG_NO_PRUNE": "1"},      # lane-per-group kernel, loops run past the window
    {"LRGE_HIP_CHAIN": "lpg", "LRGE_HIP_DEBUG_MAX_SKIP": "100000"},      # the same with the pruned scan (the default): the bound, not max_skip, ends the loops
    {"LRGE_HIP_CHAIN": "hw", "LRGE_HIP_DEBUG_MAX_SKIP": "100000"},       # half-wave kernel, same
    {"LRGE_HIP_CHAIN": "lpg", "LRGE_HIP_LPG_NOTAB": "1", "LRGE_HIP_DEBUG_MAX_ITER": "40"},
    {"LRGE_HIP_CHAIN": "lpg", "LRGE_HIP_DEBUG_MAX_SKIP": "100000", "LRGE_HIP_LPG_SLOW_BUDGET": "0", "LRGE_HIP_LPG_SLOW_RATE": "0", "LRGE_HIP_LPG_SLOW_ENTRY_EVERY": "0", "LRGE_HIP_LPG_NO_PRUNE": "1"},   # every slow-path group is redone by k_chain_hw_redo
    {"LRGE_HIP_NO_PACKED": "1", "LRGE_HIP_NO_PACKED_INDEX": "1", "LRGE_HIP_QOCC_EXACT": "1", "LRGE_HIP_BATCH_ANCHORS": "3000"},
], ids=["lpg-noskip-fullscan", "lpg-noskip", "hw-noskip", "lpg-notab-iter40", "lpg-redo", "unpacked-exact-batched"])
def test_random_small_sets_forced_paths(ctx, oracle, env):
    """The same random cases through the general / rarely taken device paths (the oracle gets the same overrides of
    max_chain_skip / max_chain_iter)."""
    import os

    @settings(max_examples=10, deadline=None, derandomize=True, suppress_health_check=list(HealthCheck))
    @given(CASE)
    def run(c):
        _check(ctx, oracle, c, skip=int(env.get("LRGE_HIP_DEBUG_MAX_SKIP", 0)), iters=int(env.get("LRGE_HIP_DEBUG_MAX_ITER", 0)))

    opts = {k[len("LRGE_HIP_"):]: v for k, v in env.items()}
    for k, v in opts.items():
        ctx.set_option(k, v)
    try:
        run()
    finally:
        for k in opts:
            ctx.set_option(k, None)
