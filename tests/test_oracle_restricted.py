"""CPU: the oracle's restricted index (lo_ridx_*, oracle/lrge_oracle.c) answers a sample of queries exactly as the oracle's
full index does.  It is what lets tests/test_gpu_configs.py::test_c5_full and bench.py compare FORWARD counts at full
H. sapiens scale (2 000 000 targets, 7.5 G minimizers -- no host holds that index): the targets stream through mm_sketch chunk by
chunk and only the entries whose key occurs in the sample queries are kept (complete position lists for those keys: mm_idx_get,
index.c, is never asked about any other key by these queries), mid_occ is the whole set's.

Checked here on sets small enough for the full index: counts, has_mapping, every chain record (coordinates, score, cnt,
mlen / blen, dv, rep_len) and the sorted anchors, both presets, clean and repeat-rich data (the repeat-rich set is where
mid_occ and the query-side occurrence filter actually drop seeds), with and without -F.
"""
import numpy as np
import pytest


def _restricted(O, opt, sample, t, mid_occ, chunk, threads=3):
    b = O.RestrictedIndexBuilder(opt, sample)
    for a in range(0, t.n, chunk):
        ch = t.slice(a, min(t.n, a + chunk))
        b.add(ch.bases, ch.offsets, ch.names, threads=threads)
    seen, kept = b.n_minimizers_seen, b.n_kept
    return b.finish(mid_occ), seen, kept


@pytest.mark.parametrize("config,scale,preset", [("tiny_twoset", 1.0, 0), ("tiny_hifi", 1.0, 1), ("tiny_hifi", 1.0, 0), ("c2_repeats", 0.04, 0)])
def test_restricted_index_answers_like_the_full_one(oracle, config, scale, preset):
    from lrge_amd import synth
    O = oracle
    _, q, t = synth.make_config(config, scale)
    opt = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
    full = O.Index(O.ReadSet(t.seqs(), t.names), opt)
    Qo = O.ReadSet(q.seqs(), q.names)
    rc, c, h = full.twoset_counts(Qo, threads=4)
    assert rc == 0 and int(c.sum()) > 0
    rcf, cf, hf = full.twoset_counts(Qo, remove_internal=True, ratio=0.2, threads=4)
    sel = list(range(0, q.n, max(1, q.n // 12)))
    seqs = q.seqs()
    sample = O.ReadSet([seqs[i] for i in sel], [q.names[i] for i in sel])
    opt2 = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
    r, seen, kept = _restricted(O, opt2, sample, t, full.mid_occ, chunk=37)
    assert seen == full.n_minimizers and 0 < kept < seen
    assert r.mid_occ == full.mid_occ
    rc2, c2, h2 = r.twoset_counts(sample, threads=2)
    assert rc2 == 0
    assert np.array_equal(c[sel], c2) and np.array_equal(h[sel], h2)
    rc3, c3, h3 = r.twoset_counts(sample, remove_internal=True, ratio=0.2, threads=2)
    assert rc3 == 0 and np.array_equal(cf[sel], c3) and np.array_equal(hf[sel], h3)
    n_regs = 0
    for k in sel:
        a, b = full.map(seqs[k], q.names[k]), r.map(seqs[k], q.names[k])
        assert np.array_equal(a, b)
        n_regs += len(a)
        assert np.array_equal(full.anchors(seqs[k], q.names[k]), r.anchors(seqs[k], q.names[k]))
    assert n_regs > 0
    # the MM2 tie policy too
    full.opt.sort_mode = r.opt.sort_mode = O.SORT_MM2
    assert np.array_equal(full.twoset_counts(Qo, threads=4)[1][sel], r.twoset_counts(sample, threads=2)[1])


def test_restricted_index_needs_the_global_mid_occ(oracle):
    from lrge_amd import synth
    O = oracle
    _, q, t = synth.make_config("tiny_twoset")
    opt = O.make_opt(O.PRESET_AVA_ONT, dual=True)
    sample = O.ReadSet(q.seqs()[:2], q.names[:2])
    b = O.RestrictedIndexBuilder(opt, sample)
    b.add(t.bases, t.offsets, t.names)
    with pytest.raises(AssertionError):
        b.finish(0)


def test_oracle_index_in_target_parts_equals_the_one_index():
    """tools/c5_allcounts.py checks EVERY forward count of full-size C5 against the oracle on a host that cannot hold the oracle's index of
    all 2 000 000 targets: the targets are indexed in P parts (each with the whole set's mid_occ), the keys that are too frequent over
    ALL parts are found from the parts' own counts (a key above mid_occ overall reaches ceil((mid_occ + 1) / P) in some part) and
    dropped in every part, and the per-part distinct-target counts add up (disjoint targets: twoset.rs:286-317).  Here the same
    procedure on a repeat-rich set that fits: equal to the one index, count for count."""
    import numpy as np
    from lrge_amd import synth
    from oracle import oracle as O
    _, qs, ts = synth.make_config("c2_repeats", 0.04)
    opt = O.make_opt(O.PRESET_AVA_ONT, dual=True)
    full = O.Index(O.ReadSet(ts.seqs(), ts.names), opt)
    Qo = O.ReadSet(qs.seqs(), qs.names)
    rc, ec, eh = full.twoset_counts(Qo, threads=2)
    assert rc == 0 and int(ec.sum()) > 0
    P = 3
    b = [ts.n * i // P for i in range(P + 1)]
    parts = []
    for p in range(P):
        sub = ts.slice(b[p], b[p + 1])
        o = O.make_opt(O.PRESET_AVA_ONT, dual=True)
        o.mid_occ = full.mid_occ
        ix = O.Index(O.ReadSet(sub.seqs(), sub.names), o)
        ix.strip()
        parts.append(ix)
    thr = full.mid_occ // P + 1
    cand = np.unique(np.concatenate([ix.keys_at_least(thr) for ix in parts]))
    tot = sum(ix.counts_of(cand).astype(np.int64) for ix in parts)
    frequent = cand[tot > full.mid_occ]
    hashes = full.minimizers()["x"] >> np.uint64(8)
    keys, cnt = np.unique(hashes, return_counts=True)
    assert np.array_equal(frequent, keys[cnt > full.mid_occ]) and frequent.size > 0
    c = np.zeros(qs.n, np.uint32); h = np.zeros(qs.n, np.uint32)
    for ix in parts:
        ix.drop_keys(frequent)
        rc, pc, ph = ix.twoset_counts(Qo, threads=2)
        assert rc == 0
        c += pc; h |= ph
    assert np.array_equal(c, ec) and np.array_equal(h, eh)
