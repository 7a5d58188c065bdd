"""GPU: the BASELINE.json configurations themselves against the CPU oracle, at their full sizes.

  C2  4.4 Mbp ONT, two-set -Q 5000 -T 10000   every count, has_mapping, mid_occ, estimate / q15 / q65 (f32, 0 ulp); again with -F;
                                               every PafRecord of the first 300 queries as PAF lines
  C3  12 Mbp ONT, all-vs-all -n 20000          every count
  C4  143 Mbp ONT, two-set -Q 50000 -T 100000  the first 1024 queries against the FULL target index, mid_occ;
                                               all 50 000 counts through size-independent properties
  C5  3.1 Gbp HiFi, -Q 100000 -T 2000000       FULL SIZE (test_c5_full: counter-based generator, reads written into HBM): inverse
                                               against the oracle on a 20 000-target range + additivity over all targets; forward:
                                               index statistics against the oracle fixture + size-independent properties
  C5/10  310 Mbp HiFi, -Q 10000 -T 200000      both presets (ava-pb = library semantics, ava-ont = what the reference CLI
                                               runs), forward (first 256 queries) and inverse (--use-min-ref: every
                                               indexed read, targets streamed) -- the full-size C5 needs ~25 minutes of
                                               data generation and is run by tools/run_config.py (profiles/)

Reference semantics: twoset.rs:286-317 (forward), ava.rs:271-306, twoset.rs:485-524 (inverse).  Each oracle run is
repeated with sort_mode = SORT_MM2 (the emulation of ksort.h's unstable radix_sort_128x): the device implements the stable
tie order, so this is where the two policies are compared on data with enough anchors for ties to occur.
The oracle is the builder's restatement of minimap2 2.30 -- parity is unpinned against the real thing.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

THREADS = 0        # the oracle's default: the CPUs the host grants (oracle.default_threads)


def _sets(ctx, q, t):
    from lrge_amd import engine
    qr, tr = engine.name_ranks(q.names, t.names)
    return ctx.upload(q.bases, q.offsets, qr), ctx.upload(t.bases, t.offsets, tr)


def _oracle_index(oracle, reads, preset, dual=True):
    opt = oracle.make_opt(oracle.PRESET_AVA_PB if preset == 1 else oracle.PRESET_AVA_ONT, dual=dual)
    return oracle.Index(oracle.ReadSet(reads.seqs(), reads.names), opt)


def _both_policies(oracle, ixo, fn):
    """fn() under the stable tie order and under the radix_sort_128x emulation."""
    out = []
    for mode in (oracle.SORT_STABLE, oracle.SORT_MM2):
        ixo.opt.sort_mode = mode
        out.append(fn())
    ixo.opt.sort_mode = oracle.SORT_STABLE
    return out


def test_c2_full(ctx, oracle):
    from lrge_amd import engine, synth
    gsize, q, t = synth.make_config("c2_bact_twoset")
    assert (q.n, t.n) == (5000, 10000)
    Qd, Td = _sets(ctx, q, t)
    Qd.presketch(0)
    ix = engine.Index(ctx, Td, 0)
    counts, has = ix.overlap_twoset(Qd)
    cn = ctx.counters()
    # the dead-pair filter (k_expand_q) drops chance matches before the sort (few on a 4.4 Mbp genome: 3 %; more than half of all
    # anchors at C4, 71 % at H. sapiens scale) and not one anchor of a pair that is chained
    assert cn["chain_anchors"] <= cn["anchors_kept"] < cn["anchors"], cn
    st = ix.stats()
    avg = np.float32(t.lens().sum()) / np.float32(t.n)
    est = ctx.estimates(counts, q.lens(), float(avg), t.n, 100)
    med = engine.median(est, True, 0.15, 0.65)
    ix.free(); Qd.free(); Td.free()

    ixo = _oracle_index(oracle, t, 0)
    assert st["mid_occ"] == ixo.mid_occ and st["n_minimizers"] == ixo.n_minimizers and st["n_keys"] == ixo.n_keys
    Qo = oracle.ReadSet(q.seqs(), q.names)
    (rc, ec, eh), (rc2, ec2, eh2) = _both_policies(oracle, ixo, lambda: ixo.twoset_counts(Qo, threads=THREADS))
    assert rc == 0 and rc2 == 0
    assert np.array_equal(counts, ec) and np.array_equal(has, eh)
    assert np.array_equal(ec, ec2) and np.array_equal(eh, eh2), "tie policies differ on %d reads" % int((ec != ec2).sum())
    # estimates: f32, bit for bit (same counts -> same multiset; per_read_estimate is evaluated with one rounding per op)
    eest = np.array([oracle.per_read_estimate(int(l), float(avg), t.n, int(c), 100) for l, c in zip(q.lens(), ec)], dtype=np.float32)
    assert np.array_equal(est.view(np.uint32), eest.view(np.uint32))
    emed = oracle.median(eest, True, 0.15, 0.65)
    assert [np.float32(x).view(np.uint32) for x in med] == [np.float32(x).view(np.uint32) for x in emed]
    assert abs(float(med[1]) - gsize) / gsize < 0.15          # 4.4 Mbp genome: the estimate lands near it


def test_c2_filter_internal_and_paf_records(ctx, oracle):
    """C2 again with -F (mapping.rs:59-77 through twoset.rs:295-299) on all 5 000 queries, and every PafRecord field of the
    first 300 queries (aligner.rs:244-291: coordinates, cm, s1, mlen / blen, dv, rl) as the multiset of PAF lines."""
    from lrge_amd import engine, paf, synth
    gsize, q, t = synth.make_config("c2_bact_twoset")
    Qd, Td = _sets(ctx, q, t)
    ix = engine.Index(ctx, Td, 0)
    counts_f, has_f = ix.overlap_twoset(Qd, remove_internal=True, max_overhang_ratio=0.2)
    counts_p, _ = ix.overlap_twoset(Qd)
    n = 300
    sub = q.slice(0, n)
    qr, tr = engine.name_ranks(q.names, t.names)
    Sd = ctx.upload(sub.bases, sub.offsets, qr[:n])
    chains = ix.chains(Sd, dual=True)
    rl, ss, nk = ix.paf_stats(Sd)
    ix.free(); Sd.free(); Qd.free(); Td.free()
    ixo = _oracle_index(oracle, t, 0)
    Qo = oracle.ReadSet(q.seqs(), q.names)
    (rc, ec, eh), (rc2, ec2, eh2) = _both_policies(oracle, ixo, lambda: ixo.twoset_counts(Qo, remove_internal=True, ratio=0.2, threads=THREADS))
    assert rc == 0 and rc2 == 0
    assert np.array_equal(counts_f, ec) and np.array_equal(has_f, eh)
    assert np.array_equal(ec, ec2)
    assert int((counts_f != counts_p).sum()) > 100            # the filter does drop overlaps on this data
    qlens, tlens = [int(x) for x in sub.lens()], [int(x) for x in t.lens()]
    got = sorted(paf.paf_lines(chains, sub.names, qlens, t.names, tlens, rl, ss, nk))
    exp = []
    seqs = sub.seqs()
    for qi in range(n):
        for r in ixo.map(seqs[qi], sub.names[qi]):
            ti = int(r["rid"])
            exp.append("\t".join([sub.names[qi].decode(), str(qlens[qi]), str(r["qs"]), str(r["qe"]), "-" if r["rev"] else "+",
                                  t.names[ti].decode(), str(tlens[ti]), str(r["rs"]), str(r["re"]), str(r["mlen"]), str(r["blen"]), "0",
                                  "tp:A:S", "cm:i:%d" % r["cnt"], "s1:i:%d" % r["score"], "dv:f:" + paf.format_dv(r["dv"]),
                                  "rl:i:%d" % r["rep_len"]]))
    exp.sort()
    assert len(exp) > 5000 and len(got) == len(exp)
    bad = [i for i, (a, b) in enumerate(zip(got, exp)) if a != b]
    assert not bad, "first differing PAF line:\n%s\n%s" % (got[bad[0]], exp[bad[0]])


def test_c3_full_ava(ctx, oracle):
    from lrge_amd import engine, synth
    gsize, reads, _ = synth.make_config("c3_yeast_ava")
    assert reads.n == 20000
    (ranks,) = engine.name_ranks(reads.names)
    Rd = ctx.upload(reads.bases, reads.offsets, ranks)
    Rd.presketch(0)
    ix = engine.Index(ctx, Rd, 0)
    counts = ix.overlap_ava()
    st = ix.stats()
    ix.free(); Rd.free()
    ixo = _oracle_index(oracle, reads, 0, dual=False)
    assert st["mid_occ"] == ixo.mid_occ
    (rc, ec), (rc2, ec2) = _both_policies(oracle, ixo, lambda: ixo.ava_counts(threads=THREADS))
    assert rc == 0 and rc2 == 0
    assert np.array_equal(counts, ec)
    assert np.array_equal(ec, ec2), "tie policies differ on %d reads" % int((ec != ec2).sum())
    assert int(counts.sum()) % 2 == 0 and int(counts.sum()) > 100000     # symmetric counting: every pair adds two


def test_c4_sampled_and_properties(ctx, oracle):
    from lrge_amd import engine, synth
    gsize, q, t = synth.make_config("c4_dmel_twoset")
    assert (q.n, t.n) == (50000, 100000)
    Qd, Td = _sets(ctx, q, t)
    Qd.presketch(0)
    ix = engine.Index(ctx, Td, 0)
    # regression guard for the ordered table (k_index.h, ht_home): the 30-bit hash leaves the top byte partly empty, and
    # byte-reversed that byte sits in the middle of the home-slot key -- unless it is stretched to a full byte the keys of every
    # 22-slot stretch crowd into its first quarter (mean displacement 3.1 slots; 1.05 with the byte stretched; 0.30 with the byte
    # also sent through the distribution function of a window minimum, the default).  Results never depend on it; only a table
    # of this size shows it.
    assert ix.build_counters["table_disp_sum"] < 0.6 * ix.stats()["n_keys"]
    counts, has = ix.overlap_twoset(Qd)
    cn = ctx.counters()
    assert cn["chain_anchors"] <= cn["anchors_kept"] < 0.6 * cn["anchors"], cn      # (the dead-pair filter: more than half of C4's anchors are chance matches)
    st = ix.stats()
    # size-independent properties on all 50 000 queries
    #  * streamed reads are independent: any sub-range of the queries gives the same counts for those reads
    lo, hi = 20000, 23000
    sub = q.slice(lo, hi)
    qr, tr = engine.name_ranks(q.names, t.names)
    Sd = ctx.upload(sub.bases, sub.offsets, qr[lo:hi])
    c_sub, h_sub = ix.overlap_twoset(Sd)
    assert np.array_equal(c_sub, counts[lo:hi]) and np.array_equal(h_sub, has[lo:hi])
    #  * a count never exceeds the number of target reads that overlap the query's true interval by >= 1 base (no false
    #    positives on a repeat-free random genome), and has_mapping <=> count > 0 without -F
    assert np.array_equal(has > 0, counts > 0)
    ts, te = np.sort(t.starts), np.sort(t.ends)
    for i in range(0, q.n, 97):
        # targets that start before the query's source interval ends, minus those that end before it starts
        n_touch = int(np.searchsorted(ts, q.ends[i], side="left")) - int(np.searchsorted(te, q.starts[i], side="right"))
        assert counts[i] <= max(n_touch, 0), (i, int(counts[i]), n_touch)
    #  * sensitivity (plausibility against the generator's truth, not parity): of the targets whose source interval overlaps
    #    a query's by >= 2 kb, the path finds more than nine in ten (6 %-error ONT reads, chains need score >= 100)
    found = true_long = 0
    t_order = np.argsort(t.starts, kind="stable")
    ts_s, te_s = t.starts[t_order], t.ends[t_order]
    for i in range(0, q.n, 499):
        a = int(np.searchsorted(ts_s, q.starts[i] - 70000, side="left")); b = int(np.searchsorted(ts_s, q.ends[i], side="left"))
        ov = np.minimum(te_s[a:b], q.ends[i]) - np.maximum(ts_s[a:b], q.starts[i])
        n_long = int((ov >= 2000).sum())
        true_long += n_long; found += min(int(counts[i]), n_long)
    assert true_long > 500 and found > 0.9 * true_long, (found, true_long)
    #  * idempotence: the same call again
    c2, h2 = ix.overlap_twoset(Qd)
    assert np.array_equal(c2, counts) and np.array_equal(h2, has)
    ix.free()
    #  * the table's home-slot function never changes a result, whatever exponent option HT_POWER asks for (out-of-range
    #    values used to shift by a negative amount / overflow, i.e. an ordered table that misses present keys -- they are clamped now)
    for pw in ("0", "1", "2", "9", "11", "50"):
        ctx.set_option("HT_POWER", pw)
        ixp = engine.Index(ctx, Td, 0)
        ctx.set_option("HT_POWER", None)
        c_p, h_p = ixp.overlap_twoset(Sd)
        assert np.array_equal(c_p, counts[lo:hi]) and np.array_equal(h_p, has[lo:hi]), pw
        ixp.free()
    Qd.free(); Td.free(); Sd.free()

    n = 1024
    ixo = _oracle_index(oracle, t, 0)
    assert st["mid_occ"] == ixo.mid_occ and st["n_minimizers"] == ixo.n_minimizers and st["n_keys"] == ixo.n_keys
    s = q.slice(0, n)
    Qo = oracle.ReadSet(s.seqs(), s.names)
    (rc, ec, eh), (rc2, ec2, eh2) = _both_policies(oracle, ixo, lambda: ixo.twoset_counts(Qo, threads=THREADS))
    assert rc == 0 and rc2 == 0
    assert np.array_equal(counts[:n], ec) and np.array_equal(has[:n], eh)
    assert np.array_equal(ec, ec2), "tie policies differ on %d reads" % int((ec != ec2).sum())


@pytest.fixture(scope="module")
def c5_tenth():
    from lrge_amd import synth
    gsize, q, t = synth.make_config("c5_human_tenth")
    assert (q.n, t.n) == (10000, 200000)
    return gsize, q, t


@pytest.mark.parametrize("preset", [1, 0], ids=["ava-pb", "ava-ont"])
def test_c5_tenth_forward_and_inverse(ctx, oracle, c5_tenth, preset, knobs):
    """Forward: the first 256 queries against the full 3-Gbase target index.  Inverse (--use-min-ref): the index holds the
    queries, the targets are streamed -- for ava-pb all 200 000 of them (every indexed read's count is checked), for
    ava-ont the first 40 000 (the oracle streams them at ~5 k reads/s).  The forward run is forced into 8 index parts and
    3 streamed views with ava-pb (what full-size C5 needs on one GPU), so the partitioned paths are exercised at scale."""
    from lrge_amd import engine
    gsize, q, t = c5_tenth
    Qd, Td = _sets(ctx, q, t)
    # ---- forward ----
    if preset == 1:
        knobs.set("PART_BASES", str(int(t.lens().sum()) // 8 + 1))      # 8 index parts, as full-size C5 needs
        knobs.set("STREAM_BASES", str(int(q.lens().sum()) // 3 + 1))    # and the queries in 3 views
    Qd.presketch(preset)
    ix = engine.Index(ctx, Td, preset)
    counts, has = ix.overlap_twoset(Qd)
    st = ix.stats()
    ix.free()
    knobs.unset("PART_BASES"); knobs.unset("STREAM_BASES")
    ixo = _oracle_index(oracle, t, preset)
    assert st["mid_occ"] == ixo.mid_occ and st["n_minimizers"] == ixo.n_minimizers and st["n_keys"] == ixo.n_keys
    n = 256
    s = q.slice(0, n)
    Qo = oracle.ReadSet(s.seqs(), s.names)
    (rc, ec, eh), (rc2, ec2, eh2) = _both_policies(oracle, ixo, lambda: ixo.twoset_counts(Qo, threads=THREADS))
    assert rc == 0 and rc2 == 0
    assert np.array_equal(counts[:n], ec) and np.array_equal(has[:n], eh)
    assert np.array_equal(ec, ec2), "tie policies differ on %d reads" % int((ec != ec2).sum())
    del ixo
    # ---- inverse ----
    n_stream = t.n if preset == 1 else 40000
    ts = t if n_stream == t.n else t.slice(0, n_stream)
    if n_stream == t.n:
        Sd = Td
    else:
        qr, tr = engine.name_ranks(q.names, t.names)
        Sd = ctx.upload(ts.bases, ts.offsets, tr[:n_stream])
    Sd.presketch(preset)
    ixq = engine.Index(ctx, Qd, preset)
    inv = ixq.overlap_inverse(Sd)
    stq = ixq.stats()
    ixq.free()
    ixo = _oracle_index(oracle, q, preset)
    assert stq["mid_occ"] == ixo.mid_occ
    To = oracle.ReadSet(ts.seqs(), ts.names)
    (rc, einv), (rc2, einv2) = _both_policies(oracle, ixo, lambda: ixo.inverse_counts(To, threads=THREADS))
    assert rc == 0 and rc2 == 0
    assert np.array_equal(inv, einv)
    assert np.array_equal(einv, einv2), "tie policies differ on %d reads" % int((einv != einv2).sum())
    assert int(inv.sum()) > 1000
    if Sd is not Td:
        Sd.free()
    Qd.free(); Td.free()


@pytest.mark.parametrize("preset", [1, 0], ids=["ava-pb", "ava-ont"])
def test_c5_full(ctx, oracle, preset):
    """BASELINE configs[4] at FULL size on one GPU: H. sapiens-scale HiFi, -Q 100 000 -T 2 000 000 (31.5 Gbases), under BOTH
    presets SURVEY 8(d) asks for: ava-pb (the library's preset for PacBio, preset.rs:24-26) and ava-ont -- what the reference CLI
    really runs on a HiFi set, because main.rs:56-85 never forwards cli.rs:32-34's -P to the strategy builders (k = 15, no HPC:
    10.19 G index minimizers instead of 7.49 G, mid_occ 140, ~44 k anchors per query).  The reads are written straight into HBM by the device twin of the counter-based generator
    (lrge_amd/synth_cb.py; the host twin feeds the oracle the very same reads, tests/test_synth_cb.py).

    inverse (--use-min-ref, twoset.rs:370-584, what the reference itself picks at this size):
      * a range of 20 000 streamed targets against the oracle's index of all 100 000 queries: every count, mid_occ,
        n_keys, n_minimizers;
      * all 2 000 000 targets through the additivity of the strategy (twoset.rs:520-523: every streamed read adds one to
        the reads it hits): counts(all) = counts(before the range) + counts(range) + counts(after the range).
    forward (one index whatever the target size, aligner.rs:111-120 -- here 8 parts with global occurrence statistics):
      * n_minimizers / n_keys / mid_occ of the partitioned index against the oracle's (tests/golden/c5_full_index_stats.json,
        made by tests/golden/make_c5_fixture.py from the host twin's reads);
      * the counts of 256 queries spread over the set, and every PAF line of 8 of them, against the oracle's index RESTRICTED
        to those queries' keys (all 2 000 000 targets through the oracle's mm_sketch; oracle/c5_sample.py);
      * size-independent properties on all 100 000 queries: sub-range independence, no count above the number of truly
        overlapping targets, sensitivity, idempotence, and agreement with the (oracle-anchored) inverse counts.
    """
    import json
    from lrge_amd import engine, synth_cb
    spec, Q, T = synth_cb.spec_of("c5_human_twoset")
    assert (Q, T) == (100000, 2000000)
    pname = "ava-pb" if preset else "ava-ont"
    dq, dt = spec.device_reads(0, Q), spec.device_reads(Q, T)
    assert dq.total_bases + dt.total_bases > 31_000_000_000
    for dr, first in ((dq, 0), (dt, Q)):                         # the twins agree on this very set (both ends of each set)
        for lo in (0, dr.n - 300):
            h = spec.host_reads(first=first + lo, n=300)
            assert np.array_equal(h.bases, dr.to_host(lo, lo + 300))
    Qd = ctx.upload(dq.ptr, dq.offsets, dq.name_ranks())
    Td = ctx.upload(dt.ptr, dt.offsets, dt.name_ranks())
    qlens = dq.lens()
    q_start, q_end = dq.starts, dq.ends
    t_start, t_end = dt.starts, dt.ends
    dq.free(); dt.free()

    # ---- inverse ----
    ixq = engine.Index(ctx, Qd, preset)
    stq = ixq.stats()
    inv_all = ixq.overlap_inverse(Td)
    n_s = 20000
    lo = (T - n_s) // 2
    parts = []
    for a, b in ((0, lo), (lo, lo + n_s), (lo + n_s, T)):
        d = spec.device_reads(Q + a, b - a)
        Sd = ctx.upload(d.ptr, d.offsets, d.name_ranks())
        d.free()
        parts.append(ixq.overlap_inverse(Sd))
        Sd.free()
    ixq.free()
    assert np.array_equal(inv_all, parts[0] + parts[1] + parts[2])
    hq = spec.host_reads(first=0, n=Q)
    ixo = _oracle_index(oracle, hq, preset)
    assert stq["mid_occ"] == ixo.mid_occ and stq["n_minimizers"] == ixo.n_minimizers and stq["n_keys"] == ixo.n_keys
    ht = spec.host_reads(first=Q + lo, n=n_s)
    rc, einv = ixo.inverse_counts(oracle.ReadSet(ht.seqs(), ht.names), threads=THREADS)
    assert rc == 0
    assert np.array_equal(parts[1], einv)
    assert int(einv.sum()) > 10000
    del ixo, hq, ht

    # ---- forward ----
    ix = engine.Index(ctx, Td, preset)
    st = ix.stats()
    with open(os.path.join(os.path.dirname(__file__), "golden", "c5_full_index_stats.json")) as f:
        fx = json.load(f)
    import zlib
    chk = spec.host_reads(idx=[0, Q - 1, Q, Q + T // 2, Q + T - 1])
    assert fx["reads_crc32"] == "%08x" % (zlib.crc32(chk.bases.tobytes()) & 0xFFFFFFFF), "fixture made with another generator"
    assert {k: st[k] for k in ("n_minimizers", "n_keys", "mid_occ")} == {k: fx[pname][k] for k in ("n_minimizers", "n_keys", "mid_occ")}
    assert st["n_minimizers"] > 2**32          # more than one index part can hold
    counts, has = ix.overlap_twoset(Qd)
    #  * FORWARD counts against the oracle at full size: 256 queries spread over the whole set (every index
    #    part and every anchor batch).  No host holds the oracle's index of 7.5 G minimizers: all 2 000 000 targets stream through
    #    the oracle's mm_sketch into an index restricted to the keys of the sample -- complete position lists for exactly the keys
    #    mm_idx_get is asked about (oracle/c5_sample.py; tests/test_oracle_restricted.py checks that form against the full
    #    index where it fits); mid_occ from the oracle's KeyStats fixture compared above.  The PAF lines of 8 of them too
    #    (aligner.rs:244-291: every PafRecord field through the 3-part index).
    from lrge_amd import paf
    from oracle import c5_sample
    idx = c5_sample.sample_indices(Q, 256)
    fs = c5_sample.forward_sample(spec, Q, T, preset, idx, fx[pname]["mid_occ"], source="device", threads=THREADS)
    assert fs["n_minimizers_seen"] == st["n_minimizers"]
    assert np.array_equal(counts[idx], fs["counts"]), "forward counts differ from the oracle on %d of %d sampled queries" % (int((counts[idx] != fs["counts"]).sum()), len(idx))
    assert np.array_equal(has[idx], fs["has_mapping"])
    assert int(fs["counts"].sum()) > 3000
    ixo_r, hs = fs["index"], fs["reads"]
    ixo_r.opt.sort_mode = oracle.SORT_MM2           # (the other tie policy gives the same counts here too)
    rc, ec2, _ = ixo_r.twoset_counts(fs["sample"], threads=THREADS)
    ixo_r.opt.sort_mode = oracle.SORT_STABLE
    assert rc == 0 and np.array_equal(ec2, fs["counts"])
    n_paf = 8
    sub = hs.slice(0, n_paf)
    Pd = ctx.upload(sub.bases, sub.offsets, idx[:n_paf].astype(np.uint32))
    chains = ix.chains(Pd, dual=True)
    rl, ss, nk = ix.paf_stats(Pd)
    Pd.free()

    class _TNames:                                   # r%08d of the read's index in the whole job (queries first)
        def __getitem__(self, i):
            return b"r%08d" % (Q + int(i))
    tnames, tlens_all = _TNames(), np.diff(dt.offsets).astype(np.int64)
    sub_lens = [int(x) for x in sub.lens()]
    got = sorted(paf.paf_lines(chains, sub.names, sub_lens, tnames, tlens_all, rl, ss, nk))
    exp = []
    seqs = sub.seqs()
    for qi in range(n_paf):
        for r in ixo_r.map(seqs[qi], sub.names[qi]):
            ti = int(r["rid"])
            exp.append("\t".join([sub.names[qi].decode(), str(sub_lens[qi]), str(r["qs"]), str(r["qe"]), "-" if r["rev"] else "+",
                                  tnames[ti].decode(), str(int(tlens_all[ti])), str(r["rs"]), str(r["re"]), str(r["mlen"]), str(r["blen"]), "0",
                                  "tp:A:S", "cm:i:%d" % r["cnt"], "s1:i:%d" % r["score"], "dv:f:" + paf.format_dv(r["dv"]),
                                  "rl:i:%d" % r["rep_len"]]))
    exp.sort()
    assert len(exp) > 100 and len(got) == len(exp), (len(got), len(exp))
    bad = [i for i, (a_, b_) in enumerate(zip(got, exp)) if a_ != b_]
    assert not bad, "first differing PAF line:\n%s\n%s" % (got[bad[0]], exp[bad[0]])
    del fs, ixo_r
    #  * sub-range independence
    a, b = 40000, 42000
    d = spec.device_reads(a, b - a)
    Sd = ctx.upload(d.ptr, d.offsets, d.name_ranks())
    d.free()
    c_sub, h_sub = ix.overlap_twoset(Sd)
    Sd.free()
    assert np.array_equal(c_sub, counts[a:b]) and np.array_equal(h_sub, has[a:b])
    #  * no count above the number of targets whose source interval touches the query's; has_mapping <=> count > 0
    assert np.array_equal(has > 0, counts > 0)
    ts, te = np.sort(t_start), np.sort(t_end)
    n_touch = np.searchsorted(ts, q_end, side="left") - np.searchsorted(te, q_start, side="right")
    assert (counts <= np.maximum(n_touch, 0)).all()
    #  * sensitivity: of the targets overlapping a query's source interval by >= 2 kb, more than 95 % are found (HiFi)
    order = np.argsort(t_start, kind="stable")
    ts_s, te_s = t_start[order], t_end[order]
    found = true_long = 0
    for i in range(0, Q, 199):
        x = int(np.searchsorted(ts_s, q_start[i] - 30000, side="left")); y = int(np.searchsorted(ts_s, q_end[i], side="left"))
        ov = np.minimum(te_s[x:y], q_end[i]) - np.maximum(ts_s[x:y], q_start[i])
        n_long = int((ov >= 2000).sum())
        true_long += n_long; found += min(int(counts[i]), n_long)
    assert true_long > 3000 and found > 0.95 * true_long, (found, true_long)
    #  * the two strategies count the same overlaps from the two sides (different indexes, different mid_occ: not identical,
    #    but the inverse counts are anchored to the oracle above)
    diff = np.abs(counts.astype(np.int64) - inv_all.astype(np.int64))
    assert (diff <= 2).mean() > 0.97 and abs(int(counts.sum()) - int(inv_all.sum())) < 0.01 * int(inv_all.sum()), ((diff <= 2).mean(), int(counts.sum()), int(inv_all.sum()))
    #  * idempotence
    c2, h2 = ix.overlap_twoset(Qd)
    assert np.array_equal(c2, counts) and np.array_equal(h2, has)
    #  * the estimate lands on the genome size
    avg = np.float32(np.diff(dt.offsets).sum()) / np.float32(T)
    est = ctx.estimates(counts, qlens, float(avg), T, 100)
    med = engine.median(est, True, 0.15, 0.65)
    assert abs(float(med[1]) - spec.gsize) < 0.03 * spec.gsize
    ix.free(); Qd.free(); Td.free()
