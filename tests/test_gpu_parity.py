"""GPU: stage-by-stage parity of the HIP path (through the C ABI) against the CPU oracle.
Bit-exact for every integer stage; f32 estimates compared with tolerance 0 ulp."""
import numpy as np
import pytest

from conftest import to_arrays

pytestmark = pytest.mark.gpu

PRESETS = {"ont": 0, "pb": 1}
TANDEM_SELF = np.uint64((1 << 42) | (1 << 43))   # MM_SEED_TANDEM / MM_SEED_SELF: unused by chaining


def _upload(ctx, seqs, ranks=None):
    b, o = to_arrays(seqs)
    return ctx.upload(b, o, ranks)


def _both_sets(ctx, oracle, qseqs, qnames, tseqs, tnames, preset, dual=True):
    from lrge_amd import engine
    qr, tr = engine.name_ranks(qnames, tnames)
    Qd, Td = _upload(ctx, qseqs, qr), _upload(ctx, tseqs, tr)
    ixd = engine.Index(ctx, Td, PRESETS[preset])
    opt = oracle.make_opt(oracle.PRESET_AVA_PB if preset == "pb" else oracle.PRESET_AVA_ONT, dual=dual)
    To, Qo = oracle.ReadSet(tseqs, tnames), oracle.ReadSet(qseqs, qnames)
    ixo = oracle.Index(To, opt)
    return Qd, Td, ixd, Qo, To, ixo


@pytest.mark.parametrize("form", ["one-pass", "two-pass", "overflow-fallback", "ranged", "tile-form", "tile-form-ranged", "tile-form-overflow"])
@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_sketch_parity(ctx, oracle, edge_set, preset, form, knobs):
    # one-pass (per-chunk slots + compaction, the default), the two-pass form (count, scan, write), the fallback from
    # the first to the second when a chunk overflows its slot (forced here by a tiny slot capacity), and the ranged one-pass
    # form of sets whose slots do not fit at once (ranges of 256 chunks here: ~12 ranges, reads crossing their borders)
    if form == "two-pass":
        knobs.set("SKETCH_TWO_PASS", "1")
    elif form == "overflow-fallback":
        knobs.set("DEBUG_SK_CAP", "9")
    elif form == "ranged":
        knobs.set("DEBUG_SK_RANGE_CHUNKS", "256")
    # "tile-form": k_sketch_tile (a lane per step; option SKETCH_TILE_FORM) in place of k_sketch_direct (a lane per chunk) in the one-pass forms
    if form.startswith("tile-form"):
        knobs.set("SKETCH_TILE_FORM", "1")
        if form.endswith("ranged"):
            knobs.set("DEBUG_SK_RANGE_CHUNKS", "256")
        elif form.endswith("overflow"):
            knobs.set("DEBUG_SK_CAP", "9")
    qseqs, qnames, tseqs, tnames = edge_set
    seqs = tseqs + [b"", b"A", b"ACGTTGCA" * 3]            # empty and tiny reads keep their rid
    S = _upload(ctx, seqs)
    x, y = S.sketch(PRESETS[preset])
    k, hpc = (19, True) if preset == "pb" else (15, False)
    exp = [oracle.sketch(s, 5, k, rid=i, is_hpc=hpc) for i, s in enumerate(seqs) if len(s)]
    ex = np.concatenate([e["x"] for e in exp]); ey = np.concatenate([e["y"] for e in exp])
    assert len(x) == len(ex), "minimizer count differs: %d vs %d" % (len(x), len(ex))
    bad = np.nonzero((x != ex) | (y != ey))[0]
    assert bad.size == 0, "first mismatch at %d: read %d" % (bad[0], int(ey[bad[0]] >> 32))


def test_hpc_sketch_run_structure(ctx, oracle):
    """The homopolymer-compressed sketch finds the steps of mm_sketch's loop as bit masks per 32-base word (k_sketch.h,
    sketch_chunk_hpc): runs that cross word and 128-base chunk boundaries, runs of 255 / 256 / 700 bases (a span >= 256
    invalidates the k-mer), ambiguous bases at run boundaries and in a row, two-letter reads (nothing but long runs), and
    read lengths around the word and chunk sizes -- every minimizer against the oracle."""
    rng = np.random.Generator(np.random.PCG64(2024))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def runs(n_runs, max_run, alphabet=4, n_rate=0.0):
        out = []
        prev = -1
        for _ in range(n_runs):
            c = int(rng.integers(alphabet))
            if c == prev:
                c = (c + 1) % alphabet
            prev = c
            ln = int(rng.integers(1, max_run + 1)) if rng.random() < 0.3 else 1
            out.append(bytes([acgt[c]]) * ln)
            if n_rate and rng.random() < n_rate:
                out.append(b"N" * int(rng.integers(1, 4)))
                prev = -1
        return b"".join(out)

    seqs = []
    for ln in (18, 19, 23, 24, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 384, 385, 1023, 1024, 1025):
        seqs.append(bytes(acgt[rng.integers(4, size=ln)]))                     # lengths around words and chunks
    seqs += [runs(400, 6), runs(300, 40), runs(200, 140), runs(300, 12, alphabet=2), runs(500, 5, n_rate=0.05), runs(300, 70, n_rate=0.1)]
    g = bytes(acgt[rng.integers(4, size=4000)])
    for big in (254, 255, 256, 257, 700):                                       # the span limit, from both sides
        seqs.append(g[:900] + b"C" * big + g[900:1800])
    seqs.append(b"N" + g[:200] + b"NN" + b"A" * 130 + b"N" + b"A" * 130 + g[200:600] + b"N")
    seqs.append(b"G" * 127 + b"T" * 129 + b"G" * 128 + g[:500])               # runs that end on / just past chunk boundaries
    seqs.append((b"A" * 31 + b"C") * 40 + (b"G" * 32) + (b"T" * 33) + g[:300])  # run ends around word boundaries
    b, o = to_arrays(seqs)
    S = ctx.upload(b, o, None)
    x, y = S.sketch(PRESETS["pb"])
    exp = [oracle.sketch(s, 5, 19, rid=i, is_hpc=True) for i, s in enumerate(seqs) if len(s)]
    ex = np.concatenate([e["x"] for e in exp]); ey = np.concatenate([e["y"] for e in exp])
    assert len(x) == len(ex), "minimizer count differs: %d vs %d" % (len(x), len(ex))
    bad = np.nonzero((x != ex) | (y != ey))[0]
    assert bad.size == 0, "first mismatch at %d: read %d" % (bad[0], int(ey[bad[0]] >> 32))
    assert len(x) > 500


@pytest.mark.parametrize("form", ["one-pass", "ranged", "lane-form"])
@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_tile_sketch_edges(ctx, oracle, preset, form, knobs):
    """k_sketch_tile.h decides per step, from x[p-w+1 .. p+w], whether mm_sketch would write the step's minimizer out (the rule of
    tests/sketch_model.py).  Its edges: equal minima inside one window (low-complexity sequence: the two flush rules), ambiguous bases
    (the valid-step thresholds w+k-1 / w+k), read lengths around the tile (2048 bases) and chunk sizes, reads of a few bases (many
    segments per workgroup), and -- HPC -- homopolymer runs across tile edges, long enough that a tile's halo does not hold its w+k
    steps (those tiles go to k_sketch_redo, whose chunks must fit between tile-form neighbours without a minimizer lost or doubled)."""
    if form != "lane-form":                # (the same reads through the default kernel: its edges are the chunk's, 128 bases)
        knobs.set("SKETCH_TILE_FORM", "1")
    if form == "ranged":
        knobs.set("DEBUG_SK_RANGE_CHUNKS", "256")
    rng = np.random.Generator(np.random.PCG64(77))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    rnd = lambda n: bytes(acgt[rng.integers(4, size=n)])
    seqs = []
    for ln in (2047, 2048, 2049, 2048 + 19, 2048 + 23, 2048 + 24, 4095, 4096, 4097, 6000, 2048 * 3 + 5, 5, 0, 33, 14, 15, 19, 23, 24, 1):
        seqs.append(rnd(ln))
    for unit in (b"AC", b"ACG", b"AAC", b"ACGTAC", b"ACGTACGTT", b"A", b"ACACACACACACACACACACACACT"):     # equal minima in every window
        seqs.append((unit * 3000)[:5000])
        s = bytearray((unit * 3000)[:4500]); s[1000] = ord("N"); s[1023] = ord("N"); s[1024] = ord("N"); s[2047] = ord("N"); s[2048] = ord("G"); s[3000:3003] = b"NNN"
        seqs.append(bytes(s))
    g = rnd(30000)
    s = bytearray(g[:9000]); s[::53] = b"N" * len(s[::53]); seqs.append(bytes(s))                          # an ambiguous base every 53
    s = bytearray(g[:9000]); s[2040:2060] = b"N" * 20; s[4090:4097] = b"N" * 7; seqs.append(bytes(s))      # ... across tile edges
    for run in (40, 90, 100, 130, 300, 2100, 5000):                                                         # runs around / across tile edges
        seqs.append(g[:2000] + b"A" * run + g[2000:4200] + b"C" * run + g[4200:9000])
        seqs.append(g[:2048 - run // 2] + b"T" * run + g[3000:9000])
        seqs.append(g[:2048] + b"G" * run + g[5000:7000])
        seqs.append(b"C" * run + g[:3000] + b"A" * run)
    seqs.append((b"A" * 7 + b"C" * 5 + b"G" * 9 + b"T" * 3) * 400)                                        # long runs only: few steps per tile
    seqs.append(b"AC" * 20 + b"N" + b"AC" * 1500 + b"N" * 3 + b"CA" * 1200)
    seqs += [rnd(int(n)) for n in rng.integers(1, 400, size=200)]                                           # many reads per workgroup
    seqs += [rnd(int(n)) for n in rng.integers(1800, 2400, size=20)]
    S = _upload(ctx, seqs)
    x, y = S.sketch(PRESETS[preset])
    k, hpc = (19, True) if preset == "pb" else (15, False)
    exp = [oracle.sketch(s, 5, k, rid=i, is_hpc=hpc) for i, s in enumerate(seqs) if len(s)]
    ex = np.concatenate([e["x"] for e in exp]); ey = np.concatenate([e["y"] for e in exp])
    if len(x) != len(ex):
        n = min(len(x), len(ex)); bad = np.nonzero((x[:n] != ex[:n]) | (y[:n] != ey[:n]))[0]
        at = int(bad[0]) if bad.size else n
        assert False, "minimizer count differs: %d vs %d; first difference at %d: read %d pos %d (got read %d pos %d)" % (
            len(x), len(ex), at, int(ey[min(at, len(ey) - 1)] >> 32), int(ey[min(at, len(ey) - 1)] & 0xffffffff) >> 1,
            int(y[min(at, len(y) - 1)] >> 32), int(y[min(at, len(y) - 1)] & 0xffffffff) >> 1)
    bad = np.nonzero((x != ex) | (y != ey))[0]
    assert bad.size == 0, "first mismatch at %d: read %d pos %d" % (bad[0], int(ey[bad[0]] >> 32), int(ey[bad[0]] & 0xffffffff) >> 1)
    assert len(x) > 5000


@pytest.mark.parametrize("packed", [True, False])
@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_index_parity(ctx, oracle, edge_set, preset, packed, knobs):
    """Both index layouts: packed 8-byte entries (the default when they fit) and (hash, y) pairs."""
    if not packed:
        knobs.set("NO_PACKED_INDEX", "1")
    qseqs, qnames, tseqs, tnames = edge_set
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, tseqs, tnames, preset)
    st = ixd.stats()
    assert st["n_minimizers"] == ixo.n_minimizers
    assert st["n_keys"] == ixo.n_keys
    assert st["mid_occ"] == ixo.mid_occ
    keys, pos = ixd.dump()
    mz = ixo.minimizers()
    order = np.lexsort((mz["y"], mz["x"] >> np.uint64(8)))       # (hash, y) ascending == mm_idx_get lists
    assert np.array_equal(keys, (mz["x"] >> np.uint64(8))[order])
    assert np.array_equal(pos, mz["y"][order])


@pytest.mark.parametrize("slot_sort", [True, False])
def test_index_of_tiny_reads(ctx, oracle, slot_sort, knobs):
    """The packed index's sort reads the sketch's per-chunk slots in its first pass (no compaction in between: k_prims.h,
    rs_load_from_slots).  A tile of 4096 entries normally spans ~95 chunks, whose offsets it keeps in LDS; reads of a few dozen
    bases make chunks of one to five entries, i.e. tiles of a thousand chunks and more -- the general path (offsets from memory,
    one search per entry), mixed here with ordinary reads, empty reads and reads shorter than k.  Against the oracle, and against
    the compact-first form (NO_SLOT_SORT)."""
    from lrge_amd import engine
    if not slot_sort:
        knobs.set("NO_SLOT_SORT", "1")
    rng = np.random.Generator(np.random.PCG64(77))
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rng.integers(4, size=400_000)]
    seqs = []
    for _ in range(30000):                                      # tiny reads: 0-5 minimizers each
        st = int(rng.integers(0, len(genome) - 100)); ln = int(rng.integers(12, 46))
        seqs.append(genome[st:st + ln].tobytes())
    for _ in range(40):                                         # ordinary ones in between and behind
        st = int(rng.integers(0, len(genome) - 9000)); ln = int(rng.integers(2000, 9000))
        seqs.insert(int(rng.integers(0, len(seqs))), genome[st:st + ln].tobytes())
    seqs[5] = b""; seqs[77] = b"ACG"
    names = [b"t%06d" % i for i in range(len(seqs))]
    b, o = to_arrays(seqs)
    Td = ctx.upload(b, o, np.arange(len(seqs), dtype=np.uint32))
    ixd = engine.Index(ctx, Td, PRESETS["ont"])
    opt = oracle.make_opt(oracle.PRESET_AVA_ONT, dual=True)
    ixo = oracle.Index(oracle.ReadSet(seqs, names), opt)
    st = ixd.stats()
    assert st["n_minimizers"] == ixo.n_minimizers and st["n_minimizers"] > 50000
    assert st["n_keys"] == ixo.n_keys and st["mid_occ"] == ixo.mid_occ
    keys, pos = ixd.dump()
    mz = ixo.minimizers()
    order = np.lexsort((mz["y"], mz["x"] >> np.uint64(8)))
    assert np.array_equal(keys, (mz["x"] >> np.uint64(8))[order])
    assert np.array_equal(pos, mz["y"][order])
    ixd.free()


@pytest.mark.parametrize("form", ["two-pass", "overflow-fallback", "ranged"])
@pytest.mark.parametrize("preset,extra", [("ont", 0), ("pb", 6)])
def test_segw_entries_through_every_sketch_form(ctx, oracle, edge_set, preset, extra, form, knobs):
    """SEGW index entries (k_sketch.h PK == 2: a word + a u32 with the sort's first two digits, 12 bytes instead of the pair's 16) leave the
    sketch through the same forms as pairs do: the two-pass form (count, scan, k_sketch_write), the fallback to it when a chunk overflows
    its slot, and the ranged one-pass form (slots of a range of chunks at a time; what a full-size C5 part takes).  The index -- keys,
    position lists, statistics -- must be the oracle's in each."""
    knobs.set("NO_PACKED_INDEX", "1")
    knobs.set("SEG_PACK_MIN", "1")
    if extra:
        knobs.set("DEBUG_SEG_EXTRA", str(extra))
    knobs.set({"two-pass": "SKETCH_TWO_PASS", "overflow-fallback": "DEBUG_SK_CAP", "ranged": "DEBUG_SK_RANGE_CHUNKS"}[form],
              {"two-pass": "1", "overflow-fallback": "9", "ranged": "256"}[form])
    qseqs, qnames, tseqs, tnames = edge_set
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, tseqs, tnames, preset)
    keys, pos = ixd.dump()
    mz = ixo.minimizers()
    order = np.lexsort((mz["y"], mz["x"] >> np.uint64(8)))
    assert np.array_equal(keys, (mz["x"] >> np.uint64(8))[order]) and np.array_equal(pos, mz["y"][order])
    st = ixd.stats()
    assert st["mid_occ"] == ixo.mid_occ and st["n_keys"] == ixo.n_keys and st["n_minimizers"] == ixo.n_minimizers
    ixd.free()


@pytest.mark.parametrize("preset,extra", [("ont", 0), ("pb", 0), ("ont", 1), ("pb", 2), ("ont", 4), ("pb", 3), ("pb", 6), ("ont", 6), ("pb", 7), ("ont", 5)])
@pytest.mark.parametrize("entries", ["segw", "pairs"])
def test_segment_packed_pair_index_is_exact(ctx, oracle, edge_set, tiny_ont, tiny_hifi, preset, extra, entries, knobs):
    """The (hash, y) pair layout sorted in its segment-packed form (k_prims.h: index_sort_segpacked: the low hash byte first, then
    one packed word per entry inside its 256 segments) only engages above 4 M entries -- C5/10 and full-size C5 run it at scale.
    Forced here onto small sets: index entries, lists, mid_occ and counts must be those of the plain pair sort and of the oracle.
    extra > 0: the form full-size C5 takes in 3 parts (read ids of 20 bits: the word is 2 bits short) -- the top `extra` bits of
    the second hash byte are implied by the segment as well (pass A2, 256 << extra segments).  Round 5: the entry keeps the low bits
    of the hash's significance string, so the keys-only passes cut it in whole bytes whatever `extra` is -- ("pb", 6) is what
    full-size C5 runs now (24 bits left: three passes), 7 the most the second byte gives.
    entries: "segw" -- the sketch writes [word, digit pair] entries of 12 bytes for this sort (k_sketch.h PK == 2, index_sort_segw: the
    default since round 5) -- or "pairs" (option NO_SEGW: (hash, y) pairs of 16 bytes through index_sort_segpacked, rounds 3-4)."""
    from lrge_amd import engine
    if entries == "pairs":
        knobs.set("NO_SEGW", "1")
    knobs.set("NO_PACKED_INDEX", "1")
    knobs.set("SEG_PACK_MIN", "1")
    if extra:
        knobs.set("DEBUG_SEG_EXTRA", str(extra))
    qseqs, qnames, tseqs, tnames = edge_set
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, tseqs, tnames, preset)
    keys, pos = ixd.dump()
    mz = ixo.minimizers()
    order = np.lexsort((mz["y"], mz["x"] >> np.uint64(8)))
    assert np.array_equal(keys, (mz["x"] >> np.uint64(8))[order]) and np.array_equal(pos, mz["y"][order])
    st = ixd.stats()
    assert st["mid_occ"] == ixo.mid_occ and st["n_keys"] == ixo.n_keys
    ds = tiny_ont if preset == "ont" else tiny_hifi
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Q2, T2 = ctx.upload(ds.q.bases, ds.q.offsets, qr), ctx.upload(ds.t.bases, ds.t.offsets, tr)
    res = {}
    if extra:      # the extra pass really runs: one scatter launch more than the plain segment-packed build
        knobs.set("DEBUG_SEG_EXTRA", "0")
        c0 = ctx.counters()["rs_scatter_launches"]
        engine.Index(ctx, T2, PRESETS[preset]).free()
        c1 = ctx.counters()["rs_scatter_launches"]
        knobs.set("DEBUG_SEG_EXTRA", str(extra))
        engine.Index(ctx, T2, PRESETS[preset]).free()
        c2 = ctx.counters()["rs_scatter_launches"]
        nbits = 30 if preset == "ont" else 38
        launches = lambda e: 1 + (1 if e else 0) + (nbits - 8 - e + 7) // 8      # A [+ A2] + the LSD passes over what is left of the hash
        delta = launches(extra) - launches(0)
        assert (c2 - c1 == c1 - c0 + delta) or (c2 == c1 + delta), (c0, c1, c2, delta)
    for seg in (True, False):
        if not seg:
            knobs.set("NO_SEG_PACK", "1")
        ix = engine.Index(ctx, T2, PRESETS[preset])
        res[seg] = (ix.overlap_twoset(Q2), ix.overlap_twoset(Q2, remove_internal=True), ix.stats())
        ch = ix.chains(Q2)
        res[seg] += (np.sort(ch, order=["query", "target", "rev", "qs", "rs"]),)
        ix.free()
    for a, b in zip(res[True][:2], res[False][:2]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert res[True][2] == res[False][2] and np.array_equal(res[True][3], res[False][3]) and int(res[True][0][0].sum()) > 0


@pytest.mark.parametrize("preset,dual", [("ont", True), ("ont", False), ("pb", True)])
def test_anchor_parity(ctx, oracle, edge_set, preset, dual):
    qseqs, qnames, tseqs, tnames = edge_set
    if not dual:   # all-vs-all flags: queries are the targets themselves (self/diagonal, NO_DUAL)
        qseqs, qnames = tseqs, tnames
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, tseqs, tnames, preset, dual)
    checked = 0
    for q in range(len(qseqs)):
        if len(qseqs[q]) == 0:
            continue
        x, y = ixd.anchors(Qd, q, dual=dual)
        a = ixo.anchors(qseqs[q], qnames[q])
        assert len(x) == len(a), "query %d: %d anchors vs oracle %d" % (q, len(x), len(a))
        assert np.array_equal(x, a["x"]), "query %d: anchor x differs" % q
        assert np.array_equal(y & ~TANDEM_SELF, a["y"] & ~TANDEM_SELF), "query %d: anchor y differs" % q
        checked += len(x)
    assert checked > 1000


def _chain_rows(arr, fields):
    rows = np.stack([arr[f].astype(np.int64) for f in fields], axis=1)
    return rows[np.lexsort(rows.T[::-1])]


def _oracle_chains(ixo, qseqs, qnames):
    rows = []
    for q in range(len(qseqs)):
        if len(qseqs[q]) == 0:
            continue
        for r in ixo.map(qseqs[q], qnames[q]):
            rows.append((q, r["rid"], r["rev"], r["score"], r["cnt"], r["qs"], r["qe"], r["rs"], r["re"], r["mlen"], r["blen"]))
    a = np.array(rows, dtype=np.int64).reshape(-1, 11)
    return a[np.lexsort(a.T[::-1])]


@pytest.mark.parametrize("preset,dual", [("ont", True), ("ont", False), ("pb", True)])
def test_chain_parity(ctx, oracle, edge_set, preset, dual):
    qseqs, qnames, tseqs, tnames = edge_set
    if not dual:
        qseqs, qnames = tseqs, tnames
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, tseqs, tnames, preset, dual)
    got = _chain_rows(ixd.chains(Qd, dual=dual),
                      ["query", "target", "rev", "score", "cnt", "qs", "qe", "rs", "re", "mlen", "blen"])
    exp = _oracle_chains(ixo, qseqs, qnames)
    assert got.shape == exp.shape, "chain count %d vs oracle %d" % (len(got), len(exp))
    bad = np.nonzero((got != exp).any(axis=1))[0]
    assert bad.size == 0, "first differing chain: got %s expected %s" % (got[bad[0]], exp[bad[0]])
    assert len(exp) > 50


@pytest.mark.parametrize("platform,preset", [("ont", "ont"), ("hifi", "pb"), ("hifi", "ont")])
def test_twoset_counts_and_estimates(ctx, oracle, tiny_ont, tiny_hifi, platform, preset):
    from lrge_amd import engine
    ds = tiny_ont if platform == "ont" else tiny_hifi
    qseqs, tseqs = ds.q.seqs(), ds.t.seqs()
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, ds.q.names, tseqs, ds.t.names, preset)
    counts, has = ixd.overlap_twoset(Qd)
    rc, ecounts, ehas = ixo.twoset_counts(Qo, threads=8)
    assert rc == 0
    assert np.array_equal(counts, ecounts), "counts differ at %s" % np.nonzero(counts != ecounts)[0][:10]
    assert np.array_equal(has, ehas)
    assert counts.sum() > 0
    # -F (remove_internal) uses chain coordinates
    for ratio in (0.2, 0.05):
        c2, h2 = ixd.overlap_twoset(Qd, remove_internal=True, max_overhang_ratio=ratio)
        rc, ec2, eh2 = ixo.twoset_counts(Qo, remove_internal=True, ratio=ratio, threads=8)
        assert np.array_equal(c2, ec2) and np.array_equal(h2, eh2)
    # per-read estimates on the device: f32, 0 ulp
    lens = ds.q.lens()
    avg = np.float32(ds.t.lens().sum()) / np.float32(ds.t.n)
    est = ctx.estimates(counts, lens, float(avg), ds.t.n, 100)
    exp = np.array([oracle.per_read_estimate(int(l), float(avg), ds.t.n, int(c), 100) for l, c in zip(lens, counts)],
                   dtype=np.float32)
    assert np.array_equal(est.view(np.uint32), exp.view(np.uint32))
    assert engine.median(est, True, 0.15, 0.65) == oracle.median(exp, True, 0.15, 0.65)


def test_inverse_counts(ctx, oracle, tiny_ont):
    ds = tiny_ont
    # index = QUERY set, streamed = TARGET set (twoset.rs:596-599)
    Sd, Id, ixd, So, Io, ixo = _both_sets(ctx, oracle, ds.t.seqs(), ds.t.names, ds.q.seqs(), ds.q.names, "ont")
    for rem in (False, True):
        got = ixd.overlap_inverse(Sd, remove_internal=rem)
        rc, exp = ixo.inverse_counts(So, remove_internal=rem, threads=8)
        assert rc == 0 and np.array_equal(got, exp)
    assert got.sum() > 0


@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_ava_counts(ctx, oracle, preset):
    from lrge_amd import engine, synth
    g, reads, _ = synth.make_config("tiny_ava")
    seqs, names = reads.seqs(), list(reads.names)
    # shuffle the names so that rank order != index order (NO_DUAL works on names)
    rng = np.random.Generator(np.random.PCG64(11))
    names = [b"x%05d" % v for v in rng.permutation(len(names))]
    (ranks,) = engine.name_ranks(names)
    Rd = _upload(ctx, seqs, ranks)
    ixd = engine.Index(ctx, Rd, PRESETS[preset])
    opt = oracle.make_opt(oracle.PRESET_AVA_PB if preset == "pb" else oracle.PRESET_AVA_ONT, dual=False)
    Ro = oracle.ReadSet(seqs, names)
    ixo = oracle.Index(Ro, opt)
    for rem in (False, True):
        got = ixd.overlap_ava(remove_internal=rem)
        rc, exp = ixo.ava_counts(remove_internal=rem, threads=8)
        assert rc == 0 and np.array_equal(got, exp), np.nonzero(got != exp)[0][:10]
    assert got.sum() > 0 and got.sum() % 2 == 0


def test_error_paths(ctx, oracle):
    from lrge_amd import engine, synth, _ffi
    g, reads, _ = synth.make_config("tiny_ava", scale=0.2)
    seqs, names = reads.seqs(), list(reads.names)
    names[3] = names[1]
    (ranks,) = engine.name_ranks(names)
    Rd = _upload(ctx, seqs, ranks)
    ixd = engine.Index(ctx, Rd, 0)
    with pytest.raises(_ffi.LrgeHipError) as ei:
        ixd.overlap_ava()
    assert ei.value.code == _ffi.ERR_DUPLICATE_ID
    # empty query sequence -> MapError "Sequence is empty" (aligner.rs:214-216)
    Qd = _upload(ctx, [seqs[0], b"", seqs[2]])
    with pytest.raises(_ffi.LrgeHipError) as ei:
        ixd.overlap_twoset(Qd)
    assert ei.value.code == _ffi.ERR_MAP and "empty" in str(ei.value)
    # forward two-set tolerates duplicate target identifiers and counts a name once
    Q2 = _upload(ctx, seqs[:10], engine.name_ranks([b"q%d" % i for i in range(10)], names)[0])
    Rd2 = _upload(ctx, seqs, engine.name_ranks([b"q%d" % i for i in range(10)], names)[1])
    ix2 = engine.Index(ctx, Rd2, 0)
    got, has = ix2.overlap_twoset(Q2)
    opt = oracle.make_opt(oracle.PRESET_AVA_ONT, dual=True)
    ixo = oracle.Index(oracle.ReadSet(seqs, names), opt)
    rc, exp, ehas = ixo.twoset_counts(oracle.ReadSet(seqs[:10], [b"q%d" % i for i in range(10)]), threads=2)
    assert np.array_equal(got, exp) and np.array_equal(has, ehas)


def test_batching_is_invisible(ctx, oracle, tiny_ont, knobs):
    """Forcing many small anchor batches must not change the counts."""
    ds = tiny_ont
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, ds.q.seqs(), ds.q.names, ds.t.seqs(), ds.t.names, "ont")
    ref, _ = ixd.overlap_twoset(Qd)
    knobs.set("BATCH_ANCHORS", "20000")
    small, _ = ixd.overlap_twoset(Qd)
    assert ctx.counters()["batches"] > 3
    assert np.array_equal(ref, small)


@pytest.mark.parametrize("kernel", ["hw", "lpg", "lpg_full_scan", "lpg_notab", "lpg_redo", "auto64"])
@pytest.mark.parametrize("max_skip,max_iter", [(25, 5000), (100000, 5000), (100000, 40), (3, 90)])
def test_chain_kernels_and_slow_paths(ctx, oracle, tiny_hifi, knobs, kernel, max_skip, max_iter):
    """Both chain kernels (half-wave, lane-per-group) and their forced variants against the oracle, including the
    paths the default heuristics almost never take: no max_skip break (the candidate loop walks the
    whole 5000-bp window, far past the 64 anchors held in registers) and a tight max_iter clamp.
    HiFi reads give dense anchor groups (hundreds of anchors inside one window).
    k_chain_lpg stops an anchor's candidate scan once its score has reached what ANY remaining candidate could lift it to
    (round 4: max f + span over the anchors behind the scanned block; exact, see k_chain_lpg.h) -- "lpg" runs with that bound under
    every max_skip / max_iter setting, "lpg_full_scan" / "lpg_redo" without it (LPG_NO_PRUNE): the full loops and their slow paths."""
    if kernel == "auto64":     # the default split (big groups -> k_chain_hw, the rest -> k_chain_lpg) at a low threshold
        knobs.unset("CHAIN")
        knobs.set("LPG_MAX", "64")
    elif kernel == "lpg_redo":   # k_chain_lpg gives every group that touches a slow path to k_chain_hw_redo
        knobs.set("CHAIN", "lpg")
        knobs.set("LPG_SLOW_BUDGET", "0"); knobs.set("LPG_SLOW_RATE", "0"); knobs.set("LPG_SLOW_ENTRY_EVERY", "0")
        knobs.set("LPG_NO_PRUNE", "1")
    elif kernel == "lpg_full_scan":
        knobs.set("CHAIN", "lpg")
        knobs.set("LPG_NO_PRUNE", "1")
    elif kernel == "lpg_notab":  # k_chain_lpg with the f32 penalty computed per candidate instead of tabulated
        knobs.set("CHAIN", "lpg")
        knobs.set("LPG_NOTAB", "1")
    else:
        knobs.set("CHAIN", kernel)
    knobs.set("DEBUG_MAX_SKIP", str(max_skip))
    knobs.set("DEBUG_MAX_ITER", str(max_iter))
    ds = tiny_hifi
    qseqs, tseqs = ds.q.seqs()[:8], ds.t.seqs()
    qnames = ds.q.names[:8]
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, tseqs, ds.t.names, "ont")
    ixo.opt.max_chain_skip = max_skip
    ixo.opt.max_chain_iter = max_iter
    got = _chain_rows(ixd.chains(Qd, dual=True),
                      ["query", "target", "rev", "score", "cnt", "qs", "qe", "rs", "re", "mlen", "blen"])
    exp = _oracle_chains(ixo, qseqs, qnames)
    assert got.shape == exp.shape, "chain count %d vs oracle %d" % (len(got), len(exp))
    bad = np.nonzero((got != exp).any(axis=1))[0]
    assert bad.size == 0, "first differing chain: got %s expected %s" % (got[bad[0]], exp[bad[0]])
    assert len(exp) > 20
    counts, has = ixd.overlap_twoset(Qd)
    rc, ecounts, ehas = ixo.twoset_counts(Qo, threads=8)
    assert np.array_equal(counts, ecounts) and np.array_equal(has, ehas)


@pytest.mark.parametrize("mode", ["twoset", "ava"])
def test_packed_anchor_path_is_invisible(ctx, oracle, tiny_ont, knobs, mode):
    """Count-only runs sort packed 8-byte anchors (keys only) and unpack in the last radix pass; the
    (key, value) pair path (LRGE_HIP_NO_PACKED=1, also what chain records use) must give the same counts."""
    ds = tiny_ont
    if mode == "twoset":
        Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, ds.q.seqs(), ds.q.names, ds.t.seqs(), ds.t.names, "ont")
        run = lambda: ixd.overlap_twoset(Qd)[0]
    else:
        Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, ds.t.seqs(), ds.t.names, ds.t.seqs(), ds.t.names, "ont", False)
        run = lambda: ixd.overlap_ava()
    a = run()
    knobs.set("NO_PACKED", "1")
    b = run()
    assert np.array_equal(np.asarray(a), np.asarray(b))
    knobs.set("NO_PACKED", "0")
    knobs.set("NO_LOCAL_SORT", "1")        # packed anchors through the tiled global sort only
    d = run()
    assert np.array_equal(np.asarray(a), np.asarray(d))
    knobs.unset("NO_LOCAL_SORT")
    knobs.set("NO_PACKED", "1")
    knobs.set("NO_PACKED_INDEX", "1")      # and the (hash, y) pair index
    from lrge_amd import engine
    ix2 = engine.Index(ctx, Td, PRESETS["ont"])
    c = ix2.overlap_twoset(Qd)[0] if mode == "twoset" else ix2.overlap_ava()
    ix2.free()
    assert np.array_equal(np.asarray(a), np.asarray(c))
    assert int(np.asarray(a).sum()) > 0


@pytest.mark.parametrize("preset", ["ont", "pb"])
@pytest.mark.parametrize("mode", ["twoset", "twoset-F", "inverse", "ava"])
def test_dead_pair_filter_is_invisible(ctx, oracle, tiny_ont, tiny_hifi, edge_set, knobs, mode, preset):
    """Count-only runs drop, inside the expansion, the anchors of (target, strand) pairs that cannot reach the min_n anchors the
    group stage asks for (k_seed.h: k_expand_q -- a hashed unary count per query in LDS, an upper bound of every pair's size).  The
    counts must be those of the plain expansion (NO_GROUP_FILTER) and of the oracle, in every mode, with the survivors sorted in
    LDS and through the tiled global passes (their first pass gathers from the sparse slots), and in several small batches; and
    the filter must actually drop something."""
    ds = tiny_ont if preset == "ont" else tiny_hifi
    if mode == "ava":
        Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, ds.t.seqs(), ds.t.names, ds.t.seqs(), ds.t.names, preset, False)
        run = lambda: ixd.overlap_ava()
        rc, exp = ixo.ava_counts(threads=8)
    elif mode == "inverse":
        Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, ds.t.seqs(), ds.t.names, ds.q.seqs(), ds.q.names, preset)     # index = the queries
        run = lambda: ixd.overlap_inverse(Qd)
        rc, exp = ixo.inverse_counts(Qo, threads=8)
    else:
        F = mode.endswith("-F")
        Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, ds.q.seqs(), ds.q.names, ds.t.seqs(), ds.t.names, preset)
        run = lambda: ixd.overlap_twoset(Qd, remove_internal=F)[0]
        rc, exp, _ = ixo.twoset_counts(Qo, remove_internal=F, threads=8)
    assert rc == 0
    a = np.asarray(run())
    cn = ctx.counters()
    assert np.array_equal(a, exp) and (int(a.sum()) > 0 or mode == "twoset-F")
    assert 0 < cn["anchors_kept"] <= cn["anchors"], cn           # (that it drops a lot where there is a lot to drop: test_c2_full)
    knobs.set("NO_LOCAL_SORT", "1")               # survivors through the tiled passes
    assert np.array_equal(np.asarray(run()), a)
    knobs.unset("NO_LOCAL_SORT")
    for smax, mmax in ((0, 1 << 30), (0, 0), (300, 1500)):      # the kernel's size classes: all mid (512 threads, 2^16 buckets), all large (1024, 2^17), a mix
        knobs.set("DEBUG_EXPQ_SMALL_MAX", smax); knobs.set("DEBUG_EXPQ_MID_MAX", mmax)
        assert np.array_equal(np.asarray(run()), a)
        assert ctx.counters()["anchors_kept"] <= cn["anchors_kept"]       # (more buckets: fewer chance survivors)
    knobs.unset("DEBUG_EXPQ_SMALL_MAX"); knobs.unset("DEBUG_EXPQ_MID_MAX")
    knobs.set("BATCH_ANCHORS", "20000")            # several batches
    assert np.array_equal(np.asarray(run()), a) and ctx.counters()["batches"] > 1
    knobs.unset("BATCH_ANCHORS")
    knobs.set("NO_GROUP_FILTER", "1")
    b = np.asarray(run())
    cn2 = ctx.counters()
    assert np.array_equal(a, b)
    assert cn2["anchors_kept"] == cn2["anchors"] == cn["anchors"]
    assert cn2["groups"] >= cn["groups"] and cn2["groups_chained"] == cn["groups_chained"] and cn2["chain_anchors"] == cn["chain_anchors"]


@pytest.mark.parametrize("world", [2, 3])
def test_ava_and_inverse_shards_sum_to_the_whole(ctx, oracle, tiny_ont, world):
    """Multi-GPU decomposition on one GPU: the shards a world of `world` ranks would process, run one
    after the other, must add up to the unsharded counts -- all-vs-all (reads dealt round-robin in name-rank
    order, counts keyed by indexed read) and inverse two-set (streamed reads cut by bases)."""
    from lrge_amd import engine, parallel
    ds = tiny_ont
    seqs, names = ds.t.seqs(), ds.t.names
    ranks, _ = engine.name_ranks(names, names)
    Td = _upload(ctx, seqs, ranks)
    ix = engine.Index(ctx, Td, PRESETS["ont"])
    full = ix.overlap_ava()
    total = np.zeros_like(full)
    seen = []
    for r in range(world):
        idx = parallel.shard_by_rank_round_robin(ranks, r, world)
        seen += idx.tolist()
        shard = _upload(ctx, [seqs[i] for i in idx], np.asarray(ranks)[idx])
        total += ix.overlap_ava(shard=shard)
    assert sorted(seen) == list(range(len(names)))
    assert np.array_equal(total, full) and int(full.sum()) > 0
    # inverse: index over the queries, targets streamed in shards
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Qd = _upload(ctx, ds.q.seqs(), qr)
    ixq = engine.Index(ctx, Qd, PRESETS["ont"])
    Tall = _upload(ctx, seqs, tr)
    inv_full = ixq.overlap_inverse(Tall)
    b = parallel.shard_by_bases(ds.t.lens(), world)
    inv = np.zeros_like(inv_full)
    for r in range(world):
        lo, hi = b[r], b[r + 1]
        if hi > lo:
            inv += ixq.overlap_inverse(_upload(ctx, seqs[lo:hi], np.asarray(tr)[lo:hi]))
    assert np.array_equal(inv, inv_full) and int(inv_full.sum()) > 0
    ix.free(); ixq.free()


def test_qocc_precheck_is_invisible(ctx, oracle, tiny_ont, knobs):
    """The conservative bucket pre-check (k_qocc_check) only decides whether the exact sort-based filter runs;
    forcing the exact pass must not change anything."""
    ds = tiny_ont
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, ds.q.seqs(), ds.q.names, ds.t.seqs(), ds.t.names, "ont")
    ref, has = ixd.overlap_twoset(Qd)
    knobs.set("QOCC_EXACT", "1")
    exact, has2 = ixd.overlap_twoset(Qd)
    assert np.array_equal(ref, exact) and np.array_equal(has, has2)
    rc, ecounts, ehas = ixo.twoset_counts(Qo, threads=8)
    assert np.array_equal(ref, ecounts) and np.array_equal(has, ehas)


def test_query_occurrence_filter(ctx, oracle):
    """mm_seed_mz_flt in isolation: the query repeats a 60-mer 30 times; the targets cover that 60-mer
    only ~3x, so the index keeps it (count <= mid_occ) and only the QUERY-side filter (cnt > mid_occ and
    > 1% of the query's minimizers) removes its anchors."""
    from lrge_amd import synth
    genome = synth.random_genome(60_000, 777)
    t = synth.sample_reads(genome, 30, "ont", seed=9)
    motif = genome[10000:10060].tobytes()
    q_rep = motif * 30 + genome[20000:23000].tobytes()
    q_ctl = genome[9000:12000].tobytes()                 # same locus, no repeat: keeps its anchors
    qseqs, qnames = [q_rep, q_ctl], [b"qrep", b"qctl"]
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, t.seqs(), t.names, "ont")
    assert ixo.mid_occ == 10
    mz = oracle.sketch(q_rep, 5, 15)
    vals, cnts = np.unique(mz["x"], return_counts=True)
    assert (cnts > 10).any(), "the query must hold minimizer values repeated more than mid_occ times"
    rep_with_hits = sum(1 for v, c in zip(vals, cnts) if c > 10 and len(ixo.get(int(v) >> 8)) > 0)
    assert rep_with_hits > 0, "and some of them must have index hits, otherwise the filter is moot"
    for q in range(2):
        x, y = ixd.anchors(Qd, q, dual=True)
        a = ixo.anchors(qseqs[q], qnames[q])
        assert np.array_equal(x, a["x"]) and np.array_equal(y & ~TANDEM_SELF, a["y"] & ~TANDEM_SELF)
    # with the filter disabled the oracle finds strictly more anchors for the repeat query
    ixo.opt.q_occ_frac = 0.0
    assert len(ixo.anchors(q_rep, b"qrep")) > len(ixd.anchors(Qd, 0, dual=True)[0])
    ixo.opt.q_occ_frac = 0.01
    got = _chain_rows(ixd.chains(Qd, dual=True), ["query", "target", "rev", "score", "cnt", "qs", "qe", "rs", "re", "mlen", "blen"])
    assert np.array_equal(got, _oracle_chains(ixo, qseqs, qnames))


def test_cpp_host_mirror_cli(ctx, oracle, tmp_path):
    """The C++ Estimate/AvaStrategy/TwoSetStrategy mirror (include/lrge_hip.hpp) driven through the
    lrge-compatible CLI: all-vs-all over every read of the file is independent of the sampling RNG, so
    the printed estimate must equal the median of the oracle's per-read estimates bit for bit."""
    import subprocess
    from lrge_amd import build as B, synth
    from lrge_amd import ava as pyava
    cli = B.build_cli()
    g, reads, _ = synth.make_config("tiny_ava")
    fa = tmp_path / "reads.fa"
    with open(fa, "wb") as fh:
        for n, s in zip(reads.names, reads.seqs()):
            fh.write(b">" + n + b" some description\n" + s + b"\n")
    out = subprocess.run([cli, "-n", str(reads.n), "-f", "-s", "3", str(fa)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    got = np.float32(float(out.stdout.strip()))
    assert "Estimated genome size" in out.stderr and "Running all-vs-all strategy" in out.stderr
    # oracle expectation
    opt = oracle.make_opt(oracle.PRESET_AVA_ONT, dual=False)
    ixo = oracle.Index(oracle.ReadSet(reads.seqs(), reads.names), opt)
    rc, counts = ixo.ava_counts(threads=8)
    avg = np.float32(reads.lens().sum()) / np.float32(reads.n - 1)
    est = np.array([oracle.per_read_estimate(int(l), float(avg), reads.n - 1, int(c), 100) for l, c in zip(reads.lens(), counts)],
                   dtype=np.float32)
    lo, med, hi = oracle.median(est, True, 0.15, 0.65)
    assert got == med
    # the Python mirror agrees as well
    r = pyava.Builder().num_reads(reads.n).seed(3).build((reads.names, reads.seqs())).estimate(True, 0.15, 0.65)
    assert np.float32(r.estimate) == med and np.float32(r.lower) == lo and np.float32(r.upper) == hi
    assert r.no_mapping_count == int((counts == 0).sum())
    # -C/-D keep overlaps.paf: same multiset of lines as the Python PAF emitter (itself checked against the oracle)
    from lrge_amd import engine, paf
    out = subprocess.run([cli, "-n", str(reads.n), "-C", "-D", str(tmp_path), str(fa)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cli_lines = sorted(open(tmp_path / "overlaps.paf").read().splitlines())
    (ranks,) = engine.name_ranks(reads.names)
    Rd = ctx.upload(reads.bases, reads.offsets, ranks)
    ixd = engine.Index(ctx, Rd, 0)
    rl, ss, nk = ixd.paf_stats(Rd)
    py_lines = sorted(paf.paf_lines(ixd.chains(Rd, dual=False), reads.names, reads.lens(), reads.names, reads.lens(), rl, ss, nk))
    assert cli_lines == py_lines and len(py_lines) > 100
    # two-set through the CLI: runs and reports a plausible size; too few reads is the reference's error
    out = subprocess.run([cli, "-T", "150", "-Q", "50", "-s", "1", str(fa)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and 0.5 * g < float(out.stdout) < 2.0 * g, (out.stdout, out.stderr)
    out = subprocess.run([cli, "-T", "10", "-Q", "500", str(fa)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 1 and "is <= query number of reads" in out.stderr


@pytest.mark.parametrize("preset,dual", [("ont", True), ("ont", False), ("pb", True)])
def test_paf_lines(ctx, oracle, edge_set, tiny_hifi, preset, dual):
    """Every PafRecord field (incl. dv, rl) for every chain: HIP path + host dv == oracle, compared as
    the multiset of formatted PAF lines (mapping.rs serialisers)."""
    from lrge_amd import paf
    qseqs, qnames, tseqs, tnames = edge_set
    if preset == "pb":     # dense HiFi chains exercise dv > 0 with long seed spans
        qseqs, qnames = tiny_hifi.q.seqs()[:10] + qseqs[-10:], list(tiny_hifi.q.names[:10]) + qnames[-10:]
        tseqs, tnames = tiny_hifi.t.seqs() + tseqs[-10:], list(tiny_hifi.t.names) + tnames[-10:]
    if not dual:
        qseqs, qnames = tseqs, tnames
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, tseqs, tnames, preset, dual)
    chains = ixd.chains(Qd, dual=dual)
    rl, ss, nk = ixd.paf_stats(Qd)
    qlens = [len(s) for s in qseqs]; tlens = [len(s) for s in tseqs]
    got = sorted(paf.paf_lines(chains, qnames, qlens, tnames, tlens, rl, ss, nk))
    exp = []
    for q in range(len(qseqs)):
        if len(qseqs[q]) == 0:
            continue
        for r in ixo.map(qseqs[q], qnames[q]):
            t = int(r["rid"])
            exp.append("\t".join([qnames[q].decode(), str(qlens[q]), str(r["qs"]), str(r["qe"]), "-" if r["rev"] else "+",
                                  tnames[t].decode(), str(tlens[t]), str(r["rs"]), str(r["re"]), str(r["mlen"]), str(r["blen"]), "0",
                                  "tp:A:S", "cm:i:%d" % r["cnt"], "s1:i:%d" % r["score"], "dv:f:" + paf.format_dv(r["dv"]),
                                  "rl:i:%d" % r["rep_len"]]))
    exp.sort()
    assert len(got) == len(exp) and len(exp) > 50
    bad = [i for i, (a, b) in enumerate(zip(got, exp)) if a != b]
    assert not bad, "first differing PAF line:\n%s\n%s" % (got[bad[0]], exp[bad[0]])
    assert any("dv:f:0." in l for l in exp)                  # non-zero divergences are exercised
    assert any(not l.endswith("rl:i:0") for l in exp)        # and so is a non-zero rl


def test_cli_alignment_inputs_and_seeded_subsampling(ctx, oracle, tmp_path):
    """The reference's CLI integration tests (lrge/tests/alignment.rs:5-67) against the C++ host mirror, plus what they
    cannot assert: with --seed the sub-sample is the one rand 0.9.4 draws (include/lrge_rand.hpp vs oracle/rand09.py),
    the LAST T sampled indices are the targets, and the printed estimate is the oracle's on exactly those reads."""
    import os
    import subprocess
    from lrge_amd import build as B, readio, twoset
    from oracle import rand09
    cli = B.build_cli()
    sam = tmp_path / "two.sam"
    sam.write_bytes(b"@HD\tVN:1.6\tSO:unsorted\nREAD1\t4\t*\t0\t0\t*\t*\t0\t0\tGATTACA\t!!!!!!!\nREAD2\t4\t*\t0\t0\t*\t*\t0\t0\tGATTACA\t!!!!!!!\n")
    out = subprocess.run([cli, str(sam), "-T", "1", "-Q", "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "No finite estimates were generated" in out.stderr, out.stderr
    msam = tmp_path / "mapped.sam"
    msam.write_bytes(b"@HD\tVN:1.6\tSO:unsorted\n@SQ\tSN:chr1\tLN:1000\nREAD1\t0\tchr1\t1\t0\t7M\t*\t0\t0\tGATTACA\t!!!!!!!\n"
                     b"READ2\t4\t*\t0\t0\t*\t*\t0\t0\tGATTACA\t!!!!!!!\n")
    out = subprocess.run([cli, str(msam), "-T", "1", "-Q", "1"], capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "Mapped records are not supported" in out.stderr, out.stderr
    # the reference's toy.bam, rebuilt from its FASTA conversion (same reads, same order: the same indices are drawn)
    from conftest import write_unaligned_bam
    fa_names, fa_seqs = readio.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "toy_reads.fa.gz"))
    bam = str(tmp_path / "toy.bam")
    write_unaligned_bam(bam, fa_names, fa_seqs)
    out = subprocess.run([cli, bam, "-T", "10", "-Q", "5", "--seed", "6", "-f"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr                       # alignment.rs:52-67 asserts exactly this
    names, seqs = readio.load(bam)

    def expect(T, Q, seed, inverse=False, F=False):
        idx = rand09.unique_random_set(T + Q, len(names), seed)
        t, q = sorted(idx[Q:]), sorted(idx[:Q])                   # split_into_hashsets pops the targets off the end; file order
        To = oracle.ReadSet([seqs[i] for i in t], [names[i] for i in t])
        Qo = oracle.ReadSet([seqs[i] for i in q], [names[i] for i in q])
        if inverse:                                               # --use-min-ref: index the queries, stream the targets
            ixo = oracle.Index(Qo, oracle.make_opt(oracle.PRESET_AVA_ONT, dual=True))
            rc, counts = ixo.inverse_counts(To, remove_internal=F, threads=4)
        else:
            ixo = oracle.Index(To, oracle.make_opt(oracle.PRESET_AVA_ONT, dual=True))
            rc, counts, has = ixo.twoset_counts(Qo, remove_internal=F, threads=4)
        avg = np.float32(sum(len(seqs[i]) for i in t)) / np.float32(T)
        est = np.array([oracle.per_read_estimate(len(seqs[i]), float(avg), T, int(c), 100) for i, c in zip(q, counts)], dtype=np.float32)
        return oracle.median(est, True, 0.15, 0.65)

    lo, med, hi = expect(10, 5, 6)
    assert med is not None and np.float32(float(out.stdout.strip())) == med
    for T, Q, seed in ((200, 100, 42), (120, 60, 2**63 + 5)):
        lo, med, hi = expect(T, Q, seed)
        out = subprocess.run([cli, bam, "-T", str(T), "-Q", str(Q), "-s", str(seed), "-f"], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and np.float32(float(out.stdout.strip())) == med, (out.stdout, out.stderr)
        r = twoset.Builder().target_num_reads(T).query_num_reads(Q).seed(seed).build(bam).estimate(True, 0.15, 0.65)
        assert np.float32(r.estimate) == med and np.float32(r.lower) == lo and np.float32(r.upper) == hi
    # --use-min-ref (twoset.rs:596-599: the query set is indexed when it holds fewer bases) and -F, through both mirrors
    for T, Q, seed, inverse, F in ((300, 100, 9, True, False), (300, 100, 9, True, True), (200, 100, 42, False, True)):
        lo, med, hi = expect(T, Q, seed, inverse, F)
        args = [cli, bam, "-T", str(T), "-Q", str(Q), "-s", str(seed), "-f"] + (["--use-min-ref"] if inverse else []) + (["-F"] if F else [])
        out = subprocess.run(args, capture_output=True, text=True, timeout=120)
        assert out.returncode == 0 and np.float32(float(out.stdout.strip())) == med, (args, out.stdout, out.stderr)
        r = twoset.Builder().target_num_reads(T).query_num_reads(Q).seed(seed).use_min_ref(inverse).remove_internal(F, 0.2).build(bam) \
            .estimate(True, 0.15, 0.65)
        assert np.float32(r.estimate) == med and np.float32(r.lower) == lo and np.float32(r.upper) == hi


@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_ultra_long_reads(ctx, oracle, preset):
    """Reads of 100-600 kb (ultra-long ONT territory) on a genome with interspersed and tandem repeats: tens of
    thousands of minimizers per read, query segments far beyond the LDS-resident sort classes, (target, strand) groups
    with thousands of anchors, chains spanning hundreds of kilobases.  Counts, chains and all-vs-all counts must equal
    the oracle's."""
    from lrge_amd import engine, synth
    genome = synth.random_genome(900_000, 515, repeats=0.05, tandem=0.01)
    base = dict(synth.PLATFORMS["ont" if preset == "ont" else "hifi"])
    synth.PLATFORMS["_ul"] = dict(base, kind="normal", mu=300_000.0, sigma=120_000.0, lo=100_000, hi=600_000)
    try:
        ul = synth.sample_reads(genome, 10, "_ul", seed=31, name_prefix="ul")
    finally:
        del synth.PLATFORMS["_ul"]
    normal = synth.sample_reads(genome, 90, "ont" if preset == "ont" else "hifi", seed=33, name_prefix="n")
    qseqs, qnames = ul.seqs()[:4] + normal.seqs()[:25], ul.names[:4] + normal.names[:25]
    tseqs, tnames = ul.seqs()[4:] + normal.seqs()[25:], ul.names[4:] + normal.names[25:]
    assert max(map(len, qseqs)) > 200_000
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, tseqs, tnames, preset, dual=True)
    for F in (False, True):
        counts, has = ixd.overlap_twoset(Qd, remove_internal=F)
        rc, ec, eh = ixo.twoset_counts(Qo, remove_internal=F, threads=8)
        assert np.array_equal(counts, ec) and np.array_equal(has, eh)
        assert int(ec.sum()) > (30 if F else 100)
    cols = ["query", "target", "rev", "score", "cnt", "qs", "qe", "rs", "re", "mlen", "blen"]
    got = _chain_rows(ixd.chains(Qd, dual=True), cols)
    exp = _oracle_chains(ixo, qseqs, qnames)
    assert got.shape == exp.shape and np.array_equal(got, exp)
    assert int((exp[:, 6] - exp[:, 5]).max()) > 100_000            # a chain over more than 100 kb of query
    inv = ixd.overlap_inverse(Qd)
    rc, einv = ixo.inverse_counts(Qo, threads=8)
    assert np.array_equal(inv, einv)
    ixd.free()
    aseqs, anames = qseqs + tseqs, qnames + tnames
    (ar,) = engine.name_ranks(anames)
    Ad = _upload(ctx, aseqs, ar)
    ixa = engine.Index(ctx, Ad, PRESETS[preset])
    ixoa = oracle.Index(oracle.ReadSet(aseqs, anames), oracle.make_opt(oracle.PRESET_AVA_PB if preset == "pb" else oracle.PRESET_AVA_ONT, dual=False))
    rc, eava = ixoa.ava_counts(threads=8)
    assert np.array_equal(ixa.overlap_ava(), eava)
    ixa.free()


@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_presketch_is_invisible(ctx, oracle, tiny_ont, tiny_hifi, preset, knobs):
    """lrge_hip_seqset_presketch is a scheduling hint: the streamed set is sketched on the side stream during the
    index build and consumed by the next overlap call.  Same minimizers, hence the same counts / chains -- also when
    the hint goes unused (another preset, a second call, a set freed with the result pending) or is switched off."""
    from lrge_amd import engine
    ds = tiny_ont if preset == "ont" else tiny_hifi
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Qd, Td = ctx.upload(ds.q.bases, ds.q.offsets, qr), ctx.upload(ds.t.bases, ds.t.offsets, tr)
    ix0 = engine.Index(ctx, Td, PRESETS[preset])
    ref_counts, ref_has = ix0.overlap_twoset(Qd)
    ref_chains = _chain_rows(ix0.chains(Qd, dual=True), ["query", "target", "rev", "score", "cnt", "qs", "qe", "rs", "re"])
    ix0.free()
    assert int(ref_counts.sum()) > 0
    for variant in ("used", "twice", "other-preset", "off", "ava"):
        if variant == "off":
            knobs.set("NO_PRESKETCH", "1")
        if variant == "ava":
            Td.presketch(PRESETS[preset])                      # the indexed set itself is the streamed set
            ix = engine.Index(ctx, Td, PRESETS[preset])
            a = ix.overlap_ava()
            ix.free()
            ix = engine.Index(ctx, Td, PRESETS[preset])
            assert np.array_equal(a, ix.overlap_ava())
            ix.free()
            continue
        Qd.presketch(PRESETS["pb" if preset == "ont" else "ont"] if variant == "other-preset" else PRESETS[preset])
        ix = engine.Index(ctx, Td, PRESETS[preset])
        counts, has = ix.overlap_twoset(Qd)
        assert np.array_equal(counts, ref_counts) and np.array_equal(has, ref_has), variant
        if variant == "twice":                                  # consumed by the first call; the next ones sketch in line
            counts, has = ix.overlap_twoset(Qd)
            assert np.array_equal(counts, ref_counts)
        got = _chain_rows(ix.chains(Qd, dual=True), ["query", "target", "rev", "score", "cnt", "qs", "qe", "rs", "re"])
        assert np.array_equal(got, ref_chains), variant
        ix.free()
    knobs.unset("NO_PRESKETCH")
    # a pending result that nobody consumes: freed with its set
    Q2 = ctx.upload(ds.q.bases, ds.q.offsets, qr)
    Q2.presketch(PRESETS[preset])
    ix = engine.Index(ctx, Td, PRESETS[preset])
    Q2.free()
    counts, has = ix.overlap_twoset(Qd)
    assert np.array_equal(counts, ref_counts)
    ix.free()
    # a request that no index build picks up before its set is freed
    Q3 = ctx.upload(ds.q.bases, ds.q.offsets, qr)
    Q3.presketch(PRESETS[preset])
    Q3.free()
    ix = engine.Index(ctx, Td, PRESETS[preset])
    assert np.array_equal(ix.overlap_twoset(Qd)[0], ref_counts)
    ix.free()


def test_repeated_steps_are_deterministic(ctx, tiny_ont):
    """Sixty index-build + overlap steps on one context (pool memory recycled, the streamed set sketched on the side stream
    in two of three): every step returns the very same counts.  A race between the two streams or a buffer recycled too
    early would show up here as a flicker (the same loop ran 700 times on the C2 set without one)."""
    from lrge_amd import engine
    ds = tiny_ont
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Qd, Td = ctx.upload(ds.q.bases, ds.q.offsets, qr), ctx.upload(ds.t.bases, ds.t.offsets, tr)
    ref = {}
    for it in range(60):
        if it % 3 != 2:
            Qd.presketch(0)
        ix = engine.Index(ctx, Td, 0)
        F = it % 5 == 4
        counts, has = ix.overlap_twoset(Qd, remove_internal=F)
        mid = ix.stats()["mid_occ"]
        ix.free()
        if F not in ref:
            ref[F] = (counts.copy(), has.copy(), mid)
        assert np.array_equal(counts, ref[F][0]) and np.array_equal(has, ref[F][1]) and mid == ref[F][2], it
    assert int(ref[False][0].sum()) > 0


@pytest.mark.parametrize("data,preset", [("tiny_ont", "ont"), ("tiny_hifi", "pb"), ("repeats", "ont")])
def test_partitioned_index_is_exact(ctx, oracle, tiny_ont, tiny_hifi, data, preset, knobs):
    """A target set above LRGE_HIP_PART_BASES bases is indexed in parts (the 2^32-entry limits of one part; minimap2 is
    given batch_size = max and always builds ONE index, aligner.rs:112-122).  With the occurrence statistics taken over
    all parts the result must be that of the single index: same mid_occ, distinct-key and minimizer totals, the same
    number of anchors and the same counts with and without -F -- forced here with parts of a few reads each.  The
    repeat-rich set has a mid_occ above the floor of 10, i.e. one that the global histogram decides."""
    from lrge_amd import engine, synth, _ffi
    if data == "repeats":
        g, q, t = synth.make_config("c2_repeats", 0.05)
        ds = type("DS", (), {"q": q, "t": t})
    else:
        ds = tiny_ont if data == "tiny_ont" else tiny_hifi
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Qd, Td = ctx.upload(ds.q.bases, ds.q.offsets, qr), ctx.upload(ds.t.bases, ds.t.offsets, tr)
    ix = engine.Index(ctx, Td, PRESETS[preset])
    ref_stats = ix.stats()
    ref, ref_anchors = {}, {}
    for F in (False, True):
        ref[F] = ix.overlap_twoset(Qd, remove_internal=F)
        ref_anchors[F] = ctx.counters()["anchors"]
    ref_inv = ix.overlap_inverse(Qd)
    ref_ava = ix.overlap_ava()
    ref_chains = _chain_rows(ix.chains(Qd, dual=True), ["query", "target", "rev", "score", "cnt", "qs", "qe", "rs", "re", "mlen", "blen", "n_seeds"])
    ref_paf = ix.paf_stats(Qd)
    ix.free()
    assert int(ref[False][0].sum()) > 0 and int(ref_inv.sum()) > 0 and int(ref_ava.sum()) > 0 and len(ref_chains) > 0
    if data == "repeats":
        assert ref_stats["mid_occ"] > 10
    total = int(ds.t.lens().sum())
    for n_parts in (2, 3, 7, 25):
        knobs.set("PART_BASES", str(total // n_parts + 1))
        if n_parts == 3:                                         # (hash, y) pair layout: a part then gives its sorted hashes back
            knobs.set("NO_PACKED_INDEX", "1")
        else:
            knobs.unset("NO_PACKED_INDEX")
        Qd.presketch(PRESETS[preset])
        ixp = engine.Index(ctx, Td, PRESETS[preset])
        st = ixp.stats()
        assert (st["n_minimizers"], st["n_keys"], st["mid_occ"]) == (ref_stats["n_minimizers"], ref_stats["n_keys"], ref_stats["mid_occ"]), n_parts
        for F in (False, True):
            counts, has = ixp.overlap_twoset(Qd, remove_internal=F)
            assert ctx.counters()["anchors"] == ref_anchors[F], (n_parts, F)
            assert np.array_equal(counts, ref[F][0]) and np.array_equal(has, ref[F][1]), (n_parts, F)
        # the other entry points against the parts: inverse counts (keyed by indexed read), all-vs-all over the indexed
        # set itself (symmetric counts, both reads of a pair possibly in different parts), chain records with global
        # target ids, and the per-query seed statistics (rl / avg_k: occurrence counts over all parts)
        assert np.array_equal(ixp.overlap_inverse(Qd), ref_inv), n_parts
        if n_parts in (2, 7):
            assert np.array_equal(ixp.overlap_ava(), ref_ava), n_parts
            got = _chain_rows(ixp.chains(Qd, dual=True), ["query", "target", "rev", "score", "cnt", "qs", "qe", "rs", "re", "mlen", "blen", "n_seeds"])
            assert np.array_equal(got, ref_chains), n_parts
            for a, b in zip(ixp.paf_stats(Qd), ref_paf):
                assert np.array_equal(a, b), n_parts
        ixp.free()
    # the part size chosen by the library (as few parts as the entry limit and the memory allow; thresholds scaled down to
    # this set), with the first attempt failing as a part above 2^32 minimizers would: the build starts over with half-size parts
    knobs.unset("PART_BASES"); knobs.unset("NO_PACKED_INDEX")
    knobs.set("DEBUG_ONE_INDEX_BASES", str(total // 4)); knobs.set("DEBUG_PART_LIMIT_BASES", str(total // 2 + 1)); knobs.set("DEBUG_PART_FAIL_ATTEMPTS", "1")
    ixp = engine.Index(ctx, Td, PRESETS[preset])
    st = ixp.stats()
    assert (st["n_minimizers"], st["n_keys"], st["mid_occ"]) == (ref_stats["n_minimizers"], ref_stats["n_keys"], ref_stats["mid_occ"])
    counts, has = ixp.overlap_twoset(Qd)
    assert np.array_equal(counts, ref[False][0]) and np.array_equal(has, ref[False][1])
    assert ctx.counters()["batches"] >= 3                      # (one batch per part at least: the parts are there)
    ixp.free()
    for k_ in ("DEBUG_ONE_INDEX_BASES", "DEBUG_PART_LIMIT_BASES", "DEBUG_PART_FAIL_ATTEMPTS"):
        knobs.unset(k_)
    # the oracle agrees with the single index (and hence with the parts)
    opt = oracle.make_opt(oracle.PRESET_AVA_PB if preset == "pb" else oracle.PRESET_AVA_ONT, dual=True)
    ixo = oracle.Index(oracle.ReadSet(ds.t.seqs(), ds.t.names), opt)
    rc, ec, eh = ixo.twoset_counts(oracle.ReadSet(ds.q.seqs(), ds.q.names), threads=8)
    assert np.array_equal(ref[False][0], ec) and ixo.mid_occ == ref_stats["mid_occ"]


@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_streamed_set_in_views_is_exact(ctx, oracle, tiny_ont, tiny_hifi, preset, knobs):
    """A streamed set above LRGE_HIP_STREAM_BASES bases goes through in views (< 2^32 minimizers per pass): the queries of
    a two-set run, the targets of an inverse (--use-min-ref) run.  Streamed reads are independent, so the result must not
    change -- forced here with a few reads per view, alone and combined with a partitioned index."""
    from lrge_amd import engine
    ds = tiny_ont if preset == "ont" else tiny_hifi
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Qd, Td = ctx.upload(ds.q.bases, ds.q.offsets, qr), ctx.upload(ds.t.bases, ds.t.offsets, tr)
    ix = engine.Index(ctx, Td, PRESETS[preset])
    ref = {F: ix.overlap_twoset(Qd, remove_internal=F) for F in (False, True)}
    ix.free()
    ixq = engine.Index(ctx, Qd, PRESETS[preset])                  # inverse: index = queries, targets streamed
    ref_inv = {F: ixq.overlap_inverse(Td, remove_internal=F) for F in (False, True)}
    assert int(ref[False][0].sum()) > 0 and int(ref_inv[False].sum()) > 0
    qb, tb = int(ds.q.lens().sum()), int(ds.t.lens().sum())
    for n_views in (2, 5, 13):
        knobs.set("STREAM_BASES", str(tb // n_views + 1))
        for F in (False, True):
            assert np.array_equal(ixq.overlap_inverse(Td, remove_internal=F), ref_inv[F]), (n_views, F)
        knobs.set("STREAM_BASES", str(qb // n_views + 1))
        for part_bases in (None, tb // 3 + 1):
            if part_bases:
                knobs.set("PART_BASES", str(part_bases))
            ixt = engine.Index(ctx, Td, PRESETS[preset])
            for F in (False, True):
                counts, has = ixt.overlap_twoset(Qd, remove_internal=F)
                assert np.array_equal(counts, ref[F][0]) and np.array_equal(has, ref[F][1]), (n_views, part_bases, F)
            ixt.free()
            knobs.unset("PART_BASES")
    ixq.free()
    # the oracle agrees on the inverse counts
    opt = oracle.make_opt(oracle.PRESET_AVA_PB if preset == "pb" else oracle.PRESET_AVA_ONT, dual=True)
    ixo = oracle.Index(oracle.ReadSet(ds.q.seqs(), ds.q.names), opt)
    rc, einv = ixo.inverse_counts(oracle.ReadSet(ds.t.seqs(), ds.t.names), threads=4)
    assert np.array_equal(ref_inv[False], einv)


def test_all_vs_all_in_views_is_exact(ctx, tiny_ava, knobs):
    """ava.rs:165-366 has no size limit; here a read set above STREAM_BASES bases goes through lrge_hip_overlap_ava in views
    (each a shard of the reads: contributions add up), alone and against a partitioned index."""
    from lrge_amd import engine
    rb = tiny_ava
    (ranks,) = engine.name_ranks(rb.names)
    Rd = ctx.upload(rb.bases, rb.offsets, ranks)
    ix = engine.Index(ctx, Rd, 0)
    ref = {F: ix.overlap_ava(remove_internal=F) for F in (False, True)}
    ix.free()
    assert int(ref[False].sum()) > 0
    tb = int(rb.lens().sum())
    for n_views, part_bases in ((3, None), (7, None), (4, tb // 3 + 1)):
        knobs.set("STREAM_BASES", str(tb // n_views + 1))
        if part_bases:
            knobs.set("PART_BASES", str(part_bases))
        ixv = engine.Index(ctx, Rd, 0)
        for F in (False, True):
            assert np.array_equal(ixv.overlap_ava(remove_internal=F), ref[F]), (n_views, part_bases, F)
        ixv.free()
        knobs.unset("PART_BASES")
    Rd.free()


def test_batch_that_does_not_fit_is_taken_in_smaller_ones(ctx, tiny_ont, knobs):
    """The anchor batches are planned from the free HBM (48 B per anchor out of 4/5 of it, at most 2^31 anchors); when a batch's
    scratch does not fit after all -- here: the batch's allocator refuses requests above three quarters of one anchor array
    (DEBUG_BATCH_ALLOC_MAX_BYTES) -- the same queries are taken in batches of half the size, and the counts are those of the one
    batch."""
    from lrge_amd import engine
    ds = tiny_ont
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Qd, Td = ctx.upload(ds.q.bases, ds.q.offsets, qr), ctx.upload(ds.t.bases, ds.t.offsets, tr)
    ix = engine.Index(ctx, Td, 0)
    ref = ix.overlap_twoset(Qd)
    cn = ctx.counters()
    A = cn["anchors"]
    assert cn["batches"] == 1 and A > 1000
    knobs.set("DEBUG_BATCH_ALLOC_MAX_BYTES", str((A + 8) * 8 * 3 // 4))       # the largest single request of a batch is (A + 8) * 8 bytes
    got = ix.overlap_twoset(Qd)
    cn2 = ctx.counters()
    assert cn2["batches"] >= 2 and cn2["anchors"] == A
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    knobs.set("NO_BATCH_RETRY", "1")                       # without the retry the same call fails, loudly
    with pytest.raises(Exception, match="device allocation"):
        ix.overlap_twoset(Qd)
    knobs.unset("NO_BATCH_RETRY"); knobs.unset("DEBUG_BATCH_ALLOC_MAX_BYTES")
    assert np.array_equal(ix.overlap_twoset(Qd)[0], ref[0])
    ix.free()


def test_allocator_retry_leaves_no_stale_error(tiny_ont):
    """Under memory pressure the pool's first hipMalloc fails, the cache is trimmed and the retry succeeds -- the failed
    attempt must not surface later as the "last error" of a launch check (it did, at C5: the overlap call of a run whose
    8-part index had been built stopped with "out of memory").  The option DEBUG_ALLOC_FAIL_EVERY (lrge_hip_ctx_set_option) makes every third
    pool miss of that context take the path for real (an impossible request first); run in a child with a context of its own."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np
from lrge_amd import engine, synth
g, q, t = synth.make_config("tiny_twoset")
import sys
ctx = engine.Context(0)
if len(sys.argv) > 1:
    ctx.set_option("DEBUG_ALLOC_FAIL_EVERY", sys.argv[1])
qr, tr = engine.name_ranks(q.names, t.names)
Qd, Td = ctx.upload(q.bases, q.offsets, qr), ctx.upload(t.bases, t.offsets, tr)
out = []
for it in range(4):
    ix = engine.Index(ctx, Td, 0)
    out.append(ix.overlap_twoset(Qd)[0].copy())
    ix.free()
assert all(np.array_equal(out[0], o) for o in out)
print("SUM", int(out[0].sum()))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sums = []
    for extra in ([], ["3"]):
        env = dict(os.environ, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code] + extra, capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        sums.append([l for l in r.stdout.splitlines() if l.startswith("SUM")][0])
    assert sums[0] == sums[1] and int(sums[0].split()[1]) > 0


@pytest.mark.parametrize("form", ["wave", "wave_overflow", "chunk_slots"])
@pytest.mark.parametrize("layout", ["segw", "packed"])
@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_wave_dense_index_sketch_and_its_fallbacks(ctx, oracle, edge_set, tiny_ont, tiny_hifi, preset, layout, form, knobs):
    """Round 6: index entries out of the wave-dense sketch (k_sketch_wave: a wavefront fills its slot densely, in emission order; the
    sort's first pass reads the slots) -- SEGW entries and packed words -- must give the index of the oracle (entries by (hash, y), lists,
    mid_occ, key and minimizer counts) and its counts; so must the fallback a wavefront takes when its slot overflows (WAVE_CAP = 64:
    every wavefront with more than 64 minimizers raises the flag and the build takes the slot-per-chunk path) and the slot-per-chunk form
    itself (NO_WAVE_SKETCH).  The edge set holds reads of 1 .. k + w bases, ambiguity runs, homopolymers and identical reads."""
    from lrge_amd import engine
    if layout == "segw":
        knobs.set("NO_PACKED_INDEX", "1"); knobs.set("SEG_PACK_MIN", "1")
    if form == "wave_overflow":
        knobs.set("WAVE_CAP", "64")
    if form == "chunk_slots":
        knobs.set("NO_WAVE_SKETCH", "1")
    qseqs, qnames, tseqs, tnames = edge_set
    Qd, Td, ixd, Qo, To, ixo = _both_sets(ctx, oracle, qseqs, qnames, tseqs, tnames, preset)
    keys, pos = ixd.dump()
    mz = ixo.minimizers()
    order = np.lexsort((mz["y"], mz["x"] >> np.uint64(8)))
    assert np.array_equal(keys, (mz["x"] >> np.uint64(8))[order]) and np.array_equal(pos, mz["y"][order])
    st = ixd.stats()
    assert st["mid_occ"] == ixo.mid_occ and st["n_keys"] == ixo.n_keys and st["n_minimizers"] == ixo.n_minimizers
    ixd.free()
    ds = tiny_ont if preset == "ont" else tiny_hifi
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Q2, T2 = ctx.upload(ds.q.bases, ds.q.offsets, qr), ctx.upload(ds.t.bases, ds.t.offsets, tr)
    launches0 = ctx.counters().get("sketch_wave_launches", 0)
    ix = engine.Index(ctx, T2, PRESETS[preset])
    cb = ix.build_counters
    if form == "chunk_slots":
        assert cb.get("sketch_wave_launches", 0) == 0
    else:
        assert cb.get("sketch_wave_launches", 0) >= 1, cb          # the wave-dense kernel really ran (and, with WAVE_CAP = 64, overflowed)
    counts, has = ix.overlap_twoset(Q2)
    opt = oracle.make_opt(oracle.PRESET_AVA_PB if preset == "pb" else oracle.PRESET_AVA_ONT, dual=True)
    ixo2 = oracle.Index(oracle.ReadSet(ds.t.seqs(), ds.t.names), opt)
    rc, ec, eh = ixo2.twoset_counts(oracle.ReadSet(ds.q.seqs(), ds.q.names), threads=8)
    assert rc == 0 and np.array_equal(counts, ec) and np.array_equal(has, eh) and ix.stats()["mid_occ"] == ixo2.mid_occ
    ix.free()
