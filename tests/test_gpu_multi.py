"""GPU: the multi-GPU mode on one device.

  * lrge_hip_index_build_for on a single rank: an index restricted to the streamed set's minimizers answers exactly like the
    full index and reports the full index's statistics (both entry layouts, direct bitmap and Bloom key sets);
  * a whole world of ranks as THREADS of this process, one context each on the same GPU, joined by the library's local
    communicator (lrge_hip_comm_create_local): every rank builds its restricted index collectively, maps its own range of
    the queries, and the gathered estimates / reduced counts equal the single-GPU result;
  * the RCCL transport with world size 1 (all a 1-GPU box can offer): unique id, communicator, both collectives and a
    collective index build, in a child process.
Reference semantics: twoset.rs:266-334 (queries are independent given the index), ava.rs:300-301 / twoset.rs:520-523
(counts keyed by indexed read add up over the streamed reads).
"""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PRESETS = {"ont": 0, "pb": 1}


def _single(ctx, ds, preset, F=False):
    from lrge_amd import engine
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Qd, Td = ctx.upload(ds.q.bases, ds.q.offsets, qr), ctx.upload(ds.t.bases, ds.t.offsets, tr)
    ix = engine.Index(ctx, Td, preset)
    counts, has = ix.overlap_twoset(Qd, remove_internal=F)
    st = ix.stats()
    ix.free()
    return Qd, Td, counts, has, st


@pytest.mark.parametrize("form", ["fused", "sweeps", "fused-overflow"])
@pytest.mark.parametrize("layout", ["packed", "pairs"])
@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_restricted_index_single_rank(ctx, tiny_ont, tiny_hifi, preset, layout, form, knobs):
    """form: the key-set test inside the target sketch (default), the general form (full sketch, first sort pass, filter
    sweeps: RESTRICT_SWEEPS), and the fall-back from the first to the second when a chunk overflows its slot."""
    from lrge_amd import _ffi, engine
    ds = tiny_ont if preset == "ont" else tiny_hifi
    if layout == "pairs":
        knobs.set("NO_PACKED_INDEX", "1")
    if form == "sweeps":
        knobs.set("RESTRICT_SWEEPS", "1")
    else:
        knobs.set("RESTRICT_FUSED", "1")           # (the default picks by world size)
        if form == "fused-overflow":
            knobs.set("DEBUG_SK_CAP", "9")
    Qd, Td, counts, has, st = _single(ctx, ds, PRESETS[preset])
    assert int(counts.sum()) > 0
    for hint in (False, True):
        if hint:
            Qd.presketch(PRESETS[preset])          # an earlier hint is reused, not repeated
        ix = engine.Index(ctx, Td, PRESETS[preset], streamed=Qd)
        assert ix.stats() == st                    # mid_occ, distinct keys, minimizers of the WHOLE target set
        for F in (False, True):
            c, h = ix.overlap_twoset(Qd, remove_internal=F)
            if not F:
                assert np.array_equal(c, counts) and np.array_equal(h, has)
        with pytest.raises(_ffi.LrgeHipError):     # built for Qd: nothing else may be streamed against it
            ix.overlap_twoset(Td)
        ix.free()
    # the restricted index really is smaller: a streamed set of three reads keeps a fraction of the entries
    sub = ds.q.slice(0, 3)
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Sd = ctx.upload(sub.bases, sub.offsets, qr[:3])
    ix = engine.Index(ctx, Td, PRESETS[preset], streamed=Sd)
    c3, h3 = ix.overlap_twoset(Sd)
    assert np.array_equal(c3, counts[:3]) and np.array_equal(h3, has[:3]) and ix.stats() == st
    ix.free()


def _run_world(world, ds, preset, mode):
    """Every rank in its own thread with its own context; returns the per-rank results."""
    from lrge_amd import engine, parallel
    grp = parallel.LocalGroup(world)
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    out, errs, shard_stats = [None] * world, [], []
    _run_world.shard_stats = shard_stats
    bounds = parallel.shard_by_bases(ds.t.lens() if mode == "inverse" else ds.q.lens(), world)
    tb = parallel.shard_by_bases(ds.t.lens(), world)        # sharded target sketch: contiguous shares in rank order
    avg_t = np.float32(ds.t.lens().sum()) / np.float32(ds.t.n)

    def rank_main(r):
        try:
            c = engine.Context(0)
            for k_, v_ in getattr(_run_world, "ctx_opts", {}).items():      # (DEBUG_* options are never read from the environment)
                c.set_option(k_, v_)
            comm = grp.comm(c, r)
            lo, hi = bounds[r], bounds[r + 1]
            if mode in ("tshard", "tshard_q"):       # forward with the TARGETS sharded: every rank maps ALL queries against its share; counts add up
                t0, t1 = tb[r], tb[r + 1]
                tsub = ds.t.slice(t0, t1)
                Td = c.upload(tsub.bases, tsub.offsets, tr[t0:t1])
                Qd = c.upload(ds.q.bases, ds.q.offsets, qr)
                if mode == "tshard_q":  # ... and sketches only ITS share of them: the minimizers are all-gathered (round 6)
                    Qd.presketch_sharded(preset, comm)
                ix = engine.Index(c, Td, preset, comm=comm, tshard=True)
                shard_stats.append(ix.shard_stats)
                counts, has = ix.overlap_twoset(Qd)
                counts = comm.all_reduce_u32(counts); has = (comm.all_reduce_u32(has) > 0).astype(np.uint32)
                est = c.estimates(counts, ds.q.lens(), float(avg_t), ds.t.n, 100)
                out[r] = (counts, has, ix.stats(), est)
            elif mode in ("twoset", "sharded"):   # forward: targets indexed (restricted per rank), this rank's queries streamed
                sub = ds.q.slice(lo, hi)
                Qd = c.upload(sub.bases, sub.offsets, qr[lo:hi])
                if mode == "sharded":      # ... and every rank sketches only its own share of the targets
                    t0, t1 = tb[r], tb[r + 1]
                    tsub = ds.t.slice(t0, t1)
                    Td = c.upload(tsub.bases, tsub.offsets, tr[t0:t1])
                    ix = engine.Index(c, Td, preset, streamed=Qd, comm=comm, shard=(ds.t.lens(), tr, t0))
                    shard_stats.append(ix.shard_stats)
                else:
                    Td = c.upload(ds.t.bases, ds.t.offsets, tr)
                    ix = engine.Index(c, Td, preset, streamed=Qd, comm=comm)
                counts, has = ix.overlap_twoset(Qd)
                est = c.estimates(counts, sub.lens(), float(avg_t), ds.t.n, 100)
                lens = [bounds[i + 1] - bounds[i] for i in range(world)]
                allest = comm.all_gather_f32(est, max(lens), lens)
                out[r] = (counts, has, ix.stats(), allest)
            else:                      # inverse: queries indexed (replicated, small), this rank's targets streamed
                Qd = c.upload(ds.q.bases, ds.q.offsets, qr)
                sub = ds.t.slice(lo, hi)
                Sd = c.upload(sub.bases, sub.offsets, tr[lo:hi])
                ix = engine.Index(c, Qd, preset, streamed=Sd, comm=comm)
                part = ix.overlap_inverse(Sd)
                out[r] = (comm.all_reduce_u32(part), ix.stats())
            ix.free()
            comm.close()
            c.close()
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            errs.append((r, repr(e)))

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    grp.close()
    assert not errs, errs
    assert all(o is not None for o in out), "a rank did not finish"
    return out, bounds


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_world_of_threads_forward(ctx, oracle, tiny_ont, tiny_hifi, preset, world):
    from lrge_amd import engine
    ds = tiny_ont if preset == "ont" else tiny_hifi
    Qd, Td, counts, has, st = _single(ctx, ds, PRESETS[preset])
    avg_t = np.float32(ds.t.lens().sum()) / np.float32(ds.t.n)
    est = ctx.estimates(counts, ds.q.lens(), float(avg_t), ds.t.n, 100)
    out, bounds = _run_world(world, ds, PRESETS[preset], "twoset")
    for r, (c, h, s, allest) in enumerate(out):
        assert s == st, (r, s, st)                                      # every rank reports the global statistics
        assert np.array_equal(c, counts[bounds[r]:bounds[r + 1]]) and np.array_equal(h, has[bounds[r]:bounds[r + 1]])
        assert np.array_equal(allest.view(np.uint32), est.view(np.uint32))   # the gathered vector, in query order, on every rank
    # and the oracle agrees with the single-GPU run (hence with every rank)
    opt = oracle.make_opt(oracle.PRESET_AVA_PB if preset == "pb" else oracle.PRESET_AVA_ONT, dual=True)
    ixo = oracle.Index(oracle.ReadSet(ds.t.seqs(), ds.t.names), opt)
    rc, ec, eh = ixo.twoset_counts(oracle.ReadSet(ds.q.seqs(), ds.q.names), threads=8)
    assert np.array_equal(counts, ec) and ixo.mid_occ == st["mid_occ"]


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("layout", ["packed", "pairs"])
@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_world_of_threads_sharded_target_sketch(ctx, tiny_ont, tiny_hifi, preset, layout, world, monkeypatch):
    """lrge_hip_index_build_sharded: every rank sketches only its share of the targets; key sets, kept entries and owned
    hashes travel through the communicator (all-gather + two variable-size all-to-alls).  Worlds of 2, 3 and 8 threads on
    one GPU give the single-GPU counts, has_mapping, estimates and index statistics on every rank, both presets and both
    entry layouts; a starved Bloom filter (many false positives: more entries travel, none is missed) changes nothing."""
    ds = tiny_ont if preset == "ont" else tiny_hifi
    if layout == "pairs":
        monkeypatch.setenv("LRGE_HIP_NO_PACKED_INDEX", "1")       # (read when the ranks' contexts are created)
    if world == 3:
        monkeypatch.setenv("LRGE_HIP_SHARD_BLOOM_BITS", "1")
    Qd, Td, counts, has, st = _single(ctx, ds, PRESETS[preset])
    avg_t = np.float32(ds.t.lens().sum()) / np.float32(ds.t.n)
    est = ctx.estimates(counts, ds.q.lens(), float(avg_t), ds.t.n, 100)
    out, bounds = _run_world(world, ds, PRESETS[preset], "sharded")
    for r, (c, h, s, allest) in enumerate(out):
        assert s == st, (r, s, st)
        assert np.array_equal(c, counts[bounds[r]:bounds[r + 1]]) and np.array_equal(h, has[bounds[r]:bounds[r + 1]])
        assert np.array_equal(allest.view(np.uint32), est.view(np.uint32))
    ss = _run_world.shard_stats
    assert len(ss) == world and sum(x["entries_sketched"] for x in ss) == st["n_minimizers"]      # every target minimizer sketched exactly once
    assert sum(x["entries_sent"] for x in ss) == sum(x["entries_recv"] for x in ss) and sum(x["hashes_sent"] for x in ss) == sum(x["hashes_recv"] for x in ss)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("layout", ["packed", "pairs", "parts"])
@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_world_of_threads_target_sharded(ctx, tiny_ont, tiny_hifi, preset, layout, world, monkeypatch):
    """lrge_hip_index_build_tsharded (round 4): every rank indexes ITS share of the targets and maps ALL queries; the count vectors
    add up to the single-GPU ones (disjoint targets), has_mapping is their OR, and the index statistics every rank reports --
    n_minimizers, n_keys, mid_occ: what mm_idx_cal_max_occ sees -- are those of the ONE index over all targets.  Worlds of 2, 3
    and 8 threads on one GPU, both presets, both entry layouts, and local indexes that are themselves partitioned (parts of a
    rank's share: their keys reach the owners once per part)."""
    ds = tiny_ont if preset == "ont" else tiny_hifi
    if layout == "pairs":
        monkeypatch.setenv("LRGE_HIP_NO_PACKED_INDEX", "1")       # (read when the ranks' contexts are created)
    if layout == "parts":
        monkeypatch.setenv("LRGE_HIP_PART_BASES", str(int(ds.t.lens().sum()) // (world * 3)))
    Qd, Td, counts, has, st = _single(ctx, ds, PRESETS[preset])
    avg_t = np.float32(ds.t.lens().sum()) / np.float32(ds.t.n)
    est = ctx.estimates(counts, ds.q.lens(), float(avg_t), ds.t.n, 100)
    out, _ = _run_world(world, ds, PRESETS[preset], "tshard")
    for r, (c, h, s, e) in enumerate(out):
        assert s == st, (r, s, st)
        assert np.array_equal(c, counts) and np.array_equal(h, has)
        assert np.array_equal(e.view(np.uint32), est.view(np.uint32))
    ss = _run_world.shard_stats
    assert len(ss) == world and sum(x["hashes_sent"] for x in ss) == sum(x["hashes_recv"] for x in ss)


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("layout", ["packed", "parts", "pairs"])
@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_world_of_threads_target_sharded_with_sharded_query_sketch(ctx, tiny_ont, tiny_hifi, preset, layout, world, monkeypatch):
    """lrge_hip_seqset_presketch_sharded (round 6) in front of the target-sharded build: rank r sketches the r-th share of the queries,
    the minimizers are all-gathered, every rank maps ALL queries from that one sketch -- the same counts, has_mapping, estimates and
    statistics as the single-GPU run, on every rank; worlds of 2, 3 and 8 (8 ranks over 60 tiny queries: uneven and near-empty shares),
    both presets, one index and a partitioned one per rank; a minimizer travelling as ONE word (the default where it fits) and as an
    (x, y) pair.
    The exchange volumes add up: every minimizer of the set is sent by its one sketcher to the world - 1 others."""
    ds = tiny_ont if preset == "ont" else tiny_hifi
    if layout == "parts":
        monkeypatch.setenv("LRGE_HIP_PART_BASES", str(int(ds.t.lens().sum()) // (world * 3)))
    if layout == "pairs":
        monkeypatch.setenv("LRGE_HIP_QSHARD_PAIRS", "1")
    Qd, Td, counts, has, st = _single(ctx, ds, PRESETS[preset])
    qx, _ = Qd.sketch(PRESETS[preset])
    avg_t = np.float32(ds.t.lens().sum()) / np.float32(ds.t.n)
    est = ctx.estimates(counts, ds.q.lens(), float(avg_t), ds.t.n, 100)
    out, _ = _run_world(world, ds, PRESETS[preset], "tshard_q")
    for r, (c, h, s, e) in enumerate(out):
        assert s == st, (r, s, st)
        assert np.array_equal(c, counts) and np.array_equal(h, has)
        assert np.array_equal(e.view(np.uint32), est.view(np.uint32))
    ss = _run_world.shard_stats
    assert len(ss) == world and ss[0]["entry_bytes"] == (16 if layout == "pairs" else 8)
    assert sum(x["entries_recv"] for x in ss) == (world - 1) * len(qx) == sum(x["entries_sent"] for x in ss)


@pytest.mark.parametrize("stage", [20, 21, 22, "alloc"])
@pytest.mark.parametrize("bad_rank", [0, 2])
def test_a_failing_rank_fails_the_sharded_query_sketch(ctx, tiny_ont, stage, bad_rank):
    """The contract of every collective call here, for lrge_hip_seqset_presketch_sharded: a rank failing in front of its share's
    sketch, at the receive buffers or behind the all-gathers joins the next collective in its shape with the status word set -- over
    the strict host-callback transport and the local one every rank gets an error, nobody hangs, no mismatched collective."""
    from lrge_amd import _ffi, engine, parallel
    ds, world = tiny_ont, 3
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    for make in (lambda: parallel.ThreadHostGroup(world, timeout=60.0), lambda: parallel.LocalGroup(world)):
        grp = make()
        res = [None] * world

        def rank_main(r):
            c = engine.Context(0)
            comm = grp.comm(c, r)
            try:
                Qd = c.upload(ds.q.bases, ds.q.offsets, qr)
                if r == bad_rank:
                    if stage == "alloc":
                        c.set_option("DEBUG_ALLOC_FAIL_ALWAYS", "1")
                    else:
                        c.set_option("DEBUG_SHARD_FAIL_AT", str(stage))
                Qd.presketch_sharded(0, comm)
                res[r] = "sketched"
            except _ffi.LrgeHipError as e:
                res[r] = "error: %s" % e
            finally:
                c.set_option("DEBUG_ALLOC_FAIL_ALWAYS", None); c.set_option("DEBUG_SHARD_FAIL_AT", None)
                comm.close(); c.close()
        th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=180)
        assert all(not t.is_alive() for t in th), "a rank is still blocked in a collective"
        assert all(isinstance(x, str) and x.startswith("error") for x in res), res
        if isinstance(grp, parallel.ThreadHostGroup):
            assert not grp.faults, grp.faults
            assert grp.log[0] == grp.log[1] == grp.log[2], grp.log
        grp.close()


def test_target_sharded_global_mid_occ_matters(ctx, oracle, monkeypatch):
    """A repeat-rich target set cut over 3 ranks: a key is dropped by its count over ALL targets (the one index's mid_occ), not by
    its count in a rank's share -- the summed counts equal the oracle's, and the shares' own thresholds would not have done."""
    from lrge_amd import synth
    _, q, t = synth.make_config("c2_repeats", 0.04)

    class DS:
        pass
    ds = DS(); ds.q, ds.t = q, t
    Qd, Td, counts, has, st = _single(ctx, ds, 0)
    opt = oracle.make_opt(oracle.PRESET_AVA_ONT, dual=True)
    ixo = oracle.Index(oracle.ReadSet(t.seqs(), t.names), opt)
    rc, ec, eh = ixo.twoset_counts(oracle.ReadSet(q.seqs(), q.names), threads=8)
    assert rc == 0 and np.array_equal(counts, ec) and st["mid_occ"] == ixo.mid_occ
    out, _ = _run_world(3, ds, 0, "tshard")
    for c, h, s, e in out:
        assert s == st and np.array_equal(c, ec) and np.array_equal(h, eh)


def test_alltoallv_behind_the_abi(ctx):
    """lrge_hip_comm_alltoallv on host buffers, world of 3 threads: rank s sends s * 10 + d repeated (s + d + 1) times to rank d."""
    from lrge_amd import engine, parallel
    world = 3
    grp = parallel.LocalGroup(world)
    got, errs = [None] * world, []

    def rank_main(r):
        try:
            c = engine.Context(0)
            comm = grp.comm(c, r)
            counts = [r + d + 1 for d in range(world)]
            send = np.concatenate([np.full(n, r * 10 + d, dtype=np.uint64) for d, n in enumerate(counts)])
            recv, rc = comm.all_to_all_v(send, counts)
            got[r] = (recv.copy(), rc.copy())
            e32, _ = comm.all_to_all_v(send.astype(np.uint32), counts)       # another element size
            assert np.array_equal(e32, recv.astype(np.uint32))
            comm.close(); c.close()
        except Exception as e:      # noqa: BLE001
            errs.append((r, repr(e)))
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    grp.close()
    assert not errs, errs
    for d in range(world):
        recv, rc = got[d]
        assert rc.tolist() == [s + d + 1 for s in range(world)]
        assert recv.tolist() == [s * 10 + d for s in range(world) for _ in range(s + d + 1)]


def test_a_failing_rank_fails_the_collective_build(ctx, tiny_ont):
    """a rank whose allocation fails inside the collective index build must not leave the others blocked.  Rank 1's
    pool refuses every new segment (DEBUG_ALLOC_FAIL_ALWAYS); all ranks return an error, nobody hangs."""
    from lrge_amd import _ffi, engine, parallel
    ds = tiny_ont
    world = 2
    grp = parallel.LocalGroup(world)
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    bounds = parallel.shard_by_bases(ds.q.lens(), world)
    res = [None] * world

    def rank_main(r):
        c = engine.Context(0)
        comm = grp.comm(c, r)
        try:
            Td = c.upload(ds.t.bases, ds.t.offsets, tr)
            sub = ds.q.slice(bounds[r], bounds[r + 1])
            Qd = c.upload(sub.bases, sub.offsets, qr[bounds[r]:bounds[r + 1]])
            if r == 1:
                c.set_option("DEBUG_ALLOC_FAIL_ALWAYS", "1")
            engine.Index(c, Td, 0, streamed=Qd, comm=comm)
            res[r] = "built"
        except _ffi.LrgeHipError as e:
            res[r] = "error: %s" % e
        finally:
            c.set_option("DEBUG_ALLOC_FAIL_ALWAYS", None)
            comm.close(); c.close()
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert all(not t.is_alive() for t in th), "a rank is still blocked in the collective"
    grp.close()
    assert all(x is not None and x.startswith("error") for x in res), res


def _sharded_world_over(grp, world, ds, preset, fail=None, replicated=False):
    """The ranks of a sharded (or replicated-sketch) collective build as threads, joined by `grp`; fail = (rank, stage) injects a
    failure (DEBUG_SHARD_FAIL_AT) or (rank, "alloc") refuses every allocation of that rank.  -> per rank "error: ..." or the results."""
    from lrge_amd import _ffi, engine, parallel
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    bounds = parallel.shard_by_bases(ds.q.lens(), world)
    tb = parallel.shard_by_bases(ds.t.lens(), world)
    res = [None] * world

    def rank_main(r):
        c = engine.Context(0)
        comm = grp.comm(c, r)
        try:
            sub = ds.q.slice(bounds[r], bounds[r + 1])
            Qd = c.upload(sub.bases, sub.offsets, qr[bounds[r]:bounds[r + 1]])
            if replicated:
                Td = c.upload(ds.t.bases, ds.t.offsets, tr)
            else:
                tsub = ds.t.slice(tb[r], tb[r + 1])
                Td = c.upload(tsub.bases, tsub.offsets, tr[tb[r]:tb[r + 1]])
            if fail and fail[0] == r:
                if fail[1] == "alloc":
                    c.set_option("DEBUG_ALLOC_FAIL_ALWAYS", "1")
                else:
                    c.set_option("DEBUG_SHARD_FAIL_AT", str(fail[1]))
            if replicated:
                ix = engine.Index(c, Td, preset, streamed=Qd, comm=comm)
            else:
                ix = engine.Index(c, Td, preset, streamed=Qd, comm=comm, shard=(ds.t.lens(), tr, tb[r]))
            counts, has = ix.overlap_twoset(Qd)
            res[r] = (counts, has, ix.stats())
        except _ffi.LrgeHipError as e:
            res[r] = "error: %s" % e
        except Exception as e:      # noqa: BLE001
            res[r] = "exception: %r" % e
        finally:
            c.set_option("DEBUG_ALLOC_FAIL_ALWAYS", None); c.set_option("DEBUG_SHARD_FAIL_AT", None)
            comm.close(); c.close()
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    assert all(not t.is_alive() for t in th), "a rank is still blocked in a collective"
    return res, bounds


@pytest.mark.parametrize("preset", ["ont", "pb"])
def test_sharded_build_over_host_callbacks(ctx, tiny_ont, tiny_hifi, preset):
    """The sharded target sketch over the HOST-CALLBACK transport (a caller's own MPI / gloo: here threads behind a barrier that
    insists on matching shapes): the single-GPU counts, has_mapping and statistics on every rank, no mismatched collective."""
    from lrge_amd import parallel
    ds = tiny_ont if preset == "ont" else tiny_hifi
    Qd, Td, counts, has, st = _single(ctx, ds, PRESETS[preset])
    grp = parallel.ThreadHostGroup(3)
    res, bounds = _sharded_world_over(grp, 3, ds, PRESETS[preset])
    assert not grp.faults, grp.faults
    for r, x in enumerate(res):
        assert not isinstance(x, str), x
        c, h, s = x
        assert s == st and np.array_equal(c, counts[bounds[r]:bounds[r + 1]]) and np.array_equal(h, has[bounds[r]:bounds[r + 1]])
    assert grp.log[0] == grp.log[1] == grp.log[2] and len(grp.log[0]) >= 6


@pytest.mark.parametrize("replicated", [False, True])
@pytest.mark.parametrize("stage", [1, 2, 3, 4, 5, 6, 7, "alloc"])
@pytest.mark.parametrize("bad_rank", [0, 2])
def test_a_failing_rank_fails_the_sharded_build_on_every_transport(ctx, tiny_ont, stage, bad_rank, replicated):
    """a rank that fails inside lrge_hip_index_build_sharded -- at any stage: before the sizes all-reduce, in the
    sketches, the routing, the exchange buffers, behind the all-to-alls, in front of the statistics -- must JOIN the collective its
    peers enter next, in that collective's shape, with the status word set; never a one-word agreement against a (W + 1)-word
    all-reduce, never no collective at all.  Through the host-callback transport (which fails loudly on a shape mismatch and times
    out on a missing rank) and through the local transport: every rank gets an error, none hangs, no mismatch is seen; the
    sequence of collectives every rank went through is the same.  `replicated`: lrge_hip_index_build_for with a communicator
    (one collective: the statistics all-reduce)."""
    from lrge_amd import parallel
    if replicated and stage in (1, 2, 3, 4, 5, 7):
        pytest.skip("stage of the sharded build only")
    world = 3
    for make in (lambda: parallel.ThreadHostGroup(world, timeout=60.0), lambda: parallel.LocalGroup(world)):
        grp = make()
        res, _ = _sharded_world_over(grp, world, tiny_ont, 0, fail=(bad_rank, stage), replicated=replicated)
        assert all(isinstance(x, str) and x.startswith("error") for x in res), res
        if isinstance(grp, parallel.ThreadHostGroup):
            assert not grp.faults, grp.faults
            assert grp.log[0] == grp.log[1] == grp.log[2], grp.log
        grp.close()


@pytest.mark.parametrize("stage", [10, 11, 12, 13, 14, 15, 16, 17, "alloc"])
@pytest.mark.parametrize("bad_rank", [0, 2])
def test_a_failing_rank_fails_the_target_sharded_build(ctx, tiny_ont, stage, bad_rank):
    """The same contract for lrge_hip_index_build_tsharded: a rank failing before the local build, in the counting pass, at the
    exchange buffers, in front of the statistics, at the list of too-frequent keys or its gathering joins the next collective in
    its shape with the status word set -- over the strict host-callback transport and the local one every rank gets an error,
    nobody hangs, no mismatched collective, the same sequence of collectives on every rank.
    Stage 17: the re-taken list of too-frequent keys -- a rank whose own list is shorter than the longest one takes a
    larger block between C5 and A2; its failure must travel through A2 like every other one.  Clean tiny data has no too-frequent
    key and lists far below the first capacity, so every rank runs with the threshold forced to 1 and a first capacity of 1."""
    from lrge_amd import _ffi, engine, parallel
    ds, world = tiny_ont, 3
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    tb = parallel.shard_by_bases(ds.t.lens(), world)
    for make in (lambda: parallel.ThreadHostGroup(world, timeout=60.0), lambda: parallel.LocalGroup(world)):
        grp = make()
        res = [None] * world

        def rank_main(r):
            c = engine.Context(0)
            comm = grp.comm(c, r)
            try:
                tsub = ds.t.slice(tb[r], tb[r + 1])
                Td = c.upload(tsub.bases, tsub.offsets, tr[tb[r]:tb[r + 1]])
                if stage == 17:
                    c.set_option("DEBUG_TS_MID_OCC", "1"); c.set_option("DEBUG_TS_LIST_CAP", "1")
                if r == bad_rank:
                    if stage == "alloc":
                        c.set_option("DEBUG_ALLOC_FAIL_ALWAYS", "1")
                    else:
                        c.set_option("DEBUG_SHARD_FAIL_AT", str(stage))
                engine.Index(c, Td, 0, comm=comm, tshard=True)
                res[r] = "built"
            except _ffi.LrgeHipError as e:
                res[r] = "error: %s" % e
            finally:
                for o in ("DEBUG_ALLOC_FAIL_ALWAYS", "DEBUG_SHARD_FAIL_AT", "DEBUG_TS_MID_OCC", "DEBUG_TS_LIST_CAP"):
                    c.set_option(o, None)
                comm.close(); c.close()
        th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=180)
        assert all(not t.is_alive() for t in th), "a rank is still blocked in a collective"
        if stage == 15 and all(x == "built" for x in res):
            pass       # (stage 15 only exists when some key is too frequent: clean tiny data has none -- nothing was injected)
        else:
            assert all(isinstance(x, str) and x.startswith("error") for x in res), res
        if isinstance(grp, parallel.ThreadHostGroup):
            assert not grp.faults, grp.faults
            assert grp.log[0] == grp.log[1] == grp.log[2], grp.log
        grp.close()


def test_world_of_threads_inverse(ctx, tiny_ont):
    from lrge_amd import engine
    ds = tiny_ont
    qr, tr = engine.name_ranks(ds.q.names, ds.t.names)
    Qd, Td = ctx.upload(ds.q.bases, ds.q.offsets, qr), ctx.upload(ds.t.bases, ds.t.offsets, tr)
    ix = engine.Index(ctx, Qd, 0)
    ref = ix.overlap_inverse(Td)
    st = ix.stats()
    ix.free()
    out, _ = _run_world(3, ds, 0, "inverse")
    for total, s in out:
        assert np.array_equal(total, ref) and s == st


def _proc_rank(rank, world, port, q):
    """One PROCESS per rank, both on GPU 0, joined by torch.distributed/gloo through the library's host transport."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from lrge_amd import engine, parallel, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g, qs, ts = synth.make_config("tiny_twoset")
        qr, tr = engine.name_ranks(qs.names, ts.names)
        ctx = engine.Context(0)
        comm = parallel.HostComm(ctx, dist)

        def overlap_fn(lo, hi):
            sub = qs.slice(lo, hi)
            Td = ctx.upload(ts.bases, ts.offsets, tr)
            Qd = ctx.upload(sub.bases, sub.offsets, qr[lo:hi])
            ix = engine.Index(ctx, Td, 0, streamed=Qd, comm=comm)       # collective: all-reduce of the occurrence statistics
            counts, has = ix.overlap_twoset(Qd)
            st = ix.stats()
            avg = np.float32(ts.lens().sum()) / np.float32(ts.n)
            est = ctx.estimates(counts, sub.lens(), float(avg), ts.n, 100)
            ix.free()
            overlap_fn.out = (counts, st)
            return est, int((has == 0).sum())
        allv, no_map, (lo, hi) = parallel.twoset_forward_sharded(overlap_fn, qs.lens(), comm)
        q.put((rank, lo, hi, overlap_fn.out[0].tolist(), overlap_fn.out[1], allv.view(np.uint32).tolist(), no_map))
        comm.close(); ctx.close()
    finally:
        dist.destroy_process_group()


def test_two_processes_host_transport(ctx, tiny_ont):
    """World of two PROCESSES (both on this box's one GPU): the sharded forward job through lrge_amd.parallel with the
    library's host transport carried by gloo -- the same code path bench.py --gpus N takes, with RCCL swapped for TCP."""
    import socket
    import torch.multiprocessing as mp
    from lrge_amd import engine
    ds = tiny_ont
    Qd, Td, counts, has, st = _single(ctx, ds, 0)
    avg_t = np.float32(ds.t.lens().sum()) / np.float32(ds.t.n)
    est = ctx.estimates(counts, ds.q.lens(), float(avg_t), ds.t.n, 100)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_proc_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == ds.q.n
    for rank, lo, hi, c, s_, allv, no_map in res:
        assert c == counts[lo:hi].tolist() and s_ == st
        assert allv == est.view(np.uint32).tolist()
        assert no_map == int((has == 0).sum())


def test_rccl_transport_world1():
    """RCCL behind the C ABI on the one GPU of this box: unique id, communicator, all-reduce, all-gather and a collective
    index build with world size 1 (in a child: RCCL keeps process-wide state)."""
    code = r'''
import numpy as np
from lrge_amd import engine, parallel, synth
ctx = engine.Context(0)
uid = parallel.RcclComm.unique_id(ctx)
assert len(uid) == 128 and any(uid)
comm = parallel.RcclComm.create(ctx, 0, 1, uid)
a = np.arange(1000, dtype=np.uint32)
assert np.array_equal(comm.all_reduce_u32(a), a)
e = np.linspace(0, 1, 37).astype(np.float32)
assert np.array_equal(comm.all_gather_f32(e, 40, [37]), e)
g, q, t = synth.make_config("tiny_twoset")
qr, tr = engine.name_ranks(q.names, t.names)
Qd, Td = ctx.upload(q.bases, q.offsets, qr), ctx.upload(t.bases, t.offsets, tr)
ix0 = engine.Index(ctx, Td, 0); ref = ix0.overlap_twoset(Qd)[0]; st = ix0.stats(); ix0.free()
ix = engine.Index(ctx, Td, 0, streamed=Qd, comm=comm)
assert ix.stats() == st and np.array_equal(ix.overlap_twoset(Qd)[0], ref)
ix.free(); comm.close(); ctx.close()
print("RCCL-OK", int(ref.sum()))
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0 and "RCCL-OK" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_rccl_world1_forced_through_every_collective_shape():
    """no multi-GPU node is available, so at least every RCCL BRANCH of comm.h runs on this box -- with
    LRGE_HIP_RCCL_WORLD1=1 a world of one goes through librccl for every collective instead of the world-1 shortcuts: ncclAllReduce /
    ncclAllGather on device and staged host vectors, the send / receive groups of the variable-size all-to-all and all-gather (a rank's own
    share through a pair with itself), the agreements.  The three collective builds and the sharded query sketch give the one-GPU results
    through it; every injected failure stage (and a device that refuses every allocation: the staging path) fails the call without a
    hang and leaves a communicator that either still works or says it was aborted; lrge_hip_comm_abort makes every later collective
    fail at once, and a fresh communicator works again.  (In a child: RCCL keeps process-wide state.)"""
    code = r'''
import numpy as np
from lrge_amd import engine, parallel, synth, _ffi
ctx = engine.Context(0)
def fresh():
    return parallel.RcclComm.create(ctx, 0, 1, parallel.RcclComm.unique_id(ctx))
comm = fresh()
ops0 = comm.rccl_ops()
a = np.arange(1000, dtype=np.uint32)
assert np.array_equal(comm.all_reduce_u32(a), a)
e = np.linspace(0, 1, 37).astype(np.float32)
assert np.array_equal(comm.all_gather_f32(e, 40, [37]), e)
r, rc = comm.all_to_all_v(np.arange(77, dtype=np.uint64), [77])
assert np.array_equal(r, np.arange(77, dtype=np.uint64)) and list(rc) == [77]
assert comm.rccl_ops() >= ops0 + 3, comm.rccl_ops()
g, q, t = synth.make_config("tiny_twoset")
qr, tr = engine.name_ranks(q.names, t.names)
Qd, Td = ctx.upload(q.bases, q.offsets, qr), ctx.upload(t.bases, t.offsets, tr)
ix0 = engine.Index(ctx, Td, 0); ref = ix0.overlap_twoset(Qd)[0]; st = ix0.stats(); ix0.free()
def build(kind):
    if kind == "for":
        return engine.Index(ctx, Td, 0, streamed=Qd, comm=comm)
    if kind == "sharded":
        return engine.Index(ctx, Td, 0, streamed=Qd, comm=comm, shard=(t.lens(), tr, 0))
    Qd.presketch_sharded(0, comm)
    return engine.Index(ctx, Td, 0, comm=comm, tshard=True)
for kind in ("for", "sharded", "tshard"):
    o0 = comm.rccl_ops()
    ix = build(kind)
    assert ix.stats() == st and np.array_equal(ix.overlap_twoset(Qd)[0], ref), kind
    ix.free()
    assert comm.rccl_ops() > o0, (kind, "no librccl call was made")
# every failure stage of the three collective calls: an error, no hang; the communicator afterwards works or says it was aborted
stages = [(s_, "sharded") for s_ in (1, 2, 3, 4, 5, 6, 7)] + [(s_, "tshard") for s_ in (10, 11, 12, 13, 14, 15, 16, 17, 20, 21, 22)] + [("alloc", "tshard"), ("alloc", "sharded")]
n_err = 0
for stage, kind in stages:
    opts = {"DEBUG_ALLOC_FAIL_ALWAYS": "1"} if stage == "alloc" else {"DEBUG_SHARD_FAIL_AT": str(stage)}
    if stage in (15, 17):
        opts.update({"DEBUG_TS_MID_OCC": "1", "DEBUG_TS_LIST_CAP": "1"})
    for k_, v_ in opts.items():
        ctx.set_option(k_, v_)
    try:
        ix = build(kind); ix.free(); failed = False
    except _ffi.LrgeHipError:
        failed = True
    for k_ in opts:
        ctx.set_option(k_, None)
    assert failed or stage in (6,), (stage, kind)        # (stage 6 belongs to the replicated-sketch build)
    n_err += failed
    try:
        ix = build(kind)
        assert ix.stats() == st and np.array_equal(ix.overlap_twoset(Qd)[0], ref), (stage, kind)
        ix.free()
    except _ffi.LrgeHipError as ex:
        assert "abort" in str(ex).lower(), (stage, kind, str(ex))
        comm.close(); comm = fresh()
        ix = build(kind); assert np.array_equal(ix.overlap_twoset(Qd)[0], ref); ix.free()
assert n_err >= len(stages) - 1
comm.abort()
try:
    comm.all_reduce_u32(a); ok = False
except _ffi.LrgeHipError:
    ok = True
assert ok, "an aborted communicator still ran a collective"
comm.close()
comm = fresh()
assert np.array_equal(comm.all_reduce_u32(a), a)
comm.close(); ctx.close()
print("RCCL-FORCED-OK", n_err)
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, LRGE_HIP_RCCL_WORLD1="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0 and "RCCL-FORCED-OK" in r.stdout, (r.stdout[-800:], r.stderr[-3000:])


def test_bench_world2_on_one_gpu():
    """bench.py exactly as the driver launches it for N = 2 (torch.distributed.run, one process per rank), with both ranks
    on this box's one GPU and the collectives on the host transport (RCCL refuses two ranks on one device): the strong-
    scaling split, the collective index build and the all-gather must reproduce the one-GPU estimate bit for bit."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=root, LRGE_BENCH_SHARE_GPU="1", LRGE_BENCH_TRANSPORT="host")
    common = ["--config", "c2_bact_twoset", "--scale", "0.2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-from-host"]
    r1 = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r1.returncode == 0, r1.stderr[-2000:]
    one = json.loads(r1.stdout.strip().splitlines()[-1])
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29547", os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                        capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r2.returncode == 0, r2.stderr[-3000:]
    two = json.loads([l for l in r2.stdout.strip().splitlines() if l.startswith("{")][-1])
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "host" in two["config"]["collectives"]
    for k in ("genome_size_estimate", "estimate_q15_q65", "mid_occ"):
        assert one[k] == two[k], (k, one[k], two[k])
