"""CPU: host-side logic above the C ABI -- strategy builders, read splitting (reference tests
twoset.rs:659-701, lib.rs:206-266), read_id (io.rs:304-347), query sharding and the gloo collective."""
import os
import socket

import numpy as np
import pytest

from lrge_amd import parallel, readio, twoset, ava


def test_unique_random_set_properties():
    # lib.rs:206-266: size, range, uniqueness, repeatability with a seed, panic when k > n
    s = twoset.unique_random_set(10, 100, seed=42)
    assert len(s) == 10 and len(set(s.tolist())) == 10 and s.max() < 100
    assert np.array_equal(s, twoset.unique_random_set(10, 100, seed=42))
    assert not np.array_equal(s, twoset.unique_random_set(10, 100, seed=43))
    assert sorted(twoset.unique_random_set(7, 7, seed=1).tolist()) == list(range(7))
    with pytest.raises(ValueError):
        twoset.unique_random_set(11, 10)


def test_split_into_sets_sizes():
    # twoset.rs:659-701
    a, b = twoset.split_into_sets([1, 2, 3, 4, 5], 2)
    assert a == {4, 5} and b == {1, 2, 3}            # the LAST sampled indices become the first set
    a, b = twoset.split_into_sets([1, 2, 3], 5)
    assert a == {1, 2, 3} and b == set()
    a, b = twoset.split_into_sets([], 2)
    assert a == set() and b == set()
    a, b = twoset.split_into_sets([1, 2, 3], 0)
    assert a == set() and b == {1, 2, 3}


def test_read_id_whitespace_rule():
    # io.rs:304-347
    assert readio.read_id(b"read1 desc more") == b"read1"
    assert readio.read_id(b"read1\tdesc") == b"read1"
    assert readio.read_id(b"read1") == b"read1"
    assert readio.read_id(b"") == b""


def test_fastx_parsing(tmp_path):
    fa = tmp_path / "x.fa"
    fa.write_bytes(b">r1 d\nACGT\nAC\n>r2\nGG\n")
    assert list(readio.iter_records(str(fa))) == [(b"r1", b"ACGTAC"), (b"r2", b"GG")]
    fq = tmp_path / "x.fq"
    fq.write_bytes(b"@r1 d\nACGT\n+\nIIII\n@r2\nGG\n+\nII\n")
    assert list(readio.iter_records(str(fq))) == [(b"r1", b"ACGT"), (b"r2", b"GG")]
    import gzip
    gz = tmp_path / "x.fa.gz"
    gz.write_bytes(gzip.compress(b">a\nAC\n"))
    assert list(readio.iter_records(str(gz))) == [(b"a", b"AC")]


def _reads(n):
    return [b"r%d" % i for i in range(n)], [b"ACGT" * (5 + i % 3) for i in range(n)]


def test_compressed_input_sniffing(tmp_path):
    """gzip / bzip2 / xz are recognised by their magic bytes, whatever the file is called (io.rs)."""
    import bz2, gzip, lzma
    from lrge_amd import readio
    fa = b">r1 desc\nACGT\nAC\n>r2\nGGTT\n"
    for name, comp in (("a.dat", gzip.compress), ("b.dat", bz2.compress), ("c.dat", lzma.compress), ("d.dat", lambda x: x)):
        p = tmp_path / name
        p.write_bytes(comp(fa))
        names, seqs = readio.load(str(p))
        assert names == [b"r1", b"r2"] and seqs == [b"ACGTAC", b"GGTT"]
    z = tmp_path / "e.zst"
    z.write_bytes(b"\x28\xb5\x2f\xfd" + b"\x00" * 8)
    with pytest.raises(ValueError):
        readio.load(str(z))


def test_twoset_split_guards():
    # twoset.rs:137-151: n <= Q -> TooFewReads; n < T+Q -> T = n - Q
    from lrge_amd import LrgeError
    s = twoset.Builder().target_num_reads(10).query_num_reads(5).seed(1).build(_reads(5))
    with pytest.raises(LrgeError) as ei:
        s.split_fastq()
    assert ei.value.kind == "TooFewReadsError"
    s = twoset.Builder().target_num_reads(10).query_num_reads(5).seed(1).build(_reads(12))
    (tn, ts), (qn, qs), avg = s.split_fastq()
    assert s.target_num_reads == 7 and len(tn) == 7 and len(qn) == 5 and not set(tn) & set(qn)
    assert avg == np.float32(sum(map(len, ts))) / np.float32(7)
    s = twoset.Builder().target_num_reads(4).query_num_reads(3).seed(7).build(_reads(50))
    (tn, ts), (qn, qs), avg = s.split_fastq()
    assert len(tn) == 4 and len(qn) == 3
    idx = [int(x[1:]) for x in tn]
    assert idx == sorted(idx)                      # written in file order (iter_records)


def test_builder_defaults_and_ava_clamp():
    b = twoset.Builder()
    assert (b._t, b._q, b._ratio, b._threads, b._use_min_ref) == (10_000, 5_000, 0.2, 1, False)   # twoset/builder.rs:22-39
    assert b.remove_internal(False, 0.5)._ratio == 0.2       # ratio only overridden when the flag is on (builder.rs:90-96)
    assert b.remove_internal(True, 0.5)._ratio == 0.5
    assert ava.Builder()._n == 25_000                        # ava.rs:62
    s = ava.Builder().num_reads(100).seed(3).build(_reads(30))
    rn, rs, sum_len = s.subsample_reads()
    assert s.num_reads == 30 and len(rn) == 30 and sum_len == sum(map(len, rs))   # ava.rs:122-128


def test_shard_by_bases():
    lens = np.array([10, 10, 10, 10, 40, 10, 10], dtype=np.int64)
    b = parallel.shard_by_bases(lens, 2)
    assert b[0] == 0 and b[-1] == 7 and len(b) == 3
    left = lens[:b[1]].sum()
    assert abs(left - lens.sum() / 2) <= 40
    b8 = parallel.shard_by_bases(lens, 8)
    assert b8 == sorted(b8) and b8[0] == 0 and b8[-1] == 7 and len(b8) == 9
    assert parallel.shard_by_bases([], 4) == [0, 0, 0, 0, 0]
    rr = [parallel.shard_round_robin(10, r, 3).tolist() for r in range(3)]
    assert sorted(sum(rr, [])) == list(range(10))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lens = np.arange(1, 12) * 100
        full = (np.arange(11, dtype=np.float32) + 0.5) * 1000

        def overlap_fn(lo, hi):                       # stands in for the per-rank GPU call
            return full[lo:hi], int(rank == 0)
        comm = parallel.TorchComm(dist)
        assert (comm.rank, comm.world) == (rank, world)
        allv, no_map, (lo, hi) = parallel.twoset_forward_sharded(overlap_fn, lens, comm)
        counts = np.zeros(5, dtype=np.uint32); counts[rank] = 3; counts[4] = 1
        tot = comm.all_reduce_u32(counts)
        # all-vs-all / inverse drivers: the per-rank GPU call is stood in for by "one count per read of the shard"
        ranks7 = np.array([4, 0, 6, 2, 5, 1, 3])
        ava, idx = parallel.ava_sharded(lambda ix: np.bincount(ix, minlength=7), ranks7, comm)
        inv, (ilo, ihi) = parallel.inverse_sharded(lambda a, b: np.bincount(np.arange(a, b) % 3, minlength=3), lens, comm)
        q.put((rank, allv.tolist(), no_map, lo, hi, tot.tolist(), ava.tolist(), idx.tolist(), inv.tolist()))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_gather_and_allreduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    full = ((np.arange(11, dtype=np.float32) + 0.5) * 1000).tolist()
    ranges = []
    shards = []
    for rank, allv, no_map, lo, hi, tot, ava, idx, inv in sorted(res):
        assert allv == full                      # every rank ends with the whole vector, in query order
        assert no_map == 1
        assert tot == [3, 3, 0, 0, 2]
        assert ava == [1] * 7                    # every read was a query on exactly one rank
        assert inv == np.bincount(np.arange(11) % 3, minlength=3).tolist()
        shards.append(idx)
        ranges.append((lo, hi))
    assert sorted(shards[0] + shards[1]) == list(range(7))
    # dealt in name-rank order: ranks 0,2,4,6 -> shard 0; 1,3,5 -> shard 1
    assert shards[0] == sorted([1, 3, 0, 2]) and shards[1] == sorted([5, 6, 4])
    assert ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][1] == 11


def _ts_worker(rank, world, port, q):
    """The target-sharded decomposition with the ORACLE standing in for a rank's GPU engine: an oracle index over the rank's share of
    the targets, mid_occ forced to the whole set's (what lrge_hip_index_build_tsharded makes global)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lrge_amd import synth
        from oracle import oracle as O
        _, qs, ts = synth.make_config("c2_repeats", 0.03)
        full_opt = O.make_opt(O.PRESET_AVA_ONT, dual=True)
        full = O.Index(O.ReadSet(ts.seqs(), ts.names), full_opt)
        Qo = O.ReadSet(qs.seqs(), qs.names)
        rc, ec, eh = full.twoset_counts(Qo, threads=2)
        assert rc == 0

        # what the collective build makes global: mid_occ, and WHICH keys exceed it over all targets
        hashes = full.minimizers()["x"] >> np.uint64(8)
        keys, cnt = np.unique(hashes, return_counts=True)
        frequent = keys[cnt > full.mid_occ]

        def overlap_fn(lo, hi):
            sub = ts.slice(lo, hi)
            opt = O.make_opt(O.PRESET_AVA_ONT, dual=True)
            opt.mid_occ = full.mid_occ                       # the GLOBAL threshold; a share's own would be smaller
            ix = O.Index(O.ReadSet(sub.seqs(), sub.names), opt)
            held = ix.drop_keys(frequent)                    # ... and a key is dropped by its count over ALL targets
            q.put(("dropped", rank, int(held), int(frequent.size)))
            rc2, c, h = ix.twoset_counts(Qo, threads=2)
            assert rc2 == 0
            own = O.Index(O.ReadSet(sub.seqs(), sub.names), O.make_opt(O.PRESET_AVA_ONT, dual=True)).mid_occ
            q.put(("own_mid_occ", rank, own, full.mid_occ))
            return c, h
        comm = parallel.TorchComm(dist)
        counts, has, (lo, hi) = parallel.twoset_forward_target_sharded(overlap_fn, ts.lens(), comm)
        q.put(("result", rank, bool(np.array_equal(counts, ec)), bool(np.array_equal(has, eh)), int(ec.sum()), lo, hi))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_target_sharded_counts_add_up():
    """lrge_amd.parallel.twoset_forward_target_sharded over gloo, world size 2, on the CPU: with the global mid_occ on every rank the
    per-shard distinct-target counts SUM to the one index's counts and has_mapping ORs (twoset.rs:286-317; the argument behind
    lrge_hip_index_build_tsharded) -- on a repeat-rich set, where the threshold actually drops keys."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ts_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    got = [q.get(timeout=300) for _ in range(6)]
    for p in procs: p.join(timeout=60)
    res = sorted(x for x in got if x[0] == "result")
    assert len(res) == 2
    for _, rank, c_ok, h_ok, total, lo, hi in res:
        assert c_ok and h_ok and total > 0
    assert res[0][5] == 0 and res[0][6] == res[1][5]
    dropped = [x for x in got if x[0] == "dropped"]
    assert all(x[3] > 0 for x in dropped) and sum(x[2] for x in dropped) > 0       # (the set is repeat-rich: some keys are too frequent, and the shards hold them)
    own = [x for x in got if x[0] == "own_mid_occ"]
    assert all(x[2] <= x[3] for x in own)                                          # (a share's own threshold is never above the whole set's)


def _fail_worker(rank, world, port, q):
    """One rank's share fails after the (stand-in) build: it must still enter the closing collective, and EVERY rank must raise."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = parallel.TorchComm(dist)
        lens = np.arange(1, 12) * 100
        out = []

        def boom(*_a):
            if rank == 1:
                raise MemoryError("rank 1 ran out of memory in its overlap call")
            return np.ones(5, np.uint32), np.ones(5, np.uint32)
        for name, call in (
                ("tshard", lambda: parallel.twoset_forward_target_sharded(boom, lens, comm, n_queries=5, build_fn=lambda lo, hi: "index")),
                ("inverse", lambda: parallel.inverse_sharded(lambda a, b: boom()[0], lens, comm, n_indexed=5)),
                ("ava", lambda: parallel.ava_sharded(lambda idx: boom()[0] if rank == 0 else boom(), np.arange(5), comm)),
                ("qshard", lambda: parallel.twoset_forward_sharded(lambda lo, hi: (boom()[0][:0], 0) if rank == 1 else (np.zeros(hi - lo, np.float32), 0), lens, comm))):
            try:
                call()
                out.append((name, "returned"))
            except parallel.RankFailed:
                out.append((name, "RankFailed"))
            except MemoryError:
                out.append((name, "MemoryError"))
        # ... and the communicator is still in step afterwards: a healthy collective works
        tot = comm.all_reduce_u32(np.array([rank + 1], np.uint32))
        q.put((rank, out, int(tot[0])))
    finally:
        dist.destroy_process_group()


def test_gloo_world2_a_failing_rank_fails_the_step_on_every_rank():
    """a rank whose overlap call fails used to return before the all-reduce that closes the step and leave its peers in
    it for ever.  Now it joins with a status word and raises afterwards; the healthy rank raises RankFailed; nobody hangs."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fail_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = dict((r, (o, t)) for r, o, t in (q.get(timeout=120) for _ in procs))
    for p in procs: p.join(timeout=60)
    assert [x[1] for x in res[0][0]] == ["RankFailed"] * 4, res[0]
    assert [x[1] for x in res[1][0]] == ["MemoryError"] * 4, res[1]
    assert res[0][1] == res[1][1] == 3


def test_cross_shard_duplicate_target_names_are_refused():
    """distinct-target counts add up over shards only if no target NAME occurs in two shards (twoset.rs:286-317 counts
    names); the target-sharded form refuses such a set before anything is built."""
    lens = np.full(8, 100)
    b = parallel.shard_by_bases(lens, 2)
    assert not parallel.cross_shard_duplicates(np.array([0, 1, 2, 2, 4, 5, 6, 7]), b)      # a duplicate INSIDE a shard: the library dedups it
    assert parallel.cross_shard_duplicates(np.array([0, 1, 2, 3, 4, 2, 6, 7]), b)
    class World2(parallel.SoloComm):
        rank, world = 0, 2
    with pytest.raises(ValueError, match="Duplicate read identifier"):
        parallel.twoset_forward_target_sharded(lambda lo, hi: (np.zeros(3, np.uint32),) * 2, lens, World2(), t_ranks=np.array([0, 1, 2, 3, 4, 2, 6, 7]))


def test_pack_side_rule():
    """lrge_hip_pack_choice: one rank on the host packs on the host; several do only when every rank has 8 of the
    CPUs the host grants (affinity mask and cgroup bandwidth) -- else the ASCII travels over each rank's own PCIe link and k_pack runs."""
    import ctypes as C
    from lrge_amd import _ffi
    L = _ffi.lib()
    g = C.c_double()
    assert L.lrge_hip_pack_choice(1, C.byref(g)) == 1 and g.value >= 1.0
    q = g.value
    for w in (2, 4, 8, 64, 4096):
        assert L.lrge_hip_pack_choice(w, None) == (1 if q / w >= 8.0 else 0), (w, q)
    assert L.lrge_hip_pack_choice(4096, None) == 0
