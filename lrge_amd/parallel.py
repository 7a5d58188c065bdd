"""Multi-GPU sharding of the overlap path: one context per GPU.

The path shards by *streamed* read: queries in two-set forward mode, streamed targets in inverse mode, reads-as-queries
in all-vs-all.  Every rank owns its range end to end; its index holds the entries its own streamed reads can ask for
(lrge_hip_index_build_for: no index data crosses the links, one small all-reduce makes mid_occ global).  One collective
closes the step:
  * forward two-set: all-gather of the per-read f32 estimate vectors;
  * all-vs-all / inverse: all-reduce(sum) of the u32 count vector keyed by indexed read.
Communicators: RcclComm (one process per GPU, RCCL through the C ABI), LocalGroup.comm (ranks = threads of one process),
TorchComm (torch.distributed; "gloo" in the CPU tests), SoloComm.  The drivers below take any of them.
"""
import numpy as np


def shard_by_bases(lens, world):
    """Contiguous read ranges with (nearly) equal base counts: returns world+1 boundaries."""
    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    if world <= 1 or n == 0:
        return [0] + [n] * max(world, 1)
    cum = np.concatenate([[0], np.cumsum(lens)])
    total = cum[-1]
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(cum, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), n))
    bounds.append(n)
    return bounds


def shard_round_robin(n, rank, world):
    """All-vs-all: NO_DUAL drops every hit onto a smaller name, so low-rank reads carry more work;
    interleaving balances it.  Returns the read indices of this rank."""
    return np.arange(rank, n, world, dtype=np.int64)


class _Comm:
    """A communicator of the C ABI (lrge_hip_comm_*): the collectives run inside liblrge_hip.so."""

    def __init__(self, ctx, h, rank, world):
        self.ctx, self.h, self.rank, self.world = ctx, h, rank, world

    def all_reduce_u32(self, counts):
        """In-place-style SUM of per-rank partial count vectors (all-vs-all / inverse)."""
        a = np.ascontiguousarray(counts, dtype=np.uint32).copy()
        self.ctx._check(self.ctx._lib.lrge_hip_comm_allreduce_u32(self.h, a.ctypes.data, a.size))
        return a

    def all_gather_f32(self, local, max_len, lens):
        """All-gather of the per-read estimate vectors; every rank knows every shard's length (the shards are cut from
        read lengths all ranks hold), so one fixed-size gather of `max_len` floats per rank does it."""
        import ctypes as C
        send = np.full(max(max_len, 1), np.nan, dtype=np.float32)
        send[:len(local)] = local
        recv = np.empty(max(max_len, 1) * self.world, dtype=np.float32)
        self.ctx._check(self.ctx._lib.lrge_hip_comm_allgather(self.h, send.ctypes.data, C.c_size_t(send.nbytes), recv.ctypes.data))
        m = max(max_len, 1)
        return np.concatenate([recv[r * m:r * m + lens[r]] for r in range(self.world)])

    def all_to_all_v(self, send, send_counts):
        """Variable-size all-to-all of a 1-d numpy array: send_counts[d] consecutive elements go to rank d.  Returns
        (received array, receive counts) -- the counts travel first (one all-gather), as the C ABI asks."""
        import ctypes as C
        send = np.ascontiguousarray(send)
        sc = np.ascontiguousarray(send_counts, dtype=np.uint64)
        allc = np.empty(self.world * self.world, dtype=np.uint64)
        self.ctx._check(self.ctx._lib.lrge_hip_comm_allgather(self.h, sc.ctypes.data, C.c_size_t(sc.nbytes), allc.ctypes.data))
        rc = allc.reshape(self.world, self.world)[:, self.rank].copy()
        so = np.zeros(self.world + 1, dtype=np.uint64); np.cumsum(sc, out=so[1:])
        ro = np.zeros(self.world + 1, dtype=np.uint64); np.cumsum(rc, out=ro[1:])
        recv = np.empty(max(int(ro[-1]), 1), dtype=send.dtype)
        self.ctx._check(self.ctx._lib.lrge_hip_comm_alltoallv(self.h, send.ctypes.data if send.size else None, so.ctypes.data, recv.ctypes.data,
                                                              ro.ctypes.data, C.c_size_t(send.dtype.itemsize)))
        return recv[:int(ro[-1])], rc

    def rccl_ranks(self):
        import ctypes as C
        n = C.c_int()
        self.ctx._check(self.ctx._lib.lrge_hip_comm_rccl_ranks(self.h, C.byref(n)))
        return n.value

    def rccl_ops(self):
        """librccl data-path calls made through this communicator so far (lrge_hip_comm_rccl_ops)."""
        import ctypes as C
        n = C.c_uint64()
        self.ctx._check(self.ctx._lib.lrge_hip_comm_rccl_ops(self.h, C.byref(n)))
        return n.value

    def turn(self, begin):
        """Serialized local groups (timing emulation of a world on one GPU): take / give back the GPU."""
        self.ctx._lib.lrge_hip_comm_local_turn(self.h, 1 if begin else 0)

    def busy_ms(self, reset=False):
        return float(self.ctx._lib.lrge_hip_comm_busy_ms(self.h, 1 if reset else 0))

    def standin_ms(self, reset=False):
        """(local groups) the part of busy_ms spent in device-to-device copies that stand in for link transfers (lrge_hip_comm_standin_ms)."""
        return float(self.ctx._lib.lrge_hip_comm_standin_ms(self.h, 1 if reset else 0))

    def abort(self):
        """This rank cannot go on (lrge_hip_comm_abort): its peers leave the collectives they are waiting in with an error instead
        of waiting for ever (local transport), every later collective here fails at once."""
        if getattr(self, "h", None):
            self.ctx._lib.lrge_hip_comm_abort(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.ctx._lib.lrge_hip_comm_destroy(self.h)
            self.h = None


class RankFailed(RuntimeError):
    """Another rank of the job failed; this rank's own work was fine."""


def _abort(comm):
    try:
        comm.abort()
    except Exception:      # noqa: BLE001 -- communicators without an abort (SoloComm, TorchComm): nothing to wake
        pass


def _close_step(comm, sizes, work):
    """The all-reduce that closes a sharded step, made failure-collective.  `work()` -> one u32 vector per entry of `sizes`; whatever
    happens inside it on THIS rank, the rank still enters the all-reduce its peers are heading for -- with zeros and a status word
    -- and raises afterwards; the peers see the status word and raise RankFailed.  (The reference's workers end the whole run on a
    MapError: twoset.rs:279-284.)  A failure of the collective itself aborts the communicator."""
    err = None
    try:
        parts = [np.ascontiguousarray(p, dtype=np.uint32) for p in work()]
        if [len(p) for p in parts] != [int(x) for x in sizes]:
            raise ValueError("a sharded step returned vectors of %r entries, expected %r" % ([len(p) for p in parts], list(sizes)))
    except BaseException as e:      # noqa: BLE001 -- re-raised below, after the collective
        err = e
        parts = [np.zeros(int(x), np.uint32) for x in sizes]
    buf = np.concatenate(parts + [np.array([1 if err is not None else 0], np.uint32)])
    try:
        buf = comm.all_reduce_u32(buf)
    except BaseException:
        _abort(comm)
        if err is not None:
            raise err
        raise
    if err is not None:
        raise err
    if int(buf[-1]):
        raise RankFailed("%d other rank(s) failed inside the step; this rank's results are discarded" % int(buf[-1]))
    out, o = [], 0
    for x in sizes:
        out.append(buf[o:o + int(x)].copy()); o += int(x)
    return out


class RcclComm(_Comm):
    """RCCL over xGMI, one process per GPU.  The 128-byte unique id travels through whatever the host has."""

    @classmethod
    def create(cls, ctx, rank, world, unique_id):
        import ctypes as C
        h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        ctx._check(ctx._lib.lrge_hip_comm_create(ctx.h, rank, world, buf, C.byref(h)))
        return cls(ctx, h, rank, world)

    @staticmethod
    def unique_id(ctx):
        import ctypes as C
        buf = (C.c_char * 128)()
        rc = ctx._lib.lrge_hip_comm_unique_id(buf)
        if rc != 0:
            from ._ffi import LrgeHipError
            raise LrgeHipError(rc, ctx._lib.lrge_hip_last_error(None).decode())
        return bytes(buf)

    @classmethod
    def bootstrap(cls, ctx, rank, world, dist):
        """Rank 0 draws the id; torch.distributed (any backend, used as a store only) hands it out."""
        box, err = [None], None
        if rank == 0:
            try:
                box[0] = cls.unique_id(ctx)
            except Exception as e:      # noqa: BLE001 -- the other ranks must not be left waiting in the broadcast
                err = e
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            raise RuntimeError("rank 0 could not draw a RCCL unique id" + (": %s" % err if err else ""))
        return cls.create(ctx, rank, world, box[0])


class HostComm(_Comm):
    """The library's collectives carried by torch.distributed on HOST buffers (gloo works wherever TCP does): the
    fall-back transport of bench.py should RCCL fail to come up, and what a host with its own MPI would plug in."""

    def __init__(self, ctx, dist):
        import ctypes as C
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)

        def allreduce(_user, buf, n, esz):
            try:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint64 if esz == 8 else C.c_uint32)), shape=(n,))
                t = torch.from_numpy(a.astype(np.int64))
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
                a[:] = t.numpy().astype(a.dtype)
                return 0
            except Exception:      # noqa: BLE001 -- reported through the return code
                return 1

        def allgather(_user, send, nbytes, recv):
            try:
                s_ = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                r_ = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
                bufs = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(bufs, torch.from_numpy(s_.copy()))
                r_[:] = np.concatenate([b.numpy() for b in bufs])
                return 0
            except Exception:      # noqa: BLE001
                return 1

        self._ar, self._ag = AR(allreduce), AG(allgather)      # keep the trampolines alive
        h = C.c_void_p()
        ctx._check(ctx._lib.lrge_hip_comm_create_host(ctx.h, rank, world, C.cast(self._ar, C.c_void_p), C.cast(self._ag, C.c_void_p),
                                                      None, C.byref(h)))
        super().__init__(ctx, h, rank, world)


class ThreadHostGroup:
    """The library's HOST-CALLBACK transport (lrge_hip_comm_create_host) between the threads of one process: the two callbacks meet
    behind a threading.Barrier.  A stand-in for a caller's own MPI / gloo, and a STRICT one: every rank of a collective must
    arrive with the same operation and the same shape, as a real transport demands -- a mismatch (or a rank that never arrives:
    the barrier times out) fails the collective on every rank and is recorded in `faults`.  tests/test_gpu_multi.py drives the
    sharded index build through it, with a rank failing at every stage, to show that no rank is ever left in a collective its
    peers do not enter."""

    def __init__(self, world, timeout=60.0):
        import threading
        self.world, self.timeout = world, timeout
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.faults = []
        self.log = [[] for _ in range(world)]        # per rank: (op, shape) of every collective entered
        self._keep = []

    def _meet(self, rank, desc, payload):
        """-> list of every rank's payload, or None on a mismatch / timeout"""
        import threading
        self.log[rank].append(desc)
        self.slots[rank] = (desc, payload)
        try:
            self.bar.wait(self.timeout)
            got = list(self.slots)
            self.bar.wait(self.timeout)
        except threading.BrokenBarrierError:
            self.faults.append(("rank %d: a peer never entered %r" % (rank, desc)))
            return None
        if any(g[0] != desc for g in got):
            if rank == 0:
                self.faults.append("mismatched collectives: %r" % [g[0] for g in got])
            return None
        return [g[1] for g in got]

    def comm(self, ctx, rank):
        import ctypes as C
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int)
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
        world = self.world

        def allreduce(_user, buf, n, esz):
            try:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint64 if esz == 8 else C.c_uint32)), shape=(n,))
                got = self._meet(rank, ("allreduce", int(n), int(esz)), a.copy())
                if got is None:
                    return 1
                a[:] = np.sum(np.stack(got), axis=0, dtype=a.dtype)
                return 0
            except Exception as e:      # noqa: BLE001 -- reported through the return code
                self.faults.append(repr(e))
                return 1

        def allgather(_user, send, nbytes, recv):
            try:
                s_ = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                got = self._meet(rank, ("allgather", int(nbytes)), s_.copy())
                if got is None:
                    return 1
                r_ = np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_uint8)), shape=(nbytes * world,))
                r_[:] = np.concatenate(got)
                return 0
            except Exception as e:      # noqa: BLE001
                self.faults.append(repr(e))
                return 1

        ar, ag = AR(allreduce), AG(allgather)
        self._keep += [ar, ag]                         # keep the trampolines alive
        h = C.c_void_p()
        ctx._check(ctx._lib.lrge_hip_comm_create_host(ctx.h, rank, world, C.cast(ar, C.c_void_p), C.cast(ag, C.c_void_p), None, C.byref(h)))
        return _Comm(ctx, h, rank, world)

    def close(self):
        pass


class LocalGroup:
    """The ranks are threads of this process (one context each); see lrge_hip_comm_create_local."""

    def __init__(self, world):
        import ctypes as C
        from . import _ffi
        self._lib, self.world = _ffi.lib(), world
        self.h = C.c_void_p()
        rc = self._lib.lrge_hip_comm_local_group_create(world, C.byref(self.h))
        if rc != 0:
            raise RuntimeError("local group: %d" % rc)

    def serialize(self, on=True):
        """The ranks take turns on the GPU (see lrge_hip_comm_local_group_serialize): clean per-rank timings on one GPU."""
        self._lib.lrge_hip_comm_local_group_serialize(self.h, int(on) if on in (0, 1, 2) else (1 if on else 0))     # 2: + idle arena segments go back to the runtime between turns

    def comm(self, ctx, rank):
        import ctypes as C
        h = C.c_void_p()
        ctx._check(self._lib.lrge_hip_comm_create_local(ctx.h, rank, self.h, C.byref(h)))
        return _Comm(ctx, h, rank, self.world)

    def close(self):
        if getattr(self, "h", None):
            self._lib.lrge_hip_comm_local_group_destroy(self.h)
            self.h = None


class TorchComm:
    """The same two collectives over torch.distributed -- backend "gloo" in the CPU tests (world size 2, no GPU), "nccl"
    (= RCCL) when a host prefers torch's communicator to the library's own.  Same interface as the C-ABI communicators."""

    def __init__(self, dist=None):
        import torch.distributed as _d
        self.dist = dist or _d
        self.rank, self.world = self.dist.get_rank(), self.dist.get_world_size()
        self.dev = "cuda" if self.dist.get_backend() == "nccl" else "cpu"

    def all_reduce_u32(self, counts):
        import torch
        t = torch.as_tensor(np.ascontiguousarray(counts).astype(np.int64)).to(self.dev)   # u32 travels as int64 (gloo / RCCL sum)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy().astype(np.uint32)

    def all_gather_f32(self, local, max_len, lens):
        import torch
        m = max(max_len, 1)
        pad = torch.full((m,), float("nan"), dtype=torch.float32, device=self.dev)
        pad[:len(local)] = torch.as_tensor(np.ascontiguousarray(local, dtype=np.float32)).to(self.dev)
        bufs = [torch.empty(m, dtype=torch.float32, device=self.dev) for _ in range(self.world)]
        self.dist.all_gather(bufs, pad)
        return np.concatenate([b[:n].cpu().numpy() for b, n in zip(bufs, lens)])

    def close(self):
        pass


class SoloComm:
    """World of one: nothing to exchange."""
    rank, world = 0, 1

    def all_reduce_u32(self, counts):
        return np.asarray(counts, dtype=np.uint32)

    def all_gather_f32(self, local, max_len, lens):
        return np.asarray(local, dtype=np.float32)

    def close(self):
        pass


def twoset_forward_sharded(overlap_fn, q_lens, comm):
    """Two-set forward over comm.world GPUs (STRONG scaling of one job): the queries are cut into contiguous ranges with equal
    base counts, `overlap_fn(lo, hi) -> (estimates f32[hi-lo], no_mapping int)` runs this rank's range end to end (its
    index restricted to its own queries: engine.Index(..., streamed=, comm=)), one all-gather of the estimate vectors
    closes the step.  Returns (all estimates in query order, total no_mapping_count, (lo, hi))."""
    b = shard_by_bases(q_lens, comm.world)
    lo, hi = b[comm.rank], b[comm.rank + 1]
    lens = [b[i + 1] - b[i] for i in range(comm.world)]
    # the index build inside overlap_fn is collective (and failure-collective: host_index_collective.inl); a failure around it on
    # this rank alone still enters the two closing collectives -- NaNs and a status word -- and raises afterwards
    err = None
    try:
        est, no_map = overlap_fn(lo, hi)
        est = np.ascontiguousarray(est, dtype=np.float32)
        if len(est) != hi - lo:
            raise ValueError("a sharded step returned %d estimates for %d queries" % (len(est), hi - lo))
    except BaseException as e:      # noqa: BLE001 -- re-raised below
        err, est, no_map = e, np.full(hi - lo, np.nan, np.float32), 0
    try:
        allv = comm.all_gather_f32(est, max(lens) if lens else 0, lens)
        nm = comm.all_reduce_u32(np.array([int(no_map), 1 if err is not None else 0], dtype=np.uint32))
    except BaseException:
        _abort(comm)
        if err is not None:
            raise err
        raise
    if err is not None:
        raise err
    if int(nm[1]):
        raise RankFailed("%d other rank(s) failed inside the step; this rank's results are discarded" % int(nm[1]))
    return allv, int(nm[0]), (lo, hi)


def cross_shard_duplicates(t_ranks, bounds):
    """True if a target identifier (name rank: equal names <=> equal rank) occurs in two DIFFERENT shards.  The shards' distinct-
    target counts add up only over disjoint NAMES (twoset.rs:286-317 inserts target_name into a HashSet and never rejects a
    duplicate id); inside one shard the library dedups by rank, across shards nobody can."""
    t_ranks = np.asarray(t_ranks)
    shard = np.searchsorted(np.asarray(bounds[1:-1]), np.arange(len(t_ranks)), side="right")
    order = np.argsort(t_ranks, kind="stable")
    r, sh = t_ranks[order], shard[order]
    return bool(((r[1:] == r[:-1]) & (sh[1:] != sh[:-1])).any())


def twoset_forward_target_sharded(overlap_fn, t_lens, comm, n_queries=None, build_fn=None, t_ranks=None):
    """Two-set forward over comm.world GPUs with the TARGETS sharded (lrge_hip_index_build_tsharded): the target reads are cut into
    contiguous ranges with equal base counts, every rank maps ALL queries against the index of its range (built with the occurrence
    statistics of the whole target set: engine.Index(ctx, target_shard, preset, comm=comm, tshard=True)); the shards hold disjoint
    targets, so the distinct-target counts of twoset.rs:286-317 add up and has_mapping ORs: ONE all-reduce closes the step.

      build_fn(t_lo, t_hi) -> index     (optional) upload + the collective index build.  A failure here -- before or inside the
                                        build -- aborts the communicator: the peers leave the build with an error.
      overlap_fn(index) -> (counts u32[Q], has_mapping u32[Q])     with build_fn; failures here are carried by the status word of
                                        the closing all-reduce (every rank raises, nobody waits)
      overlap_fn(t_lo, t_hi) -> (counts, has_mapping)              without build_fn (build inside): any failure aborts
      n_queries                         Q (needed for the status-word form: a failed rank still sends vectors of the right shape)
      t_ranks                           name ranks of ALL targets: a target identifier that occurs in two shards would be counted
                                        twice -- refused with ValueError before anything is built (use the query-sharded form, or
                                        one GPU, for such a set)

    Returns (counts, has_mapping 0/1, (t_lo, t_hi))."""
    b = shard_by_bases(t_lens, comm.world)
    lo, hi = b[comm.rank], b[comm.rank + 1]
    if t_ranks is not None and cross_shard_duplicates(t_ranks, b):       # (every rank holds the same ranks: every rank refuses)
        raise ValueError("Duplicate read identifier across target shards: the target-sharded forward form cannot count distinct names over shards")
    if build_fn is None or n_queries is None:
        try:
            counts, has = overlap_fn(lo, hi)
            n = len(counts)
            counts, has = _close_step(comm, (n, n), lambda: (counts, has))
        except BaseException:
            _abort(comm)
            raise
        return counts, (has > 0).astype(np.uint32), (lo, hi)
    try:
        ix = build_fn(lo, hi)
    except BaseException:
        _abort(comm)
        raise
    counts, has = _close_step(comm, (n_queries, n_queries), lambda: overlap_fn(ix))
    return counts, (has > 0).astype(np.uint32), (lo, hi)


def shard_by_rank_round_robin(name_ranks, rank, world):
    """All-vs-all shards: reads dealt round-robin in NAME-RANK order (NO_DUAL lets the smaller-named read of a pair
    carry it, so contiguous rank ranges would be unbalanced).  Returns this rank's read indices,
    ascending (file order is kept inside a shard)."""
    order = np.argsort(np.asarray(name_ranks), kind="stable")
    return np.sort(order[rank::max(world, 1)])


def ava_sharded(overlap_shard_fn, name_ranks, comm):
    """All-vs-all over comm.world GPUs.  `overlap_shard_fn(idx) -> u32[n_reads]`: counts keyed by indexed read that the
    reads `idx`, used as queries against the replicated index, contribute (engine.Index.overlap_ava(shard=...)).
    One all_reduce(sum) of the count vector closes the step (with a status word: a rank that fails still joins it, and every
    rank raises); returns (counts of the whole job, idx)."""
    idx = shard_by_rank_round_robin(name_ranks, comm.rank, comm.world)
    (counts,) = _close_step(comm, (len(name_ranks),), lambda: (overlap_shard_fn(idx),))
    return counts, idx


def inverse_sharded(overlap_shard_fn, streamed_lens, comm, n_indexed=None):
    """Inverse two-set (--use-min-ref): the index holds the query set (small; replicated, or restricted to the rank's
    streamed reads), the streamed target reads are cut by bases.  `overlap_shard_fn(lo, hi) -> u32[n_indexed]`; one
    all_reduce(sum) closes the step.  With n_indexed given the all-reduce carries a status word (a failing rank still joins it and
    every rank raises); without it a failure aborts the communicator."""
    b = shard_by_bases(streamed_lens, comm.world)
    lo, hi = b[comm.rank], b[comm.rank + 1]
    if n_indexed is not None:
        (counts,) = _close_step(comm, (n_indexed,), lambda: (overlap_shard_fn(lo, hi),))
        return counts, (lo, hi)
    try:
        part = np.asarray(overlap_shard_fn(lo, hi), dtype=np.uint32)
        (counts,) = _close_step(comm, (len(part),), lambda: (part,))
    except BaseException:
        _abort(comm)
        raise
    return counts, (lo, hi)
