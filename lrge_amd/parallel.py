"""Multi-GPU sharding of the overlap path: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The path shards by *streamed* read: queries in two-set forward mode, streamed targets in inverse
mode, reads-as-queries in all-vs-all.  The index is replicated (built redundantly on every GPU, no
data-path collective).  Exactly one collective closes the step:
  * forward two-set: all_gather of the per-read f32 estimate vectors (ragged -> padded);
  * all-vs-all / inverse: all_reduce(sum) of the u32 count vector keyed by indexed read.
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

    def close(self):
        if getattr(self, "h", None):
            self.ctx._lib.lrge_hip_comm_destroy(self.h)
            self.h = None


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
        box = [cls.unique_id(ctx) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return cls.create(ctx, rank, world, box[0])


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

    def comm(self, ctx, rank):
        import ctypes as C
        h = C.c_void_p()
        ctx._check(self._lib.lrge_hip_comm_create_local(ctx.h, rank, self.h, C.byref(h)))
        return _Comm(ctx, h, rank, self.world)

    def close(self):
        if getattr(self, "h", None):
            self._lib.lrge_hip_comm_local_group_destroy(self.h)
            self.h = None


def _dist():
    import torch.distributed as dist
    return dist


def gather_ragged_f32(local, device=None):
    """all_gather of variable-length f32 vectors; returns the concatenation in rank order."""
    import torch
    dist = _dist()
    world = dist.get_world_size()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    loc = torch.as_tensor(np.ascontiguousarray(local, dtype=np.float32)).to(dev)
    n_loc = torch.tensor([loc.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n_loc)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes) if sizes else 0
    pad = torch.full((m,), float("nan"), dtype=torch.float32, device=dev)
    pad[:loc.numel()] = loc
    bufs = [torch.empty(m, dtype=torch.float32, device=dev) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return np.concatenate([b[:s].cpu().numpy() for b, s in zip(bufs, sizes)]) if m else np.zeros(0, np.float32)


def allreduce_counts_u32(counts, device=None):
    """Sum of per-rank partial count vectors (AVA / inverse).  u32 travels as int64 (gloo/RCCL sum)."""
    import torch
    dist = _dist()
    dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
    t = torch.as_tensor(np.ascontiguousarray(counts).astype(np.int64)).to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy().astype(np.uint32)


def twoset_forward_sharded(overlap_fn, q_lens, rank, world):
    """Run `overlap_fn(lo, hi) -> (estimates f32[hi-lo], no_mapping int)` on this rank's query range
    and gather.  Returns (all estimates in query order, total no_mapping_count, (lo, hi))."""
    import torch
    dist = _dist()
    b = shard_by_bases(q_lens, world)
    lo, hi = b[rank], b[rank + 1]
    est, no_map = overlap_fn(lo, hi)
    allv = gather_ragged_f32(est)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    nm = torch.tensor([int(no_map)], dtype=torch.int64, device=dev)
    dist.all_reduce(nm, op=dist.ReduceOp.SUM)
    return allv, int(nm.item()), (lo, hi)


def shard_by_rank_round_robin(name_ranks, rank, world):
    """All-vs-all shards: reads dealt round-robin in NAME-RANK order (NO_DUAL lets the smaller-named read of a pair
    carry it, so contiguous rank ranges would be unbalanced).  Returns this rank's read indices,
    ascending (file order is kept inside a shard)."""
    order = np.argsort(np.asarray(name_ranks), kind="stable")
    return np.sort(order[rank::max(world, 1)])


def ava_sharded(overlap_shard_fn, name_ranks, rank, world):
    """All-vs-all over `world` GPUs.  `overlap_shard_fn(idx) -> u32[n_reads]`: counts keyed by indexed read that the
    reads `idx`, used as queries against the replicated index, contribute (engine.Index.overlap_ava(shard=...)).
    One all_reduce(sum) of the count vector closes the step; returns (counts of the whole job, idx)."""
    idx = shard_by_rank_round_robin(name_ranks, rank, world)
    part = np.asarray(overlap_shard_fn(idx), dtype=np.uint32)
    return (allreduce_counts_u32(part) if world > 1 else part), idx


def inverse_sharded(overlap_shard_fn, streamed_lens, rank, world):
    """Inverse two-set (--use-min-ref): the index holds the query set (replicated), the streamed target reads are
    sharded by bases.  `overlap_shard_fn(lo, hi) -> u32[n_indexed]`; one all_reduce(sum) closes the step."""
    b = shard_by_bases(streamed_lens, world)
    lo, hi = b[rank], b[rank + 1]
    part = np.asarray(overlap_shard_fn(lo, hi), dtype=np.uint32)
    return (allreduce_counts_u32(part) if world > 1 else part), (lo, hi)
