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
