"""Thin object wrappers over the C ABI: Context / SeqSet / Index.

The boundary mirrors liblrge's minimap2 wrapper (liblrge/src/minimap2/aligner.rs): an Index is
what AlignerWrapper::new builds (preset + dual + index over the target file), and the overlap
calls are the batched form of Aligner::map plus liblrge's counting shells.
"""
import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import LrgeHipError, Params


def name_ranks(*name_lists):
    """Lexicographic (strcmp) ranks over the union of several name lists; equal names share a rank.
    Returns one uint32 array per input list."""
    allnames = [n if isinstance(n, bytes) else n.encode() for lst in name_lists for n in lst]
    if not allnames:
        return [np.zeros(0, dtype=np.uint32) for _ in name_lists]
    arr = np.array(allnames, dtype=object)
    order = sorted(range(len(allnames)), key=lambda i: allnames[i])  # bytes compare == strcmp without NULs
    ranks = np.zeros(len(allnames), dtype=np.uint32)
    r = 0
    for j, i in enumerate(order):
        if j > 0 and allnames[i] != allnames[order[j - 1]]:
            r = j
        ranks[i] = r
    out, k = [], 0
    for lst in name_lists:
        out.append(ranks[k:k + len(lst)].copy())
        k += len(lst)
    return out


class Context:
    def __init__(self, device=0):
        self._lib = _ffi.lib()
        h = C.c_void_p()
        rc = self._lib.lrge_hip_ctx_create(device, C.byref(h))
        if rc != 0:
            raise LrgeHipError(rc, self._lib.lrge_hip_last_error(None).decode())
        self.h = h
        self.device = device

    def _check(self, rc):
        if rc != 0:
            raise LrgeHipError(rc, self._lib.lrge_hip_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self._lib.lrge_hip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        """Tuning / test option of this context; value None clears it.  The same names are
        read once from the environment (LRGE_HIP_<NAME>) when the context is created, except DEBUG_*."""
        self._check(self._lib.lrge_hip_ctx_set_option(self.h, name.encode(), None if value is None else str(value).encode()))

    def set_timer_level(self, level):
        """0 = call total + chain stage only, 1 = every stage, 2 (default) = also every k_rs_scatter launch."""
        self._check(self._lib.lrge_hip_set_timer_level(self.h, int(level)))

    def timings(self):
        a = (C.c_float * len(_ffi.T_NAMES))()
        self._lib.lrge_hip_last_timings(self.h, C.byref(a))
        return dict(zip(_ffi.T_NAMES, [float(x) for x in a]))

    def counters(self):
        a = (C.c_uint64 * len(_ffi.C_NAMES))()
        self._lib.lrge_hip_last_counters(self.h, C.byref(a))
        return dict(zip(_ffi.C_NAMES, [int(x) for x in a]))

    def upload(self, bases, offsets, ranks=None, wait=True):
        """Read set -> 2-bit packed in HBM.  `bases`: numpy uint8 (pageable or pinned host memory, see host_alloc) or an
        int device pointer to ASCII already resident in HBM.  wait=False queues the transfer + pack on the copy stream
        and returns (lrge_hip_seqset_upload_async): later calls order themselves behind it on the device."""
        return SeqSet(self, bases, offsets, ranks, wait)

    def host_alloc(self, nbytes):
        """Pinned host buffer as a numpy uint8 array (lrge_hip_host_alloc): a DMA source without staging."""
        p = C.c_void_p()
        rc = self._lib.lrge_hip_host_alloc(int(nbytes), C.byref(p))
        if rc != 0:
            raise LrgeHipError(rc, self._lib.lrge_hip_last_error(None).decode())
        return PinnedBuffer(self._lib, p, int(nbytes))

    def estimates(self, counts, read_lens, avg_target_len, n_target_reads, overlap_thresh=100):
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        lens = np.ascontiguousarray(read_lens, dtype=np.uint32)
        out = np.zeros(max(counts.size, 1), dtype=np.float32)
        self._check(self._lib.lrge_hip_estimates(self.h, counts.ctypes.data, lens.ctypes.data, counts.size,
                                                 C.c_float(avg_target_len), int(n_target_reads), overlap_thresh,
                                                 out.ctypes.data))
        return out[:counts.size]


def median(estimates, finite=True, lower=None, upper=None):
    """estimate.rs:80-132 on the host side of the library."""
    v = np.ascontiguousarray(estimates, dtype=np.float32)
    out = (C.c_float * 3)()
    ok = (C.c_int * 3)()
    rc = _ffi.lib().lrge_hip_median(v.ctypes.data if v.size else None, v.size, int(finite),
                                    int(lower is not None), lower or 0.0, int(upper is not None), upper or 0.0,
                                    C.byref(out), C.byref(ok))
    if rc != 0:
        raise LrgeHipError(rc, "invalid quantile arguments")
    return tuple(np.float32(out[i]) if ok[i] else None for i in range(3))


class PinnedBuffer:
    """Pinned host memory owned by the library; `.array` is a numpy uint8 view of it."""

    def __init__(self, lib, ptr, nbytes):
        self._lib, self.ptr, self.nbytes = lib, ptr, nbytes
        self.array = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(max(nbytes, 1),))[:nbytes]

    def free(self):
        if getattr(self, "ptr", None):
            self.array = None
            self._lib.lrge_hip_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class SeqSet:
    def __init__(self, ctx, bases, offsets, ranks=None, wait=True):
        self.ctx = ctx
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n = offsets.size - 1
        self._offsets, self._lens = offsets, None      # (lens: on first use -- 2 M reads cost 5 ms of host time per upload otherwise)
        if isinstance(bases, (int, np.integer)):          # device pointer to resident ASCII
            ptr, self._src = int(bases), None
        else:
            if isinstance(bases, PinnedBuffer):
                self._src, bases = bases, bases.array
            bases = np.ascontiguousarray(bases, dtype=np.uint8)
            ptr = bases.ctypes.data if bases.size else None
            self._src = (getattr(self, "_src", None), bases)   # an async upload reads the source until the set is consumed
        r = None if ranks is None else np.ascontiguousarray(ranks, dtype=np.uint32)
        h = C.c_void_p()
        fn = ctx._lib.lrge_hip_seqset_upload if wait else ctx._lib.lrge_hip_seqset_upload_async
        ctx._check(fn(ctx.h, ptr, offsets.ctypes.data, self.n, None if r is None else r.ctypes.data, C.byref(h)))
        self.h = h

    @property
    def lens(self):
        """read lengths (uint32), from the offsets the set was uploaded with"""
        if self._lens is None:
            self._lens = np.diff(self._offsets).astype(np.uint32)
        return self._lens

    def wait(self):
        self.ctx._check(self.ctx._lib.lrge_hip_seqset_wait(self.h))

    def free(self):
        if getattr(self, "h", None):
            self.ctx._lib.lrge_hip_seqset_free(self.h)     # (drains a pending upload before the source may go)
            self.h = None
        self._src = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def presketch(self, preset):
        """Hint (results unchanged): the next Index() built on this context sketches this set on a side stream beside its
        own sort and table passes; the next overlap call that streams the set uses it (lrge_hip_seqset_presketch)."""
        self.ctx._check(self.ctx._lib.lrge_hip_seqset_presketch(self.ctx.h, self.h, preset))

    def presketch_sharded(self, preset, comm):
        """Collective (every rank of `comm` holds the SAME set): this rank sketches its share of the reads, the minimizers are
        all-gathered, the next overlap call that streams the set uses them (lrge_hip_seqset_presketch_sharded)."""
        self.ctx._check(self.ctx._lib.lrge_hip_seqset_presketch_sharded(self.ctx.h, self.h, preset, comm.h))

    def sketch(self, preset):
        n = C.c_uint64()
        self.ctx._check(self.ctx._lib.lrge_hip_sketch_dump(self.ctx.h, self.h, preset, None, None, 0, C.byref(n)))
        x = np.zeros(max(n.value, 1), dtype=np.uint64)
        y = np.zeros(max(n.value, 1), dtype=np.uint64)
        self.ctx._check(self.ctx._lib.lrge_hip_sketch_dump(self.ctx.h, self.h, preset, x.ctypes.data, y.ctypes.data,
                                                           n.value, C.byref(n)))
        return x[:n.value], y[:n.value]


class Index:
    """AlignerWrapper::new(target_file, threads, preset, dual) -- aligner.rs:310-328."""

    def __init__(self, ctx, targets, preset=_ffi.PRESET_AVA_ONT, streamed=None, comm=None, shard=None, tshard=False):
        """shard = (all_target_lens, all_target_ranks or None, shard_first): `targets` is this rank's contiguous share of the
        target reads and the build is the collective lrge_hip_index_build_sharded (the target sketch is sharded too).
        tshard (with comm): `targets` is this rank's share of the target reads and the index holds ALL of its entries, with the
        occurrence statistics of the whole target set (lrge_hip_index_build_tsharded): the caller maps ALL queries against it
        and sums the counts over the ranks."""
        self.ctx, self.targets, self.preset = ctx, targets, preset
        h = C.c_void_p()
        if tshard:
            self.streamed = None
            ctx._check(ctx._lib.lrge_hip_index_build_tsharded(ctx.h, targets.h, preset, comm.h, C.byref(h)))
            self.h = h
            self.build_timings = ctx.timings()
            self.build_counters = ctx.counters()
            a = (C.c_uint64 * 8)()
            ctx._lib.lrge_hip_last_shard_stats(ctx.h, C.byref(a))
            # (hashes_*: (key, count) pairs, one 8-byte word each; entries_*: the 16-byte minimizers a sharded presketch of the streamed
            # set moved just before this build -- lrge_hip_seqset_presketch_sharded -- else 0)
            self.shard_stats = dict(keyset_bytes=0, entries_sketched=0, entries_sent=int(a[2]), entries_recv=int(a[3]), hashes_sent=int(a[4]), hashes_recv=int(a[5]),
                                    entry_bytes=int(a[6]) & 0xFF or 8, entries_kept=0, hash_bytes=8)
            return
        if shard is not None:
            lens = np.ascontiguousarray(shard[0], dtype=np.uint32)
            ranks = None if shard[1] is None else np.ascontiguousarray(shard[1], dtype=np.uint32)
            self.streamed = streamed
            self.n_targets = lens.size
            ctx._check(ctx._lib.lrge_hip_index_build_sharded(ctx.h, lens.ctypes.data, None if ranks is None else ranks.ctypes.data, lens.size,
                                                             targets.h, int(shard[2]), preset, streamed.h, comm.h, C.byref(h)))
            self.h = h
            self.build_timings = ctx.timings()
            self.build_counters = ctx.counters()
            a = (C.c_uint64 * 8)()
            ctx._lib.lrge_hip_last_shard_stats(ctx.h, C.byref(a))
            self.shard_stats = dict(zip(["keyset_bytes", "entries_sketched", "entries_sent", "entries_recv", "hashes_sent", "hashes_recv",
                                         "entry_bytes", "entries_kept"], [int(x) for x in a]))
            self.shard_stats["hash_bytes"] = self.shard_stats["entry_bytes"] >> 8 or 8
            self.shard_stats["entry_bytes"] &= 0xFF
            return
        # streamed / comm: an index built for ONE streamed set, occurrence statistics still over all targets; with a
        # communicator every rank passes its own range of the streamed reads (lrge_hip_index_build_for)
        self.streamed = streamed
        if streamed is None and comm is None:
            ctx._check(ctx._lib.lrge_hip_index_build(ctx.h, targets.h, preset, C.byref(h)))
        else:
            ctx._check(ctx._lib.lrge_hip_index_build_for(ctx.h, targets.h, preset, None if streamed is None else streamed.h,
                                                         None if comm is None else comm.h, C.byref(h)))
        self.h = h
        self.build_timings = ctx.timings()
        self.build_counters = ctx.counters()

    def free(self):
        if getattr(self, "h", None):
            if getattr(self.ctx, "h", None):      # a closed context has already released everything the index held
                self.ctx._lib.lrge_hip_index_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def stats(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_int32()
        self.ctx._lib.lrge_hip_index_stats(self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(n_minimizers=a.value, n_keys=b.value, mid_occ=c.value)

    def dump(self):
        n = C.c_uint64()
        self.ctx._check(self.ctx._lib.lrge_hip_index_dump(self.ctx.h, self.h, None, None, 0, C.byref(n)))
        k = np.zeros(max(n.value, 1), dtype=np.uint64)
        p = np.zeros(max(n.value, 1), dtype=np.uint64)
        self.ctx._check(self.ctx._lib.lrge_hip_index_dump(self.ctx.h, self.h, k.ctypes.data, p.ctypes.data, n.value,
                                                          C.byref(n)))
        return k[:n.value], p[:n.value]

    def _params(self, remove_internal, ratio):
        return Params(int(bool(remove_internal)), float(ratio))

    def overlap_twoset(self, queries, remove_internal=False, max_overhang_ratio=0.2):
        counts = np.zeros(max(queries.n, 1), dtype=np.uint32)
        has = np.zeros(max(queries.n, 1), dtype=np.uint32)
        p = self._params(remove_internal, max_overhang_ratio)
        self.ctx._check(self.ctx._lib.lrge_hip_overlap_twoset(self.ctx.h, self.h, queries.h, C.byref(p),
                                                              counts.ctypes.data, has.ctypes.data))
        return counts[:queries.n], has[:queries.n]

    def overlap_inverse(self, streamed, remove_internal=False, max_overhang_ratio=0.2):
        counts = np.zeros(max(self.targets.n, 1), dtype=np.uint32)
        p = self._params(remove_internal, max_overhang_ratio)
        self.ctx._check(self.ctx._lib.lrge_hip_overlap_inverse(self.ctx.h, self.h, streamed.h, C.byref(p),
                                                               counts.ctypes.data))
        return counts[:self.targets.n]

    def overlap_ava(self, remove_internal=False, max_overhang_ratio=0.2, shard=None):
        """All-vs-all counts keyed by indexed read.  `shard`: a SeqSet holding a subset of the indexed reads (name
        ranks over the whole set) -> this shard's contribution; the sum over a partition is the full result."""
        counts = np.zeros(max(self.targets.n, 1), dtype=np.uint32)
        p = self._params(remove_internal, max_overhang_ratio)
        reads = self.targets if shard is None else shard
        self.ctx._check(self.ctx._lib.lrge_hip_overlap_ava(self.ctx.h, self.h, reads.h, C.byref(p),
                                                           counts.ctypes.data))
        return counts[:self.targets.n]

    def chains(self, queries, dual=True):
        n = C.c_uint64()
        self.ctx._check(self.ctx._lib.lrge_hip_chains(self.ctx.h, self.h, queries.h, int(dual), None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=_ffi.CHAIN)
        self.ctx._check(self.ctx._lib.lrge_hip_chains(self.ctx.h, self.h, queries.h, int(dual), out.ctypes.data,
                                                      n.value, C.byref(n)))
        return out[:n.value]

    def paf_stats(self, queries):
        """Per-query (rep_len, sum_span, n_kept): the rl tag and the ingredients of mm_est_err's avg_k."""
        n = max(queries.n, 1)
        rl = np.zeros(n, dtype=np.int32); ss = np.zeros(n, dtype=np.uint64); nk = np.zeros(n, dtype=np.uint32)
        self.ctx._check(self.ctx._lib.lrge_hip_paf_stats(self.ctx.h, self.h, queries.h, rl.ctypes.data, ss.ctypes.data,
                                                         nk.ctypes.data))
        return rl[:queries.n], ss[:queries.n], nk[:queries.n]

    def anchors(self, queries, q, dual=True):
        n = C.c_uint64()
        self.ctx._check(self.ctx._lib.lrge_hip_anchors_dump(self.ctx.h, self.h, queries.h, int(dual), q, None, None, 0,
                                                            C.byref(n)))
        x = np.zeros(max(n.value, 1), dtype=np.uint64)
        y = np.zeros(max(n.value, 1), dtype=np.uint64)
        self.ctx._check(self.ctx._lib.lrge_hip_anchors_dump(self.ctx.h, self.h, queries.h, int(dual), q, x.ctypes.data,
                                                            y.ctypes.data, n.value, C.byref(n)))
        return x[:n.value], y[:n.value]
