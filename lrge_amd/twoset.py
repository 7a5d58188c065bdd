"""TwoSetStrategy + Builder: host mirror of liblrge/src/twoset.rs and twoset/builder.rs on top of the
C ABI.  Same setters, defaults and error behaviour; the overlap work runs on the MI355X."""
import logging

import numpy as np

from . import _ffi, engine, readio
from .estimate import Estimate, LrgeError

log = logging.getLogger("lrge_amd")

DEFAULT_TARGET_NUM_READS = 10_000   # twoset.rs:65
DEFAULT_QUERY_NUM_READS = 5_000     # twoset.rs:66
PLATFORM_PRESET = {"ont": _ffi.PRESET_AVA_ONT, "nanopore": _ffi.PRESET_AVA_ONT,
                   "pb": _ffi.PRESET_AVA_PB, "pacbio": _ffi.PRESET_AVA_PB}


def unique_random_set(k, n, seed=None):
    """lib.rs:189-204.  k distinct indices in [0, n) in the order rand 0.9.4's `index::sample` returns them for
    `StdRng::seed_from_u64(seed)` (OS entropy without a seed): `lrge_hip_unique_random_set`, restated in
    include/lrge_rand.hpp."""
    if k > n:
        raise ValueError("Cannot generate %d unique values from a range of 0 to %d" % (k, n))
    out = np.zeros(k, dtype=np.uint32)
    rc = _ffi.lib().lrge_hip_unique_random_set(k, n, 0 if seed is None else 1, 0 if seed is None else int(seed) & (2**64 - 1),
                                               out.ctypes.data)
    if rc:
        raise LrgeError("InvalidArgument", "unique_random_set failed (%d)" % rc)
    return out


def split_into_sets(indices, size_first):
    """twoset.rs:632-652: the LAST size_first sampled indices form the first set."""
    idx = list(indices)
    first = set(idx[len(idx) - min(size_first, len(idx)):])
    second = set(idx[:len(idx) - len(first)])
    return first, second


class Builder:
    """twoset/builder.rs:22-185."""

    def __init__(self):
        self._t, self._q = DEFAULT_TARGET_NUM_READS, DEFAULT_QUERY_NUM_READS
        self._remove_internal, self._ratio = False, 0.2
        self._use_min_ref, self._threads, self._tmpdir, self._seed = False, 1, None, None
        self._platform, self._device = "ont", 0

    def target_num_reads(self, n): self._t = int(n); return self
    def query_num_reads(self, n): self._q = int(n); return self
    def remove_internal(self, flag, max_overhang_ratio=0.2):
        self._remove_internal = bool(flag)
        if flag:
            self._ratio = float(max_overhang_ratio)
        return self
    def use_min_ref(self, flag): self._use_min_ref = bool(flag); return self
    def threads(self, n): self._threads = int(n); return self
    def tmpdir(self, d): self._tmpdir = d; return self
    def seed(self, s): self._seed = s; return self
    def platform(self, p): self._platform = str(p).lower(); return self
    def device(self, d): self._device = int(d); return self

    def build(self, input_):
        return TwoSetStrategy(input_, self)


class TwoSetStrategy(Estimate):
    def __init__(self, input_, builder=None):
        b = builder or Builder()
        self.input = input_
        self.target_num_reads, self.query_num_reads = b._t, b._q
        self.remove_internal, self.max_overhang_ratio = b._remove_internal, b._ratio
        self.use_min_ref, self.threads, self.seed = b._use_min_ref, b._threads, b._seed
        self.platform, self.device = b._platform, b._device
        self.target_num_bases = self.query_num_bases = 0
        self.timings = None

    # twoset.rs:122-201
    def split_fastq(self):
        names, seqs = readio.load(self.input)
        n = len(names)
        if n > 0xFFFFFFFF:
            raise LrgeError("TooManyReadsError", "Number of reads in input file (%d) exceeds maximum allowed value" % n)
        n_req = self.target_num_reads + self.query_num_reads
        if n <= self.query_num_reads:
            raise LrgeError("TooFewReadsError", "Number of reads in input file (%d) is <= query number of reads (%d)"
                            % (n, self.query_num_reads))
        if n < n_req:
            log.warning("Number of reads in input file (%d) is less than the sum of target and query reads (%d)", n, n_req)
            self.target_num_reads = n - self.query_num_reads
            n_req = n
            log.warning("Using %d target reads", self.target_num_reads)
        t_idx, q_idx = split_into_sets(unique_random_set(n_req, n, self.seed), self.target_num_reads)
        t = [i for i in range(n) if i in t_idx]      # file order, like iter_records
        q = [i for i in range(n) if i in q_idx]
        tn, ts = [names[i] for i in t], [seqs[i] for i in t]
        qn, qs = [names[i] for i in q], [seqs[i] for i in q]
        self.target_num_bases, self.query_num_bases = sum(map(len, ts)), sum(map(len, qs))
        avg_target_len = np.float32(self.target_num_bases) / np.float32(self.target_num_reads)
        return (tn, ts), (qn, qs), avg_target_len

    # twoset.rs:587-606
    def generate_estimates(self):
        (tn, ts), (qn, qs), avg_target_len = self.split_fastq()
        try:
            preset = PLATFORM_PRESET[self.platform]
        except KeyError:
            raise LrgeError("InvalidPlatform", self.platform)
        ctx = engine.Context(self.device)
        Q = T = ix = None
        try:
            qr, tr = engine.name_ranks(qn, tn)
            Q = ctx.upload(*readio.pack(qs), qr)
            T = ctx.upload(*readio.pack(ts), tr)
            try:
                if self.use_min_ref and self.target_num_bases > self.query_num_bases:
                    T.presketch(preset)                                   # streamed set: sketched beside the index build
                    ix = engine.Index(ctx, Q, preset)                     # index = query set
                    counts = ix.overlap_inverse(T, self.remove_internal, self.max_overhang_ratio)
                    no_mapping = int((counts == 0).sum())                 # twoset.rs:545-569
                    lens = Q.lens
                else:
                    Q.presketch(preset)
                    ix = engine.Index(ctx, T, preset)
                    counts, has = ix.overlap_twoset(Q, self.remove_internal, self.max_overhang_ratio)
                    no_mapping = int((has == 0).sum())                    # twoset.rs:303-309
                    lens = Q.lens
            except _ffi.LrgeHipError as e:
                kind = {_ffi.ERR_MAP: "MapError", _ffi.ERR_DUPLICATE_ID: "DuplicateReadIdentifier",
                        _ffi.ERR_TOO_MANY: "TooManyReadsError"}.get(e.code, "ThreadError")
                raise LrgeError(kind, str(e))
            self.timings = ctx.timings()
            est = ctx.estimates(counts, lens, float(avg_target_len), self.target_num_reads, 100)
            if no_mapping:
                log.info("%d (%.2f%%) query read(s) did not overlap any target reads", no_mapping,
                         100.0 * no_mapping / self.query_num_reads)
            return est, no_mapping
        finally:
            for h in (ix, Q, T):          # handles go before their context (lrge_hip_index_free / _seqset_free)
                if h is not None:
                    h.free()
            ctx.close()
