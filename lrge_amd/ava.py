"""AvaStrategy + Builder: host mirror of liblrge/src/ava.rs and ava/builder.rs on top of the C ABI."""
import logging

import numpy as np

from . import _ffi, engine, readio
from .estimate import Estimate, LrgeError
from .twoset import PLATFORM_PRESET, unique_random_set

log = logging.getLogger("lrge_amd")

DEFAULT_AVA_NUM_READS = 25_000   # ava.rs:62


class Builder:
    """ava/builder.rs:19-153."""

    def __init__(self):
        self._n = DEFAULT_AVA_NUM_READS
        self._remove_internal, self._ratio = False, 0.2
        self._threads, self._tmpdir, self._seed, self._platform, self._device = 1, None, None, "ont", 0

    def num_reads(self, n): self._n = int(n); return self
    def remove_internal(self, flag, max_overhang_ratio=0.2):
        self._remove_internal = bool(flag)
        if flag:
            self._ratio = float(max_overhang_ratio)
        return self
    def threads(self, n): self._threads = int(n); return self
    def tmpdir(self, d): self._tmpdir = d; return self
    def seed(self, s): self._seed = s; return self
    def platform(self, p): self._platform = str(p).lower(); return self
    def device(self, d): self._device = int(d); return self

    def build(self, input_):
        return AvaStrategy(input_, self)


class AvaStrategy(Estimate):
    def __init__(self, input_, builder=None):
        b = builder or Builder()
        self.input, self.num_reads = input_, b._n
        self.remove_internal, self.max_overhang_ratio = b._remove_internal, b._ratio
        self.threads, self.seed, self.platform, self.device = b._threads, b._seed, b._platform, b._device
        self.num_bases = 0
        self.timings = None

    # ava.rs:108-161
    def subsample_reads(self):
        names, seqs = readio.load(self.input)
        n = len(names)
        if n > 0xFFFFFFFF:
            raise LrgeError("TooManyReadsError", "Number of reads in input file (%d) exceeds maximum allowed value" % n)
        if n < self.num_reads:
            log.warning("Number of reads in input file (%d) is less than the number requested (%d)", n, self.num_reads)
            self.num_reads = n
        keep = set(unique_random_set(self.num_reads, n, self.seed).tolist())
        idx = [i for i in range(n) if i in keep]
        rn, rs = [names[i] for i in idx], [seqs[i] for i in idx]
        self.num_bases = sum(map(len, rs))
        return rn, rs, self.num_bases

    # ava.rs:369-382 + :165-366
    def generate_estimates(self):
        rn, rs, sum_len = self.subsample_reads()
        try:
            preset = PLATFORM_PRESET[self.platform]
        except KeyError:
            raise LrgeError("InvalidPlatform", self.platform)
        ctx = engine.Context(self.device)
        R = ix = None
        try:
            (ranks,) = engine.name_ranks(rn)
            R = ctx.upload(*readio.pack(rs), ranks)
            try:
                R.presketch(preset)
                ix = engine.Index(ctx, R, preset)
                counts = ix.overlap_ava(self.remove_internal, self.max_overhang_ratio)
            except _ffi.LrgeHipError as e:
                kind = {_ffi.ERR_MAP: "MapError", _ffi.ERR_DUPLICATE_ID: "DuplicateReadIdentifier",
                        _ffi.ERR_TOO_MANY: "TooManyReadsError"}.get(e.code, "ThreadError")
                raise LrgeError(kind, str(e))
            self.timings = ctx.timings()
            n_target = self.num_reads - 1                                   # ava.rs:339-346
            avg = np.float32(sum_len) / np.float32(n_target) if n_target > 0 else np.float32(0)
            est = ctx.estimates(counts, R.lens, float(avg), n_target, 100)
            no_mapping = int((counts == 0).sum())                           # ava.rs:329-331
            if no_mapping:
                log.info("%d (%.2f%%) read(s) did not overlap any other reads", no_mapping, 100.0 * no_mapping / self.num_reads)
            return est, no_mapping
        finally:
            for h in (ix, R):             # handles go before their context
                if h is not None:
                    h.free()
            ctx.close()
