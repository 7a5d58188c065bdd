"""ctypes binding of the CPU oracle (oracle/lrge_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- nowhere else.
The product package (lrge_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblrge_oracle.so")

PRESET_AVA_ONT = 0
PRESET_AVA_PB = 1
SORT_STABLE = 0
SORT_MM2 = 1


class Opt(C.Structure):
    _fields_ = [
        ("k", C.c_int32), ("w", C.c_int32), ("is_hpc", C.c_int32), ("bucket_bits", C.c_int32),
        ("flag", C.c_int64),
        ("bw", C.c_int32), ("bw_long", C.c_int32), ("max_gap", C.c_int32), ("max_gap_ref", C.c_int32),
        ("max_chain_skip", C.c_int32), ("max_chain_iter", C.c_int32),
        ("min_cnt", C.c_int32), ("min_chain_score", C.c_int32), ("min_mid_occ", C.c_int32),
        ("max_mid_occ", C.c_int32), ("mid_occ", C.c_int32), ("seed", C.c_int32),
        ("mid_occ_frac", C.c_float), ("q_occ_frac", C.c_float), ("chain_gap_scale", C.c_float),
        ("chain_skip_scale", C.c_float),
        ("sort_mode", C.c_int32),
    ]


MM128 = np.dtype([("x", "<u8"), ("y", "<u8")])
REG = np.dtype([("rid", "<i4"), ("rev", "<i4"), ("score", "<i4"), ("cnt", "<i4"),
                ("rs", "<i4"), ("re", "<i4"), ("qs", "<i4"), ("qe", "<i4"),
                ("mlen", "<i4"), ("blen", "<i4"), ("dv", "<f4"), ("rep_len", "<i4")])


def build(force=False):
    """Compile oracle/lrge_oracle.c with gcc (called by __graft_entry__.build())."""
    src = os.path.join(_HERE, "lrge_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src),
                                                   os.path.getmtime(os.path.join(_HERE, "lrge_oracle.h")))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "liblrge_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def host_cpus():
    """CPUs the host GRANTS this process: the smaller of its affinity mask and its cgroup's CPU bandwidth limit (cpu.max).  The GPU
    boxes of this pool report 256 hardware threads and grant 16 CPUs (cpu.max = "1600000 100000"): 256 OpenMP threads there spend
    most of every scheduling period throttled (tools/cpu_port_scaling.py)."""
    try:
        q = float(len(os.sched_getaffinity(0)))
    except Exception:      # noqa: BLE001
        q = float(os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, per = f.read().split()[:2]
        if a != "max" and float(per) > 0:
            q = min(q, float(a) / float(per))
    except Exception:      # noqa: BLE001
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if quota > 0 and per > 0:
                q = min(q, quota / per)
        except Exception:      # noqa: BLE001
            pass
    return max(1.0, q)


def default_threads():
    """OpenMP threads the oracle runs with unless told otherwise: LO_THREADS, else the CPUs the host grants (measured on this pool's
    boxes, 16 granted CPUs under 256 hardware threads, H. sapiens-scale HiFi reads against 50 000 targets: 8 threads 2 708 reads/s,
    16: 5 337, 24: 5 311, 32: 5 058, 64: 4 403, 256: 2 646 -- profiles/r05_cpu_port_scaling.txt)."""
    if os.environ.get("LO_THREADS"):
        return max(1, int(os.environ["LO_THREADS"]))
    return int(max(1, min(os.cpu_count() or 1, round(host_cpus()))))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.lo_set_default_threads.argtypes = [C.c_int]
        L.lo_set_default_threads(default_threads())
        vp, u64p, u32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)
        L.lo_opt_init.argtypes = [C.POINTER(Opt), C.c_int, C.c_int]
        L.lo_sketch.restype = C.c_int64
        L.lo_sketch.argtypes = [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_int32, vp, C.c_int64]
        L.lo_hash64.restype = C.c_uint64
        L.lo_hash64.argtypes = [C.c_uint64, C.c_uint64]
        L.lo_index_build.restype = vp
        L.lo_index_build.argtypes = [vp, vp, C.c_uint32, vp, C.POINTER(Opt)]
        L.lo_index_free.argtypes = [vp]
        L.lo_index_mid_occ.restype = C.c_int32
        L.lo_index_mid_occ.argtypes = [vp]
        L.lo_index_n_minimizers.restype = C.c_uint64
        L.lo_index_n_minimizers.argtypes = [vp]
        L.lo_index_n_keys.restype = C.c_uint64
        L.lo_index_n_keys.argtypes = [vp]
        L.lo_index_get.restype = C.c_int32
        L.lo_index_get.argtypes = [vp, C.c_uint64, C.POINTER(u64p)]
        L.lo_index_drop_keys.restype = C.c_uint64
        L.lo_index_drop_keys.argtypes = [vp, vp, C.c_uint64]
        L.lo_index_strip.argtypes = [vp]
        L.lo_index_keys_at_least.restype = C.c_uint64
        L.lo_index_keys_at_least.argtypes = [vp, C.c_uint32, vp, C.c_uint64]
        L.lo_index_counts_of.argtypes = [vp, vp, C.c_uint64, vp]
        L.lo_index_export_key_counts.restype = C.c_uint64
        L.lo_index_export_key_counts.argtypes = [vp, vp, vp]
        L.lo_index_dump_minimizers.restype = C.c_uint64
        L.lo_index_dump_minimizers.argtypes = [vp, vp, C.c_uint64]
        L.lo_anchors.restype = C.c_int64
        L.lo_anchors.argtypes = [vp, C.POINTER(Opt), vp, C.c_int32, C.c_char_p, vp, C.c_int64]
        L.lo_map.restype = C.c_int32
        L.lo_map.argtypes = [vp, C.POINTER(Opt), vp, C.c_int32, C.c_char_p, vp, C.c_int32]
        L.lo_twoset_counts.restype = C.c_int
        L.lo_twoset_counts.argtypes = [vp, C.POINTER(Opt), vp, vp, C.c_uint32, vp, C.c_int, C.c_float, C.c_int, vp, vp]
        L.lo_inverse_counts.restype = C.c_int
        L.lo_inverse_counts.argtypes = [vp, C.POINTER(Opt), vp, vp, C.c_uint32, vp, C.c_int, C.c_float, C.c_int, vp]
        L.lo_ava_counts.restype = C.c_int
        L.lo_ava_counts.argtypes = [vp, C.POINTER(Opt), vp, vp, C.c_uint32, vp, C.c_int, C.c_float, C.c_int, vp]
        L.lo_per_read_estimate.restype = C.c_float
        L.lo_per_read_estimate.argtypes = [C.c_uint64, C.c_float, C.c_uint64, C.c_uint64, C.c_uint32]
        L.lo_median.restype = C.c_int
        L.lo_median.argtypes = [vp, C.c_uint64, C.c_int, C.c_int, C.c_float, C.c_int, C.c_float,
                                C.POINTER(C.c_float * 3), C.POINTER(C.c_int * 3)]
        L.lo_is_internal.restype = C.c_int
        L.lo_is_internal.argtypes = [C.c_int32] * 3 + [C.c_int] + [C.c_int32] * 3 + [C.c_float]
        L.lo_inverse_skip.restype = C.c_int
        L.lo_inverse_skip.argtypes = [C.c_int32] * 3 + [C.c_int] + [C.c_int32] * 3 + [C.c_float]
        L.lo_sort128x.argtypes = [vp, C.c_int64, C.c_int]
        L.lo_kstat_new.restype = vp
        L.lo_kstat_new.argtypes = [C.POINTER(Opt)]
        L.lo_kstat_add.argtypes = [vp, vp, vp, C.c_uint32, C.c_int]
        L.lo_kstat_finish.argtypes = [vp, C.c_int, u64p, u64p, C.POINTER(C.c_int32)]
        L.lo_kstat_free.argtypes = [vp]
        L.lo_ridx_new.restype = vp
        L.lo_ridx_new.argtypes = [C.POINTER(Opt), vp, vp, C.c_uint32]
        L.lo_ridx_add.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_int]
        L.lo_ridx_finish.restype = vp
        L.lo_ridx_finish.argtypes = [vp, C.POINTER(Opt), C.c_int32, C.c_int]
        L.lo_ridx_free.argtypes = [vp]
        for f in (L.lo_ridx_n_sample_keys, L.lo_ridx_n_minimizers_seen, L.lo_ridx_n_kept):
            f.restype = C.c_uint64
            f.argtypes = [vp]
        _lib = L
    return _lib


def make_opt(preset=PRESET_AVA_ONT, dual=True, sort_mode=SORT_STABLE):
    o = Opt()
    lib().lo_opt_init(C.byref(o), preset, 1 if dual else 0)
    o.sort_mode = sort_mode
    return o


def _names_array(names):
    arr = (C.c_char_p * max(len(names), 1))()
    for i, n in enumerate(names):
        arr[i] = n if isinstance(n, bytes) else n.encode()
    return arr


def sketch(seq, w, k, rid=0, is_hpc=False):
    seq = bytes(seq)
    n = lib().lo_sketch(seq, len(seq), w, k, rid, int(is_hpc), None, 0)
    out = np.zeros(max(n, 1), dtype=MM128)
    lib().lo_sketch(seq, len(seq), w, k, rid, int(is_hpc), out.ctypes.data, n)
    return out[:n]


class ReadSet:
    """Concatenated ASCII bases + offsets + names (what Aligner::map / mm_idx_gen see)."""

    def __init__(self, seqs, names):
        self.n = len(seqs)
        self.names = [n if isinstance(n, bytes) else n.encode() for n in names]
        lens = np.array([len(s) for s in seqs], dtype=np.uint64)
        self.offsets = np.zeros(self.n + 1, dtype=np.uint64)
        np.cumsum(lens, out=self.offsets[1:])
        self.bases = np.frombuffer(b"".join(bytes(s) for s in seqs), dtype=np.uint8).copy() \
            if self.n and int(self.offsets[-1]) > 0 else np.zeros(1, dtype=np.uint8)
        self._cnames = _names_array(self.names)

    @classmethod
    def from_arrays(cls, bases, offsets, names):
        """The same set from its concatenated bases and offsets as they are (no per-read Python objects: sets of 10^6 reads)."""
        self = cls.__new__(cls)
        self.n = len(offsets) - 1
        self.names = [n if isinstance(n, bytes) else n.encode() for n in names]
        assert len(self.names) == self.n
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.bases = np.ascontiguousarray(bases, dtype=np.uint8) if len(bases) else np.zeros(1, dtype=np.uint8)
        self._cnames = _names_array(self.names)
        return self

    def seq(self, i):
        return self.bases[int(self.offsets[i]):int(self.offsets[i + 1])].tobytes()


class Index:
    def __init__(self, rs, opt, _handle=None):
        self.rs = rs
        self.opt = opt
        self.h = _handle if _handle is not None else lib().lo_index_build(rs.bases.ctypes.data, rs.offsets.ctypes.data, rs.n,
                                                                          C.cast(rs._cnames, C.c_void_p), C.byref(opt))

    def __del__(self):
        if getattr(self, "h", None):
            lib().lo_index_free(self.h)
            self.h = None

    @property
    def mid_occ(self):
        return lib().lo_index_mid_occ(self.h)

    @property
    def n_minimizers(self):
        return lib().lo_index_n_minimizers(self.h)

    @property
    def n_keys(self):
        return lib().lo_index_n_keys(self.h)

    def minimizers(self):
        n = self.n_minimizers
        out = np.zeros(max(n, 1), dtype=MM128)
        lib().lo_index_dump_minimizers(self.h, out.ctypes.data, n)
        return out[:n]

    def strip(self):
        """Drop the sketch-order copy of the minimizers (minimizers() returns nothing afterwards): a part of a set indexed in parts."""
        lib().lo_index_strip(self.h)

    def keys_at_least(self, min_count):
        n = lib().lo_index_keys_at_least(self.h, int(min_count), None, 0)
        out = np.zeros(max(n, 1), dtype=np.uint64)
        lib().lo_index_keys_at_least(self.h, int(min_count), out.ctypes.data, n)
        return out[:n]

    def counts_of(self, keys):
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        out = np.zeros(max(k.size, 1), dtype=np.uint32)
        if k.size:
            lib().lo_index_counts_of(self.h, k.ctypes.data, k.size, out.ctypes.data)
        return out[:k.size]

    def key_counts(self):
        """(distinct keys ascending, their local counts saturated at 255)"""
        n = self.n_keys
        k = np.zeros(max(n, 1), dtype=np.uint64); c = np.zeros(max(n, 1), dtype=np.uint8)
        lib().lo_index_export_key_counts(self.h, k.ctypes.data, c.ctypes.data)
        return k[:n], c[:n]

    def drop_keys(self, keys):
        """This index is a shard of a larger target set: `keys` (hashes) are too frequent over the whole set."""
        k = np.ascontiguousarray(keys, dtype=np.uint64)
        return lib().lo_index_drop_keys(self.h, k.ctypes.data if k.size else None, k.size)

    def get(self, minier):
        p = C.POINTER(C.c_uint64)()
        n = lib().lo_index_get(self.h, minier, C.byref(p))
        return np.array([p[i] for i in range(n)], dtype=np.uint64)

    def anchors(self, seq, qname):
        seq = bytes(seq)
        qn = qname if (qname is None or isinstance(qname, bytes)) else qname.encode()
        n = lib().lo_anchors(self.h, C.byref(self.opt), seq, len(seq), qn, None, 0)
        out = np.zeros(max(n, 1), dtype=MM128)
        lib().lo_anchors(self.h, C.byref(self.opt), seq, len(seq), qn, out.ctypes.data, n)
        return out[:n]

    def map(self, seq, qname):
        seq = bytes(seq)
        qn = qname if (qname is None or isinstance(qname, bytes)) else qname.encode()
        cap = 4096
        out = np.zeros(cap, dtype=REG)
        n = lib().lo_map(self.h, C.byref(self.opt), seq, len(seq), qn, out.ctypes.data, cap)
        if n > cap:
            out = np.zeros(n, dtype=REG)
            n = lib().lo_map(self.h, C.byref(self.opt), seq, len(seq), qn, out.ctypes.data, n)
        return out[:n]

    def twoset_counts(self, qs, remove_internal=False, ratio=0.2, threads=0):
        counts = np.zeros(max(qs.n, 1), dtype=np.uint32)
        has = np.zeros(max(qs.n, 1), dtype=np.uint32)
        rc = lib().lo_twoset_counts(self.h, C.byref(self.opt), qs.bases.ctypes.data, qs.offsets.ctypes.data,
                                    qs.n, C.cast(qs._cnames, C.c_void_p), int(remove_internal), ratio, threads,
                                    counts.ctypes.data, has.ctypes.data)
        return rc, counts[:qs.n], has[:qs.n]

    def inverse_counts(self, ts, remove_internal=False, ratio=0.2, threads=0):
        counts = np.zeros(max(self.rs.n, 1), dtype=np.uint32)
        rc = lib().lo_inverse_counts(self.h, C.byref(self.opt), ts.bases.ctypes.data, ts.offsets.ctypes.data,
                                     ts.n, C.cast(ts._cnames, C.c_void_p), int(remove_internal), ratio, threads,
                                     counts.ctypes.data)
        return rc, counts[:self.rs.n]

    def ava_counts(self, remove_internal=False, ratio=0.2, threads=0):
        rs = self.rs
        counts = np.zeros(max(rs.n, 1), dtype=np.uint32)
        rc = lib().lo_ava_counts(self.h, C.byref(self.opt), rs.bases.ctypes.data, rs.offsets.ctypes.data,
                                 rs.n, C.cast(rs._cnames, C.c_void_p), int(remove_internal), ratio, threads,
                                 counts.ctypes.data)
        return rc, counts[:rs.n]


class KeyStats:
    """n_minimizers / n_keys / mid_occ of a target set fed chunk by chunk (lo_kstat_*): what Index() reports for the
    whole set, for sets too large to index on the host in one piece."""

    def __init__(self, opt):
        self.h = lib().lo_kstat_new(C.byref(opt))

    def add(self, bases, offsets, threads=0):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        rc = lib().lo_kstat_add(self.h, bases.ctypes.data if bases.size else None, offsets.ctypes.data, offsets.size - 1, threads)
        assert rc == 0, rc

    def finish(self, threads=0):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_int32()
        rc = lib().lo_kstat_finish(self.h, threads, C.byref(a), C.byref(b), C.byref(c))
        assert rc == 0, rc
        return dict(n_minimizers=a.value, n_keys=b.value, mid_occ=c.value)

    def __del__(self):
        if getattr(self, "h", None):
            lib().lo_kstat_free(self.h)
            self.h = None


class RestrictedIndexBuilder:
    """lo_ridx_*: the index of a target set too large to hold on the host, restricted to the keys of a SAMPLE of query
    reads (complete position lists for exactly those keys => the answers mm_idx_get gives the sample are those of the full
    index).  Feed the targets chunk by chunk, in rid order; finish(mid_occ) hands back an Index that is valid for the sample
    queries only (twoset_counts / map / anchors); mid_occ is the whole set's (KeyStats / tests/golden/c5_full_index_stats.json)."""

    def __init__(self, opt, sample):
        self.opt = opt
        self.h = lib().lo_ridx_new(C.byref(opt), sample.bases.ctypes.data, sample.offsets.ctypes.data, sample.n)
        self.n_targets = 0

    def add(self, bases, offsets, names, threads=0):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        assert len(names) == n
        cn = _names_array(names)
        rc = lib().lo_ridx_add(self.h, bases.ctypes.data if bases.size else None, offsets.ctypes.data, n, C.cast(cn, C.c_void_p), threads)
        assert rc == 0, rc
        self.n_targets += n

    @property
    def n_minimizers_seen(self):
        return lib().lo_ridx_n_minimizers_seen(self.h)

    @property
    def n_kept(self):
        return lib().lo_ridx_n_kept(self.h)

    @property
    def n_sample_keys(self):
        return lib().lo_ridx_n_sample_keys(self.h)

    def finish(self, mid_occ, threads=0):
        h = lib().lo_ridx_finish(self.h, C.byref(self.opt), int(mid_occ), threads)
        assert h, "lo_ridx_finish: mid_occ must be the whole target set's (> 0)"
        self.h = None                # (consumed)

        class _Described:            # what Index consults of its read set: the number of reads (inverse_counts / ava_counts are not valid here)
            n = self.n_targets
        return Index(_Described(), self.opt, _handle=h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().lo_ridx_free(self.h)
            self.h = None


def per_read_estimate(read_len, avg_target_len, n_target_reads, n_ovlaps, thr):
    return float(lib().lo_per_read_estimate(read_len, avg_target_len, n_target_reads, n_ovlaps, thr))


def median(vals, finite=True, lower=None, upper=None):
    v = np.ascontiguousarray(vals, dtype=np.float32)
    out = (C.c_float * 3)()
    ok = (C.c_int * 3)()
    rc = lib().lo_median(v.ctypes.data if v.size else None, v.size, int(finite),
                         int(lower is not None), lower or 0.0, int(upper is not None), upper or 0.0,
                         C.byref(out), C.byref(ok))
    if rc != 0:
        raise ValueError("median(None, Some) is unsupported (the reference panics: estimate.rs:109)")
    return tuple(np.float32(out[i]) if ok[i] else None for i in range(3))


def is_internal(qlen, qs, qe, rev, tlen, ts, te, ratio):
    return bool(lib().lo_is_internal(qlen, qs, qe, int(rev), tlen, ts, te, ratio))


def inverse_skip(qlen, qs, qe, rev, tlen, ts, te, ratio):
    return bool(lib().lo_inverse_skip(qlen, qs, qe, int(rev), tlen, ts, te, ratio))


def sort128x(arr, mode):
    a = np.ascontiguousarray(arr, dtype=MM128).copy()
    lib().lo_sort128x(a.ctypes.data, a.size, mode)
    return a
