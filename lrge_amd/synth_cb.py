"""Counter-based synthetic read sets: the BASELINE.json configurations too large for lrge_amd.synth's sequential
generator (the H. sapiens-scale set is 31.5 Gbases: ~25 minutes of numpy), produced in seconds.

TEST / BENCH INFRASTRUCTURE (tools/synth/): every base is a pure integer function of (seed, read index, position)
(tools/synth/cb_core.h), so
  * any read -- or any sample of reads -- can be produced alone on the host (`host_reads`: cb_host.c, OpenMP), which is
    what the CPU oracle is fed with, and
  * the whole set is written straight into HBM by the device twin (`device_reads`: cb_hip.hip) without ever existing in
    host memory, bit-identical to the host twin (tests/test_gpu_synth_cb.py).
The read model is lrge_amd.synth's (uniform genome, uniform placement on either strand, platform length distribution,
independent substitution / insertion / deletion errors); the two generators do not produce the same reads.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import synth

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(_HERE), "tools", "synth")
LIB_DIR = os.path.join(_HERE, "_lib")
HOST_LIB = os.path.join(LIB_DIR, "libcbgen_host.so")
HIP_LIB = os.path.join(LIB_DIR, "libcbgen_hip.so")


class Params(C.Structure):
    _fields_ = [("gsize", C.c_uint64), ("gseed", C.c_uint64), ("rseed", C.c_uint64),
                ("t_sub", C.c_uint32), ("t_ins", C.c_uint32), ("t_del", C.c_uint32), ("pad", C.c_uint32)]


def _stale(lib, srcs):
    return not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(os.path.join(SRC, s)) for s in srcs)


def build(force=False):
    """gcc for the host twin, hipcc (gfx950) for the device twin; both into lrge_amd/_lib/ beside the product library."""
    os.makedirs(LIB_DIR, exist_ok=True)
    if force or _stale(HOST_LIB, ["cb_host.c", "cb_core.h"]):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-Wall", "-Wextra", "-fPIC", "-fopenmp", "-shared", "-o", HOST_LIB,
                               os.path.join(SRC, "cb_host.c")])
    if force or _stale(HIP_LIB, ["cb_hip.hip", "cb_core.h"]):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", HIP_LIB,
                               os.path.join(SRC, "cb_hip.hip")])
    return HOST_LIB, HIP_LIB


_libs = {}


def _host():
    if "h" not in _libs:
        if not os.path.exists(HOST_LIB):
            build()
        L = C.CDLL(HOST_LIB)
        for f in (L.cb_host_meta, L.cb_host_write, L.cb_host_genome):
            f.restype = C.c_int
        _libs["h"] = L
    return _libs["h"]


def _hip():
    if "d" not in _libs:
        if not os.path.exists(HIP_LIB):
            raise RuntimeError("libcbgen_hip.so is missing: run __graft_entry__.build()")
        L = C.CDLL(HIP_LIB)
        for f in (L.cb_hip_count, L.cb_hip_write, L.cb_hip_malloc, L.cb_hip_free, L.cb_hip_to_host, L.cb_hip_mem_info):
            f.restype = C.c_int
        _libs["d"] = L
    return _libs["d"]


def length_table(platform, gsize):
    """65 536-entry inverse CDF of the platform's read-length distribution (lrge_amd.synth.PLATFORMS), clipped like
    synth._read_lengths.  Data handed to both twins, so its floating-point provenance does not matter to their identity."""
    from scipy.stats import norm
    p = synth.PLATFORMS[platform]
    u = (np.arange(65536, dtype=np.float64) + 0.5) / 65536.0
    z = norm.ppf(u)
    l = np.exp(p["mu"] + p["sigma"] * z) if p["kind"] == "lognormal" else p["mu"] + p["sigma"] * z
    return np.ascontiguousarray(np.clip(l, p["lo"], min(p["hi"], gsize)).astype(np.uint32))


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Spec:
    """One synthetic read universe: reads 0, 1, 2, ... of a genome."""

    def __init__(self, gsize, seed, platform):
        p = synth.PLATFORMS[platform]
        self.gsize, self.seed, self.platform = int(gsize), int(seed), platform
        self.params = Params(self.gsize, (seed * 0x2545F4914F6CDD1D + 1) & (2**64 - 1), (seed * 0x9FB21C651E98DF25 + 7) & (2**64 - 1),
                             int(p["sub"] * 2**32), int(p["ins"] * 2**32), int(p["dele"] * 2**32), 0)
        self.lentab = length_table(platform, self.gsize)

    # ---- host twin ----
    def meta(self, idx=None, first=0, n=None, out_len=False):
        """(len, start, strand[, emitted length]) of reads idx[] (or first .. first + n)."""
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.uint64); n = len(idx)
        ln = np.empty(n, np.uint32); st = np.empty(n, np.uint64); sd = np.empty(n, np.uint8)
        ol = np.empty(n, np.uint32) if out_len else None
        rc = _host().cb_host_meta(C.byref(self.params), _vp(self.lentab), _vp(idx), C.c_uint64(first), C.c_uint64(n),
                                  _vp(ln), _vp(st), _vp(sd), _vp(ol))
        assert rc == 0
        return (ln, st, sd, ol) if out_len else (ln, st, sd)

    def host_reads(self, idx=None, first=0, n=None, name_prefix="r"):
        """The reads idx[] (any subset, any order) as a synth.ReadBatch, names r%08d of the read's index."""
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.uint64); n = len(idx)
        ln, st, sd, ol = self.meta(idx, first, n, out_len=True)
        offsets = np.zeros(n + 1, np.uint64)
        np.cumsum(ol, out=offsets[1:])
        bases = np.empty(int(offsets[-1]), np.uint8)
        rc = _host().cb_host_write(C.byref(self.params), _vp(self.lentab), _vp(idx), C.c_uint64(first), C.c_uint64(n), _vp(offsets), _vp(bases))
        assert rc == 0
        ids = idx if idx is not None else np.arange(first, first + n, dtype=np.uint64)
        names = [b"%s%08d" % (name_prefix.encode(), int(i)) for i in ids]
        return synth.ReadBatch(bases, offsets, names, st.astype(np.int64), st.astype(np.int64) + ln.astype(np.int64), sd.astype(np.int8))

    def genome(self, pos, n):
        out = np.empty(n, np.uint8)
        assert _host().cb_host_genome(C.byref(self.params), C.c_uint64(pos), C.c_uint64(n), _vp(out)) == 0
        return out

    # ---- device twin ----
    def emitted_lens(self, first, n, device=0):
        """Emitted length of reads [first, first + n) (the counting pass of the device twin alone)."""
        ol = np.empty(n, np.uint32)
        rc = _hip().cb_hip_count(device, C.byref(self.params), _vp(self.lentab), C.c_uint64(first), C.c_uint64(n), _vp(ol))
        if rc:
            raise RuntimeError("cb_hip_count: HIP error %d" % rc)
        return ol

    def device_reads(self, first, n, device=0):
        """Reads [first, first + n) written into HBM.  Returns a DeviceReads (device pointer + host offsets / truth)."""
        L = _hip()
        ol = np.empty(n, np.uint32)
        rc = L.cb_hip_count(device, C.byref(self.params), _vp(self.lentab), C.c_uint64(first), C.c_uint64(n), _vp(ol))
        if rc:
            raise RuntimeError("cb_hip_count: HIP error %d" % rc)
        offsets = np.zeros(n + 1, np.uint64)
        np.cumsum(ol, out=offsets[1:])
        ptr = C.c_void_p()
        rc = L.cb_hip_malloc(device, C.c_uint64(int(offsets[-1])), C.byref(ptr))
        if rc:
            raise RuntimeError("cb_hip_malloc(%d bytes): HIP error %d" % (int(offsets[-1]), rc))
        rc = L.cb_hip_write(device, C.byref(self.params), _vp(self.lentab), C.c_uint64(first), C.c_uint64(n), _vp(offsets), ptr)
        if rc:
            L.cb_hip_free(device, ptr)
            raise RuntimeError("cb_hip_write: HIP error %d" % rc)
        ln, st, sd = self.meta(first=first, n=n)
        return DeviceReads(self, device, ptr.value, offsets, first, n, ln, st, sd)


class DeviceReads:
    """A read set resident in HBM as ASCII (what lrge_hip_seqset_upload takes as a device source)."""

    def __init__(self, spec, device, ptr, offsets, first, n, src_len, starts, strands):
        self.spec, self.device, self.ptr, self.offsets, self.first, self.n = spec, device, ptr, offsets, first, n
        self.starts = starts.astype(np.int64); self.ends = self.starts + src_len.astype(np.int64); self.strands = strands.astype(np.int8)

    def lens(self):
        return np.diff(self.offsets).astype(np.int64)

    @property
    def total_bases(self):
        return int(self.offsets[-1])

    def name_ranks(self):
        """Names are r%08d of the read index: lexicographic order = numeric order, so the index itself is the rank."""
        return np.arange(self.first, self.first + self.n, dtype=np.uint32)

    def to_host(self, lo=0, hi=None):
        """ASCII of reads [lo, hi) copied back (tests)."""
        hi = self.n if hi is None else hi
        b0, b1 = int(self.offsets[lo]), int(self.offsets[hi])
        out = np.empty(b1 - b0, np.uint8)
        rc = _hip().cb_hip_to_host(self.device, _vp(out), C.c_void_p(self.ptr + b0), C.c_uint64(b1 - b0))
        if rc:
            raise RuntimeError("cb_hip_to_host: HIP error %d" % rc)
        return out

    def free(self):
        if self.ptr:
            _hip().cb_hip_free(self.device, C.c_void_p(self.ptr)); self.ptr = 0


def mem_info(device=0):
    f, t = C.c_uint64(), C.c_uint64()
    _hip().cb_hip_mem_info(device, C.byref(f), C.byref(t))
    return f.value, t.value


# Configurations (two-set: reads [0, Q) are the queries, [Q, Q + T) the targets -- file order of SURVEY.md Appendix C).
CONFIGS = {
    "c5_human_twoset": dict(genome=3_100_000_000, seed=31001, platform="hifi", Q=100000, T=2000000),   # BASELINE configs[4]
    "c5_human_half": dict(genome=1_550_000_000, seed=31001, platform="hifi", Q=50000, T=1000000),
    "c5_human_tenth": dict(genome=310_000_000, seed=31001, platform="hifi", Q=10000, T=200000),
    "c4_dmel_twoset": dict(genome=143_000_000, seed=14301, platform="ont", Q=50000, T=100000),
    "tiny_hifi": dict(genome=300_000, seed=79, platform="hifi", Q=20, T=120),
    "tiny_ont": dict(genome=200_000, seed=77, platform="ont", Q=60, T=300),
}


def spec_of(name, scale=1.0):
    c = CONFIGS[name]
    return Spec(int(c["genome"] * scale), c["seed"], c["platform"]), max(1, int(c["Q"] * scale)), max(1, int(c["T"] * scale))
