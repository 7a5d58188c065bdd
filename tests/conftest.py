import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def has_gpu():
    try:
        import ctypes as C
        from lrge_amd import _ffi
        n = C.c_int()
        return _ffi.lib().lrge_hip_device_count(C.byref(n)) == 0 and n.value > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def ctx():
    from lrge_amd import engine
    c = engine.Context(0)   # raises (no CPU fallback) when liblrge_hip.so or the device is missing
    yield c
    c.close()


class _Knobs:
    """Options of the session context set for one test (lrge_hip_ctx_set_option) and cleared afterwards."""

    def __init__(self, ctx):
        self.ctx, self.touched = ctx, set()

    def set(self, name, value):
        self.ctx.set_option(name, str(value))
        self.touched.add(name)

    def unset(self, name):
        self.ctx.set_option(name, None)


@pytest.fixture
def knobs(ctx):
    k = _Knobs(ctx)
    yield k
    for name in k.touched:
        ctx.set_option(name, None)


# ------------------------------------------------------------------------------------------
# shared small data sets (deterministic)
# ------------------------------------------------------------------------------------------
class DataSet:
    def __init__(self, name, queries, targets, platform):
        self.name, self.q, self.t, self.platform = name, queries, targets, platform


def _edge_reads(rng, genome, platform):
    """Hand-made reads exercising the edge cases of mm_sketch / skip_seed."""
    from lrge_amd import synth
    g = genome
    seqs = []
    seqs.append(g[1000:1012].tobytes())                                   # shorter than k
    seqs.append(g[2000:2019].tobytes())                                   # exactly w+k-1 (ont)
    s = g[3000:6000].copy(); s[::97] = ord("N"); seqs.append(s.tobytes())  # N-rich
    s = g[7000:9000].copy(); s[500:540] = ord("A"); s[900:1300] = ord("T"); seqs.append(s.tobytes())  # homopolymers
    seqs.append(b"ACGT" * 300)                                            # tandem repeat: query occurrence filter
    seqs.append((g[10000:10040].tobytes()) * 40)                          # 40-mer repeated 40x
    seqs.append(g[12000:15000].tobytes().lower())                         # lower case
    s = g[16000:18000].copy(); s[100] = ord("R"); s[101] = ord("u"); s[102] = ord("-"); seqs.append(s.tobytes())  # IUPAC / U
    seqs.append(b"N" * 50 + g[20000:21000].tobytes() + b"N" * 7)           # leading / trailing N
    seqs.append(b"A" * 600)                                               # one giant run (HPC span >= 256)
    return seqs


@pytest.fixture(scope="session")
def tiny_ont():
    from lrge_amd import synth
    gsize, q, t = synth.make_config("tiny_twoset")
    return DataSet("tiny_ont", q, t, "ont")


@pytest.fixture(scope="session")
def tiny_hifi():
    from lrge_amd import synth
    gsize, q, t = synth.make_config("tiny_hifi")
    return DataSet("tiny_hifi", q, t, "hifi")


@pytest.fixture(scope="session")
def tiny_ava():
    from lrge_amd import synth
    gsize, reads, _ = synth.make_config("tiny_ava")
    return reads


@pytest.fixture(scope="session")
def edge_set():
    """A target set = sampled reads + edge-case reads; queries = other sampled reads + the same edge
    reads under different names (so self/diagonal logic is NOT triggered) ."""
    from lrge_amd import synth
    genome = synth.random_genome(60_000, 4242)
    rng = np.random.Generator(np.random.PCG64(99))
    base = synth.sample_reads(genome, 150, "ont", seed=5, n_rate=1e-4)
    edge = _edge_reads(rng, genome, "ont")
    tseqs = base.seqs()[40:] + edge
    tnames = base.names[40:] + [b"edgeT%02d" % i for i in range(len(edge))]
    qseqs = base.seqs()[:40] + edge
    qnames = base.names[:40] + [b"edgeQ%02d" % i for i in range(len(edge))]
    return (qseqs, qnames, tseqs, tnames)


def to_arrays(seqs):
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    b = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if int(offs[-1]) else np.zeros(0, dtype=np.uint8)
    return b, offs


def write_unaligned_bam(path, names, seqs, bgzf_block=60000):
    """A uBAM (flag 4, no references) holding the given reads: what `samtools import` / a basecaller writes.  Used to
    rebuild the reference's toy.bam test input from its FASTA conversion (tests/golden/toy_reads.fa.gz) -- same reads,
    same order -- instead of shipping the reference's file.  Compressed as a sequence of gzip members (BGZF-like)."""
    import gzip
    import struct
    code = {c: i for i, c in enumerate(b"=ACMGRSVTWYHKDBN")}
    text = b"@HD\tVN:1.6\tSO:unknown\n"
    body = [b"BAM\x01", struct.pack("<i", len(text)), text, struct.pack("<i", 0)]
    for n, s in zip(names, seqs):
        up = s.upper()
        nib = [code.get(c, 15) for c in up]
        if len(nib) & 1:
            nib.append(0)
        packed = bytes((nib[i] << 4) | nib[i + 1] for i in range(0, len(nib), 2))
        name = n + b"\0"
        rec = struct.pack("<iiBBHHHiiii", -1, -1, len(name), 0, 4680, 0, 4, len(s), -1, -1, 0) + name + packed + b"\xff" * len(s)
        body.append(struct.pack("<i", len(rec)) + rec)
    raw = b"".join(body)
    with open(path, "wb") as fh:
        for i in range(0, len(raw), bgzf_block):
            fh.write(gzip.compress(raw[i:i + bgzf_block], 6))
        fh.write(gzip.compress(b""))
