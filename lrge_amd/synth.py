"""Synthetic long-read sets for the BASELINE.json configs.

The reference publishes no benchmark inputs, so every config is a seeded synthetic set:
uniform random genome, reads sampled uniformly on either strand, lognormal (ONT) or normal (HiFi)
lengths, independent per-base substitution / insertion / deletion errors.  numpy's PCG64 bit
generator is used throughout, so a (config, seed) pair is reproducible on the same image.
"""
from dataclasses import dataclass

import numpy as np

_COMP = np.zeros(256, dtype=np.uint8)
for _a, _b in zip(b"ACGTN", b"TGCAN"):
    _COMP[_a] = _b
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_CODE = np.full(256, 4, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    _CODE[_c] = _i


@dataclass
class ReadBatch:
    """Reads as the overlap API takes them: concatenated ASCII + offsets + names."""
    bases: np.ndarray      # uint8 ASCII, concatenated
    offsets: np.ndarray    # uint64 [n+1]
    names: list            # list[bytes]
    starts: np.ndarray     # int64 genome start of each read (truth)
    ends: np.ndarray       # int64 genome end (exclusive) of the error-free source interval
    strands: np.ndarray    # int8

    @property
    def n(self):
        return len(self.names)

    def lens(self):
        return np.diff(self.offsets).astype(np.int64)

    def slice(self, lo, hi):
        o = self.offsets
        return ReadBatch(self.bases[int(o[lo]):int(o[hi])].copy(), (o[lo:hi + 1] - o[lo]).astype(np.uint64),
                         self.names[lo:hi], self.starts[lo:hi], self.ends[lo:hi], self.strands[lo:hi])

    def seqs(self):
        o = self.offsets
        return [self.bases[int(o[i]):int(o[i + 1])].tobytes() for i in range(self.n)]


PLATFORMS = {
    # total error split sub/ins/del; length model
    "ont": dict(sub=0.024, ins=0.016, dele=0.020, kind="lognormal", mu=np.log(6000.0), sigma=0.6,
                lo=500, hi=60000),
    "hifi": dict(sub=0.002, ins=0.0015, dele=0.0015, kind="normal", mu=15000.0, sigma=2000.0,
                 lo=5000, hi=25000),
}


def random_genome(size, seed, repeats=0.0, tandem=0.0):
    """Uniform random genome.  `repeats`: fraction of the genome overwritten with copies of 300-bp and 6-kb elements
    at 1-3 % divergence (exercises mid_occ and wrong-locus chains); `tandem`: fraction made of
    short tandem repeats (2-6 bp units, 200-2000 bp long: thousands of colliding seeds, the chaining worst case)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    g = _ACGT[rng.integers(0, 4, size=size, dtype=np.uint8)]
    if repeats > 0:
        budget = int(repeats * size)
        fams = [(300, _ACGT[rng.integers(0, 4, size=300, dtype=np.uint8)]) for _ in range(8)] + \
               [(6000, _ACGT[rng.integers(0, 4, size=6000, dtype=np.uint8)]) for _ in range(3)]
        while budget > 0:
            L, unit = fams[int(rng.integers(0, len(fams)))]
            if L >= size:
                break
            copy = unit.copy()
            div = rng.uniform(0.01, 0.03)
            m = rng.random(L) < div
            copy[m] = _ACGT[rng.integers(0, 4, size=int(m.sum()), dtype=np.uint8)]
            p = int(rng.integers(0, size - L))
            g[p:p + L] = copy
            budget -= L
    if tandem > 0:
        budget = int(tandem * size)
        while budget > 0:
            ul = int(rng.integers(2, 7)); L = int(rng.integers(200, 2001))
            if L >= size:
                break
            unit = _ACGT[rng.integers(0, 4, size=ul, dtype=np.uint8)]
            p = int(rng.integers(0, size - L))
            g[p:p + L] = np.tile(unit, L // ul + 1)[:L]
            budget -= L
    return g


def _read_lengths(rng, n, p, gsize):
    if p["kind"] == "lognormal":
        l = rng.lognormal(p["mu"], p["sigma"], size=n)
    else:
        l = rng.normal(p["mu"], p["sigma"], size=n)
    l = np.clip(l, p["lo"], min(p["hi"], gsize)).astype(np.int64)
    return l


def sample_reads(genome, n, platform="ont", seed=1, name_prefix="r", name_start=0, n_rate=0.0):
    """Sample n error-laden reads.  seed drives placement (seed) and errors (seed+1)."""
    p = PLATFORMS[platform]
    g = len(genome)
    rng = np.random.Generator(np.random.PCG64(seed))
    erng = np.random.Generator(np.random.PCG64(seed + 1))
    lens = _read_lengths(rng, n, p, g)
    starts = (rng.random(n) * (g - lens + 1)).astype(np.int64)
    strands = rng.integers(0, 2, size=n, dtype=np.int8)
    ps, pi, pd = p["sub"], p["ins"], p["dele"]
    chunks, out_lens = [], np.zeros(n, dtype=np.int64)
    for i in range(n):
        src = genome[starts[i]:starts[i] + lens[i]]
        if strands[i]:
            src = _COMP[src[::-1]]
        u = erng.random(len(src))
        is_sub = u < ps
        is_ins = (u >= ps) & (u < ps + pi)
        is_del = (u >= ps + pi) & (u < ps + pi + pd)
        b = src.copy()
        if is_sub.any():
            code = _CODE[b[is_sub]]
            b[is_sub] = _ACGT[(code + erng.integers(1, 4, size=code.size, dtype=np.uint8)) & 3]
        reps = np.ones(len(src), dtype=np.int64)
        reps[is_del] = 0
        reps[is_ins] = 2
        o = np.repeat(b, reps)
        if is_ins.any():  # second copy of an "ins" base becomes a random base
            idx = np.cumsum(reps)[is_ins] - 1
            o[idx] = _ACGT[erng.integers(0, 4, size=idx.size, dtype=np.uint8)]
        if n_rate > 0:
            nm = erng.random(len(o)) < n_rate
            o[nm] = ord("N")
        if len(o) == 0:
            o = src[:1].copy()
        chunks.append(o)
        out_lens[i] = len(o)
    offsets = np.zeros(n + 1, dtype=np.uint64)
    np.cumsum(out_lens, out=offsets[1:])
    bases = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    names = [b"%s%08d" % (name_prefix.encode(), name_start + i) for i in range(n)]
    return ReadBatch(bases, offsets, names, starts, starts + lens, strands)


def _block(args):
    genome, nb, platform, seed, prefix, start, n_rate, spill = args
    rb = sample_reads(genome, nb, platform, seed=seed, name_prefix=prefix, name_start=start, n_rate=n_rate)
    if spill is None:
        return rb
    # large sets: hand the arrays over through tmpfs files instead of pickling gigabytes through the pool's pipes
    path = "%s/blk_%010d" % (spill, start)
    np.save(path + "_b.npy", rb.bases); np.save(path + "_o.npy", rb.offsets)
    np.save(path + "_s.npy", np.stack([rb.starts, rb.ends, rb.strands.astype(np.int64)]))
    return path, rb.names


def sample_reads_parallel(genome, n, platform="ont", seed=1, name_prefix="r", block=4096, procs=None):
    """The same read model for sets of hundreds of thousands of reads: blocks of `block` reads drawn by worker processes,
    block b with seed + 1000003 * b (so the set depends on (seed, block) but not on the number of processes)."""
    import multiprocessing as mp
    import os
    import shutil
    import tempfile
    spill = tempfile.mkdtemp(prefix="lrge_synth_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        jobs = [(genome, min(block, n - s0), platform, seed + 1000003 * (s0 // block), name_prefix, s0, 0.0, spill) for s0 in range(0, n, block)]
        procs = procs or min(len(jobs), max(1, (os.cpu_count() or 2) // 2), 96)
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_block, jobs, chunksize=1)
        offs = [np.load(p + "_o.npy") for p, _ in res]
        lens = np.concatenate([np.diff(o).astype(np.int64) for o in offs])
        offsets = np.zeros(n + 1, dtype=np.uint64)
        np.cumsum(lens, out=offsets[1:])
        bases = np.empty(int(offsets[-1]), dtype=np.uint8)
        pos = 0
        meta = []
        for (p, _), o in zip(res, offs):
            b = np.load(p + "_b.npy")
            bases[pos:pos + len(b)] = b
            pos += len(b)
            meta.append(np.load(p + "_s.npy"))
        meta = np.concatenate(meta, axis=1)
        names = [nm for _, nms in res for nm in nms]
        return ReadBatch(bases, offsets, names, meta[0], meta[1], meta[2].astype(np.int8))
    finally:
        shutil.rmtree(spill, ignore_errors=True)


# The five BASELINE.json configs.  "twoset": first Q reads are queries, the
# last T are targets (file order of Appendix C); "ava": all N reads.
CONFIGS = {
    "c2_bact_twoset": dict(genome=4_400_000, seed=4401, platform="ont", mode="twoset", Q=5000, T=10000),
    "c3_yeast_ava": dict(genome=12_000_000, seed=1201, platform="ont", mode="ava", N=20000),
    "c4_dmel_twoset": dict(genome=143_000_000, seed=14301, platform="ont", mode="twoset", Q=50000, T=100000),
    "c5_human_twoset": dict(genome=3_100_000_000, seed=31001, platform="hifi", mode="twoset", Q=100000, T=2000000, parallel=True),
    # C5 at one tenth of its size: same coverage (10x targets), fits the 2^32-entry limits of this round
    "c5_human_tenth": dict(genome=310_000_000, seed=31001, platform="hifi", mode="twoset", Q=10000, T=200000),
    # C5 at one quarter: 7.9 Gbases of targets -> a partitioned index (2 parts at the default limit); drawn in parallel blocks
    "c5_human_quarter": dict(genome=775_000_000, seed=31001, platform="hifi", mode="twoset", Q=25000, T=500000, parallel=True),
    "c5_human_half": dict(genome=1_550_000_000, seed=31001, platform="hifi", mode="twoset", Q=50000, T=1000000, parallel=True),
    # C2 on a repeat-rich genome (15 % interspersed 300-bp / 6-kb families, 2 % short tandem repeats): robustness run
    "c2_repeats": dict(genome=4_400_000, seed=4402, platform="ont", mode="twoset", Q=5000, T=10000, repeats=0.15, tandem=0.02),
    # reduced cases for tests / smoke
    "tiny_twoset": dict(genome=200_000, seed=77, platform="ont", mode="twoset", Q=60, T=300),
    "tiny_ava": dict(genome=100_000, seed=78, platform="ont", mode="ava", N=200),
    "tiny_hifi": dict(genome=300_000, seed=79, platform="hifi", mode="twoset", Q=20, T=120),
}


def make_config(name, scale=1.0):
    """Return (genome_size, queries, targets) for twoset or (genome_size, reads, None) for ava."""
    c = CONFIGS[name]
    gsize = int(c["genome"] * scale)
    genome = random_genome(gsize, c["seed"], c.get("repeats", 0.0), c.get("tandem", 0.0))
    if c["mode"] == "twoset":
        q, t = max(1, int(c["Q"] * scale)), max(1, int(c["T"] * scale))
        if c.get("parallel"):
            reads = sample_reads_parallel(genome, q + t, c["platform"], seed=c["seed"] + 1)
        else:
            reads = sample_reads(genome, q + t, c["platform"], seed=c["seed"] + 1)
        return gsize, reads.slice(0, q), reads.slice(q, q + t)
    n = max(2, int(c["N"] * scale))
    return gsize, sample_reads(genome, n, c["platform"], seed=c["seed"] + 1), None
