"""Forward counts of a SAMPLE of query reads at full H. sapiens scale (BASELINE configs[4]: -Q 100 000 -T 2 000 000), from the
CPU oracle.  TEST INFRASTRUCTURE ONLY (imported by tests/ and by bench.py's cpu_baseline / parity leg).

No host holds the oracle's index of 2 000 000 HiFi targets (7.5 G minimizers), so the targets stream through the oracle's
mm_sketch chunk by chunk into a RestrictedIndexBuilder (oracle.py; lo_ridx_* in lrge_oracle.c): complete position lists for
exactly the keys the sample queries carry, which is all mm_idx_get is ever asked by them (tests/test_oracle_restricted.py checks
the restricted index against the full one where the full one fits).  mid_occ -- a statistic over ALL keys -- comes from the
oracle's own KeyStats pass over the same reads, committed as tests/golden/c5_full_index_stats.json
(tests/golden/make_c5_fixture.py).

The reads are those of the counter-based generator (lrge_amd/synth_cb.py); `source="device"` copies what the device twin wrote
(bit-identical to the host twin: tests/test_synth_cb.py, and checked again by the callers on a few hundred reads), which costs a
PCIe copy instead of ~9 Mbases/s per core of host generation.
"""
import json
import os
import time
import zlib

import numpy as np

from . import oracle as O

_GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "c5_full_index_stats.json")


def fixture_stats(spec, Q, T, preset_name="ava-pb"):
    """n_minimizers / n_keys / mid_occ of the full target set from the committed oracle fixture; the checksum of five reads ties
    it to the generator it was made with."""
    with open(_GOLDEN) as f:
        fx = json.load(f)
    chk = spec.host_reads(idx=[0, Q - 1, Q, Q + T // 2, Q + T - 1])
    assert fx["reads_crc32"] == "%08x" % (zlib.crc32(chk.bases.tobytes()) & 0xFFFFFFFF), "fixture made with another generator"
    return fx[preset_name]


def sample_indices(Q, n):
    """n query reads spread evenly over [0, Q): every anchor batch and every index part of the device run is touched."""
    n = min(n, Q)
    return (np.arange(n, dtype=np.int64) * Q // n).astype(np.int64)


def forward_sample(spec, Q, T, preset, sample_idx, mid_occ, source="device", device=0, chunk=40000, threads=0, remove_internal=False):
    """-> dict(counts, has_mapping, index, sample (ReadSet), seconds, n_minimizers_seen, n_kept).  counts[i] belongs to query
    sample_idx[i] of the job."""
    t0 = time.perf_counter()
    opt = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
    hq = spec.host_reads(idx=np.asarray(sample_idx, dtype=np.uint64))
    sample = O.ReadSet(hq.seqs(), hq.names)
    b = O.RestrictedIndexBuilder(opt, sample)
    t_gen = t_sk = 0.0
    for a in range(0, T, chunk):
        n = min(chunk, T - a)
        t1 = time.perf_counter()
        if source == "device":
            d = spec.device_reads(Q + a, n, device)
            bases, offsets = d.to_host(), d.offsets
            d.free()
        else:
            h = spec.host_reads(first=Q + a, n=n)
            bases, offsets = h.bases, h.offsets
        names = [b"r%08d" % i for i in range(Q + a, Q + a + n)]
        t2 = time.perf_counter()
        b.add(bases, offsets, names, threads=threads)
        t_gen += t2 - t1
        t_sk += time.perf_counter() - t2
    seen, kept = b.n_minimizers_seen, b.n_kept
    ix = b.finish(mid_occ, threads=threads)
    t3 = time.perf_counter()
    rc, counts, has = ix.twoset_counts(sample, remove_internal=remove_internal, threads=threads)
    assert rc == 0
    return dict(counts=counts, has_mapping=has, index=ix, sample=sample, reads=hq, seconds=time.perf_counter() - t0,
                seconds_reads=t_gen, seconds_sketch=t_sk, seconds_map=time.perf_counter() - t3, n_minimizers_seen=seen, n_kept=kept)
