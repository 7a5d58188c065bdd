#!/usr/bin/env python
"""Regenerates the fixtures in this directory.  Run in the build container only (it reads the
reference's test data file); the GPU box never runs this script.

  toy_reads.fa.gz      the 500 unaligned ONT reads of /root/reference/lrge/tests/data/toy.bam (data file
                       of the reference's own integration test, lrge/tests/alignment.rs:52-67; MIT),
                       converted BAM -> FASTA: read name up to the first whitespace, sequence as stored.
  toy_expected.json    outputs of the CPU oracle (oracle/lrge_oracle.c) on those reads: index statistics,
                       all-vs-all counts, a two-set split, with both tie policies.  The reference asserts
                       nothing numeric on this file (only `.success()`), so these are oracle values,
                       not reference values -- "parity unpinned" (DESIGN.md section 5).
"""
import gzip
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

BAM = "/root/reference/lrge/tests/data/toy.bam"
NT16 = "=ACMGRSVTWYHKDBN"


def read_bam(path):
    data = gzip.open(path, "rb").read()          # BGZF = concatenated gzip members
    assert data[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", data, 4)
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", data, off); off += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", data, off); off += 4 + l_name + 4
    out = []
    while off < len(data):
        block_size, = struct.unpack_from("<i", data, off); off += 4
        rec = data[off:off + block_size]; off += block_size
        l_read_name = rec[8]
        n_cigar, flag = struct.unpack_from("<HH", rec, 12)
        l_seq, = struct.unpack_from("<i", rec, 16)
        p = 32
        name = rec[p:p + l_read_name - 1]; p += l_read_name + 4 * n_cigar
        packed = rec[p:p + (l_seq + 1) // 2]
        seq = "".join(NT16[b >> 4] + NT16[b & 15] for b in packed)[:l_seq]
        out.append((name.split()[0] if name.split() else name, seq.encode(), flag))
    return out


def main():
    import numpy as np
    from oracle import oracle as O
    recs = read_bam(BAM)
    assert len(recs) == 500 and all(f & 4 for _, _, f in recs)
    with gzip.GzipFile(os.path.join(HERE, "toy_reads.fa.gz"), "wb", mtime=0) as fh:
        for n, s, _ in recs:
            fh.write(b">" + n + b"\n" + s + b"\n")
    names = [n for n, _, _ in recs]; seqs = [s for _, s, _ in recs]
    exp = {"n_reads": len(recs), "n_bases": sum(map(len, seqs))}
    for preset, pname in ((O.PRESET_AVA_ONT, "ava-ont"), (O.PRESET_AVA_PB, "ava-pb")):
        e = {}
        for mode, mname in ((O.SORT_STABLE, "stable"), (O.SORT_MM2, "mm2")):
            opt = O.make_opt(preset, dual=False, sort_mode=mode)
            ix = O.Index(O.ReadSet(seqs, names), opt)
            rc, counts = ix.ava_counts(threads=8)
            assert rc == 0
            e["ava_counts_" + mname] = counts.tolist()
            e.update(n_minimizers=int(ix.n_minimizers), n_keys=int(ix.n_keys), mid_occ=int(ix.mid_occ))
            # two-set: first 150 reads are queries, the other 350 are targets
            opt2 = O.make_opt(preset, dual=True, sort_mode=mode)
            ix2 = O.Index(O.ReadSet(seqs[150:], names[150:]), opt2)
            Q = O.ReadSet(seqs[:150], names[:150])
            rc, c2, h2 = ix2.twoset_counts(Q, threads=8)
            e["twoset_counts_" + mname] = c2.tolist(); e["twoset_has_" + mname] = h2.tolist()
            rc, c3, h3 = ix2.twoset_counts(Q, remove_internal=True, ratio=0.2, threads=8)
            e["twoset_counts_F_" + mname] = c3.tolist()
            e["twoset_mid_occ"] = int(ix2.mid_occ)
            n_chain = sum(len(ix2.map(s, n)) for s, n in zip(seqs[:150], names[:150]))
            e["twoset_n_chains_" + mname] = int(n_chain)
        e["tie_policies_agree"] = bool(e["ava_counts_stable"] == e["ava_counts_mm2"] and
                                       e["twoset_counts_stable"] == e["twoset_counts_mm2"])
        exp[pname] = e
    json.dump(exp, open(os.path.join(HERE, "toy_expected.json"), "w"), separators=(",", ":"))
    print({k: (v if not isinstance(v, dict) else {kk: (vv if not isinstance(vv, list) else "list[%d] sum %d" % (len(vv), sum(vv)))
                                                  for kk, vv in v.items()}) for k, v in exp.items()})


if __name__ == "__main__":
    main()
