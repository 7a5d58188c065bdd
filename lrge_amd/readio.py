"""Minimal read input for the host mirror: in-memory (names, seqs) or plain/gzip FASTA/FASTQ.
The reference's full input layer (liblrge/src/io.rs: zstd/bz2/xz sniffing, BAM/CRAM/SAM) is host-side
I/O outside the hot path."""
import gzip

import numpy as np


def read_id(header: bytes) -> bytes:
    """Header up to the first ASCII whitespace (io.rs:199-204)."""
    for i, c in enumerate(header):
        if c in b" \t\n\r\x0b\x0c":
            return header[:i]
    return header


def iter_records(path):
    op = gzip.open if open(path, "rb").read(2) == b"\x1f\x8b" else open
    with op(path, "rb") as fh:
        first = fh.read(1)
        if not first:
            return
        if first == b">":
            name, chunks = fh.readline().rstrip(b"\r\n"), []
            for line in fh:
                if line.startswith(b">"):
                    yield read_id(name), b"".join(chunks)
                    name, chunks = line[1:].rstrip(b"\r\n"), []
                else:
                    chunks.append(line.rstrip(b"\r\n"))
            yield read_id(name), b"".join(chunks)
        elif first == b"@":
            name = fh.readline().rstrip(b"\r\n")
            while True:
                seq = fh.readline().rstrip(b"\r\n")
                fh.readline(); fh.readline()
                yield read_id(name), seq
                hdr = fh.readline()
                if not hdr:
                    break
                name = hdr[1:].rstrip(b"\r\n")
        else:
            raise ValueError("unrecognised sequence file: %r" % path)


def load(source):
    """source: path, (names, seqs) or an object with .names / .seqs() -> (names list[bytes], seqs list[bytes])."""
    if isinstance(source, (str, bytes)) and not isinstance(source, tuple):
        names, seqs = [], []
        for n, s in iter_records(source):
            names.append(n); seqs.append(s)
        return names, seqs
    if hasattr(source, "seqs"):
        return list(source.names), source.seqs()
    names, seqs = source
    return [n if isinstance(n, bytes) else n.encode() for n in names], [bytes(s) for s in seqs]


def pack(seqs):
    lens = np.array([len(s) for s in seqs], dtype=np.uint64)
    offs = np.zeros(len(seqs) + 1, dtype=np.uint64)
    np.cumsum(lens, out=offs[1:])
    bases = np.frombuffer(b"".join(seqs), dtype=np.uint8).copy() if int(offs[-1]) else np.zeros(0, dtype=np.uint8)
    return bases, offs
