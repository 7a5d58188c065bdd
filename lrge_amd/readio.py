"""Read input for the host mirror: in-memory (names, seqs) or FASTA/FASTQ files, plain or compressed.
Compression is sniffed from the magic bytes like the reference does (liblrge/src/io.rs): gzip, bzip2 and xz
through the Python standard library; zstd needs a module this image lacks and is reported as such.  BAM/CRAM/SAM
(io.rs `alignment` feature) stay out: host-side I/O outside the hot path."""
import bz2
import gzip
import lzma

import numpy as np


def read_id(header: bytes) -> bytes:
    """Header up to the first ASCII whitespace (io.rs:199-204)."""
    for i, c in enumerate(header):
        if c in b" \t\n\r\x0b\x0c":
            return header[:i]
    return header


def _opener(path):
    magic = open(path, "rb").read(6)
    if magic[:2] == b"\x1f\x8b":
        return gzip.open
    if magic[:3] == b"BZh":
        return bz2.open
    if magic[:6] == b"\xfd7zXZ\x00":
        return lzma.open
    if magic[:4] == b"\x28\xb5\x2f\xfd":
        raise ValueError("zstd-compressed input is not supported in this build (no zstd module): %r" % path)
    return open


def iter_records(path):
    with _opener(path)(path, "rb") as fh:
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
