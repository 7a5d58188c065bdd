"""Input formats (liblrge/src/io.rs:35-184, lrge/tests/alignment.rs:5-67): FASTA / FASTQ / unaligned
SAM / unaligned BAM, plain or gzip / bzip2 / xz / zstd, sniffed by magic bytes.  The Python mirror
(lrge_amd/readio.py) and the C++ host side (include/lrge_io.hpp, through `lrge-hip --dump-records`, a host-only mode)
must hand out the same (read id, sequence) records; mapped records are refused with the reference's message."""
import bz2
import gzip
import lzma
import os
import subprocess

import pytest

from lrge_amd import build, readio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RECORDS = [(b"read1", b"GATTACAGATTACA"), (b"r2", b"ACGTNNNNACGTacgt"), (b"third/1", b"T" * 300), (b"r4", b"C")]
MAPPED = "Mapped records are not supported. Only unaligned BAM/CRAM/SAM is allowed."


def _fasta():
    return b"".join(b">%s some comment\tx\n%s\n%s\n" % (n, s[:7], s[7:]) if len(s) > 7 else b">%s\n%s\n" % (n, s) for n, s in RECORDS)


def _fastq():
    return b"".join(b"@%s desc\n%s\n+\n%s\n" % (n, s, b"!" * len(s)) for n, s in RECORDS)


def _sam(mapped=False):
    out = b"@HD\tVN:1.6\tSO:unsorted\n" + (b"@SQ\tSN:chr1\tLN:1000\n" if mapped else b"")
    for i, (n, s) in enumerate(RECORDS):
        if mapped and i == 0:
            out += b"%s\t0\tchr1\t1\t0\t%dM\t*\t0\t0\t%s\t%s\n" % (n, len(s), s, b"!" * len(s))
        else:
            out += b"%s\t4\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\tRG:Z:x\n" % (n, s, b"!" * len(s))
    return out


PLAIN = {"fa": _fasta, "fq": _fastq, "sam": _sam}
COMPRESS = {"": lambda b: b, ".gz": gzip.compress, ".bz2": bz2.compress, ".xz": lzma.compress, ".zst": readio.zstd_compress}


@pytest.fixture(scope="module")
def cli():
    build.build_lib()
    return build.build_cli()


def _dump(cli, path):
    out = subprocess.run([cli, "--dump-records", str(path)], capture_output=True, timeout=60)
    recs = [tuple(l.split(b"\t")) for l in out.stdout.split(b"\n") if l]
    return out.returncode, recs, out.stderr.decode()


@pytest.mark.parametrize("ext", sorted(COMPRESS))
@pytest.mark.parametrize("kind", sorted(PLAIN))
def test_every_format_yields_the_same_records(tmp_path, cli, kind, ext):
    p = tmp_path / ("reads." + kind + ext + ".input")          # the extension plays no role: content is sniffed
    p.write_bytes(COMPRESS[ext](PLAIN[kind]()))
    expect = [(n, s) for n, s in RECORDS]
    assert list(readio.iter_records(str(p))) == expect
    assert readio.count_records(str(p)) == len(RECORDS)
    rc, recs, err = _dump(cli, p)
    assert rc == 0 and recs == expect, err


def test_multi_member_gzip_and_crlf(tmp_path, cli):
    fa = _fasta().replace(b"\n", b"\r\n")
    half = len(fa) // 2
    p = tmp_path / "multi.gz"
    p.write_bytes(gzip.compress(fa[:half]) + gzip.compress(fa[half:]))
    assert list(readio.iter_records(str(p))) == RECORDS
    assert _dump(cli, p)[1] == RECORDS


def test_toy_bam_matches_its_fasta_conversion(cli, tmp_path):
    """The reads of lrge/tests/data/toy.bam (the reference's own test input, alignment.rs:52-67: 500 unaligned ONT reads),
    shipped as their FASTA conversion and written back to uBAM here: both readers must return the same records."""
    from conftest import write_unaligned_bam
    fa = list(readio.iter_records(os.path.join(GOLDEN, "toy_reads.fa.gz")))
    bam_path = tmp_path / "toy.bam"
    write_unaligned_bam(bam_path, [n for n, _ in fa], [s for _, s in fa])
    bam = list(readio.iter_records(str(bam_path)))
    assert len(bam) == 500 and bam == fa
    rc, recs, err = _dump(cli, bam_path)
    assert rc == 0 and recs == fa and "500 records" in err


@pytest.mark.parametrize("ext", ["", ".gz"])
def test_mapped_records_are_refused(tmp_path, cli, ext):
    """lrge/tests/alignment.rs:29-50."""
    p = tmp_path / ("mapped.sam" + ext)
    p.write_bytes(COMPRESS[ext](_sam(mapped=True)))
    with pytest.raises(ValueError, match="Mapped records are not supported"):
        list(readio.iter_records(str(p)))
    rc, _, err = _dump(cli, p)
    assert rc != 0 and MAPPED in err


def test_mapped_bam_is_refused(tmp_path, cli):
    import struct
    name, seq = b"m1\0", b"\x12\x48"                                   # ACGT packed
    rec = struct.pack("<iiBBHHHiiii", 0, 0, len(name), 0, 0, 0, 0, 4, -1, -1, 0) + name + seq + b"\xff" * 4
    body = b"BAM\x01" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 5) + b"chr1\0" + struct.pack("<i", 1000) + \
        struct.pack("<i", len(rec)) + rec
    p = tmp_path / "mapped.bam"
    p.write_bytes(gzip.compress(body))
    with pytest.raises(ValueError, match="Mapped records are not supported"):
        list(readio.iter_records(str(p)))
    rc, _, err = _dump(cli, p)
    assert rc != 0 and MAPPED in err
    # the same record flagged unmapped is read: name without the NUL, 4-bit bases high nibble first
    rec2 = rec[:14] + struct.pack("<H", 4) + rec[16:]
    p.write_bytes(gzip.compress(body[:-len(rec)] + rec2))
    assert list(readio.iter_records(str(p))) == [(b"m1", b"ACGT")]
    assert _dump(cli, p)[1] == [(b"m1", b"ACGT")]


def test_empty_cram_and_garbage(tmp_path, cli):
    e = tmp_path / "empty.fa"
    e.write_bytes(b"")
    with pytest.raises(ValueError, match="Is the file empty"):
        readio.count_records(str(e))
    rc, _, err = _dump(cli, e)
    assert rc != 0 and "Is the file empty?" in err
    c = tmp_path / "x.cram"
    c.write_bytes(b"CRAM\x03\x00" + b"\0" * 40)          # a CRAM magic over garbage: an error, not a crash
    with pytest.raises(ValueError):
        list(readio.iter_records(str(c)))
    assert _dump(cli, c)[0] != 0
    g = tmp_path / "garbage"
    g.write_bytes(b"hello world\n")
    with pytest.raises(ValueError):
        list(readio.iter_records(str(g)))
    assert _dump(cli, g)[0] != 0
    t = tmp_path / "trunc.gz"
    t.write_bytes(gzip.compress(_fasta() * 50)[:-20])
    with pytest.raises(Exception):
        list(readio.iter_records(str(t)))
    assert _dump(cli, t)[0] != 0


# ---- unaligned CRAM 3.0 (round 6; io.rs:93,154-184 reads it through noodles): include/lrge_cram.hpp against tests/cram_writer.py, an
# independent writer of the same specification (no CRAM file and no CRAM tool exists in this image) ----
def _cram_reads(n=57):
    import numpy as np
    rng = np.random.default_rng(7)
    reads = []
    for i in range(n):
        ln = int(rng.integers(1, 4000)) if i % 11 else 1
        reads.append((b"read-%04d/%d" % (i, i % 3), bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), size=ln, p=[.24, .24, .24, .24, .01, .0075, .0075, .0075, .0075]))))
    return reads


@pytest.mark.parametrize("method", ["raw", "gzip", "bzip2", "lzma", "rans0", "rans1"])
@pytest.mark.parametrize("variant", ["external", "core"])
def test_cram_every_block_method_and_both_layouts(tmp_path, cli, method, variant):
    """Every block compression method of CRAM 3.0 x the two layouts of the per-record integers: every series in an external block of its
    own, or flags / lengths / positions / tag lines bit-packed into the core block (HUFFMAN, BETA, GAMMA, zero-bit HUFFMAN, SUBEXP).
    Names with a stop byte, one tag on every other record, qualities present, detached records with mate fields among them."""
    import cram_writer
    reads = _cram_reads()
    p = tmp_path / "u.cram"
    p.write_bytes(cram_writer.write_cram(reads, variant=variant, method=method, records_per_slice=20, slices_per_container=2))
    assert list(readio.iter_records(str(p))) == reads
    rc, recs, err = _dump(cli, p)
    assert rc == 0 and recs == reads, err
    assert readio.count_records(str(p)) == len(reads)


@pytest.mark.parametrize("opts", [dict(with_quality=False), dict(with_tags=False), dict(name_form="len"), dict(records_per_slice=1, slices_per_container=3),
                                  dict(records_per_slice=1000)])
def test_cram_shapes(tmp_path, cli, opts):
    """No qualities, no tags, names as BYTE_ARRAY_LEN, one record per slice and three slices per container, one slice for everything."""
    import cram_writer
    reads = _cram_reads(23)
    p = tmp_path / "s.cram"
    p.write_bytes(cram_writer.write_cram(reads, **opts))
    assert list(readio.iter_records(str(p))) == reads
    assert _dump(cli, p)[1] == reads
    # gzip around the whole file: io.rs:71-90 decompresses before it sniffs
    z = tmp_path / "s.cram.gz"
    z.write_bytes(gzip.compress(p.read_bytes()))
    assert list(readio.iter_records(str(z))) == reads


def test_cram_refusals(tmp_path, cli):
    """A mapped record is refused with the reference's message (io.rs:162-167); a block in one of the two CRAM 3.1 codecs this reader does not decode (the adaptive arithmetic coder, fqzcomp) is an error that names the codec
    when the reader needs the block (bases), and no obstacle when it does not (qualities are never read); truncation is an error; an
    empty file (header and EOF containers only) has no records."""
    import cram_writer
    reads = _cram_reads(9)
    m = tmp_path / "mapped.cram"
    m.write_bytes(cram_writer.write_cram(reads, mapped_at=4))
    with pytest.raises(ValueError, match="Mapped records are not supported"):
        list(readio.iter_records(str(m)))
    rc, _, err = _dump(cli, m)
    assert rc != 0 and MAPPED in err
    q = tmp_path / "qs31.cram"
    q.write_bytes(cram_writer.write_cram(reads, minor=1, method_for={"QS": "fqz"}))
    assert list(readio.iter_records(str(q))) == reads
    b = tmp_path / "ba31.cram"
    b.write_bytes(cram_writer.write_cram(reads, minor=1, method_for={"BA": "arith"}))
    with pytest.raises(ValueError, match="adaptive arithmetic coder"):
        list(readio.iter_records(str(b)))
    assert _dump(cli, b)[0] != 0
    t = tmp_path / "trunc.cram"
    t.write_bytes(cram_writer.write_cram(reads)[:-60])
    with pytest.raises(ValueError):
        list(readio.iter_records(str(t)))
    e = tmp_path / "empty.cram"
    e.write_bytes(cram_writer.write_cram([]))
    assert list(readio.iter_records(str(e))) == []
    with pytest.raises(ValueError, match="Is the file empty"):
        readio.count_records(str(e))


def test_rans4x8_round_trips():
    """The rANS 4x8 decoder against the test writer's encoder, both orders: skewed and flat distributions, every byte value, runs of
    consecutive symbols (the run-length coded frequency tables), lengths 0 .. 9 (the four interleaved states and the order-1 quarters)."""
    import ctypes as C
    import numpy as np
    import cram_writer
    from lrge_amd import _ffi
    rng = np.random.default_rng(3)
    cases = [bytes(range(256)) * 3, b"", b"A", b"AC", b"ACG", b"ACGT", b"ACGTA", b"ACGTACGTA", bytes(rng.integers(0, 256, 5000, dtype=np.uint8)),
             bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 20000, p=[.7, .1, .1, .1])), b"\x00" * 1000, bytes([5, 6, 7, 8, 9] * 400)]
    for data in cases:
        for order in (0, 1):
            # through a one-record CRAM whose base block is the data: the decoder has no other entry point
            import tempfile
            with tempfile.TemporaryDirectory() as d:
                p = os.path.join(d, "r.cram")
                seq = data.replace(b"\n", b"N") or b"A"
                open(p, "wb").write(cram_writer.write_cram([(b"x", seq)], method="rans%d" % order, with_tags=False, with_quality=False))
                assert list(readio.iter_records(p)) == [(b"x", seq)], (len(data), order)


def test_damaged_files_end_in_an_error_never_in_a_crash(tmp_path, cli):
    """Seeded damage (overwritten bytes, single bit flips, a 0x7fffffff where a length may stand, truncation) to files of every container
    the reader decodes itself: the process ends with records or with an error message -- no signal, no hang, no allocation by a length field
    the file cannot back.  (A bounded sample for the suite; the longer campaigns of the same generator are in DESIGN.md 8.)"""
    import random

    import cram_writer
    reads = _cram_reads(23)
    seeds = {"cram_" + m + "_" + v: cram_writer.write_cram(reads, variant=v, method=m, records_per_slice=10, slices_per_container=2)
             for m, v in (("raw", "core"), ("gzip", "external"), ("rans0", "core"), ("rans1", "external"))}
    seeds["fq.gz"] = gzip.compress(_fastq())
    seeds["sam"] = _sam()
    rng = random.Random(20260930)
    for name, data in sorted(seeds.items()):
        for it in range(40):
            b = bytearray(data)
            kind = it % 4
            if kind == 0:
                for _ in range(rng.randrange(1, 4)):
                    b[rng.randrange(len(b))] = rng.randrange(256)
            elif kind == 1:
                b = b[:rng.randrange(1, len(b))]
            elif kind == 2:
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            else:
                i = rng.randrange(len(b))
                b[i:i + 4] = b"\xff\xff\xff\x7f"
            p = tmp_path / ("%s.%d" % (name, it))
            p.write_bytes(bytes(b))
            out = subprocess.run([cli, "--dump-records", str(p)], capture_output=True, timeout=15)
            assert out.returncode in (0, 1), (name, it, kind, out.returncode, out.stderr[-300:])
            if out.returncode:
                assert out.stderr.strip(), (name, it)


@pytest.mark.parametrize("order", [0, 1])
def test_rans_size_fields_that_lie(tmp_path, cli, order):
    """The two cases a longer campaign of the generator above found: an uncompressed-size field of a rANS stream far beyond what its bytes
    can encode (refused before anything is allocated for it), and one within that bound but beyond the stream's content (the decoder stops
    where the stream ends instead of inventing the rest)."""
    import struct
    import time

    import cram_writer
    seq = bytes([65, 67, 71, 84] * 2000)
    data = bytearray(cram_writer.write_cram([(b"x", seq)], method="rans%d" % order, with_tags=False, with_quality=False))
    # the base block's rANS header: order byte, compressed size, uncompressed size (u32 LE each)
    hits = [i for i in range(len(data) - 9) if data[i] == order and struct.unpack_from("<I", data, i + 5)[0] == len(seq)]
    assert hits
    for lie, msg in ((0x7fffffff, "exceeds what the stream can hold"), (len(seq) * 50, "")):
        b = bytearray(data)
        for i in hits:
            struct.pack_into("<I", b, i + 5, lie)
        p = tmp_path / ("lie%d.cram" % lie)
        p.write_bytes(bytes(b))
        t0 = time.perf_counter()
        out = subprocess.run([cli, "--dump-records", str(p)], capture_output=True, timeout=15)
        assert out.returncode == 1 and msg in out.stderr.decode() and time.perf_counter() - t0 < 5, out.stderr[-300:]


# ---- CRAM 3.1 (round 6): rANS Nx16 and the name tokeniser, what samtools >= 1.22 writes by default ----
@pytest.mark.parametrize("form", sorted(__import__("cram_writer").NX16_FORMS) if os.path.isdir(os.path.dirname(os.path.abspath(__file__))) else [])
def test_cram31_rans_nx16_every_form(tmp_path, cli, form):
    """Every series of the file in one form of rANS Nx16: orders 0 and 1, 4 and 32 interleaved states, 10- and 12-bit order-1 tables, the
    table itself compressed, bit packing, run lengths (metadata raw and compressed), both, striping (the sub-streams carry no size), and
    data stored as they are; names through the name tokeniser."""
    import cram_writer
    reads = _cram_reads(31)
    p = tmp_path / "v31.cram"
    p.write_bytes(cram_writer.write_cram(reads, minor=1, method=form, method_for={"RN": "tok3"}, records_per_slice=12, slices_per_container=2))
    assert list(readio.iter_records(str(p))) == reads
    rc, recs, err = _dump(cli, p)
    assert rc == 0 and recs == reads, err


def _roundtrip_block(tmp_path, data, method, tag):
    """through a one-record CRAM whose tag value block is `data` -- no: whose BASES are the data (the decoders have no other entry point)"""
    import cram_writer
    seq = bytes(data) or b"A"
    p = tmp_path / ("rt_%s.cram" % tag)
    p.write_bytes(cram_writer.write_cram([(b"x", seq)], minor=1, method="raw", method_for={"BA": method}, with_tags=False, with_quality=False))
    assert list(readio.iter_records(str(p))) == [(b"x", seq)], (method, len(seq))


def test_rans_nx16_round_trips(tmp_path):
    """The decoder against the test writer's encoder over the shapes that matter: lengths around the interleave (0 .. 2 N + 1), one symbol,
    two / four / sixteen / seventeen symbols (the packing classes), long runs, every byte value, skewed and flat distributions."""
    import numpy as np
    import cram_writer
    rng = np.random.default_rng(5)
    nl = bytes([10])
    cases = [b"A" * n for n in (1, 2, 3, 4, 5, 9, 31, 32, 33, 65, 67)] + [b"ACGT"[:k] * 37 for k in (1, 2, 3, 4)] + \
            [bytes(range(65, 65 + 16)) * 9, bytes(range(65, 65 + 17)) * 9, bytes(b for b in range(256) if b != 10) * 2,
             bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 30000, p=[.7, .1, .1, .1])), b"A" * 3000 + b"C" * 5 + b"G" * 700 + b"ACGT" * 50,
             bytes(rng.integers(65, 91, 7000, dtype=np.uint8))]
    for ci, data in enumerate(cases):
        assert nl not in data
        for form in sorted(cram_writer.NX16_FORMS):
            if "pack" in form and len(set(data)) > 16 and False:
                continue
            _roundtrip_block(tmp_path, data, form, "%d_%s" % (ci, form))


def test_name_tokeniser_shapes(tmp_path, cli):
    """Names that exercise every token type: matches, deltas with and without leading zeros, numbers that grow a digit, duplicates,
    names of different token counts, single characters; with and without the implied TYPE streams and the duplicate-stream references;
    the token streams in several rANS Nx16 forms."""
    import cram_writer
    base = [b"m64011_190830_220126/%d/ccs" % (i * 7) for i in range(40)] + [b"m64011_190830_220126/%d/ccs" % 273] * 3 + \
           [b"read-%04d/%d" % (i, i % 3) for i in range(95, 130)] + [b"x", b"x", b"a.b.c.d.e", b"a.b.c", b"00012", b"00013", b"0009", b"0010", b"9", b"10", b"255", b"511",
            b"SRR28370649.%d" % 4294967295, b"SRR28370649.%d" % 1, b"@weird name:with spaces;and;more", b"1234567890123456789"]
    for opts in (dict(), dict(implied_type=False, dup_streams=False), dict(stream_opts=dict(order=1)), dict(stream_opts=dict(rle=True, pack=True)), dict(stream_opts=dict(cat=True))):
        reads = [(n, b"ACGT" * (1 + i % 5)) for i, n in enumerate(base)]
        body = cram_writer.tok3([n for n, _ in reads], **opts)
        p = tmp_path / "tok.cram"
        orig = cram_writer.compress
        try:
            cram_writer.compress = lambda data, method, _o=orig, _b=body: _b if method == "tok3" else _o(data, method)
            p.write_bytes(cram_writer.write_cram(reads, minor=1, method="nx16_o1", method_for={"RN": "tok3"}))
        finally:
            cram_writer.compress = orig
        assert list(readio.iter_records(str(p))) == reads, opts
        assert _dump(cli, p)[1] == reads


def test_cram31_damage_and_integrity(tmp_path, cli):
    """A flipped bit inside a rANS Nx16 body is an error (the states do not return to their start), never different bases; seeded damage
    to CRAM 3.1 files ends in records or an error message."""
    import random

    import cram_writer
    reads = _cram_reads(17)
    rng = random.Random(31)
    for form in ("nx16_o0", "nx16_o1", "nx16_packrle", "nx16_stripe", "nx16_o1ct"):
        data = cram_writer.write_cram(reads, minor=1, method=form, method_for={"RN": "tok3"})
        wrong = 0
        for it in range(40):
            b = bytearray(data)
            kind = it % 4
            if kind == 0:
                for _ in range(rng.randrange(1, 4)):
                    b[rng.randrange(len(b))] = rng.randrange(256)
            elif kind == 1:
                b = b[:rng.randrange(1, len(b))]
            elif kind == 2:
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
            else:
                i = rng.randrange(len(b))
                b[i:i + 4] = b"\xff\xff\xff\x7f"
            p = tmp_path / ("%s.%d" % (form, it))
            p.write_bytes(bytes(b))
            out = subprocess.run([cli, "--dump-records", str(p)], capture_output=True, timeout=15)
            assert out.returncode in (0, 1), (form, it, kind, out.returncode, out.stderr[-300:])
            if out.returncode == 0:
                recs = [tuple(l.split(b"\t")) for l in out.stdout.split(b"\n") if l]
                wrong += [s for _, s in recs] != [s for _, s in reads]
        # damage that leaves the file readable may change a name or a flag byte; bases that differ without an error must stay the exception
        assert wrong <= 2, (form, wrong)
