"""An independent CRAM 3.0 WRITER for the tests of include/lrge_cram.hpp (the reader): written from the encoder's side of the CRAM format
specification v3.0 (hts-specs) -- file definition, containers, slices, blocks with CRC32, ITF8 / LTF8, the compression header's three
maps, EXTERNAL / HUFFMAN / BETA / GAMMA / SUBEXP / BYTE_ARRAY_LEN / BYTE_ARRAY_STOP encodings over a core bit stream and external blocks,
block methods raw / gzip / bzip2 / lzma / rANS 4x8 (orders 0 and 1: the encoder of the published algorithm).  Unaligned records only.
No CRAM file and no CRAM tool exists in this image; the reader and this writer meet only in the specification."""
import bz2
import lzma
import struct
import zlib


def itf8(v):
    v &= 0xFFFFFFFF
    if v < 0x80:
        return bytes([v])
    if v < 0x4000:
        return bytes([0x80 | v >> 8, v & 0xFF])
    if v < 0x200000:
        return bytes([0xC0 | v >> 16, (v >> 8) & 0xFF, v & 0xFF])
    if v < 0x10000000:
        return bytes([0xE0 | v >> 24, (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    return bytes([0xF0 | (v >> 28) & 0x0F, (v >> 20) & 0xFF, (v >> 12) & 0xFF, (v >> 4) & 0xFF, v & 0x0F])


def ltf8(v):
    if v < 0x80:
        return bytes([v])
    if v < 0x4000:
        return bytes([0x80 | v >> 8, v & 0xFF])
    if v < 0x200000:
        return bytes([0xC0 | v >> 16, (v >> 8) & 0xFF, v & 0xFF])
    if v < 0x10000000:
        return bytes([0xE0 | v >> 24, (v >> 16) & 0xFF, (v >> 8) & 0xFF, v & 0xFF])
    if v < 1 << 35:
        return bytes([0xF0 | v >> 32]) + (v & 0xFFFFFFFF).to_bytes(4, "big")
    raise ValueError("ltf8 value too large for this writer")


# ---- rANS 4x8 ----
TOT, LOW = 4096, 1 << 23


def _normalise(counts):
    """256 counts -> frequencies summing to 4096, every present symbol >= 1"""
    n = sum(counts)
    F = [0] * 256
    if n == 0:
        return F
    for s in range(256):
        if counts[s]:
            F[s] = max(1, counts[s] * TOT // n)
    diff = TOT - sum(F)
    m = max(range(256), key=lambda s: F[s])
    F[m] += diff
    if F[m] <= 0:      # (many rare symbols: take the excess from the larger ones)
        F[m] -= diff
        order = sorted(range(256), key=lambda s: -F[s])
        i = 0
        while sum(F) > TOT:
            s = order[i % 256]
            if F[s] > 1:
                F[s] -= 1
            i += 1
        while sum(F) < TOT:
            F[order[0]] += 1
    assert sum(F) == TOT and all(F[s] > 0 for s in range(256) if counts[s])
    return F


def _table_bytes(F):
    out = bytearray()
    rle = 0
    for j in range(256):
        if not F[j]:
            continue
        if rle:
            rle -= 1
        else:
            out.append(j)
            if j and F[j - 1]:
                k = j + 1
                while k < 256 and F[k]:
                    k += 1
                rle = k - (j + 1)
                out.append(rle)
        if F[j] < 128:
            out.append(F[j])
        else:
            out += bytes([128 | F[j] >> 8, F[j] & 0xFF])
    out.append(0)
    return bytes(out)


def rans4x8(data, order):
    data = bytes(data)
    n = len(data)
    if n == 0:
        return bytes([order]) + struct.pack("<II", 0, 0)
    if order == 0:
        seq = [(i & 3, i, 0) for i in range(n)]
    else:
        q = n >> 2
        seq = []
        for i in range(q):
            for k in range(4):
                p = k * q + i
                seq.append((k, p, data[p - 1] if i else 0))
        for p in range(4 * q, n):
            seq.append((3, p, data[p - 1] if p else 0))
        if q == 0:      # fewer than 4 bytes: everything is the remainder, on state 3, first context 0
            seq = [(3, p, data[p - 1] if p else 0) for p in range(n)]
    counts = {}
    for _, p, ctx in seq:
        counts.setdefault(ctx, [0] * 256)[data[p]] += 1
    Fs = {c: _normalise(v) for c, v in counts.items()}
    Cs = {}
    for c, F in Fs.items():
        C, x = [0] * 256, 0
        for s in range(256):
            C[s] = x
            x += F[s]
        Cs[c] = C
    if order == 0:
        table = _table_bytes(Fs[0])
    else:
        present = [1 if c in Fs else 0 for c in range(256)]
        table = bytearray()
        rle = 0
        for i in range(256):
            if not present[i]:
                continue
            if rle:
                rle -= 1
            else:
                table.append(i)
                if i and present[i - 1]:
                    k = i + 1
                    while k < 256 and present[k]:
                        k += 1
                    rle = k - (i + 1)
                    table.append(rle)
            table += _table_bytes(Fs[i])
        table.append(0)
        table = bytes(table)
    R = [LOW] * 4
    rev = bytearray()
    for k, p, ctx in reversed(seq):
        f, c = Fs[ctx][data[p]], Cs[ctx][data[p]]
        x = R[k]
        x_max = ((LOW >> 12) << 8) * f
        while x >= x_max:
            rev.append(x & 0xFF)
            x >>= 8
        R[k] = (x // f << 12) + (x % f) + c
    payload = table + b"".join(struct.pack("<I", R[k]) for k in range(4)) + bytes(reversed(rev))
    return bytes([order]) + struct.pack("<II", len(payload), n) + payload


METHODS = {"raw": 0, "gzip": 1, "bzip2": 2, "lzma": 3, "rans0": 4, "rans1": 4, "nx16": 5}


def compress(data, method):
    if method == "raw":
        return data
    if method == "gzip":
        co = zlib.compressobj(6, zlib.DEFLATED, 31)
        return co.compress(data) + co.flush()
    if method == "bzip2":
        return bz2.compress(data)
    if method == "lzma":
        return lzma.compress(data, format=lzma.FORMAT_XZ)
    if method == "rans0":
        return rans4x8(data, 0)
    if method == "rans1":
        return rans4x8(data, 1)
    if method == "nx16":          # (a CRAM 3.1 codec this writer cannot produce: only the method byte, for the refusal test)
        return data
    raise ValueError(method)


def block(method, content_type, content_id, data):
    body = compress(data, method)
    b = bytes([METHODS[method], content_type]) + itf8(content_id) + itf8(len(body)) + itf8(len(data)) + body
    return b + struct.pack("<I", zlib.crc32(b))


def container(ref_id, start, span, n_records, counter, bases, blocks, landmarks):
    body = b"".join(blocks)
    h = struct.pack("<i", len(body)) + itf8(ref_id) + itf8(start) + itf8(span) + itf8(n_records) + ltf8(counter) + ltf8(bases) + itf8(len(blocks)) + \
        itf8(len(landmarks)) + b"".join(itf8(x) for x in landmarks)
    return h + struct.pack("<I", zlib.crc32(h)) + body


class Bits:
    def __init__(self):
        self.b, self.n = bytearray(), 0

    def put(self, v, nbits):
        for i in range(nbits - 1, -1, -1):
            if self.n % 8 == 0:
                self.b.append(0)
            if (v >> i) & 1:
                self.b[-1] |= 0x80 >> (self.n % 8)
            self.n += 1

    def bytes(self):
        return bytes(self.b)


def enc_external(cid):
    p = itf8(cid)
    return itf8(1) + itf8(len(p)) + p


def enc_huffman(alphabet, lens):
    p = itf8(len(alphabet)) + b"".join(itf8(a) for a in alphabet) + itf8(len(lens)) + b"".join(itf8(x) for x in lens)
    return itf8(3) + itf8(len(p)) + p


def enc_beta(offset, nbits):
    p = itf8(offset) + itf8(nbits)
    return itf8(6) + itf8(len(p)) + p


def enc_gamma(offset):
    p = itf8(offset)
    return itf8(9) + itf8(len(p)) + p


def enc_subexp(offset, k):
    p = itf8(offset) + itf8(k)
    return itf8(7) + itf8(len(p)) + p


def enc_bytes_stop(stop, cid):
    p = bytes([stop]) + itf8(cid)
    return itf8(5) + itf8(len(p)) + p


def enc_bytes_len(len_enc, val_enc):
    p = len_enc + val_enc
    return itf8(4) + itf8(len(p)) + p


def huffman_codes(alphabet, lens):
    order = sorted(range(len(alphabet)), key=lambda i: (lens[i], alphabet[i]))
    codes, code, cur = {}, 0, lens[order[0]]
    for i in order:
        while cur < lens[i]:
            code <<= 1
            cur += 1
        codes[alphabet[i]] = (code, lens[i])
        code += 1
    return codes


def write_cram(reads, variant="external", method="gzip", slices_per_container=1, records_per_slice=None, with_quality=True, with_tags=True,
               name_form="stop", mapped_at=None, major=3, minor=0, method_for=None):
    """reads: list of (name bytes, sequence bytes).  variant: "external" (every series in an external block of its own) or "core" (flags and
    lengths in the core bit stream: BF / CF HUFFMAN, RL BETA, AP GAMMA, RG zero-bit HUFFMAN, TL SUBEXP).  mapped_at: index of a record written
    with BAM flag 0 (mapped) -- the reader must refuse the file.  method_for: {series: method} overriding `method` for single external blocks."""
    method_for = method_for or {}
    out = bytearray(b"CRAM" + bytes([major, minor]) + b"lrge-test-cram\0\0\0\0\0\0")
    assert len(out) == 26
    hdr = b"@HD\tVN:1.6\tSO:unsorted\n"
    out += container(0, 0, 0, 0, 0, 0, [block("raw", 0, 0, struct.pack("<i", len(hdr)) + hdr)], [0])
    # content ids of the external blocks
    ID = {"BF": 1, "CF": 2, "RL": 3, "AP": 4, "RG": 5, "RN": 6, "MF": 7, "NS": 8, "NP": 9, "TS": 10, "NF": 11, "TL": 12, "BA": 13, "QS": 14, "TAG": 15, "TAGLEN": 16}
    core = variant == "core"
    tag_lines = [[b"XAZ"], []] if with_tags else [[]]
    td = b"".join(b"".join(line) + b"\0" for line in tag_lines)
    pres = [b"RN" + b"\x01", b"AP" + b"\x00", b"RR" + b"\x00", b"SM" + bytes(5), b"TD" + itf8(len(td)) + td]
    pres_b = itf8(len(pres)) + b"".join(pres)
    bf_alpha, bf_lens = [4, 0, 77], [1, 2, 2]
    cf_alpha, cf_lens = [0, 1, 2, 3], [2, 2, 2, 2]
    ds = {
        "BF": enc_huffman(bf_alpha, bf_lens) if core else enc_external(ID["BF"]),
        "CF": enc_huffman(cf_alpha, cf_lens) if core else enc_external(ID["CF"]),
        "RL": enc_beta(0, 20) if core else enc_external(ID["RL"]),
        "AP": enc_gamma(1) if core else enc_external(ID["AP"]),
        "RG": enc_huffman([-1], [0]) if core else enc_external(ID["RG"]),
        "RN": enc_bytes_stop(0, ID["RN"]) if name_form == "stop" else enc_bytes_len(enc_external(ID["TAGLEN"] + 1), enc_external(ID["RN"])),
        "MF": enc_external(ID["MF"]), "NS": enc_external(ID["NS"]), "NP": enc_external(ID["NP"]), "TS": enc_external(ID["TS"]), "NF": enc_external(ID["NF"]),
        "TL": enc_subexp(0, 0) if core else enc_external(ID["TL"]),
        "BA": enc_external(ID["BA"]), "QS": enc_external(ID["QS"]),
    }
    ds_b = itf8(len(ds)) + b"".join(k.encode() + v for k, v in ds.items())
    tag_id = ord("X") << 16 | ord("A") << 8 | ord("Z")
    tags = {tag_id: enc_bytes_len(enc_external(ID["TAGLEN"]), enc_external(ID["TAG"]))} if with_tags else {}
    tags_b = itf8(len(tags)) + b"".join(itf8(k) + v for k, v in tags.items())
    comp_header = itf8(len(pres_b)) + pres_b + itf8(len(ds_b)) + ds_b + itf8(len(tags_b)) + tags_b
    bfc, cfc = huffman_codes(bf_alpha, bf_lens), huffman_codes(cf_alpha, cf_lens)

    rps = records_per_slice or max(1, len(reads))
    slices = [reads[i:i + rps] for i in range(0, len(reads), rps)] or [[]]
    counter = 0
    for c0 in range(0, len(slices), slices_per_container):
        group = slices[c0:c0 + slices_per_container]
        blocks = [block("raw" if method == "raw" else "gzip", 1, 0, comp_header)]
        landmarks, n_rec, n_bases = [], 0, 0
        for sl in group:
            ext = {k: bytearray() for k in ID}
            ext["NAMELEN"] = bytearray()
            bits = Bits()
            for idx, (name, seq) in enumerate(sl):
                gi = counter + idx
                bf = 0 if mapped_at is not None and gi == mapped_at else (77 if gi % 5 == 4 else 4)      # 77 = paired | unmapped | mate unmapped | first
                cf = (1 if with_quality else 0) | (2 if bf == 77 else 0)
                tl = (gi % 2) if with_tags else 0
                if core:
                    bits.put(*bfc[bf]); bits.put(*cfc[cf])
                    bits.put(len(seq), 20)
                    bits.put(1, 1)                      # AP = 0: gamma of (0 + offset 1) = 1 -> "1"
                    # RG: zero bits
                else:
                    ext["BF"] += itf8(bf); ext["CF"] += itf8(cf); ext["RL"] += itf8(len(seq)); ext["AP"] += itf8(0); ext["RG"] += itf8(-1)
                if name_form == "stop":
                    ext["RN"] += name + b"\0"
                else:
                    ext["NAMELEN"] += itf8(len(name)); ext["RN"] += name
                if cf & 2:
                    ext["MF"] += itf8(2); ext["NS"] += itf8(-1); ext["NP"] += itf8(0); ext["TS"] += itf8(0)
                if core:
                    if tl == 0:
                        bits.put(0, 1)                  # subexp k = 0: u = 0, then 0 bits -> value 0
                    else:
                        bits.put(0b10, 2)               # u = 1: b = 0 bits -> value 1
                else:
                    ext["TL"] += itf8(tl)
                if with_tags and tl == 0:
                    val = b"tag-of-" + name + b"\0"
                    ext["TAGLEN"] += itf8(len(val)); ext["TAG"] += val
                if bf & 4:
                    ext["BA"] += seq
                else:                                   # a mapped record: FN = 0 features, MQ (the reader refuses before it gets here)
                    pass
                if with_quality:
                    ext["QS"] += bytes([30 + (i % 10) for i in range(len(seq))])
                n_bases += len(seq)
            n_rec += len(sl)
            use = [k for k in ID if ext[k] or k in ("BA", "RN")]
            ids = [ID[k] for k in use]
            if name_form != "stop":
                ids.append(ID["TAGLEN"] + 1)
            sh = itf8(-1) + itf8(0) + itf8(0) + itf8(len(sl)) + ltf8(counter) + itf8(1 + len(ids)) + itf8(len(ids)) + b"".join(itf8(i) for i in ids) + itf8(-1) + bytes(16)
            landmarks.append(sum(len(b) for b in blocks))
            blocks.append(block("raw", 2, 0, sh))
            blocks.append(block("raw", 5, 0, bits.bytes()))
            for k in use:
                blocks.append(block(method_for.get(k, method), 4, ID[k], bytes(ext[k])))
            if name_form != "stop":
                blocks.append(block(method, 4, ID["TAGLEN"] + 1, bytes(ext["NAMELEN"])))
            counter += len(sl)
        out += container(-1, 0, 0, n_rec, counter - n_rec, n_bases, blocks, landmarks)
    # EOF container (the specification's fixed marker: an empty compression header block, start 4542278)
    out += container(-1, 4542278, 0, 0, 0, 0, [block("raw", 1, 0, b"\x01\x00\x01\x00\x01\x00")], [])
    return bytes(out)
