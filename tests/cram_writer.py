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


# ---- rANS Nx16 (CRAM 3.1): the encoder's side of the codecs specification ----
NX_LOW = 1 << 15


def u7(v):
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append(0x80 | (v & 0x7F))
        v >>= 7
    return bytes(reversed(out))


def _pow2_at_least(n):
    p = 1
    while p < n:
        p <<= 1
    return p


def _normalise_to(counts, tot):
    """256 counts -> frequencies summing to tot (a power of two), every present symbol >= 1"""
    n = sum(counts)
    F = [0] * 256
    if n == 0:
        return F
    for s in range(256):
        if counts[s]:
            F[s] = max(1, counts[s] * tot // n)
    order = sorted(range(256), key=lambda s: -F[s])
    i = 0
    while sum(F) > tot:
        s = order[i % 256]
        if F[s] > 1:
            F[s] -= 1
        i += 1
    F[order[0]] += tot - sum(F)
    assert sum(F) == tot and all(F[s] > 0 for s in range(256) if counts[s])
    return F


def _alphabet_bytes(present):
    out = bytearray()
    rle = 0
    for j in range(256):
        if not present[j]:
            continue
        if rle:
            rle -= 1
        else:
            out.append(j)
            if j and present[j - 1]:
                k = j + 1
                while k < 256 and present[k]:
                    k += 1
                rle = k - (j + 1)
                out.append(rle)
    out.append(0)
    return bytes(out)


def _nx_encode(seq, data, Fs, shift, N):
    """seq: (state, position, context) in decoding order; returns the N states + the 16-bit words, as the decoder reads them"""
    Cs = {}
    for c, F in Fs.items():
        C, x = [0] * 256, 0
        for sy in range(256):
            C[sy] = x
            x += F[sy]
        Cs[c] = C
    R = [NX_LOW] * N
    words = []
    for k, p, ctx in reversed(seq):
        f, c = Fs[ctx][data[p]], Cs[ctx][data[p]]
        x = R[k]
        x_max = ((NX_LOW >> shift) << 16) * f
        if x >= x_max:
            words.append(x & 0xFFFF)
            x >>= 16
        R[k] = (x // f << shift) + (x % f) + c
    return b"".join(struct.pack("<I", r) for r in R) + b"".join(struct.pack("<H", w) for w in reversed(words))


def nx16_o0_body(data, N=4, stored_tot=None):
    """order-0 body (no flag byte, no size): alphabet, frequencies (summing to a power of two <= 4096, scaled up by the decoder), states, words"""
    data = bytes(data)
    counts = [0] * 256
    for b in data:
        counts[b] += 1
    tot = stored_tot or min(4096, _pow2_at_least(len(data)))
    F = _normalise_to(counts, max(tot, _pow2_at_least(sum(1 for c in counts if c))))
    table = _alphabet_bytes([1 if f else 0 for f in F]) + b"".join(u7(F[j]) for j in range(256) if F[j])
    up = 4096 // sum(F)
    Fd = {0: [f * up for f in F]}
    seq = [(i % N, i, 0) for i in range(len(data))]
    return table + _nx_encode(seq, data, Fd, 12, N)


def nx16_o1_body(data, N=4, shift=12, compress_table=False):
    data = bytes(data)
    n = len(data)
    q = n // N
    seq = []
    for i in range(q):
        for k in range(N):
            p = k * q + i
            seq.append((k, p, data[p - 1] if i else 0))
    for p in range(N * q, n):
        seq.append((N - 1, p, data[p - 1] if (p > N * q or q) else 0))
    if q == 0:
        seq = [(N - 1, p, data[p - 1] if p else 0) for p in range(n)]
    counts = {}
    for _, p, ctx in seq:
        counts.setdefault(ctx, [0] * 256)[data[p]] += 1
    present = [0] * 256
    for c, v in counts.items():
        present[c] = 1
        for sy in range(256):
            if v[sy]:
                present[sy] = 1
    Fs, Fd = {}, {}
    for c, v in counts.items():
        tot = min(1 << shift, _pow2_at_least(sum(v)))
        tot = max(tot, _pow2_at_least(sum(1 for x in v if x)))
        Fs[c] = _normalise_to(v, tot)
        Fd[c] = [f * ((1 << shift) // tot) for f in Fs[c]]
    table = bytearray(_alphabet_bytes(present))
    alpha = [j for j in range(256) if present[j]]
    for i in alpha:
        row = Fs.get(i, [0] * 256)          # a symbol that is never a context: a row of zeros
        jj = 0
        while jj < len(alpha):
            f = row[alpha[jj]]
            table += u7(f)
            if f == 0:
                run = 0
                while jj + 1 + run < len(alpha) and row[alpha[jj + 1 + run]] == 0:
                    run += 1
                table.append(run)
                jj += run
            jj += 1
    if compress_table:
        body = nx16_o0_body(bytes(table), 4)
        head = bytes([shift << 4 | 1]) + u7(len(table)) + u7(len(body)) + body
    else:
        head = bytes([shift << 4]) + bytes(table)
    return head + _nx_encode(seq, data, Fd, shift, N)


def nx16(data, order=0, x32=False, pack=False, rle=False, stripe=0, cat=False, nosz=False, shift=12, compress_table=False, compress_rle_meta=False):
    """one rANS Nx16 stream: flags, size, transform metadata, body"""
    data = bytes(data)
    N = 32 if x32 else 4
    flags = (1 if order else 0) | (4 if x32 else 0) | (8 if stripe else 0) | (16 if nosz else 0) | (32 if cat else 0) | (64 if rle else 0) | (128 if pack else 0)
    out = bytearray([flags])
    if not nosz:
        out += u7(len(data))
    if stripe:
        parts = [data[j::stripe] for j in range(stripe)]
        comp = [nx16(pt, order=order, x32=x32, nosz=True, shift=shift) for pt in parts]
        out.append(stripe)
        for cpt in comp:
            out += u7(len(cpt))
        return bytes(out) + b"".join(comp)
    if pack:
        syms = sorted(set(data))
        n = len(syms)
        assert n <= 16 or True
        out.append(n & 0xFF)
        if n <= 16:
            out += bytes(syms)
            per = 0 if n <= 1 else 8 if n <= 2 else 4 if n <= 4 else 2
            if per == 0:
                data = b""
            else:
                bits = 8 // per
                idx = {b: i for i, b in enumerate(syms)}
                packed = bytearray()
                for i in range(0, len(data), per):
                    v = 0
                    for k, b in enumerate(data[i:i + per]):
                        v |= idx[b] << (k * bits)
                    packed.append(v)
                data = bytes(packed)
        out += u7(len(data))
    if rle:
        # symbols with runs: every symbol that ever repeats
        run_syms = sorted({data[i] for i in range(1, len(data)) if data[i] == data[i - 1]}) or [data[0] if data else 0]   # (a count of 0 means 256)
        meta = bytearray([len(run_syms) & 0xFF]) + bytes(run_syms)
        lits = bytearray()
        i = 0
        while i < len(data):
            b = data[i]
            j = i + 1
            if b in run_syms:
                while j < len(data) and data[j] == b:
                    j += 1
                meta += u7(j - i - 1)
            lits.append(b)
            i = j
        if compress_rle_meta:
            body = nx16_o0_body(bytes(meta), 4)
            out += u7(len(meta) * 2) + u7(len(lits)) + u7(len(body)) + body
        else:
            out += u7(len(meta) * 2 + 1) + u7(len(lits)) + bytes(meta)
        data = bytes(lits)
    if cat:
        return bytes(out) + data
    if not data:
        return bytes(out)
    return bytes(out) + (nx16_o1_body(data, N, shift, compress_table) if order else nx16_o0_body(data, N))


# ---- the name tokeniser (CRAM 3.1): the encoder's side ----
T_TYPE, T_ALPHA, T_CHAR, T_DIGITS0, T_DZLEN, T_DUP, T_DIFF, T_DIGITS, T_DELTA, T_DELTA0, T_MATCH, T_NOP, T_END = range(13)


def _tokens(name):
    """bytes -> [(kind, text)]: runs of digits (at most 9), runs of letters, single other characters"""
    out, i = [], 0
    while i < len(name):
        c = name[i:i + 1]
        if c.isdigit():
            j = i
            while j < len(name) and name[j:j + 1].isdigit() and j - i < 9:
                j += 1
            out.append(("d", name[i:j]))
        elif c.isalpha():
            j = i
            while j < len(name) and name[j:j + 1].isalpha():
                j += 1
            out.append(("a", name[i:j]))
        else:
            j = i + 1
            out.append(("c", name[i:j]))
        i = j
    return out


def tok3(names, stream_opts=None, implied_type=True, dup_streams=True):
    """names: list of bytes (no NUL).  Every name is compared with the one before it.  implied_type: a token position whose TYPE stream
    would be [type, MATCH, MATCH, ...] for ALL names leaves it out (the first stream's type implies it); dup_streams: a stream whose bytes
    equal an earlier stream's is written as a reference to that one."""
    stream_opts = stream_opts or {}
    S = {}                                   # (position, type) -> bytearray

    def put(t, ty, b):
        S.setdefault((t, ty), bytearray()).extend(b)
    prev = None
    for name in names:
        toks = _tokens(name)
        if prev is not None and name == prev[0]:
            put(0, T_TYPE, [T_DUP]); put(0, T_DUP, struct.pack("<I", 1))
            continue
        put(0, T_TYPE, [T_DIFF]); put(0, T_DIFF, struct.pack("<I", 1 if prev is not None else 0))
        ptoks = prev[1] if prev is not None else []
        for t, (kind, text) in enumerate(toks, start=1):
            pk = ptoks[t - 1] if t - 1 < len(ptoks) else None
            if pk is not None and pk == (kind, text):
                put(t, T_TYPE, [T_MATCH])
            elif kind == "d":
                v = int(text)
                lead = len(text) > 1 and text[:1] == b"0"
                if pk is not None and pk[0] == "d" and 0 <= v - int(pk[1]) < 256 and lead and len(pk[1]) == len(text):
                    put(t, T_TYPE, [T_DELTA0]); put(t, T_DELTA0, [v - int(pk[1])])
                elif pk is not None and pk[0] == "d" and 0 <= v - int(pk[1]) < 256 and not lead and not (len(pk[1]) > 1 and pk[1][:1] == b"0"):
                    put(t, T_TYPE, [T_DELTA]); put(t, T_DELTA, [v - int(pk[1])])
                elif lead:
                    put(t, T_TYPE, [T_DIGITS0]); put(t, T_DIGITS0, struct.pack("<I", v)); put(t, T_DZLEN, [len(text)])
                else:
                    put(t, T_TYPE, [T_DIGITS]); put(t, T_DIGITS, struct.pack("<I", v))
            elif kind == "a" and len(text) > 1:
                put(t, T_TYPE, [T_ALPHA]); put(t, T_ALPHA, text + b"\0")
            else:
                put(t, T_TYPE, [T_CHAR]); put(t, T_CHAR, text)
        put(len(toks) + 1, T_TYPE, [T_END])
        prev = (name, toks)
    out = bytearray(struct.pack("<II", sum(len(n) + 1 for n in names), len(names)) + b"\0")
    written = []
    npos = 1 + max((t for t, _ in S), default=0)
    for t in range(npos):
        types = sorted(ty for (tt, ty) in S if tt == t)
        first = True
        tstream = bytes(S.get((t, T_TYPE), b""))
        skip_type = False
        if implied_type and t > 0 and len(tstream) == len(names) and len(names) > 0 and tstream[1:] == bytes([T_MATCH]) * (len(names) - 1) and \
                tstream[0] not in (T_TYPE, T_END, T_MATCH) and (t, tstream[0]) in S:
            skip_type = True
            types = [tstream[0]] + [ty for ty in types if ty not in (T_TYPE, tstream[0])]
        for ty in types:
            if ty == T_TYPE and skip_type:
                continue
            body = bytes(S[(t, ty)])
            head = ty | (128 if first else 0)
            first = False
            ref = next(((j, k) for (j, k, b) in written if b == body), None) if dup_streams else None
            if ref is not None:
                out += bytes([head | 64, ref[0], ref[1]])
            else:
                comp = nx16(body, **stream_opts)
                out += bytes([head]) + u7(len(comp)) + comp
            written.append((t, ty, body))
    return bytes(out)


METHODS = {"raw": 0, "gzip": 1, "bzip2": 2, "lzma": 3, "rans0": 4, "rans1": 4, "arith": 6, "fqz": 7, "tok3": 8}
NX16_FORMS = {
    "nx16_o0": dict(order=0), "nx16_o1": dict(order=1), "nx16_o0x32": dict(order=0, x32=True), "nx16_o1x32": dict(order=1, x32=True),
    "nx16_o1s10": dict(order=1, shift=10), "nx16_o1ct": dict(order=1, compress_table=True), "nx16_pack": dict(order=0, pack=True),
    "nx16_pack_o1": dict(order=1, pack=True), "nx16_rle": dict(order=0, rle=True), "nx16_rle_cm": dict(order=1, rle=True, compress_rle_meta=True),
    "nx16_packrle": dict(order=0, pack=True, rle=True), "nx16_stripe": dict(order=0, stripe=4), "nx16_stripe_o1": dict(order=1, stripe=3), "nx16_cat": dict(cat=True),
}
for _k in NX16_FORMS:
    METHODS[_k] = 5


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
    if method in NX16_FORMS:
        return nx16(data, **NX16_FORMS[method])
    if method == "tok3":          # (a block of NUL-terminated names)
        return tok3(bytes(data).split(b"\0")[:-1])
    if method in ("arith", "fqz"):  # (CRAM 3.1 codecs this writer cannot produce: only the method byte, for the refusal tests)
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
