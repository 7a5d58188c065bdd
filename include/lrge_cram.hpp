// lrge_cram.hpp -- unaligned CRAM 3.0 input for the C++ host side (the reference reads it through noodles:
// liblrge/src/io.rs:93 sniffs "CRAM", io.rs:154-184 iterates the records and refuses mapped ones).
//
// What basecallers and `samtools import` / `samtools view -C` of an unaligned BAM write: containers of slices whose records are all
// unmapped (reference id -1), bases in the BA data series, names in RN.  This reader decodes exactly that, from the format's own
// description (CRAM format specification v3.0, hts-specs): file definition, container and block structure, the compression header's
// preservation map / data-series encodings / tag encodings, every encoding the specification defines (EXTERNAL, HUFFMAN,
// BYTE_ARRAY_LEN, BYTE_ARRAY_STOP, BETA, SUBEXP, GAMMA, GOLOMB, GOLOMB_RICE) over the core bit stream and the external blocks, and the
// block compression methods of 3.0: raw, gzip, bzip2, lzma, rANS 4x8 (orders 0 and 1).  No reference sequence is ever needed: a
// mapped record -- the only kind that would need one -- is refused with the reference's message.  CRAM 3.1's additional codecs
// (rANS Nx16, adaptive arithmetic, fqzcomp, the name tokeniser) are NOT implemented: a block that uses one is an error naming it.
// Blocks are decompressed on first use, so series this reader never reads (qualities, tag values) may use any codec.
//
// Host-side, I/O-bound, nothing here touches the device.  Pinned by tests/test_input_formats.py against an independent CRAM writer
// (tests/cram_writer.py: the same specification, written from the encoder's side); no third-party CRAM file exists in this image.
// (included by lrge_io.hpp, behind its IoError and its gunzip_all / bunzip2_all / unxz_all)
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <vector>

namespace lrge {
namespace io {
namespace cram {

using Err = IoError;

struct Cursor {              // bounds-checked reader over a byte range
    const unsigned char *p, *e;
    Cursor(const unsigned char *b, size_t n) : p(b), e(b + n) {}
    size_t left() const { return (size_t)(e - p); }
    unsigned u8() { if (p >= e) throw Err("CRAM: truncated"); return *p++; }
    int32_t i32le() { if (left() < 4) throw Err("CRAM: truncated"); int32_t v; std::memcpy(&v, p, 4); p += 4; return v; }
    uint32_t u32le() { return (uint32_t)i32le(); }
    void skip(size_t n) { if (left() < n) throw Err("CRAM: truncated"); p += n; }
    int32_t itf8() {
        const unsigned b0 = u8();
        if (b0 < 0x80) return (int32_t)b0;
        if (b0 < 0xC0) return (int32_t)(((b0 & 0x3F) << 8) | u8());
        if (b0 < 0xE0) { unsigned v = (b0 & 0x1F) << 16; v |= u8() << 8; v |= u8(); return (int32_t)v; }
        if (b0 < 0xF0) { unsigned v = (b0 & 0x0F) << 24; v |= u8() << 16; v |= u8() << 8; v |= u8(); return (int32_t)v; }
        unsigned v = (b0 & 0x0F) << 28; v |= u8() << 20; v |= u8() << 12; v |= u8() << 4; v |= u8() & 0x0F;
        return (int32_t)v;
    }
    int64_t ltf8() {
        const unsigned b0 = u8();
        int extra; uint64_t v;
        if (b0 < 0x80) { extra = 0; v = b0; }
        else if (b0 < 0xC0) { extra = 1; v = b0 & 0x3F; }
        else if (b0 < 0xE0) { extra = 2; v = b0 & 0x1F; }
        else if (b0 < 0xF0) { extra = 3; v = b0 & 0x0F; }
        else if (b0 < 0xF8) { extra = 4; v = b0 & 0x07; }
        else if (b0 < 0xFC) { extra = 5; v = b0 & 0x03; }
        else if (b0 < 0xFE) { extra = 6; v = b0 & 0x01; }
        else if (b0 < 0xFF) { extra = 7; v = 0; }
        else { extra = 8; v = 0; }
        for (int i = 0; i < extra; ++i) v = v << 8 | u8();
        return (int64_t)v;
    }
};

// ---- rANS 4x8 (CRAM 3.0 block method 4) ----
inline std::string rans4x8_decode(const unsigned char *in, size_t n) {
    Cursor c(in, n);
    const unsigned order = c.u8();
    const uint32_t csz = c.u32le(), usz = c.u32le();
    if (order > 1) throw Err("CRAM: rANS order " + std::to_string(order));
    if (c.left() < csz) throw Err("CRAM: truncated rANS stream");
    // a symbol costs at least log2(4096 / 4095) bits: a size field beyond that is not backed by the stream (and nothing is allocated for it)
    if ((uint64_t)usz > ((uint64_t)c.left() + 16) * 24000) throw Err("CRAM: rANS size field exceeds what the stream can hold");
    std::string out(usz, '\0');
    if (usz == 0) return out;
    constexpr uint32_t TOT = 4096, LOW = 1u << 23;
    struct Tab { uint16_t F[256], C[256]; unsigned char R[TOT]; };
    auto read_table = [&](Tab &t) {
        std::memset(&t, 0, sizeof t);
        unsigned x = 0, rle = 0, j = c.u8();
        do {
            unsigned f = c.u8();
            if (f >= 128) f = ((f & 127) << 8) | c.u8();
            if (x + f > TOT) throw Err("CRAM: rANS frequencies exceed 4096");
            t.F[j] = (uint16_t)f; t.C[j] = (uint16_t)x;
            std::memset(t.R + x, (int)j, f);
            x += f;
            if (!rle && c.left() && j + 1 == *c.p) { j = c.u8(); rle = c.u8(); }
            else if (rle) { --rle; ++j; if (j > 255) throw Err("CRAM: rANS symbol run past 255"); }
            else j = c.u8();
        } while (j);
    };
    if (order == 0) {
        std::unique_ptr<Tab> t(new Tab());
        read_table(*t);
        uint32_t R[4];
        for (int k = 0; k < 4; ++k) R[k] = c.u32le();
        for (uint32_t i = 0; i < usz; ++i) {
            uint32_t &r = R[i & 3];
            const uint32_t m = r & (TOT - 1);
            const unsigned char s = t->R[m];
            out[i] = (char)s;
            r = (uint32_t)t->F[s] * (r >> 12) + m - t->C[s];
            while (r < LOW) { if (!c.left()) throw Err("CRAM: rANS stream exhausted"); r = (r << 8) | c.u8(); }
        }
        return out;
    }
    std::vector<std::unique_ptr<Tab>> T(256);
    {
        unsigned rle = 0, i = c.u8();
        do {
            T[i].reset(new Tab());
            read_table(*T[i]);
            if (!rle && c.left() && i + 1 == *c.p) { i = c.u8(); rle = c.u8(); }
            else if (rle) { --rle; ++i; if (i > 255) throw Err("CRAM: rANS context run past 255"); }
            else i = c.u8();
        } while (i);
    }
    uint32_t R[4];
    for (int k = 0; k < 4; ++k) R[k] = c.u32le();
    const uint32_t q = usz >> 2;
    uint32_t idx[4] = {0, q, 2 * q, 3 * q};
    unsigned last[4] = {0, 0, 0, 0};
    auto step = [&](int k) {
        const Tab *t = T[last[k]].get();
        if (!t) throw Err("CRAM: rANS order-1 context without a table");
        const uint32_t m = R[k] & (TOT - 1);
        const unsigned char s = t->R[m];
        out[idx[k]++] = (char)s;
        R[k] = (uint32_t)t->F[s] * (R[k] >> 12) + m - t->C[s];
        while (R[k] < LOW) { if (!c.left()) throw Err("CRAM: rANS stream exhausted"); R[k] = (R[k] << 8) | c.u8(); }
        last[k] = s;
    };
    for (uint32_t i = 0; i < q; ++i) { step(0); step(1); step(2); step(3); }
    while (idx[3] < usz) step(3);
    return out;
}

// ---- blocks ----
struct Block {
    int method = 0, content_type = 0; int32_t content_id = 0; int32_t raw_size = 0;
    const unsigned char *data = nullptr; size_t size = 0;
    bool ready = false; std::string bytes;      // decompressed on first use
    size_t pos = 0;                              // read cursor (external blocks)
    const std::string &get() {
        if (ready) return bytes;
        switch (method) {
        case 0: bytes.assign((const char *)data, size); break;
        case 1: bytes = size ? gunzip_all(std::string((const char *)data, size)) : std::string(); break;
        case 2: bytes = bunzip2_all(std::string((const char *)data, size)); break;
        case 3: bytes = unxz_all(std::string((const char *)data, size)); break;
        case 4: bytes = rans4x8_decode(data, size); break;
        case 5: throw Err("CRAM 3.1 block codec rANS Nx16 is not supported by this reader (write CRAM 3.0, or convert with `samtools fastq`)");
        case 6: throw Err("CRAM 3.1 block codec `adaptive arithmetic coder` is not supported by this reader (write CRAM 3.0, or convert with `samtools fastq`)");
        case 7: throw Err("CRAM 3.1 block codec fqzcomp is not supported by this reader (write CRAM 3.0, or convert with `samtools fastq`)");
        case 8: throw Err("CRAM 3.1 block codec `name tokeniser` is not supported by this reader (write CRAM 3.0, or convert with `samtools fastq`)");
        default: throw Err("CRAM: unknown block compression method " + std::to_string(method));
        }
        if (raw_size >= 0 && bytes.size() != (size_t)raw_size) throw Err("CRAM: block size mismatch after decompression");
        ready = true;
        return bytes;
    }
};
inline Block read_block(Cursor &c, int major) {
    Block b;
    b.method = (int)c.u8(); b.content_type = (int)c.u8(); b.content_id = c.itf8();
    const int32_t sz = c.itf8(); b.raw_size = c.itf8();
    if (sz < 0 || (size_t)sz > c.left()) throw Err("CRAM: truncated block");
    b.data = c.p; b.size = (size_t)sz; c.skip((size_t)sz);
    if (major >= 3) c.skip(4);          // CRC32
    return b;
}

// ---- bit stream over the core block (most significant bit first) ----
struct Bits {
    const std::string *s = nullptr; size_t byte = 0; int bit = 7;
    unsigned get1() {
        if (!s || byte >= s->size()) throw Err("CRAM: core block exhausted");
        const unsigned v = ((unsigned char)(*s)[byte] >> bit) & 1u;
        if (--bit < 0) { bit = 7; ++byte; }
        return v;
    }
    uint32_t get(int n) { uint32_t v = 0; for (int i = 0; i < n; ++i) v = v << 1 | get1(); return v; }
};

// ---- encodings ----
struct Slice;
struct Encoding {
    int id = 0;
    int32_t ext_id = -1;                       // EXTERNAL, BYTE_ARRAY_STOP
    int32_t offset = 0, param = 0;             // BETA (nbits), SUBEXP (k), GAMMA, GOLOMB (M), GOLOMB_RICE (log2 M)
    unsigned stop = 0;                         // BYTE_ARRAY_STOP
    std::vector<int32_t> alphabet, lens;       // HUFFMAN
    std::vector<std::pair<uint32_t, int>> codes;   // (code, index into alphabet), sorted by (len, symbol)
    std::shared_ptr<Encoding> len_enc, val_enc;    // BYTE_ARRAY_LEN
    bool set = false;
};
inline Encoding read_encoding(Cursor &c) {
    Encoding e; e.set = true;
    e.id = c.itf8();
    const int32_t n = c.itf8();
    if (n < 0 || (size_t)n > c.left()) throw Err("CRAM: truncated encoding");
    Cursor p(c.p, (size_t)n); c.skip((size_t)n);
    switch (e.id) {
    case 0: break;
    case 1: e.ext_id = p.itf8(); break;
    case 2: e.offset = p.itf8(); e.param = p.itf8(); break;
    case 3: {
        int32_t na = p.itf8(); for (int32_t i = 0; i < na; ++i) e.alphabet.push_back(p.itf8());
        int32_t nl = p.itf8(); for (int32_t i = 0; i < nl; ++i) e.lens.push_back(p.itf8());
        if (na != nl || na <= 0) throw Err("CRAM: malformed HUFFMAN encoding");
        std::vector<int> ord((size_t)na);
        for (int i = 0; i < na; ++i) ord[(size_t)i] = i;
        std::sort(ord.begin(), ord.end(), [&](int a, int b) { return e.lens[(size_t)a] != e.lens[(size_t)b] ? e.lens[(size_t)a] < e.lens[(size_t)b] : e.alphabet[(size_t)a] < e.alphabet[(size_t)b]; });
        uint32_t code = 0; int cur = e.lens[(size_t)ord[0]];
        for (int k = 0; k < na; ++k) {
            const int i = ord[(size_t)k];
            while (cur < e.lens[(size_t)i]) { code <<= 1; ++cur; }
            e.codes.emplace_back(code, i);
            ++code;
        }
        break;
    }
    case 4: e.len_enc.reset(new Encoding(read_encoding(p))); e.val_enc.reset(new Encoding(read_encoding(p))); break;
    case 5: e.stop = p.u8(); e.ext_id = p.itf8(); break;
    case 6: e.offset = p.itf8(); e.param = p.itf8(); break;
    case 7: e.offset = p.itf8(); e.param = p.itf8(); break;
    case 8: e.offset = p.itf8(); e.param = p.itf8(); break;
    case 9: e.offset = p.itf8(); break;
    default: throw Err("CRAM: unknown encoding " + std::to_string(e.id));
    }
    return e;
}

struct Slice {
    Bits core;
    std::map<int32_t, Block *> ext;
    std::set<int32_t> lazy;        // external blocks only series this reader never VALUE-reads live in (qualities, tag values): never decompressed
    Block *block(int32_t id) {
        auto it = ext.find(id);
        if (it == ext.end()) throw Err("CRAM: external block " + std::to_string(id) + " is missing from the slice");
        return it->second;
    }
    int32_t read_int(const Encoding &e) {
        switch (e.id) {
        case 1: { Block *b = block(e.ext_id); const std::string &d = b->get(); Cursor c((const unsigned char *)d.data() + b->pos, d.size() - b->pos); const int32_t v = c.itf8(); b->pos = (size_t)(c.p - (const unsigned char *)d.data()); return v; }
        case 3: {
            if (e.codes.size() == 1 && e.lens[(size_t)e.codes[0].second] == 0) return e.alphabet[(size_t)e.codes[0].second];
            uint32_t code = 0; int len = 0; size_t k = 0;
            for (;;) {
                code = code << 1 | core.get1(); ++len;
                while (k < e.codes.size() && e.lens[(size_t)e.codes[k].second] < len) ++k;
                for (size_t j = k; j < e.codes.size() && e.lens[(size_t)e.codes[j].second] == len; ++j)
                    if (e.codes[j].first == code) return e.alphabet[(size_t)e.codes[j].second];
                if (len > 31) throw Err("CRAM: invalid HUFFMAN code");
            }
        }
        case 6: return (int32_t)core.get(e.param) - e.offset;
        case 7: {
            int u = 0; while (core.get1()) ++u;
            int b; uint32_t v;
            if (u == 0) { b = e.param; v = core.get(b); } else { b = u + e.param - 1; v = (1u << b) | core.get(b); }
            return (int32_t)v - e.offset;
        }
        case 9: { int n = 0; while (!core.get1()) ++n; const uint32_t v = (1u << n) | core.get(n); return (int32_t)v - e.offset; }
        case 2: case 8: {
            const uint32_t M = e.id == 8 ? (1u << e.param) : (uint32_t)e.param;
            if (M == 0) throw Err("CRAM: GOLOMB with M = 0");
            uint32_t q = 0; while (core.get1()) ++q;
            uint32_t r;
            if (e.id == 8) r = core.get(e.param);
            else {
                int b = 0; while ((1u << b) < M) ++b;
                const uint32_t cut = (1u << b) - M;
                r = b ? core.get(b - 1) : 0;
                if (r >= cut) r = (r << 1 | core.get1()) - cut;
            }
            return (int32_t)(q * M + r) - e.offset;
        }
        default: throw Err("CRAM: encoding " + std::to_string(e.id) + " cannot yield an integer");
        }
    }
    unsigned read_byte(const Encoding &e) {
        if (e.id == 1) { Block *b = block(e.ext_id); const std::string &d = b->get(); if (b->pos >= d.size()) throw Err("CRAM: external block exhausted"); return (unsigned char)d[b->pos++]; }
        return (unsigned)read_int(e) & 0xffu;
    }
    void read_bytes(const Encoding &e, std::string *out) {          // a byte array (out may be null: skipped)
        if (e.id == 5) {
            Block *b = block(e.ext_id); const std::string &d = b->get();
            const size_t st = b->pos;
            const void *z = std::memchr(d.data() + st, (int)e.stop, d.size() - st);
            if (!z) throw Err("CRAM: BYTE_ARRAY_STOP without its stop byte");
            const size_t en = (size_t)((const char *)z - d.data());
            if (out) out->assign(d.data() + st, en - st);
            b->pos = en + 1;
            return;
        }
        if (e.id == 4) {
            const int32_t n = read_int(*e.len_enc);
            if (n < 0) throw Err("CRAM: negative byte array length");
            if (e.val_enc->id == 1 && !out && lazy.count(e.val_enc->ext_id)) return;      // nobody else reads that block: its bytes need not even be decompressed
            if (e.val_enc->id == 1) {
                Block *b = block(e.val_enc->ext_id); const std::string &d = b->get();
                if (b->pos + (size_t)n > d.size()) throw Err("CRAM: external block exhausted");
                if (out) out->assign(d.data() + b->pos, (size_t)n);
                b->pos += (size_t)n;
            } else {
                if (out) out->clear();
                for (int32_t i = 0; i < n; ++i) { const unsigned v = read_byte(*e.val_enc); if (out) out->push_back((char)v); }
            }
            return;
        }
        throw Err("CRAM: encoding " + std::to_string(e.id) + " cannot yield a byte array");
    }
};

struct CompressionHeader {
    bool rn_preserved = true;
    std::vector<std::vector<int32_t>> tag_lines;        // TD: per line, the tag ids (name << 8 | type)
    std::map<std::string, Encoding> ds;                 // data series
    std::map<int32_t, Encoding> tags;
    std::set<int32_t> lazy;                             // see Slice::lazy
    static void ext_ids(const Encoding &e, std::set<int32_t> &out, bool values_too) {
        if (e.id == 1 || e.id == 5) out.insert(e.ext_id);
        if (e.id == 4) { ext_ids(*e.len_enc, out, true); if (values_too) ext_ids(*e.val_enc, out, true); }
    }
    void find_lazy() {
        std::set<int32_t> needed, maybe;
        for (const auto &kv : ds) { if (kv.first == "QS") ext_ids(kv.second, maybe, true); else ext_ids(kv.second, needed, true); }
        for (const auto &kv : tags) { ext_ids(kv.second, needed, false); if (kv.second.id == 4) ext_ids(*kv.second.val_enc, maybe, true); else ext_ids(kv.second, needed, true); }
        for (int32_t id : maybe) if (!needed.count(id)) lazy.insert(id);
    }
    const Encoding &series(const char *k) const {
        auto it = ds.find(k);
        if (it == ds.end() || !it->second.set) throw Err(std::string("CRAM: data series ") + k + " has no encoding");
        return it->second;
    }
};
inline CompressionHeader read_compression_header(const std::string &d) {
    CompressionHeader h;
    Cursor c((const unsigned char *)d.data(), d.size());
    {   // preservation map
        const int32_t sz = c.itf8();
        if (sz < 0 || (size_t)sz > c.left()) throw Err("CRAM: truncated preservation map");
        Cursor p(c.p, (size_t)sz); c.skip((size_t)sz);
        const int32_t n = p.itf8();
        for (int32_t i = 0; i < n; ++i) {
            const char k0 = (char)p.u8(), k1 = (char)p.u8();
            if (k0 == 'R' && k1 == 'N') h.rn_preserved = p.u8() != 0;
            else if ((k0 == 'A' && k1 == 'P') || (k0 == 'R' && k1 == 'R')) p.u8();
            else if (k0 == 'S' && k1 == 'M') p.skip(5);
            else if (k0 == 'T' && k1 == 'D') {
                const int32_t len = p.itf8();
                if (len < 0 || (size_t)len > p.left()) throw Err("CRAM: truncated tag dictionary");
                std::vector<int32_t> line;
                for (int32_t j = 0; j < len;) {
                    if (p.p[j] == 0) { h.tag_lines.push_back(line); line.clear(); ++j; continue; }
                    if (j + 3 > len) throw Err("CRAM: malformed tag dictionary");
                    line.push_back((int32_t)p.p[j] << 16 | (int32_t)p.p[j + 1] << 8 | (int32_t)p.p[j + 2]);
                    j += 3;
                }
                if (!line.empty()) h.tag_lines.push_back(line);
                p.skip((size_t)len);
            } else throw Err(std::string("CRAM: unknown preservation key ") + k0 + k1);
        }
    }
    {   // data series encodings
        const int32_t sz = c.itf8();
        if (sz < 0 || (size_t)sz > c.left()) throw Err("CRAM: truncated data series map");
        Cursor p(c.p, (size_t)sz); c.skip((size_t)sz);
        const int32_t n = p.itf8();
        for (int32_t i = 0; i < n; ++i) { std::string k; k.push_back((char)p.u8()); k.push_back((char)p.u8()); h.ds[k] = read_encoding(p); }
    }
    {   // tag encodings
        const int32_t sz = c.itf8();
        if (sz < 0 || (size_t)sz > c.left()) throw Err("CRAM: truncated tag encoding map");
        Cursor p(c.p, (size_t)sz); c.skip((size_t)sz);
        const int32_t n = p.itf8();
        for (int32_t i = 0; i < n; ++i) { const int32_t k = p.itf8(); h.tags[k] = read_encoding(p); }
    }
    h.find_lazy();
    return h;
}

using Callback = std::function<void(const std::string &name, const std::string &seq)>;

// every record of the file: callback(read name, bases); a mapped record throws Err(mapped_msg)
inline void parse(const std::string &file, const Callback &cb, const char *mapped_msg) {
    Cursor f((const unsigned char *)file.data(), file.size());
    if (f.left() < 26 || std::memcmp(f.p, "CRAM", 4) != 0) throw Err("not a CRAM file");
    const int major = f.p[4], minor = f.p[5];
    if (major != 3 && major != 2) throw Err("CRAM version " + std::to_string(major) + "." + std::to_string(minor) + " is not supported (2.x / 3.x)");
    f.skip(26);
    bool first = true;
    std::string name, seq, tmp;
    while (f.left() > 0) {
        const int32_t clen = f.i32le();
        const int32_t c_ref = f.itf8(); const int32_t c_start = f.itf8(); f.itf8();
        const int32_t c_nrec = f.itf8();
        if (major >= 3) f.ltf8(); else f.itf8();
        f.ltf8();
        const int32_t n_blocks = f.itf8();
        const int32_t n_land = f.itf8(); for (int32_t i = 0; i < n_land; ++i) f.itf8();
        if (major >= 3) f.skip(4);
        if (clen < 0 || (size_t)clen > f.left()) throw Err("CRAM: truncated container");
        Cursor c(f.p, (size_t)clen); f.skip((size_t)clen);
        if (first) { first = false; continue; }                               // the SAM header container
        if (c_nrec == 0 || (c_ref == -1 && c_start == 4542278 && n_blocks <= 1)) continue;   // EOF marker / empty container
        Block chb = read_block(c, major);
        if (chb.content_type != 1) throw Err("CRAM: a data container must start with its compression header");
        const CompressionHeader H = read_compression_header(chb.get());
        while (c.left() > 0) {
            Block shb = read_block(c, major);
            if (shb.content_type != 2) throw Err("CRAM: expected a slice header block");
            const std::string &sh = shb.get();
            Cursor s((const unsigned char *)sh.data(), sh.size());
            const int32_t s_ref = s.itf8(); s.itf8(); s.itf8();
            const int32_t s_nrec = s.itf8();
            if (major >= 3) s.ltf8(); else s.itf8();
            const int32_t s_nblocks = s.itf8();
            std::vector<Block> blocks((size_t)std::max(0, s_nblocks));
            Slice S; S.lazy = H.lazy;
            for (int32_t i = 0; i < s_nblocks; ++i) blocks[(size_t)i] = read_block(c, major);
            for (Block &b : blocks) {
                if (b.content_type == 5) S.core.s = &b.get();
                else if (b.content_type == 4) S.ext[b.content_id] = &b;
            }
            for (int32_t r = 0; r < s_nrec; ++r) {
                const int32_t bf = S.read_int(H.series("BF"));
                const int32_t cf = S.read_int(H.series("CF"));
                if (s_ref == -2) (void)S.read_int(H.series("RI"));
                const int32_t rl = S.read_int(H.series("RL"));
                (void)S.read_int(H.series("AP"));
                (void)S.read_int(H.series("RG"));
                name.clear();
                if (H.rn_preserved) S.read_bytes(H.series("RN"), &name);
                if (cf & 0x2) {                                               // detached: mate data stored
                    (void)S.read_int(H.series("MF"));
                    if (!H.rn_preserved) S.read_bytes(H.series("RN"), &name);
                    (void)S.read_int(H.series("NS")); (void)S.read_int(H.series("NP")); (void)S.read_int(H.series("TS"));
                } else if (cf & 0x4) (void)S.read_int(H.series("NF"));
                const int32_t tl = S.read_int(H.series("TL"));
                if (tl < 0 || (size_t)tl >= H.tag_lines.size()) { if (!(tl == 0 && H.tag_lines.empty())) throw Err("CRAM: tag line out of range"); }
                else for (int32_t id : H.tag_lines[(size_t)tl]) {
                    auto it = H.tags.find(id);
                    if (it == H.tags.end()) throw Err("CRAM: a tag has no encoding");
                    S.read_bytes(it->second, nullptr);
                }
                if (!(bf & 0x4)) throw Err(mapped_msg);                       // io.rs:162-167: mapped records are refused
                if (rl < 0) throw Err("CRAM: negative read length");
                seq.clear();
                if (!(cf & 0x8)) { const Encoding &ba = H.series("BA"); seq.clear(); seq.reserve((size_t)std::min<int32_t>(rl, 1 << 20)); for (int32_t i = 0; i < rl; ++i) seq.push_back((char)S.read_byte(ba)); }
                if (cf & 0x1) {       // qualities: skipped -- without touching their block when it is theirs alone (any codec will do then)
                    const Encoding &qs = H.series("QS");
                    if (!(qs.id == 1 && S.lazy.count(qs.ext_id))) for (int32_t i = 0; i < rl; ++i) (void)S.read_byte(qs);
                }
                if (name == "*") name.clear();
                cb(name, seq);
            }
        }
    }
}

}  // namespace cram
}  // namespace io
}  // namespace lrge
