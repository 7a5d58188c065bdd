// lrge_cram.hpp -- unaligned CRAM 3.0 / 3.1 input for the C++ host side (the reference reads it through noodles:
// liblrge/src/io.rs:93 sniffs "CRAM", io.rs:154-184 iterates the records and refuses mapped ones).
//
// What basecallers and `samtools import` / `samtools view -C` of an unaligned BAM write: containers of slices whose records are all
// unmapped (reference id -1), bases in the BA data series, names in RN.  This reader decodes exactly that, from the format's own
// description (CRAM format specification v3.0, hts-specs): file definition, container and block structure, the compression header's
// preservation map / data-series encodings / tag encodings, every encoding the specification defines (EXTERNAL, HUFFMAN,
// BYTE_ARRAY_LEN, BYTE_ARRAY_STOP, BETA, SUBEXP, GAMMA, GOLOMB, GOLOMB_RICE) over the core bit stream and the external blocks, and the
// block compression methods of 3.0: raw, gzip, bzip2, lzma, rANS 4x8 (orders 0 and 1).  No reference sequence is ever needed: a
// mapped record -- the only kind that would need one -- is refused with the reference's message.  Of CRAM 3.1's additional codecs
// (CRAM codecs specification) the two that the default profile of samtools >= 1.22 uses are decoded: rANS Nx16 (orders 0 / 1, 4 / 32
// states, bit packing, run lengths, striping) and the name tokeniser over rANS Nx16 streams; the adaptive arithmetic coder and fqzcomp
// (archive profiles; fqzcomp holds qualities only) are NOT: a block in one of them that the reader needs is an error naming it.
// Blocks are decompressed on first use, so series this reader never reads (qualities, tag values) may use any codec.
// The 3.1 decoders are written from the specification with no third-party vector to check them against (none exists in this image):
// every rANS Nx16 body must end with its states back at the encoder's start value, so a misreading ends in an error, not in wrong bases.
//
// Host-side, I/O-bound, nothing here touches the device.  Pinned by tests/test_input_formats.py against an independent CRAM writer
// (tests/cram_writer.py: the same specification, written from the encoder's side); no third-party CRAM file exists in this image.
// (included by lrge_io.hpp, behind its IoError and its gunzip_all / bunzip2_all / unxz_all)
#pragma once
#include <algorithm>
#include <array>
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

// ---- rANS Nx16 (CRAM 3.1 block method 5; CRAM codecs specification, "rANS Nx16") ----
// A stream = one flag byte [ORDER 0x01 | X32 0x04 | STRIPE 0x08 | NOSZ 0x10 | CAT 0x20 | RLE 0x40 | PACK 0x80], the uncompressed size as a
// 7-bit variable-length integer unless NOSZ, then the transforms' metadata (bit packing, run lengths) and the entropy-coded body: 4 or 32
// interleaved 32-bit states renormalised 16 bits at a time, order 0 or order 1, frequencies of 12 (order 1: `shift`) bits.
// INTEGRITY: every rANS encoder starts its states at the lower bound (1 << 15), so a decoder that took every symbol back ends there; that
// is checked for all N states -- a stream this decoder misreads (nothing in this image can produce a third-party vector) ends in an
// error, not in wrong bases.
struct Nx16 {
    static uint32_t u7(Cursor &c) {
        uint32_t v = 0; unsigned b; int n = 0;
        do { b = c.u8(); if (++n > 5) throw Err("CRAM: rANS Nx16 integer too long"); v = (v << 7) | (b & 0x7f); } while (b & 0x80);
        return v;
    }
    static uint32_t u16le(Cursor &c) { if (c.left() < 2) throw Err("CRAM: rANS Nx16 stream exhausted"); const uint32_t v = c.p[0] | (uint32_t)c.p[1] << 8; c.p += 2; return v; }
    static constexpr uint32_t LOW = 1u << 15;
    struct Tab { uint16_t F[256], C[256]; unsigned char R[4096]; };
    static void alphabet(Cursor &c, bool A[256]) {
        std::memset(A, 0, 256 * sizeof(bool));
        unsigned rle = 0, j = c.u8();
        do {
            A[j] = true;
            if (!rle && c.left() && j + 1 == *c.p) { j = c.u8(); rle = c.u8(); }
            else if (rle) { --rle; ++j; if (j > 255) throw Err("CRAM: rANS Nx16 symbol run past 255"); }
            else j = c.u8();
        } while (j);
    }
    // frequencies of the symbols of A (already read) -> a table of 1 << shift slots; F is scaled up by a power of two when it sums to less
    static void finish(Tab &t, uint32_t F[256], unsigned shift) {
        const uint32_t tot = 1u << shift;
        uint64_t sum = 0;
        for (int j = 0; j < 256; ++j) sum += F[j];
        if (sum == 0 || sum > tot) throw Err("CRAM: rANS Nx16 frequencies do not fit the table");
        unsigned up = 0;
        while ((sum << up) < tot) ++up;
        if ((sum << up) != tot) throw Err("CRAM: rANS Nx16 frequencies do not sum to a power of two");
        uint32_t x = 0;
        for (int j = 0; j < 256; ++j) {
            const uint32_t f = F[j] << up;
            t.F[j] = (uint16_t)f; t.C[j] = (uint16_t)x;        // (f == 4096 only as the single symbol: stored as 0 in 16 bits? no: 4096 fits)
            if (f) std::memset(t.R + x, j, f);
            x += f;
        }
    }
    static void read_tab0(Cursor &c, Tab &t) {
        bool A[256]; alphabet(c, A);
        uint32_t F[256] = {0};
        for (int j = 0; j < 256; ++j) if (A[j]) F[j] = u7(c);
        finish(t, F, 12);
    }
    static std::string decode0(Cursor &c, size_t len, int N) {
        std::string out(len, '\0');
        if (!len) return out;
        std::unique_ptr<Tab> t(new Tab());
        read_tab0(c, *t);
        uint32_t R[32];
        for (int k = 0; k < N; ++k) R[k] = c.u32le();
        for (size_t i = 0; i < len; ++i) {
            uint32_t &r = R[i & (size_t)(N - 1)];
            const uint32_t m = r & 4095;
            const unsigned char s = t->R[m];
            out[i] = (char)s;
            r = (uint32_t)t->F[s] * (r >> 12) + m - t->C[s];
            if (r < LOW) r = r << 16 | u16le(c);
        }
        for (int k = 0; k < N; ++k) if (R[k] != LOW) throw Err("CRAM: rANS Nx16 stream does not end in its initial state (damaged, or a variant this reader misreads)");
        return out;
    }
    static std::string decode1(Cursor &c, size_t len, int N) {
        std::string out(len, '\0');
        if (!len) return out;
        const unsigned comp = c.u8(), shift = comp >> 4;
        if (shift < 1 || shift > 12) throw Err("CRAM: rANS Nx16 order-1 table of " + std::to_string(shift) + " bits");
        std::vector<std::unique_ptr<Tab>> T(256);
        std::string tbuf;
        auto read_tabs = [&](Cursor &tc) {
            bool A[256]; alphabet(tc, A);
            for (int i = 0; i < 256; ++i) {
                if (!A[i]) continue;
                uint32_t F[256] = {0};
                unsigned run = 0;
                for (int j = 0; j < 256; ++j) {
                    if (!A[j]) continue;
                    if (run) { --run; continue; }
                    F[j] = u7(tc);
                    if (!F[j]) run = tc.u8();
                }
                bool any = false;
                for (int j = 0; j < 256; ++j) any |= F[j] != 0;
                if (!any) continue;                               // a symbol that is never a context: a row of zeros
                T[(size_t)i].reset(new Tab());
                finish(*T[(size_t)i], F, shift);
            }
        };
        if (comp & 1) {                                           // the table itself went through the order-0 coder (4 states)
            const uint32_t ulen = u7(c), clen = u7(c);
            if (c.left() < clen) throw Err("CRAM: truncated rANS Nx16 table");
            if ((uint64_t)ulen > ((uint64_t)clen + 16) * 24000) throw Err("CRAM: rANS Nx16 table size exceeds what its bytes can hold");
            Cursor sub(c.p, clen); c.skip(clen);
            tbuf = decode0(sub, ulen, 4);
            Cursor tc((const unsigned char *)tbuf.data(), tbuf.size());
            read_tabs(tc);
        } else read_tabs(c);
        uint32_t R[32]; size_t idx[32]; unsigned last[32];
        for (int k = 0; k < N; ++k) R[k] = c.u32le();
        const size_t q = len / (size_t)N;
        for (int k = 0; k < N; ++k) { idx[k] = (size_t)k * q; last[k] = 0; }
        const uint32_t mask = (1u << shift) - 1;
        auto step = [&](int k) {
            const Tab *t = T[last[k]].get();
            if (!t) throw Err("CRAM: rANS Nx16 order-1 context without a table");
            const uint32_t m = R[k] & mask;
            const unsigned char s = t->R[m];
            out[idx[k]++] = (char)s;
            R[k] = (uint32_t)t->F[s] * (R[k] >> shift) + m - t->C[s];
            if (R[k] < LOW) R[k] = R[k] << 16 | u16le(c);
            last[k] = s;
        };
        for (size_t i = 0; i < q; ++i) for (int k = 0; k < N; ++k) step(k);
        while (idx[N - 1] < len) step(N - 1);
        for (int k = 0; k < N; ++k) if (R[k] != LOW) throw Err("CRAM: rANS Nx16 stream does not end in its initial state (damaged, or a variant this reader misreads)");
        return out;
    }
    // one stream from c; known != SIZE_MAX: the size the container of this stream expects (NOSZ streams carry none)
    static std::string decode(Cursor &c, size_t known = (size_t)-1, int depth = 0) {
        if (depth > 2) throw Err("CRAM: rANS Nx16 streams nested too deep");
        const unsigned flags = c.u8();
        size_t len;
        if (flags & 0x10) { if (known == (size_t)-1) throw Err("CRAM: rANS Nx16 stream without a size"); len = known; }
        else { len = u7(c); if (known != (size_t)-1 && len != known) throw Err("CRAM: rANS Nx16 stream size differs from its container's"); }
        const int N = (flags & 0x04) ? 32 : 4;
        if (flags & 0x08) {                                       // STRIPE: N2 streams, byte i of the data in stream i % N2
            const unsigned n2 = c.u8();
            if (!n2) throw Err("CRAM: rANS Nx16 stripe of 0 streams");
            std::vector<uint32_t> clen(n2);
            for (auto &v : clen) v = u7(c);
            std::vector<std::string> part(n2);
            for (unsigned j = 0; j < n2; ++j) {
                if (c.left() < clen[j]) throw Err("CRAM: truncated rANS Nx16 stripe");
                Cursor sub(c.p, clen[j]); c.skip(clen[j]);
                part[j] = decode(sub, len / n2 + (len % n2 > j ? 1 : 0), depth + 1);
            }
            std::string out(len, '\0');
            for (size_t i = 0; i < len; ++i) out[i] = part[i % n2][i / n2];
            return out;
        }
        // a size the remaining bytes cannot hold is refused before anything is allocated for it (PACK x8, RLE and the entropy coder multiply)
        if ((uint64_t)len > ((uint64_t)c.left() + 16) * 24000ull * 8ull * 255ull) throw Err("CRAM: rANS Nx16 size field exceeds what the stream can hold");
        const size_t final_len = len;
        unsigned char pmap[16]; unsigned per = 1; bool one = false;   // PACK: symbols per byte
        if (flags & 0x80) {
            unsigned n = c.u8();
            if (n == 0) n = 256;
            if (n <= 1) { per = 0; one = true; } else if (n <= 2) per = 8; else if (n <= 4) per = 4; else if (n <= 16) per = 2; else per = 1;
            if (n <= 16) for (unsigned i = 0; i < n; ++i) pmap[i] = (unsigned char)c.u8();
            len = u7(c);                                          // the packed length
            if (per >= 2 && len != (final_len + per - 1) / per) throw Err("CRAM: rANS Nx16 packed length does not match");
            if (per == 1 && len != final_len) throw Err("CRAM: rANS Nx16 packed length does not match");
        }
        const size_t packed_len = len;
        std::string rmeta; bool rle = false;
        if (flags & 0x40) {
            rle = true;
            const uint32_t umeta = u7(c);
            len = u7(c);                                          // the literals
            if (umeta & 1) { const size_t n = umeta >> 1; if (c.left() < n) throw Err("CRAM: truncated rANS Nx16 run lengths"); rmeta.assign((const char *)c.p, n); c.skip(n); }
            else {
                const uint32_t cmeta = u7(c);
                if (c.left() < cmeta) throw Err("CRAM: truncated rANS Nx16 run lengths");
                if ((uint64_t)(umeta >> 1) > ((uint64_t)cmeta + 16) * 24000) throw Err("CRAM: rANS Nx16 run-length size exceeds what its bytes can hold");
                Cursor sub(c.p, cmeta); c.skip(cmeta);
                rmeta = decode0(sub, umeta >> 1, 4);
            }
            if (len > packed_len) throw Err("CRAM: rANS Nx16 literals exceed the data");
        }
        if ((uint64_t)len > ((uint64_t)c.left() + 16) * 24000) throw Err("CRAM: rANS Nx16 size field exceeds what the stream can hold");
        std::string data;
        if (flags & 0x20) { if (c.left() < len) throw Err("CRAM: truncated rANS Nx16 stream"); data.assign((const char *)c.p, len); c.skip(len); }
        else data = (flags & 0x01) ? decode1(c, len, N) : decode0(c, len, N);
        if (rle) {
            Cursor m((const unsigned char *)rmeta.data(), rmeta.size());
            unsigned n = m.u8(); if (n == 0) n = 256;
            bool L[256] = {false};
            for (unsigned i = 0; i < n; ++i) L[m.u8()] = true;
            std::string out; out.reserve(std::min<size_t>(packed_len, data.size() * 4 + 64));
            for (unsigned char b : data) {
                size_t copies = 1;
                if (L[b]) copies += u7(m);
                if (out.size() + copies > packed_len) throw Err("CRAM: rANS Nx16 runs exceed the data");
                out.append(copies, (char)b);
            }
            if (out.size() != packed_len) throw Err("CRAM: rANS Nx16 runs do not add up");
            data.swap(out);
        }
        if (flags & 0x80) {
            std::string out;
            if (one) out.assign(final_len, (char)pmap[0]);
            else if (per == 1) out.swap(data);
            else {
                out.resize(final_len);
                const unsigned bits = 8 / per, msk = (1u << bits) - 1;
                size_t o = 0;
                for (unsigned char b : data) for (unsigned k = 0; k < per && o < final_len; ++k) out[o++] = (char)pmap[(b >> (k * bits)) & msk];
                if (o != final_len) throw Err("CRAM: rANS Nx16 packed data too short");
            }
            data.swap(out);
        }
        if (data.size() != final_len) throw Err("CRAM: rANS Nx16 stream size mismatch");
        return data;
    }
};

// ---- the name tokeniser (CRAM 3.1 block method 8; CRAM codecs specification, "Name tokenisation codec") ----
// A name is a list of tokens compared with an earlier name's: token t's type and values come from per-(token position, type) byte
// streams, each compressed on its own (rANS Nx16 here; streams of the adaptive arithmetic coder are refused by name).
inline std::string tok3_decode(const unsigned char *in, size_t n) {
    enum { T_TYPE = 0, T_ALPHA, T_CHAR, T_DIGITS0, T_DZLEN, T_DUP, T_DIFF, T_DIGITS, T_DELTA, T_DELTA0, T_MATCH, T_NOP, T_END };
    Cursor c(in, n);
    const uint32_t ulen = c.u32le(), nreads = c.u32le();
    const unsigned use_arith = c.u8();
    if ((uint64_t)nreads > (uint64_t)ulen) throw Err("CRAM: name tokeniser: more names than bytes");
    if ((uint64_t)ulen > ((uint64_t)n + 16) * 24000ull * 64ull) throw Err("CRAM: name tokeniser size field exceeds what the stream can hold");
    struct Stream { std::string buf; size_t pos = 0; bool present = false; };
    std::vector<std::array<Stream, 16>> desc;
    while (c.left()) {
        const unsigned ttype = c.u8(), type = ttype & 15;
        if (type > T_END) throw Err("CRAM: name tokeniser: unknown token type " + std::to_string(type));
        if (ttype & 128) {
            if (desc.size() >= 128) throw Err("CRAM: name tokeniser: too many token positions");
            desc.emplace_back();
            if (type != T_TYPE) {                                 // the TYPE stream is implied: this type for the first name, MATCH for the others
                Stream &ts = desc.back()[T_TYPE];
                ts.buf.assign(nreads, (char)T_MATCH); if (nreads) ts.buf[0] = (char)type; ts.present = true;
            }
        }
        if (desc.empty()) throw Err("CRAM: name tokeniser: a stream before the first token position");
        Stream &st = desc.back()[type];
        if (ttype & 64) {                                         // the same bytes as an earlier stream
            const unsigned j = c.u8(), k = c.u8();
            if (j >= desc.size() || k > T_END || !desc[j][k].present || (j + 1 == desc.size() && k == type)) throw Err("CRAM: name tokeniser: duplicate of a stream that does not exist");
            st.buf = desc[j][k].buf; st.present = true; st.pos = 0;
            continue;
        }
        const uint32_t clen = Nx16::u7(c);
        if (c.left() < clen) throw Err("CRAM: name tokeniser: truncated stream");
        if (use_arith) throw Err("CRAM 3.1 block codec `adaptive arithmetic coder` (inside the name tokeniser) is not supported by this reader (write CRAM 3.0, or convert with `samtools fastq`)");
        Cursor sub(c.p, clen); c.skip(clen);
        st.buf = Nx16::decode(sub); st.present = true; st.pos = 0;
    }
    auto byte_of = [&](size_t t, unsigned type) -> unsigned {
        if (t >= desc.size()) throw Err("CRAM: name tokeniser: token position without streams");
        Stream &s = desc[t][type];
        if (!s.present || s.pos >= s.buf.size()) throw Err("CRAM: name tokeniser: stream exhausted");
        return (unsigned char)s.buf[s.pos++];
    };
    auto u32_of = [&](size_t t, unsigned type) -> uint32_t { uint32_t v = 0; for (int i = 0; i < 4; ++i) v |= (uint32_t)byte_of(t, type) << (8 * i); return v; };
    struct Tok { unsigned type = T_NOP; uint32_t val = 0; std::string str; };
    std::vector<std::vector<Tok>> toks(nreads);
    std::vector<std::string> names(nreads);
    auto fixed = [](uint32_t v, size_t w) { std::string d = std::to_string(v); if (d.size() < w) d.insert(0, w - d.size(), '0'); return d; };
    std::string out; out.reserve(ulen);
    for (uint32_t cnum = 0; cnum < nreads; ++cnum) {
        const unsigned t0 = byte_of(0, T_TYPE);
        if (t0 != T_DUP && t0 != T_DIFF) throw Err("CRAM: name tokeniser: a name starts with neither DUP nor DIFF");
        const uint32_t dist = u32_of(0, t0);
        if (dist > cnum) throw Err("CRAM: name tokeniser: reference to a name that does not exist");
        const uint32_t pnum = cnum - dist;
        if (t0 == T_DUP) {
            if (pnum == cnum) throw Err("CRAM: name tokeniser: a name is a duplicate of itself");
            names[cnum] = names[pnum]; toks[cnum] = toks[pnum];
        } else {
            std::vector<Tok> cur(1);
            std::string &name = names[cnum];
            const std::vector<Tok> *prev = pnum != cnum ? &toks[pnum] : nullptr;
            auto prev_tok = [&](size_t t) -> const Tok & { if (!prev || t >= prev->size()) throw Err("CRAM: name tokeniser: token compared with nothing"); return (*prev)[t]; };
            for (size_t t = 1;; ++t) {
                if (t >= 128) throw Err("CRAM: name tokeniser: too many tokens in a name");
                const unsigned ty = byte_of(t, T_TYPE);
                Tok k; k.type = ty;
                if (ty == T_END) { cur.push_back(k); break; }
                switch (ty) {
                case T_CHAR: k.str.assign(1, (char)byte_of(t, T_CHAR)); break;
                case T_ALPHA: for (;;) { const unsigned b = byte_of(t, T_ALPHA); if (!b) break; k.str.push_back((char)b); } break;
                case T_DIGITS: k.val = u32_of(t, T_DIGITS); k.str = std::to_string(k.val); break;
                case T_DIGITS0: { k.val = u32_of(t, T_DIGITS0); const unsigned w = byte_of(t, T_DZLEN); k.str = fixed(k.val, w); } break;
                case T_DELTA: { const Tok &p = prev_tok(t); if (p.type != T_DIGITS && p.type != T_DIGITS0) throw Err("CRAM: name tokeniser: delta to a token that is no number");
                                k.val = p.val + byte_of(t, T_DELTA); k.str = std::to_string(k.val); k.type = T_DIGITS; } break;
                case T_DELTA0: { const Tok &p = prev_tok(t); if (p.type != T_DIGITS && p.type != T_DIGITS0) throw Err("CRAM: name tokeniser: delta to a token that is no number");
                                 k.val = p.val + byte_of(t, T_DELTA0); k.str = fixed(k.val, p.str.size()); k.type = T_DIGITS0; } break;
                case T_MATCH: k = prev_tok(t); if (k.type == T_END) throw Err("CRAM: name tokeniser: match with the end of a name"); break;
                case T_NOP: break;
                default: throw Err("CRAM: name tokeniser: token type " + std::to_string(ty) + " inside a name");
                }
                name += k.str;
                if (name.size() > ulen) throw Err("CRAM: name tokeniser: names exceed the stated size");
                cur.push_back(std::move(k));
            }
            toks[cnum] = std::move(cur);
        }
        out += names[cnum]; out.push_back('\0');
        if (out.size() > ulen) throw Err("CRAM: name tokeniser: names exceed the stated size");
    }
    if (out.size() != ulen) throw Err("CRAM: name tokeniser: size mismatch");
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
        case 5: { Cursor c(data, size); bytes = Nx16::decode(c, raw_size >= 0 ? (size_t)raw_size : (size_t)-1); } break;
        case 6: throw Err("CRAM 3.1 block codec `adaptive arithmetic coder` is not supported by this reader (write CRAM 3.0, or convert with `samtools fastq`)");
        case 7: throw Err("CRAM 3.1 block codec fqzcomp is not supported by this reader (write CRAM 3.0, or convert with `samtools fastq`)");
        case 8: bytes = tok3_decode(data, size); break;
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
