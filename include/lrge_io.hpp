// lrge_io.hpp -- input records for the C++ host side: the formats liblrge/src/io.rs accepts.
//
//   detect_compression_format   io.rs:36-63   magic bytes: gzip 1f 8b, bzip2 42 5a, zstd 28 b5 2f fd, xz fd 37 7a 58 5a
//   SeqReader::new              io.rs:71-121  decompress, then sniff "BAM\1" / "CRAM" / "@HD" / "@SQ" / "@RG" -> alignment
//                                             input, everything else -> FASTA / FASTQ
//   count_records, iter_records io.rs:123-184 callback(read id, sequence); mapped alignment records are refused with
//                                             "Mapped records are not supported. Only unaligned BAM/CRAM/SAM is allowed."
//   read_id                     io.rs:195-205 header up to the first ASCII whitespace
//
// Host-side and I/O-bound; nothing here touches the device.  The file is read and decompressed into memory in one
// piece (the strategies keep every sampled read in memory anyway).  gzip (incl. BGZF and multi-member) goes through
// zlib; bzip2 / xz / zstd through the system's shared libraries, bound at run time because this image ships them
// without headers (libbz2.so.1, liblzma.so.5, libzstd.so.1) -- a missing library is an error naming it, never a
// silent fallback.  Unaligned CRAM 3.0 (round 6): lrge_cram.hpp -- containers, every encoding of the specification, raw / gzip / bzip2 /
// lzma / rANS 4x8 blocks, and of CRAM 3.1 rANS Nx16 + the name tokeniser; the arithmetic coder and fqzcomp are an error naming the codec.
#pragma once
#include <dlfcn.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace lrge {
namespace io {

struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

enum class CompressionFormat { None, Gzip, Bzip2, Xz, Zstd };

inline CompressionFormat detect_compression_format(const std::string &d) {   // io.rs:36-63
    auto b = [&](size_t i) { return i < d.size() ? (unsigned char)d[i] : 0x100u; };
    if (b(0) == 0x1f && b(1) == 0x8b) return CompressionFormat::Gzip;
    if (b(0) == 0x42 && b(1) == 0x5a) return CompressionFormat::Bzip2;
    if (b(0) == 0x28 && b(1) == 0xb5 && b(2) == 0x2f && b(3) == 0xfd) return CompressionFormat::Zstd;
    if (b(0) == 0xfd && b(1) == 0x37 && b(2) == 0x7a && b(3) == 0x58 && b(4) == 0x5a) return CompressionFormat::Xz;
    return CompressionFormat::None;
}

inline std::string slurp(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw IoError("cannot open " + path);
    std::string d;
    char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) d.append(buf, n);
    fclose(f);
    return d;
}

inline std::string gunzip_all(const std::string &d) {   // MultiGzDecoder: every member, back to back
    std::string out;
    z_stream z;
    std::memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 16) != Z_OK) throw IoError("zlib: inflateInit2 failed");
    z.next_in = (Bytef *)d.data();
    z.avail_in = (uInt)0;
    size_t pos = 0;
    std::vector<char> buf(1 << 20);
    for (;;) {
        if (z.avail_in == 0 && pos < d.size()) {
            size_t take = std::min<size_t>(d.size() - pos, 1u << 30);
            z.next_in = (Bytef *)d.data() + pos; z.avail_in = (uInt)take; pos += take;
        }
        z.next_out = (Bytef *)buf.data(); z.avail_out = (uInt)buf.size();
        int rc = inflate(&z, Z_NO_FLUSH);
        out.append(buf.data(), buf.size() - z.avail_out);
        if (rc == Z_STREAM_END) {
            if (z.avail_in == 0 && pos >= d.size()) break;
            if (inflateReset(&z) != Z_OK) { inflateEnd(&z); throw IoError("zlib: inflateReset failed"); }
        } else if (rc != Z_OK) {
            inflateEnd(&z);
            throw IoError(std::string("gzip: corrupt input (") + (z.msg ? z.msg : "truncated") + ")");
        } else if (z.avail_in == 0 && pos >= d.size() && z.avail_out != 0) {
            inflateEnd(&z);
            throw IoError("gzip: unexpected end of file");
        }
    }
    inflateEnd(&z);
    return out;
}

namespace detail {
inline void *open_lib(const char *soname) {
    void *h = dlopen(soname, RTLD_NOW | RTLD_LOCAL);
    if (!h) throw IoError(std::string("cannot load ") + soname + " for this input's compression format");
    return h;
}
template <class F> F sym(void *h, const char *name) {
    void *p = dlsym(h, name);
    if (!p) throw IoError(std::string("symbol not found: ") + name);
    return reinterpret_cast<F>(p);
}
// bzlib.h's bz_stream (stable ABI of libbz2 1.0)
struct BzStream {
    char *next_in; unsigned avail_in, total_in_lo32, total_in_hi32;
    char *next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
    void *state; void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
};
// lzma/base.h's lzma_stream (stable ABI of liblzma 5)
struct LzmaStream {
    const uint8_t *next_in; size_t avail_in; uint64_t total_in;
    uint8_t *next_out; size_t avail_out; uint64_t total_out;
    const void *allocator; void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2; size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};
struct ZstdIn { const void *src; size_t size, pos; };
struct ZstdOut { void *dst; size_t size, pos; };
}  // namespace detail

inline std::string bunzip2_all(const std::string &d) {   // BzDecoder: one stream (bufread::BzDecoder stops after the first)
    using namespace detail;
    void *h = open_lib("libbz2.so.1");
    auto init = sym<int (*)(BzStream *, int, int)>(h, "BZ2_bzDecompressInit");
    auto run = sym<int (*)(BzStream *)>(h, "BZ2_bzDecompress");
    auto end = sym<int (*)(BzStream *)>(h, "BZ2_bzDecompressEnd");
    BzStream s;
    std::memset(&s, 0, sizeof s);
    if (init(&s, 0, 0) != 0) throw IoError("bzip2: init failed");
    std::string out;
    std::vector<char> buf(1 << 20);
    size_t pos = 0;
    for (;;) {
        if (s.avail_in == 0 && pos < d.size()) {
            size_t take = std::min<size_t>(d.size() - pos, 1u << 30);
            s.next_in = (char *)d.data() + pos; s.avail_in = (unsigned)take; pos += take;
        }
        s.next_out = buf.data(); s.avail_out = (unsigned)buf.size();
        int rc = run(&s);
        out.append(buf.data(), buf.size() - s.avail_out);
        if (rc == 4 /* BZ_STREAM_END */) break;
        if (rc != 0 || (s.avail_in == 0 && pos >= d.size() && s.avail_out != 0)) { end(&s); throw IoError("bzip2: corrupt or truncated input"); }
    }
    end(&s);
    return out;
}

inline std::string unxz_all(const std::string &d) {
    using namespace detail;
    void *h = open_lib("liblzma.so.5");
    auto init = sym<int (*)(LzmaStream *, uint64_t, uint32_t)>(h, "lzma_stream_decoder");
    auto code = sym<int (*)(LzmaStream *, int)>(h, "lzma_code");
    auto end = sym<void (*)(LzmaStream *)>(h, "lzma_end");
    LzmaStream s;
    std::memset(&s, 0, sizeof s);
    if (init(&s, UINT64_MAX, 0x08 /* LZMA_CONCATENATED */) != 0) throw IoError("xz: init failed");
    s.next_in = (const uint8_t *)d.data(); s.avail_in = d.size();
    std::string out;
    std::vector<uint8_t> buf(1 << 20);
    for (;;) {
        s.next_out = buf.data(); s.avail_out = buf.size();
        int rc = code(&s, s.avail_in == 0 ? 3 /* LZMA_FINISH */ : 0 /* LZMA_RUN */);
        out.append((const char *)buf.data(), buf.size() - s.avail_out);
        if (rc == 1 /* LZMA_STREAM_END */) break;
        if (rc != 0) { end(&s); throw IoError("xz: corrupt or truncated input"); }
    }
    end(&s);
    return out;
}

inline std::string unzstd_all(const std::string &d) {
    using namespace detail;
    void *h = open_lib("libzstd.so.1");
    auto create = sym<void *(*)()>(h, "ZSTD_createDStream");
    auto init = sym<size_t (*)(void *)>(h, "ZSTD_initDStream");
    auto run = sym<size_t (*)(void *, ZstdOut *, ZstdIn *)>(h, "ZSTD_decompressStream");
    auto is_err = sym<unsigned (*)(size_t)>(h, "ZSTD_isError");
    auto free_ = sym<size_t (*)(void *)>(h, "ZSTD_freeDStream");
    void *ds = create();
    if (!ds || is_err(init(ds))) throw IoError("zstd: init failed");
    std::string out;
    std::vector<char> buf(1 << 20);
    ZstdIn in{d.data(), d.size(), 0};
    size_t last = 0;
    while (in.pos < in.size || last != 0) {
        ZstdOut o{buf.data(), buf.size(), 0};
        size_t before = in.pos;
        last = run(ds, &o, &in);
        if (is_err(last)) { free_(ds); throw IoError("zstd: corrupt input"); }
        out.append(buf.data(), o.pos);
        if (in.pos == before && o.pos == 0) {
            if (last != 0) { free_(ds); throw IoError("zstd: unexpected end of file"); }
            break;
        }
    }
    free_(ds);
    return out;
}

inline std::string decompress(const std::string &raw) {
    switch (detect_compression_format(raw)) {
    case CompressionFormat::Gzip: return gunzip_all(raw);
    case CompressionFormat::Bzip2: return bunzip2_all(raw);
    case CompressionFormat::Xz: return unxz_all(raw);
    case CompressionFormat::Zstd: return unzstd_all(raw);
    default: return raw;
    }
}

inline std::string read_id(const std::string &h) {   // io.rs:195-205
    size_t i = 0;
    while (i < h.size() && !(h[i] == ' ' || h[i] == '\t' || h[i] == '\n' || h[i] == '\r' || h[i] == '\v' || h[i] == '\f')) ++i;
    return h.substr(0, i);
}

enum class Kind { Fastx, Sam, Bam, Cram };

inline Kind sniff(const std::string &d) {   // io.rs:88-98
    auto starts = [&](const char *m, size_t n) { return d.size() >= n && std::memcmp(d.data(), m, n) == 0; };
    if (starts("BAM\x01", 4)) return Kind::Bam;
    if (starts("CRAM", 4)) return Kind::Cram;
    if (starts("@HD", 3) || starts("@SQ", 3) || starts("@RG", 3)) return Kind::Sam;
    return Kind::Fastx;
}

using Callback = std::function<void(const std::string &, const std::string &)>;
static const char *const MAPPED_MSG = "Mapped records are not supported. Only unaligned BAM/CRAM/SAM is allowed.";

namespace detail {
struct Lines {
    const std::string &d;
    size_t pos = 0;
    explicit Lines(const std::string &s) : d(s) {}
    bool next(std::string &out) {
        if (pos >= d.size()) return false;
        size_t e = d.find('\n', pos);
        size_t stop = e == std::string::npos ? d.size() : e;
        out.assign(d, pos, stop - pos);
        if (!out.empty() && out.back() == '\r') out.pop_back();
        pos = e == std::string::npos ? d.size() : e + 1;
        return true;
    }
};

inline void parse_fastx(const std::string &d, const Callback &cb) {
    Lines L(d);
    std::string line, seq;
    do { if (!L.next(line)) return; } while (line.empty());
    if (line[0] == '>') {
        std::string name = read_id(line.substr(1));
        while (L.next(line)) {
            if (!line.empty() && line[0] == '>') { cb(name, seq); seq.clear(); name = read_id(line.substr(1)); }
            else seq += line;
        }
        cb(name, seq);
    } else if (line[0] == '@') {
        for (;;) {
            std::string name = read_id(line.substr(1)), s, plus, qual;
            if (!L.next(s) || !L.next(plus) || !L.next(qual)) throw IoError("truncated FASTQ record");
            if (plus.empty() || plus[0] != '+') throw IoError("malformed FASTQ record: " + name);
            cb(name, s);
            do { if (!L.next(line)) return; } while (line.empty());
            if (line[0] != '@') throw IoError("malformed FASTQ record after " + name);
        }
    } else {
        throw IoError("unrecognised sequence file");
    }
}

inline void parse_sam(const std::string &d, const Callback &cb) {
    Lines L(d);
    std::string line;
    while (L.next(line)) {
        if (line.empty() || line[0] == '@') continue;
        std::vector<std::string> f;
        size_t p = 0;
        while (f.size() < 11) {
            size_t t = line.find('\t', p);
            f.push_back(line.substr(p, t == std::string::npos ? std::string::npos : t - p));
            if (t == std::string::npos) break;
            p = t + 1;
        }
        if (f.size() < 11) throw IoError("invalid SAM record: fewer than 11 fields");
        char *endp = nullptr;
        unsigned long flag = strtoul(f[1].c_str(), &endp, 10);
        if (endp == f[1].c_str() || *endp) throw IoError("invalid SAM record: bad flag field");
        if (!(flag & 4)) throw IoError(MAPPED_MSG);
        cb(f[0] == "*" ? std::string() : f[0], f[9] == "*" ? std::string() : f[9]);
    }
}

inline void parse_bam(const std::string &d, const Callback &cb) {
    static const char NT16[] = "=ACMGRSVTWYHKDBN";
    auto need = [&](size_t off, size_t n) { if (off + n > d.size()) throw IoError("truncated BAM file"); };
    auto i32 = [&](size_t off) { need(off, 4); int32_t v; std::memcpy(&v, d.data() + off, 4); return v; };
    size_t off = 4;
    int32_t l_text = i32(off); off += 4;
    if (l_text < 0) throw IoError("invalid BAM header");
    need(off, (size_t)l_text); off += (size_t)l_text;
    int32_t n_ref = i32(off); off += 4;
    for (int32_t r = 0; r < n_ref; ++r) {
        int32_t l_name = i32(off); off += 4;
        if (l_name < 0) throw IoError("invalid BAM header");
        need(off, (size_t)l_name + 4); off += (size_t)l_name + 4;
    }
    std::string name, seq;
    while (off < d.size()) {
        int32_t block = i32(off); off += 4;
        if (block < 32) throw IoError("invalid BAM record");
        need(off, (size_t)block);
        const unsigned char *r = (const unsigned char *)d.data() + off;
        unsigned l_read_name = r[8];
        uint16_t n_cigar, flag; int32_t l_seq;
        std::memcpy(&n_cigar, r + 12, 2); std::memcpy(&flag, r + 14, 2); std::memcpy(&l_seq, r + 16, 4);
        if (l_seq < 0 || 32 + (size_t)l_read_name + 4 * (size_t)n_cigar + ((size_t)l_seq + 1) / 2 > (size_t)block) throw IoError("invalid BAM record");
        if (!(flag & 4)) throw IoError(MAPPED_MSG);
        name.assign((const char *)r + 32, l_read_name ? l_read_name - 1 : 0);   // NUL-terminated
        if (name == "*") name.clear();
        const unsigned char *s = r + 32 + l_read_name + 4 * (size_t)n_cigar;
        seq.resize((size_t)l_seq);
        for (int32_t i = 0; i < l_seq; ++i) seq[(size_t)i] = NT16[(s[i >> 1] >> ((~i & 1) << 2)) & 15];
        cb(name, seq);
        off += (size_t)block;
    }
}
}  // namespace detail

}  // namespace io
}  // namespace lrge
#include "lrge_cram.hpp"
namespace lrge {
namespace io {

inline void iter_records(const std::string &path, const Callback &cb) {   // io.rs:154-184
    std::string data = decompress(slurp(path));
    switch (sniff(data)) {
    case Kind::Bam: detail::parse_bam(data, cb); break;
    case Kind::Sam: detail::parse_sam(data, cb); break;
    case Kind::Cram: cram::parse(data, cb, MAPPED_MSG); break;
    default: detail::parse_fastx(data, cb);
    }
}

inline size_t count_records(const std::string &path) {   // io.rs:123-152
    size_t n = 0;
    iter_records(path, [&](const std::string &, const std::string &) { ++n; });
    if (n == 0) throw IoError("Is the file empty?");
    return n;
}

}  // namespace io
}  // namespace lrge
