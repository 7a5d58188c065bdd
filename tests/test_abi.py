"""CPU: the C-ABI library loads, exports every symbol include/lrge_hip.h declares, and fails
loudly (no CPU fallback) when no device is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, has_gpu


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "lrge_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lrge_hip_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported():
    from lrge_amd import _ffi
    L = _ffi.lib()
    syms = _declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(L, s), "missing export " + s
    assert set(syms) == set(_ffi.EXPORTS)
    assert b"gfx950" in L.lrge_hip_version()


def test_no_cpu_fallback_without_device():
    if has_gpu():
        pytest.skip("a GPU is present")
    from lrge_amd import engine
    with pytest.raises(Exception) as ei:
        engine.Context(0)
    assert "-5" in str(ei.value)    # LRGE_ERR_DEVICE


def test_host_median_matches_reference_kats():
    # lrge_hip_median is host code (estimate.rs:80-132): same KATs as the oracle
    from lrge_amd import engine
    inf = float("inf")
    assert engine.median(np.arange(1, 11, dtype=np.float32), False, 0.15, 0.65) == \
        (np.float32(2.35), np.float32(5.5), np.float32(6.85))
    d = np.array([1, 2, 3, 4, 5, 6, inf, inf, inf, inf], dtype=np.float32)
    assert engine.median(d, False, 0.15, 0.65) == (np.float32(2.35), np.float32(5.5), np.float32(inf))
    assert engine.median(np.array([1, 3, 5, 7], dtype=np.float32), False) == (None, np.float32(4.0), None)
    assert engine.median(np.zeros(0, dtype=np.float32)) == (None, None, None)
    assert engine.median(np.array([1, inf], dtype=np.float32), True) == (None, np.float32(1.0), None)
    with pytest.raises(Exception):
        engine.median(np.ones(3, dtype=np.float32), True, 1.1, None)   # "Quantile must be between 0.0 and 1.0"
    with pytest.raises(Exception):
        engine.median(np.ones(3, dtype=np.float32), True, None, 0.5)   # reference panics (estimate.rs:109)


def test_host_median_selection_equals_a_full_sort(oracle):
    """lrge_hip_median selects the order statistics it needs (bin histogram over the float patterns + selection inside the
    bins; comparison-based selection when negative values are present) where the reference sorts everything
    (estimate.rs:90-95): same result bit for bit, on ties, clusters, infinities, zeros and negatives."""
    from lrge_amd import engine
    rng = np.random.Generator(np.random.PCG64(11))
    cases = []
    for n in (1, 2, 3, 7, 64, 1000, 50_000):
        cases.append(rng.lognormal(18.0, 0.6, n).astype(np.float32))                      # what genome-size estimates look like
        cases.append(np.round(rng.lognormal(3.0, 0.3, n)).astype(np.float32))             # heavy ties
        c = rng.lognormal(18.0, 0.6, n).astype(np.float32); c[rng.random(n) < 0.3] = np.inf; cases.append(c)
        cases.append(np.full(n, 4.25, dtype=np.float32))                                  # one value
        c = rng.normal(0.0, 5.0, n).astype(np.float32); c[rng.random(n) < 0.1] = 0.0; cases.append(c)   # negatives, zeros
    for v in cases:
        for finite in (True, False):
            for lo, hi in ((0.15, 0.65), (0.0, 1.0), (None, None), (0.5, None)):
                got, exp = engine.median(v, finite, lo, hi), oracle.median(v, finite, lo, hi)
                # (inf * 0 in the interpolation gives NaN in the reference too: compare patterns, not values)
                bits = lambda t: tuple(None if x is None else np.float32(x).view(np.uint32) & (0xFFFFFFFF if x == x else 0x7F800000) for x in t)
                assert bits(got) == bits(exp), (v.size, finite, lo, hi, got, exp)


def test_name_ranks_follow_strcmp():
    from lrge_amd import engine
    a, b = engine.name_ranks([b"r2", b"r10", b"R1"], [b"r10", b"a"])
    # byte order: "R1" < "a" < "r10" < "r2"
    assert a.tolist() == [4, 2, 0] and b.tolist() == [2, 1]


# ---- the Rust seam (integration/liblrge_hip_shim.rs) against include/lrge_hip.h: the shim cannot be compiled here (no Rust toolchain in the
# image), so at least its `extern "C"` block and its #[repr(C)] structs are held to the header by a parser: same symbols, same arity, the same
# integer / float widths and pointer-ness per argument and return value, the same struct fields in the same order ----
_C_KIND = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "float": "f32", "double": "f64", "void": "void",
           "char": "i8"}
_RS_KIND = {"c_int": "i32", "i32": "i32", "u32": "u32", "u64": "u64", "usize": "usize", "f32": "f32", "f64": "f64", "c_char": "i8", "c_void": "void"}


def _c_decls():
    txt = open(os.path.join(ROOT, "include", "lrge_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    out = {}
    for m in re.finditer(r"\b((?:const\s+)?[A-Za-z_0-9]+\s*\**)\s*(lrge_hip_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)

        def kind(t):
            t = re.sub(r"\bconst\b", " ", t).strip()
            t = re.sub(r"\[[^\]]*\]", "*", t)                      # array parameter = pointer
            if "(" in t:
                return "ptr"                                       # function pointer
            if "*" in t:
                return "ptr"
            base = t.split()[0] if t.split() else "void"
            return _C_KIND.get(base, "ptr" if base.endswith("_fn") else base)
        arg_kinds = []
        if args.strip() and args.strip() != "void":
            depth, cur, parts = 0, "", []
            for ch in args:
                if ch == "(":
                    depth += 1
                if ch == ")":
                    depth -= 1
                if ch == "," and depth == 0:
                    parts.append(cur); cur = ""
                else:
                    cur += ch
            parts.append(cur)
            for a in parts:
                a = a.strip()
                # drop the parameter name (the last identifier) unless the declaration is just a type
                a2 = re.sub(r"\b[A-Za-z_][A-Za-z_0-9]*\s*(\[[^\]]*\])?$", lambda mm: (mm.group(1) or ""), a).strip() or a
                arg_kinds.append(kind(a2 if a2 else a))
        out[name] = (kind(ret), arg_kinds)
    return out


def _rs_decls():
    txt = open(os.path.join(ROOT, "integration", "liblrge_hip_shim.rs")).read()
    blk = re.search(r'extern "C" \{(.*?)\n\}', txt, flags=re.S).group(1)
    blk = re.sub(r"//[^\n]*", "", blk)
    out = {}
    for m in re.finditer(r"fn\s+(lrge_hip_[a-z_0-9]+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", blk, flags=re.S):
        name, args, ret = m.group(1), m.group(2), (m.group(3) or "void").strip()

        def kind(t):
            t = t.strip()
            if t.startswith("*"):
                return "ptr"
            return _RS_KIND.get(t, t)
        arg_kinds = [kind(a.split(":", 1)[1]) for a in args.split(",") if ":" in a]
        out[name] = (kind(ret), arg_kinds)
    return out, txt


def test_rust_shim_extern_block_matches_the_header():
    c = _c_decls()
    rs, txt = _rs_decls()
    assert len(rs) >= 25 and {"lrge_hip_chains", "lrge_hip_paf_stats", "lrge_hip_overlap_twoset", "lrge_hip_index_build_tsharded"} <= set(rs)
    for name, (ret, args) in rs.items():
        assert name in c, "the shim binds %s, which include/lrge_hip.h does not declare" % name
        cret, cargs = c[name]
        assert ret == cret, (name, "return", ret, cret)
        assert args == cargs, (name, "arguments", args, cargs)
    # the #[repr(C)] structs: field names, order and widths
    hdr = open(os.path.join(ROOT, "include", "lrge_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for sname in ("lrge_hip_params", "lrge_hip_chain"):
        cm = re.search(r"typedef struct \{([^}]*)\}\s*%s;" % sname, hdr, flags=re.S).group(1)
        cf = []
        for line in cm.split(";"):
            line = line.strip()
            if not line:
                continue
            ty, names = line.split(None, 1)
            cf += [(n.strip(), _C_KIND[ty]) for n in names.split(",")]
        rm = re.search(r"pub struct %s \{([^}]*)\}" % sname, txt, flags=re.S).group(1)
        rf = [(n, _RS_KIND[t]) for n, t in re.findall(r"pub\s+([a-z_0-9]+):\s*([a-z0-9_]+)", rm)]
        assert rf == cf, (sname, rf, cf)
    # the error codes the shim maps are the header's
    codes = dict(re.findall(r"#define\s+(LRGE_ERR_[A-Z_]+)\s+\(?(-?\d+)\)?", open(os.path.join(ROOT, "include", "lrge_hip.h")).read()))
    for code, variant in (("LRGE_ERR_IO", "IoError"), ("LRGE_ERR_PARSE", "FastqParseError"), ("LRGE_ERR_TOO_MANY", "TooManyReadsError"), ("LRGE_ERR_TOO_FEW", "TooFewReadsError"),
                          ("LRGE_ERR_MAP", "MapError"), ("LRGE_ERR_DUPLICATE_ID", "DuplicateReadIdentifier"), ("LRGE_ERR_PAF_WRITE", "PafWriteError")):
        assert code in codes, code
        assert re.search(r"%s\s*=>\s*LrgeError::%s" % (re.escape(codes[code]), variant), txt), (code, codes[code], variant)


def test_rust_shim_writes_overlaps_paf_where_the_reference_does():
    txt = open(os.path.join(ROOT, "integration", "liblrge_hip_shim.rs")).read()
    assert txt.count("write_overlaps_paf(job.tmpdir") == 3                 # forward, inverse, all-vs-all
    assert 'tmpdir.join("overlaps.paf")' in txt and "tp:A:S" in txt and "f32::EPSILON" in txt
