#!/usr/bin/env python
"""Damage campaign against the host-side input readers (include/lrge_io.hpp, include/lrge_cram.hpp) through `lrge-hip --dump-records`, a
host-only mode: seeded damage -- overwritten bytes, one flipped bit, 0x7fffffff where a length may stand, truncation -- to CRAM 3.0 files of
every block method and both layouts and CRAM 3.1 files of every rANS Nx16 form (tests/cram_writer.py), gzip FASTQ, FASTQ and SAM.  A case passes when the process ends by itself
within the time limit with exit code 0 or 1 (records, or an error message); a signal, another code or a timeout is reported and the file
kept under --keep.  tests/test_input_formats.py runs a bounded sample of the same generator; the two findings of the first campaigns
(a rANS size field the stream cannot back; a rANS stream that ends early) are regression tests there.

  python tools/fuzz_readers.py --seeds 1-10 --cases 80
"""
import argparse
import gzip
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1-4", help="a-b or a comma list")
    ap.add_argument("--cases", type=int, default=40, help="damaged copies per seed file and seed")
    ap.add_argument("--limit", type=float, default=8.0, help="seconds a case may take")
    ap.add_argument("--keep", default="/tmp/lrge_fuzz")
    a = ap.parse_args()
    import cram_writer
    import test_input_formats as T
    from lrge_amd import build
    build.build_lib()
    cli = build.build_cli()
    reads = T._cram_reads(23)
    files = {"cram_%s_%s" % (m, v): cram_writer.write_cram(reads, variant=v, method=m, records_per_slice=10, slices_per_container=2)
             for m in ("raw", "gzip", "bzip2", "lzma", "rans0", "rans1") for v in ("external", "core")}
    for form in sorted(cram_writer.NX16_FORMS):                       # CRAM 3.1: rANS Nx16 in every form, names through the name tokeniser
        files["cram31_" + form] = cram_writer.write_cram(reads, minor=1, method=form, method_for={"RN": "tok3"}, records_per_slice=10, slices_per_container=2)
    files["fq"], files["sam"], files["fq.gz"] = T._fastq(), T._sam(), gzip.compress(T._fastq())
    seeds = list(range(int(a.seeds.split("-")[0]), int(a.seeds.split("-")[1]) + 1)) if "-" in a.seeds else [int(x) for x in a.seeds.split(",")]
    os.makedirs(a.keep, exist_ok=True)
    bad = n = 0
    for seed in seeds:
        rng = random.Random(seed)
        for name, data in sorted(files.items()):
            for it in range(a.cases):
                b = bytearray(data)
                kind = rng.randrange(4)
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
                p = os.path.join(a.keep, "s%d_%s_%d" % (seed, name, it))
                with open(p, "wb") as f:
                    f.write(bytes(b))
                n += 1
                try:
                    rc = subprocess.run([cli, "--dump-records", p], capture_output=True, timeout=a.limit).returncode
                except subprocess.TimeoutExpired:
                    rc = "timeout"
                if rc in (0, 1):
                    os.remove(p)
                else:
                    bad += 1
                    print("BAD", p, "kind", kind, "->", rc, flush=True)
    print("%d cases, %d bad" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
