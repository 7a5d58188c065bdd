#!/usr/bin/env python
"""Generates tests/golden/c5_full_index_stats.json: what the CPU oracle reports as n_minimizers / n_keys / mid_occ for the index
over the 2 000 000 target reads of the full-size H. sapiens-scale configuration (BASELINE configs[4], counter-based
generator lrge_amd.synth_cb "c5_human_twoset", preset ava-pb and ava-ont).  The reads come from the HOST twin of the
generator chunk by chunk and go through the oracle's mm_sketch (oracle.KeyStats: only the hashes are kept -- 60 GB of
host memory, a few minutes on a many-core box); tests/test_gpu_configs.py::test_c5_full compares the device's 8-part
index with it.  A checksum of a few reads ties the fixture to the generator it was made with.

  python tests/golden/make_c5_fixture.py [--config c5_human_twoset] [--presets pb,ont]
"""
import argparse
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def reads_checksum(spec, Q, T):
    idx = [0, Q - 1, Q, Q + T // 2, Q + T - 1]
    rb = spec.host_reads(idx=idx)
    return "%08x" % (zlib.crc32(rb.bases.tobytes()) & 0xFFFFFFFF)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5_human_twoset")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--presets", default="pb,ont")
    ap.add_argument("--chunk", type=int, default=100000)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from lrge_amd import synth_cb
    from oracle import oracle as O
    spec, Q, T = synth_cb.spec_of(a.config, a.scale)
    out = {"config": a.config, "scale": a.scale, "n_query": Q, "n_target": T, "generator": "lrge_amd.synth_cb (tools/synth/cb_core.h)",
           "reads_crc32": reads_checksum(spec, Q, T), "made_by": "tests/golden/make_c5_fixture.py (oracle.KeyStats over the host twin's reads)"}
    for pname in a.presets.split(","):
        preset = O.PRESET_AVA_PB if pname == "pb" else O.PRESET_AVA_ONT
        ks = O.KeyStats(O.make_opt(preset, dual=True))
        t0 = time.perf_counter()
        for lo in range(0, T, a.chunk):
            n = min(a.chunk, T - lo)
            rb = spec.host_reads(first=Q + lo, n=n)
            ks.add(rb.bases, rb.offsets)
            print("[fixture] %s: %d / %d target reads sketched (%.0f s)" % (pname, lo + n, T, time.perf_counter() - t0), file=sys.stderr, flush=True)
        st = ks.finish()
        st["oracle_s"] = round(time.perf_counter() - t0, 1)
        out["ava-" + pname] = st
        del ks
    path = a.out or os.path.join(ROOT, "tests", "golden", "c5_full_index_stats.json" if a.scale == 1.0 and a.config == "c5_human_twoset"
                                 else "%s_x%g_index_stats.json" % (a.config, a.scale))
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
