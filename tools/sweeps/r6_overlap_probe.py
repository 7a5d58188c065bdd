#!/usr/bin/env python
"""Does an index part's sketch (VALU-bound) overlap with another part's sort (HBM-bound) when they run on two streams?  Two contexts, two
sets of one index part's size at H. sapiens scale (C5 targets), built one after the other and then from two threads, the second started
`--lag` ms behind the first so that its sketch meets the first one's sort.  A measurement for DESIGN.md section 9, not a product path."""
import argparse
import json
import sys
import os
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=600000)
    ap.add_argument("--lag", type=float, default=45.0)
    ap.add_argument("--rounds", type=int, default=3)
    a = ap.parse_args()
    import torch
    from lrge_amd import engine, synth_cb, _ffi
    spec, Q, T = synth_cb.spec_of("c5_human_twoset", 1.0)
    ctxs = [engine.Context(0), engine.Context(0)]
    devs = [spec.device_reads(Q + i * a.reads, a.reads) for i in range(2)]
    sets = [engine.SeqSet(c, d.ptr, d.offsets, d.name_ranks()) for c, d in zip(ctxs, devs)]

    def build(i, lag_ms=0.0, out=None):
        if lag_ms:
            time.sleep(lag_ms / 1e3)
        t0 = time.perf_counter()
        ix = engine.Index(ctxs[i], sets[i], preset=_ffi.PRESET_AVA_PB)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        ix.free()
        if out is not None:
            out[i] = dt
        return dt
    build(0); build(1)                       # warm-up (arena growth, first-use costs)
    res = {"reads_per_set": a.reads, "lag_ms": a.lag, "sequential_ms": [], "concurrent_wall_ms": [], "concurrent_each_ms": []}
    for _ in range(a.rounds):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); d0 = build(0); d1 = build(1); seq = (time.perf_counter() - t0) * 1e3
        res["sequential_ms"].append([round(d0, 1), round(d1, 1), round(seq, 1)])
        out = {}
        th = [threading.Thread(target=build, args=(i, a.lag * i, out)) for i in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        res["concurrent_wall_ms"].append(round((time.perf_counter() - t0) * 1e3, 1))
        res["concurrent_each_ms"].append([round(out[0], 1), round(out[1], 1)])
    print(json.dumps(res))


if __name__ == "__main__":
    main()
