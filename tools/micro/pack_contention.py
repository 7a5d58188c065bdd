#!/usr/bin/env python
"""Host-side pack under contention: N contexts of ONE process -- the ranks of a world emulated on one GPU, or of a
local communicator -- each upload their 1/N of a read set from pinned host memory (host-side 2-bit pack + PCIe), all at once, with
no GPU work beside them; against one context uploading a 1/N share alone, and against one context uploading the whole set.

  python tools/micro/pack_contention.py [--gbases 31.5] [--world 8] [--repeat 3]

What it separates: the host's capacity to pack (the CPUs the box GRANTS the process: cgroup cpu.max -- 16 on the GPU boxes of this
pool, whatever the 256 hardware threads say) from anything the emulation adds.  Since round 5 the contexts of a process share ONE
pack pool sized by that grant (csrc/host_pack.h: hp_shared_pool), so N contexts take turns chunk by chunk instead of starting N x 32
threads.  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gbases", type=float, default=31.5)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--repeat", type=int, default=3)
    ap.add_argument("--read-len", type=int, default=15000)
    a = ap.parse_args()
    from lrge_amd import engine
    from oracle import oracle as O       # (host_cpus only: what the cgroup grants)
    N, L = a.world, a.read_len
    n_share = int(a.gbases * 1e9 / N / L)
    rng = np.random.Generator(np.random.PCG64(7))
    block = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=64 << 20, dtype=np.uint8)]
    offs = np.arange(n_share + 1, dtype=np.uint64) * np.uint64(L)
    nbytes = n_share * L

    def fill(buf):
        for o in range(0, nbytes, block.size):
            m = min(block.size, nbytes - o)
            buf.array[o:o + m] = block[:m]

    ctxs = [engine.Context(0) for _ in range(N)]
    pins = []
    for c in ctxs:
        p = c.host_alloc(nbytes); fill(p); pins.append(p)

    def one_upload(i):
        S = ctxs[i].upload(pins[i], offs, None, wait=True)
        S.free()

    def timed(idx):
        best = None
        for _ in range(a.repeat + 1):
            th = [threading.Thread(target=one_upload, args=(i,)) for i in idx]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        return best
    alone = timed([0])
    together = timed(list(range(N)))
    # one context, the whole set (N shares back to back)
    t0 = time.perf_counter()
    for _ in range(N):
        one_upload(0)
    whole = (time.perf_counter() - t0) * 1e3
    out = {"what": "host-side pack + PCIe upload from pinned host memory, no GPU work beside it", "world": N, "share_gbases": nbytes / 1e9,
           "hw_threads": os.cpu_count(), "cpus_granted": O.host_cpus(),
           "one_context_one_share_ms": round(alone, 1), "all_contexts_one_share_each_ms": round(together, 1),
           "one_context_all_shares_back_to_back_ms": round(whole, 1),
           "pack_GBps_alone": round(nbytes / 1e6 / alone, 1), "pack_GBps_together": round(N * nbytes / 1e6 / together, 1),
           "reading": "together / alone = %.2f: N ranks' packs cost N single packs' worth of the granted CPUs -- the host clock of an N-rank world on THIS "
                      "host is bounded below by all_contexts_one_share_each_ms, whatever the GPUs do" % (together / alone)}
    print(json.dumps(out))
    for p in pins: p.free()
    for c in ctxs: c.close()


if __name__ == "__main__":
    main()
