#!/usr/bin/env python
"""Full-size C5 (BASELINE configs[4]: H. sapiens-scale HiFi, -Q 100 000 -T 2 000 000), FORWARD strategy: EVERY query's count from the
GPU path against the CPU oracle's index of ALL targets, with the oracle's index build and map REALLY timed (VERDICT r04 item 2:
256 sampled counts and a pro-rated 1/40 CPU sample are what existed before).  TEST / MEASUREMENT INFRASTRUCTURE: the oracle is the
checker and the CPU baseline here, never part of the product path.

  python tools/c5_allcounts.py --preset pb --out gpurun_out/c5_allcounts_pb.json [--max-map-seconds 1800]

The oracle's index of 7.5-10.2 G minimizers takes ~0.4-0.6 TB of host memory while it is built (sketch-order copy, (hash, y) pairs,
position lists): the GPU boxes of this pool have 3 TB.  Queries are mapped in chunks dealt round-robin over the whole set, so that a
run cut short by --max-map-seconds has still covered every index part and anchor batch evenly; the file is rewritten after every
chunk.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T0 = time.perf_counter()


def log(*a):
    print("[c5_allcounts %7.1f s]" % (time.perf_counter() - T0), *a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5_human_twoset")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--preset", default="pb", choices=["ont", "pb"])
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--chunks", type=int, default=16, help="the queries are mapped in this many interleaved chunks")
    ap.add_argument("--max-map-seconds", type=float, default=3000.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from lrge_amd import engine, synth_cb
    from oracle import oracle as O
    preset = 1 if a.preset == "pb" else 0
    threads = a.threads or O.default_threads()
    spec, Q, T = synth_cb.spec_of(a.config, a.scale)
    out = {"what": "forward two-set counts of EVERY query read: GPU path vs the CPU oracle's index of ALL targets; oracle times are measured, nothing pro-rated",
           "config": a.config, "scale": a.scale, "preset": "ava-pb" if preset else "ava-ont", "n_query": Q, "n_target": T, "threads": threads,
           "host": {"hw_threads": os.cpu_count(), "cpus_granted": O.host_cpus()}}

    def save():
        if a.out:
            os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
            with open(a.out + ".tmp", "w") as f:
                json.dump(out, f, indent=1)
            os.replace(a.out + ".tmp", a.out)

    # ---- the GPU path ----
    dq, dt = spec.device_reads(0, Q), spec.device_reads(Q, T)
    out["query_bases"], out["target_bases"] = dq.total_bases, dt.total_bases
    t_bases = dt.to_host()                       # the oracle's targets: what the device twin wrote (bit-identical to the host twin: tests/test_synth_cb.py)
    q_bases = dq.to_host()
    t_offsets, q_offsets = dt.offsets.copy(), dq.offsets.copy()
    log("reads: %.2f + %.2f Gbases generated in HBM and copied to the host" % (dq.total_bases / 1e9, dt.total_bases / 1e9))
    h = spec.host_reads(first=Q + T - 200, n=200)         # the twins agree on this very set
    assert np.array_equal(h.bases, t_bases[int(t_offsets[T - 200]):])
    ctx = engine.Context(0)
    Qd = ctx.upload(dq.ptr, dq.offsets, dq.name_ranks())
    Td = ctx.upload(dt.ptr, dt.offsets, dt.name_ranks())
    dq.free(); dt.free()
    best = None
    for it in range(2):                          # (the first step grows the arena)
        t1 = time.perf_counter()
        ix = engine.Index(ctx, Td, preset)
        counts, has = ix.overlap_twoset(Qd)
        dt_s = time.perf_counter() - t1
        st = ix.stats()
        ix.free()
        best = dt_s if best is None else min(best, dt_s)
    out["gpu"] = {"step_s_resident_packed": round(best, 4), "mid_occ": st["mid_occ"], "n_minimizers": st["n_minimizers"], "n_keys": st["n_keys"],
                  "counts_sum": int(counts.astype(np.int64).sum()), "no_mapping": int((has == 0).sum())}
    log("gpu:", out["gpu"])
    Qd.free(); Td.free(); ctx.close()
    save()

    # ---- the oracle: index of ALL targets, timed ----
    opt = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
    tn = [b"r%08d" % i for i in range(Q, Q + T)]
    Ts = O.ReadSet.from_arrays(t_bases, t_offsets, tn)
    t1 = time.perf_counter()
    O.lib().lo_twoset_counts  # (library loaded before the clock starts)
    ixo = O.Index(Ts, opt)
    t_index = time.perf_counter() - t1
    out["oracle_index"] = {"seconds": round(t_index, 1), "mid_occ": ixo.mid_occ, "n_minimizers": ixo.n_minimizers, "n_keys": ixo.n_keys,
                           "equal_gpu": bool(ixo.mid_occ == st["mid_occ"] and ixo.n_minimizers == st["n_minimizers"] and ixo.n_keys == st["n_keys"])}
    log("oracle index:", out["oracle_index"])
    save()

    # ---- map every query, interleaved chunks ----
    qn = [b"r%08d" % i for i in range(Q)]
    done = np.zeros(Q, bool)
    t_map, n_diff, n_has_diff, chunks = 0.0, 0, 0, []
    first_diff = []
    for c in range(a.chunks):
        if t_map > a.max_map_seconds:
            break
        idx = np.arange(c, Q, a.chunks, dtype=np.int64)
        lens = (q_offsets[idx + 1] - q_offsets[idx]).astype(np.uint64)
        offs = np.zeros(len(idx) + 1, np.uint64); np.cumsum(lens, out=offs[1:])
        b = np.empty(int(offs[-1]), np.uint8)
        for j, i in enumerate(idx):
            b[int(offs[j]):int(offs[j + 1])] = q_bases[int(q_offsets[i]):int(q_offsets[i + 1])]
        Qs = O.ReadSet.from_arrays(b, offs, [qn[i] for i in idx])
        t1 = time.perf_counter()
        rc, ec, eh = ixo.twoset_counts(Qs, threads=threads)
        dt_c = time.perf_counter() - t1
        assert rc == 0
        t_map += dt_c
        d = ec != counts[idx]
        n_diff += int(d.sum()); n_has_diff += int((eh != has[idx]).sum())
        if d.any() and len(first_diff) < 16:
            first_diff += [{"query": int(idx[k]), "gpu": int(counts[idx[k]]), "oracle": int(ec[k])} for k in np.nonzero(d)[0][:16 - len(first_diff)]]
        done[idx] = True
        chunks.append(round(dt_c, 2))
        n_done = int(done.sum())
        out["oracle_map"] = {"reads_checked": n_done, "of": Q, "seconds": round(t_map, 1), "chunk_seconds": chunks, "counts_differ": n_diff,
                             "has_mapping_differ": n_has_diff, "counts_equal": n_diff == 0 and n_has_diff == 0, "first_differences": first_diff,
                             "overlaps_checked": int(counts[done].astype(np.int64).sum()),
                             "which": "queries c, c + %d, c + 2 x %d, ... for c = 0 .. %d" % (a.chunks, a.chunks, c)}
        if n_done:
            job_s = t_index + t_map * Q / n_done
            out["cpu_port_measured"] = {"reads_per_s": round(Q / job_s, 2), "job_seconds": round(job_s, 1), "index_seconds": round(t_index, 1),
                                        "map_seconds": round(t_map * Q / n_done, 1), "map_reads_per_s": round(n_done / t_map, 1), "threads": threads,
                                        "complete": n_done == Q,
                                        "note": "oracle = this repo's C restatement of the liblrge / minimap2-2.30 path (a port, not the reference binary), OpenMP, %d threads on the %.0f CPUs the host grants (%d hardware threads)" % (threads, O.host_cpus(), os.cpu_count() or 0)
                                                + ("" if n_done == Q else "; map time of %d reads scaled to %d" % (n_done, Q))}
            out["gpu_vs_cpu_port_measured"] = round(job_s / best, 1)
        log("chunk %d: %d reads in %.1f s; %d / %d checked, %d differ" % (c, len(idx), dt_c, n_done, Q, n_diff))
        save()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
