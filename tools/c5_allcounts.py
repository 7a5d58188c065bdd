#!/usr/bin/env python
"""Full-size C5 (BASELINE configs[4]: H. sapiens-scale HiFi, -Q 100 000 -T 2 000 000), FORWARD strategy: EVERY query's count from the
GPU path against the CPU oracle, with the oracle's work REALLY timed (256 sampled counts and a pro-rated 1/40 CPU
sample are what existed before).  TEST / MEASUREMENT INFRASTRUCTURE: the oracle is the checker and the CPU baseline here, never part
of the product path.

  python tools/c5_allcounts.py --preset pb --parts 8 --out gpurun_out/c5_allcounts_pb.json

The GPU boxes of this pool cannot hold the oracle's index of all 2 000 000 targets (7.5-10.2 G minimizers: ~0.4 TB while it is
built; a first attempt at it took the box down), so the oracle indexes the targets in P PARTS:
  pass 1  every part: lo_index_build with the whole set's mid_occ (tests/golden/c5_full_index_stats.json: the oracle's own KeyStats),
          the sketch-order copy dropped; the keys whose local count reaches mid_occ // P + 1 are the candidates for "too frequent over
          all targets" (a key above mid_occ overall reaches that in some part);
  then    the candidates' counts are summed over the parts -> the keys mm_idx_get answers "too frequent" for in the ONE index;
  pass 2  every part: those keys dropped (lo_index_drop_keys), all queries mapped; the parts hold disjoint targets, so the distinct-
          target counts add up and has_mapping ORs (twoset.rs:286-317) -- tests/test_oracle_restricted.py checks the procedure against
          the one index where that fits, and the partitioned / target-sharded GPU forms rest on the same argument.
All P part indexes are alive between the passes (stripped: ~14 bytes per entry); a guard refuses to start when the estimate exceeds
--mem-frac of what the cgroup grants.  What is timed: the part index builds (sketch + sort + run-length: the work of the one index
build, cut in P) and the mapping of all queries against every part (P x the per-query fixed work of the one index: an upper bound of
the one-index map time).
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


def mem_granted():
    """bytes the cgroup grants this process (memory.max), else MemAvailable"""
    for p in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
        try:
            v = open(p).read().strip()
            if v != "max" and int(v) < (1 << 60):
                return int(v), p
        except Exception:      # noqa: BLE001
            pass
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable"):
            return int(line.split()[1]) * 1024, "/proc/meminfo MemAvailable"
    return 64 << 30, "assumed"


def rss_gb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS"):
            return int(line.split()[1]) / 1e6
    return 0.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5_human_twoset")
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--preset", default="pb", choices=["ont", "pb"])
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--parts", type=int, default=8)
    ap.add_argument("--mem-frac", type=float, default=0.55, help="refuse to start when the estimated peak exceeds this share of the granted memory")
    ap.add_argument("--mem-cap-gb", type=float, default=0.0, help="treat the granted memory as at most this (a limit the cgroup files do not show)")
    ap.add_argument("--two-pass", default="auto", choices=["auto", "yes", "no"],
                    help="yes: no part index is kept between the passes -- pass 1 keeps every part's (key, count) table only (9 bytes per distinct key), "
                         "pass 2 builds each part again to map against it (the index work twice, ~100 GB of peak memory at any scale); auto: when the "
                         "all-parts-alive plan does not pass the memory guard")
    ap.add_argument("--max-map-seconds", type=float, default=3000.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from lrge_amd import engine, synth_cb
    from oracle import oracle as O
    from oracle import c5_sample
    preset = 1 if a.preset == "pb" else 0
    pname = "ava-pb" if preset else "ava-ont"
    threads = a.threads or O.default_threads()
    spec, Q, T = synth_cb.spec_of(a.config, a.scale)
    P = max(1, a.parts)
    granted, how = mem_granted()
    if a.mem_cap_gb > 0 and granted > a.mem_cap_gb * 1e9:
        granted, how = int(a.mem_cap_gb * 1e9), "--mem-cap-gb"
    out = {"what": "forward two-set counts of EVERY query read: GPU path vs the CPU oracle indexing the targets in %d parts (one index's answers: "
                   "global mid_occ, globally too-frequent keys dropped in every part, per-part distinct-target counts summed); oracle times measured" % P,
           "config": a.config, "scale": a.scale, "preset": pname, "n_query": Q, "n_target": T, "threads": threads, "parts": P,
           "host": {"hw_threads": os.cpu_count(), "cpus_granted": O.host_cpus(), "memory_granted_GB": round(granted / 1e9, 1), "memory_source": how}}

    def save():
        if a.out:
            os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
            with open(a.out + ".tmp", "w") as f:
                json.dump(out, f, indent=1)
            os.replace(a.out + ".tmp", a.out)

    # ---- memory guard (a box that runs out of memory is lost, and the round's GPU access with it) ----
    if a.scale == 1.0 and a.config == "c5_human_twoset":
        fx = c5_sample.fixture_stats(spec, Q, T, pname)
        n_mz_est, mid_occ_global = fx["n_minimizers"], fx["mid_occ"]
    else:
        fx = None
        n_mz_est, mid_occ_global = int(0.34 * 15000 * T), None
    part_mz = n_mz_est / P
    est_resident = P * part_mz * 14.0                      # stripped parts: pos 8 B + keys / offsets ~6 B per entry
    est_build = part_mz * 16.0 * 3.2                       # one build in flight: per-read vectors (with growth slack), the concatenation, the (hash, y) pairs
    est_reads = 1.05 * (15000.0 * T / P) + 1.6e9 * a.scale + 2e9
    est_peak = est_resident + est_build + est_reads
    out["memory_plan_GB"] = {"resident_parts": round(est_resident / 1e9, 1), "one_build": round(est_build / 1e9, 1), "reads": round(est_reads / 1e9, 1),
                             "peak": round(est_peak / 1e9, 1), "limit": round(a.mem_frac * granted / 1e9, 1)}
    log("memory plan:", out["memory_plan_GB"])
    two_pass = a.two_pass == "yes" or (a.two_pass == "auto" and est_peak > a.mem_frac * granted)
    if two_pass:
        est_tables = 0.6 * n_mz_est * 9.0 * 0.5 + P * 0.0          # (about one distinct key per two entries of a part, 9 bytes each)
        est_peak = est_tables + est_build + part_mz * 14.0 + est_reads
        out["memory_plan_GB"] = {"key_count_tables": round(est_tables / 1e9, 1), "one_build": round(est_build / 1e9, 1), "one_part": round(part_mz * 14.0 / 1e9, 1),
                                 "reads": round(est_reads / 1e9, 1), "peak": round(est_peak / 1e9, 1), "limit": round(a.mem_frac * granted / 1e9, 1), "two_pass": True}
        log("memory plan (two passes):", out["memory_plan_GB"])
    if est_peak > a.mem_frac * granted:
        out["refused"] = "estimated peak above %.0f %% of the granted memory: more parts, or a smaller scale" % (100 * a.mem_frac)
        save(); print(json.dumps(out)); return

    # ---- the GPU path ----
    dq, dt = spec.device_reads(0, Q), spec.device_reads(Q, T)
    out["query_bases"], out["target_bases"] = dq.total_bases, dt.total_bases
    q_bases, q_offsets = dq.to_host(), dq.offsets.copy()
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
    if mid_occ_global is None:
        mid_occ_global = st["mid_occ"]           # (debug scales: no oracle fixture -- the GPU's own figure; said so in the output)
        out["mid_occ_source"] = "GPU (no oracle fixture at this scale: a debug run)"
    else:
        out["mid_occ_source"] = "tests/golden/c5_full_index_stats.json (the oracle's KeyStats over the same reads)"
        out["fixture_equal_gpu"] = bool(fx["mid_occ"] == st["mid_occ"] and fx["n_minimizers"] == st["n_minimizers"] and fx["n_keys"] == st["n_keys"])
    save()

    # ---- pass 1: the part indexes ----
    bounds = [T * p // P for p in range(P + 1)]
    parts, t_index, n_mz_total, t_reads = [], 0.0, 0, 0.0
    thr = mid_occ_global // P + 1
    cands, tables = [], []

    def build_part(p):
        """-> (oracle index of part p with the whole set's mid_occ, stripped; seconds of the build; seconds of getting the reads)"""
        t1 = time.perf_counter()
        d = spec.device_reads(Q + bounds[p], bounds[p + 1] - bounds[p])         # (the device twin writes the part; bit-identical to the host twin)
        bases, offs = d.to_host(), d.offsets.copy()
        d.free()
        if p == 0:
            nchk = min(100, bounds[1])
            hchk = spec.host_reads(first=Q, n=nchk)
            assert np.array_equal(hchk.bases, bases[:int(offs[nchk])]), "the generator twins disagree"
        names = [b"r%08d" % i for i in range(Q + bounds[p], Q + bounds[p + 1])]
        Ts = O.ReadSet.from_arrays(bases, offs, names)
        tr = time.perf_counter() - t1
        o = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
        o.mid_occ = int(mid_occ_global)
        t1 = time.perf_counter()
        ixp = O.Index(Ts, o)
        ixp.strip()
        return ixp, time.perf_counter() - t1, tr

    t_stats = 0.0
    for p in range(P):
        ixp, tb, tr = build_part(p)
        t_reads += tr
        n_mz_total += ixp.n_minimizers
        cands.append(ixp.keys_at_least(thr))
        if two_pass:
            t_stats += tb
            tables.append(ixp.key_counts())
            nmz = ixp.n_minimizers
            del ixp
        else:
            t_index += tb
            parts.append(ixp)
            nmz = ixp.n_minimizers
        log("part %d / %d %s: %d minimizers, %d candidates; %.1f s so far; RSS %.1f GB" % (p + 1, P, "counted" if two_pass else "indexed", nmz, cands[-1].size, t_stats + t_index, rss_gb()))
        if rss_gb() * 1e9 > 0.62 * granted:
            out["refused"] = "resident memory above 62 %% of the grant after part %d" % (p + 1)
            save(); print(json.dumps(out)); return
    t1 = time.perf_counter()
    cand = np.unique(np.concatenate(cands)) if cands else np.zeros(0, np.uint64)
    tot = np.zeros(cand.size, np.int64)
    if two_pass:
        for keys_p, cnt_p in tables:
            j = np.searchsorted(keys_p, cand)
            j[j >= keys_p.size] = max(keys_p.size - 1, 0)
            hit = keys_p[j] == cand if keys_p.size else np.zeros(cand.size, bool)
            tot += np.where(hit, cnt_p[j].astype(np.int64), 0)
        # (counts saturate at 255 > mid_occ: a saturated term alone decides "too frequent")
        tables = None
    else:
        for ixp in parts:
            tot += ixp.counts_of(cand).astype(np.int64)
    frequent = cand[tot > mid_occ_global]
    t_close = time.perf_counter() - t1
    if two_pass: t_stats += t_close
    else: t_index += t_close
    out["oracle_index"] = {"seconds": round(t_index, 1), "statistics_pass_seconds": round(t_stats, 1) if two_pass else None, "read_transfer_seconds": round(t_reads, 1),
                           "parts": P, "two_pass": two_pass, "n_minimizers": int(n_mz_total),
                           "n_minimizers_equal_gpu": bool(n_mz_total == st["n_minimizers"]), "mid_occ": int(mid_occ_global), "candidate_keys": int(cand.size),
                           "too_frequent_keys": int(frequent.size), "rss_GB": round(rss_gb(), 1)}
    log("oracle parts:", out["oracle_index"])
    save()

    # ---- pass 2: every query against every part ----
    qn = [b"r%08d" % i for i in range(Q)]
    Qs = O.ReadSet.from_arrays(q_bases, q_offsets, qn)
    ec = np.zeros(Q, np.uint32); eh = np.zeros(Q, np.uint32)
    t_map, per_part = 0.0, []
    for p in range(P):
        if two_pass:                             # the part is built again: THIS build is the index time that is reported
            ixp, tb, tr = build_part(p)
            t_index += tb; t_reads += tr
        else:
            ixp = parts[p]
        ixp.drop_keys(frequent)
        t1 = time.perf_counter()
        rc, pc, ph = ixp.twoset_counts(Qs, threads=threads)
        dt_c = time.perf_counter() - t1
        assert rc == 0
        t_map += dt_c; per_part.append(round(dt_c, 1))
        ec += pc; eh |= ph
        if not two_pass:
            parts[p] = None
        del ixp
        log("part %d / %d mapped in %.1f s (%.1f s so far)" % (p + 1, P, dt_c, t_map))
        if t_map > a.max_map_seconds and p + 1 < P:
            out["aborted"] = "map budget exceeded after part %d" % (p + 1)
            break
    complete = "aborted" not in out
    d = ec != counts
    out["oracle_map"] = {"reads_checked": Q if complete else 0, "of": Q, "seconds": round(t_map, 1), "part_seconds": per_part,
                         "counts_differ": int(d.sum()) if complete else None, "has_mapping_differ": int((eh != has).sum()) if complete else None,
                         "counts_equal": bool(complete and not d.any() and np.array_equal(eh, has)),
                         "first_differences": [{"query": int(i), "gpu": int(counts[i]), "oracle": int(ec[i])} for i in np.nonzero(d)[0][:16]] if complete else [],
                         "overlaps_checked": int(ec.astype(np.int64).sum())}
    job_s = t_index + t_map
    out["oracle_index"]["seconds"] = round(t_index, 1)
    out["cpu_port_measured"] = {"reads_per_s": round(Q / job_s, 2), "job_seconds": round(job_s, 1), "index_seconds": round(t_index, 1), "map_seconds": round(t_map, 1),
                                "threads": threads, "complete": complete,
                                "note": "oracle = this repo's C restatement of the liblrge / minimap2-2.30 path (a port, not the reference binary), OpenMP, %d threads on the "
                                        "%.0f CPUs the host grants (%d hardware threads); targets indexed in %d parts (the host cannot hold the one index): the index "
                                        "seconds are the one build's work cut in %d, the map seconds pay every query's fixed work %d times -- an upper bound of the "
                                        "one-index port" % (threads, O.host_cpus(), os.cpu_count() or 0, P, P, P)}
    out["gpu_vs_cpu_port_measured"] = round(job_s / best, 1)
    save()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
