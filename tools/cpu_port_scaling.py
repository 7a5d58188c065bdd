#!/usr/bin/env python
"""How the CPU port (oracle/, the checker that doubles as bench.py's cpu_baseline) scales over the host's threads, with and
without its malloc tuning (oracle/lrge_oracle.c: tune_malloc).  TEST / MEASUREMENT INFRASTRUCTURE.

  python tools/cpu_port_scaling.py [--targets 50000] [--queries 4096] [--threads 32,128,256] [--preset pb]

One child process per (threads, mallopt) setting: mallopt is process-wide and must be set before the first large allocation.
Prints one JSON line per setting and a summary line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(a):
    import numpy as np  # noqa: F401
    from lrge_amd import synth_cb
    from oracle import oracle as O
    spec, Q, T = synth_cb.spec_of(a.config)
    preset = 1 if a.preset == "pb" else 0
    opt = O.make_opt(O.PRESET_AVA_PB if preset else O.PRESET_AVA_ONT, dual=True)
    t = spec.host_reads(first=Q, n=a.targets)
    t0 = time.perf_counter()
    ix = O.Index(O.ReadSet(t.seqs(), t.names), opt)
    t_index = time.perf_counter() - t0
    q = spec.host_reads(first=0, n=a.queries)
    Qs = O.ReadSet(q.seqs(), q.names)
    th = int(a.child)
    ix.twoset_counts(O.ReadSet(q.seqs()[:256], q.names[:256]), threads=th)      # warm the threads' arenas
    t0 = time.perf_counter()
    rc, c, _ = ix.twoset_counts(Qs, threads=th)
    t_map = time.perf_counter() - t0
    print(json.dumps({"threads": th, "mallopt": not os.environ.get("LO_NO_MALLOPT"), "index_s": round(t_index, 2), "map_s": round(t_map, 3),
                      "reads_per_s": round(a.queries / t_map, 1), "thread_ms_per_read": round(t_map * th / a.queries * 1e3, 2),
                      "counts_sum": int(c.sum()), "targets": a.targets, "queries": a.queries, "preset": a.preset}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c5_human_twoset")
    ap.add_argument("--targets", type=int, default=50000)
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--threads", default="16,32,64")
    ap.add_argument("--preset", default="pb")
    ap.add_argument("--mallopt-ab", action="store_true")
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        child(a)
        return
    rows = []
    for th in a.threads.split(","):
        for no in ((False, True) if a.mallopt_ab else (False,)):
            env = dict(os.environ)
            env.pop("LO_NO_MALLOPT", None)
            if no:
                env["LO_NO_MALLOPT"] = "1"
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", a.config, "--targets", str(a.targets), "--queries", str(a.queries),
                                  "--preset", a.preset, "--child", th], env=env, capture_output=True, text=True)
            line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else json.dumps({"threads": int(th), "error": out.stderr[-300:]})
            print(line, flush=True)
            rows.append(json.loads(line))
    sums = {r.get("counts_sum") for r in rows if "counts_sum" in r}
    from oracle import oracle as O
    print(json.dumps({"summary": "cpu port scaling", "hw_threads": os.cpu_count(), "cpus_granted": O.host_cpus(), "same_counts_everywhere": len(sums) == 1}))


if __name__ == "__main__":
    main()
