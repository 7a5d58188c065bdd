#!/usr/bin/env python3
"""Turn one tools/profile_round.sh run (gpurun_out/<tag>_*) into the tracked files under profiles/:
   <round>_bench.json, <round>_kernel_stats.csv, <round>_pmc_sq.csv, <round>_pmc_fetch.csv, <round>_pmc_write.csv
   and <round>_hbm_traffic.json (corrected HBM bytes and VALU instructions per launch of every kernel).
   usage: tools/summarize_profiles.py <tag> <round-prefix> [suffix]     e.g.  r4p r04   (suffix "_c4": profiles/r04_*_c4.*)"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
suf = sys.argv[3] if len(sys.argv) > 3 else ""
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def agg(path):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        e = out[k][r["Counter_Name"]]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
    return out


shutil.copy(os.path.join(G, tag + "_stats", "s_kernel_stats.csv"), os.path.join(P, rnd + "_kernel_stats" + suf + ".csv"))
bench = json.loads(open(os.path.join(G, tag + "_bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(P, rnd + "_bench" + suf + ".json"), "w"), indent=1)
under = json.loads(open(os.path.join(G, tag + "_stats.json")).read().strip().splitlines()[-1])
json.dump(under, open(os.path.join(P, rnd + "_bench_under_rocprof" + suf + ".json"), "w"), indent=1)

for name, sub, pre in (("sq", "_sq", "q"), ("fetch", "_fetch", "f"), ("write", "_write", "w")):
    a = agg(os.path.join(G, tag + sub, pre + "_counter_collection.csv"))
    with open(os.path.join(P, "%s_pmc_%s%s.csv" % (rnd, name, suf)), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "counter", "dispatches", "sum", "per_dispatch"])
        for k in sorted(a):
            for c in sorted(a[k]):
                n, v = a[k][c]
                wr.writerow([k, c, n, "%.6g" % v, "%.6g" % (v / n)])

fetch = agg(os.path.join(G, tag + "_fetch", "f_counter_collection.csv"))
write = agg(os.path.join(G, tag + "_write", "w_counter_collection.csv"))
sq = agg(os.path.join(G, tag + "_sq", "q_counter_collection.csv"))

# HBM bytes per launch of every kernel: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies 128-B read requests at 64 B
# (MI355X_MICROARCH.md, HBM section): reads x2, writes x1.  The x2 is calibrated for wide coalesced streaming reads only
# (checked below on k_rs_hist, a pure 8 B/key stream); for gather patterns (k_lookup: one 16-byte slot per random line,
# k_chain_lpg: 16 B per lane from 64 lines) it is uncalibrated and the corrected figure is an upper bound -- the raw counters
# are in <round>_pmc_fetch.csv / _pmc_write.csv.
per = {}
for kern in sorted(set(fetch) | set(write)):
    f = fetch.get(kern, {}).get("FETCH_SIZE"); w = write.get(kern, {}).get("WRITE_SIZE")
    if not f or not w:
        continue
    v = sq.get(kern, {}).get("SQ_INSTS_VALU")
    per[kern] = {"launches_in_pass": f[0], "fetch_raw_bytes_per_launch": f[1] / f[0] * 1024, "write_raw_bytes_per_launch": w[1] / w[0] * 1024,
                 "hbm_bytes_per_launch_corrected": f[1] / f[0] * 1024 * 2.0 + w[1] / w[0] * 1024,
                 "valu_insts_per_launch": v[1] / v[0] if v else None}
big = 0.0
for r in csv.DictReader(open(os.path.join(G, tag + "_fetch", "f_counter_collection.csv"))):
    if r["Kernel_Name"].startswith("void k_rs_hist<false") and r["Counter_Name"] == "FETCH_SIZE":
        big = max(big, float(r["Counter_Value"]))
out = {
    "source": "tools/profile_round.sh %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_*, separate passes of bench.py --no-cpu-baseline --no-resident "
              "--parity-sample 0 --steps 1 --warmup 0; per-launch figures average the launches of the pass)" % tag,
    "workload": bench.get("config", {}).get("workload"),
    "fetch_correction": 2.0, "write_correction": 1.0,
    "calibration": {"k_rs_hist_false_largest_launch_FETCH_SIZE_KiB": big, "index_minimizers": bench.get("work_per_step", {}).get("rs_scatter_items"),
                    "note": "the largest k_rs_hist<false> launch reads 8 B x (index minimizers); FETCH_SIZE reports half of it on gfx950 "
                            "(128-B requests tallied at 64 B)"},
    "valu_issue_note": "measured (tools/micro/valu_rate.hip): a SIMD issues a wave64 v_add / v_fma / v_cndmask every 2.4-2.7 cycles, v_cmp / v_max / DPP every 4.3; one wavefront issues a dependent instruction every ~9 cycles",
    "kernels": per,
}
# whole path: every kernel's raw counters summed over the pass, per step (a pass runs the timed step and, for small jobs, the
# instrumented one that follows it: told apart by the k_lookup launches), next to the algorithmic bytes of SURVEY.md 8(d) the bench line reports
look = [v for k, v in fetch.items() if k.startswith("k_lookup")]
per_step = max(1.0, float(bench.get("work_per_step", {}).get("lookup_launches", 1) or 1))
steps_in_pass = max(1, int(round((look[0]["FETCH_SIZE"][0] if look else per_step) / per_step)))
tot_f = sum(v["FETCH_SIZE"][1] for v in fetch.values() if "FETCH_SIZE" in v) * 1024 / steps_in_pass
tot_w = sum(v["WRITE_SIZE"][1] for v in write.values() if "WRITE_SIZE" in v) * 1024 / steps_in_pass
ro_ = bench.get("roofline", {})
alg = (ro_.get("alg_GB_per_step") if ro_.get("alg_GB_per_step") is not None else ro_.get("whole_path_alg_GBps", 0.0) * bench.get("ms_per_step", 0.0) * 1e-3)       # (round 6: `roofline` is the whole-path block)
cfg_name = (bench.get("config", {}).get("workload") or "").split(":")[0]
out.update({"config": cfg_name, "inverse": "--use-min-ref" in (bench.get("config", {}).get("workload") or ""), "steps_in_pass": steps_in_pass,
            "clock": bench.get("config", {}).get("clock"), "fetch_GB_per_step": tot_f / 1e9, "write_GB_per_step": tot_w / 1e9, "algorithmic_GB_per_step": alg,
            "traffic_over_algorithmic": {"raw_counters": (tot_f + tot_w) / 1e9 / alg if alg else None,
                                         "reads_x2_correction": (2 * tot_f + tot_w) / 1e9 / alg if alg else None}})
json.dump(out, open(os.path.join(P, rnd + "_hbm_traffic" + suf + ".json"), "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch_corrected"] for k, v in per.items()}, indent=1))
