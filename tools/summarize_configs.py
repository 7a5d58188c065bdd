#!/usr/bin/env python3
"""Turn one tools/profile_configs.sh run (gpurun_out/<tag>_<name>*) into the tracked files under profiles/:
   <round>_<name>.json (the run's own line), <round>_<name>_kernel_stats.csv and <round>_configs_traffic.json (FETCH_SIZE /
   WRITE_SIZE summed over every kernel of the pass, per step, next to the algorithmic bytes of SURVEY.md 8(d); raw counters: the
   x2 read correction of gfx950 is calibrated for streaming reads only, see summarize_profiles.py).
   usage: tools/summarize_configs.py <tag> <round-prefix>      e.g.  r4c r03"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rnd = sys.argv[1], sys.argv[2]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
# steps a counter pass holds: bench.py --steps 2 --warmup 1 + its instrumented step; run_config.py --repeat 2
NAMES = {"c5t_pb_fwd": 4, "c5t_pb_inv": 4, "c5t_ont_fwd": 4, "c5t_ont_inv": 4, "c3_ava": 2}


def totals(path, counter):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            per[r["Kernel_Name"].split("(")[0].replace("void ", "")] += float(r["Counter_Value"]) * 1024.0
    return per


out = {}
for name, steps in NAMES.items():
    line = json.loads(open(os.path.join(G, "%s_%s.json" % (tag, name))).read().strip().splitlines()[-1])
    json.dump(line, open(os.path.join(P, "%s_%s.json" % (rnd, name)), "w"), indent=1)
    shutil.copy(os.path.join(G, "%s_%s_stats" % (tag, name), "s_kernel_stats.csv"), os.path.join(P, "%s_%s_kernel_stats.csv" % (rnd, name)))
    f = totals(os.path.join(G, "%s_%s_fetch" % (tag, name), "f_counter_collection.csv"), "FETCH_SIZE")
    w = totals(os.path.join(G, "%s_%s_write" % (tag, name), "w_counter_collection.csv"), "WRITE_SIZE")
    if "ms_per_step" in line:        # a bench.py line
        ms = line["ms_per_step"]; rps = line["value"]; stage = line["stage_ms_per_step"]
        anchors = line["work_per_step"]["anchors"]; alg = line["roofline"]["whole_path_alg_GBps"] * ms * 1e-3
    else:                             # tools/run_config.py
        ms = line["step_s"] * 1e3; rps = line["reads_per_s"]; stage = line["stage_ms"]; anchors = line["anchors"]
        alg = line["alg_GBps_whole_path"] * ms * 1e-3
    tf, tw = sum(f.values()) / steps, sum(w.values()) / steps
    top = sorted(set(f) | set(w), key=lambda k: -(f.get(k, 0) + w.get(k, 0)))[:8]
    out[name] = {"ms_per_step": ms, "reads_per_s": rps, "anchors_per_step": anchors, "stage_ms_per_step": {k: round(v, 2) for k, v in stage.items()},
                 "fetch_GB_per_step_raw": tf / 1e9, "write_GB_per_step_raw": tw / 1e9, "algorithmic_GB_per_step": alg,
                 "traffic_over_algorithmic_raw": (tf + tw) / 1e9 / alg if alg else None, "steps_in_pass": steps,
                 "top_kernels_GB_per_step_raw(fetch+write)": {k: round((f.get(k, 0) + w.get(k, 0)) / steps / 1e9, 2) for k in top}}
json.dump(out, open(os.path.join(P, rnd + "_configs_traffic.json"), "w"), indent=1)
for name, v in out.items():
    print("%-12s %8.2f ms  %10.0f reads/s  fetch %7.1f GB  write %7.1f GB  algorithmic %6.1f GB  ratio %.2f" % (
        name, v["ms_per_step"], v["reads_per_s"], v["fetch_GB_per_step_raw"], v["write_GB_per_step_raw"], v["algorithmic_GB_per_step"], v["traffic_over_algorithmic_raw"]))
