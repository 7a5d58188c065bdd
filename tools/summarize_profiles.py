#!/usr/bin/env python3
"""Turn one tools/profile_round.sh run (gpurun_out/<tag>_*) into the tracked files under profiles/:
   <round>_bench.json, <round>_kernel_stats.csv, <round>_pmc_sq.csv, <round>_pmc_fetch.csv, <round>_pmc_write.csv
   and chain_pmc.json (HBM bytes per launch for the kernels bench.py's roofline block names).
   usage: tools/summarize_profiles.py <tag> <round-prefix>      e.g.  r1k r01"""
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


def agg(path):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        e = out[k][r["Counter_Name"]]
        e[0] += 1
        e[1] += float(r["Counter_Value"])
    return out


shutil.copy(os.path.join(G, tag + "_stats", "s_kernel_stats.csv"), os.path.join(P, rnd + "_kernel_stats.csv"))
bench = json.loads(open(os.path.join(G, tag + "_bench.json")).read().strip().splitlines()[-1])
json.dump(bench, open(os.path.join(P, rnd + "_bench.json"), "w"), indent=1)
under = json.loads(open(os.path.join(G, tag + "_stats.json")).read().strip().splitlines()[-1])
json.dump(under, open(os.path.join(P, rnd + "_bench_under_rocprof.json"), "w"), indent=1)

for name, sub, pre in (("sq", "_sq", "q"), ("fetch", "_fetch", "f"), ("write", "_write", "w")):
    a = agg(os.path.join(G, tag + sub, pre + "_counter_collection.csv"))
    with open(os.path.join(P, "%s_pmc_%s.csv" % (rnd, name)), "w", newline="") as f:
        wr = csv.writer(f)
        wr.writerow(["kernel", "counter", "dispatches", "sum", "per_dispatch"])
        for k in sorted(a):
            for c in sorted(a[k]):
                n, v = a[k][c]
                wr.writerow([k, c, n, "%.6g" % v, "%.6g" % (v / n)])

fetch = agg(os.path.join(G, tag + "_fetch", "f_counter_collection.csv"))
write = agg(os.path.join(G, tag + "_write", "w_counter_collection.csv"))


def per_launch(kern):
    # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM
    # section; calibrated here on k_rs_hist, a pure 8 B/key stream): reads x2, writes x1
    f = fetch.get(kern, {}).get("FETCH_SIZE"); w = write.get(kern, {}).get("WRITE_SIZE")
    if not f or not w:
        return None
    return (f[1] / f[0]) * 1024 * 2.0 + (w[1] / w[0]) * 1024 * 1.0


out = {
    "source": "tools/profile_round.sh %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 0)" % tag,
    "fetch_correction": 2.0, "write_correction": 1.0,
    "fetch_correction_note": "x2 is calibrated for wide coalesced streaming reads (k_rs_hist). k_chain_lpg reads 16 B per lane from 64 "
                             "different lines per instruction; for that pattern the factor is uncalibrated, so its corrected figure is an "
                             "upper bound (raw FETCH_SIZE + WRITE_SIZE: see *_pmc_fetch.csv / *_pmc_write.csv)",
    "k_chain_lpg_hbm_bytes_per_launch": per_launch(next((k for k in fetch if k.startswith("k_chain_lpg<")), "k_chain_lpg")),
    "k_chain_hw_hbm_bytes_per_launch": per_launch("k_chain_hw"),
    "k_rs_scatter_hbm_bytes_per_launch": None,
}
# k_rs_scatter: both instantiations, averaged over all launches of the step
tot, n = 0.0, 0
for kern in fetch:
    if not kern.startswith("k_rs_scatter"):
        continue
    f = fetch.get(kern, {}).get("FETCH_SIZE"); w = write.get(kern, {}).get("WRITE_SIZE")
    if f and w:
        tot += f[1] * 1024 * 2.0 + w[1] * 1024 * 1.0; n += f[0]
if n:
    out["k_rs_scatter_hbm_bytes_per_launch"] = tot / n
# calibration of the x2 read correction: the largest k_rs_hist<false> launch of a step is the index sort's,
# a pure stream of 8 B per index minimizer
big = 0.0
for r in csv.DictReader(open(os.path.join(G, tag + "_fetch", "f_counter_collection.csv"))):
    if r["Kernel_Name"].startswith("void k_rs_hist<false>") and r["Counter_Name"] == "FETCH_SIZE":
        big = max(big, float(r["Counter_Value"]))
nmz = None
for line in open(os.path.join(G, tag + "_bench.json")):
    pass
out["calibration"] = {"k_rs_hist_false_largest_launch_FETCH_SIZE_KiB": big,
                      "note": "that launch reads 8 B x (index minimizers); FETCH_SIZE reports half of it on gfx950 (128-B requests tallied at 64 B)"}
# VALU issue of the chain stage (SQ pass): wave64 VALU instructions per launch; each occupies its SIMD for 4 cycles
sq = agg(os.path.join(G, tag + "_sq", "q_counter_collection.csv"))
for kern, key in ((next((k for k in sq if k.startswith("k_chain_lpg<")), "k_chain_lpg"), "k_chain_lpg"), ("k_chain_hw", "k_chain_hw")):
    v = sq.get(kern, {}).get("SQ_INSTS_VALU")
    out[key + "_valu_insts_per_launch"] = v[1] / v[0] if v else None
a, b = out["k_chain_lpg_hbm_bytes_per_launch"], out["k_chain_hw_hbm_bytes_per_launch"]
out["chain_stage_hbm_bytes_per_step"] = (a or 0) + (b or 0) if (a or b) else None
json.dump(out, open(os.path.join(P, "chain_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
