#!/usr/bin/env python3
"""Condensed view of bench.py lines (gpurun_out/*/bench*.json): clocks, stage times, work, roofline candidates."""
import json
import sys

for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        print(f, "unreadable:", e); continue
    r = lambda x: None if x is None else round(x, 2)
    print(f, "value", round(d["value"]), "ms", r(d["ms_per_step"]), "| resident", d.get("resident") and r(d["resident"]["ms_per_step"]),
          "| from_host", d.get("from_host") and r(d["from_host"]["ms_per_step"]))
    print("  stages", {k: r(v) for k, v in d["stage_ms_per_step"].items()})
    w = d["work_per_step"]
    print("  work", {k: w.get(k) for k in ("anchors", "anchors_kept", "groups", "groups_chained", "chain_anchors", "batches", "lpg_split")})
    ro = d["roofline"]
    print("  roofline", ro["kernel"], r(ro["ms_per_step"]), round(ro["frac"], 4), "traffic", ro.get("traffic"), "| whole path frac", round(ro["whole_path_frac"], 4))
    for o in d["roofline_other"]:
        print("     ", o["kind"], o["kernel"][:44], r(o["ms_per_step"]), round(o["frac"], 4))
    if d.get("parity_vs_oracle_sample"):
        print("  parity", {k: v for k, v in d["parity_vs_oracle_sample"].items() if k != "oracle"})
    if d.get("cpu_baseline"):
        print("  cpu", r(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
