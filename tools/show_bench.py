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
    print("  roofline", ro["kernel"], r(ro.get("avg_launch_ms")), "frac", round(ro["frac"], 4), "achieved GB/s", r(ro["achieved"]))
    dk = d.get("roofline_dominant_kernel")
    for o in ([dk] if dk else []) + d["roofline_other"]:
        print("     ", o.get("kind"), o["kernel"][:44], o.get("bound"), r(o["ms_per_step"]), round(o["frac"], 4))
    fc = d.get("from_committed_profiles") or {}
    if fc.get("valu_issue"):
        print("  valu issue", {k: round(v["issue_frac"], 3) for k, v in fc["valu_issue"].items()})
    if d.get("config", {}).get("pack"):
        print("  pack", d["config"]["pack"]["chosen"], "granted", d["config"]["pack"]["granted_cpus"])
    if d.get("parity_vs_oracle_sample"):
        print("  parity", {k: v for k, v in d["parity_vs_oracle_sample"].items() if k != "oracle"})
    if d.get("cpu_baseline"):
        print("  cpu", r(d["cpu_baseline"]["value"]), d["cpu_baseline"]["cores"])
