#!/usr/bin/env python
"""Overlap of two kernel families in a rocprofv3 kernel trace (csv): total time of each, time both run, over the last `frac` of the
trace.  usage: trace_overlap.py trace.csv nameA nameB [start_frac]"""
import csv, sys
path, na, nb = sys.argv[1], sys.argv[2], sys.argv[3]
t_lo = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
a0, a1 = min(r[0] for r in rows), max(r[1] for r in rows)
cut = a0 + (a1 - a0) * t_lo
A = sorted((s, e) for s, e, n in rows if na in n and s >= cut)
B = sorted((s, e) for s, e, n in rows if nb in n and s >= cut)
def total(iv): return sum(e - s for s, e in iv)
ov = 0; j = 0
for s, e in A:
    while j < len(B) and B[j][1] <= s: j += 1
    k = j
    while k < len(B) and B[k][0] < e:
        ov += max(0, min(e, B[k][1]) - max(s, B[k][0])); k += 1
print(na, len(A), "launches", total(A) / 1e6, "ms;", nb, len(B), "launches", total(B) / 1e6, "ms; both running", ov / 1e6, "ms")
for s, e in A[:6]: print("  A", (s - cut) / 1e6, (e - s) / 1e6)
for s, e in B[:6]: print("  B", (s - cut) / 1e6, (e - s) / 1e6)
