#!/usr/bin/env python
"""Where the device idles inside one bench step: reads a rocprofv3 kernel trace (csv), keeps the kernels of the LAST step (behind
the last long run of idle time before the last k_count... simpler: the last `frac` of the trace by time), merges their intervals
over all streams and prints the largest gaps with the kernels on either side, and the busy time per window of 50 ms."""
import csv, sys
path = sys.argv[1]; t_lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0      # fraction of the trace's span to start at
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:48], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
a0, a1 = rows[0][0], max(r[1] for r in rows)
cut = a0 + (a1 - a0) * t_lo
rows = [r for r in rows if r[0] >= cut]
print("kernels", len(rows), "span ms", (a1 - cut) / 1e6)
gaps = []; end = rows[0][1]; last = rows[0]; busy = 0; cur0 = rows[0][0]
for r in rows[1:]:
    if r[0] > end:
        gaps.append((r[0] - end, end, last[2], r[2])); busy += end - cur0; cur0 = r[0]
    if r[1] > end: end = r[1]; last = r
busy += end - cur0
print("busy ms", busy / 1e6, "idle ms", sum(g[0] for g in gaps) / 1e6, "gaps", len(gaps))
for g in sorted(gaps, reverse=True)[:25]:
    print("%8.3f ms at %9.3f  after %-48s before %s" % (g[0] / 1e6, (g[1] - cut) / 1e6, g[2], g[3]))
