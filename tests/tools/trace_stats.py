#!/usr/bin/env python
"""rocprofv3 --kernel-trace CSV (one row per dispatch) -> a kernel-stats CSV with rocprofv3's own `--stats` columns, restricted to WHOLE STEPS of the
bench: the dispatches behind the second advance_chunk_kernel of the process (plan build with its autotune trials and the first warm-up chunk are in front of
it) up to the last one.  Since round 6 a plan build above 4 streams times its candidates on the device (rvc_set_plan_autotune): those trial launches are
the same kernels as the step's and are counted in rocprofv3's whole-process `--stats` table, so `Calls / steps` and `TotalDurationNs / steps` of that table no
longer describe a step.  Both tables are kept under profiles/: `<round>_kernel_stats_bench_<cfg>.csv` (rocprofv3's own, whole process) and
`<round>_kernel_stats_bench_<cfg>_steps.csv` (this tool, same trace: every step after the first; `advance_chunk_kernel`'s Calls = the number of steps).

usage: trace_stats.py <kernel_trace.csv> <out_stats.csv>"""
import csv, math, sys, collections

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
ends = [i for i, r in enumerate(rows) if "advance_chunk" in r[1]]
if len(ends) < 3:
    raise SystemExit("not enough chunks in the trace")
body = rows[ends[1] + 1: ends[-1] + 1]
acc = collections.defaultdict(list)
for _, name, ns in body:
    acc[name].append(ns)
tot = float(sum(sum(v) for v in acc.values())) or 1.0
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        n = len(v); s = sum(v); mean = s / n
        sd = math.sqrt(sum((x - mean) ** 2 for x in v) / (n - 1)) if n > 1 else 0.0
        w.writerow([name, n, s, round(mean, 6), round(100.0 * s / tot, 2), min(v), max(v), round(sd, 6)])
print("%d dispatches of %d whole steps (of %d in the trace) -> %s" % (len(body), len(ends) - 2, len(rows), sys.argv[2]))
