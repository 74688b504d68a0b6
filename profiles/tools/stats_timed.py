#!/usr/bin/env python3
"""Per-kernel duration statistics over the LAST n launches of every ftl_* kernel in a rocprofv3 kernel-trace CSV: the timed steps of a
bench.py run, without the ageing / warm-up launches that `--stats` averages in.  usage: stats_timed.py KERNEL_TRACE.csv N_STEPS"""
import collections
import csv
import sys

rows = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ftl_" in r["Kernel_Name"]:
        rows[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
n = int(sys.argv[2])
print("kernel,launches_counted,avg_ns,min_ns,max_ns,launches_total")
for k, v in sorted(rows.items()):
    m = n if ("frames_group" in k or "rays" in k) else min(len(v), max(1, n // 4))        # the regroup kernels run every fourth step (every 2nd until round 3)
    w = v[-m:]
    print('"%s",%d,%.1f,%d,%d,%d' % (k, len(w), sum(w) / len(w), min(w), max(w), len(v)))
