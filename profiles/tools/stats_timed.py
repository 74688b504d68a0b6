#!/usr/bin/env python3
"""Per-kernel duration statistics over the LAST n steps' launches of every ftl_* kernel in a rocprofv3 kernel-trace CSV: the timed steps of a
bench.py run, without the ageing / warm-up launches that `--stats` averages in.  usage: stats_timed.py KERNEL_TRACE.csv N_STEPS [PARTS]
PARTS = bench.py's --parts (sub-batches per step, each with its own frame and ray launch; default 1).  With PARTS > 1 the launches of
different streams overlap: the last line gives the time during which at least one of the counted launches was running (their union) per
step -- the step duration as the kernel trace sees it -- and the sum of the durations over that union (the overlap factor)."""
import collections
import csv
import sys

rows = collections.defaultdict(list)
spans = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ftl_" in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0]
        rows[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        spans[k].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
n = int(sys.argv[2])
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
print("kernel,launches_counted,avg_ns,min_ns,max_ns,launches_total")
counted = []
for k, v in sorted(rows.items()):
    per_step = ("frames_group" in k or "rays" in k or "aux" in k or "tracker1" in k)
    m = min(len(v), n * parts) if per_step else min(len(v), max(1, n * parts // 4))   # the regroup kernels run every fourth step of a handle (every 2nd until round 3)
    w = v[-m:]
    counted += sorted(spans[k])[-m:]
    print('"%s",%d,%.1f,%d,%d,%d' % (k, len(w), sum(w) / len(w), min(w), max(w), len(v)))
counted.sort()
union, cur_s, cur_e = 0, None, None
for s0, e0 in counted:
    if cur_e is None or s0 > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s0, e0
    else:
        cur_e = max(cur_e, e0)
if cur_e is not None:
    union += cur_e - cur_s
total = sum(e0 - s0 for s0, e0 in counted)
print('"[union of the counted launches per step]",%d,%.1f,,,' % (n, union / n))
print('"[sum of their durations / union = overlap factor]",,%.3f,,,' % (total / max(union, 1)))
