#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: per-launch average of every counter, per ftl_* kernel (first launch = reset, dropped)."""
import csv, glob, sys, collections, json
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
out = collections.defaultdict(dict)
for f in sorted(glob.glob(root + "/pmc*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ftl_" not in r["Kernel_Name"]:
            continue
        acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, cn), v in acc.items():
        v = v[1:] if len(v) > 2 else v
        out[kn][cn] = sum(v) / len(v)
for kn in out:
    print("==", kn)
    for k, v in out[kn].items():
        print("  %-28s %16.0f per launch  %12.1f per env-step" % (k, v, v / n_envs))
json.dump(out, open(root + "/pmc_summary.json", "w"), indent=1)
