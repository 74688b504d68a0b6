#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: per-launch average of every counter for ftl_env_kernel (step launches only)."""
import csv, glob, sys, collections, json
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
out = {}
for f in sorted(glob.glob(root + "/pmc*/*/*_counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ftl_env_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[1:] if len(v) > 2 else v          # drop the reset() launch
        out[k] = sum(v) / len(v)
for k, v in out.items():
    print("%-28s %16.0f per launch  %12.1f per env-step" % (k, v, v / n_envs))
json.dump(out, open(root + "/pmc_summary.json", "w"), indent=1)
