#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes: per-launch average of every counter per ftl_* kernel over the TIMED steps of the bench command
(the launches of the ageing phase and of reset come first and are dropped), and the derived figures bench.py quotes
(profiles/pmc_current.json: HBM traffic per step with the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md, VALU-busy fraction,
instructions per env-step).  usage: summarize_pmc.py DIR N_ENVS [--json OUT WORKLOAD SOURCE_LABEL]"""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
out = collections.defaultdict(dict)
import os
files = {}
for f in glob.glob(root + "/pmc*/*/*_counter_collection.csv"):      # one file per pass: the newest, should a directory hold several runs
    k = f.split("/pmc")[-1].split("/")[0]
    if k not in files or os.path.getmtime(f) > os.path.getmtime(files[k]):
        files[k] = f
for f in sorted(files.values()):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "ftl_" not in r["Kernel_Name"]:
            continue
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (kn, cn), v in acc.items():
        v = v[-20:] if len(v) > 40 else (v[1:] if len(v) > 2 else v)        # the 20 timed steps of the pass
        out[kn][cn] = sum(v) / len(v)
for kn in out:
    print("==", kn)
    for k, v in out[kn].items():
        print("  %-28s %16.0f per launch  %12.1f per env-step" % (k, v, v / n_envs))
json.dump(out, open(root + "/pmc_summary.json", "w"), indent=1)
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    dst, workload, label = sys.argv[i + 1], sys.argv[i + 2], sys.argv[i + 3]
    hot = {k: v for k, v in out.items() if "ftl_frames_group_kernel" in k or "ftl_rays_kernel" in k}
    # gfx950: FETCH_SIZE counts 64-byte units at half rate for these access widths -> bytes = 2 * FETCH_SIZE * 1024 (guide, HBM / rocprofv3 section);
    # WRITE_SIZE is in KiB
    traffic = sum((2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 for c in hot.values())
    valu = {}
    for kn, c in hot.items():
        if "SQ_BUSY_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
            valu[kn] = dict(valu_per_env_step=c.get("SQ_INSTS_VALU", 0) / n_envs, salu_per_env_step=c.get("SQ_INSTS_SALU", 0) / n_envs,
                            # SQ_ACTIVE_INST_VALU is summed over the SIMDs in quad-cycles; SQ_BUSY_CYCLES over the 32 SQs (XCD x SE) in cycles
                            valu_busy_frac=c["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / max(c["SQ_BUSY_CYCLES"] / 32 , 1),
                            wait_any_frac=c.get("SQ_WAIT_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1))
    cur = {}
    try:
        cur = json.load(open(dst))
    except Exception:
        pass
    cur[workload] = dict(source=label, hbm_bytes_per_step=traffic, hbm_bytes_per_env_step=traffic / n_envs, valu=valu)
    json.dump(cur, open(dst, "w"), indent=1)
    print("wrote", dst, workload, "traffic/env-step %.0f" % (traffic / n_envs))
