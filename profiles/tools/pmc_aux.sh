#!/bin/bash
# pmc_aux.sh WORKLOAD : instruction / busy counters of ftl_aux_kernel on workload C, L or T (steady state; one counter set per pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
W=${1:-L}
B="python3 bench.py --workload $W --steps 10 --warmup 5 --age 300 --no-cpu-baseline --kernel-steps 0"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/paux_${W}_1 -- $B > gpurun_out/paux_${W}_1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/paux_${W}_2 -- $B > gpurun_out/paux_${W}_2.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/paux_${W}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "ftl_aux" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for cn, vals in sorted(acc.items()):
    vals = vals[len(vals) // 2:]
    print("  %-22s %14.0f per launch %10.1f per env-step" % (cn, sum(vals) / len(vals), sum(vals) / len(vals) / 65536))
PY
