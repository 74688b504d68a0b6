#!/bin/bash
# instruction-cache and stall counters of both kernels (one pass each), workload B steady state
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 10 --warmup 5 --age 300 --no-cpu-baseline --kernel-steps 0"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pic1 -- $B > gpurun_out/pic1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_SENDMSG SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d gpurun_out/pic2 -- $B > gpurun_out/pic2.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pic*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "ftl_rays" in kn or "ftl_frames" in kn:
            acc[(kn.split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (kn, cn), vals in sorted(acc.items()):
    vals = vals[len(vals) // 2:]
    print("  %-42s %-24s %14.0f per launch" % (kn, cn, sum(vals) / len(vals)))
PY
tail -3 gpurun_out/pic1.log | cut -c1-300
