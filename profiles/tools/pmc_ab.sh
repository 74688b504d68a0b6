#!/bin/bash
# pmc_ab.sh NAME... : instruction / busy counters of library variants (variants_NAME.so) on workload B, steady state; one counter set per pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  export FTL_LIB=$GRAFT_REPO_ROOT/variants_$v.so
  B="python3 bench.py --steps 10 --warmup 5 --age 300 --no-cpu-baseline --kernel-steps 0"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pab_${v}_1 -- $B > gpurun_out/pab_${v}_1.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/pab_${v}_2 -- $B > gpurun_out/pab_${v}_2.log 2>&1
  python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pab_${v}_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "ftl_rays" in kn or "ftl_frames" in kn:
            acc[(kn.split("(")[0][:44], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("== $v")
for (kn, cn), vals in sorted(acc.items()):
    vals = vals[len(vals) // 2:]            # the timed steps (the ageing steps come first)
    print("  %-46s %-22s %14.0f per launch %10.1f per env-step" % (kn, cn, sum(vals) / len(vals), sum(vals) / len(vals) / 65536))
PY
done
