# dynamic VALU/SALU instructions of the ray kernel by phase: builds that stop after phase k (variants_stopK.so, -DFTL_RAYS_STOP=K)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${STOPS:-0 1 2 3 4 5 6}; do
  FTL_LIB=$PWD/variants_stop$v.so rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d gpurun_out/stop$v -- python3 bench.py --parts 1 --steps 10 --warmup 5 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 > gpurun_out/stop$v.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/stop$v/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rays" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("stop after $v:", {k: round(sum(x[-10:])/10/65536,1) for k,x in acc.items()})
PY
done
