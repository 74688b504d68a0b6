# WRITE_SIZE of the ray kernel with and without the LDS it had before the float32 minima (is the write traffic a matter of how many wavefronts are resident?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --parts 1 --steps 20 --warmup 5 --no-cpu-baseline --kernel-steps 0 --gen-sample 0"
for pad in 0 896 2048; do
  FTL_DEBUG_LDS_PAD_RAYS=$pad rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/wr_$pad -- $B > gpurun_out/wr_$pad.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("gpurun_out/wr_$pad/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rays" in r["Kernel_Name"] or "frames_group" in r["Kernel_Name"]: acc[r["Kernel_Name"][:30]].append(float(r["Counter_Value"]))
print("pad $pad:", {k: round(sum(x[-20:])/20,0) for k,x in acc.items()})
PY
done
