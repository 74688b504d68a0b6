cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc_series -- python3 bench.py --steps 120 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_series.log 2>&1
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/pmc_series/*/*_counter_collection.csv')[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'ftl_' in r['Kernel_Name']:
        acc['frames' if 'frames' in r['Kernel_Name'] else 'rays'][r['Counter_Name']].append(float(r['Counter_Value']))
for k in acc:
    for c,v in acc[k].items():
        print(k, c, [round(x/65536,1) for x in v[::12]])
PY
