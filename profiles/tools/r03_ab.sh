run() { L=$1; shift; python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 "$@" > gpurun_out/rg_$L.log 2>&1; python3 -c "
import json
d = json.loads(open('gpurun_out/rg_$L.log').read().strip().split('\n')[-1])
print('%-22s %.1f M  step %.4f ms' % ('$L', d['value'] / 1e6, d['ms_per_step']))"; }
export FTL_NO_REGROUP=0
for e in 8 16 32; do FTL_REGROUP_EVERY=$e run B64k_every$e; done
unset FTL_NO_REGROUP
for w in "B32k --total-envs 32768" "B16k --total-envs 16384" "B8k --total-envs 8192" "D --workload D" "E --workload E"; do set -- $w; L=$1; shift
  run ${L}_auto "$@"; FTL_NO_REGROUP=0 FTL_REGROUP_EVERY=8 run ${L}_on8 "$@"; FTL_NO_REGROUP=0 FTL_REGROUP_EVERY=16 run ${L}_on16 "$@"; done
