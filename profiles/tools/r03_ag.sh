run() { L=$1; shift; python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 "$@" > gpurun_out/rg_$L.log 2>&1; python3 -c "
import json
d = json.loads(open('gpurun_out/rg_$L.log').read().strip().split('\n')[-1])
print('%-22s %.1f M  step %.4f ms errs %s' % ('$L', d['value'] / 1e6, d['ms_per_step'], d['config']['envs_with_error_flags']))"; }
for n in 16384 32768; do FTL_DEBUG_G8=0 run g4_$n --total-envs $n; FTL_DEBUG_G8=1 run g8_$n --total-envs $n; done
FTL_DEBUG_G8=0 run g4_8k_p1 --total-envs 8192 --parts 1; FTL_DEBUG_G8=1 run g8_8k_p1 --total-envs 8192 --parts 1
FTL_DEBUG_G8=1 run g8_8k_p3 --total-envs 8192 --parts 3
FTL_DEBUG_G8=1 run g8_D_p3 --workload D --parts 3
