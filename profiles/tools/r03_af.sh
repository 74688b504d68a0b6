run() { L=$1; shift; python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --kernel-steps 100 --gen-sample 0 "$@" > gpurun_out/rg_$L.log 2>&1; python3 -c "
import json
d = json.loads(open('gpurun_out/rg_$L.log').read().strip().split('\n')[-1]); k=d['roofline']['kernels_us']
print('%-22s %.1f M  step %.4f ms frames %.1f rays %.1f errs %s' % ('$L', d['value'] / 1e6, d['ms_per_step'], k['frames_us'], k['rays_us'], d['config']['envs_with_error_flags']))"; }
export FTL_LIB=$PWD/variants_g8.so
for n in 8192 16384; do run g4_$n --total-envs $n; FTL_DEBUG_G8=1 run g8_$n --total-envs $n; done
run g4_D --workload D; FTL_DEBUG_G8=1 run g8_D --workload D
