run() { L=$1; shift; python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 "$@" > gpurun_out/rg_$L.log 2>&1; python3 -c "
import json
d = json.loads(open('gpurun_out/rg_$L.log').read().strip().split('\n')[-1])
print('%-22s %.1f M  step %.4f ms' % ('$L', d['value'] / 1e6, d['ms_per_step']))"; }
run p2
run p3 --parts 3
GPU_MAX_HW_QUEUES=8 run q8_p3 --parts 3
GPU_MAX_HW_QUEUES=8 run q8_p4 --parts 4
GPU_MAX_HW_QUEUES=2 run q2_p2
