run() { L=$1; shift; python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 "$@" > gpurun_out/rg_$L.log 2>&1; python3 -c "
import json
d = json.loads(open('gpurun_out/rg_$L.log').read().strip().split('\n')[-1])
print('%-22s %.1f M  step %.4f ms' % ('$L', d['value'] / 1e6, d['ms_per_step']))"; }
run base
for p in 1024 2048 4096 8192; do FTL_DEBUG_LDS_PAD_RAYS=$p run rays_pad$p; done
for p in 2048 8192; do FTL_DEBUG_LDS_PAD=$p run frames_pad$p; done
