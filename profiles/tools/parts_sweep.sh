#!/bin/bash
# parts_sweep.sh : env-steps/s of bench.py for 1..4 independently stepped sub-batches (PipelinedVecGame) per workload / batch size
run() {  # label, args...
  L=$1; shift
  python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 "$@" > gpurun_out/ps_$L.log 2>&1
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/ps_$L.log").read().strip().split("\n")[-1])
print("%-22s %.1f M  step %.4f ms  errs %s" % ("$L", d["value"] / 1e6, d["ms_per_step"], d["config"]["envs_with_error_flags"]))
PY
}
for p in 1 2 3 4; do run B64k_p$p --parts $p; done
for p in 1 2 4; do run B32k_p$p --total-envs 32768 --parts $p; done
for p in 1 2 4; do run B16k_p$p --total-envs 16384 --parts $p; done
for p in 1 2 4; do run B8k_p$p --total-envs 8192 --parts $p; done
for p in 1 2 4; do run B4k_p$p --total-envs 4096 --parts $p; done
for p in 1 2 4; do run D_p$p --workload D --parts $p; done
for p in 1 2 4; do run E_p$p --workload E --parts $p; done
