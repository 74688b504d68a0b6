#!/bin/bash
run() {  # label, args...
  L=$1; shift
  python3 bench.py --steps 150 --warmup 20 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 "$@" > gpurun_out/ps_$L.log 2>&1
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/ps_$L.log").read().strip().split("\n")[-1])
print("%-22s %.1f M  step %.4f ms  errs %s parts %s" % ("$L", d["value"] / 1e6, d["ms_per_step"], d["config"]["envs_with_error_flags"], d["config"]["parts"]))
PY
}
run F_split --workload F
export FTL_SPLIT=0
for p in 1 2 3; do run F_nosplit_p$p --workload F --parts $p; done
unset FTL_SPLIT
for p in 1 2 3; do run C_p$p --workload C --parts $p; done
for p in 1 2; do run L_p$p --workload L --parts $p; done
for p in 1 2; do run T_p$p --workload T --parts $p; done
