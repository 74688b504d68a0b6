#!/bin/bash
# ab_bench.sh ROUNDS NAME... : interleaved A/B of library builds (variants_NAME.so) on ONE box -- box-to-box spread is larger than most
# single optimisations; prints value and per-kernel HIP-event times of every run
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do for v in "$@"; do
  FTL_LIB=$PWD/variants_$v.so python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline ${AB_ARGS} > gpurun_out/ab_${v}_$r.log 2>&1
  python3 - <<PY
import json
d = json.loads(open("gpurun_out/ab_${v}_$r.log").read().strip().split("\n")[-1])
k = d["roofline"]["kernels_us"] or {}
print("%-14s round $r  %.1f M  step %.4f ms  frames %.1f rays %.1f regroup %.1f" % ("$v", d["value"] / 1e6, d["ms_per_step"], k.get("frames_us", 0), k.get("rays_us", 0), k.get("regroup_us", 0)))
PY
done; done
