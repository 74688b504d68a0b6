#!/bin/bash
# quick loop on the GPU box: core parity tests + one steady-state bench line of workload B (usage: quick_check.sh TAG)
TAG=${1:-q}
python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/${TAG}_tests.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.log").read().strip().split("\n")[-1])
print("value %.1f M  ms/step %.4f  kernels %s  errs %s" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["kernels_us"], d["config"]["envs_with_error_flags"]))
PY
