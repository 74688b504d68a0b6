cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_scenario_gen.py tests/test_gpu_metrics.py -m gpu -x -q > gpurun_out/r03_l_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_l_tests.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --ring 8192 > gpurun_out/r03_l_ring.log 2>&1; tail -1 gpurun_out/r03_l_ring.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ring', '%.1f M' % (d['value']/1e6), d['config']['resets_per_s'], d['config']['scenario_supply'])"
