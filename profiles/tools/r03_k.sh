cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_scenario_gen.py tests/test_gpu_metrics.py tests/test_abi.py -m gpu -x -q > gpurun_out/r03_k_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r03_k_tests.log
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --ring 8192 > gpurun_out/r03_k_ring.log 2>&1; tail -1 gpurun_out/r03_k_ring.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ring', '%.1f M' % (d['value']/1e6), d['config']['resets_per_s'], d['config']['scenario_supply'])"
for n in 16384 32768; do for rg in 1 0; do FTL_NO_REGROUP=$rg FTL_LIB=$PWD/variants_cur.so python bench.py --total-envs $n --steps 200 --warmup 20 --no-cpu-baseline --gen-sample 0 > gpurun_out/r03_k_rg.log 2>&1; tail -1 gpurun_out/r03_k_rg.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B $n NO_REGROUP=$rg', '%.1f M' % (d['value']/1e6), d['roofline']['kernels_us'])"; done; done
for rg in 1 0; do FTL_NO_REGROUP=$rg python bench.py --workload E --steps 200 --warmup 20 --no-cpu-baseline --gen-sample 0 > gpurun_out/r03_k_rg.log 2>&1; tail -1 gpurun_out/r03_k_rg.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('E NO_REGROUP=$rg', '%.1f M' % (d['value']/1e6), d['roofline']['kernels_us'])"; done
