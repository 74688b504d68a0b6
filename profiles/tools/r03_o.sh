cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FTL_DEFER=2 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_configs.py -m gpu -x -q > gpurun_out/r03_o_tests.log 2>&1; echo "tests(FTL_DEFER=2) rc=$?"; tail -3 gpurun_out/r03_o_tests.log
for args in "" "--total-envs 8192" "--total-envs 32768" "--workload E" "--workload D"; do
echo "== $args"
AB_ARGS="--gen-sample 0 $args" bash profiles/tools/ab_bench.sh 1 h1 v2
FTL_DEFER=2 AB_ARGS="--gen-sample 0 $args" bash profiles/tools/ab_bench.sh 1 v2
done
