cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-tail}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/r03_t_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r03_t_tests.log
AB_ARGS="--gen-sample 0" bash profiles/tools/ab_bench.sh 2 h3 $V
AB_ARGS="--gen-sample 0 --total-envs 8192" bash profiles/tools/ab_bench.sh 2 h3 $V
AB_ARGS="--gen-sample 0 --workload E" bash profiles/tools/ab_bench.sh 1 h3 $V
AB_ARGS="--gen-sample 0 --workload D" bash profiles/tools/ab_bench.sh 1 h3 $V
