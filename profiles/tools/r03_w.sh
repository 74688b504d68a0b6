cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-wreset}
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py tests/test_gpu_metrics.py -m gpu -x -q > gpurun_out/r03_w_tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r03_w_tests.log
for args in "" "--total-envs 8192" "--total-envs 32768" "--workload E" "--workload D"; do echo "== $args"; AB_ARGS="--gen-sample 0 $args" bash profiles/tools/ab_bench.sh 2 ${BASE:-h4} $V; done
