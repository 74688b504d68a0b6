cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-def}
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/r03_e_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r03_e_tests.log
if [ $rc -eq 0 ]; then
AB_ARGS="--total-envs 8192" bash profiles/tools/ab_bench.sh 2 base $V
AB_ARGS="" bash profiles/tools/ab_bench.sh 2 base $V
AB_ARGS="--workload E" bash profiles/tools/ab_bench.sh 1 base $V
fi
