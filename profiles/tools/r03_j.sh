cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_j_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r03_j_tests.log
AB_ARGS="" bash profiles/tools/ab_bench.sh 2 base def2 def3
AB_ARGS="--total-envs 8192" bash profiles/tools/ab_bench.sh 1 base def2 def3
AB_ARGS="--workload F" bash profiles/tools/ab_bench.sh 1 base def3
AB_ARGS="--workload D" bash profiles/tools/ab_bench.sh 1 base def3
