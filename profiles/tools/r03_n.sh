cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fuzz.py > gpurun_out/r03_n_tests.log 2>&1; rc=$?; echo "tests(both) rc=$rc"; tail -4 gpurun_out/r03_n_tests.log
FTL_LIB=$PWD/variants_bearadd.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q > gpurun_out/r03_n_tests2.log 2>&1; echo "tests(bearadd) rc=$?"; tail -2 gpurun_out/r03_n_tests2.log
AB_ARGS="--gen-sample 0" bash profiles/tools/ab_bench.sh 2 h0 bearadd both
AB_ARGS="--gen-sample 0 --total-envs 8192" bash profiles/tools/ab_bench.sh 1 h0 bearadd both
