cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-pin}
FTL_LIB=$PWD/variants_$V.so python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_c_tests_$V.log 2>&1; echo "tests($V) rc=$?"; tail -3 gpurun_out/r03_c_tests_$V.log
for n in 4096 65536; do for v in baseprof ${V}prof; do FTL_DIAG_N=$n FTL_LIB=$PWD/variants_$v.so python profiles/tools/path_counts.py > gpurun_out/r03_c_paths_${v}_$n.log 2>&1; echo "$v $n"; grep "steps 150-250 cycles" gpurun_out/r03_c_paths_${v}_$n.log; done; done
AB_ARGS="--total-envs 8192" bash profiles/tools/ab_bench.sh 2 base $V
AB_ARGS="" bash profiles/tools/ab_bench.sh 2 base $V
