cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03_m_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r03_m_tests.log
[ $rc -eq 0 ] || exit 1
AB_ARGS="--gen-sample 0" bash profiles/tools/ab_bench.sh 2 cur rec
AB_ARGS="--gen-sample 0 --total-envs 8192" bash profiles/tools/ab_bench.sh 1 cur rec
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-steps 0 --gen-sample 0"
O=gpurun_out/r03_m_pmc; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc4 -- $B > $O/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc5 -- $B > $O/pmc5.log 2>&1
python3 profiles/summarize_pmc.py $O 65536 > gpurun_out/r03_m_pmc_summary.txt; cat gpurun_out/r03_m_pmc_summary.txt
