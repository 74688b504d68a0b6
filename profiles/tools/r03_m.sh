cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_fuzz.py > gpurun_out/r03_m_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 gpurun_out/r03_m_tests.log
[ $rc -eq 0 ] || exit 1
python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --gen-sample 0 | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d[\"value\"]/1e6,1), d[\"roofline\"][\"kernels_us\"])"
B="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-steps 0 --gen-sample 0"
O=gpurun_out/r03_m_pmc; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc4 -- $B > $O/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc5 -- $B > $O/pmc5.log 2>&1
python3 profiles/summarize_pmc.py $O 65536 > gpurun_out/r03_m_pmc_summary.txt; cat gpurun_out/r03_m_pmc_summary.txt
