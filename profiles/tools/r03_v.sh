cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FTL_FUZZ_SEEDS=96:400 timeout -k 10 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q > gpurun_out/r03_v_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r03_v_fuzz.log; cat gpurun_out/waivers.json; echo
timeout -k 10 600 python profiles/tools/soak.py > gpurun_out/r03_v_soak.log 2>&1; echo "soak rc=$?"; grep -v "step " gpurun_out/r03_v_soak.log | tail -4
