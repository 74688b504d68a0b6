cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03_u_tests.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r03_u_tests.log; cat gpurun_out/waivers.json
