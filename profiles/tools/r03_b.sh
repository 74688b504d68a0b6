cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > gpurun_out/r03_b_tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r03_b_tests.log
for n in 4096 65536; do FTL_DIAG_N=$n FTL_LIB=$PWD/variants_prof.so python profiles/tools/path_counts.py > gpurun_out/r03_b_paths_$n.log 2>&1; grep "steps 150-250 cycles" gpurun_out/r03_b_paths_$n.log; done
