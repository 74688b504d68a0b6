cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
FTL_LIB=$PWD/variants_def2.so timeout -k 10 120 python profiles/tools/dbg/dump_run.py gpurun_out/dump_def2.npz 256 150 && FTL_LIB=$PWD/variants_base.so timeout -k 10 120 python profiles/tools/dbg/dump_run.py gpurun_out/dump_base.npz 256 150 && python profiles/tools/dbg/dump_cmp.py gpurun_out/dump_base.npz gpurun_out/dump_def2.npz || exit 1
FTL_DIAG_N=65536 FTL_LIB=$PWD/variants_def2prof.so python profiles/tools/path_counts.py > gpurun_out/r03_i_paths.log 2>&1; grep "steps 150-250 cycles" gpurun_out/r03_i_paths.log
AB_ARGS="" bash profiles/tools/ab_bench.sh 2 base def def2
echo "--- FTL_DEFER=0"; FTL_DEFER=0 AB_ARGS="" bash profiles/tools/ab_bench.sh 1 def2
AB_ARGS="--total-envs 8192" bash profiles/tools/ab_bench.sh 1 base def def2
