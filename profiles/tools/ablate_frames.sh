# time the frame kernel with parts compiled out (variants_*.so built with -DFTL_ABLATE_*); timing only, results differ
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in base FRAMES1 FRAMES5 SENSORS GREEN AGENT SEARCH EXACT; do
  echo "== variant $v"; FTL_LIB=$PWD/variants_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abl_$v -- python3 bench.py --steps 150 --warmup 100 --no-cpu-baseline > gpurun_out/abl_$v.log 2>&1
  grep -h "ftl_" gpurun_out/abl_$v/*/*kernel_stats.csv | cut -c1-120
done
