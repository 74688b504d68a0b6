cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 2 3 4; do
  echo "== wpe $v"; FTL_LIB=$PWD/variants_wpe$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -- python3 bench.py --steps 150 --warmup 10 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
  grep -o "\"value\": [0-9.]*" gpurun_out/ab_$v.log; head -3 gpurun_out/ab_$v/*/*kernel_stats.csv | tail -2 | cut -c1-110
done
