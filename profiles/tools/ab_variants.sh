cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 2 3 4; do
  echo "== variant $v"; FTL_LIB=$PWD/variants_wpe$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abx_$v -- python3 bench.py --steps 150 --warmup 10 --no-cpu-baseline > gpurun_out/abx_$v.log 2>&1
  grep -o "\"value\": [0-9.]*" gpurun_out/abx_$v.log; grep rays gpurun_out/abx_$v/*/*kernel_stats.csv | cut -c1-110
done
