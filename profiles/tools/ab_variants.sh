# A/B of library builds: FTL_LIB=variants_<name>.so, kernel times by rocprofv3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "$@"; do
  echo "== variant $v"; FTL_LIB=$PWD/variants_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abx_$v -- python3 bench.py --steps 300 --warmup 100 --no-cpu-baseline > gpurun_out/abx_$v.log 2>&1
  grep -o "\"value\": [0-9.]*" gpurun_out/abx_$v.log; grep -h ftl_ gpurun_out/abx_$v/*/*kernel_stats.csv | cut -c1-110
done
