# one call: bench lines (B, D, E, F and the f3 workloads C, L, T), rocprofv3 kernel stats of the B bench command, PMC passes -- the round's final evidence.
# usage (on the GPU box via gpurun):  bash profiles/collect_round.sh TAG      -> gpurun_out/TAG_*
TAG=${1:-r02_z}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/${TAG}_bench.log 2>&1 && tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
for w in D E F C L T; do python3 bench.py --workload $w --steps 200 > gpurun_out/${TAG}_bench_$w.log 2>&1 && tail -1 gpurun_out/${TAG}_bench_$w.log > gpurun_out/${TAG}_bench_$w.json; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --no-cpu-baseline --kernel-steps 0 > gpurun_out/${TAG}_stats.log 2>&1
cp gpurun_out/${TAG}_stats/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
python3 profiles/tools/stats_timed.py gpurun_out/${TAG}_stats/*/*kernel_trace.csv 400 > gpurun_out/${TAG}_kernel_stats_timed.csv   # the 400 timed steps only
for w in D E; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_$w -- python3 bench.py --workload $w --steps 200 --no-cpu-baseline --kernel-steps 0 > gpurun_out/${TAG}_stats_$w.log 2>&1
  cp gpurun_out/${TAG}_stats_$w/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_$w.csv
done
TAG=$TAG bash profiles/pmc_passes.sh > gpurun_out/${TAG}_pmc_passes.log 2>&1
python3 profiles/summarize_pmc.py gpurun_out/${TAG}_pmc 65536 > gpurun_out/${TAG}_pmc_summary.txt
cat gpurun_out/${TAG}_bench.json | cut -c1-300; grep -h ftl_ gpurun_out/${TAG}_kernel_stats.csv | cut -c1-120
