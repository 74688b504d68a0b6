# one call: bench lines (B at the BASELINE size and at the strong-scaling shard sizes, D, E, F and the f3 workloads C, L, T), rocprofv3 kernel stats of the
# bench commands, PMC passes -- the round's final evidence.
# usage (on the GPU box via gpurun):  bash profiles/collect_round.sh TAG      -> gpurun_out/TAG_*
TAG=${1:-r03_z}
PART=${2:-ab}            # a: bench lines, b: rocprofv3 runs (two gpurun calls when one would exceed the time limit)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [[ $PART == *a* ]]; then
python3 bench.py > gpurun_out/${TAG}_bench.log 2>&1 && tail -1 gpurun_out/${TAG}_bench.log > gpurun_out/${TAG}_bench.json
python3 bench.py --parts 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_B_1part.log 2>&1 && tail -1 gpurun_out/${TAG}_bench_B_1part.log > gpurun_out/${TAG}_bench_B_1part.json   # one batch on one stream
for n in 4096 8192 16384 32768; do python3 bench.py --total-envs $n --steps 200 --no-cpu-baseline > gpurun_out/${TAG}_bench_B$n.log 2>&1 && tail -1 gpurun_out/${TAG}_bench_B$n.log > gpurun_out/${TAG}_bench_B$((n/1024))k.json; done
for w in D E F C L T; do python3 bench.py --workload $w --steps 200 > gpurun_out/${TAG}_bench_$w.log 2>&1 && tail -1 gpurun_out/${TAG}_bench_$w.log > gpurun_out/${TAG}_bench_$w.json; done
python3 bench.py --steps 6000 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 --ring 16384 > gpurun_out/${TAG}_bench_ring.log 2>&1 && tail -1 gpurun_out/${TAG}_bench_ring.log > gpurun_out/${TAG}_bench_ring.json
fi
if [[ $PART == *b* ]]; then
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats -- python3 bench.py --no-cpu-baseline --kernel-steps 0 --gen-sample 0 > gpurun_out/${TAG}_stats.log 2>&1
cp gpurun_out/${TAG}_stats/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
python3 profiles/tools/stats_timed.py gpurun_out/${TAG}_stats/*/*kernel_trace.csv 400 2 > gpurun_out/${TAG}_kernel_stats_timed.csv   # the 400 timed steps only (2 parts: two frame + two ray launches per step)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_1part -- python3 bench.py --parts 1 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 > gpurun_out/${TAG}_stats_1part.log 2>&1
python3 profiles/tools/stats_timed.py gpurun_out/${TAG}_stats_1part/*/*kernel_trace.csv 400 1 > gpurun_out/${TAG}_kernel_stats_timed_B_1part.csv
for n in 4096 8192 16384 32768; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_B$n -- python3 bench.py --total-envs $n --steps 200 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 > gpurun_out/${TAG}_stats_B$n.log 2>&1
  python3 profiles/tools/stats_timed.py gpurun_out/${TAG}_stats_B$n/*/*kernel_trace.csv 200 2 > gpurun_out/${TAG}_kernel_stats_timed_B$((n/1024))k.csv
done
for w in D E F; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_$w -- python3 bench.py --workload $w --steps 200 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 > gpurun_out/${TAG}_stats_$w.log 2>&1
  cp gpurun_out/${TAG}_stats_$w/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_$w.csv
  python3 profiles/tools/stats_timed.py gpurun_out/${TAG}_stats_$w/*/*kernel_trace.csv 200 2 > gpurun_out/${TAG}_kernel_stats_timed_$w.csv
done
TAG=$TAG bash profiles/pmc_passes.sh > gpurun_out/${TAG}_pmc_passes.log 2>&1
python3 profiles/summarize_pmc.py gpurun_out/${TAG}_pmc 65536 --json gpurun_out/${TAG}_pmc_current.json B "profiles/${TAG}_pmc_summary.txt (rocprofv3 --pmc passes of bench.py --steps 20 --warmup 5, per-launch averages over the timed steps)" > gpurun_out/${TAG}_pmc_summary.txt
fi
cat gpurun_out/${TAG}_bench.json | cut -c1-300; grep -h ftl_ gpurun_out/${TAG}_kernel_stats_timed.csv | cut -c1-120
