# one call: bench line + rocprofv3 kernel stats + PMC passes for the round's final state (run on the GPU box via gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/final_bench.log 2>&1 && tail -1 gpurun_out/final_bench.log > gpurun_out/final_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_stats -- python3 bench.py --no-cpu-baseline > gpurun_out/final_stats.log 2>&1
bash profiles/pmc_passes.sh > gpurun_out/pmc_passes.log 2>&1
python3 profiles/summarize_pmc.py gpurun_out 65536 > gpurun_out/final_pmc_summary.txt
cat gpurun_out/final_bench.json; grep -h ftl_ gpurun_out/final_stats/*/*kernel_stats.csv | cut -c1-120
