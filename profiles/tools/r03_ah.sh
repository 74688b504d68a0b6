# refresh of the small-batch evidence after the 8-lane form (bench lines + kernel traces): B at 4,096 / 8,192 envs, D
TAG=r03_z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in 4096 8192; do python3 bench.py --total-envs $n --steps 200 --no-cpu-baseline > gpurun_out/${TAG}_bench_B$n.log 2>&1 && tail -1 gpurun_out/${TAG}_bench_B$n.log > gpurun_out/${TAG}_bench_B$((n/1024))k.json; done
python3 bench.py --workload D --steps 200 > gpurun_out/${TAG}_bench_D.log 2>&1 && tail -1 gpurun_out/${TAG}_bench_D.log > gpurun_out/${TAG}_bench_D.json
for n in 4096 8192; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_B$n -- python3 bench.py --total-envs $n --steps 200 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 > gpurun_out/${TAG}_stats_B$n.log 2>&1
  python3 profiles/tools/stats_timed.py gpurun_out/${TAG}_stats_B$n/*/*kernel_trace.csv 200 2 > gpurun_out/${TAG}_kernel_stats_timed_B$((n/1024))k.csv
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_D -- python3 bench.py --workload D --steps 200 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 > gpurun_out/${TAG}_stats_D.log 2>&1
cp gpurun_out/${TAG}_stats_D/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_D.csv
python3 profiles/tools/stats_timed.py gpurun_out/${TAG}_stats_D/*/*kernel_trace.csv 200 2 > gpurun_out/${TAG}_kernel_stats_timed_D.csv
python3 -c "
import json
for w in ('B4k','B8k','D'):
    d=json.load(open('gpurun_out/r03_z_bench_%s.json'%w)); print(w, d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['config']['envs_with_error_flags'])"
grep -h "frames_group\|rays_kernel" gpurun_out/r03_z_kernel_stats_timed_B4k.csv gpurun_out/r03_z_kernel_stats_timed_B8k.csv gpurun_out/r03_z_kernel_stats_timed_D.csv | cut -c1-100
