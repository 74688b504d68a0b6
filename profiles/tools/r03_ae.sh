# refresh of the D / E evidence after the float32 minima (bench lines + kernel traces)
TAG=r03_z
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in D E; do python3 bench.py --workload $w --steps 200 > gpurun_out/${TAG}_bench_$w.log 2>&1 && tail -1 gpurun_out/${TAG}_bench_$w.log > gpurun_out/${TAG}_bench_$w.json; done
for w in D E; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_stats_$w -- python3 bench.py --workload $w --steps 200 --no-cpu-baseline --kernel-steps 0 --gen-sample 0 > gpurun_out/${TAG}_stats_$w.log 2>&1
  cp gpurun_out/${TAG}_stats_$w/*/*kernel_stats.csv gpurun_out/${TAG}_kernel_stats_$w.csv
  python3 profiles/tools/stats_timed.py gpurun_out/${TAG}_stats_$w/*/*kernel_trace.csv 200 2 > gpurun_out/${TAG}_kernel_stats_timed_$w.csv
done
python3 -c "
import json
for w in 'DE':
    d=json.load(open('gpurun_out/r03_z_bench_%s.json'%w)); print(w, d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['config']['envs_with_error_flags'])"
grep -h "frames_group\|rays_kernel" gpurun_out/r03_z_kernel_stats_timed_D.csv gpurun_out/r03_z_kernel_stats_timed_E.csv | cut -c1-100
