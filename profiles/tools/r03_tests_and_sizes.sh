cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r03_a_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03_a_tests.log
for n in 4096 8192 16384 32768 65536; do
  python bench.py --total-envs $n --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r03_a_bench_B$n.log 2>&1; tail -1 gpurun_out/r03_a_bench_B$n.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$n', '%.1f M' % (d['value']/1e6), d['ms_per_step'], d['roofline']['kernels_us'])"
done
