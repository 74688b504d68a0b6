cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r03_q_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03_q_tests.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 50 --warmup 5 --backend gloo --gen-sample 0 > gpurun_out/r03_q_gloo2.log 2>&1; echo "gloo2 rc=$?"; tail -1 gpurun_out/r03_q_gloo2.log | cut -c1-900
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_q_driver_like.log 2>&1; echo "driver-like rc=$?"; tail -1 gpurun_out/r03_q_driver_like.log | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()"
