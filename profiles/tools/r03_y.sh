cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for q in 0 7 5 3 2; do echo "FTL_QUIET_FRAMES=$q"; FTL_QUIET_FRAMES=$q AB_ARGS="--gen-sample 0" bash profiles/tools/ab_bench.sh 1 qf; FTL_QUIET_FRAMES=$q AB_ARGS="--gen-sample 0 --total-envs 8192" bash profiles/tools/ab_bench.sh 1 qf; done
