cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AB_ARGS="--gen-sample 0" bash profiles/tools/ab_bench.sh 1 h2 wpe3 wpe4
FTL_NO_REGROUP=1 AB_ARGS="--gen-sample 0" bash profiles/tools/ab_bench.sh 1 wpe4
AB_ARGS="--gen-sample 0 --total-envs 32768" bash profiles/tools/ab_bench.sh 1 h2 wpe4
