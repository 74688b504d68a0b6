cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ev in 2 3 4 6 8; do echo "FTL_REGROUP_EVERY=$ev"; FTL_REGROUP_EVERY=$ev AB_ARGS="--gen-sample 0" bash profiles/tools/ab_bench.sh 2 fin; done
