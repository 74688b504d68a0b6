cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for sp in 0 1 2; do echo "FTL_SPLIT=$sp"; FTL_SPLIT=$sp AB_ARGS="--gen-sample 0 --kernel-steps 0" bash profiles/tools/ab_bench.sh 2 stag; done
echo "E:"; for sp in 0 2; do FTL_SPLIT=$sp AB_ARGS="--gen-sample 0 --kernel-steps 0 --workload E" bash profiles/tools/ab_bench.sh 1 stag; done
echo "F:"; for sp in 1 2; do FTL_SPLIT=$sp AB_ARGS="--gen-sample 0 --kernel-steps 0 --workload F" bash profiles/tools/ab_bench.sh 1 stag; done
