cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AB_ARGS="" bash profiles/tools/ab_bench.sh 2 def defk0 defk1
echo "--- no regroup"
FTL_NO_REGROUP=1 AB_ARGS="" bash profiles/tools/ab_bench.sh 1 base def
