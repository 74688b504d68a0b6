cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=${1:-pin}; N=${2:-256}; T=${3:-150}
FTL_LIB=$PWD/variants_base.so timeout -k 10 120 python profiles/tools/dbg/dump_run.py gpurun_out/dump_base.npz $N $T && \
FTL_LIB=$PWD/variants_$V.so timeout -k 10 120 python profiles/tools/dbg/dump_run.py gpurun_out/dump_$V.npz $N $T && \
python profiles/tools/dbg/dump_cmp.py gpurun_out/dump_base.npz gpurun_out/dump_$V.npz
