# PMC passes of the B bench command in its steady state: one counter set per pass, never combined with any trace domain but --kernel-trace
TAG=${TAG:-r03_z}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --parts 1 --steps 20 --warmup 5 --no-cpu-baseline --kernel-steps 0 --gen-sample 0"     # one batch per launch: the counters are per kernel, and every launch covers all 65,536 envs
O=gpurun_out/${TAG}_pmc; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 --output-format csv -d $O/pmc3 -- $B > $O/pmc3.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc4 -- $B > $O/pmc4.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc5 -- $B > $O/pmc5.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc6 -- $B > $O/pmc6.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $O/pmc7 -- $B > $O/pmc7.log 2>&1     # lane utilisation
ls $O/pmc*/*/ | head -30
