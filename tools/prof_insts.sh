# instruction mix of chain_kernel (separate --pmc passes; kernel-trace only):  bash tools/prof_insts.sh   (inside one GPU call)
: "${GRAFT_REPO_ROOT:?}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${MVMC_PROF_OUT:-insts}; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --cpu-frames 0 --sustain 0 --steps 2 --warmup 1 $MVMC_PROF_ARGS"; K=${MVMC_PROF_KERNEL:-chain_kernel}
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/a -- $B > /dev/null 2> $O/a.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/b -- $B > /dev/null 2> $O/b.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 -d $O/c -- $B > /dev/null 2> $O/c.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU -d $O/d -- $B > /dev/null 2> $O/d.err || true
for p in a b c d; do python3 $R/tools/pmc_kernel.py $O/$p $K || true; done
