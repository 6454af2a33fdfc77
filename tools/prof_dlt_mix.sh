# instruction mix of config 2's kernel:  bash tools/prof_dlt_mix.sh <out name>   (inside one GPU call)
: "${GRAFT_REPO_ROOT:?}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --no-other-configs --cpu-frames 0 --sustain 0 --steps 3 --warmup 1 --workload dlt --frames 2000000 --tile-from 10000 --people 1 --seed 20260101"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 -d $O/b -- $B > /dev/null 2> $O/b.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 -d $O/c -- $B > /dev/null 2> $O/c.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $O/d -- $B > /dev/null 2> $O/d.err || true
for p in b c d; do python3 $R/tools/pmc_kernel.py $O/$p ingest_dlt || true; done
