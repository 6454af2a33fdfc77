# SQ counters of config 2's kernel (separate --pmc passes; kernel-trace only):  [MVMC_INGEST_DLT_V1=1] bash tools/prof_dlt.sh <out name>   (inside one GPU call)
: "${GRAFT_REPO_ROOT:?}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --no-other-configs --cpu-frames 0 --sustain 0 --steps 3 --warmup 1 --workload dlt --frames 2000000 --tile-from 10000 --people 1 --seed 20260101"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d $O/a -- $B > /dev/null 2> $O/a.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS -d $O/b -- $B > /dev/null 2> $O/b.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_IFETCH_LEVEL -d $O/c -- $B > /dev/null 2> $O/c.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- $B > /dev/null 2> $O/t.err
for p in a b c; do python3 $R/tools/pmc_kernel.py $O/$p ingest_dlt || true; done
grep -h "ingest_dlt" $O/t/*/*kernel_stats.csv || true
