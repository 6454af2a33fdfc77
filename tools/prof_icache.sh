# instruction / scalar cache counters of the chain kernel (one --pmc pass each):  bash tools/prof_icache.sh
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/icache; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1"
rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH -d $O/a -- $B > /dev/null 2> $O/a.err
rocprofv3 --kernel-trace --output-format csv --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_WAVES -d $O/b -- $B > /dev/null 2> $O/b.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN SQ_INSTS_SENDMSG SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_EXP_GDS -d $O/c -- $B > /dev/null 2> $O/c.err || true
for p in a b c; do python3 $R/tools/pmc_kernel.py $O/$p chain_kernel || true; done
