# LDS counters of the chain kernel:  bash tools/pmc_lds.sh [bench flags]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lds; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/p -- python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1 "$@" > /dev/null 2> $O/err.txt
python3 $R/tools/pmc_kernel.py $O/p chain_kernel
