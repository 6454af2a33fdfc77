# LDS bank conflicts of the chain kernels (one --pmc pass per configuration):  bash tools/prof_lds.sh
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lds; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $O/c4 -- python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1 > /dev/null 2> $O/c4.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $O/c5 -- python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1 --views 8 --people 8 --frames 8192 > /dev/null 2> $O/c5.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $O/st -- python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1 --path stages > /dev/null 2> $O/st.err
for p in c4 c5 st; do python3 $R/tools/pmc_kernel.py $O/$p _kernel | grep -v "rocprim\|at::" || true; done
