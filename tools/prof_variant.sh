# Counters of chain_kernel for ONE library variant (same-box comparison of builds):  bash tools/prof_variant.sh <name>   (inside one GPU call)
#   -> gpurun_out/pv_<name>/{a,sq,mem}: instruction counts, SQ busy / wait cycles, FETCH_SIZE / WRITE_SIZE.  Kernel-trace only, counters
#   in passes of their own, python3 directly after "--".
set -e
: "${GRAFT_REPO_ROOT:?}"
N=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pv_$N; rm -rf $O; mkdir -p $O
export MVMC_LIB_PATH=$R/multiview_motion_capture_amd/lib/libmvmc_$N.so
B="python3 $R/bench.py --cpu-frames 0 --sustain 0 --no-other-configs --steps 2 --warmup 1 $MVMC_PROF_ARGS"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $O/a -- $B > /dev/null 2> $O/a.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/sq -- $B > /dev/null 2> $O/sq.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/f -- $B > /dev/null 2> $O/f.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/w -- $B > /dev/null 2> $O/w.err
for p in a sq f w; do python3 $R/tools/pmc_kernel.py $O/$p chain_kernel || true; done
