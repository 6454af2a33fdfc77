# A round's profiling session (ONE GPU call):  bash tools/prof.sh <tag, e.g. r06> [calib|c4|c5|c3|dlt ...]  -> gpurun_out/<tag>s/*, then (here)
# tools/refresh_profiles.sh <tag> writes profiles/<tag>_*, and (one more GPU call) tools/records.sh <tag> the bench lines that cite them.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
# Every rocprofv3 run has the program itself after "--" (python3 / a binary), kernel-trace only, counters in passes of their own.
set -e
cd /tmp && export TMPDIR=/tmp
TAG=${1:?usage: prof.sh <tag> [calib|c4|c5|c3|dlt ...]}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${TAG}s; mkdir -p $O
WHAT="${*:-calib c4 c5 c3 dlt}"
N="--no-other-configs --cpu-frames 0 --sustain 0"
C5="--views 8 --people 8 --frames 25008 --seed 20260104 --steps 3 --warmup 1 $N"
C3="--workload assoc_dlt --seed 20260102 --steps 6 --warmup 2 $N"
DLT="--workload dlt --people 1 --frames 2000000 --tile-from 10000 --seed 20260101 $N"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
passes() {   # <tag> <bench args for the counter passes> : kernel stats of the full command, then FETCH / WRITE / SQ passes
  tag=$1; shift
  rm -rf $O/stats_$tag $O/fetch_$tag $O/write_$tag $O/sq_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$tag -- python3 $R/bench.py $STATS_ARGS > $O/bench_${tag}_stats.json 2> $O/stats_$tag.err
  rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_$tag -- python3 $R/bench.py "$@" > /dev/null 2> $O/fetch_$tag.err
  rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write_$tag -- python3 $R/bench.py "$@" > /dev/null 2> $O/write_$tag.err
  rocprofv3 --kernel-trace --output-format csv --pmc $SQ -d $O/sq_$tag -- python3 $R/bench.py "$@" > /dev/null 2> $O/sq_$tag.err
  echo "$tag done"
}
for w in $WHAT; do
  case $w in
    calib)   # FETCH_SIZE calibration (tools/fetch_calib.hip)
      rm -rf $O/calib
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o /tmp/fetch_calib 2> $O/calib_build.err
      rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/calib -- /tmp/fetch_calib > $O/calib.txt 2> $O/calib.err
      echo "calibration done" ;;
    c4) STATS_ARGS="$N" passes c4 $N --steps 2 --warmup 1
        MVMC_PROF_OUT=${TAG}s/insts_c4 MVMC_PROF_ARGS="--no-other-configs" bash $R/tools/prof_insts.sh > $O/insts_c4.txt 2>&1 ;;
    c5) STATS_ARGS="$C5" passes c5 $C5 --steps 2
        rm -rf $O/mfma_c5
        rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/mfma_c5 -- python3 $R/bench.py $C5 --steps 2 > /dev/null 2> $O/mfma_c5.err || true
        MVMC_PROF_OUT=${TAG}s/insts_c5 MVMC_PROF_ARGS="--no-other-configs --views 8 --people 8 --frames 25008 --seed 20260104" bash $R/tools/prof_insts.sh > $O/insts_c5.txt 2>&1 ;;
    c3) STATS_ARGS="$C3" passes c3 $C3
        MVMC_PROF_OUT=${TAG}s/insts_c3 MVMC_PROF_KERNEL=als4_kernel MVMC_PROF_ARGS="--no-other-configs --workload assoc_dlt --seed 20260102" bash $R/tools/prof_insts.sh > $O/insts_c3.txt 2>&1 ;;
    dlt) STATS_ARGS="$DLT" passes dlt $DLT --steps 2 --warmup 1 ;;
  esac
done
ls $O
