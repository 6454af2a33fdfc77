# Round-4 profiling session (ONE GPU call):  bash tools/prof_r04.sh     -> gpurun_out/r04s/*, then tools/refresh_profiles_r04.sh r04
: "${GRAFT_REPO_ROOT:?}"
# Every rocprofv3 run has the program itself after "--" (python3 / a binary), kernel-trace only, counters in passes of their own.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04s; rm -rf $O; mkdir -p $O
N="--no-other-configs --cpu-frames 0 --sustain 0"
C5="--views 8 --people 8 --frames 25008 --seed 20260104 --steps 3 --warmup 1 $N"
DLT="--workload dlt --people 1 --frames 2000000 --tile-from 10000 --seed 20260101 $N"
# FETCH_SIZE calibration (tools/fetch_calib.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/fetch_calib.hip -o /tmp/fetch_calib 2> $O/calib_build.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/calib -- /tmp/fetch_calib > $O/calib.txt 2> $O/calib.err
echo "calibration done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 $R/bench.py $N > $O/bench_c4_stats.json 2> $O/stats_c4.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_c4 -- python3 $R/bench.py $N --steps 2 --warmup 1 > /dev/null 2> $O/fetch_c4.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write_c4 -- python3 $R/bench.py $N --steps 2 --warmup 1 > /dev/null 2> $O/write_c4.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/sq_c4 -- python3 $R/bench.py $N --steps 2 --warmup 1 > /dev/null 2> $O/sq_c4.err
echo "config 4 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $R/bench.py $C5 > /dev/null 2> $O/stats_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_c5 -- python3 $R/bench.py $C5 --steps 2 > /dev/null 2> $O/fetch_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write_c5 -- python3 $R/bench.py $C5 --steps 2 > /dev/null 2> $O/write_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/sq_c5 -- python3 $R/bench.py $C5 --steps 2 > /dev/null 2> $O/sq_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/mfma_c5 -- python3 $R/bench.py $C5 --steps 2 > /dev/null 2> $O/mfma_c5.err || true
echo "config 5 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dlt -- python3 $R/bench.py $DLT > /dev/null 2> $O/stats_dlt.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_dlt -- python3 $R/bench.py $DLT --steps 2 --warmup 1 > /dev/null 2> $O/fetch_dlt.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write_dlt -- python3 $R/bench.py $DLT --steps 2 --warmup 1 > /dev/null 2> $O/write_dlt.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS -d $O/sq_dlt -- python3 $R/bench.py $DLT --steps 2 --warmup 1 > /dev/null 2> $O/sq_dlt.err
echo "dlt done"
ls $O
