# Round-3 profiling session (ONE GPU call):  bash tools/prof_r03.sh     -> gpurun_out/r03s/*, then tools/refresh_profiles_r03.sh r03
# Every rocprofv3 run has the program itself after "--" (python3 ...), kernel-trace only, counters in passes of their own.
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; rm -rf $O; mkdir -p $O
C5="--views 8 --people 8 --frames 25008 --seed 20260104 --steps 3 --warmup 1 --cpu-frames 0 --sustain 0"
DLT="--workload dlt --people 1 --frames 2000000 --seed 20260101 --cpu-frames 0 --sustain 0"
python3 $R/bench.py > $O/bench_c4.json 2> $O/bench_c4.err
echo "bench c4 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 $R/bench.py --cpu-frames 0 --sustain 0 > $O/bench_c4_stats.json 2> $O/stats_c4.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_c4 -- python3 $R/bench.py --cpu-frames 0 --sustain 0 --steps 2 --warmup 1 > /dev/null 2> $O/fetch_c4.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write_c4 -- python3 $R/bench.py --cpu-frames 0 --sustain 0 --steps 2 --warmup 1 > /dev/null 2> $O/write_c4.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/sq_c4 -- python3 $R/bench.py --cpu-frames 0 --sustain 0 --steps 2 --warmup 1 > /dev/null 2> $O/sq_c4.err
echo "config 4 done"
python3 $R/bench.py $C5 --steps 5 > $O/bench_c5.json 2> $O/bench_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $R/bench.py $C5 > /dev/null 2> $O/stats_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_c5 -- python3 $R/bench.py $C5 --steps 2 > /dev/null 2> $O/fetch_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write_c5 -- python3 $R/bench.py $C5 --steps 2 > /dev/null 2> $O/write_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/sq_c5 -- python3 $R/bench.py $C5 --steps 2 > /dev/null 2> $O/sq_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -d $O/mfma_c5 -- python3 $R/bench.py $C5 --steps 2 > /dev/null 2> $O/mfma_c5.err || true
echo "config 5 done"
python3 $R/bench.py $DLT > $O/bench_dlt.json 2> $O/bench_dlt.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dlt -- python3 $R/bench.py $DLT > /dev/null 2> $O/stats_dlt.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_dlt -- python3 $R/bench.py $DLT --steps 2 --warmup 1 > /dev/null 2> $O/fetch_dlt.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write_dlt -- python3 $R/bench.py $DLT --steps 2 --warmup 1 > /dev/null 2> $O/write_dlt.err
echo "dlt done"
python3 $R/bench.py --cpu-frames 0 --occlusion 0.05 --spurious 0.2 > $O/bench_occ.json 2> $O/bench_occ.err
ls $O
