: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02s; rm -rf $O; mkdir -p $O
python3 $R/bench.py > $O/bench_c4.json 2> $O/bench_c4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c4 -- python3 $R/bench.py --cpu-frames 0 > $O/bench_c4_stats.json 2> $O/stats_c4.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_c4 -- python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1 > /dev/null 2> $O/fetch_c4.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write_c4 -- python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1 > /dev/null 2> $O/write_c4.err
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $O/sq_c4 -- python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1 > /dev/null 2> $O/sq_c4.err
echo "config 4 done"
python3 $R/bench.py --views 8 --people 8 --frames 8192 --steps 3 --warmup 1 --cpu-frames 0 > $O/bench_c5.json 2> $O/bench_c5.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c5 -- python3 $R/bench.py --views 8 --people 8 --frames 8192 --steps 3 --warmup 1 --cpu-frames 0 > /dev/null 2> $O/stats_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch_c5 -- python3 $R/bench.py --views 8 --people 8 --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 > /dev/null 2> $O/fetch_c5.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write_c5 -- python3 $R/bench.py --views 8 --people 8 --frames 8192 --steps 2 --warmup 1 --cpu-frames 0 > /dev/null 2> $O/write_c5.err
python3 $R/bench.py --views 8 --people 8 --frames 25008 --steps 3 --warmup 1 --cpu-frames 0 > $O/bench_c5_full.json 2> $O/bench_c5_full.err
echo "config 5 done"
ls $O
