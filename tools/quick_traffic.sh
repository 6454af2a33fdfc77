# bench + FETCH_SIZE / WRITE_SIZE of the chain kernel in one GPU call:  bash tools/quick_traffic.sh [bench flags]
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/qt; rm -rf $O; mkdir -p $O
python3 $R/bench.py --cpu-frames 0 "$@" > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fetch -- python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1 "$@" > /dev/null 2> $O/fetch.err
rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/write -- python3 $R/bench.py --cpu-frames 0 --steps 2 --warmup 1 "$@" > /dev/null 2> $O/write.err
python3 - <<PY
import json, sys
sys.path.insert(0, "$R/tools")
r = json.load(open("$O/bench.json"))
print("frames/s", r["value"], "ms/step", r["ms_per_step"], "one step alone", r["stages_ms"].get("one_step_alone"), "chain Mcycles", r["stages_ms"].get("chain_mcycles_mean_max"), r["stages_ms"].get("chain_cycle_shares"))
PY
python3 $R/tools/pmc_kernel.py $O/fetch chain_kernel
python3 $R/tools/pmc_kernel.py $O/write chain_kernel
