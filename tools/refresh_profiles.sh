#!/bin/bash
# After `bash tools/prof_r02.sh && MVMC_PROF_OUT=r02s_insts bash tools/prof_insts.sh` on the GPU box (outputs merged into gpurun_out/):
#   tools/refresh_profiles.sh <new tag, e.g. r02m> [<old tag to remove>]
# writes profiles/<tag>_* (kernel stats, PMC traffic, SQ counters, instruction mix, the bench lines) and profiles/pmc_traffic.json.
# Clear gpurun_out/r02s and gpurun_out/r02s_insts BEFORE the GPU call: gpurun merges new files into them and never deletes old ones.
set -e
TAG=$1; OLD=$2
cd "$(dirname "$0")/.."
O=gpurun_out/r02s
[ -n "$OLD" ] && { git rm -q --cached profiles/${OLD}_* 2>/dev/null || true; rm -f profiles/${OLD}_*; }
python3 tools/aggregate_profiles.py $O/stats_c4 $O/fetch_c4 $O/write_c4 ${TAG}_fused_10k_C5P4 > /dev/null
python3 tools/aggregate_profiles.py $O/stats_c5 $O/fetch_c5 $O/write_c5 ${TAG}_fused_8k_C8P8 8192x8x8 > /dev/null
python3 tools/pmc_kernel.py $O/sq_c4 chain_kernel > profiles/${TAG}_fused_10k_C5P4_sq_counters.txt
python3 tools/aggregate_insts.py gpurun_out/r02s_insts $O/sq_c4 ${TAG}_fused_10k_C5P4 chain:10000x5x4 > /dev/null
cp $O/bench_c5.json profiles/${TAG}_bench_fused_8k_C8P8.json
cp $O/bench_c5_full.json profiles/${TAG}_bench_fused_25k_C8P8.json
cp $O/bench_c4.json profiles/${TAG}_bench_fused_10k_C5P4.json
python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_c4", "bench_c4_stats", "bench_c5", "bench_c5_full"):
    r = json.load(open(f"{O}/{f}.json")); s = r["stages_ms"]
    print(f, "steps", r["steps"], "frames/s %.0f ms/step %.2f" % (r["value"], r["ms_per_step"]), "launch_ms %.2f" % r["roofline"]["launch_ms"],
          "alone %.2f" % s["one_step_alone"], "all launches %.2f" % s["chain_kernel_all_launches"], [round(x, 1) for x in s["chain_mcycles_mean_max"]],
          {k: round(v, 3) for k, v in s["chain_cycle_shares"].items()})
    if r.get("cpu_baseline"):
        print("   cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["one_core_value"], r["cpu_baseline"]["numpy_port"]["value"])
PY
head -2 profiles/${TAG}_fused_10k_C5P4_kernel_stats.csv | tail -1
head -2 profiles/${TAG}_fused_8k_C8P8_kernel_stats.csv | tail -1
grep chain_kernel profiles/${TAG}_fused_10k_C5P4_pmc_traffic.csv profiles/${TAG}_fused_8k_C8P8_pmc_traffic.csv
cat profiles/${TAG}_fused_10k_C5P4_sq_counters.txt
grep -E "flop_per_launch|valu_busy|lds_busy|valu_insts|wave_cycles" profiles/${TAG}_fused_10k_C5P4_inst_mix.txt
