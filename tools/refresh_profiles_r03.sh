#!/bin/bash
# After `bash tools/prof_r03.sh && MVMC_PROF_OUT=r03s_insts bash tools/prof_insts.sh` on the GPU box (outputs merged into gpurun_out/):
#   tools/refresh_profiles_r03.sh <tag, e.g. r03>
# writes profiles/<tag>_* (kernel stats, PMC traffic, SQ counters, instruction mix, the bench lines) and profiles/pmc_traffic.json.
set -e
TAG=$1
cd "$(dirname "$0")/.."
O=gpurun_out/r03s
python3 tools/aggregate_profiles.py $O/stats_c4 $O/fetch_c4 $O/write_c4 ${TAG}_fused_10k_C5P4 > /dev/null
python3 tools/aggregate_profiles.py $O/stats_c5 $O/fetch_c5 $O/write_c5 ${TAG}_fused_25k_C8P8 25008x8x8 > /dev/null
python3 tools/aggregate_profiles.py $O/stats_dlt $O/fetch_dlt $O/write_dlt ${TAG}_dlt_2M_C5P1 2000000x5x1 > /dev/null
python3 tools/pmc_kernel.py $O/sq_c4 chain_kernel > profiles/${TAG}_fused_10k_C5P4_sq_counters.txt
python3 tools/pmc_kernel.py $O/sq_c5 chain_kernel > profiles/${TAG}_fused_25k_C8P8_sq_counters.txt
python3 tools/pmc_kernel.py $O/mfma_c5 chain_kernel > profiles/${TAG}_fused_25k_C8P8_mfma_counters.txt 2>/dev/null || true
[ -d gpurun_out/r03s_insts ] && python3 tools/aggregate_insts.py gpurun_out/r03s_insts $O/sq_c4 ${TAG}_fused_10k_C5P4 chain:10000x5x4 > /dev/null
cp $O/bench_c5.json profiles/${TAG}_bench_fused_25k_C8P8.json
cp $O/bench_c4.json profiles/${TAG}_bench_fused_10k_C5P4.json
cp $O/bench_dlt.json profiles/${TAG}_bench_dlt_2M_C5P1.json
cp $O/bench_occ.json profiles/${TAG}_bench_fused_10k_C5P4_occluded.json
python3 - "$O" <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_c4", "bench_c4_stats", "bench_c5", "bench_occ", "bench_dlt"):
    r = json.load(open(f"{O}/{f}.json")); s = r["stages_ms"]
    print(f, "steps", r["steps"], "frames/s %.0f ms/step %.3f" % (r["value"], r["ms_per_step"]), "sustained", (r.get("sustained") or {}).get("value"),
          "launch_ms %.3f" % r["roofline"]["launch_ms"], "frac %.2e" % r["roofline"]["frac"], "traffic", r["roofline"]["traffic"],
          [round(x, 1) for x in s.get("chain_mcycles_mean_max", [])], {k: round(v, 3) for k, v in s.get("chain_cycle_shares", {}).items()}, r.get("tracker_events_per_step"))
    if r.get("cpu_baseline"):
        print("   cpu", r["cpu_baseline"]["value"], r["cpu_baseline"].get("one_core_value"), (r["cpu_baseline"].get("numpy_port") or {}).get("value"))
PY
for f in profiles/${TAG}_*_kernel_stats.csv; do echo $f; head -3 $f | tail -2; done
grep -E "chain_kernel|dlt" profiles/${TAG}_*_pmc_traffic.csv
