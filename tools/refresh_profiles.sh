#!/bin/bash
# After `bash tools/prof.sh <tag>` on the GPU box (outputs merged into gpurun_out/<tag>s):   tools/refresh_profiles.sh <tag, e.g. r06>
# writes profiles/<tag>_* (kernel stats, PMC traffic, SQ counters, instruction mix) and profiles/pmc_traffic.json.
set -e
TAG=$1
cd "$(dirname "$0")/.."
O=gpurun_out/${TAG}s
python3 tools/aggregate_profiles.py $O/stats_c4 $O/fetch_c4 $O/write_c4 ${TAG}_fused_10k_C5P4 > /dev/null
python3 tools/aggregate_profiles.py $O/stats_c5 $O/fetch_c5 $O/write_c5 ${TAG}_fused_25k_C8P8 25008x8x8 > /dev/null
python3 tools/aggregate_profiles.py $O/stats_c3 $O/fetch_c3 $O/write_c3 ${TAG}_assoc_dlt_10k_C5P4 assoc_dlt:10000x5x4 > /dev/null
python3 tools/aggregate_profiles.py $O/stats_dlt $O/fetch_dlt $O/write_dlt ${TAG}_dlt_2M_C5P1 2000000x5x1
python3 tools/pmc_kernel.py $O/sq_c4 chain_kernel > profiles/${TAG}_fused_10k_C5P4_sq_counters.txt
python3 tools/pmc_kernel.py $O/sq_c5 chain_kernel > profiles/${TAG}_fused_25k_C8P8_sq_counters.txt
python3 tools/pmc_kernel.py $O/sq_c3 als4_kernel > profiles/${TAG}_assoc_dlt_10k_C5P4_sq_counters.txt
python3 tools/pmc_kernel.py $O/mfma_c5 chain_kernel > profiles/${TAG}_fused_25k_C8P8_mfma_counters.txt 2>/dev/null || true
python3 tools/pmc_kernel.py $O/sq_dlt ingest_dlt > profiles/${TAG}_dlt_2M_C5P1_sq_counters.txt
python3 tools/pmc_kernel.py $O/calib calib_ > profiles/${TAG}_fetch_size_calibration.txt; cat $O/calib.txt >> profiles/${TAG}_fetch_size_calibration.txt
python3 tools/aggregate_insts.py $O/insts_c4 $O/sq_c4 ${TAG}_fused_10k_C5P4 chain:10000x5x4 > /dev/null
python3 tools/aggregate_insts.py $O/insts_c5 $O/sq_c5 ${TAG}_fused_25k_C8P8 chain:25008x8x8 chain_kernel $O/mfma_c5 > /dev/null
python3 tools/aggregate_insts.py $O/insts_c3 $O/sq_c3 ${TAG}_assoc_dlt_10k_C5P4 assoc:10000x5x4 als4_kernel > /dev/null
for f in profiles/${TAG}_*_kernel_stats.csv; do echo $f; head -3 $f | tail -2; done
grep -E "chain_kernel|dlt|als4" profiles/${TAG}_*_pmc_traffic.csv
cat profiles/${TAG}_fetch_size_calibration.txt
