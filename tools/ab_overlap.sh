#!/bin/bash
# Same-box A/B of library variants over the number of steps in flight:  [BENCH_ARGS="..."] tools/ab_overlap.sh "2 3 4 5" name1 name2 ...  (name "hip" = the shipped library)
OV=$1; shift
for o in $OV; do
  for n in "$@"; do
    MVMC_LIB_PATH=$PWD/multiview_motion_capture_amd/lib/libmvmc_$n.so timeout -k 10 200 python bench.py --cpu-frames 0 --no-other-configs --sustain 0 --overlap $o $BENCH_ARGS 2>/dev/null > gpurun_out/ab_$n.json || exit 1
    python - "$n" $o <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
s = d["stages_ms"]; sh = s.get("chain_cycle_shares", {}); mc = s.get("chain_mcycles_mean_max", [0, 0])
print("overlap %s %-8s %8.0f frames/s  %.2f ms  ALS %.2f Mcyc  IK %.2f Mcyc  chain mean %.2f max %.2f" % (
    sys.argv[2], sys.argv[1], d["value"], d["ms_per_step"], sh.get("als", 0) * mc[0], sh.get("ik", 0) * mc[0], mc[0], mc[1]), flush=True)
PY
  done
done
