#!/bin/bash
# Same-box A/B of library variants: [BENCH_ARGS="--views 8 --people 8 --frames 25008 --seed 20260104 --steps 3 --warmup 1"] tools/lib_ab.sh name1 name2 ...
# (multiview_motion_capture_amd/lib/libmvmc_<name>.so, built by tools/full_variant.sh or tools/build_variant.sh), two rounds each on the default
# bench line (config 4) or on BENCH_ARGS.  A library named this way is not checked against lib/BUILD_INFO.json (_cabi.load).
for round in 1 2; do
  for n in "$@"; do
    MVMC_LIB_PATH=$PWD/multiview_motion_capture_amd/lib/libmvmc_$n.so timeout -k 10 200 python bench.py --cpu-frames 0 --no-other-configs --sustain 0 $BENCH_ARGS 2>/dev/null > gpurun_out/ab_$n.json || exit 1
    python - "$n" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_%s.json" % sys.argv[1]))
s = d["stages_ms"]; sh = s.get("chain_cycle_shares", {}); mc = s.get("chain_mcycles_mean_max", [0, 0])
print("%-8s %8.0f frames/s  %.2f ms  ALS %.2f Mcyc  IK %.2f Mcyc  chain mean %.2f max %.2f" % (
    sys.argv[1], d["value"], d["ms_per_step"], sh.get("als", 0) * mc[0], sh.get("ik", 0) * mc[0], mc[0], mc[1]), flush=True)
PY
  done
done
