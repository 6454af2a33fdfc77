#!/bin/bash
# same-box A/B of library variants on config 5 (25,008 frames): bash tools/lib_variants_c5.sh v1 v2 ... (lib/libmvmc_<v>.so; "hip" = the shipped one)
for v in "$@" "$@"; do
  MVMC_LIB_PATH=multiview_motion_capture_amd/lib/libmvmc_$v.so python bench.py --cpu-frames 0 --sustain 0 --views 8 --people 8 --frames 25008 --seed 20260104 --steps 5 --warmup 1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c5 $v', round(r['value']), r['stages_ms'].get('chain_mcycles_mean_max'))"
done
