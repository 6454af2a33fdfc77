#!/bin/bash
# same-box A/B of library variants on config 4 (the default bench line): bash tools/lib_variants_c4.sh base hip ... (lib/libmvmc_<v>.so)
for v in "$@" "$@"; do
  MVMC_LIB_PATH=multiview_motion_capture_amd/lib/libmvmc_$v.so python bench.py --cpu-frames 0 --sustain 0 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c4 $v', round(r['value']), r['stages_ms'].get('chain_mcycles_mean_max'), {k: round(v, 4) for k, v in (r['stages_ms'].get('chain_cycle_shares') or {}).items()})"
done
