#!/bin/bash
# same-box A/B of library variants on config 2 (2 M frames):  bash tools/dlt_ab.sh hip nodma ...   (V1=1 in front selects the first kernel)
for v in "$@" "$@"; do
  MVMC_LIB_PATH=multiview_motion_capture_amd/lib/libmvmc_$v.so python bench.py --no-other-configs --cpu-frames 0 --sustain 0 --workload dlt --frames 2000000 --tile-from 10000 --people 1 --seed 20260101 $DLT_AB_EXTRA 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('dlt $v', round(r['value'] / 1e6, 1), 'M frames/s', round(r['ms_per_step'], 3), 'ms', round(r['roofline']['frac'], 4))"
done
