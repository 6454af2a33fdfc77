#!/bin/bash
# config 5 after a change of als5: the oracle gates, the phase profile of an iteration, then the bench line (same box)
set -o pipefail
python -m pytest tests/test_gpu_config5_c8p8.py -m gpu -q -x > gpurun_out/t_c5.log 2>&1; rc=$?; tail -5 gpurun_out/t_c5.log
[ $rc -ne 0 ] && exit $rc
make -C multiview_motion_capture_amd/csrc prof-als > /dev/null 2>&1
MVMC_LIB_PATH=multiview_motion_capture_amd/lib/libmvmc_prof.so python tools/als5_profile.py > gpurun_out/als5_profile.txt 2>&1; tail -25 gpurun_out/als5_profile.txt
for i in 1 2; do
  python bench.py --cpu-frames 0 --sustain 0 --views 8 --people 8 --frames 25008 --seed 20260104 --steps 5 --warmup 1 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c5', round(r['value']), r['stages_ms'].get('chain_mcycles_mean_max'), r['tracker_events_per_step'])"
done
