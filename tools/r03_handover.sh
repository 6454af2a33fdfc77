#!/bin/bash
# the three hand-over protocols: tests, then same-box A/B on config 4 and config 5
python -m pytest tests -m gpu -q -x -k "chain_fused or parallel or capacity or synth_tracker" > gpurun_out/t3.log 2>&1; tail -3 gpurun_out/t3.log
for h in static ticket queue static ticket queue; do
  python bench.py --cpu-frames 0 --sustain 0 --hand-over $h 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c4 $h', round(r['value']), r['stages_ms'].get('chain_mcycles_mean_max'), r['stages_ms'].get('one_step_alone'))"
done
for h in static ticket static ticket; do
  python bench.py --cpu-frames 0 --sustain 0 --views 8 --people 8 --frames 25008 --seed 20260104 --steps 5 --warmup 1 --hand-over $h 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read()); print('c5 $h', round(r['value']))"
done
