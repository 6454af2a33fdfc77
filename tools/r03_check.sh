#!/bin/bash
# One GPU call: the GPU test suite, then the default bench of this tree and of the round-2 tree (_r02/, same box), then config 5.
mkdir -p gpurun_out
python -m pytest tests -m gpu -q > gpurun_out/t_all.log 2>&1; tail -15 gpurun_out/t_all.log
python bench.py --cpu-frames 0 > gpurun_out/b_c4_new.json 2> gpurun_out/b_c4_new.err; tail -c 600 gpurun_out/b_c4_new.err
if [ -d _r02 ]; then (cd _r02 && python bench.py --cpu-frames 0 > ../gpurun_out/b_c4_r02.json 2> ../gpurun_out/b_c4_r02.err); fi
python bench.py --cpu-frames 0 --views 8 --people 8 --frames 25008 --seed 20260104 --steps 5 --warmup 1 > gpurun_out/b_c5_new.json 2> gpurun_out/b_c5_new.err; tail -c 600 gpurun_out/b_c5_new.err
python bench.py --cpu-frames 0 --occlusion 0.05 --spurious 0.2 --steps 10 --warmup 2 > gpurun_out/b_occ_new.json 2> gpurun_out/b_occ_new.err; tail -c 600 gpurun_out/b_occ_new.err
python - <<'PY'
import json
for n in ("b_c4_new", "b_c4_r02", "b_c5_new", "b_occ_new"):
    try:
        r = json.load(open(f"gpurun_out/{n}.json"))
        print(n, round(r["value"]), "frames/s", round(r["ms_per_step"], 2), "ms", "sustained", r.get("sustained"), r.get("tracker_events_per_step"))
    except Exception as e:
        print(n, "FAILED", e)
PY
