#!/bin/bash
# The bench lines kept under profiles/ (run AFTER tools/refresh_profiles.sh <tag> has written profiles/pmc_traffic.json for the current
# kernel sources, so that the lines carry their traffic / fp64 records):  bash tools/records.sh <tag>   (one GPU call) -> gpurun_out/rec/;
# copy gpurun_out/rec/<tag>_bench_*.json into profiles/ afterwards.
T=${1:?usage: records.sh <tag>}
O=gpurun_out/rec; rm -rf $O; mkdir -p $O
python bench.py > $O/${T}_bench_fused_10k_C5P4.json 2> $O/c4.err
python bench.py --no-other-configs --cpu-frames 0 --views 8 --people 8 --frames 25008 --seed 20260104 --steps 5 --warmup 1 > $O/${T}_bench_fused_25k_C8P8.json 2> $O/c5.err
python bench.py --no-other-configs --cpu-frames 0 --workload dlt --people 1 --frames 2000000 --tile-from 10000 --seed 20260101 --steps 200 --warmup 20 > $O/${T}_bench_dlt_2M_C5P1.json 2> $O/dlt.err
python bench.py --no-other-configs --workload dlt --people 1 --seed 20260101 > $O/${T}_bench_dlt_10k_C5P1.json 2> $O/dlt10k.err
python bench.py --no-other-configs --cpu-frames 0 --occlusion 0.05 --spurious 0.2 > $O/${T}_bench_fused_10k_C5P4_occluded.json 2> $O/occ.err
python bench.py --no-other-configs --cpu-frames 0 --workload assoc_dlt --seed 20260102 > $O/${T}_bench_assoc_dlt_10k_C5P4.json 2> $O/c3.err
python bench.py --cpu-frames 0 --gpus 2 --backend gloo --share-gpu --steps 3 --warmup 1 --sustain 0 2> $O/two.err | grep -v "^\[Gloo\]" > $O/${T}_bench_two_ranks_one_gpu_gloo.json
python bench.py --no-other-configs --cpu-frames 0 --gpus 2 --backend gloo --share-gpu --views 8 --people 8 --frames 2048 --occlusion 0.05 --spurious 0.2 --steps 3 --warmup 1 --sustain 0 2> $O/two_occ.err | grep -v "^\[Gloo\]" > $O/${T}_bench_two_ranks_one_gpu_gloo_repairs.json
python bench.py --no-other-configs --cpu-frames 0 --views 8 --people 8 --frames 8192 --seed 20260104 --occlusion 0.05 --spurious 0.2 --steps 5 --warmup 1 > $O/${T}_bench_fused_8k_C8P8_occluded.json 2> $O/c5occ.err
python bench.py --workload shelf --steps 3 --warmup 1 > $O/${T}_bench_shelf_update_4d.json 2> $O/shelf.err
python bench.py --no-other-configs --cpu-frames 0 --force-collective > $O/${T}_bench_fused_10k_C5P4_forced_rccl_world1.json 2> $O/force.err
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/rec/*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(r["value"]), "ms %.3f" % r["ms_per_step"], "frac %.4g" % r["roofline"]["frac"], "traffic", r["roofline"]["traffic"], "sustained", (r.get("sustained") or {}).get("value"),
              [(o.get("name", "")[:9], round(o.get("value", 0)), round(o["roofline"]["frac"], 4), o["roofline"]["traffic"]) for o in r.get("other_configs", []) if "roofline" in o])
    except Exception as e:
        print(f, "FAILED", e)
PY
