# kernel stats of config 5 with occlusion (the repair tier's launches):  bash tools/prof_c5occ.sh   (inside one GPU call)
: "${GRAFT_REPO_ROOT:?}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c5occ; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --no-other-configs --cpu-frames 0 --sustain 0 --views 8 --people 8 --frames 4096 --occlusion 0.05 --spurious 0.2 --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench.json 2> $O/stats.err
python3 - <<PY
import json, csv, glob
r = json.load(open("$O/bench.json")); print("frames/s", round(r["value"]), "ms/step", r["ms_per_step"], r["tracker_events_per_step"])
f = glob.glob("$O/stats/*/*kernel_stats.csv")[0]
for row in list(csv.DictReader(open(f)))[:12]:
    print(row["Name"][:80], row["Calls"], round(float(row["AverageNs"])/1e6, 4), round(float(row["TotalDurationNs"])/1e6,1), row["Percentage"])
PY
