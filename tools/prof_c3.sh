# kernel stats of config 3 (association + triangulation, every frame cold):  bash tools/prof_c3.sh   (inside one GPU call)
: "${GRAFT_REPO_ROOT:?}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/c3; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --no-other-configs --cpu-frames 0 --sustain 0 --workload assoc_dlt --seed 20260102"
$B > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > /dev/null 2> $O/stats.err
python3 - <<PY
import json, csv, glob
r = json.load(open("$O/bench.json")); print("frames/s", round(r["value"]), "ms/step", r["ms_per_step"], r["stages_ms"], r["roofline"]["kernel"], r["roofline"]["frac"])
f = glob.glob("$O/stats/*/*kernel_stats.csv")[0]
for row in list(csv.DictReader(open(f)))[:10]:
    print(row["Name"][:70], row["Calls"], round(float(row["AverageNs"])/1e6, 4), row["Percentage"])
PY
