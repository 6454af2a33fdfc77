# VALU / SALU / LDS instructions per KERNEL of the launch-per-stage path (the same device functions as the chain kernel's phases):
# where the chain kernel's instructions come from.   bash tools/prof_stage_insts.sh   (inside one GPU call)
: "${GRAFT_REPO_ROOT:?run on the GPU box (gpurun exports it), or set it to the repo root}"
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/stage_insts; rm -rf $O; mkdir -p $O
B="python3 $R/bench.py --cpu-frames 0 --sustain 0 --steps 1 --warmup 1 --path stages --overlap 1 $MVMC_PROF_ARGS"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES -d $O/a -- $B > /dev/null 2> $O/a.err
python3 - $O/a <<'PY'
import collections, csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if n.startswith("at::"): continue
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVES": cnt[n] += 1
tot = sum(v["SQ_INSTS_VALU"] for v in acc.values())
for n, v in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_INSTS_VALU"]):
    print("%-60s launches %4d VALU %8.1f M (%4.1f %%) SALU %7.1f M LDS %7.1f M" % (n[:60], cnt[n], v["SQ_INSTS_VALU"] / 1e6, 100 * v["SQ_INSTS_VALU"] / tot, v["SQ_INSTS_SALU"] / 1e6, v["SQ_INSTS_LDS"] / 1e6))
print("total VALU %.1f M over all launches of the process (2 steps)" % (tot / 1e6))
PY
