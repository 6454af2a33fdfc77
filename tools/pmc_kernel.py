"""Per-kernel means of the counters in a rocprofv3 --pmc output directory:  python tools/pmc_kernel.py <dir> [kernel substring]"""
import collections, csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)   # newest run
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if sub in n and not n.startswith("at::"):
        acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, cs in acc.items():
    print(n, {k: round(sum(v) / len(v)) for k, v in sorted(cs.items())}, "launches", len(next(iter(cs.values()))))
