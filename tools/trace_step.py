"""Per-launch kernel durations of the LAST bench step from a rocprofv3 --kernel-trace csv:
  python tools/trace_step.py <dir with *_kernel_trace.csv> [launches_per_step=16]"""
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
L = int(sys.argv[2]) if len(sys.argv) > 2 else 16


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


names = [short(r["Kernel_Name"]) for r in rows]
ik = [i for i, n in enumerate(names) if n == "ik_kernel"]
last = ik[-L:]
start = max(i for i, n in enumerate(names[:last[0]]) if n.startswith("ingest_kernel"))
t0 = int(rows[start]["Start_Timestamp"])
tot = {}
for r, n in zip(rows[start:], names[start:]):
    if n.startswith("at::") or n.startswith("__amd"):
        n = "torch/copy"
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot.setdefault(n, []).append(d)
print("step wall %.3f ms" % ((int(rows[-1]["End_Timestamp"]) - t0) / 1e6))
for n, v in sorted(tot.items(), key=lambda kv: -sum(kv[1])):
    print("%-34s n=%4d sum %8.3f ms  per-launch: %s" % (n, len(v), sum(v), " ".join("%.2f" % x for x in v[:20])))
