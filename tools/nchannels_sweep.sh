#!/bin/bash
# RCCL channel cap beside the persistent chain kernel: config 4 at world size 1 with the step's all-gather FORCED through RCCL
# (bench.py --force-collective), NCCL_MAX_NCHANNELS swept; first line = no collective at all.  One line per run -> $1.
set -u
R="$(cd "$(dirname "$0")/.." && pwd)"
O="${1:-$R/gpurun_out/r05_nchannels_sweep.txt}"
: > "$O"
run() {
  python "$R/bench.py" --cpu-frames 0 --no-other-configs --sustain 100 "$@" 2>/dev/null | python -c '
import json, sys
p = json.loads(sys.stdin.read().strip().splitlines()[-1])
c = p["collective"]
print("%-28s value %.1f k  sustained %.1f k  gather p50 %.3f ms max %.3f ms  backend %s  env %s" % (sys.argv[1], p["value"] / 1e3,
      p["sustained"]["value"] / 1e3, c["gather_ms"]["p50"], c["gather_ms"]["max"], c["backend"], c["env"]["NCCL_MAX_NCHANNELS"]))' "$LABEL" >> "$O" || echo "$LABEL FAILED" >> "$O"
}
LABEL="no collective (world 1)" run || exit 1
for n in 1 2 4 8 16 32; do
  LABEL="forced, NCCL_MAX_NCHANNELS=$n" NCCL_MAX_NCHANNELS=$n run --force-collective || exit 1
done
cat "$O"
