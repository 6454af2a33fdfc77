for x in 0 12000 30000 42000; do
  MVMC_CHAIN_EXTRA_LDS=$x timeout -k 10 300 python bench.py --cpu-frames 0 --no-other-configs --sustain 0 --host-io 0 2>/dev/null > gpurun_out/occ.json || exit 1
  python - $x <<'PY'
import json, sys
d = json.load(open("gpurun_out/occ.json"))
s = d["stages_ms"]; sh = s.get("chain_cycle_shares", {}); mc = s.get("chain_mcycles_mean_max", [0, 0])
x = int(sys.argv[1]); wg = int(163840 // ((40944 + x + 1279) // 1280 * 1280))
print("extra LDS %6d B -> %d workgroups per CU: %8.0f frames/s  %.2f ms  ALS %.2f Mcyc  IK %.2f Mcyc  chain mean %.2f" % (
    x, min(wg, 4), d["value"], d["ms_per_step"], sh.get("als", 0) * mc[0], sh.get("ik", 0) * mc[0], mc[0]), flush=True)
PY
done
